#!/bin/bash
# quick GPU iteration: bench lines for the three workloads (+ optional pytest of the parity file)
#   gpurun -- 'bash tools/gpu_quick.sh [test]'
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "$1" = "test" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -4 gpurun_out/pytest_gpu.log
fi
for w in config3 config2 config5; do
  P=5000000; [ $w = config5 ] && P=3000000
  timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --workload $w --pairs $P > gpurun_out/bench_$w.log 2> gpurun_out/bench_$w.err; echo "bench $w rc=$?"
  python - $w <<'PY'
import json, sys
w = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/bench_%s.log" % w).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s value %.1f Mreads/s  step %.3f ms  kernel %.4f ms  frac %.4f  qc_stat %.3f ms good %.4f" % (w, d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["qc_stat_kernel_ms"], d["good_reads_frac"]))
except Exception as e:
    print("bench parse failed", e)
PY
  tail -2 gpurun_out/bench_$w.err | grep -v amdgpu.ids
done
