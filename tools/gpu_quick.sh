#!/bin/bash
# quick GPU iteration: parity tests of the hot path + one bench line
#   gpurun -- 'bash tools/gpu_quick.sh [pytest-args]'
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value %.1f Mreads/s  step %.3f ms  kernel %.4f ms  frac %.4f  qc_stat %.3f ms" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["qc_stat_kernel_ms"]))
except Exception as e:
    print("bench parse failed", e)
PY
tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
