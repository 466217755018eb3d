#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipe.py -m gpu -x -q > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu2.log
for w in config5 config2; do
  P=5000000; [ $w = config5 ] && P=3000000
  timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pipe-runs 1 --workload $w --pairs $P > gpurun_out/bench_$w.log 2> gpurun_out/bench_$w.err; echo "bench $w rc=$?"
done
timeout 600 python bench.py --steps 3 --warmup 1 --contexts 4 --cpu-sample 0 --pipe-runs 2 --device-steps 3 > gpurun_out/bench_ctx4.log 2> gpurun_out/bench_ctx4.err; echo "bench ctx4 rc=$?"
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --pipe-runs 2 --device-steps 3 --gz-runs 2 > gpurun_out/bench_gz.log 2> gpurun_out/bench_gz.err; echo "bench gz rc=$?"
python - <<'PY'
import json
for f in ("bench_config5", "bench_config2", "bench_ctx4", "bench_gz"):
    try:
        d = json.loads(open("gpurun_out/%s.log" % f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "ms", d["ms_per_step"], "device_step", d["device_step_mreads_s"], d["device_step"]["ms_per_step"], "p2p", d["pinned_to_pinned_mreads_s"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
        print("   f2f", json.dumps(d["file_to_file"])[:400])
        if d.get("file_to_file_gz"): print("   gz", json.dumps(d["file_to_file_gz"]))
    except Exception as e:
        print(f, "parse failed", e)
PY
tail -3 gpurun_out/bench_gz.err
