#!/bin/bash
# round-3 checkpoint: every -m gpu test, the driver's bench line, the one-input multi-context variant
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err | grep -v amdgpu.ids
timeout 600 python bench.py --steps 3 --warmup 1 --contexts 4 --cpu-sample 0 --pipe-runs 2 --device-steps 3 > gpurun_out/bench_ctx4.log 2> gpurun_out/bench_ctx4.err; echo "bench ctx4 rc=$?"
python - <<'PY'
import json
for f in ("bench", "bench_ctx4"):
    try:
        d = json.loads(open("gpurun_out/%s.log" % f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "ms", d["ms_per_step"], "device_step", d["device_step_mreads_s"], "p2p", d["pinned_to_pinned_mreads_s"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
        print("   f2f", json.dumps(d["file_to_file"]))
        print("   cpu", json.dumps(d.get("cpu_baseline", {}))[:300], d["host"])
    except Exception as e:
        print(f, "parse failed", e)
PY
