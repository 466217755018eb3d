#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for w in config3 config2 config5; do
  P=5000000; [ $w = config5 ] && P=3000000
  timeout 600 python bench.py --device-only --device-steps 20 --cpu-sample 0 --workload $w --pairs $P > gpurun_out/dev_$w.log 2> gpurun_out/dev_$w.err; echo "dev $w rc=$?"
done
python - <<'PY'
import json
for w in ("config3", "config2", "config5"):
    try:
        d = json.loads(open("gpurun_out/dev_%s.log" % w).read().strip().splitlines()[-1])
        print(w, "device_step", d["device_step_mreads_s"], "ms", d["device_step"]["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "qc_ms", d["roofline"]["qc_stat_ms_per_call"])
    except Exception as e:
        print(w, "parse failed", e)
PY
