#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for f in afterqc_amd/csrc/libafterqc_hip.so build/ablate/*.so; do
  AQC_LIB=$PWD/$f python bench.py --device-only --device-steps 10 --cpu-sample 0 --no-pmc "$@" > gpurun_out/b.log 2>gpurun_out/b.err
  grep PROF gpurun_out/b.err
  python - "$f" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/b.log").read().strip().splitlines()[-1]); r = d["roofline"]
print("%-40s kernel %.4f ms  frac %.4f  qc_stat %.4f ms" % (sys.argv[1][-40:], r["kernel_ms"], r["frac"], r.get("qc_stat_kernel_ms", 0.0)))
PY
done
