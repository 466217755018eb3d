#!/bin/bash
# the driver-comparable bench line + a config-5 kernel-trace with the statRead kernels on the slot's own stream (AQC_QC_STREAM=0)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${1:-r5b}; mkdir -p $D; export TMPDIR=/tmp
python bench.py > $D/bench.json 2> $D/bench.err; tail -c 300 $D/bench.err
python - <<PY
import json
d = json.loads(open("$D/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "device_step", "device_step_spans", "roofline", "pinned_to_pinned", "file_to_file", "file_to_file_100M", "multi_input_file_to_file", "file_to_file_gz", "file_to_gz", "cpu_baseline"):
    print(k, d.get(k))
PY
AQC_QC_STREAM=0 EXTRA=--text-step-only bash tools/gpu_profile.sh config5 $D/profile_config5_qc_inline.txt 14
