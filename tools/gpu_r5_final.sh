#!/bin/bash
# round 5, closing run: smoke, the whole -m gpu suite, the driver-comparable bench line, the three device steps under rocprofv3
# (text, spans, and the text step with AQC_FUSED=1), each with its own PMC passes
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${1:-r5final}; mkdir -p $D; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $D/smoke.log
timeout 1200 python -m pytest tests -m gpu -q > $D/all.log 2>&1; echo rc=$? >> $D/all.log; tail -3 $D/all.log
python bench.py > $D/bench.json 2> $D/bench.err; tail -c 200 $D/bench.err
python - <<PY
import json
d = json.loads(open("$D/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "device_step", "device_step_spans", "device_step_fused", "roofline", "pinned_to_pinned", "file_to_file", "file_to_file_100M", "multi_input_file_to_file", "file_to_file_gz", "file_to_gz"):
    print(k, d.get(k))
PY
EXTRA=--spans-step-only bash tools/gpu_profile.sh config3 $D/profile_spans_config3.txt 14
EXTRA=--text-step-only bash tools/gpu_profile.sh config3 $D/profile_text_config3.txt 14
AQC_FUSED=1 EXTRA=--text-step-only bash tools/gpu_profile.sh config3 $D/profile_fused_config3.txt 14
