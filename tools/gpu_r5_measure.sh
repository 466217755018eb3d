#!/bin/bash
# round 5: the driver-comparable bench line, then rocprofv3 summaries (kernel trace + HBM counters in their own passes) of the device
# step in its two variants — the text step (aqc_format) and the step the pipe runs for plain-text outputs (aqc_format_spans)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${1:-r5m}; mkdir -p $D; export TMPDIR=/tmp
python bench.py > $D/bench.json 2> $D/bench.err; tail -c 600 $D/bench.err
EXTRA=--spans-step-only bash tools/gpu_profile.sh config3 $D/profile_spans_config3.txt 60 > /dev/null
EXTRA=--text-step-only bash tools/gpu_profile.sh config3 $D/profile_text_config3.txt 60 > /dev/null
python - <<PY
import json
d = json.loads(open("$D/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "device_step_mreads_s", "pinned_to_pinned_mreads_s")})
print(d.get("device_step")); print(d.get("device_step_spans")); print(d.get("roofline")); print(d.get("pinned_to_pinned")); print(d.get("file_to_file"))
PY
