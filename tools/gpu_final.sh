#!/bin/bash
# closing run of a round: smoke, the whole -m gpu suite (the soak too), the driver-comparable bench line, the device steps under rocprofv3
# (text = the default, spans, the text step with AQC_FUSED=1, config 5, config 2), each with its own PMC passes, the CLI table.
#   usage: tools/gpu_final.sh [tag] [parts]      parts: any of  smoke tests bench steps cli   (default: all)
cd "$GRAFT_REPO_ROOT"; TAG=${1:-final}; PARTS=${2:-smoke tests bench steps cli}; D=gpurun_out/$TAG; mkdir -p $D; export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
rm -rf /tmp/pytest-of-* /tmp/aqc_* 2>/dev/null      # (a box may be one of this round's earlier ones: pytest keeps the last three runs' files, 39 GB each)
if has smoke; then python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $D/smoke.log; fi
if has tests; then timeout 2400 python -m pytest tests -m gpu -q > $D/all.log 2>&1; echo rc=$? >> $D/all.log; grep -v "^{\|options:$" $D/all.log | tail -4 | cut -c1-300; rm -rf /tmp/pytest-of-* 2>/dev/null; fi
if has bench; then
  python bench.py > $D/bench.json 2> $D/bench.err; tail -c 200 $D/bench.err
  python - <<PY
import json
d = json.loads(open("$D/bench.json").read().strip().splitlines()[-1])
for k in ("value", "value_median", "value_best", "ms_per_step", "device_step", "device_step_spans", "device_step_fused", "roofline", "cpu_baseline", "pinned_to_pinned", "file_to_file", "file_to_file_100M", "multi_input_file_to_file", "file_to_file_gz", "file_to_gz", "host"):
    print(k, d.get(k))
PY
fi
if has steps; then
  EXTRA=--text-step-only bash tools/gpu_profile.sh config3 $D/profile_text_config3.txt 16
  EXTRA=--spans-step-only bash tools/gpu_profile.sh config3 $D/profile_spans_config3.txt 12
  AQC_FUSED=1 EXTRA=--text-step-only bash tools/gpu_profile.sh config3 $D/profile_fused_config3.txt 12
  AQC_PLACE_COPY=0 EXTRA=--text-step-only bash tools/gpu_profile.sh config3 $D/profile_text_config3_round5_writer.txt 12
  EXTRA=--text-step-only bash tools/gpu_profile.sh config5 $D/profile_text_config5.txt 14
  EXTRA=--text-step-only bash tools/gpu_profile.sh config2 $D/profile_text_config2.txt 14
  python tools/kt_step.py gpurun_out/prof_kt/kt_kernel_trace.csv 3 > $D/timeline_config2.txt 2>&1
fi
if has cli; then bash tools/gpu_e2e_table.sh; cp gpurun_out/e2e_cli.txt $D/e2e_cli.txt; fi
