#!/bin/bash
# A/B of the pipe's plain-text output path on ONE box: AQC_SPANS=1 (good records written from the input buffers) vs 0 (formatted + fetched)
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${1:-r5ab}; mkdir -p $D; export TMPDIR=/tmp
B="python bench.py --cpu-sample 0 --no-pmc --device-steps 1 --gz-runs 0 --inputs ${INPUTS:-0} --steps 6 --warmup 2 --pipe-runs 3"
for rep in 1 2; do
  for sp in 1 0; do
    AQC_SPANS=$sp $B > $D/ab_spans${sp}_$rep.json 2> $D/ab_spans${sp}_$rep.err
    python - <<PY
import json
d = json.loads(open("$D/ab_spans${sp}_$rep.json").read().strip().splitlines()[-1])
f = d["file_to_file"]; t = f["thread_seconds_last_run"]
print("spans=$sp rep=$rep value", d["value"], "pinned", d["pinned_to_pinned_mreads_s"], "f2f s", f["seconds_mean"], f["seconds_min"], {k: round(v, 3) for k, v in t.items()}, "multi", (d.get("multi_input_file_to_file") or {}).get("mreads_s"))
PY
  done
done
