#!/bin/bash
# per-build PMC profile of the hot kernel: instruction mix per wave batch and wait/active split
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for f in afterqc_amd/csrc/libafterqc_hip.so build/ablate/*.so; do
  tag=$(basename $f .so)
  (cd /tmp && AQC_LIB=$GRAFT_REPO_ROOT/$f rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pa_$tag -o x -- python $GRAFT_REPO_ROOT/bench.py --device-only --device-steps 3 --cpu-sample 0 --no-pmc $AQC_BENCH_ARGS > /dev/null 2>&1)
  python - "$tag" <<'PY'
import csv, sys, glob, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pa_%s/*counter_collection.csv" % tag):
    for row in csv.DictReader(open(f)):
        if "fast_filter" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
# per PAIR numbers (5 M pairs per dispatch)
print(tag, {k: round(v[1] / v[0] / 5e6, 2) for k, v in sorted(agg.items())})
PY
done
