#!/bin/bash
# round 4, GPU call 6: hybrid gunzip with enough slices for GNU gzip's 32 K-token blocks; decoder PMC profile; config-5 / config-2 profiles
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c6; mkdir -p $O
AQC_PIPE_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 2 --warmup 1 --device-steps 3 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c6/bench.log").read().strip().splitlines()[-1])
    for k in ("value", "roofline", "file_to_file_gz", "file_to_gz"):
        print(k, json.dumps(d.get(k))[:800])
except Exception as e:
    print("bench parse failed", e)
PY
grep -E "gunzip|CPU seconds" $O/bench.err | tail -6
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_gunzip -o g -- python $GRAFT_REPO_ROOT/tools/gpu_gunzip_dev.py 64 6 default 16 1048576 268435456 > $GRAFT_REPO_ROOT/$O/pmc_gunzip.log 2>&1)
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r4c6/pmc_gunzip/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        if "gzb" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    print(k, {c: round(x) for c, x in v.items()}, "dispatches", max(cnt[(k, c)] for c in v))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_gunzip -o g -- python $GRAFT_REPO_ROOT/tools/gpu_gunzip_dev.py 64 6 default 16 1048576 268435456 > /dev/null 2>&1); python tools/pmc_summary.py $O/kt_gunzip 2>/dev/null | grep gzb | cut -c1-160
timeout 600 bash tools/gpu_profile.sh config5 gpurun_out/r4c6/profile_config5.txt 14
timeout 600 bash tools/gpu_profile.sh config2 gpurun_out/r4c6/profile_config2.txt 12
