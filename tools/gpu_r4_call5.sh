#!/bin/bash
# round 4, GPU call 5: sliced gunzip decoder (global tables) -> does the hybrid finally beat the host alone; decoder PMC profile; four-context test
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 300 python tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456 > $O/gunzip_419_l6_g256.log 2>&1; echo "gunzip419 g256 rc=$?"; tail -2 $O/gunzip_419_l6_g256.log
AQC_PIPE_DEBUG=1 timeout 900 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 2 --warmup 1 --device-steps 3 > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4c5/bench.log").read().strip().splitlines()[-1])
    for k in ("value", "roofline", "file_to_file", "file_to_file_gz", "file_to_gz"):
        print(k, json.dumps(d.get(k))[:800])
except Exception as e:
    print("bench parse failed", e)
PY
grep -E "gunzip|CPU seconds" $O/bench.err | tail -8
for G in 33554432 134217728; do
  AQC_GZ_GROUP=$G timeout 600 python bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 2 > $O/bench_g$G.log 2> $O/bench_g$G.err
  python - $G <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/r4c5/bench_g%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
    print("group", sys.argv[1], json.dumps(d.get("file_to_file_gz"))[:500])
except Exception as e:
    print("bench parse failed", e)
PY
done
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_gunzip -o g -- python $GRAFT_REPO_ROOT/tools/gpu_gunzip_dev.py 64 6 default 16 1048576 268435456 > $GRAFT_REPO_ROOT/$O/pmc_gunzip.log 2>&1)
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r4c5/pmc_gunzip/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        if "gzb" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    print(k, {c: round(x) for c, x in v.items()}, "dispatches", max(cnt[(k, c)] for c in v))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_gunzip -o g -- python $GRAFT_REPO_ROOT/tools/gpu_gunzip_dev.py 64 6 default 16 1048576 268435456 > /dev/null 2>&1); python tools/pmc_summary.py $O/kt_gunzip 2>/dev/null | grep gzb | cut -c1-160
