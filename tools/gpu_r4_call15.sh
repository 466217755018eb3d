#!/bin/bash
# round 4, GPU call 15: PMC counters of the device gunzip's decode / expand / scan kernels (per dispatch)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c15; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/tools/gpu_gunzip_dev.py 419 6 default 16 1048576 268435456"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O/p1 -o x -- $CMD > /dev/null 2>&1); echo "p1 rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/$O/p2 -o x -- $CMD > /dev/null 2>&1); echo "p2 rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH --output-format csv -d $GRAFT_REPO_ROOT/$O/p3 -o x -- $CMD > /dev/null 2>&1); echo "p3 rc=$?"
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    per = collections.defaultdict(dict)
    for f in glob.glob("gpurun_out/r4c15/%s/*counter_collection.csv" % p):
        for row in csv.DictReader(open(f)):
            if "gzb" in row["Kernel_Name"]:
                per[(int(row["Dispatch_Id"]), row["Kernel_Name"][5:22])][row["Counter_Name"]] = float(row["Counter_Value"])
    keys = sorted(per)[:24]
    for k in keys:
        print(p, k, {a: int(b) for a, b in sorted(per[k].items())})
PY
