#!/bin/bash
# where do the dominant kernel's cycles go: active-VALU / SALU / LDS cycles and waits (separate PMC pass)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0"
(cd /tmp && rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/prof_act -o a -- $B > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH --output-format csv -d $OUT/prof_act2 -o a -- $B > /dev/null 2>&1)
python tools/pmc_summary.py $OUT/prof_act $OUT/prof_act2 | grep -E "fast_filter" | cut -c1-150
