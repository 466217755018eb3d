#!/bin/bash
# round 4, GPU call 24: kernel trace of a bench run with .gz output and .gz -> .gz (what the GPU does beside the filter)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c24; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --no-pmc --inputs 0 --pipe-runs 0 --steps 1 --warmup 1 --device-steps 0 --gz-runs 2"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o g -- $B > $GRAFT_REPO_ROOT/$O/bench.log 2> $GRAFT_REPO_ROOT/$O/bench.err); echo "rc=$?"
python tools/pmc_summary.py $O/kt 2>/dev/null | cut -c1-170 | head -40
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -o p -- $B > /dev/null 2>&1); echo "pmc rc=$?"
python tools/pmc_summary.py $O/pmc 2>/dev/null | grep -E "gz_encode|gz_hist|gz_pack" | cut -c1-170
