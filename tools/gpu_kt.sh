#!/bin/bash
# kernel-trace summary of the bench's HBM-resident step (no pipe / file runs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --pipe-runs 0 --file-runs 0 $@ > /dev/null 2>&1)
python tools/pmc_summary.py gpurun_out/prof_kt | grep -v rocclr
