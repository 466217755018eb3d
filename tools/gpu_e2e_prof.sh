#!/bin/bash
# kernel-trace summary of one file-to-file run (text path)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for mb in 4 8 16 32; do python tools/e2e_bench.py --pairs 5000000 --mode text --chunk-mb $mb 2>&1 | tail -1 | cut -c1-330; done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_e2e -o kt -- python $GRAFT_REPO_ROOT/tools/e2e_bench.py --pairs 5000000 --mode text --chunk-mb 16 > /dev/null 2>&1)
python tools/pmc_summary.py gpurun_out/prof_e2e | grep -v rocclr
