#!/bin/bash
# round 4, GPU call 22: where the config-5 flavour's time goes through the CLI (thread seconds per stage)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c22; mkdir -p $O
AQC_PIPE_DEBUG=1 timeout 600 python tools/e2e_bench.py --pairs 2000000 --config5 > $O/cfg5_plain.log 2> $O/cfg5_plain.err; echo "cfg5 plain rc=$?"; tail -1 $O/cfg5_plain.log | cut -c1-1500
grep -E "^pipe:" $O/cfg5_plain.err | tail -30 | cut -c1-300
AQC_PIPE_DEBUG=1 timeout 600 python tools/e2e_bench.py --pairs 5000000 > $O/cfg3_plain.log 2> $O/cfg3_plain.err; echo "cfg3 plain rc=$?"; tail -1 $O/cfg3_plain.log | cut -c1-1500
grep -E "^pipe:" $O/cfg3_plain.err | tail -12 | cut -c1-300
