#!/bin/bash
# single-member gzip -6 inputs (made once), CLI cold: default, then section / in-flight variants on the same files
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --keep --reuse --dir /tmp/e2e_gz6 "$@" > gpurun_out/e2e_gz6_$tag.log 2>&1; echo "$tag rc=$?"; tail -1 gpurun_out/e2e_gz6_$tag.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('gen_s','wall_s','pass1_s','pass2_s','pass2_mreads_s','pass2_cores_busy','pipe_threads')})"; }
run default
run default2
AQC_GZ_SECTION=1048576 run sec1m
AQC_GZ_SECTION=4194304 run sec4m
run default3
