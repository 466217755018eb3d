#!/bin/bash
# kernel-trace of the bench step for the in-tree library and every build/ablate/*.so (A/B builds)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for f in afterqc_amd/csrc/libafterqc_hip.so build/ablate/*.so; do
  tag=$(basename $f .so)
  (cd /tmp && AQC_LIB=$GRAFT_REPO_ROOT/$f rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kta_$tag -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --pipe-runs 0 --file-runs 0 > /dev/null 2>&1)
  echo "== $tag"; python tools/pmc_summary.py gpurun_out/kta_$tag | grep -E "${1:-fmt_|text_index|frame_rec|fast_filter}" | cut -c1-150
done
