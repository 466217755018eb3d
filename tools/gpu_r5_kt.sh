#!/bin/bash
# round 5: kernel-trace summaries (no counters) of the device step: text step, spans step, fused text step
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
B="python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --device-only --device-steps 10 --workload config3 --pairs 5000000 --no-pmc --no-fused-step"
for v in text spans ${KT_FUSED:+fused}; do
  rm -rf $OUT/kt_$v
  case $v in
    text) (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$v -o kt -- $B --text-step-only > /dev/null 2>&1) ;;
    spans) (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$v -o kt -- $B --spans-step-only > /dev/null 2>&1) ;;
    fused) (cd /tmp && AQC_FUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$v -o kt -- $B --text-step-only > /dev/null 2>&1) ;;
  esac
  echo "== $v"
  python tools/pmc_summary.py $OUT/kt_$v | grep -v -E "rocclr|kmer_compact" | cut -c1-150 | tee $OUT/kt_$v.txt | head -14
done
