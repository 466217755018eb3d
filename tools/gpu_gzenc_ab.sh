#!/bin/bash
# .gz OUTPUT on the device: the wave-per-member encoder (round 6) against the thread-per-segment one (AQC_GZ_ENCODER=seg), interleaved
# on one box: plain -> .gz and .gz -> .gz through the pipe, then the kernels' own times (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "AQC_GZ_ENCODER=wave" "AQC_GZ_ENCODER=seg" "AQC_GZ_ENCODER=wave" "AQC_GZ_ENCODER=seg"; do
  env $v timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pipe-runs 0 --device-steps 2 --gz-runs 4 --no-pmc --no-fused-step --inputs 1 --big-copies 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['file_to_file_gz']
print('$v', 'file_to_gz', d['file_to_gz']['mreads_s'], 'gz->gz', g['mreads_s'], 'median_s', g['median_seconds'], 'host_only', g['host_only_mreads_s'], 'share', g['gunzip_text_share_from_device'], 'out_gb', g['output_gz_gb'])" | tee -a gpurun_out/gzenc_ab.log
done
OUT=$GRAFT_REPO_ROOT/gpurun_out; rm -rf $OUT/prof_gz
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pipe-runs 0 --device-steps 1 --gz-runs 3 --no-pmc --no-fused-step --inputs 1 --big-copies 0"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_gz -o kt -- $B > /dev/null 2>&1)
python tools/pmc_summary.py $OUT/prof_gz | sort -t= -k3 -n -r | head -24 | cut -c1-170 | tee gpurun_out/gz_prof_summary2.txt
rm -f $OUT/prof_gz/kt_kernel_trace.csv
