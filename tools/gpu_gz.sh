#!/bin/bash
# gzip INPUT on the device (round 6: resident results, buffers by need): tests, the decoder alone, the pipe, the cold CLI.
#   tools/gpu_gz.sh [tests] [dev] [bench] [cold] [ab] [hbm] [prof]      (no argument: all but prof)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
WHAT="${*:-tests dev bench cold ab hbm}"
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1200 python -m pytest tests/test_gpu_pipe.py -m gpu -x -q -k "gunzip or gzip or gz" > gpurun_out/gz_pytest.log 2>&1; echo "pytest gz rc=$?"; tail -5 gpurun_out/gz_pytest.log
fi
if has dev; then
  # the decoder alone: one group of ~130 MB compressed (level 6), then two groups at level 1; device-only schedule
  AQC_GZ_DEBUG=1 timeout 600 python tools/gpu_gunzip_dev.py 420 6 default 16 1048576 268435456 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gz_dev.log
  timeout 600 python tools/gpu_gunzip_dev.py 1250 1 default 16 1048576 268435456 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gz_dev.log
fi
if has bench; then
  AQC_PIPE_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pipe-runs 0 --device-steps 2 --gz-runs 4 --no-pmc --no-fused-step --inputs 1 --big-copies 0 > gpurun_out/gz_bench.log 2> gpurun_out/gz_bench.err; echo "bench gz rc=$?"
  grep "gunzip\|resolved\|CPU seconds" gpurun_out/gz_bench.err | tail -24
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/gz_bench.log").read().strip().splitlines()[-1])
print("value", d["value"], "file_to_gz", d["file_to_gz"]["mreads_s"], "gz", json.dumps({k: v for k, v in d["file_to_file_gz"].items() if k != "thread_seconds_last_run"}))
PY
fi
if has cold; then
  # a fresh process per run (the CLI): pass 2 of a one-member .gz -> .gz run, default settings, then the host pool alone
  P=${GZ_PAIRS:-5000000}
  python tools/e2e_bench.py --pairs $P --gz --gz-level 6 --keep --dir /tmp/aqc_gzcold 2>&1 | tail -1 | cut -c1-900 | tee gpurun_out/gz_cold.log
  for k in 1 2; do
    AQC_GZ_DEBUG=1 python tools/e2e_bench.py --pairs $P --gz --gz-level 6 --keep --reuse --dir /tmp/aqc_gzcold 2>&1 | grep "gz dev\|mode" | cut -c1-900 | tee -a gpurun_out/gz_cold.log
  done
  AQC_GZ_DEVICE_IN=0 python tools/e2e_bench.py --pairs $P --gz --gz-level 6 --keep --reuse --dir /tmp/aqc_gzcold 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/gz_cold.log
  python tools/e2e_bench.py --pairs 2000000 --gz --config5 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/gz_cold.log
  AQC_GZ_DEVICE_IN=0 python tools/e2e_bench.py --pairs 2000000 --gz --config5 2>&1 | tail -1 | cut -c1-900 | tee -a gpurun_out/gz_cold.log
  rm -rf /tmp/aqc_gzcold
fi
if has ab; then
  # resident results against the round-5 way (symbols back to the host, translated and checksummed there), warm pipe, same box
  for r in 1 0 1 0; do
    AQC_GZ_RESIDENT=$r timeout 600 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --pipe-runs 0 --device-steps 1 --gz-runs 3 --no-pmc --no-fused-step --inputs 1 --big-copies 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['file_to_file_gz']
print('AQC_GZ_RESIDENT=$r', 'mreads_s', g['mreads_s'], 'median_s', g['median_seconds'], 'first', g['first_run_seconds'], 'host_only', g['host_only_mreads_s'], 'device share', g['gunzip_text_share_from_device'])" | tee -a gpurun_out/gz_ab.log
  done
fi
if has hbm; then
  # the device-decoded text staying in HBM (aqc_frame_mixed) against fetching it into the chunk buffers (AQC_GZ_HBM=0), and the
  # group size, warm pipe, same box, interleaved
  for v in "AQC_GZ_HBM=1" "AQC_GZ_HBM=0" "AQC_GZ_GROUP=100663296" "AQC_GZ_HBM=1" "AQC_GZ_HBM=0" "AQC_GZ_GROUP=100663296" "AQC_GZ_GROUP=16777216"; do
    env $v timeout 600 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --pipe-runs 0 --device-steps 1 --gz-runs 4 --no-pmc --no-fused-step --inputs 1 --big-copies 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['file_to_file_gz']
print('$v', 'mreads_s', g['mreads_s'], 'median_s', g['median_seconds'], 'first', g['first_run_seconds'], 'host_only', g['host_only_mreads_s'], 'device share', g['gunzip_text_share_from_device'], 'file_to_gz', d['file_to_gz']['mreads_s'])" | tee -a gpurun_out/gz_hbm_ab.log
  done
fi
if has prof; then
  # which kernels a .gz -> .gz run spends the GPU's time in
  OUT=$GRAFT_REPO_ROOT/gpurun_out; rm -rf $OUT/prof_gz
  B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pipe-runs 0 --device-steps 1 --gz-runs 3 --no-pmc --no-fused-step --inputs 1 --big-copies 0"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_gz -o kt -- $B > /dev/null 2>&1)
  python tools/pmc_summary.py $OUT/prof_gz | sort -t= -k3 -n -r | head -40 | cut -c1-170 | tee gpurun_out/gz_prof_summary.txt
  python - <<'PY'
import csv, glob
# GPU busy: union of the kernel intervals over the trace
rows = []
for f in glob.glob("gpurun_out/prof_gz/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
if rows:
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, _ in rows[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("kernels: %d launches, GPU busy (union of kernel intervals) %.1f ms of a %.1f ms trace; sum of durations %.1f ms" % (len(rows), busy / 1e6, (rows[-1][1] - rows[0][0]) / 1e6, sum(e - s for s, e, _ in rows) / 1e6))
PY
fi
