#!/bin/bash
# round 6: tests named by $K (default: everything but the soak), then $WHAT of: bench (the driver's line), ab (AQC_SPANS 0 / 1 / 2 interleaved), dmaw (tools/ubench/dma_write_rate)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
WHAT="${WHAT:-bench}"
if [ "${TESTS:-1}" = 1 ]; then
  timeout 2800 python -m pytest ${FILES:-tests} -m gpu -q -x -k "${K:-not soak}" > gpurun_out/pytest_r6.log 2>&1; echo "pytest rc=$?"; grep -v "^{\|options:$" gpurun_out/pytest_r6.log | tail -${TAILN:-8} | cut -c1-400
fi
if [[ " $WHAT " == *" dmaw "* ]]; then
  hipcc --offload-arch=gfx950 -O2 -o /tmp/dma_write_rate tools/ubench/dma_write_rate.hip -lpthread 2>/dev/null && /tmp/dma_write_rate /tmp 1.7 | tee gpurun_out/dma_write_rate.txt
fi
if [[ " $WHAT " == *" bench "* ]]; then
  timeout 900 python bench.py > gpurun_out/bench_r6.json 2> gpurun_out/bench_r6.err; echo "bench rc=$?"
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r6.json").read().strip().splitlines()[-1])
print("value", d["value"], "median", d.get("value_median"), "best", d.get("value_best"), "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["kernel_ms"])
print("device_step", d["device_step"]["ms_per_step"], "spans", d["device_step_spans"]["ms_per_step"], "fused", (d.get("device_step_fused") or {}).get("ms_per_step"))
print("p2p", d["pinned_to_pinned_mreads_s"], "multi", d["multi_input_file_to_file_mreads_s"], "100M", (d.get("file_to_file_100M") or {}).get("mreads_s"), "f2gz", (d.get("file_to_gz") or {}).get("mreads_s"))
g = d.get("file_to_file_gz") or {}
print("gz", g.get("mreads_s"), "median_s", g.get("median_seconds"), "host_only", g.get("host_only_mreads_s"), "share", g.get("gunzip_text_share_from_device"), "first", g.get("first_run_seconds"))
print("host", d["host"])
PY
fi
if [[ " $WHAT " == *" ab "* ]]; then
  for v in "AQC_SPANS=0" "AQC_SPANS=2" "AQC_SPANS=1" "AQC_SPANS=0" "AQC_SPANS=2" "AQC_SPANS=1"; do
    env $v timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pipe-runs 3 --device-steps 5 --gz-runs 0 --no-pmc --no-fused-step --big-copies 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'value', d['value'], 'median', d['value_median'], 'best', d['value_best'], 'p2p', d['pinned_to_pinned_mreads_s'], 'multi', d['multi_input_file_to_file_mreads_s'], 'threads', d['file_to_file']['thread_seconds_last_run'])" | tee -a gpurun_out/spans_ab.log
  done
fi
