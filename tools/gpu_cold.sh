#!/bin/bash
# a fresh process per run (the CLI): pass 2 of a one-member .gz -> .gz run with the default settings, then the host pool alone;
# config 3 (5 M pairs) and the config-5 flavour (2 M pairs, barcodes + --debubble)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=${GZ_LEVEL:-1}
pick='import json,sys
for ln in sys.stdin:
    if ln.startswith("{\"mode\""):
        d=json.loads(ln); print(sys.argv[1], "pass2_s", d["pass2_s"], "pass2_mreads_s", d["pass2_mreads_s"], "cores_busy", d["pass2_cores_busy"], "wall_s", d["wall_s"], "pipe_s", (d.get("pipe_threads") or {}).get("seconds"))'
python tools/e2e_bench.py --pairs 5000000 --gz --gz-level $L --keep --dir /tmp/aqc_c3 2>/dev/null | python -c "$pick" "config3 default (warm-up: files just made)" | tee gpurun_out/gz_cold.log
for k in 1 2 3; do
  python tools/e2e_bench.py --pairs 5000000 --gz --gz-level $L --keep --reuse --dir /tmp/aqc_c3 2>/dev/null | python -c "$pick" "config3 default" | tee -a gpurun_out/gz_cold.log
  AQC_GZ_DEVICE_IN=0 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level $L --keep --reuse --dir /tmp/aqc_c3 2>/dev/null | python -c "$pick" "config3 host pool alone" | tee -a gpurun_out/gz_cold.log
done
AQC_GZ_DEBUG=1 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level $L --keep --reuse --dir /tmp/aqc_c3 2>&1 | grep "gz dev" | cut -c1-220 | tee -a gpurun_out/gz_cold.log
rm -rf /tmp/aqc_c3
python tools/e2e_bench.py --pairs 2000000 --gz --gz-level $L --config5 --keep --dir /tmp/aqc_c5 2>/dev/null | python -c "$pick" "config5 flavour default (warm-up)" | tee -a gpurun_out/gz_cold.log
for k in 1 2 3; do
  python tools/e2e_bench.py --pairs 2000000 --gz --gz-level $L --config5 --keep --reuse --dir /tmp/aqc_c5 2>/dev/null | python -c "$pick" "config5 flavour default" | tee -a gpurun_out/gz_cold.log
  AQC_GZ_DEVICE_IN=0 python tools/e2e_bench.py --pairs 2000000 --gz --gz-level $L --config5 --keep --reuse --dir /tmp/aqc_c5 2>/dev/null | python -c "$pick" "config5 flavour host pool alone" | tee -a gpurun_out/gz_cold.log
done
rm -rf /tmp/aqc_c5
