#!/bin/bash
# round 4, GPU call 32: the CLI (fresh process) on one-member gzip -2 inputs: host alone vs pool + device (what the device's set-up costs a one-shot run)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c32; mkdir -p $O
AQC_GZ_DEVICE_IN=0 timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 2 --keep --dir /tmp/e2e_a 2> $O/host.err | tail -1 > $O/host.json
AQC_PIPE_DEBUG=1 timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 2 --keep --reuse --dir /tmp/e2e_a 2> $O/dev.err | tail -1 > $O/dev.json
AQC_GZ_DEVICE_IN=0 timeout 600 python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 2 --keep --reuse --dir /tmp/e2e_a 2> $O/host2.err | tail -1 > $O/host2.json
python - <<'PY'
import json
for n in ("host", "dev", "host2"):
    d = json.loads(open("gpurun_out/r4c32/%s.json" % n).read())
    print(n, "wall", d["wall_s"], "pass1", d["pass1_s"], "pass2", d["pass2_s"], "Mreads/s", d["pass2_mreads_s"], "pipe", d["pipe_threads"])
PY
grep -E "gunzip consumer|device gunzip so far|readers / workers done|pipe: gunzip —" $O/dev.err | cut -c1-300
