#!/bin/bash
# the CLI end to end on six inputs: config 3 / config-5 flavour x plain / ONE-member .gz (gzip -6) / BGZF .gz  -> gpurun_out/e2e_cli.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/e2e_cli.txt
run() { timeout 900 python tools/e2e_bench.py "$@" 2>>gpurun_out/e2e.err | tail -1 >> gpurun_out/e2e_cli.txt; }
run --pairs 5000000
run --pairs 5000000 --gz --gz-level 6 --keep --dir /tmp/e2e_a
run --pairs 5000000 --gz --gz-level 6 --keep --reuse --dir /tmp/e2e_a
rm -rf /tmp/e2e_a
run --pairs 5000000 --gz --bgzf
run --pairs 2000000 --config5
run --pairs 2000000 --config5 --gz --gz-level 6 --keep --dir /tmp/e2e_b
run --pairs 2000000 --config5 --gz --gz-level 6 --keep --reuse --dir /tmp/e2e_b
rm -rf /tmp/e2e_b
run --pairs 2000000 --config5 --gz --bgzf
python - <<'PY'
import json
for line in open("gpurun_out/e2e_cli.txt"):
    d = json.loads(line)
    print("cfg5" if d["config5"] else "cfg3", d["gz"], "gen", d["gen_s"], "wall", d["wall_s"], "pass1", d["pass1_s"], "pass2", d["pass2_s"], "Mreads/s", d["pass2_mreads_s"], "cores", d.get("pass2_cores_busy"), "pipe", d["pipe_threads"]["seconds"])
PY
