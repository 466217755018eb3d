#!/bin/bash
# round 4, final GPU call: smoke, every -m gpu test, the bench line, then the CLI end to end on .gz inputs (gzip -2: the level bench.py uses)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_final.log 2> gpurun_out/bench_final.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/bench_final.log
: > gpurun_out/e2e_cli_r4.txt
run() { timeout 600 python tools/e2e_bench.py "$@" 2>>gpurun_out/e2e_r4.err | tail -1 >> gpurun_out/e2e_cli_r4.txt; }
run --pairs 5000000
run --pairs 5000000 --gz --gz-level 2 --keep --dir /tmp/e2e_a
run --pairs 5000000 --gz --gz-level 2 --keep --reuse --dir /tmp/e2e_a
rm -rf /tmp/e2e_a
run --pairs 2000000 --config5 --gz --gz-level 2
python - <<'PY'
import json
for line in open("gpurun_out/e2e_cli_r4.txt"):
    d = json.loads(line)
    print("cfg5" if d["config5"] else "cfg3", d["gz"], "gen", d["gen_s"], "wall", d["wall_s"], "pass1", d["pass1_s"], "pass2", d["pass2_s"], "Mreads/s", d["pass2_mreads_s"], "cores", d.get("pass2_cores_busy"), "pipe", d["pipe_threads"]["seconds"])
PY
