#!/bin/bash
# round 5: config 5 at its own shape — the bench line of --workload config5 (2 x 267 with barcodes; its .gz legs on), then the CLI
# path (fresh processes) over the config-5 flavour plain -> plain and one-member .gz -> .gz, and a .bz2 input through the pipe
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${1:-r5c5}; mkdir -p $D; export TMPDIR=/tmp
python bench.py --workload config5 --pairs 3000000 --big-copies 0 --no-pmc --cpu-sample 0 > $D/bench_config5.json 2> $D/bench_config5.err; tail -c 300 $D/bench_config5.err
python - <<PY
import json
d = json.loads(open("$D/bench_config5.json").read().strip().splitlines()[-1])
for k in ("metric", "value", "ms_per_step", "device_step", "device_step_spans", "roofline", "pinned_to_pinned", "file_to_file", "multi_input_file_to_file", "file_to_file_gz", "file_to_gz"):
    print(k, d.get(k))
PY
{
for rep in 1 2; do python tools/e2e_bench.py --pairs 2000000 --config5 --dir /tmp/aqc_e2e_c5; done
for rep in 1 2; do python tools/e2e_bench.py --pairs 2000000 --config5 --gz --dir /tmp/aqc_e2e_c5gz; done
python tools/e2e_bench.py --pairs 2000000 --config5 --gz --bgzf --dir /tmp/aqc_e2e_c5bg
for rep in 1 2; do python tools/e2e_bench.py --pairs 5000000 --gz --gz-level 6 --dir /tmp/aqc_e2e_gz; done
python tools/e2e_bench.py --pairs 5000000 --dir /tmp/aqc_e2e_plain
python tools/e2e_bench.py --pairs 1000000 --bz2 --dir /tmp/aqc_e2e_bz2
python tools/e2e_bench.py --pairs 1000000 --bz2 --mode text --dir /tmp/aqc_e2e_bz2t
} > $D/e2e_cli.txt 2> $D/e2e_cli.err
python - <<PY
import json
for ln in open("$D/e2e_cli.txt"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print({k: d[k] for k in ("mode", "gz", "config5", "reads", "wall_s", "pass1_s", "pass2_s", "e2e_mreads_s", "pass2_mreads_s", "pass2_cores_busy", "used_pipe")}, d["input_bytes"])
PY
tail -3 $D/e2e_cli.err
