#!/bin/bash
# round 5: GPU tests, the PCIe ceiling (tools/ubench/pcie_rate.hip), and an A/B of the I/O threads' NUMA binding through the CLI path
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${1:-r5misc}; mkdir -p $D; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $D/all.log 2>&1; echo rc=$? >> $D/all.log; tail -4 $D/all.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie_rate.hip -o /tmp/pcie_rate 2>/dev/null && timeout 120 /tmp/pcie_rate > $D/pcie_rate.txt 2>&1; cat $D/pcie_rate.txt
{
python tools/e2e_bench.py --pairs 5000000 --dir /tmp/aqc_e2e_ab --keep
for rep in 1 2; do
  python tools/e2e_bench.py --pairs 5000000 --dir /tmp/aqc_e2e_ab --keep --reuse
  AQC_PIPE_NUMA=0 python tools/e2e_bench.py --pairs 5000000 --dir /tmp/aqc_e2e_ab --keep --reuse
done
rm -rf /tmp/aqc_e2e_ab
} > $D/numa_ab.txt 2> $D/numa_ab.err
python - <<PY
import json
for k, ln in enumerate(l for l in open("$D/numa_ab.txt") if l.startswith('{"mode"')):
    d = json.loads(ln)
    print("run", k, "(default)" if k in (0, 1, 3) else "(AQC_PIPE_NUMA=0)", {x: d[x] for x in ("wall_s", "pass1_s", "pass2_s", "pass2_mreads_s", "init_s")}, d["pipe_threads"])
PY
