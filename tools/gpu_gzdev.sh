#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, sys, gzip
import numpy as np
sys.path.insert(0, ".")
from afterqc_amd import capi, synth
d = synth.make_pairs(2000000, 150, seed=1003, workers=32)
eng = capi.Engine(0, 2)
cfg = capi.Config(); cfg.paired = 1
cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
cfg.barcode_length = 12; cfg.set_verify("CAGTA"); cfg.qc_kmer = 8
eng.set_config(cfg); eng.reset_stats()
w = synth.fixed_record_width(150)
hb1 = eng.host_buffer(len(d["len1"]) * w + 4096); hb2 = eng.host_buffer(len(d["len1"]) * w + 4096)
_, n1 = synth.render_fastq_fixed(d["seq1"], d["qual1"], 1, out=hb1.array)
_, n2 = synth.render_fastq_fixed(d["seq2"], d["qual2"], 2, out=hb2.array)
info = eng.frame(0, hb1.array, n1, True, hb2.array, n2, True)
eng.run(0); sizes = eng.format(0, int(info.n), False); eng.sync(0)
for rep in range(3):
    t0 = time.perf_counter(); gz = eng.compress(0, 2); t1 = time.perf_counter()
    print("compress: %.2f ms for %.2f GB of text -> %.2f GB (ratio %.3f): %.1f GB/s" % (1e3 * (t1 - t0), sum(sizes) / 1e9, sum(gz) / 1e9, sum(sizes) / sum(gz), sum(sizes) / (t1 - t0) / 1e9))
# host codec on the same text for comparison
text = np.zeros(sizes[0] + 64, dtype=np.uint8); eng.fetch_text(0, 0, 0, text, text.size)
sample = text[:64 << 20].tobytes()
t0 = time.perf_counter(); z = capi.bgzf_compress(sample, 2); t1 = time.perf_counter()
print("host codec (1 thread): ratio %.3f, %.0f MB/s" % (len(sample) / len(z), len(sample) / (t1 - t0) / 1e6))
import zlib
t0 = time.perf_counter(); zz = zlib.compress(sample[:16 << 20], 2); t1 = time.perf_counter()
print("zlib level 2: ratio %.3f, %.0f MB/s" % ((16 << 20) / len(zz), (16 << 20) / (t1 - t0) / 1e6))
PY
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --pipe-runs 1 --device-steps 3 --gz-runs 2 > gpurun_out/bench_gz.log 2> gpurun_out/bench_gz.err; echo "bench gz rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_gz.log").read().strip().splitlines()[-1])
print("value", d["value"], "gz", json.dumps(d["file_to_file_gz"]))
PY
python tools/e2e_bench.py --pairs 5000000 --gz 2>>gpurun_out/e2e_gz.err | tail -1 | cut -c1-700
python tools/e2e_bench.py --pairs 2000000 --gz --config5 2>>gpurun_out/e2e_gz.err | tail -1 | cut -c1-700
