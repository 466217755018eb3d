#!/usr/bin/env python3
"""the device text step (reframe -> run -> statRead -> format, bench.py's step A) on the 2 x 150 workload, with and without
AQC_FUSED=1, interleaved: ms per step, which kernels ran for how long, how many pairs the general kernel had to finish, and
that both ways leave the same bytes.  Usage: fused_step.py [pairs] [steps]"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afterqc_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = synth.make_pairs(n, 150, seed=1003)
cfg = capi.Config()
cfg.paired = 1
cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
cfg.barcode_length = 12
cfg.set_verify("CAGTA")
cfg.qc_kmer = 8
engines = {}
for name, env in (("two_step", None), ("fused", "1")):
    if env:
        os.environ["AQC_FUSED"] = env
    else:
        os.environ.pop("AQC_FUSED", None)
    engines[name] = capi.Engine(0, 2)
os.environ.pop("AQC_FUSED", None)
w = synth.fixed_record_width(d["seq1"].shape[1])
crc = {}
for name, eng in engines.items():
    eng.set_config(cfg)
    eng.reset_stats()
    texts = []
    for mate in (1, 2):
        hb = eng.host_buffer(n * w + 4096)
        _, nbytes = synth.render_fastq_fixed(d["seq%d" % mate], d["qual%d" % mate], mate, out=hb.array, index0=0)
        texts.append((hb, nbytes))
    info = eng.frame(0, texts[0][0].array, texts[0][1], True, texts[1][0].array, texts[1][1], True)
    assert int(info.n) == n
    eng.sync(0)
    engines[name] = (eng, texts)


n_qc = min(n, 199_999)


def step(eng):
    eng.reframe(0)
    eng.run(0)
    eng.qc_stat(0, capi.QC_R1_POST, 0, 0, n_qc, 1)        # (bench.py's --qc-sample default)
    eng.qc_stat(0, capi.QC_R2_POST, 1, 0, n_qc, 1)
    return eng.format(0, n, False)


for name, (eng, _) in engines.items():
    for _ in range(3):
        sizes = step(eng)
    eng.sync(0)
    out = []
    for q in (0, 1, 3, 4):
        buf = np.zeros(sizes[q] + 1, dtype=np.uint8)
        eng.fetch_text(0, q // 3, q % 3, buf, sizes[q])
        out.append(zlib.crc32(buf[:sizes[q]].tobytes()))
    crc[name] = (sizes, out)
    print(name, "fused placement taken:", eng.format_fused(0), "deferred pairs:", eng.last_deferred(0), "sizes:", sizes, flush=True)
print("same bytes both ways:", crc["two_step"] == crc["fused"], flush=True)
res = {k: [] for k in engines}
for rnd in range(3):
    for name, (eng, _) in engines.items():
        eng.timing_reset(0)
        eng.sync(0)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(eng)
        eng.sync(0)
        ms = 1000 * (time.perf_counter() - t0) / steps
        res[name].append(ms)
        kms, kn = eng.timing_mean(0)
        print("round %d %-9s %.3f ms/step   verdict kernel %.3f ms" % (rnd, name, ms, float(kms[capi.K_FILTER_OVERLAP])), flush=True)
for name in res:
    print("%-9s best %.3f ms/step  (%.1f Mreads/s)" % (name, min(res[name]), 2 * n / min(res[name]) / 1e3))
