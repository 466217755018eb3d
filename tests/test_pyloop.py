"""The pure-Python stand-in for the reference's loop (oracle/pyloop.py, used by bench.py's cpu_baseline as the
"CPython path" figure) against the C oracle, verdict for verdict, on dirty synthetic pairs under several option sets."""
import numpy as np
import pytest

from afterqc_amd import capi, synth
from oracle import oracle, pyloop


def cfg_of(paired=True, **kw):
    cfg = capi.Config()
    cfg.paired = 1 if paired else 0
    cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
    cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
    cfg.barcode_length = 12
    cfg.set_verify("CAGTA")
    cfg.qc_kmer = 8
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("case", ["default", "trim", "nocorr", "mask", "strict", "ragged", "nooverlap", "se"])
def test_pyloop_matches_c_oracle(case):
    kw = dict(n=1200, L=150, seed=77, dirty=True, workers=1)
    cfgkw = {}
    paired = case != "se"
    if case == "trim":
        cfgkw = dict(trim_front=3, trim_tail=2, trim_front2=1, trim_tail2=4)
    elif case == "nocorr":
        cfgkw = dict(no_correction=1)
    elif case == "mask":
        cfgkw = dict(mask_mismatch=1)
    elif case == "strict":
        cfgkw = dict(qualified_quality_phred=20, unqualified_base_limit=20, poly_size_limit=20, allow_mismatch_in_poly=1,
                     n_base_limit=1, seq_len_req=100)
    elif case == "ragged":
        kw.update(ragged=True, short_frac=0.3)
        cfgkw = dict(seq_len_req=20, trim_front=1)
    elif case == "nooverlap":
        cfgkw = dict(no_overlap=1)
    d = synth.make_pairs(**kw)
    if paired:
        batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    else:
        batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"])
    cfg = cfg_of(paired, **cfgkw)
    eng = oracle.OracleEngine()
    eng.set_config(cfg)
    eng.upload(0, batch)
    eng.run(0)
    res = eng.fetch_results(0)
    py = pyloop.run_batch(batch, cfg)
    assert len(py) == batch.n
    for i, (p, r) in enumerate(zip(py, res)):
        assert p["flag"] == int(r["flag"]), (i, p["flag"], r)
        s1, q1 = batch.read1(i)
        f1 = oracle.final_read(s1, q1, r, 1)
        assert (p["seq1"].encode("latin-1"), p["qual1"].encode("latin-1")) == f1, i
        if paired:
            s2, q2 = batch.read2(i)
            assert (p["seq2"].encode("latin-1"), p["qual2"].encode("latin-1")) == oracle.final_read(s2, q2, r, 2), i
            if p["flag"] in (pyloop.GOOD, pyloop.BADDIFF, pyloop.BADMISMATCH):
                assert (p["offset"], p["overlap_len"], p["distance"]) == (int(r["offset"]), int(r["overlap_len"]), int(r["distance"])), i
            assert len(p["edits"]) == int(r["n_edits"])
            for k, (o, kind, base, qual) in enumerate(p["edits"]):
                e = r["edits"][k]
                assert (o, kind) == (int(e["o"]), int(e["kind"]))
                if kind != pyloop.EDIT_MASK:
                    assert (ord(base), ord(qual)) == (int(e["base"]), int(e["qual"]))
    flags = np.bincount(res["flag"], minlength=12)
    assert flags[capi.GOOD] > 0
