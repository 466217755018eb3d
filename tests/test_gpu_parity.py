"""GPU parity (-m gpu): the HIP path, called through the C ABI, against (a) the golden vectors of
the real reference and (b) the oracle on seeded inputs.  Bit-exact everywhere (integer/byte work)."""
import numpy as np
import pytest

from afterqc_amd import capi, synth

pytestmark = pytest.mark.gpu


def default_cfg(paired=True, **kw):
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


# ---- function seams vs the reference's golden vectors -------------------------------------------------
def test_overlap_golden(gpu_engine, fvec):
    vec = fvec["overlap"]
    b = capi.Batch.from_strings([v[0] for v in vec], None, [v[1] for v in vec], None)
    off, ol, df = gpu_engine.overlap(b)
    got = np.stack([off, ol, df], axis=1).tolist()
    exp = [v[2] for v in vec]
    bad = [(vec[i][0], vec[i][1], exp[i], got[i]) for i in range(len(vec)) if got[i] != exp[i]]
    assert not bad, bad[:3]


def test_read_stats_golden(gpu_engine, fvec):
    for mp, mm in ((35, 2), (20, 0), (10, 3), (50, 5)):
        vec = [v for v in fvec["polyx"] if v[1] == mp and v[2] == mm and len(v[0]) > 0]
        b = capi.Batch.from_strings([v[0] for v in vec])
        px, _, _ = gpu_engine.read_stats(b, mp, mm, 15)
        exp = [0 if v[3] is None else ord(v[3]) for v in vec]
        assert px.tolist() == exp
    for qv in sorted(set(v[2] for v in fvec["counts"])):
        vec = [v for v in fvec["counts"] if v[2] == qv]
        b = capi.Batch.from_strings([v[0] for v in vec], [v[1] for v in vec])
        _, lq, nn = gpu_engine.read_stats(b, 35, 2, qv)
        assert lq.tolist() == [v[3] for v in vec]
        assert nn.tolist() == [v[4] for v in vec]


def test_edit_distance_golden(gpu_engine, fvec):
    vec = [v for v in fvec["editdistance"] if min(len(v[0]), len(v[1])) <= 64]
    b = capi.Batch.from_strings([v[0] for v in vec], None, [v[1] for v in vec], None)
    d = gpu_engine.edit_distance(b)
    assert d.tolist() == [v[2] for v in vec]


def test_libed_compatible_symbols(fvec):
    """The reference's own native seam (editdistance/_editdistance.h:16,23), loaded the way util.py:16-24 loads libed.so:
    edit_distance against the reference's editDistance vectors (any length), seek_overlap(r1, reverse_r2, 3, 50, 30)
    against the reference's util.overlap vectors."""
    from ctypes import cdll
    ed_ctypes = cdll.LoadLibrary(capi.LIB_PATH)
    for a, b, d in fvec["editdistance"]:
        assert ed_ctypes.edit_distance(a.encode(), len(a), b.encode(), len(b)) == d
    comp = {"A": "T", "T": "A", "C": "G", "G": "C", "a": "t", "t": "a", "c": "g", "g": "c", "N": "N"}
    vec = fvec["overlap"]
    for r1, r2, exp in vec[:400] + vec[5200:]:
        rr2 = "".join(comp.get(c, "N") for c in reversed(r2))
        ret = ed_ctypes.seek_overlap(r1.encode("latin-1"), len(r1), rr2.encode("latin-1"), len(r2), 3, 50, 30)
        if ret == 0x7FFFFFFF:
            got = [0, 0, 0]
        else:
            offset, diff = ret >> 8, ret & 0xFF                      # decoded like util.py:224-225
            ol = min(len(r1) - offset, len(r2)) if offset >= 0 else min(len(r1), len(r2) - abs(offset))
            got = [offset, ol, diff]
        assert got == exp, (r1, r2, exp, got)


def test_overlap_golden_through_the_production_kernel(gpu_engine, fvec):
    """The reference's util.overlap vectors (incl. the i = 49/50/51 and L = 51/52 boundary set, util.py:183,206) through
    aqc_run — i.e. through the lane-per-read kernel and its closed form, not the function seam: every filter off, no trim,
    the result record's (offset, overlap_len, distance) must be the golden triple unless the adapter cut re-ran the scan
    (offset < 0 and overlap_len > 30, preprocesser.py:520-534), in which case the record holds the second call's triple,
    which is util.overlap of the two reads cut to overlap_len — also a golden-checkable value via the seam's semantics."""
    comp_keys = set("ATCGatcgN")          # util.COMP (util.py:27): anything else raises KeyError in the walk upstream
    vec = [v for v in fvec["overlap"] if len(v[0]) > 0 and len(v[1]) > 0 and set(v[0] + v[1]) <= comp_keys]
    b = capi.Batch.from_strings([v[0] for v in vec], None, [v[1] for v in vec], None)
    cfg = default_cfg(True, seq_len_req=0, poly_size_limit=0, unqualified_base_limit=0, n_base_limit=0, no_correction=1)
    gpu_engine.set_config(cfg)
    gpu_engine.reset_stats()
    gpu_engine.upload(0, b)
    gpu_engine.run(0)
    res = gpu_engine.fetch_results(0)
    n_def, deferred = gpu_engine.last_deferred(0, want_indices=True)
    from oracle import oracle
    checked = 0
    for i, (r1, r2, exp) in enumerate(vec):
        r = res[i]
        got = [int(r["offset"]), int(r["overlap_len"]), int(r["distance"])]
        if exp[0] < 0 and exp[1] > 30:
            # adapter cut: both reads := [0:overlap_len], util.overlap again (preprocesser.py:520-534); the second call is
            # checked against the (golden-pinned) oracle
            assert int(r["len1"]) == exp[1] and int(r["len2"]) == exp[1]
            assert got == list(oracle.overlap(r1[:exp[1]], r2[:exp[1]])), (i, r1, r2, exp, got)
        else:
            assert got == exp, (i, r1, r2, exp, got)
        checked += 1
    assert checked > 5000
    # the lane-per-read kernel must really have decided them: of the pairs it is built for (A C G T N only, both reads at
    # least 16 bases) it may hand on only the rare second-scan / walk corners
    plain = set("ACGTN")
    eligible = np.array([set(v[0]) <= plain and set(v[1]) <= plain and min(len(v[0]), len(v[1])) >= 16 for v in vec])
    was_deferred = np.zeros(len(vec), dtype=bool)
    was_deferred[deferred] = True
    assert eligible.sum() > 4000
    assert was_deferred[eligible].mean() < 0.05, was_deferred[eligible].mean()


# ---- whole batches vs the oracle ---------------------------------------------------------------------------
def run_both(gpu_engine, cfg, batch, circles=(), qc=True, accum_limit=capi.UINT64_MAX):
    from oracle import oracle
    out = []
    for eng in (gpu_engine, oracle.OracleEngine()):
        eng.set_config(cfg)
        eng.set_circles(list(circles))
        eng.reset_stats()
        eng.upload(0, batch)
        if qc:
            eng.qc_stat(0, capi.QC_R1_PRE, 0, 0, batch.n, 0)
            if cfg.paired:
                eng.qc_stat(0, capi.QC_R2_PRE, 1, 0, batch.n, 0)
        eng.run(0, accum_limit)
        if qc:
            eng.qc_stat(0, capi.QC_R1_POST, 0, 0, batch.n, 1)
            if cfg.paired:
                eng.qc_stat(0, capi.QC_R2_POST, 1, 0, batch.n, 1)
        res = eng.fetch_results(0)
        kms = []
        if qc:
            for w in ((0, 1, 2, 3) if cfg.paired else (capi.QC_R1_PRE, capi.QC_R1_POST)):
                keys, counts, order = eng.kmers(w)
                idx = np.argsort(order, kind="stable")
                kms.append((keys[idx].tolist(), counts[idx].tolist()))
        out.append(dict(res=res, counters=eng.counters(), hist=eng.histograms(), qc=[eng.qc(w) for w in range(4)], kmers=kms))
    return out


def assert_same(g, o):
    diff = np.flatnonzero(g["res"].view(np.uint8).reshape(-1, 32) != o["res"].view(np.uint8).reshape(-1, 32))
    if len(diff):
        i = diff[0] // 32
        raise AssertionError("record %d differs:\n gpu    %s\n oracle %s" % (i, g["res"][i], o["res"][i]))
    assert g["counters"].tolist() == o["counters"].tolist()
    assert g["hist"][0].tolist() == o["hist"][0].tolist()
    assert g["hist"][1].tolist() == o["hist"][1].tolist()
    for w in range(4):
        assert np.array_equal(g["qc"][w], o["qc"][w]), "QC accumulator %d" % w
    assert g["kmers"] == o["kmers"]


@pytest.mark.parametrize("case", ["default", "trim", "nocorr", "mask", "nocorr_mask", "nooverlap", "strict", "ragged",
                                  "short", "lowercase", "l250", "l100", "index2", "l400", "l700_ragged", "l1000_max"])
def test_pairs_vs_oracle(gpu_engine, case):
    kw = dict(n=6000, L=150, seed=4242, dirty=True)
    cfgkw = {}
    if case == "trim":
        cfgkw = dict(trim_front=3, trim_tail=2, trim_front2=1, trim_tail2=4)
    elif case == "nocorr":
        cfgkw = dict(no_correction=1)
    elif case == "mask":
        cfgkw = dict(mask_mismatch=1)
    elif case == "nocorr_mask":
        cfgkw = dict(no_correction=1, mask_mismatch=1)
    elif case == "nooverlap":
        cfgkw = dict(no_overlap=1)
    elif case == "strict":
        cfgkw = dict(qualified_quality_phred=20, unqualified_base_limit=20, poly_size_limit=20, allow_mismatch_in_poly=1,
                     n_base_limit=1, seq_len_req=100)
    elif case == "ragged":
        kw.update(ragged=True)
        cfgkw = dict(seq_len_req=20)
    elif case == "short":
        kw.update(ragged=True, short_frac=0.6)
        cfgkw = dict(seq_len_req=20, trim_front=1, trim_front2=1)
    elif case == "lowercase":
        kw.update(lowercase=0.2)
        cfgkw = dict(no_correction=1)
    elif case == "l250":
        kw.update(L=250, n=2500)
    elif case == "l100":
        kw.update(L=100)
    elif case == "l400":
        # beyond the lane-per-read kernel's 256 bases: general kernel, multi-pass QC / k-mer kernels (unfused)
        kw.update(L=400, n=1200)
    elif case == "l1000_max":
        # AQC_MAX_READ_LEN: the longest read the ABI takes
        kw.update(L=1000, n=300)
    elif case == "l700_ragged":
        kw.update(L=700, n=600, ragged=True)
        cfgkw = dict(seq_len_req=20, trim_front=2, trim_tail=1, trim_front2=3, trim_tail2=2)
    elif case == "index2":
        cfgkw = dict(count_r2_bases=1)
    d = synth.make_pairs(**kw)
    batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    g, o = run_both(gpu_engine, default_cfg(True, **cfgkw), batch)
    assert_same(g, o)


def test_every_length_pairs_vs_oracle(gpu_engine):
    """Read 2 is packed in 16-byte chunks counted from its END (and both mates are padded behind their last base): every
    length 1..70 for either mate, the shortest ones first so that the first record's leading chunk begins BEFORE the
    arena, with overlapping tails so that the scan / verification / correction walk see the stream ends."""
    rng = np.random.default_rng(20260927)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    seqs1, quals1, seqs2, quals2 = [], [], [], []
    for l1 in list(range(1, 71)) + [150, 160]:
        for l2 in (1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 70, 150, 160, l1):
            ins = "".join(rng.choice(list("ACGT"), size=max(l1, l2) + 7))
            r1 = list(ins[:l1])
            r2 = [comp[c] for c in reversed(ins[max(0, len(ins) - l2 - 7):len(ins) - 7])][:l2]
            r2 += list(rng.choice(list("ACGT"), size=l2 - len(r2)))
            for r in (r1, r2):                           # a few mismatches / N
                for k in rng.integers(0, len(r), size=len(r) // 25):
                    r[k] = "ACGTN"[rng.integers(0, 5)]
            seqs1.append("".join(r1)); seqs2.append("".join(r2))
            quals1.append("".join(rng.choice(list("#/5?I"), size=l1))); quals2.append("".join(rng.choice(list("#/5?I"), size=l2)))
    batch = capi.Batch.from_strings(seqs1, quals1, seqs2, quals2)
    for cfgkw in (dict(seq_len_req=1), dict(seq_len_req=1, trim_front=1, trim_tail=2, trim_front2=2, trim_tail2=1)):
        g, o = run_both(gpu_engine, default_cfg(True, **cfgkw), batch, qc=False)
        assert_same(g, o)


def test_single_end_vs_oracle(gpu_engine):
    d = synth.make_single(8000, 150, seed=99)
    batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"])
    g, o = run_both(gpu_engine, default_cfg(False, trim_front=5, trim_tail=5), batch)
    assert_same(g, o)


@pytest.mark.parametrize("L,ragged", [(120, False), (250, False), (239, False), (100, True)])
def test_barcode_and_bubble_vs_oracle(gpu_engine, L, ragged):
    """barcode mode on the lane-per-read kernel (detectBarcode / moveAndTrimPair / cleanBarcodeTail incl. mates that read
    into each other's barcode), at 137, 267 (BASELINE config 5: the 18-word variant), 256 and ragged lengths"""
    import cases
    d = synth.make_pairs(5000 if L < 200 else 2500, L, seed=515 + L, dirty=True, ragged=ragged)
    d = synth.add_barcodes(d, 516, tail_frac=0.25)
    batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    lane, tile, x, y = d["meta"]
    which = tile % len(cases.CIRCLES)
    lane = np.array([c[3] for c in cases.CIRCLES])[which]
    tile = np.array([c[4] for c in cases.CIRCLES])[which]
    ok = (x % 7 != 0).astype(np.uint8)
    batch.set_aux(lane, tile, x, y, ok)
    cfg = default_cfg(True, barcode=1, debubble=1, trim_tail=3 if L == 239 else 0, trim_tail2=2 if L == 239 else 0, seq_len_req=20 if ragged else 35)
    # (ragged: reads that are shorter than 5 bases once the barcode is gone make statRead raise upstream, so no QC there)
    g, o = run_both(gpu_engine, cfg, batch, circles=cases.CIRCLES, qc=not ragged)
    assert_same(g, o)
    flags = np.bincount(g["res"]["flag"], minlength=12)
    assert flags[capi.BADBCD1] > 0 and flags[capi.BADBCD2] > 0 and flags[capi.BADBBL] > 0
    # the lane-per-read kernel decided (nearly) all of them, and barcode tails were cut
    assert gpu_engine.last_deferred(0) < 0.05 * batch.n
    r = g["res"]
    moved = (r["flag"] != capi.BADBCD1) & (r["flag"] != capi.BADBCD2)
    cut = d["len1"].astype(np.int64) - r["start1"] - r["len1"]
    assert (cut[moved & (r["flag"] != capi.BADLEN)] > 4).sum() > 50 or cfg.trim_tail
    # single-end barcode
    b1 = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"])
    g, o = run_both(gpu_engine, default_cfg(False, barcode=1), b1)
    assert_same(g, o)


def test_accum_limit(gpu_engine):
    d = synth.make_pairs(3000, 150, seed=31, dirty=True)
    batch = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    g, o = run_both(gpu_engine, default_cfg(True), batch, qc=False, accum_limit=1234)
    assert_same(g, o)
    assert int(g["counters"][capi.C_TOTAL_READS]) == 1234


def test_zero_copy_text_arena(gpu_engine, tmp_path):
    """Batches that address the raw FASTQ text in place (unaligned offsets, separate quality offsets)."""
    import os
    from afterqc_amd import fastq
    d = synth.make_pairs(3000, 150, seed=77, ragged=True, dirty=True)
    lane, tile, x, y = d["meta"]
    p1, p2 = str(tmp_path / "a_R1.fq"), str(tmp_path / "a_R2.fq")
    synth.write_fastq(p1, synth.render_names(lane, tile, x, y, 1), d["seq1"], d["qual1"], d["len1"])
    synth.write_fastq(p2, synth.render_names(lane, tile, x, y, 2), d["seq2"], d["qual2"], d["len2"])
    rb1 = fastq.Reader(p1).next_batch(10000)
    rb2 = fastq.Reader(p2).next_batch(10000)
    batch = capi.Batch.from_raw(rb1, rb2)
    g, o = run_both(gpu_engine, default_cfg(True, seq_len_req=20), batch)
    assert_same(g, o)
    packed = capi.Batch.from_matrices(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    g2, _ = run_both(gpu_engine, default_cfg(True, seq_len_req=20), packed)
    assert g2["res"].tobytes() == g["res"].tobytes()


def test_lowcomplexity_vs_oracle(gpu_engine, tmp_path):
    """Many accepted diagonals, tail-anchored walk disagreeing with the scan -> BADMISMATCH (App. B-6)."""
    import cases
    from afterqc_amd import fastq
    cases.write_lowcomplex(dict(seed=909, n=4000), str(tmp_path))
    rb1 = fastq.Reader(str(tmp_path / "R1.fq")).next_batch(10000)
    rb2 = fastq.Reader(str(tmp_path / "R2.fq")).next_batch(10000)
    batch = capi.Batch.from_raw(rb1, rb2)
    for kw in (dict(), dict(mask_mismatch=1)):
        g, o = run_both(gpu_engine, default_cfg(True, poly_size_limit=0, seq_len_req=10, **kw), batch)
        assert_same(g, o)
        assert np.bincount(g["res"]["flag"], minlength=12)[capi.BADMISMATCH] > 0
        # the lane-per-read kernel walks the shifted diagonal itself (round 3): these pairs are no longer handed to the
        # general kernel
        n_def = gpu_engine.last_deferred(0)
        assert n_def < 0.05 * batch.n, (n_def, batch.n)       # (what is left are adapter cuts that need a second scan)


def test_errors_are_loud(gpu_engine):
    cfg = default_cfg(True)
    gpu_engine.set_config(cfg)
    long_read = "A" * 1200
    b = capi.Batch.from_strings([long_read], ["I" * 1200], ["ACGT" * 10], ["I" * 40])
    gpu_engine.upload(0, b)
    gpu_engine.run(0)
    with pytest.raises(capi.AqcError):
        gpu_engine.fetch_results(0)
    bad = capi.Config()
    bad.qc_kmer = 12
    with pytest.raises(capi.AqcError):
        gpu_engine.set_config(bad)


def test_empty_inputs(gpu_engine):
    """zero records through every entry point of the path: nothing is counted, nothing is written, nothing fails"""
    cfg = default_cfg(True)
    gpu_engine.set_config(cfg)
    gpu_engine.reset_stats()
    empty = np.zeros((0, 150), dtype=np.uint8)
    lens = np.zeros(0, dtype=np.uint32)
    b = capi.Batch.from_matrices(empty, empty, lens, empty, empty, lens)
    gpu_engine.upload(0, b)
    gpu_engine.run(0)
    assert len(gpu_engine.fetch_results(0)) == 0
    assert not gpu_engine.counters().any()
    text = np.zeros(64, dtype=np.uint8)
    info = gpu_engine.frame(0, text, 0, True, text, 0, True)
    assert int(info.n) == 0 and int(info.avail1) == 0 and not info.eof1
    gpu_engine.run(0)
    assert gpu_engine.format(0, 0, True) == [0] * 6
    assert not gpu_engine.counters().any()


def test_qc_stat_from_two_threads(gpu_engine):
    """aqc_qc_stat for two slots of one context from two threads at once (the CLI samples both mates of a pair side by side):
    the calls share one slice buffer and one stream inside, so they must queue up — accumulators and k-mer dictionaries equal a
    serial run's, call by call interleaved as it comes"""
    import threading
    eng = gpu_engine
    eng.set_config(default_cfg(False))
    d1 = synth.make_pairs(6000, 150, seed=901, dirty=True)
    d2 = synth.make_pairs(6000, 150, seed=902, dirty=True)
    b = [capi.Batch.from_matrices(d1["seq1"], d1["qual1"], d1["len1"]), capi.Batch.from_matrices(d2["seq1"], d2["qual1"], d2["len1"])]
    which = [capi.QC_R1_PRE, capi.QC_R2_PRE]
    pieces = [(k * 250, 250) for k in range(24)]

    def snapshot(w):
        # the dictionary comes back in table order (whichever wave inserted first): compare it sorted by key
        keys, counts, order = eng.kmers(w)
        rank = np.argsort(keys, kind="stable")
        return eng.qc(w).copy(), [keys[rank].copy(), counts[rank].copy(), order[rank].copy()]

    def serial():
        eng.reset_stats()
        for s in (0, 1):
            eng.upload(s, b[s])
            for first, count in pieces:
                eng.qc_stat(s, which[s], 0, first, count, 0)
            eng.sync(s)
        return [snapshot(w) for w in which]

    def threaded():
        eng.reset_stats()
        errs = []

        def work(s):
            try:
                eng.upload(s, b[s])
                for first, count in pieces:
                    eng.qc_stat(s, which[s], 0, first, count, 0)
                eng.sync(s)
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(s,)) for s in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        return [snapshot(w) for w in which]

    ref = serial()
    for _ in range(3):
        got = threaded()
        for (acc_r, km_r), (acc_g, km_g) in zip(ref, got):
            assert np.array_equal(acc_r, acc_g)
            for a, c in zip(km_r, km_g):
                assert np.array_equal(a, c)
