"""Randomised parity of the whole text path (-m gpu): for random option sets and dirty / ragged synthetic pairs the
device pipeline aqc_frame -> aqc_run -> aqc_qc_stat -> aqc_format must reproduce, byte for byte, what the oracle
engine (scalar restatement of fastq.Reader / the per-read loop / writeReads) makes of the same text chunks:
verdict records, counters, histograms, QC accumulators, k-mer dictionaries and the six output streams."""
import numpy as np
import pytest

from afterqc_amd import capi, synth

pytestmark = pytest.mark.gpu


def render(d, mate, rng, crlf):
    n = len(d["len" + mate])
    eol = "\r\n" if crlf else "\n"
    out = []
    for i in range(n):
        l = int(d["len" + mate][i])
        name = "@SIM:1:FC1:%d:%d:%d:%d %s:N:0:ACGT" % (1 + i % 3, 1101 + i % 7, 1000 + 7 * i, 2000 + 3 * i, mate)
        plus = "+" if i % 5 else "+" + name[1:]
        out.append(name + eol + d["seq" + mate][i, :l].tobytes().decode("latin-1") + eol + plus + eol +
                   d["qual" + mate][i, :l].tobytes().decode("latin-1") + eol)
    return "".join(out).encode("latin-1")


def random_cfg(rng, paired):
    cfg = capi.Config()
    cfg.paired = 1 if paired else 0
    cfg.trim_front, cfg.trim_tail = int(rng.integers(0, 6)), int(rng.integers(0, 6))
    cfg.trim_front2, cfg.trim_tail2 = int(rng.integers(0, 6)), int(rng.integers(0, 6))
    cfg.seq_len_req = int(rng.choice([0, 20, 35, 60]))
    cfg.poly_size_limit = int(rng.choice([0, 20, 35]))
    cfg.allow_mismatch_in_poly = int(rng.integers(0, 4))
    cfg.qualified_quality_phred = int(rng.choice([5, 15, 20, 30]))
    cfg.unqualified_base_limit = int(rng.choice([0, 20, 60]))
    cfg.n_base_limit = int(rng.choice([0, 1, 5]))
    cfg.no_overlap = int(rng.random() < 0.15)
    cfg.no_correction = int(rng.random() < 0.3)
    cfg.mask_mismatch = int(rng.random() < 0.3)
    cfg.count_r2_bases = int(rng.random() < 0.5)
    cfg.barcode_length = 12
    cfg.set_verify("CAGTA")
    cfg.qc_kmer = int(rng.choice([4, 6, 8]))
    return cfg


def pad(data):
    a = np.zeros(len(data) + 64, dtype=np.uint8)
    a[:len(data)] = np.frombuffer(data, dtype=np.uint8)
    return a


def run_text(eng, cfg, t1, t2, chunk, store_overlap=False):
    """feed the texts in chunks (lock step, carry-over) exactly like preprocesser._run_text; collect everything"""
    eng.set_config(cfg)
    eng.set_circles([])
    eng.reset_stats()
    paired = t2 is not None
    texts = [t1] + ([t2] if paired else [])
    pos = [0] * len(texts)
    left = [b""] * len(texts)
    done = [False] * len(texts)
    streams = [b""] * 6
    results = []
    total = 0
    while True:
        bufs, finals = [], []
        for k, t in enumerate(texts):
            new = b"" if done[k] else t[pos[k]:pos[k] + chunk]
            pos[k] += len(new)
            finals.append(done[k] or pos[k] >= len(t))
            bufs.append(left[k] + new)
        if paired:
            info = eng.frame(0, pad(bufs[0]), len(bufs[0]), finals[0], pad(bufs[1]), len(bufs[1]), finals[1], first_index=total)
        else:
            info = eng.frame(0, pad(bufs[0]), len(bufs[0]), finals[0], first_index=total)
        n = int(info.n)
        if n:
            eng.run(0)
            eng.qc_stat(0, capi.QC_R1_POST, 0, 0, n, 1)
            if paired:
                eng.qc_stat(0, capi.QC_R2_POST, 1, 0, n, 1)
            results.append(eng.fetch_results(0)[:n].copy())
            sizes = eng.format(0, n, store_overlap)
            for q, nb in enumerate(sizes):
                if nb:
                    out = np.zeros(nb, dtype=np.uint8)
                    eng.fetch_text(0, q // 3, q % 3, out, nb)
                    streams[q] += out.tobytes()
            total += n
        done1 = (info.eof1 or finals[0]) and info.avail1 == n
        done2 = paired and (info.eof2 or finals[1]) and info.avail2 == n
        if done1 or (done2 and info.avail1 > n):
            break
        cons = [int(info.consumed1), int(info.consumed2)]
        for k in range(len(texts)):
            left[k] = bufs[k][cons[k]:]
            if (info.eof1, info.eof2)[k] or (k == 1 and done2):
                done[k], left[k] = True, b""
    kms = []
    for w in ((capi.QC_R1_POST, capi.QC_R2_POST) if paired else (capi.QC_R1_POST,)):
        keys, counts, order = eng.kmers(w)
        idx = np.argsort(order, kind="stable")
        kms.append((keys[idx].tolist(), counts[idx].tolist()))
    res = np.concatenate(results) if results else np.zeros(0, dtype=capi.RESULT_DTYPE)
    return dict(res=res, streams=streams, counters=eng.counters().tolist(), hist=[h.tolist() for h in eng.histograms()],
                qc=[eng.qc(w).tolist() for w in (capi.QC_R1_POST, capi.QC_R2_POST)], kmers=kms)


@pytest.mark.parametrize("seed", range(12))
def test_text_path_random_options(gpu_engine, seed):
    from oracle import oracle
    rng = np.random.default_rng(1000 + seed)
    paired = seed % 4 != 3
    L = int(rng.choice([100, 150, 250]))
    n = 1500
    d = synth.make_pairs(n=n, L=L, seed=500 + seed, dirty=True, ragged=bool(seed % 2), short_frac=0.1 if seed % 3 == 0 else 0.03,
                         lowercase=0.05 if seed % 5 == 0 else 0.0)
    crlf = seed % 6 == 0
    t1 = render(d, "1", rng, crlf)
    t2 = render(d, "2", rng, crlf) if paired else None
    cfg = random_cfg(rng, paired)
    chunk = int(rng.choice([7000, 50_000, 10_000_000]))
    store = bool(seed % 2)
    g = run_text(gpu_engine, cfg, t1, t2, chunk, store)
    o = run_text(oracle.OracleEngine(), cfg, t1, t2, chunk, store)
    assert len(g["res"]) == len(o["res"]) == n
    assert np.array_equal(g["res"].view(np.uint8), o["res"].view(np.uint8))
    for q in range(6):
        assert g["streams"][q] == o["streams"][q], "stream %d" % q
    if store and paired and not cfg.no_overlap:
        assert len(o["streams"][2]) > 0 and len(o["streams"][5]) > 0
    assert g["counters"] == o["counters"]
    assert g["hist"] == o["hist"]
    assert g["qc"] == o["qc"]
    assert g["kmers"] == o["kmers"]


@pytest.mark.parametrize("paired", [True, False])
def test_whole_record_windows_around_the_32_window_limit(gpu_engine, paired):
    """the whole-record copy cuts a record into a head window + 16-byte windows on the SOURCE's 16-byte grid when that takes at most 32
    of them (records of <= 496 bytes at any alignment), plain windows otherwise (497..512 bytes off the grid), and leaves longer records
    to the general kernel: records of 470..540 bytes at every source alignment, untrimmed and good (corrections of the overlap walk
    included: their byte patches land in windows of both kinds), byte for byte against the oracle"""
    from oracle import oracle
    rng = np.random.default_rng(77)
    n = 1200
    d = synth.make_pairs(n=n, L=250, seed=4242, dirty=False)
    for m in ("1", "2"):
        d["len" + m] = np.minimum(d["len" + m], 210 + (np.arange(n) * 7 + (3 if m == "2" else 0)) % 36).astype(d["len" + m].dtype)
    t1 = render(d, "1", rng, False)
    t2 = render(d, "2", rng, False) if paired else None
    sizes = np.diff([0] + [i + 1 for i, c in enumerate(t1) if c == 10][3::4])
    assert sizes.min() < 480 and ((sizes > 496) & (sizes <= 512)).any() and sizes.max() > 512
    cfg = capi.Config()
    cfg.paired = 1 if paired else 0
    cfg.qualified_quality_phred = 15
    cfg.qc_kmer = 8
    cfg.barcode_length = 12
    cfg.set_verify("CAGTA")
    g = run_text(gpu_engine, cfg, t1, t2, 10_000_000, False)
    o = run_text(oracle.OracleEngine(), cfg, t1, t2, 10_000_000, False)
    assert np.array_equal(g["res"].view(np.uint8), o["res"].view(np.uint8))
    for q in range(6):
        assert g["streams"][q] == o["streams"][q], "stream %d" % q
    assert len(g["streams"][0]) > 0.9 * len(t1)          # (nearly everything is good and goes out as its own bytes)
    assert g["counters"] == o["counters"]
