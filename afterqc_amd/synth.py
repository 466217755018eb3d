"""Seeded synthetic FASTQ workloads.

Everything here is numpy-vectorised so that the 10 M-read bench workload is
generated in well under a minute.  The same generators feed
  * tests/golden/make_golden.py  (inputs handed to the imported reference),
  * the parity tests             (same seeds -> same bytes),
  * bench.py                     (SoA arrays straight to the device).

Reads are produced as fixed-width uint8 matrices (n x L) plus a length vector,
which is exactly the padded form the SoA packer wants; `write_fastq` renders
them as the 4-line text records the reference's fastq.Reader consumes
(fastq.py:37-49).
"""
import gzip
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP_IDX = np.array([3, 2, 1, 0], dtype=np.uint8)  # A<->T, C<->G on indices into BASES
# quality alphabet / weights measured on the reference's testdata
QUAL_CHARS = np.frombuffer(b"#/6<AE", dtype=np.uint8)
QUAL_WEIGHTS = np.array([0.01, 0.09, 0.02, 0.06, 0.18, 0.64])
ADAPTER1 = np.frombuffer(b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", dtype=np.uint8)
ADAPTER2 = np.frombuffer(b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT", dtype=np.uint8)
# 256-entry lookup: a uniform byte -> quality character with the testdata weights (1/256 resolution)
_QUAL_LUT = np.repeat(QUAL_CHARS, [3, 23, 5, 15, 46, 164]).astype(np.uint8)


def _quals(rng, shape):
    return _QUAL_LUT[rng.integers(0, 256, shape, dtype=np.uint8)]


def make_names(rng, n, mate=1, start=0):
    """@SIM:1:FC1:<lane 1-4>:<tile 1101-2316>:<x 1000-25000>:<y 1000-20000> <mate>:N:0:ACGT"""
    lane = rng.integers(1, 5, n)
    tile = rng.integers(1101, 2317, n)
    x = rng.integers(1000, 25001, n)
    y = rng.integers(1000, 20001, n)
    return lane, tile, x, y


def render_names(lane, tile, x, y, mate):
    return ["@SIM:1:FC1:%d:%d:%d:%d %d:N:0:ACGT" % (lane[i], tile[i], x[i], y[i], mate)
            for i in range(len(lane))]


def _single_chunk(args):
    (n, L, seedseq) = args
    rng = np.random.Generator(np.random.PCG64(seedseq))
    seq = BASES[rng.integers(0, 4, (n, L), dtype=np.uint8)]
    qual = _QUAL_LUT[rng.integers(0, 256, (n, L), dtype=np.uint8)]
    # per-base N
    nmask = rng.integers(0, 65536, (n, L), dtype=np.uint16) < 131
    # 0.5 % of reads with 8-20 N
    for r in np.nonzero(rng.random(n) < 0.005)[0]:
        nmask[r, rng.choice(L, int(rng.integers(8, 21)), replace=False)] = True
    seq[nmask] = ord("N")
    qual[nmask] = ord("#")
    # 1 % poly-G / poly-A tails of 35-80
    for r in np.nonzero(rng.random(n) < 0.01)[0]:
        k = int(rng.integers(35, min(81, L)))
        seq[r, L - k:] = ord("G") if rng.random() < 0.7 else ord("A")
    # 2 % of reads with 61-100 low quality bases
    for r in np.nonzero(rng.random(n) < 0.02)[0]:
        k = int(rng.integers(min(61, L // 2), min(101, L)))
        qual[r, rng.choice(L, k, replace=False)] = ord("#")
    return seq, qual, make_names(rng, n)


def make_single(n, L=150, seed=1002, chunk=250_000, workers=None):
    """Config 2: single-end reads with N bursts, poly-X tails and low-quality stretches
    (independently seeded chunks, generated in parallel like make_pairs)."""
    sizes = [min(chunk, n - a) for a in range(0, n, chunk)]
    seeds = np.random.SeedSequence(seed).spawn(len(sizes))
    jobs = [(m, L, sq) for m, sq in zip(sizes, seeds)]
    if len(jobs) > 1 and (workers is None or workers > 1):
        import concurrent.futures as cf
        import multiprocessing as mp
        import os
        nw = min(len(jobs), workers or os.cpu_count() or 1)
        with cf.ProcessPoolExecutor(max_workers=nw, mp_context=mp.get_context("fork")) as ex:
            parts = list(ex.map(_single_chunk, jobs))
    else:
        parts = [_single_chunk(jb) for jb in jobs]
    seq = parts[0][0] if len(parts) == 1 else np.concatenate([p[0] for p in parts])
    qual = parts[0][1] if len(parts) == 1 else np.concatenate([p[1] for p in parts])
    meta = tuple(np.concatenate([p[2][i] for p in parts]) for i in range(4))
    return dict(seq1=seq, qual1=qual, len1=np.full(n, L, dtype=np.uint32), meta=meta)


def _pairs_chunk(args):
    """One independently seeded chunk of make_pairs (so chunks can be generated in parallel)."""
    (m, L, seedseq, ragged, lowercase, dirty, short_frac) = args
    rng = np.random.Generator(np.random.PCG64(seedseq))
    T = 2 * L
    j = np.arange(L, dtype=np.int32)[None, :]
    ov = np.clip(np.rint(rng.normal(30.0, 8.0, m)), 0, L).astype(np.int32)
    ins = (2 * L - ov).astype(np.int32)
    short = rng.random(m) < short_frac
    ins[short] = rng.integers(60, L, int(short.sum()))
    frag = rng.integers(0, 4, (m, T), dtype=np.uint8)          # base codes 0..3 = A,C,G,T
    rows = np.arange(m, dtype=np.int32)[:, None]
    past = j - ins[:, None]                                      # >= 0: past the insert end
    filler = frag[:, ::-1][:, :L]                                # pseudo-random tail behind the adapter
    # read 1 = fragment[0:L]; adapter1 + filler past the insert end
    c1 = frag[:, :L].copy()
    # read 2 = revcomp(fragment[0:ins])[0:L]; adapter2 + filler past the insert end
    idx = ins[:, None] - 1 - j
    c2 = (3 - frag[rows, np.clip(idx, 0, T - 1)]).astype(np.uint8)
    rt = np.flatnonzero(ins < L)                                 # rows with read-through
    if len(rt):
        pr = past[rt]
        for c, ad in ((c1, ADAPTER1), (c2, ADAPTER2)):
            adc = np.searchsorted(BASES, ad).astype(np.uint8)
            sub = np.where(pr < len(adc), adc[np.clip(pr, 0, len(adc) - 1)], filler[rt])
            c[rt] = np.where(pr >= 0, sub, c[rt])
    reads = []
    for c in (c1, c2):
        u = rng.integers(0, 65536, (m, L), dtype=np.uint16)      # one draw decides error / N per base
        v = rng.integers(0, 256, (m, L), dtype=np.uint8)         # substitution offset + low-quality flavour
        w = rng.integers(0, 256, (m, L), dtype=np.uint8)         # quality
        q = _QUAL_LUT[w]
        err_lo = u < 328                                         # 0.5 %
        err_hi = (u >= 328) & (u < 393)                          # 0.1 %
        nm = (u >= 393) & (u < 524)                              # 0.2 %
        err = err_lo | err_hi
        c[err] = (c[err] + (v[err] % 3) + 1) & 3
        r = BASES[c]
        q[err_lo] = np.where(v[err_lo] & 128, ord("/"), ord("#")).astype(np.uint8)
        q[err_hi] = ord("E")
        r[nm] = ord("N")
        q[nm] = ord("#")
        reads.append((r, q))
    (r1, q1), (r2, q2) = reads
    if dirty:
        for r, q in ((r1, q1), (r2, q2)):
            for row in np.nonzero(rng.random(m) < 0.01)[0]:
                k = int(rng.integers(30, min(81, L)))
                r[row, L - k:] = ord("G") if rng.random() < 0.7 else ord("A")
            for row in np.nonzero(rng.random(m) < 0.02)[0]:
                k = int(rng.integers(min(55, L // 2), min(101, L)))
                q[row, rng.choice(L, k, replace=False)] = ord("#")
            for row in np.nonzero(rng.random(m) < 0.01)[0]:
                k = int(rng.integers(3, 12))
                r[row, rng.choice(L, k, replace=False)] = ord("N")
    if lowercase > 0:
        lc = rng.random(m) < lowercase
        r1[lc, :20] |= 0x20
        r2[lc, :20] |= 0x20
        # 'n' is not in the reference's COMP table; keep N upper case
        r1[r1 == ord("n")] = ord("N")
        r2[r2 == ord("n")] = ord("N")
    if ragged:
        l1 = rng.integers(20, L + 1, m).astype(np.uint32)
        l2 = rng.integers(20, L + 1, m).astype(np.uint32)
        keep = rng.random(m) < 0.5
        l1[keep] = L
        l2[keep] = L
    else:
        l1 = np.full(m, L, dtype=np.uint32)
        l2 = np.full(m, L, dtype=np.uint32)
    return r1, q1, l1, r2, q2, l2, make_names(rng, m)


def make_pairs(n, L=150, seed=1003, ragged=False, chunk=250_000, lowercase=0.0, dirty=False, short_frac=0.03,
               workers=None):
    """Config 3: paired-end reads, insert = 2L - ov with ov ~ round(N(30, 8)) in [0, L],
    3 % short inserts (60 .. L-1) with adapter read-through, 0.5 %/base substitution errors
    carrying low quality ('/' or '#') and 0.1 %/base carrying high quality, N at 0.2 %/base.

    Chunks of `chunk` pairs are seeded independently (SeedSequence(seed).spawn), so the result does
    not depend on `workers` (processes used to generate chunks in parallel; default: all cores when
    there is more than one chunk).

    ragged=True additionally truncates every read to a random length (tests only).
    lowercase>0 soft-masks that fraction of reads' first 20 bases to lower case (tests only).
    dirty=True adds config-2 style artefacts to both mates (poly-X tails, low-quality stretches,
    N bursts) so that every filter fires in the small golden cases (tests only).
    """
    sizes = [min(chunk, n - a) for a in range(0, n, chunk)]
    seeds = np.random.SeedSequence(seed).spawn(len(sizes))
    jobs = [(m, L, sq, ragged, lowercase, dirty, short_frac) for m, sq in zip(sizes, seeds)]
    if len(jobs) > 1 and (workers is None or workers > 1):
        import concurrent.futures as cf
        import multiprocessing as mp
        import os
        nw = min(len(jobs), workers or os.cpu_count() or 1)
        with cf.ProcessPoolExecutor(max_workers=nw, mp_context=mp.get_context("fork")) as ex:
            parts = list(ex.map(_pairs_chunk, jobs))
    else:
        parts = [_pairs_chunk(jb) for jb in jobs]
    keys = ("seq1", "qual1", "len1", "seq2", "qual2", "len2")
    res = {k: (parts[0][i] if len(parts) == 1 else np.concatenate([p[i] for p in parts])) for i, k in enumerate(keys)}
    res["meta"] = tuple(np.concatenate([p[6][i] for p in parts]) for i in range(4))
    return res


def add_barcodes(d, seed, barcode_len=12, verify=b"CAGTA", tail_frac=0.0):
    """Config 5 flavour: prepend <barcode><verify> to both mates; 10 % of R1 verify sequences get
    one mismatch, 3 % are destroyed (-> BADBCD1); a few R2 are destroyed too (-> BADBCD2).
    tail_frac > 0: that fraction of the pairs reads through into the mate's barcode (the last k bases of each mate are
    the reverse complement of the other mate's first k bases, k = 1 .. barcode + verify + 2, with an occasional
    substitution / deletion), which is what cleanBarcodeTail (barcodeprocesser.py:47-75) exists to cut."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n, L = d["seq1"].shape
    ver = np.frombuffer(verify, dtype=np.uint8)
    P = barcode_len + len(ver)
    for k, frac_bad in (("1", 0.03), ("2", 0.01)):
        pre = np.empty((n, P), dtype=np.uint8)
        pre[:, :barcode_len] = BASES[rng.integers(0, 4, (n, barcode_len), dtype=np.uint8)]
        pre[:, barcode_len:] = ver[None, :]
        one = np.nonzero(rng.random(n) < 0.10)[0]
        pos = rng.integers(0, len(ver), len(one))
        cur = pre[one, barcode_len + pos]
        pre[one, barcode_len + pos] = np.where(cur == ord("A"), ord("C"), ord("A"))
        bad = np.nonzero(rng.random(n) < frac_bad)[0]
        pre[bad, barcode_len:] = BASES[rng.integers(0, 4, (len(bad), len(ver)), dtype=np.uint8)]
        # some barcodes are one base short / long (verify found at 11 or 13)
        shift = rng.random(n)
        seq = np.concatenate([pre, d["seq" + k]], axis=1)
        qual = np.concatenate([_quals(rng, (n, P)), d["qual" + k]], axis=1)
        shorter = np.nonzero(shift < 0.02)[0]
        seq[shorter, :-1] = seq[shorter, 1:]
        longer = np.nonzero((shift >= 0.02) & (shift < 0.04))[0]
        seq[longer, 1:] = seq[longer, :-1]
        d["seq" + k] = seq
        d["qual" + k] = qual
        d["len" + k] = (d["len" + k] + P).astype(np.uint32)
    if tail_frac > 0:
        # (drawn after everything else so that tail_frac = 0 reproduces the older fixtures bit for bit)
        comp = np.zeros(256, dtype=np.uint8)
        comp[:] = ord("N")
        for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
            comp[a] = b
        s1, s2 = d["seq1"], d["seq2"]
        for r in np.nonzero(rng.random(n) < tail_frac)[0]:
            k = int(rng.integers(1, P + 3))
            l1, l2 = int(d["len1"][r]), int(d["len2"][r])
            if k + 2 >= min(l1, l2):
                continue
            t1 = comp[s2[r, :k]][::-1].copy()
            t2 = comp[s1[r, :k]][::-1].copy()
            for t in (t1, t2):
                u = rng.random()
                if u < 0.25:
                    t[int(rng.integers(0, k))] = BASES[int(rng.integers(0, 4))]
                elif u < 0.35 and k > 3:
                    j = int(rng.integers(0, k - 1))
                    t[j:-1] = t[j + 1:].copy()
            s1[r, l1 - k:l1] = t1
            s2[r, l2 - k:l2] = t2
    return d


def write_fastq(path, names, seq, qual, lens, plus="+"):
    """Render records as 4-line FASTQ text; a '.gz' / '.bz2' suffix selects gzip / bzip2 (fastq.py:23-28,63-76)."""
    import bz2
    opener = gzip.open if path.endswith(".gz") else bz2.open if path.endswith(".bz2") else open
    with opener(path, "wb") as f:
        buf = []
        for i in range(len(names)):
            l = int(lens[i])
            buf.append(names[i].encode() + b"\n" + seq[i, :l].tobytes() + b"\n" + plus.encode() + b"\n"
                       + qual[i, :l].tobytes() + b"\n")
            if len(buf) >= 4096:
                f.write(b"".join(buf)); buf = []
        f.write(b"".join(buf))


def fixed_record_width(L, tile=1101, mate=1):
    """bytes of one record as render_fastq_fixed / write_fastq_fixed lay it out"""
    return len("@SIM:1:FC1:1:%d:" % tile) + 7 + 1 + 5 + len(" %d:N:0:ACGT\n" % mate) + L + 1 + 2 + L + 1


def render_fastq_fixed(seq, qual, mate=1, tile=1101, out=None, index0=0):
    """The text write_fastq_fixed writes, rendered into memory (numpy uint8, e.g. a page-locked HostBuffer.array): returns
    (array, nbytes).  index0 = global index of record 0 (names carry it modulo 10^7)."""
    n, L = seq.shape
    w = fixed_record_width(L, tile, mate)
    if out is None:
        out = np.empty(n * w + 64, dtype=np.uint8)
    assert out.size >= n * w
    head = np.frombuffer(("@SIM:1:FC1:1:%d:" % tile).encode(), dtype=np.uint8)
    tail = np.frombuffer((" %d:N:0:ACGT\n" % mate).encode(), dtype=np.uint8)
    chunk = 1 << 18
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        m = out[a * w:b * w].reshape(b - a, w)
        idx = (np.arange(a, b, dtype=np.int64) + index0)
        c = 0
        m[:, c:c + len(head)] = head; c += len(head)
        x = idx % 10 ** 7
        for d in range(7):
            m[:, c + 6 - d] = (x // (10 ** d)) % 10 + 48
        c += 7
        m[:, c] = ord(":"); c += 1
        y = (idx * 7919) % 100000
        for d in range(5):
            m[:, c + 4 - d] = (y // (10 ** d)) % 10 + 48
        c += 5
        m[:, c:c + len(tail)] = tail; c += len(tail)
        m[:, c:c + L] = seq[a:b]; c += L
        m[:, c] = 10; c += 1
        m[:, c] = ord("+"); m[:, c + 1] = 10; c += 2
        m[:, c:c + L] = qual[a:b]; c += L
        m[:, c] = 10
    return out, n * w


def write_fastq_fixed(path, seq, qual, mate=1, tile=1101):
    """Fixed-length reads as FASTQ text, vectorised (benchmarks: millions of records).  Names follow SURVEY.md
    §8d: @SIM:1:FC1:1:<tile>:<x = 7-digit record index>:<y> <mate>:N:0:ACGT"""
    n, L = seq.shape
    head = ("@SIM:1:FC1:1:%d:" % tile).encode()
    tail = (" %d:N:0:ACGT\n" % mate).encode()
    idx = np.arange(n, dtype=np.int64)
    digits = np.empty((n, 7), dtype=np.uint8)
    for d in range(7):
        digits[:, 6 - d] = (idx // (10 ** d)) % 10 + 48
    y = (idx * 7919) % 100000
    ydig = np.empty((n, 5), dtype=np.uint8)
    for d in range(5):
        ydig[:, 4 - d] = (y // (10 ** d)) % 10 + 48
    w = len(head) + 7 + 1 + 5 + len(tail) + L + 1 + 2 + L + 1
    chunk = 1 << 18
    if path.endswith(".gz"):
        from .fastq import ParallelGzipFile
        f = ParallelGzipFile(path, 2)
    else:
        f = open(path, "wb")
    try:
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            m = np.empty((b - a, w), dtype=np.uint8)
            c = 0
            m[:, c:c + len(head)] = np.frombuffer(head, dtype=np.uint8); c += len(head)
            m[:, c:c + 7] = digits[a:b]; c += 7
            m[:, c] = ord(":"); c += 1
            m[:, c:c + 5] = ydig[a:b]; c += 5
            m[:, c:c + len(tail)] = np.frombuffer(tail, dtype=np.uint8); c += len(tail)
            m[:, c:c + L] = seq[a:b]; c += L
            m[:, c] = 10; c += 1
            m[:, c] = ord("+"); m[:, c + 1] = 10; c += 2
            m[:, c:c + L] = qual[a:b]; c += L
            m[:, c] = 10
            f.write(m.reshape(-1))
    finally:
        f.close()
