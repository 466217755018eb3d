"""-m gpu: AQC_FUSED=1 — the verdict kernel places every record of a device-framed 2 x <= 160 chunk in its output stream itself
(sizes scanned inside the batch, decoupled look-back over the batches) and copies the good records that go out as their own
bytes; aqc_format then only rebuilds the rest (aqc_fast.hpp FUSE).  Whatever the options, the six streams must be
byte for byte what a context WITHOUT the variant hands out (itself pinned to the reference by the golden cases: seqFilter.writeReads,
preprocesser.py:206-232) — for untrimmed runs (nearly every good record is copied by the verdict kernel), trimmed runs (none is),
corrections and masks, strict filters (many bad records), ragged lengths, names of every length, chunks of one record and of
300 000 (the look-back crosses thousands of batches).  Chunks the variant cannot place (CR LF lines, a pair the general kernel
has to finish, irregular records) must fall back to the other writer — same bytes again."""
import os

import numpy as np
import pytest

from afterqc_amd import capi, synth
from test_gpu_spans import cfg_default, fetch, pad, texts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fused_engine():
    old = os.environ.get("AQC_FUSED")
    os.environ["AQC_FUSED"] = "1"            # read by aqc_create
    try:
        eng = capi.Engine(0, 2)
    finally:
        if old is None:
            os.environ.pop("AQC_FUSED", None)
        else:
            os.environ["AQC_FUSED"] = old
    yield eng
    eng.close()


def six_streams(eng, t1, t2, cfg, store=False):
    eng.set_config(cfg)
    eng.reset_stats()
    info = eng.frame(0, pad(t1), len(t1), True, pad(t2), len(t2), True)
    eng.run(0)
    sizes = eng.format(0, int(info.n), store)
    out = [fetch(eng, 0, sizes, q) for q in range(6)]
    counters = np.array(eng.counters()).copy()
    deferred = eng.last_deferred(0)
    eng.reset_stats()
    return int(info.n), out, counters, deferred


CASES = {
    # name: (options, texts() arguments, n, must the fused placement be taken?)
    "untrimmed": (dict(), dict(), 5000, True),
    "untrimmed_odd_count": (dict(), dict(), 4999, True),
    "one_batch": (dict(), dict(), 31, True),
    "one_record": (dict(), dict(), 1, True),
    "trimmed": (dict(trim_front=3, trim_tail=2, trim_front2=3, trim_tail2=2), dict(), 5000, True),
    "mask": (dict(mask_mismatch=1), dict(), 5000, True),
    "no_correction": (dict(no_correction=1), dict(), 3000, True),
    "strict_quality": (dict(qualified_quality_phred=36, unqualified_base_limit=20), dict(), 5000, True),
    "ragged": (dict(), dict(ragged=True), 5000, None),          # (reads under 5 / 31 bases defer: either way is right)
    "crlf_lines": (dict(), dict(crlf_every=5), 2000, False),
    "no_quality_filter": (dict(unqualified_base_limit=0), dict(), 2000, True),
    "exotic_base": (dict(), dict(), 3000, False),               # (one '.' among the bases: that pair is the general kernel's, the chunk the other writer's)
}


@pytest.mark.parametrize("case", list(CASES))
def test_fused_placement_equals_the_two_step_writer(case, gpu_engine, fused_engine):
    opts, tk, n, must = CASES[case]
    t1, t2 = texts(n, 7300 + len(case), **tk)
    if case == "exotic_base":
        rec = t1.index(b"\n@SIM", len(t1) // 2) + 1                            # a record in the middle of file 1 ...
        at = t1.index(b"\n", rec) + 21                                         # ... base 20 of its sequence line
        t1 = t1[:at] + b"." + t1[at + 1:]
    n_a, want, cnt_a, _ = six_streams(gpu_engine, t1, t2, cfg_default(**opts))
    assert not gpu_engine.format_fused(0)
    n_b, got, cnt_b, deferred = six_streams(fused_engine, t1, t2, cfg_default(**opts))
    assert n_a == n_b == n
    if must is not None:
        # (a chunk with a pair the general kernel has to finish is not placed by the verdict kernel)
        assert fused_engine.format_fused(0) == (must and deferred == 0), (case, deferred)
    if case == "exotic_base":
        assert deferred >= 1
    for q in range(6):
        assert got[q] == want[q], (case, q, len(got[q]), len(want[q]))
    assert (cnt_a == cnt_b).all()
    if case == "untrimmed":
        assert len(want[0]) > 0.5 * len(t1)


def test_fused_long_names_and_many_batches(gpu_engine, fused_engine):
    """300 000 pairs (9 375 batches: the look-back runs over hundreds of windows) with names of 20 .. 120 bytes — records of up to
    ~430 bytes take the 64-window copy pass"""
    n = 300_000
    d = synth.make_pairs(n, 150, seed=7411, workers=1)
    rng = np.random.default_rng(7)
    extra = rng.integers(0, 100, size=n)
    out = []
    for mate, (sq, ql) in enumerate(((d["seq1"], d["qual1"]), (d["seq2"], d["qual2"])), 1):
        recs = []
        for i in range(n):
            recs.append(b"@S:%d:%d %d:N:0:" % (i, 3 * i, mate) + b"A" * int(extra[i]) + b"\n" + sq[i].tobytes() + b"\n+\n" + ql[i].tobytes() + b"\n")
        out.append(b"".join(recs))
    t1, t2 = out
    _, want, cnt_a, _ = six_streams(gpu_engine, t1, t2, cfg_default())
    _, got, cnt_b, deferred = six_streams(fused_engine, t1, t2, cfg_default())
    assert fused_engine.format_fused(0) == (deferred == 0)
    for q in range(6):
        assert got[q] == want[q], q
    assert (cnt_a == cnt_b).all()
    # ... and the spans assembly of the same chunk (events crossing the same 9 375 batches)
    eng = gpu_engine
    eng.set_config(cfg_default())
    eng.reset_stats()
    info = eng.frame(0, pad(t1), len(t1), True, pad(t2), len(t2), True)
    eng.run(0)
    sizes, n_ev = eng.format_spans(0, n, False)
    sp = [fetch(eng, 0, sizes, q) for q in range(6)]
    end = eng.span_end(0, n)
    spans_good = [capi.assemble_spans(chunk, end[f], eng.fetch_span_events(0, f, n_ev[f]), sp[3 * f]) for f, chunk in enumerate((t1, t2))]
    eng.reset_stats()
    assert spans_good[0] == want[0] and spans_good[1] == want[3] and sp[1] == want[1] and sp[4] == want[4]
    # All three writers against the ORACLE's streams, not only against each other (round-5 review: the look-back over thousands of
    # batches had never been checked against anything but the other HIP writer): the scalar C restatement of the reference's
    # loop + the reference's writeReads (oracle/oracle.py), over the very same 300 000 pairs
    from oracle import oracle
    oe = oracle.OracleEngine()
    oe.set_config(cfg_default())
    oe.reset_stats()
    oinfo = oe.frame(0, pad(t1), len(t1), True, pad(t2), len(t2), True)
    assert int(oinfo.n) == n
    oe.run(0)
    osizes = oe.format(0, n, False)
    for q in range(6):
        obuf = np.zeros(osizes[q] + 64, dtype=np.uint8)
        if osizes[q]:
            oe.fetch_text(0, q // 3, q % 3, obuf, obuf.size)
        assert obuf[:osizes[q]].tobytes() == want[q], ("oracle", q)
    assert (np.array(oe.counters()) == cnt_a).all()
    oe.close()


def test_fused_then_partial_and_span_formats(gpu_engine, fused_engine):
    """a format of fewer records, a spans format or an overlap-store format after a fused run go through the other writer and leave
    the right bytes; formatting all records again afterwards still does"""
    n = 4000
    t1, t2 = texts(n, 7555)
    cfg = cfg_default()
    eng = fused_engine
    eng.set_config(cfg)
    eng.reset_stats()
    eng.frame(0, pad(t1), len(t1), True, pad(t2), len(t2), True)
    eng.run(0)
    ref = gpu_engine
    ref.set_config(cfg)
    ref.reset_stats()
    ref.frame(0, pad(t1), len(t1), True, pad(t2), len(t2), True)
    ref.run(0)
    for m, store in ((n, False), (n - 7, False), (n, False), (n, True), (n, False)):
        sz_a, sz_b = ref.format(0, m, store), eng.format(0, m, store)
        assert sz_a == sz_b
        for q in range(6):
            assert fetch(eng, 0, sz_b, q) == fetch(ref, 0, sz_a, q), (m, store, q)
    eng.reset_stats()
    ref.reset_stats()


def test_fused_contexts_sharing_one_device_through_the_pipe(tmp_path):
    """four contexts on ONE device, all created with AQC_FUSED=1, over one 600 k-pair input: their verdict kernels — persistent grids
    with in-order look-backs — compete for the same CUs (a workgroup of one only starts where a workgroup of another has finished),
    the chunks are 20 000 records: byte-identical files and an identical statistics JSON to the plain one-context run"""
    from test_gpu_pipe import run
    work = str(tmp_path)
    n = 600_000
    d = synth.make_pairs(n, 150, seed=7813, workers=4)
    r1, r2 = os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq")
    synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
    synth.write_fastq_fixed(r2, d["seq2"], d["qual2"], 2)
    del d
    extra = ["-f", "0", "-t", "0"]
    a_files, a_stat, a = run(work, r1, r2, extra, tag="plain", use_pipe=True, devices=[0], chunk_records=1 << 15)
    old = os.environ.get("AQC_FUSED")
    os.environ["AQC_FUSED"] = "1"
    try:
        b_files, b_stat, b = run(work, r1, r2, extra, tag="fused4", use_pipe=True, devices=[0] * 4, chunk_records=20000, pipe_slots=3)
    finally:
        if old is None:
            os.environ.pop("AQC_FUSED", None)
        else:
            os.environ["AQC_FUSED"] = old
    assert a.used_pipe and b.used_pipe
    assert a_files == b_files
    assert a_stat == b_stat
    assert a_stat["afterqc_main_summary"]["total_reads"] == n
    chunks, fused_chunks = b.timing["pipe_chunks"]
    assert chunks >= n // 20000 and fused_chunks == chunks, (chunks, fused_chunks)       # (no pair of this input is deferred)
    assert a.timing["pipe_chunks"][1] == 0
