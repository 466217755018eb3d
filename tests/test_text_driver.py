"""Host-side streaming driver (preprocesser._run_text) corner cases on CPU (oracle engine injected): mate files of
unequal length, an empty line ending one file early, gzip in / multi-member gzip out — the text path must agree
with the host path (which the end-to-end goldens pin against the real reference) in every output byte and statistic."""
import gzip
import json
import os

import numpy as np
import pytest

from afterqc_amd import after, fastq, preprocesser, synth


def write_pair(work, n1, n2, cut2_at=None, gz=False, seed=77):
    d = synth.make_pairs(n=max(n1, n2), L=100, seed=seed, dirty=True, workers=1)
    ext = ".fq.gz" if gz else ".fq"
    p1, p2 = os.path.join(work, "R1" + ext), os.path.join(work, "R2" + ext)
    meta = d["meta"]
    names1 = synth.render_names(meta[0][:n1], meta[1][:n1], meta[2][:n1], meta[3][:n1], 1)
    names2 = synth.render_names(meta[0][:n2], meta[1][:n2], meta[2][:n2], meta[3][:n2], 2)
    synth.write_fastq(p1, names1, d["seq1"][:n1], d["qual1"][:n1], d["len1"][:n1])
    synth.write_fastq(p2, names2, d["seq2"][:n2], d["qual2"][:n2], d["len2"][:n2])
    if cut2_at is not None:
        op = gzip.open if gz else open
        with op(p2, "rb") as f:
            lines = f.read().split(b"\n")
        lines.insert(4 * cut2_at + 1, b"   ")          # a blank line inside record cut2_at: EOF for that reader
        with op(p2, "wb") as f:
            f.write(b"\n".join(lines))
    return p1, p2


def run(work, p1, p2, mode, sub, engine=None):
    from oracle import oracle
    out = os.path.join(work, sub)
    argv = ["-1", p1, "-2", p2, "-f", "0", "-t", "0", "-g", os.path.join(out, "good"), "-b", os.path.join(out, "bad"),
            "-r", os.path.join(out, "QC")]
    options, _ = after.parseCommand(argv)
    after.finalize_options(options)
    options.barcode = False
    kw = {"text": dict(use_text_path=True), "tiny": dict(use_text_path=True, chunk_bytes=900), "host": dict(use_text_path=False)}[mode]
    flt = preprocesser.seqFilter(options, engine=engine if engine is not None else oracle.OracleEngine(), **kw)
    stat = flt.run()
    assert flt.text_path == (mode != "host")
    files = {}
    for sub_dir in ("good", "bad"):
        for fn in sorted(os.listdir(os.path.join(out, sub_dir))):
            path = os.path.join(out, sub_dir, fn)
            op = gzip.open if fn.endswith(".gz") else open
            with op(path, "rb") as f:
                files[sub_dir + "/" + fn] = f.read()
    return stat, files


@pytest.mark.parametrize("shape", ["r2_short", "r1_short", "r2_blank_line", "gz"])
def test_text_driver_matches_host_driver(shape, tmp_path):
    work = str(tmp_path)
    if shape == "r2_short":
        p1, p2 = write_pair(work, 60, 53)
    elif shape == "r1_short":
        p1, p2 = write_pair(work, 41, 60)
    elif shape == "r2_blank_line":
        p1, p2 = write_pair(work, 60, 60, cut2_at=37)
    else:
        p1, p2 = write_pair(work, 500, 500, gz=True)
    ref_stat, ref_files = run(work, p1, p2, "host", "host")
    n_expected = {"r2_short": 53, "r1_short": 41, "r2_blank_line": 37, "gz": 500}[shape]
    assert ref_stat["afterqc_main_summary"]["total_reads"] == 2 * n_expected or ref_stat["afterqc_main_summary"]["total_reads"] == n_expected
    for mode in ("text", "tiny"):
        stat, files = run(work, p1, p2, mode, mode)
        for key in ref_stat:
            if key != "command":                 # (the option dump carries this run's output folders)
                assert json.dumps(stat[key], sort_keys=True) == json.dumps(ref_stat[key], sort_keys=True), (shape, mode, key)
        assert files.keys() == ref_files.keys()
        for k in files:
            assert files[k] == ref_files[k], (shape, mode, k)


def test_parallel_gzip_members_roundtrip(tmp_path):
    rng = np.random.default_rng(3)
    data = bytes(rng.integers(65, 90, 3_500_000, dtype=np.uint8))
    p = str(tmp_path / "x.fq.gz")
    w = fastq.Writer(p, gzip_compression=2)
    for a in range(0, len(data), 1_200_000):
        w.write_bytes(memoryview(data)[a:a + 1_200_000])
    w.close()
    with gzip.open(p, "rb") as f:
        assert f.read() == data
    # and the reader side takes multi-member files too
    r = fastq.open_binary(p)
    buf = bytearray(len(data) + 10)
    got = 0
    while True:
        k = r.readinto(memoryview(buf)[got:])
        if not k:
            break
        got += k
    assert bytes(buf[:got]) == data
    e = str(tmp_path / "empty.fq.gz")
    fastq.Writer(e).close()
    with gzip.open(e, "rb") as f:
        assert f.read() == b""


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["r2_short", "r1_short", "r2_blank_line"])
def test_text_driver_on_gpu(shape, tmp_path, gpu_engine):
    """the same corners with aqc_frame's lock-step bookkeeping (avail / consumed / eof / next_len1) coming from the device"""
    work = str(tmp_path)
    if shape == "r2_short":
        p1, p2 = write_pair(work, 60, 53)
    elif shape == "r1_short":
        p1, p2 = write_pair(work, 41, 60)
    else:
        p1, p2 = write_pair(work, 60, 60, cut2_at=37)
    ref_stat, ref_files = run(work, p1, p2, "host", "host")
    for mode in ("text", "tiny"):
        stat, files = run(work, p1, p2, mode, "gpu_" + mode, engine=gpu_engine)
        for key in ref_stat:
            if key != "command":
                assert json.dumps(stat[key], sort_keys=True) == json.dumps(ref_stat[key], sort_keys=True), (shape, mode, key)
        assert files == ref_files


def test_directory_mode_runs_pairs_concurrently(tmp_path):
    """after.py -d DIR: every R1/R2 pair of the folder, one worker per (here: injected) engine; outputs equal to
    running each pair on its own"""
    from oracle import oracle
    work = str(tmp_path / "in")
    os.makedirs(work)
    pairs = []
    for k in range(3):
        sub = os.path.join(work, "s%d" % k)
        os.makedirs(sub)
        p1, p2 = write_pair(sub, 80 + 10 * k, 80 + 10 * k, seed=100 + k)
        q1, q2 = os.path.join(work, "sample%d_R1.fq" % k), os.path.join(work, "sample%d_R2.fq" % k)
        os.rename(p1, q1); os.rename(p2, q2)
        pairs.append((q1, q2))
    argv = ["-d", work, "-f", "0", "-t", "0", "-g", os.path.join(work, "good"), "-b", os.path.join(work, "bad"),
            "-r", os.path.join(work, "QC")]
    options, _ = after.parseCommand(argv)
    after.finalize_options(options)
    stats = after.processDir(work, options, engine_factory=lambda dev: oracle.OracleEngine(), n_workers=2)
    assert len(stats) == 3 and all(s is not None for s in stats)
    for k, (q1, q2) in enumerate(pairs):
        ref_stat, ref_files = run(str(tmp_path), q1, q2, "text", "solo%d" % k)
        for sub_dir in ("good", "bad"):
            for fn, data in ref_files.items():
                if fn.startswith(sub_dir + "/"):
                    with open(os.path.join(work, fn), "rb") as f:
                        assert f.read() == data, fn


@pytest.mark.gpu
def test_directory_mode_two_workers_on_gpu0(tmp_path, monkeypatch):
    """after.py -d DIR on the HIP engine (after.py:101-171): AQC_DEVICES=0,0 = two workers, each with its own context on GPU 0,
    taking file pairs off one queue; outputs, statistics and reports equal to running each pair on its own"""
    monkeypatch.setenv("AQC_DEVICES", "0,0")
    work = str(tmp_path / "in")
    os.makedirs(work)
    pairs = []
    for k in range(4):
        sub = os.path.join(work, "s%d" % k)
        os.makedirs(sub)
        p1, p2 = write_pair(sub, 300 + 40 * k, 300 + 40 * k, seed=200 + k)
        q1, q2 = os.path.join(work, "sample%d_R1.fq" % k), os.path.join(work, "sample%d_R2.fq" % k)
        os.rename(p1, q1); os.rename(p2, q2)
        pairs.append((q1, q2))
    argv = ["-d", work, "-f", "0", "-t", "0", "-g", os.path.join(work, "good"), "-b", os.path.join(work, "bad"),
            "-r", os.path.join(work, "QC")]
    options, _ = after.parseCommand(argv)
    after.finalize_options(options)
    assert after.visible_devices() == [0, 0]
    stats = after.processDir(work, options)
    assert len(stats) == 4 and all(s is not None for s in stats)
    by_file = {os.path.basename(s["command"]["read1_file"]): s for s in stats}       # (the folder is listed in no particular order)
    for k, (q1, q2) in enumerate(pairs):
        stat_k = by_file[os.path.basename(q1)]
        ref_stat, ref_files = run(str(tmp_path), q1, q2, "text", "solo%d" % k)       # (oracle engine, serial loop)
        for fn, data in ref_files.items():
            if fn.startswith(("good/", "bad/")):
                with open(os.path.join(work, fn), "rb") as f:
                    assert f.read() == data, fn
        for key in ref_stat:
            if key != "command":
                assert json.dumps(stat_k[key], sort_keys=True) == json.dumps(ref_stat[key], sort_keys=True), (k, key)
        assert os.path.exists(os.path.join(work, "QC", os.path.basename(q1) + ".html"))


@pytest.mark.gpu
def test_cli_one_input_over_two_contexts(tmp_path, monkeypatch, e2e):
    """`after.py -1 R1 -2 R2` with AQC_DEVICES=0,0: the CLI deals ONE input over two contexts (here both on GPU 0) through
    the whole-input pipe; outputs and statistics are the reference's (golden case pe_default)"""
    import cases
    from test_host_golden import CASE_TABLE, check_case
    monkeypatch.setenv("AQC_DEVICES", "0,0")
    name = "pe_default"
    _, argv, spec, _ = CASE_TABLE[name]
    work = str(tmp_path)
    cases.materialize(spec, work)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        after.main(list(argv))
    finally:
        os.chdir(cwd)
    rec = e2e[name]
    with open(os.path.join(work, rec["stat_file"])) as f:
        stat = json.load(f)
    check_case(name, work, stat, e2e)
