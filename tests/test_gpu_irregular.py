"""Records whose quality line is not as long as their sequence line, on the HIP path (-m gpu): the REFERENCE's behaviour —
captured by running it, tests/golden/make_irregular.py -> tests/golden/irregular_cases.json.gz (thirteen runs: short / long /
front-clipped quality lines in either mate, with trims, masking, strict quality, no correction, --store_overlap, a quality line
shorter than the overlap it is read in (python's negative-index wrap), adapter read-through, barcodes, single-end, and the
IndexError death) — reproduced by the product
through every driver: the serial chunk loop (production-size and tiny chunks), the host cross-check path (packed upload with
aqc_batch.qlen*, Python writer fed by aqc_fetch_quality_views) and the whole-input pipe (one context, tiny chunks over three
slots, two contexts).

What the reference does (fastq.py:37-49, preprocesser.py:19-28,61-76,563-598, qualitycontrol.py:81-88): nothing compares the
two lengths; every slice is a python slice of each string by ITS OWN length; lowQualityNum counts the quality line; the overlap
walk indexes each quality string from its own end (a negative index wraps, an index outside the string is an IndexError that
ENDS THE RUN with everything before that record written); statRead counts a cycle, then skips it when the quality line has no
character there.  tests/test_irregular_oracle.py pins oracle/pyloop.py to the same fixture on the CPU."""
import gzip
import json
import math
import os

import numpy as np
import pytest

from afterqc_amd import after, capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

MODES = {
    "text": dict(use_text_path=True, use_pipe=False),
    "text_tiny_chunks": dict(use_text_path=True, use_pipe=False, chunk_bytes=1500),
    "host": dict(use_text_path=False),
    "pipe": dict(use_text_path=True, use_pipe=True),
    "pipe_small_chunks": dict(use_text_path=True, use_pipe=True, chunk_records=5, pipe_slots=3),
    "pipe_two_contexts": dict(use_text_path=True, use_pipe=True, chunk_records=7, devices=[0, 0], own_engines=True),
    "pipe_spans": dict(use_text_path=True, use_pipe=True, chunk_records=6, pipe_slots=2, spans=True),      # AQC_SPANS=1 (aqc_format_spans)
}


def _cases():
    with gzip.open(os.path.join(HERE, "golden", "irregular_cases.json.gz"), "rt") as f:
        return json.load(f)["cases"]


def _run(case, tmp_path, engine, mode, info):
    from afterqc_amd import preprocesser
    work = str(tmp_path)
    argv = list(case["argv"])
    with open(os.path.join(work, argv[argv.index("-1") + 1]), "w") as f:
        f.write(case["r1"])
    if "-2" in argv:
        with open(os.path.join(work, argv[argv.index("-2") + 1]), "w") as f:
            f.write(case["r2"])
    cwd = os.getcwd()
    os.chdir(work)
    try:
        (options, args) = after.parseCommand(argv)
        after.finalize_options(options)
        if options.barcode_flag in options.read1_file and after.parseBool(options.barcode):      # after.py:215-221
            options.barcode = True
            options.trim_front = 0
            options.trim_front2 = 0
        else:
            options.barcode = False
        kw = dict(MODES[mode])
        if kw.pop("own_engines", False):
            engine = None
        spans = kw.pop("spans", False)
        old_spans = os.environ.get("AQC_SPANS")
        if spans:
            os.environ["AQC_SPANS"] = "1"
        flt = preprocesser.seqFilter(options, engine=engine, **kw)
        try:
            stat = flt.run()
        finally:
            info["used_pipe"] = flt.used_pipe
            info["text_path"] = getattr(flt, "text_path", None)
            if spans:
                if old_spans is None:
                    os.environ.pop("AQC_SPANS", None)
                else:
                    os.environ["AQC_SPANS"] = old_spans
    finally:
        os.chdir(cwd)
    return stat


def _files(work):
    out = {}
    for sub in ("good", "bad", "overlap"):
        d = os.path.join(work, sub)
        if os.path.isdir(d):
            for fn in sorted(os.listdir(d)):
                with open(os.path.join(d, fn)) as f:
                    out[sub + "/" + fn] = f.read()
    return out


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("case", [c["case"] for c in _cases()])
def test_irregular_records_like_the_reference(case, mode, tmp_path, gpu_engine):
    c = [x for x in _cases() if x["case"] == case][0]
    info = {}
    if c["returncode"] != 0:
        # the reference died with IndexError in the overlap walk of record 1 (a one-character quality line): so does the
        # product — AQC_ERR_INDEX — with record 0 written, as upstream left it
        assert "IndexError" in c["error"]
        with pytest.raises(capi.AqcError) as ei:
            _run(c, tmp_path, gpu_engine, mode, info)
        assert ei.value.code == capi.ERR_INDEX, ei.value
        gpu_engine.reset_stats()
    else:
        stat = _run(c, tmp_path, gpu_engine, mode, info)
        exp = json.loads(json.dumps(c["stat"]))
        # (the fixture was captured under py3's true division; the reference under py2 floors it, and so does the product)
        if "afterqc_overlap" in exp:
            exp["afterqc_overlap"]["average_overlap_length"] = float(math.floor(exp["afterqc_overlap"]["average_overlap_length"]))
        assert stat.keys() == exp.keys()
        for k in exp:
            if k == "command":
                continue
            assert stat[k] == exp[k], (case, mode, k)
    got = _files(str(tmp_path))
    assert got.keys() == c["files"].keys(), (got.keys(), c["files"].keys())
    for name, want in c["files"].items():
        assert got[name] == want, (case, mode, name)
    if mode.startswith("pipe"):
        assert info["used_pipe"] or c["returncode"] != 0, info     # no fallback to the serial loop: the pipe takes these inputs itself
    if mode == "host":
        assert info["text_path"] is False


def test_irregular_records_are_deferred_not_refused(gpu_engine):
    """the lane-per-read kernel hands every marked record to the general kernel (aqc_last_deferred), and only those"""
    c = _cases()[0]
    text1, text2 = c["r1"].encode(), c["r2"].encode()
    cfg = capi.Config()
    cfg.paired = 1
    cfg.seq_len_req = 35
    cfg.poly_size_limit = 35
    cfg.allow_mismatch_in_poly = 2
    cfg.qualified_quality_phred = 15
    cfg.unqualified_base_limit = 60
    cfg.n_base_limit = 5
    cfg.qc_kmer = 8
    gpu_engine.set_config(cfg)
    gpu_engine.reset_stats()
    pad = lambda b: np.frombuffer(b + b"\0" * 4096, dtype=np.uint8).copy()
    info = gpu_engine.frame(0, pad(text1), len(text1), True, pad(text2), len(text2), True)
    assert int(info.n) == 24
    gpu_engine.run(0)
    n_def, idx = gpu_engine.last_deferred(0, want_indices=True)
    r1 = [x.split("\n") for x in c["r1"].split("@IRR")[1:]]
    r2 = [x.split("\n") for x in c["r2"].split("@IRR")[1:]]
    irregular = sorted(i for i, (a, b) in enumerate(zip(r1, r2)) if len(a[1]) != len(a[3]) or len(b[1]) != len(b[3]))
    assert sorted(int(i) for i in idx[:n_def]) == irregular and len(irregular) >= 6
    gpu_engine.reset_stats()
