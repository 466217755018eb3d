"""End-to-end golden cases on CPU: the product's host driver (afterqc_amd.after / preprocesser / qc /
fastq) runs with the ORACLE engine injected, and its outputs are compared with what the real
reference produced for the same inputs (tests/golden/e2e_cases.json.gz): byte-identical good / bad /
overlap FASTQ (sha256 of the decompressed files) and an identical stats JSON.

This pins (a) the oracle end to end and (b) every piece of host logic — framing, sampling policy,
auto-trim floats, writers, JSON schema.  The same cases run against the HIP engine in
tests/test_gpu_e2e.py (-m gpu)."""
import gzip
import hashlib
import json
import math
import os

import pytest

import cases
from afterqc_amd import after

CASE_TABLE = {c[0]: c for c in cases.CASES}


# how pass 2 is driven: device-side framing/formatting with production-size chunks, the same with chunks of a few
# records (every carry-over / lock-step corner is hit many times per file), and the host-side framing / writer
MODES = {"text": dict(use_text_path=True, use_pipe=False), "text_tiny_chunks": dict(use_text_path=True, use_pipe=False, chunk_bytes=1500),
         "host": dict(use_text_path=False)}
# GPU only (tests/test_gpu_e2e.py): the whole-input pipe (C++ threads), with production-size chunks, with chunks of a few
# hundred records over three slots, and with ONE input dealt over two contexts (here: two contexts on GPU 0) whose
# statistics are summed on the host
PIPE_MODES = {"pipe": dict(use_text_path=True, use_pipe=True),
              "pipe_small_chunks": dict(use_text_path=True, use_pipe=True, chunk_records=257, pipe_slots=3),
              "pipe_two_contexts": dict(use_text_path=True, use_pipe=True, chunk_records=300, devices=[0, 0], own_engines=True),
              # AQC_SPANS=1: plain-text good files written from the pipe's input buffers (aqc_format_spans), rebuilt records between them
              "pipe_spans": dict(use_text_path=True, use_pipe=True, chunk_records=211, pipe_slots=3, spans=True),
              # AQC_SPANS=2: the same step, the good files' chunks put together on the host by the slot workers (one write() per chunk)
              "pipe_spans_assemble": dict(use_text_path=True, use_pipe=True, chunk_records=223, pipe_slots=3, spans=2),
              # AQC_FUSED=1 (read when the context is created: the filter makes its own): where the chunk allows it the verdict kernel
              # places the records and copies the whole good ones itself; every other case must come out the same through the fallback
              "pipe_fused": dict(use_text_path=True, use_pipe=True, chunk_records=263, pipe_slots=3, own_engines=True, fused=True),
              "text_fused": dict(use_text_path=True, use_pipe=False, own_engines=True, fused=True)}


def run_case(name, tmp_path, engine, mode="text", info=None):
    from afterqc_amd import preprocesser
    _, argv, spec, _ = CASE_TABLE[name]
    work = str(tmp_path)
    cases.materialize(spec, work)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        (options, args) = after.parseCommand(list(argv))
        after.finalize_options(options)
        if options.barcode_flag in options.read1_file and after.parseBool(options.barcode):
            options.barcode = True
            options.trim_front = 0
            options.trim_front2 = 0
        else:
            options.barcode = False
        kw = dict(MODES[mode] if mode in MODES else PIPE_MODES[mode])
        if kw.pop("own_engines", False):
            engine = None                                                        # the filter creates one engine per device
        sp = kw.pop("spans", False)
        env = {"AQC_SPANS": "2" if sp == 2 else "1"} if sp else {}
        if kw.pop("fused", False):
            env["AQC_FUSED"] = "1"
        old_env = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            flt = preprocesser.seqFilter(options, engine=engine, **kw)          # what after.processOptions does
            stat = flt.run()
        finally:
            for k, v in old_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if info is not None:
            info["text_path"] = flt.text_path
            info["used_pipe"] = flt.used_pipe
    finally:
        os.chdir(cwd)
    return work, stat


def digest(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        data = f.read()
    return {"sha256": hashlib.sha256(data).hexdigest(), "lines": data.count(b"\n"), "bytes": len(data)}


def check_case(name, work, stat, golden):
    rec = golden[name]
    # ---- files: same set, same decompressed bytes
    for rel, exp in rec["files"].items():
        path = os.path.join(work, rel)
        assert os.path.exists(path), "missing output " + rel
        got = digest(path)
        assert got == exp, (name, rel, got, exp)
    for sub in ("good", "bad", "overlap", "gout", "bout"):
        d = os.path.join(work, sub)
        if os.path.isdir(d):
            for fn in os.listdir(d):
                assert sub + "/" + fn in rec["files"], "unexpected output %s/%s" % (sub, fn)
    # ---- stats JSON: identical after the two documented py2/py3 deltas
    with open(os.path.join(work, rec["stat_file"])) as f:
        mine = json.load(f)
    exp = rec["stat"]
    if "afterqc_overlap" in exp:
        # golden was captured under py3 (true division); the reference under py2 floors it
        exp = json.loads(json.dumps(exp))
        exp["afterqc_overlap"]["average_overlap_length"] = float(math.floor(exp["afterqc_overlap"]["average_overlap_length"]))
    assert mine.keys() == exp.keys()
    for k in exp:
        assert mine[k] == exp[k], (name, k)


CPU_CASES = [c[0] for c in cases.CASES]


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", CPU_CASES)
def test_host_with_oracle_engine(name, mode, tmp_path, e2e):
    from oracle import oracle
    info = {}
    work, stat = run_case(name, tmp_path, oracle.OracleEngine(), mode, info)
    check_case(name, work, stat, e2e)
    if mode == "host":
        assert not info["text_path"]


def test_text_path_is_the_default_path():
    """every case — barcodes, bubbles, overlap store, qc_only, index files included — goes through aqc_frame /
    aqc_format; the host framing / writer only runs when asked for (use_text_path=False)"""
    from oracle import oracle
    import tempfile
    import pathlib
    used = {}
    for name in CPU_CASES:
        with tempfile.TemporaryDirectory() as d:
            info = {}
            run_case(name, pathlib.Path(d), oracle.OracleEngine(), "text", info)
            used[name] = info["text_path"]
    assert all(used.values()), used
