"""Records whose quality line is not as long as their sequence line: the REFERENCE's behaviour, captured by running it
(tests/golden/make_irregular.py -> tests/golden/irregular_cases.json.gz), against the pure-Python restatement of the per-pair
loop (oracle/pyloop.py).  The reference never compares the two lengths: every string is trimmed, counted and indexed by its own
length (preprocesser.py:19-28, 61-76, 565-568 — negative indices wrap the Python way), the record is written through, and
when the overlap walk reaches a position the quality line does not have the run dies with IndexError.

The product does the same on the device: tests/test_gpu_irregular.py (-m gpu) runs every case of the fixture through the HIP
path.  This test pins the pure-Python restatement on the CPU (the paired cases without barcodes: pyloop has no barcode stage)."""
import gzip
import json
import os

import pytest

from oracle import pyloop

HERE = os.path.dirname(os.path.abspath(__file__))
FLAG_NAME = {pyloop.BADTRIM1: "BADTRIM1", pyloop.BADTRIM2: "BADTRIM2", pyloop.BADLEN: "BADLEN", pyloop.BADPOL: "BADPOL", pyloop.BADLQC: "BADLQC",
             pyloop.BADNCT: "BADNCT", pyloop.BADDIFF: "BADDIFF", pyloop.BADMISMATCH: "BADMISMATCH"}


def _cases():
    with gzip.open(os.path.join(HERE, "golden", "irregular_cases.json.gz"), "rt") as f:
        return json.load(f)["cases"]


def _options(argv):
    """after.py:17-92's defaults for what the loop consults, then the case's flags"""
    opt = dict(trim_front=0, trim_tail=0, seq_len_req=35, poly_size_limit=35, allow_mismatch_in_poly=2, qualified_quality_phred=15,
               unqualified_base_limit=60, n_base_limit=5, no_overlap=0, no_correction=0, mask_mismatch=0, store_overlap=0)
    it = iter(argv)
    for a in it:
        if a in ("-1", "-2"):
            next(it)
        elif a == "-f":
            opt["trim_front"] = int(next(it))
        elif a == "-t":
            opt["trim_tail"] = int(next(it))
        elif a == "-q":
            opt["qualified_quality_phred"] = int(next(it))
        elif a == "-u":
            opt["unqualified_base_limit"] = int(next(it))
        elif a == "--mask_mismatch":
            opt["mask_mismatch"] = 1
        elif a == "--no_correction":
            opt["no_correction"] = 1
        elif a == "--store_overlap":
            opt["store_overlap"] = 1 if next(it) == "on" else 0
        else:
            raise AssertionError("flag not modelled: " + a)
    opt["trim_front2"], opt["trim_tail2"] = opt["trim_front"], opt["trim_tail"]          # after.py:205-206 (trim_pair_same)
    return opt


def _records(text):
    lines = text.split("\n")
    assert lines[-1] == ""
    return [lines[k:k + 4] for k in range(0, len(lines) - 1, 4)]


@pytest.mark.parametrize("case", [c["case"] for c in _cases() if c.get("kind", "pairs") in ("pairs", "adapter")])
def test_pyloop_equals_the_reference_on_irregular_records(case):
    c = [x for x in _cases() if x["case"] == case][0]
    opt = _options(c["argv"])
    r1, r2 = _records(c["r1"]), _records(c["r2"])
    assert len(r1) == len(r2) == 24
    irregular = sum(1 for a, b in zip(r1, r2) if len(a[1]) != len(a[3]) or len(b[1]) != len(b[3]))
    assert irregular >= 1
    out = {"good/R1.good.fq": [], "good/R2.good.fq": [], "bad/R1.bad.fq": [], "bad/R2.bad.fq": [], "overlap/R1.overlap.fq": [], "overlap/R2.overlap.fq": []}
    died = None
    # statRead (qualitycontrol.py:73-110) over the raw reads (pre-filter: the 24-record file is sampled whole, :352-355) and over
    # the GOOD reads as the loop leaves them (post-filter, preprocesser.py:619-622)
    qc = dict((k, pyloop.CycleQC()) for k in ("read1_prefilter", "read2_prefilter", "read1_postfilter", "read2_postfilter"))
    for a, b in zip(r1, r2):
        qc["read1_prefilter"].stat_read(a[1], a[3])
        qc["read2_prefilter"].stat_read(b[1], b[3])
    for i, (a, b) in enumerate(zip(r1, r2)):
        try:
            res = pyloop.process_pair(a[1], a[3], b[1], b[3], opt)
        except IndexError as e:
            died = (i, str(e))
            break
        flag = res["flag"]
        n1, n2 = a[0], b[0]
        if flag != pyloop.GOOD:                                  # preprocesser.py:206-220: the flag goes into the name
            n1, n2 = "@" + FLAG_NAME[flag] + n1[1:], "@" + FLAG_NAME[flag] + n2[1:]
        where = "good" if flag == pyloop.GOOD else "bad"
        if flag == pyloop.GOOD:
            qc["read1_postfilter"].stat_read(res["seq1"], res["qual1"])
            qc["read2_postfilter"].stat_read(res["seq2"], res["qual2"])
        if flag == pyloop.GOOD and opt["store_overlap"] and res["overlap_len"] > 30:
            # getOverlap (preprocesser.py:78-84,614-616): the last overlap_len characters of EACH string — a start index below
            # zero (a quality line shorter than the overlap) counts from the end
            corrected = sum(1 for e in res["edits"] if e[1] in (pyloop.EDIT_FIX_R1, pyloop.EDIT_FIX_R2))
            if res["distance"] == 0 or res["distance"] == corrected:
                ov = res["overlap_len"]
                for m, nm, sq, pl, ql in ((1, n1, res["seq1"], a[2], res["qual1"]), (2, n2, res["seq2"], b[2], res["qual2"])):
                    out["overlap/R%d.overlap.fq" % m].append("%s\n%s\n%s\n%s\n" % (nm, sq[len(sq) - ov:], pl, ql[len(ql) - ov:]))
        out["%s/R1.%s.fq" % (where, where)].append("%s\n%s\n%s\n%s\n" % (n1, res["seq1"], a[2], res["qual1"]))
        out["%s/R2.%s.fq" % (where, where)].append("%s\n%s\n%s\n%s\n" % (n2, res["seq2"], b[2], res["qual2"]))
    if c["returncode"] != 0:
        # the reference died in the overlap walk (a quality line too short for the position it reads): so does the restatement,
        # at the same record, and what was written before it is the same
        assert "IndexError" in (c["error"] or "")
        assert died is not None and died[0] == 1
    else:
        assert died is None
    for name, want in c["files"].items():
        assert "".join(out[name]) == want, (case, name)
    if c["returncode"] == 0:
        summ = c["stat"]["afterqc_main_summary"]
        assert summ["total_reads"] == 24 and summ["good_reads"] == len(out["good/R1.good.fq"]) and summ["bad_reads"] == len(out["bad/R1.bad.fq"])
        assert summ["bad_reads_with_low_quality"] == sum(1 for x in out["bad/R1.bad.fq"] if x.startswith("@BADLQC"))
        # good_bases: the SEQUENCE lines of the good reads 1, whatever their quality lines are (preprocesser.py:621-623: read 2's
        # are only added when there is an index-2 file — an upstream quirk the product keeps too)
        assert summ["good_bases"] == sum(len(x.split("\n")[1]) for x in out["good/R1.good.fq"])
    if c["returncode"] == 0:
        # the per-cycle lists of the stats JSON, to the last bit: positions a short quality line does not cover are counted in the
        # mean quality's denominator and nowhere else
        for key, q in qc.items():
            d = q.derived()
            for section in ("mean_quality", "gc_content", "base_content", "base_quality"):
                assert d[section] == c["stat"][section][key], (case, section, key)
    if c["returncode"] == 0:
        # some of the irregular records really went through the correction walk with shifted quality indices
        changed = sum(1 for k, (a, b) in enumerate(zip(r1, r2)) if (len(a[1]) != len(a[3]) or len(b[1]) != len(b[3])) and
                      any(("%s\n%s\n" % (x[1], x[3])) not in "".join(out["good/R%d.good.fq" % m] + out["bad/R%d.bad.fq" % m]) for m, x in ((1, a), (2, b))))
        if case in ("irr_default", "irr_mask"):
            assert changed >= 2, changed
