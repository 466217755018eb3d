#!/usr/bin/env python3
"""Golden vectors for records whose QUALITY line is not as long as their SEQUENCE line, produced by the REAL reference
(/root/reference, under the py3 shim of make_golden.py) — run in the build container only:

    python tests/golden/make_irregular.py        ->  tests/golden/irregular_cases.json.gz

The reference has no length check: fastq.py:37-49 hands the four lines over as they are, preprocesser.py:19-28 trims each
string by its own length, :61-76 counts each line on its own, the overlap walk (:565-568) indexes every quality string from ITS
OWN end (negative indices wrap the Python way, an index beyond the string raises IndexError and ends the run), statRead
(qualitycontrol.py:81-88) swallows the positions the quality line does not have, and the record is written through as it is.
Each case below is a small paired input (inputs kept in the fixture as text) with a handful of such records between regular
ones; the fixture keeps the reference's output FILES as text (they are a few kilobytes), its stats JSON, and — for the one
case where the walk reaches a missing position — the fact that the reference died with IndexError.
tests/test_gpu_irregular.py runs every case through the HIP path (serial loop, host cross-check, pipe) against this fixture;
tests/test_irregular_oracle.py pins oracle/pyloop.py to the plain paired cases of it on the CPU."""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402


def revcomp(s):
    return "".join({"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}[c] for c in reversed(s))


def make_pairs(rng, n, L):
    """pairs that overlap by 40 .. 70 bases, some with one or two mismatches in the overlap at contrasting qualities (so the
    correction walk runs), some without overlap"""
    recs = []
    for i in range(n):
        ov = rng.choice([0, 45, 52, 60, 70])
        frag_len = 2 * L - ov if ov else 2 * L + 30
        frag = "".join(rng.choice("ACGT") for _ in range(frag_len))
        s1 = frag[:L]
        s2 = revcomp(frag[-L:])
        q1 = [rng.choice("FFFFFFFFGGGHIIJ") for _ in range(L)]
        q2 = [rng.choice("FFFFFFFFGGGHIIJ") for _ in range(L)]
        if ov and i % 3 != 0:
            # a mismatch inside the overlap: read 2 wrong and bad there, read 1 good
            o = rng.randrange(5, ov - 5)
            p2 = L - o - 1
            wrong = rng.choice([c for c in "ACGT" if c != s2[p2]])
            s2 = s2[:p2] + wrong + s2[p2 + 1:]
            q2[p2] = "#"
            q1[L - ov + o] = "J"
        recs.append(["@IRR:1:FC:1:%d:%d:%d 1:N:0:ACGT" % (1101 + i, 1000 + 7 * i, 2000 + 3 * i), s1, "+", "".join(q1),
                     "@IRR:1:FC:1:%d:%d:%d 2:N:0:ACGT" % (1101 + i, 1000 + 7 * i, 2000 + 3 * i), s2, "+", "".join(q2)])
    return recs


def make_adapter_pairs(rng, n, L):
    """inserts SHORTER than the reads (50 .. 74 bases): both mates read through into adapter sequence, util.overlap reports a
    negative offset and the loop cuts all four strings to [0:overlap_len] (preprocesser.py:520-524) — each by its own length;
    every third pair carries a correctable mismatch"""
    recs = []
    for i in range(n):
        f = rng.choice([50, 56, 63, 74])
        frag = "".join(rng.choice("ACGT") for _ in range(f))
        ad1 = "".join(rng.choice("ACGT") for _ in range(L - f))
        ad2 = "".join(rng.choice("ACGT") for _ in range(L - f))
        s1 = frag + ad1
        s2 = revcomp(frag) + ad2
        q1 = [rng.choice("FFFFFFFFGGGHIIJ") for _ in range(L)]
        q2 = [rng.choice("FFFFFFFFGGGHIIJ") for _ in range(L)]
        if i % 3 == 1:
            p2 = rng.randrange(8, f - 8)
            wrong = rng.choice([c for c in "ACGT" if c != s2[p2]])
            s2 = s2[:p2] + wrong + s2[p2 + 1:]
            q2[p2] = "#"
            q1[f - 1 - p2] = "J"
        recs.append(["@IRA:1:FC:1:%d:%d:%d 1:N:0:ACGT" % (1101 + i, 1000 + 7 * i, 2000 + 3 * i), s1, "+", "".join(q1),
                     "@IRA:1:FC:1:%d:%d:%d 2:N:0:ACGT" % (1101 + i, 1000 + 7 * i, 2000 + 3 * i), s2, "+", "".join(q2)])
    return recs


def make_barcode_pairs(rng, n, L):
    """barcoded pairs (12-base barcode + verify CAGTA in front of both mates, the default --barcode_length / --barcode_verify):
    moveBarcodeToName (barcodeprocesser.py:34-45) slices read[1] and read[3] by removeLen, cleanBarcodeTail (:47-75) by -compLen —
    each string by its own length.  Some inserts are short enough for the mates to read into each other's barcode."""
    recs = []
    for i in range(n):
        body = L - 17
        ov = rng.choice([0, 45, 52, body, body + 6])
        frag_len = 2 * body - ov if ov else 2 * body + 30
        frag = "".join(rng.choice("ACGT") for _ in range(frag_len))
        bc1 = "".join(rng.choice("ACGT") for _ in range(12))
        bc2 = "".join(rng.choice("ACGT") for _ in range(12))
        # the fragment as the sequencer sees it: barcode + verify on both ends
        full = bc1 + "CAGTA" + frag + revcomp(bc2 + "CAGTA")
        s1 = full[:L]
        s2 = revcomp(full)[:L]
        q1 = "".join(rng.choice("FFFFFFFFGGGHIIJ") for _ in range(L))
        q2 = "".join(rng.choice("FFFFFFFFGGGHIIJ") for _ in range(L))
        recs.append(["@IRB:1:FC:1:%d:%d:%d 1:N:0:ACGT" % (1101 + i, 1000 + 7 * i, 2000 + 3 * i), s1, "+", q1,
                     "@IRB:1:FC:1:%d:%d:%d 2:N:0:ACGT" % (1101 + i, 1000 + 7 * i, 2000 + 3 * i), s2, "+", q2])
    return recs


# (record index, mate, how): "short k" drops the last k quality characters, "long k" appends k, "front k" drops the FIRST k
EDITS = {
    "short_long_r1_r2": [(2, 1, ("short", 5)), (5, 1, ("long", 7)), (8, 2, ("short", 4)), (11, 2, ("long", 9)), (14, 1, ("short", 1)), (14, 2, ("long", 1)),
                         (17, 1, ("front", 6)), (20, 2, ("front", 3))],
    "walk_reaches_missing_quality": [(1, 2, ("short", 79))],       # read 2's quality line is ONE character: r2[3][-o-1] raises at o = 1
    # read 1's quality line SHORTER THAN THE OVERLAP: r1[3][len(r1[3]) - overlap_len + o] starts at a negative index, which python
    # wraps to the END of the string — the walk compares (and edits) qualities of quite other positions, and may read one it
    # has just written
    "negative_index_wraps": [(k, 1, ("short", 25)) for k in range(1, 24, 3)] + [(6, 2, ("short", 12)), (9, 2, ("long", 5))],
}
# (name, edit set, flags[, input kind])
CASES = [
    ("irr_default", "short_long_r1_r2", ["-f", "0", "-t", "0"]),
    ("irr_trim", "short_long_r1_r2", ["-f", "3", "-t", "2"]),
    ("irr_mask", "short_long_r1_r2", ["-f", "0", "-t", "0", "--mask_mismatch"]),
    ("irr_strict_quality", "short_long_r1_r2", ["-f", "0", "-t", "0", "-q", "38", "-u", "42"]),
    ("irr_no_correction", "short_long_r1_r2", ["-f", "0", "-t", "0", "--no_correction"]),
    ("irr_index_error", "walk_reaches_missing_quality", ["-f", "0", "-t", "0"]),
    # round 5: the rest of what slices or indexes a quality string
    ("irr_store_overlap", "short_long_r1_r2", ["-f", "0", "-t", "0", "--store_overlap", "on"]),          # getOverlap: r[3][len(r[3]) - overlap_len:]
    ("irr_wrap", "negative_index_wraps", ["-f", "0", "-t", "0"]),
    ("irr_wrap_mask_overlap", "negative_index_wraps", ["-f", "0", "-t", "0", "--mask_mismatch", "--store_overlap", "on"]),
    ("irr_adapter", "short_long_r1_r2", ["-f", "0", "-t", "0"], "adapter"),                          # [0:overlap_len] of each string
    ("irr_adapter_trim", "negative_index_wraps", ["-f", "2", "-t", "1", "--store_overlap", "on"], "adapter"),
    ("irr_barcode", "short_long_r1_r2", ["-f", "0", "-t", "0"], "barcode"),                          # moveBarcodeToName / cleanBarcodeTail
    ("irr_single_end", "short_long_r1_r2", ["-f", "3", "-t", "2"], "single"),
]


def apply_edits(recs, edits):
    recs = [list(r) for r in recs]
    for idx, mate, (how, k) in edits:
        col = 3 if mate == 1 else 7
        q = recs[idx][col]
        if how == "short":
            q = q[:len(q) - k]
        elif how == "long":
            q = q + "".join("FGHIJ"[(idx + j) % 5] for j in range(k))
        else:
            q = q[k:]
        recs[idx][col] = q
    return recs


def texts(recs):
    r1 = "".join("%s\n%s\n%s\n%s\n" % tuple(r[0:4]) for r in recs)
    r2 = "".join("%s\n%s\n%s\n%s\n" % tuple(r[4:8]) for r in recs)
    return r1, r2


def run_reference(r1_text, r2_text, argv, kind="pairs"):
    work = tempfile.mkdtemp(prefix="aqc_irr_")
    try:
        # (barcode mode is switched on by the FILE NAME, after.py:215-221)
        n1, n2 = ("R1.barcode.fq", "R2.barcode.fq") if kind == "barcode" else ("R1.fq", "R2.fq")
        with open(os.path.join(work, n1), "w") as f:
            f.write(r1_text)
        if kind != "single":
            with open(os.path.join(work, n2), "w") as f:
                f.write(r2_text)
        full = ["-1", n1] + (["-2", n2] if kind != "single" else []) + argv
        p = subprocess.run([sys.executable, os.path.join(HERE, "make_golden.py"), "--run-ref"] + full, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        rec = {"argv": full, "returncode": p.returncode, "files": {}, "stat": None, "error": None}
        if p.returncode != 0:
            tail = p.stdout.strip().splitlines()
            rec["error"] = tail[-1] if tail else "?"
        for sub in ("good", "bad", "overlap", "QC"):
            d = os.path.join(work, sub)
            if not os.path.isdir(d):
                continue
            for fn in sorted(os.listdir(d)):
                path = os.path.join(d, fn)
                if fn.endswith(".json"):
                    with open(path) as f:
                        rec["stat"] = json.load(f)
                elif fn.endswith(".html"):
                    continue
                else:
                    with open(path) as f:
                        rec["files"][sub + "/" + fn] = f.read()
        return rec
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    out = {"what": __doc__.split("\n")[0], "cases": []}
    for case in CASES:
        name, edit_key, argv = case[:3]
        kind = case[3] if len(case) > 3 else "pairs"
        rng = random.Random(20260927)
        maker = {"pairs": make_pairs, "single": make_pairs, "adapter": make_adapter_pairs, "barcode": make_barcode_pairs}[kind]
        recs = apply_edits(maker(rng, 24, 80), EDITS[edit_key])
        r1, r2 = texts(recs)
        rec = run_reference(r1, r2, argv, kind)
        rec.update(case=name, kind=kind, edits=[[i, m, list(h)] for i, m, h in EDITS[edit_key]], r1=r1, r2=r2)
        out["cases"].append(rec)
        print(name, "returncode", rec["returncode"], rec["error"] or "", {k: v.count("\n") // 4 for k, v in rec["files"].items()})
    with gzip.open(os.path.join(HERE, "irregular_cases.json.gz"), "wt") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
