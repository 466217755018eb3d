"""End-to-end golden cases: name, after.py argv, input spec, keep-outputs flag.

Pure data + a materialiser that rebuilds the inputs from seeds (afterqc_amd.synth) or from the
reference's own test data files, so make_golden.py (reference run) and the parity tests (our run)
see byte-identical inputs.
"""
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# NB the reference compares int(<tile field>[1:]) with the CSV tile (preprocesser.py:189-190): "1101" -> 101
CIRCLES = [(5000.0, 6000.0, 1500.0, 1, 101), (12000.5, 9000.25, 2500.75, 2, 405), (8000.0, 8000.0, 3000.0, 3, 205),
           (15000.0, 12000.0, 4000.0, 4, 808), (20000.0, 5000.0, 2000.0, 1, 215), (7000.0, 15000.0, 3500.0, 2, 101),
           (18000.0, 15000.0, 3999.5, 3, 610), (3000.0, 3000.0, 1999.0, 4, 10)]


def pe(seed, n=2000, L=150, ragged=False, r1="R1.fq", r2="R2.fq", lowercase=0.0, barcode=False, index=False,
       circles=False, dirty=True, short_frac=0.03, tail_frac=0.0):
    return dict(kind="pe", seed=seed, n=n, L=L, ragged=ragged, r1=r1, r2=r2, lowercase=lowercase, barcode=barcode,
                index=index, circles=circles, dirty=dirty, short_frac=short_frac, tail_frac=tail_frac)


def se(seed, n=3000, L=150):
    return dict(kind="se", seed=seed, n=n, L=L)


def lowcomplex(seed, n=1500):
    return dict(kind="lowcomplex", seed=seed, n=n)


TESTDATA = dict(kind="testdata")
PE = ["-1", "R1.fq", "-2", "R2.fq"]
F0 = ["-f", "0", "-t", "0"]

CASES = [
    ("g1_testdata", ["-1", "R1.fq.gz", "-2", "R2.fq.gz"], TESTDATA, True),
    ("g1_testdata_se", ["-1", "R1.fq.gz"], TESTDATA, False),
    ("se_default", ["-1", "R1.fq"], se(1002), False),
    ("se_cfg2_flags", ["-1", "R1.fq", "-f", "5", "-t", "5", "-q", "15", "-u", "60", "-p", "35", "-a", "2", "-n", "5",
                       "-s", "35"], se(2002), False),
    ("pe_default", PE, pe(1003), False),
    ("pe_clean_cfg3", PE + F0, pe(1004, dirty=False), False),
    ("pe_f0t0", PE + F0, pe(1013), False),
    ("pe_nocorr", PE + F0 + ["--no_correction"], pe(1023), False),
    ("pe_mask", PE + F0 + ["--mask_mismatch"], pe(1033), False),
    ("pe_nocorr_mask", PE + F0 + ["--mask_mismatch", "--no_correction"], pe(1034), False),
    ("pe_nooverlap", PE + ["--no_overlap"], pe(1043), False),
    ("pe_store_overlap", PE + F0 + ["--store_overlap", "on"], pe(1053), False),
    ("pe_trim_explicit", PE + ["-f", "3", "-t", "2"], pe(1063), False),
    ("pe_trim_tail_only", PE + ["-f", "0", "-t", "4"], pe(1064), False),
    ("pe_trim_pair_diff", PE + ["--trim_pair_same", "false"], pe(1073), False),
    ("pe_index", PE + ["-7", "I1.fq", "-5", "I2.fq"] + F0, pe(1083, index=True), False),
    ("pe_gz", ["-1", "R1.fq.gz", "-2", "R2.fq.gz"] + F0, pe(1093, r1="R1.fq.gz", r2="R2.fq.gz"), False),
    ("pe_gz_out", PE + F0 + ["-z", "--compression", "4"], pe(1094), False),
    # .bz2 inputs (fastq.py:25-26): read through bz2.BZ2File upstream; the outputs are plain (only ".gz" selects a codec)
    ("pe_bz2", ["-1", "R1.fq.bz2", "-2", "R2.fq.bz2"] + F0, pe(1095, r1="R1.fq.bz2", r2="R2.fq.bz2"), False),
    ("pe_barcode", ["-1", "barcode_R1.fq", "-2", "barcode_R2.fq", "-t", "0"],
     pe(1103, L=120, r1="barcode_R1.fq", r2="barcode_R2.fq", barcode=True), False),
    ("se_barcode", ["-1", "barcode_R1.fq", "-t", "0"],
     pe(1104, L=120, r1="barcode_R1.fq", r2="unused_R2.fq", barcode=True), False),
    ("pe_debubble", PE + F0 + ["--debubble", "--debubble_dir", "D"], pe(1113, circles=True), False),
    ("pe_qc_only", PE + ["--qc_only"], pe(1123), False),
    ("pe_qc_sample_small", PE + F0 + ["--qc_sample", "500"], pe(1133), False),
    ("pe_qc_sample_zero", PE + F0 + ["--qc_sample", "0"], pe(1143, n=1500), False),
    ("pe_ragged", PE + F0, pe(1153, ragged=True), False),
    ("pe_ragged_short", PE + F0 + ["-s", "20"], pe(1154, ragged=True, short_frac=0.6), False),
    ("pe_lowcomplex", PE + F0 + ["-p", "0", "-s", "10"], lowcomplex(1155), False),
    ("pe_lowcomplex_mask", PE + F0 + ["-p", "0", "-s", "10", "--mask_mismatch"], lowcomplex(1156), False),
    ("pe_ragged_trim", PE + ["-f", "2", "-t", "3", "-s", "20"], pe(1163, ragged=True), False),
    ("pe_lowercase", PE + F0 + ["--no_correction"], pe(1173, lowercase=0.2), False),
    ("pe_filters_off", PE + F0 + ["-p", "0", "-u", "0", "-n", "0"], pe(1183), False),
    ("pe_strict", PE + F0 + ["-q", "20", "-u", "20", "-p", "20", "-a", "1", "-n", "1", "-s", "100"], pe(1193), False),
    ("pe_l100", PE + F0, pe(1203, L=100), False),
    ("pe_l250", PE + F0, pe(1213, n=800, L=250), False),
    ("pe_outdirs", PE + F0 + ["-g", "gout", "-b", "bout", "-r", "rout"], pe(1223, n=300), False),
    ("pe_qc_kmer5", PE + F0 + ["--qc_kmer", "5"], pe(1233, n=600), False),
    # BASELINE.json config 5 in one piece: 2x250 + 17-base barcode/verify prefix on both mates, "barcode" in the file
    # names (after.py:215-221), gzip in, -z out, --debubble with a circles.csv, default (auto) tail trim
    ("pe_cfg5", ["-1", "barcode_R1.fq.gz", "-2", "barcode_R2.fq.gz", "--debubble", "--debubble_dir", "D", "-z"],
     pe(1305, n=1200, L=250, r1="barcode_R1.fq.gz", r2="barcode_R2.fq.gz", barcode=True, circles=True, tail_frac=0.12), False),
    ("pe_barcode_tails", ["-1", "barcode_R1.fq", "-2", "barcode_R2.fq", "-t", "0"],
     pe(1306, n=1500, L=100, r1="barcode_R1.fq", r2="barcode_R2.fq", barcode=True, tail_frac=0.3), False),
]


def materialize(spec, work):
    from afterqc_amd import synth
    if spec["kind"] == "testdata":
        shutil.copy(os.path.join(HERE, "testdata", "R1.fq.gz"), work)
        shutil.copy(os.path.join(HERE, "testdata", "R2.fq.gz"), work)
        return
    if spec["kind"] == "se":
        d = synth.make_single(spec["n"], spec["L"], spec["seed"])
        lane, tile, x, y = d["meta"]
        synth.write_fastq(os.path.join(work, "R1.fq"), synth.render_names(lane, tile, x, y, 1), d["seq1"], d["qual1"],
                          d["len1"])
        return
    if spec["kind"] == "lowcomplex":
        write_lowcomplex(spec, work)
        return
    n, seed = spec["n"], spec["seed"]
    d = synth.make_pairs(n, spec["L"], seed, ragged=spec["ragged"], lowercase=spec["lowercase"],
                         dirty=spec["dirty"], short_frac=spec["short_frac"])
    if spec["barcode"]:
        d = synth.add_barcodes(d, seed + 7, tail_frac=spec.get("tail_frac", 0.0))
    lane, tile, x, y = d["meta"]
    if spec["circles"]:
        # put the clusters on the circles' lanes/tiles so that BADBBL actually fires
        which = tile % len(CIRCLES)
        lane = np.where(x % 5 == 0, lane, np.array([c[3] for c in CIRCLES])[which])
        tile = np.array([c[4] for c in CIRCLES])[which] + 1000 * (1 + (y % 2))
    synth.write_fastq(os.path.join(work, spec["r1"]), synth.render_names(lane, tile, x, y, 1), d["seq1"], d["qual1"],
                      d["len1"])
    synth.write_fastq(os.path.join(work, spec["r2"]), synth.render_names(lane, tile, x, y, 2), d["seq2"], d["qual2"],
                      d["len2"])
    if spec["index"]:
        rng = np.random.Generator(np.random.PCG64(seed + 99))
        i1 = synth.BASES[rng.integers(0, 4, (n, 8), dtype=np.uint8)]
        i2 = synth.BASES[rng.integers(0, 4, (n, 8), dtype=np.uint8)]
        iq = synth._quals(rng, (n, 8))
        l8 = np.full(n, 8)
        synth.write_fastq(os.path.join(work, "I1.fq"), synth.render_names(lane, tile, x, y, 1), i1, iq, l8)
        synth.write_fastq(os.path.join(work, "I2.fq"), synth.render_names(lane, tile, x, y, 2), i2, iq, l8)
    if spec["circles"]:
        os.makedirs(os.path.join(work, "D"), exist_ok=True)
        with open(os.path.join(work, "D", "circles.csv"), "w") as f:
            f.write("x,y,radius,lane,tile\n")
            for (cx, cy, r, ln, tl) in CIRCLES:
                f.write("%s,%s,%s,%d,%d\n" % (repr(cx), repr(cy), repr(r), ln, tl))


def write_lowcomplex(spec, work):
    """Short-period repeats with a few substitutions and ragged lengths: many diagonals pass the
    overlap test, and the tail-anchored correction walk (preprocesser.py:563-598) sees a different
    column set than the scan did -> BADMISMATCH and odd edits."""
    import random
    rng = random.Random(spec["seed"])
    comp = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N"}
    with open(os.path.join(work, "R1.fq"), "w") as f1, open(os.path.join(work, "R2.fq"), "w") as f2:
        for i in range(spec["n"]):
            unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 1, 2, 2, 3, 4, 7])))
            l1 = rng.randint(40, 150)
            l2 = rng.randint(12, 150)
            base = (unit * 200)
            st = rng.randint(0, 10)
            r1 = list(base[st:st + l1])
            r2 = list("".join(comp[c] for c in reversed(base[:rng.randint(l2, 160)]))[:l2])
            q1 = [rng.choice("#/6<AEEE") for _ in range(l1)]
            q2 = [rng.choice("#/6<AEEE") for _ in range(l2)]
            for r, q in ((r1, q1), (r2, q2)):
                for _ in range(rng.choice([0, 0, 1, 1, 2, 3])):
                    p = rng.randrange(len(r))
                    r[p] = rng.choice("ACGTN")
                    q[p] = rng.choice("#/E")
            nm = "@LC:1:FC1:1:1101:%d:%d" % (1000 + i, 2000 + i)
            f1.write("%s 1:N:0:ACGT\n%s\n+\n%s\n" % (nm, "".join(r1), "".join(q1)))
            f2.write("%s 2:N:0:ACGT\n%s\n+\n%s\n" % (nm, "".join(r2), "".join(q2)))
