#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container: it imports the Python reference from /root/reference under
CPython 3 with the 5-point shim of SURVEY.md §8c and refuses to run when the reference is absent
(so it is inert on the GPU box).  Nothing from the reference is copied: the outputs are *data* —
function inputs/outputs and end-to-end counters / digests — plus the reference's own test data
files (testdata/R1.fq.gz, R2.fq.gz), which are data too.

    python tests/golden/make_golden.py            # regenerate everything
"""
import builtins
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


class Py2Int(int):
    """an int whose `/` is python 2's: floor division between integers (results stay Py2Int through + - * //)"""
    def __truediv__(self, o):
        return Py2Int(int(self) // int(o)) if isinstance(o, int) else int(self) / o
    def __rtruediv__(self, o):
        return Py2Int(int(o) // int(self)) if isinstance(o, int) else o / int(self)
    def __add__(self, o):
        return Py2Int(int(self) + int(o)) if isinstance(o, int) else int(self) + o
    __radd__ = __add__
    def __sub__(self, o):
        return Py2Int(int(self) - int(o)) if isinstance(o, int) else int(self) - o
    def __rsub__(self, o):
        return Py2Int(int(o) - int(self)) if isinstance(o, int) else o - int(self)
    def __mul__(self, o):
        return Py2Int(int(self) * int(o)) if isinstance(o, int) else int(self) * o
    __rmul__ = __mul__
    def __floordiv__(self, o):
        return Py2Int(int(self) // int(o)) if isinstance(o, int) else int(self) // o


def install_shim():
    """(1) xrange, (2) text-mode gzip.open, (3) force the pure-Python editDistance."""
    builtins.xrange = range
    _open = gzip.open

    def gzopen(fn, mode="r", *a, **k):
        if mode in ("r", "w"):
            mode += "t"
        return _open(fn, mode, *a, **k)
    gzip.open = gzopen
    # (2b) the same for bz2.BZ2File (fastq.py:25-26): python 2 hands out str lines, python 3 bytes
    import bz2
    import io
    _bz = bz2.BZ2File
    bz2.BZ2File = lambda fn, mode="r", *a, **k: io.TextIOWrapper(_bz(fn, mode, *a, **k)) if mode in ("r", "w") else _bz(fn, mode, *a, **k)
    sys.path.insert(0, REF)
    import util
    util.EDIT_DISTANCE_MODULE_EXISTS = False
    util.EDIT_DISTANCE_CTYPES_LOADED = False


# --------------------------------------------------------------------------------------------
# end-to-end driver (runs in a subprocess: `make_golden.py --run-ref <argv...>` inside a work dir)
# --------------------------------------------------------------------------------------------
def run_ref_main(argv):
    """Replay after.py:195-222 (the py3 guard at after.py:189-191 is bypassed) and tolerate the
    py2-only TypeError raised by strandBiasPlotly *after* FASTQ + JSON are written."""
    install_shim()
    import after
    sys.argv = ["after.py"] + argv
    (options, args) = after.parseCommand()
    options.version = after.AFTERQC_VERSION
    options.trim_pair_same = after.parseBool(options.trim_pair_same)
    options.draw = after.parseBool(options.draw)
    options.store_overlap = after.parseBool(options.store_overlap)
    options.trim_front2 = options.trim_front
    options.trim_tail2 = options.trim_tail
    if options.barcode_flag in options.read1_file and after.parseBool(options.barcode):
        options.barcode = True
        options.trim_front = 0
        options.trim_front2 = 0
    else:
        options.barcode = False
    # (6) the report: strandBiasPlotly (qualitycontrol.py:238-270) divides list lengths with `/` and feeds the result to
    # xrange — python 2 integer division.  Inside qualitycontrol ONLY, len() hands out ints whose `/` floors like python 2's,
    # so the reference's own code runs to the end and writes its HTML report (the one py2-vs-py3 delta of the report path).
    import qualitycontrol
    qualitycontrol.len = lambda x: Py2Int(builtins.len(x))
    try:
        after.processOptions(options)
    except TypeError as e:  # (should not fire any more; kept so that FASTQ + JSON goldens survive a report problem)
        print("tolerated:", e)


def js_to_json(text):
    """the object literals the reference's *Plotly emitters write (bare keys, single quotes) -> JSON text"""
    import re
    text = text.replace("'", '"')
    return re.sub(r'([{,]\s*)([A-Za-z_][A-Za-z0-9_]*)\s*:', r'\1"\2":', text)


def parse_report(html):
    """the reference's HTML report as data: menu titles, summary rows, and per Plotly div the traces + layout"""
    import re
    rep = {"menu": re.findall(r"<li class='menu-item'><a href='#([^']*)'>(\d+), (.*?)</a> </li>", html),
           "summary": re.findall(r"<tr><td class='col1'>(.*?)</td><td class='col2'>(.*?)</td></tr>", html),
           "sections": re.findall(r"<div class='figure-title'><a name='([^']*)'>(\d+), (.*?)</a></div>\n<div id='([^']*)' class='plotly-div'></div>", html),
           "figures": {}}
    for m in re.finditer(r"var data=(\[.*?\]);\s*var layout=(\{.*?\});\s*Plotly\.newPlot\('([A-Za-z0-9_]+)', data, layout\);", html, re.S):
        rep["figures"][m.group(3)] = {"data": json.loads(js_to_json(m.group(1))), "layout": json.loads(js_to_json(m.group(2)))}
    return rep


def sha_lines(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        data = f.read()
    return {"sha256": hashlib.sha256(data).hexdigest(), "lines": data.count(b"\n"), "bytes": len(data)}


def run_case(name, argv, setup, keep_outputs=False):
    """setup(workdir) writes the inputs; returns the golden record for this case."""
    work = tempfile.mkdtemp(prefix="aqc_gold_")
    try:
        setup(work)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--run-ref"] + argv, cwd=work,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            print(p.stdout[-3000:])
            raise RuntimeError("reference failed on case " + name)
        rec = {"case": name, "argv": argv, "files": {}, "stat": None}
        for sub in ("good", "bad", "overlap", "QC", "gout", "bout", "oout", "rout"):
            d = os.path.join(work, sub)
            if not os.path.isdir(d):
                continue
            for fn in sorted(os.listdir(d)):
                path = os.path.join(d, fn)
                if fn.endswith(".json"):
                    with open(path) as f:
                        rec["stat"] = json.load(f)
                    rec["stat_file"] = sub + "/" + fn
                elif fn.endswith(".html"):
                    with open(path) as f:
                        rec["report"] = parse_report(f.read())
                    continue
                else:
                    rec["files"][sub + "/" + fn] = sha_lines(path)
                    if keep_outputs:
                        dst = os.path.join(HERE, name + "_out", sub)
                        os.makedirs(dst, exist_ok=True)
                        op = gzip.open if fn.endswith(".gz") else open
                        with op(path, "rb") as f:
                            data = f.read()
                        with open(os.path.join(dst, fn[:-3] if fn.endswith(".gz") else fn), "wb") as f:
                            f.write(data)
        return rec
    finally:
        shutil.rmtree(work, ignore_errors=True)


# --------------------------------------------------------------------------------------------
# function-level vectors (G2)
# --------------------------------------------------------------------------------------------
def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def revcomp(s):
    m = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N", "a": "t", "t": "a", "c": "g", "g": "c"}
    return "".join(m.get(c, "N") for c in reversed(s))


def mutate(rng, s, k, alphabet="ACGT"):
    s = list(s)
    for p in rng.sample(range(len(s)), min(k, len(s))):
        s[p] = rng.choice([c for c in alphabet if c != s[p]])
    return "".join(s)


def overlap_inputs(rng):
    """Pairs covering: no overlap, forward overlaps of every size, adapter read-through (negative
    offsets), unequal/short lengths, N and lower case, mismatch positions around i=49/50/51,
    overlap_len 51/52, late mismatches (diff > 3 accepted)."""
    cases = []
    cases.append(("CAGCGCCTACGGGCCCCTTTTTCTGCGCGACCGCGTGGCTGTGGGCGCGGATGCCTTTGAGCGCGGTGACTTCTCACTGCGTATCGAGC",
                  "ACCTCCAGCGGCTCGATACGCAGTGAGAAGTCACCGCGCTCAAAGGCATCCGCGCCCACAGCCACGCGGTCGCGCAGAAAAAGGGGTCC"))
    for _ in range(5200):
        L1 = rng.choice([150, 150, 150, 100, 75, 151, rng.randint(20, 160)])
        L2 = rng.choice([L1, L1, 150, rng.randint(5, 160)])
        kind = rng.random()
        if kind < 0.15:
            r1 = rand_seq(rng, L1); r2 = rand_seq(rng, L2)
        else:
            ins = rng.randint(25, L1 + L2 + 20)
            frag = rand_seq(rng, ins)
            pad1 = rand_seq(rng, L1); pad2 = rand_seq(rng, L2)
            r1 = (frag + pad1)[:L1]
            r2 = (revcomp(frag) + pad2)[:L2]
        nm = rng.choice([0, 0, 0, 1, 2, 3, 4, 6])
        r1 = mutate(rng, r1, nm)
        if rng.random() < 0.3:
            r2 = mutate(rng, r2, rng.choice([1, 2, 3, 5]))
        if rng.random() < 0.15:
            r1 = mutate(rng, r1, rng.randint(1, 6), "N")
            r2 = mutate(rng, r2, rng.randint(1, 6), "N")
        if rng.random() < 0.05:
            k = rng.randint(1, len(r1)); r1 = r1[:k].lower() + r1[k:]
        if rng.random() < 0.05:
            k = rng.randint(1, len(r2)); r2 = r2[:k].lower() + r2[k:]
        if rng.random() < 0.02:
            r1 = mutate(rng, r1, 2, "XnR.")
        if rng.random() < 0.02:
            r2 = mutate(rng, r2, 2, "XnR.")
        cases.append((r1, r2))
    # targeted: mismatches placed at exact diagonal positions
    for L, ov in ((150, 100), (150, 51), (150, 52), (150, 53), (120, 60), (150, 150), (100, 100)):
        for pos_set in ((47, 48, 49), (48, 49, 50), (49, 50, 51), (50, 51, 52), (0, 1, 50), (0, 49, 51),
                        (10, 60, 70), (51, 52, 53, 54, 55, 56), (0, 1), (49, 50), (3, 50, 51, 52)):
            frag = rand_seq(rng, 2 * L - ov)
            r1 = list(frag[:L])
            r2 = revcomp(frag)[:L]
            off = L - ov
            for p in pos_set:
                if p < ov:
                    c = r1[off + p]
                    r1[off + p] = {"A": "C", "C": "G", "G": "T", "T": "A"}[c]
            cases.append(("".join(r1), r2))
    # short reads and degenerate lengths
    for L1, L2 in ((31, 31), (30, 30), (31, 5), (5, 31), (32, 1), (1, 40), (35, 2), (31, 150), (150, 31),
                   (30, 150), (150, 30), (60, 16), (16, 60), (33, 15), (15, 15), (1, 1)):
        for _ in range(6):
            frag = rand_seq(rng, max(L1, L2) + 10)
            r1 = frag[:L1]
            r2 = revcomp(frag[:max(L1, L2)])[:L2] if rng.random() < 0.7 else rand_seq(rng, L2)
            cases.append((r1, r2))
    # low complexity (many prefix survivors)
    for _ in range(60):
        unit = rand_seq(rng, rng.randint(1, 4))
        r1 = mutate(rng, (unit * 200)[:150], rng.randint(0, 8))
        r2 = mutate(rng, revcomp((unit * 200)[:150]), rng.randint(0, 8))
        cases.append((r1, r2))
    return cases


def gen_function_vectors():
    install_shim()
    import util, preprocesser, barcodeprocesser
    from qualitycontrol import QualityControl
    rng = random.Random(20260926)
    out = {}

    ov = []
    for r1, r2 in overlap_inputs(rng):
        ov.append([r1, r2, list(util.overlap(r1, r2))])
    out["overlap"] = ov

    poly = []
    seqs = ["A" * 33 + "C" * 40, "A" * 40 + "X", "X" + "A" * 40, "N" * 40, "A" * 34, "A" * 35, "a" * 40,
            "A" * 32 + "CG" + "A", "AC" * 40, "G" * 16 + "A" + "G" * 16 + "T" + "G" * 20,
            "ACGT" * 5 + "G" * 33 + "AC", "T" * 33, "T" * 32 + "C" * 3, "C" * 31 + "GG" + "CC", ""]
    for _ in range(1500):
        L = rng.choice([150, 150, 100, 60, 40, 35, 34, rng.randint(1, 160)])
        s = rand_seq(rng, L)
        if rng.random() < 0.7 and L > 10:
            k = rng.randint(20, 80); st = rng.randint(0, max(0, L - 10))
            ch = rng.choice("ACGTNacgt")
            s = (s[:st] + ch * k + s[st:])[:L]
            s = mutate(rng, s, rng.randint(0, 6))
        if rng.random() < 0.1 and L > 0:
            s = mutate(rng, s, 1, "X-.n")
        seqs.append(s)
    for s in seqs:
        for mp, mm in ((35, 2), (35, 2), (20, 0), (10, 3), (50, 5)):
            poly.append([s, mp, mm, preprocesser.hasPolyX(s, mp, mm)])
            if rng.random() < 0.8:
                break
    out["polyx"] = poly

    cnt = []
    for _ in range(400):
        L = rng.randint(1, 200)
        s = rand_seq(rng, L, "ACGTNNn")
        q = "".join(chr(rng.randint(33, 75)) for _ in range(L))
        qv = rng.choice([15, 15, 20, 0, 2, 30, 41])
        cnt.append([s, q, qv, preprocesser.lowQualityNum(["@n", s, "+", q], qv), preprocesser.nNumber(["@n", s, "+", q])])
    out["counts"] = cnt

    tr = []
    for _ in range(200):
        L = rng.randint(1, 60)
        s = rand_seq(rng, L); q = "".join(chr(rng.randint(33, 75)) for _ in range(L))
        f = rng.randint(0, 20); t = rng.randint(0, 20)
        r = preprocesser.trim(["@n", s, "+", q], f, t)
        tr.append([s, q, f, t, r[1], r[3]])
    out["trim"] = tr

    ed = []
    for _ in range(1500):
        a = rand_seq(rng, rng.randint(0, 24), "ACGTN")
        if rng.random() < 0.6:
            b = mutate(rng, a, rng.randint(0, 4)) if a else ""
            if rng.random() < 0.5 and len(b) > 2:
                k = rng.randint(0, len(b) - 1); b = b[:k] + b[k + 1:]
            if rng.random() < 0.5:
                k = rng.randint(0, len(b)); b = b[:k] + rng.choice("ACGT") + b[k:]
        else:
            b = rand_seq(rng, rng.randint(0, 24), "ACGTN")
        if len(a) == 0 or len(b) == 0:
            continue  # util.editDistance's DP is undefined for empty strings (util.py:83)
        ed.append([a, b, util.editDistance(a, b)])
    for n in (64, 65, 100, 130, 200):
        a = rand_seq(rng, n); b = mutate(rng, a, 7)[3:]
        ed.append([a, b, util.editDistance(a, b)])
    out["editdistance"] = ed

    bc = []
    verify = "CAGTA"
    for _ in range(1200):
        L = rng.choice([150, 100, 60, 18, 19, 17, rng.randint(1, 40)])
        s = rand_seq(rng, L)
        if L > 20 and rng.random() < 0.85:
            at = rng.choice([12, 12, 12, 11, 13, 10, 14])
            v = verify if rng.random() < 0.6 else mutate(rng, verify, rng.choice([1, 1, 2]))
            s = (s[:at] + v + s[at + len(v):])[:L]
        bc.append([s, 12, verify, barcodeprocesser.detectBarcode(s, 12, verify)])
    out["detect_barcode"] = bc

    mv = []
    for _ in range(800):
        L = rng.choice([150, 100, 60, 40])
        ins = rng.randint(20, 2 * L)
        b1 = rand_seq(rng, 12); b2 = rand_seq(rng, 12)
        frag = rand_seq(rng, ins)
        tpl = b1 + verify + frag + revcomp(b2 + verify)
        r1 = (tpl + rand_seq(rng, 2 * L))[:L]
        r2 = (revcomp(tpl) + rand_seq(rng, 2 * L))[:L]
        r1 = mutate(rng, r1, rng.choice([0, 0, 1, 3]))
        r2 = mutate(rng, r2, rng.choice([0, 0, 1, 3]))
        q1 = "".join(chr(rng.randint(35, 73)) for _ in range(L))
        q2 = "".join(chr(rng.randint(35, 73)) for _ in range(L))
        n1 = "@SIM:1:FC1:1:1101:%d:%d 1:N:0:ACGT" % (rng.randint(1000, 9999), rng.randint(1000, 9999))
        n2 = n1.replace(" 1:", " 2:")
        bl1 = barcodeprocesser.detectBarcode(r1, 12, verify)
        bl2 = barcodeprocesser.detectBarcode(r2, 12, verify)
        if bl1 == 0 or bl2 == 0:
            continue
        a = [n1, r1, "+", q1]; b = [n2, r2, "+", q2]
        barcodeprocesser.moveAndTrimPair(a, b, bl1, bl2, verify)
        mv.append([[n1, r1, q1], [n2, r2, q2], bl1, bl2, verify, [a[0], a[1], a[3]], [b[0], b[1], b[3]]])
    # single-end move
    for _ in range(50):
        r1 = rand_seq(rng, 12) + verify + rand_seq(rng, 60)
        q1 = "".join(chr(rng.randint(35, 73)) for _ in range(len(r1)))
        a = ["@M01:23:FC:1:1101:5:6 1:N:0:AC", r1, "+", q1]
        barcodeprocesser.moveBarcodeToName(a, 12, verify)
        mv.append([["@M01:23:FC:1:1101:5:6 1:N:0:AC", r1, q1], None, 12, 0, verify, [a[0], a[1], a[3]], None])
    out["move_barcode"] = mv

    # QC accumulators + derived floats + autoTrim on small read sets
    qcs = []
    for case in range(12):
        n = rng.choice([1, 5, 40, 200])
        L = rng.choice([151, 100, 36, 9, 5])
        k = rng.choice([8, 8, 8, 4, 3])
        reads = []
        for _ in range(n):
            l = L if rng.random() < 0.7 else rng.randint(5, L)
            s = rand_seq(rng, l, "ACGT" * 12 + "N")
            if case % 4 == 3:
                s = mutate(rng, s, 2, "acgt")
            if rng.random() < 0.2:
                s = (s[:l // 2] + "G" * l)[:l]
            q = "".join(rng.choice("#/6<AE") for _ in range(l))
            reads.append([s, q])
        qc = QualityControl(200000, k)
        for s, q in reads:
            qc.statRead(["@r", s, "+", q])
        qc.qc()
        trimv = list(qc.autoTrim())
        rl = qc.readLen
        qcs.append({
            "kmer": k, "reads": reads, "readLen": rl,
            "totalNum": qc.totalNum[:rl + 2], "totalQual": qc.totalQual[:rl + 2],
            "baseCounts": {b: qc.baseCounts[b][:rl + 2] for b in "ATCG"},
            "baseTotalQual": {b: qc.baseTotalQual[b][:rl + 2] for b in "ATCG"},
            "totalDiscontinuity": qc.totalDiscontinuity[:rl + 2],
            "gcHistogram": qc.gcHistogram[:L + 2], "totalKmer": qc.totalKmer,
            "topKmer": [list(x) for x in qc.topKmerCount[:40]],
            "nKmerKeys": len(qc.kmerCount),
            "meanQual": qc.meanQual[:rl], "gcPercents": qc.gcPercents[:rl],
            "percents": {b: qc.percents[b][:rl] for b in "ATCG"},
            "baseMeanQual": {b: qc.baseMeanQual[b][:rl] for b in "ATCG"},
            "meanDiscontinuity": qc.meanDiscontinuity[:rl], "autoTrim": trimv})
    out["qc"] = qcs

    # isInBubble
    opt = type("O", (), {})()
    opt.read2_file = None; opt.index1_file = None
    sf = preprocesser.seqFilter(opt)
    circles = [(5000.0, 6000.0, 1500.0, 1, 101), (12000.5, 9000.25, 2500.75, 2, 101), (8000.0, 8000.0, 3000.0, 3, 205)]
    for c in circles:
        if c[4] not in sf.bubbleTiles:
            sf.bubbleTiles.append(c[4]); sf.bubbleCircles[c[4]] = []
        sf.bubbleCircles[c[4]].append(c)
    bub = []
    names = ["@SIM:1:FC1:1:1101:5000:7499 1:N:0:ACGT", "@SIM:1:FC1:1:1101:5000:7500 1:N:0:ACGT",
             "@SIM:1:FC1:2:1101:5000:6000 1:N:0:ACGT", "@noformat", "@a:1:b:1:1101:5000 x", "@x y:1:FC:3:2205:8000:8000",
             "@SIM:1:FC1:3:2205:10999:8000 1:N", "@SIM:1:FC1:3:12205:8000:8000 1:N", "@SIM:1:FC1:3:205:8000:8000"]
    for _ in range(300):
        names.append("@SIM:1:FC1:%d:%d:%d:%d 1:N:0:ACGT" % (rng.randint(1, 3), rng.choice([1101, 2205, 1102]),
                                                            rng.randint(1000, 16000), rng.randint(1000, 12000)))
    for nm in names:
        bub.append([nm, bool(sf.isInBubble(nm))])
    out["bubble"] = {"circles": circles, "names": bub}
    return out


# --------------------------------------------------------------------------------------------
# record framing (fastq.Reader.nextRead, fastq.py:37-49) on awkward texts
# --------------------------------------------------------------------------------------------
def framing_texts():
    import random
    rng = random.Random(77)
    rec = lambda i, L=12, name_extra="": "@r%d%s\n%s\n+\n%s\n" % (i, name_extra, rand_seq(rng, L), "I" * L)
    base = "".join(rec(i, rng.randint(5, 40)) for i in range(6))
    texts = {
        "plain": base,
        "crlf": base.replace("\n", "\r\n"),
        "trailing_ws": "@a x \t\nACGTACGT \n+ \nIIIIIIII\t \n@b\nAC\n+\nII\n",
        "no_final_newline": base[:-1],
        "no_final_newline_ws": base[:-1] + "  ",
        "partial_record": base + "@tail\nACGT\n",
        "partial_one_line": base + "@tail",
        "empty_line_mid": rec(0) + rec(1) + "\n" + rec(2) + rec(3),
        "ws_only_line_mid": rec(0) + "@x\nACGT\n \t \nIIII\n" + rec(2),
        "empty_seq_line": rec(0) + "@x\n\n+\n\n" + rec(2),
        "empty_first_line": "\n" + base,
        "empty_file": "",
        "only_newlines": "\n\n\n\n",
        "one_record": rec(0),
        "long_lines": rec(0, 300) + rec(1, 1000) + rec(2, 65) + rec(3, 64) + rec(4, 63),
        "name_with_spaces": rec(0, 10, " 1:N:0:ACGT extra  words") + rec(1, 10, "\tTAB"),
        "plus_with_name": "@a\nACGT\n+a comment\nIIII\n@b\nAC\n+\nII\n",
        "vt_ff_tail": "@a\x0b\nACGT\x0c\n+\nIIII\n",
        "inner_ws_kept": "@a b\nAC GT\n+\nII II\n",
        "at_in_quality": "@a\nACGT\n+\n@@@@\n@b\nACGT\n+\n@III\n",
        "five_lines": rec(0) + "@x\nACGT\n+\n",
        "many": "".join(rec(i, rng.randint(5, 120)) for i in range(300)),
    }
    return texts


def gen_text_vectors():
    install_shim()
    import fastq as ref_fastq
    out = {}
    work = tempfile.mkdtemp(prefix="aqc_txt_")
    for name, text in framing_texts().items():
        path = os.path.join(work, name + ".fq")
        with open(path, "w", newline="") as f:
            f.write(text)
        r = ref_fastq.Reader(path)
        recs = []
        while True:
            rec = r.nextRead()
            if rec is None:
                break
            recs.append(list(rec))
        out[name] = {"text": text, "records": recs}
    shutil.rmtree(work)
    return out


# --------------------------------------------------------------------------------------------
# end-to-end cases (G1, G3): the table lives in cases.py so the tests rebuild the same inputs
# --------------------------------------------------------------------------------------------
# cases whose HTML report (qcreporter.py + qualitycontrol.py:158-322 figures) is kept as a fixture
REPORT_CASES = ("g1_testdata", "se_default", "pe_barcode", "pe_cfg5")


def e2e_cases():
    import cases
    return [(name, argv, (lambda work, spec=spec: cases.materialize(spec, work)), keep)
            for (name, argv, spec, keep) in cases.CASES]


def main():
    if not os.path.isdir(REF):
        print("reference not present at %s: golden vectors can only be regenerated in the build container" % REF)
        sys.exit(2)
    if len(sys.argv) > 1 and sys.argv[1] == "--run-ref":
        run_ref_main(sys.argv[2:])
        return
    only = set(sys.argv[1:])
    # the reference's own test data files are fixtures (data, not source)
    os.makedirs(os.path.join(HERE, "testdata"), exist_ok=True)
    for fn in ("R1.fq.gz", "R2.fq.gz"):
        shutil.copy(os.path.join(REF, "testdata", fn), os.path.join(HERE, "testdata", fn))
    if not only or "func" in only:
        vec = gen_function_vectors()
        with gzip.open(os.path.join(HERE, "function_vectors.json.gz"), "wt") as f:
            json.dump(vec, f)
        print("function vectors:", {k: len(v) for k, v in vec.items()})
    if not only or "text" in only:
        vec = gen_text_vectors()
        with gzip.open(os.path.join(HERE, "text_vectors.json.gz"), "wt") as f:
            json.dump(vec, f, sort_keys=True)
        print("framing vectors:", {k: len(v["records"]) for k, v in vec.items()})
    if not only or "e2e" in only or any(o.startswith("case:") for o in only):
        path = os.path.join(HERE, "e2e_cases.json.gz")
        rpath = os.path.join(HERE, "report_vectors.json.gz")
        recs, reports = {}, {}
        if os.path.exists(path):
            with gzip.open(path, "rt") as f:
                recs = json.load(f)
        if os.path.exists(rpath):
            with gzip.open(rpath, "rt") as f:
                reports = json.load(f)
        for name, argv, setup, keep in e2e_cases():
            if any(o.startswith("case:") for o in only) and ("case:" + name) not in only:
                continue
            recs[name] = run_case(name, argv, setup, keep)
            rep = recs[name].pop("report", None)
            if name in REPORT_CASES:
                if rep is None or not rep["figures"]:
                    raise RuntimeError("the reference wrote no report for " + name)
                reports[name] = rep
            s = recs[name]["stat"]["afterqc_main_summary"]
            print(name, {k: s[k] for k in ("total_reads", "good_reads", "bad_reads")}, "report figures:", len(rep["figures"]) if rep else 0)
        with gzip.open(path, "wt") as f:
            json.dump(recs, f, sort_keys=True)
        with gzip.open(rpath, "wt") as f:
            json.dump(reports, f, sort_keys=True)


if __name__ == "__main__":
    main()
