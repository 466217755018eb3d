"""Pure-Python restatement of the reference's per-pair loop — TEST / MEASUREMENT INFRASTRUCTURE only.

Purpose ("CPU baseline beside it", item 2): the reference itself is CPython and cannot travel to the
GPU box, so the CPU baseline there is (a) the scalar C oracle and (b) this module, which performs the reference's work
the way the reference does it — Python strings, per-base Python loops — and therefore runs at the reference's kind of
speed.  It is a third, independent statement of the path; tests/test_pyloop.py checks it verdict for verdict against
the C oracle.  Plain paired-end / single-end path only (no barcode, no bubble).

Every function cites the reference lines it restates; nothing here is imported by the product.
"""

COMP = {"A": "T", "T": "A", "C": "G", "G": "C", "a": "t", "t": "a", "c": "g", "g": "c", "N": "N"}   # util.py:27
POLY_SYMBOLS = "ATCGatcgN"                                                                      # preprocesser.py:35
ALL_BASES = ("A", "T", "C", "G")                                                                # qualitycontrol.py:24
GOOD, BADTRIM1, BADTRIM2, BADLEN, BADPOL, BADLQC, BADNCT, BADDIFF, BADMISMATCH = 0, 3, 4, 6, 7, 8, 9, 10, 11
EDIT_FIX_R2, EDIT_FIX_R1, EDIT_MASK = 1, 2, 3


def reverse_complement(seq):
    """util.py:42-51: reversed, mapped through COMP, anything unknown becomes 'N'"""
    out = []
    for c in reversed(seq):
        out.append(COMP.get(c, "N"))
    return "".join(out)


def trim(seq, qual, front, tail):
    """preprocesser.py:19-28 (python slice semantics)"""
    if tail > 0:
        return seq[front:-tail], qual[front:-tail]
    return seq[front:], qual[front:]


def has_polyx(seq, max_poly, mismatch):
    """preprocesser.py:30-51"""
    if len(seq) < max_poly:
        return None
    count = dict((c, 0) for c in POLY_SYMBOLS)
    for x in range(len(seq)):
        c = seq[x]
        if c not in count:
            return None
        count[c] += 1
        if x >= max_poly:
            count[seq[x - max_poly]] -= 1
        if count[c] >= max_poly - mismatch:
            return c
    return None


def low_quality_num(qual, q):
    """preprocesser.py:61-68"""
    n = 0
    for c in qual:
        if ord(c) < q + 33:
            n += 1
    return n


def n_number(seq):
    """preprocesser.py:70-76"""
    n = 0
    for c in seq:
        if c == "N":
            n += 1
    return n


def _diagonal_ok(a, i0, b, j0, n):
    """One diagonal of util.overlap_hm in closed form (checked there against the imported reference on
    6,000 random pairs): with `tot` mismatches among a[i0 : i0 + n] vs b[j0 : j0 + n] and `c50` of them in the first 50
    columns, the diagonal is accepted iff tot < 3, or c50 < 3 and n >= 52 (the third mismatch must not fall into columns
    0..49, and the scan must get past column 50).  Returns (accepted, tot) — tot is only complete when accepted."""
    tot = c50 = 0
    for k in range(n):
        if a[i0 + k] != b[j0 + k]:
            tot += 1
            if k < 50:
                c50 += 1
                if c50 >= 3:
                    return False, tot        # no diagonal with three early mismatches is ever accepted
    return tot < 3 or n >= 52, tot


def overlap_hm(r1, r2):
    """util.overlap (util.py:88-89,158-212) restated from its closed form: the forward diagonals d = 0, 1, ... (read 1 from d
    on against reverse_r2 from its start) in order, then the reverse ones (read 1 from its start against reverse_r2 from
    |d| on), each while at least 31 columns remain; the first accepted one wins.  -> (offset, overlap_len, diff)"""
    rr2 = reverse_complement(r2)
    n1, n2 = len(r1), len(r2)
    for d in range(max(0, n1 - 30)):
        n = min(n1 - d, n2)
        ok, tot = _diagonal_ok(r1, d, rr2, 0, n)
        if ok:
            return d, n, tot
    for d in range(max(0, n2 - 30)):
        n = min(n1, n2 - d)
        ok, tot = _diagonal_ok(r1, 0, rr2, d, n)
        if ok:
            return -d, n, tot
    return 0, 0, 0


def change_string(s, pos, val):
    """util.py:236-239: list assignment — a negative position counts from the end, one outside the string raises IndexError"""
    lst = list(s)
    lst[pos] = val
    return "".join(lst)


def process_pair(s1, q1, s2, q2, opt):
    """preprocesser.py:436-631 for one record without barcode / bubble.  `opt`: dict with the option names of the CLI.
    Returns dict(flag, seq1, qual1, seq2, qual2, offset, overlap_len, distance, edits=[(o, kind, base, qual)])"""
    paired = s2 is not None
    res = {"flag": GOOD, "offset": 0, "overlap_len": 0, "distance": 0, "edits": []}

    def done(flag):
        res.update(flag=flag, seq1=s1, qual1=q1, seq2=s2, qual2=q2)
        return res

    if opt["trim_front"] > 0 or opt["trim_tail"] > 0:          # :455-466
        s1, q1 = trim(s1, q1, opt["trim_front"], opt["trim_tail"])
        if len(s1) < 5:
            return done(BADTRIM1)
        if paired:
            s2, q2 = trim(s2, q2, opt["trim_front2"], opt["trim_tail2"])
            if len(s2) < 5:
                return done(BADTRIM2)
    if len(s1) < opt["seq_len_req"]:                           # :476-479
        return done(BADLEN)
    if opt["poly_size_limit"] > 0:                             # :482-490
        p1 = has_polyx(s1, opt["poly_size_limit"], opt["allow_mismatch_in_poly"])
        p2 = has_polyx(s2, opt["poly_size_limit"], opt["allow_mismatch_in_poly"]) if paired else None
        if p1 is not None or p2 is not None:
            return done(BADPOL)
    if opt["unqualified_base_limit"] > 0:                      # :493-501 (read 1 only, upstream quirk)
        if low_quality_num(q1, opt["qualified_quality_phred"]) > opt["unqualified_base_limit"]:
            return done(BADLQC)
    if opt["n_base_limit"] > 0:                                # :504-512
        if n_number(s1) > opt["n_base_limit"] or (paired and n_number(s2) > opt["n_base_limit"]):
            return done(BADNCT)
    if paired and not opt["no_overlap"]:                       # :515-617
        offset, overlap_len, distance = overlap_hm(s1, s2)
        res["first_overlap_len"] = overlap_len
        if offset < 0 and overlap_len > 30:                    # adapter read-through
            s1, q1, s2, q2 = s1[0:overlap_len], q1[0:overlap_len], s2[0:overlap_len], q2[0:overlap_len]
            res["adapter_bases"] = 2 * abs(offset)
            if len(s1) < opt["seq_len_req"]:
                return done(BADLEN)
            offset, overlap_len, distance = overlap_hm(s1, s2)
        res.update(offset=offset, overlap_len=overlap_len, distance=distance)
        if distance > 3:
            return done(BADDIFF)
        if overlap_len > 30 and distance > 0:
            corrected = masked = skipped = 0
            for o in range(overlap_len):
                b1 = s1[len(s1) - overlap_len + o]
                b2 = COMP[s2[-o - 1]]                          # util.complement: KeyError on anything else
                qa, qb = q1[len(q1) - overlap_len + o], q2[-o - 1]
                if b1 != b2:
                    fixed = False
                    if ord(qa) - 33 >= 30 and ord(qb) - 33 <= 14:
                        if not opt["no_correction"]:
                            # (:578-579: EACH string by its own end — they differ when the quality line is not as long as the
                            #  sequence line, which the reference never checks)
                            s2 = change_string(s2, -o - 1, COMP[b1])
                            q2 = change_string(q2, -o - 1, qa)
                            res["edits"].append((o, EDIT_FIX_R2, COMP[b1], qa))
                            corrected += 1
                            fixed = True
                    elif ord(qb) - 33 >= 30 and ord(qa) - 33 <= 14:
                        if not opt["no_correction"]:
                            s1 = change_string(s1, len(s1) - overlap_len + o, b2)           # :586-587
                            q1 = change_string(q1, len(q1) - overlap_len + o, qb)
                            res["edits"].append((o, EDIT_FIX_R1, b2, qb))
                            corrected += 1
                            fixed = True
                    if not fixed:
                        if opt["mask_mismatch"]:
                            q2 = change_string(q2, -o - 1, "!")                                # :594-595
                            q1 = change_string(q1, len(q1) - overlap_len + o, "!")
                            res["edits"].append((o, EDIT_MASK, "\0", "!"))
                            masked += 1
                        else:
                            skipped += 1
                    if corrected + masked + skipped >= distance:
                        break
            if corrected + masked + skipped != distance:
                return done(BADMISMATCH)
    return done(GOOD)


class CycleQC:
    """qualitycontrol.py:73-156 without the k-mer dictionary: the per-cycle accumulators of statRead and what qc() derives from
    them (the lists the stats JSON carries, squeezed to the read length, preprocesser.py:703-740)."""
    MAX_LEN = 1000                                                                              # qualitycontrol.py:20

    def __init__(self):
        n = self.MAX_LEN
        self.total_num = [0] * n
        self.total_qual = [0] * n
        self.base_counts = dict((b, [0] * n) for b in ALL_BASES)
        self.base_total_qual = dict((b, [0] * n) for b in ALL_BASES)
        self.total_discontinuity = [0] * n

    def stat_read(self, seq, qual):
        """:73-110.  The position is COUNTED before the quality is looked at; a quality line that has no character there
        (IndexError, swallowed at :84) skips everything else of the position — quality sum, base count, discontinuity."""
        seqlen = len(seq)
        for i in range(seqlen):
            self.total_num[i] += 1
            if i >= len(qual):
                continue
            qnum = ord(qual[i]) - 33
            self.total_qual[i] += qnum
            b = seq[i]
            if b in ALL_BASES:
                self.base_counts[b][i] += 1
                self.base_total_qual[b][i] += qnum
            left, right = i - 2, i + 3
            if left < 0:
                left, right = 0, 5
            elif right >= seqlen:
                right, left = seqlen, seqlen - 5
            self.total_discontinuity[i] += sum(1 for j in range(left, right - 1) if seq[j] != seq[j + 1])

    def derived(self):
        """:124-156 + squeeze(): read length = first cycle without any A/T/C/G; then the four lists of the stats JSON"""
        read_len = 0
        for pos in range(self.MAX_LEN):
            if not any(self.base_counts[b][pos] > 0 for b in ALL_BASES):
                read_len = pos
                break
        percents = dict((b, [0.0] * read_len) for b in ALL_BASES)
        gc = [0.0] * read_len
        mean_qual = [0.0] * read_len
        base_mean_qual = dict((b, [0.0] * read_len) for b in ALL_BASES)
        for pos in range(read_len):
            total = sum(self.base_counts[b][pos] for b in ALL_BASES)
            for b in ALL_BASES:
                percents[b][pos] = float(self.base_counts[b][pos]) / float(total)
            gc[pos] = float(self.base_counts["G"][pos] + self.base_counts["C"][pos]) / float(total)
            mean_qual[pos] = float(self.total_qual[pos]) / float(self.total_num[pos])
            for b in ALL_BASES:
                if self.base_counts[b][pos] > 0:
                    base_mean_qual[b][pos] = float(self.base_total_qual[b][pos]) / float(self.base_counts[b][pos])
        return {"readlen": read_len, "base_content": percents, "gc_content": gc, "mean_quality": mean_qual, "base_quality": base_mean_qual}


def options_from_config(cfg):
    """the fields of afterqc_amd.capi.Config this restatement consults"""
    return dict((k, int(getattr(cfg, k))) for k in (
        "trim_front", "trim_tail", "trim_front2", "trim_tail2", "seq_len_req", "poly_size_limit", "allow_mismatch_in_poly",
        "qualified_quality_phred", "unqualified_base_limit", "n_base_limit", "no_overlap", "no_correction", "mask_mismatch"))


def run_batch(batch, cfg, first=0, count=None):
    """records [first, first+count) of an afterqc_amd.capi.Batch -> list of result dicts"""
    opt = options_from_config(cfg)
    paired = bool(cfg.paired)
    out = []
    hi = batch.n if count is None else min(batch.n, first + count)
    for i in range(first, hi):
        s1, q1 = batch.read1(i)
        if paired:
            s2, q2 = batch.read2(i)
            out.append(process_pair(s1.decode("latin-1"), q1.decode("latin-1"), s2.decode("latin-1"), q2.decode("latin-1"), opt))
        else:
            out.append(process_pair(s1.decode("latin-1"), q1.decode("latin-1"), None, None, opt))
    return out
