"""ctypes front end of oracle/liboracle.so — TEST INFRASTRUCTURE (checker, never the product path).

`OracleEngine` offers the same methods as afterqc_amd.capi.Engine (the C-ABI wrapper), so the host
driver and the tests can run the identical call sequence against either and compare bit for bit.
Function-level helpers mirror the reference's Python signatures (SURVEY.md §8b "Python function seams").
"""
import ctypes as C
import os
import subprocess

import numpy as np

from afterqc_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "aqc_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        P = C.c_void_p
        L.orc_overlap_hm.argtypes = [P, C.c_int, P, C.c_int, P, P, P]
        L.orc_overlap_hm.restype = None
        L.orc_has_polyx.argtypes = [P, C.c_int, C.c_int, C.c_int]
        L.orc_low_quality_num.argtypes = [P, C.c_int, C.c_int]
        L.orc_n_number.argtypes = [P, C.c_int]
        L.orc_edit_distance.argtypes = [P, C.c_int, P, C.c_int]
        L.orc_detect_barcode.argtypes = [P, C.c_int, C.c_int, P, C.c_int]
        L.orc_clean_barcode_tail.argtypes = [P, C.c_int, P, C.c_int, P, C.c_int, P, C.c_int]
        L.orc_in_bubble.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, C.c_int]
        L.orc_process_batch.argtypes = [C.POINTER(capi.Config), C.POINTER(capi.BatchStruct), P, P, P, P, P, C.c_int32,
                                        P, P, P, P, C.c_uint64]
        L.orc_qc_new.argtypes = [C.c_int]
        L.orc_qc_new.restype = P
        L.orc_qc_free.argtypes = [P]
        L.orc_qc_free.restype = None
        L.orc_qc_stat_read.argtypes = [P, P, P, C.c_int, C.c_uint64]
        L.orc_qc_get.argtypes = [P, P]
        L.orc_qc_get.restype = None
        L.orc_qc_kmer_count.argtypes = [P]
        L.orc_qc_kmer_count.restype = C.c_uint64
        L.orc_qc_get_kmers.argtypes = [P, P, P, P, C.c_uint64]
        L.orc_qc_get_kmers.restype = C.c_uint64
        _lib = L
    return _lib


def _b(s):
    return s.encode("latin-1") if isinstance(s, str) else bytes(s)


# ---- reference-signature function seams ----------------------------------------------------------
def overlap(r1, r2):
    """util.overlap(r1, r2) -> (offset, overlap_len, diff)"""
    a, b = _b(r1), _b(r2)
    o, l, d = C.c_int(), C.c_int(), C.c_int()
    lib().orc_overlap_hm(a, len(a), b, len(b), C.byref(o), C.byref(l), C.byref(d))
    return (o.value, l.value, d.value)


def hasPolyX(seq, maxPoly, mismatch):
    s = _b(seq)
    r = lib().orc_has_polyx(s, len(s), maxPoly, mismatch)
    return None if r == 0 else chr(r)


def lowQualityNum(read, qual):
    q = _b(read[3])
    return lib().orc_low_quality_num(q, len(q), qual)


def nNumber(read):
    s = _b(read[1])
    return lib().orc_n_number(s, len(s))


def editDistance(s1, s2):
    a, b = _b(s1), _b(s2)
    return lib().orc_edit_distance(a, len(a), b, len(b))


def detectBarcode(seq, barcodeLen, verify):
    s, v = _b(seq), _b(verify)
    return lib().orc_detect_barcode(s, len(s), barcodeLen, v, len(v))


def isInBubble(lane, tile, x, y, circles):
    n = len(circles)
    cx = np.array([c[0] for c in circles], dtype=np.float64); cy = np.array([c[1] for c in circles], dtype=np.float64)
    cr = np.array([c[2] for c in circles], dtype=np.float64); ln = np.array([c[3] for c in circles], dtype=np.int32)
    tl = np.array([c[4] for c in circles], dtype=np.int32)
    return bool(lib().orc_in_bubble(lane, tile, x, y, cx.ctypes.data, cy.ctypes.data, cr.ctypes.data, ln.ctypes.data,
                                    tl.ctypes.data, n))


class OracleQC:
    """QualityControl accumulators + k-mer dict (qualitycontrol.py:33-122)."""

    def __init__(self, kmer_len=8):
        self.k = kmer_len
        self.h = lib().orc_qc_new(kmer_len)
        self._t = 0

    def __del__(self):
        try:
            lib().orc_qc_free(self.h)
        except Exception:
            pass

    def statRead(self, seq, qual, t0=None):
        """t0: global scan time of the read's first base (see orc_qc_stat_read); default: sequential"""
        s, q = _b(seq), _b(qual)
        if t0 is None:
            t0 = self._t
            self._t += 1024
        rc = lib().orc_qc_stat_read(self.h, s, q, len(s), t0)
        if rc != 0:
            raise capi.AqcError(rc, "oracle statRead")

    def acc(self):
        out = np.zeros((capi.QC_ROWS, capi.AQC_QC_COLS), dtype=np.int64)
        lib().orc_qc_get(self.h, out.ctypes.data)
        return out

    def kmers(self, with_order=False):
        """[(kmer bytes, count)] in dict insertion order (optionally with the time key of the insertion)"""
        n = lib().orc_qc_kmer_count(self.h)
        keys = np.zeros(max(1, n) * self.k, dtype=np.uint8)
        counts = np.zeros(max(1, n), dtype=np.int64)
        orders = np.zeros(max(1, n), dtype=np.uint64)
        lib().orc_qc_get_kmers(self.h, keys.ctypes.data, counts.ctypes.data, orders.ctypes.data, n)
        if with_order:
            return [(keys[i * self.k:(i + 1) * self.k].tobytes(), int(counts[i]), int(orders[i])) for i in range(n)]
        return [(keys[i * self.k:(i + 1) * self.k].tobytes(), int(counts[i])) for i in range(n)]


class OracleEngine:
    """Drop-in for afterqc_amd.capi.Engine backed by the C restatement (CPU, scalar, 1 thread)."""

    def __init__(self, device=0, n_slots=2):
        lib()
        self.n_slots = n_slots
        self.slots = [None] * n_slots
        self.results = [None] * n_slots
        self.cfg = None
        self.circles = []
        self.reset_stats()

    def close(self):
        pass

    def device_name(self):
        return "oracle-cpu"

    def set_config(self, cfg):
        c = capi.Config()
        C.memmove(C.byref(c), C.byref(cfg), C.sizeof(capi.Config))
        self.cfg = c

    def set_circles(self, circles):
        self.circles = list(circles)

    def reset_stats(self):
        self._qc_last_end = [0] * 4
        self._qc_epoch = [0] * 4
        self._counters = np.zeros(capi.N_COUNTERS, dtype=np.int64)
        self._ovl = np.zeros(capi.AQC_QC_COLS, dtype=np.int64)
        self._dist = np.zeros(capi.AQC_QC_COLS, dtype=np.int64)
        self._qc = [None] * 4

    def upload(self, slot, batch):
        self.slots[slot] = batch
        self.results[slot] = None

    def run(self, slot, accum_limit=capi.UINT64_MAX):
        b = self.slots[slot]
        res = np.zeros(b.n, dtype=capi.RESULT_DTYPE)
        s = b.as_struct()
        n = len(self.circles)
        cx = np.array([c[0] for c in self.circles], dtype=np.float64); cy = np.array([c[1] for c in self.circles], dtype=np.float64)
        cr = np.array([c[2] for c in self.circles], dtype=np.float64); ln = np.array([c[3] for c in self.circles], dtype=np.int32)
        tl = np.array([c[4] for c in self.circles], dtype=np.int32)
        rc = lib().orc_process_batch(C.byref(self.cfg), C.byref(s), cx.ctypes.data, cy.ctypes.data, cr.ctypes.data,
                                     ln.ctypes.data, tl.ctypes.data, n, res.ctypes.data, self._counters.ctypes.data,
                                     self._ovl.ctypes.data, self._dist.ctypes.data, accum_limit)
        if rc != 0:
            raise capi.AqcError(rc, "oracle process_batch")
        self.results[slot] = res

    def qc_stat(self, slot, which, mate, first, count, post):
        b = self.slots[slot]
        qc = self._get_qc(which)
        r2 = mate == 1
        res = self.results[slot]
        # same time keys as the device: epoch (bumped when a call goes back in the file) | global index | position
        g0 = b.first_index + first
        if g0 < self._qc_last_end[which]:
            self._qc_epoch[which] += 1
        self._qc_last_end[which] = g0 + count
        for i in range(first, min(first + count, b.n)):
            seq, qual = b.read2(i) if r2 else b.read1(i)
            if post:
                r = res[i]
                if r["flag"] != capi.GOOD:
                    continue
                seq, qual = final_read(seq, qual, r, 2 if r2 else 1)
            qc.statRead(seq, qual, (self._qc_epoch[which] << 44) | ((b.first_index + i) << 10))

    def fetch_results(self, slot):
        return self.results[slot].copy()

    def sync(self, slot):
        pass

    def kernel_ms(self, slot):
        return np.zeros(capi.N_KERNELS, dtype=np.float32)

    def counters(self):
        return self._counters.copy()

    def histograms(self, n=capi.AQC_QC_COLS):
        return self._ovl[:n].copy(), self._dist[:n].copy()

    def _get_qc(self, which):
        if self._qc[which] is None:
            self._qc[which] = OracleQC(self.cfg.qc_kmer if self.cfg is not None and self.cfg.qc_kmer > 0 else 8)
        return self._qc[which]

    def qc(self, which):
        return self._get_qc(which).acc()

    def kmers(self, which, cap=1 << 22):
        items = self._get_qc(which).kmers(with_order=True)
        keys = np.zeros(len(items), dtype=np.uint64)
        for i, (kb, _, _) in enumerate(items):
            keys[i] = int.from_bytes(kb.ljust(8, b"\0"), "little")
        counts = np.array([c for _, c, _ in items], dtype=np.int64)
        order = np.array([o for _, _, o in items], dtype=np.uint64)
        return keys, counts, order

    def overlap(self, batch):
        n = batch.n
        off = np.zeros(n, np.int32); ol = np.zeros(n, np.int32); df = np.zeros(n, np.int32)
        for i in range(n):
            off[i], ol[i], df[i] = overlap(batch.read1(i)[0], batch.read2(i)[0])
        return off, ol, df

    def read_stats(self, batch, max_poly, mismatch, qual):
        n = batch.n
        px = np.zeros(n, np.uint8); lq = np.zeros(n, np.int32); nn = np.zeros(n, np.int32)
        for i in range(n):
            s, q = batch.read1(i)
            p = hasPolyX(s, max_poly, mismatch)
            px[i] = 0 if p is None else ord(p)
            lq[i] = lowQualityNum([None, s, None, q], qual)
            nn[i] = nNumber([None, s])
        return px, lq, nn

    def edit_distance(self, batch):
        return np.array([editDistance(batch.read1(i)[0], batch.read2(i)[0]) for i in range(batch.n)], dtype=np.int32)


# ---- text in / text out: scalar restatement of fastq.Reader.nextRead and seqFilter.writeReads ------------------
FLAG_NAMES_B = [n.encode() for n in capi.FLAG_NAMES]


def frame_text(buf, final):
    """fastq.py:37-49 over one chunk of text.  Returns (records, avail, eof, line_ends) where records[r] = four
    (start, stripped_length) pairs, avail = complete records before the first empty line, eof = such a line was
    met, line_ends[i] = byte offset just behind line i (what a reader has consumed after readline() number i)."""
    buf = bytes(buf)
    lines, ends = [], []
    pos, n = 0, len(buf)
    while pos < n:
        nl = buf.find(b"\n", pos)
        if nl < 0:
            if not final:
                break                       # incomplete line: wait for more text
            end, nxt = n, n                 # unterminated last line of the file: readline() still returns it
        else:
            end, nxt = nl, nl + 1
        lines.append((pos, len(buf[pos:end].rstrip())))     # bytes.rstrip(): space \t \n \r \v \f
        ends.append(nxt)
        pos = nxt
    nrec = len(lines) // 4
    records = [lines[4 * r:4 * r + 4] for r in range(nrec)]
    avail, eof = nrec, False
    for r in range(nrec):
        if any(l == 0 for _, l in records[r]):              # fastq.py:44-47
            avail, eof = r, True
            break
    return records, avail, eof, ends


class _HostBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.array = np.zeros(self.nbytes, dtype=np.uint8)
        self.view = memoryview(self.array)

    def free(self):
        pass


def _oracle_frame(self, slot, text1, bytes1, final1, text2=None, bytes2=0, final2=False, max_records=capi.UINT64_MAX,
                  first_index=0):
    files = [(text1, bytes1, final1)] + ([(text2, bytes2, final2)] if text2 is not None else [])
    framed = []
    for text, nbytes, final in files:
        buf = np.asarray(text)[:nbytes].tobytes()
        framed.append((buf,) + frame_text(buf, final))
    n = min(f[2] for f in framed)
    n = min(n, max_records)
    info = capi.FrameInfo()
    info.n = n
    b = capi.Batch(n, first_index)
    lines = []
    mx = 0
    for k, (buf, records, avail, eof, ends) in enumerate(framed):
        arena = np.zeros(len(buf) + 64, dtype=np.uint8)
        arena[:len(buf)] = np.frombuffer(buf, dtype=np.uint8)
        so = np.array([records[r][1][0] for r in range(n)], dtype=np.uint64)
        sl = np.array([records[r][1][1] for r in range(n)], dtype=np.uint32)
        qo = np.array([records[r][3][0] for r in range(n)], dtype=np.uint64)
        ql = [records[r][3][1] for r in range(n)]
        if list(sl) != ql:
            raise capi.AqcError(-2, "malformed FASTQ: sequence and quality lines differ in length")
        for r in range(len(records)):
            mx = max(mx, records[r][1][1])
        consumed = min(ends[4 * n - 1], len(buf)) if n else 0
        if k == 0:
            b.seq1 = b.qual1 = arena
            b.off1, b.qoff1, b.len1 = so, qo, sl
            info.avail1, info.eof1, info.consumed1 = avail, int(eof), consumed
            info.next_len1 = records[n][1][1] if avail > n else 0
        else:
            b.seq2 = b.qual2 = arena
            b.off2, b.qoff2, b.len2 = so, qo, sl
            info.avail2, info.eof2, info.consumed2 = avail, int(eof), consumed
        lines.append((buf, records))
    info.max_len = mx
    if self.cfg is not None and self.cfg.debubble:
        # the host half of isInBubble (preprocesser.py:180-192) with Python's own regex engine
        from afterqc_amd import fastq
        buf, records = lines[0]
        try:
            aux = [fastq.parse_illumina_name(buf[records[r][0][0]:records[r][0][0] + records[r][0][1]]) for r in range(n)]
        except ValueError as e:
            raise capi.AqcError(-2, "read name: %s" % e)
        if n:
            ok, lane, tile, x, y = zip(*aux)
            b.set_aux(lane, tile, x, y, ok)
    self.upload(slot, b)
    self._text = getattr(self, "_text", {})
    self._text[slot] = lines
    self._fmt = getattr(self, "_fmt", {})
    self._fmt.pop(slot, None)
    return info


def _oracle_format(self, slot, n, store_overlap=False):
    """seqFilter.writeReads (preprocesser.py:206-232) + Writer.writeLines (fastq.py:87-93) + getOverlap (:78-84);
    index files aside.  Streams [file * 3 + stream], stream 0 good / 1 bad / 2 overlap"""
    res = self.results[slot]
    out = [bytearray() for _ in range(6)]
    paired = len(self._text[slot]) == 2
    for k, (buf, records) in enumerate(self._text[slot]):
        for r in range(n):
            (no, nl), (so, sl), (po, pl), (qo, ql) = records[r]
            name, plus = buf[no:no + nl], buf[po:po + pl]
            rr = res[r]
            seq, qual = final_read(buf[so:so + sl], buf[qo:qo + ql], rr, k + 1)
            flag = int(rr["flag"])
            if self.cfg.barcode and flag not in (capi.BADBCD1, capi.BADBCD2):
                # moveBarcodeToName (barcodeprocesser.py:34-45): detected length for pairs, design length single-end
                code = (int(rr["barcode"]) & 15) if k == 0 else (int(rr["barcode"]) >> 4)
                blen = code - 2 + self.cfg.barcode_length if self.cfg.paired else self.cfg.barcode_length
                name = b"@" + buf[so:so + sl][0:max(blen, 0)] + name[name.find(b":"):]
            ov, dist = int(rr["overlap_len"]), int(rr["distance"])
            if store_overlap and paired and flag == capi.GOOD and ov > 30:
                corrected = sum(1 for e in range(int(rr["n_edits"])) if int(rr["edits"][e]["kind"]) in (capi.EDIT_FIX_R1, capi.EDIT_FIX_R2))
                if dist == 0 or dist == corrected:               # preprocesser.py:614-616
                    out[3 * k + 2] += name + b"\n" + seq[len(seq) - ov:] + b"\n" + plus + b"\n" + qual[len(qual) - ov:] + b"\n"
            stream = 0 if flag == capi.GOOD else 1
            if stream:
                name = b"@" + FLAG_NAMES_B[flag] + name[1:]
            out[3 * k + stream] += name + b"\n" + seq + b"\n" + plus + b"\n" + qual + b"\n"
    self._fmt[slot] = [bytes(o) for o in out]
    return [len(o) for o in out]


def _oracle_format_plain(self, slot, verdict_slot, n, store_overlap=False):
    """writeReads for the index files (preprocesser.py:222-232,616): whole records, bad ones renamed, overlap copies"""
    res = self.results[verdict_slot]
    paired = self.slots[verdict_slot].seq2 is not None
    out = [bytearray() for _ in range(6)]
    for k, (buf, records) in enumerate(self._text[slot]):
        for r in range(n):
            rec = [buf[o:o + l] for (o, l) in records[r]]
            rr = res[r]
            flag = int(rr["flag"])
            ov, dist = int(rr["overlap_len"]), int(rr["distance"])
            if store_overlap and paired and flag == capi.GOOD and ov > 30:
                corrected = sum(1 for e in range(int(rr["n_edits"])) if int(rr["edits"][e]["kind"]) in (capi.EDIT_FIX_R1, capi.EDIT_FIX_R2))
                if dist == 0 or dist == corrected:
                    out[3 * k + 2] += b"\n".join(rec) + b"\n"
            stream = 0 if flag == capi.GOOD else 1
            if stream:
                rec[0] = b"@" + FLAG_NAMES_B[flag] + rec[0][1:]
            out[3 * k + stream] += b"\n".join(rec) + b"\n"
    self._fmt[slot] = [bytes(o) for o in out]
    return [len(o) for o in out]


def _oracle_fetch_text(self, slot, file, stream, dst, cap):
    data = self._fmt[slot][3 * file + stream]
    if len(data) > cap:
        raise capi.AqcError(-2, "fetch_text: destination too small")
    if data:
        dst[:len(data)] = np.frombuffer(data, dtype=np.uint8)


def final_read(seq, qual, r, which):
    """Apply a result record (trim extents + edits) to the original read -> final (seq, qual)."""
    if which == 1:
        st, ln = int(r["start1"]), int(r["len1"])
    else:
        st, ln = int(r["start2"]), int(r["len2"])
    s = bytearray(seq[st:st + ln]); q = bytearray(qual[st:st + ln])
    ov = int(r["overlap_len"])
    for k in range(int(r["n_edits"])):
        e = r["edits"][k]
        o = int(e["o"]); kind = int(e["kind"])
        p = (ln - ov + o) if which == 1 else (ln - 1 - o)
        if kind == capi.EDIT_MASK:
            q[p] = ord("!")
        elif (kind == capi.EDIT_FIX_R1 and which == 1) or (kind == capi.EDIT_FIX_R2 and which == 2):
            s[p] = int(e["base"]); q[p] = int(e["qual"])
    return bytes(s), bytes(q)


OracleEngine.host_buffer = lambda self, nbytes: _HostBuffer(nbytes)
OracleEngine.frame = _oracle_frame
OracleEngine.format = _oracle_format
OracleEngine.format_plain = _oracle_format_plain
OracleEngine.fetch_text = _oracle_fetch_text
