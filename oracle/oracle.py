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
