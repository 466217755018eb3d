"""Host half of the reference's QualityControl objects (qualitycontrol.py:31-408).

The per-read accumulation (statRead, qualitycontrol.py:73-122) runs on the GPU (aqc_qc_stat).
This module only
  * decides WHICH reads are stat'd — statFile's sampling policy (qualitycontrol.py:331-357),
  * pulls the int64 accumulators + k-mer dictionary back through the C ABI, and
  * derives the float statistics (qualitycontrol.py:124-156) and the auto-trim decision
    (qualitycontrol.py:359-408) with numpy float64 arithmetic.  int -> float64 conversion and one
    IEEE division per value are exactly what `float(a)/float(b)` does in the reference, so every
    derived number is bit-identical.
The Plotly string emitters (qualitycontrol.py:158-322) are report rendering: out of scope.
"""
import numpy as np

from . import capi

MAX_LEN = 1000                      # qualitycontrol.py:23
ALL_BASES = ("A", "T", "C", "G")    # qualitycontrol.py:24 (row order of the device accumulators)
READ_TO_SKIP = 1000                 # qualitycontrol.py:333

# thresholds of isAbnormalCycle (qualitycontrol.py:391-395)
_BASE_TOP, _BASE_BOTTOM, _GC_TOP, _GC_BOTTOM, _QUAL_BOTTOM = 0.4, 0.15, 0.7, 0.3, 20.0


class _KmerView:
    """(k-mer, count) pairs in sortKmer order without materialising 4^k Python strings"""

    def __init__(self, qc, order):
        self.qc, self.order = qc, order

    def __len__(self):
        return len(self.order)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [(self.qc._kmer_str(int(j)), int(self.qc._kmer_cnt[j])) for j in self.order[i]]
        j = int(self.order[i])
        return (self.qc._kmer_str(j), int(self.qc._kmer_cnt[j]))

    def __iter__(self):
        return iter(self[:])


class QualityControl:
    """One of the four QC objects of preprocesser.py:247-254, backed by a device accumulator block."""

    def __init__(self, qc_sample, qc_kmer, engine, which):
        self.engine = engine
        self.which = which
        self.sampleLimit = qc_sample
        self.kmerLen = qc_kmer
        self.readCount = 0
        self.readLen = 0
        self.acc = np.zeros((capi.QC_ROWS, capi.AQC_QC_COLS), dtype=np.int64)
        self._kmer_raw = b""            # k-mer keys (8 bytes each) in dictionary (= first insertion) order
        self._kmer_cnt = np.zeros(0, dtype=np.int64)
        self._kmer_sorted = None

    # ---- device -> host -------------------------------------------------------------------------
    def pull(self):
        self.acc = self.engine.qc(self.which)
        keys, counts, order = self.engine.kmers(self.which)
        rank = np.argsort(order, kind="stable")
        # insertion order of the reference's dict under py3 (ties: App. B-12); strings are only built on demand
        self._kmer_raw = np.ascontiguousarray(keys[rank]).astype("<u8").tobytes()
        self._kmer_cnt = np.ascontiguousarray(counts[rank])
        self._kmer_sorted = None

    def _kmer_str(self, j):
        return self._kmer_raw[8 * j:8 * j + self.kmerLen].decode("latin-1")

    @property
    def kmerCount(self):
        """the reference's kmerCount dict (insertion-ordered)"""
        cnt = self._kmer_cnt.tolist()
        return {self._kmer_str(j): cnt[j] for j in range(len(cnt))}

    def kmer_pairs(self, indices):
        """(forward counts, reverse-complement counts) of the k-mers at positions `indices` of the count-sorted list — what
        strandBiasPlotly looks up one by one in the kmerCount dict (qualitycontrol.py:238-270: complement through COMP with 'N'
        for any other character, 0 for a reverse complement that was never seen); vectorised: the dict of all 4^k strings costs
        more to build than the rest of the report"""
        if self._kmer_sorted is None:
            self._kmer_sorted = np.argsort(-self._kmer_cnt, kind="stable")
        j = self._kmer_sorted[np.asarray(indices, dtype=np.int64)]
        k = self.kmerLen
        raw = np.frombuffer(self._kmer_raw, dtype=np.uint8).reshape(-1, 8)
        comp = np.full(256, ord("N"), dtype=np.uint8)
        for a, b in ("AT", "TA", "CG", "GC", "at", "ta", "cg", "gc", "NN"):
            comp[ord(a)] = ord(b)
        rc = np.zeros((len(j), 8), dtype=np.uint8)
        rc[:, :k] = comp[raw[j, :k]][:, ::-1]
        want = rc.view("<u8").ravel()
        keys = np.frombuffer(self._kmer_raw, dtype="<u8")
        order = np.argsort(keys, kind="stable")
        pos = np.minimum(np.searchsorted(keys[order], want), max(len(keys) - 1, 0))
        hit = keys[order][pos] == want if len(keys) else np.zeros(len(want), dtype=bool)
        rev = np.where(hit, self._kmer_cnt[order][pos] if len(keys) else 0, 0)
        return self._kmer_cnt[j].tolist(), [int(x) for x in rev]

    @property
    def topKmerCount(self):
        """sortKmer (qualitycontrol.py:155-156): stable, count-descending over dict order; a lazy sequence"""
        if self._kmer_sorted is None:
            self._kmer_sorted = np.argsort(-self._kmer_cnt, kind="stable")
        return _KmerView(self, self._kmer_sorted)

    # ---- derived statistics ----------------------------------------------------------------------
    def qc(self):
        """calcReadLen / calcPercents / calcQualities / calcDiscontinuity / sortKmer (qualitycontrol.py:324-329)"""
        if self.engine is not None:
            self.pull()
        a = self.acc
        counts = a[capi.QC_BASE_COUNT_A:capi.QC_BASE_COUNT_A + 4, :MAX_LEN]
        empty = np.flatnonzero(counts.sum(axis=0) == 0)
        # first cycle without any A/T/C/G (qualitycontrol.py:124-132); stays 0 if none is empty
        self.readLen = int(empty[0]) if len(empty) else 0
        n = self.readLen
        cnt = counts[:, :n].astype(np.float64)
        tot = counts[:, :n].sum(axis=0).astype(np.float64)
        self.percents = {b: cnt[i] / tot for i, b in enumerate(ALL_BASES)}
        gc_int = (counts[3, :n] + counts[2, :n]).astype(np.float64)   # G + C, summed as ints first
        self.gcPercents = gc_int / tot
        num = a[capi.QC_TOTAL_NUM, :n].astype(np.float64)
        self.meanQual = a[capi.QC_TOTAL_QUAL, :n].astype(np.float64) / num
        self.baseMeanQual = {}
        for i, b in enumerate(ALL_BASES):
            q = a[capi.QC_BASE_QUAL_A + i, :n].astype(np.float64)
            out = np.zeros(n, dtype=np.float64)
            np.divide(q, cnt[i], out=out, where=cnt[i] > 0)   # left at 0.0 where the base never occurs
            self.baseMeanQual[b] = out
        self.meanDiscontinuity = a[capi.QC_DISCONTINUITY, :n].astype(np.float64) / num
        self.totalKmer = int(a[capi.QC_SCALARS, 0])
        self._kmer_sorted = None        # sortKmer happens lazily (topKmerCount)

    def autoTrim(self):
        """qualitycontrol.py:359-408 as two masked scans outward from the centre cycle."""
        n = self.readLen
        centre = n // 2
        pct = np.stack([self.percents[b] for b in ALL_BASES]) if n else np.zeros((4, 0))
        bmq = np.stack([self.baseMeanQual[b] for b in ALL_BASES]) if n else np.zeros((4, 0))
        gc = self.gcPercents
        static_bad = (gc > _GC_TOP) | (gc < _GC_BOTTOM) | ((pct > _BASE_TOP) | (pct < _BASE_BOTTOM) | (bmq < _QUAL_BOTTOM)).any(axis=0)
        front_trim = tail_trim = 0
        if centre > 0:
            c = np.arange(0, centre)
            jump = (np.abs(pct[:, c] - pct[:, c + 1]) > 0.10).any(axis=0)
            hit = np.flatnonzero(static_bad[c] | jump)
            if len(hit):
                front_trim = int(hit[-1]) + 1            # scanning centre-1 .. 0, first abnormal cycle
        if centre + 1 < n:
            c = np.arange(centre + 1, n)
            jump = (np.abs(pct[:, c] - pct[:, c - 1]) > 0.05).any(axis=0)
            hit = np.flatnonzero(static_bad[c] | jump)
            if len(hit):
                tail_trim = n - int(c[hit[0]])
        return (min(int(n * 0.1), front_trim), min(int(n * 0.05), tail_trim))

    # ---- JSON views (squeeze, qualitycontrol.py:59-71: everything cut to readLen) -------------------
    def json_base_quality(self):
        return {b: self.baseMeanQual[b].tolist() for b in ALL_BASES}

    def json_mean_quality(self):
        return self.meanQual.tolist()

    def json_base_content(self):
        return {b: self.percents[b].tolist() for b in ALL_BASES}

    def json_gc_content(self):
        return self.gcPercents.tolist()

    def json_top_kmers(self, top=10):
        return [[k, c] for k, c in self.topKmerCount[:top]]

    # ---- sampling policy ----------------------------------------------------------------------------
    def statFileText(self, filename, chunk_bytes, slot=0):
        """statFile (qualitycontrol.py:331-357) with the records framed on the device (aqc_frame): the host only reads
        the file into a page-locked buffer.  Same policy as statFile below.  `slot`: the engine slot to work in (the two
        files of a pair are sampled side by side, each in its own slot)."""
        from . import fastq
        eng = self.engine
        cap = max(int(chunk_bytes), 4 << 20)          # the first chunk must hold the 999 skipped reads (fallback below)
        f = fastq.open_binary(filename, sample=self.sampleLimit > 0)
        buf = eng.host_buffer(cap)
        lo = READ_TO_SKIP - 1
        hi = lo + self.sampleLimit if self.sampleLimit > 0 else None    # stat 0-based [lo, hi)
        stop = None if hi is None else hi + 1      # the read whose arrival triggers the break is still consumed
        head, head_n = None, 0
        seen = 0
        left = 0
        file_eof = False
        fill = cap                                  # bytes to have in the buffer before framing
        try:
            while stop is None or seen < stop:
                end = left
                while not file_eof and end < fill:
                    got = f.readinto(buf.view[end:fill])
                    if not got:
                        file_eof = True
                    else:
                        end += got
                info = eng.frame(slot, buf.array, end, file_eof, max_records=(2 ** 64 - 1) if stop is None else stop - seen,
                                 first_index=seen)
                n = int(info.n)
                if head is None:
                    head, head_n = buf.array[:int(info.consumed1)].copy(), n
                a = max(lo, seen)
                b = seen + n if hi is None else min(hi, seen + n)
                if b > a:
                    eng.qc_stat(slot, self.which, 0, a - seen, b - a, 0)
                    eng.sync(slot)
                seen += n
                if info.eof1 or (file_eof and int(info.avail1) == n):
                    break
                if n == 0 and int(info.avail1) == 0:
                    # not one record fits: double the buffer
                    nb = eng.host_buffer(cap * 2)
                    nb.array[:end] = buf.array[:end]
                    buf.free()
                    buf, cap, left = nb, cap * 2, end
                    fill = cap
                    continue
                left = end - int(info.consumed1)
                if left:
                    buf.array[:left] = buf.array[int(info.consumed1):end].copy()
                if stop is not None and n:
                    # only the sample is wanted: read on for about as many bytes as the records still missing take (a .gz is
                    # decoded for every byte asked for), not for another buffer-full
                    fill = min(cap, left + int(int(info.consumed1) / n * (stop - seen) * 1.1) + (256 << 10))
            self.readCount = seen
            if max(0, seen - lo) < READ_TO_SKIP and head is not None and min(lo, seen) > 0:
                if head_n < min(lo, seen):
                    raise RuntimeError("the first chunk must hold at least %d records" % lo)
                pad = np.zeros(len(head) + 64, dtype=np.uint8)
                pad[:len(head)] = head
                eng.frame(slot, pad, len(head), True, max_records=min(lo, seen), first_index=0)
                eng.qc_stat(slot, self.which, 0, 0, min(lo, seen), 0)
                eng.sync(slot)
        finally:
            f.close()
            buf.free()
        self.qc()

    def statFile(self, filename, open_reader, to_batch, batch_records):
        """statFile (qualitycontrol.py:331-357): reads #1..999 are skipped, the next `sampleLimit`
        reads are stat'd; when fewer than 1000 reads followed the skipped ones, the skipped reads are
        stat'd as well — afterwards, which fixes the k-mer dictionary's insertion order."""
        reader = open_reader(filename)
        lo = READ_TO_SKIP - 1
        hi = lo + self.sampleLimit if self.sampleLimit > 0 else None    # stat 0-based [lo, hi)
        stop = None if hi is None else hi + 1      # the read whose arrival triggers the break is still consumed
        head = None
        seen = 0
        while stop is None or seen < stop:
            rb = reader.next_batch(batch_records if stop is None else min(batch_records, stop - seen))
            if rb is None:
                break
            if head is None:
                head = rb
            a = max(lo, seen)
            b = seen + rb.n if hi is None else min(hi, seen + rb.n)
            if b > a:
                self.engine.upload(0, to_batch(rb))
                self.engine.qc_stat(0, self.which, 0, a - seen, b - a, 0)
                self.engine.sync(0)
            seen += rb.n
        reader.close()
        self.readCount = seen
        if max(0, seen - lo) < READ_TO_SKIP and head is not None and min(lo, seen) > 0:
            if head.n < min(lo, seen):
                raise RuntimeError("batch_records must be at least %d" % lo)
            self.engine.upload(0, to_batch(head))
            self.engine.qc_stat(0, self.which, 0, 0, min(lo, seen), 0)
            self.engine.sync(0)
        self.qc()
