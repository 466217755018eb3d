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
        self.kmerCount = {}
        self.topKmerCount = []

    # ---- device -> host -------------------------------------------------------------------------
    def pull(self):
        self.acc = self.engine.qc(self.which)
        keys, counts, order = self.engine.kmers(self.which)
        rank = np.argsort(order, kind="stable")
        raw = np.ascontiguousarray(keys[rank]).astype("<u8").tobytes()
        cnt = counts[rank].tolist()
        k = self.kmerLen
        # python dicts keep insertion order: this IS the reference's dict under py3 (ties: App. B-12)
        self.kmerCount = {raw[8 * j:8 * j + k].decode("latin-1"): cnt[j] for j in range(len(cnt))}

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
        # sortKmer: stable, count-descending over dict order (qualitycontrol.py:155-156)
        self.topKmerCount = sorted(self.kmerCount.items(), key=lambda kv: -kv[1])

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
