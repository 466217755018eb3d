"""Per-file driver: the batch-pipelined counterpart of the reference's seqFilter
(preprocesser.py:141-783).  Same constructor / run() interface and the same outputs
(good/bad/overlap FASTQ, QC/<R1 basename>.json), but the per-read loop of preprocesser.py:411-631
is not executed here: records are framed in bulk (afterqc_amd.fastq), shipped to the GPU as SoA
batches, judged by the HIP kernels behind the C ABI (afterqc_amd.capi.Engine) and only the 32-byte
verdict records come back.  This module is I/O and bookkeeping:

  pass 1  pre-filter QC sampling + auto-trim            (preprocesser.py:247-280)
  pass 2  upload -> aqc_run -> aqc_qc_stat(post) -> fetch -> write good/bad/overlap
  stats   counters -> JSON with the reference's schema  (preprocesser.py:660-778)

There is no CPU compute path: `engine` defaults to the HIP engine, which raises if the library or
the GPU is missing.  (Tests inject the oracle engine to exercise THIS file's host logic on CPU.)
"""
import json
import os

import numpy as np

from . import capi, fastq
from .qc import QualityControl, ALL_BASES

FLAG_BYTES = [b"", b"BADBCD1", b"BADBCD2", b"BADTRIM1", b"BADTRIM2", b"BADBBL", b"BADLEN", b"BADPOL", b"BADLQC",
              b"BADNCT", b"BADDIFF", b"BADMISMATCH"]


def getMainName(filename):
    """preprocesser.py:14-17"""
    base = os.path.basename(filename)
    for ext in (".fastq", ".fq", ".gz"):
        base = base.replace(ext, "")
    return base


# the keys echoed into the JSON "command" block (makeDict, preprocesser.py:86-123)
COMMAND_KEYS = ("index2_flag", "draw", "barcode", "index1_flag", "seq_len_req", "index1_file",
                "overlap_output_folder", "trim_tail", "trim_pair_same", "poly_size_limit", "good_output_folder",
                "debubble_dir", "index2_file", "qualified_quality_phred", "barcode_flag", "trim_front",
                "barcode_verify", "read2_file", "n_base_limit", "barcode_length", "trim_tail2",
                "unqualified_base_limit", "allow_mismatch_in_poly", "input_dir", "read1_file", "read2_flag",
                "store_overlap", "debubble", "read1_flag", "trim_front2", "bad_output_folder", "qc_only", "qc_sample",
                "qc_kmer")


def makeDict(opt):
    return {k: getattr(opt, k) for k in COMMAND_KEYS}


def load_circles(debubble_dir):
    """loadBubbleCircles (preprocesser.py:157-174): rows of x,y,radius,lane,tile after a header line."""
    path = os.path.join(debubble_dir, "circles.csv")
    circles = []
    if not os.path.exists(path):
        return circles
    with open(path) as f:
        for row in f.readlines()[1:]:
            r = row.split(",")
            circles.append((float(r[0]), float(r[1]), float(r[2]), int(r[3]), int(r[4])))
    return circles


def build_config(opt, paired, has_index2):
    cfg = capi.Config()
    cfg.paired = 1 if paired else 0
    cfg.count_r2_bases = 1 if has_index2 else 0
    cfg.trim_front, cfg.trim_tail = max(opt.trim_front, 0), max(opt.trim_tail, 0)
    cfg.trim_front2, cfg.trim_tail2 = max(opt.trim_front2, 0), max(opt.trim_tail2, 0)
    cfg.seq_len_req = opt.seq_len_req
    cfg.poly_size_limit = opt.poly_size_limit
    cfg.allow_mismatch_in_poly = opt.allow_mismatch_in_poly
    cfg.qualified_quality_phred = opt.qualified_quality_phred
    cfg.unqualified_base_limit = opt.unqualified_base_limit
    cfg.n_base_limit = opt.n_base_limit
    cfg.no_overlap = 1 if opt.no_overlap else 0
    cfg.no_correction = 1 if opt.no_correction else 0
    cfg.mask_mismatch = 1 if opt.mask_mismatch else 0
    cfg.barcode = 1 if opt.barcode else 0
    cfg.barcode_length = opt.barcode_length
    cfg.set_verify(opt.barcode_verify)
    cfg.debubble = 1 if opt.debubble else 0
    cfg.qc_kmer = opt.qc_kmer
    return cfg


class _Outputs:
    """The good / bad / overlap writers of preprocesser.py:323-371 for up to four input files."""

    def __init__(self, opt, files, good_dir, bad_dir, overlap_dir, gzip_out, gzip_comp):
        self.good, self.bad, self.overlap = [], [], []
        store = opt.store_overlap and opt.read2_file is not None
        for f in files:
            if f is None or opt.qc_only:
                self.good.append(None); self.bad.append(None); self.overlap.append(None)
                continue
            main = getMainName(f)
            self.good.append(fastq.Writer(os.path.join(good_dir, main + ".good.fq"), gzip_out, gzip_comp))
            self.bad.append(fastq.Writer(os.path.join(bad_dir, main + ".bad.fq"), gzip_out, gzip_comp))
            # upstream opens the R1 overlap writer whenever store_overlap is on, the others only when paired
            if (opt.store_overlap and f == files[0]) or store:
                self.overlap.append(fastq.Writer(os.path.join(overlap_dir, main + ".overlap.fq"), gzip_out, gzip_comp))
            else:
                self.overlap.append(None)

    def close(self):
        for group in (self.good, self.bad, self.overlap):
            for w in group:
                if w is not None:
                    w.close()


class seqFilter:
    """seqFilter(options).run() — preprocesser.py:141-155,234-783."""

    def __init__(self, opt, engine=None, batch_records=1 << 20, device=0):
        self.options = opt
        self.engine = engine
        self.device = device
        self.own_engine = False
        self.batch_records = max(int(batch_records), 1000)
        self.paired = opt.read2_file is not None
        self.bubbleCircles = []
        self.stat = None

    # ---- helpers ---------------------------------------------------------------------------------
    def _engine(self):
        if self.engine is None:
            self.engine = capi.Engine(self.device, 2)
            self.own_engine = True
        return self.engine

    def _aux(self, batch, rb1):
        """Parse lane/tile/x/y out of the R1 names for the bubble filter (preprocesser.py:180-192)."""
        n = rb1.n
        ok = np.zeros(n, np.uint8); lane = np.zeros(n, np.int32); tile = np.zeros(n, np.int32)
        x = np.zeros(n, np.int32); y = np.zeros(n, np.int32)
        for i in range(n):
            ok[i], lane[i], tile[i], x[i], y[i] = fastq.parse_illumina_name(rb1.line("name", i))
        batch.set_aux(lane, tile, x, y, ok)

    # ---- the run -----------------------------------------------------------------------------------
    def run(self):
        opt = self.options
        eng = self._engine()
        paired = self.paired
        if opt.debubble:
            self.bubbleCircles = load_circles(opt.debubble_dir)
        # no front trim if the sequence is barcoded (preprocesser.py:242-243)
        if opt.barcode:
            opt.trim_front = 0

        has_i1 = opt.index1_file is not None
        has_i2 = opt.index2_file is not None
        eng.set_config(build_config(opt, paired, has_i2))
        eng.set_circles(self.bubbleCircles)
        eng.reset_stats()

        # ---- pass 1: pre-filter QC on a sample of each file (preprocesser.py:247-251)
        r1pre = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R1_PRE)
        r2pre = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R2_PRE)
        r1post = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R1_POST)
        r2post = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R2_POST)
        single = lambda rb: capi.Batch.from_raw(rb)
        r1pre.statFile(opt.read1_file, fastq.Reader, single, self.batch_records)
        if paired:
            # the R2 file is stat'd through the same single-read path into its own accumulator
            r2pre.statFile(opt.read2_file, fastq.Reader, single, self.batch_records)
        readLen = r1pre.readLen

        # ---- auto trim (preprocesser.py:261-280)
        if opt.trim_front == -1 or opt.trim_tail == -1:
            tf, tt = r1pre.autoTrim()
            if opt.trim_front == -1:
                opt.trim_front = tf
            if opt.trim_tail == -1:
                opt.trim_tail = tt
            if paired:
                if opt.trim_pair_same:
                    opt.trim_front2 = opt.trim_front
                    opt.trim_tail2 = opt.trim_tail
                else:
                    tf2, tt2 = r2pre.autoTrim()
                    if opt.trim_front2 == -1:
                        opt.trim_front2 = tf2
                    if opt.trim_tail2 == -1:
                        opt.trim_tail2 = tt2
        print(opt.read1_file + " options:")
        print(opt)

        # ---- output layout (preprocesser.py:285-321)
        good_dir = opt.good_output_folder
        if good_dir is None:
            good_dir = os.path.dirname(opt.read1_file)
        parent = os.path.dirname(os.path.dirname(good_dir + "/"))
        bad_dir = opt.bad_output_folder if opt.bad_output_folder is not None else os.path.join(parent, "bad")
        overlap_dir = opt.overlap_output_folder if opt.overlap_output_folder is not None else os.path.join(parent, "overlap")
        qc_dir = opt.report_output_folder if opt.report_output_folder is not None else os.path.join(parent, "QC")
        for d in (qc_dir, good_dir, bad_dir):
            if not os.path.exists(d):
                os.makedirs(d)
        if opt.store_overlap and paired and not os.path.exists(overlap_dir):
            os.makedirs(overlap_dir)
        gzip_out = bool(opt.gzip) or opt.read1_file.endswith(".gz")
        files = [opt.read1_file, opt.read2_file, opt.index1_file, opt.index2_file]
        if opt.store_overlap and not opt.qc_only and not os.path.exists(overlap_dir):
            os.makedirs(overlap_dir)   # single-end + store_overlap: upstream opens the writer without the dir
        outs = _Outputs(opt, files, good_dir, bad_dir, overlap_dir, gzip_out, opt.compression)

        # ---- pass 2: the main loop (preprocesser.py:411-631), one batch at a time
        # the per-read settings now include the resolved trim values
        eng.set_config(build_config(opt, paired, has_i2))
        readers = [fastq.Reader(f) if f is not None else None for f in files]
        total = 0          # TOTAL_READS so far
        extra_bases = 0    # R1 bases read for a record that a shorter mate file then cut off (:416-421)
        stop = False
        slot = 0
        while not stop:
            rbs = [r.next_batch(self.batch_records) if r is not None else None for r in readers]
            if rbs[0] is None:
                break
            # lock-step reading: the first file to run dry ends the loop (preprocesser.py:412-429); R1's
            # record was already counted into TOTAL_BASES by then (:416)
            n = min(rbs[k].n if rbs[k] is not None else 0 for k in range(4) if readers[k] is not None)
            if n < rbs[0].n:
                extra_bases = int(rbs[0].seq_len[n])
                stop = True
            if n == 0:
                break
            batch = capi.Batch.from_raw(rbs[0], rbs[1] if paired else None, first_index=total)
            if n < batch.n:
                batch.n = n
            if opt.debubble:
                self._aux(batch, rbs[0])
            eng.upload(slot, batch)
            # --qc_only stops at the first good record whose 1-based index reaches qc_sample (:630-631)
            limit = capi.UINT64_MAX
            results = None
            if opt.qc_only:
                eng.run(slot, 0)
                results = eng.fetch_results(slot)[:n]
                idx1 = total + 1 + np.arange(n)
                hit = np.flatnonzero((results["flag"] == capi.GOOD) & (idx1 >= opt.qc_sample))
                if len(hit):
                    n = int(hit[0]) + 1
                    stop = True
                limit = n
            eng.run(slot, limit)
            # post-filter QC on good records while TOTAL_READS < qc_sample (preprocesser.py:624-627)
            n_qc = n if opt.qc_sample <= 0 else max(0, min(n, opt.qc_sample - 1 - total))
            if n_qc > 0:
                eng.qc_stat(slot, capi.QC_R1_POST, 0, 0, n_qc, 1)
                if paired:
                    eng.qc_stat(slot, capi.QC_R2_POST, 1, 0, n_qc, 1)
            results = eng.fetch_results(slot)[:n]
            if not opt.qc_only:
                self._write(outs, rbs, results, n)
            total += n
        for r in readers:
            if r is not None:
                r.close()
        outs.close()

        r1post.qc()
        if paired:
            r2post.qc()

        self.stat = self._stats(eng, r1pre, r2pre, r1post, r2post, readLen, extra_bases)
        stat_path = os.path.join(qc_dir, os.path.basename(opt.read1_file) + ".json")
        with open(stat_path, "w") as f:
            f.write(json.dumps(self.stat, sort_keys=True, indent=4, separators=(',', ': ')))
        if self.own_engine:
            eng.close()
            self.engine = None
        return self.stat

    # ---- output formatting (writeReads, preprocesser.py:206-232; fastq.Writer.writeLines) ------------
    def _write(self, outs, rbs, results, n):
        opt = self.options
        paired = self.paired
        vlen = len(opt.barcode_verify)
        bl = opt.barcode_length
        store = opt.store_overlap
        chunks = [[[], [], []] for _ in range(4)]      # per file: good, bad, overlap byte pieces
        rb1, rb2, rbi1, rbi2 = rbs
        flags = results["flag"]
        for i in range(n):
            r = results[i]
            flag = int(flags[i])
            name1, s1, p1, q1 = rb1.record(i)
            if paired:
                name2, s2, p2, q2 = rb2.record(i)
            bc = int(r["barcode"])
            moved = opt.barcode and flag not in (capi.BADBCD1, capi.BADBCD2)
            if moved:
                # moveBarcodeToName (barcodeprocesser.py:34-45): '@' + barcode + name[first ':' :]
                # single-end moves the DESIGN length (preprocesser.py:444), paired the detected lengths (:452)
                b1 = (bc & 15) - 2 + bl if paired else bl
                name1 = b"@" + s1[0:b1] + name1[name1.find(b":"):]
                if paired:
                    b2 = (bc >> 4) - 2 + bl
                    name2 = b"@" + s2[0:b2] + name2[name2.find(b":"):]
            st, ln = int(r["start1"]), int(r["len1"])
            s1 = bytearray(s1[st:st + ln]); q1 = bytearray(q1[st:st + ln])
            if paired:
                st2, ln2 = int(r["start2"]), int(r["len2"])
                s2 = bytearray(s2[st2:st2 + ln2]); q2 = bytearray(q2[st2:st2 + ln2])
            ov = int(r["overlap_len"])
            ne = int(r["n_edits"])
            corrected = 0
            for k in range(ne):
                e = r["edits"][k]
                o = int(e["o"]); kind = int(e["kind"])
                if kind == capi.EDIT_FIX_R2:
                    s2[ln2 - 1 - o] = int(e["base"]); q2[ln2 - 1 - o] = int(e["qual"]); corrected += 1
                elif kind == capi.EDIT_FIX_R1:
                    s1[ln - ov + o] = int(e["base"]); q1[ln - ov + o] = int(e["qual"]); corrected += 1
                else:
                    q2[ln2 - 1 - o] = 33; q1[ln - ov + o] = 33
            which = 0 if flag == capi.GOOD else 1
            if which == 1:
                fb = FLAG_BYTES[flag]
                name1 = b"@" + fb + name1[1:]
                if paired:
                    name2 = b"@" + fb + name2[1:]
            if store and paired and not opt.no_overlap and flag == capi.GOOD and ov > 30:
                dist = int(r["distance"])
                if dist == 0 or dist == corrected:
                    # getOverlap (preprocesser.py:78-84): the last overlap_len bases of both reads
                    chunks[0][2].append(name1 + b"\n" + bytes(s1[ln - ov:]) + b"\n" + p1 + b"\n" + bytes(q1[ln - ov:]) + b"\n")
                    chunks[1][2].append(name2 + b"\n" + bytes(s2[ln2 - ov:]) + b"\n" + p2 + b"\n" + bytes(q2[ln2 - ov:]) + b"\n")
                    for k, rbx in ((2, rbi1), (3, rbi2)):
                        if rbx is not None:
                            chunks[k][2].append(b"\n".join(rbx.record(i)) + b"\n")
            chunks[0][which].append(name1 + b"\n" + bytes(s1) + b"\n" + p1 + b"\n" + bytes(q1) + b"\n")
            if paired:
                chunks[1][which].append(name2 + b"\n" + bytes(s2) + b"\n" + p2 + b"\n" + bytes(q2) + b"\n")
            for k, rbx in ((2, rbi1), (3, rbi2)):
                if rbx is not None:
                    rec = rbx.record(i)
                    if which == 1:
                        rec[0] = b"@" + FLAG_BYTES[flag] + rec[0][1:]
                    chunks[k][which].append(b"\n".join(rec) + b"\n")
        for k in range(4):
            for which, group in ((0, outs.good), (1, outs.bad), (2, outs.overlap)):
                if group[k] is not None and chunks[k][which]:
                    group[k].write_bytes(b"".join(chunks[k][which]))

    # ---- statistics -> JSON (preprocesser.py:660-778) ---------------------------------------------------
    def _stats(self, eng, r1pre, r2pre, r1post, r2post, readLen, extra_bases):
        opt = self.options
        C = eng.counters()
        c = lambda k: int(C[k])
        f = lambda k: int(C[capi.C_FLAG0 + k])
        total_reads = c(capi.C_TOTAL_READS)
        good_reads = c(capi.C_GOOD_READS)
        summary = {
            'total_bases': c(capi.C_TOTAL_BASES) + extra_bases,
            'good_bases': c(capi.C_GOOD_BASES),
            'total_reads': total_reads,
            'good_reads': good_reads,
            'bad_reads': total_reads - good_reads,
            'bad_reads_with_bad_barcode': f(capi.BADBCD1) + f(capi.BADBCD2),
            'bad_reads_with_reads_in_bubble': f(capi.BADBBL),
            'bad_reads_with_bad_read_length': f(capi.BADLEN) + f(capi.BADTRIM1) + f(capi.BADTRIM2),
            'bad_reads_with_polyX': f(capi.BADPOL),
            'bad_reads_with_low_quality': f(capi.BADLQC),
            'bad_reads_with_too_many_N': f(capi.BADNCT),
            'bad_reads_with_bad_overlap': f(capi.BADMISMATCH) + f(capi.BADDIFF),   # BADINDEL is always 0 upstream
            'readlen': readLen,
        }
        stat = {"afterqc_main_summary": summary, "command": makeDict(opt)}
        pairs = [("read1_prefilter", r1pre), ("read1_postfilter", r1post)]
        if self.paired:
            pairs += [("read2_prefilter", r2pre), ("read2_postfilter", r2post)]
        stat["kmer_content"] = {k: q.json_top_kmers(10) for k, q in pairs}
        stat["base_quality"] = {k: q.json_base_quality() for k, q in pairs}
        stat["mean_quality"] = {k: q.json_mean_quality() for k, q in pairs}
        stat["base_content"] = {k: q.json_base_content() for k, q in pairs}
        stat["gc_content"] = {k: q.json_gc_content() for k, q in pairs}
        if self.paired:
            overlapped = c(capi.C_OVERLAPPED)
            base_sum = c(capi.C_OVERLAP_BASE_SUM)
            _, dist_hist = eng.histograms(capi.AQC_QC_COLS)
            ov = {
                'overlapped_pairs': overlapped,
                # python-2 integer division upstream: float(OVERLAP_LEN_SUM/OVERLAPPED) (preprocesser.py:752)
                'average_overlap_length': float(c(capi.C_OVERLAP_LEN_SUM) // overlapped) if overlapped > 0 else 0.0,
                'bad_mismatch_reads': f(capi.BADMISMATCH),
                'bad_diff': f(capi.BADDIFF),
                'bad_indel_reads': 0,
                'corrected_reads': c(capi.C_READ_CORRECTED),
                'corrected_bases': c(capi.C_BASE_CORRECTED),
                'skipped_correction_bases': c(capi.C_BASE_SKIPPED_CORRECTION),
                'zero_qual_masked': c(capi.C_BASE_ZERO_QUAL_MASKED),
                'zero_qual_skipped': c(capi.C_BASE_ZERO_QUAL_MASKED),
                'trimmed_adapter_bases': c(capi.C_TRIMMED_ADAPTER_BASE),
                'trimmed_adapter_reads': c(capi.C_TRIMMED_ADAPTER_READ),
                'error_rate': float(c(capi.C_OVERLAP_BASE_ERR)) / float(base_sum) if base_sum > 0 else 0.0,
                'edit_distance_histogram': [int(v) for v in dist_hist[0:min(10, readLen + 1)]],
            }
            mtx = {}
            for i, a in enumerate(ALL_BASES):
                mtx[a] = {b: c(capi.C_ERR_MATRIX0 + 4 * i + j) for j, b in enumerate(ALL_BASES) if b != a}
            ov['error_matrix'] = mtx
            stat["afterqc_overlap"] = ov
        return stat
