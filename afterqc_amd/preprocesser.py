"""Per-file driver: the batch-pipelined counterpart of the reference's seqFilter
(preprocesser.py:141-783).  Same constructor / run() interface and the same outputs
(good/bad/overlap FASTQ, QC/<R1 basename>.json), but the per-read loop of preprocesser.py:411-631
is not executed here: records are framed in bulk (afterqc_amd.fastq), shipped to the GPU as SoA
batches, judged by the HIP kernels behind the C ABI (afterqc_amd.capi.Engine) and only the 32-byte
verdict records come back.  This module is I/O and bookkeeping:

  pass 1  pre-filter QC sampling + auto-trim            (preprocesser.py:247-280)
  pass 2  every run takes the text path: raw text chunk -> aqc_frame -> aqc_run -> aqc_qc_stat(post) -> aqc_format ->
          good / bad / overlap text back -> files; framing and formatting happen on the device.
            * whole-input pipe (aqc_pipe_run, C++ reader / slot-worker / writer threads; the default for read files
              without index files and without --qc_only): chunks of exactly `chunk_records` records, dealt round robin
              over every engine (= GPU) the filter was given, outputs stitched in chunk order, statistics summed on
              the host;
            * serial chunk loop in this file (_run_text / _run_text_indexed): index files, --qc_only, inputs of
              irregular shape (the pipe reports them), injected engines;
          the host path (_run_host: numpy framing, Python writer) only exists as a cross-check (use_text_path=False)
  stats   counters -> JSON with the reference's schema  (preprocesser.py:660-778)

There is no CPU compute path: `engine` defaults to the HIP engine, which raises if the library or
the GPU is missing.  (Tests inject the oracle engine to exercise THIS file's host logic on CPU.)
"""
import json
import os
import sys
import queue
import threading

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


class _TextInput:
    """One input file as a stream of raw text chunks in two page-locked buffers.  A side thread reads the file into
    the buffer the main loop is not using; the unconsumed tail of a chunk is carried to the front of the next."""

    def __init__(self, eng, fname, chunk_bytes):
        self.eng = eng
        self.f = fastq.open_binary(fname)
        self.cap = max(int(chunk_bytes), 256)
        self.bufs = [eng.host_buffer(self.cap), eng.host_buffer(self.cap)]
        self.req = queue.Queue()
        self.done = queue.Queue()
        self.file_eof = False
        self.thread = threading.Thread(target=self._reader, daemon=True)
        self.thread.start()

    def _reader(self):
        while True:
            job = self.req.get()
            if job is None:
                return
            which, off = job
            try:
                view = self.bufs[which].view
                end = off
                final = self.file_eof
                while not final and end < self.cap:
                    got = self.f.readinto(view[end:self.cap])
                    if not got:
                        final = self.file_eof = True
                    else:
                        end += got
                self.done.put((end, final))
            except BaseException as e:      # surfaced by wait_fill on the main thread
                self.done.put(e)

    def start_fill(self, which, off):
        self.req.put((which, off))

    def wait_fill(self):
        r = self.done.get()
        if isinstance(r, BaseException):
            raise r
        return r

    def grow(self, which):
        """double both buffers (a single record did not fit)"""
        ncap = self.cap * 2
        nb = [self.eng.host_buffer(ncap), self.eng.host_buffer(ncap)]
        nb[which].array[:self.cap] = self.bufs[which].array[:self.cap]
        for b in self.bufs:
            b.free()
        self.bufs, self.cap = nb, ncap

    def carry(self, cur, consumed, nbytes, final, finished):
        """tail of buffer `cur` -> head of the other buffer, then ask the reader for the rest of it.
        `finished`: an empty line ended this file (fastq.py:44-47): nothing after it is ever read"""
        nxt = 1 - cur
        if finished:
            self.file_eof = True
            self.done.put((0, True))
            return
        left = nbytes - consumed
        if left:
            self.bufs[nxt].array[:left] = self.bufs[cur].array[consumed:nbytes]
        if final:
            self.done.put((left, True))
        else:
            self.start_fill(nxt, left)

    def close(self):
        self.req.put(None)
        self.thread.join()
        self.f.close()
        for b in self.bufs:
            b.free()


class _TextSink:
    """The good / bad writers of every file fed from the device.  The main loop only queues (slot, sizes) after
    aqc_format; a fetch thread copies the four formatted streams of that slot into page-locked buffers (and then
    releases the slot for the next chunk), a write thread writes them in order.  With two slots the upload and
    kernels of chunk i+1 overlap the download and file writes of chunk i."""

    N_SETS = 3

    def __init__(self, eng, writers, n_slots, store_overlap=False):
        self.eng = eng
        self.store_overlap = store_overlap
        self.writers = writers                       # per file (good, bad, overlap); None = not written
        self.sets = [[None] * 6 for _ in range(self.N_SETS)]
        self.set_free = [threading.Semaphore(1) for _ in range(self.N_SETS)]
        self.slot_free = [threading.Semaphore(1) for _ in range(n_slots)]
        self.fq = queue.Queue()
        self.wq = queue.Queue()
        self.err = None
        self.death = None                            # upstream's run ends inside a chunk (death_record): raised once all is written
        self.dead = False
        self.fetcher = threading.Thread(target=self._fetch, daemon=True)
        self.writer = threading.Thread(target=self._write, daemon=True)
        self.fetcher.start()
        self.writer.start()

    def _fetch(self):
        which = 0
        while True:
            job = self.fq.get()
            if job is None:
                self.wq.put(None)
                return
            slot, sizes = job[0], job[1]
            died = len(job) > 2 and job[2]           # (the main loop cut this chunk at the record upstream dies at)

            def fetch_all(sizes):
                for q, nbytes in enumerate(sizes):
                    if q // 3 >= len(self.writers) or nbytes == 0 or self.writers[q // 3][q % 3] is None:
                        continue
                    buf = self.sets[which][q]
                    if buf is None or buf.nbytes < nbytes:
                        if buf is not None:
                            buf.free()
                        buf = self.sets[which][q] = self.eng.host_buffer(nbytes + nbytes // 4 + 4096)
                    self.eng.fetch_text(slot, q // 3, q % 3, buf.array, buf.nbytes)

            try:
                self.set_free[which].acquire()           # the writer is done with this buffer set
                if self.err is None:
                    try:
                        fetch_all(sizes)
                    except capi.AqcError as e:
                        # an exception inside upstream's loop: its run ends at that record, what came before is written
                        k = death_record(self.eng, slot, e)
                        if k is None:
                            raise
                        sizes = self.eng.format(slot, k, self.store_overlap)
                        fetch_all(sizes)
                        self.death = e
                        died = True
            except BaseException as e:
                self.err = e
            finally:
                self.slot_free[slot].release()           # the slot may take the next chunk
                self.wq.put((which, list(sizes), died))
                which = (which + 1) % self.N_SETS

    def _write(self):
        while True:
            job = self.wq.get()
            if job is None:
                return
            which, sizes, died = job
            try:
                if self.err is None and not self.dead:
                    if died:
                        self.dead = True                 # this chunk is the last one written
                    for q, nbytes in enumerate(sizes):
                        if nbytes and q // 3 < len(self.writers) and self.writers[q // 3][q % 3] is not None:
                            self.writers[q // 3][q % 3].write_bytes(self.sets[which][q].view[:nbytes])
            except BaseException as e:
                self.err = e
            finally:
                self.set_free[which].release()

    def acquire_slot(self, slot):
        """block until the previous chunk of this slot has been fetched"""
        self.slot_free[slot].acquire()
        if self.err is not None:
            self.slot_free[slot].release()
            raise self.err

    def release_slot(self, slot):
        self.slot_free[slot].release()

    def emit(self, slot, sizes, death=None):
        if death is not None:
            self.death = death
        self.fq.put((slot, list(sizes), death is not None))

    def close(self):
        self.fq.put(None)
        self.fetcher.join()
        self.writer.join()
        for st in self.sets:
            for b in st:
                if b is not None:
                    b.free()
        if self.err is not None:
            raise self.err
        if self.death is not None:
            raise self.death


def death_record(eng, slot, e):
    """An exception INSIDE the reference's loop (KeyError / IndexError of the overlap walk, int() of a name field) ends its run
    at that record, with everything before it written.  -> the index of that record in the slot, or None when the error is
    not of that kind."""
    if isinstance(e, capi.AqcError) and e.code in capi.RECORD_ERRORS and hasattr(eng, "error_record"):
        return eng.error_record(slot)
    return None


class seqFilter:
    """seqFilter(options).run() — preprocesser.py:141-155,234-783."""

    def __init__(self, opt, engine=None, batch_records=1 << 20, device=0, chunk_bytes=64 << 20, use_text_path=True,
                 devices=None, use_pipe=True, chunk_records=1 << 17, pipe_slots=3, io_threads=0):
        self.options = opt
        self.chunk_bytes = chunk_bytes
        self.use_text_path = use_text_path
        self.text_path = False
        self.engine = engine
        self.device = device
        # devices: the GPUs one input is dealt over (default: just `device`); engines[0] is self.engine
        self.devices = list(devices) if devices else [device]
        self.extra_engines = []
        self.use_pipe = use_pipe
        self.used_pipe = False
        self.chunk_records = int(chunk_records)
        self.pipe_slots = int(pipe_slots)
        self.io_threads = int(io_threads)
        self.own_engine = False
        self.batch_records = max(int(batch_records), 1000)
        self.paired = opt.read2_file is not None
        self.bubbleCircles = []
        self.stat = None

    # ---- helpers ---------------------------------------------------------------------------------
    def _engine(self):
        if self.engine is None:
            self.engine = capi.Engine(self.devices[0], max(2, self.pipe_slots))
            self.own_engine = True
        return self.engine

    def _engines(self, all_devices=False):
        """engines[0] is the run's engine; the contexts on the other devices are only created once the whole-input pipe has
        been chosen (all_devices=True) — a run that stays on one engine (index files, --qc_only, .bz2, the serial fallback)
        never touches another GPU"""
        eng = self._engine()
        if all_devices and self.own_engine and not self.extra_engines and len(self.devices) > 1:
            cfg, circles = self._last_cfg
            for d in self.devices[1:]:
                e = capi.Engine(d, max(2, self.pipe_slots))
                e.set_config(cfg)
                e.set_circles(circles)
                e.reset_stats()
                self.extra_engines.append(e)
        return [eng] + self.extra_engines

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
        import time
        opt = self.options
        t_init = time.perf_counter()
        eng = self._engine()
        t_init = time.perf_counter() - t_init        # (a fresh process: HIP runtime start + the library's code object)
        paired = self.paired
        if opt.debubble:
            self.bubbleCircles = load_circles(opt.debubble_dir)
        # no front trim if the sequence is barcoded (preprocesser.py:242-243)
        if opt.barcode:
            opt.trim_front = 0

        self.timing = {"init_s": t_init}
        t_run = time.perf_counter()
        has_i1 = opt.index1_file is not None
        has_i2 = opt.index2_file is not None
        self._last_cfg = (build_config(opt, paired, has_i2), self.bubbleCircles)
        for e in self._engines():
            e.set_config(self._last_cfg[0])
            e.set_circles(self.bubbleCircles)
            e.reset_stats()

        # ---- pass 1: pre-filter QC on a sample of each file (preprocesser.py:247-251)
        r1pre = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R1_PRE)
        r2pre = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R2_PRE)
        r1post = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R1_POST)
        r2post = QualityControl(opt.qc_sample, opt.qc_kmer, eng, capi.QC_R2_POST)
        single = lambda rb: capi.Batch.from_raw(rb)
        if self.use_text_path and hasattr(eng, "frame"):
            side, side_err = None, []
            if paired and isinstance(eng, capi.Engine) and eng.n_slots >= 2:
                # read 2 is sampled at the same time in slot 1 (a .gz spends most of this pass decoding)
                def _sample_r2():
                    try:
                        r2pre.statFileText(opt.read2_file, self.chunk_bytes, slot=1)
                    except BaseException as e:       # re-raised on the main thread
                        side_err.append(e)
                side = threading.Thread(target=_sample_r2, name="aqc-sample-r2")
                side.start()
            try:
                r1pre.statFileText(opt.read1_file, self.chunk_bytes)
            finally:
                if side is not None:
                    side.join()
            if side_err:
                raise side_err[0]
            if paired and side is None:
                r2pre.statFileText(opt.read2_file, self.chunk_bytes)
        else:
            r1pre.statFile(opt.read1_file, fastq.Reader, single, self.batch_records)
            if paired:
                # the R2 file is stat'd through the same single-read path into its own accumulator
                r2pre.statFile(opt.read2_file, fastq.Reader, single, self.batch_records)
        readLen = r1pre.readLen
        self.timing["pass1_s"] = time.perf_counter() - t_run

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
            os.makedirs(d, exist_ok=True)          # (directory mode runs several files at once)
        if opt.store_overlap and paired:
            os.makedirs(overlap_dir, exist_ok=True)
        gzip_out = bool(opt.gzip) or opt.read1_file.endswith(".gz")
        files = [opt.read1_file, opt.read2_file, opt.index1_file, opt.index2_file]
        if opt.store_overlap and not opt.qc_only:
            os.makedirs(overlap_dir, exist_ok=True)   # single-end + store_overlap: upstream opens the writer without the dir
        # ---- pass 2: the main loop (preprocesser.py:411-631)
        # the per-read settings now include the resolved trim values
        self._last_cfg = (build_config(opt, paired, has_i2), self.bubbleCircles)
        for e in self._engines():
            e.set_config(self._last_cfg[0])
        # text in / text out on the device (aqc_frame / aqc_format); use_text_path=False keeps the host-side framing and
        # writer below as a cross-check.  An injected engine without the text calls takes the host path.
        self.text_path = self.use_text_path and hasattr(eng, "frame")
        t_p2 = time.perf_counter()
        cpu_p2 = sum(os.times()[:2])
        outs = None
        extra_bases = None
        readers = []
        if self.text_path and self.use_pipe and not (has_i1 or has_i2) and not opt.qc_only and isinstance(eng, capi.Engine):
            extra_bases = self._run_pipe(opt, files, good_dir, bad_dir, overlap_dir, gzip_out, paired)
            if extra_bases is None:
                # not the regular shape (empty line inside, mates of different lengths, ...): start over, chunk by chunk
                print("afterqc_amd: %s is not of the regular shape the whole-input pipe takes (blank line inside / mates of "
                      "different lengths); running it again through the serial chunk loop" % opt.read1_file, file=sys.stderr)
                self.timing["pipe_fallback"] = True
                for e in self._engines():
                    e.reset_stats()
        try:
            if extra_bases is not None:
                pass
            elif self.text_path and (has_i1 or has_i2):
                outs = _Outputs(opt, files, good_dir, bad_dir, overlap_dir, gzip_out, opt.compression)
                extra_bases = self._run_text_indexed(eng, opt, outs, paired)
            elif self.text_path:
                outs = _Outputs(opt, files, good_dir, bad_dir, overlap_dir, gzip_out, opt.compression)
                extra_bases = self._run_text(eng, opt, outs, paired)
            else:
                outs = _Outputs(opt, files, good_dir, bad_dir, overlap_dir, gzip_out, opt.compression)
                readers, extra_bases = self._run_host(eng, opt, outs, paired, files)
        finally:
            # (also when the run ends in an exception: what was written up to the record upstream dies at stays written)
            for r in readers:
                if r is not None:
                    r.close()
            if outs is not None:
                outs.close()
        self.timing["pass2_s"] = time.perf_counter() - t_p2
        self.timing["pass2_cpu_s"] = sum(os.times()[:2]) - cpu_p2          # user + system, all threads: how many cores pass 2 kept busy

        t_stats = time.perf_counter()
        try:
            # statistics: per-GPU integers summed on the host (only the pipe spreads a run over several engines)
            stat_eng = capi.MergedEngines(self._engines()) if self.extra_engines else eng
            r1post.engine = r2post.engine = stat_eng
            r1post.qc()
            if paired:
                r2post.qc()

            self.stat = self._stats(stat_eng, r1pre, r2pre, r1post, r2post, readLen, extra_bases)
            stat_path = os.path.join(qc_dir, os.path.basename(opt.read1_file) + ".json")
            with open(stat_path, "w") as f:
                f.write(json.dumps(self.stat, sort_keys=True, indent=4, separators=(',', ': ')))
            self.timing["stats_s"] = time.perf_counter() - t_stats
            t_report = time.perf_counter()
            # the HTML report next to it (preprocesser.py:780-783, qcreporter.py).  The FASTQ outputs and the statistics are
            # complete at this point: a problem in the report is reported, it does not fail the run
            try:
                from . import qcreporter
                ovl_hist, _ = stat_eng.histograms(capi.AQC_QC_COLS)
                figures = qcreporter.build_figures(self.stat, opt, r1pre, r2pre, r1post, r2post, ovl_hist, readLen)
                with open(os.path.join(qc_dir, os.path.basename(opt.read1_file) + ".html"), "w") as f:
                    f.write(qcreporter.render(self.stat, figures, getattr(opt, "version", "")))
            except Exception as e:      # noqa: BLE001 — whatever it is, the run's results stand
                if os.environ.get("AQC_REPORT_STRICT"):      # the test suites: a broken report must not pass unseen
                    raise
                print("afterqc_amd: the HTML report could not be written (%s: %s); outputs and %s are complete"
                      % (type(e).__name__, e, stat_path), file=sys.stderr)
                self.timing["report_error"] = "%s: %s" % (type(e).__name__, e)
            self.timing["report_s"] = time.perf_counter() - t_report
            self.timing["total_s"] = time.perf_counter() - t_run
        finally:
            t_close = time.perf_counter()
            if self.own_engine:
                for e in [self.engine] + self.extra_engines:
                    if e is not None:
                        e.close()
                self.engine = None
                self.extra_engines = []
            self.timing["close_s"] = time.perf_counter() - t_close
        return self.stat

    # ---- pass 2 through the whole-input pipe (aqc_pipe_run) ---------------------------------------------------------------
    def _run_pipe(self, opt, files, good_dir, bad_dir, overlap_dir, gzip_out, paired):
        """Hands the read file(s) to the C++ pipe: chunks of `chunk_records` records dealt over every engine, outputs written
        in chunk order by its writer thread.  Returns the extra-bases quirk value (preprocesser.py:416-421) or None when the
        pipe met an input it cannot chunk (the caller falls back to the serial chunk loop)."""
        nfiles = 2 if paired else 1
        outputs = []
        for k in range(nfiles):
            main = getMainName(files[k])
            ext = ".gz" if gzip_out else ""
            # upstream opens the R1 overlap writer whenever store_overlap is on, the others only when paired (_Outputs)
            want_ovl = (opt.store_overlap and k == 0) or (opt.store_overlap and paired)
            outputs.append((os.path.join(good_dir, main + ".good.fq" + ext), os.path.join(bad_dir, main + ".bad.fq" + ext),
                            os.path.join(overlap_dir, main + ".overlap.fq" + ext) if want_ovl else None))
        engines = self._engines(all_devices=True)
        pipe = capi.Pipe(engines, slots=min([self.pipe_slots] + [e.n_slots for e in engines]), io_threads=self.io_threads)
        try:
            # (fastq.py:23-26: .gz through gzip.open, .bz2 through bz2.BZ2File upstream — here the pipe's own decoders: 1 gzip, 2 bzip2)
            res = pipe.run(files[:nfiles], outputs, gzip_in=[1 if f.endswith(".gz") else 2 if f.endswith(".bz2") else 0 for f in files[:nfiles]], gzip_out=gzip_out,
                           gzip_level=opt.compression, chunk_records=self.chunk_records, qc_sample=opt.qc_sample,
                           store_overlap=bool(opt.store_overlap) and paired)
        except capi.AqcError as e:
            # (an exception inside upstream's loop — capi.RECORD_ERRORS — ends the run at that record: the pipe has written
            #  everything before it and says so; there is nothing to rerun)
            self.used_pipe = e.code in capi.RECORD_ERRORS
            raise
        finally:
            pipe.close()
        self.used_pipe = not res.anomaly
        if res.anomaly:
            return None
        self.timing["pipe_s"] = res.seconds
        self.timing["pipe_threads"] = res.breakdown()
        self.timing["pipe_chunks"] = (int(res.chunks), int(res.fused_chunks))       # (all, placed by the verdict kernel: AQC_FUSED=1)
        # (R1's record that upstream had read and counted before a shorter R2 ended its loop, preprocesser.py:416-421: the pipe
        #  applies fastq.Reader's end-of-file rules itself since round 6)
        return int(res.extra_bases)

    # ---- pass 2, text path with index files (-7 / -5): four lock-stepped inputs, two device slots ---------------------
    def _run_text_indexed(self, eng, opt, outs, paired):
        """Like _run_text, for runs with index files: the reads are framed into slot 0, the index reads into slot 1 (capped
        at the reads' record count), the index records are formatted whole under the reads' verdicts
        (aqc_format_plain).  Rare in practice, so this variant is kept simple: one chunk at a time, fetch and write inline."""
        files = [opt.read1_file, opt.read2_file, opt.index1_file, opt.index2_file]
        present = [k for k in range(4) if files[k] is not None]
        groups = [[k for k in present if k < 2], [k for k in present if k >= 2]]
        inputs = dict((k, _TextInput(eng, files[k], self.chunk_bytes)) for k in present)
        writers = dict((k, (outs.good[k], outs.bad[k], outs.overlap[k])) for k in present)
        hold = {}
        total = 0
        extra_bases = 0
        cur = 0
        UNLIMITED = capi.UINT64_MAX

        def frame_group(slot, g, fills, cap):
            a = g[0]
            if len(g) > 1:
                b = g[1]
                return eng.frame(slot, inputs[a].bufs[cur].array, fills[a][0], fills[a][1], inputs[b].bufs[cur].array, fills[b][0],
                                 fills[b][1], max_records=cap, first_index=total)
            return eng.frame(slot, inputs[a].bufs[cur].array, fills[a][0], fills[a][1], max_records=cap, first_index=total)

        def per_file(info, g):
            d = {g[0]: (int(info.avail1), bool(info.eof1), int(info.consumed1))}
            if len(g) > 1:
                d[g[1]] = (int(info.avail2), bool(info.eof2), int(info.consumed2))
            return d

        def write_streams(slot, g, sizes):
            for q, nbytes in enumerate(sizes):
                if nbytes == 0 or q // 3 >= len(g):
                    continue
                w = writers[g[q // 3]][q % 3]
                if w is None:
                    continue
                buf = hold.get(q)
                if buf is None or buf.nbytes < nbytes:
                    if buf is not None:
                        buf.free()
                    buf = hold[q] = eng.host_buffer(nbytes + nbytes // 4 + 4096)
                eng.fetch_text(slot, q // 3, q % 3, buf.array, buf.nbytes)
                w.write_bytes(buf.view[:nbytes])

        try:
            for k in present:
                inputs[k].start_fill(cur, 0)
            while True:
                fills = dict((k, inputs[k].wait_fill()) for k in present)
                info_a = frame_group(0, groups[0], fills, UNLIMITED)
                info_b = frame_group(1, groups[1], fills, int(info_a.n))
                if int(info_b.n) < int(info_a.n):
                    info_a = frame_group(0, groups[0], fills, int(info_b.n))     # the index chunk holds fewer records
                n = int(info_b.n)
                state = per_file(info_a, groups[0])
                state.update(per_file(info_b, groups[1]))
                done = dict((k, (state[k][1] or fills[k][1]) and state[k][0] == n) for k in present)
                stop = False
                if done[0]:
                    stop = True
                elif any(done[k] for k in present if k != 0) and state[0][0] > n:
                    # R1's next record was read (and counted into TOTAL_BASES) before another reader ran dry (:416-429)
                    extra_bases = int(info_a.next_len1)
                    stop = True
                if not stop:
                    for k in present:
                        if n == 0 and not fills[k][1] and not state[k][1] and state[k][0] == 0:
                            inputs[k].grow(cur)
                        inputs[k].carry(cur, state[k][2], fills[k][0], fills[k][1], state[k][1] or (k != 0 and done[k]))
                if n:
                    death = None
                    try:
                        limit = UNLIMITED
                        if opt.qc_only:
                            eng.run(0, 0)
                            flags = eng.fetch_results(0)[:n]["flag"]
                            hit = np.flatnonzero((flags == capi.GOOD) & (total + 1 + np.arange(n) >= opt.qc_sample))
                            if len(hit):
                                n = int(hit[0]) + 1
                                stop = True
                            limit = n
                        eng.run(0, limit)
                        n_qc = n if opt.qc_sample <= 0 else max(0, min(n, opt.qc_sample - 1 - total))
                        if n_qc > 0:
                            eng.qc_stat(0, capi.QC_R1_POST, 0, 0, n_qc, 1)
                            if paired:
                                eng.qc_stat(0, capi.QC_R2_POST, 1, 0, n_qc, 1)
                        eng.sync(0)
                    except capi.AqcError as e:
                        k = death_record(eng, 0, e)          # upstream's run ends at that record, what came before is written
                        if k is None or k >= n:
                            raise
                        n, death = k, e
                    if not opt.qc_only:
                        write_streams(0, groups[0], eng.format(0, n, bool(opt.store_overlap)))
                        write_streams(1, groups[1], eng.format_plain(1, 0, n, bool(opt.store_overlap)))
                    if death is not None:
                        raise death
                    total += n
                if stop:
                    break
                cur = 1 - cur
        finally:
            for k in present:
                inputs[k].close()
            for b in hold.values():
                b.free()
        return extra_bases

    # ---- pass 2 with host-side framing / formatting (general: barcodes, index files, overlap store, bubbles) ------
    def _run_host(self, eng, opt, outs, paired, files):
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
            death = None
            try:
                results = eng.fetch_results(slot)[:n]
            except capi.AqcError as e:
                k = death_record(eng, slot, e)               # upstream's run ends at that record, what came before is written
                if k is None or k >= n:
                    raise
                n, death = k, e
                results = eng.fetch_results(slot)[:n]
            if not opt.qc_only:
                qviews = None
                if batch.qlen1 is not None:
                    qviews = [eng.fetch_quality_views(slot, 0), eng.fetch_quality_views(slot, 1) if paired else None]
                self._write(outs, rbs, results, n, qviews)
            if death is not None:
                raise death
            total += n
        return readers, extra_bases

    # ---- pass 2, text in / text out: framing and formatting on the device -------------------
    def _run_text(self, eng, opt, outs, paired):
        """The loop of preprocesser.py:411-631 over raw text chunks.  Per chunk the host only moves bytes: file ->
        page-locked buffer -> aqc_frame (records found on the device) -> aqc_run / aqc_qc_stat -> aqc_format
        (good / bad text built on the device) -> page-locked buffer -> file.  File reads and writes run on side
        threads while the main thread sits in the (GIL-free) C-ABI calls."""
        files = [opt.read1_file] + ([opt.read2_file] if paired else [])
        inputs = [_TextInput(eng, f, self.chunk_bytes) for f in files]
        n_slots = min(2, getattr(eng, "n_slots", 1))
        sink = _TextSink(eng, [(outs.good[k], outs.bad[k], outs.overlap[k]) for k in range(len(files))], n_slots, bool(opt.store_overlap))
        total = 0
        extra_bases = 0
        slot = 0
        cur = 0
        try:
            for inp in inputs:
                inp.start_fill(cur, 0)
            while True:
                fills = [inp.wait_fill() for inp in inputs]          # (bytes in buffer `cur`, final)
                sink.acquire_slot(slot)                               # its previous chunk has left the device
                a1, n1, f1 = inputs[0].bufs[cur].array, fills[0][0], fills[0][1]
                if paired:
                    info = eng.frame(slot, a1, n1, f1, inputs[1].bufs[cur].array, fills[1][0], fills[1][1], first_index=total)
                else:
                    info = eng.frame(slot, a1, n1, f1, first_index=total)
                n = int(info.n)
                # lock step (preprocesser.py:412-429): R1 is read first; the first reader to return None ends the loop,
                # and an R1 record read just before R2 ran dry has already been counted into TOTAL_BASES (:416)
                done1 = (info.eof1 or f1) and info.avail1 == n
                done2 = paired and (info.eof2 or fills[1][1]) and info.avail2 == n
                stop = False
                if done1:
                    stop = True
                elif done2 and info.avail1 > n:
                    extra_bases = int(info.next_len1)
                    stop = True
                if not stop:
                    consumed = [int(info.consumed1), int(info.consumed2)]
                    eofs = [bool(info.eof1), bool(info.eof2)]
                    for k, inp in enumerate(inputs):
                        if n == 0 and not fills[k][1] and not eofs[k] and (info.avail1, info.avail2)[k] == 0:
                            inp.grow(cur)                                  # not even one record fits the buffer
                        inp.carry(cur, consumed[k], fills[k][0], fills[k][1], eofs[k] or (k == 1 and done2))
                if n:
                    death = None
                    try:
                        limit = capi.UINT64_MAX
                        if opt.qc_only:
                            # --qc_only stops at the first good record whose 1-based index reaches qc_sample (:630-631):
                            # verdicts first (nothing accumulated), then the accumulating run up to that record
                            eng.run(slot, 0)
                            flags = eng.fetch_results(slot)[:n]["flag"]
                            hit = np.flatnonzero((flags == capi.GOOD) & (total + 1 + np.arange(n) >= opt.qc_sample))
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
                            eng.sync(slot)       # the sampling kernels share per-context scratch: never two chunks at once
                        if opt.qc_only:
                            eng.sync(slot)
                    except capi.AqcError as e:
                        # an exception inside upstream's loop ends its run AT that record: what came before is written
                        k = death_record(eng, slot, e)
                        if k is None or k >= n:
                            raise
                        n, death, stop = k, e, True
                    if opt.qc_only:
                        sink.release_slot(slot)
                        if death is not None:
                            raise death
                    else:
                        sizes = eng.format(slot, n, bool(opt.store_overlap))
                        sink.emit(slot, sizes, death)   # (the fetch thread releases the slot)
                    total += n
                else:
                    sink.release_slot(slot)
                if stop:
                    break
                cur = 1 - cur
                slot = (slot + 1) % n_slots
        finally:
            sink.close()
            for inp in inputs:
                inp.close()
        return extra_bases

    # ---- output formatting (writeReads, preprocesser.py:206-232; fastq.Writer.writeLines) ------------
    def _write(self, outs, rbs, results, n, qviews=None):
        """qviews: per mate (starts, lengths) of the quality-string slices (Engine.fetch_quality_views) when some record's quality
        line is not as long as its sequence line; None: the slices of the reads"""
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
            qs, ql = (st, ln) if qviews is None else (int(qviews[0][0][i]), int(qviews[0][1][i]))
            s1 = bytearray(s1[st:st + ln]); q1 = bytearray(q1[qs:qs + ql])
            if paired:
                st2, ln2 = int(r["start2"]), int(r["len2"])
                qs2, ql2 = (st2, ln2) if qviews is None else (int(qviews[1][0][i]), int(qviews[1][1][i]))
                s2 = bytearray(s2[st2:st2 + ln2]); q2 = bytearray(q2[qs2:qs2 + ql2])
            ov = int(r["overlap_len"])
            ne = int(r["n_edits"])
            corrected = 0
            # (each string is edited at its OWN index, preprocesser.py:575-592: the quality strings from their own ends — the
            #  same positions as the bases' unless a quality line has a length of its own; a negative index wraps, as upstream)
            for k in range(ne):
                e = r["edits"][k]
                o = int(e["o"]); kind = int(e["kind"])
                if kind == capi.EDIT_FIX_R2:
                    s2[ln2 - 1 - o] = int(e["base"]); q2[len(q2) - 1 - o] = int(e["qual"]); corrected += 1
                elif kind == capi.EDIT_FIX_R1:
                    s1[ln - ov + o] = int(e["base"]); q1[len(q1) - ov + o] = int(e["qual"]); corrected += 1
                else:
                    q2[len(q2) - 1 - o] = 33; q1[len(q1) - ov + o] = 33
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
                    chunks[0][2].append(name1 + b"\n" + bytes(s1[ln - ov:]) + b"\n" + p1 + b"\n" + bytes(q1[len(q1) - ov:]) + b"\n")
                    chunks[1][2].append(name2 + b"\n" + bytes(s2[ln2 - ov:]) + b"\n" + p2 + b"\n" + bytes(q2[len(q2) - ov:]) + b"\n")
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
