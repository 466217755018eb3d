"""ctypes binding of include/afterqc_hip.h (the C ABI of libafterqc_hip.so).

The structures and enums below mirror the header field by field.  Loading is strict: if the HIP
library has not been built, or no GPU is visible, the product path raises — there is NO CPU
fallback anywhere in afterqc_amd (the CPU restatement lives in oracle/ and is test-only).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AQC_LIB") or os.path.join(HERE, "csrc", "libafterqc_hip.so")   # AQC_LIB: A/B builds only

AQC_MAX_READ_LEN = 1000
AQC_QC_COLS = 1024

FLAG_NAMES = ["GOOD", "BADBCD1", "BADBCD2", "BADTRIM1", "BADTRIM2", "BADBBL", "BADLEN", "BADPOL", "BADLQC",
              "BADNCT", "BADDIFF", "BADMISMATCH"]
(GOOD, BADBCD1, BADBCD2, BADTRIM1, BADTRIM2, BADBBL, BADLEN, BADPOL, BADLQC, BADNCT, BADDIFF,
 BADMISMATCH) = range(12)
N_FLAGS = 12
EDIT_FIX_R2, EDIT_FIX_R1, EDIT_MASK = 1, 2, 3

# enum aqc_counter
C_TOTAL_READS, C_TOTAL_BASES, C_GOOD_READS, C_GOOD_BASES, C_FLAG0 = 0, 1, 2, 3, 4
C_READ_CORRECTED = C_FLAG0 + N_FLAGS
(C_BASE_CORRECTED, C_BASE_SKIPPED_CORRECTION, C_BASE_ZERO_QUAL_MASKED, C_OVERLAPPED, C_OVERLAP_LEN_SUM,
 C_OVERLAP_BASE_SUM, C_OVERLAP_BASE_ERR, C_TRIMMED_ADAPTER_BASE, C_TRIMMED_ADAPTER_READ,
 C_ERR_MATRIX0) = range(C_READ_CORRECTED + 1, C_READ_CORRECTED + 11)
N_COUNTERS = C_ERR_MATRIX0 + 16

# enum aqc_qc_row
(QC_TOTAL_NUM, QC_TOTAL_QUAL, QC_BASE_COUNT_A, QC_BASE_COUNT_T, QC_BASE_COUNT_C, QC_BASE_COUNT_G, QC_BASE_QUAL_A,
 QC_BASE_QUAL_T, QC_BASE_QUAL_C, QC_BASE_QUAL_G, QC_DISCONTINUITY, QC_GC_HIST, QC_SCALARS, QC_ROWS) = range(14)
QC_R1_PRE, QC_R2_PRE, QC_R1_POST, QC_R2_POST = range(4)
K_FILTER_OVERLAP, K_QC_STAT, N_KERNELS = 0, 1, 2
UINT64_MAX = (1 << 64) - 1

ERRORS = {-1: "HIP runtime error", -2: "bad argument", -3: "read longer than AQC_MAX_READ_LEN", -4: "no gfx950 device",
          -5: "bad call sequence", -6: "byte outside the reference's COMP alphabet", -7: "unsupported option value",
          -8: "quality line too short for the overlap walk (IndexError upstream)"}
ERR_ARG, ERR_ALPHABET, ERR_INDEX = -2, -6, -8
# errors that end the reference's run AT A RECORD (an exception inside its loop): the records before it were written
RECORD_ERRORS = (ERR_ARG, ERR_ALPHABET, ERR_INDEX)

# numpy view of struct aqc_result (packed, 32 bytes)
EDIT_DTYPE = np.dtype([("o", "<u2"), ("kind", "u1"), ("base", "u1"), ("qual", "u1")])
RESULT_DTYPE = np.dtype([("flag", "u1"), ("n_edits", "u1"), ("start1", "<u2"), ("len1", "<u2"), ("start2", "<u2"),
                         ("len2", "<u2"), ("offset", "<i2"), ("overlap_len", "<u2"), ("distance", "<u2"),
                         ("edits", EDIT_DTYPE, (3,)), ("barcode", "u1")])
assert RESULT_DTYPE.itemsize == 32
SPAN_EVENT_DTYPE = np.dtype([("in_start", "<u4"), ("in_len", "<u4"), ("out_len", "<u4")])      # struct aqc_span_event


def assemble_spans(chunk, end, events, stream0):
    """The good output of one file from an aqc_format_spans result: the chunk's own bytes between the events, the rebuilt records
    (stream 0, in order) at them — what aqc_pipe_run's file writers do with writev (include/afterqc_hip.h)."""
    out = []
    cursor = poff = 0
    for e in events:
        a, ln, ol = int(e["in_start"]), int(e["in_len"]), int(e["out_len"])
        out.append(bytes(chunk[cursor:a]))
        out.append(bytes(stream0[poff:poff + ol]))
        poff += ol
        cursor = a + ln
    out.append(bytes(chunk[cursor:end]))
    return b"".join(out)


class Config(C.Structure):
    """struct aqc_config"""
    _fields_ = [(n, C.c_int32) for n in (
        "paired", "count_r2_bases", "trim_front", "trim_tail", "trim_front2", "trim_tail2", "seq_len_req",
        "poly_size_limit", "allow_mismatch_in_poly", "qualified_quality_phred", "unqualified_base_limit",
        "n_base_limit", "no_overlap", "no_correction", "mask_mismatch", "barcode", "barcode_length",
        "barcode_verify_len")] + [("barcode_verify", C.c_uint8 * 32), ("debubble", C.c_int32), ("qc_kmer", C.c_int32)]

    def set_verify(self, verify):
        v = verify.encode() if isinstance(verify, str) else bytes(verify)
        if len(v) > 32:
            raise ValueError("barcode_verify longer than 32 bytes")
        self.barcode_verify_len = len(v)
        for i, ch in enumerate(v):
            self.barcode_verify[i] = ch


class BatchStruct(C.Structure):
    """struct aqc_batch"""
    _fields_ = [("n", C.c_uint64), ("first_index", C.c_uint64),
                ("seq1", C.c_void_p), ("qual1", C.c_void_p), ("off1", C.c_void_p), ("qoff1", C.c_void_p),
                ("len1", C.c_void_p), ("bytes1", C.c_uint64), ("qbytes1", C.c_uint64),
                ("seq2", C.c_void_p), ("qual2", C.c_void_p), ("off2", C.c_void_p), ("qoff2", C.c_void_p),
                ("len2", C.c_void_p), ("bytes2", C.c_uint64), ("qbytes2", C.c_uint64),
                ("aux_lane", C.c_void_p), ("aux_tile", C.c_void_p), ("aux_x", C.c_void_p), ("aux_y", C.c_void_p),
                ("aux_ok", C.c_void_p), ("qlen1", C.c_void_p), ("qlen2", C.c_void_p)]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class TextChunk(C.Structure):
    """struct aqc_text_chunk (include/afterqc_hip.h)"""
    _fields_ = [("text1", C.c_void_p), ("bytes1", C.c_uint64), ("text2", C.c_void_p), ("bytes2", C.c_uint64),
                ("final1", C.c_int32), ("final2", C.c_int32), ("max_records", C.c_uint64), ("first_index", C.c_uint64)]


class TextExtent(C.Structure):
    """struct aqc_text_extent: a stretch of a chunk whose bytes are in device memory already (aqc_frame_mixed)"""
    _fields_ = [("offset", C.c_uint64), ("bytes", C.c_uint64), ("device_text", C.c_void_p)]


class FrameInfo(C.Structure):
    """struct aqc_frame_info"""
    _fields_ = [("n", C.c_uint64), ("avail1", C.c_uint64), ("avail2", C.c_uint64), ("consumed1", C.c_uint64),
                ("consumed2", C.c_uint64), ("eof1", C.c_int32), ("eof2", C.c_int32), ("max_len", C.c_uint32),
                ("next_len1", C.c_uint32)]


class PipeIO(C.Structure):
    """struct aqc_pipe_io"""
    _fields_ = [("in_path", C.c_char_p * 2), ("in_mem", C.c_void_p * 2), ("in_mem_bytes", C.c_uint64 * 2),
                ("gzip_in", C.c_int32 * 2), ("out_path", (C.c_char_p * 3) * 2), ("gzip_out", C.c_int32), ("gzip_level", C.c_int32)]


class PipeOpts(C.Structure):
    """struct aqc_pipe_opts"""
    _fields_ = [("chunk_records", C.c_uint64), ("qc_sample", C.c_int64), ("store_overlap", C.c_int32), ("no_output", C.c_int32),
                ("chunk_index0", C.c_uint64), ("chunk_index_stride", C.c_uint64)]


class PipeResult(C.Structure):
    """struct aqc_pipe_result"""
    _fields_ = [("records", C.c_uint64), ("chunks", C.c_uint64), ("bytes_out", C.c_uint64 * 6), ("anomaly", C.c_int32),
                ("fused_chunks", C.c_int32), ("extra_bases", C.c_uint64), ("seconds", C.c_double)] + [(k, C.c_double) for k in (
                    "t_read", "t_count", "t_wait_ring", "t_frame", "t_kernels", "t_wait_set", "t_fetch", "t_write")]

    def breakdown(self):
        return {k: round(getattr(self, k), 4) for k in ("seconds", "t_read", "t_count", "t_wait_ring", "t_frame", "t_kernels",
                                                        "t_wait_set", "t_fetch", "t_write")}


class HostBuffer:
    """Page-locked host memory from aqc_host_alloc, exposed as a writable numpy uint8 array / memoryview."""

    def __init__(self, lib, nbytes):
        self._lib = lib
        self.nbytes = int(nbytes)
        self.ptr = lib.aqc_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("aqc_host_alloc(%d) failed" % self.nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr))
        self.view = memoryview(self.array)

    def free(self):
        if self.ptr:
            self.array = None
            self.view = None
            self._lib.aqc_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Batch:
    """Host-side packed SoA batch (numpy owned).  Arenas are 16-byte aligned per record and padded
    with 64 zero bytes so that vector loads past the last record stay inside the allocation."""

    ALIGN = 16
    PAD = 64

    def __init__(self, n, first_index=0):
        self.n = n
        self.first_index = first_index
        self.seq1 = self.qual1 = self.off1 = self.len1 = self.qoff1 = None
        self.seq2 = self.qual2 = self.off2 = self.len2 = self.qoff2 = None
        self.qlen1 = self.qlen2 = None     # lengths of the quality strings where some differ from the reads' (None: all equal)
        self.aux = None  # (lane, tile, x, y, ok) int32 x4 + uint8
        self._keep = None

    @staticmethod
    def _offsets(lens):
        padded = (lens.astype(np.uint64) + np.uint64(Batch.ALIGN - 1)) & ~np.uint64(Batch.ALIGN - 1)
        off = np.zeros(len(lens), dtype=np.uint64)
        if len(lens) > 1:
            np.cumsum(padded[:-1], out=off[1:])
        total = int(padded.sum()) + Batch.PAD
        return off, total

    @classmethod
    def from_matrices(cls, seq1, qual1, len1, seq2=None, qual2=None, len2=None, first_index=0):
        """From fixed-width uint8 matrices (synthetic workloads)."""
        b = cls(len(len1), first_index)
        b.seq1, b.qual1, b.off1, b.len1 = cls._pack_matrix(seq1, qual1, len1)
        if seq2 is not None:
            b.seq2, b.qual2, b.off2, b.len2 = cls._pack_matrix(seq2, qual2, len2)
        return b

    @staticmethod
    def _pack_matrix(seq, qual, lens):
        n, W = seq.shape
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        stride = (W + Batch.ALIGN - 1) // Batch.ALIGN * Batch.ALIGN
        if np.all(lens == W):
            # uniform lengths: the padded matrix IS the arena
            off = (np.arange(n, dtype=np.uint64) * np.uint64(stride))
            sa = np.zeros(n * stride + Batch.PAD, dtype=np.uint8)
            qa = np.zeros(n * stride + Batch.PAD, dtype=np.uint8)
            sa[:n * stride].reshape(n, stride)[:, :W] = seq
            qa[:n * stride].reshape(n, stride)[:, :W] = qual
            return sa, qa, off, lens
        off, total = Batch._offsets(lens)
        sa = np.zeros(total, dtype=np.uint8)
        qa = np.zeros(total, dtype=np.uint8)
        for i in range(n):
            l = int(lens[i]); o = int(off[i])
            sa[o:o + l] = seq[i, :l]
            qa[o:o + l] = qual[i, :l]
        return sa, qa, off, lens

    @classmethod
    def from_strings(cls, seqs1, quals1=None, seqs2=None, quals2=None, first_index=0):
        """From Python bytes/str lists (tests, function seams)."""
        b = cls(len(seqs1), first_index)
        b.seq1, b.qual1, b.off1, b.len1 = cls._pack_strings(seqs1, quals1)
        if seqs2 is not None:
            b.seq2, b.qual2, b.off2, b.len2 = cls._pack_strings(seqs2, quals2)
        return b

    @staticmethod
    def _pack_strings(seqs, quals):
        enc = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
        lens = np.array([len(s) for s in enc], dtype=np.uint32)
        off, total = Batch._offsets(lens)
        sa = np.zeros(total, dtype=np.uint8)
        qa = np.zeros(total, dtype=np.uint8)
        for i, s in enumerate(enc):
            o = int(off[i])
            sa[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
        if quals is not None:
            for i, q in enumerate(quals):
                q = q.encode("latin-1") if isinstance(q, str) else bytes(q)
                if len(q) != int(lens[i]):
                    raise ValueError("record %d: quality length %d != sequence length %d" % (i, len(q), int(lens[i])))
                o = int(off[i])
                qa[o:o + len(q)] = np.frombuffer(q, dtype=np.uint8)
        else:
            for i in range(len(enc)):
                o = int(off[i])
                qa[o:o + int(lens[i])] = ord("I")
        return sa, qa, off, lens

    @classmethod
    def from_raw(cls, rb1, rb2=None, first_index=0):
        """Zero-copy view of framed FASTQ text (afterqc_amd.fastq.RawBatch): the text chunk is the arena."""
        b = cls(rb1.n, first_index)
        b.seq1 = b.qual1 = rb1.text
        b.off1, b.qoff1, b.len1 = rb1.seq_off, rb1.qual_off, rb1.seq_len
        # a quality line that is not as long as its sequence line is no error upstream (fastq.py:37-49 does not look): the
        # quality strings travel with lengths of their own (aqc_batch.qlen1 / qlen2)
        irregular = not np.array_equal(rb1.seq_len, rb1.qual_len)
        if rb2 is not None:
            b.seq2 = b.qual2 = rb2.text
            b.off2, b.qoff2, b.len2 = rb2.seq_off, rb2.qual_off, rb2.seq_len
            irregular = irregular or not np.array_equal(rb2.seq_len, rb2.qual_len)
        if irregular:
            b.qlen1 = np.ascontiguousarray(rb1.qual_len, dtype=np.uint32)
            if rb2 is not None:
                b.qlen2 = np.ascontiguousarray(rb2.qual_len, dtype=np.uint32)
        return b

    def set_aux(self, lane, tile, x, y, ok):
        self.aux = (np.ascontiguousarray(lane, dtype=np.int32), np.ascontiguousarray(tile, dtype=np.int32),
                    np.ascontiguousarray(x, dtype=np.int32), np.ascontiguousarray(y, dtype=np.int32),
                    np.ascontiguousarray(ok, dtype=np.uint8))

    def max_len(self):
        m = int(self.len1.max()) if self.n else 0
        if self.len2 is not None and self.n:
            m = max(m, int(self.len2.max()))
        return m

    def as_struct(self):
        s = BatchStruct()
        s.n = self.n
        s.first_index = self.first_index
        s.seq1, s.qual1, s.off1, s.len1 = _ptr(self.seq1), _ptr(self.qual1), _ptr(self.off1), _ptr(self.len1)
        s.qoff1 = _ptr(self.qoff1)
        s.bytes1 = 0 if self.seq1 is None else self.seq1.size
        s.qbytes1 = 0 if self.qual1 is None else self.qual1.size
        s.seq2, s.qual2, s.off2, s.len2 = _ptr(self.seq2), _ptr(self.qual2), _ptr(self.off2), _ptr(self.len2)
        s.qoff2 = _ptr(self.qoff2)
        s.bytes2 = 0 if self.seq2 is None else self.seq2.size
        s.qbytes2 = 0 if self.qual2 is None else self.qual2.size
        if self.aux is not None:
            s.aux_lane, s.aux_tile, s.aux_x, s.aux_y, s.aux_ok = (_ptr(a) for a in self.aux)
        s.qlen1, s.qlen2 = _ptr(self.qlen1), _ptr(self.qlen2)
        return s

    def read1(self, i):
        o = int(self.off1[i]); l = int(self.len1[i])
        q = o if self.qoff1 is None else int(self.qoff1[i])
        ql = l if self.qlen1 is None else int(self.qlen1[i])
        return self.seq1[o:o + l].tobytes(), self.qual1[q:q + ql].tobytes()

    def read2(self, i):
        o = int(self.off2[i]); l = int(self.len2[i])
        q = o if self.qoff2 is None else int(self.qoff2[i])
        ql = l if self.qlen2 is None else int(self.qlen2[i])
        return self.seq2[o:o + l].tobytes(), self.qual2[q:q + ql].tobytes()


class AqcError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "afterqc_hip error %d (%s): %s" % (code, ERRORS.get(code, "?"), msg))
        self.code = code


_lib = None


def load_library():
    """Load libafterqc_hip.so or raise: the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP library %s is missing: build it with `python __graft_entry__.py build` "
                           "(hipcc --offload-arch=gfx950); afterqc_amd has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    lib.aqc_abi_version.restype = C.c_int
    lib.aqc_device_count.restype = C.c_int
    lib.aqc_last_error.restype = C.c_char_p
    lib.aqc_create.argtypes = [C.c_int, C.c_int, C.POINTER(P)]
    lib.aqc_destroy.argtypes = [P]
    lib.aqc_destroy.restype = None
    lib.aqc_device_name.argtypes = [P, C.c_char_p, C.c_int]
    lib.aqc_set_config.argtypes = [P, C.POINTER(Config)]
    lib.aqc_set_circles.argtypes = [P, P, P, P, P, P, C.c_int32]
    lib.aqc_reset_stats.argtypes = [P]
    lib.aqc_upload.argtypes = [P, C.c_int, C.POINTER(BatchStruct)]
    lib.aqc_run.argtypes = [P, C.c_int, C.c_uint64]
    lib.aqc_qc_stat.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int]
    lib.aqc_fetch_results.argtypes = [P, C.c_int, P, C.c_uint64]
    lib.aqc_format_spans.argtypes = [P, C.c_int, C.c_uint64, C.c_int32, P, P]
    lib.aqc_fetch_span_events.argtypes = [P, C.c_int, C.c_int, P, C.c_uint64]
    lib.aqc_span_end.argtypes = [P, C.c_int, C.c_uint64, P]
    lib.aqc_format_fused.argtypes = [P, C.c_int]
    lib.aqc_fetch_quality_views.argtypes = [P, C.c_int, C.c_int, P, C.c_uint64]
    lib.aqc_error_record.argtypes = [P, C.c_int, C.POINTER(C.c_uint64)]
    lib.aqc_sync.argtypes = [P, C.c_int]
    lib.aqc_last_deferred.argtypes = [P, C.c_int, P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.aqc_last_deferred.restype = C.c_int
    lib.aqc_kernel_ms.argtypes = [P, C.c_int, P]
    lib.aqc_timing_reset.argtypes = [P, C.c_int]
    lib.aqc_timing_mean.argtypes = [P, C.c_int, P, P]
    lib.aqc_get_counters.argtypes = [P, P]
    lib.aqc_get_histograms.argtypes = [P, P, P, C.c_int32]
    lib.aqc_get_qc.argtypes = [P, C.c_int, P]
    lib.aqc_get_kmers.argtypes = [P, C.c_int, P, P, P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.aqc_overlap.argtypes = [P, C.POINTER(BatchStruct), P, P, P]
    lib.aqc_read_stats.argtypes = [P, C.POINTER(BatchStruct), C.c_int32, C.c_int32, C.c_int32, P, P, P]
    lib.aqc_edit_distance.argtypes = [P, C.POINTER(BatchStruct), P]
    lib.aqc_frame.argtypes = [P, C.c_int, C.POINTER(TextChunk), C.POINTER(FrameInfo)]
    lib.aqc_frame_mixed.argtypes = [P, C.c_int, C.POINTER(TextChunk), C.POINTER(TextExtent), C.c_uint64, C.c_uint8, C.POINTER(TextExtent), C.c_uint64, C.c_uint8,
                                    C.POINTER(FrameInfo)]
    lib.aqc_frame_mixed.restype = C.c_int
    lib.aqc_reframe.argtypes = [P, C.c_int, C.POINTER(FrameInfo)]
    lib.aqc_reframe.restype = C.c_int
    lib.aqc_format.argtypes = [P, C.c_int, C.c_uint64, C.c_int32, P]
    lib.aqc_format_plain.argtypes = [P, C.c_int, C.c_int, C.c_uint64, C.c_int32, P]
    lib.aqc_fetch_text.argtypes = [P, C.c_int, C.c_int, C.c_int, P, C.c_uint64]
    lib.aqc_fetch_streams.argtypes = [P, C.c_int, C.c_int32, C.POINTER(C.c_void_p * 6), C.POINTER(C.c_uint64 * 6)]
    lib.aqc_gunzip_dev.argtypes = [C.c_int, P, C.c_uint64, P, C.c_uint64, C.POINTER(C.c_uint64), P, C.c_int, C.c_uint64, C.c_uint64]
    lib.aqc_compress.argtypes = [P, C.c_int, C.c_int32, P]
    lib.aqc_fetch_gz.argtypes = [P, C.c_int, C.c_int, C.c_int, P, C.c_uint64]
    lib.aqc_pipe_create.argtypes = [C.POINTER(P), C.c_int32, C.c_int32, C.c_int32, C.POINTER(P)]
    lib.aqc_pipe_create.restype = C.c_int
    lib.aqc_pipe_destroy.argtypes = [P]
    lib.aqc_pipe_destroy.restype = None
    lib.aqc_pipe_run.argtypes = [P, C.POINTER(PipeIO), C.POINTER(PipeOpts), C.POINTER(PipeResult)]
    lib.aqc_pipe_run.restype = C.c_int
    lib.aqc_pipe_last_error.restype = C.c_char_p
    lib.aqc_source_open.argtypes = [C.c_char_p, C.c_int32, C.c_int32]
    lib.aqc_source_open.restype = P
    lib.aqc_source_read.argtypes = [P, P, C.c_uint64]
    lib.aqc_source_read.restype = C.c_int64
    lib.aqc_source_close.argtypes = [P]
    lib.aqc_source_close.restype = None
    lib.aqc_source_open2.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_uint64]
    lib.aqc_source_open2.restype = P
    lib.aqc_source_error.argtypes = [P]
    lib.aqc_source_error.restype = C.c_char_p
    lib.aqc_source_gz_stats.argtypes = [P, C.POINTER(C.c_uint64 * 4)]
    lib.aqc_source_gz_stats.restype = C.c_int
    lib.aqc_gz_input_stats.argtypes = [C.POINTER(C.c_uint64 * 4)]
    lib.aqc_gz_input_stats.restype = C.c_int
    lib.aqc_gz_deflate_block.argtypes = [P, C.c_uint64, C.c_int32, P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.aqc_gz_deflate_block.restype = C.c_int
    lib.aqc_gz_inflate_raw.argtypes = [P, C.c_uint64, P, C.c_uint64]
    lib.aqc_gz_inflate_raw.restype = C.c_int64
    lib.aqc_gz_crc32.argtypes = [C.c_uint32, P, C.c_uint64]
    lib.aqc_gz_crc32.restype = C.c_uint32
    lib.aqc_host_count_newlines.argtypes = [P, C.c_uint64]
    lib.aqc_host_count_newlines.restype = C.c_uint64
    lib.aqc_bgzf_compress.argtypes = [P, C.c_uint64, C.c_int32, P, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.aqc_bgzf_compress.restype = C.c_int
    lib.aqc_pipe_split.argtypes = [C.POINTER(PipeIO), C.c_int32, C.c_uint64, C.c_int32, P, P, C.c_uint64, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint32)]
    lib.aqc_pipe_split.restype = C.c_int
    lib.aqc_host_alloc.argtypes = [C.c_uint64]
    lib.aqc_host_alloc.restype = C.c_void_p
    lib.aqc_host_free.argtypes = [P]
    lib.aqc_host_free.restype = None
    for name in ("aqc_create", "aqc_device_name", "aqc_set_config", "aqc_set_circles", "aqc_reset_stats", "aqc_upload",
                 "aqc_run", "aqc_qc_stat", "aqc_fetch_results", "aqc_sync", "aqc_kernel_ms", "aqc_timing_reset",
                 "aqc_timing_mean", "aqc_get_counters", "aqc_get_histograms", "aqc_get_qc", "aqc_get_kmers",
                 "aqc_overlap", "aqc_read_stats", "aqc_edit_distance", "aqc_frame", "aqc_frame_mixed", "aqc_reframe", "aqc_format", "aqc_format_plain",
                 "aqc_fetch_text", "aqc_fetch_quality_views", "aqc_error_record", "aqc_format_spans", "aqc_fetch_span_events", "aqc_span_end", "aqc_format_fused"):
        getattr(lib, name).restype = C.c_int
    if lib.aqc_abi_version() != 3:
        raise RuntimeError("libafterqc_hip.so ABI version mismatch")
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ["aqc_abi_version", "aqc_device_count", "aqc_device_index", "aqc_device_numa_node", "aqc_device_numa_node_of", "aqc_bind_thread_to_node", "aqc_last_error", "aqc_create", "aqc_destroy",
                    "aqc_device_name", "aqc_set_config", "aqc_set_circles", "aqc_reset_stats", "aqc_upload", "aqc_run",
                    "aqc_qc_stat", "aqc_fetch_results", "aqc_fetch_quality_views", "aqc_error_record", "aqc_sync", "aqc_last_deferred", "aqc_kernel_ms", "aqc_timing_reset",
                    "aqc_timing_mean", "aqc_get_counters",
                    "aqc_get_histograms", "aqc_get_qc", "aqc_get_kmers", "aqc_overlap", "aqc_read_stats",
                    "aqc_edit_distance", "aqc_frame", "aqc_frame_mixed", "aqc_reframe", "aqc_format", "aqc_format_spans", "aqc_fetch_span_events", "aqc_span_end", "aqc_format_fused", "aqc_format_plain", "aqc_fetch_text", "aqc_fetch_streams", "aqc_compress", "aqc_fetch_gz", "aqc_gunzip_dev", "aqc_host_alloc",
                    "aqc_host_free",
                    "aqc_pipe_create", "aqc_pipe_destroy", "aqc_pipe_run", "aqc_pipe_last_error",
                    "aqc_host_count_newlines", "aqc_bgzf_compress", "aqc_pipe_split",
                    "aqc_source_open", "aqc_source_open2", "aqc_source_read", "aqc_source_error", "aqc_source_gz_stats", "aqc_gz_input_stats", "aqc_source_close",
                    "aqc_gz_deflate_block", "aqc_gz_inflate_raw", "aqc_gz_crc32",
                    # the reference's own native seam (editdistance/_editdistance.h:16,23), same names and signatures
                    "edit_distance", "seek_overlap"]


class Engine:
    """One GPU context (struct aqc_ctx).  Thin, 1:1 with the C ABI."""

    def __init__(self, device=0, n_slots=2):
        self.lib = load_library()
        if self.lib.aqc_device_count() <= 0:
            raise RuntimeError("afterqc_amd: no AMD GPU visible (aqc_device_count() == 0); the HIP path is the only "
                               "path — there is no CPU fallback")
        h = C.c_void_p()
        self._check(self.lib.aqc_create(device, n_slots, C.byref(h)))
        self.h = h
        self.n_slots = n_slots
        self.slot_n = [0] * n_slots

    def _check(self, rc):
        if rc != 0:
            raise AqcError(rc, (self.lib.aqc_last_error() or b"").decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "h", None):
            self.lib.aqc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.aqc_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_config(self, cfg):
        self._check(self.lib.aqc_set_config(self.h, C.byref(cfg)))

    def set_circles(self, circles):
        """circles: list of (x, y, radius, lane, tile) as loaded from circles.csv"""
        n = len(circles)
        cx = np.array([c[0] for c in circles], dtype=np.float64)
        cy = np.array([c[1] for c in circles], dtype=np.float64)
        cr = np.array([c[2] for c in circles], dtype=np.float64)
        ln = np.array([c[3] for c in circles], dtype=np.int32)
        tl = np.array([c[4] for c in circles], dtype=np.int32)
        self._check(self.lib.aqc_set_circles(self.h, _ptr(cx), _ptr(cy), _ptr(cr), _ptr(ln), _ptr(tl), n))

    def reset_stats(self):
        self._check(self.lib.aqc_reset_stats(self.h))

    def upload(self, slot, batch):
        s = batch.as_struct()
        self._check(self.lib.aqc_upload(self.h, slot, C.byref(s)))
        self.slot_n[slot] = batch.n

    def run(self, slot, accum_limit=UINT64_MAX):
        self._check(self.lib.aqc_run(self.h, slot, accum_limit))

    def qc_stat(self, slot, which, mate, first, count, post):
        self._check(self.lib.aqc_qc_stat(self.h, slot, which, mate, first, count, 1 if post else 0))

    def fetch_results(self, slot):
        out = np.zeros(self.slot_n[slot], dtype=RESULT_DTYPE)
        self._check(self.lib.aqc_fetch_results(self.h, slot, _ptr(out), self.slot_n[slot]))
        return out

    def sync(self, slot):
        self._check(self.lib.aqc_sync(self.h, slot))

    def fetch_quality_views(self, slot, mate):
        """(start, length) of the quality-string slice that goes with the final read of every record of the slot"""
        out = np.zeros(self.slot_n[slot], dtype=np.uint32)
        self._check(self.lib.aqc_fetch_quality_views(self.h, slot, mate, _ptr(out), self.slot_n[slot]))
        return (out & 0xffff).astype(np.int64), (out >> 16).astype(np.int64)

    def error_record(self, slot):
        """after an AqcError whose code is in RECORD_ERRORS: the index (in the slot) of the earliest record at which the
        reference's run would have died, or None when the error is not tied to a record"""
        rec = C.c_uint64(0)
        self._check(self.lib.aqc_error_record(self.h, slot, C.byref(rec)))
        return None if rec.value == UINT64_MAX else int(rec.value)

    def last_deferred(self, slot, want_indices=False):
        """records of the slot's last run() that the lane-per-read kernel handed to the general kernel"""
        n = C.c_uint64(0)
        cap = self.slot_n[slot] if want_indices else 0
        idx = np.zeros(max(cap, 1), dtype=np.uint32)
        self._check(self.lib.aqc_last_deferred(self.h, slot, _ptr(idx), cap, C.byref(n)))
        return (int(n.value), idx[:min(int(n.value), cap)]) if want_indices else int(n.value)

    def kernel_ms(self, slot):
        ms = np.zeros(N_KERNELS, dtype=np.float32)
        self._check(self.lib.aqc_kernel_ms(self.h, slot, _ptr(ms)))
        return ms

    def timing_reset(self, slot):
        self._check(self.lib.aqc_timing_reset(self.h, slot))

    def timing_mean(self, slot):
        ms = np.zeros(N_KERNELS, dtype=np.float32)
        n = np.zeros(N_KERNELS, dtype=np.int32)
        self._check(self.lib.aqc_timing_mean(self.h, slot, _ptr(ms), _ptr(n)))
        return ms, n

    def counters(self):
        out = np.zeros(N_COUNTERS, dtype=np.int64)
        self._check(self.lib.aqc_get_counters(self.h, _ptr(out)))
        return out

    def histograms(self, n=AQC_QC_COLS):
        a = np.zeros(n, dtype=np.int64)
        b = np.zeros(n, dtype=np.int64)
        self._check(self.lib.aqc_get_histograms(self.h, _ptr(a), _ptr(b), n))
        return a, b

    def qc(self, which):
        out = np.zeros((QC_ROWS, AQC_QC_COLS), dtype=np.int64)
        self._check(self.lib.aqc_get_qc(self.h, which, _ptr(out)))
        return out

    def kmers(self, which, cap=1 << 22):
        keys = np.zeros(cap, dtype=np.uint64)
        counts = np.zeros(cap, dtype=np.int64)
        order = np.zeros(cap, dtype=np.uint64)
        n = C.c_uint64(0)
        self._check(self.lib.aqc_get_kmers(self.h, which, _ptr(keys), _ptr(counts), _ptr(order), cap, C.byref(n)))
        m = n.value
        return keys[:m], counts[:m], order[:m]

    # ---- text in / text out ------------------------------------------------------------------
    def host_buffer(self, nbytes):
        return HostBuffer(self.lib, nbytes)

    def frame(self, slot, text1, bytes1, final1, text2=None, bytes2=0, final2=False, max_records=UINT64_MAX, first_index=0):
        """aqc_frame: text1/text2 are numpy uint8 arrays (or HostBuffer.array) holding raw FASTQ text."""
        ch = TextChunk()
        ch.text1, ch.bytes1, ch.final1 = text1.ctypes.data, int(bytes1), 1 if final1 else 0
        if text2 is not None:
            ch.text2, ch.bytes2, ch.final2 = text2.ctypes.data, int(bytes2), 1 if final2 else 0
        ch.max_records, ch.first_index = int(max_records), int(first_index)
        info = FrameInfo()
        self._check(self.lib.aqc_frame(self.h, slot, C.byref(ch), C.byref(info)))
        self.slot_n[slot] = info.n
        return info

    def reframe(self, slot):
        """aqc_reframe: frame the text the slot already holds in HBM again (no host copy)"""
        info = FrameInfo()
        self._check(self.lib.aqc_reframe(self.h, slot, C.byref(info)))
        self.slot_n[slot] = info.n
        return info

    def format(self, slot, n, store_overlap=False):
        """sizes[file * 3 + stream], stream 0 good / 1 bad / 2 overlap"""
        sizes = np.zeros(6, dtype=np.uint64)
        self._check(self.lib.aqc_format(self.h, slot, int(n), 1 if store_overlap else 0, _ptr(sizes)))
        return [int(x) for x in sizes]

    def format_spans(self, slot, n, store_overlap=False):
        """aqc_format_spans: like format(), but the good records that go out as their own bytes are left out of stream 0;
        -> (bytes per stream, events per file)"""
        sizes = (C.c_uint64 * 6)()
        n_ev = (C.c_uint64 * 2)()
        self._check(self.lib.aqc_format_spans(self.h, slot, n, 1 if store_overlap else 0, sizes, n_ev))
        return [int(x) for x in sizes], [int(x) for x in n_ev]

    def fetch_span_events(self, slot, file, n_events):
        ev = np.zeros(max(n_events, 1), dtype=SPAN_EVENT_DTYPE)
        self._check(self.lib.aqc_fetch_span_events(self.h, slot, file, _ptr(ev), n_events))
        return ev[:n_events]

    def format_fused(self, slot):
        """did the slot's last format() take the placement the verdict kernel made (AQC_FUSED=1)?"""
        rc = self.lib.aqc_format_fused(self.h, slot)
        if rc < 0:
            self._check(rc)
        return rc == 1

    def span_end(self, slot, n):
        end = (C.c_uint64 * 2)()
        self._check(self.lib.aqc_span_end(self.h, slot, n, end))
        return [int(x) for x in end]

    def format_plain(self, slot, verdict_slot, n, store_overlap=False):
        """index files: whole records of `slot`, routed / renamed by the verdicts of `verdict_slot`"""
        sizes = np.zeros(6, dtype=np.uint64)
        self._check(self.lib.aqc_format_plain(self.h, slot, verdict_slot, int(n), 1 if store_overlap else 0, _ptr(sizes)))
        return [int(x) for x in sizes]

    def compress(self, slot, level=2):
        """the slot's formatted streams as gzip members, built on the device -> compressed bytes per stream"""
        sizes = np.zeros(6, dtype=np.uint64)
        self._check(self.lib.aqc_compress(self.h, slot, int(level), _ptr(sizes)))
        return [int(x) for x in sizes]

    def fetch_gz(self, slot, file, stream, dst, cap):
        self._check(self.lib.aqc_fetch_gz(self.h, slot, file, stream, dst.ctypes.data if dst is not None else None, int(cap)))

    def fetch_text(self, slot, file, stream, dst, cap):
        """dst: numpy uint8 array with room for the stream (sizes from format())"""
        self._check(self.lib.aqc_fetch_text(self.h, slot, file, stream, dst.ctypes.data if dst is not None else None, int(cap)))

    # ---- function seams --------------------------------------------------------------------
    def overlap(self, batch):
        n = batch.n
        off = np.zeros(n, dtype=np.int32); ol = np.zeros(n, dtype=np.int32); df = np.zeros(n, dtype=np.int32)
        s = batch.as_struct()
        self._check(self.lib.aqc_overlap(self.h, C.byref(s), _ptr(off), _ptr(ol), _ptr(df)))
        return off, ol, df

    def read_stats(self, batch, max_poly, mismatch, qual):
        n = batch.n
        px = np.zeros(n, dtype=np.uint8); lq = np.zeros(n, dtype=np.int32); nn = np.zeros(n, dtype=np.int32)
        s = batch.as_struct()
        self._check(self.lib.aqc_read_stats(self.h, C.byref(s), max_poly, mismatch, qual, _ptr(px), _ptr(lq), _ptr(nn)))
        return px, lq, nn

    def edit_distance(self, batch):
        d = np.zeros(batch.n, dtype=np.int32)
        s = batch.as_struct()
        self._check(self.lib.aqc_edit_distance(self.h, C.byref(s), _ptr(d)))
        return d


class Pipe:
    """struct aqc_pipe: the whole-input pipeline (C++ reader / slot-worker / writer threads) over one or more engines —
    one engine per GPU; chunk i of the input goes to engine i % len(engines)."""

    def __init__(self, engines, slots=3, io_threads=0):
        self.lib = load_library()
        self.engines = list(engines)
        arr = (C.c_void_p * len(self.engines))(*[e.h for e in self.engines])
        h = C.c_void_p()
        rc = self.lib.aqc_pipe_create(arr, len(self.engines), slots, io_threads, C.byref(h))
        if rc != 0:
            raise AqcError(rc, "aqc_pipe_create failed")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.aqc_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, inputs, outputs=None, gzip_in=(False, False), gzip_out=False, gzip_level=2, chunk_records=0, qc_sample=200000,
            store_overlap=False, no_output=False, chunk_index0=0, chunk_index_stride=1):
        """inputs: list of 1-2 file names, or of (numpy uint8 array / HostBuffer.array, nbytes) tuples (text in memory);
        outputs: per input (good, bad, overlap) file names or None.  Returns a PipeResult."""
        io = PipeIO()
        keep = []
        for k, src in enumerate(inputs):
            if src is None:
                continue
            if isinstance(src, (str, bytes)):
                b = src.encode() if isinstance(src, str) else src
                keep.append(b)
                io.in_path[k] = b
                io.gzip_in[k] = int(gzip_in[k]) if gzip_in[k] else 0        # 1 gzip, 2 bzip2
            else:
                arr, nbytes = src
                keep.append(arr)
                io.in_mem[k] = arr.ctypes.data
                io.in_mem_bytes[k] = int(nbytes)
        if outputs is not None:
            for k, trio in enumerate(outputs):
                for st, name in enumerate(trio or ()):
                    if name is not None:
                        b = name.encode() if isinstance(name, str) else name
                        keep.append(b)
                        io.out_path[k][st] = b
        io.gzip_out = 1 if gzip_out else 0
        io.gzip_level = int(gzip_level)
        opts = PipeOpts(int(chunk_records), int(qc_sample), 1 if store_overlap else 0, 1 if no_output else 0, int(chunk_index0),
                        int(chunk_index_stride))
        res = PipeResult()
        rc = self.lib.aqc_pipe_run(self.h, C.byref(io), C.byref(opts), C.byref(res))
        if rc != 0:
            raise AqcError(rc, (self.lib.aqc_pipe_last_error() or b"").decode("utf-8", "replace"))
        for e in self.engines:
            e.slot_n = [0] * e.n_slots
        return res


class NativeSource:
    """A read file as a binary stream with readinto() — the pipe's readers behind Python's file protocol (parallel pread,
    member-parallel BGZF inflate): what afterqc_amd.fastq.open_binary hands out for plain and .gz files."""

    def __init__(self, path, gzip_in, io_threads=0, gz_section_bytes=0):
        self.lib = load_library()
        self.h = self.lib.aqc_source_open2(path.encode() if isinstance(path, str) else path, int(gzip_in) if gzip_in else 0, int(io_threads),
                                           int(gz_section_bytes))
        if not self.h:
            raise IOError("cannot open " + str(path))

    def gz_stats(self):
        """(sections accepted, sections discarded, bytes decoded sequentially, bytes out) of the parallel gunzip"""
        out = (C.c_uint64 * 4)()
        self.lib.aqc_source_gz_stats(self.h, C.byref(out))
        return tuple(int(x) for x in out)

    def readinto(self, view):
        mv = memoryview(view)
        n = mv.nbytes
        if n == 0:
            return 0
        arr = np.frombuffer(mv, dtype=np.uint8)
        got = self.lib.aqc_source_read(self.h, arr.ctypes.data, n)
        if got < 0:
            raise IOError((self.lib.aqc_source_error(self.h) or b"read error").decode("utf-8", "replace"))
        return int(got)

    def read(self, n=-1):
        out = bytearray()
        step = n if n is not None and n >= 0 else (8 << 20)
        while True:
            buf = bytearray(step)
            got = self.readinto(buf)
            out += buf[:got]
            if got < step or (n is not None and n >= 0):
                break
        return bytes(out)

    def close(self):
        if getattr(self, "h", None):
            self.lib.aqc_source_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bgzf_compress(data, level=2):
    """the pipe's .gz writer on a bytes object: BGZF-style members (valid gzip)"""
    lib = load_library()
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    cap = len(data) + len(data) // 100 + 64 * (len(data) // 0xff00 + 2)
    dst = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    rc = lib.aqc_bgzf_compress(src.ctypes.data, len(data), level, dst.ctypes.data, cap, C.byref(n))
    if rc != 0:
        raise AqcError(rc, "aqc_bgzf_compress failed")
    return dst[:n.value].tobytes()


def pipe_split(source, chunk_records, gzip_in=False, io_threads=4, cap=1 << 16):
    """the pipe's reader half alone (no GPU): (bytes per chunk, lines per chunk, crc32 of the concatenated chunks)"""
    lib = load_library()
    io = PipeIO()
    keep = []
    if isinstance(source, (str, bytes)) and not isinstance(source, bytes):
        b = source.encode()
        keep.append(b)
        io.in_path[0] = b
        io.gzip_in[0] = int(gzip_in) if gzip_in else 0        # 1 gzip, 2 bzip2
    else:
        arr = np.frombuffer(source, dtype=np.uint8) if len(source) else np.zeros(1, dtype=np.uint8)
        keep.append(arr)
        io.in_mem[0] = arr.ctypes.data
        io.in_mem_bytes[0] = len(source)
    nbytes = np.zeros(cap, dtype=np.uint64)
    lines = np.zeros(cap, dtype=np.uint64)
    n = C.c_uint64(0)
    crc = C.c_uint32(0)
    rc = lib.aqc_pipe_split(C.byref(io), 0, int(chunk_records), io_threads, nbytes.ctypes.data, lines.ctypes.data, cap, C.byref(n),
                            C.byref(crc))
    if rc != 0:
        raise AqcError(rc, (lib.aqc_pipe_last_error() or b"").decode("utf-8", "replace"))
    k = min(int(n.value), cap)
    return nbytes[:k].tolist(), lines[:k].tolist(), int(crc.value)


def merge_kmer_parts(parts):
    """k-mer dictionaries of several engines / ranks as one: counts add up, the first-seen key (the GLOBAL index of the read that
    inserted the k-mer, stamped by the device) takes the minimum — so that ties order exactly like a sequential run
    (qualitycontrol.py:113-122,155-156).  parts: (keys, counts, order) triples; returns one."""
    parts = list(parts)
    if len(parts) == 1:
        return parts[0]
    keys = np.concatenate([p[0] for p in parts])
    counts = np.concatenate([p[1] for p in parts])
    order = np.concatenate([p[2] for p in parts])
    uk, inv = np.unique(keys, return_inverse=True)
    c = np.zeros(len(uk), dtype=np.int64)
    np.add.at(c, inv, counts)
    o = np.full(len(uk), np.iinfo(np.uint64).max, dtype=np.uint64)
    np.minimum.at(o, inv, order)
    return uk, c, o


def collect_stats(engine, paired=True):
    """Every statistic of one engine as plain numpy (picklable: what a rank of a multi-process run sends to rank 0)."""
    whichs = (0, 1, 2, 3) if paired else (QC_R1_PRE, QC_R1_POST)
    ovl, dist = engine.histograms()
    return {"counters": engine.counters(), "ovl": ovl, "dist": dist, "qc": {w: engine.qc(w) for w in whichs},
            "kmers": {w: tuple(np.asarray(a) for a in engine.kmers(w)) for w in whichs}}


def merge_stats(parts):
    """SURVEY.md §8e, the ONE merge rule of this package (MergedEngines uses the same pieces): counters, histograms and QC rows
    are summed, k-mer dictionaries go through merge_kmer_parts.  No collective: integers on the host."""
    parts = list(parts)
    out = {"counters": sum(p["counters"] for p in parts), "ovl": sum(p["ovl"] for p in parts), "dist": sum(p["dist"] for p in parts),
           "qc": {}, "kmers": {}}
    for w in parts[0]["qc"]:
        out["qc"][w] = sum(p["qc"][w] for p in parts)
        out["kmers"][w] = merge_kmer_parts([p["kmers"][w] for p in parts])
    return out


def top_kmers(kmers, kmer_len, top=10):
    """sortKmer (qualitycontrol.py:155-156) on a (keys, counts, order) triple: count descending, insertion order for ties"""
    keys, counts, order = kmers
    idx = sorted(range(len(keys)), key=lambda i: (-int(counts[i]), int(order[i])))[:top]
    return [[int(keys[i]).to_bytes(8, "little")[:kmer_len].decode("latin-1"), int(counts[i])] for i in idx]


class MergedEngines:
    """Statistics of several engines seen as one (per-GPU integers are summed on the host; k-mer dictionaries
    merge by count sum and smallest first-seen key — merge_kmer_parts).  Read-only: counters / histograms / qc / kmers."""

    def __init__(self, engines):
        self.engines = list(engines)

    def counters(self):
        return sum(e.counters() for e in self.engines)

    def histograms(self, n=AQC_QC_COLS):
        hs = [e.histograms(n) for e in self.engines]
        return sum(h[0] for h in hs), sum(h[1] for h in hs)

    def qc(self, which):
        return sum(e.qc(which) for e in self.engines)

    def kmers(self, which, cap=1 << 22):
        return merge_kmer_parts([e.kmers(which, cap) for e in self.engines])
