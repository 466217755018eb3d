"""FASTQ record framing and writing with the reference's exact rules (fastq.py:37-49, 87-93),
vectorised with numpy so that a whole chunk of text is framed at once and handed to the GPU as a
zero-copy SoA batch: the raw text chunk IS the byte arena, (offset, length) pairs address the
sequence and quality lines inside it.

Rules reproduced:
  * a record is 4 lines, each `readline().rstrip()` (Python-2 byte semantics: strips " \\t\\n\\r\\v\\f");
  * a line that is empty after stripping ends the file: the partial record is dropped;
  * `.gz` / `.bz2` inputs are decoded transparently (fastq.py:23-28);
  * the writer emits line + "\\n" for each of the 4 lines; `.gz` output iff the name ends in .gz or
    gzip is forced, at the given compression level (fastq.py:63-76).
"""
import bz2
import gzip
import os
import re
import sys
import zlib

import numpy as np

_WS = np.zeros(256, dtype=bool)
for _c in b" \t\n\r\x0b\x0c":
    _WS[_c] = True


def isFastq(f):
    """fastq.py:7-12"""
    for ext in (".fq", ".fastq", ".fq.gz", ".fastq.gz", ".fq.bz2", ".fastq.bz2"):
        if f.endswith(ext):
            return True
    return False


def open_binary(fname, sample=False):
    """the byte stream fastq.Reader reads (fastq.py:23-28): .gz / .bz2 decoded transparently.  sample=True: the caller reads
    only the head of the file (the pre-filter sampling pass): a .gz is then decoded with a few small sections in flight instead
    of speculating hundreds of megabytes ahead"""
    try:
        if fname.endswith(".bz2"):
            return bz2.BZ2File(fname)
        try:
            # the native readers (parallel pread; .gz: the pipe's own decoder, many threads per stream)
            from . import capi
            if sample and fname.endswith(".gz"):
                return capi.NativeSource(fname, True, io_threads=6, gz_section_bytes=1 << 20)
            return capi.NativeSource(fname, fname.endswith(".gz"))
        except (ImportError, RuntimeError, AttributeError):
            pass
        if fname.endswith(".gz"):
            return gzip.open(fname, "rb")
        return open(fname, "rb", buffering=0)
    except (IOError, OSError):
        print("Failed to open file " + fname)
        sys.exit(1)


class RawBatch:
    """n framed records inside one text buffer: per line kind an (offset, length) array pair."""

    __slots__ = ("text", "n", "name_off", "name_len", "seq_off", "seq_len", "plus_off", "plus_len", "qual_off",
                 "qual_len")

    def line(self, kind, i):
        off = getattr(self, kind + "_off")[i]
        ln = getattr(self, kind + "_len")[i]
        return self.text[off:off + ln].tobytes()

    def record(self, i):
        """[name, seq, plus, qual] as bytes — the reference's 4-element read list"""
        return [self.line("name", i), self.line("seq", i), self.line("plus", i), self.line("qual", i)]


class Reader:
    """Chunked, vectorised counterpart of fastq.Reader.  `next_batch(n)` returns up to n records
    (fewer only at end of file, None when exhausted)."""

    PAD = 64  # zero bytes kept after the text so device vector loads never leave the buffer

    def __init__(self, fname, bytes_per_record_hint=400):
        self.filename = fname
        try:
            if fname.endswith(".gz"):
                self._f = gzip.open(fname, "rb")
            elif fname.endswith(".bz2"):
                self._f = bz2.BZ2File(fname)
            else:
                self._f = open(fname, "rb")
        except (IOError, OSError):
            print("Failed to open file " + fname)
            sys.exit(1)
        self._left = b""
        self._file_eof = False
        self._eof = False
        self._bpr = bytes_per_record_hint

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def _frame(self, data, final):
        """Frame the complete 4-line groups in `data` (`final`: nothing more will be read).
        Returns (starts[nrec,4], lens[nrec,4], nrec, consumed_bytes, eof_marker)."""
        buf = np.frombuffer(data, dtype=np.uint8)
        nl = np.flatnonzero(buf == 10)
        nlines = len(nl)
        starts = np.empty(nlines + 1, dtype=np.int64)
        starts[0] = 0
        starts[1:] = nl + 1
        ends = np.empty(nlines + 1, dtype=np.int64)
        ends[:nlines] = nl
        ends[nlines] = len(buf)
        if final and len(buf) > 0 and (nlines == 0 or nl[-1] != len(buf) - 1):
            nlines += 1  # unterminated last line of the file
        nrec = nlines // 4
        starts = starts[:nrec * 4]
        ends = ends[:nrec * 4].copy()
        # rstrip: peel trailing whitespace (usually at most one "\r")
        while True:
            idx = np.flatnonzero(ends > starts)
            if len(idx) == 0:
                break
            w = _WS[buf[ends[idx] - 1]]
            if not w.any():
                break
            ends[idx[w]] -= 1
        lens = ends - starts
        consumed = int(nl[nrec * 4 - 1]) + 1 if (nrec > 0 and nrec * 4 - 1 < len(nl)) else (len(buf) if nrec > 0 else 0)
        # the first empty line ends the file (fastq.py:44-47); a partial group at EOF is dropped
        marker = False
        empties = np.flatnonzero(lens == 0)
        if len(empties):
            nrec = int(empties[0]) // 4
            marker = True
        return starts[:nrec * 4].reshape(nrec, 4), lens[:nrec * 4].reshape(nrec, 4), nrec, consumed, marker

    def next_batch(self, nmax):
        if self._eof:
            return None
        want = nmax
        data = self._left
        while True:
            s, l, nrec, consumed, marker = self._frame(data, self._file_eof)
            if nrec >= want or self._file_eof or marker:
                break
            need = max(1 << 16, int((want - nrec) * self._bpr * 1.05) + 4096)
            more = self._f.read(need)
            if not more:
                self._file_eof = True
            else:
                data = data + more if data else more
        if nrec > 0:
            self._bpr = max(16.0, float(s[nrec - 1, 3] + l[nrec - 1, 3] + 1) / nrec)
        take = min(nrec, want)
        if take == 0:
            self._eof = True
            self._left = b""
            return None
        if take < nrec:
            # more records framed than asked for: keep the tail for the next call
            cut = int(s[take, 0])
            self._left = data[cut:]
        else:
            if marker or self._file_eof:
                self._eof = True
                self._left = b""
            else:
                self._left = data[consumed:]
        rb = RawBatch()
        end_text = int(s[take - 1, 3] + l[take - 1, 3])
        text = np.zeros(end_text + self.PAD, dtype=np.uint8)
        text[:end_text] = np.frombuffer(data, dtype=np.uint8, count=end_text)
        rb.text = text
        rb.n = take
        s = s[:take]
        l = l[:take]
        rb.name_off = np.ascontiguousarray(s[:, 0]).astype(np.uint64); rb.name_len = np.ascontiguousarray(l[:, 0]).astype(np.uint32)
        rb.seq_off = np.ascontiguousarray(s[:, 1]).astype(np.uint64); rb.seq_len = np.ascontiguousarray(l[:, 1]).astype(np.uint32)
        rb.plus_off = np.ascontiguousarray(s[:, 2]).astype(np.uint64); rb.plus_len = np.ascontiguousarray(l[:, 2]).astype(np.uint32)
        rb.qual_off = np.ascontiguousarray(s[:, 3]).astype(np.uint64); rb.qual_len = np.ascontiguousarray(l[:, 3]).astype(np.uint32)
        return rb

    def nextRead(self):
        """fastq.Reader.nextRead (fastq.py:37-49): one record as [name, seq, strand, qual] (bytes) or None."""
        rb = self.next_batch(1)
        if rb is None:
            return None
        return rb.record(0)


_POOL = None


def _pool():
    """shared worker threads for gzip members (zlib releases the GIL)"""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(2, min(32, (os.cpu_count() or 4))))
    return _POOL


def _gzip_member(data, level):
    """one block of text as gzip members.  With the native library present: BGZF-style 64 KiB members (the pipe's writer,
    aqc_bgzf_compress — a reader can inflate them in parallel); else one plain member from Python's zlib."""
    try:
        from . import capi
        return capi.bgzf_compress(bytes(data), level)
    except Exception:
        c = zlib.compressobj(level, zlib.DEFLATED, 31)       # wbits 31: a complete gzip member (header + CRC trailer)
        return c.compress(data) + c.flush()


class ParallelGzipFile:
    """Write-only .gz file made of independent gzip members (RFC 1952 allows concatenation; gzip / zcat / Python's
    gzip module read them as one stream), so that compression runs on many cores: every write() is cut into
    blocks that are deflated concurrently and written in order.  The DECOMPRESSED bytes are what the reference's
    gzip.open(..., "w", compresslevel) would have produced; the container bytes differ (upstream's embed mtime and
    file name anyway)."""

    BLOCK = 1 << 20

    def __init__(self, filename, level):
        self._f = open(filename, "wb")
        self._level = level
        self._wrote = False

    def write(self, data):
        mv = memoryview(data)
        if len(mv) == 0:
            return
        if len(mv) <= self.BLOCK:
            self._f.write(_gzip_member(mv, self._level))
        else:
            futs = [_pool().submit(_gzip_member, mv[o:o + self.BLOCK], self._level) for o in range(0, len(mv), self.BLOCK)]
            for fu in futs:
                self._f.write(fu.result())
        self._wrote = True

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._wrote:
            self._f.write(_gzip_member(b"", self._level))      # an empty but valid gzip file
        self._f.close()


class Writer:
    """fastq.Writer (fastq.py:56-104) over bytes."""

    def __init__(self, fname, force_gzip=False, gzip_compression=2):
        self.filename = fname
        if not self.filename.endswith(".gz") and force_gzip:
            self.filename = self.filename + ".gz"
        if self.filename.endswith(".gz"):
            self._f = ParallelGzipFile(self.filename, gzip_compression)
        elif self.filename.endswith(".bz2"):
            print("ERROR: Write bzip2 stream is not supported")
            sys.exit(1)
        else:
            self._f = open(self.filename, "wb")

    def flush(self):
        if self._f is not None:
            self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.flush()
            self._f.close()
            self._f = None

    def writeLines(self, lines):
        if self._f is None:
            return False
        self._f.write(b"".join(line + b"\n" for line in lines))
        return True

    def write_bytes(self, data):
        self._f.write(data)


# preprocesser.py:155 — the Illumina name pattern searched by isInBubble
_NAME_RE = re.compile(r'\S+\:\d+\:\S+\:\d+\:\d+\:\d+\:\d+')


def parse_illumina_name(name):
    """The host half of isInBubble (preprocesser.py:180-192): (ok, lane, tile, x, y).
    tile drops the first character of the tile field exactly like `int(tile_no[1:])`."""
    if isinstance(name, bytes):
        name = name.decode("latin-1")
    m = _NAME_RE.search(name)
    if not m:
        return (0, 0, 0, 0, 0)
    items = m.group().split(":")
    if len(items) < 7:
        return (0, 0, 0, 0, 0)
    lane = int(items[3])
    tile = int(items[4][1:])
    x = int(items[5])
    y = int(items[6])
    return (1, lane, tile, x, y)
