"""CPU: the host halves of the whole-input pipe (aqc_pipe.cpp) — newline counting, the reader's record-aligned chunking
over plain / gzip / BGZF / memory sources, the BGZF writer — through the host-only entry points of the C ABI (no GPU)."""
import gzip
import os
import random
import zlib

import numpy as np
import pytest

from afterqc_amd import capi


def fastq_text(n, seed, ragged=True, crlf=False, final_newline=True):
    rng = random.Random(seed)
    eol = "\r\n" if crlf else "\n"
    out = []
    for i in range(n):
        l = rng.randint(5, 300) if ragged else 150
        out.append("@r%d some:name:%d%s%s%s+%s%s%s" % (i, rng.randint(0, 10 ** 6), eol, "".join(rng.choice("ACGTN") for _ in range(l)), eol,
                                                    eol, "".join(rng.choice("#/6<AE") for _ in range(l)), eol))
    t = "".join(out)
    return (t if final_newline else t[:-len(eol)]).encode()


def test_count_newlines_matches_python():
    lib = capi.load_library()
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 8, 9, 63, 64, 127, 128, 129, 1000, 100003):
        a = rng.integers(0, 40, n, dtype=np.uint8)           # plenty of 0x0a among them
        for off in (0, 1, 3):
            v = a[off:]
            buf = np.ascontiguousarray(v)
            assert lib.aqc_host_count_newlines(buf.ctypes.data if len(buf) else None, len(buf)) == int((buf == 10).sum())


@pytest.mark.parametrize("K", [1, 7, 100, 4096])
@pytest.mark.parametrize("kind", ["mem", "file", "gz_stream", "gz_members", "bgzf"])
def test_reader_cuts_record_aligned_chunks(tmp_path, K, kind):
    text = fastq_text(5000, 11 + K)
    total_lines = text.count(b"\n")
    if kind == "mem":
        src, gz = text, False
    else:
        p = str(tmp_path / ("in.fq" + ("" if kind == "file" else ".gz")))
        if kind == "file":
            open(p, "wb").write(text)
        elif kind == "gz_stream":
            with gzip.open(p, "wb", compresslevel=1) as f:
                f.write(text)
        elif kind == "gz_members":                              # concatenated plain members (no BGZF extra field)
            with open(p, "wb") as f:
                for o in range(0, len(text), 70001):
                    f.write(gzip.compress(text[o:o + 70001], 1))
        else:
            open(p, "wb").write(capi.bgzf_compress(text, 2))
        src, gz = p, kind != "file"
    nbytes, lines, crc = capi.pipe_split(src, K, gzip_in=gz)
    assert crc == zlib.crc32(text)
    assert sum(nbytes) == len(text) and sum(lines) == total_lines
    assert all(l == 4 * K for l in lines[:-1]) and 0 < lines[-1] <= 4 * K
    # every chunk ends at a record boundary: replay the cuts on the text
    pos = 0
    for b, l in zip(nbytes, lines):
        piece = text[pos:pos + b]
        assert piece.count(b"\n") == l and piece.endswith(b"\n")
        pos += b


def test_reader_edge_shapes(tmp_path):
    # empty input, no final newline, CRLF, a partial trailing record
    assert capi.pipe_split(b"", 10)[:2] == ([0], [0])
    t = fastq_text(37, 3, final_newline=False)
    nbytes, lines, crc = capi.pipe_split(t, 10)
    assert sum(nbytes) == len(t) and crc == zlib.crc32(t) and lines == [40, 40, 40, 28]     # the unterminated last line counts
    t = fastq_text(25, 4, crlf=True)
    nbytes, lines, crc = capi.pipe_split(t, 8)
    assert lines == [32, 32, 32, 4] and crc == zlib.crc32(t)
    t = fastq_text(12, 5) + b"@tail\nACGT\n"
    nbytes, lines, crc = capi.pipe_split(t, 5)
    assert lines == [20, 20, 10] and sum(nbytes) == len(t)                                   # the dispatcher flags lines % 4 != 0 as irregular


def test_bgzf_writer_is_valid_gzip():
    rng = random.Random(9)
    for n in (0, 1, 1000, 0xff00, 0xff01, 300000):
        data = bytes(rng.choice(b"ACGTN#E\n") for _ in range(n))
        z = capi.bgzf_compress(data, 2)
        assert gzip.decompress(z) == data
        assert z[:4] == b"\x1f\x8b\x08\x04" and z[12:14] == b"BC"
        # every member announces its own size (BSIZE) and holds at most 64 KiB of text
        pos, members = 0, 0
        while pos < len(z):
            bsize = z[pos + 16] + (z[pos + 17] << 8) + 1
            isize = int.from_bytes(z[pos + bsize - 4:pos + bsize], "little")
            assert isize <= 0xff00
            pos += bsize
            members += 1
        assert pos == len(z) and members == max(1, (n + 0xff00 - 1) // 0xff00)
