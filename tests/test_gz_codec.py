"""The pipe's own gzip codec (csrc/aqc_inflate.cpp, aqc_gunzip.cpp, aqc_deflate.cpp) against zlib / Python's gzip module — what
fastq.py:23-24,65-68 uses upstream.  No GPU: the codec is host code inside libafterqc_hip.so."""
import gzip
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from afterqc_amd import capi, synth  # noqa: E402


def _fastq_text(n_pairs, seed):
    d = synth.make_pairs(n_pairs, 150, seed=seed, dirty=True)
    buf, n = synth.render_fastq_fixed(d["seq1"], d["qual1"], 1)
    return bytes(memoryview(buf)[:n])


@pytest.fixture(scope="module")
def gzb_selftest_exe(tmp_path_factory):
    """tests/native/gzb_selftest.cpp built once for the tests of this module that run it"""
    exe = str(tmp_path_factory.mktemp("gzb") / "gzb_selftest")
    src = [os.path.join(ROOT, "tests", "native", "gzb_selftest.cpp")] + [os.path.join(ROOT, "afterqc_amd", "csrc", f) for f in ("aqc_inflate.cpp", "aqc_gunzip.cpp")]
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-pthread", "-Wno-stringop-overflow"] + src + ["-lz", "-o", exe])
    return exe


def test_native_selftest(tmp_path):
    """inflate vs zlib streams of every level / strategy, deflate_block -> zlib, ParallelGunzip over single- and multi-member
    files in small sections, truncated / corrupted / bad-CRC / trailing-garbage inputs (tests/native/gz_selftest.cpp)"""
    exe = str(tmp_path / "gz_selftest")
    src = [os.path.join(ROOT, "tests", "native", "gz_selftest.cpp")] + [os.path.join(ROOT, "afterqc_amd", "csrc", f) for f in
                                                                      ("aqc_inflate.cpp", "aqc_gunzip.cpp", "aqc_deflate.cpp")]
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-pthread", "-Wno-stringop-overflow"] + src + ["-lz", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "all gz codec checks passed" in out.stdout


def test_device_gunzip_logic_on_the_cpu(gzb_selftest_exe):
    """csrc/aqc_gunzip_dev.hpp's per-lane functions (block-start scan, table builder, block decoder, chain walk, marker
    re-basing) dealt out by plain loops and plugged into the real ParallelGunzip through the SectionOffload interface the GPU
    uses (tests/native/gzb_selftest.cpp): zlib streams of every level and strategy, flush-separated blocks, concatenated
    members, damaged and truncated files — byte-identical or a loud error, and the offloaded sections really are committed"""
    exe = gzb_selftest_exe
    # (every case runs in two modes — symbols handed back / resolved by the decoder: two processes side by side, half the wall time)
    procs = [subprocess.Popen([exe, str(mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for mode in (0, 1)]
    for pr in procs:
        so, se = pr.communicate(timeout=900)
        assert pr.returncode == 0, so[-3000:] + se[-2000:]
        assert "all device-gunzip logic checks passed" in so


@pytest.mark.parametrize("level", [1, 6, 9])
def test_device_gunzip_logic_on_gnu_gzip_files(tmp_path, gzb_selftest_exe, level):
    """files written by the `gzip` PROGRAM (GNU gzip closes a block every 32 K tokens, twice zlib's) through the same CPU emulation
    of the device decoder in its file mode, which runs with the kernels' own budget (6 slices of 2048 tokens per lane, 12 x symbol
    space): byte-identical, and the emulated device supplied sections"""
    import shutil
    if not shutil.which("gzip"):
        pytest.skip("no gzip program")
    exe = gzb_selftest_exe
    d = synth.make_pairs(30000, 150, seed=300 + level, dirty=True)
    buf, n = synth.render_fastq_fixed(d["seq1"], d["qual1"], 1)
    plain = str(tmp_path / "t.fq")
    with open(plain, "wb") as f:
        f.write(memoryview(buf)[:n])
    with open(plain + ".gz", "wb") as g:
        subprocess.check_call(["gzip", "-%d" % level, "-c", plain], stdout=g)
    out = subprocess.run([exe, plain + ".gz", plain, str(256 << 10), str(2 << 20), "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]
    line = out.stdout.strip().splitlines()[-1]
    assert " ok " in line and "failed 0" in line, line


def test_pool_lane_policy(tmp_path):
    """csrc/aqc_pool.hpp: queued front jobs (somebody waits for them: the gunzip consumer's translation pieces) are all started
    before any further background job (speculative sections); help_front() lets the waiting thread run them itself;
    parallel_for completes with every worker stuck in background work (tests/native/pool_selftest.cpp)"""
    exe = str(tmp_path / "pool_selftest")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "pool_selftest.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]
    assert "all pool checks passed" in out.stdout


def _read_all(path, section=0, threads=4, piece=1 << 20):
    s = capi.NativeSource(path, True, io_threads=threads, gz_section_bytes=section)
    out = bytearray()
    buf = bytearray(piece)
    try:
        while True:
            k = s.readinto(buf)
            out += buf[:k]
            if k < piece:
                break
        return bytes(out), s.gz_stats()
    finally:
        s.close()


@pytest.mark.parametrize("level", [1, 2, 6, 9])
def test_python_gzip_single_member_parallel(tmp_path, level):
    """a single-member .gz as Python's gzip module writes it (what upstream's own outputs and most inputs are): decoded by
    speculative sections on several threads, byte-identical, and the sections really are used"""
    text = _fastq_text(12000, 31 + level)
    p = str(tmp_path / "a.fq.gz")
    with gzip.open(p, "wb", compresslevel=level) as f:
        f.write(text)
    for section in (64 << 10, 256 << 10, 0):
        got, st = _read_all(p, section)
        assert got == text
        if section == 64 << 10:
            assert st[0] >= 8 and st[3] == len(text), st       # accepted sections, bytes out
            assert st[2] < len(text) // 4, st                   # little was decoded sequentially


def test_multi_member_and_bgzf_and_mixed(tmp_path):
    a, b, c = _fastq_text(3000, 1), _fastq_text(2000, 2), _fastq_text(2500, 3)
    # cat of three gzip files, the middle one written by the pipe's own writer (BGZF members)
    p = str(tmp_path / "cat.gz")
    with open(p, "wb") as f:
        f.write(gzip.compress(a, 4))
        f.write(capi.bgzf_compress(b, 2))
        f.write(gzip.compress(c, 1))
    assert _read_all(p, 64 << 10)[0] == a + b + c
    # BGZF first (member-parallel path), a plain member behind it
    p2 = str(tmp_path / "bgzf_first.gz")
    with open(p2, "wb") as f:
        f.write(capi.bgzf_compress(b, 2))
        f.write(gzip.compress(c, 6))
    assert _read_all(p2)[0] == b + c
    # the pipe's writer output reads back with Python's gzip too (and levels 0 .. 9 all give valid members)
    for level in (0, 1, 2, 6, 9):
        assert gzip.decompress(capi.bgzf_compress(a, level)) == a
    assert gzip.decompress(capi.bgzf_compress(b"", 2)) == b""


def test_empty_and_tiny(tmp_path):
    p = str(tmp_path / "e.gz")
    open(p, "wb").close()
    assert _read_all(p)[0] == b""
    with gzip.open(p, "wb") as f:
        pass
    assert _read_all(p)[0] == b""
    with gzip.open(p, "wb") as f:
        f.write(b"@r\nA\n+\nI\n")
    assert _read_all(p)[0] == b"@r\nA\n+\nI\n"


def test_damaged_inputs_fail_loudly(tmp_path):
    """a truncated or corrupt input must be an error (gzip.open raises upstream), never a clean end of file"""
    text = _fastq_text(8000, 5)
    gz = gzip.compress(text, 6)
    cases = {"truncated": gz[:len(gz) * 2 // 3], "flipped": gz[:len(gz) // 2] + bytes([gz[len(gz) // 2] ^ 0x20]) + gz[len(gz) // 2 + 1:],
             "bad_crc": gz[:-6] + bytes([gz[-6] ^ 1]) + gz[-5:], "garbage_behind": gz + b"junk", "not_gzip": b"@r\nACGT\n+\nIIII\n" * 100}
    for name, data in cases.items():
        p = str(tmp_path / (name + ".gz"))
        with open(p, "wb") as f:
            f.write(data)
        with pytest.raises(IOError):
            _read_all(p, 64 << 10)
    # BGZF: one corrupted byte inside a member, and a member cut short
    bg = capi.bgzf_compress(text, 2)
    bad = bytearray(bg)
    bad[len(bg) // 2] ^= 0x40
    for name, data in (("bgzf_flipped", bytes(bad)), ("bgzf_truncated", bg[:len(bg) - 100])):
        p = str(tmp_path / (name + ".gz"))
        with open(p, "wb") as f:
            f.write(data)
        with pytest.raises(IOError):
            _read_all(p)


def test_deflate_block_and_crc_through_the_c_abi():
    lib = capi.load_library()
    rng = np.random.default_rng(4)
    text = np.frombuffer(_fastq_text(3000, 8), dtype=np.uint8)
    for n in (0, 1, 17, 4096, 0xff00, 200000):
        src = np.ascontiguousarray(text[:n])
        for level in (0, 1, 2, 6):
            dst = np.zeros(n + n // 1000 + 400, dtype=np.uint8)
            m = capi.C.c_uint64(0)
            assert lib.aqc_gz_deflate_block(src.ctypes.data, n, level, dst.ctypes.data, dst.size, capi.C.byref(m)) == 0
            raw = dst[:m.value].tobytes()
            assert zlib.decompress(raw, -15) == src.tobytes()
            back = np.zeros(n + 1, dtype=np.uint8)
            assert lib.aqc_gz_inflate_raw(dst.ctypes.data, m.value, back.ctypes.data, n) == n
            assert back[:n].tobytes() == src.tobytes()
    for n in (0, 1, 15, 16, 127, 128, 129, 1000, 65536, 1 << 20):
        b = rng.integers(0, 256, n, dtype=np.uint8)
        c0 = int(rng.integers(0, 1 << 32))
        assert lib.aqc_gz_crc32(c0, b.ctypes.data, n) == zlib.crc32(b.tobytes(), c0)


def _read_all_bz2(path, threads=4, piece=777777):
    s = capi.NativeSource(path, 2, io_threads=threads)
    out = bytearray()
    buf = bytearray(piece)
    try:
        while True:
            k = s.readinto(buf)
            out += buf[:k]
            if k < piece:
                break
        return bytes(out)
    finally:
        s.close()


def test_bzip2_input_through_the_pipes_own_source(tmp_path):
    """fastq.py:25-26 opens .bz2 inputs with bz2.BZ2File; here libbz2 (loaded at run time) decodes on the pipe's own threads
    (csrc/aqc_pipe.cpp, Bz2Source): a plain bzip2 file (one stream), a pbzip2-style file (many streams, decoded in parallel,
    delivered in order), an empty stream in between, levels 1 and 9 — byte for byte what python's bz2 module makes of them; a
    truncated file, a corrupted block and a file that is no bzip2 file are errors, never a clean end of input"""
    import bz2
    text = _fastq_text(30000, 12)
    one = bz2.compress(text, 9)
    pieces = [text[i:i + 700001] for i in range(0, len(text), 700001)]
    many = b"".join(bz2.compress(p, 1 if k % 2 else 9) for k, p in enumerate(pieces))
    with_empty = bz2.compress(pieces[0]) + bz2.compress(b"") + bz2.compress(b"".join(pieces[1:]))
    for name, data, want in (("one", one, text), ("many", many, text), ("with_empty", with_empty, text), ("empty", bz2.compress(b""), b"")):
        p = str(tmp_path / (name + ".fq.bz2"))
        with open(p, "wb") as f:
            f.write(data)
        assert bz2.decompress(data) == want
        assert _read_all_bz2(p) == want, name
        assert _read_all_bz2(p, threads=1, piece=4096 if len(want) < (1 << 20) else 1 << 20) == want, name
    bad = bytearray(one)
    bad[len(one) // 2] ^= 0x55
    for name, data in (("truncated", one[:len(one) * 2 // 3]), ("flipped", bytes(bad)), ("not_bzip2", b"@r\nACGT\n+\nIIII\n" * 100),
                       ("many_truncated", many[:len(many) - 1000])):
        p = str(tmp_path / (name + ".fq.bz2"))
        with open(p, "wb") as f:
            f.write(data)
        with pytest.raises(IOError):
            _read_all_bz2(p)
    # ... and the reader half of the pipe over it: chunks of exactly K records, the same bytes (CRC-32 of the concatenated chunks)
    p = str(tmp_path / "many.fq.bz2")
    nbytes, lines, crc = capi.pipe_split(p, 4096, gzip_in=2)
    assert sum(nbytes) == len(text) and sum(lines) == text.count(b"\n") and all(x == 4 * 4096 for x in lines[:-1]) and crc == (zlib.crc32(text) & 0xffffffff)
