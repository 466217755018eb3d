"""-m gpu: ONE input dealt over several contexts by the whole-input pipe gives byte-identical outputs and
an identical stats JSON to the serial single-context chunk loop — chunk i carries first_index = i * chunk_records, so the
post-filter sampling rule (TOTAL_READS < qc_sample, preprocesser.py:624) and the k-mer dictionary's insertion order survive
the sharding; counters / histograms / QC rows are plain sums, k-mers merge by (count sum, min first-seen key)."""
import hashlib
import json
import os

import numpy as np
import pytest

from afterqc_amd import after, capi, preprocesser, synth

pytestmark = pytest.mark.gpu


def run(work, r1, r2, extra, **kw):
    out = os.path.join(work, kw.pop("tag"))
    argv = ["-1", r1] + (["-2", r2] if r2 else []) + ["-g", os.path.join(out, "good"), "-b", os.path.join(out, "bad"),
                                                       "-r", os.path.join(out, "QC")] + extra
    options, _ = after.parseCommand(argv)
    after.finalize_options(options)
    options.barcode = False
    flt = preprocesser.seqFilter(options, **kw)
    stat = flt.run()
    files = {}
    for sub in ("good", "bad"):
        for fn in sorted(os.listdir(os.path.join(out, sub))):
            with open(os.path.join(out, sub, fn), "rb") as f:
                files[sub + "/" + fn] = hashlib.sha256(f.read()).hexdigest()
    stat = json.loads(json.dumps(stat))
    for k in ("good_output_folder", "bad_output_folder", "report_output_folder"):
        stat["command"].pop(k, None)
    return files, stat, flt


@pytest.mark.parametrize("paired", [True, False])
def test_one_input_over_two_contexts(tmp_path, paired):
    work = str(tmp_path)
    n = 60000
    d = synth.make_pairs(n, 150, seed=8801, dirty=True)
    r1, r2 = os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq")
    synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
    synth.write_fastq_fixed(r2, d["seq2"], d["qual2"], 2)
    extra = ["-f", "2", "-t", "3", "--qc_sample", "30000"]        # the sample ends inside a chunk, several chunks behind it
    a_files, a_stat, a = run(work, r1, r2 if paired else None, extra, tag="serial", use_pipe=False, devices=[0])
    b_files, b_stat, b = run(work, r1, r2 if paired else None, extra, tag="two", use_pipe=True, devices=[0, 0], chunk_records=4096, pipe_slots=3)
    c_files, c_stat, c = run(work, r1, r2 if paired else None, extra, tag="one", use_pipe=True, devices=[0], chunk_records=7000, pipe_slots=2)
    assert b.used_pipe and c.used_pipe and not a.used_pipe
    assert a_files == b_files == c_files
    assert a_stat == b_stat == c_stat
    assert a_stat["afterqc_main_summary"]["total_reads"] == n


def test_one_million_pairs_over_four_contexts_on_one_device(tmp_path):
    """the N-GPU code path at a size where every context sees hundreds of chunks: ONE 1 M-pair input dealt over FOUR contexts on
    GPU 0 (what AQC_DEVICES=0,0,0,0 sets up: per-device DMA gates, NUMA binding of the slot workers, host-side merge of four
    contexts' statistics) — outputs byte-identical to the one-context run, statistics JSON identical"""
    work = str(tmp_path)
    n = 1_000_000
    d = synth.make_pairs(n, 150, seed=8844, workers=4)
    r1, r2 = os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq")
    synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
    synth.write_fastq_fixed(r2, d["seq2"], d["qual2"], 2)
    extra = ["-f", "0", "-t", "0"]
    a_files, a_stat, a = run(work, r1, r2, extra, tag="one", use_pipe=True, devices=[0], chunk_records=1 << 14)
    b_files, b_stat, b = run(work, r1, r2, extra, tag="four", use_pipe=True, devices=[0, 0, 0, 0], chunk_records=3000, pipe_slots=3)
    assert a.used_pipe and b.used_pipe
    assert a_files == b_files
    assert a_stat == b_stat
    assert a_stat["afterqc_main_summary"]["total_reads"] == n


def test_two_million_pairs_over_eight_contexts_on_one_device(tmp_path):
    """what an 8-GPU node sets up, on one device: AQC_DEVICES=0,0,0,0,0,0,0,0 — eight contexts, 24 slot workers, one pair of DMA
    gates, the I/O threads bound to the (one) NUMA node of all eight — over ONE 2 M-pair input: byte-identical outputs and an
    identical statistics JSON to the one-context run"""
    work = str(tmp_path)
    n = 2_000_000
    d = synth.make_pairs(n, 150, seed=8888, workers=8)
    r1, r2 = os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq")
    synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
    synth.write_fastq_fixed(r2, d["seq2"], d["qual2"], 2)
    del d
    extra = ["-f", "0", "-t", "0"]
    a_files, a_stat, a = run(work, r1, r2, extra, tag="one", use_pipe=True, devices=[0], chunk_records=1 << 15)
    b_files, b_stat, b = run(work, r1, r2, extra, tag="eight", use_pipe=True, devices=[0] * 8, chunk_records=5000, pipe_slots=3)
    assert a.used_pipe and b.used_pipe
    assert a_files == b_files
    assert a_stat == b_stat
    assert a_stat["afterqc_main_summary"]["total_reads"] == n


def test_pipe_gzip_in_and_out(tmp_path):
    """.gz in (our own BGZF-style multi-member files: inflated member-parallel; and a single-member stream) -> .gz out"""
    import gzip
    work = str(tmp_path)
    d = synth.make_pairs(20000, 150, seed=8802, dirty=True)
    r1, r2 = os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq")
    synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
    synth.write_fastq_fixed(r2, d["seq2"], d["qual2"], 2)
    ref_files, ref_stat, _ = run(work, r1, r2, ["-f", "0", "-t", "0"], tag="plain", use_pipe=False, devices=[0])
    # single-member gzip inputs -> BGZF outputs; then those outputs' siblings as BGZF inputs
    for p in (r1, r2):
        with open(p, "rb") as f, gzip.open(p + ".gz", "wb", compresslevel=1) as g:
            g.write(f.read())
    out = os.path.join(work, "gz1")
    argv = ["-1", r1 + ".gz", "-2", r2 + ".gz", "-f", "0", "-t", "0", "-g", os.path.join(out, "good"), "-b", os.path.join(out, "bad"), "-r", os.path.join(out, "QC")]
    options, _ = after.parseCommand(argv)
    after.finalize_options(options)
    options.barcode = False
    flt = preprocesser.seqFilter(options, use_pipe=True, devices=[0], chunk_records=3000)
    flt.run()
    assert flt.used_pipe
    good1 = os.path.join(out, "good", "R1.good.fq.gz")
    with gzip.open(good1, "rb") as f:
        data = f.read()
    assert hashlib.sha256(data).hexdigest() == ref_files["good/R1.good.fq"]
    with open(good1, "rb") as f:
        head = f.read(16)
    assert head[:4] == b"\x1f\x8b\x08\x04" and head[12:14] == b"BC"       # BGZF members
    # BGZF in: feed the good outputs back in (every read is good again or at least the pipe must frame them all)
    out2 = os.path.join(work, "gz2")
    argv = ["-1", good1, "-2", os.path.join(out, "good", "R2.good.fq.gz"), "-f", "0", "-t", "0", "-g", os.path.join(out2, "good"),
            "-b", os.path.join(out2, "bad"), "-r", os.path.join(out2, "QC")]
    options, _ = after.parseCommand(argv)
    after.finalize_options(options)
    options.barcode = False
    flt2 = preprocesser.seqFilter(options, use_pipe=True, devices=[0], chunk_records=2500)
    st2 = flt2.run()
    assert flt2.used_pipe
    assert st2["afterqc_main_summary"]["total_reads"] == ref_stat["afterqc_main_summary"]["good_reads"]
    flt3 = preprocesser.seqFilter(options, use_pipe=False, devices=[0])
    options.good_output_folder = os.path.join(work, "gz3", "good")
    options.bad_output_folder = os.path.join(work, "gz3", "bad")
    options.report_output_folder = os.path.join(work, "gz3", "QC")
    st3 = flt3.run()
    for k in ("good_output_folder", "bad_output_folder", "report_output_folder"):
        st2["command"].pop(k, None); st3["command"].pop(k, None)
    assert json.dumps(st2, sort_keys=True) == json.dumps(st3, sort_keys=True)
    for sub, fn in (("good", "R1.good.good.fq.gz"), ("bad", "R2.good.bad.fq.gz")):
        with gzip.open(os.path.join(out2, sub, fn), "rb") as f, gzip.open(os.path.join(work, "gz3", sub, fn), "rb") as g:
            assert f.read() == g.read()


def test_pipe_reports_irregular_inputs(tmp_path):
    """mates of different lengths / an empty line inside: the pipe steps aside, the serial loop reproduces the reference"""
    work = str(tmp_path)
    d = synth.make_pairs(3000, 100, seed=8803)
    r1, r2 = os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq")
    synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
    synth.write_fastq_fixed(r2, d["seq2"][:2500], d["qual2"][:2500], 2)
    files, stat, flt = run(work, r1, r2, ["-f", "0", "-t", "0"], tag="short", use_pipe=True, devices=[0], chunk_records=512)
    assert not flt.used_pipe
    assert stat["afterqc_main_summary"]["total_reads"] == 2500
    files2, stat2, _ = run(work, r1, r2, ["-f", "0", "-t", "0"], tag="short_serial", use_pipe=False, devices=[0])
    assert files == files2 and stat == stat2


def test_device_gzip_members(gpu_engine):
    """aqc_compress / aqc_fetch_gz: the formatted streams as gzip members built on the device decompress (Python's gzip
    module = zlib: header, dynamic-Huffman block, CRC-32, ISIZE all checked by it) to exactly what aqc_fetch_text hands out;
    odd sizes: a stream of less than one member, of exactly one, with a short tail member; and the ratio is sane"""
    import gzip
    for n_pairs, L, seed in ((40, 100, 5), (94, 150, 6), (30000, 150, 7), (2500, 250, 8)):
        d = synth.make_pairs(n_pairs, L, seed=seed, dirty=True)
        t1, n1 = synth.render_fastq_fixed(d["seq1"], d["qual1"], 1)
        t2, n2 = synth.render_fastq_fixed(d["seq2"], d["qual2"], 2)
        cfg = capi.Config()
        cfg.paired = 1
        cfg.seq_len_req, cfg.poly_size_limit, cfg.allow_mismatch_in_poly = 35, 35, 2
        cfg.qualified_quality_phred, cfg.unqualified_base_limit, cfg.n_base_limit = 15, 60, 5
        cfg.barcode_length = 12
        cfg.set_verify("CAGTA")
        cfg.qc_kmer = 8
        gpu_engine.set_config(cfg)
        gpu_engine.reset_stats()
        info = gpu_engine.frame(0, t1, n1, True, t2, n2, True)
        gpu_engine.run(0)
        sizes = gpu_engine.format(0, int(info.n), True)
        gz = gpu_engine.compress(0, 2)
        assert sum(sizes) > 0
        for q, (nb, zb) in enumerate(zip(sizes, gz)):
            assert (nb == 0) == (zb == 0)
            if not nb:
                continue
            text = np.zeros(nb + 64, dtype=np.uint8)
            gpu_engine.fetch_text(0, q // 3, q % 3, text, text.size)
            comp = np.zeros(zb + 64, dtype=np.uint8)
            gpu_engine.fetch_gz(0, q // 3, q % 3, comp, comp.size)
            back = gzip.decompress(comp[:zb].tobytes())
            assert back == text[:nb].tobytes(), (n_pairs, q, nb, zb, len(back))
            if nb > 200000:
                assert zb < nb / 2.2, (nb, zb)                  # FASTQ: well below half


@pytest.mark.parametrize("level,strategy,sec_kb", [(1, "default", 64), (6, "default", 64), (9, "default", 256), (6, "huffman", 64), (6, "fixed", 64), (6, "rle", 64),
                                                   (0, "default", 64)])
def test_device_gunzip_sections_are_exact(level, strategy, sec_kb):
    """csrc/aqc_gunzip_dev.hpp through aqc_gunzip_dev: ONE gzip member, every section handed to the device (a lane per deflate
    block; the blocks are found by scanning every bit position), committed only when they chain up bit for bit, markers
    resolved and CRC-32 checked on the host — the result is exactly zlib's for dynamic, literal-only, run-length, fixed-Huffman
    and stored streams; and for the dynamic ones the device really supplied the sections"""
    import zlib
    d = synth.make_pairs(9000, 150, seed=40 + level, dirty=True)
    buf, n = synth.render_fastq_fixed(d["seq1"], d["qual1"], 1)
    text = bytes(memoryview(buf)[:n])
    st = {"default": zlib.Z_DEFAULT_STRATEGY, "huffman": zlib.Z_HUFFMAN_ONLY, "fixed": zlib.Z_FIXED, "rle": zlib.Z_RLE}[strategy]
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, st)
    gz = c.compress(text) + c.flush()
    lib = capi.load_library()
    src = np.frombuffer(gz, dtype=np.uint8)
    out = np.zeros(len(text) + 4096, dtype=np.uint8)
    n_out = capi.C.c_uint64(0)
    stats = np.zeros(8, dtype=np.uint64)
    rc = lib.aqc_gunzip_dev(0, src.ctypes.data, len(gz), out.ctypes.data, out.size, capi.C.byref(n_out), stats.ctypes.data, 4, sec_kb << 10, 1 << 20)
    assert rc == 0, (lib.aqc_last_error() or b"").decode()
    assert n_out.value == len(text) and out[:len(text)].tobytes() == text
    if strategy not in ("fixed",) and level > 0:
        # (fixed-Huffman and stored streams carry no block header the scan looks for: the host decodes them)
        assert stats[0] >= 1 and stats[0] >= stats[1], stats
        assert stats[2] < len(text) // 2, stats


def test_device_gunzip_takes_concatenated_members_and_pigz_style_sync_blocks():
    """members glued together (cat a.gz b.gz) and empty stored blocks between the deflate blocks (Z_FULL_FLUSH, what pigz writes
    between its chunks): the chain steps over stored blocks on the device, member ends go through the host — exact either way"""
    import zlib
    d = synth.make_pairs(12000, 150, seed=77, dirty=True)
    buf, n = synth.render_fastq_fixed(d["seq1"], d["qual1"], 1)
    text = bytes(memoryview(buf)[:n])
    third = len(text) // 3
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = b""
    for k in range(0, third, 200_000):
        gz += c.compress(text[k:min(third, k + 200_000)]) + c.flush(zlib.Z_FULL_FLUSH)
    gz += c.flush()
    c2 = zlib.compressobj(1, zlib.DEFLATED, 31)
    gz += c2.compress(text[third:2 * third]) + c2.flush()
    c3 = zlib.compressobj(9, zlib.DEFLATED, 31)
    gz += c3.compress(text[2 * third:]) + c3.flush()
    lib = capi.load_library()
    src = np.frombuffer(gz, dtype=np.uint8)
    out = np.zeros(len(text) + 4096, dtype=np.uint8)
    n_out = capi.C.c_uint64(0)
    stats = np.zeros(8, dtype=np.uint64)
    rc = lib.aqc_gunzip_dev(0, src.ctypes.data, len(gz), out.ctypes.data, out.size, capi.C.byref(n_out), stats.ctypes.data, 4, 64 << 10, 512 << 10)
    assert rc == 0, (lib.aqc_last_error() or b"").decode()
    assert n_out.value == len(text) and out[:len(text)].tobytes() == text
    assert stats[0] >= 3, stats
    # a damaged stream is an error, not silence
    bad = bytearray(gz)
    bad[len(bad) // 2] ^= 0x5a
    srcb = np.frombuffer(bytes(bad), dtype=np.uint8)
    rc = lib.aqc_gunzip_dev(0, srcb.ctypes.data, len(bad), out.ctypes.data, out.size, capi.C.byref(n_out), stats.ctypes.data, 4, 64 << 10, 512 << 10)
    assert rc != 0


def test_pipe_single_member_gzip_input_is_shared_with_the_device(tmp_path):
    """one-member `gzip -6` inputs big enough to be cut into many sections, through aqc_pipe_run: the GPU takes groups of sections
    off the host pool (ParallelGunzip + DeviceInflate), the outputs equal the plain-input run byte for byte, and the device
    supplied more than 30 % of the committed sections; with AQC_GZ_DEVICE_IN=0 the host does it all, same bytes"""
    import gzip
    work = str(tmp_path)
    d = synth.make_pairs(400_000, 150, seed=8811, workers=4)
    r1, r2 = os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq")
    synth.write_fastq_fixed(r1, d["seq1"], d["qual1"], 1)
    synth.write_fastq_fixed(r2, d["seq2"], d["qual2"], 2)
    ref_files, ref_stat, _ = run(work, r1, r2, ["-f", "0", "-t", "0"], tag="plain", use_pipe=True, devices=[0])
    for p in (r1, r2):
        with open(p, "rb") as f, gzip.open(p + ".gz", "wb", compresslevel=6) as g:
            g.write(f.read())
    lib = capi.load_library()

    def gz_run(tag):
        before = (capi.C.c_uint64 * 4)()
        lib.aqc_gz_input_stats(capi.C.byref(before))
        out = os.path.join(work, tag)
        argv = ["-1", r1 + ".gz", "-2", r2 + ".gz", "-f", "0", "-t", "0", "--compression", "0", "-g", os.path.join(out, "good"), "-b", os.path.join(out, "bad"),
                "-r", os.path.join(out, "QC")]
        options, _ = after.parseCommand(argv)
        after.finalize_options(options)
        options.barcode = False
        flt = preprocesser.seqFilter(options, use_pipe=True, devices=[0])
        flt.run()
        assert flt.used_pipe
        after_ = (capi.C.c_uint64 * 4)()
        lib.aqc_gz_input_stats(capi.C.byref(after_))
        for name in ("good/R1.good.fq", "good/R2.good.fq", "bad/R1.bad.fq", "bad/R2.bad.fq"):
            with gzip.open(os.path.join(out, name + ".gz"), "rb") as f:
                assert hashlib.sha256(f.read()).hexdigest() == ref_files[name], (tag, name)
        return [int(a) - int(b) for a, b in zip(after_, before)]

    os.environ["AQC_GZ_GROUP"] = str(16 << 20)       # (a 40 MB file: one group of 64 sections per mate, a third of the file)
    os.environ["AQC_GZ_DEVICE_MIN"] = "0"        # (files this small are normally left to the pool)
    os.environ["AQC_GZ_KEEP"] = "5"              # (... and the pool's head start of 32 sections is a sixth of them: one group in front of it will do)
    try:
        sections, from_device, text_bytes, device_bytes = gz_run("gzdev")
    finally:
        del os.environ["AQC_GZ_GROUP"]
        del os.environ["AQC_GZ_DEVICE_MIN"]
        del os.environ["AQC_GZ_KEEP"]
    assert sections > 20 and from_device > 0.3 * sections and device_bytes > 0.2 * text_bytes, (sections, from_device, text_bytes, device_bytes)
    os.environ["AQC_GZ_DEVICE_IN"] = "0"
    try:
        sections, from_device, _, _ = gz_run("gzhost")
    finally:
        del os.environ["AQC_GZ_DEVICE_IN"]
    assert sections > 20 and from_device == 0


def test_one_gigabyte_member_is_inflated_exactly(tmp_path):
    """ONE gzip member of more than 1 GB of FASTQ text (zlib level 1, a single deflate stream) read back through the pipe's
    parallel decoder (aqc_gunzip.cpp: sections started at block boundaries found mid-stream, committed only when they chain
    up bit for bit): byte-identical, and the sections really were used.  No kernel is involved — the test runs with the
    GPU-box suite because of its size (zlib needs ~10 s there to make the member)."""
    import zlib
    d = synth.make_pairs(200_000, 150, seed=4242, workers=4)
    plain = str(tmp_path / "piece.fq")
    synth.write_fastq_fixed(plain, d["seq1"], d["qual1"], 1)
    piece = np.fromfile(plain, dtype=np.uint8)
    os.unlink(plain)
    reps = (1 << 30) // len(piece) + 1
    gz = str(tmp_path / "big.fq.gz")
    want = hashlib.sha256()
    total = 0
    comp = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(gz, "wb") as f:
        for r in range(reps):
            # every repetition gets its own tile number in the read names, so that no two stretches of the file are equal
            blk = piece.copy()
            tile = ("%04d" % (1101 + r)).encode()
            pos = np.flatnonzero(blk == ord("@"))
            pos = pos[(pos == 0) | (blk[np.maximum(pos, 1) - 1] == 10)]          # '@' at a line start: the name lines (and a few quality lines: harmless)
            for k, c in enumerate(tile):
                blk[pos + 13 + k] = np.where(blk[pos + 12] == ord(":"), c, blk[pos + 13 + k])
            b = blk.tobytes()
            want.update(b)
            total += len(b)
            f.write(comp.compress(b))
        f.write(comp.flush())
    assert total > (1 << 30)
    src = capi.NativeSource(gz, True)
    got = hashlib.sha256()
    n = 0
    buf = np.empty(64 << 20, dtype=np.uint8)
    while True:
        k = src.readinto(memoryview(buf))
        if not k:
            break
        got.update(memoryview(buf)[:k])
        n += k
    accepted, discarded, sequential, out = src.gz_stats()
    src.close()
    assert n == total and got.digest() == want.digest()
    assert out == total and accepted > 50 and sequential < total // 20, (accepted, discarded, sequential)
    # the same member with the DEVICE decoding its sections (aqc_gunzip_dev: a lane per deflate block, chain-validated): exact
    # again, and the device supplied nearly all of it
    lib = capi.load_library()
    comp_bytes = np.fromfile(gz, dtype=np.uint8)
    text = np.empty(total + 4096, dtype=np.uint8)
    n_out = capi.C.c_uint64(0)
    stats = np.zeros(8, dtype=np.uint64)
    rc = lib.aqc_gunzip_dev(0, comp_bytes.ctypes.data, comp_bytes.size, text.ctypes.data, text.size, capi.C.byref(n_out), stats.ctypes.data, 8, 1 << 20, 64 << 20)
    assert rc == 0, (lib.aqc_last_error() or b"").decode()
    assert n_out.value == total and hashlib.sha256(memoryview(text)[:total]).digest() == want.digest()
    assert stats[0] > 50 and stats[0] > 3 * stats[1] and stats[2] < total // 10, stats
