#!/usr/bin/env python3
"""device gunzip self-test / timing: python tools/gpu_gunzip_dev.py [MB of text] [gzip level]"""
import gzip, subprocess, sys, time, zlib
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afterqc_amd import capi, synth

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
strategy = {"default": zlib.Z_DEFAULT_STRATEGY, "huffman": zlib.Z_HUFFMAN_ONLY, "rle": zlib.Z_RLE}[sys.argv[3] if len(sys.argv) > 3 else "default"]
n_pairs = mb * (1 << 20) // 347
d = synth.make_pairs(n_pairs, 150, seed=11, workers=16)
buf, n = synth.render_fastq_fixed(d["seq1"], d["qual1"], 1)
text = bytes(memoryview(buf)[:n])
t0 = time.time()
c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
gz = c.compress(text) + c.flush()
print("text %.1f MB -> gz %.1f MB (level %d, %.1f s)" % (len(text) / 1e6, len(gz) / 1e6, level, time.time() - t0), flush=True)
lib = capi.load_library()
src = np.frombuffer(gz, dtype=np.uint8)
out = np.zeros(len(text) + 4096, dtype=np.uint8)
n_out = capi.C.c_uint64(0)
stats = np.zeros(8, dtype=np.uint64)
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 16
sec = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 20
group = int(sys.argv[6]) if len(sys.argv) > 6 else 64 << 20
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.aqc_gunzip_dev(0, src.ctypes.data, len(gz), out.ctypes.data, out.size, capi.C.byref(n_out), stats.ctypes.data, threads, sec, group)
    dt = time.perf_counter() - t0
    ok = rc == 0 and n_out.value == len(text) and out[:len(text)].tobytes() == text
    print("rc %d  %s  %d of %d bytes  %.3f s = %.2f GB/s of text; sections device %d host %d, bridged %d B; kernels ms: scan %.1f decode %.1f chain+gather %.1f h2d %.1f d2h %.1f; err: %s" % (
        rc, "EXACT" if ok else "MISMATCH", n_out.value, len(text), dt, len(text) / dt / 1e9, stats[0], stats[1], stats[2], stats[3] / 1e3, stats[4] / 1e3, stats[5] / 1e3, stats[6] / 1e3, stats[7] / 1e3,
        (lib.aqc_last_error() or b"").decode() if rc else ""), flush=True)
    if not ok and rc == 0:
        a = np.frombuffer(text, dtype=np.uint8); b = out[:len(text)]
        m = min(len(a), int(n_out.value))
        diff = np.flatnonzero(a[:m] != b[:m])
        print("first difference at", int(diff[0]) if len(diff) else None, "of", m)
        if len(diff):
            i = int(diff[0]); print(bytes(a[max(0, i - 40):i + 40])); print(bytes(b[max(0, i - 40):i + 40]))
        break
