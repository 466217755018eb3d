"""CPU: the C-ABI library loads and exports every symbol include/afterqc_hip.h declares (no compute)."""
import ctypes
import os
import re

import numpy as np

from afterqc_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "afterqc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aqc_[a-z_0-9]+|edit_distance|seek_overlap)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(capi.EXPORTED_SYMBOLS) == syms
    assert lib.aqc_abi_version() == 3          # (2: aqc_batch.qlen1 / qlen2, AQC_ERR_INDEX, aqc_fetch_quality_views, aqc_error_record; 3: aqc_frame_mixed)


def test_loads_the_way_the_reference_loads_libed():
    """util.py:16-24 does `ed_ctypes = cdll.LoadLibrary(<path>)` and then calls ed_ctypes.edit_distance(a, len(a), b, len(b))
    (util.py:70) / ed_ctypes.seek_overlap(r1, len1, reverse_r2, len2, 3, 30, 50) (util.py:223) with default (int) restype:
    the two symbols must resolve on this library with exactly those names (no compute here: no GPU)."""
    from ctypes import cdll
    ed_ctypes = cdll.LoadLibrary(capi.LIB_PATH)
    assert ed_ctypes.edit_distance is not None and ed_ctypes.seek_overlap is not None
    hdr = open(os.path.join(ROOT, "include", "afterqc_hip.h")).read()
    assert "unsigned int edit_distance(const char* a, const unsigned int asize, const char* b, const unsigned int bsize);" in hdr
    assert re.search(r"int seek_overlap\(const char\* r1, const int len1, const char\* r2_revcomp, const int len2, const int limit_distance,\s*"
                     r"const int complete_compare_require, const int overlap_require\);", hdr)


def test_struct_layouts_match_header():
    assert capi.RESULT_DTYPE.itemsize == 32
    assert ctypes.sizeof(capi.Config) == 18 * 4 + 32 + 2 * 4
    assert ctypes.sizeof(capi.BatchStruct) == 8 * 2 + 8 * 7 * 2 + 8 * 5 + 8 * 2
    assert capi.N_COUNTERS == 4 + 12 + 10 + 16
    assert ctypes.sizeof(capi.TextChunk) == 8 * 4 + 4 * 2 + 8 * 2          # struct aqc_text_chunk
    assert ctypes.sizeof(capi.FrameInfo) == 8 * 5 + 4 * 4                  # struct aqc_frame_info


def test_no_gpu_means_loud_failure():
    """Without a GPU the product path must raise, never fall back (there is no CPU path)."""
    lib = capi.load_library()
    if lib.aqc_device_count() > 0:
        return
    try:
        capi.Engine(0, 1)
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("Engine() must fail without a GPU")


def test_batch_packing_roundtrip():
    seqs = [b"ACGT" * 5, b"A", b"ACGTN" * 30]
    quals = [b"I" * 20, b"#", b"E" * 150]
    b = capi.Batch.from_strings(seqs, quals)
    assert all(int(o) % 16 == 0 for o in b.off1)
    for i in range(3):
        assert b.read1(i) == (seqs[i], quals[i])
    m = np.frombuffer(b"ACGTACGTAC" * 3, dtype=np.uint8).reshape(3, 10).copy()
    b2 = capi.Batch.from_matrices(m, m, np.array([10, 10, 10]))
    assert b2.read1(2)[0] == b"ACGTACGTAC"
    b3 = capi.Batch.from_matrices(m, m, np.array([10, 3, 7]))
    assert b3.read1(1)[0] == b"ACG" and b3.read1(2)[0] == b"ACGTACG"
