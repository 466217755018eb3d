// aqc_kernels.hpp — gfx950 device code of the AfterQC hot path (wave64, LDS-staged).
//
// Generation 1 ("wave per record"): one 64-lane wavefront owns one read pair, stages the four byte
// strings of the pair in LDS and runs the whole per-read pipeline of preprocesser.py:436-631 on
// them with ballot / popcount reductions.  Every stage is an exact restatement of the reference
// arithmetic; comments cite the reference lines.  Integer / byte work only — no MFMA, no floats
// except the f64 circle test of isInBubble.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/afterqc_hip.h"

namespace aqc {

// ---- loads / stores that SAY the pointer is global memory ---------------------------------------------------------------------------
// A pointer that reached a lane through LDS, v_readlane or a select between arrays is "generic" to the compiler: it emits flat_load /
// flat_store, which count on lgkmcnt as well as vmcnt — every wait for an LDS read behind one then waits for the memory round trip
// too (the general copy kernel's four window loads and the k-mer kernel's prefetch were serialised that way, rounds 1 - 5).  These
// helpers cast to address space 1 first: global_load / global_store at any alignment, vmcnt only.
#ifdef AQC_FLAT_AS
#define AQC_GLOBAL_AS      // (A/B builds only: the pointers stay generic, as in rounds 1 - 5)
#else
#define AQC_GLOBAL_AS __attribute__((address_space(1)))
#endif
typedef uint32_t aqc_u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t aqc_u32x2_u __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t aqc_u32_u __attribute__((aligned(1)));
typedef uint16_t aqc_u16_u __attribute__((aligned(1)));
__device__ __forceinline__ uint4 gload_u128(const uint8_t* p) {
    const aqc_u32x4_u t = *(const AQC_GLOBAL_AS aqc_u32x4_u*)p;
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint32_t gload_u32(const uint8_t* p) { return *(const AQC_GLOBAL_AS aqc_u32_u*)p; }
__device__ __forceinline__ void gstore_u128(uint8_t* p, uint4 v) {
    aqc_u32x4_u t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(AQC_GLOBAL_AS aqc_u32x4_u*)p = t;
}
__device__ __forceinline__ void gstore_u64(uint8_t* p, uint32_t a, uint32_t b) {
    aqc_u32x2_u t;
    t.x = a; t.y = b;
    *(AQC_GLOBAL_AS aqc_u32x2_u*)p = t;
}
__device__ __forceinline__ void gstore_u32(uint8_t* p, uint32_t a) { *(AQC_GLOBAL_AS aqc_u32_u*)p = a; }
__device__ __forceinline__ void gstore_u16(uint8_t* p, uint16_t a) { *(AQC_GLOBAL_AS aqc_u16_u*)p = a; }
__device__ __forceinline__ void gstore_u8(uint8_t* p, uint8_t a) { *(AQC_GLOBAL_AS uint8_t*)p = a; }

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WPB = BLOCK / WAVE;      // waves (= records in flight) per workgroup
constexpr int LSTR = 1024;             // LDS bytes per staged string (AQC_MAX_READ_LEN = 1000)

struct DevBatch {
    const uint8_t *seq1, *qual1, *seq2, *qual2;
    const uint32_t *off1, *qoff1, *off2, *qoff2;   // byte offsets into the arenas (a chunk / batch arena is < 4 GiB); qoff NULL = off
    const uint32_t *len1, *len2;
    const int32_t *aux_lane, *aux_tile, *aux_x, *aux_y;
    const uint8_t* aux_ok;
    uint64_t n;
    uint64_t first_index;
    // Records whose QUALITY line is not as long as their SEQUENCE line (the reference never compares the two: fastq.py:37-49
    // hands the lines over as they are, preprocesser.py:19-28 slices each string by its own length, :565-568 index each quality
    // string from its own end).  Such a mate carries LEN_IRR in its len word; qlen holds the quality line's length (low 31
    // bits; NULL: the batch has no such record) and the general kernel leaves the FINAL quality view of both mates of such a
    // record in qview (start | length << 16, relative to the quality line) for the writer and the post-filter statRead.
    const uint32_t *qlen1, *qlen2;
    uint32_t *qview1, *qview2;
};
constexpr uint32_t LEN_IRR = 0x80000000u, LEN_MASK = 0x7fffffffu;
// the quality-length words of a framed chunk (frame_records_kernel) carry two flags above the length: the quality line ends right
// at its '\n' (nothing stripped), and ALL FOUR lines of the record do — the record stands in the chunk exactly as a writer would
// write it
constexpr uint32_t QLEN_TAILNL = 0x80000000u, QLEN_CONTIG = 0x40000000u, QLEN_MASK = 0x3fffffffu;

struct DevCircles {
    const double *cx, *cy, *cr;
    const int32_t *lane, *tile;
    int32_t n;
};

struct DevStats {
    unsigned long long* counters;   // [AQC_N_COUNTERS]
    unsigned long long* ovl_hist;   // [AQC_QC_COLS]
    unsigned long long* dist_hist;  // [AQC_QC_COLS]
    int* status;                    // first error code raised on the device (0 = ok)
    // errors that END THE RUN AT A RECORD upstream (an exception inside the loop of preprocesser.py:411-631: KeyError of
    // util.complement / the error matrix, IndexError of a quality string too short for the walk, int() of a name field):
    // min over (record << 8 | -code), so that the host learns the EARLIEST such record — everything before it was written
    // upstream when the exception flew (~0 = none)
    unsigned long long* err_key;
};
__device__ __forceinline__ void raise_at_record(const DevStats& st, uint64_t rec, int code) {
    atomicCAS(st.status, 0, code);
    atomicMin(st.err_key, ((unsigned long long)rec << 8) | (unsigned long long)(unsigned int)(-code));
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, WAVE);
    return v;
}

// util.py:27 COMP; returns 0 for bytes outside the table (KeyError upstream)
__device__ __forceinline__ uint8_t comp_strict(uint8_t c) {
    switch (c) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'a': return 't';
        case 't': return 'a';
        case 'c': return 'g';
        case 'g': return 'c';
        case 'N': return 'N';
        default: return 0;
    }
}

// util.py:47-50 reverseComplement's per-base rule: unknown -> 'N'
__device__ __forceinline__ uint8_t comp_or_n(uint8_t c) {
    uint8_t r = comp_strict(c);
    return r ? r : (uint8_t)'N';
}

// ALL_BASES index A,T,C,G -> 0..3 (qualitycontrol.py:24), -1 otherwise
__device__ __forceinline__ int base_idx(uint8_t c) {
    return c == 'A' ? 0 : c == 'T' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : -1;
}

// the 9 symbols hasPolyX counts (preprocesser.py:35)
__device__ __forceinline__ bool poly_symbol(uint8_t c) { return comp_strict(c) != 0; }

// ------------------------------------------------------------------------------------------------
// util.overlap_hm (util.py:158-212) for one pair, executed by one wavefront.
//   r1  : LDS pointer to the current read 1 (len1 bytes)
//   c2  : LDS pointer to complement-or-N of the current read 2 in ORIGINAL orientation (len2 bytes)
//         -> reverse_r2[i] == c2[len2 - 1 - i]
// Candidates are enumerated in the reference's order (forward offsets 0..len1-31, then reverse
// 0,-1,..,-(len2-31)); 64 candidates per step, one per lane:
//   phase A  each lane counts mismatches over the first min(16, L) columns of its diagonal; a
//            diagonal with >= 3 of them can never be accepted (both accept branches of
//            util.py:183 need fewer than 3 mismatches among the first 50 columns);
//   phase B  survivors are verified in order by the whole wave: tot = mismatches over all L
//            columns, c50 = those at i < 50; accept iff tot < 3 or (c50 < 3 and L >= 52), which is
//            the loop of util.py:177-183 in closed form: the loop breaks at
//            the 3rd mismatch only if it falls at i < 50, otherwise it runs to i = L-1 and the
//            test `i > 50` needs L >= 52.  diff reported = tot.
// All lanes return the same (offset, overlap_len, diff).
// ------------------------------------------------------------------------------------------------
__device__ inline void overlap_hm_wave(const uint8_t* r1, int len1, const uint8_t* c2, int len2, int& o_offset,
                                       int& o_len, int& o_diff) {
    const int lane = lane_id();
    const int nf = len1 > 30 ? len1 - 30 : 0;   // forward offsets: offset < len1 - 30
    const int nr = len2 > 30 ? len2 - 30 : 0;   // reverse offsets: offset > -(len2 - 30)
    const int ncand = nf + nr;
    const uint8_t* rr2_last = c2 + len2 - 1;    // reverse_r2[i] = rr2_last[-i]
    for (int base = 0; base < ncand; base += WAVE) {
        const int c = base + lane;
        const bool valid = c < ncand;
        int p1 = 0, p2 = 0, L = 0;
        if (valid) {
            if (c < nf) { p1 = c; p2 = 0; L = min(len1 - c, len2); }
            else { p1 = 0; p2 = c - nf; L = min(len1, len2 - p2); }
        }
        int cnt = 0;
        const int P = min(16, L);
        for (int i = 0; i < P; i++) cnt += (r1[p1 + i] != rr2_last[-(p2 + i)]) ? 1 : 0;
        unsigned long long surv = __ballot(valid && cnt < 3);
        while (surv) {
            const int l = __ffsll((long long)surv) - 1;
            surv &= surv - 1;
            const int q1 = __shfl(p1, l, WAVE), q2 = __shfl(p2, l, WAVE), QL = __shfl(L, l, WAVE);
            int tot = 0, c50 = 0;
            for (int i0 = 0; i0 < QL; i0 += WAVE) {
                const int i = i0 + lane;
                const bool mm = i < QL && r1[q1 + i] != rr2_last[-(q2 + i)];
                const unsigned long long b = __ballot(mm);
                tot += __popcll(b);
                if (i0 == 0) c50 = __popcll(b & ((1ull << 50) - 1));
            }
            if (tot < 3 || (c50 < 3 && QL >= 52)) {
                const int cand = base + l;
                o_offset = cand < nf ? cand : -(cand - nf);
                o_len = QL;
                o_diff = tot;
                return;
            }
        }
    }
    o_offset = 0; o_len = 0; o_diff = 0;
}

// hasPolyX (preprocesser.py:30-51) by one wave: the byte that fires first, or 0 for None.
// Position x fires iff seq[x] occurs >= maxPoly - mismatch times in seq[max(0,x-maxPoly+1) .. x];
// scanning stops (with None) at the first byte outside the 9 symbols.
__device__ inline int has_polyx_wave(const uint8_t* s, int len, int maxPoly, int mismatch) {
    if (len < maxPoly) return 0;
    const int lane = lane_id();
    const int need = maxPoly - mismatch;
    // first invalid position (scan range is [0, vend))
    int vend = len;
    for (int x0 = 0; x0 < len; x0 += WAVE) {
        const int x = x0 + lane;
        const unsigned long long bad = __ballot(x < len && !poly_symbol(s[x]));
        if (bad) { vend = x0 + __ffsll((long long)bad) - 1; break; }
    }
    for (int x0 = 0; x0 < vend; x0 += WAVE) {
        const int x = x0 + lane;
        bool fire = false;
        if (x < vend) {
            const uint8_t f = s[x];
            const int lo = x - maxPoly + 1 > 0 ? x - maxPoly + 1 : 0;
            int cnt = 0;
            for (int j = lo; j <= x; j++) cnt += (s[j] == f) ? 1 : 0;
            fire = cnt >= need;
        }
        const unsigned long long b = __ballot(fire);
        if (b) return s[x0 + __ffsll((long long)b) - 1];
    }
    return 0;
}

// lowQualityNum (preprocesser.py:61-68): count of ord(q) < qual + 33
__device__ inline int low_quality_wave(const uint8_t* q, int len, int qual) {
    const int lane = lane_id();
    const int thr = qual + 33;
    int n = 0;
    for (int i0 = 0; i0 < len; i0 += WAVE) {
        const int i = i0 + lane;
        n += __popcll(__ballot(i < len && (int)q[i] < thr));
    }
    return n;
}

// nNumber (preprocesser.py:70-76)
__device__ inline int n_number_wave(const uint8_t* s, int len) {
    const int lane = lane_id();
    int n = 0;
    for (int i0 = 0; i0 < len; i0 += WAVE) {
        const int i = i0 + lane;
        n += __popcll(__ballot(i < len && s[i] == 'N'));
    }
    return n;
}

// Levenshtein distance of two short strings by ONE LANE, Myers/Hyyro bit-vector form (the
// algorithm of editdistance/_editdistance.cpp:29-60 for a single 64-bit block): A(i) is the pattern
// (la <= 64 bits), B(j) the text, both given as accessors so that views (reverse complements, LDS
// or global pointers) need no copy.  Equals the DP of util.py:72-83.
template <typename FA, typename FB>
__device__ __forceinline__ int edit_distance_lane(FA A, int la, FB B, int lb) {
    if (la == 0) return lb;
    if (lb == 0) return la;
    unsigned long long Pv = la >= 64 ? ~0ull : ((1ull << la) - 1), Mv = 0;
    const unsigned long long top = 1ull << (la - 1);
    int score = la;
    for (int j = 0; j < lb; j++) {
        const uint8_t ch = B(j);
        unsigned long long Eq = 0;
        for (int i = 0; i < la; i++) Eq |= (unsigned long long)(A(i) == ch) << i;
        const unsigned long long Xv = Eq | Mv;
        const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        unsigned long long Ph = Mv | ~(Xh | Pv);
        unsigned long long Mh = Pv & Xh;
        if (Ph & top) score++;
        else if (Mh & top) score--;
        Ph = (Ph << 1) | 1ull;
        Mh <<= 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
    }
    return score;
}

// detectBarcode (barcodeprocesser.py:19-32) by one wave
__device__ inline int detect_barcode_wave(const uint8_t* s, int len, int bl, const uint8_t* verify, int vl) {
    if (len <= vl + bl + 1) return 0;
    const int lane = lane_id();
    const bool in = lane < vl;
    const uint8_t v = in ? verify[lane] : 0;
    const int dc = __popcll(__ballot(in && s[bl + lane] != v));
    if (dc <= 1) return bl;
    const int dl = __popcll(__ballot(in && s[bl - 1 + lane] != v));
    if (dl == 0) return bl - 1;
    const int dr = __popcll(__ballot(in && s[bl + 1 + lane] != v));
    if (dr == 0) return bl + 1;
    return 0;
}

// cleanBarcodeTail (barcodeprocesser.py:47-75): lane i evaluates iteration i of the loop
// (compLen = min(n1,n2) - i) with two Levenshtein distances; the first i that satisfies both
// thresholds wins.  rs1/rs2 = readStart strings (barcode + verify), r1/r2 = the moved reads.
__device__ inline int clean_barcode_tail_wave(const uint8_t* r1, int r1len, const uint8_t* r2, int r2len,
                                              const uint8_t* rs1, int n1, const uint8_t* rs2, int n2) {
    const int lane = lane_id();
    const int bsl = min(n1, n2);
    bool ok = false;
    int compLen = 0;
    if (lane < bsl) {
        compLen = bsl - lane;
        if (!(compLen >= r1len || compLen >= r2len)) {
            // reverse2[i:] = revcomp(readStart2)[i:], n2 - i chars; reverse1 likewise
            const int m2 = n2 - lane, m1 = n1 - lane;
            const uint8_t* t1p = r1 + r1len - compLen;
            const uint8_t* t2p = r2 + r2len - compLen;
            const int d1 = edit_distance_lane([&](int i) { return t1p[i]; }, compLen,
                                              [&](int k) { return comp_or_n(rs2[n2 - 1 - (lane + k)]); }, m2);
            const int d2 = edit_distance_lane([&](int i) { return t2p[i]; }, compLen,
                                              [&](int k) { return comp_or_n(rs1[n1 - 1 - (lane + k)]); }, m1);
            ok = (d1 * 5 <= compLen) && (d2 * 5 <= compLen);   // distance <= compLen/5
        }
    }
    const unsigned long long b = __ballot(ok);
    if (!b) return 0;
    return bsl - (__ffsll((long long)b) - 1);
}

// isInBubble's geometric half (preprocesser.py:193-204), IEEE double, no contraction
__device__ inline bool in_bubble_wave(int lane_no, int tile, int x, int y, const DevCircles& c) {
    const int lane = lane_id();
    bool hit = false;
    for (int i = lane; i < c.n; i += WAVE) {
        if (c.tile[i] == tile && c.lane[i] == lane_no) {
            const double dx = __dsub_rn(c.cx[i], (double)x), dy = __dsub_rn(c.cy[i], (double)y);
            const double lhs = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
            if (lhs < __dmul_rn(c.cr[i], c.cr[i])) hit = true;
        }
    }
    return __ballot(hit) != 0;
}

// stage `len` bytes from global memory into LDS (coalesced byte loads: lane i -> byte i)
__device__ __forceinline__ void stage(uint8_t* dst, const uint8_t* src, int len) {
    for (int i = lane_id(); i < len; i += WAVE) dst[i] = src[i];
}

struct BlockAcc {
    unsigned long long counters[AQC_N_COUNTERS];
    unsigned int ovl_hist[AQC_QC_COLS];
    unsigned int dist_hist[AQC_QC_COLS];
};

// ------------------------------------------------------------------------------------------------
// One record through preprocesser.py:436-631, executed by ONE wavefront on byte strings staged in
// LDS ("generation 1", fully general: any alphabet, any length <= AQC_MAX_READ_LEN, barcodes,
// bubbles).  Used by the generic kernel for every record and by the fast kernel (aqc_fast.hpp) for
// the records it defers.
// ------------------------------------------------------------------------------------------------
struct WaveLds {
    uint8_t *s1, *q1, *s2, *q2, *c2;   // 5 x LSTR bytes
    uint8_t *rs1, *rs2;                // 2 x 64 bytes (barcode readStart strings)
};

template <class Cfg>      // aqc_config, or aqc_config in the kernarg segment (address space 4: its fields are scalar loads where they are used)
__device__ __forceinline__ void process_record_wave(const DevBatch& b, uint64_t rec, const Cfg& cfg, const DevCircles& circ,
                                           const WaveLds& w, aqc_result* __restrict__ results, BlockAcc& acc,
                                           const DevStats& st, bool accum) {
    const int lane = lane_id();
    uint8_t* s1 = w.s1;
    uint8_t* q1 = w.q1;
    uint8_t* s2 = w.s2;
    uint8_t* q2 = w.q2;
    uint8_t* c2 = w.c2;   // complement-or-N of s2, same orientation
    const bool paired = cfg.paired != 0;
    {
        const uint32_t l1w = b.len1[rec], l2w = paired ? b.len2[rec] : 0u;
        const int L1 = (int)(l1w & LEN_MASK);
        const int L2 = (int)(l2w & LEN_MASK);
        // a quality line that is not as long as its sequence line (either mate): every string keeps its own view
        const bool irr = b.qlen1 != nullptr && ((l1w | l2w) & LEN_IRR) != 0u;
        const int QL1 = irr ? (int)(b.qlen1[rec] & QLEN_MASK) : L1;
        const int QL2 = (irr && paired) ? (int)(b.qlen2[rec] & QLEN_MASK) : L2;
        if (L1 > AQC_MAX_READ_LEN || L2 > AQC_MAX_READ_LEN || QL1 > AQC_MAX_READ_LEN || QL2 > AQC_MAX_READ_LEN) {
            if (lane == 0) atomicCAS(st.status, 0, AQC_ERR_READ_TOO_LONG);
            return;
        }
        stage(s1, b.seq1 + b.off1[rec], L1);
        stage(q1, b.qual1 + (b.qoff1 ? b.qoff1[rec] : b.off1[rec]), QL1);
        if (paired) {
            stage(s2, b.seq2 + b.off2[rec], L2);
            stage(q2, b.qual2 + (b.qoff2 ? b.qoff2[rec] : b.off2[rec]), QL2);
            for (int i = lane; i < L2; i += WAVE) c2[i] = comp_or_n(b.seq2[b.off2[rec] + i]);
        }
        // (wave-private LDS region: no barrier needed, the compiler orders LDS ops of one wave)
        __builtin_amdgcn_wave_barrier();

        int a1 = 0, len1 = L1, a2 = 0, len2 = L2;     // current views: s1[a1 .. a1+len1), s2[a2 .. a2+len2)
        // ... and of the quality strings: q1[qa1 .. qa1+ql1), q2[qa2 .. qa2+ql2).  Every slice upstream is a python slice of
        // EACH string (preprocesser.py:19-28,521-524, barcodeprocesser.py:42-43,66-69): the same cut applied to a string of
        // another length.  For a regular record they stay equal to the sequence views.
        int qa1 = 0, ql1 = QL1, qa2 = 0, ql2 = QL2;
        auto cut_front = [](int& a, int& l, int k) { const int m = min(k, l); a += m; l -= m; };      // s[k:]
        auto cut_tail = [](int& l, int k) { if (k > 0) l = max(l - k, 0); };                          // s[:-k]  (k > 0)
        int flag = -1;
        int offset = 0, ovl = 0, dist = 0, n_edits = 0;
        uint8_t bcode = 0;
        // counters this record contributes (wave-uniform values, committed by lane 0)
        int c_adapter_base = 0, c_adapter_read = 0, c_overlapped = 0, c_corrected = 0, c_masked = 0, c_skipped = 0;
        int c_read_corrected = 0, ovl0 = -1, dist_final = -1;
        int em[3] = {-1, -1, -1};
        // (written through constant indices only, so that both arrays stay in registers: indexed by n_edits / handled they lived in
        //  20 bytes of scratch — round 5 review)
        unsigned long long ed0 = 0, ed1 = 0, ed2 = 0;      // o | kind << 16 | base << 24 | qual << 32
        auto put_edit = [&](int k, const aqc_edit& e) {
            const unsigned long long x = (unsigned long long)e.o | ((unsigned long long)e.kind << 16) | ((unsigned long long)e.base << 24) | ((unsigned long long)e.qual << 32);
            ed0 = k == 0 ? x : ed0; ed1 = k == 1 ? x : ed1; ed2 = k >= 2 ? x : ed2;
        };
        auto get_edit = [](unsigned long long x) { return aqc_edit{(uint16_t)(x & 0xffffu), (uint8_t)(x >> 16), (uint8_t)(x >> 24), (uint8_t)(x >> 32)}; };
        auto put_em = [&](int k, int val) { em[0] = k == 0 ? val : em[0]; em[1] = k == 1 ? val : em[1]; em[2] = k >= 2 ? val : em[2]; };

        // ---- barcode (preprocesser.py:436-452)
        if (cfg.barcode) {
            const int bl = cfg.barcode_length, vl = cfg.barcode_verify_len;
            const int b1 = detect_barcode_wave(s1, len1, bl, cfg.barcode_verify, vl);
            if (b1 == 0) flag = AQC_BADBCD1;
            else {
                bcode = (uint8_t)(b1 - bl + 2);
                if (!paired) {
                    const int rm = vl + bl;   // single-end moves the design length (preprocesser.py:444)
                    a1 += min(rm, len1); len1 = max(len1 - rm, 0);
                    cut_front(qa1, ql1, rm);
                } else {
                    const int b2 = detect_barcode_wave(s2, len2, bl, cfg.barcode_verify, vl);
                    if (b2 == 0) flag = AQC_BADBCD2;
                    else {
                        bcode |= (uint8_t)((b2 - bl + 2) << 4);
                        // readStart = seq[0:barcodeLen] + verify (barcodeprocesser.py:78-79)
                        uint8_t* rs1 = w.rs1;
                        uint8_t* rs2 = w.rs2;
                        if (lane < b1) rs1[lane] = s1[lane];
                        if (lane < vl) rs1[b1 + lane] = cfg.barcode_verify[lane];
                        if (lane < b2) rs2[lane] = s2[lane];
                        if (lane < vl) rs2[b2 + lane] = cfg.barcode_verify[lane];
                        __builtin_amdgcn_wave_barrier();
                        a1 += vl + b1; len1 -= vl + b1;
                        a2 += vl + b2; len2 -= vl + b2;
                        cut_front(qa1, ql1, vl + b1); cut_front(qa2, ql2, vl + b2);
                        const int cut = clean_barcode_tail_wave(s1 + a1, len1, s2 + a2, len2, rs1, b1 + vl, rs2, b2 + vl);
                        len1 -= cut; len2 -= cut;
                        cut_tail(ql1, cut); cut_tail(ql2, cut);
                    }
                }
            }
        }
        // ---- trim (preprocesser.py:455-466, python slice semantics of trim() :19-28)
        if (flag < 0 && (cfg.trim_front > 0 || cfg.trim_tail > 0)) {
            int end = cfg.trim_tail > 0 ? max(len1 - cfg.trim_tail, 0) : len1;
            int stt = min(cfg.trim_front, len1);
            int nl = max(end - stt, 0);
            a1 += stt; len1 = nl;
            {
                const int qend = cfg.trim_tail > 0 ? max(ql1 - cfg.trim_tail, 0) : ql1, qst = min(cfg.trim_front, ql1);
                qa1 += qst; ql1 = max(qend - qst, 0);
            }
            if (len1 < 5) flag = AQC_BADTRIM1;
            else if (paired) {
                end = cfg.trim_tail2 > 0 ? max(len2 - cfg.trim_tail2, 0) : len2;
                stt = min(cfg.trim_front2, len2);
                nl = max(end - stt, 0);
                a2 += stt; len2 = nl;
                const int qend = cfg.trim_tail2 > 0 ? max(ql2 - cfg.trim_tail2, 0) : ql2, qst = min(cfg.trim_front2, ql2);
                qa2 += qst; ql2 = max(qend - qst, 0);
                if (len2 < 5) flag = AQC_BADTRIM2;
            }
        }
        // ---- bubble (preprocesser.py:469-473)
        if (flag < 0 && cfg.debubble && b.aux_ok && b.aux_ok[rec]) {
            if (b.aux_ok[rec] == 2) { if (lane == 0) raise_at_record(st, rec, AQC_ERR_ARG); }     // int() raises upstream (preprocesser.py:187-192)
            else if (in_bubble_wave(b.aux_lane[rec], b.aux_tile[rec], b.aux_x[rec], b.aux_y[rec], circ)) flag = AQC_BADBBL;
        }
        // ---- length (preprocesser.py:476-479)
        if (flag < 0 && len1 < cfg.seq_len_req) flag = AQC_BADLEN;
        // ---- polyX (preprocesser.py:482-490)
        if (flag < 0 && cfg.poly_size_limit > 0) {
            int p = has_polyx_wave(s1 + a1, len1, cfg.poly_size_limit, cfg.allow_mismatch_in_poly);
            if (p == 0 && paired) p = has_polyx_wave(s2 + a2, len2, cfg.poly_size_limit, cfg.allow_mismatch_in_poly);
            if (p != 0) flag = AQC_BADPOL;
        }
        // ---- low quality: only read 1 is tested (preprocesser.py:498, upstream quirk)
        if (flag < 0 && cfg.unqualified_base_limit > 0) {
            if (low_quality_wave(q1 + qa1, ql1, cfg.qualified_quality_phred) > cfg.unqualified_base_limit) flag = AQC_BADLQC;      // (the QUALITY line is what is counted, :61-68)
        }
        // ---- N (preprocesser.py:504-512)
        if (flag < 0 && cfg.n_base_limit > 0) {
            const int n1 = n_number_wave(s1 + a1, len1);
            const int n2 = paired ? n_number_wave(s2 + a2, len2) : 0;
            if (n1 > cfg.n_base_limit || n2 > cfg.n_base_limit) flag = AQC_BADNCT;
        }
        // ---- overlap + correction (preprocesser.py:515-617)
        if (flag < 0 && paired && !cfg.no_overlap) {
            overlap_hm_wave(s1 + a1, len1, c2 + a2, len2, offset, ovl, dist);
            ovl0 = ovl;
            if (offset < 0 && ovl > 30) {
                len1 = ovl; len2 = ovl;                      // all four strings := [0:overlap_len]
                ql1 = min(ql1, ovl); ql2 = min(ql2, ovl);
                c_adapter_base = 2 * (-offset); c_adapter_read = 1;
                if (len1 < cfg.seq_len_req) { flag = AQC_BADLEN; offset = 0; ovl = 0; dist = 0; }   // record carries no overlap
                else overlap_hm_wave(s1 + a1, len1, c2 + a2, len2, offset, ovl, dist);
            }
            if (flag < 0) {
                dist_final = dist;
                if (dist > 3) flag = AQC_BADDIFF;
                else if (ovl > 30) {
                    c_overlapped = 1;
                    if (dist > 0 && irr) {
                        // The walk of preprocesser.py:563-598 for a record whose quality views differ from its sequence views,
                        // as upstream runs it: one position after the other, every string indexed from ITS OWN end
                        // (r1[3][len(r1[3]) - overlap_len + o] with python's wrap for a negative index, r2[3][-o-1]), the
                        // quality strings edited in place — a wrapped index can meet a position a later step reads again.
                        // An index outside a string is upstream's IndexError: the run ends at this record.  One lane; such
                        // records are rare.
                        int handled = 0, err = 0;
                        if (lane == 0) {
                            const uint8_t* S1 = s1 + a1;
                            const uint8_t* S2 = s2 + a2;
                            uint8_t* Q1 = q1 + qa1;
                            uint8_t* Q2 = q2 + qa2;
                            for (int o = 0; o < ovl && handled < dist; ++o) {
                                const uint8_t bA = S1[len1 - ovl + o];
                                const uint8_t r2o = S2[len2 - 1 - o];
                                const uint8_t bB = comp_strict(r2o);
                                if (bB == 0) { err = AQC_ERR_ALPHABET; break; }            // util.complement (:565)
                                int i1 = ql1 - ovl + o;
                                if (i1 < 0) i1 += ql1;                                      // python: a negative index counts from the end
                                const int i2 = ql2 - 1 - o;
                                if (i1 < 0 || i2 < 0) { err = AQC_ERR_INDEX; break; }      // IndexError (:566-567)
                                const int qa = Q1[i1], qb = Q2[i2];
                                if (bA == bB) continue;
                                bool fixed = false;
                                if (qa - 33 >= 30 && qb - 33 <= 14) {
                                    const uint8_t cA = comp_strict(bA);
                                    if (bA != 'N' && bB != 'N') {
                                        const int i0 = base_idx(cA), ix = base_idx(r2o);
                                        if (cA == 0 || i0 < 0 || ix < 0) { err = AQC_ERR_ALPHABET; break; }
                                        put_em(handled, i0 * 4 + ix);
                                    }
                                    if (!cfg.no_correction) {
                                        if (cA == 0) { err = AQC_ERR_ALPHABET; break; }
                                        put_edit(n_edits, aqc_edit{(uint16_t)o, AQC_EDIT_FIX_R2, cA, (uint8_t)qa});
                                        Q2[i2] = (uint8_t)qa;
                                        n_edits++; c_corrected++; fixed = true;
                                    }
                                } else if (qb - 33 >= 30 && qa - 33 <= 14) {
                                    if (bA != 'N' && bB != 'N') {
                                        const int i0 = base_idx(bB), ix = base_idx(bA);
                                        if (i0 < 0 || ix < 0) { err = AQC_ERR_ALPHABET; break; }
                                        put_em(handled, i0 * 4 + ix);
                                    }
                                    if (!cfg.no_correction) {
                                        put_edit(n_edits, aqc_edit{(uint16_t)o, AQC_EDIT_FIX_R1, bB, (uint8_t)qb});
                                        Q1[i1] = (uint8_t)qb;
                                        n_edits++; c_corrected++; fixed = true;
                                    }
                                }
                                if (!fixed) {
                                    if (cfg.mask_mismatch) {
                                        put_edit(n_edits, aqc_edit{(uint16_t)o, AQC_EDIT_MASK, 0, (uint8_t)'!'});
                                        Q2[i2] = (uint8_t)'!'; Q1[i1] = (uint8_t)'!';
                                        n_edits++; c_masked++;
                                    } else c_skipped++;
                                }
                                handled++;
                            }
                            if (err) raise_at_record(st, rec, err);
                        }
                        // (lane 0 writes the result record and the counters; the other lanes only need the verdict)
                        handled = __shfl(handled, 0, WAVE);
                        if (handled == dist) {
                            if (c_corrected > 0) c_read_corrected = 1;
                        } else {
                            flag = AQC_BADMISMATCH;
                            em[0] = em[1] = em[2] = -1;
                            c_corrected = c_masked = c_skipped = 0;
                        }
                    } else if (dist > 0) {
                        // the tail-anchored walk of preprocesser.py:563-598
                        int handled = 0;
                        bool bad_alpha = false;
                        const uint8_t* w1 = s1 + a1 + len1 - ovl;       // b1 = w1[o]
                        const uint8_t* x1 = q1 + a1 + len1 - ovl;       // q1 = x1[o]
                        const uint8_t* w2 = s2 + a2 + len2 - 1;         // r2[-o-1] = w2[-o]
                        const uint8_t* x2 = q2 + a2 + len2 - 1;
                        for (int o0 = 0; o0 < ovl && handled < dist; o0 += WAVE) {
                            const int o = o0 + lane;
                            const bool in = o < ovl;
                            const uint8_t r2b = in ? w2[-o] : (uint8_t)'A';
                            const uint8_t bb2 = comp_strict(r2b);
                            const unsigned long long inval = __ballot(in && bb2 == 0);
                            unsigned long long mm = __ballot(in && w1[o] != bb2);
                            int last = WAVE - 1;
                            while (mm && handled < dist) {
                                const int l = __ffsll((long long)mm) - 1;
                                mm &= mm - 1;
                                last = l;
                                const int oo = o0 + l;
                                const uint8_t bA = w1[oo];
                                const uint8_t r2o = w2[-oo];
                                const uint8_t bB = comp_strict(r2o);
                                const int qa = x1[oo], qb = x2[-oo];
                                bool fixed = false;
                                if (qa - 33 >= 30 && qb - 33 <= 14) {
                                    if (bA != 'N' && bB != 'N') {
                                        const uint8_t cA = comp_strict(bA);
                                        const int i0 = base_idx(cA), i1 = base_idx(r2o);
                                        if (cA == 0 || i0 < 0 || i1 < 0) bad_alpha = true;
                                        else put_em(handled, i0 * 4 + i1);          // err[comp(b1)][comp(b2)]
                                    }
                                    if (!cfg.no_correction) {
                                        const uint8_t cA = comp_strict(bA);
                                        if (cA == 0) bad_alpha = true;
                                        put_edit(n_edits, aqc_edit{(uint16_t)oo, AQC_EDIT_FIX_R2, cA, (uint8_t)qa});
                                        n_edits++; c_corrected++; fixed = true;
                                    }
                                } else if (qb - 33 >= 30 && qa - 33 <= 14) {
                                    if (bA != 'N' && bB != 'N') {
                                        const int i0 = base_idx(bB), i1 = base_idx(bA);
                                        if (i0 < 0 || i1 < 0) bad_alpha = true;
                                        else put_em(handled, i0 * 4 + i1);          // err[b2][b1]
                                    }
                                    if (!cfg.no_correction) {
                                        put_edit(n_edits, aqc_edit{(uint16_t)oo, AQC_EDIT_FIX_R1, bB, (uint8_t)qb});
                                        n_edits++; c_corrected++; fixed = true;
                                    }
                                }
                                if (!fixed) {
                                    if (cfg.mask_mismatch) {
                                        put_edit(n_edits, aqc_edit{(uint16_t)oo, AQC_EDIT_MASK, 0, (uint8_t)'!'});
                                        n_edits++; c_masked++;
                                    } else c_skipped++;
                                }
                                handled++;
                            }
                            // util.complement raises on every visited r2 byte outside COMP (preprocesser.py:565)
                            const unsigned long long visited = (handled >= dist) ? ((last == 63) ? ~0ull : ((2ull << last) - 1)) : ~0ull;
                            if (inval & visited) bad_alpha = true;
                        }
                        if (bad_alpha && lane == 0) raise_at_record(st, rec, AQC_ERR_ALPHABET);
                        if (handled == dist) {
                            if (c_corrected > 0) c_read_corrected = 1;
                        } else {
                            flag = AQC_BADMISMATCH;
                            em[0] = em[1] = em[2] = -1;
                            c_corrected = c_masked = c_skipped = 0;   // edits stay (written to bad/), counters do not
                        }
                    }
                }
            }
        }
        if (flag < 0) flag = AQC_GOOD;

        // ---- result record + counters (lane 0)
        if (lane == 0) {
            aqc_result r;
            r.flag = (uint8_t)flag; r.n_edits = (uint8_t)n_edits;
            r.start1 = (uint16_t)a1; r.len1 = (uint16_t)len1;
            r.start2 = (uint16_t)a2; r.len2 = (uint16_t)len2;
            r.offset = (int16_t)offset; r.overlap_len = (uint16_t)ovl; r.distance = (uint16_t)dist;
            r.edits[0] = get_edit(ed0); r.edits[1] = get_edit(ed1); r.edits[2] = get_edit(ed2);
            r.barcode = bcode;
            results[rec] = r;
            if (irr) {
                b.qview1[rec] = (uint32_t)qa1 | ((uint32_t)ql1 << 16);
                if (paired) b.qview2[rec] = (uint32_t)qa2 | ((uint32_t)ql2 << 16);
            }
            if (accum) {
                unsigned long long* C = acc.counters;
                atomicAdd(&C[AQC_C_TOTAL_READS], 1ull);
                atomicAdd(&C[AQC_C_TOTAL_BASES], (unsigned long long)(L1 + ((paired && cfg.count_r2_bases) ? L2 : 0)));
                atomicAdd(&C[AQC_C_FLAG0 + flag], 1ull);
                if (flag == AQC_GOOD) {
                    atomicAdd(&C[AQC_C_GOOD_READS], 1ull);
                    atomicAdd(&C[AQC_C_GOOD_BASES], (unsigned long long)(len1 + ((paired && cfg.count_r2_bases) ? len2 : 0)));
                }
                if (ovl0 >= 0) atomicAdd(&acc.ovl_hist[ovl0], 1u);
                if (dist_final >= 0) atomicAdd(&acc.dist_hist[min(dist_final, AQC_QC_COLS - 1)], 1u);
                if (c_adapter_read) {
                    atomicAdd(&C[AQC_C_TRIMMED_ADAPTER_BASE], (unsigned long long)c_adapter_base);
                    atomicAdd(&C[AQC_C_TRIMMED_ADAPTER_READ], 1ull);
                }
                if (c_overlapped) {
                    atomicAdd(&C[AQC_C_OVERLAPPED], 1ull);
                    atomicAdd(&C[AQC_C_OVERLAP_LEN_SUM], (unsigned long long)ovl);
                    atomicAdd(&C[AQC_C_OVERLAP_BASE_SUM], (unsigned long long)(2 * ovl));
                    atomicAdd(&C[AQC_C_OVERLAP_BASE_ERR], (unsigned long long)dist);
                    if (c_read_corrected) atomicAdd(&C[AQC_C_READ_CORRECTED], 1ull);
                    if (c_corrected) atomicAdd(&C[AQC_C_BASE_CORRECTED], (unsigned long long)c_corrected);
                    if (c_masked) atomicAdd(&C[AQC_C_BASE_ZERO_QUAL_MASKED], (unsigned long long)(2 * c_masked));
                    if (c_skipped) atomicAdd(&C[AQC_C_BASE_SKIPPED_CORRECTION], (unsigned long long)(2 * c_skipped));
                    for (int k = 0; k < 3; k++)
                        if (em[k] >= 0) atomicAdd(&C[AQC_C_ERR_MATRIX0 + em[k]], 1ull);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        }
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: grid-stride over records, one wave per record, block-private counters flushed once.
// ------------------------------------------------------------------------------------------------
__device__ inline void flush_block_acc(BlockAcc& acc, const DevStats& st, int tid = -1) {
    if (tid < 0) tid = (int)threadIdx.x;
    for (int i = tid; i < AQC_N_COUNTERS; i += blockDim.x)
        if (acc.counters[i]) atomicAdd(&st.counters[i], acc.counters[i]);
    for (int i = tid; i < AQC_QC_COLS; i += blockDim.x) {
        if (acc.ovl_hist[i]) atomicAdd(&st.ovl_hist[i], (unsigned long long)acc.ovl_hist[i]);
        if (acc.dist_hist[i]) atomicAdd(&st.dist_hist[i], (unsigned long long)acc.dist_hist[i]);
    }
}

// the leading arguments of the two kernels below as they stand in the kernarg segment (each at its natural alignment)
struct FilterArgs {
    DevBatch b;
    aqc_config cfg;
    DevCircles circ;
    aqc_result* results;
    DevStats st;
    uint64_t accum_limit;
};

__global__ __launch_bounds__(BLOCK) void filter_overlap_kernel(DevBatch b, aqc_config cfg, DevCircles circ,
                                                               aqc_result* __restrict__ results, DevStats st,
                                                               uint64_t accum_limit) {
    __shared__ uint8_t lds[WPB][5][LSTR];
    __shared__ uint8_t rsbuf[WPB][2][64];
    __shared__ BlockAcc acc;
    const int wave = threadIdx.x / WAVE;
    for (int i = threadIdx.x; i < (int)(sizeof(BlockAcc) / 4); i += BLOCK) ((unsigned int*)&acc)[i] = 0;
    __syncthreads();
    const WaveLds w{lds[wave][0], lds[wave][1], lds[wave][2], lds[wave][3], lds[wave][4], rsbuf[wave][0], rsbuf[wave][1]};
    const uint64_t nwaves = (uint64_t)gridDim.x * WPB;
    const uint64_t n_rec = b.n;
    for (uint64_t rec = (uint64_t)blockIdx.x * WPB + wave; rec < n_rec; rec += nwaves) {
        // (the arguments — forty pointers and the configuration — are read from the kernarg segment where a record is worked on, not
        //  held in scalar registers across the loop: qc_stat_kernel's trick, round 5)
#if defined(__HIP_DEVICE_COMPILE__)
        const FilterArgs __attribute__((address_space(4)))* ka = (const FilterArgs __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const DevBatch bb = ka->b;
        const DevCircles ci = ka->circ;
        const DevStats ss = ka->st;
        // (the configuration is read in place: barcode_verify is indexed at run time, a copy would live in scratch)
        process_record_wave(bb, rec, ka->cfg, ci, w, ka->results, acc, ss, rec < ka->accum_limit);
#else
        process_record_wave(b, rec, cfg, circ, w, results, acc, st, rec < accum_limit);
#endif
    }
    __syncthreads();
    flush_block_acc(acc, st);
}

// The same pipeline over an explicit list of record indices (the pairs the lane-per-read kernel deferred);
// the list length lives in device memory, so the launch needs no host round trip.
__global__ __launch_bounds__(BLOCK) void filter_overlap_list_kernel(DevBatch b, aqc_config cfg, DevCircles circ,
                                                                    aqc_result* __restrict__ results, DevStats st,
                                                                    uint64_t accum_limit, const uint32_t* __restrict__ list,
                                                                    const unsigned int* __restrict__ n_list) {
    __shared__ uint8_t lds[WPB][5][LSTR];
    __shared__ uint8_t rsbuf[WPB][2][64];
    __shared__ BlockAcc acc;
    const unsigned int n = *n_list;
    if (n == 0) return;
    const int wave = threadIdx.x / WAVE;
    for (int i = threadIdx.x; i < (int)(sizeof(BlockAcc) / 4); i += BLOCK) ((unsigned int*)&acc)[i] = 0;
    __syncthreads();
    const WaveLds w{lds[wave][0], lds[wave][1], lds[wave][2], lds[wave][3], lds[wave][4], rsbuf[wave][0], rsbuf[wave][1]};
    const unsigned int nwaves = gridDim.x * WPB;
    for (unsigned int i = blockIdx.x * WPB + wave; i < n; i += nwaves) {
        const uint64_t rec = list[i];
#if defined(__HIP_DEVICE_COMPILE__)
        const FilterArgs __attribute__((address_space(4)))* ka = (const FilterArgs __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const DevBatch bb = ka->b;
        const DevCircles ci = ka->circ;
        const DevStats ss = ka->st;
        // (the configuration is read in place: barcode_verify is indexed at run time, a copy would live in scratch)
        process_record_wave(bb, rec, ka->cfg, ci, w, ka->results, acc, ss, rec < ka->accum_limit);
#else
        process_record_wave(b, rec, cfg, circ, w, results, acc, st, rec < accum_limit);
#endif
    }
    __syncthreads();
    flush_block_acc(acc, st);
}

// ------------------------------------------------------------------------------------------------
// QualityControl.statRead (qualitycontrol.py:73-122): one wave per read, lane = cycle.
// Block-private u32 accumulators in LDS, flushed with 64-bit global atomics at the end.
// k-mers go to an open-addressing table in HBM keyed by the k raw bytes (k <= 8).
// ------------------------------------------------------------------------------------------------
struct KmerTable {
    // open-addressing table for k-mers containing anything but A,C,G,T (rare): keyed by the k raw bytes
    unsigned long long* keys;    // 0 = empty
    unsigned long long* counts;
    unsigned long long* order;   // min over 2*t (seen) / 2*t+1 (inserted as reverse complement)
    uint64_t mask;               // capacity - 1
    // dense tables for pure A/C/G/T k-mers, 4^k entries.  Index = (bit-1 plane << k) | bit-0 plane of the
    // per-base code (c >> 1) & 3 (A=0 C=1 T=2 G=3); base j of the k-mer sits at bit j of each plane.
    // One copy of the dense tables PER XCD (8 on MI355X): a wave updates the copy of the XCD it runs on with
    // atomics that execute in that XCD's L2 (workgroup scope is enough: every accessor of a copy shares the L2),
    // instead of device-scope atomics that have to travel to the memory side.  Copies are summed / min-ed when
    // the dictionary is read back.
    unsigned int* dense_count;         // [N_XCD][4^k]
    unsigned long long* dense_first;   // [N_XCD][4^k] smallest scan time t at which the k-mer was seen (~0 = never)
    // complete[b] != 0: every dense entry of reduce-workgroup b has a first-seen time (written by kmer_reduce_kernel).
    // Time keys only grow from launch to launch, so once every entry has one no later launch can lower any of them
    // and kmer_count_kernel stops probing the first-seen table (for random DNA that is after ~10^4 reads).
    unsigned int* complete;            // [DENSE_ENTRIES / KRED_ENTRIES]
};
constexpr int N_XCD = 8;
constexpr uint32_t DENSE_ENTRIES = 1u << 16;   // 4^8
constexpr int KRED_BLOCK = 256;
constexpr int KRED_ENTRIES = 256;       // dense entries per kmer_reduce_kernel workgroup

// id of the XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (N_XCD - 1); }

// reverse complement of a dense k-mer index: complement flips the code's high bit, the order of bases reverses
__device__ __host__ inline uint32_t dense_rc(uint32_t idx, int k) {
    // reverse the order of the 2-bit codes and complement each (A0 <-> T2, C1 <-> G3: code ^ 2)
    uint32_t r = 0;
    for (int j = 0; j < k; ++j) r |= (((idx >> (2 * j)) & 3u) ^ 2u) << (2 * (k - 1 - j));
    return r;
}

__device__ __forceinline__ uint64_t hash64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__device__ inline long long kmer_slot(const KmerTable& t, unsigned long long key) {
    uint64_t h = hash64(key) & t.mask;
    for (uint64_t probe = 0; probe <= t.mask; probe++) {
        unsigned long long cur = t.keys[h];
        if (cur == key) return (long long)h;
        if (cur == 0) {
            unsigned long long prev = atomicCAS(&t.keys[h], 0ull, key);
            if (prev == 0 || prev == key) return (long long)h;
        }
        h = (h + 1) & t.mask;
    }
    return -1;
}

// both slots of a k-mer and its reverse complement: the two first probes travel together (the common case, both
// keys already present, costs ONE memory round trip)
__device__ __forceinline__ void kmer_slot2(const KmerTable& t, unsigned long long key, unsigned long long rkey,
                                           long long& h, long long& hr) {
    const uint64_t a = hash64(key) & t.mask, b = hash64(rkey) & t.mask;
    const unsigned long long ka = t.keys[a], kb = t.keys[b];
    h = ka == key ? (long long)a : kmer_slot(t, key);
    hr = kb == rkey ? (long long)b : kmer_slot(t, rkey);
}

// ------------------------------------------------------------------------------------------------
// Read descriptors for the sampling kernels.  A wavefront that walks its reads one by one pays the chain
// len/offset -> verdict record -> bases -> table probe as four DEPENDENT memory round trips per read (~6 us on
// this chip), which is what bounded both sampling kernels.  Instead lane j fetches the descriptor of the
// wave's j-th read (64 descriptors per two round trips), the loop broadcasts one descriptor per iteration with
// v_readlane, and the bases of read j+1 are loaded into registers while read j is being accumulated.
//   len  < 0 : not part of the sample (verdict not GOOD / beyond the range)
//   e[k]     : the walk's k-th edit as it applies to THIS mate in final-read coordinates,
//              pos << 16 | new base << 8 | new quality   (base 0 = keep the base: a mask edit; pos 0xffff = none)
// ------------------------------------------------------------------------------------------------
struct ReadDesc {
    unsigned long long s, q;
    int len;
    int qlen;                // length of the quality view (== len unless the record's quality line has a length of its own)
    unsigned int e[3];
};

__device__ __forceinline__ ReadDesc lane_desc(const DevBatch& b, int mate, uint64_t rec, bool valid, int post,
                                              const aqc_result* __restrict__ results) {
    ReadDesc d;
    d.s = d.q = 0ull;
    d.len = -1;
    d.qlen = -1;
    d.e[0] = d.e[1] = d.e[2] = 0xffff0000u;
    if (!valid) return d;
    int len, st = 0;
    uint32_t lw;
    if (mate == 0) {
        lw = b.len1[rec];
        const uint64_t o = b.off1[rec];
        d.s = (unsigned long long)(b.seq1 + o);
        d.q = (unsigned long long)(b.qual1 + (b.qoff1 ? b.qoff1[rec] : o));
    } else {
        lw = b.len2[rec];
        const uint64_t o = b.off2[rec];
        d.s = (unsigned long long)(b.seq2 + o);
        d.q = (unsigned long long)(b.qual2 + (b.qoff2 ? b.qoff2[rec] : o));
    }
    len = (int)(lw & LEN_MASK);
    // this mate's quality line has a length of its own: its view comes from qlen (raw read) / qview (final read)
    const bool irr = (lw & LEN_IRR) != 0u && b.qlen1 != nullptr;
    int qst = 0, qlen = irr ? (int)((mate == 0 ? b.qlen1[rec] : b.qlen2[rec]) & QLEN_MASK) : len;
    if (post) {
        // the 32-byte verdict record as two 16-byte loads; fields by shifts (aqc_result is packed, see the header)
        const uint4* rp = reinterpret_cast<const uint4*>(results + rec);
        const uint4 w0 = rp[0], w1 = rp[1];
        if ((w0.x & 0xffu) != (unsigned int)AQC_GOOD) return d;   // only good records reach preprocesser.py:624-627
        const int n_edits = (int)((w0.x >> 8) & 0xffu);
        const int len1 = (int)(w0.y & 0xffffu), len2 = (int)(w0.z & 0xffffu), ovl = (int)(w0.w & 0xffffu);
        st = mate == 0 ? (int)(w0.x >> 16) : (int)(w0.y >> 16);
        len = mate == 0 ? len1 : len2;
        qst = st; qlen = len;
        if (irr) {
            const uint32_t qv = mate == 0 ? b.qview1[rec] : b.qview2[rec];
            qst = (int)(qv & 0xffffu); qlen = (int)(qv >> 16);
        }
        const unsigned long long e_lo = ((unsigned long long)w1.y << 32) | w1.x, e_hi = ((unsigned long long)w1.w << 32) | w1.z;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            if (e < n_edits) {
                // edit e = 5 bytes at byte 5e of the 16: o (u16), kind, base, qual
                const int bit = 40 * e;
                unsigned long long v = bit < 64 ? e_lo >> bit : 0ull;
                if (bit + 40 > 64) v |= bit < 64 ? e_hi << (64 - bit) : e_hi >> (bit - 64);
                const int o = (int)(v & 0xffffu);
                const unsigned int kind = (unsigned int)(v >> 16) & 0xffu, base = (unsigned int)(v >> 24) & 0xffu, qual = (unsigned int)(v >> 32) & 0xffu;
                const unsigned int pos = mate == 0 ? (unsigned int)(len1 - ovl + o) : (unsigned int)(len2 - 1 - o);
                if (kind == AQC_EDIT_MASK) d.e[e] = (pos << 16) | (unsigned int)'!';
                else if ((kind == AQC_EDIT_FIX_R1 && mate == 0) || (kind == AQC_EDIT_FIX_R2 && mate == 1))
                    d.e[e] = (pos << 16) | (base << 8) | qual;
            }
        }
    }
    d.s += (unsigned int)st;
    d.q += (unsigned int)(post ? qst : 0);
    d.len = len;
    d.qlen = qlen;
    return d;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int j) {
    const unsigned int lo = __builtin_amdgcn_readlane((int)(unsigned int)v, j);
    const unsigned int hi = __builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), j);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ ReadDesc bcast_desc(const ReadDesc& d, int j) {
    ReadDesc o;
    o.s = readlane64(d.s, j);
    o.q = readlane64(d.q, j);
    o.len = __builtin_amdgcn_readlane(d.len, j);
    o.qlen = __builtin_amdgcn_readlane(d.qlen, j);
    o.e[0] = (unsigned int)__builtin_amdgcn_readlane((int)d.e[0], j);
    o.e[1] = (unsigned int)__builtin_amdgcn_readlane((int)d.e[1], j);
    o.e[2] = (unsigned int)__builtin_amdgcn_readlane((int)d.e[2], j);
    return o;
}

// four consecutive bytes of a read starting at byte x; bytes at or beyond `len` read as the (byte-uniform) `pad`.
// Branch-free so that a prefetch stays asynchronous: ONE unaligned dword load from an address clamped into the
// read (len >= 4), then a funnel shift brings the pad in from the top.
__device__ __forceinline__ uint32_t load4(const uint8_t* p, int x, int len, uint32_t pad) {
    const int xa = min(x, len - 4);
    uint32_t dw = pad;
    if (x < len) dw = gload_u32(p + xa);      // (global_load: the pointer came through v_readlane and would otherwise be a flat_load)
    return __builtin_amdgcn_alignbit(pad, dw, (unsigned)(8 * (x - xa)) & 31u);
}

// the walk's edits that fall into the dword at byte offset x (d is wave-uniform, so the outer tests are scalar)
// (IRR: the caller may meet reads whose quality view has a length of its own — qc_stat_kernel; the fused k-mer kernel, which runs at
//  its register limit, leaves such reads' per-cycle statistics to that kernel and only ever needs the bases here)
template <bool IRR>
__device__ __forceinline__ void apply_edits(const ReadDesc& d, int x, uint32_t& ws, uint32_t& wq) {
    if (IRR && d.qlen != d.len) {
        // a quality view of its own length: the walk indexed it from ITS end (preprocesser.py:566-567) — position + (qlen - len),
        // a negative index wrapped the python way; edits in order, a later one wins (d is wave-uniform: scalar branches)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const unsigned int ed = d.e[e];
            if ((ed >> 16) != 0xffffu) {
                const unsigned int rel = (ed >> 16) - (unsigned int)x;
                if (rel < 4u && ((ed >> 8) & 0xffu)) ws = (ws & ~(0xffu << (8u * rel))) | (((ed >> 8) & 0xffu) << (8u * rel));
                int qp = (int)(ed >> 16) + d.qlen - d.len;
                if (qp < 0) qp += d.qlen;
                const unsigned int relq = (unsigned int)(qp - x);
                if (qp >= 0 && relq < 4u) wq = (wq & ~(0xffu << (8u * relq))) | ((ed & 0xffu) << (8u * relq));
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const unsigned int ed = d.e[e];
        if ((ed >> 16) != 0xffffu) {
            const unsigned int rel = (ed >> 16) - (unsigned int)x;
            if (rel < 4u) {
                const unsigned int sh = 8u * rel;
                if ((ed >> 8) & 0xffu) ws = (ws & ~(0xffu << sh)) | (((ed >> 8) & 0xffu) << sh);
                wq = (wq & ~(0xffu << sh)) | ((ed & 0xffu) << sh);
            }
        }
    }
}

// 0x80 in every byte of x that is not zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) {
    return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}

constexpr uint32_t CODE_TO_BASE = 0x47544341u;   // 2-bit code (c >> 1) & 3 -> 'A' 'C' 'T' 'G'

// ------------------------------------------------------------------------------------------------
// Per-cycle accumulators of statRead (qualitycontrol.py:73-111).  A lane owns FOUR consecutive cycles of the
// read: one unaligned dword of bases and one of qualities live in registers (no LDS staging), the neighbours'
// dwords come over the wave for the 5-wide discontinuity window, and a base's count and quality sum travel in
// ONE LDS atomic (count << 20 | quality sum; at most 4095 reads per workgroup).  total_num / total_qual are
// column sums of the five rows (A T C G other), formed when the workgroup flushes.
// ------------------------------------------------------------------------------------------------
constexpr int QC_BLOCK = 1024;
constexpr int QC_WPB = QC_BLOCK / WAVE;
constexpr int QC_LDS_ROWS = 6;            // A T C G other | discontinuity   (+ gc histogram)
constexpr int QC_MAX_READS_PER_BLOCK = 4095;

struct QcLds {
    unsigned int* accs;          // [QC_LDS_ROWS][cols], columns permuted (see qc_accumulate_read)
    unsigned int* gch;           // [cols] GC histogram
    unsigned long long* scal;    // [0] totalKmer, [1] reads
    int cols;
};

// statRead's per-cycle part for ONE read (wave-wide).  ws / wq: the lane's dword of bases / qualities of pass 0
// (bytes 4*lane .. 4*lane+3, zero beyond the read), without the walk's edits.
// Column i lives at word (i & 3) * (cols / 4) + (i >> 2): the four cycles a lane owns are cols/4 words apart and
// neighbouring lanes hit neighbouring banks (the natural layout would be a 4-way bank conflict on every add).
template <bool IRR>
__device__ __forceinline__ void qc_accumulate_read(const ReadDesc& cur, uint32_t ws, uint32_t wq, const QcLds& L, int kmer_len) {
    const int lane = lane_id();
    const int len = cur.len;
    const int qlen = IRR ? cur.qlen : cur.len;               // (the pass-0 dword of qualities was loaded against it)
    const int cols = L.cols, cq = L.cols >> 2;
    unsigned int* const accs = L.accs;
    unsigned int* const gch = L.gch;
    unsigned long long* const scal = L.scal;
    const uint8_t* gs = reinterpret_cast<const uint8_t*>(cur.s);
    const uint8_t* gq = reinterpret_cast<const uint8_t*>(cur.q);
    int gc = 0;
    unsigned int d_head = 0, d_tail = 0;      // discontinuity of cycle 2 / cycle len-3: the clamped windows
    for (int base0 = 0; base0 < len; base0 += 4 * WAVE) {
        const int x = base0 + 4 * lane;
        if (base0 > 0) { ws = load4(gs, x, len, 0); wq = load4(gq, x, qlen, 0); }
        apply_edits<IRR>(cur, x, ws, wq);
        uint32_t prev = __shfl_up(ws, 1), next = __shfl_down(ws, 1);
        if (base0 > 0 && lane == 0) { uint32_t dq = 0; prev = load4(gs, x - 4, len, 0); apply_edits<false>(cur, x - 4, prev, dq); }
        if (base0 + 4 * WAVE < len && lane == WAVE - 1) { uint32_t dq = 0; next = load4(gs, x + 4, len, 0); apply_edits<false>(cur, x + 4, next, dq); }
        // bytes x-2 .. x+5 ; byte k of (v ^ v >> 8) is non-zero iff bases x-2+k and x-1+k differ
        const uint32_t vlo = __builtin_amdgcn_alignbit(ws, prev, 16), vhi = __builtin_amdgcn_alignbit(next, ws, 16);
        const uint32_t tlo = nonzero_bytes(vlo ^ __builtin_amdgcn_alignbit(vhi, vlo, 8));
        const uint32_t thi = nonzero_bytes(vhi ^ (vhi >> 8));
        unsigned int dpk = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) dpk |= (unsigned int)__popc(__builtin_amdgcn_alignbit(thi, tlo, 8 * j)) << (3 * j);
        if (base0 == 0) d_head = ((unsigned int)__builtin_amdgcn_readlane((int)dpk, 0) >> 6) & 7u;
        const int tl = len - 3 - base0;                 // cycle len-3 relative to this pass (uniform)
        if (tl >= 0 && tl < 4 * WAVE)
            d_tail = ((unsigned int)__builtin_amdgcn_readlane((int)dpk, tl >> 2) >> (3 * (tl & 3))) & 7u;
        // the lane's four cycles at once: accumulator row per base (A T C G = 0..3, anything else 4), quality - 33 per
        // byte, the clamped discontinuity windows patched into the packed fields; then per cycle only two field
        // extractions, one multiply-add for the address and the atomics remain
        if (base0 == 0 && lane == 0) dpk = (dpk & ~0x3fu) | d_head | (d_head << 3);     // cycles 0, 1: window [0, 5)
        {
            const int over = min(max(x + 3 - (len - 3), 0), 4);                         // cycles beyond len-3: window [len-5, len)
            const unsigned int m = over ? (0xfffu << (3 * (4 - over))) & 0xfffu : 0u;
            dpk = (dpk & ~m) | ((d_tail * 0x249u) & m);
        }
        if (IRR && qlen < len) {
            // A quality line shorter than the read (qualitycontrol.py:81-88): totalNum[i] is counted, then qual[i] raises and the
            // rest of the position is skipped — no quality sum, no base count, no G/C, no discontinuity.  In these accumulators
            // that is a "foreign" base (byte 0: row 4, which only feeds the total_num / total_qual column sums) of quality '!' = 0
            // whose discontinuity is dropped; the discontinuities of the cycles before it were formed from the real bases above.
            // (Edited in place: the kernels that inline this run at their register limit.)
            const int qin = min(max(qlen - x, 0), 4);
            const uint32_t qm = qin >= 4 ? 0xffffffffu : ((1u << (8 * qin)) - 1u);
            ws &= qm;
            wq = (wq & qm) | (0x21212121u & ~qm);
            dpk &= qin >= 4 ? 0xfffu : ((1u << (3 * qin)) - 1u);
        }
        const uint32_t codes = (ws >> 1) & 0x03030303u;
        const uint32_t bad = __builtin_amdgcn_perm(0u, CODE_TO_BASE, codes) ^ ws;       // 0 where the byte is A/C/G/T
        const uint32_t nz = nonzero_bytes(bad);                                         // 0x80 per foreign byte
        const uint32_t foreign = nz | (nz - (nz >> 7));                                 // 0xff per foreign byte
        // code (A0 C1 T2 G3) -> row (A0 T1 C2 G3); foreign -> 4
        const uint32_t rows4 = (__builtin_amdgcn_perm(0u, 0x03010200u, codes) & ~foreign) | (0x04040404u & foreign);
        const uint32_t qn4 = wq - 0x21212121u;                                          // (qualities are >= '!' in FASTQ)
        const int nin = min(max(len - x, 0), 4);                                        // cycles of this lane inside the read
        const unsigned int col0 = (unsigned int)lane + (unsigned int)(base0 >> 2);
        // G / C among the lane's cycles inside the read (C = code 1, G = code 3: low code bit), not foreign
        const uint32_t gcm = codes & 0x01010101u & ~foreign & (nin >= 4 ? 0x01010101u : ((1u << (8 * nin)) - 1u));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nin) {
                const unsigned int row = (rows4 >> (8 * j)) & 0xffu;
                atomicAdd(&accs[row * (unsigned int)cols + (unsigned int)(j * cq) + col0], (1u << 20) + ((qn4 >> (8 * j)) & 0xffu));
                // discontinuity over the 5-wide window clamped to the read (qualitycontrol.py:97-109)
                const unsigned int d = (dpk >> (3 * j)) & 7u;
                if (d) atomicAdd(&accs[5u * (unsigned int)cols + (unsigned int)(j * cq) + col0], d);
            }
            gc += __popcll(__ballot((gcm >> (8 * j)) & 1u));
        }
    }
    if (lane == 0) {
        atomicAdd(&gch[gc], 1u);
        atomicAdd(&scal[1], 1ull);
        if (len > kmer_len) atomicAdd(&scal[0], (unsigned long long)(len - kmer_len));
    }
}

__global__ __launch_bounds__(QC_BLOCK) void qc_stat_kernel(DevBatch b, int mate, uint64_t first, uint64_t count, int post,
                                                           const aqc_result* __restrict__ results, int kmer_len,
                                                           unsigned long long* __restrict__ qc /* [QC_ROWS*QC_COLS] */,
                                                           int* status, int cols, int only_irr) {
    // (only_irr: the reads whose quality view has a length of its own, and nothing else — the fused k-mer kernel has left exactly
    //  those reads' per-cycle statistics to this one)
    // dynamic LDS, sized by the longest read of the batch (cols = multiple of 64 <= 1024)
    extern __shared__ __attribute__((aligned(16))) unsigned int qc_smem[];
    unsigned int* const accs = qc_smem;                               // accs[row * cols + i]
    unsigned int* const gch = qc_smem + QC_LDS_ROWS * cols;
    unsigned long long* const scal = reinterpret_cast<unsigned long long*>(gch + cols);
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    for (int i = threadIdx.x; i < (QC_LDS_ROWS + 1) * cols; i += QC_BLOCK) qc_smem[i] = 0;
    if (threadIdx.x < 2) scal[threadIdx.x] = 0;
    __syncthreads();
    const int cq = cols >> 2;
    const uint64_t nwaves = (uint64_t)gridDim.x * QC_WPB;
    if ((count + nwaves - 1) / nwaves * QC_WPB > (uint64_t)QC_MAX_READS_PER_BLOCK) {      // host sizes the grid; never silently overflow
        if (threadIdx.x == 0) atomicCAS(status, 0, AQC_ERR_STATE);
        return;
    }
    auto usable = [&](const ReadDesc& d) { return d.len >= 5 && d.len <= AQC_MAX_READ_LEN && d.len <= cols && (!only_irr || d.qlen != d.len); };
    for (uint64_t kb = (uint64_t)blockIdx.x * QC_WPB + wave; kb < count; kb += nwaves * WAVE) {
        const uint64_t myk = kb + (uint64_t)lane * nwaves;
#if defined(__HIP_DEVICE_COMPILE__)
        const DevBatch __attribute__((address_space(4)))* kb_args = (const DevBatch __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb_args));          // (read where it is used, not held across the loop: see kmer_count_kernel)
        const DevBatch bb = *kb_args;              // (the batch descriptor is the kernel's first argument)
#else
        const DevBatch bb = b;
#endif
        const ReadDesc mine = lane_desc(bb, mate, first + myk, myk < count, post, results);
        const int nr = (int)min((uint64_t)WAVE, (count - kb + nwaves - 1) / nwaves);
        ReadDesc cur = bcast_desc(mine, 0);
        uint32_t pre_s = 0, pre_q = 0;
        if (usable(cur)) {
            pre_s = load4(reinterpret_cast<const uint8_t*>(cur.s), 4 * lane, cur.len, 0);
            pre_q = load4(reinterpret_cast<const uint8_t*>(cur.q), 4 * lane, cur.qlen, 0);
        }
        for (int r = 0; r < nr; ++r) {
            uint32_t ws = pre_s, wq = pre_q;
            ReadDesc nxt = cur;
            if (r + 1 < nr) {
                nxt = bcast_desc(mine, r + 1);
                if (usable(nxt)) {
                    pre_s = load4(reinterpret_cast<const uint8_t*>(nxt.s), 4 * lane, nxt.len, 0);
                    pre_q = load4(reinterpret_cast<const uint8_t*>(nxt.q), 4 * lane, nxt.qlen, 0);
                }
            }
            const int len = cur.len;
            if (only_irr) { }                                                                    // (the fused kernel has reported these)
            else if (len > AQC_MAX_READ_LEN || len > cols) { if (lane == 0) atomicCAS(status, 0, AQC_ERR_READ_TOO_LONG); }
            else if (len < 5 && len > 0) { if (lane == 0) atomicCAS(status, 0, AQC_ERR_ARG); }   // IndexError upstream (:106-107)
            if (usable(cur)) qc_accumulate_read<true>(cur, ws, wq, QcLds{accs, gch, scal, cols}, kmer_len);
            cur = nxt;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cols; i += QC_BLOCK) {
        const int ci = (i & 3) * cq + (i >> 2);
        unsigned long long tn = 0, tq = 0;
#pragma unroll
        for (int row = 0; row < 5; ++row) {
            const unsigned int v = accs[row * cols + ci];
            const unsigned long long cnt = v >> 20, qs = v & 0xfffffu;
            if (row < 4 && v) {
                atomicAdd(&qc[(AQC_QC_BASE_COUNT_A + row) * AQC_QC_COLS + i], cnt);
                atomicAdd(&qc[(AQC_QC_BASE_QUAL_A + row) * AQC_QC_COLS + i], qs);
            }
            tn += cnt; tq += qs;
        }
        if (tn) {
            atomicAdd(&qc[AQC_QC_TOTAL_NUM * AQC_QC_COLS + i], tn);
            atomicAdd(&qc[AQC_QC_TOTAL_QUAL * AQC_QC_COLS + i], tq);
        }
        const unsigned int dv = accs[5 * cols + ci];
        if (dv) atomicAdd(&qc[AQC_QC_DISCONTINUITY * AQC_QC_COLS + i], (unsigned long long)dv);
        if (gch[i]) atomicAdd(&qc[AQC_QC_GC_HIST * AQC_QC_COLS + i], (unsigned long long)gch[i]);
    }
    if (threadIdx.x < 2 && scal[threadIdx.x]) atomicAdd(&qc[AQC_QC_SCALARS * AQC_QC_COLS + threadIdx.x], scal[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// k-mer dictionary of statRead (qualitycontrol.py:113-122), counting part.
// Global atomics top out near 3e10 /s on this chip — 28 M k-mers of a 200 k-read sample would cost ~0.9 ms —
// so the 4^k counters live in LDS: one 1024-thread workgroup per CU keeps a private table of 65536 u16 counters
// (128 KiB) and works in ROUNDS of at most 65535 k-mers (no counter can overflow), then streams the table to
// its own slice of `partial` with plain coalesced stores; kmer_reduce_kernel adds the slices up.  No global
// atomic is issued for a pure A/C/G/T k-mer except the (rare, load-guarded) first-seen minimum.
// A lane owns four consecutive k-mer start positions: its dword of bases is packed to 4 x 2-bit codes, two
// wave shifts assemble the codes of 16 consecutive bases, and the dense index of position j is
// (window >> 2j) & (4^k - 1) — base q of the k-mer at bits 2q..2q+1, code (c >> 1) & 3 = A0 C1 T2 G3.
// K-mers containing anything else go to the open-addressing table (byte keys).
// Descriptors are fetched lane-parallel and the bases of the next read are prefetched (see ReadDesc).
// ------------------------------------------------------------------------------------------------
#ifdef AQC_PROFILE
__device__ unsigned long long g_kprof[16];
#define KPROF_DECL unsigned long long kp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long kp_last = __builtin_amdgcn_s_memtime();
#define KPROF(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); kp_t[k] += now_ - kp_last; kp_last = now_; } while (0)
#define KPROF_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_kprof[k_], kp_t[k_]); } while (0)
#else
#define KPROF_DECL
#define KPROF(k)
#define KPROF_FLUSH
#endif

constexpr int KMER_BLOCK = 1024;                // (round 6: 512 threads at <= 128 registers, to leave room for the writer's kernels beside it, measured slower —
                                                //  the step 3.90 - 3.95 -> 4.10 - 4.22 ms, the k-mer launches 0.265 -> 0.35 ms: profiles/r06_copy_window_grid.txt)
constexpr int KMER_WPB = KMER_BLOCK / WAVE;
constexpr int KMER_EXQ = 1536;                  // LDS queue of k-mers bound for the open-addressing table (24 KiB)
constexpr size_t KMER_LDS_BYTES = DENSE_ENTRIES * 2 + (size_t)KMER_EXQ * 16 + 16;
constexpr int KMER_FUSED_MAX_COLS = 256;         // widest QC accumulator block that still fits behind the table (160 KiB LDS)
constexpr size_t KMER_FUSED_LDS_BYTES = KMER_LDS_BYTES + sizeof(unsigned int) * (QC_LDS_ROWS + 1) * KMER_FUSED_MAX_COLS + 16;
constexpr int KMER_PASS_LANES = WAVE - 2;        // the last two lanes of a pass only supply bases to their neighbours

__global__ __launch_bounds__(KMER_BLOCK) void kmer_count_kernel(DevBatch b, int mate, uint64_t first, uint64_t count, int post,
                                                                const aqc_result* __restrict__ results, int kmer_len,
                                                                KmerTable kt, unsigned long long order_base,
                                                                uint16_t* __restrict__ partial, uint32_t reads_per_round,
                                                                uint32_t n_rounds, int* status,
                                                                unsigned long long* __restrict__ qc /* [QC_ROWS*QC_COLS] or null */,
                                                                int cols) {
    extern __shared__ __attribute__((aligned(16))) unsigned int ktab[];     // 32768 words = 65536 u16 counters
    unsigned long long* const exq_key = reinterpret_cast<unsigned long long*>(ktab + DENSE_ENTRIES / 2);   // parked exotic k-mers
    unsigned long long* const exq_t = exq_key + KMER_EXQ;
    unsigned int* const exq_n = reinterpret_cast<unsigned int*>(exq_t + KMER_EXQ);
    // fused mode (qc != null, reads <= 256 bases): the per-cycle accumulators of statRead ride along — same descriptor,
    // same dword of bases — in the LDS left over behind the k-mer table, and fill the issue slots this kernel idles in
    const bool fused = qc != nullptr;
    unsigned int* const q_accs = exq_n + 4;
    unsigned int* const q_gch = q_accs + QC_LDS_ROWS * cols;
    unsigned long long* const q_scal = reinterpret_cast<unsigned long long*>(q_gch + cols);
    const QcLds qlds{q_accs, q_gch, q_scal, cols};
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    if (fused) {
        for (int i = threadIdx.x; i < (QC_LDS_ROWS + 1) * cols; i += KMER_BLOCK) q_accs[i] = 0;
        if (threadIdx.x < 2) q_scal[threadIdx.x] = 0;
    }
    const unsigned long long kmask = kmer_len >= 8 ? ~0ull : (1ull << (8 * kmer_len)) - 1ull;
    // k-mer with a symbol outside A/C/G/T -> open-addressing table, keyed by its bytes (first seen at time t)
    auto exotic_insert = [&](unsigned long long key, unsigned long long t) {
        unsigned long long rkey = 0;
        for (int qq = 0; qq < kmer_len; qq++)
            rkey |= (unsigned long long)comp_or_n((uint8_t)(key >> (8 * (kmer_len - 1 - qq)))) << (8 * qq);
        long long h, hr;
        kmer_slot2(kt, key, rkey, h, hr);
        if (h < 0 || hr < 0) atomicCAS(status, 0, AQC_ERR_UNSUPPORTED);
        else {
            // (the minima are load-guarded: after its first occurrence a k-mer costs one atomic, not three)
            const unsigned long long oa = kt.order[h], ob = kt.order[hr];
            atomicAdd(&kt.counts[h], 1ull);
            if (oa > 2 * t) atomicMin(&kt.order[h], 2 * t);
            if (ob > 2 * t + 1) atomicMin(&kt.order[hr], 2 * t + 1);
        }
    };
    const uint32_t imask = (1u << (2 * kmer_len)) - 1u, kbits = (1u << kmer_len) - 1u;
    unsigned long long* const my_first = kt.dense_first + (size_t)xcc_id() * DENSE_ENTRIES;
    auto usable = [&](const ReadDesc& d) { return d.len >= 5 && d.len <= AQC_MAX_READ_LEN && d.len > kmer_len; };
    // (a read whose quality view has a length of its own is left to qc_stat_kernel's only_irr pass: its per-cycle statistics, not its k-mers)
    auto qc_usable = [&](const ReadDesc& d) { return fused && d.len >= 5 && d.len <= AQC_MAX_READ_LEN && d.len <= cols && d.qlen == d.len; };
    KPROF_DECL
    // every dense k-mer already has a first-seen time from an earlier launch: nothing this launch sees can be earlier
    const bool complete = __syncthreads_and(threadIdx.x < (int)(DENSE_ENTRIES / KRED_ENTRIES) ? (int)kt.complete[threadIdx.x] : 1) != 0;
    constexpr uint32_t PAD = 0x41414141u;      // 'AAAA': bases beyond the read never reach a counted k-mer
    for (uint32_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        for (int i = threadIdx.x; i < (int)(DENSE_ENTRIES / 2); i += KMER_BLOCK) ktab[i] = 0;
        if (threadIdx.x == 0) *exq_n = 0;
        __syncthreads();
        KPROF(0);
        const uint64_t r_lo = (uint64_t)round * reads_per_round;
        const uint64_t r_hi = min(r_lo + reads_per_round, count);
        for (uint64_t kb = r_lo + wave; kb < r_hi; kb += (uint64_t)KMER_WPB * WAVE) {
            const uint64_t myk = kb + (uint64_t)lane * KMER_WPB;
            // (the batch descriptor — twenty pointers — is read from the kernarg segment where it is used, once per 64 reads: held in
            //  scalar registers across the loop it was most of this kernel's SGPR spills)
#if defined(__HIP_DEVICE_COMPILE__)
            const DevBatch __attribute__((address_space(4)))* kb_args = (const DevBatch __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kb_args));
            const DevBatch bb = *kb_args;              // (the batch descriptor is the kernel's first argument)
#else
            const DevBatch bb = b;
#endif
            const ReadDesc mine = lane_desc(bb, mate, first + myk, myk < r_hi, post, results);
            const int nr = (int)min((uint64_t)WAVE, (r_hi - kb + KMER_WPB - 1) / KMER_WPB);
            ReadDesc cur = bcast_desc(mine, 0);
            uint32_t pre = PAD, pre_q = 0;
            if (qc_usable(cur) || usable(cur)) {
                pre = load4(reinterpret_cast<const uint8_t*>(cur.s), 4 * lane, cur.len, PAD);
                if (fused) pre_q = load4(reinterpret_cast<const uint8_t*>(cur.q), 4 * lane, cur.qlen, 0);
            }
            KPROF(1);
            for (int r = 0; r < nr; ++r) {
                uint32_t ws = pre;
                const uint32_t wq0 = pre_q;
                ReadDesc nxt = cur;
                if (r + 1 < nr) {
                    nxt = bcast_desc(mine, r + 1);
                    if (qc_usable(nxt) || usable(nxt)) {
                        pre = load4(reinterpret_cast<const uint8_t*>(nxt.s), 4 * lane, nxt.len, PAD);
                        if (fused) pre_q = load4(reinterpret_cast<const uint8_t*>(nxt.q), 4 * lane, nxt.qlen, 0);
                    }
                }
                if (fused) {
                    const int len = cur.len;
                    if (len > AQC_MAX_READ_LEN || len > cols) { if (lane == 0) atomicCAS(status, 0, AQC_ERR_READ_TOO_LONG); }
                    else if (len < 5 && len > 0) { if (lane == 0) atomicCAS(status, 0, AQC_ERR_ARG); }   // IndexError upstream (:106-107)
                    if (qc_usable(cur)) {
                        // (the bases beyond the read are 'A' here, 0 in qc_stat_kernel: neither is ever looked at)
                        qc_accumulate_read<false>(cur, ws, wq0, qlds, kmer_len);
                    }
                }
                if (usable(cur)) {
                    const int len = cur.len;
                    const int nk = len - kmer_len;
                    const uint64_t k = kb + (uint64_t)r * KMER_WPB;
                    const unsigned long long t0 = (order_base + k) * (unsigned long long)AQC_QC_COLS;
                    for (int base0 = 0; base0 < nk; base0 += 4 * KMER_PASS_LANES) {
                        const int x = base0 + 4 * lane;
                        if (base0 > 0) ws = load4(reinterpret_cast<const uint8_t*>(cur.s), x, len, PAD);
                        uint32_t dq = 0;
                        apply_edits<false>(cur, x, ws, dq);                                           // (only the bases matter here)
                        const uint32_t codes = (ws >> 1) & 0x03030303u;
                        const uint32_t bad = __builtin_amdgcn_perm(0u, CODE_TO_BASE, codes) ^ ws;    // 0 where the byte is A/C/G/T
                        const uint32_t c8 = (codes | (codes >> 6) | (codes >> 12) | (codes >> 18)) & 0xffu;
                        const uint32_t c16 = c8 | ((uint32_t)__shfl_down(c8, 1) << 8);
                        const uint32_t win = c16 | ((uint32_t)__shfl_down(c16, 2) << 16);           // codes of bases x .. x+15
                        uint32_t badwin = 0;                                                         // bit q: base x+q is not A/C/G/T
                        if (__ballot(bad != 0u)) {
                            const uint32_t nz = nonzero_bytes(bad);
                            const uint32_t b4 = ((nz >> 7) | (nz >> 14) | (nz >> 21) | (nz >> 28)) & 0xfu;
                            const uint32_t b8 = b4 | ((uint32_t)__shfl_down(b4, 1) << 4);
                            badwin = b8 | ((uint32_t)__shfl_down(b8, 2) << 8);
                        }
                        KPROF(2);
                        uint32_t idx[4];
                        bool dense[4], exotic[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool act = lane < KMER_PASS_LANES && x + j < nk;
                            idx[j] = (win >> (2 * j)) & imask;
                            exotic[j] = act && ((badwin >> j) & kbits) != 0u;
                            dense[j] = act && !exotic[j];
                            if (dense[j]) atomicAdd(&ktab[idx[j] >> 1], 1u << (16 * (idx[j] & 1)));
                        }
                        KPROF(3);
                        if (!complete) {
                            unsigned long long seen[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) seen[j] = dense[j] ? my_first[idx[j]] : 0ull;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const unsigned long long t = t0 + (unsigned long long)(x + j);
                                if (dense[j] && seen[j] > t) __hip_atomic_fetch_min(&my_first[idx[j]], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                        KPROF(4);
                        if (__ballot(badwin != 0u)) {
                            // byte keys straight from registers: bases x .. x+11 = own dword + the next two lanes'.
                            // The table insert is a chain of global round trips, so it is not done here: the
                            // k-mer is parked in the workgroup's LDS queue and inserted when the round ends.
                            const uint32_t w1 = (uint32_t)__shfl_down(ws, 1), w2 = (uint32_t)__shfl_down(ws, 2);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (exotic[j]) {
                                    const uint32_t lo = __builtin_amdgcn_alignbit(w1, ws, 8 * j), hi = __builtin_amdgcn_alignbit(w2, w1, 8 * j);
                                    const unsigned long long key = (((unsigned long long)hi << 32) | lo) & kmask;
                                    const unsigned long long t = t0 + (unsigned long long)(x + j);
                                    const unsigned int slot = atomicAdd(exq_n, 1u);
                                    if (slot < (unsigned int)KMER_EXQ) { exq_key[slot] = key; exq_t[slot] = t; }
                                    else exotic_insert(key, t);                       // queue full: insert in place
                                }
                            }
                        }
                    }
                }
                KPROF(5);
                cur = nxt;
            }
        }
        __syncthreads();
        KPROF(6);
        {
            const unsigned int nq = min(*exq_n, (unsigned int)KMER_EXQ);
            for (unsigned int e = threadIdx.x; e < nq; e += KMER_BLOCK) exotic_insert(exq_key[e], exq_t[e]);
        }
        KPROF(5);
        uint4* dst = reinterpret_cast<uint4*>(partial + (size_t)round * DENSE_ENTRIES);
        const uint4* srcv = reinterpret_cast<const uint4*>(ktab);
        for (int i = threadIdx.x; i < (int)(DENSE_ENTRIES * 2 / 16); i += KMER_BLOCK) dst[i] = srcv[i];
        // the table and the queue may be reused once every wave has READ them out of LDS: wait for the LDS reads
        // only (lgkmcnt), not for the slice stores and table atomics still in flight — they drain under the next round
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        KPROF(7);
    }
    if (fused) {
        __syncthreads();
        const int cq = cols >> 2;
        for (int i = threadIdx.x; i < cols; i += KMER_BLOCK) {
            const int ci = (i & 3) * cq + (i >> 2);
            unsigned long long tn = 0, tq = 0;
#pragma unroll
            for (int row = 0; row < 5; ++row) {
                const unsigned int v = q_accs[row * cols + ci];
                const unsigned long long cnt = v >> 20, qs = v & 0xfffffu;
                if (row < 4 && v) {
                    atomicAdd(&qc[(AQC_QC_BASE_COUNT_A + row) * AQC_QC_COLS + i], cnt);
                    atomicAdd(&qc[(AQC_QC_BASE_QUAL_A + row) * AQC_QC_COLS + i], qs);
                }
                tn += cnt; tq += qs;
            }
            if (tn) {
                atomicAdd(&qc[AQC_QC_TOTAL_NUM * AQC_QC_COLS + i], tn);
                atomicAdd(&qc[AQC_QC_TOTAL_QUAL * AQC_QC_COLS + i], tq);
            }
            const unsigned int dv = q_accs[5 * cols + ci];
            if (dv) atomicAdd(&qc[AQC_QC_DISCONTINUITY * AQC_QC_COLS + i], (unsigned long long)dv);
            if (q_gch[i]) atomicAdd(&qc[AQC_QC_GC_HIST * AQC_QC_COLS + i], (unsigned long long)q_gch[i]);
        }
        if (threadIdx.x < 2 && q_scal[threadIdx.x]) atomicAdd(&qc[AQC_QC_SCALARS * AQC_QC_COLS + threadIdx.x], q_scal[threadIdx.x]);
    }
    KPROF_FLUSH;
}

// dense_count[XCD 0 copy][idx] += sum over rounds of partial[round][idx]
// A workgroup owns 256 adjacent entries; its four waves take every fourth round each, a lane adds four entries
// (one 8-byte load per round, 512 contiguous bytes per wave) and the four partial sums meet in LDS.
__global__ __launch_bounds__(KRED_BLOCK) void kmer_reduce_kernel(const uint16_t* __restrict__ partial, uint32_t n_rounds,
                                                                 KmerTable kt, int kmer_len) {
    __shared__ unsigned int part[4][KRED_ENTRIES];
    const int quad = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const uint32_t e0 = blockIdx.x * KRED_ENTRIES + 4 * quad;
    unsigned int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 8
    for (uint32_t r = grp; r < n_rounds; r += 4) {
        const uint2 v = *reinterpret_cast<const uint2*>(partial + (size_t)r * DENSE_ENTRIES + e0);
        s0 += v.x & 0xffffu; s1 += v.x >> 16; s2 += v.y & 0xffffu; s3 += v.y >> 16;
    }
    part[grp][4 * quad + 0] = s0; part[grp][4 * quad + 1] = s1; part[grp][4 * quad + 2] = s2; part[grp][4 * quad + 3] = s3;
    __syncthreads();
    const int i = threadIdx.x;
    const uint32_t idx = blockIdx.x * KRED_ENTRIES + i;
    kt.dense_count[idx] += part[0][i] + part[1][i] + part[2][i] + part[3][i];
    // does every entry of this workgroup have a first-seen time by now (in any XCD's copy)?
    bool seen = idx >= (1u << (2 * kmer_len));
    if (!kt.complete[blockIdx.x]) {
        for (int x = 0; x < N_XCD && !seen; ++x) seen = kt.dense_first[(size_t)x * DENSE_ENTRIES + idx] != ~0ull;
        const int all = __syncthreads_and(seen ? 1 : 0);
        if (threadIdx.x == 0 && all) kt.complete[blockIdx.x] = 1u;
    }
}

// compact the occupied k-mer slots into dense arrays
__global__ void kmer_compact_kernel(KmerTable kt, unsigned long long* keys, unsigned long long* counts,
                                    unsigned long long* order, unsigned long long cap, unsigned long long* n_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > kt.mask) return;
    const unsigned long long key = kt.keys[i];
    if (key == 0) return;
    const unsigned long long w = atomicAdd(n_out, 1ull);
    if (w < cap) { keys[w] = key; counts[w] = kt.counts[i]; order[w] = kt.order[i]; }
}

// ... and the dense A/C/G/T table: k-mer X is in the dictionary iff X or its reverse complement was scanned;
// its insertion rank is min(2 * first(X), 2 * first(rc X) + 1) (qualitycontrol.py:116-122)
__global__ void kmer_compact_dense_kernel(KmerTable kt, int k, unsigned long long* keys, unsigned long long* counts,
                                          unsigned long long* order, unsigned long long cap, unsigned long long* n_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << (2 * k))) return;
    const unsigned long long never = ~0ull;
    unsigned long long f = never, fr = never, cnt = 0;
    const uint32_t ir = dense_rc(i, k);
    for (int x = 0; x < N_XCD; ++x) {
        const unsigned long long a = kt.dense_first[(size_t)x * DENSE_ENTRIES + i], b = kt.dense_first[(size_t)x * DENSE_ENTRIES + ir];
        f = a < f ? a : f;
        fr = b < fr ? b : fr;
        cnt += kt.dense_count[(size_t)x * DENSE_ENTRIES + i];
    }
    if (f == never && fr == never) return;
    unsigned long long ord = never;
    if (f != never) ord = 2 * f;
    if (fr != never && 2 * fr + 1 < ord) ord = 2 * fr + 1;
    unsigned long long key = 0;
    for (int j = 0; j < k; ++j) {
        const uint32_t code = (i >> (2 * j)) & 3u;
        key |= (unsigned long long)((CODE_TO_BASE >> (8 * code)) & 0xffu) << (8 * j);
    }
    const unsigned long long w = atomicAdd(n_out, 1ull);
    if (w < cap) { keys[w] = key; counts[w] = cnt; order[w] = ord; }
}

// ------------------------------------------------------------------------------------------------
// function seams: the same device functions, one result per input record
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void overlap_seam_kernel(DevBatch b, int32_t* off, int32_t* ol, int32_t* df) {
    __shared__ uint8_t lds[WPB][2][LSTR];
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const uint64_t rec = (uint64_t)blockIdx.x * WPB + wave;
    if (rec >= b.n) return;
    const int L1 = (int)(b.len1[rec] & LEN_MASK), L2 = (int)(b.len2[rec] & LEN_MASK);
    stage(lds[wave][0], b.seq1 + b.off1[rec], L1);
    for (int i = lane; i < L2; i += WAVE) lds[wave][1][i] = comp_or_n(b.seq2[b.off2[rec] + i]);
    __builtin_amdgcn_wave_barrier();
    int o, l, d;
    overlap_hm_wave(lds[wave][0], L1, lds[wave][1], L2, o, l, d);
    if (lane == 0) { off[rec] = o; ol[rec] = l; df[rec] = d; }
}

__global__ __launch_bounds__(BLOCK) void read_stats_seam_kernel(DevBatch b, int max_poly, int mismatch, int qual,
                                                                uint8_t* polyx, int32_t* lowq, int32_t* ncount) {
    __shared__ uint8_t lds[WPB][2][LSTR];
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const uint64_t rec = (uint64_t)blockIdx.x * WPB + wave;
    if (rec >= b.n) return;
    const int L1 = (int)(b.len1[rec] & LEN_MASK);
    stage(lds[wave][0], b.seq1 + b.off1[rec], L1);
    stage(lds[wave][1], b.qual1 + (b.qoff1 ? b.qoff1[rec] : b.off1[rec]), L1);
    __builtin_amdgcn_wave_barrier();
    const int p = has_polyx_wave(lds[wave][0], L1, max_poly, mismatch);
    const int lq = low_quality_wave(lds[wave][1], L1, qual);
    const int nn = n_number_wave(lds[wave][0], L1);
    if (lane == 0) { polyx[rec] = (uint8_t)p; lowq[rec] = lq; ncount[rec] = nn; }
}

__global__ void edit_distance_seam_kernel(DevBatch b, int32_t* dist, int* status) {
    const uint64_t rec = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rec >= b.n) return;
    const int la = (int)(b.len1[rec] & LEN_MASK), lb = (int)(b.len2[rec] & LEN_MASK);
    const uint8_t* a = b.seq1 + b.off1[rec];
    const uint8_t* c = b.seq2 + b.off2[rec];
    // the bit-vector form needs the pattern in one 64-bit word; Levenshtein is symmetric
    auto fa = [&](int i) { return a[i]; };
    auto fc = [&](int i) { return c[i]; };
    if (la <= 64) dist[rec] = edit_distance_lane(fa, la, fc, lb);
    else if (lb <= 64) dist[rec] = edit_distance_lane(fc, lb, fa, la);
    else { dist[rec] = -1; atomicCAS(status, 0, AQC_ERR_UNSUPPORTED); }
}

// the caller's 64-bit byte offsets (struct aqc_batch) -> the 32-bit device form
// an uploaded batch whose quality strings have lengths of their own (aqc_batch::qlen*): mark the mates that differ
__global__ void mark_irregular_kernel(uint32_t* __restrict__ len, const uint32_t* __restrict__ qlen, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && qlen[i] != len[i]) len[i] |= LEN_IRR;
}

// aqc_fetch_quality_views: the slice of the quality string that goes with the final read of every record
__global__ void quality_views_kernel(DevBatch b, const aqc_result* __restrict__ results, int mate, uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t lw = mate == 0 ? b.len1[i] : b.len2[i];
    if ((lw & LEN_IRR) && b.qlen1) { out[i] = mate == 0 ? b.qview1[i] : b.qview2[i]; return; }
    const aqc_result r = results[i];
    out[i] = mate == 0 ? ((uint32_t)r.start1 | ((uint32_t)r.len1 << 16)) : ((uint32_t)r.start2 | ((uint32_t)r.len2 << 16));
}

__global__ void narrow_offsets_kernel(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}

// ---- libed.so-compatible seams (editdistance/_editdistance.h:16,23): one call = one tiny launch ----------------
// Levenshtein distance of two strings of any length: one workgroup, the DP row of util.py:72-83 in global scratch
// (`row`, lb + 1 ints), anti-dependencies resolved by walking the row in order on lane 0 — a compatibility seam, not a
// hot path (the hot path's Levenshtein is edit_distance_lane above).
__global__ void edit_distance_any_kernel(const uint8_t* a, int la, const uint8_t* b, int lb, int* row, int* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (la <= 64 || lb <= 64) {
        auto fa = [&](int i) { return a[i]; };
        auto fb = [&](int i) { return b[i]; };
        *out = la <= 64 ? edit_distance_lane(fa, la, fb, lb) : edit_distance_lane(fb, lb, fa, la);
        return;
    }
    for (int j = 0; j <= lb; ++j) row[j] = j;
    for (int i = 1; i <= la; ++i) {
        int diag = row[0];
        row[0] = i;
        const uint8_t ca = a[i - 1];
        for (int j = 1; j <= lb; ++j) {
            const int up = row[j];
            const int v = min(min(up + 1, row[j - 1] + 1), diag + (ca != b[j - 1] ? 1 : 0));
            diag = up;
            row[j] = v;
        }
    }
    *out = row[lb];
}

// seek_overlap(r1, len1, reverse_r2, len2, limit_distance, complete_compare_require, overlap_require) with the semantics of
// the LIVE scan util.overlap_hm (util.py:158-212), parameters made explicit: per offset the loop counts mismatches and
// breaks at the limit-th one if it falls at a column < complete_compare_require; the offset is accepted iff diff < limit,
// or the loop ran to the end (no break) and its last column L-1 is > complete_compare_require.  In closed form:
// tot < limit, or (mismatches among the first `ccr` columns < limit and L - 1 > ccr).  One candidate per lane, 64 per
// step, in the reference's order (forward offsets, then reverse).  Result (offset << 8) + diff, 0x7FFFFFFF for none
// (_editdistance.cpp:150,177,181).
__global__ void seek_overlap_kernel(const uint8_t* r1, int len1, const uint8_t* rr2, int len2, int limit, int ccr, int ovr, int* out) {
    const int lane = lane_id();
    const int nf = len1 > ovr ? len1 - ovr : 0;
    const int nr = len2 > ovr ? len2 - ovr : 0;
    for (int base = 0; base < nf + nr; base += WAVE) {
        const int c = base + lane;
        bool ok = false;
        int tot = 0;
        if (c < nf + nr) {
            const int p1 = c < nf ? c : 0, p2 = c < nf ? 0 : c - nf;
            const int L = c < nf ? min(len1 - c, len2) : min(len1, len2 - p2);
            int early = 0;
            for (int i = 0; i < L; ++i) {
                const int mm = r1[p1 + i] != rr2[p2 + i] ? 1 : 0;
                tot += mm;
                if (i < ccr) early += mm;
            }
            ok = tot < limit || (early < limit && L - 1 > ccr);
        }
        const unsigned long long b = __ballot(ok);
        if (b) {
            const int l = __ffsll((long long)b) - 1;
            const int cand = base + l;
            const int t = __shfl(tot, l, WAVE);
            if (lane == 0) *out = (int)((unsigned int)(cand < nf ? cand : -(cand - nf)) << 8) + t;
            return;
        }
    }
    if (lane == 0) *out = 0x7FFFFFFF;
}

}  // namespace aqc
