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

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WPB = BLOCK / WAVE;      // waves (= records in flight) per workgroup
constexpr int LSTR = 1024;             // LDS bytes per staged string (AQC_MAX_READ_LEN = 1000)

struct DevBatch {
    const uint8_t *seq1, *qual1, *seq2, *qual2;
    const uint64_t *off1, *qoff1, *off2, *qoff2;
    const uint32_t *len1, *len2;
    const int32_t *aux_lane, *aux_tile, *aux_x, *aux_y;
    const uint8_t* aux_ok;
    uint64_t n;
    uint64_t first_index;
};

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
};

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

// stage two strings of the same length at once: all loads are issued before the first LDS store, so a read of up
// to 256 bytes costs ONE memory round trip instead of one per 64 bytes and string
__device__ __forceinline__ void stage2(uint8_t* d0, const uint8_t* s0, uint8_t* d1, const uint8_t* s1, int len) {
    const int lane = lane_id();
    if (len <= 4 * WAVE) {
        uint8_t a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + WAVE * j;
            a[j] = i < len ? s0[i] : (uint8_t)0;
            b[j] = i < len ? s1[i] : (uint8_t)0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + WAVE * j;
            if (i < len) { d0[i] = a[j]; d1[i] = b[j]; }
        }
    } else {
        for (int i = lane; i < len; i += WAVE) { d0[i] = s0[i]; d1[i] = s1[i]; }
    }
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

__device__ inline void process_record_wave(const DevBatch& b, uint64_t rec, const aqc_config& cfg, const DevCircles& circ,
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
        const int L1 = (int)b.len1[rec];
        const int L2 = paired ? (int)b.len2[rec] : 0;
        if (L1 > AQC_MAX_READ_LEN || L2 > AQC_MAX_READ_LEN) {
            if (lane == 0) atomicCAS(st.status, 0, AQC_ERR_READ_TOO_LONG);
            return;
        }
        stage(s1, b.seq1 + b.off1[rec], L1);
        stage(q1, b.qual1 + (b.qoff1 ? b.qoff1[rec] : b.off1[rec]), L1);
        if (paired) {
            stage(s2, b.seq2 + b.off2[rec], L2);
            stage(q2, b.qual2 + (b.qoff2 ? b.qoff2[rec] : b.off2[rec]), L2);
            for (int i = lane; i < L2; i += WAVE) c2[i] = comp_or_n(b.seq2[b.off2[rec] + i]);
        }
        // (wave-private LDS region: no barrier needed, the compiler orders LDS ops of one wave)
        __builtin_amdgcn_wave_barrier();

        int a1 = 0, len1 = L1, a2 = 0, len2 = L2;     // current views: s1[a1 .. a1+len1), s2[a2 .. a2+len2)
        int flag = -1;
        int offset = 0, ovl = 0, dist = 0, n_edits = 0;
        uint8_t bcode = 0;
        aqc_edit edits[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        // counters this record contributes (wave-uniform values, committed by lane 0)
        int c_adapter_base = 0, c_adapter_read = 0, c_overlapped = 0, c_corrected = 0, c_masked = 0, c_skipped = 0;
        int c_read_corrected = 0, ovl0 = -1, dist_final = -1;
        int em[3] = {-1, -1, -1};

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
                        const int cut = clean_barcode_tail_wave(s1 + a1, len1, s2 + a2, len2, rs1, b1 + vl, rs2, b2 + vl);
                        len1 -= cut; len2 -= cut;
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
            if (len1 < 5) flag = AQC_BADTRIM1;
            else if (paired) {
                end = cfg.trim_tail2 > 0 ? max(len2 - cfg.trim_tail2, 0) : len2;
                stt = min(cfg.trim_front2, len2);
                nl = max(end - stt, 0);
                a2 += stt; len2 = nl;
                if (len2 < 5) flag = AQC_BADTRIM2;
            }
        }
        // ---- bubble (preprocesser.py:469-473)
        if (flag < 0 && cfg.debubble && b.aux_ok && b.aux_ok[rec]) {
            if (in_bubble_wave(b.aux_lane[rec], b.aux_tile[rec], b.aux_x[rec], b.aux_y[rec], circ)) flag = AQC_BADBBL;
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
            if (low_quality_wave(q1 + a1, len1, cfg.qualified_quality_phred) > cfg.unqualified_base_limit) flag = AQC_BADLQC;
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
                c_adapter_base = 2 * (-offset); c_adapter_read = 1;
                if (len1 < cfg.seq_len_req) { flag = AQC_BADLEN; offset = 0; ovl = 0; dist = 0; }   // record carries no overlap
                else overlap_hm_wave(s1 + a1, len1, c2 + a2, len2, offset, ovl, dist);
            }
            if (flag < 0) {
                dist_final = dist;
                if (dist > 3) flag = AQC_BADDIFF;
                else if (ovl > 30) {
                    c_overlapped = 1;
                    if (dist > 0) {
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
                                        else em[handled] = i0 * 4 + i1;          // err[comp(b1)][comp(b2)]
                                    }
                                    if (!cfg.no_correction) {
                                        const uint8_t cA = comp_strict(bA);
                                        if (cA == 0) bad_alpha = true;
                                        edits[n_edits] = aqc_edit{(uint16_t)oo, AQC_EDIT_FIX_R2, cA, (uint8_t)qa};
                                        n_edits++; c_corrected++; fixed = true;
                                    }
                                } else if (qb - 33 >= 30 && qa - 33 <= 14) {
                                    if (bA != 'N' && bB != 'N') {
                                        const int i0 = base_idx(bB), i1 = base_idx(bA);
                                        if (i0 < 0 || i1 < 0) bad_alpha = true;
                                        else em[handled] = i0 * 4 + i1;          // err[b2][b1]
                                    }
                                    if (!cfg.no_correction) {
                                        edits[n_edits] = aqc_edit{(uint16_t)oo, AQC_EDIT_FIX_R1, bB, (uint8_t)qb};
                                        n_edits++; c_corrected++; fixed = true;
                                    }
                                }
                                if (!fixed) {
                                    if (cfg.mask_mismatch) {
                                        edits[n_edits] = aqc_edit{(uint16_t)oo, AQC_EDIT_MASK, 0, (uint8_t)'!'};
                                        n_edits++; c_masked++;
                                    } else c_skipped++;
                                }
                                handled++;
                            }
                            // util.complement raises on every visited r2 byte outside COMP (preprocesser.py:565)
                            const unsigned long long visited = (handled >= dist) ? ((last == 63) ? ~0ull : ((2ull << last) - 1)) : ~0ull;
                            if (inval & visited) bad_alpha = true;
                        }
                        if (bad_alpha && lane == 0) atomicCAS(st.status, 0, AQC_ERR_ALPHABET);
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
            r.edits[0] = edits[0]; r.edits[1] = edits[1]; r.edits[2] = edits[2];
            r.barcode = bcode;
            results[rec] = r;
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
__device__ inline void flush_block_acc(BlockAcc& acc, const DevStats& st) {
    for (int i = threadIdx.x; i < AQC_N_COUNTERS; i += blockDim.x)
        if (acc.counters[i]) atomicAdd(&st.counters[i], acc.counters[i]);
    for (int i = threadIdx.x; i < AQC_QC_COLS; i += blockDim.x) {
        if (acc.ovl_hist[i]) atomicAdd(&st.ovl_hist[i], (unsigned long long)acc.ovl_hist[i]);
        if (acc.dist_hist[i]) atomicAdd(&st.dist_hist[i], (unsigned long long)acc.dist_hist[i]);
    }
}

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
    for (uint64_t rec = (uint64_t)blockIdx.x * WPB + wave; rec < b.n; rec += nwaves)
        process_record_wave(b, rec, cfg, circ, w, results, acc, st, rec < accum_limit);
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
        process_record_wave(b, rec, cfg, circ, w, results, acc, st, rec < accum_limit);
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
};
constexpr int N_XCD = 8;
constexpr uint32_t DENSE_ENTRIES = 1u << 16;   // 4^8

// id of the XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (N_XCD - 1); }

// reverse complement of a dense k-mer index: complement flips the code's high bit, the order of bases reverses
__device__ __host__ inline uint32_t dense_rc(uint32_t idx, int k) {
    const uint32_t m = (1u << k) - 1u;
    uint32_t b0 = idx & m, b1 = (idx >> k) & m, r0 = 0, r1 = 0;
    for (int j = 0; j < k; ++j) {
        r0 |= ((b0 >> j) & 1u) << (k - 1 - j);
        r1 |= ((b1 >> j) & 1u) << (k - 1 - j);
    }
    return ((~r1 & m) << k) | r0;
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

constexpr int QC_LDS_ROWS = 11;   // TOTAL_NUM .. DISCONTINUITY (gc histogram kept separately)

__global__ __launch_bounds__(BLOCK) void qc_stat_kernel(DevBatch b, int mate, uint64_t first, uint64_t count, int post,
                                                        const aqc_result* __restrict__ results, int kmer_len,
                                                        unsigned long long* __restrict__ qc /* [QC_ROWS*QC_COLS] */,
                                                        int* status, int cols) {
    // dynamic LDS, sized by the longest read of the batch (cols = multiple of 64 <= 1024) so that short reads get
    // many resident workgroups: [QC_LDS_ROWS][cols] + gc histogram [cols] u32, scalars, 2 staging strings per wave
    extern __shared__ __attribute__((aligned(16))) unsigned int qc_smem[];
    unsigned int* const accs = qc_smem;                               // accs[row * cols + i]
    unsigned int* const gch = qc_smem + QC_LDS_ROWS * cols;
    unsigned long long* const scal = reinterpret_cast<unsigned long long*>(gch + cols);
    uint8_t* const strings = reinterpret_cast<uint8_t*>(scal + 2);
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    for (int i = threadIdx.x; i < (QC_LDS_ROWS + 1) * cols; i += BLOCK) qc_smem[i] = 0;
    if (threadIdx.x < 2) scal[threadIdx.x] = 0;
    __syncthreads();
    uint8_t* s = strings + (size_t)(2 * wave) * (cols + 16);
    uint8_t* q = s + (cols + 16);
    const uint64_t nwaves = (uint64_t)gridDim.x * WPB;
    for (uint64_t k = (uint64_t)blockIdx.x * WPB + wave; k < count; k += nwaves) {
        const uint64_t rec = first + k;
        int st = 0, len;
        const uint8_t *gs, *gq;
        if (mate == 0) {
            len = (int)b.len1[rec];
            gs = b.seq1 + b.off1[rec];
            gq = b.qual1 + (b.qoff1 ? b.qoff1[rec] : b.off1[rec]);
        } else {
            len = (int)b.len2[rec];
            gs = b.seq2 + b.off2[rec];
            gq = b.qual2 + (b.qoff2 ? b.qoff2[rec] : b.off2[rec]);
        }
        aqc_result r;
        if (post) {
            r = results[rec];
            if (r.flag != AQC_GOOD) continue;                 // only good records reach :624-627
            st = mate == 0 ? r.start1 : r.start2;
            len = mate == 0 ? r.len1 : r.len2;
        }
        if (len > AQC_MAX_READ_LEN || len > cols) { if (lane == 0) atomicCAS(status, 0, AQC_ERR_READ_TOO_LONG); continue; }
        if (len < 5) { if (lane == 0 && len > 0) atomicCAS(status, 0, AQC_ERR_ARG); continue; }   // IndexError upstream (:106-107)
        stage2(s, gs + st, q, gq + st, len);
        __builtin_amdgcn_wave_barrier();
        if (post && lane == 0) {
            // apply the <= 3 edits of the correction walk to the staged copy
#pragma unroll
            for (int e = 0; e < 3; e++) {
                if (e >= r.n_edits) break;
                const aqc_edit ed = r.edits[e];
                const int p1 = (int)r.len1 - (int)r.overlap_len + ed.o, p2 = (int)r.len2 - 1 - ed.o;
                if (ed.kind == AQC_EDIT_MASK) q[mate == 0 ? p1 : p2] = '!';
                else if (ed.kind == AQC_EDIT_FIX_R1 && mate == 0) { s[p1] = ed.base; q[p1] = ed.qual; }
                else if (ed.kind == AQC_EDIT_FIX_R2 && mate == 1) { s[p2] = ed.base; q[p2] = ed.qual; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        int gc = 0;
#if defined(AQC_ABLATE) && AQC_ABLATE == 21
        if (false)
#endif
        for (int i0 = 0; i0 < len; i0 += WAVE) {
            const int i = i0 + lane;
            const bool in = i < len;
            if (in) {
                const int qn = (int)q[i] - 33;
                const uint8_t c = s[i];
                atomicAdd(&accs[AQC_QC_TOTAL_NUM * cols + i], 1u);
                atomicAdd(&accs[AQC_QC_TOTAL_QUAL * cols + i], (unsigned int)qn);
                const int bi = base_idx(c);
                if (bi >= 0) {
                    atomicAdd(&accs[(AQC_QC_BASE_COUNT_A + bi) * cols + i], 1u);
                    atomicAdd(&accs[(AQC_QC_BASE_QUAL_A + bi) * cols + i], (unsigned int)qn);
                }
                // discontinuity over the 5-wide window clamped to the read (qualitycontrol.py:97-109)
                int left = i - 2, right = i + 3;
                if (left < 0) { left = 0; right = 5; }
                else if (right >= len) { right = len; left = len - 5; }
                int d = 0;
                for (int j = left; j < right - 1; j++) d += s[j] != s[j + 1];
                if (d) atomicAdd(&accs[AQC_QC_DISCONTINUITY * cols + i], (unsigned int)d);
            }
            gc += __popcll(__ballot(in && (s[i] == 'G' || s[i] == 'C')));
        }
        if (lane == 0) {
            atomicAdd(&gch[gc], 1u);
            atomicAdd(&scal[1], 1ull);
            if (len > kmer_len) atomicAdd(&scal[0], (unsigned long long)(len - kmer_len));
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < QC_LDS_ROWS * cols; i += BLOCK) {
        const unsigned int v = accs[i];
        if (v) atomicAdd(&qc[(i / cols) * AQC_QC_COLS + (i % cols)], (unsigned long long)v);
    }
    for (int i = threadIdx.x; i < cols; i += BLOCK)
        if (gch[i]) atomicAdd(&qc[AQC_QC_GC_HIST * AQC_QC_COLS + i], (unsigned long long)gch[i]);
    if (threadIdx.x < 2 && scal[threadIdx.x]) atomicAdd(&qc[AQC_QC_SCALARS * AQC_QC_COLS + threadIdx.x], scal[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// k-mer dictionary of statRead (qualitycontrol.py:113-122), counting part.
// Global atomics top out near 3e10 /s on this chip — 28 M k-mers of a 200 k-read sample would cost ~0.9 ms —
// so the 4^k counters live in LDS: one 1024-thread workgroup per CU keeps a private table of 65536 u16 counters
// (128 KiB) and works in ROUNDS of at most 65535 k-mers (no counter can overflow), then streams the table to
// its own slice of `partial` with plain coalesced stores; kmer_reduce_kernel adds the slices up.  No global
// atomic is issued for a pure A/C/G/T k-mer except the (rare, load-guarded) first-seen minimum.
// Per 64 positions three ballots give the two code bit planes and the "is A/C/G/T" plane; lane i shifts its
// k-mer out of them.  K-mers containing anything else go to the open-addressing table.
// ------------------------------------------------------------------------------------------------
constexpr int KMER_BLOCK = 1024;
constexpr int KMER_WPB = KMER_BLOCK / WAVE;

__global__ __launch_bounds__(KMER_BLOCK) void kmer_count_kernel(DevBatch b, int mate, uint64_t first, uint64_t count, int post,
                                                                const aqc_result* __restrict__ results, int kmer_len,
                                                                KmerTable kt, unsigned long long order_base,
                                                                uint16_t* __restrict__ partial, uint32_t reads_per_round,
                                                                uint32_t n_rounds, int* status) {
    extern __shared__ __attribute__((aligned(16))) unsigned int ktab[];     // 32768 words = 65536 u16 counters
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    const unsigned long long km = (1ull << kmer_len) - 1ull;
    unsigned long long* const my_first = kt.dense_first + (size_t)xcc_id() * DENSE_ENTRIES;
    for (uint32_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        for (int i = threadIdx.x; i < (int)(DENSE_ENTRIES / 2); i += KMER_BLOCK) ktab[i] = 0;
        __syncthreads();
        const uint64_t r_lo = (uint64_t)round * reads_per_round;
        const uint64_t r_hi = min(r_lo + reads_per_round, count);
        for (uint64_t k = r_lo + wave; k < r_hi; k += KMER_WPB) {
            const uint64_t rec = first + k;
            int st = 0, len;
            const uint8_t* gs;
            if (mate == 0) { len = (int)b.len1[rec]; gs = b.seq1 + b.off1[rec]; }
            else { len = (int)b.len2[rec]; gs = b.seq2 + b.off2[rec]; }
            int e_pos[3] = {-1, -1, -1};
            uint8_t e_base[3] = {0, 0, 0};
            if (post) {
#if defined(AQC_ABLATE) && AQC_ABLATE == 33
                aqc_result r; r.flag = 0; r.start1 = r.start2 = 0; r.len1 = r.len2 = (uint16_t)len; r.n_edits = 0; r.overlap_len = 0;
#else
                const aqc_result r = results[rec];
#endif
                if (r.flag != AQC_GOOD) continue;                 // only good records reach preprocesser.py:624-627
                st = mate == 0 ? r.start1 : r.start2;
                len = mate == 0 ? r.len1 : r.len2;
                // base corrections of the walk that touch this mate (final-read coordinates)
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    if (e < r.n_edits) {
                        const aqc_edit ed = r.edits[e];
                        if (ed.kind == AQC_EDIT_FIX_R1 && mate == 0) { e_pos[e] = (int)r.len1 - (int)r.overlap_len + ed.o; e_base[e] = ed.base; }
                        if (ed.kind == AQC_EDIT_FIX_R2 && mate == 1) { e_pos[e] = (int)r.len2 - 1 - ed.o; e_base[e] = ed.base; }
                    }
                }
            }
            if (len > AQC_MAX_READ_LEN || len < 5) continue;       // (reported by qc_stat_kernel)
            const int nk = len - kmer_len;
            if (nk <= 0) continue;
            const uint8_t* src = gs + st;
            auto base_at = [&](int x) -> uint8_t {
                uint8_t c = x < len ? src[x] : (uint8_t)0;
                if (x == e_pos[0]) c = e_base[0];
                if (x == e_pos[1]) c = e_base[1];
                if (x == e_pos[2]) c = e_base[2];
                return c;
            };
            const unsigned long long t0 = (order_base + k) * (unsigned long long)AQC_QC_COLS;
            // 256 positions per pass: the four byte loads, the four LDS adds and the four first-seen probes of a
            // pass are each issued back to back, so a pass costs two memory round trips, not eight
            for (int base0 = 0; base0 < nk; base0 += 4 * WAVE) {
                unsigned long long m0[5], m1[5], mv[5];
                uint8_t cb[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) cb[j] = base_at(base0 + WAVE * j + lane);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    m0[j] = __ballot((cb[j] >> 1) & 1); m1[j] = __ballot((cb[j] >> 2) & 1);
                    mv[j] = __ballot(cb[j] == 'A' || cb[j] == 'C' || cb[j] == 'G' || cb[j] == 'T');
                }
                uint32_t idx[4];
                bool dense[4], exotic[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = base0 + WAVE * j + lane;
                    const unsigned long long p0 = ((m0[j] >> lane) | (lane ? m0[j + 1] << (64 - lane) : 0ull)) & km;
                    const unsigned long long p1 = ((m1[j] >> lane) | (lane ? m1[j + 1] << (64 - lane) : 0ull)) & km;
                    const unsigned long long pv = ((mv[j] >> lane) | (lane ? mv[j + 1] << (64 - lane) : 0ull)) & km;
                    idx[j] = (uint32_t)((p1 << kmer_len) | p0);
                    dense[j] = i < nk && pv == km;
                    exotic[j] = i < nk && pv != km;
#if !(defined(AQC_ABLATE) && AQC_ABLATE == 31)
                    if (dense[j]) atomicAdd(&ktab[idx[j] >> 1], 1u << (16 * (idx[j] & 1)));
#endif
                }
                unsigned long long seen[4];
#if defined(AQC_ABLATE) && AQC_ABLATE == 32
#pragma unroll
                for (int j = 0; j < 4; ++j) seen[j] = 0ull;
#else
#pragma unroll
                for (int j = 0; j < 4; ++j) seen[j] = dense[j] ? my_first[idx[j]] : 0ull;
#endif
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long t = t0 + (unsigned long long)(base0 + WAVE * j + lane);
                    if (dense[j] && seen[j] > t) __hip_atomic_fetch_min(&my_first[idx[j]], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!__ballot(exotic[j])) continue;
                    if (exotic[j]) {
                        const int i = base0 + WAVE * j + lane;
                        unsigned long long key = 0, rkey = 0;
                        for (int q = 0; q < kmer_len; q++) {
                            key |= (unsigned long long)base_at(i + q) << (8 * q);
                            rkey |= (unsigned long long)comp_or_n(base_at(i + kmer_len - 1 - q)) << (8 * q);
                        }
                        const long long h = kmer_slot(kt, key);
                        const long long hr = kmer_slot(kt, rkey);
                        if (h < 0 || hr < 0) atomicCAS(status, 0, AQC_ERR_UNSUPPORTED);
                        else {
                            atomicAdd(&kt.counts[h], 1ull);
                            atomicMin(&kt.order[h], 2 * (t0 + i));
                            atomicMin(&kt.order[hr], 2 * (t0 + i) + 1);
                        }
                    }
                }
            }
        }
        __syncthreads();
        uint4* dst = reinterpret_cast<uint4*>(partial + (size_t)round * DENSE_ENTRIES);
        const uint4* srcv = reinterpret_cast<const uint4*>(ktab);
        for (int i = threadIdx.x; i < (int)(DENSE_ENTRIES * 2 / 16); i += KMER_BLOCK) dst[i] = srcv[i];
        __syncthreads();
    }
}

// dense_count[XCD 0 copy][idx] += sum over rounds of partial[round][idx]
__global__ void kmer_reduce_kernel(const uint16_t* __restrict__ partial, uint32_t n_rounds, unsigned int* __restrict__ dense_count) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= DENSE_ENTRIES) return;
    unsigned int sum = 0;
    for (uint32_t r = 0; r < n_rounds; ++r) sum += partial[(size_t)r * DENSE_ENTRIES + idx];
    dense_count[idx] += sum;
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
        const uint32_t code = ((i >> j) & 1u) | (((i >> (k + j)) & 1u) << 1);
        key |= (unsigned long long)((0x47544341u >> (8 * code)) & 0xffu) << (8 * j);      // code -> A C T G
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
    const int L1 = (int)b.len1[rec], L2 = (int)b.len2[rec];
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
    const int L1 = (int)b.len1[rec];
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
    const int la = (int)b.len1[rec], lb = (int)b.len2[rec];
    const uint8_t* a = b.seq1 + b.off1[rec];
    const uint8_t* c = b.seq2 + b.off2[rec];
    // the bit-vector form needs the pattern in one 64-bit word; Levenshtein is symmetric
    auto fa = [&](int i) { return a[i]; };
    auto fc = [&](int i) { return c[i]; };
    if (la <= 64) dist[rec] = edit_distance_lane(fa, la, fc, lb);
    else if (lb <= 64) dist[rec] = edit_distance_lane(fc, lb, fa, la);
    else { dist[rec] = -1; atomicCAS(status, 0, AQC_ERR_UNSUPPORTED); }
}

}  // namespace aqc
