// aqc_fast.hpp — generation 2 of the hot kernel: "one LANE per read pair".
//
// Why: the wave-per-record kernel (aqc_kernels.hpp) spends ~1500 wave-instructions per pair, which
// caps it at ~1 % of the HBM roofline.  To stream pairs at a useful fraction of 8 TB/s the whole
// pipeline must cost on the order of 100 wave-instructions per pair (256 CUs x 4 SIMDs x ~1.1 G
// wave-instr/s / 5 G pairs/s), i.e. every lane has to do useful work all the time and byte
// compares have to become word compares.  Design:
//
//   phase 1 (cooperative, coalesced)   the wave owns 64 consecutive pairs.  Each lane loads one
//       16-byte chunk of one string per step (global_load_dwordx4: 16 lanes cover a 160-byte read,
//       4 reads per instruction) and converts it with SWAR + v_dot4/v_perm into
//         lo  : 2 bits per base  ((c >> 1) & 3: A=0 C=1 T=2 G=3; N shares 3)            32 bits/chunk
//         e   : 1 bit per base (odd bit of the 2-bit field) set for 'N'                32 bits/chunk
//       Read 2 is stored complemented and reversed, so that reverse_r2 (util.py:161) is a forward
//       2-bit stream.  Low-quality counts of read 1 are reduced per chunk.  The planes go to LDS.
//   phase 2 (lane per pair)            every lane pulls the planes of ITS pair into registers,
//       normalises them (trim offsets, reverse-complement alignment) and runs the pipeline of
//       preprocesser.py:455-617 on 32-bit words:
//         * overlap scan: one diagonal = v_alignbit + v_xor + v_bcnt on a 16-base prefix window;
//           >= 5 differing bits imply >= 3 mismatching bases, which util.py:180-183 can never accept;
//         * the rare survivors are verified exactly over the full diagonal (lo and e planes);
//         * the correction walk reads the <= 3 mismatch positions off the same words.
//   exactness: bytes outside {A,C,G,T,N}, reads longer than 16*NW, reads shorter than the 16-base
//       prefix, barcodes, and the one adapter-trim corner case that needs a second scan are not
//       handled here: the lane DEFERS its pair and the wave runs the fully general generation-1
//       pipeline (process_record_wave) for it afterwards, in the same launch.  Results are
//       bit-identical either way; only the speed differs.
//
// Input is the engine's canonical device layout (built by aqc_upload): records 16-byte aligned,
// offsets in 16-byte units, the unused bytes of a sequence's last chunk filled with 'A'.
#pragma once
#include "aqc_kernels.hpp"

namespace aqc {

struct FastBatch {
    const uint8_t *seq1, *qual1, *seq2, *qual2;
    const uint32_t *o1, *o2;       // record offsets / 16
    const uint32_t *len1, *len2;
    const int32_t *aux_lane, *aux_tile, *aux_x, *aux_y;
    const uint8_t* aux_ok;
    uint64_t n;
};

constexpr uint32_t ODD = 0xAAAAAAAAu;
constexpr int NONE_CAND = 0x7fffffff;

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// mask with the low 2*nb bits set, nb in 0..16
__device__ __forceinline__ uint32_t base_mask(int nb) { return nb >= 16 ? 0xffffffffu : ((1u << (2 * nb)) - 1u); }

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v = max(v, __shfl_xor(v, s, WAVE));
    return v;
}

// 16 sequence bytes -> lo plane (2 bits/base), e plane ('N' flag on the odd bit), bad != 0 iff a byte
// is outside {A,C,G,T,N}.  Expected byte by 3-bit index (c>>1)&7 via v_perm: A C T G - - - N.
__device__ __forceinline__ void pack_dword(uint32_t d, uint32_t& lo8, uint32_t& e8, uint32_t& bad) {
    const uint32_t h = d >> 1;
    lo8 = udot4(h & 0x03030303u, 0x40100401u, 0u);
    e8 = udot4((d >> 3) & 0x01010101u, 0x80200802u, 0u);
    bad |= d ^ __builtin_amdgcn_perm(0x4e000000u, 0x47544341u, h & 0x07070707u);
}

__device__ __forceinline__ void pack_chunk(const uint4 v, uint32_t& lo, uint32_t& e, uint32_t& bad) {
    uint32_t l0, l1, l2, l3, e0, e1, e2, e3;
    bad = 0;
    pack_dword(v.x, l0, e0, bad);
    pack_dword(v.y, l1, e1, bad);
    pack_dword(v.z, l2, e2, bad);
    pack_dword(v.w, l3, e3, bad);
    lo = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24);
    e = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
}

// reverse the order of the sixteen 2-bit fields of a word
__device__ __forceinline__ uint32_t rev2(uint32_t x) {
    const uint32_t y = __builtin_bitreverse32(x);
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}

// trim() of preprocesser.py:19-28 with python slice semantics -> (start, new length)
__device__ __forceinline__ void trim_view(int len, int front, int tail, int& st, int& nl) {
    const int end = tail > 0 ? max(len - tail, 0) : len;
    st = min(front, len);
    nl = max(end - st, 0);
}

template <int NW>
struct FastWaveLds {
    static constexpr int STRIDE = 4 * NW + 5;   // odd: conflict-free lane-strided access
    uint32_t planes[WAVE][STRIDE];
    uint32_t o1[WAVE], o2[WAVE], l1[WAVE], l2[WAVE];
    uint32_t lq[WAVE];
    uint32_t exo[WAVE];
    uint8_t stage[16 * NW + 16];
    uint8_t rs[2][64];
};

// mismatch word j of a diagonal: moving stream (lo/e at word index k+j, sub-word shift s) against the
// fixed stream's word j; returns one flag per base on the ODD bits, limited to the first nb bases.
// N pairs with N (both e) as equal, N against a base as different (byte equality on {A,C,G,T,N}).
__device__ __forceinline__ uint32_t mm_word(uint32_t mlo0, uint32_t mlo1, uint32_t me0, uint32_t me1, uint32_t s,
                                            uint32_t flo, uint32_t fe, int nb) {
    const uint32_t mlo = alignbit(mlo1, mlo0, s), me = alignbit(me1, me0, s);
    const uint32_t x = mlo ^ flo;
    const uint32_t ld = ((x << 1) | x) & ODD;
    const uint32_t eo = me | fe, ex = me ^ fe;
    const uint32_t mm = (eo & ex) | (~eo & ld);
    return mm & base_mask(nb);
}

#ifdef AQC_PROFILE
#define PROF_DECL unsigned long long prof_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#define PROF(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof_t[k] += now_ - prof_last; prof_last = now_; } while (0)
#define PROF_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 10; ++k_) atomicAdd(&st.counters[AQC_N_COUNTERS + k_], prof_t[k_]); } while (0)
#else
#define PROF_DECL
#define PROF(k)
#define PROF_FLUSH
#endif

template <int NW, bool PAIRED, int WPBT>
__global__ __launch_bounds__(WPBT * WAVE) void fast_filter_overlap_kernel(FastBatch fb, DevBatch raw, aqc_config cfg, DevCircles circ,
                                                                    aqc_result* __restrict__ results, DevStats st,
                                                                    uint64_t accum_limit) {
    using WL = FastWaveLds<NW>;
    constexpr int STRIDE = WL::STRIDE;
    __shared__ WL wls[WPBT];
    __shared__ BlockAcc acc;
#if defined(AQC_ABLATE) && AQC_ABLATE == 9   /* occupancy probe: one workgroup per CU */
    __shared__ uint32_t occ_pad[11000];
    if (threadIdx.x == 0 && fb.n == 1234567891234ull) occ_pad[blockIdx.x % 11000] = 1;
    if (fb.n == 1234567891235ull && occ_pad[threadIdx.x] == 77) return;
#endif
    static_assert(sizeof(uint32_t) * WAVE * STRIDE >= 5 * LSTR, "generation-1 staging must fit into the plane region");
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    for (int i = threadIdx.x; i < (int)(sizeof(BlockAcc) / 4); i += WPBT * WAVE) ((unsigned int*)&acc)[i] = 0;
    __syncthreads();
    WL& L = wls[wave];
    uint32_t* const my = L.planes[lane];
    const int thr4 = (cfg.qualified_quality_phred + 33) * 0x01010101;
    const bool do_trim = cfg.trim_front > 0 || cfg.trim_tail > 0;
    // longest run of identical bases any firing polyX window must contain (pigeonhole over the mismatches)
    const int need = cfg.poly_size_limit - cfg.allow_mismatch_in_poly;
    const int run_req = cfg.allow_mismatch_in_poly >= 0 ? (need + cfg.allow_mismatch_in_poly) / (cfg.allow_mismatch_in_poly + 1) : 0;

    PROF_DECL
    const uint64_t stride = (uint64_t)gridDim.x * WPBT * WAVE;
    const uint64_t base0 = ((uint64_t)blockIdx.x * WPBT + wave) * WAVE;
    // record descriptors of the NEXT batch are fetched one iteration ahead (no round trip at the top of the loop)
    uint32_t m_o1 = 0, m_l1 = 0, m_o2 = 0, m_l2 = 0;
    if (base0 + lane < fb.n) {
        m_o1 = fb.o1[base0 + lane]; m_l1 = fb.len1[base0 + lane];
        if (PAIRED) { m_o2 = fb.o2[base0 + lane]; m_l2 = fb.len2[base0 + lane]; }
    }
    for (uint64_t base = base0; base < fb.n; base += stride) {
        const uint64_t rec = base + lane;
        const bool valid = rec < fb.n;
        // ------------------------------------------------------------------ phase 1: load + pack
        L.o1[lane] = m_o1;
        L.l1[lane] = m_l1;
        if (PAIRED) { L.o2[lane] = m_o2; L.l2[lane] = m_l2; }
        L.lq[lane] = 0;
        L.exo[lane] = 0;
        {
            const uint64_t nrec = rec + stride;
            m_o1 = m_l1 = m_o2 = m_l2 = 0;
            if (nrec < fb.n) {
                m_o1 = fb.o1[nrec]; m_l1 = fb.len1[nrec];
                if (PAIRED) { m_o2 = fb.o2[nrec]; m_l2 = fb.len2[nrec]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // each pass: descriptors from LDS, then all NW 16-byte loads in flight at once, then the packing
        {
            uint4 v[NW];
            int sp[NW], len[NW];
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const int t = it * WAVE + lane;
                sp[it] = t / NW;
                len[it] = (int)L.l1[sp[it]];
                v[it] = *reinterpret_cast<const uint4*>(fb.seq1 + ((uint64_t)(L.o1[sp[it]] + (t - sp[it] * NW)) << 4));
            }
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const int c = it * WAVE + lane - sp[it] * NW;
                uint32_t lo, e, bad;
                pack_chunk(v[it], lo, e, bad);
                L.planes[sp[it]][c] = lo;
                L.planes[sp[it]][NW + c] = e;
                if (bad && c * 16 < len[it]) L.exo[sp[it]] = 1;
            }
        }
        if (PAIRED) {
            uint4 v[NW];
            int sp[NW], len[NW];
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const int t = it * WAVE + lane;
                sp[it] = t / NW;
                len[it] = (int)L.l2[sp[it]];
                v[it] = *reinterpret_cast<const uint4*>(fb.seq2 + ((uint64_t)(L.o2[sp[it]] + (t - sp[it] * NW)) << 4));
            }
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const int c = it * WAVE + lane - sp[it] * NW;
                uint32_t lo, e, bad;
                pack_chunk(v[it], lo, e, bad);
                // complement (A<->T, C<->G: flip the high bit of the field), keep N at code 3, then reverse the chunk
                lo = (lo ^ ODD) | e | (e >> 1);
                L.planes[sp[it]][2 * NW + (NW - 1 - c)] = rev2(lo);
                L.planes[sp[it]][3 * NW + (NW - 1 - c)] = __builtin_bitreverse32(e) << 1;
                if (bad && c * 16 < len[it]) L.exo[sp[it]] = 1;
            }
        }
        if (cfg.unqualified_base_limit > 0) {
            uint4 v[NW];
            int sp[NW], len[NW];
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const int t = it * WAVE + lane;
                sp[it] = t / NW;
                len[it] = (int)L.l1[sp[it]];
                v[it] = *reinterpret_cast<const uint4*>(fb.qual1 + ((uint64_t)(L.o1[sp[it]] + (t - sp[it] * NW)) << 4));
            }
#pragma unroll
            for (int it = 0; it < NW; ++it) {
                const int c = it * WAVE + lane - sp[it] * NW;
                int a = 0, nl = len[it];
                if (do_trim) trim_view(len[it], cfg.trim_front, cfg.trim_tail, a, nl);
                // byte < thr  <=>  high bit of ((byte | 0x80) - thr) clear   (bytes < 0x80, thr <= 0x7f)
                const uint32_t f0 = (~((v[it].x | 0x80808080u) - thr4) & 0x80808080u) >> 7;
                const uint32_t f1 = (~((v[it].y | 0x80808080u) - thr4) & 0x80808080u) >> 7;
                const uint32_t f2 = (~((v[it].z | 0x80808080u) - thr4) & 0x80808080u) >> 7;
                const uint32_t f3 = (~((v[it].w | 0x80808080u) - thr4) & 0x80808080u) >> 7;
                const uint32_t f16 = udot4(f0, 0x08040201u, 0u) | (udot4(f1, 0x08040201u, 0u) << 4) |
                                     (udot4(f2, 0x08040201u, 0u) << 8) | (udot4(f3, 0x08040201u, 0u) << 12);
                const int lo_b = min(max(a - 16 * c, 0), 16), hi_b = min(max(a + nl - 16 * c, 0), 16);
                const uint32_t m16 = ((1u << hi_b) - 1u) & ~((1u << lo_b) - 1u);
                const int cnt = __popc(f16 & m16);
                if (cnt) atomicAdd(&L.lq[sp[it]], (uint32_t)cnt);
                if (((v[it].x | v[it].y | v[it].z | v[it].w) & 0x80808080u) && c * 16 < len[it]) L.exo[sp[it]] = 1;   // non-ASCII quality byte
            }
        }
        __builtin_amdgcn_wave_barrier();
        PROF(0);

#if defined(AQC_ABLATE) && AQC_ABLATE == 1   /* phase 1 only */
        if (valid) results[rec].flag = (uint8_t)(L.planes[lane][0] + L.lq[lane] + L.exo[lane]);
        continue;
#endif
        // ------------------------------------------------------------------ phase 2: lane per pair
        const int L1 = (int)L.l1[lane];
        const int L2 = PAIRED ? (int)L.l2[lane] : 0;
        const bool accum = valid && rec < accum_limit;
        bool defer = valid && (L.exo[lane] != 0 || L1 > 16 * NW || L2 > 16 * NW || L1 == 0 || (PAIRED && L2 == 0));
        int a1 = 0, len1 = L1, a2 = 0, len2 = L2;
        int flag = -1;
        if (do_trim) {
            trim_view(L1, cfg.trim_front, cfg.trim_tail, a1, len1);
            if (len1 < 5) flag = AQC_BADTRIM1;
            else if (PAIRED) {
                trim_view(L2, cfg.trim_front2, cfg.trim_tail2, a2, len2);
                if (len2 < 5) flag = AQC_BADTRIM2;
            }
        }
        // ---- normalise the planes into registers: W1*[j] holds read1 bases 16j..16j+15, W2* the same for reverse_r2
        uint32_t W1lo[NW + 1], W1e[NW + 1], W2lo[NW + 1], W2e[NW + 1];
        {
            const int k0 = a1 >> 4;
            const uint32_t s = (uint32_t)(a1 & 15) * 2;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int i0 = k0 + j, i1 = k0 + j + 1;
                const uint32_t lo0 = i0 < NW ? my[i0] : 0u, lo1 = i1 < NW ? my[i1] : 0u;
                const uint32_t e0 = i0 < NW ? my[NW + i0] : 0u, e1 = i1 < NW ? my[NW + i1] : 0u;
                const uint32_t m = base_mask(min(max(len1 - 16 * j, 0), 16));
                W1lo[j] = alignbit(lo1, lo0, s) & m;
                W1e[j] = alignbit(e1, e0, s) & m;
            }
            W1lo[NW] = 0; W1e[NW] = 0;
        }
        if (PAIRED) {
            // reverse_r2[i] sits at stream position p0 + i of the reversed chunk sequence
            const int tl = a2 + len2 - 1;                       // last base of the current read 2
            const int p0 = 16 * (NW - 1 - (tl >> 4)) + 15 - (tl & 15);
            const int k0 = p0 >> 4;
            const uint32_t s = (uint32_t)(p0 & 15) * 2;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int i0 = k0 + j, i1 = k0 + j + 1;
                const bool ok0 = i0 >= 0 && i0 < NW, ok1 = i1 >= 0 && i1 < NW;
                const uint32_t lo0 = ok0 ? my[2 * NW + i0] : 0u, lo1 = ok1 ? my[2 * NW + i1] : 0u;
                const uint32_t e0 = ok0 ? my[3 * NW + i0] : 0u, e1 = ok1 ? my[3 * NW + i1] : 0u;
                const uint32_t m = base_mask(min(max(len2 - 16 * j, 0), 16));
                W2lo[j] = alignbit(lo1, lo0, s) & m;
                W2e[j] = alignbit(e1, e0, s) & m;
            }
            W2lo[NW] = 0; W2e[NW] = 0;
        } else {
#pragma unroll
            for (int j = 0; j <= NW; ++j) { W2lo[j] = 0; W2e[j] = 0; }
        }
        __builtin_amdgcn_wave_barrier();
        // keep the normalised streams in LDS as well: the verify / walk steps index them with per-lane offsets
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            my[j] = W1lo[j]; my[NW + j] = W1e[j]; my[2 * NW + j] = W2lo[j]; my[3 * NW + j] = W2e[j];
        }
        my[4 * NW] = 0; my[4 * NW + 1] = 0; my[4 * NW + 2] = 0; my[4 * NW + 3] = 0; my[4 * NW + 4] = 0;
        __builtin_amdgcn_wave_barrier();

        PROF(1);
        // ---- bubble (preprocesser.py:469-473)
        if (cfg.debubble && circ.n > 0 && fb.aux_ok) {
            bool hit = false;
            if (valid && flag < 0 && fb.aux_ok[rec]) {
                const int ln = fb.aux_lane[rec], tl = fb.aux_tile[rec], x = fb.aux_x[rec], y = fb.aux_y[rec];
                for (int i = 0; i < circ.n; ++i) {
                    if (circ.tile[i] == tl && circ.lane[i] == ln) {
                        const double dx = __dsub_rn(circ.cx[i], (double)x), dy = __dsub_rn(circ.cy[i], (double)y);
                        if (__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)) < __dmul_rn(circ.cr[i], circ.cr[i])) hit = true;
                    }
                }
            }
            if (hit) flag = AQC_BADBBL;
        }
        // ---- length (preprocesser.py:476-479)
        if (flag < 0 && len1 < cfg.seq_len_req) flag = AQC_BADLEN;
#if defined(AQC_ABLATE) && AQC_ABLATE == 4   /* no polyX */
        if (false) {
#else
        // ---- polyX (preprocesser.py:482-490): run-length screen per lane, exact check by the wave for the few hits
        if (cfg.poly_size_limit > 0) {
#endif
            bool sus1 = false, sus2 = false;
            if (run_req < 2) {
                sus1 = len1 >= cfg.poly_size_limit;
                sus2 = PAIRED && len2 >= cfg.poly_size_limit;
            } else {
#pragma unroll
                for (int which = 0; which < (PAIRED ? 2 : 1); ++which) {
                    const uint32_t* Wl = which ? W2lo : W1lo;
                    const uint32_t* We = which ? W2e : W1e;
                    const int ln = which ? len2 : len1;
                    uint32_t r[NW + 1];
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const uint32_t x = Wl[j] ^ alignbit(Wl[j + 1], Wl[j], 2);
                        const uint32_t ex = We[j] ^ alignbit(We[j + 1], We[j], 2);
                        // base i equals base i+1, only for i <= ln-2
                        r[j] = ~(((x << 1) | x | ex)) & ODD & base_mask(min(max(ln - 1 - 16 * j, 0), 16));
                    }
                    r[NW] = 0;
                    int covered = 1;
                    while (covered < run_req - 1) {
                        const int step = min(min(covered, run_req - 1 - covered), 15);
#pragma unroll
                        for (int j = 0; j < NW; ++j) r[j] &= alignbit(r[j + 1], r[j], 2 * step);
                        covered += step;
                    }
                    uint32_t any = 0;
#pragma unroll
                    for (int j = 0; j < NW; ++j) any |= r[j];
                    if (which) sus2 = any != 0 && ln >= cfg.poly_size_limit;
                    else sus1 = any != 0 && ln >= cfg.poly_size_limit;
                }
            }
            bool poly = false;
            unsigned long long todo = __ballot(valid && !defer && flag < 0 && (sus1 || sus2));
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const uint64_t r2 = base + l;
                const int s1f = __shfl((int)sus1, l, WAVE), s2f = __shfl((int)sus2, l, WAVE);
                const int ta1 = __shfl(a1, l, WAVE), tl1 = __shfl(len1, l, WAVE), ta2 = __shfl(a2, l, WAVE), tl2 = __shfl(len2, l, WAVE);
                int p = 0;
                if (s1f) {
                    stage(L.stage, fb.seq1 + ((uint64_t)fb.o1[r2] << 4) + ta1, tl1);
                    __builtin_amdgcn_wave_barrier();
                    p = has_polyx_wave(L.stage, tl1, cfg.poly_size_limit, cfg.allow_mismatch_in_poly);
                    __builtin_amdgcn_wave_barrier();
                }
                if (p == 0 && s2f) {
                    // hasPolyX runs on read 2 as sequenced (not reverse-complemented)
                    stage(L.stage, fb.seq2 + ((uint64_t)fb.o2[r2] << 4) + ta2, tl2);
                    __builtin_amdgcn_wave_barrier();
                    p = has_polyx_wave(L.stage, tl2, cfg.poly_size_limit, cfg.allow_mismatch_in_poly);
                    __builtin_amdgcn_wave_barrier();
                }
                if (p != 0 && lane == l) poly = true;
            }
            if (poly) flag = AQC_BADPOL;
        }
        PROF(2);
        // ---- low quality: read 1 only (preprocesser.py:498)
        if (flag < 0 && cfg.unqualified_base_limit > 0 && (int)L.lq[lane] > cfg.unqualified_base_limit) flag = AQC_BADLQC;
        // ---- N (preprocesser.py:504-512)
        if (flag < 0 && cfg.n_base_limit > 0) {
            int n1 = 0, n2 = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) { n1 += __popc(W1e[j]); n2 += __popc(W2e[j]); }
            if (n1 > cfg.n_base_limit || n2 > cfg.n_base_limit) flag = AQC_BADNCT;
        }

        PROF(3);
        // ---- overlap (util.py:158-212) --------------------------------------------------------------
        int offset = 0, ovl = 0, dist = 0, ovl0 = -1, dist_final = -1, n_edits = 0;
        int c_adapter_base = 0, c_adapter_read = 0, c_overlapped = 0, c_corrected = 0, c_masked = 0, c_skipped = 0, c_read_corrected = 0;
        int em0 = -1, em1 = -1, em2 = -1;
        aqc_edit ed0 = {0, 0, 0, 0}, ed1 = {0, 0, 0, 0}, ed2 = {0, 0, 0, 0};
        if (PAIRED && !cfg.no_overlap) {
            bool scan = valid && !defer && flag < 0;
            const int nf = len1 > 30 ? len1 - 30 : 0, nr = len2 > 30 ? len2 - 30 : 0;
            // the 16-base prefix test needs 16 columns on every diagonal
            if (scan && ((nf > 0 && len2 < 16) || (nr > 0 && len1 < 16))) { defer = true; scan = false; }
            int from = 0;               // first candidate (in the reference's enumeration order) still to be examined
            bool found = false;
            const uint32_t F2 = W2lo[0], F1 = W1lo[0];
#if defined(AQC_ABLATE) && AQC_ABLATE == 3   /* no scan */
            scan = false;
#endif
            while (true) {
                const int wmax_f = wave_max_i(scan && !found ? nf : 0);
                const int wmax_r = wave_max_i(scan && !found ? nr : 0);
                if (wmax_f == 0 && wmax_r == 0) break;
                int s0 = NONE_CAND, s1 = NONE_CAND, s2 = NONE_CAND;   // first three prefix survivors >= from
                const bool live = scan && !found;
                // forward diagonals d = 16k + r: read1[d + i] against reverse_r2[i].  Eight diagonals are
                // evaluated back to back (independent alignbit/xor/popcount chains) and share one branch.
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    if (16 * k >= wmax_f) break;
#pragma unroll
                    for (int r0 = 0; r0 < 16; r0 += 8) {
                        uint32_t cnt[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) cnt[q] = __popc(alignbit(W1lo[k + 1], W1lo[k], 2 * (r0 + q)) ^ F2);
                        const uint32_t best = min(min(min(cnt[0], cnt[1]), min(cnt[2], cnt[3])), min(min(cnt[4], cnt[5]), min(cnt[6], cnt[7])));
                        if (__ballot(best < 5)) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int c = 16 * k + r0 + q;
                                if (cnt[q] < 5 && live && c < nf && c >= from) {
                                    if (s0 == NONE_CAND) s0 = c; else if (s1 == NONE_CAND) s1 = c; else if (s2 == NONE_CAND) s2 = c;
                                }
                            }
                        }
                    }
                }
                // reverse diagonals a = 16k + r: read1[i] against reverse_r2[a + i]
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    if (16 * k >= wmax_r) break;
#pragma unroll
                    for (int r0 = 0; r0 < 16; r0 += 8) {
                        uint32_t cnt[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) cnt[q] = __popc(alignbit(W2lo[k + 1], W2lo[k], 2 * (r0 + q)) ^ F1);
                        const uint32_t best = min(min(min(cnt[0], cnt[1]), min(cnt[2], cnt[3])), min(min(cnt[4], cnt[5]), min(cnt[6], cnt[7])));
                        if (__ballot(best < 5)) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int a = 16 * k + r0 + q;
                                const int c = nf + a;
                                if (cnt[q] < 5 && live && a < nr && c >= from) {
                                    if (s0 == NONE_CAND) s0 = c; else if (s1 == NONE_CAND) s1 = c; else if (s2 == NONE_CAND) s2 = c;
                                }
                            }
                        }
                    }
                }
                PROF(4);
                // exact verification of up to three survivors per lane, in order (util.py:177-184 / 200-207)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int c = v == 0 ? s0 : v == 1 ? s1 : s2;
                    const bool check = live && !found && c != NONE_CAND;
                    if (!__ballot(check)) break;
                    const bool fwd = c < nf;
                    const int off = fwd ? c : c - nf;
                    const int QL = fwd ? min(len1 - off, len2) : min(len1, len2 - off);
                    const uint32_t* mv = my + (fwd ? 0 : 2 * NW);      // moving stream: lo at mv[], e at mv[NW + ]
                    const uint32_t* fx = my + (fwd ? 2 * NW : 0);      // fixed stream
                    const int k = off >> 4;
                    const uint32_t s = (uint32_t)(off & 15) * 2;
                    int tot = 0, c50 = 0;
                    if (check) {
                        uint32_t lo0 = mv[k], e0 = mv[NW + k];
#pragma unroll
                        for (int j = 0; j < NW; ++j) {
                            const uint32_t lo1 = (k + j + 1 < NW) ? mv[k + j + 1] : 0u, e1 = (k + j + 1 < NW) ? mv[NW + k + j + 1] : 0u;
                            const uint32_t mm = mm_word(lo0, lo1, e0, e1, s, fx[j], fx[NW + j], min(max(QL - 16 * j, 0), 16));
                            const int pc = __popc(mm);
                            tot += pc;
                            if (j < 3) c50 += pc;
                            else if (j == 3) c50 += __popc(mm & 0xFu);
                            lo0 = lo1; e0 = e1;
                        }
                        if (tot < 3 || (c50 < 3 && QL >= 52)) {
                            found = true;
                            offset = fwd ? off : -off;
                            ovl = QL;
                            dist = tot;
                        }
                    }
                }
                PROF(5);
                // a lane whose three survivors all failed and that may have more continues after the third one
                const bool more = live && !found && s2 != NONE_CAND;
                if (more) from = s2 + 1;
                if (live && !found && s2 == NONE_CAND) scan = false;   // exhausted: (0, 0, 0)
                if (!__ballot(more)) break;
            }

            PROF(5);
            // ---- post-processing (preprocesser.py:516-617)
            const bool reached = valid && !defer && flag < 0;
            if (reached) {
                ovl0 = ovl;
                if (offset < 0 && ovl > 30) {
                    // adapter read-through: both reads are cut to overlap_len and util.overlap runs again.  When
                    // overlap_len == len2 - |offset| the second call's first candidate (offset 0) is exactly the
                    // diagonal just accepted, so it returns (0, overlap_len, diff) again; otherwise defer.
                    if (ovl != len2 + offset) defer = true;
                    else {
                        c_adapter_base = -2 * offset; c_adapter_read = 1;
                        // the accepted diagonal in the cut reads: read1[i] vs reverse_r2[|offset| + i]
                        if (ovl < cfg.seq_len_req) { flag = AQC_BADLEN; len1 = ovl; len2 = ovl; offset = 0; ovl = 0; dist = 0; }
                    }
                }
            }
            const bool adapter = reached && !defer && c_adapter_read;
            const int walk_a = adapter ? -offset : 0;      // reverse diagonal of the walk in the adapter case
            if (adapter && flag < 0) { len1 = ovl; len2 = ovl; offset = 0; }
            if (reached && !defer && flag < 0) {
                dist_final = dist;
                if (dist > 3) flag = AQC_BADDIFF;
                else if (ovl > 30) {
                    c_overlapped = 1;
                }
            }
            // ---- correction walk (preprocesser.py:563-598): first `dist` mismatches of the tail-anchored diagonal
#if defined(AQC_ABLATE) && AQC_ABLATE == 5   /* no correction walk */
            const bool walk = false;
#else
            const bool walk = reached && !defer && flag < 0 && c_overlapped && dist > 0;
#endif
            if (__ballot(walk)) {
                int p0 = -1, p1 = -1, p2 = -1, nfound = 0;
                // coordinates in the ORIGINAL (pre adapter cut) normalised streams held in LDS
                const bool fwd = !adapter;
                const int off = fwd ? (len1 - ovl) : walk_a;
                const uint32_t* mv = my + (fwd ? 0 : 2 * NW);
                const uint32_t* fx = my + (fwd ? 2 * NW : 0);
                const int k = off >> 4;
                const uint32_t s = (uint32_t)(off & 15) * 2;
                if (walk) {
                    uint32_t lo0 = mv[k], e0 = mv[NW + k];
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const uint32_t lo1 = (k + j + 1 < NW) ? mv[k + j + 1] : 0u, e1 = (k + j + 1 < NW) ? mv[NW + k + j + 1] : 0u;
                        uint32_t mm = mm_word(lo0, lo1, e0, e1, s, fx[j], fx[NW + j], min(max(ovl - 16 * j, 0), 16));
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            if (mm != 0 && nfound < 3) {
                                const int pos = 16 * j + ((__ffs((int)mm) - 1) >> 1);
                                if (nfound == 0) p0 = pos; else if (nfound == 1) p1 = pos; else p2 = pos;
                                nfound++;
                                mm &= mm - 1;
                            }
                        }
                        lo0 = lo1; e0 = e1;
                    }
                }
                const int handled = min(nfound, dist);
                // bytes of the (<= 3) handled mismatches straight from the canonical arenas: all loads issued
                // before any is consumed, so the walk costs one memory round trip
                const uint8_t* g1 = fb.seq1 + ((uint64_t)L.o1[lane] << 4) + a1;
                const uint8_t* h1 = fb.qual1 + ((uint64_t)L.o1[lane] << 4) + a1;
                const uint8_t* g2 = fb.seq2 + ((uint64_t)L.o2[lane] << 4) + a2;
                const uint8_t* h2 = fb.qual2 + ((uint64_t)L.o2[lane] << 4) + a2;
                uint8_t wb1[3], wb2[3], wq1[3], wq2[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int oo = q == 0 ? p0 : q == 1 ? p1 : p2;
                    const bool use = walk && q < handled;
                    const int i1 = use ? len1 - ovl + oo : 0, i2 = use ? len2 - 1 - oo : 0;
                    wb1[q] = g1[i1]; wb2[q] = g2[i2]; wq1[q] = h1[i1]; wq2[q] = h2[i2];
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int oo = q == 0 ? p0 : q == 1 ? p1 : p2;
                    if (walk && q < handled) {
                        const uint8_t bA = wb1[q], r2o = wb2[q];
                        const uint8_t bB = comp_strict(r2o);
                        const int qa = wq1[q], qb = wq2[q];
                        bool fixed = false;
                        int em = -1;
                        aqc_edit ed = {0, 0, 0, 0};
                        bool have_edit = false;
                        if (qa - 33 >= 30 && qb - 33 <= 14) {
                            if (bA != 'N' && bB != 'N') em = base_idx(comp_strict(bA)) * 4 + base_idx(r2o);
                            if (!cfg.no_correction) {
                                ed = aqc_edit{(uint16_t)oo, AQC_EDIT_FIX_R2, comp_strict(bA), (uint8_t)qa};
                                have_edit = true; c_corrected++; fixed = true;
                            }
                        } else if (qb - 33 >= 30 && qa - 33 <= 14) {
                            if (bA != 'N' && bB != 'N') em = base_idx(bB) * 4 + base_idx(bA);
                            if (!cfg.no_correction) {
                                ed = aqc_edit{(uint16_t)oo, AQC_EDIT_FIX_R1, bB, (uint8_t)qb};
                                have_edit = true; c_corrected++; fixed = true;
                            }
                        }
                        if (!fixed) {
                            if (cfg.mask_mismatch) { ed = aqc_edit{(uint16_t)oo, AQC_EDIT_MASK, 0, (uint8_t)'!'}; have_edit = true; c_masked++; }
                            else c_skipped++;
                        }
                        if (q == 0) em0 = em; else if (q == 1) em1 = em; else em2 = em;
                        if (have_edit) {
                            if (n_edits == 0) ed0 = ed; else if (n_edits == 1) ed1 = ed; else ed2 = ed;
                            n_edits++;
                        }
                    }
                }
                if (walk) {
                    if (handled == dist) { if (c_corrected > 0) c_read_corrected = 1; }
                    else {
                        flag = AQC_BADMISMATCH;
                        em0 = em1 = em2 = -1;
                        c_corrected = c_masked = c_skipped = 0;
                    }
                }
            }
        }
        if (flag < 0) flag = AQC_GOOD;
        PROF(6);

        // ------------------------------------------------------------------ results + counters
        const bool mine = valid && !defer;
        if (mine) {
            // struct aqc_result (packed, 32 bytes) assembled in registers and written as two 16-byte stores
            auto e40 = [](const aqc_edit& e) {
                return (unsigned long long)e.o | ((unsigned long long)e.kind << 16) | ((unsigned long long)e.base << 24) |
                       ((unsigned long long)e.qual << 32);
            };
            const unsigned long long E0 = e40(ed0), E1 = e40(ed1), E2 = e40(ed2);
            uint4 lo, hi;
            lo.x = (uint32_t)flag | ((uint32_t)n_edits << 8) | ((uint32_t)(a1 & 0xffff) << 16);
            lo.y = (uint32_t)(len1 & 0xffff) | ((uint32_t)(a2 & 0xffff) << 16);
            lo.z = (uint32_t)(len2 & 0xffff) | ((uint32_t)(offset & 0xffff) << 16);
            lo.w = (uint32_t)(ovl & 0xffff) | ((uint32_t)(dist & 0xffff) << 16);
            const unsigned long long q0 = E0 | (E1 << 40), q1 = (E1 >> 24) | (E2 << 16);   // byte 31 (barcode) = 0
            hi.x = (uint32_t)q0; hi.y = (uint32_t)(q0 >> 32); hi.z = (uint32_t)q1; hi.w = (uint32_t)(q1 >> 32);
            uint4* out = reinterpret_cast<uint4*>(results + rec);
            out[0] = lo;
            out[1] = hi;
        }
        const bool cnt = mine && accum;
        {
            unsigned long long* C = acc.counters;
            const unsigned long long mcnt = __ballot(cnt);
            if (mcnt) {
                const int r2b = (PAIRED && cfg.count_r2_bases) ? 1 : 0;
                const int tb = wave_sum(cnt ? L1 + r2b * L2 : 0);
                const int gb = wave_sum(cnt && flag == AQC_GOOD ? len1 + r2b * len2 : 0);
                const int ng = __popcll(__ballot(cnt && flag == AQC_GOOD));
                if (lane == 0) {
                    atomicAdd(&C[AQC_C_TOTAL_READS], (unsigned long long)__popcll(mcnt));
                    atomicAdd(&C[AQC_C_TOTAL_BASES], (unsigned long long)tb);
                    atomicAdd(&C[AQC_C_GOOD_READS], (unsigned long long)ng);
                    atomicAdd(&C[AQC_C_GOOD_BASES], (unsigned long long)gb);
                }
                for (int f = 0; f < AQC_N_FLAGS; ++f) {
                    const int k = __popcll(__ballot(cnt && flag == f));
                    if (k && lane == 0) atomicAdd(&C[AQC_C_FLAG0 + f], (unsigned long long)k);
                }
                if (PAIRED) {
                    if (cnt && ovl0 >= 0) atomicAdd(&acc.ovl_hist[ovl0], 1u);
                    if (cnt && dist_final >= 0) atomicAdd(&acc.dist_hist[min(dist_final, AQC_QC_COLS - 1)], 1u);
                    const int s_ab = wave_sum(cnt ? c_adapter_base : 0), s_ar = wave_sum(cnt ? c_adapter_read : 0);
                    const int s_ov = wave_sum(cnt ? c_overlapped : 0), s_ol = wave_sum(cnt && c_overlapped ? ovl : 0);
                    const int s_od = wave_sum(cnt && c_overlapped ? dist : 0);
                    const int s_rc = wave_sum(cnt ? c_read_corrected : 0), s_bc = wave_sum(cnt ? c_corrected : 0);
                    const int s_mk = wave_sum(cnt ? c_masked : 0), s_sk = wave_sum(cnt ? c_skipped : 0);
                    if (lane == 0) {
                        if (s_ar) { atomicAdd(&C[AQC_C_TRIMMED_ADAPTER_BASE], (unsigned long long)s_ab); atomicAdd(&C[AQC_C_TRIMMED_ADAPTER_READ], (unsigned long long)s_ar); }
                        if (s_ov) {
                            atomicAdd(&C[AQC_C_OVERLAPPED], (unsigned long long)s_ov);
                            atomicAdd(&C[AQC_C_OVERLAP_LEN_SUM], (unsigned long long)s_ol);
                            atomicAdd(&C[AQC_C_OVERLAP_BASE_SUM], (unsigned long long)(2 * s_ol));
                            atomicAdd(&C[AQC_C_OVERLAP_BASE_ERR], (unsigned long long)s_od);
                        }
                        if (s_rc) atomicAdd(&C[AQC_C_READ_CORRECTED], (unsigned long long)s_rc);
                        if (s_bc) atomicAdd(&C[AQC_C_BASE_CORRECTED], (unsigned long long)s_bc);
                        if (s_mk) atomicAdd(&C[AQC_C_BASE_ZERO_QUAL_MASKED], (unsigned long long)(2 * s_mk));
                        if (s_sk) atomicAdd(&C[AQC_C_BASE_SKIPPED_CORRECTION], (unsigned long long)(2 * s_sk));
                    }
                    if (cnt && em0 >= 0) atomicAdd(&C[AQC_C_ERR_MATRIX0 + em0], 1ull);
                    if (cnt && em1 >= 0) atomicAdd(&C[AQC_C_ERR_MATRIX0 + em1], 1ull);
                    if (cnt && em2 >= 0) atomicAdd(&C[AQC_C_ERR_MATRIX0 + em2], 1ull);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        PROF(7);
        // ------------------------------------------------------------------ deferred pairs: general pipeline, one at a time
        unsigned long long dmask = __ballot(valid && defer);
        if (dmask) {
            uint8_t* area = reinterpret_cast<uint8_t*>(&L.planes[0][0]);
            const WaveLds w{area, area + LSTR, area + 2 * LSTR, area + 3 * LSTR, area + 4 * LSTR, L.rs[0], L.rs[1]};
            while (dmask) {
                const int l = __ffsll((long long)dmask) - 1;
                dmask &= dmask - 1;
                const uint64_t r2 = base + l;
                process_record_wave(raw, r2, cfg, circ, w, results, acc, st, r2 < accum_limit);
                __builtin_amdgcn_wave_barrier();
            }
        }
        __builtin_amdgcn_wave_barrier();
        PROF(8);
    }
    PROF_FLUSH;
    __syncthreads();
    flush_block_acc(acc, st);
}

// ------------------------------------------------------------------------------------------------
// canonicalisation (part of aqc_upload, not of the hot path): copy every record of a raw arena
// (arbitrary alignment, e.g. FASTQ text addressed in place) to a 16-byte aligned slot and fill the
// rest of the slot's last 16-byte chunk with `pad`.
// ------------------------------------------------------------------------------------------------
__global__ void canonicalize_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ off,
                                    const uint32_t* __restrict__ len, const uint32_t* __restrict__ o16, uint64_t n,
                                    uint8_t* __restrict__ dst, uint8_t pad) {
    // 16 lanes per record, lane = 16-byte chunk of the record (looping for reads longer than 256 bytes)
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t rec = gid >> 4;
    if (rec >= n) return;
    const int l = (int)len[rec];
    const uint8_t* s = src + off[rec];
    uint8_t* d = dst + ((uint64_t)o16[rec] << 4);
    const int nchunks = (l + 15) >> 4;
    for (int c = (int)(gid & 15); c < nchunks; c += 16) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 16 * c + 4 * k + j;
                v |= (uint32_t)(i < l ? s[i] : pad) << (8 * j);
            }
            w[k] = v;
        }
        *reinterpret_cast<uint4*>(d + 16 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

}  // namespace aqc
