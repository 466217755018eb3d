// aqc_fast.hpp — the hot kernel: "one LANE per READ" (two lanes per pair), generation 5.
//
// Why: the wave-per-record kernel (aqc_kernels.hpp) spends ~1500 wave-instructions per pair, which
// caps it at ~1 % of the HBM roofline.  To stream pairs at a useful fraction of 8 TB/s the whole
// pipeline must cost on the order of 100 wave-instructions per pair (256 CUs x 4 SIMDs x ~1.1 G
// wave-instr/s / 5 G pairs/s), i.e. every lane has to do useful work all the time and byte
// compares have to become word compares.  It is at 77 per pair now.  Design:
//
//   phase 1 (cooperative, coalesced)   the wave owns 32 consecutive pairs.  Chunk t = it * 64 + lane of the batch is
//       chunk t % NW of record t / NW: NW neighbouring lanes cover one read.  Each lane loads its 16-byte chunk
//       (global_load_dwordx4 at the read's own alignment) and converts it with SWAR + v_dot4/v_perm into
//         lo  : 2 bits per base  ((c >> 1) & 3: A=0 C=1 T=2 G=3; N shares 3)            32 bits/chunk
//         e   : 1 bit per base (odd bit of the 2-bit field) set for 'N'                32 bits/chunk
//       Read 2 is cut into chunks from its END, complemented and reversed: chunk c IS word c of reverse_r2
//       (util.py:161), a forward 2-bit stream that starts at bit 0.  Behind a read both streams continue A C A C ...
//       Low-quality counts of read 1 are reduced per chunk.  The planes and the record's descriptor (offsets, lengths,
//       alphabet verdict, low-quality count) share one LDS row per record.
//   phase 2 (lane per read)            lanes 2p / 2p+1 own read 1 / reverse_r2 of pair p and run the pipeline of
//       preprocesser.py:455-617 on 32-bit words fetched from the row as needed (a trim / barcode stage that moves a
//       view re-aligns the stream in place first).  The two roles are symmetric: the read-1 lane scans the forward
//       offsets of util.py:172-186 (its stream moves over the partner's 16-base prefix), the read-2 lane the reverse
//       offsets of :194-209 — same instructions, different data, values exchanged with one DPP quad_perm.
//         * overlap scan: one diagonal = v_alignbit + v_xor + v_bcnt on a 16-base prefix window;
//           >= 5 differing bits imply >= 3 mismatching bases, which util.py:180-183 can never accept;
//         * the rare survivors are verified exactly over the full diagonal (lo and e planes);
//         * the correction walk reads the <= 3 mismatch positions off the same words, in 2-bit codes.
//   exactness: bytes outside {A,C,G,T,N}, reads longer than 16*NW, reads shorter than the 16-base
//       prefix, and the adapter-trim / walk-anchor corner cases that need a second scan are not
//       handled here: the lane DEFERS its pair to a device queue and filter_overlap_list_kernel runs the fully
//       general wave-per-record pipeline (process_record_wave) for it right behind this kernel.  Results are
//       bit-identical either way; only the speed differs.
//
// Input is the batch exactly as it sits in HBM (DevBatch): byte arenas + 32-bit byte offsets + lengths — the raw FASTQ
// text chunk addressed in place, or the caller's packed SoA arenas.  Nothing is copied or re-laid-out first: a lane
// replaces the bytes outside the read by the pad symbols with one v_bfi per dword (mask from a 17-entry LDS table) and
// validates the alphabet on the fly (v_perm round trip; a record with any other byte is deferred).  Generation 2 needed a
// separate canonicalisation pass (read 3.0 GB + write 3.2 GB per 5 M pairs, 2.8x the time of this kernel) for that.
#pragma once
#include "aqc_kernels.hpp"

namespace aqc {

constexpr uint32_t ODD = 0xAAAAAAAAu;
// What stands behind a read in its 2-bit stream: A C A C ... by stream position (no two neighbours equal, so the padding
// never looks like a homopolymer run to the polyX screen and that screen needs no length masks).  PAD1 / PAD2 are the text
// bytes that pack to it: "ACAC" for read 1, "GTGT" for read 2 (complemented and reversed: position p comes from byte 15 - p).
constexpr uint32_t PADLO = 0x44444444u, PAD1 = 0x43414341u, PAD2 = 0x54475447u;
constexpr int NONE_CAND = 0x7fffffff;

// 16 bytes from an arbitrarily aligned address: one global_load_dwordx4
__device__ __forceinline__ uint4 load16u(const uint8_t* p) {
    uint4 v;
#if defined(AQC_ABL) && (AQC_ABL & 128)
    p = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)15);      // (ablation: what would loads on the 16-byte grid give?)
#endif
    __builtin_memcpy(&v, p, 16);
    return v;
}
// (mask & a) | (~mask & b): v_bfi_b32
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
// bytes of d that are none of A C G T N come back non-zero: table round trip through the 3-bit code (c >> 1) & 7
// (A0 C1 T2 G3 N7; codes 4..6 map to 0x00, which no text byte equals)
__device__ __forceinline__ uint32_t not_acgtn(uint32_t d) {
    return d ^ __builtin_amdgcn_perm(0x4e000000u, 0x47544341u, (d >> 1) & 0x07070707u);
}

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// mask with the low 2*nb bits set, nb in 0..16
__device__ __forceinline__ uint32_t base_mask(int nb) { return nb >= 16 ? 0xffffffffu : ((1u << (2 * nb)) - 1u); }

// Wave-wide max / sum of a lane value, in uniform control flow (all 64 lanes active), as a wave-uniform result: four DPP
// steps inside the rows of 16 lanes (lane ^ 1, lane ^ 2, half-row mirror, row mirror), then the four row results through
// v_readlane and scalar arithmetic.  The __shfl_xor form goes through the LDS crossbar (six ds_bpermute round trips) and
// keeps its six lane-address registers alive for the whole kernel.
template <int CTRL>
__device__ __forceinline__ int dpp_move(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_move<0xB1>(v));       // quad_perm [1,0,3,2]
    v = max(v, dpp_move<0x4E>(v));       // quad_perm [2,3,0,1]
    v = max(v, dpp_move<0x141>(v));      // row_half_mirror
    v = max(v, dpp_move<0x140>(v));      // row_mirror
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ void store16f(uint8_t* p, uint4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ int wave_sum_u(int v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

// Inclusive prefix sum / running maximum over the 64 lanes, in uniform control flow: four DPP row_shr steps scan the rows of 16
// lanes (lanes shifted in from outside a row read 0), the three row totals come through v_readlane.
template <int CTRL>
__device__ __forceinline__ int dpp_shr0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int wave_incl_sum(int v, int lane) {
    v += dpp_shr0<0x111>(v);
    v += dpp_shr0<0x112>(v);
    v += dpp_shr0<0x114>(v);
    v += dpp_shr0<0x118>(v);
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {      // (values >= 0)
    v = max(v, dpp_shr0<0x111>(v));
    v = max(v, dpp_shr0<0x112>(v));
    v = max(v, dpp_shr0<0x114>(v));
    v = max(v, dpp_shr0<0x118>(v));
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return max(max(v, row >= 1 ? t0 : 0), max(row >= 2 ? t1 : 0, row >= 3 ? t2 : 0));
}

// 16 sequence bytes -> lo plane (2 bits/base: (c >> 1) & 3), e plane ('N' flag, bit 3 of the byte, on the odd bit)
__device__ __forceinline__ void pack_dword(uint32_t d, uint32_t& lo8, uint32_t& e8) {
    lo8 = udot4((d >> 1) & 0x03030303u, 0x40100401u, 0u);
    e8 = udot4((d >> 3) & 0x01010101u, 0x80200802u, 0u);
}

__device__ __forceinline__ void pack_chunk(const uint4 v, uint32_t& lo, uint32_t& e) {
    uint32_t l0, l1, l2, l3, e0, e1, e2, e3;
    pack_dword(v.x, l0, e0);
    pack_dword(v.y, l1, e1);
    pack_dword(v.z, l2, e2);
    pack_dword(v.w, l3, e3);
    // (three shift-or steps each; written as a chain so that it stays one)
    lo = (((((l3 << 8) | l2) << 8) | l1) << 8) | l0;
    e = (((((e3 << 8) | e2) << 8) | e1) << 8) | e0;
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

// the length this kernel works with: a mate marked LEN_IRR (its quality line has a length of its own: the general kernel's
// business) counts as EMPTY — one v_max per lane and batch; phase 1 then sees padding only and never forms an address from it
#ifdef AQC_NO_IRR_VMAX      // measurement builds only (tools/build_ablate.sh): the kernel without that v_max — wrong for irregular records
__device__ __forceinline__ uint32_t lane_len(uint32_t len_word) { return len_word; }
#else
__device__ __forceinline__ uint32_t lane_len(uint32_t len_word) { return (uint32_t)max((int)len_word, 0); }
#endif

template <int NW, bool PAIRED, bool FUSE = false>
struct FastWaveLds {
    static constexpr int PPW = PAIRED ? 32 : 64;               // records per wave batch
    static constexpr int GUARD = (PAIRED ? 4 : 2) * NW;        // 5 zero words behind the planes
    static constexpr int LQ0 = GUARD + 5;                      // read 1's low-quality bit plane: 16 bits per chunk
    // the record's descriptor lives in its row too: one row address serves the planes and these (immediate offsets)
    static constexpr int D_O1 = LQ0 + (NW + 1) / 2, D_L1 = D_O1 + 1, D_Q1 = D_O1 + 2, D_O2 = D_O1 + 3, D_L2 = D_O1 + 4, D_Q2 = D_O1 + 5;
    static constexpr int D_EXO = D_O1 + 6, D_LQ = D_O1 + 7;    // alphabet verdict (non-zero: defer), read 1's low-quality count
    // FUSE (the kernel also places every record and copies the whole ones): where each mate's record starts in its text
    // (bit 31: all four lines end right at their '\n')
    static constexpr int D_N1 = D_O1 + 8, D_N2 = D_O1 + 9;
    static constexpr int STRIDE = (D_O1 + (FUSE ? 10 : 8)) | 1;   // odd: conflict-free lane-strided access
    uint32_t planes[PPW][STRIDE];
    uint8_t stage[16 * NW + 16];
    // FUSE: the batch whose records are copied one batch later (by then its predecessors have long published their sums): per
    // record and file the first byte in the text | copied here << 31, and its length | offset inside the batch's bytes << 16
    uint32_t pend[FUSE ? 2 : 1][FUSE ? 4 * PPW : 1];
};

// reverse the order of the thirty-two 2-bit fields of a 64-bit value given as (w0 = fields 0..15, w1 = fields 16..31)
__device__ __forceinline__ void rev2_64(uint32_t& w0, uint32_t& w1) {
    const uint32_t a = rev2(w1), b = rev2(w0);
    w0 = a; w1 = b;
}
// the 32 fields [start, start + 32) of a plane (16 fields per word); reads plane[start / 16 .. + 2]
__device__ __forceinline__ void win64(const uint32_t* plane, int start, uint32_t& w0, uint32_t& w1) {
    const int k = start >> 4;
    const uint32_t s = (uint32_t)(start & 15) * 2;
    const uint32_t a = plane[k], b = plane[k + 1], c = plane[k + 2];
    w0 = alignbit(b, a, s);
    w1 = alignbit(c, b, s);
}
// the sixteen even bits of w, compacted into the low half
__device__ __forceinline__ uint32_t even_bits16(uint32_t w) {
    uint32_t x = w & 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

// complement / ALL_BASES index of a byte KNOWN to be one of A,C,G,T,N (the fast path only sees such bytes):
// index (c >> 1) & 7 is A=0 C=1 T=2 G=3 N=7
__device__ __forceinline__ uint32_t comp_acgtn(uint32_t c) {
    return __builtin_amdgcn_perm(0x4e000000u, 0x43414754u, (c >> 1) & 7u) & 0xffu;   // T G A C . . . N
}
__device__ __forceinline__ int base_idx_acgt(uint32_t c) { return (int)((0x3120u >> (((c >> 1) & 3u) * 4u)) & 0xfu); }   // A T C G -> 0 1 2 3

// exchange a value with the partner lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ int xchg(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
// the partner lane's predicate (all lanes must call it).  One DPP move: the ballot form needs the lane's 64-bit bit mask,
// a loop invariant the compiler keeps in two registers for the whole kernel
__device__ __forceinline__ bool xchg_pred(bool b) { return xchg(b ? 1 : 0) != 0; }
// number of set bits of m below this lane
__device__ __forceinline__ unsigned int rank_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

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
#define PROF_DEFER(k, cond) do { if (cond) atomicAdd(&R->st.counters[AQC_N_COUNTERS + 10 + (k)], 1ull); } while (0)
#define PROF_DECL unsigned long long prof_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_last = __builtin_amdgcn_s_memtime(); const unsigned long long prof_t0 = prof_last;
#define PROF(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof_t[k] += now_ - prof_last; prof_last = now_; } while (0)
#define PROF_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 10; ++k_) atomicAdd(&R->st.counters[AQC_N_COUNTERS + k_], prof_t[k_]); } while (0)
#else
#define PROF_DEFER(k, cond)
#define PROF_DECL
#define PROF(k)
#define PROF_FLUSH
#endif

#ifndef AQC_MIN_WAVES
#define AQC_MIN_WAVES 4
#endif
#ifndef AQC_PRIO2
#define AQC_PRIO2 2
#endif
#ifndef AQC_PRIO1
#define AQC_PRIO1 0
#endif
#ifndef AQC_COOP_VERIFY
#define AQC_COOP_VERIFY 1      // 1: the survivors' diagonals are verified by the whole wave, a lane per (survivor, 16-base word); 0: every lane its own
#endif
#ifndef AQC_ABL
#define AQC_ABL 0      // ablation builds only (tools/gpu_ablate.sh; results are WRONG, only instruction counts / times mean anything):
                       // 1 no alphabet validation, 2 no length masks, 4 no polyX screen, 8 no diagonal scan, 16 no correction walk,
                       // 32 no N count, 64 no phase 1 packing (loads only), 128 every 16-byte load moved down to the 16-byte grid
#endif

// what the barcode stage needs of aqc_config, decoded once per kernel (uniform): verify as 2-bit codes
struct BarcodeCodes {
    uint32_t v0, v1;      // verify, field j = (verify[j] >> 1) & 3, fields 0..15 / 16..31
    uint32_t m0, m1;      // even-bit mask of the verify's fields
};

// The kernel's arguments, one struct.  The hot loop keeps what it uses in every batch in scalar registers (K.field); what
// only rare branches need — the bubble circles and name fields, the trim amounts, the deferral queue, the device status
// word, the statistics arrays — is read from the kernarg segment where it is used (R->field, an s_load): held in SGPRs
// across the loop those ~40 values pushed the kernel past the 102 it has, and the spill code (v_writelane / v_readlane
// + hazard nops) was ~6 % of all vector instructions issued.
// FUSE: what the verdict kernel needs to place every record of the chunk in its output stream and to copy the good records that
// go out as their own bytes itself.  Batches are committed in index order: a batch publishes the bytes it adds to
// the four streams (good / bad of either file) and looks back over its predecessors' (decoupled look-back, as text_index_kernel).
struct FuseArgs {
    const uint32_t *name_off1, *name_off2;       // where each record starts in its file's text
    uint8_t *out1, *out2;                        // the good streams
    uint32_t *fstate1, *fstate2;                 // per record and file: offset inside its batch's share of its stream (16 bits) | FUSE_WHOLE (its own
                                                 // bytes, written here) | FUSE_PATCH (... but for the walk's byte patches, which the plan pass stores)
    unsigned long long* state;                   // [2 * batches]: flag (2 bits) | good1 (31) | good2 (31), flag | bad1 | bad2: the batch's sums (A), then
                                                 // — final — the streams' bytes up to and including the batch (P): the plan pass reads those
    unsigned int* ticket;                        // rounds handed out (a round = WPBT consecutive batches)
    int* abort;                                  // != 0: this placement is void — a deferred pair, a record that is not contiguous text, a
                                                 // look-back that ran out of patience: the host formats the chunk the other way
    unsigned long long* totals;                  // [4] bytes of good1, good2, bad1, bad2
};
constexpr uint32_t FUSE_WHOLE = 0x80000000u, FUSE_POS = 0x7fffffffu, FUSE_PATCH = 0x40000000u;
constexpr unsigned long long FUSE_FLAG_A = 1ull << 62, FUSE_FLAG_P = 2ull << 62;

struct FastArgs {
    DevBatch fb;
    aqc_config cfg;
    DevCircles circ;
    aqc_result* results;
    DevStats st;
    uint64_t accum_limit;
    uint32_t* deferred;
    unsigned int* n_deferred;
    FuseArgs fz;
};
typedef const FastArgs __attribute__((address_space(4))) * FastArgsRare;
#ifndef AQC_FUSE_UNROLL
#define AQC_FUSE_UNROLL 12
#endif
#ifndef AQC_FUSE_ABL
#define AQC_FUSE_ABL 0          // measurement builds only: 1 no copy, 2 no stores, 4 no loads, 8 no look-back
#endif

// (one workgroup of WPBT waves per CU — its LDS rows allow no second one — is WPBT / 4 waves per SIMD: that is the occupancy the
//  register budget is sized for: 128 VGPRs for the 16-wave 2 x 150 variant, 168 for the 12-wave ones)
template <int NW, bool PAIRED, int WPBT, bool BARCODE, bool FUSE = false>
__global__ __launch_bounds__(WPBT * WAVE, (WPBT + 3) / 4 < AQC_MIN_WAVES ? (WPBT + 3) / 4 : AQC_MIN_WAVES) void fast_filter_overlap_kernel(FastArgs K) {
    static_assert(!FUSE || (PAIRED && !BARCODE), "the fused variant is the plain paired one");
    const DevBatch& fb = K.fb;
    const aqc_config& cfg = K.cfg;
    aqc_result* __restrict__ const results = K.results;
    const uint64_t accum_limit = K.accum_limit;
    FastArgsRare R = (FastArgsRare)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(R));                   // opaque: nothing read through R is hoisted out of its branch
    using WL = FastWaveLds<NW, PAIRED, FUSE>;
    constexpr int PPW = WL::PPW;
    constexpr int ITERS = PPW * NW / WAVE;        // 16-byte chunk tasks per lane and string kind
    static_assert(PPW * NW % WAVE == 0 && (!PAIRED || NW % 2 == 0), "chunk tasks must tile the wave");
    __shared__ WL wls[WPBT];
    __shared__ BlockAcc acc;
    __shared__ unsigned int batch_ticket;
    __shared__ unsigned long long round_slot[16];  // FUSE: (local round + 1) << 32 | the global round it was given
    __shared__ uint4 mtab[17];                    // mtab[nb]: byte mask of the first nb bytes of a 16-byte chunk
    // (lane from mbcnt, wave in a scalar register: nothing derived from the work-item id has to survive the main loop)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    for (int i = threadIdx.x; i < (int)(sizeof(BlockAcc) / 4); i += WPBT * WAVE) ((unsigned int*)&acc)[i] = 0;
    if (threadIdx.x == 0) batch_ticket = FUSE ? 0u : (unsigned int)WPBT;      // tickets 0 .. WPBT-1 are the waves' first batches
    if (FUSE && threadIdx.x < 16) round_slot[threadIdx.x] = 0ull;
    if (threadIdx.x < 17) {
        const int nb = threadIdx.x;
        uint32_t m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int vb = min(max(nb - 4 * k, 0), 4);
            m[k] = vb >= 4 ? 0xffffffffu : ((1u << (8 * vb)) - 1u);
        }
        mtab[nb] = make_uint4(m[0], m[1], m[2], m[3]);
    }
    __syncthreads();
    WL& L = wls[wave];
    const int p = PAIRED ? lane >> 1 : lane;      // record of this lane within the batch
    const int role = PAIRED ? lane & 1 : 0;       // 0: owns read 1 (forward offsets), 1: owns reverse_r2 (reverse offsets)
    uint32_t* const pr = L.planes[p];
    uint32_t* const own = pr + (role ? 2 * NW : 0);         // own stream: lo at own[j], e at own[NW + j]
    const uint32_t* const par = pr + (role ? 0 : 2 * NW);   // partner stream
    const uint32_t* const qo1 = fb.qoff1 ? fb.qoff1 : fb.off1;
    const uint32_t* const qo2 = PAIRED ? (fb.qoff2 ? fb.qoff2 : fb.off2) : nullptr;
    const int thr4 = __builtin_amdgcn_readfirstlane((cfg.qualified_quality_phred + 33) * 0x01010101);
    const int do_trim = __builtin_amdgcn_readfirstlane((cfg.trim_front > 0 || cfg.trim_tail > 0) ? 1 : 0);
    // read 1's low-quality count: with a view that is only known per read (trim, barcode) phase 1 leaves one bit per base
    // and the owner counts inside its view; otherwise the chunks are simply summed
    // longest run of identical bases any firing polyX window must contain (pigeonhole over the mismatches)
    const int need = cfg.poly_size_limit - cfg.allow_mismatch_in_poly;
    // (the division runs on the vector unit: bring the result back to a scalar register once)
    const int run_req = __builtin_amdgcn_readfirstlane(cfg.allow_mismatch_in_poly >= 0 ? (need + cfg.allow_mismatch_in_poly) / (cfg.allow_mismatch_in_poly + 1) : 0);
    const int r2b = __builtin_amdgcn_readfirstlane((PAIRED && cfg.count_r2_bases) ? 1 : 0);
    // (the one-word polyX screen needs the firing span to hold a whole aligned word and the word test to mean something)
    const int poly_word = __builtin_amdgcn_readfirstlane((cfg.allow_mismatch_in_poly >= 0 && cfg.poly_size_limit - cfg.allow_mismatch_in_poly >= 31 && cfg.allow_mismatch_in_poly <= 5) ? 1 : 0);
    BarcodeCodes bc{0, 0, 0, 0};
    if (BARCODE) {
        for (int j = 0; j < cfg.barcode_verify_len && j < 32; ++j) {
            const uint32_t code = ((uint32_t)cfg.barcode_verify[j] >> 1) & 3u;
            if (j < 16) { bc.v0 |= code << (2 * j); bc.m0 |= 1u << (2 * j); }
            else { bc.v1 |= code << (2 * (j - 16)); bc.m1 |= 1u << (2 * (j - 16)); }
        }
    }

    // per-lane running totals, reduced once at the end of the kernel
    // (packed: R0 = records | good | adapter reads | overlapped, 8 bits each; R1 = total bases | good bases, R2 = adapter bases |
    //  overlap length, 16 bits each; R3 = distance | corrected reads | corrected bases | masked, R4 = skipped, 8 bits each.  A
    //  lane adds at most 1 / 576 / 3 to a field per batch, so the wave flushes every 64 batches — five live registers, not 13)
    uint32_t R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0;
    int since_flush = 0;
    auto flush_totals = [&]() {
        unsigned long long* C = acc.counters;
        auto fld = [&](uint32_t r, int sh, uint32_t mask) -> unsigned long long { return (unsigned long long)(uint32_t)wave_sum_u((int)((r >> sh) & mask)); };
        const unsigned long long t_n = fld(R0, 0, 0xff), t_good = fld(R0, 8, 0xff), t_ar = fld(R0, 16, 0xff), t_ov = fld(R0, 24, 0xff);
        const unsigned long long t_tb = fld(R1, 0, 0xffff), t_gb = fld(R1, 16, 0xffff), t_ab = fld(R2, 0, 0xffff), t_ol = fld(R2, 16, 0xffff);
        const unsigned long long t_od = fld(R3, 0, 0xff), t_rc = fld(R3, 8, 0xff), t_bc = fld(R3, 16, 0xff), t_mk = fld(R3, 24, 0xff), t_sk = fld(R4, 0, 0xff);
        if (lane == 0) {
            atomicAdd(&C[AQC_C_TOTAL_READS], t_n);
            atomicAdd(&C[AQC_C_TOTAL_BASES], t_tb);
            atomicAdd(&C[AQC_C_GOOD_READS], t_good);
            atomicAdd(&C[AQC_C_GOOD_BASES], t_gb);
            atomicAdd(&C[AQC_C_FLAG0 + AQC_GOOD], t_good);
            atomicAdd(&C[AQC_C_TRIMMED_ADAPTER_BASE], t_ab);
            atomicAdd(&C[AQC_C_TRIMMED_ADAPTER_READ], t_ar);
            atomicAdd(&C[AQC_C_OVERLAPPED], t_ov);
            atomicAdd(&C[AQC_C_OVERLAP_LEN_SUM], t_ol);
            atomicAdd(&C[AQC_C_OVERLAP_BASE_SUM], 2ull * t_ol);
            atomicAdd(&C[AQC_C_OVERLAP_BASE_ERR], t_od);
            atomicAdd(&C[AQC_C_READ_CORRECTED], t_rc);
            atomicAdd(&C[AQC_C_BASE_CORRECTED], t_bc);
            atomicAdd(&C[AQC_C_BASE_ZERO_QUAL_MASKED], 2ull * t_mk);
            atomicAdd(&C[AQC_C_BASE_SKIPPED_CORRECTION], 2ull * t_sk);
        }
        R0 = R1 = R2 = R3 = R4 = 0;
        since_flush = 0;
    };

    PROF_DECL
    // Wave batches (PPW consecutive records) are NOT dealt statically: a saturated SIMD serves its resident waves by age,
    // the oldest wave of four runs ~35 % faster than the youngest, and with equal shares the SIMD idles while the
    // stragglers finish alone (measured: mean wave lifetime 79 % of the kernel).  The workgroup owns one batch of every
    // group of gridDim.x batches (batch_of(t)) and its waves draw the tickets t from an LDS counter — one LDS atomic per batch.
    const uint32_t n_batches = (uint32_t)((fb.n + PPW - 1) / PPW);
    auto draw = [&]() -> uint32_t {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&batch_ticket, 1u);
        return t;                                      // (lane 0 holds it; broadcast when consumed)
    };
    // (the workgroup's column rotates with t, so that its batches are spread over all memory channels)
    // (record and batch numbers are 32-bit: a batch's byte offsets are, so it has fewer than 2^32 records)
    auto batch_of = [&](uint32_t t) -> uint32_t { return gridDim.x * t + (blockIdx.x + 61u * t) % gridDim.x; };
    // FUSE: batches are committed in INDEX order, so they are handed out in index order too: the workgroup takes a ROUND of WPBT
    // consecutive batches with one global ticket and its waves take one each (whoever draws a round's first local ticket fetches
    // the global one; the others wait for it in LDS).  A batch only ever waits for batches with lower indices, which running
    // waves hold: no deadlock whatever else occupies the chip.
    auto fdraw = [&]() -> uint32_t {
        uint32_t b = 0;
        if (lane == 0) {
            const uint32_t t = atomicAdd(&batch_ticket, 1u);
            const uint32_t j = t / WPBT, w = t % WPBT;
            unsigned long long slot;
            if (w == 0) {
                const unsigned int g = atomicAdd(R->fz.ticket, 1u);
                slot = ((unsigned long long)(j + 1u) << 32) | g;
                __hip_atomic_store(&round_slot[j & 15u], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                do { slot = __hip_atomic_load(&round_slot[j & 15u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while ((uint32_t)(slot >> 32) != j + 1u);
            }
            b = (uint32_t)slot * (uint32_t)WPBT + w;
        }
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    };
    const uint32_t n_rec = (uint32_t)fb.n;
    uint32_t cur = FUSE ? fdraw() : batch_of((uint32_t)wave);
    // chunk task `it` of this lane: chunk t = it * 64 + lane of the batch, i.e. chunk t % NW of record t / NW — ten
    // neighbouring lanes cover one read, every load instruction of the wave reads 6.4 whole reads (dense in memory: dealing
    // each lane the chunks of its OWN pair costs a third fewer instructions and runs slower, its loads touch 32 lines
    // each).  Kept as ONE packed word per task — byte offset of the record's LDS row | chunk << 16 — and re-opened inside
    // the loop behind an opaque barrier: left to itself the compiler hoists every derived address (15 64-bit values per
    // lane), runs out of registers and reloads them from scratch in every iteration.
    uint32_t task[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int t = it * WAVE + lane;
        task[it] = (uint32_t)((t / NW) * (WL::STRIDE * 4)) | ((uint32_t)(t % NW) << 16);
    }
    // the descriptor of this lane's read in the NEXT batch is fetched one iteration ahead
    uint32_t m_o = 0, m_l = 0, m_q = 0, m_n = 0;
    // (FUSE: the record's first byte in its text, and in bit 31 whether all four lines end right at their '\n' — QLEN_CONTIG)
    auto start_word = [&](uint32_t r) -> uint32_t {
        const uint32_t n0 = role ? R->fz.name_off2[r] : R->fz.name_off1[r];
        const uint32_t qw = role ? R->fb.qlen2[r] : R->fb.qlen1[r];
        return (n0 & FUSE_POS) | ((qw & QLEN_CONTIG) ? FUSE_WHOLE : 0u);
    };
    // FUSE: a batch's place — its predecessors' sums — is looked up, and its whole records are copied, TWO iterations after its
    // verdicts (behind the NEXT-but-one batch's phase 1): by then every batch drawn before it has published its sums and the
    // look-back does not wait.  (Committed on the spot, every wave ran at the pace of the slowest of the ~3000 batches in flight
    // before it: the kernel took twice as long.)  Two batches wait at any time: the older one (its look-back window is loaded at
    // the top of the iteration, under phase 1's loads) and the newer one; their record lists alternate between L.pend[0 / 1].
    struct FusePending { uint32_t b; unsigned long long wa, wb; int mx; };     // batch (0xffffffff: none), its sums, its longest whole record
    FusePending fz_old{0xffffffffu, 0, 0, 0}, fz_new{0xffffffffu, 0, 0, 0};
    uint32_t fz_iter = 0;                                                      // iterations done: L.pend[fz_iter & 1] is the older batch's / the next to fill
    unsigned long long fz_sa = 0, fz_sb = 0;                                   // the older batch's first look-back window, per lane
    auto fuse_peek = [&](const FusePending& P) {
        const long long idx = (long long)P.b - 1 - lane;
        fz_sa = fz_sb = FUSE_FLAG_P;                                           // before batch 0: prefix 0
        if (P.b != 0xffffffffu && idx >= 0) {
            fz_sa = __hip_atomic_load(&R->fz.state[2 * idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fz_sb = __hip_atomic_load(&R->fz.state[2 * idx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto fuse_finish = [&](const FusePending& P, const uint32_t* pend_list, bool peeked) {
        // ---- the batch's place: its predecessors' sums (decoupled look-back; batches are handed out in index order)
        unsigned long long* const stt = R->fz.state;
        uint32_t bg1 = 0, bg2 = 0, bb1 = 0, bb2 = 0;
        bool dead = false;
        long long j = (AQC_FUSE_ABL & 8) ? -1ll : (long long)P.b - 1;
        unsigned int spins = 0;
        while (j >= 0) {
            const long long idx = j - lane;
            unsigned long long sa = FUSE_FLAG_P, sb = FUSE_FLAG_P;            // before batch 0: prefix 0
            if (peeked) { sa = fz_sa; sb = fz_sb; }
            else if (idx >= 0) {
                sa = __hip_atomic_load(&stt[2 * idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sb = __hip_atomic_load(&stt[2 * idx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            peeked = false;
            // (the two words of a batch are stored one after the other: seen in different states, they are not there yet)
            const unsigned int fl = (sa >> 62) == (sb >> 62) ? (unsigned int)(sa >> 62) : 0u;
            const unsigned long long pnd = __ballot(fl == 0u), pfx = __ballot(fl == 2u);
            const int fp = pfx ? __ffsll((long long)pfx) - 1 : 63;
            const unsigned long long upto = fp == 63 ? ~0ull : ((2ull << fp) - 1ull);
            if (pnd & upto) {
                // somebody before us is not there yet.  Patience has an end: a look-back that cannot finish voids the
                // placement (the host formats the chunk the other way) instead of hanging the device.
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023u) == 0u && (spins > (1u << 22) || __hip_atomic_load(R->fz.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    if (lane == 0) atomicExch(R->fz.abort, 1);
                    dead = true;
                    break;
                }
                continue;
            }
            const bool in = lane <= fp;
            bg1 += (uint32_t)wave_sum_u(in ? (int)((sa >> 31) & 0x7fffffffull) : 0);
            bg2 += (uint32_t)wave_sum_u(in ? (int)(sa & 0x7fffffffull) : 0);
            bb1 += (uint32_t)wave_sum_u(in ? (int)((sb >> 31) & 0x7fffffffull) : 0);
            bb2 += (uint32_t)wave_sum_u(in ? (int)(sb & 0x7fffffffull) : 0);
            if (pfx) break;
            j -= WAVE;
        }
        if (dead) return;
        const uint32_t ig1 = bg1 + (uint32_t)(P.wa >> 31), ig2 = bg2 + (uint32_t)(P.wa & 0x7fffffffull);
        const uint32_t ib1 = bb1 + (uint32_t)(P.wb >> 31), ib2 = bb2 + (uint32_t)(P.wb & 0x7fffffffull);
        if (lane == 0) {
            __hip_atomic_store(&stt[2 * (size_t)P.b], FUSE_FLAG_P | ((unsigned long long)(ig1 & FUSE_POS) << 31) | (ig2 & FUSE_POS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&stt[2 * (size_t)P.b + 1], FUSE_FLAG_P | ((unsigned long long)(ib1 & FUSE_POS) << 31) | (ib2 & FUSE_POS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (P.b + 1 == n_batches) {
                unsigned long long* tt = R->fz.totals;
                tt[0] = ig1; tt[1] = ig2; tt[2] = ib1; tt[3] = ib2;
            }
        }
        if (P.mx == 0 || (AQC_FUSE_ABL & 1)) return;
        // ---- the whole records, copied with fresh loads (the chunks phase 1 loaded were packed and dropped long ago — keeping 60
        // registers of text alive through phase 2 costs the kernel a wave per SIMD; the text is some microseconds old and still in
        // the last-level cache): the batch's 64 records x XW sixteen-byte windows, window w of a record = its bytes
        // [min(16 w, len - 16), + 16), dealt to the lanes window-fastest (a wave instruction moves 1 KiB of neighbouring bytes).
        // Records of up to 384 bytes — 2 x 150 with names of up to 80 — go in ONE round: 24 loads per lane in flight, then 24 stores
        // (this is the point of the iteration where the registers are free: phase 1's are dead, phase 2's not yet alive).
        // (the window numbers only depend on the lane: re-opened per batch behind an opaque barrier, or the compiler keeps every
        //  address of the loop below across the whole batch loop — in scratch)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        uint8_t* const o1 = R->fz.out1 + bg1;
        uint8_t* const o2 = R->fz.out2 + bg2;
        auto copy_pass = [&](auto xw_c, auto u_c) {
            constexpr int XW = decltype(xw_c)::value, U = decltype(u_c)::value;
            for (int t0 = 0; t0 < 2 * PPW * XW; t0 += WAVE * U) {
                uint4 xv[U];
                uint32_t xd[U];                        // offset from the batch's place | file << 31; 0xffffffff: nothing to store
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u * WAVE + lane_o;
                    const int rr = t / XW, w = t % XW;
                    const uint32_t w0 = pend_list[2 * rr], w1 = pend_list[2 * rr + 1];
                    const int n0 = (int)(w0 & FUSE_POS), ln0 = (int)(w1 & 0xffffu);
                    const bool ok = (w0 >> 31) != 0u && 16 * w < ln0;
                    const int o = ok ? min(16 * w, ln0 - 16) : 0;
                    xd[u] = ok ? (((w1 >> 16) + (uint32_t)o) | ((uint32_t)(rr & 1) << 31)) : 0xffffffffu;
                    if (AQC_FUSE_ABL & 4) xv[u] = make_uint4(w0, w1, (uint32_t)t, 0u);
                    else xv[u] = load16u(((rr & 1) ? fb.seq2 : fb.seq1) + (uint32_t)(ok ? n0 + o : 0));
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint8_t* const d = ((xd[u] >> 31) ? o2 : o1) + (xd[u] & FUSE_POS);
                    if (AQC_FUSE_ABL & 2) { if (xd[u] != 0xffffffffu && xv[u].x == 0x12345678u && xv[u].w == 0x9abcdef0u) store16f(d, xv[u]); }
                    else if (xd[u] != 0xffffffffu) store16f(d, xv[u]);
                }
            }
        };
        if (P.mx <= 384) copy_pass(std::integral_constant<int, 24>{}, std::integral_constant<int, AQC_FUSE_UNROLL>{});
        else copy_pass(std::integral_constant<int, 64>{}, std::integral_constant<int, 16>{});
        __builtin_amdgcn_wave_barrier();
    };
    if (cur < n_batches && cur * PPW + p < n_rec) {
        const uint32_t r0 = cur * PPW + p;
        m_o = role ? fb.off2[r0] : fb.off1[r0];
        m_l = lane_len(role ? fb.len2[r0] : fb.len1[r0]);
        m_q = role ? qo2[r0] : qo1[r0];
        if (FUSE) m_n = start_word(r0);
    }
    while (cur < n_batches) {
        // The options are read where they are used, from the kernarg segment through a pointer that is opaque per batch
        // (Rb->cfg.x: an s_load next to the vector work).  As loop invariants the compiler turns every "option > 0" into a
        // 64-bit lane mask held across the loop (two SGPRs per condition, ~15 conditions), runs out of scalar registers and
        // spills them to VGPR lanes — v_writelane / v_readlane + hazard nops were ~6 % of the vector instructions issued.
        FastArgsRare Rb = R;
        asm volatile("" : "+s"(Rb));
        const int o_do_trim = do_trim, o_run_req = run_req, o_r2b = r2b, o_thr4 = thr4, o_poly_word = poly_word;
        __builtin_amdgcn_s_setprio(AQC_PRIO1);
        const uint32_t base = cur * PPW;
        // exactly ONE batch of lookahead (its descriptors travel while this batch is processed): a slow wave never sits
        // on more than one batch the faster waves could have taken
        const uint32_t nxt = FUSE ? fdraw() : batch_of((uint32_t)__builtin_amdgcn_readfirstlane((int)draw()));
        const uint32_t rec = base + p;
        const bool valid = rec < n_rec;
        // ------------------------------------------------------------------ phase 1: load + pack
        if (role == 0) { pr[WL::D_O1] = m_o; pr[WL::D_L1] = m_l; pr[WL::D_Q1] = m_q; pr[WL::D_EXO] = 0; pr[WL::D_LQ] = 0; }
        else { pr[WL::D_O2] = m_o; pr[WL::D_L2] = m_l; pr[WL::D_Q2] = m_q; }
        if (FUSE) {
            pr[role ? WL::D_N2 : WL::D_N1] = m_n;
            fuse_peek(fz_old);                        // (two 8-byte loads per lane, in flight under phase 1's)
        }
        {
            const uint32_t nrec = nxt * PPW + p;
            m_o = m_l = m_q = m_n = 0;
            if (nxt < n_batches && nrec < n_rec) {
                m_o = role ? fb.off2[nrec] : fb.off1[nrec];
                m_l = lane_len(role ? fb.len2[nrec] : fb.len1[nrec]);
                m_q = role ? qo2[nrec] : qo1[nrec];
                if (FUSE) m_n = start_word(nrec);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // each pass: descriptors from LDS, then all 16-byte loads of the pass in flight at once, then the packing.
        // Bytes behind the end of a read (the rest of the text line, the next record) become the pad symbol; every byte
        // that is kept is checked against the alphabet the packed arithmetic takes (A C G T N; qualities < 0x80).  The
        // verdict goes to the record's flag word with one LDS OR per chunk (no compare, no branch).
        uint8_t* const rows = reinterpret_cast<uint8_t*>(&L.planes[0][0]);
        // the task words are re-opened once per batch where five of them fit the registers with what is derived from them
        // (the 2 x 150 variant), else once per use
        constexpr bool TASK_ONCE = ITERS <= 5;
        if (TASK_ONCE) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) asm volatile("" : "+v"(task[it]));
        }
#define AQC_TASK_ROW(q) reinterpret_cast<uint32_t*>(rows + ((q) & 0xffffu))
        // Where the registers allow it (2 x 150: 15 loads, 60 registers that nothing else needs at this point) ALL the
        // 16-byte loads of the batch are issued first — read 1, read 2, read 1's qualities: one memory latency per batch
        // instead of three, the wave has only three neighbours on its SIMD to hide them behind.  Each pass then packs its
        // chunks as they arrive.  The longer / single-end variants load pass by pass.
        constexpr bool ALL_UP = (PAIRED ? 3 : 2) * ITERS * 4 <= (WPBT >= 16 ? 64 : 108);      // (three waves per SIMD have 168 registers each)
        uint4 v1[ITERS], v2[PAIRED ? ITERS : 1], v3[ITERS];
        const bool want_qual = Rb->cfg.unqualified_base_limit > 0;
        auto issue1 = [&]() {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                uint32_t q = task[it];
                if (!TASK_ONCE) asm volatile("" : "+v"(q));
                v1[it] = load16u(fb.seq1 + (uint32_t)(AQC_TASK_ROW(q)[WL::D_O1] + ((q >> 16) << 4)));
            }
        };
        auto issue2 = [&]() {
            const uint8_t* const seq2b = fb.seq2 - 64;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                uint32_t q = task[it];
                if (!TASK_ONCE) asm volatile("" : "+v"(q));
                const uint32_t* const row = AQC_TASK_ROW(q);
                const int st = max((int)row[WL::D_L2] - 16 * ((int)(q >> 16) + 1), -16);
                v2[PAIRED ? it : 0] = load16u(seq2b + (uint32_t)((int)row[WL::D_O2] + st + 64));
            }
        };
        auto issue3 = [&]() {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                uint32_t q = task[it];
                if (!TASK_ONCE) asm volatile("" : "+v"(q));
                v3[it] = load16u(fb.qual1 + (uint32_t)(AQC_TASK_ROW(q)[WL::D_Q1] + ((q >> 16) << 4)));
            }
        };
        issue1();
        if (ALL_UP) {
            if (PAIRED) issue2();
            if (want_qual) issue3();
        }
        {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                uint32_t q = task[it];
                if (!TASK_ONCE) asm volatile("" : "+v"(q));
                uint32_t* const row = AQC_TASK_ROW(q);
                const int c = (int)(q >> 16);
                const uint4 m = mtab[min(max((int)row[WL::D_L1] - 16 * c, 0), 16)];
                uint4 d;
                d.x = bfi(m.x, v1[it].x, PAD1); d.y = bfi(m.y, v1[it].y, PAD1);
                d.z = bfi(m.z, v1[it].z, PAD1); d.w = bfi(m.w, v1[it].w, PAD1);
                if (AQC_ABL & 2) d = v1[it];
                asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));      // (the padded chunk itself feeds both uses below)
                if (!(AQC_ABL & 1)) atomicOr(&row[WL::D_EXO], (not_acgtn(d.x) | not_acgtn(d.y)) | (not_acgtn(d.z) | not_acgtn(d.w)));
                uint32_t lo, e;
                pack_chunk(d, lo, e);
                row[c] = lo;
                row[NW + c] = e;
            }
        }
        if (PAIRED) {
            if (!ALL_UP) issue2();
            // Read 2 is cut into chunks from its END: chunk c holds the bases [L2 - 16 (c + 1), L2 - 16 c), so that reversed
            // (and complemented) it IS word c of reverse_r2 — the stream starts at bit 0 of word 0 whatever the length,
            // and an untrimmed pair needs no alignment pass in phase 2.  (Unaligned 16-byte loads were taken to cost the same as
            // aligned ones — round 6's ablation says they cost this kernel ~9 %, AQC_ABL 128, profiles/r06_copy_window_grid.txt, not acted
            // on here; the chunk of the read's first bases may begin up to 15 bytes before the read, chunks wholly before it
            // are clamped to 16 bytes before: hence the 64-byte bias, every arena has that much readable space in front.)
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                uint32_t q = task[it];
                if (!TASK_ONCE) asm volatile("" : "+v"(q));
                uint32_t* const row = AQC_TASK_ROW(q);
                const int c = (int)(q >> 16);
                // the chunk's LAST nb bytes belong to the read: the first 16 - nb become the pad symbol ('T': complemented it
                // is code 0, like the 'A' padding of read 1)
                const uint4 m = mtab[16 - min(max((int)row[WL::D_L2] - 16 * c, 0), 16)];
                uint4 d;
                d.x = bfi(m.x, PAD2, v2[it].x); d.y = bfi(m.y, PAD2, v2[it].y);
                d.z = bfi(m.z, PAD2, v2[it].z); d.w = bfi(m.w, PAD2, v2[it].w);
                if (AQC_ABL & 2) d = v2[it];
                asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
                if (!(AQC_ABL & 1)) atomicOr(&row[WL::D_EXO], (not_acgtn(d.x) | not_acgtn(d.y)) | (not_acgtn(d.z) | not_acgtn(d.w)));
                uint32_t lo, e;
                pack_chunk(d, lo, e);
                // complement (A<->T, C<->G: flip the high bit of the field), keep N at code 3, then reverse the chunk
                lo = (lo ^ ODD) | e | (e >> 1);
                row[2 * NW + c] = rev2(lo);
                row[3 * NW + c] = __builtin_bitreverse32(e) << 1;
            }
        }
        if (want_qual) {
            if (!ALL_UP) issue3();
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                uint32_t q = task[it];
                if (!TASK_ONCE) asm volatile("" : "+v"(q));
                uint32_t* const row = AQC_TASK_ROW(q);
                const int c = (int)(q >> 16);
                const uint4 m = mtab[min(max((int)row[WL::D_L1] - 16 * c, 0), 16)];
                uint4 d;
                d.x = bfi(m.x, v3[it].x, 0x7f7f7f7fu); d.y = bfi(m.y, v3[it].y, 0x7f7f7f7fu);
                d.z = bfi(m.z, v3[it].z, 0x7f7f7f7fu); d.w = bfi(m.w, v3[it].w, 0x7f7f7f7fu);
                if (AQC_ABL & 2) d = v3[it];
                if (!(AQC_ABL & 1)) atomicOr(&row[WL::D_EXO], ((d.x | d.y) | (d.z | d.w)) & 0x80808080u);
                // byte >= thr  <=>  high bit of ((byte | 0x80) - thr) set   (bytes < 0x80, thr <= 0x7f)
                const uint32_t g0 = ((d.x | 0x80808080u) - o_thr4) & 0x80808080u, g1 = ((d.y | 0x80808080u) - o_thr4) & 0x80808080u;
                const uint32_t g2 = ((d.z | 0x80808080u) - o_thr4) & 0x80808080u, g3 = ((d.w | 0x80808080u) - o_thr4) & 0x80808080u;
                if (!(BARCODE || o_do_trim)) {
                    // the whole read counts: the 0x7f padding is "qualified", chunks wholly behind the read are all padding
                    const int cnt = 16 - (__popc(g0) + __popc(g1) + __popc(g2) + __popc(g3));
                    if (cnt) atomicAdd(&row[WL::D_LQ], (uint32_t)cnt);
                } else {
                    const uint32_t f16 = udot4((g0 ^ 0x80808080u) >> 7, 0x08040201u, 0u) | (udot4((g1 ^ 0x80808080u) >> 7, 0x08040201u, 0u) << 4) |
                                         (udot4((g2 ^ 0x80808080u) >> 7, 0x08040201u, 0u) << 8) | (udot4((g3 ^ 0x80808080u) >> 7, 0x08040201u, 0u) << 12);
                    reinterpret_cast<uint16_t*>(row + WL::LQ0)[c] = (uint16_t)f16;
                }
            }
        }
#undef AQC_TASK_ROW
        __builtin_amdgcn_wave_barrier();
        // FUSE: the batch of two iterations ago gets its place and its whole records are copied (memory work, still at phase 1's
        // priority; the list it was kept in is the one this iteration fills at its end)
        if (FUSE && fz_old.b != 0xffffffffu) {
            fuse_finish(fz_old, L.pend[fz_iter & 1u], true);
            fz_old.b = 0xffffffffu;
        }
        // phase 2 is pure arithmetic on LDS: run it ahead of the waves that are still waiting for their chunk loads (a wave
        // back in phase 1 drops to priority 0 again) — measured -3 % on the 2 x 150 workload, interleaved A/B on one box
        __builtin_amdgcn_s_setprio(AQC_PRIO2);
        PROF(0);

        // ------------------------------------------------------------------ phase 2: lane per read
        // (a mate whose quality line has a length of its own — LEN_IRR in its length word, aqc_kernels.hpp — arrives here with
        //  length 0, see lane_len: the pair goes to the general kernel like an empty read)
        const int L1 = (int)pr[WL::D_L1];
        const int L2 = PAIRED ? (int)pr[WL::D_L2] : 0;
        const bool accum = valid && rec < accum_limit;
        bool defer = valid && (pr[WL::D_EXO] != 0 || L1 > 16 * NW || L2 > 16 * NW || L1 == 0 || (PAIRED && L2 == 0));
        PROF_DEFER(0, defer && role == 0);
        const int Lown = role ? L2 : L1;
        int a_own = 0, len_own = Lown;
        int flag = -1;
        uint32_t bcode = 0;
        // ---- barcode (preprocesser.py:436-452, barcodeprocesser.py): detectBarcode on the packed first 32 bases of the own
        //      read, moveBarcodeToName as a front shift of the view, cleanBarcodeTail with ONE bit-vector Levenshtein
        //      pass per lane (see below)
        if (BARCODE) {
            const int bl = cfg.barcode_length, vl = cfg.barcode_verify_len;
            // Two 32-base windows of the own stream serve both mates: its head A (fields [0, 32)) and its tail B, reversed
            // and complemented (fields [Lown - 32, Lown) read backwards; for reads under 32 bases what lies before the
            // stream is shifted out).  Read 1's stream is the read: its first bases are A, rc(read 1) starts with B.  Read
            // 2's stream is rc(read 2): its first bases as sequenced are B, rc(read 2) starts with A.
            uint32_t a_lo0 = own[0], a_lo1 = own[1], a_e0 = own[NW], a_e1 = own[NW + 1];
            uint32_t b_lo0, b_lo1, b_e0, b_e1;
            {
                const int start = max(Lown - 32, 0);
                win64(own, start, b_lo0, b_lo1);
                win64(own + NW, start, b_e0, b_e1);
                rev2_64(b_lo0, b_lo1);
                rev2_64(b_e0, b_e1);
                b_lo0 ^= ODD; b_lo1 ^= ODD;                 // (an N comes out as code 1; its flag decides)
                const uint32_t sh = (uint32_t)(2 * (start + 32 - Lown));          // > 0 only for reads under 32 bases
                if (sh) {
                    const unsigned long long x = (((unsigned long long)b_lo1 << 32) | b_lo0) >> sh;
                    const unsigned long long y = (((unsigned long long)b_e1 << 32) | b_e0) >> sh;
                    b_lo0 = (uint32_t)x; b_lo1 = (uint32_t)(x >> 32); b_e0 = (uint32_t)y; b_e1 = (uint32_t)(y >> 32);
                }
            }
            const uint32_t f0 = role ? b_lo0 : a_lo0, f1 = role ? b_lo1 : a_lo1, g0 = role ? b_e0 : a_e0, g1 = role ? b_e1 : a_e1;
            const unsigned long long F = ((unsigned long long)f1 << 32) | f0, G = ((unsigned long long)g1 << 32) | g0;
            const unsigned long long V = ((unsigned long long)bc.v1 << 32) | bc.v0, VM = ((unsigned long long)bc.m1 << 32) | bc.m0;
            // diffNumber(seq[s : s + len(verify)], verify) (barcodeprocesser.py:9-14): differing codes or an N
            auto ndiff = [&](int s) -> int {
                const unsigned long long x = (F >> (2 * s)) ^ V;
                return __popcll(((x | (x >> 1)) | (G >> (2 * s + 1))) & VM);
            };
            int b_own = 0;
            if (Lown > vl + bl + 1) {
                if (ndiff(bl) <= 1) b_own = bl;
                else if (ndiff(bl - 1) == 0) b_own = bl - 1;
                else if (ndiff(bl + 1) == 0) b_own = bl + 1;
            }
            if (!PAIRED) {
                if (b_own == 0) flag = AQC_BADBCD1;
                else {
                    bcode = (uint32_t)(b_own - bl + 2);
                    const int rm = vl + bl;                 // single-end moves the design length (preprocesser.py:444)
                    a_own = min(rm, Lown); len_own = max(Lown - rm, 0);
                }
            } else {
                const int b_par = xchg(b_own);
                const int b1 = role ? b_par : b_own, b2 = role ? b_own : b_par;
                if (b1 == 0) flag = AQC_BADBCD1;
                else if (b2 == 0) { flag = AQC_BADBCD2; bcode = (uint32_t)(b1 - bl + 2); }
                else bcode = (uint32_t)(b1 - bl + 2) | ((uint32_t)(b2 - bl + 2) << 4);
                const bool moved = b1 != 0 && b2 != 0;
                if (moved) { a_own = vl + b_own; len_own = Lown - a_own; }
                const int len_par = xchg(len_own);
                // ---- cleanBarcodeTail (barcodeprocesser.py:47-75).  For compLen = bsl .. 1 upstream computes
                //      d = editDistance(own[-compLen:], rc(readStart_partner)[bsl - compLen:]) for both mates and cuts compLen
                //      from both tails at the first compLen where both d <= compLen / 5.  Reversing both strings and
                //      complementing both leaves d unchanged, and turns "drop the first i characters of both" into "take
                //      prefixes": with P = rc(own read)[0 .. bsl) (the tail read backwards, complemented) and
                //      T = readStart_partner (its barcode + the verify sequence, forward), d(compLen) is the cell
                //      D[compLen][n_par - bsl + compLen] of ONE Levenshtein matrix of P against T — a diagonal that the
                //      bit-vector recurrence (Myers / Hyyro, as in edit_distance_lane) yields column by column:
                //      D[x][y] = y + popc(Pv & low(x)) - popc(Mv & low(x)).
                const int n_own = b_own + vl, n_par = b_par + vl, bsl = min(n_own, n_par);
                // readStart of the own read: first b_own bases + verify, as fields; handed to the partner
                const unsigned long long keep = b_own >= 32 ? ~0ull : ((1ull << (2 * b_own)) - 1ull);
                const unsigned long long RS = (F & keep) | (b_own >= 32 ? 0ull : (V << (2 * b_own)));
                const unsigned long long RE = G & keep;
                uint32_t t0 = (uint32_t)xchg((int)(uint32_t)RS), t1 = (uint32_t)xchg((int)(uint32_t)(RS >> 32));
                uint32_t u0 = (uint32_t)xchg((int)(uint32_t)RE), u1 = (uint32_t)xchg((int)(uint32_t)(RE >> 32));
                // the pattern P: rc(own read) from its start.  Read 2's stream already is rc(read 2); read 1's tail is
                // fetched, reversed and complemented.
                const uint32_t p_lo0 = role ? a_lo0 : b_lo0, p_lo1 = role ? a_lo1 : b_lo1, p_e0 = role ? a_e0 : b_e0, p_e1 = role ? a_e1 : b_e1;
                // bit planes of the pattern: code bit 0, code bit 1, N flag; position j at bit j
                const uint32_t P0 = even_bits16(p_lo0) | (even_bits16(p_lo1) << 16);
                const uint32_t P1 = even_bits16(p_lo0 >> 1) | (even_bits16(p_lo1 >> 1) << 16);
                const uint32_t PN = even_bits16(p_e0 >> 1) | (even_bits16(p_e1 >> 1) << 16);
                uint32_t Pv = 0xffffffffu, Mv = 0u, pass = 0u;
                const int delta = n_par - bsl;
                const int ymax = bl + 1 + vl;                     // longest readStart
                // the diagonal's score is carried from cell to cell: D[x][y+1] = D[x-1][y] + 1 - D0[x-1], D0 = Xh | Mv the
                // "diagonal delta is zero" vector of the recurrence; it starts at D[0][delta] = delta.  Columns behind the
                // partner's readStart (y >= n_par) only ever reach x > bsl, which the bound below excludes.
                int dsc = delta;
                const int xmax = min(bsl, min(len_own, len_par) - 1);        // compLen < both reads' lengths
                for (int y = 0; y < ymax; ++y) {
                    const uint32_t tc = t0 & 3u, tn = (u0 >> 1) & 1u;
                    t0 = alignbit(t1, t0, 2); t1 >>= 2;
                    u0 = alignbit(u1, u0, 2); u1 >>= 2;
                    const uint32_t s0 = 0u - (tc & 1u), s1 = 0u - (tc >> 1);
                    const uint32_t Eq = tn ? PN : (~(P0 ^ s0) & ~(P1 ^ s1) & ~PN);
                    const uint32_t Xv = Eq | Mv;
                    const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                    const uint32_t D0 = Xh | Mv;
                    uint32_t Ph = Mv | ~(Xh | Pv);
                    uint32_t Mh = Pv & Xh;
                    Ph = (Ph << 1) | 1u;
                    Mh <<= 1;
                    Pv = Mh | ~(Xv | Ph);
                    Mv = Ph & Xv;
                    const int x = y + 1 - delta;                  // compLen whose cell sits in this column
                    if (x >= 1) {
                        dsc += 1 - (int)((D0 >> (x - 1)) & 1u);
                        if (dsc * 5 <= x && x <= xmax) pass |= 1u << x;
                    }
                    // the score never falls along a diagonal: once 5 * score > xmax on every lane (unrelated tails get there
                    // within a handful of columns) no later compLen can pass
                    if (!__ballot(dsc * 5 <= xmax && x < xmax)) break;
                }
                const uint32_t both = pass & (uint32_t)xchg((int)pass);
                const int cut = (moved && both) ? 31 - __clz((int)both) : 0;
                len_own -= cut;
            }
        }
        // trim (preprocesser.py:455-466): every lane trims its own read
        const int a_pre = a_own, len_pre = len_own;
        if (o_do_trim && flag < 0) {
            int st_ = 0, nl_ = 0;
            trim_view(len_own, role ? R->cfg.trim_front2 : R->cfg.trim_front, role ? R->cfg.trim_tail2 : R->cfg.trim_tail, st_, nl_);
            a_own += st_; len_own = nl_;
        }
        int a_par = 0, len_par = 0;
        if (PAIRED) { a_par = xchg(a_own); len_par = xchg(len_own); }
        int a1 = role ? a_par : a_own, len1 = role ? len_par : len_own;
        int a2 = role ? a_own : a_par, len2 = role ? len_own : len_par;
        if (o_do_trim) {
            // (every exchange runs on all lanes: a DPP read from a masked-off partner returns 0)
            const int xa_pre = PAIRED ? xchg(a_pre) : 0, xl_pre = PAIRED ? xchg(len_pre) : 0;
            if (flag < 0) {
                if (len1 < 5) { flag = AQC_BADTRIM1; a2 = role ? a_pre : xa_pre; len2 = role ? len_pre : xl_pre; }   // read 2 is not trimmed in this case
                else if (PAIRED && len2 < 5) flag = AQC_BADTRIM2;
            }
        }
        // ---- read 1's low-quality count inside its final view
        int lq_cnt = 0;
        if (Rb->cfg.unqualified_base_limit > 0) {
            if (!(BARCODE || o_do_trim)) lq_cnt = (int)pr[WL::D_LQ];
            else {
                int cnt = 0;
                if (role == 0) {
#pragma unroll
                    for (int j = 0; j < (NW + 1) / 2; ++j) {
                        const int lo_b = min(max(a_own - 32 * j, 0), 32), hi_b = min(max(a_own + len_own - 32 * j, 0), 32);
                        const uint32_t mh = hi_b >= 32 ? 0xffffffffu : ((1u << hi_b) - 1u), ml = lo_b >= 32 ? 0xffffffffu : ((1u << lo_b) - 1u);
                        cnt += __popc(pr[WL::LQ0 + j] & mh & ~ml);
                    }
                }
                const int xc = PAIRED ? xchg(cnt) : 0;
                lq_cnt = role ? xc : cnt;
            }
        }
        // ---- normalise the own stream IN PLACE in LDS: afterwards own[j] / own[NW + j] hold bases 16j..16j+15 of
        //      read 1 (role 0) or of reverse_r2 (role 1); everything beyond the read's length is the padding (A C A C ..., N flags zero).  Later
        //      stages fetch the few words they need from LDS instead of pinning 2 x NW registers per lane.
        //      Phase 1 leaves both streams starting at bit 0 and zero behind the full read, so there is nothing to do
        //      unless a trim / barcode stage moved this read's view: read 1's stream starts a_own bases in, reverse_r2
        //      starts as many bases in as were cut from read 2's tail.
        {
            const int b0 = role ? Lown - (a_own + len_own) : a_own;
            if (__ballot(b0 != 0 || len_own != Lown)) {
                const int k0 = b0 >> 4;
                const uint32_t s = (uint32_t)(b0 & 15) * 2;
                uint32_t lo0 = k0 < NW ? own[k0] : 0u, e0 = k0 < NW ? own[NW + k0] : 0u;
#pragma nounroll
                for (int j = 0; j < NW; ++j) {
                    const int i1 = k0 + j + 1;
                    const uint32_t lo1 = i1 < NW ? own[i1] : 0u, e1 = i1 < NW ? own[NW + i1] : 0u;
                    const uint32_t m = base_mask(min(max(len_own - 16 * j, 0), 16));
                    own[j] = bfi(m, alignbit(lo1, lo0, s), PADLO);   // index j <= k0 + j: never overwrites a word still to be read
                    own[NW + j] = alignbit(e1, e0, s) & m;
                    lo0 = lo1; e0 = e1;
                }
            }
        }
        if (role == 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j) pr[WL::GUARD + j] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        PROF(1);

        // ---- bubble (preprocesser.py:469-473)
        if (Rb->cfg.debubble && R->circ.n > 0 && R->fb.aux_ok) {
            bool hit = false;
            const uint8_t* const aux_ok = R->fb.aux_ok;
            const uint8_t ok = valid && flag < 0 ? aux_ok[rec] : (uint8_t)0;
            if (ok == 2) {                                          // int() raises upstream
                int code = AQC_ERR_ARG;
                asm volatile("" : "+v"(code));                      // (built here, not hoisted out of the loop as a register pair)
                atomicCAS(R->st.status, 0, code);
                atomicMin(R->st.err_key, ((unsigned long long)rec << 8) | (unsigned long long)(unsigned int)(-code));
            }
            else if (ok) {
                const int ln = R->fb.aux_lane[rec], tl = R->fb.aux_tile[rec], x = R->fb.aux_x[rec], y = R->fb.aux_y[rec];
                const int n_circ = R->circ.n;
                const int32_t *c_tile = R->circ.tile, *c_lane = R->circ.lane;
                const double *c_x = R->circ.cx, *c_y = R->circ.cy, *c_r = R->circ.cr;
                for (int i = 0; i < n_circ; ++i) {
                    if (c_tile[i] == tl && c_lane[i] == ln) {
                        const double dx = __dsub_rn(c_x[i], (double)x), dy = __dsub_rn(c_y[i], (double)y);
                        if (__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)) < __dmul_rn(c_r[i], c_r[i])) hit = true;
                    }
                }
            }
            if (hit) flag = AQC_BADBBL;
        }
        // ---- length (preprocesser.py:476-479)
        if (flag < 0 && len1 < Rb->cfg.seq_len_req) flag = AQC_BADLEN;
        // ---- polyX (preprocesser.py:482-490): run-length screen per read, exact check by the wave for the few hits
        if (Rb->cfg.poly_size_limit > 0 && !(AQC_ABL & 4)) {
            bool sus = false;
            if (o_run_req < 2) sus = len_own >= Rb->cfg.poly_size_limit;
            else if (o_poly_word) {
                // hasPolyX fires on a span of >= maxPoly - mismatch >= 31 bases that holds at most `mismatch` foreign ones
                // (preprocesser.py:37-50).  Such a span contains a whole aligned 16-base word of the stream, and in that word at
                // most 2 * mismatch of the 15 neighbour pairs differ: one word test — XOR with the stream moved on by a base, count
                // the non-zero fields — instead of the run-length doubling over every word (6 of the kernel's 78 vector
                // instructions per pair).  The exact check below decides, as before.
                int fewest = 15;
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const uint32_t w = own[j];
                    const uint32_t x = w ^ (w >> 2);
                    fewest = min(fewest, (int)__popc((x | (x >> 1)) & 0x15555555u));
                }
                sus = fewest <= 2 * Rb->cfg.allow_mismatch_in_poly && len_own >= Rb->cfg.poly_size_limit;
            } else {
                uint32_t r[NW + 1];
                uint32_t lo0 = own[0], e0 = own[NW];
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const uint32_t lo1 = j + 1 < NW ? own[j + 1] : 0u, e1 = j + 1 < NW ? own[NW + j + 1] : 0u;
                    const uint32_t x = lo0 ^ alignbit(lo1, lo0, 2);
                    const uint32_t ex = e0 ^ alignbit(e1, e0, 2);
                    // base i equals base i+1 (behind the read the stream alternates: at most the read's last base pairs
                    // with the padding, which lengthens a run by one — the exact check decides)
                    r[j] = ~(((x << 1) | x | ex)) & ODD;
                    lo0 = lo1; e0 = e1;
                }
                r[NW] = 0;
                int covered = 1;
                while (covered < o_run_req - 1) {
                    const int step = min(min(covered, o_run_req - 1 - covered), 15);
#pragma unroll
                    for (int j = 0; j < NW; ++j) r[j] &= alignbit(r[j + 1], r[j], 2 * step);
                    covered += step;
                }
                uint32_t any = 0;
#pragma unroll
                for (int j = 0; j < NW; ++j) any |= r[j];
                sus = any != 0 && len_own >= Rb->cfg.poly_size_limit;
            }
            if (AQC_ABL & 4) sus = false;
            bool poly = false;
            unsigned long long todo = __ballot(valid && !defer && flag < 0 && sus);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int lp = PAIRED ? l >> 1 : l, lr = PAIRED ? l & 1 : 0;
                const int ta = __builtin_amdgcn_readlane(a_own, l), tl = __builtin_amdgcn_readlane(len_own, l);
                // hasPolyX runs on the read as sequenced (read 2 is NOT reverse-complemented)
                const uint8_t* src = (lr ? fb.seq2 + L.planes[lp][WL::D_O2] : fb.seq1 + L.planes[lp][WL::D_O1]) + ta;
                stage(L.stage, src, tl);
                __builtin_amdgcn_wave_barrier();
                const int px = has_polyx_wave(L.stage, tl, Rb->cfg.poly_size_limit, Rb->cfg.allow_mismatch_in_poly);
                __builtin_amdgcn_wave_barrier();
                if (px != 0 && lane == l) poly = true;
            }
            // (the exchange must run on all lanes: a DPP read from a masked-off partner returns 0)
            const bool poly_par = PAIRED ? xchg_pred(poly) : false;
            const bool poly_pair = poly || poly_par;
            if (flag < 0 && poly_pair) flag = AQC_BADPOL;
        }
        PROF(2);
        // ---- low quality: read 1 only (preprocesser.py:498)
        if (flag < 0 && Rb->cfg.unqualified_base_limit > 0 && lq_cnt > Rb->cfg.unqualified_base_limit) flag = AQC_BADLQC;
        // ---- N (preprocesser.py:504-512)
        if (Rb->cfg.n_base_limit > 0 && !(AQC_ABL & 32)) {
            int n_own = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) n_own += __popc(own[NW + j]);
            const int n_par = PAIRED ? xchg(n_own) : 0;
            if (flag < 0 && (n_own > Rb->cfg.n_base_limit || n_par > Rb->cfg.n_base_limit)) flag = AQC_BADNCT;
        }
        PROF(3);
        // ---- overlap (util.py:158-212) --------------------------------------------------------------
        int offset = 0, ovl = 0, dist = 0, ovl0 = -1, dist_final = -1, n_edits = 0;
        int c_adapter_base = 0, c_adapter_read = 0, c_overlapped = 0, c_corrected = 0, c_masked = 0, c_skipped = 0, c_read_corrected = 0;
        int em0 = -1, em1 = -1, em2 = -1, walk_a = 0;
        uint32_t EW0 = 0, EW1 = 0, EW2 = 0, EW3 = 0;       // the walk's edits as they go into the result (bytes 16..30)
        bool walk_pair = false, walker = false;
        if (PAIRED && !Rb->cfg.no_overlap) {
            // own candidates: offsets c = 0 .. len_own - 31, the own stream moving over the partner's prefix
            const int n_own = len_own > 30 ? len_own - 30 : 0;
            bool scan = valid && !defer && flag < 0 && !(AQC_ABL & 8);
            {
                // the 16-base prefix test needs 16 columns on every diagonal
                const bool short16 = scan && n_own > 0 && len_par < 16;
                const bool short16_par = xchg_pred(short16);
                PROF_DEFER(1, (short16 || short16_par) && !defer && role == 0);
                if (short16 || short16_par) { defer = true; scan = false; }
            }
            bool i_found_it = false;    // this lane's stream moves on the accepted diagonal
            int from = 0;               // first own candidate still to be examined
            bool found = false;
            int f_off = 0, f_len = 0, f_tot = 0, f_p0 = 0, f_p1 = 0, f_p2 = 0;   // accepted candidate + its first mismatch columns
            uint32_t f_wm = 0;          // (cooperative verification: which 16-base words of the accepted diagonal hold a mismatch)
            const uint32_t F = par[0];                          // partner's first 16 bases
            // mismatches of one diagonal — the own stream from base c on against the partner's first QL bases: their number,
            // how many lie in columns 0..49, and the columns of the first three (the correction walk needs them)
            // (results come back by value, packed: columns 9 bits each | total << 27 | c50 << 36 — with reference parameters the
            //  compiler turns the three columns into an indexed private array, i.e. scratch memory)
            auto diag_mismatches = [&](int c, int QL) __attribute__((always_inline)) -> unsigned long long {
                const int k = c >> 4;
                const uint32_t s = (uint32_t)(c & 15) * 2;
                uint32_t lo0 = own[k], e0 = own[NW + k];
                int rem = QL, tot = 0, c50 = 0, vp0 = 0, vp1 = 0, vp2 = 0;
#pragma nounroll
                for (int j = 0; 16 * j < QL; ++j) {
                    const bool in = k + j + 1 < NW;
                    const uint32_t lo1 = in ? own[k + j + 1] : 0u, e1 = in ? own[NW + k + j + 1] : 0u;
                    uint32_t mm = mm_word(lo0, lo1, e0, e1, s, par[j], par[NW + j], rem);
                    if (j < 4) c50 += __popc(j < 3 ? mm : (mm & 0xFu));      // columns 0..49 (wave-uniform branch)
                    while (mm != 0 && tot < 3) {
                        const int col = 16 * j + ((__ffs((int)mm) - 1) >> 1);
                        if (tot == 0) vp0 = col; else if (tot == 1) vp1 = col; else vp2 = col;
                        tot++;
                        mm &= mm - 1;
                    }
                    tot += __popc(mm);
                    rem = max(rem - 16, 0);
                    lo0 = lo1; e0 = e1;
                }
                return (unsigned long long)(uint32_t)(vp0 | (vp1 << 9) | (vp2 << 18)) | ((unsigned long long)(uint32_t)tot << 27) | ((unsigned long long)(uint32_t)c50 << 36);
            };
            while (true) {
                const int wmax = wave_max_i(scan && !found ? n_own : 0);
                if (wmax == 0) break;
                int s0 = NONE_CAND, s1 = NONE_CAND, s2 = NONE_CAND;   // first three prefix survivors >= from
                const bool live = scan && !found;
                // sixteen diagonals per 32-bit word pair; eight are evaluated back to back (independent
                // alignbit / xor / popcount chains) per branch
                uint32_t w0 = own[0];
#pragma nounroll
                for (int k = 0; 16 * k < wmax; ++k) {
                    const uint32_t w1 = (k + 1 < NW) ? own[k + 1] : 0u;
#pragma unroll
                    for (int r0 = 0; r0 < 16; r0 += 8) {
                        // popcount - 5 (the popcount instruction adds its second operand): negative for a survivor, and the
                        // OR of the eight keeps a sign bit — three v_or3 instead of a seven-deep min tree
                        int32_t cnt[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) cnt[q] = (int32_t)__popc(alignbit(w1, w0, 2 * (r0 + q)) ^ F) - 5;
                        const int32_t any = ((cnt[0] | cnt[1] | cnt[2]) | (cnt[3] | cnt[4] | cnt[5])) | (cnt[6] | cnt[7]);
                        if (__ballot(any < 0)) {
                            // Not rare at all: half the pairs of a 2 x 150 run with ~30-base overlaps have their true offset
                            // in the scan's last two words, so four of the sixteen half-steps come through here in nearly every
                            // batch (measured, round 4: a rolled loop over the eight diagonals with a branch each cost ~100 vector
                            // and ~120 scalar instructions per visit, a fifth of the kernel).  The eight sign bits become a mask
                            // with one v_alignbit each, the candidates this lane still wants ([from, n_own)) a v_bfm, and the
                            // survivors — one per lane as a rule — are taken off the mask lowest first.
                            uint32_t m8 = 0;
#pragma unroll
                            for (int q = 7; q >= 0; --q) m8 = alignbit(m8, (uint32_t)cnt[q], 31);
                            const int base8 = 16 * k + r0;
                            const int lo_q = min(max(from - base8, 0), 8), hi_q = min(max(n_own - base8, 0), 8);
                            const uint32_t want = live && hi_q > lo_q ? ((1u << hi_q) - 1u) & ~((1u << lo_q) - 1u) : 0u;
                            m8 &= want;
                            while (__ballot(m8 != 0)) {
                                if (m8 != 0) {
                                    const int c = base8 + __builtin_ctz(m8);
                                    m8 &= m8 - 1;
                                    if (s0 == NONE_CAND) s0 = c; else if (s1 == NONE_CAND) s1 = c; else if (s2 == NONE_CAND) s2 = c;
                                }
                            }
                        }
                    }
                    w0 = w1;
                }
                PROF(4);
                // exact verification of up to three survivors per lane, in order (util.py:177-184 / 200-207)
#if AQC_COOP_VERIFY
                // By the WHOLE wave: a diagonal is QL / 16 words long — two for the ~30-base overlaps of a long insert, ten for an
                // adapter read-through — and a lane that walks its own diagonal keeps the other 63 waiting for the longest one
                // (measured: scan + verification 36 of the kernel's 78 vector instructions per pair, the scan itself 16).  So the
                // (survivor, word) pairs of a round become tasks dealt to consecutive lanes: the owners' word counts are scanned,
                // each owner marks its first task lane in the wave's staging bytes, a running maximum tells every lane whose word
                // it has, the owner's candidate comes over the LDS crossbar (ds_bpermute), and the word's mismatch count goes to
                // the owner with one LDS atomic (into the row's low-quality words, which nobody needs any more at this point).
                // The mismatch COLUMNS the correction walk needs are looked up later, by the lanes that walk, in the words this
                // pass has flagged.
#pragma nounroll
                for (int v = 0; v < 3; ++v) {
                    const int c = v == 0 ? s0 : v == 1 ? s1 : s2;
                    const bool check = live && !found && c != NONE_CAND;
                    if (!__ballot(check)) break;
                    const int QL = min(len_own - c, len_par);
                    const int nwv = check ? (QL + 15) >> 4 : 0;
                    const int desc = c | (QL << 10);
                    uint32_t* const acc = pr + WL::LQ0 + 2 * role;
                    bool pend = check;
                    while (__ballot(pend)) {
                        // owners whose words fit the 64 lanes of this sub-round (a prefix of the pending ones)
                        const int incl = wave_incl_sum(pend ? nwv : 0, lane);
                        const bool now = pend && incl <= WAVE;
                        const unsigned long long nowm = __ballot(now);
                        const int n_task = __builtin_amdgcn_readlane(incl, 63 - __builtin_clzll(nowm));
                        L.stage[lane] = 0;
                        __builtin_amdgcn_wave_barrier();
                        if (now) {
                            L.stage[incl - nwv] = (uint8_t)(lane + 1);
                            acc[0] = 0; acc[1] = 0;
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int mark = (int)L.stage[lane];
                        const int key = wave_incl_max(mark ? ((lane << 8) | mark) : 0, lane);
                        const bool task = lane < n_task;
                        const int owner = task ? (key & 0xff) - 1 : 0;
                        const int j = task ? lane - (key >> 8) : 0;
                        const int od = __builtin_amdgcn_ds_bpermute(owner << 2, desc);
                        const int oc = od & 0x3ff, oq = od >> 10;
                        uint32_t* const orow = L.planes[owner >> 1];
                        const uint32_t* const oown = orow + ((owner & 1) ? 2 * NW : 0);
                        const uint32_t* const opar = orow + ((owner & 1) ? 0 : 2 * NW);
                        const int k = (oc >> 4) + j;
                        const uint32_t sh = (uint32_t)(oc & 15) * 2;
                        const bool in1 = k + 1 < NW;
                        const uint32_t lo0 = oown[min(k, NW - 1)], lo1 = in1 ? oown[min(k + 1, NW - 1)] : 0u;
                        const uint32_t e0 = oown[NW + min(k, NW - 1)], e1 = in1 ? oown[NW + min(k + 1, NW - 1)] : 0u;
                        const uint32_t mm = task ? mm_word(lo0, lo1, e0, e1, sh, opar[min(j, NW - 1)], opar[NW + min(j, NW - 1)], min(max(oq - 16 * j, 0), 16)) : 0u;
                        const uint32_t cnt_all = (uint32_t)__popc(mm);
                        const uint32_t cnt50 = j < 3 ? cnt_all : j == 3 ? (uint32_t)__popc(mm & 0xFu) : 0u;      // columns 0..49
                        if (mm) {
                            uint32_t* const oacc = orow + WL::LQ0 + 2 * (owner & 1);
                            atomicAdd(&oacc[0], cnt_all | (cnt50 << 16));
                            atomicOr(&oacc[1], 1u << j);
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (now) {
                            const uint32_t a0 = acc[0];
                            const int tot = (int)(a0 & 0xffffu), c50 = (int)(a0 >> 16);
                            if (tot < 3 || (c50 < 3 && QL >= 52)) { found = true; f_off = c; f_len = QL; f_tot = tot; f_wm = acc[1]; }
                            pend = false;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
#else
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int c = v == 0 ? s0 : v == 1 ? s1 : s2;
                    const bool check = live && !found && c != NONE_CAND;
                    if (!__ballot(check)) break;
                    const int QL = min(len_own - c, len_par);
                    if (check) {
                        const unsigned long long dm = diag_mismatches(c, QL);
                        const int tot = (int)((dm >> 27) & 0x1ffu), c50 = (int)(dm >> 36);
                        if (tot < 3 || (c50 < 3 && QL >= 52)) {
                            found = true; f_off = c; f_len = QL; f_tot = tot;
                            f_p0 = (int)(dm & 0x1ffu); f_p1 = (int)((dm >> 9) & 0x1ffu); f_p2 = (int)((dm >> 18) & 0x1ffu);
                        }
                    }
                }
#endif
                PROF(5);
                // a lane whose three survivors all failed and that may have more continues after the third one
                const bool more = live && !found && s2 != NONE_CAND;
                if (more) from = s2 + 1;
                if (live && !found && s2 == NONE_CAND) scan = false;   // exhausted
                if (!__ballot(more)) break;
            }
            PROF(5);
            // ---- merge: an accepted forward offset wins over any reverse offset (util.py:170-212)
            {
                const bool x_found = xchg_pred(found);
                const int x_off = xchg(f_off), x_len = xchg(f_len), x_tot = xchg(f_tot);
                const bool fw = role ? x_found : found, rv = role ? found : x_found;
                i_found_it = fw ? role == 0 : (rv && role == 1);
                if (fw) { offset = role ? x_off : f_off; ovl = role ? x_len : f_len; dist = role ? x_tot : f_tot; }
                else if (rv) { offset = -(role ? f_off : x_off); ovl = role ? f_len : x_len; dist = role ? f_tot : x_tot; }
            }
            // ---- post-processing (preprocesser.py:516-617), identical on both lanes of the pair
            const bool reached = valid && !defer && flag < 0;
            if (reached) {
                ovl0 = ovl;
                if (offset < 0 && ovl > 30) {
                    // adapter read-through: both reads are cut to overlap_len and util.overlap runs again.  When
                    // overlap_len == len2 - |offset| the second call's first candidate (offset 0) is exactly the
                    // diagonal just accepted, so it returns (0, overlap_len, diff) again; otherwise defer.
                    PROF_DEFER(2, reached && offset < 0 && ovl > 30 && ovl != len2 + offset && role == 0);
                    if (ovl != len2 + offset) defer = true;
                    else {
                        c_adapter_base = -2 * offset; c_adapter_read = 1;
                        walk_a = -offset;
                        len1 = ovl; len2 = ovl; offset = 0;
                        if (len1 < Rb->cfg.seq_len_req) { flag = AQC_BADLEN; ovl = 0; dist = 0; }
                    }
                }
            }
            if (reached && !defer && flag < 0) {
                dist_final = dist;
                if (dist > 3) flag = AQC_BADDIFF;
                else if (ovl > 30) c_overlapped = 1;
            }
            // ---- correction walk (preprocesser.py:563-598).  The walk is anchored at the tails: b1 = r1[len1 - ovl + o],
            //      b2 = complement(r2[-o-1]).  Whenever overlap_len == len1 - offset (always after an adapter cut) those
            //      are exactly the columns of the accepted diagonal, whose first three mismatches the verification
            //      already located; otherwise (read 2 shorter than the rest of read 1) the pair is deferred.
            walk_pair = reached && !defer && flag < 0 && c_overlapped && dist > 0;
            // Read 2 shorter than the rest of read 1 (overlap_len == len2 < len1 - offset, App. B-6): the walk then runs along
            // ANOTHER diagonal than the one util.overlap accepted — read 1 from len1 - len2 on against all of reverse_r2 — and
            // handles the first `distance` mismatches it meets there; fewer than that on the whole diagonal => BADMISMATCH
            // (preprocesser.py:563-617).  The read-1 lane looks that diagonal up with the same routine as the scan.
            int w_tot = dist;
            {
                const bool shifted = walk_pair && !c_adapter_read && ovl != len1 - offset;
                PROF_DEFER(3, shifted && role == 0);
                if (__ballot(shifted)) {
                    if (shifted && role == 0) {
                        const unsigned long long dm = diag_mismatches(len1 - ovl, ovl);
                        w_tot = (int)((dm >> 27) & 0x1ffu);
                        f_p0 = (int)(dm & 0x1ffu); f_p1 = (int)((dm >> 9) & 0x1ffu); f_p2 = (int)((dm >> 18) & 0x1ffu);      // (the accepted diagonal's columns are not needed any more)
                    }
                }
            }
            const int w_n = min(dist, w_tot);                 // mismatches the walk handles
            walker = walk_pair && i_found_it && !(AQC_ABL & 16);
#if AQC_COOP_VERIFY
            {
                // the columns of the walker's first w_n mismatches: only the words the verification flagged are looked at (a pair
                // walked along a diagonal of its own got its columns from diag_mismatches above)
                const bool lookup = walker && !(walk_pair && !c_adapter_read && ovl != len1 - offset);
                uint32_t wm = lookup ? f_wm : 0u;
                int have = 0;
                while (__ballot(wm != 0 && have < w_n)) {
                    const bool act = wm != 0 && have < w_n;
                    const int j = act ? __builtin_ctz(wm) : 0;
                    wm &= wm - 1;
                    const int k = (f_off >> 4) + j;
                    const uint32_t sh = (uint32_t)(f_off & 15) * 2;
                    const bool in1 = k + 1 < NW;
                    const uint32_t lo0 = own[min(k, NW - 1)], lo1 = in1 ? own[min(k + 1, NW - 1)] : 0u;
                    const uint32_t e0 = own[NW + min(k, NW - 1)], e1 = in1 ? own[NW + min(k + 1, NW - 1)] : 0u;
                    uint32_t mm = act ? mm_word(lo0, lo1, e0, e1, sh, par[min(j, NW - 1)], par[NW + min(j, NW - 1)], min(max(f_len - 16 * j, 0), 16)) : 0u;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        if (mm != 0 && have < 3) {
                            const int col = 16 * j + ((__ffs((int)mm) - 1) >> 1);
                            if (have == 0) f_p0 = col; else if (have == 1) f_p1 = col; else f_p2 = col;
                            ++have;
                            mm &= mm - 1;
                        }
                    }
                }
            }
#endif
            if (__ballot(walker)) {
                // Straight-line and in the streams' 2-bit codes (A 0, C 1, T 2, G 3; complement = code ^ 2): lanes that do not
                // walk run along with column 0 and switch their results off.
                const int shift1 = c_adapter_read ? 0 : len1 - ovl;        // read-1 position of walk column o: shift1 + o
                const int shift2 = c_adapter_read ? walk_a : 0;            // reverse_r2 position of column o
                const uint32_t* s1w = pr;                                  // read-1 stream
                const uint32_t* s2w = pr + 2 * NW;                         // reverse_r2 stream
                const bool no_corr = Rb->cfg.no_correction != 0, mask_mm = Rb->cfg.mask_mismatch != 0;
                // byte offsets of column 0's qualities (32-bit, from the arenas' bases: one add per load, no 64-bit address math)
                const uint32_t qa0 = walker ? pr[WL::D_Q1] + (uint32_t)(a1 + (len1 - ovl)) : 0u;
                const uint32_t qb0 = walker ? pr[WL::D_Q2] + (uint32_t)(a2 + (len2 - 1)) : 0u;
                // (a second / third mismatch is walked only when some pair of the batch has one: with ~0.3 mismatches per
                //  overlapping pair most batches stop after the first — the three steps used to run unconditionally, 8 of the
                //  kernel's 78 vector instructions per pair)
                const bool any_q[3] = {true, __ballot(walker && w_n > 1) != 0, __ballot(walker && w_n > 2) != 0};
                int wcol[3] = {0, 0, 0};
                uint32_t wq1[3] = {0, 0, 0}, wq2[3] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (!any_q[q]) continue;
                    wcol[q] = walker && q < w_n ? (q == 0 ? f_p0 : q == 1 ? f_p1 : f_p2) : 0;
                    wq1[q] = fb.qual1[qa0 + (uint32_t)wcol[q]];
                    wq2[q] = fb.qual2[qb0 - (uint32_t)wcol[q]];
                }
                uint32_t El0 = 0, El1 = 0, El2 = 0, Eh0 = 0, Eh1 = 0, Eh2 = 0;      // edits: column | kind << 16 | base << 24, quality
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (!any_q[q]) continue;
                    const bool on = walker && q < w_n;
                    const int oo = wcol[q];
                    const int x1 = on ? shift1 + oo : 0, x2 = on ? shift2 + oo : 0;
                    const uint32_t h1 = (uint32_t)(x1 & 15) * 2u, h2 = (uint32_t)(x2 & 15) * 2u;
                    const uint32_t c1 = (s1w[x1 >> 4] >> h1) & 3u, n1 = (s1w[NW + (x1 >> 4)] >> (h1 + 1u)) & 1u;
                    const uint32_t c2 = (s2w[x2 >> 4] >> h2) & 3u, n2 = (s2w[NW + (x2 >> 4)] >> (h2 + 1u)) & 1u;      // c2: complement(r2 base)
                    const uint32_t qa = wq1[q], qb = wq2[q];
                    const bool r2_wrong = qa >= 63u && qb <= 47u;                   // trust read 1 (preprocesser.py:571: q1 >= 30, q2 <= 14)
                    const bool r1_wrong = !r2_wrong && qb >= 63u && qa <= 47u;      // trust read 2 (:579)
                    const bool trust = on && (r2_wrong || r1_wrong);
                    // the base that is written (and its N flag) / the base it replaces, as codes:
                    // read 2 wrong: r2 := complement(b1), was complement(b2');  read 1 wrong: r1 := b2', was b1
                    const uint32_t X = r2_wrong ? (c1 ^ 2u) : c2, Y = r2_wrong ? (c2 ^ 2u) : c1, nX = r2_wrong ? n1 : n2;
                    // error matrix cell (preprocesser.py:573,581): ALL_BASES index A T C G = 0 1 2 3 of either base
                    const int em = (trust && !(n1 | n2)) ? (int)((((0xD8u >> (2u * X)) & 3u) << 2) | ((0xD8u >> (2u * Y)) & 3u)) : -1;
                    const bool fix = trust && !no_corr;
                    const bool mask = on && !fix && mask_mm;
                    const uint32_t kind = fix ? (r2_wrong ? (uint32_t)AQC_EDIT_FIX_R2 : (uint32_t)AQC_EDIT_FIX_R1) : (uint32_t)AQC_EDIT_MASK;
                    const uint32_t base = fix ? (nX ? (uint32_t)'N' : ((0x47544341u >> (8u * X)) & 0xffu)) : 0u;       // code -> A C T G
                    const uint32_t el = (uint32_t)oo | (kind << 16) | (base << 24);
                    const uint32_t eh = fix ? (r2_wrong ? qa : qb) : (uint32_t)'!';
                    c_corrected += fix ? 1 : 0;
                    c_masked += mask ? 1 : 0;
                    c_skipped += (on && !fix && !mask) ? 1 : 0;
                    if (q == 0) em0 = em; else if (q == 1) em1 = em; else em2 = em;
                    const bool put = fix || mask;
                    if (put && n_edits == 0) { El0 = el; Eh0 = eh; }
                    if (put && n_edits == 1) { El1 = el; Eh1 = eh; }
                    if (put && n_edits == 2) { El2 = el; Eh2 = eh; }
                    n_edits += put ? 1 : 0;
                }
                if (walker && w_tot < dist) {
                    // the walk ran out of mismatches before it had handled `distance` of them: the pair goes to bad/ with the
                    // edits made so far, the correction counters do not move (preprocesser.py:600-612)
                    flag = AQC_BADMISMATCH;
                    em0 = em1 = em2 = -1;
                    c_corrected = c_masked = c_skipped = 0;
                }
                if (walker && c_corrected > 0) c_read_corrected = 1;
                // the three 40-bit edits, packed into the result's upper 16 bytes
                EW0 = El0; EW1 = Eh0 | (El1 << 8); EW2 = (El1 >> 24) | (Eh1 << 8) | (El2 << 16); EW3 = (El2 >> 16) | (Eh2 << 16);
            }
        }
        if (flag < 0) flag = AQC_GOOD;
        PROF(6);

        // ------------------------------------------------------------------ results + counters
        // one lane per pair writes: the walker if there was a walk (it holds the edits), the read-1 lane otherwise
        const bool mine = valid && !defer && (walk_pair ? walker : role == 0);
        if (mine) {
            // struct aqc_result (packed, 32 bytes) assembled in registers and written as two 16-byte stores
            uint4 lo, hi;
            lo.x = (uint32_t)flag | ((uint32_t)n_edits << 8) | ((uint32_t)(a1 & 0xffff) << 16);
            lo.y = (uint32_t)(len1 & 0xffff) | ((uint32_t)(a2 & 0xffff) << 16);
            lo.z = (uint32_t)(len2 & 0xffff) | ((uint32_t)(offset & 0xffff) << 16);
            lo.w = (uint32_t)(ovl & 0xffff) | ((uint32_t)(dist & 0xffff) << 16);
            hi.x = EW0; hi.y = EW1; hi.z = EW2; hi.w = EW3 | (bcode << 24);   // byte 31: barcode nibbles
            uint4* out = reinterpret_cast<uint4*>(results + rec);
            out[0] = lo;
            out[1] = hi;
        }
        const bool cnt = mine && accum;
        if (cnt) {
            R0 += 1u;
            R1 += (uint32_t)(L1 + o_r2b * L2);
            if (flag == AQC_GOOD) { R0 += 1u << 8; R1 += (uint32_t)(len1 + o_r2b * len2) << 16; }
            else atomicAdd(&acc.counters[AQC_C_FLAG0 + flag], 1ull);
            if (PAIRED) {
                R2 += (uint32_t)c_adapter_base; R0 += (uint32_t)c_adapter_read << 16;
                if (c_overlapped) { R0 += 1u << 24; R2 += (uint32_t)ovl << 16; R3 += (uint32_t)dist; }
                R3 += ((uint32_t)c_read_corrected << 8) + ((uint32_t)c_corrected << 16) + ((uint32_t)c_masked << 24); R4 += (uint32_t)c_skipped;
                if (em0 >= 0) atomicAdd(&acc.counters[AQC_C_ERR_MATRIX0 + em0], 1ull);
                if (em1 >= 0) atomicAdd(&acc.counters[AQC_C_ERR_MATRIX0 + em1], 1ull);
                if (em2 >= 0) atomicAdd(&acc.counters[AQC_C_ERR_MATRIX0 + em2], 1ull);
            }
        }
        if (PAIRED) {
            // histograms (preprocesser.py:517,536): the dominant bins (no overlap, distance 0) are counted with a ballot
            const unsigned long long z0 = __ballot(cnt && ovl0 == 0), d0 = __ballot(cnt && dist_final == 0);
            if (lane == 0) {
                if (z0) atomicAdd(&acc.ovl_hist[0], (unsigned int)__popcll(z0));
                if (d0) atomicAdd(&acc.dist_hist[0], (unsigned int)__popcll(d0));
            }
            if (cnt && ovl0 > 0) atomicAdd(&acc.ovl_hist[ovl0], 1u);
            if (cnt && dist_final > 0) atomicAdd(&acc.dist_hist[min(dist_final, AQC_QC_COLS - 1)], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        PROF(7);
        // ------------------------------------------------------------------ FUSE: place every record, copy the whole ones
        if (FUSE) {
            // (the walker holds the pair's last word on verdict and edits; everything else of the post-processing is identical
            //  on both lanes of a pair)
            // (every lane takes what the lane that wrote the pair's result holds)
            const int pf = xchg(flag), pn = xchg(n_edits), pa1 = xchg(a1), pl1 = xchg(len1), pa2 = xchg(a2), pl2 = xchg(len2);
            const bool take = walk_pair ? !walker : role == 1;
            const int flagF = take ? pf : flag, neF = take ? pn : n_edits;
            const int fa1 = take ? pa1 : a1, fl1 = take ? pl1 : len1, fa2 = take ? pa2 : a2, fl2 = take ? pl2 : len2;
            const uint32_t nw = pr[role ? WL::D_N2 : WL::D_N1];
            const int startO = (int)(nw & FUSE_POS);
            const int Qo = (int)pr[role ? WL::D_Q2 : WL::D_Q1];
            const int endO = Qo + Lown + 1;                       // one past the '\n' of the quality line
            const int recLen = endO - startO;
            const int st = role ? fa2 : fa1, ln = role ? fl2 : fl1;
            const bool live = valid && !defer;
            // what this scheme cannot place: a pair the general kernel still has to decide, a record that is not four lines of
            // contiguous text (stripped blanks: the writer rebuilds it from pieces), records beyond the packed sums' range
            const bool odd = valid && (defer || !(nw >> 31) || recLen < 48 || recLen > 1000);
            // (a pair the correction walk edited is still its own bytes but for up to six of them: copied here like the others — a
            //  dense stream of stores — and patched byte by byte by the writer's plan pass, which sees FUSE_PATCH.  Copying those
            //  records later, 8 % of them scattered over the stream, ran at a third of the dense rate.)
            const bool whole = live && !odd && flagF == AQC_GOOD && st == 0 && ln == Lown;
            // bytes the record adds to its stream: its own bytes, or name + strand line + 4 newlines + the final view twice
            // (+ the flag text in the name of a bad record: "@" + FLAG + name[1:], preprocesser.py:213-219)
            const int flen = (int)((0xB76666688774ull >> (4 * (flagF & 15))) & 15ull);
            const int szr = whole ? recLen : (recLen - 2 * Lown) + 2 * ln + (flagF == AQC_GOOD ? 0 : flen);
            const uint32_t gsz = (live && !odd && flagF == AQC_GOOD) ? (uint32_t)szr : 0u, bsz = (live && !odd && flagF != AQC_GOOD) ? (uint32_t)szr : 0u;
            const int packed = (int)(gsz | (bsz << 16));
            const int s_lo = role ? 0 : packed, s_hi = role ? packed : 0;      // file 1's sums in one word, file 2's in the other
            const int i_lo = wave_incl_sum(s_lo, lane), i_hi = wave_incl_sum(s_hi, lane);
            const uint32_t t_lo = (uint32_t)__builtin_amdgcn_readlane(i_lo, 63), t_hi = (uint32_t)__builtin_amdgcn_readlane(i_hi, 63);
            const uint32_t ex = (uint32_t)(role ? i_hi - s_hi : i_lo - s_lo);
            const uint32_t ag1 = t_lo & 0xffffu, ab1 = t_lo >> 16, ag2 = t_hi & 0xffffu, ab2 = t_hi >> 16;
            if (__ballot(odd) && lane == 0) atomicExch(R->fz.abort, 1);
            // ---- the batch's sums go out at once (state A); its place — its predecessors' sums — is looked up and its whole
            //      records are copied two iterations later (fuse_finish, above)
            const unsigned long long wa = ((unsigned long long)ag1 << 31) | ag2, wb = ((unsigned long long)ab1 << 31) | ab2;
            if (lane == 0) {
                __hip_atomic_store(&R->fz.state[2 * (size_t)cur], FUSE_FLAG_A | wa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&R->fz.state[2 * (size_t)cur + 1], FUSE_FLAG_A | wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // every record's offset inside its batch's share of its stream; the host-side writer adds the batch's place
            if (live && !odd) (role ? R->fz.fstate2 : R->fz.fstate1)[rec] = (flagF == AQC_GOOD ? (ex & 0xffffu) : (ex >> 16)) | (whole ? FUSE_WHOLE : 0u) | ((whole && neF > 0) ? FUSE_PATCH : 0u);
            const int mx = wave_max_i(whole ? recLen : 0);
            uint32_t* const pl = L.pend[fz_iter & 1u];             // (the batch that had this list was finished behind this iteration's phase 1)
            pl[2 * lane] = (uint32_t)startO | (whole ? FUSE_WHOLE : 0u);
            pl[2 * lane + 1] = (uint32_t)recLen | ((ex & 0xffffu) << 16);
            fz_old = fz_new;
            fz_new = FusePending{cur, wa, wb, mx};
            ++fz_iter;
            __builtin_amdgcn_wave_barrier();
        }
        // ------------------------------------------------------------------ deferred pairs: queued for the general kernel
        {
            const unsigned long long dmask = __ballot(valid && defer && role == 0);
            if (dmask) {
                unsigned int slot0 = 0;
                if (lane == 0) slot0 = atomicAdd(R->n_deferred, (unsigned int)__popcll(dmask));
                slot0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)slot0);
                if (valid && defer && role == 0)
                    R->deferred[slot0 + rank_below(dmask)] = (uint32_t)(rec - 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        PROF(8);
        if (++since_flush == 64 || nxt >= n_batches) flush_totals();     // (also the wave's last batch: the only copy of the flush)
        cur = nxt;
    }
    if (FUSE) {
        // the two batches still waiting, the older first (a loop that stays a loop: one more copy of fuse_finish, not two)
#pragma nounroll
        for (uint32_t k = 0; k < 2u; ++k) {
            const FusePending P = k == 0u ? fz_old : fz_new;
            if (P.b != 0xffffffffu) fuse_finish(P, L.pend[(fz_iter + k) & 1u], false);
        }
    }
    PROF_FLUSH;
#ifdef AQC_PROFILE
    // per-wave start / end stamps (100 MHz reference clock) at the tail of the deferral queue buffer: load balance
    if (lane == 0) {
        const uint64_t gw = (uint64_t)blockIdx.x * WPBT + wave;
        R->deferred[fb.n - 2 * (gw + 1)] = (uint32_t)prof_t0;
        R->deferred[fb.n - 2 * (gw + 1) + 1] = (uint32_t)__builtin_amdgcn_s_memtime();
    }
#endif
    __syncthreads();
    DevStats st;
    st.counters = R->st.counters; st.ovl_hist = R->st.ovl_hist; st.dist_hist = R->st.dist_hist; st.status = R->st.status;
    flush_block_acc(acc, st, wave * WAVE + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
}

}  // namespace aqc
