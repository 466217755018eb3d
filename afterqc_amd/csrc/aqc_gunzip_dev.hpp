// aqc_gunzip_dev.hpp — gzip INPUT decoded on the device (round 3): the sections of aqc_gunzip.cpp's ParallelGunzip, one WAVE
// per section instead of one host thread (fastq.py:23-24 upstream: gzip.open + readline on the one CPU thread).
//
// Why: the MI355X boxes grant a container 16 CPUs; the host decoder makes ~4 GB/s of text out of that, which is what bounds
// a `.gz -> .gz` run once the writer's deflate is on the device (aqc_gzdev.hpp).  The structure of the host decoder carries over:
//
//   gzd_find_kernel      a wave per section tests 64 bit positions at a time for "a non-final dynamic-Huffman block starts here"
//                        (header fields, Kraft sum of the code-length code; survivors: the code lengths must parse and give
//                        complete literal/length and distance codes).  No trial decoding: the commit rule below catches a
//                        false start.
//   gzd_decode_kernel    a wave per section decodes sequentially from its start to the first block boundary at or behind the
//                        next section's nominal start, in 16-bit SYMBOLS (>= 0x8000: "byte j of the 32 KiB before my start").
//                        Control flow is wave-uniform — bit buffer, table indices, lengths and distances live in scalar
//                        registers; the Huffman root tables (2048 + 512 entries) and a ring of the last 8192 symbols are in
//                        LDS; literals collect in one register across the lanes and leave as 128-byte stores; match copies are
//                        shared by the lanes (from the ring; from memory only beyond it, where every store has long landed).
//   gzd_chain_kernel     ONE workgroup walks the sections in order: section k counts only if it starts at the very bit section
//                        k - 1 ended on (decoding is deterministic from a block boundary: exact, not heuristic); it places the
//                        sections in the output and resolves the markers of each section's LAST 32 KiB (they point into the
//                        32 KiB before the section, resolved one step earlier).
//   gzd_resolve_kernel   everything else, in parallel: a marker's byte is in a region the chain pass has finished.
//
// The host (aqc_capi.hip: DeviceGunzip) feeds compressed windows, takes the text of the sections that chained up and hands
// anything else — a block the search did not find, a final block, an error, an overflow — back to the host decoder, which
// also checks the members' CRC-32 / ISIZE over the bytes it receives.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aqc_kernels.hpp"

namespace aqc {

constexpr uint32_t GZD_MARKER = 0x8000u;
constexpr int GZD_LROOT = 11, GZD_DROOT = 9;
constexpr int GZD_RING = 32768;                    // the whole window behind the write position stays in LDS (64 KiB per wave)
constexpr uint64_t GZD_NONE = ~0ull;
// section flags
constexpr uint32_t GZD_FINAL = 1u, GZD_ERROR = 2u, GZD_OVERFLOW = 4u;

struct GzdJob {
    const uint8_t* comp;        // compressed window; comp[0] is byte `base` of the file (not used by the kernels), 4-byte aligned
    uint64_t comp_bytes;        // readable bytes (the buffer is padded with 64 zero bytes behind)
    uint64_t start_bit;         // section 0 starts here (a known block boundary), relative to comp[0]
    uint32_t n_sections;
    uint32_t section_bytes;     // section k >= 1 is searched from bit k * section_bytes * 8 on
    uint64_t* sec_start;        // [n] found start bit (GZD_NONE: none)
    uint64_t* sec_end;          // [n] block boundary the section stopped at
    uint32_t* sec_flags;        // [n]
    uint32_t* sec_nsym;         // [n] symbols produced
    uint16_t* sym;              // [n][sym_cap]
    uint32_t sym_cap;
    // chain / resolve
    uint8_t* text;              // output bytes; text[-32768 .. 0) holds the window before the batch (right-aligned)
    uint32_t window_valid_from; // markers below this index of the first section's window point before the member's start
    uint64_t* sec_off;          // [n + 1] offset of each section's bytes in text
    uint32_t* result;           // [0] sections accepted, [1] a marker reached before the member start (corrupt), [2] the last
                                // accepted section ended a member (BFINAL), [4..5] bit position reached, [6..7] bytes of text
};

__device__ __constant__ uint16_t GZD_LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ __constant__ uint8_t GZD_LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ __constant__ uint16_t GZD_DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ __constant__ uint8_t GZD_DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ __constant__ uint8_t GZD_CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// >= 57 bits of the stream from bit position p on (any alignment; the buffer is padded)
__device__ __forceinline__ unsigned long long gzd_peek(const uint8_t* comp, unsigned long long p) {
    unsigned long long v;
    __builtin_memcpy(&v, comp + (p >> 3), 8);
    return v >> (p & 7);
}

__device__ __forceinline__ uint32_t gzd_rev(uint32_t c, int len) { return __builtin_bitreverse32(c) >> (32 - len); }

// One wave's decoding tables (LDS).  Root entries: low 4 bits = code length (0: not a root code: use the canonical search),
// bits 4.. = symbol.  The canonical search (codes longer than the root) uses first[] / offs[] / sorted[].
struct GzdTables {
    uint32_t lit[1 << GZD_LROOT];
    uint32_t dist[1 << GZD_DROOT];
    uint16_t lsorted[288], dsorted[32];
    uint16_t lcount[16], dcount[16];
    uint8_t lens[320];
    uint8_t cll[32];          // the 19 lengths of the code-length code
    uint32_t cl[128];
    uint32_t cnt[16], next[16], offs[16];      // scratch of the table builder (LDS: indexed by code length at run time)
};

// Canonical Huffman code from lens[0, n) -> root table + sorted symbols / counts.  Uniform (every lane runs it; the table fill
// is shared by the lanes).  Returns 0 ok, 1 over-subscribed, 2 incomplete (the caller applies zlib's single-code rule).
__device__ __forceinline__ uint32_t gzd_wave_sum(uint32_t v) {
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) v += (uint32_t)__shfl_xor((int)v, sft, WAVE);
    return v;
}
__device__ __forceinline__ uint32_t gzd_wave_max(uint32_t v) {
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, sft, WAVE));
    return v;
}

__device__ inline int gzd_build(GzdTables& T, const uint8_t* lens, int n, int R, uint32_t* root, uint16_t* sorted, uint16_t* count, int lane, int* max_len) {
    uint32_t* const cnt = T.cnt; uint32_t* const next = T.next; uint32_t* const offs = T.offs;
    if (lane < 16) cnt[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < n; s += WAVE) { const int l = lens[s]; if (l) atomicAdd(&cnt[l], 1u); }
    __builtin_amdgcn_wave_barrier();
    int left = 1, mx = 0;
    for (int l = 1; l <= 15; ++l) {
        const int c = (int)cnt[l];
        left = (left << 1) - c;
        if (left < 0) return 1;
        if (c) mx = l;
    }
    *max_len = mx;
    for (int i = lane; i < (1 << R); i += WAVE) root[i] = 0;
    {
        uint32_t code = 0, o = 0, prev = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (code + prev) << 1;
            prev = cnt[l];
            if (lane == 0) { next[l] = code; offs[l] = o; count[l] = (uint16_t)prev; }
            o += prev;
        }
        if (lane == 0) count[0] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l == 0) continue;
        const uint32_t c = next[l], o = offs[l];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) { sorted[o] = (uint16_t)s; next[l] = c + 1; offs[l] = o + 1; }
        if (l <= R) {
            const uint32_t e = ((uint32_t)s << 4) | (uint32_t)l;
            const uint32_t r = gzd_rev(c, l);
            for (uint32_t i = r + ((uint32_t)lane << l); i < (1u << R); i += (uint32_t)WAVE << l) root[i] = e;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return left == 0 ? 0 : 2;
}

// the code lengths of a dynamic block header at bit p (behind the 3 header bits) into T.lens; returns 0 and sets hlit / hdist /
// the bit position behind the header, or 1 when the header is not valid
__device__ inline int gzd_code_lengths(const uint8_t* comp, unsigned long long limit_bit, unsigned long long& p, GzdTables& T, int lane, int& hlit, int& hdist) {
    const unsigned long long w = gzd_peek(comp, p);
    hlit = (int)(w & 31u) + 257; hdist = (int)((w >> 5) & 31u) + 1;
    const int hclen = (int)((w >> 10) & 15u) + 4;
    p += 14;
    if (hlit > 286 || hdist > 30) return 1;
    uint8_t* cl_lens = T.cll;
    if (lane < 19) cl_lens[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane < hclen) cl_lens[GZD_CL_ORDER[lane]] = (uint8_t)(gzd_peek(comp, p + 3ull * (unsigned)lane) & 7u);
    p += 3ull * (unsigned)hclen;
    __builtin_amdgcn_wave_barrier();
    // the code-length code: complete, at most 7 bits -> a flat 128-entry table
    {
        uint32_t* const cnt = T.cnt; uint32_t* const next = T.next;
        if (lane < 16) cnt[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane < 19 && cl_lens[lane]) atomicAdd(&cnt[cl_lens[lane]], 1u);
        __builtin_amdgcn_wave_barrier();
        int left = 1, used = 0;
        for (int l = 1; l <= 7; ++l) { const int c = (int)cnt[l]; left = (left << 1) - c; if (left < 0) return 1; used += c; }
        if (used == 0 || left != 0) return 1;
        {
            uint32_t code = 0, prev = 0;
            for (int l = 1; l <= 7; ++l) { code = (code + prev) << 1; prev = cnt[l]; if (lane == 0) next[l] = code; }
        }
        for (int i = lane; i < 128; i += WAVE) T.cl[i] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int s = 0; s < 19; ++s) {
            const int l = cl_lens[s];
            if (!l) continue;
            const uint32_t c = next[l];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) next[l] = c + 1;
            const uint32_t r = gzd_rev(c, l);
            const uint32_t e = ((uint32_t)s << 4) | (uint32_t)l;
            for (uint32_t i = r + ((uint32_t)lane << l); i < 128u; i += (uint32_t)WAVE << l) T.cl[i] = e;
            __builtin_amdgcn_wave_barrier();
        }
    }
    const int total = hlit + hdist;
    int i = 0;
    uint32_t prev = 0;
    while (i < total) {
        if (p > limit_bit) return 1;
        const unsigned long long v = gzd_peek(comp, p);
        const uint32_t e = T.cl[v & 127u];
        const uint32_t l = e & 15u;
        if (l == 0) return 1;
        const uint32_t sym = e >> 4;
        p += l;
        if (sym < 16) {
            if (lane == 0) T.lens[i] = (uint8_t)sym;
            prev = sym;
            ++i;
        } else {
            uint32_t rep, val = 0;
            const uint32_t x = (uint32_t)(v >> l);
            if (sym == 16) { if (i == 0) return 1; val = prev; rep = 3 + (x & 3u); p += 2; }
            else if (sym == 17) { rep = 3 + (x & 7u); p += 3; prev = 0; }
            else { rep = 11 + (x & 127u); p += 7; prev = 0; }
            if (i + (int)rep > total) return 1;
            for (uint32_t k = (uint32_t)lane; k < rep; k += WAVE) T.lens[i + (int)k] = (uint8_t)val;
            i += (int)rep;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (T.lens[256] == 0) return 1;
    return 0;
}

// zlib's acceptance rule for a literal/length or distance code: not over-subscribed; incomplete only as a single 1-bit code.
// (Kraft sum in units of 2^-15 over the lanes.)
__device__ inline bool gzd_code_ok(const uint8_t* lens, int n, int lane) {
    uint32_t kraft = 0, mx = 0;
    for (int s = lane; s < n; s += WAVE) {
        const uint32_t l = lens[s];
        if (l) { kraft += 32768u >> l; mx = max(mx, l); }
    }
    kraft = gzd_wave_sum(kraft);
    mx = gzd_wave_max(mx);
    return kraft == 32768u || (kraft < 32768u && mx <= 1u);
}

// ---- block starts in the middle of the stream -----------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void gzd_find_kernel(GzdJob J) {
    __shared__ GzdTables T;
    const uint32_t k = blockIdx.x;
    const int lane = lane_id();
    if (k == 0) { if (lane == 0) J.sec_start[0] = J.start_bit; return; }
    const unsigned long long from = (unsigned long long)k * J.section_bytes * 8ull;
    const unsigned long long total_bits = J.comp_bytes * 8ull;
    unsigned long long to = from + (unsigned long long)J.section_bytes * 8ull;
    if (to + 600 > total_bits) to = total_bits > 600 ? total_bits - 600 : 0;
    unsigned long long found = GZD_NONE;
    for (unsigned long long base = from; base < to && found == GZD_NONE; base += WAVE) {
        const unsigned long long p = base + (unsigned)lane;
        bool cand = false;
        if (p < to) {
            const unsigned long long w = gzd_peek(J.comp, p);
            // BFINAL = 0, BTYPE = 2; HLIT <= 29, HDIST <= 29
            if ((w & 7u) == 4u && ((w >> 3) & 31u) <= 29u && ((w >> 8) & 31u) <= 29u) {
                const int hclen = (int)((w >> 13) & 15u) + 4;
                const unsigned long long v = gzd_peek(J.comp, p + 17);
                uint32_t kraft = 0;
                for (int i = 0; i < 19; ++i) {
                    const uint32_t l = i < hclen ? (uint32_t)((v >> (3 * i)) & 7u) : 0u;
                    kraft += l ? 128u >> l : 0u;
                }
                cand = kraft == 128u;
            }
        }
        unsigned long long m = __ballot(cand);
        while (m && found == GZD_NONE) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            unsigned long long q = base + (unsigned)l + 3;
            int hlit, hdist;
            if (gzd_code_lengths(J.comp, total_bits, q, T, lane, hlit, hdist) == 0 && gzd_code_ok(T.lens, hlit, lane) && gzd_code_ok(T.lens + hlit, hdist, lane))
                found = base + (unsigned)l;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) J.sec_start[k] = found;
}

// ---- one section -----------------------------------------------------------------------------------------------------------------
constexpr int GZD_CWIN = 256;              // dwords of the compressed stream staged in LDS at a time (1 KiB)

struct GzdWave {
    GzdTables T;
    uint16_t ring[GZD_RING];
    uint32_t cwin[GZD_CWIN];
};

__global__ __launch_bounds__(WAVE) void gzd_decode_kernel(GzdJob J) {
    __shared__ GzdWave W;
    const uint32_t k = blockIdx.x;
    const int lane = lane_id();
    const unsigned long long start = J.sec_start[k];
    if (start == GZD_NONE) {
        if (lane == 0) { J.sec_end[k] = GZD_NONE; J.sec_nsym[k] = 0; J.sec_flags[k] = 0; }
        return;
    }
    const unsigned long long total_bits = J.comp_bytes * 8ull;
    // (the last section of a batch stops like the others: at the first block boundary at or behind its nominal end — the
    //  window holds some megabytes beyond it — or at the stream's final block)
    const unsigned long long stop = (unsigned long long)(k + 1) * J.section_bytes * 8ull;
    const uint8_t* const comp = J.comp;
    uint16_t* const out = J.sym + (unsigned long long)k * J.sym_cap;
    const uint32_t cap = J.sym_cap;
    unsigned long long p = start;            // bit position (uniform)
    uint32_t op = 0;                         // symbols produced (uniform)
#ifdef GZD_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    uint32_t flags = 0;
    // the symbol at section position i (i < op): before the section = a marker, else the ring (distances are <= 32768)
    auto fetch = [&](long long i) -> uint32_t {
        if (i < 0) return GZD_MARKER | (uint32_t)(32768 + i);
        return W.ring[(uint32_t)i & (GZD_RING - 1)];
    };
    // `len` symbols from `dist` back to position op, the lanes side by side (positions that the copy itself produces repeat
    // the pattern of the `dist` symbols before it)
    auto copy_match = [&](uint32_t len, uint32_t dist) {
        const long long src0 = (long long)op - (long long)dist;
        for (uint32_t c0 = 0; c0 < len; c0 += WAVE) {
            const uint32_t i = c0 + (uint32_t)lane;
            uint32_t rel = i;
            if (dist < WAVE && dist <= i) { if (dist == 1) rel = 0; else while (rel >= dist) rel -= dist; }
            uint32_t val = 0;
            if (i < len) val = fetch(src0 + (long long)rel);
            __builtin_amdgcn_wave_barrier();
            if (i < len) { out[op + i] = (uint16_t)val; W.ring[(op + i) & (GZD_RING - 1)] = (uint16_t)val; }
            __builtin_amdgcn_wave_barrier();
        }
        op += len;
    };
    bool done = false;
    uint32_t guard_blocks = 0;
    while (!done) {
        if (p >= stop) break;                                   // a block boundary at or behind the next section's start
        if (p + 3 > total_bits || ++guard_blocks > (1u << 20)) { flags |= GZD_ERROR; break; }
        const unsigned long long hw = gzd_peek(comp, p);
        const uint32_t bfinal = (uint32_t)(hw & 1u), btype = (uint32_t)((hw >> 1) & 3u);
        p += 3;
        if (btype == 3) { flags |= GZD_ERROR; break; }
        if (btype == 0) {
            // stored: to the byte boundary, LEN, ~LEN, bytes
            p = (p + 7) & ~7ull;
            const unsigned long long v = gzd_peek(comp, p);
            const uint32_t len = (uint32_t)(v & 0xffffu), nlen = (uint32_t)((v >> 16) & 0xffffu);
            p += 32;
            if (len != (~nlen & 0xffffu) || p + 8ull * len > total_bits) { flags |= GZD_ERROR; break; }
            if (op + len + 64 > cap) { flags |= GZD_OVERFLOW; break; }
            const uint8_t* src = comp + (p >> 3);
            for (uint32_t i = (uint32_t)lane; i < len; i += WAVE) {
                const uint16_t b = src[i];
                out[op + i] = b;
                W.ring[(op + i) & (GZD_RING - 1)] = b;
            }
            __builtin_amdgcn_wave_barrier();
            op += len;
            p += 8ull * len;
        } else {
            int hlit = 288, hdist = 30;
            if (btype == 1) {
                for (int i = lane; i < 288; i += WAVE) W.T.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
                for (int i = lane; i < 30; i += WAVE) W.T.lens[288 + i] = 5;
                if (lane < 2) W.T.lens[288 + 30 + lane] = 5;
                hdist = 32;
                __builtin_amdgcn_wave_barrier();
            } else if (gzd_code_lengths(comp, total_bits, p, W.T, lane, hlit, hdist) != 0) { flags |= GZD_ERROR; break; }
            int lmax, dmax;
            const int rl = gzd_build(W.T, W.T.lens, hlit, GZD_LROOT, W.T.lit, W.T.lsorted, W.T.lcount, lane, &lmax);
            // (the distance lengths sit behind the literal/length ones)
            const int rd = gzd_build(W.T, W.T.lens + hlit, hdist, GZD_DROOT, W.T.dist, W.T.dsorted, W.T.dcount, lane, &dmax);
            if (rl == 1 || rd == 1 || (rl == 2 && lmax != 1) || (rd == 2 && dmax > 1)) { flags |= GZD_ERROR; break; }
            // ---- symbols, a ROUND at a time: lane i decodes the token that would start at bit p + i (literal, end of block, or
            //      length + distance with their extra bits: <= 48 bits), a scalar walk from lane 0 follows the tokens' lengths
            //      and marks the ones that are real (10 - 25 per round of 64 bit positions), a lane scan places their output.
            //      Literals are stored by their lanes; matches are applied in order, the lanes sharing each copy.  A token whose
            //      code is longer than the root index ends the round and is decoded on its own (canonical search).
            const uint32_t* const comp32 = reinterpret_cast<const uint32_t*>(comp);
            const uint32_t total_dwords = (uint32_t)((J.comp_bytes + 64) >> 2);
            uint32_t cbase = (uint32_t)(p >> 5);
            auto load_window = [&]() {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < GZD_CWIN / WAVE; ++r) {
                    const uint32_t i = cbase + (uint32_t)(r * WAVE + lane);
                    W.cwin[r * WAVE + lane] = i < total_dwords ? comp32[i] : 0u;
                }
                __builtin_amdgcn_wave_barrier();
            };
            load_window();
            bool eob = false;
            uint32_t guard = 0;
#ifdef GZD_PROFILE
#define GZD_T(k_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof[k_] += n_ - tlast; tlast = n_; } while (0)
            unsigned long long tlast = __builtin_amdgcn_s_memtime();
#else
#define GZD_T(k_)
#endif
            while (!eob) {
                if ((p >> 5) > total_dwords || ++guard > (1u << 24)) { flags |= GZD_ERROR; break; }
                if (op + 64u * 258u + 64u > cap) { flags |= GZD_OVERFLOW; break; }
                // the window must hold the dwords of bits [p, p + 64 + 64)
                if ((uint32_t)(p >> 5) - cbase + 6u > (uint32_t)GZD_CWIN) { cbase = (uint32_t)(p >> 5); load_window(); }
                unsigned long long w;
                {
                    const unsigned long long q = p + (unsigned)lane;
                    const uint32_t di = (uint32_t)(q >> 5) - cbase, sh = (uint32_t)q & 31u;
                    const unsigned long long lo = ((unsigned long long)W.cwin[di + 1] << 32) | W.cwin[di];
                    const uint32_t hi = W.cwin[di + 2];
                    w = sh ? (lo >> sh) | ((unsigned long long)hi << (64u - sh)) : lo;
                }
                const uint32_t e = W.T.lit[(uint32_t)w & ((1u << GZD_LROOT) - 1u)];
                const uint32_t l = e & 15u, sym = e >> 4;
                uint32_t tb = l, kind = 0, tlen = 1, tdist = 0;          // bits of the token; 0 literal, 1 end of block, 2 match, 3 not decodable here
                if (l == 0) kind = 3;
                else if (sym == 256) kind = 1;
                else if (sym > 256) {
                    if (sym > 285) kind = 3;
                    else {
                        const uint32_t ls = sym - 257;
                        const uint32_t lxe = ls < 8 || ls == 28 ? 0u : (ls - 4) >> 2;
                        const uint32_t lbase = ls < 8 ? 3u + ls : ls == 28 ? 258u : 3u + ((4u + (ls & 3u)) << lxe);
                        tlen = lbase + (uint32_t)((w >> l) & ((1u << lxe) - 1u));
                        const uint32_t o = l + lxe;
                        const uint32_t de = W.T.dist[(uint32_t)(w >> o) & ((1u << GZD_DROOT) - 1u)];
                        const uint32_t dl = de & 15u, dsym = de >> 4;
                        if (dl == 0 || dsym > 29) kind = 3;
                        else {
                            const uint32_t dxe = dsym < 4 ? 0u : (dsym >> 1) - 1u;
                            const uint32_t dbase = dsym < 4 ? dsym + 1u : 1u + ((2u + (dsym & 1u)) << dxe);
                            tdist = dbase + (uint32_t)((w >> (o + dl)) & ((1u << dxe) - 1u));
                            tb = o + dl + dxe;
                            kind = 2;
                        }
                    }
                }
                GZD_T(0);
                // ---- the walk: which lanes start a real token
                const uint32_t packed = tb | (kind << 8);
                unsigned long long tok = 0;
                uint32_t cur = 0;
                bool slow = false;
                while (cur < (uint32_t)WAVE) {
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)packed, (int)cur);
                    const uint32_t k2 = pk >> 8;
                    if (k2 == 3) { slow = true; break; }
                    tok |= 1ull << cur;
                    cur += pk & 0xffu;
                    if (k2 == 1) { eob = true; break; }
                }
                GZD_T(1);
                // ---- output positions: exclusive scan of the real tokens' output lengths
                const bool mine = (tok >> lane) & 1ull;
                const uint32_t olen = mine ? (kind == 0 ? 1u : kind == 2 ? tlen : 0u) : 0u;
                uint32_t inc = olen;
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) {
                    const uint32_t o2 = (uint32_t)__shfl_up((int)inc, d, WAVE);
                    if (lane >= d) inc += o2;
                }
                const uint32_t total_out = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
                const uint32_t my_pos = op + inc - olen;
                GZD_T(2);
                // matches that reach before the window are errors
                const unsigned long long mm_all = __ballot(mine && kind == 2);
                if (__ballot(mine && kind == 2 && (long long)my_pos - (long long)tdist < -32768)) { flags |= GZD_ERROR; break; }
                if (mm_all == 0) {
                    // literals only: all at once
                    if (mine && kind == 0) { out[my_pos] = (uint16_t)sym; W.ring[my_pos & (GZD_RING - 1)] = (uint16_t)sym; }
                    op += total_out;
                } else {
                    if (mine && kind == 0) { out[my_pos] = (uint16_t)sym; W.ring[my_pos & (GZD_RING - 1)] = (uint16_t)sym; }
                    __builtin_amdgcn_wave_barrier();
                    unsigned long long mm = mm_all;
                    const uint32_t op0 = op;
                    while (mm) {
                        const int ml = __ffsll((long long)mm) - 1;
                        mm &= mm - 1;
                        const uint32_t mlen = (uint32_t)__builtin_amdgcn_readlane((int)tlen, ml);
                        const uint32_t mdist = (uint32_t)__builtin_amdgcn_readlane((int)tdist, ml);
                        op = (uint32_t)__builtin_amdgcn_readlane((int)my_pos, ml);
                        copy_match(mlen, mdist);
                    }
                    op = op0 + total_out;
                }
                p += cur;
                GZD_T(3);
#ifdef GZD_PROFILE
                prof[5] += 1; prof[6] += (unsigned long long)__popcll(tok); prof[7] += (unsigned long long)__popcll(mm_all);
#endif
                if (slow) {
                    // one token with a code longer than the root index, decoded on its own
                    const unsigned long long v0 = gzd_peek(comp, p);
                    unsigned long long v = v0;
                    uint32_t e2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.T.lit[(uint32_t)v & ((1u << GZD_LROOT) - 1u)]);
                    uint32_t l2 = e2 & 15u, sym2 = e2 >> 4;
                    if (l2 == 0) {
                        uint32_t code = gzd_rev((uint32_t)v & ((1u << GZD_LROOT) - 1u), GZD_LROOT), first = 0, index = 0;
                        for (int q = 1; q <= GZD_LROOT; ++q) { const uint32_t c = W.T.lcount[q]; first = (first + c) << 1; index += c; }
                        bool ok = false;
                        for (int q = GZD_LROOT + 1; q <= 15; ++q) {
                            code = (code << 1) | (uint32_t)((v >> (q - 1)) & 1u);
                            const uint32_t c = W.T.lcount[q];
                            if (code - first < c) { sym2 = W.T.lsorted[index + (code - first)]; l2 = (uint32_t)q; ok = true; break; }
                            index += c;
                            first = (first + c) << 1;
                        }
                        if (!ok) { flags |= GZD_ERROR; break; }
                        sym2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sym2);
                    }
                    p += l2;
                    v >>= l2;
                    if (sym2 < 256) {
                        if (lane == 0) { out[op] = (uint16_t)sym2; W.ring[op & (GZD_RING - 1)] = (uint16_t)sym2; }
                        ++op;
                    } else if (sym2 == 256) eob = true;
                    else if (sym2 > 285) { flags |= GZD_ERROR; break; }
                    else {
                        const uint32_t ls = sym2 - 257;
                        const uint32_t lx = GZD_LEN_EXTRA[ls];
                        const uint32_t len = GZD_LEN_BASE[ls] + (uint32_t)(v & ((1u << lx) - 1u));
                        p += lx;
                        v = gzd_peek(comp, p);
                        uint32_t de = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.T.dist[(uint32_t)v & ((1u << GZD_DROOT) - 1u)]);
                        uint32_t dl = de & 15u, dsym = de >> 4;
                        if (dl == 0) {
                            uint32_t code = gzd_rev((uint32_t)v & ((1u << GZD_DROOT) - 1u), GZD_DROOT), first = 0, index = 0;
                            for (int q = 1; q <= GZD_DROOT; ++q) { const uint32_t c = W.T.dcount[q]; first = (first + c) << 1; index += c; }
                            bool ok = false;
                            for (int q = GZD_DROOT + 1; q <= 15; ++q) {
                                code = (code << 1) | (uint32_t)((v >> (q - 1)) & 1u);
                                const uint32_t c = W.T.dcount[q];
                                if (code - first < c) { dsym = W.T.dsorted[index + (code - first)]; dl = (uint32_t)q; ok = true; break; }
                                index += c;
                                first = (first + c) << 1;
                            }
                            if (!ok) { flags |= GZD_ERROR; break; }
                            dsym = (uint32_t)__builtin_amdgcn_readfirstlane((int)dsym);
                        }
                        if (dsym > 29) { flags |= GZD_ERROR; break; }
                        p += dl;
                        v >>= dl;
                        const uint32_t dx = GZD_DIST_EXTRA[dsym];
                        const uint32_t dist = GZD_DIST_BASE[dsym] + (uint32_t)(v & ((1u << dx) - 1u));
                        p += dx;
                        if ((long long)op - (long long)dist < -32768) { flags |= GZD_ERROR; break; }
                        __builtin_amdgcn_wave_barrier();
                        copy_match(len, dist);
                    }
                }
            }
            if (flags) break;
        }
        if (bfinal) { flags |= GZD_FINAL; done = true; }
    }
    if (lane == 0) { J.sec_end[k] = p; J.sec_nsym[k] = op; J.sec_flags[k] = flags; }
#ifdef GZD_PROFILE
    if (lane == 0 && k == 1)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(J.result + 8)[i] = prof[i];
#endif
}

// ---- commit in order + the last 32 KiB of every section -------------------------------------------------------------------------
constexpr int GZD_CHAIN_THREADS = 1024;

__global__ __launch_bounds__(GZD_CHAIN_THREADS) void gzd_chain_kernel(GzdJob J) {
    __shared__ uint32_t bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    unsigned long long off = 0, expect = J.start_bit;
    uint32_t accepted = 0, final_seen = 0;
    for (uint32_t k = 0; k < J.n_sections; ++k) {
        const unsigned long long st = J.sec_start[k];
        const uint32_t fl = J.sec_flags[k];
        // a section without a start inside a long block is simply skipped: its predecessor ran through it
        if (st == GZD_NONE && k > 0) {
            if (expect >= (unsigned long long)(k + 1) * J.section_bytes * 8ull) { if (threadIdx.x == 0) J.sec_off[k] = off; ++accepted; continue; }
            break;
        }
        if (st != expect || (fl & (GZD_ERROR | GZD_OVERFLOW))) break;
        const uint32_t n = J.sec_nsym[k];
        if (threadIdx.x == 0) J.sec_off[k] = off;
        const uint32_t tail = n < 32768u ? n : 32768u;
        const uint16_t* s = J.sym + (unsigned long long)k * J.sym_cap + (n - tail);
        uint8_t* d = J.text + off + (n - tail);
        const uint8_t* win = J.text + off - 32768;             // the 32 KiB before this section
        const uint32_t valid_from = k == 0 ? J.window_valid_from : 0u;   // (later sections: the member started before them)
        // (32 symbols per thread: all loads first, so that one memory latency covers them, then the few marker look-ups)
        uint32_t v[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const uint32_t i = threadIdx.x + (uint32_t)r * GZD_CHAIN_THREADS;
            v[r] = i < tail ? (uint32_t)s[i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const uint32_t i = threadIdx.x + (uint32_t)r * GZD_CHAIN_THREADS;
            if (i < tail) {
                uint8_t b;
                if (v[r] < GZD_MARKER) b = (uint8_t)v[r];
                else {
                    const uint32_t j = v[r] & 0x7fffu;
                    if (j < valid_from) bad = 1;
                    b = win[j];
                }
                d[i] = b;
            }
        }
        __threadfence();
        __syncthreads();
        off += n;
        expect = J.sec_end[k];
        ++accepted;
        if (fl & GZD_FINAL) { final_seen = 1; break; }
    }
    if (threadIdx.x == 0) {
        J.sec_off[J.n_sections] = off;
        J.result[0] = accepted;
        J.result[1] = bad;
        J.result[2] = final_seen;
        J.result[4] = (uint32_t)expect; J.result[5] = (uint32_t)(expect >> 32);
        J.result[6] = (uint32_t)off; J.result[7] = (uint32_t)(off >> 32);
    }
}

// the rest of every accepted section (all but its last 32 KiB), in parallel
__global__ __launch_bounds__(256) void gzd_resolve_kernel(GzdJob J) {
    const uint32_t k = blockIdx.y;
    if (k >= J.result[0]) return;
    if (J.sec_start[k] == GZD_NONE) return;
    const uint32_t n = J.sec_nsym[k];
    const uint32_t body = n > 32768u ? n - 32768u : 0u;
    const unsigned long long off = J.sec_off[k];
    const uint16_t* s = J.sym + (unsigned long long)k * J.sym_cap;
    uint8_t* d = J.text + off;
    const uint8_t* win = J.text + off - 32768;
    const uint32_t valid_from = k == 0 ? J.window_valid_from : 0u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < body; i += gridDim.x * 256u) {
        const uint32_t v = s[i];
        uint8_t b;
        if (v < GZD_MARKER) b = (uint8_t)v;
        else {
            const uint32_t j = v & 0x7fffu;
            if (j < valid_from) J.result[1] = 1;
            b = win[j];
        }
        d[i] = b;
    }
}

}  // namespace aqc
