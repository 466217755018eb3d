// aqc_text.hpp — FASTQ text in, FASTQ text out, on the device.
//
//   framing    fastq.Reader.nextRead (fastq.py:37-49): a record is 4 lines, each `readline().rstrip()`; a line
//              that is empty after stripping ends the file.  The raw text chunk is the byte arena; the kernels
//              here find the newlines, strip trailing whitespace and emit (offset, length) per line.
//   formatting seqFilter.writeReads (preprocesser.py:206-232) + fastq.Writer.writeLines (fastq.py:87-93):
//              name, bases, strand line, qualities, each followed by "\n"; a bad record's name becomes
//              "@" + FLAG + name[1:]; bases/qualities are the trimmed / adapter-cut slices with the <= 3 edits of
//              the correction walk applied.  Good and bad records of each file are compacted into their own
//              contiguous text streams in record order (sizes -> exclusive scan -> copy).
//
// All of it is byte shuffling bound by HBM bandwidth; no data-dependent host work remains per record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afterqc_hip.h"
#include "aqc_kernels.hpp"
#include "aqc_fast.hpp"      // the DPP wave sums / scans

namespace aqc {

constexpr int TXT_BLOCK = 256;
constexpr int TXT_BYTES_PER_THREAD = 64;
constexpr int TXT_TILE = TXT_BLOCK * TXT_BYTES_PER_THREAD;    // 16 KiB of text per workgroup
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = TXT_BLOCK * SCAN_ITEMS;             // 4096 values per workgroup

// exclusive prefix of one value per thread over a 256-thread workgroup; `total` = sum over the workgroup
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* lds /* [4] */,
                                                              unsigned long long& total) {
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();                       // lds may still be read by the previous call
    if (lane == WAVE - 1) lds[wave] = inc;
    __syncthreads();
    unsigned long long base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < TXT_BLOCK / WAVE; ++w) {
        if (w >= (int)(blockDim.x / WAVE)) break;            // (workgroups of 64 .. 256 threads)
        const unsigned long long t = lds[w];
        if (w < wave) base += t;
        total += t;
    }
    return base + inc - v;
}

// ---- generic three-pass exclusive scan of u32 values produced by a functor -----------------------------------
template <class F>
__global__ __launch_bounds__(TXT_BLOCK) void scan_tile_sums_kernel(F f, uint64_t n, unsigned long long* __restrict__ tile_sum) {
    __shared__ unsigned long long lds[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (i0 + k < n) s += f(i0 + k);
    unsigned long long total;
    (void)block_excl_scan(s, lds, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// in-place exclusive scan of the tile sums by ONE workgroup; the grand total goes to *total_out
__global__ __launch_bounds__(TXT_BLOCK) void scan_tile_bases_kernel(unsigned long long* __restrict__ tile_sum, uint64_t n_tiles,
                                                                    unsigned long long* __restrict__ total_out) {
    __shared__ unsigned long long lds[4];
    unsigned long long carry = 0;
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += TXT_BLOCK) {
        const uint64_t t = t0 + threadIdx.x;
        const unsigned long long v = t < n_tiles ? tile_sum[t] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan(v, lds, total);
        if (t < n_tiles) tile_sum[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

template <class F, class OutT>
__global__ __launch_bounds__(TXT_BLOCK) void scan_apply_kernel(F f, uint64_t n, const unsigned long long* __restrict__ tile_base,
                                                               OutT* __restrict__ out, unsigned long long add) {
    __shared__ unsigned long long lds[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = i0 + k < n ? f(i0 + k) : 0u;
        s += v[k];
    }
    unsigned long long total;
    unsigned long long run = tile_base[blockIdx.x] + add + block_excl_scan(s, lds, total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (i0 + k < n) out[i0 + k] = (OutT)run;
        run += v[k];
    }
}

// ---- framing ----------------------------------------------------------------------------------------------------
// 64-bit mask of the '\n' bytes among the 64 bytes at p (16-byte aligned, readable: the device buffer is padded)
__device__ __forceinline__ unsigned long long newline_mask64(const uint8_t* p, uint64_t pos, uint64_t bytes) {
    unsigned long long mask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 v = reinterpret_cast<const uint4*>(p)[k];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t x = w[d] ^ 0x0a0a0a0au;
            const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;       // 0x80 where the byte is '\n'
            const uint32_t nib = ((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xfu;
            mask |= (unsigned long long)nib << (16 * k + 4 * d);
        }
    }
    if (pos + 64 > bytes) mask &= pos >= bytes ? 0ull : ((1ull << (bytes - pos)) - 1ull);
    return mask;
}

// ---- line index of a text chunk in ONE pass over the bytes ----------------------------------------------------------
// line_end[i] = byte position of the i-th '\n' (bit 31: the byte before it is a blank or control character, i.e. the
// line MAY end in whitespace that readline().rstrip() removes — the framing kernel looks at the text only then).
// Tiles of 32 KiB are claimed in arrival order (ticket) and chained with a decoupled look-back: a tile publishes its
// newline count (flag A), adds up its predecessors' counts until it meets one that already knows its inclusive prefix
// (flag P), publishes its own prefix and emits.  The text is read once; the old count / scan / emit trio read it twice
// and needed three launches per file.
#ifndef AQC_IDX_SUB
#define AQC_IDX_SUB 4
#endif
constexpr int IDX_K = 8;                              // 16-byte pieces a lane loads at a time (one sub-tile)
constexpr int IDX_SUB = AQC_IDX_SUB;                  // sub-tiles per tile
constexpr int IDX_P = IDX_K * IDX_SUB;                // pieces per lane per tile
constexpr int IDX_TILE = TXT_BLOCK * 16 * IDX_P;      // 128 KiB: one ticket and one look-back per tile (same-address atomics
                                                      // serialise in L2 at ~8 ns each, so 32 KiB tiles capped the kernel near 4 TB/s)
constexpr unsigned long long IDX_FLAG_A = 1ull << 62, IDX_FLAG_P = 2ull << 62, IDX_VAL = (1ull << 62) - 1ull;
constexpr uint32_t LINE_WS = 0x80000000u, LINE_POS = 0x7fffffffu;

struct IndexFile {
    const uint8_t* text;
    uint64_t bytes;
    uint32_t* line_end;
    uint64_t cap;              // entries line_end can take (more are counted, not written)
    unsigned long long* total; // out: number of '\n' in the chunk
    uint32_t tile0, tiles;     // this file's tiles are [tile0, tile0 + tiles) of the launch
};

// newline flags and "< 0x21" flags of the 16 bytes of v, one bit per byte
__device__ __forceinline__ void piece_masks(const uint4 v, uint32_t& nl16, uint32_t& bl16) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t zb[4], zn[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t t = (w[d] & 0x7f7f7f7fu) + 0x5f5f5f5fu;                            // bit 7 where the low 7 bits are >= 0x21
        zb[d] = ~(t | w[d]) & 0x80808080u;                                                 // 0x80 where the byte is < 0x21
        // '\n' is one of those bytes, and they all have bit 7 clear: byte ^ 0x0a is zero iff adding 0x7f does not reach bit 7
        const uint32_t x = w[d] ^ 0x0a0a0a0au;
        zn[d] = ~((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) & zb[d];                          // 0x80 where the byte is '\n'
    }
    // gather the bit 7s: dot products with weights 1, 2, 4, ... (0x80 * mask), two words per chain
    const uint32_t n_lo = __builtin_amdgcn_udot4(zn[1], 0x80402010u, __builtin_amdgcn_udot4(zn[0], 0x08040201u, 0u, false), false);
    const uint32_t n_hi = __builtin_amdgcn_udot4(zn[3], 0x80402010u, __builtin_amdgcn_udot4(zn[2], 0x08040201u, 0u, false), false);
    const uint32_t b_lo = __builtin_amdgcn_udot4(zb[1], 0x80402010u, __builtin_amdgcn_udot4(zb[0], 0x08040201u, 0u, false), false);
    const uint32_t b_hi = __builtin_amdgcn_udot4(zb[3], 0x80402010u, __builtin_amdgcn_udot4(zb[2], 0x08040201u, 0u, false), false);
    nl16 = (n_lo >> 7) | (n_hi << 1);
    bl16 = (b_lo >> 7) | (b_hi << 1);
}

// Layout inside a tile: a wave owns IDX_SUB * 8 KiB; its lane i reads the 16-byte pieces at  wave base + p * 1024 + i * 16,
// p = 0..IDX_P-1, eight at a time — every load instruction of the wave is one contiguous KiB (the earlier "contiguous bytes
// per thread" made each instruction touch 64 different cache lines and ran at 2.3 TB/s whatever the arithmetic cost).  The
// text order of the pieces is (p, lane), so the rank of a piece's first newline is  tile prefix + waves before + pieces
// (p' < p) + lanes before within p.  Between the load and the emit only the 16-bit newline / blank masks of a piece are
// kept, in LDS (32 KiB per workgroup); the emit pass rebuilds the lane ranks from them with one packed lane scan per two
// pieces.  Both passes are rolled loops: fully unrolled, the compiler kept the whole tile's state live (300 registers).
__global__ __launch_bounds__(TXT_BLOCK) void text_index_kernel(IndexFile f0, IndexFile f1, unsigned long long* __restrict__ state,
                                                               unsigned int* __restrict__ ticket) {
    __shared__ unsigned int s_tile;
    __shared__ unsigned int s_wave_tot[TXT_BLOCK / WAVE];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_nl[IDX_P / 2][TXT_BLOCK], s_ws[IDX_P / 2][TXT_BLOCK];     // two pieces per word
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const IndexFile& f = tile >= f1.tile0 && f1.tiles ? f1 : f0;
    const uint32_t lt = tile - f.tile0;                                   // tile within the file
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t wbase = (uint64_t)lt * IDX_TILE + (uint64_t)wave * (WAVE * 16 * IDX_P);
    const uint64_t lbase = wbase + (uint64_t)lane * 16;
    // (the buffer is zero-filled for more than a tile behind the text; loads are still bounded by the text's end)
    // (round 6: a wave whose 32 KiB lie inside the text — all but a file's last — loads and masks without looking at the text's end;
    //  "the byte before is blank" comes from a ballot of the lanes' last bytes instead of a lane shift through LDS; the lane scans
    //  of the emit pass are DPP scans: 146 -> ~115 instructions per 16 bytes of an instruction-bound kernel)
    const bool full = wbase + (uint64_t)(WAVE * 16 * IDX_P) <= f.bytes;          // (wave-uniform)
    uint32_t carry_in = (wbase > 0 && wbase <= f.bytes) ? (f.text[wbase - 1] < 0x21 ? 1u : 0u) : 0u;      // (wave-uniform too)
    uint32_t mine = 0;                            // newlines in this lane's pieces
    uint4 v[IDX_K];
    auto load_piece = [&](int p) -> uint4 {
        const uint64_t q = lbase + (uint64_t)p * (WAVE * 16);
        if (full) return *reinterpret_cast<const uint4*>(f.text + q);
        return q < f.bytes ? *reinterpret_cast<const uint4*>(f.text + q) : make_uint4(0, 0, 0, 0);
    };
#pragma unroll
    for (int k = 0; k < IDX_K; ++k) v[k] = load_piece(k);
#pragma unroll 1
    for (int sub = 0; sub < IDX_SUB; ++sub) {
        uint4 nx[IDX_K];
#pragma unroll
        for (int k = 0; k < IDX_K; ++k)          // the next sub-tile is on its way while this one is worked on
            nx[k] = sub + 1 < IDX_SUB ? load_piece((sub + 1) * IDX_K + k) : make_uint4(0, 0, 0, 0);
        uint32_t nlp = 0, wsp = 0;
#pragma unroll
        for (int k = 0; k < IDX_K; ++k) {
            const int p = sub * IDX_K + k;
            uint32_t nl, bl;
            piece_masks(v[k], nl, bl);
            // mask the bytes behind the end of the text (the last piece may be partial)
            if (!full) {
                const uint64_t q = lbase + (uint64_t)p * (WAVE * 16);
                if (q + 16 > f.bytes) { const uint32_t keep = q >= f.bytes ? 0u : ((1u << (f.bytes - q)) - 1u); nl &= keep; }
            }
            // "the byte before is blank": this piece's flags moved up one byte; the byte before the piece is the last byte
            // of the piece of the lane before (same p), for lane 0 of the last lane's piece of p - 1
            const unsigned long long tops = __ballot((bl >> 15) != 0u);
            const uint32_t prev = (uint32_t)((((tops << 1) | carry_in) >> lane) & 1ull);
            carry_in = (uint32_t)(tops >> (WAVE - 1));
            const uint32_t ws = ((bl << 1) | prev) & 0xffffu;
            mine += (uint32_t)__popc(nl);
            if (k & 1) {
                s_nl[p / 2][threadIdx.x] = nlp | (nl << 16);
                s_ws[p / 2][threadIdx.x] = wsp | (ws << 16);
            } else { nlp = nl; wsp = ws; }
        }
#pragma unroll
        for (int k = 0; k < IDX_K; ++k) v[k] = nx[k];
    }
    const uint32_t wtot = (uint32_t)wave_sum_u((int)mine);
    if (lane == 0) s_wave_tot[wave] = wtot;
    __syncthreads();
    unsigned long long total = 0, wave_off = 0;
#pragma unroll
    for (int w = 0; w < TXT_BLOCK / WAVE; ++w) {
        if (w < wave) wave_off += s_wave_tot[w];
        total += s_wave_tot[w];
    }
    if (threadIdx.x < WAVE) {
        unsigned long long base = 0;
        if (lt == 0) {
            if (lane == 0) __hip_atomic_store(&state[tile], IDX_FLAG_P | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&state[tile], IDX_FLAG_A | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long j = (long long)tile - 1;                            // look back from the predecessor
            const long long first = (long long)f.tile0;
            while (true) {
                const long long idx = j - lane;
                unsigned long long sv = IDX_FLAG_P;                       // before the file's first tile: prefix 0
                if (idx >= first) sv = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int flag = (unsigned int)(sv >> 62);
                const unsigned long long pend = __ballot(flag == 0), pfx = __ballot(flag == 2);
                if (pfx) {
                    const int fp = __ffsll((long long)pfx) - 1;
                    const unsigned long long upto = fp == 63 ? ~0ull : ((2ull << fp) - 1ull);
                    if (pend & upto) { __builtin_amdgcn_s_sleep(1); continue; }
                    unsigned long long part = lane <= fp ? (sv & IDX_VAL) : 0ull;
#pragma unroll
                    for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft, WAVE);
                    base += part;
                    break;
                }
                if (pend) { __builtin_amdgcn_s_sleep(1); continue; }
                unsigned long long part = sv & IDX_VAL;
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft, WAVE);
                base += part;
                j -= WAVE;
            }
            if (lane == 0) __hip_atomic_store(&state[tile], IDX_FLAG_P | (base + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_base = base;
            if (lt + 1 == f.tiles) *f.total = base + total;
        }
    }
    __syncthreads();
    unsigned long long run = s_base + wave_off;      // rank of the first newline of the wave's pieces in row p (wave-uniform)
#pragma unroll 2
    for (int j = 0; j < IDX_P / 2; ++j) {
        const uint32_t nlp = s_nl[j][threadIdx.x], wsp = s_ws[j][threadIdx.x];
        const uint32_t own = (uint32_t)__popc(nlp & 0xffffu) | ((uint32_t)__popc(nlp >> 16) << 16);
        const uint32_t c = (uint32_t)wave_incl_sum((int)own, lane);      // inclusive lane scan, two rows at once (each < 2^11)
        const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)c, WAVE - 1);
        const uint32_t ex = c - own;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t m = h ? (nlp >> 16) : (nlp & 0xffffu);
            const uint32_t ws = h ? (wsp >> 16) : (wsp & 0xffffu);
            unsigned long long i = run + (h ? (ex >> 16) : (ex & 0xffffu));
            const uint32_t q = (uint32_t)lbase + (uint32_t)(2 * j + h) * (WAVE * 16);      // (chunks are < 2 GiB)
            // (a lane's 16 bytes hold at most two line ends but for one-base lines: two predicated stores, then the loop for the rest)
            if (m) {
                const int bit = __builtin_ctz(m);
                if (i < f.cap) f.line_end[i] = (q + (uint32_t)bit) | (((ws >> bit) & 1u) ? LINE_WS : 0u);
                m &= m - 1;
                if (m) {
                    const int bit2 = __builtin_ctz(m);
                    if (i + 1 < f.cap) f.line_end[i + 1] = (q + (uint32_t)bit2) | (((ws >> bit2) & 1u) ? LINE_WS : 0u);
                    m &= m - 1;
                }
                i += 2;
            }
            if (__ballot(m != 0u)) {
                while (m) {
                    const int bit = __builtin_ctz(m);
                    if (i < f.cap) f.line_end[i] = (q + (uint32_t)bit) | (((ws >> bit) & 1u) ? LINE_WS : 0u);
                    ++i;
                    m &= m - 1;
                }
            }
            run += h ? (last >> 16) : (last & 0xffffu);
        }
    }
}

// the whitespace bytes.rstrip() removes: space, \t \n \v \f \r
__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

struct FrameMeta {
    unsigned int first_empty;   // first record with an empty line (0xffffffff = none)
    unsigned int max_len;       // longest sequence line
    unsigned int first_mismatch; // first record whose quality line is not as long as its sequence line
    unsigned int pad_;
};

// the four lines of every complete group of the chunk (fastq.py:37-49); thread per record
struct FramedFile {
    uint32_t* seq_off;
    uint32_t* qual_off;
    uint32_t* seq_len;
    uint32_t* name_off;
    uint32_t* name_len;
    uint32_t* plus_off;
    uint32_t* plus_len;
    uint32_t* qual_len;      // QLEN_TAILNL: the byte behind the (stripped) quality line is its '\n'; QLEN_CONTIG: that holds for all four lines
};

// (the number of lines comes from the index pass's device-side total: nothing of it goes through the host first.  virt_end != 0:
//  the file's unterminated last line ends at this virtual line end — readline() returns it — which is line number *d_total)
__global__ __launch_bounds__(TXT_BLOCK) void frame_records_kernel(const uint8_t* __restrict__ text,
                                                                  const uint32_t* __restrict__ line_end, const unsigned long long* __restrict__ d_total,
                                                                  uint32_t virt_end, FramedFile out, FrameMeta* __restrict__ meta, uint64_t cap) {
    // (`cap`: entries the line table holds.  The index pass COUNTS every line and writes the first `cap`: when a chunk of very
    //  short lines overflows the table the host indexes it again with the exact size — until then nothing beyond the table may
    //  be read, and no record beyond it written: the output arrays are sized for cap / 4 records — round-4 advisory)
    const uint64_t real = *d_total < cap ? *d_total : cap;
    const uint64_t n_rec = (real + ((virt_end && *d_total <= cap) ? 1u : 0u)) / 4;
    if ((uint64_t)blockIdx.x * TXT_BLOCK >= n_rec) return;          // (the grid is sized for the most lines the chunk could hold)
    const uint64_t r = (uint64_t)blockIdx.x * TXT_BLOCK + threadIdx.x;
    const bool in = r < n_rec;
    const int lane = lane_id();
    // the record's four line ends in one 16-byte load; the end of the line before it is the neighbour lane's fourth
    uint4 le4 = make_uint4(0, 0, 0, 0);
    if (in) {
        le4 = reinterpret_cast<const uint4*>(line_end)[r];
        if (virt_end && 4 * r + 3 == real && *d_total <= cap) le4.w = virt_end;         // (a virtual line can only be the last line of the last record)
    }
    uint32_t before = (uint32_t)__shfl_up((int)le4.w, 1, WAVE);
    if (lane == 0) before = (in && r > 0) ? line_end[4 * r - 1] : 0u;
    const uint32_t le[4] = {le4.x, le4.y, le4.z, le4.w};
    uint32_t s[4] = {0, 0, 0, 0}, l[4] = {1, 1, 1, 1};
    uint32_t tail_nl = 0;                                   // the quality line ends right at its '\n' (nothing stripped)
    bool all_nl = true;                                     // ... and so do the other three lines
    if (in) {
        uint32_t b = r == 0 ? 0u : (before & LINE_POS) + 1u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t e = le[k] & LINE_POS;
            const uint32_t nl_at = e;
            bool at_nl = true;
            if (le[k] & LINE_WS) {                          // only lines that may end in whitespace touch the text
                while (e > b && is_space(text[e - 1])) --e;
                at_nl = e == nl_at && text[nl_at] == '\n';   // (the file's unterminated last line ends at a virtual '\n')
            }
            s[k] = b;
            l[k] = e - b;
            all_nl = all_nl && at_nl;
            if (k == 3) tail_nl = at_nl ? QLEN_TAILNL : 0u;
            b = nl_at + 1u;
        }
        out.name_off[r] = s[0]; out.name_len[r] = l[0];
        // a quality line of another length than its sequence line: the reference does not mind (fastq.py:37-49 hands the lines
        // over as they are) — the record is marked, every later stage keeps a view per string (aqc_kernels.hpp, LEN_IRR)
        out.seq_off[r] = s[1];  out.seq_len[r] = l[1] | (l[1] != l[3] ? LEN_IRR : 0u);
        out.plus_off[r] = s[2]; out.plus_len[r] = l[2] | (l[1] != l[3] ? LEN_IRR : 0u);      // (the mark once more, where the writer's sizing pass reads anyway)
        out.qual_off[r] = s[3]; out.qual_len[r] = min(l[3], QLEN_MASK) | tail_nl | (all_nl ? QLEN_CONTIG : 0u);
    }
    // chunk-wide reductions: at most one atomic per workgroup and only when it has something to say.  (Same-address
    // atomics serialise in L2 at ~8 ns each; the running maximum is read with an L2-coherent load — a plain load is served
    // from the CU's L1, which kept saying 0 for most of the kernel and let nearly every wave through to the atomic.)
    __shared__ unsigned int s_mx[TXT_BLOCK / WAVE];
    const bool empty = in && (l[0] == 0 || l[1] == 0 || l[2] == 0 || l[3] == 0);
    const bool mism = in && !empty && l[1] != l[3];
    const unsigned long long be = __ballot(empty), bm = __ballot(mism);
    unsigned int mx = in ? l[1] : 0u;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, sft, WAVE));
    if (lane == 0) {
        const uint64_t rw = r;                                     // first record of this wave
        if (be) atomicMin(&meta->first_empty, (unsigned int)(rw + (uint64_t)__builtin_ctzll(be)));
        if (bm) atomicMin(&meta->first_mismatch, (unsigned int)(rw + (uint64_t)__builtin_ctzll(bm)));
        s_mx[threadIdx.x / WAVE] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < TXT_BLOCK / WAVE; ++w) mx = max(mx, s_mx[w]);
        if (mx > __hip_atomic_load(&meta->max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&meta->max_len, mx);
    }
}

// What aqc_frame reports, worked out on the device behind the framing kernels (one thread): the lock-step record count of
// preprocesser.py:412-429, the bytes the n records take, R1's next sequence length.  The host reads it with ONE copy and ONE wait
// per chunk (rounds 1 - 3: three round trips — line totals, frame meta, tail values).
struct FrameOut {
    unsigned long long n, avail[2], lines[2], consumed[2];
    unsigned int eof[2], first_mismatch[2], max_len, next_len1;
};

__global__ void frame_finish_kernel(const unsigned long long* __restrict__ d_total, const FrameMeta* __restrict__ meta, const uint32_t* __restrict__ line_end0,
                                    const uint32_t* __restrict__ line_end1, const uint32_t* __restrict__ seq_len0, uint32_t virt0, uint32_t virt1,
                                    unsigned long long bytes0, unsigned long long bytes1, int nf, unsigned long long max_records, FrameOut* __restrict__ out,
                                    unsigned long long cap0, unsigned long long cap1) {
    const unsigned long long cap[2] = {cap0, cap1};
    const uint32_t* const le[2] = {line_end0, line_end1};
    const uint32_t virt[2] = {virt0, virt1};
    const unsigned long long bytes[2] = {bytes0, bytes1};
    FrameOut o{};
    unsigned long long nrec[2] = {0, 0};
    for (int k = 0; k < nf; ++k) {
        o.lines[k] = d_total[k] + (virt[k] ? 1u : 0u);
        // (a table that overflowed: the host sees lines > cap and frames the chunk again; what is reported until then stays inside it)
        nrec[k] = (d_total[k] <= cap[k] ? o.lines[k] : cap[k]) / 4;
        o.avail[k] = meta[k].first_empty < nrec[k] ? meta[k].first_empty : nrec[k];
        o.eof[k] = meta[k].first_empty < nrec[k] ? 1u : 0u;
        o.first_mismatch[k] = meta[k].first_mismatch;
        o.max_len = meta[k].max_len > o.max_len ? meta[k].max_len : o.max_len;
    }
    unsigned long long n = o.avail[0];
    if (nf == 2 && o.avail[1] < n) n = o.avail[1];
    if (max_records < n) n = max_records;
    o.n = n;
    for (int k = 0; k < nf; ++k) {
        if (!n) continue;
        const unsigned long long i = 4 * n - 1;
        const uint32_t e = (virt[k] && i == d_total[k]) ? virt[k] : le[k][i];
        const unsigned long long end = (unsigned long long)(e & LINE_POS) + 1;
        o.consumed[k] = end < bytes[k] ? end : bytes[k];
    }
    o.next_len1 = o.avail[0] > n ? (seq_len0[n] & LEN_MASK) : 0u;
    *out = o;
}

// ---- Illumina read names for the bubble filter (preprocesser.py:155,176-192) ---------------------------------------
// re.search(r'\S+\:\d+\:\S+\:\d+\:\d+\:\d+\:\d+', name), then items = match.split(':'), lane = int(items[3]),
// tile = int(items[4][1:]), x = int(items[5]), y = int(items[6]).  The search is reproduced with the regex engine's
// own order: leftmost start; first \S+ greedy (longest first, backing off to the previous ':'); \d+ runs are maximal
// (a shorter run is followed by a digit, never by ':'); second \S+ greedy.  ok = 0 no match, 1 parsed, 2 the
// reference would raise (a non-numeric items[k], e.g. more than seven fields, or an empty items[4][1:]) — the
// kernels turn 2 into AQC_ERR_ARG only if the record actually reaches the bubble stage, like the exception upstream.
__device__ __forceinline__ bool is_digit(uint8_t c) { return c >= '0' && c <= '9'; }

// int() of name[a:b): digits only; returns false for anything else; values beyond int32 saturate (no circle can
// match such a lane / tile, and such a coordinate is outside every circle)
__device__ __forceinline__ bool parse_uint(const uint8_t* name, int a, int b, int32_t& out) {
    if (b <= a) return false;
    unsigned long long v = 0;
    for (int i = a; i < b; ++i) {
        if (!is_digit(name[i])) return false;
        v = v * 10ull + (unsigned long long)(name[i] - '0');
        if (v > 0x7fffffffull) v = 0x7fffffffull;
    }
    out = (int32_t)v;
    return true;
}

__global__ __launch_bounds__(TXT_BLOCK) void parse_names_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ name_off,
                                                                const uint32_t* __restrict__ name_len, const unsigned long long* __restrict__ n_dev,
                                                                int32_t* __restrict__ lane_out, int32_t* __restrict__ tile_out,
                                                                int32_t* __restrict__ x_out, int32_t* __restrict__ y_out,
                                                                uint8_t* __restrict__ ok_out) {
    const uint64_t r = (uint64_t)blockIdx.x * TXT_BLOCK + threadIdx.x;
    if (r >= *n_dev) return;                              // (FrameOut::n: the grid is sized for the most records the chunk could hold)
    const uint8_t* name = text + name_off[r];
    const int len = (int)name_len[r];
    int m_s = -1, m_e = -1;
    for (int s = 0; s < len && m_s < 0; ++s) {
        if (is_space(name[s])) continue;
        int E = s;
        while (E < len && !is_space(name[E])) ++E;           // the match stays inside this blank-delimited token
        for (int e1 = E - 1; e1 > s && m_s < 0; --e1) {
            if (name[e1] != ':') continue;
            int p = e1 + 1;
            while (p < E && is_digit(name[p])) ++p;
            if (p == e1 + 1 || p >= E || name[p] != ':') continue;
            const int s3 = p + 1;
            for (int e3 = E - 1; e3 > s3 && m_s < 0; --e3) {
                if (name[e3] != ':') continue;
                int q = e3 + 1;
                bool good = true;
                for (int g = 0; g < 3 && good; ++g) {
                    const int a = q;
                    while (q < E && is_digit(name[q])) ++q;
                    if (q == a || q >= E || name[q] != ':') good = false;
                    else ++q;
                }
                if (good) {
                    const int a = q;
                    while (q < E && is_digit(name[q])) ++q;
                    if (q > a) { m_s = s; m_e = q; }
                }
            }
        }
    }
    int32_t lane = 0, tile = 0, x = 0, y = 0;
    uint8_t ok = 0;
    if (m_s >= 0) {
        // items = match.split(':') : the 4th..7th field from the LEFT
        int fs[8], fe[8], nf = 0, a = m_s;
        for (int i = m_s; i <= m_e && nf < 8; ++i) {
            if (i == m_e || name[i] == ':') { fs[nf] = a; fe[nf] = i; ++nf; a = i + 1; }
        }
        ok = 1;
        if (!parse_uint(name, fs[3], fe[3], lane) || !parse_uint(name, fs[4] + 1, fe[4], tile) || !parse_uint(name, fs[5], fe[5], x) ||
            !parse_uint(name, fs[6], fe[6], y))
            ok = 2;
    }
    lane_out[r] = lane; tile_out[r] = tile; x_out[r] = x; y_out[r] = y; ok_out[r] = ok;
}

// ---- formatting -----------------------------------------------------------------------------------------------------
__device__ __constant__ char FLAG_TEXT[AQC_N_FLAGS][12] = {"GOOD", "BADBCD1", "BADBCD2", "BADTRIM1", "BADTRIM2", "BADBBL",
                                                            "BADLEN", "BADPOL", "BADLQC", "BADNCT", "BADDIFF", "BADMISMATCH"};
__device__ __constant__ int FLAG_TEXT_LEN[AQC_N_FLAGS] = {4, 7, 7, 8, 8, 6, 6, 6, 6, 6, 7, 11};

struct TextFile {
    const uint8_t* text;
    const uint32_t *seq_off, *qual_off;
    const uint32_t *seq_len;      // bit 31 (LEN_IRR): the quality line has a length of its own -> qual_len / qview
    const uint32_t *name_off, *name_len, *plus_off, *plus_len;      // plus_len: bit 31 = LEN_IRR again
    const uint32_t *qual_len;     // bit 31: the quality line's '\n' follows it directly (frame_records_kernel)
    const uint32_t *qview;        // final quality view (start | length << 16) of the records marked LEN_IRR, left by the verdict kernels
};

// The slice of the QUALITY line that record r of this file writes, when that line is not as long as the sequence line (LEN_IRR):
// every slice upstream is a python slice of each string by its own length, the final view is what the verdict kernel left in
// qview; getOverlap (preprocesser.py:78-84) takes r[3][len(r[3]) - overlap_len:] — a NEGATIVE start counts from the end.
__device__ __forceinline__ void irregular_quality_slice(const TextFile& tf, uint64_t r, int plain, int overlap_pass, int ovl, int& qst, int& qlen) {
    if (plain) { qst = 0; qlen = (int)(tf.qual_len[r] & QLEN_MASK); return; }
    const uint32_t qv = tf.qview[r];
    const int vs = (int)(qv & 0xffffu), vl = (int)(qv >> 16);
    qst = vs; qlen = vl;
    if (overlap_pass) {
        const int k = vl - ovl;
        if (k >= 0) { qst = vs + k; qlen = ovl; }
        else if (-k <= vl) { qst = vs + vl + k; qlen = -k; }
    }
}

struct FormatView {
    TextFile f[2];
    const aqc_result* results;
    int paired;
    int barcode;          // options.barcode: moveBarcodeToName (barcodeprocesser.py:34-45) rewrites the names
    int barcode_length;
    int store_overlap;    // --store_overlap: third stream with the overlapped tails of good pairs (preprocesser.py:78-84,614-616)
    int plain;            // index files (-7 / -5): records are written whole (no trim, no edits, no barcode move); only
                          // the verdicts — of the read pairs in `results` — route them and rename the bad ones
    int verdict_paired;   // the verdicts belong to read PAIRS (overlap stream exists)
    int spans;            // aqc_format_spans: good records that go out as their own bytes are NOT copied (they already stand in the
                          // chunk the caller framed): stream 0 holds only the good records that had to be rebuilt, and every
                          // record that is not such a "whole" record leaves an event (SpanEvent) saying where it stood
    uint32_t consumed[2]; // bytes of each file's chunk that the framed records take (the end of the last record)
    uint64_t n_framed;    // records framed into the slot (>= the n being formatted)
    int fused;            // the verdict kernel placed every record itself and copied the whole good ones (aqc_fast.hpp, FUSE): fstate[file][r] =
                          // the record's offset inside its batch's share of its stream | bit 31: already written; fbatch[2 b], [2 b + 1] =
                          // the bytes of the good / bad streams up to and including batch b (2 bits of state | file 0: 31 bits | file 1: 31 bits)
    const uint32_t* fstate[2];
    const unsigned long long* fbatch;
    int fbatch_shift;     // records per batch = 1 << fbatch_shift
};
constexpr uint32_t FMT_FUSED_DONE = 0x80000000u, FMT_FUSED_PATCH = 0x40000000u, FMT_FUSED_OFF = 0xffffu;      // (bit 30: written, but for the walk's byte patches)

// event k of a file = the k-th record (in order) that is bad or had to be rebuilt: it stood at chunk bytes [in_start, in_start +
// in_len) and contributes out_len bytes to stream 0 (0: a bad record).  The good output of the file is, in order: the chunk's
// bytes up to event 0 | out_len bytes of stream 0 | the chunk's bytes behind event 0 up to event 1 | ... up to the end of record n - 1.
struct SpanEvent { uint32_t in_start, in_len, out_len; };

// A good record that is written as its own bytes: not trimmed, not renamed, no edit of the walk in this mate, and all four lines
// followed directly by their '\n' in the chunk (QLEN_CONTIG) — the bulk of a run without trimming.
__device__ __forceinline__ bool record_is_whole(const FormatView& v, const TextFile& t, uint64_t r, int file, const uint4& w0) {
    if (v.plain || v.barcode || (int)(w0.x & 0xffu) != AQC_GOOD) return false;
    const uint32_t slw = t.seq_len[r];
    const uint32_t st = file == 0 ? (w0.x >> 16) : (w0.y >> 16), len = file == 0 ? (w0.y & 0xffffu) : (w0.z & 0xffffu);
    if (st != 0u || len != slw) return false;                      // (a mate marked LEN_IRR never equals its length word)
    if (!(t.qual_len[r] & QLEN_CONTIG)) return false;
    const int n_edits = (int)((w0.x >> 8) & 0xffu);
    if (n_edits) {
        const uint4 w1 = *(reinterpret_cast<const uint4*>(v.results + r) + 1);
        const unsigned long long e_lo = ((unsigned long long)w1.y << 32) | w1.x, e_hi = ((unsigned long long)w1.w << 32) | w1.z;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            if (e < n_edits) {
                const int bit = 40 * e + 16;                       // the edit's kind byte
                const unsigned int kind = (unsigned int)((bit < 64 ? e_lo >> bit : e_hi >> (bit - 64)) & 0xffu);
                if (kind == AQC_EDIT_MASK || (kind == AQC_EDIT_FIX_R1 && file == 0) || (kind == AQC_EDIT_FIX_R2 && file == 1)) return false;
            }
        }
    }
    return true;
}

// does record r go to the overlap stream?  paired, GOOD, overlap_len > 30 and every mismatch of the overlap was
// corrected (distance == 0 or distance == corrected bases, preprocesser.py:614)
__device__ __forceinline__ bool in_overlap_stream(const FormatView& v, const uint4& w0, const uint4& w1) {
    if (!v.store_overlap || !v.verdict_paired || (int)(w0.x & 0xffu) != AQC_GOOD) return false;
    const int ovl = (int)(w0.w & 0xffffu), dist = (int)(w0.w >> 16), n_edits = (int)((w0.x >> 8) & 0xffu);
    if (ovl <= 30) return false;
    const unsigned long long e_lo = ((unsigned long long)w1.y << 32) | w1.x, e_hi = ((unsigned long long)w1.w << 32) | w1.z;
    int corrected = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        if (e < n_edits) {
            const int bit = 40 * e + 16;                       // the edit's kind byte
            const unsigned int kind = (unsigned int)((bit < 64 ? e_lo >> bit : e_hi >> (bit - 64)) & 0xffu);
            corrected += (kind == AQC_EDIT_FIX_R1 || kind == AQC_EDIT_FIX_R2) ? 1 : 0;
        }
    }
    return dist == 0 || dist == corrected;
}


// moveBarcodeToName for one read: the name becomes '@' + bases[0:b] + name[first ':' :]; b is the detected barcode
// length for pairs (preprocesser.py:452), the design length for single-end input (:444).  Records flagged
// BADBCD1 / BADBCD2 keep their names.  Returns b (bases moved, clipped to the read) or -1 when the name stays.
__device__ __forceinline__ int moved_barcode_len(const FormatView& v, int file, int flag, uint32_t barcode_byte, uint32_t seq_len) {
    if (!v.barcode || flag == AQC_BADBCD1 || flag == AQC_BADBCD2) return -1;
    const int code = file == 0 ? (int)(barcode_byte & 15u) : (int)(barcode_byte >> 4);
    const int b = v.paired ? code - 2 + v.barcode_length : v.barcode_length;
    return min(max(b, 0), (int)seq_len);
}

// name.find(':') over a name of nlen bytes, 16 bytes per step (an unaligned 16-byte load, the exact zero-byte test on name ^ "::::"); nlen - 1
// when there is none — find() == -1 slices the last character (barcodeprocesser.py:41).  Reads up to 15 bytes behind the name: text.
// (rounds 2 - 5: a byte load per character — the sizing pass of a barcode run took 0.35 ms per 6 M records, six times the plain run's)
__device__ __forceinline__ uint32_t find_colon(const uint8_t* name, uint32_t nlen) {
    for (uint32_t i = 0; i < nlen; i += 16u) {
        const uint4 v = load16u(name + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        unsigned long long z[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t x0 = w[2 * h] ^ 0x3a3a3a3au, x1 = w[2 * h + 1] ^ 0x3a3a3a3au;
            const uint32_t z0 = ~(((x0 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x0) & 0x80808080u, z1 = ~(((x1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x1) & 0x80808080u;
            z[h] = ((unsigned long long)z1 << 32) | z0;
        }
        const uint32_t at = z[0] ? (uint32_t)(__builtin_ctzll(z[0]) >> 3) : z[1] ? 8u + (uint32_t)(__builtin_ctzll(z[1]) >> 3) : 16u;
        if (at < 16u && i + at < nlen) return i + at;
    }
    return nlen - 1u;
}

// bytes of (file, stream) that record r contributes: the sizes of all three streams of one file at once
// (sz[0] good, sz[1] bad, sz[2] overlap); the name's first ':' is only searched when a barcode was moved
__device__ __forceinline__ void fmt_sizes(const FormatView& v, uint64_t r, int file, uint32_t sz[3], uint32_t& event) {
    const uint4 w0 = *reinterpret_cast<const uint4*>(v.results + r);
    const int flag = (int)(w0.x & 0xffu);
    const TextFile& t = v.f[file];
    event = 0u;
    bool whole = false;                                     // spans mode: the record stays where it is, stream 0 does not get it
    if (v.spans) {
        whole = record_is_whole(v, t, r, file, w0);
        event = whole ? 0u : 1u;
        if (whole && !v.store_overlap) { sz[0] = sz[1] = sz[2] = 0u; return; }
    }
    uint32_t len = file == 0 ? (w0.y & 0xffffu) : (w0.z & 0xffffu);
    const uint32_t plw = t.plus_len[r];
    // (the sequence line's own length only where it is needed: index records, the barcode move, a quality line of another length)
    const uint32_t slw = (v.plain || v.barcode || (plw & LEN_IRR)) ? t.seq_len[r] : 0u;
    if (v.plain) len = slw & LEN_MASK;                      // index records go out whole
    uint32_t nlen = t.name_len[r];
    if (v.barcode && !v.plain) {
        const uint32_t bc = reinterpret_cast<const uint8_t*>(v.results + r)[31];
        const int b = moved_barcode_len(v, file, flag, bc, slw & LEN_MASK);
        if (b >= 0) {
            // name[str.find(':'):] — find() == -1 slices the last character
            const uint32_t cpos = find_colon(t.text + t.name_off[r], nlen);
            nlen = 1u + (uint32_t)b + (nlen - cpos);
        }
    }
    const uint32_t body = nlen + (plw & LEN_MASK) + 4u;
    uint32_t qlen = len;                                    // the quality line written beside `len` bases
    if (slw & LEN_IRR) {
        int qs_, ql_;
        irregular_quality_slice(t, r, v.plain, 0, 0, qs_, ql_);
        qlen = (uint32_t)ql_;
    }
    sz[0] = (flag == AQC_GOOD && !whole) ? body + len + qlen : 0u;
    sz[1] = flag == AQC_GOOD ? 0u : body + (uint32_t)FLAG_TEXT_LEN[flag] + len + qlen;
    sz[2] = 0u;
    if (v.store_overlap) {
        const uint4 w1 = *(reinterpret_cast<const uint4*>(v.results + r) + 1);
        if (in_overlap_stream(v, w0, w1)) {
            const uint32_t olen = v.plain ? len : (w0.w & 0xffffu);                                   // getOverlap: the last overlap_len bases
            uint32_t oq = v.plain ? qlen : olen;
            if ((slw & LEN_IRR) && !v.plain) {
                int qs_, ql_;
                irregular_quality_slice(t, r, 0, 1, (int)olen, qs_, ql_);
                oq = (uint32_t)ql_;
            }
            sz[2] = body + olen + oq;
        }
    }
}

constexpr int FMT_TILE = 128;           // records per workgroup of the plan pass (one thread per record)
constexpr int FMT_SUPER = 8;            // tiles per workgroup of the sizing pass = per entry of the second-level scan
constexpr int FMT_STREAMS = 8;          // file * 3 + {good, bad, overlap}, then (spans mode) the two files' event counts
constexpr int FMT_EVENT_STREAM = 6;

// Sizing pass (round 6: one workgroup per FMT_SUPER tiles, wave sums through DPP, no scans).  A wave takes the 64 records of half a
// tile; what it leaves behind, per stream q = file * 3 + stream:
//     tile_sum[q * n_tiles + tile]   the bytes of the tiles BEFORE this one inside its super-tile (a prefix the plan pass adds to ...)
//     super_sum[q * n_super + s]     ... the bytes of super-tile s, turned into the bytes before it by fmt_tile_bases_kernel
// (rounds 2 - 5: a workgroup of 128 threads per tile, four block scans of two barriers each to get four sums — 78 k workgroups and
//  0.26 ms per 10 M reads for 0.1 GB of input; the scan over all 78 k tile sums per stream, six workgroups, was another 0.11 ms)
__global__ __launch_bounds__(TXT_BLOCK) void fmt_tile_sums_kernel(FormatView v, uint64_t n, uint64_t n_tiles, uint64_t n_super,
                                                                  unsigned long long* __restrict__ tile_sum, unsigned long long* __restrict__ super_sum) {
    constexpr int HALVES = FMT_SUPER * FMT_TILE / WAVE;                   // waves' worth of records per super-tile
    constexpr int ROUNDS = FMT_SUPER * FMT_TILE / TXT_BLOCK;
    static_assert(FMT_TILE == 2 * WAVE && HALVES * WAVE == ROUNDS * TXT_BLOCK, "a tile is two waves' records");
    __shared__ uint32_t part[HALVES][FMT_STREAMS];
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const int nfiles = v.paired ? 2 : 1;
    for (int i = threadIdx.x; i < HALVES * FMT_STREAMS; i += TXT_BLOCK) (&part[0][0])[i] = 0u;
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * (FMT_SUPER * FMT_TILE);
#pragma unroll 1
    for (int it = 0; it < ROUNDS; ++it) {
        const uint64_t r = r0 + (uint64_t)it * TXT_BLOCK + threadIdx.x;
        const int half = it * (TXT_BLOCK / WAVE) + wave;
        if (r0 + (uint64_t)half * WAVE >= n) break;                      // (wave-uniform: nothing of this wave's records exists)
        for (int file = 0; file < nfiles; ++file) {
            uint32_t sz[3] = {0, 0, 0}, ev = 0;
            if (r < n) fmt_sizes(v, r, file, sz, ev);
            // (a record is < 64 KiB, a wave's sum < 4 MiB: int arithmetic; all 64 lanes are here)
            const int g = wave_sum_u((int)sz[0]), b = wave_sum_u((int)sz[1]);
            if (lane == 0) { part[half][file * 3 + 0] = (uint32_t)g; part[half][file * 3 + 1] = (uint32_t)b; }
            if (v.store_overlap) {
                const int o = wave_sum_u((int)sz[2]);
                if (lane == 0) part[half][file * 3 + 2] = (uint32_t)o;
            }
            if (v.spans) {                                               // streams 6, 7: the files' event counts
                const int e = wave_sum_u((int)ev);
                if (lane == 0) part[half][FMT_EVENT_STREAM + file] = (uint32_t)e;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < FMT_STREAMS) {
        const int q = threadIdx.x;
        unsigned long long run = 0;
        for (int t = 0; t < FMT_SUPER; ++t) {
            const uint64_t tile = (uint64_t)blockIdx.x * FMT_SUPER + t;
            if (tile < n_tiles) tile_sum[(uint64_t)q * n_tiles + tile] = run;
            run += (unsigned long long)part[2 * t][q] + part[2 * t + 1][q];
        }
        super_sum[(uint64_t)q * n_super + blockIdx.x] = run;
    }
}

// exclusive scan of each stream's super-tile sums (workgroup q handles stream q, eight entries per thread per round); totals to total_out[q]
__global__ __launch_bounds__(TXT_BLOCK) void fmt_tile_bases_kernel(unsigned long long* __restrict__ tile_sum, uint64_t n_tiles,
                                                                   unsigned long long* __restrict__ total_out) {
    __shared__ unsigned long long lds[4];
    constexpr int PER = 8;
    unsigned long long* ts = tile_sum + (uint64_t)blockIdx.x * n_tiles;
    unsigned long long carry = 0;
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += (uint64_t)TXT_BLOCK * PER) {
        const uint64_t t = t0 + (uint64_t)threadIdx.x * PER;
        unsigned long long val[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            val[k] = t + k < n_tiles ? ts[t + k] : 0ull;
            sum += val[k];
        }
        unsigned long long total;
        unsigned long long run = carry + block_excl_scan(sum, lds, total);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (t + k < n_tiles) ts[t + k] = run;
            run += val[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) total_out[blockIdx.x] = carry;
}

__device__ __forceinline__ void store16u(uint8_t* p, uint4 v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ uint4 load16u_t(const uint8_t* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// ---- one record of one file, as the writer sees it --------------------------------------------------------------------
// The output record is a sequence of pieces
//     '@' FLAG | barcode bases | name tail | \n | bases | \n | strand line | \n | qualities | \n
// each copied from a source: the chunk's text, or a small table of literals ("@BADPOL", "\n", "@").  Neighbouring pieces
// that are neighbours in the text as well are merged while the list is built, so an untrimmed good record is ONE piece
// (the record's own bytes), a tail-trimmed one three, a renamed (bad) one two more.
constexpr int FMT_MAXP = 10;
#ifndef AQC_GEN_ALIGN
#define AQC_GEN_ALIGN 1       // the general copy kernel's pieces of >= GRID_MIN bytes: windows on the source's grid, one work item more per piece
#endif
#ifndef AQC_GEN_SMALL
#define AQC_GEN_SMALL 1       // the general copy kernel's pieces of < 16 bytes: loaded with the windows, stored straight-line (0: rounds 2 - 5, a branch per size)
#endif
#ifndef AQC_GEN_DECODE_LDS
#define AQC_GEN_DECODE_LDS 1
#endif
constexpr uint32_t GRID_MIN = 48;
// work items of a piece of `len` bytes in the general copy kernel: a short piece is one, a long one a 16-byte window per item —
// on the source's grid (head window + aligned windows, the last one end-aligned) that is one more than len / 16 rounded up
__host__ __device__ constexpr uint32_t piece_items(uint32_t len) {
    return len >= 16u ? ((len + 15u) >> 4) + ((AQC_GEN_ALIGN && len >= GRID_MIN) ? 1u : 0u) : (len > 0u ? 1u : 0u);
}
constexpr uint32_t FMT_LIT_BIT = 0x80000000u;
struct FmtPiece {
    uint32_t src;          // byte offset from the file's text base; FMT_LIT_BIT: offset into FMT_LIT instead
    uint16_t dst, len;     // position in the output record, bytes
};
struct FmtTask {
    uint32_t pos;          // offset of the record in its output stream
    uint8_t stream;        // 0 good / 1 bad / 2 overlap, 0xff: not written in this pass
    uint8_t np, n_patch, pad_;
    uint16_t total, items; // bytes of the output record; work items (16-byte windows + short pieces)
    uint32_t patch[6];     // the walk's edits: output position | new byte << 16
    FmtPiece p[FMT_MAXP];
};

// literals: row f < 12 = "@" + FLAG text, row 12 = "\n", row 13 = "@"
__device__ uint8_t FMT_LIT[16][16] = {"@GOOD", "@BADBCD1", "@BADBCD2", "@BADTRIM1", "@BADTRIM2", "@BADBBL", "@BADLEN", "@BADPOL", "@BADLQC",
                                      "@BADNCT", "@BADDIFF", "@BADMISMATCH", "\n", "@", "", ""};

__device__ __forceinline__ void fmt_add(FmtTask& t, int& o, int len, uint32_t src) {
    if (len <= 0) return;
    if (t.np > 0) {
        FmtPiece& q = t.p[t.np - 1];
        if (!((q.src | src) & FMT_LIT_BIT) && q.src + q.len == src && (int)q.len + len <= 0xffff) {       // neighbours in the text too
            q.len = (uint16_t)(q.len + len);
            o += len;
            return;
        }
    }
    if (t.np < FMT_MAXP) {
        t.p[t.np].src = src; t.p[t.np].dst = (uint16_t)o; t.p[t.np].len = (uint16_t)len;
        t.np++;
    }
    o += len;
}

// the piece list of record r of `file` for this pass (main: good / bad, overlap_pass: the overlap stream)
__device__ inline void fmt_build(const FormatView& v, uint64_t r, int file, int overlap_pass, FmtTask& t, int* status) {
    const uint4 w0 = *reinterpret_cast<const uint4*>(v.results + r);
    const uint4 w1 = *(reinterpret_cast<const uint4*>(v.results + r) + 1);
    const int flag = (int)(w0.x & 0xffu);
    const int n_edits = v.plain ? 0 : (int)((w0.x >> 8) & 0xffu);
    t.np = 0; t.n_patch = 0; t.pad_ = 0; t.items = 0; t.total = 0;
    if (overlap_pass) t.stream = in_overlap_stream(v, w0, w1) ? 2 : 0xff;
    else t.stream = flag == AQC_GOOD ? 0 : 1;
    if (t.stream == 0xff) return;
    const int len1 = (int)(w0.y & 0xffffu), len2 = (int)(w0.z & 0xffffu), ovl = (int)(w0.w & 0xffffu);
    const TextFile& tf = v.f[file];
    const uint32_t name_off = tf.name_off[r], seq_off = tf.seq_off[r], plus_off = tf.plus_off[r], qual_off = tf.qual_off[r];
    const uint32_t slw = tf.seq_len[r];
    const int nlen = (int)tf.name_len[r], plen = (int)(tf.plus_len[r] & LEN_MASK), slen = (int)(slw & LEN_MASK);
    // the slice of the original read that is written: the final read, or its last overlap_len bases (getOverlap)
    const int cut = v.plain ? 0 : (overlap_pass ? (file == 0 ? len1 : len2) - ovl : 0);
    const int st = v.plain ? 0 : (file == 0 ? (int)(w0.x >> 16) : (int)(w0.y >> 16)) + cut;
    const int len = v.plain ? slen : (overlap_pass ? ovl : (file == 0 ? len1 : len2));
    const int flen = t.stream == 1 ? FLAG_TEXT_LEN[flag] : 0;
    // barcode moved into the name: '@' + [FLAG] + bases[0:mb] + name[cpos:]  (name[str.find(':'):]; find() == -1 slices the last character)
    const int mb = (v.barcode && !v.plain) ? moved_barcode_len(v, file, flag, w1.w >> 24, (uint32_t)slen) : -1;
    const int cpos = mb >= 0 ? (int)find_colon(tf.text + name_off, (uint32_t)nlen) : nlen - 1;
    const uint32_t NL = FMT_LIT_BIT | (12 * 16);
    int o = 0;
    // "@" + FLAG + name[1:] for a bad record (preprocesser.py:213-219), the name itself for a good one
    const bool renamed = t.stream == 1 || mb >= 0;
    if (renamed) fmt_add(t, o, 1 + flen, FMT_LIT_BIT | (uint32_t)((t.stream == 1 ? flag : 13) * 16));
    if (mb >= 0) { fmt_add(t, o, mb, seq_off); fmt_add(t, o, nlen - cpos, name_off + (uint32_t)cpos); }
    else if (renamed) fmt_add(t, o, nlen - 1, name_off + 1);
    else fmt_add(t, o, nlen, name_off);
    // the newlines come from the text where the text has them right there (no stripped whitespace), else from the table
    fmt_add(t, o, 1, seq_off == name_off + (uint32_t)nlen + 1 ? name_off + (uint32_t)nlen : NL);
    const int seq_dst = o;
    fmt_add(t, o, len, seq_off + (uint32_t)st);
    fmt_add(t, o, 1, plus_off == seq_off + (uint32_t)slen + 1 ? seq_off + (uint32_t)slen : NL);
    fmt_add(t, o, plen, plus_off);
    fmt_add(t, o, 1, qual_off == plus_off + (uint32_t)plen + 1 ? plus_off + (uint32_t)plen : NL);
    const int qual_dst = o;
    // the quality line: the same slice as the bases, unless this record's quality line has a length of its own
    int qst = st, qlen = len, qline = slen;
    const bool irr = (slw & LEN_IRR) != 0u;
    if (irr) {
        qline = (int)(tf.qual_len[r] & QLEN_MASK);
        irregular_quality_slice(tf, r, v.plain, overlap_pass, ovl, qst, qlen);
    }
    fmt_add(t, o, qlen, qual_off + (uint32_t)qst);
    fmt_add(t, o, 1, (qst + qlen == qline && (tf.qual_len[r] >> 31)) ? qual_off + (uint32_t)qline : NL);
    if (o > 0xffff) { atomicCAS(status, 0, AQC_ERR_UNSUPPORTED); t.stream = 0xff; return; }      // (a 64 KiB FASTQ record)
    t.total = (uint16_t)o;
    int items = 0;
    for (int k = 0; k < t.np; ++k) items += (int)piece_items(t.p[k].len);
    t.items = (uint16_t)items;
    // the walk's edits in this mate's slice coordinates -> byte patches of the output record
    const unsigned long long e_lo = ((unsigned long long)w1.y << 32) | w1.x, e_hi = ((unsigned long long)w1.w << 32) | w1.z;
    for (int e = 0; e < n_edits && e < 3; ++e) {
        const int bit = 40 * e;
        unsigned long long x = bit < 64 ? e_lo >> bit : 0ull;
        if (bit + 40 > 64) x |= bit < 64 ? e_hi << (64 - bit) : e_hi >> (bit - 64);
        const int oo = (int)(x & 0xffffu);
        const uint32_t kind = (uint32_t)(x >> 16) & 0xffu, base = (uint32_t)(x >> 24) & 0xffu, qual = (uint32_t)(x >> 32) & 0xffu;
        const int pp = (file == 0 ? len1 - ovl + oo : len2 - 1 - oo) - cut;
        if (irr) {
            // each string was edited at its OWN index (preprocesser.py:575-576,583-584,591-592): the bases at pp, the quality
            // view (start vs, length vl) at vl - overlap_len + o (a negative index wraps) resp. vl - 1 - o; two edits may meet
            // in one quality character — the later one stands
            const uint32_t qv = tf.qview[r];
            const int vs = (int)(qv & 0xffffu), vl = (int)(qv >> 16);
            int iq = file == 0 ? vl - ovl + oo : vl - 1 - oo;
            if (iq < 0) iq += vl;
            const int qp = vs + iq - qst;                    // in the slice that is written
            const bool mine = (kind == AQC_EDIT_FIX_R1 && file == 0) || (kind == AQC_EDIT_FIX_R2 && file == 1);
            if (mine && base && pp >= 0 && pp < len) t.patch[t.n_patch++] = (uint32_t)(seq_dst + pp) | (base << 16);
            if ((mine || kind == AQC_EDIT_MASK) && iq >= 0 && qp >= 0 && qp < qlen) {
                const uint32_t at = (uint32_t)(qual_dst + qp), val = kind == AQC_EDIT_MASK ? (uint32_t)'!' : qual;
                bool merged = false;
                for (int k = 0; k < (int)t.n_patch; ++k)
                    if ((t.patch[k] & 0xffffu) == at) { t.patch[k] = at | (val << 16); merged = true; }
                if (!merged) t.patch[t.n_patch++] = at | (val << 16);
            }
            continue;
        }
        if (pp < 0 || pp >= len) continue;
        if (kind == AQC_EDIT_MASK) t.patch[t.n_patch++] = (uint32_t)(qual_dst + pp) | ((uint32_t)'!' << 16);
        else if ((kind == AQC_EDIT_FIX_R1 && file == 0) || (kind == AQC_EDIT_FIX_R2 && file == 1)) {
            if (base) t.patch[t.n_patch++] = (uint32_t)(seq_dst + pp) | (base << 16);
            t.patch[t.n_patch++] = (uint32_t)(qual_dst + pp) | (qual << 16);
        }
    }
}

struct FormatOut {
    uint8_t* p[6];        // [file * 3 + stream]
};

template <int N>
__device__ __forceinline__ void copy_small(uint8_t* d, const uint8_t* s_, int len) {
    // len in [N, 2N): two overlapping N-byte moves
    uint8_t a[N], b[N];
    __builtin_memcpy(a, s_, N);
    __builtin_memcpy(b, s_ + len - N, N);
    __builtin_memcpy(d, a, N);
    __builtin_memcpy(d + len - N, b, N);
}

// ---- the writer: plan, then copy ------------------------------------------------------------------------------------------
// fmt_plan_kernel (thread = record, workgroup = tile of FMT_TILE records): the record's offset in its stream (block scans
// over the sizes, tile bases from fmt_tile_bases_kernel) and its piece list, written as a PLAN of six 16-byte words per
// (record, file):
//     q0  offset in the stream | stream, piece count, patch count | source of piece 0 | lengths of pieces 0, 1
//     q1  sources of pieces 1..4          q2  sources of pieces 5..7 | lengths of pieces 2, 3
//     q3  lengths of pieces 4..7 | cumulative work items of pieces 0..7 (a byte each)
//     q4  output offsets of pieces 1..7 (16 bits each) | total work items      q5  up to four byte patches (position | byte << 16)
// The search a copy lane would otherwise repeat (which piece is my work item in, where does that piece start in the
// output) is done here once per record.  Records with more than eight pieces / four patches / 32 work items keep their
// full FmtTask in an overflow array (q0.y bit 31).  A record that goes out as ONE piece — its own bytes — needs q0 only.
// fmt_copy_whole_kernel takes those, fmt_copy_kernel everything else (listed by the plan kernel).
constexpr uint32_t PLAN_SKIP = 0xffffffffu, PLAN_OVER = 0x80000000u;
constexpr unsigned int GEN_LISTS = 256;      // lists of "general" records (capacity gen_cap each), see fmt_plan_kernel
constexpr int PLAN_Q = 6;                    // 16-byte words per plan
constexpr int PLAN_MAXP = 8;
constexpr int GEN_PASSES = 2;                // work items per lane of the general copy kernel (32 lanes per record)

// the plans fmt_copy_whole_kernel takes: one piece of 16..512 bytes from the text — the record's own bytes — with up to four
// byte patches (a pair the correction walk edited is still its own bytes but for those: ~8 % of the records of a 2 x 150 run,
// which used to go through the general kernel, 0.7 of the text step's 4.6 ms)
__device__ __forceinline__ bool plan_is_whole(const uint4& q0) {
    return (q0.y & 0xff00ff00u) == 0x100u && ((q0.y >> 16) & 0xffu) <= 4u && !(q0.z & FMT_LIT_BIT) && (q0.w & 0xffffu) >= 16u && (q0.w & 0xffffu) <= 512u;
}

// the six plan words of a record's piece list `t` placed at `pos` of its stream (layout: above); false: the record does not fit a
// plan — more than eight pieces, four patches or 64 work items, or a patch inside a piece of < 16 bytes — q[0] then says PLAN_OVER
// and the caller keeps the full FmtTask in the overflow array
struct PlanWords { uint4 q0, q1, q2, q3, q4, q5; bool inline_ok; };
__device__ __forceinline__ PlanWords plan_words(const FmtTask& t, uint32_t pos) {
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint4 q0 = zero4, q1 = zero4, q2 = zero4, q3 = zero4, q4 = zero4, q5 = zero4;
    // inline: up to eight pieces, four byte patches (each inside a piece of >= 16 bytes), 64 work items (1 KiB)
    bool inline_ok = t.np <= PLAN_MAXP && t.n_patch <= 4 && t.items <= 32 * GEN_PASSES;
    for (int e = 0; e < (int)t.n_patch && inline_ok; ++e) {
        const int pp = (int)(t.patch[e] & 0xffffu);
        int o = 0;
        for (int k = 0; k < (int)t.np; ++k) {
            if (pp >= o && pp < o + (int)t.p[k].len && t.p[k].len < 16) inline_ok = false;
            o += t.p[k].len;
        }
    }
    if (inline_ok) {
        uint32_t src[PLAN_MAXP], len[PLAN_MAXP], cum[PLAN_MAXP], off[PLAN_MAXP];
        uint32_t ci = 0, doff = 0;
        for (int k = 0; k < PLAN_MAXP; ++k) {
            const bool in = k < (int)t.np;
            src[k] = in ? t.p[k].src : 0u;
            len[k] = in ? (uint32_t)t.p[k].len : 0u;
            off[k] = doff;
            ci += piece_items(len[k]);
            cum[k] = ci;
            doff += len[k];
        }
        q0 = make_uint4(pos, (uint32_t)t.stream | ((uint32_t)t.np << 8) | ((uint32_t)t.n_patch << 16), src[0], len[0] | (len[1] << 16));
        q1 = make_uint4(src[1], src[2], src[3], src[4]);
        q2 = make_uint4(src[5], src[6], src[7], len[2] | (len[3] << 16));
        q3 = make_uint4(len[4] | (len[5] << 16), len[6] | (len[7] << 16), cum[0] | (cum[1] << 8) | (cum[2] << 16) | (cum[3] << 24),
                          cum[4] | (cum[5] << 8) | (cum[6] << 16) | (cum[7] << 24));
        q4 = make_uint4(off[1] | (off[2] << 16), off[3] | (off[4] << 16), off[5] | (off[6] << 16), off[7] | (ci << 16));
        q5 = make_uint4(t.n_patch > 0 ? t.patch[0] : 0u, t.n_patch > 1 ? t.patch[1] : 0u, t.n_patch > 2 ? t.patch[2] : 0u,
                          t.n_patch > 3 ? t.patch[3] : 0u);
    } else q0 = make_uint4(pos, PLAN_OVER | (uint32_t)t.stream, 0, 0);
    return PlanWords{q0, q1, q2, q3, q4, q5, inline_ok};
}

// exclusive prefixes of two values per thread over the FMT_TILE threads of a plan workgroup (two waves): DPP lane scans, the other
// wave's totals through LDS (`lds`: 2 x 2 words of its own per call site — one barrier, none for reuse)
__device__ __forceinline__ void tile_excl_scan2(uint32_t a, uint32_t b, uint32_t (*lds)[2], uint32_t& ea, uint32_t& eb) {
    static_assert(FMT_TILE == 2 * WAVE, "two waves per plan workgroup");
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const int ia = wave_incl_sum((int)a, lane), ib = wave_incl_sum((int)b, lane);
    if (lane == WAVE - 1) { lds[wave][0] = (uint32_t)ia; lds[wave][1] = (uint32_t)ib; }
    __syncthreads();
    ea = (uint32_t)ia - a + (wave ? lds[0][0] : 0u);
    eb = (uint32_t)ib - b + (wave ? lds[0][1] : 0u);
}

__global__ __launch_bounds__(FMT_TILE) void fmt_plan_kernel(FormatView v, uint64_t n, uint64_t n_tiles, uint64_t n_super,
                                                            const unsigned long long* __restrict__ tile_base, const unsigned long long* __restrict__ super_base, int overlap_pass,
                                                            int* __restrict__ status, uint4* __restrict__ plan0, uint4* __restrict__ plan_patch, uint4* __restrict__ plan_gen,
                                                            FmtTask* __restrict__ over, uint32_t* __restrict__ gen_list,
                                                            unsigned int* __restrict__ n_gen, uint64_t gen_cap,
                                                            uint4* __restrict__ whole_plan, unsigned int* __restrict__ n_whole, uint8_t* __restrict__ good0, uint8_t* __restrict__ good1,
                                                            SpanEvent* __restrict__ events0, SpanEvent* __restrict__ events1) {
    __shared__ unsigned long long lds[4];
    __shared__ uint32_t lds2[2][2][2];          // [file][wave][good, bad]: tile_excl_scan2
    __shared__ FmtTask tasks[FMT_TILE];
    const int nfiles = v.paired ? 2 : 1;
    const uint64_t r = (uint64_t)blockIdx.x * FMT_TILE + threadIdx.x;
    // bytes of stream q before this tile: the super-tiles before (fmt_tile_bases_kernel) + the tiles before inside the super-tile
    auto base_of = [&](int q) -> unsigned long long {
        return super_base[(uint64_t)q * n_super + blockIdx.x / FMT_SUPER] + tile_base[(uint64_t)q * n_tiles + blockIdx.x];
    };
    for (int file = 0; file < nfiles; ++file) {
        FmtTask& t = tasks[threadIdx.x];
        t.stream = 0xff;
        t.total = 0;
        uint32_t sz[3] = {0, 0, 0};
        uint32_t event = 0;                     // spans mode, main pass: this record is not one that stays where it is
        if (r < n) {
            // (spans mode: a whole record needs no piece list — it is not copied — except for the overlap stream's slice of it)
            const uint32_t fs = v.fused ? v.fstate[file][r] : 0u;
            const bool whole = v.fused ? (fs & (FMT_FUSED_DONE | FMT_FUSED_PATCH)) == FMT_FUSED_DONE
                                       : v.spans && !overlap_pass && record_is_whole(v, v.f[file], r, file, *reinterpret_cast<const uint4*>(v.results + r));
            if (!whole) fmt_build(v, r, file, overlap_pass, t, status);
            event = (v.spans && !overlap_pass && !whole) ? 1u : 0u;
            if (!overlap_pass) sz[t.stream == 1 ? 1 : 0] = t.stream == 0xff ? 0u : (uint32_t)t.total;
            else sz[2] = t.stream == 2 ? (uint32_t)t.total : 0u;
        }
        if (v.spans && !overlap_pass) {
            // the event list of the file, in record order: where the record stood in the chunk, what it gives to stream 0
            unsigned long long te;
            const unsigned long long ee = block_excl_scan((unsigned long long)event, lds, te);
            if (event) {
                const TextFile& tf = v.f[file];
                const uint32_t a = tf.name_off[r];
                const uint32_t b = r + 1 < v.n_framed ? tf.name_off[r + 1] : v.consumed[file];
                SpanEvent* const ev = file == 0 ? events0 : events1;
                ev[base_of(FMT_EVENT_STREAM + file) + ee] = SpanEvent{a, b - a, t.stream == 0 ? (uint32_t)t.total : 0u};
            }
        }
        unsigned int pos;
        bool general = false, listed_whole = false;
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        uint4 q0 = zero4, q1 = zero4, q2 = zero4, q3 = zero4, q4 = zero4, q5 = zero4;      // (named, not an array: an array of them ended up in scratch)
        if (v.fused) {
            // (placed by the verdict kernel: no scans, no tile bases)
            pos = 0u;
            if (r < n && t.stream != 0xff) {
                const uint64_t b = r >> v.fbatch_shift;
                const unsigned long long w = b ? v.fbatch[2 * (b - 1) + (t.stream == 1 ? 1 : 0)] : 0ull;
                pos = (unsigned int)((file == 0 ? (w >> 31) : w) & 0x7fffffffull) + (v.fstate[file][r] & FMT_FUSED_OFF);
                if (v.fstate[file][r] & FMT_FUSED_DONE) {
                    // the verdict kernel wrote the record's own bytes; the walk's edits go on top (its launch is long complete)
                    uint8_t* const rec_out = (file == 0 ? good0 : good1) + pos;
                    for (int e = 0; e < (int)t.n_patch; ++e) rec_out[t.patch[e] & 0xffffu] = (uint8_t)(t.patch[e] >> 16);
                    t.stream = 0xff;
                }
            }
        } else if (!overlap_pass) {
            // good and bad records interleave: two scans, each record keeps the offset of the stream it goes to
            uint32_t eg, eb;
            tile_excl_scan2(sz[0], sz[1], lds2[file], eg, eb);
            pos = sz[1] ? (unsigned int)(base_of(file * 3 + 1) + eb) : (unsigned int)(base_of(file * 3 + 0) + eg);      // (offsets inside a chunk's stream fit 32 bits)
        } else {
            unsigned long long to;
            const unsigned long long eo = block_excl_scan((unsigned long long)sz[2], lds, to);
            pos = (unsigned int)(base_of(file * 3 + 2) + eo);
        }
        if (r < n) {
            const uint64_t ti = r * nfiles + file;
            q0 = make_uint4(pos, PLAN_SKIP, 0, 0);
            if (t.stream != 0xff) {
                t.pos = pos;
                const PlanWords pw = plan_words(t, pos);
                q0 = pw.q0; q1 = pw.q1; q2 = pw.q2; q3 = pw.q3; q4 = pw.q4; q5 = pw.q5;
                if (!pw.inline_ok) over[ti] = t;
            }
            // (spans / fused mode: the records that stay where they are / that the verdict kernel copied have no plan: PLAN_SKIP)
            // Text mode: nearly every record is one piece — fmt_copy_whole_kernel walks the dense plan0.  Spans / fused mode: few
            // are left (the pairs the walk edited) — their plans are LISTED (q0 with the file in bit 24 | the patches) and
            // fmt_copy_whole_list_kernel walks the lists (a dense walk over 10 M mostly empty plans cost 0.8 ms).
            // (a barcode run has no one-piece record — every good name is rewritten, every bad one flagged: no dense plans, and the
            //  host does not launch the kernel that would walk them: 0.33 ms per 6 M records of config 5 for nothing; should a plan
            //  be one piece after all it goes the general way)
            const bool no_dense = v.barcode && !v.plain;
            const bool whole = plan_is_whole(q0) && !no_dense;
            const bool sparse = v.spans || v.fused;
            if (!sparse && !no_dense) {
                plan0[ti] = q0;
                if (whole && (q0.y & 0x00ff0000u)) plan_patch[ti] = q5;        // (written and read for the patched records only)
            }
            general = q0.y != PLAN_SKIP && !whole;
            listed_whole = sparse && whole;
        }
        // the records fmt_copy_whole_kernel does not take are listed (one atomic per wave) for the general copy kernel
        {
            const unsigned long long gm = __ballot(general);
            if (gm) {
                unsigned int base = 0;
                // (GEN_LISTS separate lists, tile t appends to list t % GEN_LISTS: one shared counter would serialise
                //  ~10^5 same-address atomics in L2 — that alone cost 1.3 ms)
                const unsigned int lj = blockIdx.x % GEN_LISTS;
                if (lane_id() == 0) base = atomicAdd(&n_gen[lj], (unsigned int)__popcll(gm));
                base = (unsigned int)__shfl((int)base, 0, WAVE);
                if (general) {
                    // the general kernel reads its plans in list order: all six words go where the record is listed
                    const uint64_t slot = (uint64_t)lj * gen_cap + base + (unsigned int)__popcll(gm & ((1ull << lane_id()) - 1ull));
                    gen_list[slot] = (uint32_t)(r * nfiles + file);
                    uint4* const pg = plan_gen + slot * PLAN_Q;
                    pg[0] = q0; pg[1] = q1; pg[2] = q2; pg[3] = q3; pg[4] = q4; pg[5] = q5;
                }
            }
        }
        {
            const unsigned long long wm = __ballot(listed_whole);
            if (wm) {
                unsigned int base = 0;
                const unsigned int lj = blockIdx.x % GEN_LISTS;
                if (lane_id() == 0) base = atomicAdd(&n_whole[lj], (unsigned int)__popcll(wm));
                base = (unsigned int)__shfl((int)base, 0, WAVE);
                if (listed_whole) {
                    const uint64_t slot = (uint64_t)lj * gen_cap + base + (unsigned int)__popcll(wm & ((1ull << lane_id()) - 1ull));
                    whole_plan[2 * slot] = make_uint4(q0.x, q0.y | ((uint32_t)file << 24), q0.z, q0.w);
                    whole_plan[2 * slot + 1] = q5;
                }
            }
        }
        __syncthreads();            // (tasks[] is reused for the second file)
    }
}

#ifndef AQC_FMT_UNROLL
#define AQC_FMT_UNROLL 4
#endif
constexpr int FMT_UNROLL = AQC_FMT_UNROLL;
constexpr int COPY_BLOCK = 256;
#ifndef AQC_COPY_ALIGN
#define AQC_COPY_ALIGN 2      // window grid of the whole-record copy: 0 none (rounds 2 - 5), 1 the destination's, 2 the source's (copy_whole_tasks)
#endif

// Records that are ONE piece (untrimmed, unedited, not renamed: the bulk of a -f 0 -t 0 run): 32 lanes, window
// min(16 * lane, len - 16), load, store — as lean as a copy gets (tools/ubench/copy_rate.hip: this shape moves 6.9 GB in
// 1.4 ms without the plan read, 1.6 ms with it).
// pa[u]: the plan's first word (PLAN_SKIP: nothing), file[u]: its file, pq[u]: where its patch word stands
template <int NU>
__device__ __forceinline__ void copy_whole_tasks(const FormatView& v, const uint4 (&pa)[NU], const int (&file_of)[NU], const uint4* const (&pq)[NU],
                                                 const FormatOut& outs, int lane32) {
    uint4 val[NU];
    uint8_t* dptr[NU];
    bool on[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int file = file_of[u];
        const int len = (int)(pa[u].w & 0xffffu);
        uint8_t* const d0 = outs.p[file * 3 + (int)(pa[u].y & 0xffu)] + pa[u].x;
        const uint8_t* const s0 = v.f[file].text + pa[u].z;
        int nw = (len + 15) >> 4;
        int off = 16 * lane32;
#if AQC_COPY_ALIGN
        // round 6: the windows stand on the 16-byte grid of the SOURCE (AQC_COPY_ALIGN 2; 1: of the destination, measured slower than no
        // grid at all — profiles/r06_copy_window_grid.txt): lane 0 takes the record's first 16 bytes wherever they stand, lane k >= 1 the
        // k-th aligned window behind them, the last window end-aligned as before.  A wave's load instruction then touches every 64-byte
        // line once (off the grid each quad of lanes straddles two).  A record of > 496 bytes off the grid would take 33 windows: it
        // keeps the plain ones
        {
            const int a = (16 - (int)((AQC_COPY_ALIGN == 1 ? (uintptr_t)d0 : (uintptr_t)s0) & 15u)) & 15;
            const int nwa = 1 + ((len - a + 15) >> 4);
            const bool grid = a != 0 && nwa <= 32;
            nw = grid ? nwa : nw;
            off = grid && lane32 ? a + 16 * (lane32 - 1) : off;
        }
#endif
        on[u] = plan_is_whole(pa[u]) && lane32 < nw;
        off = min(off, len - 16);
        dptr[u] = d0 + off;
        if (on[u]) val[u] = load16u_t(s0 + off);
    }
    // the correction walk's edits: byte patches applied in registers (windows that overlap carry the same patch)
    // (a wave-level test per record in flight and per patch: 8 % of a 2 x 150 run's records carry one or two, and half the rounds of a wave
    //  — eight records — met one: all 16 patch slots were worked through for them)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const uint32_t np = on[u] ? (pa[u].y >> 16) & 0xffu : 0u;
        if (__ballot(np != 0)) {
            if (np) {
                const uint4 q5 = *pq[u];
                const uint32_t pt[4] = {q5.x, q5.y, q5.z, q5.w};
                const uint32_t wpos = (uint32_t)(dptr[u] - (outs.p[file_of[u] * 3 + (int)(pa[u].y & 0xffu)] + pa[u].x));      // (the window's place in the record)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e > 0 && __ballot((uint32_t)e < np) == 0) break;
                    const uint32_t i = (pt[e] & 0xffffu) - wpos;
                    const bool hit = (uint32_t)e < np && i < 16u;
                    const uint32_t sh = (i & 3u) * 8u, m = hit ? 0xffu << sh : 0u, cb = hit ? ((pt[e] >> 16) & 0xffu) << sh : 0u;
                    const uint32_t wd = i >> 2;
                    val[u].x = wd == 0 ? (val[u].x & ~m) | cb : val[u].x;
                    val[u].y = wd == 1 ? (val[u].y & ~m) | cb : val[u].y;
                    val[u].z = wd == 2 ? (val[u].z & ~m) | cb : val[u].z;
                    val[u].w = wd == 3 ? (val[u].w & ~m) | cb : val[u].w;
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
        if (on[u]) store16u(dptr[u], val[u]);      // (non-temporal loads / stores here: 3.95 -> 4.23 ms per step either way — measured, left out)
}

__global__ __launch_bounds__(COPY_BLOCK) void fmt_copy_whole_kernel(FormatView v, uint64_t n_tasks, const uint4* __restrict__ plan,
                                                                   const uint4* __restrict__ plan_patch, FormatOut outs) {
    const int nfiles = v.paired ? 2 : 1;
    const int lane32 = threadIdx.x & 31;
    // (grid-strided: the host launches one workgroup per 32 records, or — AQC_COPY_PERSIST builds — a fixed grid that loops)
    const uint64_t n_hw = ((uint64_t)gridDim.x * COPY_BLOCK) >> 5;
    for (uint64_t hw = ((uint64_t)blockIdx.x * COPY_BLOCK + threadIdx.x) >> 5; hw * FMT_UNROLL < n_tasks; hw += n_hw) {
        uint4 pa[FMT_UNROLL];
        int file_of[FMT_UNROLL];
        const uint4* pq[FMT_UNROLL];
#pragma unroll
        for (int u = 0; u < FMT_UNROLL; ++u) {
            const uint64_t ti = hw * FMT_UNROLL + u;
            pa[u] = make_uint4(0, PLAN_SKIP, 0, 0);
            if (ti < n_tasks) pa[u] = plan[ti];
            file_of[u] = nfiles == 2 ? (int)(ti & 1) : 0;
            pq[u] = plan_patch + ti;
        }
        copy_whole_tasks(v, pa, file_of, pq, outs, lane32);
    }
}

// ... the same for the LISTED one-piece records of a spans / fused format (fmt_plan_kernel): workgroup b walks list b % GEN_LISTS, whose
// plans stand in list order (two 16-byte words each: one coalesced load per round, then the text)
__global__ __launch_bounds__(COPY_BLOCK) void fmt_copy_whole_list_kernel(FormatView v, const uint4* __restrict__ whole_plans, FormatOut outs,
                                                                        const unsigned int* __restrict__ n_whole, uint64_t gen_cap) {
    const int lane32 = threadIdx.x & 31, hwi = threadIdx.x >> 5;
    const unsigned int lj = blockIdx.x % GEN_LISTS;
    const uint4* wp = whole_plans + 2 * (uint64_t)lj * gen_cap;
    const uint32_t n_list = n_whole[lj];
    constexpr uint32_t PER_WG = (COPY_BLOCK / 32) * FMT_UNROLL;
    const uint32_t stride = (gridDim.x / GEN_LISTS) * PER_WG;
    for (uint32_t r0 = (blockIdx.x / GEN_LISTS) * PER_WG; r0 < n_list; r0 += stride) {
        uint4 pa[FMT_UNROLL];
        int file_of[FMT_UNROLL];
        const uint4* pq[FMT_UNROLL];
#pragma unroll
        for (int u = 0; u < FMT_UNROLL; ++u) {
            const uint32_t idx = r0 + (uint32_t)(hwi * FMT_UNROLL + u);
            pa[u] = make_uint4(0, PLAN_SKIP, 0, 0);
            if (idx < n_list) pa[u] = wp[2 * (uint64_t)idx];
            file_of[u] = (int)((pa[u].y >> 24) & 1u);
            if (pa[u].y != PLAN_SKIP) pa[u].y &= ~(1u << 24);
            pq[u] = wp + 2 * (uint64_t)min(idx, n_list - 1u) + 1;
        }
        copy_whole_tasks(v, pa, file_of, pq, outs, lane32);
    }
}

// Everything else, from the plan kernel's lists: 32 lanes per plan, two plans in flight per half-wave, a lane owns one work
// item — a 16-byte window of a long piece (16-byte load + 16-byte store at any alignment, the piece's last window
// end-aligned) or a whole short piece.  Stages (list, plans, decode, loads, patches, stores) run over both plans so that
// each stage's memory operations travel together.
#ifndef AQC_GEN_U
#define AQC_GEN_U 2
#endif
constexpr int GEN_U = AQC_GEN_U;                           // plans per half-wave per round
constexpr int GEN_ROUND = (COPY_BLOCK / 32) * GEN_U;       // plans per workgroup per round
static_assert(GEN_ROUND * PLAN_Q <= COPY_BLOCK, "one 16-byte word per thread stages a round's plans");

// Everything that is not "one piece": the plans arrive in list order (fmt_plan_kernel), so a workgroup stages the 16 plans
// of a round with ONE coalesced load into LDS (1.5 KB) and its eight half-waves take two plans each: per lane up to four
// 16-byte windows in flight (32 lanes x 2 work items per plan), held as 32-bit offsets — the earlier version kept the
// plans in registers (48 of them), had two records in flight per half-wave and three dependent memory round trips per
// iteration (list -> plan -> data): 1.6 TB/s.
__global__ __launch_bounds__(COPY_BLOCK) void fmt_copy_kernel(FormatView v, const uint4* __restrict__ plan_gen, const FmtTask* __restrict__ over,
                                                             FormatOut outs, const uint32_t* __restrict__ gen_lists,
                                                             const unsigned int* __restrict__ n_gen, uint64_t gen_cap) {
    __shared__ uint4 s_plan[GEN_ROUND * PLAN_Q];
    __shared__ uint32_t s_ti[GEN_ROUND];
    __shared__ const uint8_t* s_ptr[8];                       // [0..5] output streams (file * 3 + stream), [6..7] the files' texts
    const int nfiles = v.paired ? 2 : 1;
    const int lane32 = threadIdx.x & 31, hwi = threadIdx.x >> 5;
    if (threadIdx.x < 6) s_ptr[threadIdx.x] = outs.p[threadIdx.x];
    if (threadIdx.x >= 6 && threadIdx.x < 8) s_ptr[threadIdx.x] = v.f[threadIdx.x - 6].text;
    // workgroup b works on list b % GEN_LISTS together with the other workgroups of that list
    const unsigned int lj = blockIdx.x % GEN_LISTS;
    const uint32_t* gen_list = gen_lists + (uint64_t)lj * gen_cap;
    const uint4* pg = plan_gen + (uint64_t)lj * gen_cap * PLAN_Q;
    const uint32_t n_list = n_gen[lj];
    const uint32_t stride = (gridDim.x / GEN_LISTS) * GEN_ROUND;
    for (uint32_t r0 = (blockIdx.x / GEN_LISTS) * GEN_ROUND; r0 < n_list; r0 += stride) {
        const uint32_t cnt = min((uint32_t)GEN_ROUND, n_list - r0);
        __syncthreads();                                     // (the previous round's plans are no longer read)
        if (threadIdx.x < cnt * PLAN_Q) s_plan[threadIdx.x] = pg[(uint64_t)r0 * PLAN_Q + threadIdx.x];
        if (threadIdx.x < cnt) s_ti[threadIdx.x] = gen_list[r0 + threadIdx.x];
        __syncthreads();
        // (round 6, measured twice and left out: the next round's plans prefetched into registers while this round is worked on — with the
        //  79-register kernel a wave less per SIMD, config 5 2.32 -> 2.41 ms; with this one 4.82 -> 4.80 ms per step: not what a round waits for)
        constexpr int NWIN = GEN_U * GEN_PASSES;          // windows in flight per lane: plan u, pass j -> slot u * GEN_PASSES + j
        uint4 val[NWIN];
        uint32_t so[NWIN], dof[NWIN], mw[NWIN];           // source offset (FMT_LIT_BIT: literal table), offset in the output stream,
                                                          // mode | position in the record << 5 | file * 3 + stream << 21 | file << 24
                                                          // (mode: 0 nothing, 16 a window, 1..15 a short piece of that many bytes)
        uint32_t more = 0;                                // bit u: plan u lives in the overflow array
        const uint8_t* const tp0 = s_ptr[6];
        const uint8_t* const tp1 = s_ptr[7];
        // (straight-line on purpose: with a branch per plan / window the instruction stream was one saveexec - branch - nop
        //  sequence after the other and the kernel spent its time on instruction latency)
#pragma unroll
        for (int u = 0; u < GEN_U; ++u) {
            const uint32_t idx = (uint32_t)(hwi * GEN_U + u);
            const bool live = idx < cnt;
            const uint4* P = s_plan + (live ? idx : 0u) * PLAN_Q;
#if AQC_GEN_DECODE_LDS
            // (round 6) the fields of my piece are READ from the plan in LDS at an address worked out from k — a 32-bit source, two 16-bit
            // reads — and k itself is a packed byte compare: the eight-way selects and seven compares per window they replace were a
            // third of this kernel's vector instructions
            const uint8_t* const pb = reinterpret_cast<const uint8_t*>(P);
            const uint4 q0 = P[0];
            const uint2 cumw = *reinterpret_cast<const uint2*>(pb + 56);                                 // q3.z, q3.w: cumulative items, a byte per piece
            const uint32_t q4w = *reinterpret_cast<const uint32_t*>(pb + 76);
            const bool ovf = live && (q0.y & PLAN_OVER) != 0;
            more |= ovf ? 1u << u : 0u;
            const uint32_t file = nfiles == 2 ? (s_ti[live ? idx : 0u] & 1u) : 0u;
            const uint32_t fs = file * 3u + (q0.y & 0xffu);
            const uint32_t items = (live && !ovf) ? q4w >> 16 : 0u;
            const unsigned long long cum = ((unsigned long long)cumw.y << 32) | cumw.x;
#pragma unroll
            for (int j = 0; j < GEN_PASSES; ++j) {
                const int w = u * GEN_PASSES + j;
                const uint32_t item = (uint32_t)lane32 + 32u * j;
                const bool on = item < items;
                // my piece: the pieces whose cumulative item count I am at or beyond (bytes 0..6; counts and items are < 128)
                const uint32_t rep = (item * 0x01010101u) | 0x80808080u;
                int k = __popc((rep - cumw.x) & 0x80808080u) + __popc((rep - cumw.y) & 0x00808080u);
                k = on ? k : 0;
                const int first_item = (int)(((cum << 8) >> (8 * k)) & 0xffu);
                const int lk = (int)*reinterpret_cast<const uint16_t*>(pb + (k < 2 ? 12 : 40) + 2 * k);                    // lengths: q0.w | q2.w, q3.x, q3.y
                const int dst_rd = (int)*reinterpret_cast<const uint16_t*>(pb + 62 + 2 * max(k, 1));                      // output offsets of pieces 1..7: q4
                const int dst_off = k == 0 ? 0 : dst_rd;                                                                  // (read, then chosen: no branch around the read)
                const uint32_t sk = *reinterpret_cast<const uint32_t*>(pb + (k == 0 ? 8 : 12 + 4 * k));                    // sources: q0.z | q1, q2.xyz
#else
            const uint4 q0 = P[0], q1 = P[1], q2 = P[2], q3 = P[3], q4 = P[4];
            const bool ovf = live && (q0.y & PLAN_OVER) != 0;
            more |= ovf ? 1u << u : 0u;
            const uint32_t file = nfiles == 2 ? (s_ti[live ? idx : 0u] & 1u) : 0u;
            const uint32_t fs = file * 3u + (q0.y & 0xffu);
            const uint32_t items = (live && !ovf) ? q4.w >> 16 : 0u;
            const unsigned long long cum = ((unsigned long long)q3.w << 32) | q3.z;
#pragma unroll
            for (int j = 0; j < GEN_PASSES; ++j) {
                const int w = u * GEN_PASSES + j;
                const uint32_t item = (uint32_t)lane32 + 32u * j;
                const bool on = item < items;
                // my piece: count the pieces whose cumulative item count I am at or beyond
                int k = 0;
#pragma unroll
                for (int jj = 0; jj < PLAN_MAXP - 1; ++jj) k += item >= (uint32_t)((cum >> (8 * jj)) & 0xffu) ? 1 : 0;
                k = on ? k : 0;
                const int first_item = (int)(((cum << 8) >> (8 * k)) & 0xffu);
                const uint32_t lw = k < 2 ? q0.w : k < 4 ? q2.w : k < 6 ? q3.x : q3.y;                  // lengths, two per word
                const int lk = (int)((lw >> (16 * (k & 1))) & 0xffffu);
                const uint32_t ow = k < 1 ? 0u : k < 3 ? q4.x : k < 5 ? q4.y : k < 7 ? q4.z : q4.w;     // output offsets of pieces 1..7
                const int dst_off = k == 0 ? 0 : (int)((ow >> (16 * ((k - 1) & 1))) & 0xffffu);
                const uint32_t sk = k == 0 ? q0.z : k == 1 ? q1.x : k == 2 ? q1.y : k == 3 ? q1.z : k == 4 ? q1.w : k == 5 ? q2.x : k == 6 ? q2.y : q2.z;
#endif
                // a long piece: my 16-byte window of it, the last one aligned to the piece's end; a short piece: all of it
                int off = 16 * ((int)item - first_item);
#if AQC_GEN_ALIGN
                // (round 6) a piece of >= GRID_MIN bytes has one item more (piece_items): its first window where the piece starts, the
                // others on the 16-byte grid of the SOURCE, so that a load instruction touches each 64-byte line once
                {
                    const uint32_t tb = (uint32_t)(uintptr_t)(file ? tp1 : tp0);
                    const int a = (int)((0u - (tb + sk)) & 15u);
                    const bool grid = lk >= (int)GRID_MIN && !(sk & FMT_LIT_BIT) && off;
                    off += (a - 16) & -(int)grid;                                    // (arithmetic, not a branch around eight instructions)
                }
#endif
                off = lk >= 16 ? min(off, lk - 16) : 0;
                so[w] = sk + (uint32_t)off;
                dof[w] = q0.x + (uint32_t)(dst_off + off);
                mw[w] = on ? ((uint32_t)min(lk, 16) | ((uint32_t)(dst_off + off) << 5) | (fs << 21) | (file << 24)) : 0u;
            }
        }
        const uint8_t* const lit = &FMT_LIT[0][0];
        auto src_of = [&](int w) -> const uint8_t* {
            const uint8_t* base = ((mw[w] >> 24) & 1u) ? tp1 : tp0;
            base = (so[w] & FMT_LIT_BIT) ? lit : base;
            return base + (so[w] & ~FMT_LIT_BIT);
        };
        auto dst_of = [&](int w) -> uint8_t* { return const_cast<uint8_t*>(s_ptr[(mw[w] >> 21) & 7u]) + dof[w]; };
        // every lane loads (a lane without a window reads the first bytes of the text: harmless, and no branch)
        uint32_t any_small = 0;
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            const uint32_t md = mw[w] & 31u;
#if AQC_GEN_SMALL
            val[w] = gload_u128(md ? src_of(w) : tp0);       // (a short piece's 16 bytes too: the text is padded, a literal is a 16-byte row)
#else
            val[w] = load16u_t(md == 16u ? src_of(w) : tp0);
#endif
            any_small |= (md - 1u) < 15u ? 1u : 0u;
        }
        // the correction walk's edits: byte patches applied in registers (windows that overlap carry the same patch)
        {
            uint32_t np_[GEN_U];
            uint32_t any_patch = 0;
#pragma unroll
            for (int u = 0; u < GEN_U; ++u) {
                const uint32_t idx = (uint32_t)(hwi * GEN_U + u);
                np_[u] = (idx < cnt && !((more >> u) & 1u)) ? (s_plan[idx * PLAN_Q].y >> 16) & 0xffu : 0u;
                any_patch |= np_[u];
            }
            if (__ballot(any_patch != 0)) {
#pragma unroll
                for (int u = 0; u < GEN_U; ++u) {
                    // (a wave-level test per plan and per patch: a record carries two patches or none, and few records any)
                    if (__ballot(np_[u] != 0) == 0) continue;
                    const uint32_t idx = min((uint32_t)(hwi * GEN_U + u), cnt - 1u);
                    const uint4 q5 = s_plan[idx * PLAN_Q + 5];
                    const uint32_t pt[4] = {q5.x, q5.y, q5.z, q5.w};
#pragma unroll
                    for (int j = 0; j < GEN_PASSES; ++j) {
                        const int w = u * GEN_PASSES + j;
                        const bool win = (mw[w] & 31u) == 16u;
                        const uint32_t wpos = (mw[w] >> 5) & 0xffffu;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (e > 0 && __ballot((uint32_t)e < np_[u]) == 0) break;
                            const uint32_t i = (pt[e] & 0xffffu) - wpos;
                            const bool hit = win && (uint32_t)e < np_[u] && i < 16u;
                            const uint32_t sh = (i & 3u) * 8u, m = hit ? 0xffu << sh : 0u, cb = hit ? ((pt[e] >> 16) & 0xffu) << sh : 0u;
                            const uint32_t wd = i >> 2;
                            val[w].x = wd == 0 ? (val[w].x & ~m) | cb : val[w].x;
                            val[w].y = wd == 1 ? (val[w].y & ~m) | cb : val[w].y;
                            val[w].z = wd == 2 ? (val[w].z & ~m) | cb : val[w].z;
                            val[w].w = wd == 3 ? (val[w].w & ~m) | cb : val[w].w;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int w = 0; w < NWIN; ++w)
            if ((mw[w] & 31u) == 16u) gstore_u128(dst_of(w), val[w]);
#if AQC_GEN_SMALL
        // pieces of 1..15 bytes — a literal '@', a moved barcode, a stray newline: their bytes came with the windows' loads; 8 + 4 + 2 + 1
        // bytes stored as the length's bits say.  (Rounds 2 - 5 copied them behind the windows, a branch per size class with its own
        // load -> store round trip: two or three memory latencies per round of a barcode run, where every record has two of them.)
        if (__ballot(any_small != 0)) {
#pragma unroll
            for (int w = 0; w < NWIN; ++w) {
                const uint32_t md = mw[w] & 31u;
                const bool sm = (md - 1u) < 15u;
                uint8_t* const d = dst_of(w);
                const uint32_t i2 = (md >> 2) & 3u;
                uint32_t vx = val[w].x, vy = val[w].y, vz = val[w].z, vw = val[w].w;
                asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz), "+v"(vw));      // (values, not addresses: a select of loads would put val[] into scratch)
                const uint32_t pick = i2 == 0u ? vx : i2 == 1u ? vy : i2 == 2u ? vz : vw;      // the word of byte (md & 12)
                if (sm && (md & 8u)) gstore_u64(d, vx, vy);
                if (sm && (md & 4u)) gstore_u32(d + (md & 8u), (md & 8u) ? vz : vx);
                if (sm && (md & 2u)) gstore_u16(d + (md & 12u), (uint16_t)pick);
                if (sm && (md & 1u)) gstore_u8(d + (md & 14u), (uint8_t)(pick >> ((md & 2u) * 8u)));
            }
        }
#else
        if (__ballot(any_small != 0)) {                    // pieces of 1..15 bytes: a literal '@', a moved barcode, a stray newline
#pragma unroll
            for (int w = 0; w < NWIN; ++w) {
                const int md = (int)(mw[w] & 31u);
                if (md >= 16 || md == 0) continue;
                if (md >= 8) copy_small<8>(dst_of(w), src_of(w), md);
                else if (md >= 4) copy_small<4>(dst_of(w), src_of(w), md);
                else if (md >= 2) copy_small<2>(dst_of(w), src_of(w), md);
                else dst_of(w)[0] = src_of(w)[0];
            }
        }
#endif
        // ---- overflow records: any number of pieces / work items, piece by piece (records of more than 1 KiB, more than
        //      eight pieces or four patches)
        if (more) {
            for (int u = 0; u < GEN_U; ++u) {
                if (!((more >> u) & 1u)) continue;
                const uint64_t ti = s_ti[hwi * GEN_U + u];
                const FmtTask& t = over[ti];
                const int file = nfiles == 2 ? (int)(ti & 1) : 0;
                uint8_t* out0 = outs.p[file * 3 + (int)t.stream] + t.pos;
                const int np = (int)t.np;
                for (int k = 0; k < np; ++k) {
                    const uint32_t sk = t.p[k].src;
                    const int lk = (int)t.p[k].len;
                    const uint8_t* src = (sk & FMT_LIT_BIT) ? &FMT_LIT[0][0] + (sk & ~FMT_LIT_BIT) : v.f[file].text + sk;
                    uint8_t* dst = out0 + t.p[k].dst;
                    if (lk >= 16) {
                        for (int w0 = 16 * lane32; w0 < lk; w0 += 16 * 32) {
                            const int off = min(w0, lk - 16);
                            store16u(dst + off, load16u_t(src + off));
                        }
                    } else if (lane32 < lk) dst[lane32] = src[lane32];
                }
                if (t.n_patch) {
                    // byte patches on top of the copies (the copies of this wave are complete first)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_s_waitcnt(0);
                    if (lane32 < (int)t.n_patch) {
                        const uint32_t pt = t.patch[lane32];
                        out0[pt & 0xffffu] = (uint8_t)(pt >> 16);
                    }
                }
            }
        }
    }
}

// ---- round 6: the text mode's default writer — place + copy in one kernel, piece lists only for the records that need them ------
// (rounds 2 - 5: fmt_plan_kernel built the piece list of EVERY record — fmt_build, a 116-byte task in LDS, six plan words — wrote a
//  16-byte plan per record and file to HBM, and fmt_copy_whole_kernel read it back: 0.31 + 2.06 ms and 0.64 GB of plan traffic per
//  10 M reads, although 97 % of the records of a run without trimming go out as their own bytes.)
// A workgroup takes a tile of FMT_TILE records, thread = (record, file):
//   1. the record's bytes in its streams (fmt_sizes — the sizing pass's routine, so the two agree by construction), two DPP scans
//      per file, the tile's bases -> the record's place;
//   2. a good record that is its own bytes (record_is_whole's conditions, but edits of the walk in this mate become <= 4 byte
//      patches): its plan word (+ patch word) stays in LDS;  every other record is LISTED for fmt_plan_listed_kernel with its place;
//   3. the workgroup copies its own-bytes records: 32 lanes per record, four records in flight per half-wave (copy_whole_tasks).
// fmt_plan_listed_kernel then builds the piece lists / plans of the listed records only (thread = list entry), fmt_copy_kernel
// copies them as before.  Barcode runs (every name rewritten), index files, the overlap pass, spans and fused formats keep
// fmt_plan_kernel.
__device__ __forceinline__ int own_bytes_patches(const uint4& w0, const uint4& w1, int file, int len, int seq_dst, int qual_dst, uint32_t (&patch)[6]) {
    // fmt_build's edit loop for a record whose quality line is as long as its sequence line, written from base 0 (cut == 0)
    const int n_edits = (int)((w0.x >> 8) & 0xffu);
    const int len1 = (int)(w0.y & 0xffffu), len2 = (int)(w0.z & 0xffffu), ovl = (int)(w0.w & 0xffffu);
    const unsigned long long e_lo = ((unsigned long long)w1.y << 32) | w1.x, e_hi = ((unsigned long long)w1.w << 32) | w1.z;
    int np = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        if (e >= n_edits) break;
        const int bit = 40 * e;
        unsigned long long x = bit < 64 ? e_lo >> bit : 0ull;
        if (bit + 40 > 64) x |= bit < 64 ? e_hi << (64 - bit) : e_hi >> (bit - 64);
        const int oo = (int)(x & 0xffffu);
        const uint32_t kind = (uint32_t)(x >> 16) & 0xffu, base = (uint32_t)(x >> 24) & 0xffu, qual = (uint32_t)(x >> 32) & 0xffu;
        const int pp = file == 0 ? len1 - ovl + oo : len2 - 1 - oo;
        if (pp < 0 || pp >= len) continue;
        if (kind == AQC_EDIT_MASK) patch[np++] = (uint32_t)(qual_dst + pp) | ((uint32_t)'!' << 16);
        else if ((kind == AQC_EDIT_FIX_R1 && file == 0) || (kind == AQC_EDIT_FIX_R2 && file == 1)) {
            if (base) patch[np++] = (uint32_t)(seq_dst + pp) | (base << 16);
            patch[np++] = (uint32_t)(qual_dst + pp) | (qual << 16);
        }
    }
    return np;
}

constexpr int PC_BLOCK = 2 * FMT_TILE;       // thread = (record of the tile, file)
#ifndef AQC_PC_UNROLL
#define AQC_PC_UNROLL 4
#endif
constexpr int PC_UNROLL = AQC_PC_UNROLL;     // records in flight per half-wave in the copy phase
static_assert(PC_BLOCK == COPY_BLOCK, "the copy phase is fmt_copy_whole_kernel's");

#ifndef AQC_PC_WAVES
#define AQC_PC_WAVES 1
#endif
__global__ __launch_bounds__(PC_BLOCK, AQC_PC_WAVES) void fmt_place_copy_kernel(FormatView v, uint64_t n, uint64_t n_tiles, uint64_t n_super,
                                                                   const unsigned long long* __restrict__ tile_base, const unsigned long long* __restrict__ super_base,
                                                                   uint4* __restrict__ plan_gen, uint32_t* __restrict__ gen_list, unsigned int* __restrict__ n_gen,
                                                                   uint64_t gen_cap, FormatOut outs) {
    __shared__ uint4 s_q0[PC_BLOCK], s_q5[PC_BLOCK];
    __shared__ uint32_t s_tot[PC_BLOCK / WAVE][2];
    const int nfiles = v.paired ? 2 : 1;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const int file = wave >> 1;                                           // waves 0, 1: the tile's records of file 0; waves 2, 3: of file 1
    const uint64_t r = (uint64_t)blockIdx.x * FMT_TILE + (threadIdx.x & (FMT_TILE - 1));
    const bool have = file < nfiles && r < n;
    uint32_t sz[3] = {0, 0, 0}, ev = 0;
    uint4 q0 = make_uint4(0, PLAN_SKIP, 0, 0), q5 = make_uint4(0, 0, 0, 0);
    bool own = false;
    uint32_t own_src = 0, n_patch = 0;
    if (have) {
        fmt_sizes(v, r, file, sz, ev);
        const TextFile& tf = v.f[file];
        const uint4 w0 = *reinterpret_cast<const uint4*>(v.results + r);
        const uint32_t slw = tf.seq_len[r];
        const uint32_t st = file == 0 ? (w0.x >> 16) : (w0.y >> 16), len = file == 0 ? (w0.y & 0xffffu) : (w0.z & 0xffffu);
        // its own bytes: good, the whole read (a mate marked LEN_IRR never equals its length word), every line followed directly by
        // its '\n' — then sz[0] = name + bases + strand line + qualities + 4 is the distance from its name to behind its last '\n'
        if ((int)(w0.x & 0xffu) == AQC_GOOD && st == 0u && len == slw && (tf.qual_len[r] & QLEN_CONTIG) && sz[0] >= 16u && sz[0] <= 512u) {
            own = true;
            own_src = tf.name_off[r];
            if ((w0.x >> 8) & 0xffu) {
                const uint4 w1 = *(reinterpret_cast<const uint4*>(v.results + r) + 1);
                const int nlen = (int)tf.name_len[r], plen = (int)(tf.plus_len[r] & LEN_MASK);
                uint32_t patch[6] = {0, 0, 0, 0, 0, 0};
                n_patch = (uint32_t)own_bytes_patches(w0, w1, file, (int)len, nlen + 1, nlen + 1 + (int)len + 1 + plen + 1, patch);
                q5 = make_uint4(patch[0], patch[1], patch[2], patch[3]);
                own = n_patch <= 4u;                                      // (three corrections in one mate: the general kernel's overflow path)
            }
        }
    }
    // the record's place: exclusive prefixes over the tile's records of this file (two waves), good and bad apart
    const int ig = wave_incl_sum((int)sz[0], lane), ib = wave_incl_sum((int)sz[1], lane);
    if (lane == WAVE - 1) { s_tot[wave][0] = (uint32_t)ig; s_tot[wave][1] = (uint32_t)ib; }
    __syncthreads();
    uint32_t pos = 0;
    if (have) {
        const int q = file * 3 + (sz[1] ? 1 : 0);
        const unsigned long long base = super_base[(uint64_t)q * n_super + blockIdx.x / FMT_SUPER] + tile_base[(uint64_t)q * n_tiles + blockIdx.x];
        const uint32_t ex = sz[1] ? (uint32_t)ib - sz[1] + ((wave & 1) ? s_tot[wave - 1][1] : 0u) : (uint32_t)ig - sz[0] + ((wave & 1) ? s_tot[wave - 1][0] : 0u);
        pos = (uint32_t)(base + ex);                                      // (offsets inside a chunk's stream fit 32 bits)
        if (own) q0 = make_uint4(pos, 0x100u | (n_patch << 16), own_src, sz[0]);      // stream 0, one piece: what fmt_build + plan_words make of it
    }
    s_q0[threadIdx.x] = q0;
    s_q5[threadIdx.x] = q5;
    // everything else is listed for fmt_plan_listed_kernel, with its place (one atomic per wave; the lists: see fmt_plan_kernel)
    {
        const bool general = have && !own;
        const unsigned long long gm = __ballot(general);
        if (gm) {
            unsigned int b0 = 0;
            const unsigned int lj = blockIdx.x % GEN_LISTS;
            if (lane == 0) b0 = atomicAdd(&n_gen[lj], (unsigned int)__popcll(gm));
            b0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
            if (general) {
                const uint64_t slot = (uint64_t)lj * gen_cap + b0 + (unsigned int)__popcll(gm & ((1ull << lane) - 1ull));
                gen_list[slot] = (uint32_t)(r * nfiles + file);
                plan_gen[slot * PLAN_Q] = make_uint4(pos, PLAN_SKIP, 0, 0);
            }
        }
    }
    __syncthreads();
    // the copy: a half-wave per record, FMT_UNROLL records in flight, plans and patch words from LDS
    const int lane32 = threadIdx.x & 31, hwi = threadIdx.x >> 5;
    const int n_plans = nfiles * FMT_TILE;
    constexpr int PER_ROUND = (PC_BLOCK / 32) * PC_UNROLL;
#pragma unroll 1
    for (int p0 = 0; p0 < n_plans; p0 += PER_ROUND) {
        uint4 pa[PC_UNROLL];
        int file_of[PC_UNROLL];
        const uint4* pq[PC_UNROLL];
#pragma unroll
        for (int u = 0; u < PC_UNROLL; ++u) {
            const int p = p0 + hwi * PC_UNROLL + u;                       // (< PC_BLOCK: a single-end tile's upper half says PLAN_SKIP)
            pa[u] = s_q0[p];
            file_of[u] = p / FMT_TILE;
            pq[u] = &s_q5[p];
        }
        copy_whole_tasks(v, pa, file_of, pq, outs, lane32);
    }
}

// the piece lists and plans of the records fmt_place_copy_kernel listed: workgroup b works on list b % GEN_LISTS, thread = entry
__global__ __launch_bounds__(FMT_TILE) void fmt_plan_listed_kernel(FormatView v, uint4* __restrict__ plan_gen, FmtTask* __restrict__ over,
                                                                   const uint32_t* __restrict__ gen_lists, const unsigned int* __restrict__ n_gen,
                                                                   uint64_t gen_cap, int* __restrict__ status) {
    __shared__ FmtTask tasks[FMT_TILE];
    const int nfiles = v.paired ? 2 : 1;
    const unsigned int lj = blockIdx.x % GEN_LISTS;
    const uint32_t n_list = n_gen[lj];
    const uint32_t stride = (gridDim.x / GEN_LISTS) * FMT_TILE;
    FmtTask& t = tasks[threadIdx.x];
    for (uint32_t i = (blockIdx.x / GEN_LISTS) * FMT_TILE + threadIdx.x; i < n_list; i += stride) {
        const uint64_t slot = (uint64_t)lj * gen_cap + i;
        const uint64_t ti = gen_lists[slot];
        const uint32_t pos = plan_gen[slot * PLAN_Q].x;
        const uint64_t r = nfiles == 2 ? ti >> 1 : ti;
        const int file = nfiles == 2 ? (int)(ti & 1) : 0;
        fmt_build(v, r, file, 0, t, status);
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        uint4 q0 = make_uint4(pos, PLAN_SKIP, 0, 0), q1 = zero4, q2 = zero4, q3 = zero4, q4 = zero4, q5 = zero4;
        if (t.stream != 0xff) {
            t.pos = pos;
            const PlanWords pw = plan_words(t, pos);
            q0 = pw.q0; q1 = pw.q1; q2 = pw.q2; q3 = pw.q3; q4 = pw.q4; q5 = pw.q5;
            if (!pw.inline_ok) over[ti] = t;
        }
        uint4* const pg = plan_gen + slot * PLAN_Q;
        pg[0] = q0; pg[1] = q1; pg[2] = q2; pg[3] = q3; pg[4] = q4; pg[5] = q5;
    }
}

}  // namespace aqc
