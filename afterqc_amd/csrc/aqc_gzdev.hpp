// aqc_gzdev.hpp — gzip output built ON THE DEVICE: the good / bad / overlap text streams that aqc_format leaves in HBM
// become BGZF-compatible gzip members there, so `-z` output costs no host CPU and a third of the PCIe traffic
// (fastq.py:63-68 + `--compression`, after.py:91-92 upstream: Python's gzip module on the one CPU thread).
//
// The host's CPU budget is what bounds the pipe's own codec (aqc_deflate.cpp): the MI355X boxes grant a container 16 CPUs'
// worth of run time, ~4 GB/s of deflate in total — which the inflate side of a .gz -> .gz run needs for itself.
//
//   member     <= 0xff00 = 255 x 256 bytes of text -> one gzip member with the BGZF extra field, ONE final dynamic-Huffman
//              block.  A workgroup of 256 threads per member, a thread per 255-byte segment: the text is staged in LDS, every
//              thread tokenises its own segment (so no token crosses a segment border), a block scan of the segments' bit
//              counts places them, the threads write their bits (whole words stored, border words OR-ed in).
//   matches    two kinds, both found without any search structure: RUNS (distance 1: quality strings) and the SAME COLUMN OF
//              THE LINE FOUR LINES UP (the previous record's name / strand line; the distance is that line's start + column).
//              A match is taken when its code is shorter than the literals' it replaces, by the actual code lengths.
//   code       ONE literal/length + distance code per stream and launch, shared by all its members (every symbol has a code):
//              gz_hist_kernel tokenises a sample of the members, the HOST builds the code from the counts with the same
//              routines as its own encoder (aqc_deflate.cpp: code_lengths / canonical codes / run-length header) and hands back
//              the codes + the ready-made block header bits.
//   CRC-32     per segment byte-wise from a 256-entry table, then a tree of "advance by 255 * 2^k bytes" operators (32 x 32 bit
//              matrices precomputed on the host); the member's first four bytes are complemented instead of starting the
//              register at ~0, which makes leading padding harmless (the last member of a stream is right-aligned in the grid).
//   compaction member sizes -> exclusive scan per stream -> contiguous stream (gz_pack_kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aqc_kernels.hpp"
#include "aqc_text.hpp"

namespace aqc {

constexpr int GZ_TEXT = 0xff00;                  // text bytes per member = GZ_THREADS * GZ_SEG
constexpr int GZ_THREADS = 256, GZ_SEG = 255;
constexpr int GZ_SLOT = 65536;                   // staging bytes per member; the member starts at byte 2 (deflate data 4-aligned)
constexpr int GZ_MAX_LINES = 4096;
constexpr int GZ_HDR_WORDS = 192;

struct GzCodebookDev {
    uint32_t lit[286];       // bit-reversed code | length << 16
    uint32_t dist[30];
    uint32_t hdr[GZ_HDR_WORDS];   // BFINAL, BTYPE and the dynamic header, LSB first
    uint32_t hdr_bits;
    uint32_t pad_[3];
};

struct GzCrcTables {
    uint32_t byte_table[256];
    uint32_t shift[8][32];   // operator "advance the register by 255 * 2^k zero bytes", column j = image of bit j
};

struct GzJob {
    const uint8_t* text[6];
    uint64_t bytes[6];
    uint32_t first_block[7];     // members of stream q: [first_block[q], first_block[q + 1])
    uint8_t* stage;              // n_members * GZ_SLOT
    uint32_t* sizes;             // bytes of each member
    const GzCodebookDev* code;   // [6]
    uint32_t* hist;              // [6][320]: 286 literal/length counts, then 30 distance counts
    const GzCrcTables* crc;
    uint32_t member_text;        // text bytes per member: GZ_TEXT (gz_encode_kernel) or GZW_TEXT (gz_encode_wave_kernel)
    uint32_t slot_bytes;         // staging bytes per member: GZ_SLOT or GZW_SLOT
    uint8_t* packed[6];          // contiguous streams (gz_pack_kernel)
    uint64_t* offsets;           // [n_members] start of each member inside its packed stream; totals in total[6]
    uint64_t* total;
};

// base value and number of extra bits of length symbol 257 + i / distance symbol i (RFC 1951 3.2.5) in closed form: a table in
// constant memory indexed per lane is a vector load from memory per token (measured on the inflate side, aqc_gunzip_dev.hpp)
__device__ __forceinline__ int gz_len_extra(int i) { return (i < 8 || i == 28) ? 0 : (i - 4) >> 2; }
__device__ __forceinline__ int gz_len_base(int i) { return i == 28 ? 258 : i < 8 ? 3 + i : 3 + ((4 + (i & 3)) << ((i - 4) >> 2)); }
__device__ __forceinline__ int gz_dist_extra(int i) { return i < 4 ? 0 : (i - 2) >> 1; }
__device__ __forceinline__ int gz_dist_base(int i) { return i < 4 ? 1 + i : 1 + ((2 + (i & 1)) << ((i - 2) >> 1)); }

// length 3..258 -> length symbol - 257; distance 1..32768 -> distance symbol (closed forms: no tables)
__device__ __forceinline__ int gz_len_sym(int len) {
    if (len == 258) return 28;
    if (len < 11) return len - 3;
    const int v = len - 3;                        // 8 .. 254
    const int e = 31 - __clz(v) - 2;              // extra bits: v in [8,16) -> 1, [16,32) -> 2, ...
    return 4 + 4 * e + ((v >> e) & 3);
}
__device__ __forceinline__ int gz_dist_sym(int d) {
    if (d < 5) return d - 1;
    const int v = d - 1;                          // >= 4
    const int e = 31 - __clz(v) - 1;              // extra bits
    return 2 + 2 * e + ((v >> e) & 1);
}

// which member is this, and of which stream
__device__ __forceinline__ int gz_stream_of(const GzJob& J, uint32_t member) {
    int q = 0;
#pragma unroll
    for (int k = 1; k < 6; ++k) q += member >= J.first_block[k] ? 1 : 0;
    return q;
}

struct GzSinkHist {
    uint32_t* h;     // LDS [320]
    __device__ __forceinline__ void lit(uint32_t b) { atomicAdd(&h[b], 1u); }
    __device__ __forceinline__ void match(int len, int dist) { atomicAdd(&h[257 + gz_len_sym(len)], 1u); atomicAdd(&h[286 + gz_dist_sym(dist)], 1u); }
};
struct GzSinkSize {
    const uint32_t* lc; const uint32_t* dc;     // LDS codebook
    uint32_t bits = 0;
    __device__ __forceinline__ void lit(uint32_t b) { bits += lc[b] >> 16; }
    __device__ __forceinline__ void match(int len, int dist) {
        const int ls = gz_len_sym(len), ds = gz_dist_sym(dist);
        bits += (lc[257 + ls] >> 16) + gz_len_extra(ls) + (dc[ds] >> 16) + gz_dist_extra(ds);
    }
};
struct GzSinkEmit {
    const uint32_t* lc; const uint32_t* dc;
    uint32_t* out;            // deflate data as words (zeroed)
    uint32_t w;               // next word
    unsigned long long acc = 0;
    int nacc;
    bool first = true;
    __device__ __forceinline__ void flush_words() {
        while (nacc >= 32) {
            const uint32_t v = (uint32_t)acc;
            if (first) { atomicOr(&out[w], v); first = false; }
            else out[w] = v;
            ++w;
            acc >>= 32;
            nacc -= 32;
        }
    }
    __device__ __forceinline__ void put(uint32_t code, int len) { acc |= (unsigned long long)code << nacc; nacc += len; flush_words(); }
    __device__ __forceinline__ void lit(uint32_t b) { const uint32_t c = lc[b]; put(c & 0xffffu, (int)(c >> 16)); }
    __device__ __forceinline__ void match(int len, int dist) {
        const int ls = gz_len_sym(len), ds = gz_dist_sym(dist);
        const uint32_t a = lc[257 + ls], d = dc[ds];
        put(a & 0xffffu, (int)(a >> 16));
        if (gz_len_extra(ls)) put((uint32_t)(len - gz_len_base(ls)), gz_len_extra(ls));
        put(d & 0xffffu, (int)(d >> 16));
        if (gz_dist_extra(ds)) put((uint32_t)(dist - gz_dist_base(ds)), gz_dist_extra(ds));
    }
    __device__ __forceinline__ void finish() { if (nacc > 0) atomicOr(&out[w], (uint32_t)acc); }
};

// What the size pass decided, for the emit pass: bit i of `start` = a token begins at byte a + i of the segment (its length is the
// distance to the next one: 1 = literal, >= 3 = match), bit i of `col` = that match copies the column four lines up (else the
// byte before: distance 1), bit i of `nl` = byte a + i is a line end.  Four 64-bit words each, never indexed by a variable (that
// would put them in scratch memory): the emit pass costs a few instructions per TOKEN instead of the whole search per byte again.
struct GzTokMask {
    unsigned long long start[4] = {0, 0, 0, 0}, col[4] = {0, 0, 0, 0}, nl[4] = {0, 0, 0, 0};
};
__device__ __forceinline__ void gz_mask_set(unsigned long long (&m)[4], int i) {
    const unsigned long long bit = 1ull << (i & 63);
    const int w = i >> 6;
    m[0] |= w == 0 ? bit : 0ull; m[1] |= w == 1 ? bit : 0ull; m[2] |= w == 2 ? bit : 0ull; m[3] |= w == 3 ? bit : 0ull;
}
__device__ __forceinline__ bool gz_mask_test(const unsigned long long (&m)[4], int i) {
    const int w = i >> 6;
    const unsigned long long v = w == 0 ? m[0] : w == 1 ? m[1] : w == 2 ? m[2] : m[3];
    return (v >> (i & 63)) & 1ull;
}
// set bits below position i
__device__ __forceinline__ int gz_mask_count_below(const unsigned long long (&m)[4], int i) {
    int c = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int k = i - 64 * w;                          // bits of word w below i: all (k >= 64), none (k <= 0), the low k
        const unsigned long long v = k >= 64 ? m[w] : k <= 0 ? 0ull : m[w] & ((1ull << k) - 1ull);
        c += __popcll(v);
    }
    return c;
}

// Tokens of the segment s[a, b) of a member's text s[0, n): the same decisions whatever the sink.  EXACT: matches are judged
// by the code lengths (size / emit passes, which therefore agree); otherwise by fixed thresholds (the sampling pass, before a
// code exists).  ls[0 .. n_lines) = line starts (ls[l + 1] - 1 is line l's '\n'), line = index of the line position `a` is in.
template <bool EXACT, class Sink, bool REC = false>
__device__ __forceinline__ void gz_tokenize(const uint8_t* s, int a, int b, int line, const uint16_t* ls, int n_lines, bool use_lines,
                                            const uint32_t* lc, const uint32_t* dc, Sink& sink, GzTokMask* rec = nullptr) {
    int p = a;
    while (p < b) {
        const uint32_t c0 = s[p];
        int best_len = 0, best_dist = 0, best_gain = 0;
        // a run of the byte before
        if (p > 0 && c0 == s[p - 1]) {
            int r = 1;
            const int lim = min(b - p, 258);
            while (r < lim && s[p + r] == c0) ++r;
            if (r >= 3) {
                int gain;
                if (EXACT) {
                    const int lsym = gz_len_sym(r);
                    gain = r * (int)(lc[c0] >> 16) - (int)((lc[257 + lsym] >> 16) + gz_len_extra(lsym) + (dc[0] >> 16));
                } else gain = r >= 5 ? r : 0;
                if (gain > best_gain) { best_gain = gain; best_len = r; best_dist = 1; }
            }
        }
        // the same column four lines up
        if (use_lines && line >= 4) {
            const int q0 = (int)ls[line - 4] + (p - (int)ls[line]);
            const int dist = p - q0;
            if (q0 < (int)ls[line - 3] && dist <= 32768 && s[q0] == c0) {
                int m = 1, lit_bits = EXACT ? (int)(lc[c0] >> 16) : 0;
                const int lim = min(b - p, 258);
                while (m < lim && s[p + m] == s[q0 + m]) { if (EXACT) lit_bits += (int)(lc[s[p + m]] >> 16); ++m; }
                if (m >= 3) {
                    int gain;
                    if (EXACT) {
                        const int lsym = gz_len_sym(m), dsym = gz_dist_sym(dist);
                        gain = lit_bits - (int)((lc[257 + lsym] >> 16) + gz_len_extra(lsym) + (dc[dsym] >> 16) + gz_dist_extra(dsym));
                    } else gain = m >= 6 ? m : 0;
                    if (gain > best_gain) { best_gain = gain; best_len = m; best_dist = dist; }
                }
            }
        }
        if (REC) {
            gz_mask_set(rec->start, p - a);
            if (best_len && best_dist != 1) gz_mask_set(rec->col, p - a);
        }
        if (best_len) {
            sink.match(best_len, best_dist);
            if (use_lines)
                for (int i = 0; i < best_len; ++i) {
                    const bool e = s[p + i] == '\n';
                    line += e ? 1 : 0;
                    if (REC && e) gz_mask_set(rec->nl, p + i - a);
                }
            p += best_len;
        } else {
            sink.lit(c0);
            const bool e = use_lines && c0 == '\n';
            line += e ? 1 : 0;
            if (REC && e) gz_mask_set(rec->nl, p - a);
            ++p;
        }
        if (line >= n_lines) line = n_lines - 1;       // (cannot happen: kept so that a bad table can never index past ls)
    }
}

// The emit pass: the tokens the size pass recorded, replayed.  line_a = the line byte `a` is in.
template <class Sink>
__device__ __forceinline__ void gz_replay(const uint8_t* s, int a, int b, int line_a, const uint16_t* ls, const GzTokMask& M, Sink& sink) {
    const int nseg = b - a;
    int prev = -1;                       // start of the token waiting for its end
    auto emit = [&](int at, int len) {
        if (len == 1) { sink.lit(s[a + at]); return; }
        int dist = 1;
        if (gz_mask_test(M.col, at)) {
            const int p = a + at;
            const int line = line_a + gz_mask_count_below(M.nl, at);
            dist = p - ((int)ls[line - 4] + (p - (int)ls[line]));
        }
        sink.match(len, dist);
    };
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned long long m = M.start[w];
        while (m) {
            const int at = 64 * w + (int)__builtin_ctzll(m);
            m &= m - 1;
            if (prev >= 0) emit(prev, at - prev);
            prev = at;
        }
    }
    if (prev >= 0) emit(prev, nseg - prev);
}

// text of one member into LDS + its line table.  The member's n bytes are RIGHT-aligned in the 256 x 255 grid: thread t owns
// the text bytes [seg_a, seg_b) = grid positions [255 t, 255 (t + 1)) minus the padding in front.
struct alignas(16) GzStage {
    uint8_t text[GZ_TEXT + 16];
    uint16_t ls[GZ_MAX_LINES + 8];
    uint32_t scan[GZ_THREADS / WAVE + 1];
    uint32_t n_lines;
};

__device__ __forceinline__ uint32_t gz_block_excl_scan(uint32_t v, uint32_t* lds /* [5] */, uint32_t& total) {
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == WAVE - 1) lds[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < GZ_THREADS / WAVE; ++w) {
        const uint32_t t = lds[w];
        if (w < wave) base += t;
        total += t;
    }
    return base + inc - v;
}

// returns this thread's segment [a, b), the line its first byte is in, and whether the line table is usable
__device__ __forceinline__ void gz_stage_member(GzStage& S, const uint8_t* src, int n, int& a, int& b, int& line, bool& use_lines) {
    for (int i = threadIdx.x * 16; i < n; i += GZ_THREADS * 16) {
        const uint4 v = load16u_t(src + i);        // (the formatted streams are followed by 64 readable bytes)
        *reinterpret_cast<uint4*>(S.text + i) = v;
    }
    const int pad = GZ_TEXT - n;
    a = max(0, (int)threadIdx.x * GZ_SEG - pad);
    b = max(0, ((int)threadIdx.x + 1) * GZ_SEG - pad);
    __syncthreads();
    uint32_t nl = 0;
    for (int p = a; p < b; ++p) nl += S.text[p] == '\n' ? 1u : 0u;
    uint32_t total;
    const uint32_t before = gz_block_excl_scan(nl, S.scan, total);
    // line starts: ls[0] = 0, ls[k] = position behind the k-th '\n'
    use_lines = total + 2 <= GZ_MAX_LINES;
    if (threadIdx.x == 0) { S.ls[0] = 0; S.n_lines = total + 1; }
    if (use_lines) {
        uint32_t k = before + 1;
        for (int p = a; p < b; ++p)
            if (S.text[p] == '\n') S.ls[k++] = (uint16_t)(p + 1);
    }
    line = (int)before;
    __syncthreads();
    if (use_lines && threadIdx.x == 0) S.ls[total + 1] = (uint16_t)n;      // (sentinel: end of the last line)
    __syncthreads();
}

// ---- sampling pass: symbol counts of some members of every stream -------------------------------------------------------------
constexpr int GZ_SAMPLES = 16;           // members sampled per stream (evenly spaced)

__global__ __launch_bounds__(GZ_THREADS) void gz_hist_kernel(GzJob J) {
    __shared__ GzStage S;
    __shared__ uint32_t h[320];
    // workgroup (q, k) samples member k * stride of stream q
    const int q = blockIdx.x / GZ_SAMPLES, k = blockIdx.x % GZ_SAMPLES;
    const uint32_t nb = (uint32_t)((J.bytes[q] + GZ_TEXT - 1) / GZ_TEXT);      // (its own pieces of GZ_TEXT bytes, whatever the encoder's members are)
    const uint32_t stride = nb > GZ_SAMPLES ? nb / GZ_SAMPLES : 1u;
    if ((uint32_t)k * stride >= nb) return;
    const uint32_t local = (uint32_t)k * stride;
    const uint64_t off = (uint64_t)local * GZ_TEXT;
    const int n = (int)min<uint64_t>(GZ_TEXT, J.bytes[q] - off);
    for (int i = threadIdx.x; i < 320; i += GZ_THREADS) h[i] = 0;
    int a, b, line; bool use_lines;
    gz_stage_member(S, J.text[q] + off, n, a, b, line, use_lines);
    GzSinkHist sink{h};
    gz_tokenize<false>(S.text, a, b, line, S.ls, (int)S.n_lines, use_lines, nullptr, nullptr, sink);
    __syncthreads();
    for (int i = threadIdx.x; i < 320; i += GZ_THREADS)
        if (h[i]) atomicAdd(&J.hist[q * 320 + i], h[i]);
}

// ---- one gzip member per workgroup ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GZ_THREADS) void gz_encode_kernel(GzJob J) {
    __shared__ GzStage S;
    __shared__ uint32_t lc[286], dc[30];
    __shared__ uint32_t crc_tab[256];
    __shared__ uint32_t crc_part[GZ_THREADS];
    const uint32_t member = blockIdx.x;
    const int q = gz_stream_of(J, member);
    const uint32_t local = member - J.first_block[q];
    const uint64_t off = (uint64_t)local * GZ_TEXT;
    const int n = (int)min<uint64_t>(GZ_TEXT, J.bytes[q] - off);
    const GzCodebookDev& cb = J.code[q];
    for (int i = threadIdx.x; i < 286; i += GZ_THREADS) lc[i] = cb.lit[i];
    if (threadIdx.x < 30) dc[threadIdx.x] = cb.dist[threadIdx.x];
    crc_tab[threadIdx.x] = J.crc->byte_table[threadIdx.x];
    int a, b, line; bool use_lines;
    gz_stage_member(S, J.text[q] + off, n, a, b, line, use_lines);
    // ---- sizes -> positions
    GzSinkSize sz{lc, dc};
    GzTokMask tm;
    gz_tokenize<true, GzSinkSize, true>(S.text, a, b, line, S.ls, (int)S.n_lines, use_lines, lc, dc, sz, &tm);
    uint32_t total_bits;
    const uint32_t my_bit = gz_block_excl_scan(sz.bits, S.scan, total_bits);
    const uint32_t hdr_bits = cb.hdr_bits;
    const uint32_t eob = lc[256];
    const uint32_t all_bits = hdr_bits + total_bits + (eob >> 16);
    uint32_t dbytes = (all_bits + 7) >> 3;
    const bool stored = dbytes >= (uint32_t)n + 5u;
    if (stored) dbytes = (uint32_t)n + 5u;
    uint8_t* const mem = J.stage + (uint64_t)member * GZ_SLOT + 2;       // the member; its deflate data at +18 is 4-byte aligned
    uint32_t* const dwords = reinterpret_cast<uint32_t*>(mem + 18);
    if (!stored) {
        const uint32_t nw = (all_bits + 31) >> 5;
        for (uint32_t i = threadIdx.x; i <= nw; i += GZ_THREADS) dwords[i] = 0;
        __threadfence();
        __syncthreads();
        GzSinkEmit em{lc, dc, dwords, (hdr_bits + my_bit) >> 5};
        em.nacc = (int)((hdr_bits + my_bit) & 31u);
        gz_replay(S.text, a, b, line, S.ls, tm, em);
        if (threadIdx.x == GZ_THREADS - 1) em.put(eob & 0xffffu, (int)(eob >> 16));       // end of block, behind the last segment
        em.finish();
        // the shared block header: whole words stored, the last (partial) one OR-ed in
        for (uint32_t i = threadIdx.x; i * 32 < hdr_bits; i += GZ_THREADS) {
            const uint32_t v = cb.hdr[i];
            if ((i + 1) * 32 <= hdr_bits) atomicOr(&dwords[i], v);         // (word 0 .. may be shared with nobody, OR is simply safe)
            else atomicOr(&dwords[i], v & ((1u << (hdr_bits & 31u)) - 1u));
        }
    } else {
        // incompressible with this code: one stored block (BFINAL = 1, BTYPE = 0, LEN, ~LEN, bytes)
        if (threadIdx.x == 0) {
            mem[18] = 1;
            mem[19] = (uint8_t)n; mem[20] = (uint8_t)(n >> 8); mem[21] = (uint8_t)~n; mem[22] = (uint8_t)(~n >> 8);
        }
        for (int i = threadIdx.x; i < n; i += GZ_THREADS) mem[23 + i] = S.text[i];
    }
    // ---- CRC-32 of the member's text: raw CRCs of the 256 grid segments (the first four text bytes complemented), combined
    {
        uint32_t c = 0;
        for (int p = a; p < b; ++p) {
            uint32_t x = S.text[p];
            if (p < 4) x ^= 0xffu;                          // = starting the register at 0xffffffff, in a form leading zeros do not disturb
            c = crc_tab[(c ^ x) & 0xffu] ^ (c >> 8);
        }
        crc_part[threadIdx.x] = c;
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
            const int step = 1 << k;
            uint32_t r = 0;
            const bool active = (threadIdx.x & (2 * step - 1)) == 0;
            if (active) {
                const uint32_t left = crc_part[threadIdx.x];
                const uint32_t* M = J.crc->shift[k];
                for (int j = 0; j < 32; ++j) r ^= ((left >> j) & 1u) ? M[j] : 0u;
                r ^= crc_part[threadIdx.x + step];
            }
            __syncthreads();
            if (active) crc_part[threadIdx.x] = r;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        uint32_t crc = ~crc_part[0];
        if (n < 4) {
            // (fewer than four bytes: the complement trick does not apply; do it the plain way)
            uint32_t c = 0xffffffffu;
            for (int p = 0; p < n; ++p) c = crc_tab[(c ^ S.text[p]) & 0xffu] ^ (c >> 8);
            crc = ~c;
        }
        const uint32_t bsize = 18u + dbytes + 8u;
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        for (int i = 0; i < 16; ++i) mem[i] = hdr[i];
        mem[16] = (uint8_t)((bsize - 1) & 0xffu);
        mem[17] = (uint8_t)((bsize - 1) >> 8);
        uint8_t* t = mem + 18 + dbytes;
        for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)((uint32_t)n >> (8 * k)); }
        J.sizes[member] = bsize;
    }
}

// ---- member sizes -> offsets inside each packed stream (one workgroup per stream) ------------------------------------------------
__global__ __launch_bounds__(GZ_THREADS) void gz_offsets_kernel(GzJob J) {
    __shared__ unsigned long long lds[4];
    const int q = blockIdx.x;
    const uint32_t m0 = J.first_block[q], m1 = J.first_block[q + 1];
    unsigned long long carry = 0;
    for (uint32_t t0 = m0; t0 < m1; t0 += GZ_THREADS) {
        const uint32_t m = t0 + threadIdx.x;
        const unsigned long long v = m < m1 ? (unsigned long long)J.sizes[m] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan(v, lds, total);
        if (m < m1) J.offsets[m] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) J.total[q] = carry;
}

__global__ __launch_bounds__(GZ_THREADS) void gz_pack_kernel(GzJob J) {
    const uint32_t member = blockIdx.x;
    const int q = gz_stream_of(J, member);
    const uint8_t* src = J.stage + (uint64_t)member * J.slot_bytes + 2;
    uint8_t* dst = J.packed[q] + J.offsets[member];
    const int n = (int)J.sizes[member];
    for (int i = threadIdx.x * 16; i < n; i += GZ_THREADS * 16) {
        if (i + 16 <= n) store16u(dst + i, load16u_t(src + i));
        else for (int k = i; k < n; ++k) dst[k] = src[k];
    }
}


// ---- round 6: one gzip member per WAVE, 64 bytes per step ------------------------------------------------------------------------
// gz_encode_kernel gives every thread 255 bytes of its own: 64 lanes, each in a state of its own (inside a run, comparing a column,
// emitting a literal) — the wave executes every path for every step, 142 K vector instructions per wave and 16 KB, and a
// .gz -> .gz run spends a third of the GPU's time in it (profiles/r06_gz_prof_summary.txt).  Here a wave owns a member of
// GZW_TEXT = 64 x 255 bytes and walks it in windows of 64 consecutive bytes, a lane per byte:
//   * what can match — the byte before (runs), the same column four lines up — is one comparison per lane and a ballot: a match's
//     length is the run of ones in that mask from the lane's bit on (masks of the next four windows are kept ahead, so a match may
//     be 258 long; a column match ends with its line);
//   * the greedy parse is a walk over the window's token starts with scalar steps (v_readlane of "my token ends at"); a window
//     without a match — the bases of a read — takes none: every lane is a literal;
//   * the tokens' bits are placed by a lane scan of their lengths and OR-ed into a 128-word staging ring in LDS, whole words go out
//     to the member with one coalesced store per lane.  The wave writes its member front to back: no sizing pass.
// Same format as before — BGZF-compatible members, ONE final dynamic block with the stream's shared code, stored when that is
// smaller — with members a quarter the size: +~2.5 % bytes (header and code per 16 KB, no column match in a member's first record).
constexpr int GZW_TEXT = 64 * 255;               // 16320 text bytes per member
constexpr int GZW_SLOT = 16896;                  // staging bytes per member (its text stored + headers, and a window's worth of slack)
constexpr int GZW_MAX_LINES = 512;
constexpr int GZW_RING = 128;                    // staging words

__device__ __forceinline__ int gzw_ctz64(unsigned long long x) { return x ? (int)__builtin_ctzll(x) : 64; }
// ones of the masks M[0..5) from bit l of M[0] on, up to the first zero
__device__ __forceinline__ int gzw_ones_from(const unsigned long long (&M)[5], int l) {
    int t = gzw_ctz64(~(M[0] >> l) | (l ? 0ull : 0ull));
    if (l == 0 && M[0] == ~0ull) t = 64;
    bool go = t >= 64 - l;
    t = min(t, 64 - l);
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        const int ck = gzw_ctz64(~M[k]);            // (wave-uniform)
        if (go) { t += ck; go = ck == 64; }
    }
    return t;
}
// zeros of the masks from bit l of M[0] on, up to the first one (the distance to the next set bit)
__device__ __forceinline__ int gzw_zeros_from(const unsigned long long (&M)[5], int l) {
    const unsigned long long x = M[0] >> l;
    int t = x ? (int)__builtin_ctzll(x) : 64 - l;
    bool go = x == 0;
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        const int ck = gzw_ctz64(M[k]);
        if (go) { t += ck; go = ck == 64; }
    }
    return t;
}

struct alignas(16) GzwStage {
    uint8_t text[GZW_TEXT + 64];
    uint16_t ls[GZW_MAX_LINES];
    uint32_t lc[286], dc[30];
    uint32_t crc_tab[256];
    uint32_t ring[GZW_RING];
};

__global__ __launch_bounds__(WAVE) void gz_encode_wave_kernel(GzJob J) {
    __shared__ GzwStage S;
    const int lane = (int)threadIdx.x;
    const uint32_t member = blockIdx.x;
    const int q = gz_stream_of(J, member);
    const uint32_t local = member - J.first_block[q];
    const uint64_t off = (uint64_t)local * GZW_TEXT;
    const int n = (int)min<uint64_t>(GZW_TEXT, J.bytes[q] - off);
    const GzCodebookDev& cb = J.code[q];
    for (int i = lane; i < 286; i += WAVE) S.lc[i] = cb.lit[i];
    if (lane < 30) S.dc[lane] = cb.dist[lane];
    for (int i = lane; i < 256; i += WAVE) S.crc_tab[i] = J.crc->byte_table[i];
    for (int i = lane; i < GZW_RING; i += WAVE) S.ring[i] = 0;
    {
        const uint8_t* src = J.text[q] + off;
        for (int i = lane * 16; i < n; i += WAVE * 16) *reinterpret_cast<uint4*>(S.text + i) = load16u_t(src + i);      // (64 readable bytes follow a stream)
        // what lies behind the member's end never matches anything
        for (int i = n + lane; i < ((n + 15) & ~15) + 64 && i < GZW_TEXT + 64; i += WAVE) S.text[i] = 0;
    }
    if (lane == 0) S.ls[0] = 0;
    __syncthreads();
    uint8_t* const mem = J.stage + (uint64_t)member * GZW_SLOT + 2;         // the member; its deflate data at +18 is 4-byte aligned
    uint32_t* const dwords = reinterpret_cast<uint32_t*>(mem + 18);
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;     // bits of the lanes before this one
    // ---- the block header: whole words straight to the member, the partial one opens the ring
    const uint32_t hdr_bits = cb.hdr_bits;
    uint32_t bitpos = hdr_bits;
    for (uint32_t i = (uint32_t)lane; i < (hdr_bits >> 5); i += WAVE) dwords[i] = cb.hdr[i];
    if (lane == 0 && (hdr_bits & 31u)) S.ring[0] = cb.hdr[hdr_bits >> 5] & ((1u << (hdr_bits & 31u)) - 1u);
    const uint32_t limit_bits = ((uint32_t)n + 5u) * 8u;                      // beyond this a stored block is smaller
    bool stored = false;
    // ---- masks of window f: what matches its predecessor byte, what matches the column four lines up, where lines end
    unsigned long long M1[5], Mc[5], NL[5];
    int line_far = 0;                                                        // line of window f's first byte
    bool use_lines = true;
    auto far_masks = [&](int f, unsigned long long& m1, unsigned long long& mc, unsigned long long& nlm) {
        const int p = 64 * f + lane;
        const bool valid = p < n;
        const uint32_t c = valid ? S.text[p] : 0u;
        const uint32_t prev = (valid && p > 0) ? S.text[p - 1] : 0x100u;
        const bool nl = valid && c == '\n';
        nlm = __ballot(nl);
        m1 = __ballot(valid && c == prev);
        const int line = line_far + __popcll(nlm & lt);
        // this window's line ends become line starts (a later lane of this very window may need the one an earlier lane writes)
        if (nl && line + 1 < GZW_MAX_LINES) S.ls[line + 1] = (uint16_t)(p + 1);
        bool eq = false;
        if (valid && line >= 4 && line < GZW_MAX_LINES) {
            const int l0 = (int)S.ls[line], l4 = (int)S.ls[line - 4], l3 = (int)S.ls[line - 3];
            const int q0 = l4 + (p - l0);
            eq = q0 < l3 && S.text[q0] == c;
        }
        mc = __ballot(eq);
        line_far += __popcll(nlm);
        if (line_far >= GZW_MAX_LINES - 1) use_lines = false;               // (too many short lines: no column matches from here on)
    };
    const int n_win = (n + 63) >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) far_masks(k, M1[k], Mc[k], NL[k]);
    int line_cur = 0, skip = 0;
    for (int w = 0; w < n_win && !stored; ++w) {
        const int p = 64 * w + lane;
        const bool valid = p < n;
        if (skip >= 64) skip -= 64;
        else {
            const uint32_t c = valid ? S.text[p] : 0u;
            // ---- the best match at every byte
            int blen = 1, dist = 0;
            {
                const uint32_t lb = S.lc[c] >> 16;
                const int r = min(gzw_ones_from(M1, lane), 258);
                int best_gain = 0;
                if (r >= 3) {
                    const int lsym = gz_len_sym(r);
                    const int gain = r * (int)lb - (int)((S.lc[257 + lsym] >> 16) + gz_len_extra(lsym) + (S.dc[0] >> 16));
                    if (gain > 0) { best_gain = gain; blen = r; dist = 1; }
                }
                if (use_lines && ((Mc[0] >> lane) & 1ull)) {
                    const int m = min(min(gzw_ones_from(Mc, lane), gzw_zeros_from(NL, lane) + 1), 258);
                    if (m >= 3) {
                        const int line = line_cur + __popcll(NL[0] & lt);
                        const int d = (int)S.ls[line] - (int)S.ls[line - 4];
                        const int lsym = gz_len_sym(m), dsym = gz_dist_sym(d);
                        // (the literals' bits taken as the first one's: exact for the long matches of a name against the name before
                        //  it would need the window's prefix sums and four more windows' — the decision is not close there)
                        const int gain = m * (int)max(lb, 5u) - (int)((S.lc[257 + lsym] >> 16) + gz_len_extra(lsym) + (S.dc[dsym] >> 16) + gz_dist_extra(dsym));
                        if (gain > best_gain) { best_gain = gain; blen = m; dist = d; }
                    }
                }
            }
            // ---- the greedy parse: token starts of this window
            unsigned long long marks;
            const unsigned long long vmask = __ballot(valid);
            int e = skip;
            if (__ballot(valid && blen > 1) == 0ull) {
                marks = vmask & ~((1ull << skip) - 1ull);                   // literals all the way
                e = 64;
            } else {
                marks = 0;
                const int last = 64 - (int)__builtin_clzll(vmask | 1ull);    // one behind the last valid lane
                while (e < last) {
                    marks |= 1ull << e;
                    e += __builtin_amdgcn_readlane(blen, e);
                }
                if (e < 64) e = 64;                                          // (the member ends inside this window)
            }
            skip = e - 64;
            // ---- the tokens' bits
            const bool tok = (marks >> lane) & 1ull;
            unsigned long long bits = 0;
            uint32_t nb = 0;
            if (tok) {
                if (blen == 1) {
                    const uint32_t a = S.lc[c];
                    bits = a & 0xffffu; nb = a >> 16;
                } else {
                    const int lsym = gz_len_sym(blen), dsym = gz_dist_sym(dist);
                    const uint32_t a = S.lc[257 + lsym], d = S.dc[dsym];
                    bits = a & 0xffffu; nb = a >> 16;
                    const int xl = gz_len_extra(lsym);
                    bits |= (unsigned long long)(uint32_t)(blen - gz_len_base(lsym)) << nb; nb += (uint32_t)xl;
                    bits |= (unsigned long long)(d & 0xffffu) << nb; nb += d >> 16;
                    const int xd = gz_dist_extra(dsym);
                    bits |= (unsigned long long)(uint32_t)(dist - gz_dist_base(dsym)) << nb; nb += (uint32_t)xd;
                }
            }
            uint32_t inc = nb;
#pragma unroll
            for (int dlt = 1; dlt < WAVE; dlt <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)inc, dlt);
                if (lane >= dlt) inc += o;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (tok) {
                const uint32_t at = bitpos + inc - nb;
                const uint32_t wi = (at >> 5) - (bitpos >> 5), sh = at & 31u;
                const unsigned long long lo = bits << sh;
                atomicOr(&S.ring[wi], (uint32_t)lo);
                if (sh + nb > 32u) atomicOr(&S.ring[wi + 1], (uint32_t)(lo >> 32));
                if (sh + nb > 64u) atomicOr(&S.ring[wi + 2], (uint32_t)(bits >> (64u - sh)));
            }
            // whole words leave the ring, the open one moves to its front
            const uint32_t w0 = bitpos >> 5, w1 = (bitpos + total) >> 5, nfull = w1 - w0;
            __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the ring's atomics are done
            uint32_t keep[2] = {0, 0};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t j = (uint32_t)lane + 64u * (uint32_t)r;
                keep[r] = S.ring[j];
                if (j < nfull) dwords[w0 + j] = keep[r];
            }
            const uint32_t open = nfull < (uint32_t)GZW_RING ? (uint32_t)__builtin_amdgcn_readlane((int)keep[nfull >> 6], (int)(nfull & 63u)) : 0u;
#pragma unroll
            for (int r = 0; r < 2; ++r) S.ring[(uint32_t)lane + 64u * (uint32_t)r] = 0;
            if (lane == 0) S.ring[0] = open;
            bitpos += total;
            if (bitpos > limit_bits) stored = true;
        }
        // ---- the masks move on by a window
        line_cur += __popcll(NL[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { M1[k] = M1[k + 1]; Mc[k] = Mc[k + 1]; NL[k] = NL[k + 1]; }
        far_masks(w + 5, M1[4], Mc[4], NL[4]);
    }
    uint32_t dbytes;
    if (!stored) {
        // end of block, then the open word
        const uint32_t eob = S.lc[256];
        if (lane == 0) {
            const uint32_t sh = bitpos & 31u;
            const unsigned long long v = (unsigned long long)S.ring[0] | ((unsigned long long)(eob & 0xffffu) << sh);
            dwords[bitpos >> 5] = (uint32_t)v;
            if (sh + (eob >> 16) > 32u) dwords[(bitpos >> 5) + 1] = (uint32_t)(v >> 32);
        }
        bitpos += eob >> 16;
        dbytes = (bitpos + 7u) >> 3;
        stored = dbytes >= (uint32_t)n + 5u;
    }
    if (stored) {
        // incompressible with this code: one stored block (BFINAL = 1, BTYPE = 0, LEN, ~LEN, bytes)
        dbytes = (uint32_t)n + 5u;
        if (lane == 0) {
            mem[18] = 1;
            mem[19] = (uint8_t)n; mem[20] = (uint8_t)(n >> 8); mem[21] = (uint8_t)~n; mem[22] = (uint8_t)(~n >> 8);
        }
        for (int i = lane; i < n; i += WAVE) mem[23 + i] = S.text[i];
    }
    // ---- CRC-32 of the member's text: raw CRCs of the 64 grid segments (the first four text bytes complemented), combined
    uint32_t crc;
    {
        const int pad = GZW_TEXT - n;
        const int a = max(0, lane * GZ_SEG - pad), b = max(0, (lane + 1) * GZ_SEG - pad);
        uint32_t c = 0;
        for (int p = a; p < b; ++p) {
            uint32_t x = S.text[p];
            if (p < 4) x ^= 0xffu;                          // = starting the register at 0xffffffff, in a form leading zeros do not disturb
            c = S.crc_tab[(c ^ x) & 0xffu] ^ (c >> 8);
        }
#pragma unroll 1
        for (int k = 0; k < 6; ++k) {
            const int step = 1 << k;
            const uint32_t right = (uint32_t)__shfl_down((int)c, step);
            if ((lane & (2 * step - 1)) == 0) {
                const uint32_t* M = J.crc->shift[k];
                uint32_t r = right;
                for (int j = 0; j < 32; ++j) r ^= ((c >> j) & 1u) ? M[j] : 0u;
                c = r;
            }
        }
        crc = ~c;
    }
    if (lane == 0) {
        if (n < 4) {
            uint32_t c = 0xffffffffu;
            for (int p = 0; p < n; ++p) c = S.crc_tab[(c ^ S.text[p]) & 0xffu] ^ (c >> 8);
            crc = ~c;
        }
        const uint32_t bsize = 18u + dbytes + 8u;
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        for (int i = 0; i < 16; ++i) mem[i] = hdr[i];
        mem[16] = (uint8_t)((bsize - 1) & 0xffu);
        mem[17] = (uint8_t)((bsize - 1) >> 8);
        uint8_t* t = mem + 18 + dbytes;
        for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)((uint32_t)n >> (8 * k)); }
        J.sizes[member] = bsize;
    }
}

}  // namespace aqc
