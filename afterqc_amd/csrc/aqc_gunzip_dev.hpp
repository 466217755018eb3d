// aqc_gunzip_dev.hpp — gzip INPUT decoded on the device (round 4): the unit of work is the DEFLATE BLOCK, read by 32 lanes
// (fastq.py:23-24 upstream: gzip.open + readline on the one CPU thread).
//
// Why the block.  Huffman decoding is one dependency chain per stream — bit position -> table entry -> next bit
// position — and nothing inside a block breaks it.  Round 3's decoder gave a whole WAVE to one stream (64 lanes guessing token
// starts, a scalar walk picking the real ones): 5 tokens per ~4,800-cycle round, 4 MB/s per wave, 2.75 GB/s per GPU.  But one
// gzip stream is tens of thousands of blocks (zlib closes a block every 16 K tokens: ~110 KB of FASTQ text at level 6), every
// dynamic-Huffman block carries its own code, and where a block begins can be RECOGNISED without decoding anything before it
// (the header must describe three complete prefix codes: about one false hit per gigabyte).  So:
//
//   gzb_scan_kernel     EVERY bit position of the batch is tested for "a non-final dynamic-Huffman block begins here": the
//                       three header bits and HLIT / HDIST <= 29 for 32 positions at a time with bitwise logic on 64-bit words
//                       (~1 instruction per position), the Kraft sum of the code-length code for the ~11 % that remain (seven
//                       LDS look-ups of three 3-bit fields each), and for the ~0.2 % that remain the code lengths themselves: they
//                       must parse and form a complete literal/length code with an end-of-block symbol and a usable distance code.
//   gzb_compact_kernel  the per-tile hits become one sorted candidate list; a candidate's output space is sized from the
//                       compressed bytes up to the next candidate.
//   gzb_tables_kernel   a lane per candidate builds the block's tables (11-bit literal/length and 10-bit distance roots, 16-bit
//                       entries; longer codes: canonical search) and plans where its GZB_K = 32 lanes start and stop.
//   gzb_decode_kernel   a lane per (candidate, entry point) turns its stretch of the block into TOKENS (literal | length,
//                       distance; each with the bit it starts at) the way a CPU thread reads a Huffman stream — from a GUESSED
//                       bit inside the block: what follows a bit position depends on nothing but the tables, and Huffman
//                       streams re-synchronise within tens of tokens, so each lane reads on into its successor's share.
//                       Tables and a window of the stream in LDS, in slices of 2048 tokens per launch.
//   gzb_expand_kernel   a WAVE per block stitches the lanes' lists together — lane k + 1's list takes over at the first bit
//                       position both lists hold a token at: found by search, never assumed, so the block is exact or counts
//                       as failed — and applies the tokens, 64 at a time: literals stored at once, the matches that copy from
//                       before the chunk all together, the others one by one with the 64 lanes sharing the copy; into 16-bit
//                       SYMBOLS (>= 0x8000: "byte j of the 32 KiB before this BLOCK").
//   gzb_chain_kernel    a lane per SECTION (the host's unit of work, aqc_gunzip.cpp): from the first candidate at or behind the
//                       section's nominal start it follows  end of block == start of a candidate  until the section's stop bit;
//                       stored blocks (pigz's sync markers) are stepped over in place.  A false candidate never chains up.
//   gzb_gather_kernel   a workgroup per section copies the chained blocks into one symbol stream and re-bases their markers
//                       from "before my block" to "before my section" (or resolves them: the byte is in an earlier block).
//
// What comes back is exactly what a host pool thread would have produced for the section — start bit, end bit, symbols — so
// the consumer's commit rule (a section counts only if it starts at the very bit its predecessor ended on) is unchanged and
// the result stays exact.  Final blocks, fixed-Huffman blocks and anything the scan does not recognise end a chain; the host
// decodes on from there (ParallelGunzip::bridge).
//
// Everything a LANE does is a plain __host__ __device__ function (GZB_HD): tests/native/gzb_selftest.cpp compiles this header
// with the host compiler and runs the same scan tests, table builder, block decoder, chain walk and marker re-basing over
// zlib streams on the CPU (`-m "not gpu"`); the kernels below only deal the work to lanes.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GZB_HD __host__ __device__
#else
#define GZB_HD
#endif

namespace aqc {

GZB_HD inline uint32_t gzb_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
GZB_HD inline uint32_t gzb_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
// the low `len` bits of c, reversed
GZB_HD inline uint32_t gzb_rev(uint32_t c, uint32_t len) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(c) >> (32u - len);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; ++i) r |= ((c >> i) & 1u) << (len - 1u - i);
    return r;
#endif
}

constexpr uint32_t GZB_MARKER = 0x8000u;
constexpr int GZB_LROOT = 11, GZB_DROOT = 10;
// A block's decoding tables, 16-bit entries: literal/length root (2^11), distance root (2^10), symbols sorted by code (288 + 32),
// codes per length (16 + 16), in the candidate's own 7 KiB of global memory (with its 320 code lengths behind them); the token
// kernel copies them into LDS (two blocks' tables per workgroup of 64 lanes: 14 KiB).  Roots this wide make a code longer than
// the root — the canonical search, a dozen dependent look-ups — a rarity: with 9 bits SOME lane of a wave needed it in most
// iterations, and the whole wave waits for it.
constexpr int GZB_E_DIST = 1 << GZB_LROOT, GZB_E_LSORT = GZB_E_DIST + (1 << GZB_DROOT), GZB_E_DSORT = GZB_E_LSORT + 288, GZB_E_LCOUNT = GZB_E_DSORT + 32,
              GZB_E_DCOUNT = GZB_E_LCOUNT + 16, GZB_TAB_ENTRIES = GZB_E_DCOUNT + 16;
constexpr int GZB_TAB_WORDS = GZB_TAB_ENTRIES / 2 + 80;          // 32-bit words per candidate: the tables, then the code lengths
constexpr int GZB_SCAN_THREADS = 256, GZB_SCAN_TILE = GZB_SCAN_THREADS * 16, GZB_TILE_CAND = 16;
constexpr int GZB_DEC_THREADS = 64;
constexpr int GZB_K = 32;                          // lanes per block (entry points guessed inside it, see gzb_decode_kernel)
constexpr uint32_t GZB_OVERLAP_BITS = 4096;        // how far a lane reads into its successor's share to meet its token list
constexpr uint32_t GZB_PLAIN_BITS = 8192;           // a block shorter than this is read by one lane
constexpr uint32_t GZB_SYM_EXTRA = 4104u;           // symbol space per candidate beyond ratio_cap x its bytes
constexpr uint32_t GZB_OVERLAP_TOKENS = 512u;       // token entries a lane has for what it reads of its successor's share (8 bits a token: FASTQ's tokens take ~16 — its bases go out as
                                                    // matches of 7 - 8, measured 0.5 - 0.6 tokens per compressed byte at every level; less: overflow, the group is decoded again with more)
constexpr uint32_t GZB_T_EOB = 0x40000000u, GZB_T_JUNK = 0x20000000u;
constexpr int GZB_SEC_BLOCKS = 4096;                // chain entries per section (3 words each)
constexpr int GZB_GATHER_THREADS = 1024;
constexpr uint32_t GZB_F_ERROR = 1u, GZB_F_OVERFLOW = 2u, GZB_F_SKIP = 4u, GZB_F_MORE = 8u;     // MORE: the slice ended inside the block
constexpr uint32_t GZB_NONE = 0xffffffffu;
constexpr uint32_t GZB_STORED = 0x80000000u;

// One batch: a window of the compressed file on the device.  Bit positions are 32-bit, relative to comp[0]: a window is < 512 MiB.
struct GzbJob {
    const uint8_t* comp;         // 16-byte aligned, zero-padded for 64 bytes behind comp_bytes
    uint32_t comp_bytes;
    uint32_t scan_byte0;         // first byte whose bit positions are scanned (multiple of 16)
    uint32_t first_bit, last_bit;// candidates are kept in [first_bit, last_bit)
    uint32_t n_tiles;
    uint32_t* tile_cnt;          // [n_tiles]
    uint32_t* tile_cand;         // [n_tiles][GZB_TILE_CAND]
    uint32_t cand_cap;
    uint32_t* n_cand;            // [0] candidates, [1] 1 when the symbol space ran out
    uint32_t* c_start;           // [cand_cap] header bit
    uint32_t* c_end;             // [cand_cap] bit behind the end-of-block code
    uint32_t* c_nsym;
    uint32_t* c_flags;
    uint64_t* c_symoff;          // [cand_cap] first symbol of the candidate in blk_sym
    uint32_t* c_symcap;
    uint16_t* blk_sym;
    uint64_t blk_sym_cap;        // symbols
    unsigned long long* blk_tp;  // tokens of candidate c at [c_tokoff[c], + c_tokcap[c]), GZB_K equal shares for its GZB_K lanes: the token in the
                                 // low word, the bit it starts at in the high one (one store per token)
    uint64_t* c_tokoff;          // [cand_cap]
    uint32_t* c_tokcap;          // [cand_cap] a multiple of GZB_K
    uint64_t blk_tp_cap;         // entries
    uint32_t overlap_tokens;     // token entries a lane has for what it reads of its successor's share (GZB_OVERLAP_TOKENS; more for streams of literals)
    uint32_t tok_ratio;          // token entries per compressed byte (round 6: no longer tied to the symbol space — tokens are bounded by
                                 // the compressed bits, symbols by the text: sized together they cost 12 x 6 bytes per compressed byte)
    uint32_t* c_lanes;           // [cand_cap] lanes that read this block: GZB_K from guessed entry points, or 1 (plain)
    uint32_t* l_p;               // [cand_cap * GZB_K] per lane: bit position reached ...
    uint32_t* l_stop;            //   ... where it stops (a lane reads on behind its share until it has met its successor's list)
    uint32_t* l_start;           //   ... where it started
    uint32_t* l_ntok;            //   ... tokens written
    uint32_t* l_flags;           //   ... GZB_F_MORE while it has work left, 0 done, GZB_F_OVERFLOW
    uint32_t ratio_cap;          // a block may expand to ratio_cap x its compressed size (+ 4096 symbols)
    uint32_t* tables;            // [cand_cap][GZB_TAB_WORDS]: a candidate's tables and code lengths
    uint32_t slice_tokens;       // tokens a lane decodes per launch (the decoder runs in slices: see gzb_decode_kernel)
    // sections
    uint32_t n_sec;
    const uint32_t* s_nominal;   // [n_sec] search from this bit
    const uint32_t* s_stop;      // [n_sec] stop at the first block boundary at or behind this bit
    const uint32_t* s_exact;     // [n_sec] 1: the section must start AT s_nominal (a known boundary)
    uint32_t* s_start;           // [n_sec] GZB_NONE: nothing found
    uint32_t* s_end;
    uint32_t* s_nsym;
    uint32_t* s_nblk;
    uint32_t* s_blocks;          // [n_sec][GZB_SEC_BLOCKS][3]: candidate (or GZB_STORED | length), source byte (stored), offset
    uint64_t* s_off;             // [n_sec + 1] first symbol of each section in s_sym (gzb_place), [n_sec] = symbols in all
    uint16_t* s_sym;             // the sections' symbols, packed (64-byte aligned starts): ONE copy takes them to the host
    uint64_t s_sym_total;        // symbols s_sym holds
    uint32_t s_symcap;           // most symbols one section may have
};

// base value and number of extra bits of length symbol 257 + i and of distance symbol i (RFC 1951 3.2.5), COMPUTED: a table in
// memory is a trip to the cache per token for every lane (two trips in a row for a match: that alone was 0.8 us per token)
GZB_HD inline uint32_t gzb_len_extra(uint32_t i) { return (i < 8u || i == 28u) ? 0u : (i - 4u) >> 2; }
GZB_HD inline uint32_t gzb_len_base(uint32_t i) { return i == 28u ? 258u : i < 8u ? 3u + i : 3u + ((4u + (i & 3u)) << ((i - 4u) >> 2)); }
GZB_HD inline uint32_t gzb_dist_extra(uint32_t i) { return i < 4u ? 0u : (i - 2u) >> 1; }
GZB_HD inline uint32_t gzb_dist_base(uint32_t i) { return i < 4u ? 1u + i : 1u + ((2u + (i & 1u)) << ((i - 2u) >> 1)); }
// the order the code-length code's lengths come in (16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15), five bits each in two constants
GZB_HD inline uint32_t gzb_cl_order(uint32_t i) {
    return i < 12u ? (uint32_t)(0x22caa324e804a30ull >> (5u * i)) & 31u : (uint32_t)(0x3c2e1346cull >> (5u * (i - 12u))) & 31u;
}

// >= 57 bits of the stream from bit position p on (any alignment; the buffer is padded)
GZB_HD inline unsigned long long gzb_peek(const uint8_t* comp, uint32_t p) {
    unsigned long long v;
    memcpy(&v, comp + (p >> 3), 8);
    return v >> (p & 7u);
}

// Table entries (16 bits).  Literal/length: bits 3:0 code length (0: not a root code), 4 literal, 5 end of block, 6 invalid symbol,
// 15:8 the literal byte or the length symbol - 257.  Distance: bits 3:0 code length, 8:4 distance symbol, 15 invalid.
GZB_HD inline uint32_t gzb_lit_entry(uint32_t s, uint32_t l) {
    if (s < 256u) return l | 0x10u | (s << 8);
    if (s == 256u) return l | 0x20u;
    if (s > 285u) return l | 0x40u;
    return l | ((s - 257u) << 8);
}
GZB_HD inline uint32_t gzb_dist_entry(uint32_t s, uint32_t l) {
    if (s > 29u) return l | 0x8000u;
    return l | (s << 4);
}

// The header of a dynamic-Huffman block whose three header bits sit at bit p: HLIT, HDIST, HCLEN, the code-length code, the
// HLIT + HDIST code lengths.  Returns false unless the code-length code is complete, the lengths parse, the literal/length
// code is complete and has an end-of-block symbol, and the distance code is complete, a single 1-bit code, or empty (what
// zlib's inflate accepts, minus the incomplete single-code literal case no compressor writes).  cl: this lane's 128-entry
// table of the code-length code, entry i at cl[i * stride]; lens (may be null): the lengths as bytes, 320 of them.
GZB_HD inline bool gzb_header(const uint8_t* comp, uint32_t limit_bit, uint32_t p, uint8_t* cl, int stride, uint8_t* lens,
                                  uint32_t& data_bit, uint32_t& hlit_out, uint32_t& hdist_out) {
    if (p + 17u + 64u > limit_bit) return false;
    const unsigned long long w = gzb_peek(comp, p);
    const uint32_t hlit = (uint32_t)((w >> 3) & 31u) + 257u, hdist = (uint32_t)((w >> 8) & 31u) + 1u, hclen = (uint32_t)((w >> 13) & 15u) + 4u;
    if (hlit > 286u || hdist > 30u) return false;
    uint32_t q = p + 17u;
    unsigned long long v = gzb_peek(comp, q);
    q += 3u * hclen;
    // the 19 code-length-code lengths, 3 bits each, by symbol; codes per length, 8 bits each
    unsigned long long cl3 = 0, cnt = 0;
    for (uint32_t i = 0; i < hclen; ++i) {
        const unsigned long long l = v & 7u;
        v >>= 3;
        cl3 |= l << (3u * gzb_cl_order(i));
        cnt += 1ull << (8u * (uint32_t)l);
    }
    unsigned long long next = 0;
    {
        uint32_t code = 0, left = 1, prevc = 0;
        for (uint32_t l = 1; l <= 7; ++l) {
            const uint32_t c = (uint32_t)(cnt >> (8u * l)) & 255u;
            code = (code + prevc) << 1;
            prevc = c;
            next |= (unsigned long long)(code & 255u) << (8u * l);
            left <<= 1;
            if (c > left) return false;
            left -= c;
        }
        if (left != 0) return false;
    }
    for (uint32_t s = 0; s < 19; ++s) {
        const uint32_t l = (uint32_t)(cl3 >> (3u * s)) & 7u;
        if (!l) continue;
        const uint32_t c = (uint32_t)(next >> (8u * l)) & 255u;
        next += 1ull << (8u * l);
        const uint32_t r = gzb_rev(c, l);
        const uint8_t e = (uint8_t)((s << 3) | l);
        for (uint32_t i = r; i < 128u; i += 1u << l) cl[i * stride] = e;
    }
    const uint32_t total = hlit + hdist;
    uint32_t i = 0, prev = 0, klit = 0, kdist = 0, maxd = 0;
    bool has_eob = false;
    while (i < total) {
        if (q + 64u > limit_bit) return false;
        const unsigned long long x = gzb_peek(comp, q);
        const uint32_t e = cl[((uint32_t)x & 127u) * stride];
        const uint32_t l = e & 7u, sym = e >> 3;
        q += l;
        uint32_t rep = 1, val = sym;
        if (sym >= 16u) {
            const uint32_t y = (uint32_t)(x >> l);
            if (sym == 16u) { if (i == 0) return false; val = prev; rep = 3u + (y & 3u); q += 2; }
            else if (sym == 17u) { val = 0; rep = 3u + (y & 7u); q += 3; }
            else { val = 0; rep = 11u + (y & 127u); q += 7; }
            if (i + rep > total) return false;
        }
        prev = val;
        if (val) {
            const uint32_t k = 32768u >> val;
            const uint32_t nl = i < hlit ? gzb_min(rep, hlit - i) : 0u;
            klit += nl * k;
            kdist += (rep - nl) * k;
            if (rep > nl) maxd = gzb_max(maxd, val);
            if (i <= 256u && i + rep > 256u) has_eob = true;
            // (an over-subscribed code is no code: what is not a header gets here within a few dozen lengths — the scan's
            //  lanes that parse run in step with the slowest of them, which used to be the full 300 lengths)
            if (klit > 32768u || kdist > 32768u) return false;
        }
        if (lens)
            for (uint32_t k = 0; k < rep; ++k) lens[i + k] = (uint8_t)val;
        i += rep;
    }
    if (!has_eob || klit != 32768u) return false;
    if (!(kdist == 32768u || (kdist <= 16384u && maxd <= 1u))) return false;
    data_bit = q; hlit_out = hlit; hdist_out = hdist;
    return true;
}

// ---- what a lane does, as plain functions ------------------------------------------------------------------------------------------
// 32 positions at a time (v = 64 bits of the stream from the first position on): BFINAL = 0 and BTYPE = 2 (bits 0 0 1),
// HLIT <= 29 and HDIST <= 29 (not: the upper four bits of the five all set)
GZB_HD inline uint32_t gzb_quick32(unsigned long long v) {
    const unsigned long long z = ~v & ~(v >> 1) & (v >> 2);
    const unsigned long long hl = (v >> 4) & (v >> 5) & (v >> 6) & (v >> 7);
    const unsigned long long hd = (v >> 9) & (v >> 10) & (v >> 11) & (v >> 12);
    return (uint32_t)(z & ~hl & ~hd);
}
// kraft9[i] = sum over the three 3-bit lengths packed in i of 128 >> l (0 for l == 0)
GZB_HD inline uint32_t gzb_kraft9(uint32_t i) {
    uint32_t k = 0;
    for (int f = 0; f < 3; ++f) { const uint32_t l = (i >> (3 * f)) & 7u; k += l ? 128u >> l : 0u; }
    return k;
}
// the code-length code of the header at bit p must be complete: Kraft sum of its `hclen` 3-bit lengths (v: the 64 bits of the
// stream from bit p + 17 on)
GZB_HD inline bool gzb_kraft_ok_v(unsigned long long v, uint32_t hclen, const uint8_t* kraft9) {
    v &= (1ull << (3u * hclen)) - 1ull;
    const uint32_t kraft = (uint32_t)kraft9[v & 511u] + kraft9[(v >> 9) & 511u] + kraft9[(v >> 18) & 511u] + kraft9[(v >> 27) & 511u] +
                           kraft9[(v >> 36) & 511u] + kraft9[(v >> 45) & 511u] + kraft9[(v >> 54) & 511u];
    return kraft == 128u;
}
GZB_HD inline bool gzb_kraft_ok(const uint8_t* comp, uint32_t p, uint32_t hclen, const uint8_t* kraft9) {
    return gzb_kraft_ok_v(gzb_peek(comp, p + 17u), hclen, kraft9);
}

// One lane's tables: entry e at base[e * S] (S = 64 on the device: see GZB_E_*; 1 on the host).
template <int S>
struct GzbLaneTab {
    uint16_t* base;
    GZB_HD uint16_t& at(int e) const { return base[e * S]; }
};

// canonical Huffman code from lens[0, n): root table + symbols sorted by code + codes per length.  cnt / nxt / off: three arrays
// of 16 counters, element l at [l * CS] (the lanes' columns of LDS arrays).
template <bool LIT, int S>
GZB_HD inline void gzb_build(const uint8_t* lens, uint32_t n, const GzbLaneTab<S>& T, uint32_t* cnt, uint32_t* nxt, uint32_t* off, int CS) {
    const uint32_t R = LIT ? GZB_LROOT : GZB_DROOT;
    const int root = LIT ? 0 : GZB_E_DIST, sorted = LIT ? GZB_E_LSORT : GZB_E_DSORT, count = LIT ? GZB_E_LCOUNT : GZB_E_DCOUNT;
    for (int l = 0; l < 16; ++l) cnt[l * CS] = 0;
    for (uint32_t s = 0; s < n; ++s) cnt[(uint32_t)lens[s] * CS] += 1u;
    {
        uint32_t code = 0, prev = 0, o = 0;
        T.at(count) = 0;
        for (int l = 1; l <= 15; ++l) {
            code = (code + prev) << 1;
            prev = cnt[l * CS];
            nxt[l * CS] = code;
            off[l * CS] = o;
            T.at(count + l) = (uint16_t)prev;
            o += prev;
        }
    }
    for (uint32_t i = 0; i < (1u << R); ++i) T.at(root + (int)i) = 0;
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens[s];
        if (!l) continue;
        const uint32_t c = nxt[l * CS];
        nxt[l * CS] = c + 1;
        const uint32_t o = off[l * CS];
        off[l * CS] = o + 1;
        T.at(sorted + (int)o) = (uint16_t)s;
        if (l <= R) {
            const uint16_t e = (uint16_t)(LIT ? gzb_lit_entry(s, l) : gzb_dist_entry(s, l));
            for (uint32_t i = gzb_rev(c, l); i < (1u << R); i += 1u << l) T.at(root + (int)i) = e;
        }
    }
}

// a code longer than the root index: canonical search, one bit at a time (w: the stream from the code's first bit on)
template <bool LIT, int S>
GZB_HD inline uint32_t gzb_slow(unsigned long long w, const GzbLaneTab<S>& T) {
    const int sorted = LIT ? GZB_E_LSORT : GZB_E_DSORT, count = LIT ? GZB_E_LCOUNT : GZB_E_DCOUNT;
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t len = 1; len <= 15; ++len) {
        code |= (uint32_t)(w & 1u);
        w >>= 1;
        const uint32_t c = T.at(count + (int)len);
        if (code - first < c) {
            const uint32_t s = T.at(sorted + (int)(index + (code - first)));
            return LIT ? gzb_lit_entry(s, len) : gzb_dist_entry(s, len);
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return 0;
}

GZB_HD inline uint32_t gzb_lower_bound(const uint32_t* a, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// the same over the bit positions of a token list (high words)
GZB_HD inline uint32_t gzb_lower_bound_pos(const unsigned long long* a, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)(a[mid] >> 32) < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

GZB_HD inline unsigned long long gzb_load64(const uint8_t* p) {
    unsigned long long v;
    memcpy(&v, p, 8);
    return v;
}

// PHASE 1 — TOKENS of a stretch of a block, from bit p on: a literal is 0x80000000 | byte, a match length << 16 | distance - 1,
// the end-of-block code GZB_T_EOB; the bit every token starts at goes into the entry's high word.  Nothing is copied here: what a lane waits for per
// token is one table look-up (two for a match) and, every few tokens, the next word of the stream, which was asked for when the
// bit buffer was last refilled.  (The first version copied the matches as it went: 41 scattered memory instructions per wave
// step, 4 microseconds per token.)
//
// A block is read by SEVERAL lanes: lane k starts at a guessed bit inside the block.  Until its reading frame happens to fall
// on a real token boundary it produces nonsense (including "end of block" and codes no symbol has: GZB_T_JUNK, one bit
// consumed); from then on — Huffman streams re-synchronise within tens of tokens — it produces the true tokens, because what
// follows a bit position depends on nothing but the tables.  So a lane never stops at an end-of-block code (spec) and reads on
// to stop_bit, some way into its successor's share; gzb_stitch picks, per lane, the tokens from the first bit position it SHARES
// with its predecessor's list — a check, not a guess.  A plain lane (spec == false) stops behind the end-of-block code.
// Returns 0 done, GZB_F_MORE (max_tokens written: call again with the same p / nt), GZB_F_OVERFLOW, GZB_F_ERROR (plain only).
//
// The stream is read 32 bits at a time through `in`: the host reads memory; a device lane reads an LDS window of 128 bytes of
// its stretch (GzbInLds) that is topped up at in.sync(), which every lane of the wave reaches at the same step.  That is the
// point: the hardware counts a WAVE's outstanding loads, not a lane's.  Whenever lanes ask for the next piece of their streams
// at steps of their own — eight bytes per refill in the first version, 64 bytes whenever a lane left a half of its window in
// the second — nearly every step has some lane asking and some lane needing what it asked for long ago, and the wave waits
// for the newest request to come back from memory: 0.8 us per token, whatever the tables and the stores were doing.
struct GzbInMem {
    const uint8_t* comp;
    uint32_t wi;                 // the next word to hand out
    GZB_HD void start(uint32_t word) { wi = word; }
    GZB_HD void sync() {}
    GZB_HD uint32_t next() {
        uint32_t v;
        memcpy(&v, comp + 4ull * wi, 4);
        ++wi;
        return v;
    }
};
constexpr uint32_t GZB_SYNC_STEPS = 8;             // the tokenizer calls in.sync() every so many tokens (a token takes <= 48 bits)

template <int S, class In>
GZB_HD inline uint32_t gzb_tokenize(In& in, uint32_t limit_bit, const GzbLaneTab<S>& T, unsigned long long* tp, uint32_t tok_cap, uint32_t& p, uint32_t& nt,
                                    uint32_t stop_bit, uint32_t max_tokens, bool spec) {
    uint32_t fl = 0, tokens = 0;
    in.start(p >> 5);
    unsigned long long bb = in.next();
    bb |= (unsigned long long)in.next() << 32;
    bb >>= p & 31u;
    uint32_t bn = 64u - (p & 31u);                            // valid bits in bb; the next unread bit of the stream is in.wi * 32
#define GZB_EMIT(at_, t_) (tp[nt++] = ((unsigned long long)(at_) << 32) | (unsigned long long)(t_))
#define GZB_REFILL()                                                 \
    do {                                                             \
        if (bn <= 32u) {                                             \
            bb |= (unsigned long long)in.next() << bn;               \
            bn += 32u;                                               \
        }                                                            \
    } while (0)
    for (;;) {
        const uint32_t at = in.wi * 32u - bn;                 // the bit this token starts at
        if (at + 64u > limit_bit) { fl = spec ? 0u : GZB_F_ERROR; break; }      // (the buffer is padded: reading on is harmless, but nothing ends there)
        if (at >= stop_bit) break;
        if (nt >= tok_cap) { fl = GZB_F_OVERFLOW; break; }
        if (tokens++ >= max_tokens) { fl = GZB_F_MORE; break; }
        if ((tokens & (GZB_SYNC_STEPS - 1u)) == 0u) in.sync();
        GZB_REFILL();                                         // > 32 bits now; a literal/length code + its extra bits take <= 20
        uint32_t e = T.at((int)((uint32_t)bb & ((1u << GZB_LROOT) - 1u)));
        if ((e & 15u) == 0u) e = gzb_slow<true, S>(bb, T);
        if (!e || (e & 0x40u)) {                              // no such code / a symbol that does not exist
            if (!spec) { fl = GZB_F_ERROR; break; }
            bb >>= 1; bn -= 1;
            GZB_EMIT(at, GZB_T_JUNK);
            continue;
        }
        const uint32_t l = e & 15u;
        bb >>= l;
        bn -= l;
        if (e & 0x10u) { GZB_EMIT(at, 0x80000000u | (e >> 8)); continue; }
        if (e & 0x20u) {                                      // end of block
            GZB_EMIT(at, GZB_T_EOB);
            if (spec) continue;
            break;
        }
        const uint32_t ls = e >> 8;
        const uint32_t xb = gzb_len_extra(ls);
        const uint32_t len = gzb_len_base(ls) + ((uint32_t)bb & ((1u << xb) - 1u));
        bb >>= xb;
        bn -= xb;
        GZB_REFILL();                                         // a distance code + its extra bits take <= 28
        uint32_t de = T.at(GZB_E_DIST + (int)((uint32_t)bb & ((1u << GZB_DROOT) - 1u)));
        if ((de & 15u) == 0u) de = gzb_slow<false, S>(bb, T);
        if (!de || (de & 0x8000u)) {
            if (!spec) { fl = GZB_F_ERROR; break; }
            GZB_EMIT(at, GZB_T_JUNK);            // (the length code's bits are gone: any rule will do before the frames meet)
            continue;
        }
        const uint32_t dl = de & 15u, ds = (de >> 4) & 31u;
        bb >>= dl;
        const uint32_t dxb = gzb_dist_extra(ds);
        const uint32_t dd = gzb_dist_base(ds) + ((uint32_t)bb & ((1u << dxb) - 1u));
        bb >>= dxb;
        bn -= dl + dxb;
        GZB_EMIT(at, (len << 16) | (dd - 1u));
    }
#undef GZB_REFILL
#undef GZB_EMIT
    p = in.wi * 32u - bn;                                     // the first bit not consumed
    return fl;
}

// Where the lanes of candidate c (of n) start and stop.  data_bit: the block's first data bit (behind its header); the block is
// taken to end where the next candidate starts (the window's end for the last one).  If it ends earlier — stored and fixed-code
// blocks are no candidates — the lanes behind its end read nonsense nobody looks at; if it ends LATER (the next candidate was
// a false hit inside it) its last list runs out and the block counts as failed: the host inflates that section.  A small block
// is read by one plain lane.
GZB_HD inline void gzb_plan_lanes(const GzbJob& J, uint32_t c, uint32_t n, uint32_t data_bit) {
    const uint32_t limit = J.comp_bytes * 8u;
    // (the window's last candidate: blocks of one stream are of similar size — half as much again as its predecessor, not the
    // megabytes of slack up to the window's end, which sixteen lanes would read to the last bit)
    uint32_t est_end = limit;
    if (c + 1 < n) est_end = J.c_start[c + 1];
    else if (c > 0) {
        const uint32_t gap = J.c_start[c] - J.c_start[c - 1];
        est_end = gzb_min(limit, J.c_start[c] + gap + gap / 2u + 4096u);
    }
    const bool plain = est_end <= data_bit || est_end - data_bit < GZB_PLAIN_BITS;
    J.c_lanes[c] = plain ? 1u : (uint32_t)GZB_K;
    for (uint32_t k = 0; k < (uint32_t)GZB_K; ++k) {
        const uint32_t i = c * (uint32_t)GZB_K + k;
        uint32_t st = data_bit, sp = 0xffffffffu;
        if (!plain) {
            const uint32_t share = (est_end - data_bit) / (uint32_t)GZB_K;
            st = data_bit + k * share;
            sp = k + 1 < (uint32_t)GZB_K ? data_bit + (k + 1) * share + GZB_OVERLAP_BITS : gzb_min(est_end + 64u, limit);
        }
        J.l_start[i] = st; J.l_p[i] = st; J.l_stop[i] = sp; J.l_ntok[i] = 0;
        J.l_flags[i] = (plain && k > 0) ? 0u : GZB_F_MORE;
    }
}

// PHASE 2 as the host tests run it (the kernel does the same a wave per block, the lanes sharing every copy and every search):
// the block's symbols from its lanes' token lists.  Lane k's list counts from the first bit position it shares with lane
// k - 1's; the block ends at the first end-of-block token met on the way.  A symbol >= 0x8000 is "byte j of the 32 KiB before
// this block".  Returns the block's flags; n_sym / end_bit as gzb_chain_section wants them.
GZB_HD inline uint32_t gzb_stitch_expand(const GzbJob& J, uint32_t c, uint32_t& n_sym, uint32_t& end_bit) {
    const uint32_t lanes = J.c_lanes[c], cap = J.c_symcap[c];
    const uint32_t share = J.c_tokcap[c] / (uint32_t)GZB_K;
    const unsigned long long* const tp0 = J.blk_tp + J.c_tokoff[c];
    uint16_t* const out = J.blk_sym + J.c_symoff[c];
    for (uint32_t k = 0; k < lanes; ++k)
        if (J.l_flags[c * GZB_K + k] != 0u) return J.l_flags[c * GZB_K + k] == GZB_F_MORE ? GZB_F_OVERFLOW : J.l_flags[c * GZB_K + k];
    uint32_t k = 0, i = 0, op = 0;
    for (;;) {
        const unsigned long long* const tp = tp0 + (size_t)k * share;
        const uint32_t n = J.l_ntok[c * GZB_K + k];
        if (i >= n) return GZB_F_ERROR;                       // the list ran out before it met the next one / an end of block
        const uint32_t t = (uint32_t)tp[i], at = (uint32_t)(tp[i] >> 32);
        if (k + 1 < lanes && at >= J.l_start[c * GZB_K + k + 1]) {
            const unsigned long long* const ntp = tp0 + (size_t)(k + 1) * share;
            const uint32_t nn = J.l_ntok[c * GZB_K + k + 1];
            const uint32_t j = gzb_lower_bound_pos(ntp, nn, at);
            if (j < nn && (uint32_t)(ntp[j] >> 32) == at) { ++k; i = j; continue; }        // the two reading frames have met: go on in the next list
        }
        if (t == GZB_T_EOB) { end_bit = i + 1 < n ? (uint32_t)(tp[i + 1] >> 32) : J.l_p[c * GZB_K + k]; break; }
        if (t & GZB_T_JUNK) return GZB_F_ERROR;
        if (t >> 31) {
            if (op + 1u > cap) return GZB_F_OVERFLOW;
            out[op++] = (uint16_t)(t & 0xffu);
        } else {
            const uint32_t len = (t >> 16) & 0x1ffu, dd = (t & 0x7fffu) + 1u;
            if (op + len > cap) return GZB_F_OVERFLOW;
            if ((int)op - (int)dd < -32768) return GZB_F_ERROR;
            for (uint32_t q = 0; q < len; ++q) {
                const int src = (int)op - (int)dd;
                out[op] = src >= 0 ? out[src] : (uint16_t)(GZB_MARKER | (uint32_t)(32768 + src));
                ++op;
            }
        }
        ++i;
    }
    n_sym = op;
    return 0;
}

// symbol space of candidate c of n: ratio_cap x the compressed bytes up to the next candidate (the last one: to the window's end);
// token space: tok_ratio entries per compressed byte, and per lane GZB_OVERLAP_TOKENS more for what it reads of its successor's
// share.  Whatever does not fit overflows, and that block's section is the host's.  (Budgets: a candidate per 16 KiB at most.)
GZB_HD inline uint64_t gzb_sym_budget(uint64_t span, uint32_t ratio_cap) { return span * ratio_cap + (span / 16384u + 32u) * (uint64_t)GZB_SYM_EXTRA; }
GZB_HD inline uint64_t gzb_tok_budget(uint64_t span, uint32_t tok_ratio, uint32_t overlap_tokens) { return span * tok_ratio + (span / 16384u + 32u) * (uint64_t)(GZB_K * overlap_tokens + 64u); }
GZB_HD inline uint32_t gzb_cand_span(const GzbJob& J, uint32_t c, uint32_t n) {
    const uint32_t nxt = c + 1 < n ? J.c_start[c + 1] : J.comp_bytes * 8u;
    return gzb_min((nxt - J.c_start[c] + 7u) >> 3, 2u << 20);        // (a block of more than 2 MiB: overflow, host)
}
GZB_HD inline uint32_t gzb_symcap_of(const GzbJob& J, uint32_t c, uint32_t n) { return ((gzb_cand_span(J, c, n) * J.ratio_cap + GZB_SYM_EXTRA - 8u) + 7u) & ~7u; }
GZB_HD inline uint32_t gzb_tokcap_of(const GzbJob& J, uint32_t c, uint32_t n) {
    return (gzb_cand_span(J, c, n) * J.tok_ratio + (uint32_t)GZB_K * J.overlap_tokens + 2u * (uint32_t)GZB_K - 1u) & ~((uint32_t)GZB_K - 1u);
}

// the usable candidate that starts exactly at bit x (GZB_NONE: none)
GZB_HD inline uint32_t gzb_cand_at(const GzbJob& J, uint32_t n, uint32_t x) {
    const uint32_t i = gzb_lower_bound(J.c_start, n, x);
    return (i < n && J.c_start[i] == x && J.c_flags[i] == 0u) ? i : GZB_NONE;
}
// a non-final stored block at bit e: returns true and its data byte / length / the bit behind it
GZB_HD inline bool gzb_stored_at(const GzbJob& J, uint32_t e, uint32_t& byte, uint32_t& len, uint32_t& next) {
    const uint32_t limit = J.comp_bytes * 8u;
    if (e + 3u + 7u + 32u + 64u > limit) return false;
    const uint32_t h = (uint32_t)(gzb_peek(J.comp, e) & 7u);
    if (h != 0u) return false;                              // BFINAL = 0, BTYPE = 0
    const uint32_t q = (e + 3u + 7u) & ~7u;
    const uint32_t v = (uint32_t)gzb_peek(J.comp, q);
    len = v & 0xffffu;
    if (len != (~(v >> 16) & 0xffffu)) return false;
    byte = (q >> 3) + 4u;
    if ((unsigned long long)byte + len > J.comp_bytes) return false;
    next = q + 32u + 8u * len;
    return true;
}

// section k: which blocks, in which order
GZB_HD inline void gzb_chain_section(const GzbJob& J, uint32_t k) {
    const uint32_t n = J.n_cand[0];
    const uint32_t nom = J.s_nominal[k], stop = J.s_stop[k];
    const bool exact = J.s_exact[k] != 0u;
    // the first usable candidate at or behind the nominal start; one whose end is itself a block start (or reaches the stop
    // bit) is preferred over one that merely parses: that is what a false hit practically never does
    uint32_t pick = GZB_NONE, fallback = GZB_NONE;
    {
        uint32_t i = gzb_lower_bound(J.c_start, n, nom);
        for (uint32_t tries = 0; i < n && tries < 64u; ++i, ++tries) {
            const uint32_t st = J.c_start[i];
            if (exact && st != nom) break;
            if (st >= stop) break;
            if (J.c_flags[i] == 0u) {
                const uint32_t e = J.c_end[i];
                uint32_t b, l, nx;
                if (e >= stop || gzb_cand_at(J, n, e) != GZB_NONE || gzb_stored_at(J, e, b, l, nx)) { pick = i; break; }
                if (fallback == GZB_NONE) fallback = i;
            }
            if (exact) break;
        }
        if (pick == GZB_NONE) pick = fallback;
    }
    uint32_t nb = 0, off = 0, end = 0, start = GZB_NONE;
    if (pick != GZB_NONE) {
        start = J.c_start[pick];
        uint32_t* const blocks = J.s_blocks + (size_t)k * GZB_SEC_BLOCKS * 3u;
        uint32_t cur = pick;
        end = start;
        for (;;) {
            if (cur != GZB_NONE) {
                const uint32_t ns = J.c_nsym[cur];
                if (nb >= (uint32_t)GZB_SEC_BLOCKS || off + ns > J.s_symcap) break;
                blocks[3u * nb] = cur; blocks[3u * nb + 1] = 0; blocks[3u * nb + 2] = off;
                ++nb;
                off += ns;
                end = J.c_end[cur];
            }
            if (end >= stop) break;
            cur = gzb_cand_at(J, n, end);
            if (cur == GZB_NONE) {
                uint32_t b, l, nx;
                if (!gzb_stored_at(J, end, b, l, nx)) break;
                if (nb >= (uint32_t)GZB_SEC_BLOCKS || off + l > J.s_symcap) break;
                blocks[3u * nb] = GZB_STORED | l; blocks[3u * nb + 1] = b; blocks[3u * nb + 2] = off;
                ++nb;
                off += l;
                end = nx;
            }
        }
        if (nb == 0) start = GZB_NONE;
    }
    J.s_start[k] = start;
    J.s_end[k] = end;
    J.s_nsym[k] = off;
    J.s_nblk[k] = nb;
}

// where each section's symbols go in the packed buffer; a section that does not fit any more is dropped (the host decodes it)
GZB_HD inline void gzb_place(const GzbJob& J) {
    unsigned long long o = 0;
    for (uint32_t k = 0; k < J.n_sec; ++k) {
        J.s_off[k] = o;
        if (J.s_start[k] == GZB_NONE) continue;
        const unsigned long long n = ((unsigned long long)J.s_nsym[k] + 31ull) & ~31ull;
        if (o + n > J.s_sym_total) { J.s_start[k] = GZB_NONE; J.s_nsym[k] = 0; J.s_nblk[k] = 0; continue; }
        o += n;
    }
    J.s_off[J.n_sec] = o;
}

// symbol x of a block that begins `off` symbols into its section, for the section's stream dst: a marker j stands for section
// position off - 32768 + j — in an earlier block (already in dst), or still before the section
GZB_HD inline uint16_t gzb_rebase(uint32_t x, uint32_t off, const uint16_t* dst) {
    if (x & GZB_MARKER) {
        const int q = (int)off - 32768 + (int)(x & 0x7fffu);
        x = q >= 0 ? (uint32_t)dst[q] : (GZB_MARKER | (uint32_t)(q + 32768));
    }
    return (uint16_t)x;
}

// ---- RESIDENT results (round 6): markers resolved and CRC-32 computed where the symbols are --------------------------------------
// A run = consecutive sections of one group, each starting at the bit its predecessor ended on.  Section k's markers point into
// W_k, the 32 KiB of output before it: W_0 comes from the consumer (it has committed everything in front of the run), W_{k+1} is
// the last 32 KiB of (W_k ++ text_k).  gzb_window_byte yields the windows one after the other — the only sequential part, 32 KiB
// per section — and with every W_k at hand each symbol of the run resolves on its own (gzb_resolve_sym).  A window entry below
// `valid` does not exist (the member began less than 32 KiB before): a marker that points there is corrupt data.
constexpr uint32_t GZB_WINDOW = 32768u;
constexpr uint32_t GZB_CRC_PIECE = 65536u, GZB_CRC_THREADS = 256u, GZB_CRC_SLOT = GZB_CRC_PIECE / GZB_CRC_THREADS;
struct GzbResolveJob {
    const uint16_t* sym;         // the group's symbols (gzb_gather's output)
    uint8_t* text;               // text[i] = the byte of symbol i
    uint8_t* wins;               // [n_run + 1][GZB_WINDOW]: wins[0] the window before the run (right-aligned), wins[k + 1] the one behind section k
    const uint64_t* off;         // [n_run] first symbol of each section of the run
    const uint32_t* nsym;        // [n_run]
    uint32_t n_run;
    uint32_t valid0;             // entries [0, valid0) of wins[0] do not exist
    uint32_t* bad;               // [0] set to 1 by a marker that points at a byte that does not exist
    // CRC-32: pieces of GZB_CRC_PIECE bytes, RIGHT-aligned in their section (only a section's first piece is short: leading zeros do
    // not disturb a raw CRC); piece p belongs to section piece_sec[p] and is that section's piece_idx[p]-th of piece_cnt
    const uint32_t* piece_sec;
    const uint32_t* piece_idx;
    uint32_t* piece_crc;         // raw CRC (register starts at 0, no final complement) of each piece
    uint32_t* piece_nl;          // line feeds in each piece (the pipe cuts its chunks at the 4K-th one without looking at the text)
    uint32_t n_pieces;
    const uint32_t* crc_tab;     // [GZB_CRC_TAB_WORDS] (gzb_crc_tables)
};
// entries of W_k that do not exist, given W_0's
GZB_HD inline uint32_t gzb_window_valid(const GzbResolveJob& R, uint32_t k) {
    uint32_t v = R.valid0;
    for (uint32_t i = 0; i < k && v; ++i) v = R.nsym[i] >= v ? 0u : v - R.nsym[i];
    return v;
}
// W_{k+1}[j] from W_k (w: GZB_WINDOW bytes) and section k's last symbols; *bad |= 1 on a marker into the void
GZB_HD inline uint8_t gzb_window_byte(const GzbResolveJob& R, uint32_t k, uint32_t j, const uint8_t* w, uint32_t valid, uint32_t* bad) {
    const uint32_t n = R.nsym[k];
    if (n < GZB_WINDOW && j < GZB_WINDOW - n) return w[j + n];
    const uint32_t i = n >= GZB_WINDOW ? n - GZB_WINDOW + j : j - (GZB_WINDOW - n);
    const uint32_t v = R.sym[R.off[k] + i];
    if (v & GZB_MARKER) {
        const uint32_t q = v & 0x7fffu;
        if (q < valid) *bad = 1u;
        return w[q];
    }
    return (uint8_t)v;
}
GZB_HD inline uint8_t gzb_resolve_sym(uint32_t v, const uint8_t* w, uint32_t valid, uint32_t* bad) {
    if (v & GZB_MARKER) {
        const uint32_t q = v & 0x7fffu;
        if (q < valid) *bad = 1u;
        return w[q];
    }
    return (uint8_t)v;
}
// raw CRC of slot t of piece (section k, index pi of cnt): bytes [lo, lo + GZB_CRC_SLOT) of the section's text, bytes before the
// section's start counting as zeros (lo may be < 0 in a section's first piece).  Four bytes per step (slicing: tab = T0 T1 T2 T3,
// 256 words each — T0 the byte table, T_{i+1}[x] = T0[T_i[x] & 255] ^ T_i[x] >> 8): one dependent table look-up per word, not per byte.
typedef uint32_t gzb_u32_unaligned __attribute__((aligned(1)));
GZB_HD inline uint32_t gzb_crc_slot(const uint8_t* text, uint32_t n, uint32_t cnt, uint32_t pi, uint32_t t, const uint32_t* tab, uint32_t* nl) {
    const long long lo = (long long)n - (long long)(cnt - pi) * GZB_CRC_PIECE + (long long)t * GZB_CRC_SLOT;
    uint32_t c = 0, lf = 0;
    for (uint32_t b = 0; b < GZB_CRC_SLOT; b += 4u) {
        const long long p = lo + b;
        uint32_t x = 0;
        if (p >= 0) x = *reinterpret_cast<const gzb_u32_unaligned*>(text + p);
        else if (p > -4) for (int q = (int)-p; q < 4; ++q) x |= (uint32_t)text[p + q] << (8 * q);
        {
            // line feeds among the four bytes (a byte that does not exist is 0, never 0x0a)
            const uint32_t y = x ^ 0x0a0a0a0au;
            const uint32_t z = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);      // 0x80 in every zero byte of y
#if defined(__HIP_DEVICE_COMPILE__)
            lf += (uint32_t)__popc(z);
#else
            lf += (uint32_t)__builtin_popcount(z);
#endif
        }
        c ^= x;
        c = tab[768u + (c & 0xffu)] ^ tab[512u + ((c >> 8) & 0xffu)] ^ tab[256u + ((c >> 16) & 0xffu)] ^ tab[c >> 24];
    }
    *nl = lf;
    return c;
}
constexpr uint32_t GZB_CRC_TAB_WORDS = 1024u + 8u * 32u;    // T0..T3, then [8][32]: "advance the register by GZB_CRC_SLOT << k zero bytes", column j = image of bit j
// left ++ right, where right is `GZB_CRC_SLOT << level` bytes long: the left register advanced over them, then added
GZB_HD inline uint32_t gzb_crc_join(uint32_t left, uint32_t right, int level, const uint32_t* tab) {
    const uint32_t* M = tab + 1024 + 32 * level;
    uint32_t r = right;
    for (int j = 0; j < 32; ++j) r ^= ((left >> j) & 1u) ? M[j] : 0u;
    return r;
}
// the tables: `advance(x, n)` = the CRC register x after n zero bytes (zlib: crc32_combine(x, 0, n))
template <class Advance>
inline void gzb_crc_tables(uint32_t* tab, Advance advance) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t v = i;
        for (int k = 0; k < 8; ++k) v = (v >> 1) ^ (0xEDB88320u & (0u - (v & 1u)));
        tab[i] = v;
    }
    for (int s = 1; s < 4; ++s)
        for (uint32_t i = 0; i < 256; ++i) tab[256 * s + i] = tab[tab[256 * (s - 1) + i] & 0xffu] ^ (tab[256 * (s - 1) + i] >> 8);
    for (int k = 0; k < 8; ++k)
        for (int j = 0; j < 32; ++j) tab[1024 + 32 * k + j] = advance(1u << j, (uint64_t)GZB_CRC_SLOT << k);
}
// a section's CRC-32 from its pieces' raw CRCs (in order): the register advanced over each next piece, then the start value
// 0xffffffff advanced over the whole section and the final complement (crc32(M) = raw(M) ^ ~advance(~0, |M|))
template <class Advance>
inline uint32_t gzb_crc_fold(const uint32_t* piece, uint32_t cnt, uint64_t n, const uint32_t* adv_piece /* [32]: advance by GZB_CRC_PIECE, by column */, Advance advance) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
        uint32_t a = 0;
        for (int j = 0; j < 32; ++j) a ^= ((r >> j) & 1u) ? adv_piece[j] : 0u;
        r = a ^ piece[i];
    }
    return r ^ ~advance(0xffffffffu, n);
}

#if defined(__HIPCC__)
// the windows, one after the other: ONE workgroup, W_k in LDS
constexpr int GZB_WIN_THREADS = 1024;
__global__ __launch_bounds__(GZB_WIN_THREADS) void gzb_windows_kernel(GzbResolveJob R) {
    __shared__ uint8_t s_w[GZB_WINDOW];
    const uint32_t tid = threadIdx.x;
    for (uint32_t j = tid * 16u; j < GZB_WINDOW; j += GZB_WIN_THREADS * 16u) *reinterpret_cast<uint4*>(s_w + j) = *reinterpret_cast<const uint4*>(R.wins + j);
    __syncthreads();
    uint32_t valid = R.valid0, bad = 0;
    for (uint32_t k = 0; k < R.n_run; ++k) {
        uint8_t v[GZB_WINDOW / GZB_WIN_THREADS];
#pragma unroll
        for (uint32_t r = 0; r < GZB_WINDOW / GZB_WIN_THREADS; ++r) v[r] = gzb_window_byte(R, k, tid + r * GZB_WIN_THREADS, s_w, valid, &bad);
        __syncthreads();
        uint8_t* const out = R.wins + (size_t)(k + 1) * GZB_WINDOW;
#pragma unroll
        for (uint32_t r = 0; r < GZB_WINDOW / GZB_WIN_THREADS; ++r) { s_w[tid + r * GZB_WIN_THREADS] = v[r]; out[tid + r * GZB_WIN_THREADS] = v[r]; }
        __syncthreads();
        const uint32_t n = R.nsym[k];
        valid = n >= valid ? 0u : valid - n;
    }
    if (bad) R.bad[0] = 1u;
}
// every symbol of the run on its own: 16 per thread, blockIdx.y = section
constexpr int GZB_RES_THREADS = 256;
__global__ __launch_bounds__(GZB_RES_THREADS) void gzb_resolve_kernel(GzbResolveJob R) {
    const uint32_t k = blockIdx.y;
    const uint32_t n = R.nsym[k];
    const uint32_t i0 = (blockIdx.x * (uint32_t)GZB_RES_THREADS + threadIdx.x) * 16u;
    if (i0 >= n) return;
    const uint8_t* const w = R.wins + (size_t)k * GZB_WINDOW;
    const uint32_t valid = gzb_window_valid(R, k);
    const uint16_t* const s = R.sym + R.off[k] + i0;        // (section starts are multiples of 32 symbols: 16-byte loads and stores)
    uint8_t* const d = R.text + R.off[k] + i0;
    const uint4 a = reinterpret_cast<const uint4*>(s)[0], b = reinterpret_cast<const uint4*>(s)[1];
    const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t bad = 0, o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t b0 = gzb_resolve_sym(x[2 * q] & 0xffffu, w, valid, &bad), b1 = gzb_resolve_sym(x[2 * q] >> 16, w, valid, &bad),
                       b2 = gzb_resolve_sym(x[2 * q + 1] & 0xffffu, w, valid, &bad), b3 = gzb_resolve_sym(x[2 * q + 1] >> 16, w, valid, &bad);
        o[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    // (symbols behind the section's end are padding: whatever they resolve to is never read, and never counts as corrupt)
    if (i0 + 16u <= n) { if (bad) R.bad[0] = 1u; *reinterpret_cast<uint4*>(d) = make_uint4(o[0], o[1], o[2], o[3]); }
    else {
        uint32_t bad2 = 0;
        for (uint32_t i = 0; i < n - i0; ++i) d[i] = gzb_resolve_sym(s[i], w, valid, &bad2);
        if (bad2) R.bad[0] = 1u;
    }
}
// raw CRC of every piece: a thread per slot, then a tree of joins
__global__ __launch_bounds__(GZB_CRC_THREADS) void gzb_crc_kernel(GzbResolveJob R) {
    __shared__ uint32_t s_tab[GZB_CRC_TAB_WORDS];
    __shared__ uint32_t s_part[GZB_CRC_THREADS];
    __shared__ uint32_t s_nl;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < GZB_CRC_TAB_WORDS; i += GZB_CRC_THREADS) s_tab[i] = R.crc_tab[i];
    if (tid == 0) s_nl = 0;
    __syncthreads();
    const uint32_t p = blockIdx.x, k = R.piece_sec[p], n = R.nsym[k];
    const uint32_t cnt = (n + GZB_CRC_PIECE - 1u) / GZB_CRC_PIECE;
    uint32_t lf = 0;
    s_part[tid] = gzb_crc_slot(R.text + R.off[k], n, cnt, R.piece_idx[p], tid, s_tab, &lf);
    if (lf) atomicAdd(&s_nl, lf);
    __syncthreads();
#pragma unroll 1
    for (int level = 0; level < 8; ++level) {
        const uint32_t step = 1u << level;
        const bool active = (tid & (2u * step - 1u)) == 0u;
        uint32_t r = 0;
        if (active) r = gzb_crc_join(s_part[tid], s_part[tid + step], level, s_tab);
        __syncthreads();
        if (active) s_part[tid] = r;
        __syncthreads();
    }
    if (tid == 0) { R.piece_crc[p] = s_part[0]; R.piece_nl[p] = s_nl; }
}

// ---- every block start of the batch -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GZB_SCAN_THREADS) void gzb_scan_kernel(GzbJob J) {
    __shared__ uint8_t s_kraft[512];
    __shared__ uint8_t s_cl[128 * GZB_SCAN_THREADS];
    __shared__ uint32_t s_found[GZB_TILE_CAND];
    __shared__ uint32_t s_n;
    const int tid = threadIdx.x;
    for (int i = tid; i < 512; i += GZB_SCAN_THREADS) s_kraft[i] = (uint8_t)gzb_kraft9((uint32_t)i);
    if (tid == 0) s_n = 0;
    __syncthreads();
    const uint32_t byte0 = J.scan_byte0 + blockIdx.x * (uint32_t)GZB_SCAN_TILE + (uint32_t)tid * 16u;
    const uint32_t limit_bit = J.comp_bytes * 8u;
    uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (byte0 < J.comp_bytes) {                             // (the buffer is padded: 32 bytes are readable from any byte of the data)
        const uint4 a = *reinterpret_cast<const uint4*>(J.comp + byte0), b = *reinterpret_cast<const uint4*>(J.comp + byte0 + 16);
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    }
    uint32_t m[4], k2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t mm = gzb_quick32(((unsigned long long)d[i + 1] << 32) | d[i]);
        // keep [first_bit, last_bit)
        const uint32_t b0 = (byte0 + 4u * (uint32_t)i) * 8u;
        if (b0 + 32u <= J.first_bit || b0 >= J.last_bit || byte0 >= J.comp_bytes) mm = 0;
        else {
            if (b0 < J.first_bit) mm &= ~0u << (J.first_bit - b0);
            if (b0 + 32u > J.last_bit) mm &= (1u << (J.last_bit - b0)) - 1u;
        }
        m[i] = mm;
    }
    // (a lane pops its own survivors: every iteration of these loops does useful work on most lanes)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (the 57 bits the test looks at, from 17 to 48 bits behind this word's start, are in the lane's registers: a survivor
        //  costs no trip to memory — there are a dozen per lane)
        const unsigned long long v64 = ((unsigned long long)d[i + 1] << 32) | d[i];
        const unsigned long long hi64 = ((unsigned long long)d[i + 3 < 8 ? i + 3 : 7] << 32) | d[i + 2];
        uint32_t mm = m[i];
        while (mm) {
            const uint32_t bit = (uint32_t)__builtin_ctz(mm);
            mm &= mm - 1;
            const uint32_t hclen = ((uint32_t)(v64 >> (bit + 13u)) & 15u) + 4u;
            const uint32_t o = bit + 17u;                                       // 17 .. 48
            if (gzb_kraft_ok_v((v64 >> o) | (hi64 << (64u - o)), hclen, s_kraft)) k2[i] |= 1u << bit;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t mm = k2[i];
        while (mm) {
            const uint32_t bit = (uint32_t)__builtin_ctz(mm);
            mm &= mm - 1;
            const uint32_t p = (byte0 + 4u * (uint32_t)i) * 8u + bit;
            uint32_t data_bit, hlit, hdist;
            if (gzb_header(J.comp, limit_bit, p, s_cl + tid, GZB_SCAN_THREADS, nullptr, data_bit, hlit, hdist)) {
                const uint32_t at = atomicAdd(&s_n, 1u);
                if (at < (uint32_t)GZB_TILE_CAND) s_found[at] = p;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t n = gzb_min(s_n, (uint32_t)GZB_TILE_CAND);
        for (uint32_t a = 1; a < n; ++a) {                  // (a handful at most: insertion sort)
            const uint32_t x = s_found[a];
            uint32_t b = a;
            while (b > 0 && s_found[b - 1] > x) { s_found[b] = s_found[b - 1]; --b; }
            s_found[b] = x;
        }
        J.tile_cnt[blockIdx.x] = n;
        for (uint32_t a = 0; a < n; ++a) J.tile_cand[blockIdx.x * (uint32_t)GZB_TILE_CAND + a] = s_found[a];
    }
}

// ---- one sorted candidate list + each candidate's symbol space -------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gzb_compact_kernel(GzbJob J) {
    __shared__ uint32_t s_part[1024];
    __shared__ unsigned long long s_part64[1024];
    __shared__ uint32_t s_total;
    const uint32_t tid = threadIdx.x;
    // (1) tile counts -> offsets (a thread owns a contiguous run of tiles)
    const uint32_t per = (J.n_tiles + 1023u) / 1024u;
    const uint32_t t0 = gzb_min(J.n_tiles, tid * per), t1 = gzb_min(J.n_tiles, t0 + per);
    uint32_t mine = 0;
    for (uint32_t t = t0; t < t1; ++t) mine += J.tile_cnt[t];
    s_part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint32_t c = s_part[i]; s_part[i] = run; run += c; }
        s_total = gzb_min(run, J.cand_cap);
        J.n_cand[0] = s_total;
        J.n_cand[1] = 0;
    }
    __syncthreads();
    {
        uint32_t o = s_part[tid];
        for (uint32_t t = t0; t < t1; ++t) {
            const uint32_t c = J.tile_cnt[t];
            for (uint32_t a = 0; a < c; ++a, ++o)
                if (o < J.cand_cap) J.c_start[o] = J.tile_cand[t * (uint32_t)GZB_TILE_CAND + a];
        }
    }
    __threadfence_block();
    __syncthreads();
    // (2) symbol space
    const uint32_t n = s_total;
    const uint32_t per2 = (n + 1023u) / 1024u;
    const uint32_t c0 = gzb_min(n, tid * per2), c1 = gzb_min(n, c0 + per2);
    unsigned long long sum = 0, tsum = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t cap = gzb_symcap_of(J, c, n), tcap = gzb_tokcap_of(J, c, n);
        J.c_symcap[c] = cap;
        J.c_tokcap[c] = tcap;
        sum += cap;
        tsum += tcap;
    }
    s_part64[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned long long c = s_part64[i]; s_part64[i] = run; run += c; }
        if (run > J.blk_sym_cap) J.n_cand[1] = 1;
    }
    __syncthreads();
    unsigned long long o = s_part64[tid];
    __syncthreads();
    s_part64[tid] = tsum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { const unsigned long long c = s_part64[i]; s_part64[i] = run; run += c; }
        if (run > J.blk_tp_cap) J.n_cand[1] = 1;
    }
    __syncthreads();
    unsigned long long to = s_part64[tid];
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t cap = J.c_symcap[c], tcap = J.c_tokcap[c];
        J.c_symoff[c] = o;
        J.c_tokoff[c] = to;
        // candidates that do not fit the symbol / token buffer are not decoded (their sections fall back to the host)
        if (o + cap > J.blk_sym_cap || to + tcap > J.blk_tp_cap) J.c_symcap[c] = 0;
        o += cap;
        to += tcap;
    }
}

// ---- a lane per block --------------------------------------------------------------------------------------------------------------
// PHASE 1 runs in SLICES of J.slice_tokens tokens per lane and launch: a kernel that sits on its CUs for tens of milliseconds
// keeps the filter's kernels — which want every CU, with all its registers — waiting behind it (measured: the first wiring made
// `.gz -> .gz` slower than the host alone).  Between two slices they get their turn.
// gzb_tables_kernel: a lane per candidate — header, tables, where its GZB_K lanes start and stop.
__global__ __launch_bounds__(GZB_DEC_THREADS) void gzb_tables_kernel(GzbJob J) {
    __shared__ uint8_t s_cl[128 * GZB_DEC_THREADS];
    __shared__ uint32_t s_cnt[16 * GZB_DEC_THREADS], s_nxt[16 * GZB_DEC_THREADS], s_off[16 * GZB_DEC_THREADS];
    const int tid = threadIdx.x;
    const uint32_t c = blockIdx.x * (uint32_t)GZB_DEC_THREADS + (uint32_t)tid;
    const uint32_t n = J.n_cand[0];
    if (c >= n) return;
    const uint32_t limit_bit = J.comp_bytes * 8u;
    uint32_t* const tw = J.tables + (size_t)c * GZB_TAB_WORDS;
    const GzbLaneTab<1> T{reinterpret_cast<uint16_t*>(tw)};
    uint8_t* const lens = reinterpret_cast<uint8_t*>(tw + GZB_TAB_ENTRIES / 2);
    uint32_t p = 0, hlit = 0, hdist = 0, fl = 0;
    if (J.c_symcap[c] == 0) fl = GZB_F_SKIP;
    else if (!gzb_header(J.comp, limit_bit, J.c_start[c], s_cl + tid, GZB_DEC_THREADS, lens, p, hlit, hdist)) fl = GZB_F_ERROR;
    if (!fl) {
        gzb_build<true>(lens, hlit, T, s_cnt + tid, s_nxt + tid, s_off + tid, GZB_DEC_THREADS);
        gzb_build<false>(lens + hlit, hdist, T, s_cnt + tid, s_nxt + tid, s_off + tid, GZB_DEC_THREADS);
        gzb_plan_lanes(J, c, n, p);
    } else {
        J.c_lanes[c] = 0;
        for (uint32_t k = 0; k < (uint32_t)GZB_K; ++k) J.l_flags[c * GZB_K + k] = 0;
    }
    J.c_flags[c] = fl;
    J.c_nsym[c] = 0;
    J.c_end[c] = 0;
}

// gzb_decode_kernel: a lane per (candidate, entry point) reads one slice of tokens.  The 64 lanes of a workgroup belong to
// 64 / GZB_K blocks, whose tables (11 KB each) are copied into LDS first: what a lane waits for per token is then an LDS
// look-up, not a trip to L2 (1.4 us per token measured with the tables in global memory, the lanes spending their time waiting).
constexpr int GZB_DEC_BLOCKS = GZB_DEC_THREADS / GZB_K;
// A lane's window of the stream: 32 words in a column of an LDS array (word j at [(j % 32) * GZB_DEC_THREADS]).  At sync(), every
// GZB_SYNC_STEPS tokens and for all lanes at once, the pieces asked for at the previous sync() (32 bytes each, <= 2) are put
// into the window and new ones are asked for, as many as there is room for.  A token takes <= 48 bits, GZB_SYNC_STEPS of them
// <= 12 words: a lane leaves sync() with more than 12 words at hand or on their way, so it never runs dry, and what arrives
// only overwrites words it had consumed when it asked.
struct GzbInLds {
    const uint4* comp16;
    uint32_t* ring;
    uint4 pend[4];
    uint32_t wi, wf, kp;         // next word to hand out; words [.., wf) are in the window or on their way; pieces on their way
    __device__ void put(uint32_t word, const uint4& v) {
        const uint32_t w = word & 31u;
        ring[(w + 0u) * GZB_DEC_THREADS] = v.x; ring[(w + 1u) * GZB_DEC_THREADS] = v.y;
        ring[(w + 2u) * GZB_DEC_THREADS] = v.z; ring[(w + 3u) * GZB_DEC_THREADS] = v.w;
    }
    __device__ void start(uint32_t word) {
        wi = word;
        const uint32_t base = word & ~3u;           // (>= 29 words at hand: the bit buffer takes two, eight tokens <= 12, > 12 are left at the first sync())
        uint4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = comp16[(base >> 2) + (uint32_t)j];
#pragma unroll
        for (int j = 0; j < 8; ++j) put(base + 4u * (uint32_t)j, a[j]);
        wf = base + 32u;
        kp = 0;
    }
    __device__ void sync() {
        // what was asked for last time has had GZB_SYNC_STEPS tokens' time to arrive
        if (kp > 0u) { put(wf - 8u * kp, pend[0]); put(wf - 8u * kp + 4u, pend[1]); }
        if (kp > 1u) { put(wf - 8u, pend[2]); put(wf - 4u, pend[3]); }
        const uint32_t room = 32u - (wf - wi);           // (wf - wi <= 32 always)
        kp = room >> 3;
        if (kp > 2u) kp = 2u;
        if (kp > 0u) { pend[0] = comp16[wf >> 2]; pend[1] = comp16[(wf >> 2) + 1u]; }
        if (kp > 1u) { pend[2] = comp16[(wf >> 2) + 2u]; pend[3] = comp16[(wf >> 2) + 3u]; }
        wf += 8u * kp;
    }
    __device__ uint32_t next() {
        const uint32_t v = ring[(wi & 31u) * GZB_DEC_THREADS];
        ++wi;
        return v;
    }
};

__global__ __launch_bounds__(GZB_DEC_THREADS) void gzb_decode_kernel(GzbJob J) {
    __shared__ uint32_t s_tab[GZB_DEC_BLOCKS][GZB_TAB_ENTRIES / 2];
    __shared__ uint32_t s_ring[32 * GZB_DEC_THREADS];
    const uint32_t i = blockIdx.x * (uint32_t)GZB_DEC_THREADS + threadIdx.x;
    const uint32_t c = i / (uint32_t)GZB_K, k = i % (uint32_t)GZB_K;
    const bool live = c < J.n_cand[0] && J.l_flags[i] == GZB_F_MORE;
    if (!__syncthreads_or(live ? 1 : 0)) return;
    {
        // GZB_K lanes copy their block's tables, four words at a time (a block none of whose lanes has work left is skipped)
        const uint32_t b = threadIdx.x / (uint32_t)GZB_K;
        const bool any = __ballot(live) & (((1ull << GZB_K) - 1ull) << (b * GZB_K));
        if (any) {
            const uint4* const src = reinterpret_cast<const uint4*>(J.tables + (size_t)c * GZB_TAB_WORDS);
            uint4* const dst = reinterpret_cast<uint4*>(s_tab[b]);
            for (uint32_t w = k; w < (uint32_t)(GZB_TAB_ENTRIES / 8); w += (uint32_t)GZB_K) dst[w] = src[w];
        }
    }
    __syncthreads();
    if (!live) return;
    const uint32_t lanes = J.c_lanes[c];
    const GzbLaneTab<1> T{reinterpret_cast<uint16_t*>(s_tab[threadIdx.x / (uint32_t)GZB_K])};
    const uint32_t share = J.c_tokcap[c] / (uint32_t)GZB_K;
    const size_t at = J.c_tokoff[c] + (size_t)k * share;
    uint32_t p = J.l_p[i], nt = J.l_ntok[i];
    GzbInLds in;
    in.comp16 = reinterpret_cast<const uint4*>(J.comp);
    in.ring = s_ring + threadIdx.x;
    const uint32_t fl = gzb_tokenize(in, J.comp_bytes * 8u, T, J.blk_tp + at, lanes == 1u ? share * (uint32_t)GZB_K : share, p, nt, J.l_stop[i], J.slice_tokens, lanes != 1u);
    J.l_p[i] = p;
    J.l_ntok[i] = nt;
    J.l_flags[i] = fl;
}

// PHASE 2 — a WAVE per block turns its lanes' token lists into symbols (gzb_stitch_expand is the same thing on the host): 64
// tokens at a time.  Where the chunk reaches into the next lane's share every lane looks its token's bit position up in that
// lane's list (a binary search); the first lane that finds it is where the two reading frames have met: the tokens before it
// are applied and the walk goes on in the next list.  Output positions come from a lane scan, the chunk's literals are stored
// at once, then the matches that copy from before the chunk all together, then the few that copy from the chunk itself in
// order with the 64 lanes sharing each copy (a copy that overlaps itself repeats its period).
// A wave's loads and stores to global memory are performed in issue order, so a copy sees what the one before it wrote.
constexpr int GZB_EXP_WAVES = 4;
__global__ __launch_bounds__(64 * GZB_EXP_WAVES) void gzb_expand_kernel(GzbJob J) {
    const uint32_t c = blockIdx.x * (uint32_t)GZB_EXP_WAVES + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63u);
    if (c >= J.n_cand[0] || J.c_flags[c] != 0u) return;
    const uint32_t lanes = J.c_lanes[c], cap = J.c_symcap[c];
    const uint32_t share = J.c_tokcap[c] / (uint32_t)GZB_K;
    const unsigned long long* const tp0 = J.blk_tp + J.c_tokoff[c];
    uint16_t* const out = J.blk_sym + J.c_symoff[c];
    uint32_t fl = 0;
    for (uint32_t q = 0; q < lanes; ++q) {
        const uint32_t f = J.l_flags[c * GZB_K + q];
        if (f != 0u) fl = f == GZB_F_MORE ? GZB_F_OVERFLOW : f;
    }
    uint32_t k = 0, i = 0, op = 0, end_bit = 0;
    bool done = false;
    while (!fl && !done) {
        const unsigned long long* const tp = tp0 + (size_t)k * share;
        const uint32_t n = J.l_ntok[c * GZB_K + k];
        if (i >= n) { fl = GZB_F_ERROR; break; }              // the list ran out before it met the next one / an end of block
        const bool have = i + (uint32_t)lane < n;
        const unsigned long long e = have ? tp[i + (uint32_t)lane] : (unsigned long long)GZB_T_JUNK;
        const uint32_t t = (uint32_t)e, at = (uint32_t)(e >> 32);
        // does the next lane's list have a token at this very bit?
        uint32_t jn = GZB_NONE;
        if (k + 1 < lanes) {
            const uint32_t nstart = J.l_start[c * GZB_K + k + 1];
            if (__ballot(have && at >= nstart)) {
                const unsigned long long* const ntp = tp0 + (size_t)(k + 1) * share;
                const uint32_t nn = J.l_ntok[c * GZB_K + k + 1];
                if (have && at >= nstart) {
                    const uint32_t j = gzb_lower_bound_pos(ntp, nn, at);
                    if (j < nn && (uint32_t)(ntp[j] >> 32) == at) jn = j;
                }
            }
        }
        const unsigned long long m_meet = __ballot(jn != GZB_NONE), m_eob = __ballot(have && t == GZB_T_EOB),
                                 m_junk = __ballot((t & GZB_T_JUNK) != 0u);        // (lanes behind the list's end hold JUNK too)
        const unsigned long long m_cut = m_meet | m_eob | m_junk;
        const int cut = m_cut ? __ffsll((long long)m_cut) - 1 : 64;       // tokens of lanes < cut are applied
        const bool use = lane < cut;
        const bool lit = (t >> 31) != 0u;
        const uint32_t len = use ? (lit ? 1u : (t >> 16) & 0x1ffu) : 0u;
        uint32_t inc = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
            if (lane >= d) inc += o;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (op + total > cap) { fl = GZB_F_OVERFLOW; break; }
        const uint32_t opos = op + inc - len;
        if (use && lit) out[opos] = (uint16_t)(t & 0xffu);
        if (__ballot(use && !lit && (int)opos - (int)((t & 0x7fffu) + 1u) < -32768)) { fl = GZB_F_ERROR; break; }
        // The chunk's matches.  Most of them copy from before the chunk (its output is a few hundred symbols, FASTQ's distances
        // are thousands): those do not depend on anything this chunk writes and are done TOGETHER — their symbols numbered
        // through by a second lane scan, 64 of them per step whichever match they belong to (found by a search over the lanes'
        // running sums), four steps' loads in flight before the first store.  (One match at a time, the lanes sharing its copy:
        // a trip to the cache per match, 12 us per chunk, most lanes idle.)
        const uint32_t dd = (t & 0x7fffu) + 1u;
        const bool is_match = use && !lit;
        const bool indep_ok = is_match && ((int)opos - (int)dd + (int)gzb_min(len, dd) <= (int)op);
        uint32_t isum = indep_ok ? len : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)isum, d, 64);
            if (lane >= d) isum += o;
        }
        const uint32_t itotal = (uint32_t)__builtin_amdgcn_readlane((int)isum, 63);
        const uint32_t iex = isum - (indep_ok ? len : 0u);
        for (uint32_t j0 = 0; j0 < itotal; j0 += 256u) {
            uint32_t dst_at[4];
            int src_at[4];
            uint16_t val[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t j = j0 + 64u * (uint32_t)r + (uint32_t)lane;
                // the first lane whose running sum exceeds j owns symbol j
                uint32_t own = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const uint32_t probe = own + (uint32_t)step - 1u;
                    const uint32_t v = (uint32_t)__shfl((int)isum, (int)probe, 64);
                    if (v <= j) own += (uint32_t)step;
                }
                own &= 63u;
                const uint32_t o_pos = (uint32_t)__shfl((int)opos, (int)own, 64), o_dd = (uint32_t)__shfl((int)dd, (int)own, 64),
                               o_ex = (uint32_t)__shfl((int)iex, (int)own, 64);
                const uint32_t rel = j - o_ex;
                dst_at[r] = j < itotal ? o_pos + rel : 0xffffffffu;
                src_at[r] = (int)o_pos - (int)o_dd + (int)(rel < o_dd ? rel : rel % o_dd);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                val[r] = (dst_at[r] != 0xffffffffu && src_at[r] >= 0) ? out[src_at[r]] : (uint16_t)(GZB_MARKER | (uint32_t)(32768 + src_at[r]));
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (dst_at[r] != 0xffffffffu) out[dst_at[r]] = val[r];
        }
        // the others (their source lies in this chunk's output) in order, the lanes sharing each copy
        unsigned long long mm = __ballot(is_match && !indep_ok);
        while (mm) {
            const int l = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const uint32_t m_len = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
            const uint32_t m_dd = (uint32_t)__builtin_amdgcn_readlane((int)dd, l);
            const uint32_t m_pos = (uint32_t)__builtin_amdgcn_readlane((int)opos, l);
            const int src = (int)m_pos - (int)m_dd;
            for (uint32_t j0 = 0; j0 < m_len; j0 += 64u) {
                const uint32_t j = j0 + (uint32_t)lane;
                if (j < m_len) {
                    const uint32_t rel = m_dd < m_len ? j % m_dd : j;
                    const int q = src + (int)rel;
                    out[m_pos + j] = q >= 0 ? out[q] : (uint16_t)(GZB_MARKER | (uint32_t)(32768 + q));
                }
            }
        }
        op += total;
        if (cut == 64) { i += 64u; continue; }
        // what stopped the chunk at lane `cut`: the frames met (first: the next list takes over AT this token), the block's end, or nonsense
        if ((m_meet >> cut) & 1ull) {
            i = (uint32_t)__builtin_amdgcn_readlane((int)jn, cut);
            ++k;
        } else if ((m_eob >> cut) & 1ull) {
            const uint32_t ci = i + (uint32_t)cut;
            end_bit = ci + 1u < n ? (uint32_t)(tp[ci + 1u] >> 32) : J.l_p[c * GZB_K + k];
            done = true;
        } else fl = GZB_F_ERROR;
    }
    if (lane == 0) {
        J.c_flags[c] = fl;
        J.c_nsym[c] = fl ? 0u : op;
        J.c_end[c] = end_bit;
    }
}

// ---- a lane per section --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gzb_chain_kernel(GzbJob J) {
    const uint32_t k = blockIdx.x * 64u + threadIdx.x;
    if (k < J.n_sec) gzb_chain_section(J, k);
}

__global__ void gzb_place_kernel(GzbJob J) { gzb_place(J); }

// ---- a workgroup per section: the blocks' symbols as one stream, markers relative to the section ---------------------------------
__global__ __launch_bounds__(GZB_GATHER_THREADS) void gzb_gather_kernel(GzbJob J) {
    const uint32_t k = blockIdx.x;
    const uint32_t nb = J.s_nblk[k];
    uint16_t* const dst = J.s_sym + J.s_off[k];
    const uint32_t* const blocks = J.s_blocks + (size_t)k * GZB_SEC_BLOCKS * 3u;
    for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t w0 = blocks[3u * b], w1 = blocks[3u * b + 1], off = blocks[3u * b + 2];
        if (w0 & GZB_STORED) {
            const uint32_t len = w0 & 0xffffu;
            const uint8_t* const s = J.comp + w1;
            for (uint32_t i = threadIdx.x; i < len; i += GZB_GATHER_THREADS) dst[off + i] = s[i];
        } else {
            const uint32_t n = J.c_nsym[w0];
            const uint16_t* const s = J.blk_sym + J.c_symoff[w0];
            for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 4u * GZB_GATHER_THREADS) {
                uint32_t v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const uint32_t i = i0 + (uint32_t)r * GZB_GATHER_THREADS; v[r] = i < n ? (uint32_t)s[i] : 0u; }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t i = i0 + (uint32_t)r * GZB_GATHER_THREADS;
                    if (i < n) dst[off + i] = gzb_rebase(v[r], off, dst);
                }
            }
        }
        // (the next block's markers may point into this one)
        __threadfence_block();
        __syncthreads();
    }
}
#endif  // __HIPCC__

}  // namespace aqc
