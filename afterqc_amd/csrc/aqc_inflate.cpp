// aqc_inflate.cpp — DEFLATE decoding for the pipe's gzip source (see aqc_gz.hpp).  Written from RFC 1951 / RFC 1952.
#include <zlib.h>      // crc32 / crc32_combine only

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "aqc_gz.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace aqcgz {

namespace {

inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline void store64(void* p, uint64_t v) { memcpy(p, &v, 8); }

inline uint32_t rev16(uint32_t x) {
    x = ((x & 0x5555u) << 1) | ((x >> 1) & 0x5555u);
    x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
    x = ((x & 0x0f0fu) << 4) | ((x >> 4) & 0x0f0fu);
    return ((x & 0x00ffu) << 8) | (x >> 8);
}
inline uint32_t revbits(uint32_t c, int len) { return rev16(c) >> (16 - len); }

// litlen entries: low byte = bits to consume; bits 8-12 extra bits; bits 16-24 literal / base length
// a literal entry may carry a SECOND literal (bits 8-15, flag E_PAIR): both codes fit the root index together, one lookup
// yields two bytes — FASTQ's frequent symbols have 2-4 bit codes, so most lookups do
constexpr uint32_t E_LIT = 0x80000000u, E_SUB = 0x40000000u, E_EOB = 0x20000000u, E_BAD = 0x10000000u, E_PAIR = 0x08000000u;
// distance entries: bits 0-3 bits to consume, 4-7 extra bits, 8-23 base distance
constexpr uint32_t D_SUB = 0x80000000u, D_BAD = 0x40000000u;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t lit_value(int sym) {
    if (sym < 256) return E_LIT | ((uint32_t)sym << 16);
    if (sym == 256) return E_EOB;
    if (sym < 286) return ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 8);
    return E_BAD;
}
inline uint32_t dist_value(int sym) {
    if (sym < 30) return ((uint32_t)DIST_BASE[sym] << 8) | ((uint32_t)DIST_EXTRA[sym] << 4);
    return D_BAD;
}

// Canonical Huffman code (lens[0, n), 0 = unused) -> root table of 2^R entries + subtables for the longer codes.
// *complete: the code uses the whole code space.  false: over-subscribed, or the table does not fit.  Entries no code leads to
// hold `empty` (the decoders' BAD flag, so that the hot loops need no separate test for them).
template <class Value>
bool build_table(const uint8_t* lens, int n, int R, uint32_t* table, int cap, uint32_t sub_flag, uint32_t empty, Value value, bool* complete, int* max_len) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    int mx = 15;
    while (mx > 0 && count[mx] == 0) --mx;
    *max_len = mx;
    for (int i = 0; i < (1 << R); ++i) table[i] = empty;
    if (mx == 0) { *complete = false; return true; }
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    *complete = left == 0;
    uint32_t next[16];
    {
        uint32_t code = 0;
        count[0] = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    }
    uint32_t codes[320];
    for (int s = 0; s < n; ++s) codes[s] = lens[s] ? next[lens[s]]++ : 0u;
    // short codes
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l == 0 || l > R) continue;
        const uint32_t e = value(s) | (uint32_t)l;
        for (uint32_t i = revbits(codes[s], l); i < (1u << R); i += 1u << l) table[i] = e;
    }
    if (mx <= R) return true;
    // long codes: one subtable per root prefix, as wide as the longest code under it
    uint8_t deep[1 << 11];
    memset(deep, 0, (size_t)1 << R);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= R) continue;
        const uint32_t p = revbits(codes[s] >> (l - R), R);
        if (l > deep[p]) deep[p] = (uint8_t)l;
    }
    int free_at = 1 << R;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= R) continue;
        const uint32_t r = revbits(codes[s], l);
        const uint32_t p = r & ((1u << R) - 1u);
        const int sb = deep[p] - R;
        if (!(table[p] & sub_flag)) {
            if (free_at + (1 << sb) > cap) return false;
            table[p] = sub_flag | ((uint32_t)free_at << 8) | (uint32_t)sb;
            for (int i = 0; i < (1 << sb); ++i) table[free_at + i] = empty;
            free_at += 1 << sb;
        }
        const uint32_t start = (table[p] >> 8) & 0xfffffu;
        const uint32_t e = value(s) | (uint32_t)(l - R);
        for (uint32_t i = r >> R; i < (1u << sb); i += 1u << (l - R)) table[start + i] = e;
    }
    return true;
}

// bits from an absolute position, with end-of-input bookkeeping (the bytes behind the input read as zero)
struct BitIn {
    const uint8_t* p;
    size_t n;
    uint64_t pos;
    uint64_t peek() const {
        const size_t b = (size_t)(pos >> 3);
        if (b + 8 <= n) return load64(p + b) >> (pos & 7);
        uint64_t v = 0;
        for (size_t i = 0; i < 8 && b + i < n; ++i) v |= (uint64_t)p[b + i] << (8 * i);
        return v >> (pos & 7);
    }
    uint32_t get(int k) {
        const uint32_t v = (uint32_t)(peek() & ((1ull << k) - 1ull));
        pos += (uint64_t)k;
        return v;
    }
    bool over() const { return pos > (uint64_t)n * 8; }
};

struct FixedTables {
    uint32_t lit[LIT_TABLE];
    uint32_t dist[DIST_TABLE];
    FixedTables() {
        uint8_t l[288];
        for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        bool c; int m;
        build_table(l, 288, LIT_ROOT, lit, LIT_TABLE, E_SUB, E_BAD, lit_value, &c, &m);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(d, 32, DIST_ROOT, dist, DIST_TABLE, D_SUB, D_BAD, dist_value, &c, &m);
    }
};
const FixedTables& fixed_tables() {
    static const FixedTables t;
    return t;
}

// the code lengths of a dynamic block header (behind the 3 header bits); b.pos is advanced
int read_code_lengths(BitIn& b, uint8_t* lens, int& hlit, int& hdist) {
    hlit = (int)b.get(5) + 257; hdist = (int)b.get(5) + 1;
    const int hclen = (int)b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return GZ_ERR_DATA;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)b.get(3);
    uint32_t clt[128];
    bool complete; int mx;
    if (!build_table(cl, 19, 7, clt, 128, 0u, 0u, [](int s) { return (uint32_t)s << 8; }, &complete, &mx)) return GZ_ERR_DATA;
    if (mx > 0 && !complete) return GZ_ERR_DATA;          // (zlib: an incomplete code-length code is an error)
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        const uint32_t e = clt[b.peek() & 127u];
        if ((e & 0xffu) == 0) return GZ_ERR_DATA;
        b.pos += e & 0xffu;
        const int sym = (int)(e >> 8);
        if (sym < 16) lens[i++] = (uint8_t)sym;
        else {
            int rep; uint8_t v = 0;
            if (sym == 16) { if (i == 0) return GZ_ERR_DATA; v = lens[i - 1]; rep = 3 + (int)b.get(2); }
            else if (sym == 17) rep = 3 + (int)b.get(3);
            else rep = 11 + (int)b.get(7);
            if (i + rep > total) return GZ_ERR_DATA;
            while (rep--) lens[i++] = v;
        }
        if (b.over()) return GZ_ERR_DATA;
    }
    if (lens[256] == 0) return GZ_ERR_DATA;
    return GZ_OK;
}

// zlib's rule for the litlen and the distance code: over-subscribed is an error, incomplete only as a single 1-bit code
bool code_acceptable(const uint8_t* lens, int n) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    int left = 1, mx = 0;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
        if (count[l]) mx = l;
    }
    return left == 0 || mx <= 1;
}

// second literals: a root entry whose literal code leaves room for another whole literal code inside the root index
// yields both (the entry for index i >> l1 is decided by its low l2 bits alone when l1 + l2 <= LIT_ROOT)
void add_literal_pairs(uint32_t* lit) {
    uint32_t single[1 << LIT_ROOT];
    memcpy(single, lit, sizeof(single));
    for (uint32_t i = 0; i < (1u << LIT_ROOT); ++i) {
        const uint32_t e = single[i];
        if (!(e & E_LIT)) continue;
        const uint32_t l1 = e & 0xffu;
        if (l1 >= (uint32_t)LIT_ROOT) continue;
        const uint32_t e2 = single[i >> l1];
        if (!(e2 & E_LIT)) continue;
        const uint32_t l2 = e2 & 0xffu;
        if (l1 + l2 > (uint32_t)LIT_ROOT) continue;
        lit[i] = E_LIT | E_PAIR | (e & 0x00ff0000u) | (((e2 >> 16) & 0xffu) << 8) | (l1 + l2);
    }
}

int build_block_tables(const uint8_t* lens, int hlit, int hdist, uint32_t* lit, uint32_t* dist) {
    bool complete; int mx;
    if (!build_table(lens, hlit, LIT_ROOT, lit, LIT_TABLE, E_SUB, E_BAD, lit_value, &complete, &mx)) return GZ_ERR_DATA;
    if (!complete && mx != 1) return GZ_ERR_DATA;
    if (!build_table(lens + hlit, hdist, DIST_ROOT, dist, DIST_TABLE, D_SUB, D_BAD, dist_value, &complete, &mx)) return GZ_ERR_DATA;
    if (!complete && mx > 1) return GZ_ERR_DATA;
    add_literal_pairs(lit);
    return GZ_OK;
}

int read_dynamic_header(BitIn& b, uint32_t* lit, uint32_t* dist) {
    uint8_t lens[320];
    int hlit, hdist;
    if (read_code_lengths(b, lens, hlit, hdist) != GZ_OK) return GZ_ERR_DATA;
    return build_block_tables(lens, hlit, hdist, lit, dist);
}

template <typename OutT> inline void fill_run(OutT* d, OutT v, size_t len);
template <> inline void fill_run<uint8_t>(uint8_t* d, uint8_t v, size_t len) {
    const uint64_t w = 0x0101010101010101ull * v;
    for (size_t i = 0; i < len; i += 8) store64(d + i, w);
}
template <> inline void fill_run<uint16_t>(uint16_t* d, uint16_t v, size_t len) {
    const uint64_t w = 0x0001000100010001ull * v;
    for (size_t i = 0; i < len; i += 4) store64(d + i, w);
}

}  // namespace

template <typename OutT>
int Inflater<OutT>::read_header() {
    BitIn b{in, in_size, bitpos};
    bfinal = b.get(1) != 0;
    const uint32_t type = b.get(2);
    if (type == 0) {
        b.pos = (b.pos + 7) & ~7ull;
        const uint32_t len = b.get(16), nlen = b.get(16);
        if (b.over() || len != (~nlen & 0xffffu)) return GZ_ERR_DATA;
        stored_left = len;
        in_block = 1;
    } else if (type == 1) {
        const FixedTables& f = fixed_tables();
        memcpy(lit, f.lit, sizeof(uint32_t) * ((1 << LIT_ROOT)));      // (fixed codes are at most 9 bits: no subtables)
        memcpy(dist, f.dist, sizeof(uint32_t) * (1 << DIST_ROOT));
        in_block = 2;
    } else if (type == 2) {
        if (read_dynamic_header(b, lit, dist) != GZ_OK) return GZ_ERR_DATA;
        in_block = 2;
    } else return GZ_ERR_DATA;
    if (b.over()) return GZ_ERR_DATA;
    bitpos = b.pos;
    return GZ_OK;
}

// ---- the fast loop's state and body --------------------------------------------------------------------------------------
// One stream position held in locals: 64-bit bit buffer (bb, bc valid bits, ip next byte to load), output cursor, the block's
// tables.  fast_step() decodes one "group" — up to three table lookups of literals (each up to two bytes), or one match — with
// no per-bit checks: the caller guarantees >= 32 input bytes and >= 320 output elements of headroom (ip <= ip_end, op < op_end).
// It is a function of its own (always inlined) because TWO streams can be stepped alternately in one loop: a DEFLATE decoder
// is one long dependency chain (bit buffer -> table load -> shift -> table load ...) that leaves most of a core's issue slots
// empty, and FASTQ streams are nearly all matches (94 % of the bytes at gzip -6: two dependent lookups per 8 bytes), so a
// second, independent chain in the same instruction window comes almost for free (decode_pair below).
constexpr int FAST_CONT = -100, FAST_BOUNDS = -101;
template <typename OutT>
struct FastStream {
    // hot: five registers per stream (two streams fit the sixteen of x86-64 with room for the temporaries)
    uint64_t bb;
    int bc;
    const uint8_t* ip;
    OutT* op;                   // output cursor
    const uint32_t* tab;        // the block's tables: litlen at tab[0 ..], distance at tab[LIT_TABLE ..] (Inflater::lit, ::dist)
    // cold: only compared against
    const uint8_t* ip_end;
    OutT* op_end;
    OutT* lo;                   // lowest readable output element (out - hist)
};

// headroom the loops keep in front of a GROUP of steps (checked once per group): a step takes at most two refills (<= 16 input
// bytes are touched beyond ip, <= 8 consumed each) and emits at most 6 literals or a 258-element match whose copies may run 31
// bytes past it.  What lies behind the headroom is left to the careful loop: sections (symbols, megabytes long) take groups of
// four; byte streams are often BGZF members of 64 KiB decoded straight into their slot of the output, where a kilobyte of
// careful decoding per member would cost more than the saved tests.
template <typename OutT> struct FastCfg;
template <> struct FastCfg<uint16_t> { static constexpr int GROUP = 4; };
template <> struct FastCfg<uint8_t> { static constexpr int GROUP = 1; };
template <typename OutT> constexpr size_t fast_in_room() { return 32 + 16 * (size_t)FastCfg<OutT>::GROUP; }
template <typename OutT> constexpr size_t fast_out_room() { return 320 * (size_t)FastCfg<OutT>::GROUP; }

// FAST_CONT: go on; GZ_OK: end-of-block code consumed; GZ_ERR_DATA.  No bounds are looked at in here (see fast_room).
template <typename OutT>
__attribute__((always_inline)) inline int fast_step(FastStream<OutT>& s) {
    constexpr size_t COPY_W = 8 / sizeof(OutT);            // elements per 8-byte move
    constexpr uint64_t LMASK = (1u << LIT_ROOT) - 1u, DMASK = (1u << DIST_ROOT) - 1u;
    const uint32_t* const L = s.tab;
    const uint32_t* const D = s.tab + LIT_TABLE;
#define AQC_REFILL() do { s.bb |= load64(s.ip) << s.bc; s.ip += (63 - s.bc) >> 3; s.bc |= 56; } while (0)
#define AQC_DROP(k) do { const int k_ = (int)(k); s.bb >>= k_; s.bc -= k_; } while (0)
    // (both bytes are always stored: without E_PAIR the second one is overwritten by the next symbol)
#define AQC_LITS() do { s.op[0] = (OutT)((e >> 16) & 0xffu); s.op[1] = (OutT)((e >> 8) & 0xffu); s.op += 1 + ((e >> 27) & 1u); AQC_DROP(e & 0xffu); } while (0)
    AQC_REFILL();
    uint32_t e = L[s.bb & LMASK];
    if (e & E_LIT) {
        AQC_LITS();
        e = L[s.bb & LMASK];
        if (e & E_LIT) {
            AQC_LITS();
            e = L[s.bb & LMASK];
            if (e & E_LIT) {
                AQC_LITS();
                return FAST_CONT;
            }
        }
        AQC_REFILL();
    }
    if (e & (E_SUB | E_BAD | E_EOB)) {           // everything that is not a plain length code: one test in the common case
        if (e & E_SUB) {
            AQC_DROP(LIT_ROOT);
            e = L[((e >> 8) & 0xfffffu) + (uint32_t)(s.bb & ((1u << (e & 0xffu)) - 1u))];
            if (e & E_LIT) { AQC_LITS(); return FAST_CONT; }
        }
        if (e & E_BAD) return GZ_ERR_DATA;       // (entries no code leads to carry E_BAD as well)
        if (e & E_EOB) { AQC_DROP(e & 0xffu); return GZ_OK; }
    }
    AQC_DROP(e & 0xffu);
    const uint32_t xb = (e >> 8) & 0x1fu;
    const size_t len = (size_t)(e >> 16) + (size_t)(s.bb & ((1u << xb) - 1u));
    AQC_DROP(xb);
    uint32_t d = D[s.bb & DMASK];
    if (d & (D_SUB | D_BAD)) {
        if (d & D_SUB) {
            AQC_DROP(DIST_ROOT);
            d = D[((d >> 8) & 0xffffu) + (uint32_t)(s.bb & ((1u << (d & 0xfu)) - 1u))];
        }
        if (d & D_BAD) return GZ_ERR_DATA;
    }
    AQC_DROP(d & 0xfu);
    const uint32_t db = (d >> 4) & 0xfu;
    const size_t dd = (size_t)((d >> 8) & 0xffffu) + (size_t)(s.bb & ((1u << db) - 1u));
    AQC_DROP(db);
    if (dd > (size_t)(s.op - s.lo)) return GZ_ERR_DATA;
    OutT* dst = s.op;
    const OutT* src = dst - dd;
    s.op += len;
    if (dd >= 16 / sizeof(OutT)) {
        // 16 bytes per move.  The first two moves are unconditional (a loop that runs once or twice by the data's whim is a
        // mispredicted branch per match, and FASTQ is nearly all matches of 4 - 20 bytes); the headroom covers the up to 31
        // bytes they may run past the match.  The second move may read what the first one wrote (dd < 32 bytes): it is behind it.
        constexpr size_t W16 = 16 / sizeof(OutT);
        memcpy(dst, src, 16);
        memcpy(dst + W16, src + W16, 16);
        if (len > 2 * W16) {
            OutT* const end = dst + len;
            dst += 2 * W16; src += 2 * W16;
            do { memcpy(dst, src, 16); dst += W16; src += W16; } while (dst < end);
        }
    } else if (dd >= COPY_W) {
        // 8 bytes at a time; the moves may run up to 7 bytes past the match (headroom), never into unread source
        OutT* const end = dst + len;
        do { store64(dst, load64((const uint8_t*)src)); dst += COPY_W; src += COPY_W; } while (dst < end);
    } else if (dd == 1) {
        fill_run<OutT>(dst, src[0], len);
    } else {
        for (size_t i = 0; i < len; ++i) dst[i] = src[i];
    }
    return FAST_CONT;
#undef AQC_REFILL
#undef AQC_DROP
#undef AQC_LITS
}

// the stream's position as a FastStream (false: too little input or output left for the fast loop)
template <typename OutT>
inline bool fast_open(const Inflater<OutT>& f, FastStream<OutT>& s) {
    static_assert(offsetof(Inflater<OutT>, dist) == offsetof(Inflater<OutT>, lit) + sizeof(uint32_t) * LIT_TABLE, "dist must follow lit");
    if (f.in_size < fast_in_room<OutT>() + 8 || f.out_cap < fast_out_room<OutT>()) return false;
    s.ip = f.in + (f.bitpos >> 3);
    s.ip_end = f.in + f.in_size - fast_in_room<OutT>();
    s.op = f.out + f.out_pos;
    s.op_end = f.out + (f.out_cap - fast_out_room<OutT>());
    s.lo = f.out - f.hist;
    s.tab = f.lit;
    if (s.ip > s.ip_end || s.op >= s.op_end) return false;
    s.bb = load64(s.ip);
    s.ip += 7;
    s.bc = 56;
    const int k = (int)(f.bitpos & 7);
    s.bb >>= k;
    s.bc -= k;
    return true;
}
template <typename OutT>
inline void fast_close(Inflater<OutT>& f, const FastStream<OutT>& s) {
    f.bitpos = (uint64_t)(s.ip - f.in) * 8 - (uint64_t)s.bc;
    f.out_pos = (size_t)(s.op - f.out);
}

template <typename OutT>
__attribute__((always_inline)) inline bool fast_room(const FastStream<OutT>& s) { return s.ip <= s.ip_end && s.op < s.op_end; }

// The loops themselves exist twice: for BMI2 (shrx / shlx / bzhi: the variable shifts and masks the body is made of, one
// micro-operation each and any register for the count — 5 - 7 % faster) and for plain x86-64; the dynamic linker picks one.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define AQC_CLONES __attribute__((target_clones("bmi2", "default")))
#else
#define AQC_CLONES
#endif
// (the state is copied into locals — registers — for the loop: through the reference every byte store of the u8 flavour could
// alias it)
#define AQC_FAST_RUN(NAME, T) \
    AQC_CLONES int NAME(FastStream<T>& ref) { \
        FastStream<T> st = ref; \
        int rc = FAST_BOUNDS; \
        while (fast_room(st)) { \
            if ((rc = fast_step(st)) != FAST_CONT) break; \
            if (FastCfg<T>::GROUP == 4) { \
                if ((rc = fast_step(st)) != FAST_CONT) break; \
                if ((rc = fast_step(st)) != FAST_CONT) break; \
                if ((rc = fast_step(st)) != FAST_CONT) break; \
            } \
            rc = FAST_BOUNDS; \
        } \
        ref = st; \
        return rc; \
    }
#define AQC_FAST_RUN2(NAME, T) \
    AQC_CLONES int NAME(FastStream<T>& ra, FastStream<T>& rb, int* rc) { \
        FastStream<T> a = ra, b = rb; \
        int which = 0, r = FAST_BOUNDS; \
        for (;;) { \
            if (!fast_room(a)) { which = 0; r = FAST_BOUNDS; break; } \
            if (!fast_room(b)) { which = 1; r = FAST_BOUNDS; break; } \
            if ((r = fast_step(a)) != FAST_CONT) { which = 0; break; } \
            if ((r = fast_step(b)) != FAST_CONT) { which = 1; break; } \
            if (FastCfg<T>::GROUP == 4) { \
                if ((r = fast_step(a)) != FAST_CONT) { which = 0; break; } \
                if ((r = fast_step(b)) != FAST_CONT) { which = 1; break; } \
                if ((r = fast_step(a)) != FAST_CONT) { which = 0; break; } \
                if ((r = fast_step(b)) != FAST_CONT) { which = 1; break; } \
                if ((r = fast_step(a)) != FAST_CONT) { which = 0; break; } \
                if ((r = fast_step(b)) != FAST_CONT) { which = 1; break; } \
            } \
        } \
        ra = a; rb = b; \
        *rc = r; \
        return which; \
    }
static_assert(FastCfg<uint16_t>::GROUP == 4 && FastCfg<uint8_t>::GROUP == 1, "the loops below are written out for groups of one and of four steps");
AQC_FAST_RUN(fast_run_u8, uint8_t)
AQC_FAST_RUN(fast_run_u16, uint16_t)
inline int fast_run(FastStream<uint8_t>& st) { return fast_run_u8(st); }
inline int fast_run(FastStream<uint16_t>& st) { return fast_run_u16(st); }
// two streams alternately until one of them stops: 0 / 1 = which, *rc = its FAST_* / GZ_* code
AQC_FAST_RUN2(fast_run2_u8, uint8_t)
AQC_FAST_RUN2(fast_run2_u16, uint16_t)
inline int fast_run2(FastStream<uint8_t>& a, FastStream<uint8_t>& b, int* rc) { return fast_run2_u8(a, b, rc); }
inline int fast_run2(FastStream<uint16_t>& a, FastStream<uint16_t>& b, int* rc) { return fast_run2_u16(a, b, rc); }

// symbols of the current Huffman block until its end-of-block code (GZ_OK), the output runs short (GZ_NEED_OUTPUT) or an error
template <typename OutT>
int Inflater<OutT>::decode_huffman() {
    OutT* const o = out;
    size_t op = out_pos;
    const uint32_t* const L = lit;
    const uint32_t* const D = dist;
    constexpr uint64_t LMASK = (1u << LIT_ROOT) - 1u, DMASK = (1u << DIST_ROOT) - 1u;
    // ---- fast loop: >= 32 input bytes and >= 320 output elements of headroom, no per-bit checks
    {
        FastStream<OutT> st;
        if (fast_open(*this, st)) {
            const int rc = fast_run(st);
            fast_close(*this, st);
            op = out_pos;
            if (rc != FAST_BOUNDS) return rc;
        }
    }
    // ---- careful loop: every symbol checked against the end of the input and of the output
    BitIn b{in, in_size, bitpos};
    for (;;) {
        const uint64_t sym_start = b.pos;
        const uint64_t w = b.peek();
        uint32_t e = L[w & LMASK];
        uint64_t used = 0;
        if (e & E_SUB) {
            used = LIT_ROOT;
            e = L[((e >> 8) & 0xfffffu) + (uint32_t)((w >> LIT_ROOT) & ((1u << (e & 0xffu)) - 1u))];
        }
        if ((e & 0xffu) == 0 || (e & E_BAD)) return GZ_ERR_DATA;
        used += e & 0xffu;
        if (e & E_LIT) {
            const size_t k = 1 + ((e >> 27) & 1u);
            if (op + k > out_cap) { bitpos = sym_start; out_pos = op; return GZ_NEED_OUTPUT; }
            b.pos += used;
            if (b.over()) return GZ_ERR_DATA;
            o[op] = (OutT)((e >> 16) & 0xffu);
            if (k == 2) o[op + 1] = (OutT)((e >> 8) & 0xffu);
            op += k;
            continue;
        }
        b.pos += used;
        if (e & E_EOB) {
            if (b.over()) return GZ_ERR_DATA;
            bitpos = b.pos; out_pos = op;
            return GZ_OK;
        }
        const uint32_t xb = (e >> 8) & 0x1fu;
        const size_t len = (size_t)(e >> 16) + (size_t)b.get((int)xb);
        const uint64_t w2 = b.peek();
        uint32_t d = D[w2 & DMASK];
        if (d & D_SUB) {
            b.pos += DIST_ROOT;
            d = D[((d >> 8) & 0xffffu) + (uint32_t)((w2 >> DIST_ROOT) & ((1u << (d & 0xfu)) - 1u))];
        }
        if ((d & 0xfu) == 0 || (d & D_BAD)) return GZ_ERR_DATA;
        b.pos += d & 0xfu;
        const size_t dd = (size_t)((d >> 8) & 0xffffu) + (size_t)b.get((int)((d >> 4) & 0xfu));
        if (b.over() || dd > op + hist) return GZ_ERR_DATA;
        if (op + len > out_cap) { bitpos = sym_start; out_pos = op; return GZ_NEED_OUTPUT; }
        OutT* dst = o + op;
        const OutT* src = dst - dd;
        for (size_t i = 0; i < len; ++i) dst[i] = src[i];
        op += len;
    }
}

template <typename OutT>
int Inflater<OutT>::run_step(uint64_t stop_bit) {
    if (in_block == 0) {
        if (final_done) return GZ_FINAL;
        if (bitpos >= stop_bit) return GZ_STOPPED;
        if (bitpos + 3 > (uint64_t)in_size * 8) return GZ_ERR_DATA;
        const int rc = read_header();
        return rc != GZ_OK ? rc : GZ_CONTINUE;
    }
    if (in_block == 1) {
        const size_t byte = (size_t)(bitpos >> 3);
        if (byte + stored_left > in_size) return GZ_ERR_DATA;
        const size_t room = out_cap - out_pos;
        const size_t k = std::min<size_t>(stored_left, room);
        for (size_t i = 0; i < k; ++i) out[out_pos + i] = (OutT)in[byte + i];
        out_pos += k;
        stored_left -= (uint32_t)k;
        bitpos += (uint64_t)k * 8;
        if (stored_left) return GZ_NEED_OUTPUT;
    } else {
        const int rc = decode_huffman();
        if (rc != GZ_OK) return rc;
    }
    end_block();
    return GZ_CONTINUE;
}

template <typename OutT>
int Inflater<OutT>::run(uint64_t stop_bit) {
    for (;;) {
        const int rc = run_step(stop_bit);
        if (rc != GZ_CONTINUE) return rc;
    }
}

template <typename OutT>
int decode_pair(Inflater<OutT>& A, Inflater<OutT>& B, int* rc) {
    FastStream<OutT> a, b;
    if (!fast_open(A, a)) { *rc = GZ_SLOW; return 0; }
    if (!fast_open(B, b)) { *rc = GZ_SLOW; return 1; }
    int r;
    const int which = fast_run2(a, b, &r);
    fast_close(A, a);
    fast_close(B, b);
    *rc = r == FAST_BOUNDS ? GZ_SLOW : r;
    return which;
}

template struct Inflater<uint8_t>;
template struct Inflater<uint16_t>;
template int decode_pair<uint8_t>(Inflater<uint8_t>&, Inflater<uint8_t>&, int*);
template int decode_pair<uint16_t>(Inflater<uint16_t>&, Inflater<uint16_t>&, int*);

size_t parse_gzip_header(const uint8_t* d, size_t n, size_t pos) {
    if (pos + 10 > n || d[pos] != 0x1f || d[pos + 1] != 0x8b || d[pos + 2] != 8) return 0;
    const uint8_t flg = d[pos + 3];
    if (flg & 0xe0) return 0;
    size_t p = pos + 10;
    if (flg & 4) {
        if (p + 2 > n) return 0;
        const size_t xlen = (size_t)d[p] | ((size_t)d[p + 1] << 8);
        p += 2 + xlen;
        if (p > n) return 0;
    }
    for (int k = 0; k < 2; ++k)
        if (flg & (k == 0 ? 8 : 16)) {
            while (p < n && d[p] != 0) ++p;
            if (p >= n) return 0;
            ++p;
        }
    if (flg & 2) p += 2;
    if (p > n) return 0;
    return p;
}

int64_t inflate_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    // (heap: the tables are ~20 KiB, callers run on pool threads with default stacks — fine either way, but keep frames small)
    std::unique_ptr<Inflater<uint8_t>> inf(new Inflater<uint8_t>());
    inf->reset(src, n, 0);
    inf->out = dst; inf->out_pos = 0; inf->out_cap = cap; inf->hist = 0;
    const int rc = inf->run(UINT64_MAX);
    if (rc != GZ_FINAL) return -1;
    return (int64_t)inf->out_pos;
}

// two independent raw-deflate streams (two BGZF members) decoded alternately, see decode_pair; got[k] = bytes written or -1
void inflate_raw2(const uint8_t* const src[2], const size_t n[2], uint8_t* const dst[2], const size_t cap[2], int64_t got[2]) {
    std::unique_ptr<Inflater<uint8_t>> inf[2];
    bool done[2] = {false, false};
    for (int k = 0; k < 2; ++k) {
        inf[k].reset(new Inflater<uint8_t>());
        inf[k]->reset(src[k], n[k], 0);
        inf[k]->out = dst[k]; inf[k]->out_pos = 0; inf[k]->out_cap = cap[k]; inf[k]->hist = 0;
        got[k] = -1;
    }
    auto settle = [&](int k, int rc) {           // a return code other than "go on" ends the stream: complete or broken
        done[k] = true;
        if (rc == GZ_FINAL) got[k] = (int64_t)inf[k]->out_pos;
    };
    while (!done[0] || !done[1]) {
        if (!done[0] && !done[1] && inf[0]->in_block == 2 && inf[1]->in_block == 2) {
            int rc;
            const int k = decode_pair(*inf[0], *inf[1], &rc);
            if (rc == GZ_OK) inf[k]->end_block();
            else if (rc == GZ_SLOW) {
                const int r2 = inf[k]->decode_huffman();
                if (r2 == GZ_OK) inf[k]->end_block();
                else settle(k, r2);
            } else settle(k, rc);
            continue;
        }
        const int k = (!done[0] && (done[1] || inf[0]->in_block != 2)) ? 0 : 1;
        const int rc = inf[k]->run_step(UINT64_MAX);
        if (rc != GZ_CONTINUE) settle(k, rc);
    }
}

// ---- block boundaries in the middle of a stream ----------------------------------------------------------------------------
namespace {

inline bool texty(uint32_t c) { return c == '\n' || (c >= 0x20 && c < 0x7f) || c == '\t' || c == '\r'; }

// does a valid block header (of any type) start at `bit`?
bool plausible_header(const uint8_t* data, size_t size, uint64_t bit, uint32_t* lit, uint32_t* dist) {
    BitIn b{data, size, bit};
    if (bit + 3 > (uint64_t)size * 8) return false;
    (void)b.get(1);
    const uint32_t type = b.get(2);
    if (type == 3) return false;
    if (type == 1) return true;
    if (type == 0) {
        b.pos = (b.pos + 7) & ~7ull;
        const uint32_t len = b.get(16), nlen = b.get(16);
        return !b.over() && len == (~nlen & 0xffffu);
    }
    return read_dynamic_header(b, lit, dist) == GZ_OK && !b.over();
}

}  // namespace

uint64_t find_block_start(const uint8_t* data, size_t size, uint64_t from_bit, uint64_t to_bit) {
    const uint64_t end_bit = std::min<uint64_t>(to_bit, size >= 16 ? (uint64_t)(size - 16) * 8 : 0);
    // (per thread, kept: the scratch output of the trial decode and its marker prefix)
    static thread_local std::unique_ptr<Inflater<uint16_t>> inf;
    static thread_local std::vector<uint16_t> scratch;
    // which of the 8 bit offsets of a byte position can be a block start at all: BFINAL = 0, BTYPE = 2 (bits 0, then 0 1 LSB
    // first -> value 0b100 = 4) and HLIT <= 29, from the 16 bits at that byte — one table lookup per BYTE of the section
    // instead of a load and three tests per BIT (the search was 7 % of a .gz run's CPU time)
    static const struct StartTab {
        uint8_t t[65536];
        StartTab() {
            for (uint32_t x = 0; x < 65536; ++x) {
                uint8_t m = 0;
                for (int k = 0; k < 8; ++k)
                    if (((x >> k) & 7u) == 4u && ((x >> (k + 3)) & 31u) <= 29u) m |= (uint8_t)(1u << k);
                t[x] = m;
            }
        }
    } ST;
    uint32_t cand = 0;              // candidate offsets left in the current byte
    uint64_t byte = from_bit >> 3;
    bool first = true;
    for (;;) {
        if (!cand) {
            if (!first) ++byte;
            // skip bytes without candidates
            for (;; ++byte) {
                if (byte * 8 >= end_bit) return UINT64_MAX;
                uint16_t x;
                memcpy(&x, data + byte, 2);
                cand = ST.t[x];
                if (first) { cand &= 0xffu << (from_bit & 7); first = false; }
                if (cand) break;
            }
        }
        const int k = __builtin_ctz(cand);
        cand &= cand - 1;
        const uint64_t bit = byte * 8 + (uint64_t)k;
        if (bit >= end_bit) return UINT64_MAX;
        const uint64_t w = load64(data + byte) >> k;
        if (((w >> 8) & 31u) > 29u) continue;          // HDIST
        // the code-length code must be complete: sum of 2^(7 - len) over its used lengths == 128
        const int hclen = (int)((w >> 13) & 15u) + 4;
        {
            // four 3-bit lengths per table lookup
            static const struct KraftTab {
                uint16_t t[4096];
                KraftTab() {
                    for (uint32_t i = 0; i < 4096; ++i) {
                        uint32_t k = 0;
                        for (int j = 0; j < 4; ++j) { const uint32_t l = (i >> (3 * j)) & 7u; if (l) k += 128u >> l; }
                        t[i] = (uint16_t)k;
                    }
                }
            } KT;
            const uint64_t p0 = bit + 17;
            const uint64_t v0 = load64(data + (p0 >> 3)) >> (p0 & 7);                 // >= 57 bits: lengths 0..18
            uint32_t kraft = 0;
            const int nbits = hclen * 3;                                            // 12 .. 57
            const uint64_t v = nbits >= 64 ? v0 : (v0 & ((1ull << nbits) - 1ull));
            kraft = KT.t[v & 4095u] + KT.t[(v >> 12) & 4095u] + KT.t[(v >> 24) & 4095u] + KT.t[(v >> 36) & 4095u] + KT.t[(v >> 48) & 4095u];
            if (kraft != 128u) continue;
        }
        // the code lengths must parse and give acceptable codes (no decode table is built before that)
        {
            BitIn b{data, size, bit + 3};
            uint8_t lens[320];
            int hlit, hdist;
            if (read_code_lengths(b, lens, hlit, hdist) != GZ_OK || b.over()) continue;
            if (!code_acceptable(lens, hlit) || !code_acceptable(lens + hlit, hdist)) continue;
        }
        // the whole block
        if (!inf) inf.reset(new Inflater<uint16_t>());
        const size_t cap = 1u << 18;
        if (scratch.empty()) {
            scratch.resize(WINDOW + cap + 512);
            for (size_t j = 0; j < WINDOW; ++j) scratch[j] = (uint16_t)(MARKER | j);      // what lies before the block: unknown
        }
        inf->reset(data, size, bit);
        inf->out = scratch.data() + WINDOW; inf->out_pos = 0; inf->out_cap = cap; inf->hist = WINDOW;
        const int rc = inf->run(bit + 1);
        if (rc == GZ_ERR_DATA) continue;
        // literals must be text (a marker is whatever the window holds: unknown).  Only what THIS block produced counts.
        bool ok = true;
        const uint16_t* o = scratch.data() + WINDOW;
        const size_t n = inf->out_pos;
        if (n == 0 && rc != GZ_NEED_OUTPUT) continue;
        for (size_t i = 0; i < n && ok; ++i) {
            const uint32_t c = o[i];
            if (c < MARKER && !texty(c)) ok = false;
        }
        if (!ok) continue;
        if (rc == GZ_STOPPED) {
            if (!plausible_header(data, size, inf->bitpos, inf->lit, inf->dist)) continue;
        } else if (rc == GZ_FINAL) {
            continue;      // (BFINAL was 0: cannot happen)
        }
        return bit;
    }
    return UINT64_MAX;
}

// ---- CRC-32 (the gzip polynomial, reflected 0xEDB88320) -----------------------------------------------------------------------
// Carry-less multiplication folding (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ"): four
// 128-bit lanes folded 64 bytes at a time, then 128 -> 64 -> 32 bits with a Barrett reduction.  zlib 1.2.11's table-driven
// crc32 (~1 GB/s) would otherwise cost as much as a third of the parallel inflate.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t crc, const uint8_t* buf, size_t len) {
    // len >= 64 and a multiple of 16; crc is the running value in its inverted (register) form
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = k1k2;
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = k3k4;
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

uint32_t crc32_fast(uint32_t crc, const uint8_t* p, size_t n) {
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && n >= 128) {
        const size_t body = n & ~(size_t)15;
        crc = ~crc32_clmul(~crc, p, body);
        p += body; n -= body;
    }
#endif
    while (n) {
        const size_t k = std::min<size_t>(n, 1u << 30);
        crc = (uint32_t)::crc32(crc, p, (uInt)k);
        p += k; n -= k;
    }
    return crc;
}
uint32_t crc32_combine_fast(uint32_t a, uint32_t b, uint64_t len2) { return (uint32_t)::crc32_combine(a, b, (z_off_t)len2); }

}  // namespace aqcgz
