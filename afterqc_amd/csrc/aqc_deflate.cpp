// aqc_deflate.cpp — DEFLATE encoding for the pipe's .gz writer (fastq.py:65-68 + `--compression`, after.py:91-92 upstream:
// Python's gzip module, i.e. zlib).  One call = one raw DEFLATE stream for one block of text (the writer makes one BGZF
// member of <= 0xff00 bytes out of it).  Written from RFC 1951.
//
//   parse    one greedy pass: 6-byte hash, one candidate per bucket, match accepted only when it is cheaper than the
//            literals it replaces — estimated from the block's own byte histogram, so the same input always gives the same
//            bytes.  FASTQ is mostly 4-symbol text at ~2.2 bits per literal: the short far matches zlib's fast levels take
//            there cost more bits than they save (and most of the time).
//   codes    dynamic Huffman codes (two-queue construction on the sorted frequencies, lengths limited to 15 / 7 by a Kraft
//            repair), canonical, sent with the usual run-length header.
//   levels   <= 0 stored; 1 .. 3 one probe per position; >= 4 also tries the previous occupant of the bucket (2-way).
#include <algorithm>
#include <cstring>

#include "aqc_gz.hpp"

namespace aqcgz {

namespace {

inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline void store64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }

inline uint32_t rev16(uint32_t x) {
    x = ((x & 0x5555u) << 1) | ((x >> 1) & 0x5555u);
    x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
    x = ((x & 0x0f0fu) << 4) | ((x >> 4) & 0x0f0fu);
    return ((x & 0x00ffu) << 8) | (x >> 8);
}

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct StaticTables {
    uint8_t len_sym[259];      // match length -> length symbol - 257
    uint8_t dist_sym_lo[257];  // distance 1..256 -> distance symbol
    uint8_t dist_sym_hi[256];  // (distance - 1) >> 7 for distances > 256
    StaticTables() {
        for (int s = 0; s < 29; ++s)
            for (int l = LEN_BASE[s]; l <= (s == 28 ? 258 : LEN_BASE[s] + (1 << LEN_EXTRA[s]) - 1) && l <= 258; ++l) len_sym[l] = (uint8_t)s;
        len_sym[258] = 28;
        for (int s = 0; s < 30; ++s)
            for (int d = DIST_BASE[s]; d < DIST_BASE[s] + (1 << DIST_EXTRA[s]); ++d) {
                if (d <= 256) dist_sym_lo[d] = (uint8_t)s;
                else dist_sym_hi[(d - 1) >> 7] = (uint8_t)s;
            }
    }
};
const StaticTables& tabs() {
    static const StaticTables t;
    return t;
}
inline int dist_symbol(uint32_t d) {
    const StaticTables& t = tabs();
    return d <= 256 ? t.dist_sym_lo[d] : t.dist_sym_hi[(d - 1) >> 7];
}

// Huffman code lengths for freq[0, n), each <= max_bits; symbols with zero frequency get length 0
void code_lengths(const uint32_t* freq, int n, int max_bits, uint8_t* lens) {
    struct Node { uint32_t f; int16_t sym; };
    Node leaf[288];
    int m = 0;
    for (int s = 0; s < n; ++s) {
        lens[s] = 0;
        if (freq[s]) leaf[m++] = Node{freq[s], (int16_t)s};
    }
    if (m == 0) return;
    if (m == 1) { lens[leaf[0].sym] = 1; return; }
    std::sort(leaf, leaf + m, [](const Node& a, const Node& b) { return a.f != b.f ? a.f < b.f : a.sym < b.sym; });
    // two-queue construction: internal nodes come out in non-decreasing weight
    uint32_t w[576];
    int16_t parent[576];
    for (int i = 0; i < m; ++i) w[i] = leaf[i].f;
    int li = 0, ni = m, nn = m;         // next leaf, next internal node, nodes so far
    auto take = [&]() -> int {
        if (li < m && (ni >= nn || w[li] <= w[ni])) return li++;
        return ni++;
    };
    while ((m - li) + (nn - ni) > 1) {
        const int a = take(), b = take();
        w[nn] = w[a] + w[b];
        parent[a] = parent[b] = (int16_t)nn;
        nn++;
    }
    // depths from the root down (a parent's index is above its children's).  Length limit, the classic repair: depths are
    // clamped on the way down — internal nodes too, every clamped node counts — then for every two clamped nodes one leaf of
    // the deepest level above the limit moves a step down and an overflowing leaf becomes its sibling; the code stays
    // complete.  Lengths are then dealt out again by frequency (leaf 0 is the rarest: it gets the longest code).
    uint8_t depth[576];
    depth[nn - 1] = 0;
    int bl[32] = {0};
    int overflow = 0;
    for (int i = nn - 2; i >= 0; --i) {
        int d = depth[parent[i]] + 1;
        if (d > max_bits) { d = max_bits; overflow++; }
        depth[i] = (uint8_t)d;
        if (i < m) bl[d]++;
    }
    if (overflow > 0) {
        do {
            int bits = max_bits - 1;
            while (bl[bits] == 0) --bits;
            bl[bits]--;
            bl[bits + 1] += 2;
            bl[max_bits]--;
            overflow -= 2;
        } while (overflow > 0);
        int idx = 0;
        for (int bits = max_bits; bits >= 1; --bits)
            for (int k = 0; k < bl[bits]; ++k) depth[idx++] = (uint8_t)bits;
    }
    for (int i = 0; i < m; ++i) lens[leaf[i].sym] = depth[i];
}

// canonical codes for the lengths, bit-reversed (DEFLATE sends Huffman codes most significant bit first)
void make_codes(const uint8_t* lens, int n, uint16_t* codes) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    count[0] = 0;
    uint32_t next[16], code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    for (int s = 0; s < n; ++s) codes[s] = lens[s] ? (uint16_t)(rev16(next[lens[s]]++) >> (16 - lens[s])) : (uint16_t)0;
}

struct BitOut {
    uint8_t* p;
    uint64_t bb = 0;
    int bc = 0;
    inline void put(uint64_t v, int n) { bb |= v << bc; bc += n; }
    inline void flush() { store64(p, bb); p += bc >> 3; bb >>= bc & ~7; bc &= 7; }      // keeps < 8 bits
    uint8_t* finish() { flush(); if (bc) { *p++ = (uint8_t)bb; bb = 0; bc = 0; } return p; }
};

struct Seq { uint32_t lits; uint16_t len, dist; };      // `lits` literals, then (if len) one match

constexpr int HASH_BITS = 13;      // 16 KiB of u16 positions: the table stays in the L1 cache next to the block

inline uint32_t hash6(uint64_t v) { return (uint32_t)((v * 0x9E3779B185EBCA87ull) >> (64 - HASH_BITS)); }

// byte histogram of the block (four interleaved tables: no store-to-load stalls on runs) and from it the average literal
// cost in 1/8 bits
void byte_histogram(const uint8_t* src, size_t n, uint32_t* h /* [256] */) {
    uint32_t t[4][256];
    memset(t, 0, sizeof(t));
    size_t i = 0;
    for (; i + 4 <= n; i += 4) { t[0][src[i]]++; t[1][src[i + 1]]++; t[2][src[i + 2]]++; t[3][src[i + 3]]++; }
    for (; i < n; ++i) t[0][src[i]]++;
    for (int s = 0; s < 256; ++s) h[s] = t[0][s] + t[1][s] + t[2][s] + t[3][s];
}

int literal_cost8(const uint32_t* h, size_t cnt) {
    if (cnt == 0) return 64;
    uint64_t bits8 = 0;
    for (int s = 0; s < 256; ++s) {
        if (!h[s]) continue;
        // log2(cnt / h) in 1/8 bits: integer part from the leading zeros, fraction from the three bits below the leading one
        const uint64_t q = ((uint64_t)cnt << 16) / h[s];         // 16.16 fixed, >= 1.0
        const int ip = 63 - __builtin_clzll(q);                   // >= 16
        const uint64_t frac = ((q << (63 - ip)) >> 60) & 7u;
        bits8 += (uint64_t)h[s] * ((uint64_t)(ip - 16) * 8 + frac);
    }
    const int c = (int)(bits8 / cnt);
    return c < 8 ? 8 : c;
}

}  // namespace

size_t deflate_bound(size_t n) { return n + (n / 65535 + 1) * 5 + 16 + 320; }

static size_t stored_stream(const uint8_t* src, size_t n, uint8_t* dst) {
    uint8_t* p = dst;
    size_t i = 0;
    do {
        const size_t k = std::min<size_t>(n - i, 65535);
        *p++ = (i + k == n) ? 1 : 0;
        *p++ = (uint8_t)k; *p++ = (uint8_t)(k >> 8); *p++ = (uint8_t)~k; *p++ = (uint8_t)(~k >> 8);
        if (k) memcpy(p, src + i, k);
        p += k; i += k;
    } while (i < n);
    return (size_t)(p - dst);
}

size_t deflate_block(const uint8_t* src, size_t n, int level, uint8_t* dst) {
    if (n == 0) { dst[0] = 0x03; dst[1] = 0x00; return 2; }       // final fixed-Huffman block holding only end-of-block
    if (level <= 0 || n < 16) return stored_stream(src, n, dst);
    static thread_local uint16_t tl_head[1 << HASH_BITS], tl_prev[1 << HASH_BITS];
    static thread_local std::vector<Seq> tl_seqs;
    uint16_t* const head = tl_head;
    uint16_t* const prev_occ = tl_prev;
    std::vector<Seq>& seqs = tl_seqs;
    const bool two_way = level >= 4;
    seqs.clear();
    uint32_t freq[286];
    byte_histogram(src, n, freq);                  // every byte; what the matches cover is taken out again below
    uint32_t lenf[29] = {0}, distf[30] = {0};
    const StaticTables& T = tabs();
    const int lit8 = literal_cost8(freq, n);
    // cheap text (few symbols, ~2 bits each): only matches of 8+ bytes can pay, so the first 8 bytes must agree; otherwise 6
    const uint64_t need_mask = lit8 < 28 ? ~0ull : 0x0000ffffffffffffull;
    const size_t min_len = lit8 < 28 ? 8 : 6;
    size_t lit_run_start = 0;
    // u16 table positions cover 65535 bytes: longer inputs are parsed in windows with the table cleared between them
    for (size_t w0 = 0; w0 < n; w0 += 65535 - 258) {
        const size_t w1 = std::min(n, w0 + 65535 - 258);
        memset(head, 0, sizeof(tl_head));
        if (two_way) memset(prev_occ, 0, sizeof(tl_prev));
        const uint8_t* const base = src + w0;
        const size_t wn = w1 - w0;
        size_t i = 0;
        const size_t hash_end = wn >= 8 ? wn - 8 : 0;      // positions with 8 readable bytes inside the window
        uint32_t miss = 0;                                 // probes since the last match: the stride grows with it
        while (i < hash_end) {
            const uint64_t v = load64(base + i);
            const uint32_t h = hash6(v & need_mask);
            const uint32_t cand = head[h];
            const uint32_t cand2 = two_way ? prev_occ[h] : 0u;
            if (two_way) prev_occ[h] = (uint16_t)cand;
            head[h] = (uint16_t)(i + 1);
            size_t best_len = 0, best_dist = 0;
            for (int probe = 0; probe < (two_way ? 2 : 1); ++probe) {
                const uint32_t c = probe ? cand2 : cand;
                if (!c) continue;
                const size_t cp = c - 1;
                const size_t d = i - cp;
                if (d > 32768) continue;
                const uint64_t x = load64(base + cp) ^ v;
                if (x & need_mask) continue;                 // the first 6 / 8 bytes must agree
                size_t len;
                if (x) len = (size_t)(__builtin_ctzll(x) >> 3);
                else {
                    len = 8;
                    const size_t maxl = std::min<size_t>(258, wn - i);
                    while (len + 8 <= maxl) {
                        const uint64_t y = load64(base + cp + len) ^ load64(base + i + len);
                        if (y) { len += (size_t)(__builtin_ctzll(y) >> 3); break; }
                        len += 8;
                    }
                    if (len + 8 > maxl) { while (len < maxl && base[cp + len] == base[i + len]) ++len; }
                    if (len > maxl) len = maxl;
                }
                if (len > best_len) { best_len = len; best_dist = d; }
            }
            if (best_len >= min_len) {
                // worth it?  the literals it replaces against ~13 bits + the distance's extra bits
                const int ds = dist_symbol((uint32_t)best_dist);
                if ((int)best_len * lit8 > (13 + DIST_EXTRA[ds]) * 8) {
                    seqs.push_back(Seq{(uint32_t)(w0 + i - lit_run_start), (uint16_t)best_len, (uint16_t)(best_dist - 1)});
                    lenf[T.len_sym[best_len]]++;
                    distf[ds]++;
                    for (size_t k = 0; k < best_len; ++k) freq[base[i + k]]--;       // not literals after all
                    // a few positions inside the match keep the table warm
                    const size_t e = i + best_len;
                    for (size_t k = i + best_len / 2; k + 1 < e && k < hash_end; k += 4) head[hash6(load64(base + k) & need_mask)] = (uint16_t)(k + 1);
                    i = e;
                    lit_run_start = w0 + i;
                    if (best_len >= 16) miss = 0;          // (short matches come and go inside literal-heavy lines)
                    continue;
                }
            }
            // no match here: after a while of that (sequence / quality lines) only every 2nd .. 4th position is probed
            // — but never across a line end: what repeats in line-structured text (read names) starts right behind one
            const uint32_t step = 1 + std::min<uint32_t>(miss >> 5, 3u);
            if (step > 1) {
                uint64_t z = v ^ 0x0a0a0a0a0a0a0a0aull;
                z = (z - 0x0101010101010101ull) & ~z & 0x8080808080808080ull & ((1ull << (8 * step)) - 1ull);
                if (z) { i += (size_t)(__builtin_ctzll(z) >> 3) + 1; miss = 0; continue; }
            }
            i += step;
            ++miss;
        }
    }
    seqs.push_back(Seq{(uint32_t)(n - lit_run_start), 0, 0});
    // ---- codes
    freq[256] = 1;
    for (int s = 0; s < 29; ++s) freq[257 + s] = lenf[s];
    uint32_t dfreq[30];
    int nd = 0;
    for (int s = 0; s < 30; ++s) { dfreq[s] = distf[s]; nd += distf[s] ? 1 : 0; }
    if (nd == 1) { dfreq[distf[0] ? 1 : 0] = 1; }      // (a lone distance code would be an incomplete code: give it a sibling)
    uint8_t ll[286], dl[30];
    code_lengths(freq, 286, 15, ll);
    code_lengths(dfreq, 30, 15, dl);
    uint16_t lc[286], dc[30];
    make_codes(ll, 286, lc);
    make_codes(dl, 30, dc);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && ll[hlit - 1] == 0) --hlit;
    while (hdist > 1 && dl[hdist - 1] == 0) --hdist;
    // run-length coded code lengths
    uint8_t all[316];
    memcpy(all, ll, (size_t)hlit);
    memcpy(all + hlit, dl, (size_t)hdist);
    const int total = hlit + hdist;
    uint8_t rsym[316], rext[316];
    int nr = 0;
    uint32_t clf[19] = {0};
    for (int i = 0; i < total;) {
        const uint8_t v = all[i];
        int run = 1;
        while (i + run < total && all[i + run] == v) ++run;
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int k = std::min(left, 138); rsym[nr] = 18; rext[nr++] = (uint8_t)(k - 11); clf[18]++; left -= k; }
            if (left >= 3) { rsym[nr] = 17; rext[nr++] = (uint8_t)(left - 3); clf[17]++; left = 0; }
            while (left--) { rsym[nr] = 0; rext[nr++] = 0; clf[0]++; }
        } else {
            rsym[nr] = v; rext[nr++] = 0; clf[v]++; left--;
            while (left >= 3) { const int k = std::min(left, 6); rsym[nr] = 16; rext[nr++] = (uint8_t)(k - 3); clf[16]++; left -= k; }
            while (left-- > 0) { rsym[nr] = v; rext[nr++] = 0; clf[v]++; }
        }
        i += run;
    }
    uint8_t cll[19];
    uint16_t clc[19];
    code_lengths(clf, 19, 7, cll);
    {
        // a code-length code with a single symbol must still be complete for zlib: give it a sibling
        int used = 0, first = -1;
        for (int s = 0; s < 19; ++s) if (cll[s]) { used++; if (first < 0) first = s; }
        if (used == 1) cll[first == 0 ? 1 : 0] = 1;
    }
    make_codes(cll, 19, clc);
    {
        // belt and braces: all three codes must be complete (zlib rejects anything else); if a construction bug ever gave
        // something different, store the block rather than write a stream nobody can read
        auto complete = [](const uint8_t* l, int cnt, int maxb) {
            uint32_t k = 0; int used = 0;
            for (int s = 0; s < cnt; ++s) if (l[s]) { k += (1u << maxb) >> l[s]; used++; }
            return used == 0 || k == (1u << maxb);
        };
        if (!complete(ll, 286, 15) || !complete(dl, 30, 15) || !complete(cll, 19, 7)) return stored_stream(src, n, dst);
    }
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cll[order[hclen - 1]] == 0) --hclen;
    // ---- size estimate: would stored blocks be smaller?
    {
        uint64_t bits = 17 + 3 * (uint64_t)hclen;
        for (int k = 0; k < nr; ++k) bits += cll[rsym[k]] + (rsym[k] == 16 ? 2 : rsym[k] == 17 ? 3 : rsym[k] == 18 ? 7 : 0);
        for (int s = 0; s < 286; ++s) bits += (uint64_t)freq[s] * ll[s];
        for (int s = 0; s < 29; ++s) bits += (uint64_t)lenf[s] * LEN_EXTRA[s];
        for (int s = 0; s < 30; ++s) bits += (uint64_t)distf[s] * (dl[s] + DIST_EXTRA[s]);
        if ((bits + 7) / 8 >= n + (n / 65535 + 1) * 5) return stored_stream(src, n, dst);
    }
    // ---- emit
    BitOut o{dst};
    o.put(1, 1); o.put(2, 2);
    o.put((uint64_t)(hlit - 257), 5); o.put((uint64_t)(hdist - 1), 5); o.put((uint64_t)(hclen - 4), 4);
    o.flush();
    for (int k = 0; k < hclen; ++k) { o.put(cll[order[k]], 3); if ((k & 7) == 7) o.flush(); }
    o.flush();
    for (int k = 0; k < nr; ++k) {
        o.put(clc[rsym[k]], cll[rsym[k]]);
        if (rsym[k] >= 16) o.put(rext[k], rsym[k] == 16 ? 2 : rsym[k] == 17 ? 3 : 7);
        o.flush();
    }
    // literal code + length in one word each
    uint32_t lcode[256];
    for (int s = 0; s < 256; ++s) lcode[s] = (uint32_t)lc[s] | ((uint32_t)ll[s] << 16);
    const uint8_t* p = src;
    for (const Seq& q : seqs) {
        uint32_t k = q.lits;
        while (k >= 3) {
            const uint32_t a = lcode[p[0]], b = lcode[p[1]], c = lcode[p[2]];
            o.put(a & 0xffffu, (int)(a >> 16)); o.put(b & 0xffffu, (int)(b >> 16)); o.put(c & 0xffffu, (int)(c >> 16));
            o.flush();
            p += 3; k -= 3;
        }
        while (k--) { const uint32_t a = lcode[*p++]; o.put(a & 0xffffu, (int)(a >> 16)); }
        o.flush();
        if (q.len) {
            const int ls = T.len_sym[q.len];
            o.put(lc[257 + ls], ll[257 + ls]);
            o.put((uint64_t)(q.len - LEN_BASE[ls]), LEN_EXTRA[ls]);
            const uint32_t d = (uint32_t)q.dist + 1;
            const int ds = dist_symbol(d);
            o.put(dc[ds], dl[ds]);
            o.put((uint64_t)(d - DIST_BASE[ds]), DIST_EXTRA[ds]);
            o.flush();
            p += q.len;
        }
    }
    o.put(lc[256], ll[256]);
    return (size_t)(o.finish() - dst);
}

// ---- a code for the DEVICE encoder (aqc_gzdev.hpp): one literal/length + distance code in which EVERY symbol has a code (it
// is shared by all members of a stream, whatever bytes they hold), built from sampled symbol counts with the routines above,
// plus the ready-made bits of the block header that announces it (BFINAL = 1, BTYPE = 2, HLIT, HDIST, HCLEN, code lengths).
bool build_codebook(const uint32_t* lit_freq, const uint32_t* dist_freq, GzCodebook* cb) {
    uint32_t freq[286], dfreq[30];
    for (int s = 0; s < 286; ++s) freq[s] = lit_freq[s] + 1u;
    for (int s = 0; s < 30; ++s) dfreq[s] = dist_freq[s] + 1u;
    uint8_t ll[286], dl[30];
    code_lengths(freq, 286, 15, ll);
    code_lengths(dfreq, 30, 15, dl);
    uint16_t lc[286], dc[30];
    make_codes(ll, 286, lc);
    make_codes(dl, 30, dc);
    const int hlit = 286, hdist = 30;
    uint8_t all[316];
    memcpy(all, ll, 286);
    memcpy(all + 286, dl, 30);
    const int total = hlit + hdist;
    uint8_t rsym[316], rext[316];
    int nr = 0;
    uint32_t clf[19] = {0};
    for (int i = 0; i < total;) {
        const uint8_t v = all[i];
        int run = 1;
        while (i + run < total && all[i + run] == v) ++run;
        int left = run;
        rsym[nr] = v; rext[nr++] = 0; clf[v]++; left--;            // (no zero lengths here: every symbol is coded)
        while (left >= 3) { const int k = std::min(left, 6); rsym[nr] = 16; rext[nr++] = (uint8_t)(k - 3); clf[16]++; left -= k; }
        while (left-- > 0) { rsym[nr] = v; rext[nr++] = 0; clf[v]++; }
        i += run;
    }
    uint8_t cll[19];
    uint16_t clc[19];
    code_lengths(clf, 19, 7, cll);
    {
        int used = 0, first = -1;
        for (int s = 0; s < 19; ++s) if (cll[s]) { used++; if (first < 0) first = s; }
        if (used == 1) cll[first == 0 ? 1 : 0] = 1;
    }
    make_codes(cll, 19, clc);
    auto complete = [](const uint8_t* l, int cnt, int maxb) {
        uint32_t k = 0;
        for (int s = 0; s < cnt; ++s) if (l[s]) k += (1u << maxb) >> l[s];
        return k == (1u << maxb);
    };
    if (!complete(ll, 286, 15) || !complete(dl, 30, 15) || !complete(cll, 19, 7)) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cll[order[hclen - 1]] == 0) --hclen;
    uint8_t buf[sizeof(cb->hdr) + 16];
    memset(buf, 0, sizeof(buf));
    BitOut o{buf};
    uint64_t nbits = 0;
    auto put = [&](uint64_t v, int n) { o.put(v, n); o.flush(); nbits += (uint64_t)n; };
    put(1, 1); put(2, 2);
    put((uint64_t)(hlit - 257), 5); put((uint64_t)(hdist - 1), 5); put((uint64_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) put(cll[order[k]], 3);
    for (int k = 0; k < nr; ++k) {
        put(clc[rsym[k]], cll[rsym[k]]);
        if (rsym[k] == 16) put(rext[k], 2);
    }
    (void)o.finish();
    if (nbits > 8 * sizeof(cb->hdr)) return false;
    memset(cb, 0, sizeof(*cb));
    memcpy(cb->hdr, buf, sizeof(cb->hdr));
    cb->hdr_bits = (uint32_t)nbits;
    for (int s = 0; s < 286; ++s) cb->lit[s] = (uint32_t)lc[s] | ((uint32_t)ll[s] << 16);
    for (int s = 0; s < 30; ++s) cb->dist[s] = (uint32_t)dc[s] | ((uint32_t)dl[s] << 16);
    return true;
}

}  // namespace aqcgz
