// gzb_selftest.cpp — the device gunzip's per-lane logic (afterqc_amd/csrc/aqc_gunzip_dev.hpp), run on the CPU.
//
// The header's GZB_HD functions are what the kernels' lanes execute: the block-start tests, the table builder, the block
// decoder, the chain walk, the marker re-basing.  CpuOffload below deals them out with plain loops in the kernels' order
// (scan -> compact -> decode -> chain -> gather, same buffers, same GzbJob) and plugs into the real ParallelGunzip through
// the SectionOffload interface the GPU uses (aqc_capi.hip: DeviceInflate).  So this checks, without a GPU:
//   * every zlib stream below comes out byte-identical (ParallelGunzip also verifies CRC-32 / ISIZE itself);
//   * the offloaded sections really are committed (the chain rule accepts them) for dynamic-Huffman streams;
//   * stored blocks inside a chain, concatenated members, fixed-Huffman and stored-only streams, false section starts,
//     tiny symbol space (overflow -> host) and tiny candidate space all degrade to host decoding, never to wrong bytes.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <memory>
#include <vector>

#include "../../afterqc_amd/csrc/aqc_gunzip_dev.hpp"
#include "../../afterqc_amd/csrc/aqc_gz.hpp"

using namespace aqc;

namespace {

struct CpuOffload : aqcgz::SectionOffload {
    size_t group;
    uint32_t ratio_cap = 20, tok_ratio = 8, overlap_tokens = 2048;
    uint32_t cand_div = 4096;            // candidate capacity = span / cand_div + 256
    uint32_t slice_tokens = 300, max_slices = 1u << 20;
    uint64_t groups = 0, sections = 0, found = 0, candidates = 0, false_ends = 0, spec_lanes = 0, failed_blocks = 0, stitched_blocks = 0;
    std::vector<std::vector<uint16_t>*> live;
    std::vector<aqcgz::OffloadResult> pending_results;
    // break_after >= 0: the "device" fails on its (break_after + 1)-th group — the group's sections come back empty, as
    // DeviceInflate hands them back after a HIP error — and takes no work from then on (ready() == false)
    long break_after = -1;
    bool broken = false;
    // resident mode (round 6): the symbols stay "on the device" — a GroupRes — until the consumer asks for the run to be resolved
    // (gzb_window_byte / gzb_resolve_sym / gzb_crc_slot + gzb_crc_join, dealt out the way gzb_windows_kernel, gzb_resolve_kernel
    // and gzb_crc_kernel deal them) and then fetches bytes
    bool resident = false;
    uint64_t runs_resolved = 0, sections_resolved = 0;
    long resolve_breaks_after = -1;       // >= 0: resolve() fails (as after a HIP error) from its (n + 1)-th call on
    struct GroupRes {
        std::vector<uint16_t> sym;
        std::vector<uint8_t> text;
        std::vector<uint64_t> off;
        std::vector<uint32_t> nsym;
    };
    struct ResTok { std::shared_ptr<GroupRes> g; int k; };
    explicit CpuOffload(size_t g) : group(g) {}
    size_t group_bytes() const override { return group; }
    bool ready() override { return !broken; }
    bool gave_up() override { return broken; }
    void release(void* token) override {
        if (resident) delete (ResTok*)token;
        else delete (std::vector<uint16_t>*)token;
    }
    const uint8_t* text_ptr(void* token, int* device) override {
        ResTok* t = (ResTok*)token;
        if (device) *device = 0;
        return t->g->text.data() + t->g->off[t->k];
    }
    int resolve(void* const* tokens, int n, const uint8_t* win, size_t wlen, uint32_t* crc, uint8_t* tail, size_t* tail_len, uint32_t* piece_nl) override {
        if (resolve_breaks_after >= 0 && (long)runs_resolved >= resolve_breaks_after) return -2;
        ++runs_resolved;
        sections_resolved += (uint64_t)n;
        GroupRes& G = *((ResTok*)tokens[0])->g;
        for (int k = 0; k < n; ++k) if (((ResTok*)tokens[k])->g.get() != &G) return -2;
        std::vector<uint64_t> off((size_t)n);
        std::vector<uint32_t> nsym((size_t)n);
        for (int k = 0; k < n; ++k) { const int sk = ((ResTok*)tokens[k])->k; off[k] = G.off[sk]; nsym[k] = G.nsym[sk]; }
        std::vector<uint8_t> wins((size_t)(n + 1) * GZB_WINDOW, 0);
        if (wlen) memcpy(wins.data() + GZB_WINDOW - wlen, win, wlen);
        uint32_t bad = 0;
        GzbResolveJob R{};
        R.sym = G.sym.data(); R.text = G.text.data(); R.wins = wins.data(); R.off = off.data(); R.nsym = nsym.data(); R.n_run = (uint32_t)n;
        R.valid0 = (uint32_t)(GZB_WINDOW - wlen); R.bad = &bad;
        // gzb_windows_kernel
        uint32_t valid = R.valid0;
        std::vector<uint32_t> valids((size_t)n);
        for (int k = 0; k < n; ++k) {
            valids[k] = valid;
            const uint8_t* w = wins.data() + (size_t)k * GZB_WINDOW;
            uint8_t* o = wins.data() + (size_t)(k + 1) * GZB_WINDOW;
            for (uint32_t j = 0; j < GZB_WINDOW; ++j) o[j] = gzb_window_byte(R, (uint32_t)k, j, w, valid, &bad);
            valid = nsym[k] >= valid ? 0u : valid - nsym[k];
        }
        // gzb_resolve_kernel
        for (int k = 0; k < n; ++k) {
            const uint32_t v = gzb_window_valid(R, (uint32_t)k);
            if (v != valids[k]) { printf("gzb_window_valid disagrees with the window pass\n"); return -2; }
            const uint8_t* w = wins.data() + (size_t)k * GZB_WINDOW;
            for (uint32_t i = 0; i < nsym[k]; ++i) G.text[off[k] + i] = gzb_resolve_sym(G.sym[off[k] + i], w, v, &bad);
        }
        if (bad) return aqcgz::GZ_ERR_DATA;
        // gzb_crc_kernel + the host's fold
        static std::vector<uint32_t> tab, adv_piece;
        auto advance = [](uint32_t x, uint64_t len) { return (uint32_t)crc32_combine((uLong)x, 0UL, (z_off_t)len); };
        if (tab.empty()) {
            tab.resize(GZB_CRC_TAB_WORDS);
            gzb_crc_tables(tab.data(), advance);
            adv_piece.resize(32);
            for (int j = 0; j < 32; ++j) adv_piece[j] = advance(1u << j, GZB_CRC_PIECE);
        }
        for (int k = 0; k < n; ++k) {
            const uint32_t cnt = (nsym[k] + GZB_CRC_PIECE - 1u) / GZB_CRC_PIECE;
            std::vector<uint32_t> pieces(cnt);
            for (uint32_t pi = 0; pi < cnt; ++pi) {
                uint32_t part[GZB_CRC_THREADS], lines = 0;
                for (uint32_t t = 0; t < GZB_CRC_THREADS; ++t) { uint32_t lf = 0; part[t] = gzb_crc_slot(G.text.data() + off[k], nsym[k], cnt, pi, t, tab.data(), &lf); lines += lf; }
                if (piece_nl) *piece_nl++ = lines;
                for (int level = 0; level < 8; ++level) {
                    const uint32_t step = 1u << level;
                    for (uint32_t t = 0; t < GZB_CRC_THREADS; t += 2u * step) part[t] = gzb_crc_join(part[t], part[t + step], level, tab.data());
                }
                pieces[pi] = part[0];
            }
            crc[k] = gzb_crc_fold(pieces.data(), cnt, nsym[k], adv_piece.data(), advance);
            if (crc[k] != (uint32_t)crc32(0L, G.text.data() + off[k], nsym[k])) { printf("device CRC of a section differs from zlib's\n"); return -2; }
        }
        uint64_t total = 0;
        for (int k = 0; k < n; ++k) total += nsym[k];
        const size_t tl = (size_t)std::min<uint64_t>(GZB_WINDOW, wlen + total);
        memcpy(tail, wins.data() + (size_t)n * GZB_WINDOW + GZB_WINDOW - tl, tl);
        *tail_len = tl;
        return 0;
    }
    std::vector<std::pair<std::pair<const uint8_t*, uint8_t*>, size_t>> queued;      // fetch(): copies land at fetch_wait(), as a stream's would
    bool fetch(void* token, size_t off, size_t len, uint8_t* dst) override {
        ResTok* t = (ResTok*)token;
        if (off + len > t->g->nsym[t->k]) return false;
        queued.push_back({{t->g->text.data() + t->g->off[t->k] + off, dst}, len});
        return true;
    }
    bool fetch_wait() override {
        for (auto& q : queued) memcpy(q.first.second, q.first.first, q.second);
        queued.clear();
        return true;
    }

    bool submit(const uint8_t* data, size_t size, int n, const uint64_t* nominal, const uint64_t* stop, const uint8_t* exact,
                std::function<void(int, const aqcgz::OffloadResult&)> done) override {
        if (broken) return false;
        if (break_after >= 0 && (long)groups >= break_after) {
            broken = true;
            ++groups;
            aqcgz::OffloadResult none;
            std::thread([n, done, none] { for (int k = 0; k < n; ++k) done(k, none); }).detach();
            return true;
        }
        const uint64_t SLACK = 256u << 10;
        const uint64_t byte0 = (nominal[0] >> 3) & ~(uint64_t)15;
        const uint64_t end_byte = std::min<uint64_t>(size, (stop[n - 1] >> 3) + 1 + SLACK);
        const size_t span = (size_t)(end_byte - byte0);
        std::vector<uint8_t> comp(span + 256, 0);
        memcpy(comp.data(), data + byte0, span);
        GzbJob J{};
        J.comp = comp.data(); J.comp_bytes = (uint32_t)span; J.scan_byte0 = 0;
        J.first_bit = (uint32_t)(nominal[0] - byte0 * 8);
        J.last_bit = (uint32_t)std::min<uint64_t>(stop[n - 1] - byte0 * 8, (uint64_t)span * 8);
        J.n_tiles = (uint32_t)(((size_t)(J.last_bit >> 3) + 1 + GZB_SCAN_TILE - 1) / GZB_SCAN_TILE);
        J.cand_cap = (uint32_t)(span / cand_div + 256);
        J.ratio_cap = ratio_cap;
        std::vector<uint32_t> tile_cnt(J.n_tiles), tile_cand((size_t)J.n_tiles * GZB_TILE_CAND), n_cand(2), c_start(J.cand_cap), c_end(J.cand_cap), c_nsym(J.cand_cap),
            c_flags(J.cand_cap), c_symcap(J.cand_cap);
        std::vector<uint64_t> c_symoff(J.cand_cap);
        J.tile_cnt = tile_cnt.data(); J.tile_cand = tile_cand.data(); J.n_cand = n_cand.data(); J.c_start = c_start.data(); J.c_end = c_end.data();
        J.c_nsym = c_nsym.data(); J.c_flags = c_flags.data(); J.c_symcap = c_symcap.data(); J.c_symoff = c_symoff.data();
        J.blk_sym_cap = gzb_sym_budget(span, ratio_cap);
        std::unique_ptr<uint16_t[]> blk_sym(new uint16_t[J.blk_sym_cap + 64]);
        J.blk_sym = blk_sym.get();
        J.tok_ratio = tok_ratio;
        J.overlap_tokens = overlap_tokens;
        J.blk_tp_cap = gzb_tok_budget(span, tok_ratio, overlap_tokens);
        std::unique_ptr<unsigned long long[]> blk_tp(new unsigned long long[J.blk_tp_cap + 64]);
        std::vector<uint64_t> c_tokoff(J.cand_cap);
        std::vector<uint32_t> c_tokcap(J.cand_cap);
        J.c_tokoff = c_tokoff.data(); J.c_tokcap = c_tokcap.data();
        std::vector<uint32_t> c_lanes(J.cand_cap), l_u32((size_t)5 * J.cand_cap * GZB_K);
        J.blk_tp = blk_tp.get(); J.c_lanes = c_lanes.data();
        J.l_p = l_u32.data(); J.l_stop = J.l_p + (size_t)J.cand_cap * GZB_K; J.l_start = J.l_stop + (size_t)J.cand_cap * GZB_K;
        J.l_ntok = J.l_start + (size_t)J.cand_cap * GZB_K; J.l_flags = J.l_ntok + (size_t)J.cand_cap * GZB_K;
        std::vector<uint32_t> tables((size_t)J.cand_cap * GZB_TAB_WORDS);
        J.tables = tables.data();
        // ---- scan: every lane of every tile
        uint8_t kraft[512];
        for (int i = 0; i < 512; ++i) kraft[i] = (uint8_t)gzb_kraft9((uint32_t)i);
        std::vector<uint8_t> cl(128);
        const uint32_t limit_bit = J.comp_bytes * 8u;
        for (uint32_t tile = 0; tile < J.n_tiles; ++tile) {
            std::vector<uint32_t> hits;
            for (uint32_t tid = 0; tid < (uint32_t)GZB_SCAN_THREADS; ++tid) {
                const uint32_t b0 = tile * (uint32_t)GZB_SCAN_TILE + tid * 16u;
                if (b0 >= J.comp_bytes) continue;
                uint32_t d[8];
                memcpy(d, comp.data() + b0, 32);
                for (int i = 0; i < 4; ++i) {
                    const unsigned long long v64 = ((unsigned long long)d[i + 1] << 32) | d[i];
                    uint32_t mm = gzb_quick32(v64);
                    const uint32_t bb = (b0 + 4u * (uint32_t)i) * 8u;
                    if (bb + 32u <= J.first_bit || bb >= J.last_bit) mm = 0;
                    else {
                        if (bb < J.first_bit) mm &= ~0u << (J.first_bit - bb);
                        if (bb + 32u > J.last_bit) mm &= (1u << (J.last_bit - bb)) - 1u;
                    }
                    while (mm) {
                        const uint32_t bit = (uint32_t)__builtin_ctz(mm);
                        mm &= mm - 1;
                        const uint32_t p = bb + bit;
                        const uint32_t hclen = ((uint32_t)(v64 >> (bit + 13u)) & 15u) + 4u;
                        if (!gzb_kraft_ok(comp.data(), p, hclen, kraft)) continue;
                        uint32_t db, hl, hd;
                        if (gzb_header(comp.data(), limit_bit, p, cl.data(), 1, nullptr, db, hl, hd)) hits.push_back(p);
                    }
                }
            }
            std::sort(hits.begin(), hits.end());
            const uint32_t c = (uint32_t)std::min<size_t>(hits.size(), (size_t)GZB_TILE_CAND);
            tile_cnt[tile] = c;
            for (uint32_t a = 0; a < c; ++a) tile_cand[(size_t)tile * GZB_TILE_CAND + a] = hits[a];
        }
        // ---- compact
        uint32_t nc = 0;
        for (uint32_t t = 0; t < J.n_tiles; ++t)
            for (uint32_t a = 0; a < tile_cnt[t]; ++a)
                if (nc < J.cand_cap) c_start[nc++] = tile_cand[(size_t)t * GZB_TILE_CAND + a];
        n_cand[0] = nc; n_cand[1] = 0;
        candidates += nc;
        {
            unsigned long long o = 0, to = 0;
            for (uint32_t c = 0; c < nc; ++c) {
                const uint32_t cap = gzb_symcap_of(J, c, nc), tcap = gzb_tokcap_of(J, c, nc);
                c_symoff[c] = o;
                c_tokoff[c] = to;
                c_tokcap[c] = tcap;
                c_symcap[c] = (o + cap > J.blk_sym_cap || to + tcap > J.blk_tp_cap) ? 0u : cap;
                o += cap;
                to += tcap;
            }
        }
        // ---- decode: tables per candidate, then GZB_K lanes per candidate in slices, then the stitch
        std::vector<uint32_t> cnt(16), nxt(16), off(16);
        for (uint32_t c = 0; c < nc; ++c) {
            uint32_t* const tw = J.tables + (size_t)c * GZB_TAB_WORDS;
            const GzbLaneTab<1> T{reinterpret_cast<uint16_t*>(tw)};
            uint8_t* const lens = reinterpret_cast<uint8_t*>(tw + GZB_TAB_ENTRIES / 2);
            uint32_t p = 0, hlit = 0, hdist = 0, fl = 0;
            if (c_symcap[c] == 0) fl = GZB_F_SKIP;
            else if (!gzb_header(J.comp, limit_bit, c_start[c], cl.data(), 1, lens, p, hlit, hdist)) fl = GZB_F_ERROR;
            if (!fl) {
                gzb_build<true>(lens, hlit, T, cnt.data(), nxt.data(), off.data(), 1);
                gzb_build<false>(lens + hlit, hdist, T, cnt.data(), nxt.data(), off.data(), 1);
                gzb_plan_lanes(J, c, nc, p);
            } else {
                c_lanes[c] = 0;
                for (uint32_t k = 0; k < (uint32_t)GZB_K; ++k) J.l_flags[c * GZB_K + k] = 0;
            }
            c_flags[c] = fl; c_nsym[c] = 0; c_end[c] = 0;
        }
        // in slices, like the kernels: a slice ends after slice_tokens tokens, the next one resumes at the saved bit / token count
        for (uint32_t sl = 0; sl < max_slices; ++sl)
            for (uint32_t i = 0; i < nc * (uint32_t)GZB_K; ++i) {
                const uint32_t c = i / (uint32_t)GZB_K, k = i % (uint32_t)GZB_K;
                if (J.l_flags[i] != GZB_F_MORE) continue;
                const uint32_t lanes = c_lanes[c];
                if (lanes > 1) ++spec_lanes;
                const GzbLaneTab<1> T{reinterpret_cast<uint16_t*>(J.tables + (size_t)c * GZB_TAB_WORDS)};
                const uint32_t share = c_tokcap[c] / (uint32_t)GZB_K;
                const size_t at = c_tokoff[c] + (size_t)k * share;
                uint32_t p = J.l_p[i], nt = J.l_ntok[i];
                GzbInMem in{J.comp, 0};
                J.l_flags[i] = gzb_tokenize(in, limit_bit, T, J.blk_tp + at, lanes == 1u ? share * (uint32_t)GZB_K : share, p, nt, J.l_stop[i], slice_tokens, lanes != 1u);
                J.l_p[i] = p; J.l_ntok[i] = nt;
            }
        for (uint32_t c = 0; c < nc; ++c) {
            if (c_flags[c]) continue;
            uint32_t ns = 0, eb = 0;
            const uint32_t fl = gzb_stitch_expand(J, c, ns, eb);
            c_flags[c] = fl; c_nsym[c] = fl ? 0u : ns; c_end[c] = eb;
            if (fl) ++failed_blocks;
            if (!fl && c_lanes[c] > 1) ++stitched_blocks;
        }
        // ---- chain + gather
        uint64_t sec_max = 0;
        for (int k = 0; k < n; ++k) sec_max = std::max<uint64_t>(sec_max, (stop[k] - nominal[k]) >> 3);
        J.s_symcap = (uint32_t)((sec_max * 12 + (2u << 20) + 7) & ~(uint64_t)7);
        J.n_sec = (uint32_t)n;
        std::vector<uint32_t> s_nom(n), s_stop(n), s_exact(n), s_start(n), s_end(n), s_nsym(n), s_nblk(n), s_blocks((size_t)n * GZB_SEC_BLOCKS * 3);
        for (int k = 0; k < n; ++k) {
            s_nom[k] = (uint32_t)(nominal[k] - byte0 * 8);
            s_stop[k] = (uint32_t)std::min<uint64_t>(stop[k] - byte0 * 8, (uint64_t)span * 8);
            s_exact[k] = exact[k];
        }
        J.s_nominal = s_nom.data(); J.s_stop = s_stop.data(); J.s_exact = s_exact.data(); J.s_start = s_start.data(); J.s_end = s_end.data();
        J.s_nsym = s_nsym.data(); J.s_nblk = s_nblk.data(); J.s_blocks = s_blocks.data();
        std::vector<uint64_t> s_off((size_t)n + 1);
        J.s_off = s_off.data();
        J.s_sym_total = (uint64_t)span * 12 + (uint64_t)n * 64 + (1u << 20);
        std::vector<uint16_t> s_sym(J.s_sym_total + 64);
        J.s_sym = s_sym.data();
        ++groups;
        sections += (uint64_t)n;
        for (int k = 0; k < n; ++k) gzb_chain_section(J, (uint32_t)k);
        gzb_place(J);
        std::shared_ptr<GroupRes> res;
        if (resident) {
            res.reset(new GroupRes());
            res->off.assign(s_off.begin(), s_off.begin() + n);
            res->nsym.assign(s_nsym.begin(), s_nsym.end());
        }
        for (int k = 0; k < n; ++k) {
            aqcgz::OffloadResult r;
            if (s_start[k] != GZB_NONE && s_nsym[k] != 0) {
                uint16_t* const dst = s_sym.data() + s_off[k];
                const uint32_t* const blocks = s_blocks.data() + (size_t)k * GZB_SEC_BLOCKS * 3u;
                for (uint32_t b = 0; b < s_nblk[k]; ++b) {
                    const uint32_t w0 = blocks[3u * b], w1 = blocks[3u * b + 1], o = blocks[3u * b + 2];
                    if (w0 & GZB_STORED) { for (uint32_t i = 0; i < (w0 & 0xffffu); ++i) dst[o + i] = comp[w1 + i]; }
                    else {
                        const uint16_t* const s = J.blk_sym + c_symoff[w0];
                        for (uint32_t i = 0; i < c_nsym[w0]; ++i) dst[o + i] = gzb_rebase(s[i], o, dst);
                    }
                }
                r.found = true;
                r.start_bit = byte0 * 8 + s_start[k];
                r.end_bit = byte0 * 8 + s_end[k];
                r.n_sym = s_nsym[k];
                if (!resident) {
                    auto* keep = new std::vector<uint16_t>(dst, dst + s_nsym[k]);
                    r.sym = keep->data();
                    r.token = keep;
                }
                ++found;
            }
            if (!resident) done(k, r);
            else if (r.found) { r.resident = true; r.token = new ResTok{res, k}; }
            if (resident) pending_results.push_back(r);
        }
        if (resident) {
            // (the symbols of all sections are final only now: a later section's gather does not touch an earlier one's, but the
            //  buffer is handed over whole)
            res->sym.assign(s_sym.begin(), s_sym.begin() + (long)s_off[n] + 64);
            res->text.assign((size_t)s_off[n] + 64, 0);
            for (int k = 0; k < n; ++k) done(k, pending_results[(size_t)k]);
            pending_results.clear();
        }
        return true;
    }
};

std::vector<uint8_t> fastq_like(size_t n_reads, unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<uint8_t> t;
    const char B[] = "ACGT", Q[] = "#/6<AEEEEEEEEE";
    for (size_t r = 0; r < n_reads; ++r) {
        char name[96];
        const int nl = snprintf(name, sizeof(name), "@SIM:1:FC1:%u:%u:%u:%u 1:N:0:ACGT\n", (unsigned)(1 + rng() % 4), (unsigned)(1101 + rng() % 1200), (unsigned)(1000 + rng() % 24000), (unsigned)(1000 + rng() % 19000));
        t.insert(t.end(), name, name + nl);
        for (int i = 0; i < 150; ++i) t.push_back((uint8_t)(rng() % 500 == 0 ? 'N' : B[rng() & 3]));
        t.push_back('\n'); t.push_back('+'); t.push_back('\n');
        uint8_t q = 'E';
        for (int i = 0; i < 150; ++i) { if (rng() % 7 == 0) q = (uint8_t)Q[rng() % 14]; t.push_back(q); }
        t.push_back('\n');
    }
    return t;
}

std::vector<uint8_t> gz_of(const std::vector<uint8_t>& text, int level, int strategy, size_t flush_every = 0, int flush = Z_FULL_FLUSH) {
    z_stream zs{};
    deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, (uLong)text.size()) + (flush_every ? text.size() / flush_every * 16 + 64 : 0) + 64);
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    size_t pos = 0;
    while (pos < text.size()) {
        const size_t k = flush_every ? std::min(flush_every, text.size() - pos) : text.size() - pos;
        zs.next_in = (Bytef*)text.data() + pos; zs.avail_in = (uInt)k;
        deflate(&zs, pos + k == text.size() ? Z_FINISH : flush);
        pos += k;
    }
    if (text.empty()) deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

int failures = 0;
uint32_t g_max_slices = 1u << 20, g_slice_tokens = 300;      // (a case may cut the decoder off: unfinished blocks then go to the host)

// decode gz through ParallelGunzip with the CPU emulation of the device as its offloader; returns the offloader's counters
long g_break_after = -1, g_resolve_breaks_after = -1;
bool g_resident = false;
int g_case_no = 0;
uint64_t g_segment_bytes = 0;
uint32_t g_tok_ratio = 8;
uint64_t g_sections_resolved = 0;
bool run_case(const char* what, const std::vector<uint8_t>& gz, const std::vector<uint8_t>& text, size_t section, size_t group, int threads, bool want_device,
              uint32_t ratio_cap = 20, uint32_t cand_div = 4096, bool expect_fail = false, bool hybrid = false) {
    CpuOffload off(group);
    ++g_case_no;
    off.break_after = g_break_after;
    off.resident = g_resident;
    off.resolve_breaks_after = g_resolve_breaks_after;
    off.tok_ratio = g_tok_ratio;
    off.overlap_tokens = g_tok_ratio <= 2 ? GZB_OVERLAP_TOKENS : 2048u;       // (the product's lean budget in file mode)
    off.ratio_cap = ratio_cap; off.cand_div = cand_div; off.max_slices = g_max_slices; off.slice_tokens = g_slice_tokens;
    aqc_host::Pool pool(threads);
    std::vector<uint8_t> out(text.size() + 65536);
    size_t produced = 0;
    bool failed = false;
    uint64_t dev_acc = 0, host_acc = 0, bridged = 0, res_bytes = 0, off_bytes = 0;
    {
        aqcgz::ParallelGunzip pg(gz.data(), gz.size(), threads ? &pool : nullptr, 4, section, &off, !hybrid);
        for (;;) {
            // resident mode, every other case: the text of resolved sections is LISTED, not copied (what the pipe's reader asks for);
            // the "device memory" of the emulation is host memory, so the check copies it into place itself — and counts the line
            // feeds of every whole piece against what resolve() reported
            std::vector<aqcgz::DevSegment> segs;
            const bool want_segs = g_resident && (g_case_no & 1);
            const size_t got = pg.read(out.data() + produced, std::min<size_t>(out.size() - produced, 777777), want_segs ? &segs : nullptr);
            if (pg.failed()) { failed = true; break; }
            if (!got) break;
            size_t last_end = 0;
            for (const aqcgz::DevSegment& g : segs) {
                if (g.dst_off < last_end || g.dst_off + g.len > got || g.sec_off + g.len > g.sec_len || !g.keep) { printf("bad device segment\n"); failed = true; break; }
                last_end = g.dst_off + g.len;
                memcpy(out.data() + produced + g.dst_off, g.dev, g.len);
                g_segment_bytes += g.len;
                const size_t pieces = (g.sec_len + aqcgz::NL_PIECE - 1) / aqcgz::NL_PIECE;
                for (size_t j = 0; j < pieces; ++j) {
                    const size_t hi = g.sec_len - (pieces - 1 - j) * aqcgz::NL_PIECE, lo = hi > aqcgz::NL_PIECE ? hi - aqcgz::NL_PIECE : 0;
                    if (lo < g.sec_off || hi > g.sec_off + g.len) continue;
                    const uint8_t* p = g.dev - g.sec_off + lo;
                    uint32_t c = 0;
                    for (size_t i = 0; i < hi - lo; ++i) c += p[i] == '\n';
                    if (c != g.piece_nl[j]) { printf("line feeds of piece %zu: %u counted, %u reported\n", j, c, g.piece_nl[j]); failed = true; }
                }
            }
            if (failed) break;
            produced += got;
        }
        dev_acc = pg.offloaded_accepted; host_acc = pg.sections_accepted - pg.offloaded_accepted; bridged = pg.bridged_bytes;
        res_bytes = pg.resident_bytes; off_bytes = pg.offloaded_bytes;
    }
    g_sections_resolved += off.sections_resolved;
    bool ok;
    if (expect_fail) ok = failed;
    else {
        ok = !failed && produced == text.size() && (text.empty() || memcmp(out.data(), text.data(), text.size()) == 0);
        if (ok && want_device && dev_acc == 0) ok = false;
        if (ok && hybrid && want_device && host_acc == 0) ok = false;      // the pool and the device both supplied sections
        if (ok && g_resident && res_bytes != off_bytes) ok = false;        // resident mode: no device section went through the host's translation
    }
    printf("%-46s %s  gz %8zu -> %9zu B  sec %7zu  device sections %3llu  host %3llu  bridged %9llu B  candidates %llu  groups %llu  stitched %llu  failed %llu\n", what, ok ? "ok  " : "FAIL", gz.size(),
           text.size(), section, (unsigned long long)dev_acc, (unsigned long long)host_acc, (unsigned long long)bridged, (unsigned long long)off.candidates, (unsigned long long)off.groups, (unsigned long long)off.stitched_blocks,
           (unsigned long long)off.failed_blocks);
    if (!ok) ++failures;
    return ok;
}

}  // namespace

// gzb_selftest FILE.gz TEXT [section [group [ratio_cap [tok_ratio]]]]: one single-member file through the emulation with the kernels' own slice budget
static std::vector<uint8_t> slurp(const char* path) {
    std::vector<uint8_t> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    {
        // the computed length / distance bases against RFC 1951's tables
        static const uint16_t LB[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t LX[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t DB[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t DX[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (uint32_t i = 0; i < 29; ++i)
            if (gzb_len_base(i) != LB[i] || gzb_len_extra(i) != LX[i]) { printf("length symbol %u: base / extra wrong\n", i); return 1; }
        for (uint32_t i = 0; i < 30; ++i)
            if (gzb_dist_base(i) != DB[i] || gzb_dist_extra(i) != DX[i]) { printf("distance symbol %u: base / extra wrong\n", i); return 1; }
    }
    if (argc > 2) {
        const std::vector<uint8_t> gz = slurp(argv[1]), text = slurp(argv[2]);
        g_max_slices = 6; g_slice_tokens = 2048;          // (DeviceInflate::run_group's defaults: slices, 1 token entry per compressed byte, resident results)
        g_tok_ratio = argc > 6 ? (uint32_t)atoi(argv[6]) : 1;
        g_resident = true;
        const bool ok = run_case(argv[1], gz, text, argc > 3 ? (size_t)atol(argv[3]) : 65536, argc > 4 ? (size_t)atol(argv[4]) : 1 << 20, 4, true, argc > 5 ? (uint32_t)atoi(argv[5]) : 12);
        return ok ? 0 : 1;
    }
    // `gzb_selftest 0` / `gzb_selftest 1`: one of the two modes only (the test runs them as two processes side by side)
    const int only_mode = argc == 2 ? atoi(argv[1]) : -1;
    for (int mode = 0; mode < 2; ++mode) {
    if (only_mode >= 0 && mode != only_mode) continue;
    // every case twice: the symbols come back to the host (mode 0, rounds 4 - 5), or stay with the decoder, which resolves their
    // markers and CRCs when the consumer arrives (mode 1, round 6: what DeviceInflate does)
    g_resident = mode == 1;
    printf("---- %s ----\n", g_resident ? "resident results: markers and CRC-32 resolved by the decoder" : "symbols handed back to the host");
    const std::vector<uint8_t> fq = fastq_like(14000, 3);        // ~4.9 MB
    // dynamic-Huffman streams of every level: the device must supply sections
    for (int level : {1, 2, 4, 6, 9}) {
        const std::vector<uint8_t> gz = gz_of(fq, level, Z_DEFAULT_STRATEGY);
        char what[80];
        snprintf(what, sizeof(what), "fastq level %d, 64 KiB sections", level);
        run_case(what, gz, fq, 64 << 10, 512 << 10, 3, true);
    }
    {
        const std::vector<uint8_t> gz = gz_of(fq, 6, Z_DEFAULT_STRATEGY);
        run_case("fastq level 6, 256 KiB sections, one group", gz, fq, 256 << 10, 64 << 20, 2, true);
        run_case("fastq level 6, no host threads", gz, fq, 128 << 10, 1 << 20, 0, true);
        run_case("fastq level 6, symbol space 2x (overflow)", gz, fq, 64 << 10, 512 << 10, 2, false, 2);
        run_case("fastq level 6, 8 candidates per MiB", gz, fq, 64 << 10, 512 << 10, 2, false, 20, 1u << 30);
        g_max_slices = 3;
        run_case("fastq level 6, decoder cut off after 3 slices", gz, fq, 64 << 10, 512 << 10, 2, false);
        g_max_slices = 1u << 20;
        std::vector<uint8_t> bad = gz;
        bad[bad.size() / 2] ^= 0x21;
        run_case("fastq level 6, one byte damaged", bad, fq, 64 << 10, 512 << 10, 2, false, 20, 4096, true);
        std::vector<uint8_t> cut(gz.begin(), gz.begin() + (long)(gz.size() * 2 / 3));
        run_case("fastq level 6, truncated", cut, fq, 64 << 10, 512 << 10, 2, false, 20, 4096, true);
    }
    run_case("fastq huffman-only", gz_of(fq, 6, Z_HUFFMAN_ONLY), fq, 64 << 10, 512 << 10, 2, true);
    run_case("fastq run-length", gz_of(fq, 6, Z_RLE), fq, 64 << 10, 512 << 10, 2, true);
    run_case("fastq fixed codes (host only)", gz_of(fq, 6, Z_FIXED), fq, 64 << 10, 512 << 10, 2, false);
    run_case("fastq stored (host only)", gz_of(fq, 0, Z_DEFAULT_STRATEGY), fq, 64 << 10, 512 << 10, 2, false);
    // pigz-style: an empty stored block between the deflate blocks every 128 KiB of text; and sync flushes
    run_case("fastq level 6, full flush every 128 KiB", gz_of(fq, 6, Z_DEFAULT_STRATEGY, 128 << 10, Z_FULL_FLUSH), fq, 64 << 10, 512 << 10, 2, true);
    run_case("fastq level 1, sync flush every 40 KB", gz_of(fq, 1, Z_DEFAULT_STRATEGY, 40000, Z_SYNC_FLUSH), fq, 64 << 10, 512 << 10, 2, true);
    {
        // concatenated members of different levels + zero padding behind the last one
        std::vector<uint8_t> cat, text;
        int lv = 1;
        for (size_t a = 0; a < fq.size(); a += fq.size() / 3 + 1) {
            std::vector<uint8_t> part(fq.begin() + (long)a, fq.begin() + (long)std::min(fq.size(), a + fq.size() / 3 + 1));
            const std::vector<uint8_t> g = gz_of(part, lv, Z_DEFAULT_STRATEGY);
            cat.insert(cat.end(), g.begin(), g.end());
            text.insert(text.end(), part.begin(), part.end());
            lv += 4;
        }
        cat.insert(cat.end(), 100, 0);
        run_case("three members (levels 1, 5, 9) + zero padding", cat, text, 64 << 10, 512 << 10, 2, true);
    }
    {
        // data in which block headers are easy to mistake: random bytes (stored by zlib), long runs, a tiny alphabet
        std::mt19937 rng(9);
        std::vector<uint8_t> r(1500000);
        for (auto& b : r) b = (uint8_t)rng();
        run_case("random bytes level 6", gz_of(r, 6, Z_DEFAULT_STRATEGY), r, 64 << 10, 512 << 10, 2, false);
        std::vector<uint8_t> runs;
        while (runs.size() < 3000000) { const int k = 1 + (int)(rng() % 900); runs.insert(runs.end(), (size_t)k, (uint8_t)('a' + rng() % 3)); }
        run_case("long runs level 6 (ratio > 20: overflow)", gz_of(runs, 6, Z_DEFAULT_STRATEGY), runs, 64 << 10, 512 << 10, 2, false);
        run_case("long runs level 6, symbol space 400x", gz_of(runs, 6, Z_DEFAULT_STRATEGY), runs, 64 << 10, 512 << 10, 2, false, 400);
        std::vector<uint8_t> two(2000000);
        for (auto& b : two) b = (uint8_t)("AC"[rng() & 1]);
        run_case("two-symbol text level 9", gz_of(two, 9, Z_DEFAULT_STRATEGY), two, 64 << 10, 512 << 10, 2, true);
        std::vector<uint8_t> one(1, 'x'), none;
        run_case("one byte", gz_of(one, 6, Z_DEFAULT_STRATEGY), one, 64 << 10, 512 << 10, 2, false);
        run_case("empty", gz_of(none, 6, Z_DEFAULT_STRATEGY), none, 64 << 10, 512 << 10, 2, false);
    }
    {
        const std::vector<uint8_t> big = fastq_like(60000, 11);   // ~21 MB: several groups of 2 MiB, distances near 32 KiB do occur at level 9
        run_case("fastq 21 MB level 9, 256 KiB sections", gz_of(big, 9, Z_DEFAULT_STRATEGY), big, 256 << 10, 2 << 20, 4, true);
        run_case("fastq 21 MB level 1, 1 MiB sections", gz_of(big, 1, Z_DEFAULT_STRATEGY), big, 1 << 20, 4 << 20, 4, true);
    }
    {
        // HYBRID: the pool takes the window's sections from the bottom up, the device groups from the top down (what the pipe runs)
        const std::vector<uint8_t> big = fastq_like(60000, 12);
        const std::vector<uint8_t> g6 = gz_of(big, 6, Z_DEFAULT_STRATEGY), g1 = gz_of(big, 1, Z_DEFAULT_STRATEGY);
        run_case("hybrid: 21 MB level 6, 64 KiB sections, groups of 4", g6, big, 64 << 10, 256 << 10, 3, true, 20, 4096, false, true);
        run_case("hybrid: 21 MB level 1, 128 KiB sections, groups of 8", g1, big, 128 << 10, 1 << 20, 2, true, 20, 4096, false, true);
        run_case("hybrid: 21 MB level 6, no pool threads", g6, big, 64 << 10, 512 << 10, 0, true, 20, 4096, false, true);
        run_case("hybrid: full flush every 128 KiB", gz_of(big, 6, Z_DEFAULT_STRATEGY, 128 << 10, Z_FULL_FLUSH), big, 64 << 10, 256 << 10, 3, true, 20, 4096, false, true);
        std::vector<uint8_t> bad = g6;
        bad[bad.size() / 3] ^= 0x44;
        run_case("hybrid: one byte damaged", bad, big, 64 << 10, 256 << 10, 3, false, 20, 4096, true, true);
        const std::vector<uint8_t> small = fastq_like(300, 13);
        run_case("hybrid: 100 KB file", gz_of(small, 6, Z_DEFAULT_STRATEGY), small, 64 << 10, 256 << 10, 2, false, 20, 4096, false, true);
        // the device gives up (its first / its third group fails, it takes no more work): the host decodes on, in both modes — the
        // device-only mode used to wait forever for a section nobody would make (round-4 advisory)
        g_break_after = 0;
        run_case("device breaks at group 1 (device only)", g6, big, 64 << 10, 256 << 10, 3, false);
        run_case("device breaks at group 1 (device only, no pool)", g6, big, 64 << 10, 256 << 10, 0, false);
        run_case("device breaks at group 1 (hybrid)", g6, big, 64 << 10, 256 << 10, 3, false, 20, 4096, false, true);
        g_break_after = 2;
        run_case("device breaks at group 3 (device only)", g6, big, 64 << 10, 256 << 10, 3, true);
        run_case("device breaks at group 3 (hybrid)", g1, big, 128 << 10, 1 << 20, 2, false, 20, 4096, false, true);
        g_break_after = -1;
        if (g_resident) {
            // the decoder fails when asked to resolve (a HIP error at that point): the run's sections are dropped, the host decodes the stretch
            g_resolve_breaks_after = 1;
            run_case("resolve fails from the 2nd run on (device only)", g6, big, 64 << 10, 256 << 10, 3, true);
            run_case("resolve fails from the 2nd run on (hybrid)", g6, big, 64 << 10, 256 << 10, 3, true, 20, 4096, false, true);
            g_resolve_breaks_after = -1;
            // a member that starts less than 32 KiB before a device section: window entries that do not exist
            std::vector<uint8_t> cat, text;
            const std::vector<uint8_t> head = fastq_like(40, 21);
            const std::vector<uint8_t> g0 = gz_of(head, 6, Z_DEFAULT_STRATEGY);
            cat.insert(cat.end(), g0.begin(), g0.end()); text.insert(text.end(), head.begin(), head.end());
            cat.insert(cat.end(), g6.begin(), g6.end()); text.insert(text.end(), big.begin(), big.end());
            run_case("short member, then a long one", cat, text, 64 << 10, 256 << 10, 3, true);
        }
    }
    }
    // (only the resident mode resolves sections and hands over segments)
    if (only_mode != 0 && g_sections_resolved == 0) { printf("no section was ever resolved by the emulated device\n"); ++failures; }
    if (only_mode != 0 && g_segment_bytes == 0) { printf("no text was ever handed over as a device segment\n"); ++failures; }
    printf("%.1f MB of text handed over as device segments\n", 1e-6 * (double)g_segment_bytes);
    if (failures) { printf("%d FAILED\n", failures); return 1; }
    printf("all device-gunzip logic checks passed\n");
    return 0;
}
