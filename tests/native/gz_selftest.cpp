// gz_selftest.cpp — the pipe's own gzip codec against zlib (test infrastructure; built and run by tests/test_gz_codec.py).
//   g++ -O2 -std=c++17 -pthread tests/native/gz_selftest.cpp afterqc_amd/csrc/aqc_{inflate,gunzip,deflate}.cpp -lz -o gz_selftest
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "../../afterqc_amd/csrc/aqc_gz.hpp"

using namespace aqcgz;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); failures++; } } while (0)

static std::vector<uint8_t> fastq_like(size_t n_rec, unsigned seed, int L = 150) {
    std::mt19937 rng(seed);
    std::vector<uint8_t> t;
    const char B[4] = {'A', 'C', 'G', 'T'};
    const char Q[6] = {'E', 'A', '/', '<', '6', '#'};
    char name[128];
    for (size_t r = 0; r < n_rec; ++r) {
        const int k = snprintf(name, sizeof(name), "@SIM:1:FC1:%d:%d:%d:%d 1:N:0:ACGT\n", 1 + (int)(rng() % 4), 1101 + (int)(rng() % 1200), 1000 + (int)(rng() % 24000), 1000 + (int)(rng() % 19000));
        t.insert(t.end(), name, name + k);
        for (int i = 0; i < L; ++i) t.push_back((rng() % 500) ? B[rng() & 3] : 'N');
        t.push_back('\n'); t.push_back('+'); t.push_back('\n');
        for (int i = 0; i < L; ++i) { const unsigned x = rng() % 100; t.push_back(Q[x < 64 ? 0 : x < 82 ? 1 : x < 91 ? 2 : x < 97 ? 3 : x < 99 ? 4 : 5]); }
        t.push_back('\n');
    }
    return t;
}

static std::vector<uint8_t> zlib_deflate(const std::vector<uint8_t>& src, int level, int wbits, int strategy = Z_DEFAULT_STRATEGY, int memlevel = 8) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, wbits, memlevel, strategy);
    std::vector<uint8_t> out(deflateBound(&z, (uLong)src.size()) + 64);
    z.next_in = const_cast<uint8_t*>(src.data()); z.avail_in = (uInt)src.size();
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

static bool zlib_inflate_raw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
    z_stream z{};
    inflateInit2(&z, -15);
    out.assign(expect + 16, 0);
    z.next_in = const_cast<uint8_t*>(src); z.avail_in = (uInt)n;
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    const int rc = inflate(&z, Z_FINISH);
    const size_t got = z.total_out;
    inflateEnd(&z);
    out.resize(got);
    return rc == Z_STREAM_END;
}

int main(int argc, char** argv) {
    const bool bench = argc > 1 && !strcmp(argv[1], "bench");
    std::vector<std::pair<std::string, std::vector<uint8_t>>> sets;
    sets.push_back({"fastq", fastq_like(bench ? 60000 : 6000, 1)});
    {
        std::mt19937 rng(5);
        std::vector<uint8_t> r(300000);
        for (auto& b : r) b = (uint8_t)rng();
        sets.push_back({"random", r});
        std::vector<uint8_t> z(400000, 0);
        sets.push_back({"zeros", z});
        std::vector<uint8_t> t;
        const char* words[] = {"the ", "quick ", "brown ", "fox ", "jumps ", "over ", "lazy ", "dog\n", "GATTACA", "EEEEEEEEEEEEEEEE"};
        while (t.size() < 500000) { const char* w = words[rng() % 10]; t.insert(t.end(), w, w + strlen(w)); }
        sets.push_back({"words", t});
        std::vector<uint8_t> s1(1, 'x'), s0;
        sets.push_back({"one", s1});
        sets.push_back({"empty", s0});
        std::vector<uint8_t> runs;
        while (runs.size() < 200000) { const int k = 1 + (int)(rng() % 600); runs.insert(runs.end(), (size_t)k, (uint8_t)('a' + rng() % 3)); }
        sets.push_back({"runs", runs});
        // Fibonacci-like symbol frequencies: the unlimited Huffman code would be deeper than 15 bits (the length limit and its
        // repair are exercised, for the code-length code's 7-bit limit as well)
        std::vector<uint8_t> fib;
        {
            uint64_t a = 1, b = 1;
            for (int sym = 0; sym < 24; ++sym) {
                fib.insert(fib.end(), (size_t)std::min<uint64_t>(a, 40000), (uint8_t)(33 + sym));
                const uint64_t c = a + b; a = b; b = c;
            }
            std::shuffle(fib.begin(), fib.end(), rng);
        }
        sets.push_back({"fibonacci", fib});
    }
    // ---- A. inflate_raw vs zlib-made streams
    for (auto& ds : sets) {
        for (int level = 0; level <= 9; ++level) {
            for (int strat : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED}) {
                if (strat != Z_DEFAULT_STRATEGY && level != 6) continue;
                for (int ml : {8, 1}) {
                    if (ml == 1 && level != 2) continue;
                    const auto comp = zlib_deflate(ds.second, level, -15, strat, ml);
                    std::vector<uint8_t> out(ds.second.size() + 1);
                    const int64_t got = inflate_raw(comp.data(), comp.size(), out.data(), ds.second.size());
                    CHECK(got == (int64_t)ds.second.size() && !memcmp(out.data(), ds.second.data(), ds.second.size()), "inflate_raw %s level %d strat %d ml %d: got %lld of %zu",
                          ds.first.c_str(), level, strat, ml, (long long)got, ds.second.size());
                    // truncated stream must fail (not crash)
                    if (comp.size() > 8) {
                        const int64_t g2 = inflate_raw(comp.data(), comp.size() / 2, out.data(), ds.second.size());
                        CHECK(g2 < 0, "truncated %s level %d accepted", ds.first.c_str(), level);
                    }
                }
            }
        }
    }
    // ---- A2. two streams decoded alternately (inflate_raw2 / decode_pair): every pairing of the data sets, mixed levels and
    //          strategies (stored, fixed and dynamic blocks meet each other), and a broken partner must not disturb the good one
    {
        std::vector<std::vector<uint8_t>> comp;
        std::vector<const std::vector<uint8_t>*> plain;
        int v = 0;
        for (auto& ds : sets)
            for (int level : {0, 1, 6, 9}) {
                const int strat = (v++ % 3 == 2) ? Z_FIXED : Z_DEFAULT_STRATEGY;
                comp.push_back(zlib_deflate(ds.second, level, -15, strat));
                plain.push_back(&ds.second);
            }
        for (size_t i = 0; i < comp.size(); ++i)
            for (size_t j = 0; j < comp.size(); j += (i % 3) + 1) {
                std::vector<uint8_t> o0(plain[i]->size() + 1), o1(plain[j]->size() + 1);
                const uint8_t* src[2] = {comp[i].data(), comp[j].data()};
                const size_t n[2] = {comp[i].size(), comp[j].size()}, cap[2] = {plain[i]->size(), plain[j]->size()};
                uint8_t* dst[2] = {o0.data(), o1.data()};
                int64_t got[2];
                inflate_raw2(src, n, dst, cap, got);
                CHECK(got[0] == (int64_t)cap[0] && !memcmp(o0.data(), plain[i]->data(), cap[0]) && got[1] == (int64_t)cap[1] && !memcmp(o1.data(), plain[j]->data(), cap[1]),
                      "inflate_raw2 of streams %zu and %zu: %lld of %zu, %lld of %zu", i, j, (long long)got[0], cap[0], (long long)got[1], cap[1]);
                if (comp[j].size() > 8 && (i + j) % 5 == 0) {
                    const size_t n2[2] = {comp[i].size(), comp[j].size() / 2};
                    inflate_raw2(src, n2, dst, cap, got);
                    CHECK(got[0] == (int64_t)cap[0] && !memcmp(o0.data(), plain[i]->data(), cap[0]) && got[1] < 0, "inflate_raw2 with a truncated partner (%zu, %zu)", i, j);
                }
            }
    }
    // ---- B. deflate_block -> zlib inflate
    for (auto& ds : sets) {
        for (size_t n : {(size_t)0, (size_t)1, (size_t)15, (size_t)16, (size_t)17, (size_t)100, (size_t)4096, (size_t)0xff00, (size_t)65535, (size_t)200000}) {
            if (n > ds.second.size()) continue;
            for (int level : {0, 1, 2, 6}) {
                std::vector<uint8_t> comp(deflate_bound(n));
                const size_t cn = deflate_block(ds.second.data(), n, level, comp.data());
                CHECK(cn <= comp.size() - 8, "deflate_block overran its bound");
                std::vector<uint8_t> back;
                const bool ok = zlib_inflate_raw(comp.data(), cn, back, n);
                CHECK(ok && back.size() == n && !memcmp(back.data(), ds.second.data(), n), "deflate_block %s n %zu level %d: zlib says %d, %zu bytes", ds.first.c_str(), n, level, (int)ok, back.size());
                std::vector<uint8_t> mine(n + 1);
                const int64_t got = inflate_raw(comp.data(), cn, mine.data(), n);
                CHECK(got == (int64_t)n && !memcmp(mine.data(), ds.second.data(), n), "own round trip %s n %zu level %d", ds.first.c_str(), n, level);
            }
        }
    }
    // ---- C. ParallelGunzip
    {
        const auto text = fastq_like(bench ? 400000 : 40000, 9);
        aqc_host::Pool pool(bench ? 7 : 3);
        for (int level : {1, 2, 6, 9}) {
            auto gz = zlib_deflate(text, level, 31);
            for (size_t section : {(size_t)64 << 10, (size_t)300 << 10, (size_t)4 << 20}) {
                ParallelGunzip pg(gz.data(), gz.size(), &pool, 8, section);
                std::vector<uint8_t> out(text.size() + 100);
                size_t got = 0;
                // odd read sizes: outputs straddle sections
                for (;;) {
                    const size_t want = std::min<size_t>(out.size() - got, 777777);
                    const size_t k = pg.read(out.data() + got, want);
                    got += k;
                    if (k < want) break;
                }
                CHECK(!pg.failed() && got == text.size() && !memcmp(out.data(), text.data(), text.size()), "ParallelGunzip level %d section %zu: %zu of %zu (%s)", level, section, got,
                      text.size(), pg.error());
                if (section == ((size_t)64 << 10))
                    printf("gunzip level %d: %zu B -> %zu B, sections accepted %llu discarded %llu bridged %llu B\n", level, gz.size(), text.size(),
                           (unsigned long long)pg.sections_accepted, (unsigned long long)pg.sections_discarded, (unsigned long long)pg.bridged_bytes);
                if (level == 6 && section == ((size_t)64 << 10)) CHECK(pg.sections_accepted > 4, "speculative sections were not used");
            }
        }
        // several members back to back (cat a.gz b.gz), an empty member in between, zero padding behind
        {
            std::vector<uint8_t> cat, all;
            for (int m = 0; m < 5; ++m) {
                auto part = m == 2 ? std::vector<uint8_t>() : fastq_like(3000 + 2000 * m, 40 + m);
                auto gz = zlib_deflate(part, 3, 31);
                cat.insert(cat.end(), gz.begin(), gz.end());
                all.insert(all.end(), part.begin(), part.end());
            }
            cat.insert(cat.end(), 37, 0);
            ParallelGunzip pg(cat.data(), cat.size(), &pool, 6, 64 << 10);
            std::vector<uint8_t> out(all.size() + 10);
            const size_t got = pg.read(out.data(), out.size());
            CHECK(!pg.failed() && got == all.size() && !memcmp(out.data(), all.data(), all.size()), "multi-member: %zu of %zu (%s)", got, all.size(), pg.error());
        }
        // one record over and over (ratio in the hundreds): a section's output outgrows its symbol buffer many times over
        {
            const auto rec = fastq_like(1, 77);
            std::vector<uint8_t> rep;
            while (rep.size() < (bench ? 200u : 48u) * 1000 * 1000) rep.insert(rep.end(), rec.begin(), rec.end());
            auto gz = zlib_deflate(rep, 6, 31);
            ParallelGunzip pg(gz.data(), gz.size(), &pool, 6, 64 << 10);
            std::vector<uint8_t> out(rep.size() + 10);
            size_t got = 0;
            for (;;) {
                const size_t want = std::min<size_t>(out.size() - got, 5u << 20);
                const size_t k = pg.read(out.data() + got, want);
                got += k;
                if (k < want) break;
            }
            CHECK(!pg.failed() && got == rep.size() && !memcmp(out.data(), rep.data(), rep.size()), "repeated record: %zu of %zu (%s)", got, rep.size(), pg.error());
            printf("gunzip of a repeated record: %zu B -> %zu B, sections accepted %llu discarded %llu bridged %llu B\n", gz.size(), rep.size(), (unsigned long long)pg.sections_accepted,
                   (unsigned long long)pg.sections_discarded, (unsigned long long)pg.bridged_bytes);
        }
        // non-text payload: the block finder finds nothing, everything is bridged, still exact
        {
            auto gz = zlib_deflate(sets[1].second, 6, 31);
            ParallelGunzip pg(gz.data(), gz.size(), &pool, 4, 64 << 10);
            std::vector<uint8_t> out(sets[1].second.size() + 10);
            const size_t got = pg.read(out.data(), out.size());
            CHECK(!pg.failed() && got == sets[1].second.size() && !memcmp(out.data(), sets[1].second.data(), got), "binary payload: %zu (%s)", got, pg.error());
        }
        // damage: truncated, one flipped byte in the middle, bad CRC, garbage behind the member
        {
            auto gz = zlib_deflate(text, 6, 31);
            std::vector<uint8_t> out(text.size() + 10);
            {
                ParallelGunzip pg(gz.data(), gz.size() * 2 / 3, &pool, 6, 64 << 10);
                size_t got = pg.read(out.data(), out.size());
                CHECK(pg.failed(), "truncated .gz accepted (%zu bytes)", got);
            }
            {
                auto bad = gz; bad[bad.size() / 2] ^= 0x10;
                ParallelGunzip pg(bad.data(), bad.size(), &pool, 6, 64 << 10);
                (void)pg.read(out.data(), out.size());
                CHECK(pg.failed(), "flipped byte accepted");
            }
            {
                auto bad = gz; bad[bad.size() - 6] ^= 1;
                ParallelGunzip pg(bad.data(), bad.size(), &pool, 6, 64 << 10);
                (void)pg.read(out.data(), out.size());
                CHECK(pg.failed(), "bad CRC accepted");
            }
            {
                auto bad = gz; bad.insert(bad.end(), {'j', 'u', 'n', 'k'});
                ParallelGunzip pg(bad.data(), bad.size(), &pool, 6, 64 << 10);
                (void)pg.read(out.data(), out.size());
                CHECK(pg.failed(), "trailing garbage accepted");
            }
        }
        if (bench) {
            auto gz = zlib_deflate(text, 2, 31);
            std::vector<uint8_t> out(text.size() + 10);
            double t0 = now();
            { ParallelGunzip pg(gz.data(), gz.size(), &pool, 16, 1 << 20); (void)pg.read(out.data(), out.size()); }
            double t1 = now();
            printf("ParallelGunzip 8 threads: %.1f MB/s of text\n", text.size() / (t1 - t0) / 1e6);
            { aqc_host::Pool none(0); ParallelGunzip pg(gz.data(), gz.size(), &none, 1, 1 << 30); t0 = now(); (void)pg.read(out.data(), out.size()); t1 = now(); }
            printf("one section (16-bit symbols + resolve), 1 thread: %.1f MB/s\n", text.size() / (t1 - t0) / 1e6);
            auto raw = zlib_deflate(text, 2, -15);
            t0 = now(); (void)inflate_raw(raw.data(), raw.size(), out.data(), text.size()); t1 = now();
            printf("inflate_raw: %.1f MB/s\n", text.size() / (t1 - t0) / 1e6);
            std::vector<uint8_t> back;
            t0 = now(); zlib_inflate_raw(raw.data(), raw.size(), back, text.size()); t1 = now();
            printf("zlib inflate: %.1f MB/s\n", text.size() / (t1 - t0) / 1e6);
            for (int level : {1, 2, 6}) {
                size_t total = 0;
                std::vector<uint8_t> comp(deflate_bound(0xff00));
                t0 = now();
                for (size_t o = 0; o < text.size(); o += 0xff00) total += deflate_block(text.data() + o, std::min<size_t>(0xff00, text.size() - o), level, comp.data());
                t1 = now();
                printf("deflate_block level %d: %.1f MB/s, ratio %.3f\n", level, text.size() / (t1 - t0) / 1e6, (double)text.size() / total);
                t0 = now();
                auto z = zlib_deflate(text, level, -15);
                t1 = now();
                printf("zlib deflate level %d: %.1f MB/s, ratio %.3f\n", level, text.size() / (t1 - t0) / 1e6, (double)text.size() / z.size());
            }
        }
    }
    printf(failures ? "FAILED: %d\n" : "all gz codec checks passed\n", failures);
    return failures ? 1 : 0;
}
