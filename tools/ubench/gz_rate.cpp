// gz_rate.cpp — how the pipe's gzip codec scales with threads on this host (no GPU):
//   g++ -O3 -std=c++17 -pthread [-DAQC_GZ_PROFILE] tools/ubench/gz_rate.cpp afterqc_amd/csrc/aqc_{inflate,gunzip,deflate}.cpp -lz -o /tmp/gz_rate
//   /tmp/gz_rate [MiB of FASTQ text, default 1024]
// inflate: ONE single-member .gz (zlib level 2, what Python's gzip writes) through ParallelGunzip with T pool threads;
// deflate: the same text as 0xff00-byte BGZF-style blocks over T threads.
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../afterqc_amd/csrc/aqc_gz.hpp"

using namespace aqcgz;
#ifdef AQC_GZ_PROFILE
#include <atomic>
namespace aqcgz { extern std::atomic<long> gz_prof[6]; }
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t want = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    std::vector<uint8_t> text;
    {
        // one 16 MiB piece of synthetic FASTQ, varied per repetition by its read names only (cheap to make, compresses like the real thing)
        std::mt19937 rng(7);
        const char B[4] = {'A', 'C', 'G', 'T'};
        std::vector<uint8_t> piece;
        char name[128];
        while (piece.size() < (16u << 20)) {
            const int k = snprintf(name, sizeof(name), "@SIM:1:FC1:%d:%d:%d:%d 1:N:0:ACGT\n", 1 + (int)(rng() % 4), 1101 + (int)(rng() % 1200), 1000 + (int)(rng() % 24000), 1000 + (int)(rng() % 19000));
            piece.insert(piece.end(), name, name + k);
            for (int i = 0; i < 150; ++i) piece.push_back(B[rng() & 3]);
            piece.push_back('\n'); piece.push_back('+'); piece.push_back('\n');
            for (int i = 0; i < 150; ++i) { const unsigned x = rng() % 100; piece.push_back("EA/<6#"[x < 64 ? 0 : x < 82 ? 1 : x < 91 ? 2 : x < 97 ? 3 : x < 99 ? 4 : 5]); }
            piece.push_back('\n');
        }
        text.reserve(want + piece.size());
        while (text.size() < want) {
            const size_t o = text.size();
            text.insert(text.end(), piece.begin(), piece.end());
            for (size_t i = o; i < text.size(); i += 977) if (text[i] >= 'A' && text[i] <= 'T' && text[i] != 'N') text[i] = B[rng() & 3];   // no 16 MiB-periodic repeats
        }
    }
    // single-member gzip, zlib level 2, made in parallel-free fashion (one stream): time it as the zlib baseline
    std::vector<uint8_t> gz;
    {
        z_stream z{};
        deflateInit2(&z, 2, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
        gz.resize(deflateBound(&z, (uLong)std::min<size_t>(text.size(), 1u << 30)) * (text.size() / (1u << 30) + 1) + 1024);
        z.next_out = gz.data(); z.avail_out = (uInt)std::min<size_t>(gz.size(), 0xffffffffu);
        const double t0 = now();
        for (size_t o = 0; o < text.size(); o += 1u << 30) {
            z.next_in = text.data() + o; z.avail_in = (uInt)std::min<size_t>(1u << 30, text.size() - o);
            deflate(&z, o + (1u << 30) >= text.size() ? Z_FINISH : Z_NO_FLUSH);
        }
        const double dt = now() - t0;
        gz.resize(z.total_out);
        deflateEnd(&z);
        printf("text %.2f GB -> .gz %.2f GB (zlib level 2, one stream: %.0f MB/s)\n", text.size() / 1e9, gz.size() / 1e9, text.size() / dt / 1e6);
    }
    std::vector<uint8_t> out(text.size() + 64);
    if (const char* m = getenv("GZ_MATRIX")) {
        // GZ_MATRIX="T:section KiB:in flight ..." — inflate only, one line per combination
        for (const char* q = m; *q;) {
            int T = 0, kib = 0, fl = 0, used = 0;
            if (sscanf(q, "%d:%d:%d%n", &T, &kib, &fl, &used) != 3) break;
            q += used;
            while (*q == ' ') ++q;
            aqc_host::Pool pool(T);
            const double t0 = now();
            ParallelGunzip pg(gz.data(), gz.size(), &pool, fl, (size_t)kib << 10);
            size_t got = 0;
            for (;;) {
                const size_t k = pg.read(out.data() + got, std::min<size_t>(out.size() - got, 48u << 20));
                got += k;
                if (k == 0) break;
            }
            const double dt = now() - t0;
            printf("inflate  T=%-3d section %5d KiB, %3d in flight: %7.0f MB/s of text (%s)\n", T, kib, fl, text.size() / dt / 1e6,
                   !pg.failed() && got == text.size() && !memcmp(out.data(), text.data(), got) ? "exact" : "MISMATCH");
#ifdef AQC_GZ_PROFILE
            printf("         thread-ms: find %ld, decode %ld, translate %ld, crc %ld, consumer waiting %ld, accept %ld\n", gz_prof[0].exchange(0) / 1000, gz_prof[1].exchange(0) / 1000,
                   gz_prof[2].exchange(0) / 1000, gz_prof[3].exchange(0) / 1000, gz_prof[4].exchange(0) / 1000, gz_prof[5].exchange(0) / 1000);
#endif
        }
        return 0;
    }
    for (int T : {1, 4, 8, 16, 32, 64, 96, 128}) {
        if (T > (int)std::thread::hardware_concurrency()) break;
        aqc_host::Pool pool(T);
        // ---- inflate
        {
            const int inflight = std::max(4, std::min(T + T / 2, 96));
            const size_t sec = std::min<size_t>(4u << 20, std::max<size_t>(256u << 10, gz.size() / (size_t)(4 * inflight)));
            const double t0 = now();
            ParallelGunzip pg(gz.data(), gz.size(), &pool, inflight, sec);
            size_t got = 0;
            for (;;) {
                const size_t k = pg.read(out.data() + got, std::min<size_t>(out.size() - got, 48u << 20));      // chunk-sized reads, as the pipe's reader does
                got += k;
                if (k == 0) break;
            }
            const double dt = now() - t0;
            const bool ok = !pg.failed() && got == text.size() && !memcmp(out.data(), text.data(), got);
            printf("inflate  T=%-3d %7.0f MB/s of text  (%s; sections %llu ok, %llu discarded, %.1f MB sequential; section %zu KiB, %d in flight)\n", T, text.size() / dt / 1e6,
                   ok ? "exact" : "MISMATCH", (unsigned long long)pg.sections_accepted, (unsigned long long)pg.sections_discarded, pg.bridged_bytes / 1e6, sec >> 10, inflight);
#ifdef AQC_GZ_PROFILE
            printf("         thread-ms: find %ld, decode %ld, translate %ld, crc %ld, consumer waiting %ld, accept %ld\n", gz_prof[0].exchange(0) / 1000, gz_prof[1].exchange(0) / 1000,
                   gz_prof[2].exchange(0) / 1000, gz_prof[3].exchange(0) / 1000, gz_prof[4].exchange(0) / 1000, gz_prof[5].exchange(0) / 1000);
#endif
        }
        // ---- deflate
        {
            const size_t blk = 0xff00, nb = (text.size() + blk - 1) / blk;
            std::vector<uint32_t> clen(nb);
            const double t0 = now();
            pool.parallel_for(nb, [&](size_t i) {
                static thread_local std::vector<uint8_t> comp;
                comp.resize(deflate_bound(blk));
                clen[i] = (uint32_t)deflate_block(text.data() + i * blk, std::min(blk, text.size() - i * blk), 2, comp.data()) + (uint32_t)(crc32_fast(0, text.data() + i * blk, std::min(blk, text.size() - i * blk)) & 0);
            });
            const double dt = now() - t0;
            size_t total = 0;
            for (auto c : clen) total += c;
            printf("deflate  T=%-3d %7.0f MB/s of text  (ratio %.3f)\n", T, text.size() / dt / 1e6, (double)text.size() / total);
        }
    }
    return 0;
}
