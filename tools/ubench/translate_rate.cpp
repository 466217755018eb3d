// translate_rate.cpp — marker translation variants on the symbols of a real section (host only):
//   g++ -O3 -std=c++17 -pthread tools/ubench/translate_rate.cpp afterqc_amd/csrc/aqc_inflate.cpp -lz -o /tmp/translate_rate
// A FASTQ-like text is compressed with zlib (levels 1 and 6), a section from a block boundary found mid-stream is decoded in
// symbol form (aqc_gz.hpp) and translated against a window by: the scalar loop, AVX2 narrowing + scalar halves (round 3's first
// version), AVX2 masked gathers (what aqc_gunzip.cpp does), AVX-512 masked gathers.
#include <immintrin.h>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../afterqc_amd/csrc/aqc_gz.hpp"

using namespace aqcgz;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool tr_generic(const uint16_t* s, size_t n, uint8_t* d, const uint8_t* win, size_t valid_from) {
    uint32_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t v = s[i], j = v & 0x7fffu, m = v >> 15;
        const uint8_t w = win[j];
        d[i] = m ? w : (uint8_t)v;
        bad |= m & (uint32_t)(j < valid_from);
    }
    return bad == 0;
}
__attribute__((target("avx2"))) static bool tr_halves(const uint16_t* s, size_t n, uint8_t* d, const uint8_t* win, size_t valid_from) {
    bool ok = true;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i));
        if ((uint32_t)_mm256_movemask_epi8(a) & 0xAAAAAAAAu) ok &= tr_generic(s + i, 16, d + i, win, valid_from);
        else _mm_storeu_si128((__m128i*)(d + i), _mm_packus_epi16(_mm256_castsi256_si128(a), _mm256_extracti128_si256(a, 1)));
    }
    return ok & tr_generic(s + i, n - i, d + i, win, valid_from);
}
__attribute__((target("avx2"))) static bool tr_gather(const uint16_t* s, size_t n, uint8_t* d, const uint8_t* win, size_t valid_from) {
    size_t i = 0;
    const __m256i k7fff = _mm256_set1_epi32(0x7fff), kff = _mm256_set1_epi32(0xff), vfrom = _mm256_set1_epi32((int)valid_from);
    __m256i bad = _mm256_setzero_si256();
    for (; i + 16 <= n; i += 16) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i));
        if (!((uint32_t)_mm256_movemask_epi8(a) & 0xAAAAAAAAu)) {
            _mm_storeu_si128((__m128i*)(d + i), _mm_packus_epi16(_mm256_castsi256_si128(a), _mm256_extracti128_si256(a, 1)));
            continue;
        }
        const __m256i lo = _mm256_cvtepu16_epi32(_mm256_castsi256_si128(a)), hi = _mm256_cvtepu16_epi32(_mm256_extracti128_si256(a, 1));
        const __m256i mlo = _mm256_cmpgt_epi32(lo, k7fff), mhi = _mm256_cmpgt_epi32(hi, k7fff);
        const __m256i ilo = _mm256_and_si256(lo, k7fff), ihi = _mm256_and_si256(hi, k7fff);
        const __m256i glo = _mm256_mask_i32gather_epi32(lo, (const int*)win, ilo, mlo, 1), ghi = _mm256_mask_i32gather_epi32(hi, (const int*)win, ihi, mhi, 1);
        bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_and_si256(mlo, _mm256_cmpgt_epi32(vfrom, ilo)), _mm256_and_si256(mhi, _mm256_cmpgt_epi32(vfrom, ihi))));
        const __m256i p16 = _mm256_permute4x64_epi64(_mm256_packus_epi32(_mm256_and_si256(glo, kff), _mm256_and_si256(ghi, kff)), 0xD8);
        _mm_storeu_si128((__m128i*)(d + i), _mm_packus_epi16(_mm256_castsi256_si128(p16), _mm256_extracti128_si256(p16, 1)));
    }
    return (_mm256_testz_si256(bad, bad) != 0) & tr_generic(s + i, n - i, d + i, win, valid_from);
}
__attribute__((target("avx512f,avx512bw,avx512vl"))) static bool tr_gather512(const uint16_t* s, size_t n, uint8_t* d, const uint8_t* win, size_t valid_from) {
    size_t i = 0;
    const __m512i k7fff = _mm512_set1_epi32(0x7fff), vfrom = _mm512_set1_epi32((int)valid_from);
    __mmask16 bad = 0;
    for (; i + 32 <= n; i += 32) {
        const __m512i a = _mm512_loadu_si512((const void*)(s + i));
        const __mmask32 m = _mm512_movepi16_mask(a);
        if (!m) { _mm256_storeu_si256((__m256i*)(d + i), _mm512_cvtepi16_epi8(a)); continue; }
        const __m512i lo = _mm512_cvtepu16_epi32(_mm512_castsi512_si256(a)), hi = _mm512_cvtepu16_epi32(_mm512_extracti64x4_epi64(a, 1));
        const __mmask16 mlo = (__mmask16)m, mhi = (__mmask16)(m >> 16);
        const __m512i ilo = _mm512_and_si512(lo, k7fff), ihi = _mm512_and_si512(hi, k7fff);
        const __m512i glo = _mm512_mask_i32gather_epi32(lo, mlo, ilo, win, 1), ghi = _mm512_mask_i32gather_epi32(hi, mhi, ihi, win, 1);
        bad |= _mm512_mask_cmplt_epi32_mask(mlo, ilo, vfrom) | _mm512_mask_cmplt_epi32_mask(mhi, ihi, vfrom);
        _mm_storeu_si128((__m128i*)(d + i), _mm512_cvtepi32_epi8(glo));
        _mm_storeu_si128((__m128i*)(d + i + 16), _mm512_cvtepi32_epi8(ghi));
    }
    return (bad == 0) & tr_generic(s + i, n - i, d + i, win, valid_from);
}

int main() {
    std::vector<uint8_t> text;
    {
        std::mt19937 rng(7);
        const char B[4] = {'A', 'C', 'G', 'T'};
        char name[128];
        while (text.size() < (96u << 20)) {
            const int k = snprintf(name, sizeof(name), "@SIM:1:FC1:%d:%d:%d:%d 1:N:0:ACGT\n", 1 + (int)(rng() % 4), 1101 + (int)(rng() % 1200), 1000 + (int)(rng() % 24000), 1000 + (int)(rng() % 19000));
            text.insert(text.end(), name, name + k);
            for (int i = 0; i < 150; ++i) text.push_back(B[rng() & 3]);
            text.push_back('\n'); text.push_back('+'); text.push_back('\n');
            for (int i = 0; i < 150; ++i) { const unsigned x = rng() % 100; text.push_back("EA/<6#"[x < 64 ? 0 : x < 82 ? 1 : x < 91 ? 2 : x < 97 ? 3 : x < 99 ? 4 : 5]); }
            text.push_back('\n');
        }
    }
    for (int level : {1, 6}) {
        std::vector<uint8_t> gz(compressBound((uLong)text.size()) + 64);
        z_stream z{};
        deflateInit2(&z, level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
        z.next_in = text.data(); z.avail_in = (uInt)text.size(); z.next_out = gz.data(); z.avail_out = (uInt)gz.size();
        deflate(&z, Z_FINISH);
        gz.resize(z.total_out);
        deflateEnd(&z);
        const size_t sec = 4u << 20, nominal = sec;
        const uint64_t start = find_block_start(gz.data(), gz.size(), nominal * 8, (nominal + sec) * 8);
        if (start == UINT64_MAX) { printf("level %d: no block start found\n", level); continue; }
        std::vector<uint16_t> buf(WINDOW + sec * 8 + 512);
        for (size_t j = 0; j < WINDOW; ++j) buf[j] = (uint16_t)(MARKER | j);
        auto* inf = new Inflater<uint16_t>();
        inf->reset(gz.data(), gz.size(), start);
        inf->out = buf.data() + WINDOW; inf->out_pos = 0; inf->out_cap = sec * 8; inf->hist = WINDOW;
        inf->run((nominal + sec) * 8);
        const size_t n = inf->out_pos;
        delete inf;
        size_t markers = 0;
        for (size_t i = 0; i < n; ++i) markers += buf[WINDOW + i] >= MARKER;
        std::vector<uint8_t> win(WINDOW + 64, 0), o1(n + 64), o2(n + 64);
        for (size_t j = 0; j < WINDOW; ++j) win[j] = (uint8_t)("ACGT\n@+EA/<6#"[j % 13]);
        printf("zlib level %d: a section of %zu symbols, %.1f %% of them markers\n", level, n, 100.0 * markers / n);
        typedef bool (*fn)(const uint16_t*, size_t, uint8_t*, const uint8_t*, size_t);
        const struct { const char* name; fn f; bool need512; } v[] = {{"scalar", tr_generic, false}, {"AVX2 narrowing, scalar where a half holds markers", tr_halves, false},
                                                                      {"AVX2 masked gathers", tr_gather, false}, {"AVX-512 masked gathers", tr_gather512, true}};
        tr_generic(buf.data() + WINDOW, n, o1.data(), win.data(), 0);
        for (auto& x : v) {
            if (x.need512 && !__builtin_cpu_supports("avx512bw")) continue;
            double best = 1e9;
            for (int r = 0; r < 7; ++r) { const double t0 = now(); x.f(buf.data() + WINDOW, n, o2.data(), win.data(), 0); best = std::min(best, now() - t0); }
            printf("  %-52s %6.2f GB/s of text  %s\n", x.name, n / best / 1e9, memcmp(o1.data(), o2.data(), n) ? "MISMATCH" : "(same bytes)");
        }
    }
    return 0;
}
