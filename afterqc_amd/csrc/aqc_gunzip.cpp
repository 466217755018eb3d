// aqc_gunzip.cpp — ParallelGunzip: one gzip stream decoded by many threads, exactly (see aqc_gz.hpp for the idea).
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <mutex>
#include <thread>

#include <sys/mman.h>

#include "aqc_gz.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace aqcgz {

// -DAQC_GZ_PROFILE (tools/ubench/gz_rate.cpp; AQC_PIPE_DEBUG in a pipe built with it): microseconds of thread CPU time per phase,
// summed over all threads
#ifdef AQC_GZ_PROFILE
}  // namespace aqcgz
#include <time.h>

#include <atomic>
namespace aqcgz {
std::atomic<long> gz_prof[6];       // find, decode (find included), translate, crc, consumer waits for the front section, accept
struct ProfScope {
    int k;
    double t0;
    static double cpu() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    explicit ProfScope(int kk) : k(kk), t0(cpu()) {}
    ~ProfScope() { gz_prof[k] += (long)((cpu() - t0) * 1e6); }
};
#define GZ_PROF(k) ProfScope prof_scope_##k(k)
#else
#define GZ_PROF(k) do { } while (0)
#endif

namespace {

// symbols -> bytes: literals pass, a marker j is byte j of the (right-aligned) 32 KiB window before the section; markers
// below valid_from point before the start of the member: corrupt data.  Returns false on such a marker.
// (no branch on "is it a marker": in FASTQ every read name is a copy of the one before it, so markers stay frequent through a
// whole section and come in no predictable pattern)
bool translate_generic(const uint16_t* s, size_t n, uint8_t* d, const uint8_t* win, size_t valid_from) {
    uint32_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t v = s[i];
        const uint32_t j = v & 0x7fffu;
        const uint32_t m = v >> 15;                       // 1: marker
        const uint8_t w = win[j];                         // always readable: the window buffer holds all 32 Ki entries
        d[i] = m ? w : (uint8_t)v;
        bad |= m & (uint32_t)(j < valid_from);
    }
    return bad == 0;
}

#if defined(__x86_64__)
// 16 symbols per step: a half without markers is narrowed, one with markers looks its bytes up with two masked 8-lane gathers
// (win must be readable up to WINDOW + 3: the gathers load 32 bits per lane).  2.7 - 3 x the scalar loop on marker-rich data
// (gzip -1 FASTQ: 37 % of the symbols of a section are markers, gzip -6: 12 %).
__attribute__((target("avx2"))) bool translate_avx2(const uint16_t* s, size_t n, uint8_t* d, const uint8_t* win, size_t valid_from) {
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
    bool ok = _mm256_testz_si256(bad, bad) != 0;
    ok &= translate_generic(s + i, n - i, d + i, win, valid_from);
    return ok;
}
#endif

bool translate(const uint16_t* s, size_t n, uint8_t* d, const uint8_t* win, size_t valid_from) {
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) return translate_avx2(s, n, d, win, valid_from);
#endif
    return translate_generic(s, n, d, win, valid_from);
}

struct Event {
    int kind;            // 0: piece of output (crc over len bytes), 1: end of a member (expected crc / isize)
    uint32_t crc;
    uint64_t len;
};

constexpr size_t BRIDGE_CAP = 1u << 20;

}  // namespace

// a section's symbols: WINDOW marker entries, then the output.  Anonymous mappings on transparent huge pages in 2 Mi-symbol size
// classes, recycled through Shared::free_bufs — a std::vector would zero-fill and page-fault 30 MB per section in 4 KiB steps,
// which costs as much as decoding it.
struct SymBuf {
    uint16_t* p = nullptr;
    size_t cap = 0;                             // symbols behind the WINDOW prefix (512 more are mapped as slack)
    SymBuf() = default;
    SymBuf(SymBuf&& o) noexcept : p(o.p), cap(o.cap), base_(o.base_), len_(o.len_) { o.forget(); }
    SymBuf& operator=(SymBuf&& o) noexcept {
        if (this != &o) { drop(); p = o.p; cap = o.cap; base_ = o.base_; len_ = o.len_; o.forget(); }
        return *this;
    }
    SymBuf(const SymBuf&) = delete;
    SymBuf& operator=(const SymBuf&) = delete;
    ~SymBuf() { drop(); }
    static size_t round_up(size_t n) { return (n + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1); }
    bool grow(size_t want) {                    // keeps the contents
        want = round_up(want);
        if (want <= cap) return true;
        const size_t HUGE = 2u << 20;
        const size_t bytes = ((WINDOW + want + 512) * sizeof(uint16_t) + HUGE - 1) & ~(HUGE - 1);
        void* m = mmap(nullptr, bytes + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return false;
        uint16_t* q = (uint16_t*)(((uintptr_t)m + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
        (void)madvise(q, bytes, MADV_HUGEPAGE);
        if (p) memcpy(q, p, (WINDOW + cap + 512) * sizeof(uint16_t));
        drop();
        p = q; cap = want; base_ = m; len_ = bytes + HUGE;
        return true;
    }

private:
    void* base_ = nullptr;
    size_t len_ = 0;
    void forget() { p = nullptr; cap = 0; base_ = nullptr; len_ = 0; }
    void drop() { if (base_) munmap(base_, len_); forget(); }
};

struct ParallelGunzip::Shared {
    std::mutex mu;
    std::condition_variable cv;
    int pending = 0;                            // translation tasks in flight
    bool marker_error = false;
    std::vector<SymBuf> free_bufs;
    std::deque<Event> events;                   // in commit order (filled by the consumer; piece CRCs by the tasks)
    size_t events_base = 0;                     // absolute index of events.front()
};

struct ParallelGunzip::Section {
    size_t index = 0;
    bool known_start = false;
    uint64_t nominal_bit = 0, start_bit = 0, stop_bit = 0;
    bool done = false, found = false, error = false, hit_eof = false;
    uint64_t end_bit = 0;
    SymBuf buf;
    size_t n_out = 0;
    // decoded by a SectionOffload (the device): the symbols live in its buffer until the section is dropped
    bool offloaded = false;
    const uint16_t* ext_sym = nullptr;
    void* ext_token = nullptr;
    SectionOffload* ext_owner = nullptr;
    // ... whose symbols stay there (OffloadResult::resident): the decoder resolves them once the consumer knows the window before the
    // run this section belongs to, and hands out bytes
    bool resident = false, resolved = false;
    uint64_t group_id = 0;                      // the group it was submitted with
    uint32_t crc = 0;                           // CRC-32 of its bytes (resolved)
    std::vector<uint32_t> piece_nl;             // line feeds per NL_PIECE piece (resolved)
    struct MemberEnd { size_t out_pos; uint32_t crc, isize; };
    std::vector<MemberEnd> ends;
    std::mutex mu;
    std::condition_variable cv;
    std::shared_ptr<Shared> sh;
    ~Section() {
        if (ext_owner && ext_token) ext_owner->release(ext_token);
        if (sh && buf.p) {
            std::lock_guard<std::mutex> g(sh->mu);
            if (sh->free_bufs.size() < 256) sh->free_bufs.push_back(std::move(buf));
        }
    }
};

namespace {

void run_sections(const uint8_t* data, size_t size, const std::shared_ptr<ParallelGunzip::Section>& s0, const std::shared_ptr<ParallelGunzip::Section>& s1);

}  // namespace

ParallelGunzip::ParallelGunzip(const uint8_t* data, size_t size, aqc_host::Pool* pool, int inflight, size_t section_bytes, SectionOffload* offload,
                               bool offload_only)
    : data_(data), size_(size), pool_(pool), inflight_(inflight < 1 ? 1 : inflight), section_bytes_(section_bytes < (64u << 10) ? (64u << 10) : section_bytes),
      offload_(offload), offload_only_(offload_only && offload), sh_(new Shared()) {
    // the hybrid schedule's two numbers: groups per window, and how many fifths of a group the pool must still have
    // in front of it for the device to be given another one
    if (const char* e = getenv("AQC_GZ_WINDOW")) win_groups_ = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("AQC_GZ_KEEP")) keep_fifths_ = (size_t)std::max(0, atoi(e));
}

ParallelGunzip::~ParallelGunzip() {
    if (fetch_dirty_ && offload_) (void)offload_->fetch_wait();      // (copies out of sections that are about to be released)
    // speculative sections still running hold their own references; wait for them (they read data_)
    for (auto& kv : q_) {
        const std::shared_ptr<Section>& s = kv.second;
        if (s == unlaunched_) continue;             // (never went to the pool)
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->done; });
    }
    std::unique_lock<std::mutex> lk(sh_->mu);
    sh_->cv.wait(lk, [&] { return sh_->pending == 0; });
}

void ParallelGunzip::fail(const char* what) {
    if (!bad_) snprintf(err_, sizeof(err_), "%s", what);
    bad_ = true;
}

namespace {

// One section being decoded: the block-start search, the symbol buffer, the decoder and what its return codes mean for the
// section.  Two of them can be stepped side by side (run_sections below).
struct SectionRun {
    ParallelGunzip::Section& s;
    const uint8_t* data;
    size_t size;
    std::unique_ptr<Inflater<uint16_t>> inf;
    size_t cap = 0, cap_limit = 0;  // symbols the buffer holds / may ever hold (64 x the compressed span: a speculative section that
                                    // wants more — a gzip bomb, a false start in constant data — is given up, the consumer then
                                    // decodes that stretch sequentially through its 1 MiB bridge buffer)
    size_t base_off = 0;            // symbols of earlier members of this section (a new member has no history at all)
    bool done = false;

    SectionRun(ParallelGunzip::Section& sec, const uint8_t* d, size_t n) : s(sec), data(d), size(n) {}

    // block start, buffer, decoder; false: nothing to decode (no start found / out of memory)
    bool start() {
        uint64_t start = s.start_bit;
        if (!s.known_start) {
            GZ_PROF(0);
            start = find_block_start(data, size, s.nominal_bit, s.stop_bit);
            if (start == UINT64_MAX) { done = true; return false; }
        }
        s.start_bit = start;
        s.found = true;
        inf.reset(new Inflater<uint16_t>());
        const size_t span = (size_t)((std::min<uint64_t>(s.stop_bit, (uint64_t)size * 8) - std::min<uint64_t>(start, (uint64_t)size * 8)) >> 3);
        const size_t need = SymBuf::round_up(span * 4 + (256u << 10));
        cap_limit = SymBuf::round_up(span * 64 + (8u << 20));
        if (s.sh) {
            // a recycled buffer that is big enough (sections are all alike, so nearly any is), else the last one: it grows
            std::lock_guard<std::mutex> g(s.sh->mu);
            auto& fb = s.sh->free_bufs;
            if (!fb.empty()) {
                size_t pick = fb.size() - 1;
                for (size_t i = 0; i < fb.size(); ++i) if (fb[i].cap >= need) { pick = i; break; }
                s.buf = std::move(fb[pick]);
                fb.erase(fb.begin() + (long)pick);
            }
        }
        if (!s.buf.grow(need)) { s.error = true; done = true; return false; }
        cap = s.buf.cap;
        for (size_t j = 0; j < WINDOW; ++j) s.buf.p[j] = (uint16_t)(MARKER | j);
        inf->reset(data, size, start);
        inf->out = s.buf.p + WINDOW; inf->out_pos = 0; inf->out_cap = cap; inf->hist = WINDOW;
        return true;
    }

    // what a return code of the decoder (other than GZ_CONTINUE) means here
    void on_rc(int rc) {
        if (rc == GZ_NEED_OUTPUT) {
            if (cap >= cap_limit || !s.buf.grow(std::min(cap_limit, cap + cap / 2 + (1u << 20)))) { s.error = true; done = true; return; }
            cap = s.buf.cap;
            inf->out = s.buf.p + WINDOW + base_off; inf->out_cap = cap - base_off;
            return;
        }
        if (rc == GZ_STOPPED) { s.end_bit = inf->bitpos; done = true; return; }
        if (rc == GZ_FINAL) {
            const size_t byte = (size_t)((inf->bitpos + 7) >> 3);
            if (byte + 8 > size) { s.error = true; done = true; return; }
            uint32_t crc, isz;
            memcpy(&crc, data + byte, 4); memcpy(&isz, data + byte + 4, 4);
            base_off += inf->out_pos;
            s.ends.push_back({base_off, crc, isz});
            size_t p = byte + 8;
            while (p < size && data[p] == 0) ++p;
            if (p >= size) { s.hit_eof = true; s.end_bit = (uint64_t)size * 8; inf->out_pos = 0; done = true; return; }
            const size_t h = parse_gzip_header(data, size, p);
            if (!h) { s.error = true; inf->out_pos = 0; done = true; return; }
            // the next member starts with no history: distances reaching before it are errors, no marker can appear in it
            inf->reset(data, size, (uint64_t)h * 8);
            inf->out = s.buf.p + WINDOW + base_off; inf->out_pos = 0; inf->out_cap = cap - base_off; inf->hist = 0;
            return;
        }
        s.error = true;
        done = true;
    }

    // the section is complete (or given up): publish it to the consumer
    void finish() {
        if (inf) s.n_out = base_off + inf->out_pos;
        {
            std::lock_guard<std::mutex> g(s.mu);
            s.done = true;
        }
        s.cv.notify_all();
    }
};

// One or two sections on this thread.  Two are decoded ALTERNATELY while both are inside a Huffman block (decode_pair: two
// independent dependency chains in one instruction window, 1.2 - 1.4 x the rate of one after the other); headers, stored
// blocks, member ends and whatever is left of the longer one go through the ordinary single-stream steps.  Each section is
// published the moment it is complete.
void run_sections(const uint8_t* data, size_t size, const std::shared_ptr<ParallelGunzip::Section>& s0, const std::shared_ptr<ParallelGunzip::Section>& s1) {
    GZ_PROF(1);
    SectionRun a(*s0, data, size);
    if (!s1) {
        if (a.start())
            while (!a.done) a.on_rc(a.inf->run(a.s.stop_bit));
        a.finish();
        return;
    }
    SectionRun b(*s1, data, size);
    a.start();
    b.start();
    bool a_pub = false, b_pub = false;
    for (;;) {
        if (a.done && !a_pub) { a.finish(); a_pub = true; }
        if (b.done && !b_pub) { b.finish(); b_pub = true; }
        if (a.done && b.done) break;
        if (!a.done && !b.done && a.inf->in_block == 2 && b.inf->in_block == 2) {
            int rc;
            SectionRun& r = decode_pair(*a.inf, *b.inf, &rc) ? b : a;
            if (rc == GZ_OK) r.inf->end_block();
            else if (rc == GZ_SLOW) {
                const int r2 = r.inf->decode_huffman();         // the last stretch of its input / output, with every check
                if (r2 == GZ_OK) r.inf->end_block();
                else r.on_rc(r2);
            } else r.on_rc(rc);
            continue;
        }
        SectionRun& r = (!a.done && (b.done || a.inf->in_block != 2)) ? a : b;
        const int rc = r.inf->run_step(r.s.stop_bit);
        if (rc != GZ_CONTINUE) r.on_rc(rc);
    }
}

}  // namespace

std::shared_ptr<ParallelGunzip::Section> ParallelGunzip::make_section(size_t idx) {
    std::shared_ptr<Section> s(new Section());
    s->index = idx;
    if (idx == start_idx_) {
        s->known_start = true;
        s->start_bit = start_bit0_;
        s->nominal_bit = start_bit0_;
    } else s->nominal_bit = (uint64_t)idx * section_bytes_ * 8;
    s->stop_bit = idx < last_idx_ ? (uint64_t)(idx + 1) * section_bytes_ * 8 : UINT64_MAX;
    s->sh = sh_;
    return s;
}

// sections go to the pool two at a time (run_sections decodes a pair alternately); an odd one waits for its partner
void ParallelGunzip::to_pool(const std::shared_ptr<Section>& s) {
    q_[s->index] = s;
    auto launch = [this](const std::shared_ptr<Section>& s0, const std::shared_ptr<Section>& s1) {
        const uint8_t* data = data_;
        const size_t size = size_;
        if (pool_) pool_->submit([data, size, s0, s1] { run_sections(data, size, s0, s1); }, true);
        else run_sections(data, size, s0, s1);
    };
    if (unlaunched_) { launch(unlaunched_, s); unlaunched_.reset(); }
    else unlaunched_ = s;
}

// the next window of section indices: [win_lo_, win_hi_); the file's last section runs to the end of the file (its stop bit is
// "none"), which the device cannot take: it goes to the pool at once
void ParallelGunzip::next_window() {
    const size_t per_group = std::max<size_t>(1, (offload_ ? std::min(offload_->group_bytes(), std::max<size_t>(16u << 20, size_ / 6)) : section_bytes_) / section_bytes_);
    win_hi_ = std::min(last_idx_ + 1, win_lo_ + win_groups_ * per_group);
    pool_next_ = win_lo_;
    dev_hi_ = win_hi_;
    if (win_hi_ == last_idx_ + 1) {
        dev_hi_ = last_idx_;
        to_pool(make_section(last_idx_));
    }
}

// need_front: the consumer has nothing it may commit next — the lowest section not yet created is made now, whatever the quotas say
void ParallelGunzip::top_up(bool need_front) {
    if (!started_) {
        started_ = true;
        start_bit0_ = cur_bit_;
        start_idx_ = (size_t)((cur_bit_ >> 3) / section_bytes_);
        last_idx_ = start_idx_;
        // index i > start_idx_ exists while byte i * section_bytes + 64 lies inside the file
        if (size_ > 64 && (size_ - 65) / section_bytes_ > start_idx_) last_idx_ = (size_ - 65) / section_bytes_;
        win_lo_ = start_idx_;
        next_window();
        if (offload_) {
            const size_t per_group = std::max<size_t>(1, (offload_only_ ? offload_->group_bytes() : std::min(offload_->group_bytes(), std::max<size_t>(16u << 20, size_ / 6))) / section_bytes_);
            offload_->prepare(per_group * section_bytes_);
        }
    }
    for (;;) {
        if (pool_next_ >= dev_hi_) {                          // this window is handed out
            if (win_hi_ > last_idx_) break;
            win_lo_ = win_hi_;
            next_window();
            continue;
        }
        size_t on_pool = 0, on_device = 0;
        for (auto& kv : q_) (kv.second->offloaded ? on_device : on_pool)++;
        const bool front_missing = need_front && (q_.empty() || q_.begin()->first >= lowest_uncreated());
        const size_t per_group = std::max<size_t>(1, (offload_ ? (offload_only_ ? offload_->group_bytes() : std::min(offload_->group_bytes(), std::max<size_t>(16u << 20, size_ / 6)))
                                                               : section_bytes_) / section_bytes_);
        // The pool's share, from the bottom of the window up: the sections the consumer wants next.  (With a device decoder it
        // may work twice as far ahead: it is the only one feeding the consumer while a device group is under way.)
        if (!offload_only_ && (on_pool < (size_t)inflight_ * (offload_ ? 2u : 1u) || front_missing)) {
            const size_t idx = pool_next_++;
            if (idx != start_idx_ && (uint64_t)idx * section_bytes_ * 8 <= cur_bit_) continue;     // (its predecessor ran through it)
            to_pool(make_section(idx));
            need_front = false;
            continue;
        }
        // The device's share: a GROUP of sections whenever it is free — from the TOP of the window down, so that the consumer,
        // who commits in index order, gets there last (offload_only: from the bottom up, nobody else feeds the consumer)
        if (offload_ && offload_only_ && front_missing && on_device == 0 && offload_->gave_up()) {
            // The device has given up (a failed hipMalloc of its lane buffers, any HIP error — DeviceInflate::broken_).  With the
            // device as the ONLY decoder nobody would ever make the section the consumer waits for (round-4 advisory:
            // aqc_gunzip_dev hung here): the host takes over from this section on.  (A decoder that is merely busy — the other
            // input's group, its buffers still being set up — is waited for: round-5 advisory.)
            offload_only_ = false;
            continue;
        }
        if (!offload_ || on_device >= 4 * per_group || !offload_->ready()) break;
        // ... but only while the pool still has more than two groups' worth of sections in front of it: a group takes the device
        // a fixed 100 - 200 ms (a block is decoded by one lane from start to end), and one that is started when the pool is about
        // to arrive makes the consumer wait for it
        if (!offload_only_ && dev_hi_ - pool_next_ < per_group * keep_fifths_ / 5) break;
        size_t lo, hi;
        if (offload_only_) { lo = pool_next_; hi = std::min(dev_hi_, lo + per_group); }
        else { hi = dev_hi_; lo = hi > pool_next_ + per_group ? hi - per_group : pool_next_; }
        std::vector<std::shared_ptr<Section>> group;
        for (size_t idx = lo; idx < hi; ++idx) {
            if (idx != start_idx_ && (uint64_t)idx * section_bytes_ * 8 <= cur_bit_) continue;
            group.push_back(make_section(idx));
        }
        if (offload_only_) pool_next_ = hi; else dev_hi_ = lo;
        if (group.empty()) continue;
        std::vector<uint64_t> nominal(group.size()), stop(group.size());
        std::vector<uint8_t> exact(group.size());
        ++group_seq_;
        for (size_t k = 0; k < group.size(); ++k) {
            group[k]->offloaded = true;
            group[k]->group_id = group_seq_;
            nominal[k] = group[k]->nominal_bit; stop[k] = group[k]->stop_bit; exact[k] = group[k]->known_start ? 1 : 0;
        }
        SectionOffload* const off = offload_;
        auto done = [group, off](int k, const OffloadResult& r) {
            Section& s = *group[(size_t)k];
            s.found = r.found;
            s.start_bit = r.start_bit;
            s.end_bit = r.end_bit;
            s.ext_sym = r.sym;
            s.resident = r.resident;
            s.n_out = r.n_sym;
            s.ext_token = r.token;
            s.ext_owner = off;
            {
                std::lock_guard<std::mutex> g(s.mu);
                s.done = true;
            }
            s.cv.notify_all();
        };
        if (offload_->submit(data_, size_, (int)group.size(), nominal.data(), stop.data(), exact.data(), done)) {
            for (auto& g : group) q_[g->index] = g;
            sections_offloaded += group.size();
        } else {
            for (auto& g : group) { g->offloaded = false; to_pool(g); }
        }
    }
    // an odd pool section is launched alone when nothing will follow it, and always when it is the one the consumer waits for
    if (unlaunched_ && (pool_next_ >= dev_hi_ || q_.begin()->second == unlaunched_ || !pool_ || offload_only_)) {
        const std::shared_ptr<Section> s0 = unlaunched_;
        unlaunched_.reset();
        const uint8_t* data = data_;
        const size_t size = size_;
        if (pool_) pool_->submit([data, size, s0] { run_sections(data, size, s0, nullptr); }, true);
        else run_sections(data, size, s0, nullptr);
    }
}

void ParallelGunzip::push_window(const uint8_t* p, size_t n) {
    if (n >= WINDOW) { window_.assign(p + n - WINDOW, p + n); return; }
    if (window_.size() + n > WINDOW) window_.erase(window_.begin(), window_.begin() + (window_.size() + n - WINDOW));
    window_.insert(window_.end(), p, p + n);
}

void ParallelGunzip::emit(const uint8_t* p, size_t n, uint8_t* dst, size_t& out, size_t want) {
    const size_t k = std::min(n, want - out);
    if (k) memcpy(dst + out, p, k);
    out += k;
    if (k < n) spill_.insert(spill_.end(), p + k, p + n);
}

bool ParallelGunzip::member_end(uint64_t& bit) {
    const size_t byte = (size_t)((bit + 7) >> 3);
    if (byte + 8 > size_) { fail("gzip stream ends before its trailer (truncated file)"); return false; }
    uint32_t crc, isz;
    memcpy(&crc, data_ + byte, 4); memcpy(&isz, data_ + byte + 4, 4);
    {
        std::lock_guard<std::mutex> g(sh_->mu);
        sh_->events.push_back(Event{1, crc, isz});
    }
    size_t p = byte + 8;
    while (p < size_ && data_[p] == 0) ++p;
    if (p >= size_) return false;
    const size_t h = parse_gzip_header(data_, size_, p);
    if (!h) { fail("data behind the gzip member is not a gzip member"); return false; }
    bit = (uint64_t)h * 8;
    return true;
}

// fold the finished events, in order, into the running member CRC; wait_all: wait for every translation task first
void ParallelGunzip::drain_events(bool wait_all) {
    std::unique_lock<std::mutex> lk(sh_->mu);
    if (wait_all && pool_) {
        // the consumer does not sleep while its translation pieces queue up behind the pool's sections: it takes pieces itself
        while (sh_->pending != 0) {
            lk.unlock();
            const bool did = pool_->help_front();
            lk.lock();
            if (!did && sh_->pending != 0) sh_->cv.wait_for(lk, std::chrono::microseconds(100));
        }
    } else if (wait_all) sh_->cv.wait(lk, [&] { return sh_->pending == 0; });
    if (sh_->pending != 0) return;
    if (sh_->marker_error) fail("corrupt gzip data: a back-reference reaches before the start of its member");
    while (!sh_->events.empty()) {
        const Event e = sh_->events.front();
        sh_->events.pop_front();
        sh_->events_base++;
        if (e.kind == 0) {
            crc_ = isize_ == 0 ? e.crc : crc32_combine_fast(crc_, e.crc, e.len);
            isize_ += e.len;
        } else {
            if (crc_ != e.crc || (uint32_t)isize_ != (uint32_t)e.len) fail("gzip member fails its CRC-32 / length check");
            crc_ = 0;
            isize_ = 0;
        }
    }
}

void ParallelGunzip::accept(Section& s, uint8_t* dst, size_t& out, size_t want) {
    const uint16_t* sym = s.ext_sym ? s.ext_sym : s.buf.p + WINDOW;
    const size_t n = s.n_out;
    const size_t to_dst = std::min(n, want - out);
    // (the spill buffer is empty here: read() serves it before anything else)
    spill_.resize(n - to_dst);
    spill_lo_ = 0;
    uint8_t* const d0 = dst + out;
    size_t pos = 0;
    std::shared_ptr<Section> keep;
    for (auto& kv : q_) if (kv.second.get() == &s) keep = kv.second;
    for (size_t me = 0; me <= s.ends.size(); ++me) {
        const size_t seg_end = me < s.ends.size() ? s.ends[me].out_pos : n;
        if (seg_end > pos) {
            // window of this segment: the consumer's for the part that continues the member, none behind a member start
            std::shared_ptr<std::vector<uint8_t>> win(new std::vector<uint8_t>(WINDOW + 32, 0));     // (+ slack: translate_avx2 loads 4 bytes per lane)
            const size_t wl = window_.size();
            if (wl) memcpy(win->data() + WINDOW - wl, window_.data(), wl);
            const size_t valid_from = WINDOW - wl;
            // the window behind this segment: its last <= 32 KiB, resolved here (cheap) so that the next section can go on
            {
                const size_t tail = std::min<size_t>(seg_end - pos, WINDOW);
                uint8_t tmp[WINDOW];
                if (!translate(sym + seg_end - tail, tail, tmp, win->data(), valid_from)) {
                    std::lock_guard<std::mutex> g(sh_->mu);
                    sh_->marker_error = true;
                }
                push_window(tmp, tail);
            }
            const size_t PIECE = 1u << 19;
            for (size_t a = pos; a < seg_end;) {
                size_t b = std::min(seg_end, a + PIECE);
                if (a < to_dst && b > to_dst) b = to_dst;          // a piece lies wholly in dst or wholly in the spill buffer
                uint8_t* const d = a < to_dst ? d0 + a : spill_.data() + (a - to_dst);
                size_t ev;
                {
                    std::lock_guard<std::mutex> g(sh_->mu);
                    sh_->events.push_back(Event{0, 0u, (uint64_t)(b - a)});
                    ev = sh_->events_base + sh_->events.size() - 1;
                    sh_->pending++;
                }
                std::shared_ptr<Shared> sh = sh_;
                const uint16_t* src = sym + a;
                const size_t len = b - a;
                auto job = [sh, keep, win, src, len, d, valid_from, ev] {
                    bool ok;
                    uint32_t c;
                    { GZ_PROF(2); ok = translate(src, len, d, win->data(), valid_from); }
                    { GZ_PROF(3); c = crc32_fast(0u, d, len); }
                    std::lock_guard<std::mutex> g(sh->mu);
                    if (!ok) sh->marker_error = true;
                    sh->events[ev - sh->events_base].crc = c;
                    if (--sh->pending == 0) sh->cv.notify_all();
                };
                if (pool_) pool_->submit(job, false);
                else job();
                a = b;
            }
        }
        if (me < s.ends.size()) {
            std::lock_guard<std::mutex> g(sh_->mu);
            sh_->events.push_back(Event{1, s.ends[me].crc, (uint64_t)s.ends[me].isize});
            window_.clear();
        }
        pos = seg_end;
    }
    out += to_dst;
    total_out += n;
    cur_bit_ = s.end_bit;
    if (s.hit_eof) done_ = true;
    sections_accepted++;
    if (s.offloaded) { offloaded_accepted++; offloaded_bytes += n; }
}

// The front section's symbols are with the decoder (resident): take the RUN it begins — the sections of its group behind it that
// chain on, each starting at the bit its predecessor ended on — and have the decoder resolve the run's markers against the window
// the consumer has just arrived with, CRC-32 per section included.  false: nothing of the run can be used (the decoder failed);
// its sections are dropped and the stretch is decoded here.
bool ParallelGunzip::resolve_run() {
    std::vector<std::shared_ptr<Section>> run;
    uint64_t bit = cur_bit_;
    const uint64_t gid = q_.begin()->second->group_id;
    for (auto& kv : q_) {
        const std::shared_ptr<Section>& s = kv.second;
        if (!s->offloaded || s->group_id != gid) break;
        {
            std::unique_lock<std::mutex> lk(s->mu);           // (a group's sections are handed back one after the other, microseconds apart)
            s->cv.wait(lk, [&] { return s->done; });
        }
        if (!s->found || s->error || !s->resident || s->resolved || s->start_bit != bit) break;
        run.push_back(s);
        bit = s->end_bit;
    }
    if (run.empty()) return false;
    std::vector<void*> tokens(run.size());
    std::vector<uint32_t> crc(run.size());
    size_t pieces = 0;
    for (size_t k = 0; k < run.size(); ++k) { tokens[k] = run[k]->ext_token; pieces += (run[k]->n_out + NL_PIECE - 1) / NL_PIECE; }
    std::vector<uint32_t> piece_nl(pieces);
    run_tail_.resize(WINDOW);
    size_t tail_len = 0;
    const int rc = offload_->resolve(tokens.data(), (int)run.size(), window_.data(), window_.size(), crc.data(), run_tail_.data(), &tail_len, piece_nl.data());
    if (rc == GZ_ERR_DATA) { fail("corrupt gzip data: a back-reference reaches before the start of its member"); return false; }
    if (rc != 0) {
        for (auto& s : run) { q_.erase(s->index); sections_discarded++; }
        return false;
    }
    run_tail_.resize(tail_len);
    run_last_ = run.back().get();
    size_t p0 = 0;
    for (size_t k = 0; k < run.size(); ++k) {
        const size_t cnt = (run[k]->n_out + NL_PIECE - 1) / NL_PIECE;
        run[k]->resolved = true;
        run[k]->crc = crc[k];
        run[k]->piece_nl.assign(piece_nl.begin() + (long)p0, piece_nl.begin() + (long)(p0 + cnt));
        p0 += cnt;
    }
    return true;
}

// [sec_off, sec_off + len) of a resolved resident section belongs at dst_off of the caller's buffer: listed, not copied — when the
// caller takes segments and the text is in device memory
bool ParallelGunzip::push_segment(const std::shared_ptr<Section>& sp, size_t sec_off, size_t len, size_t dst_off) {
    if (!segs_) return false;
    int dev = -1;
    const uint8_t* p = offload_->text_ptr(sp->ext_token, &dev);
    if (!p) return false;
    DevSegment g;
    g.dst_off = dst_off; g.len = len; g.dev = p + sec_off; g.device = dev; g.sec_off = sec_off; g.sec_len = sp->n_out;
    g.piece_nl = sp->piece_nl.data(); g.token = sp->ext_token; g.owner = offload_; g.keep = sp;
    segs_->push_back(std::move(g));
    return true;
}

// a resolved resident section: its bytes come straight from the decoder (fetch) or stay with it (a segment), its CRC-32 is known
void ParallelGunzip::accept_resident(const std::shared_ptr<Section>& sp, uint8_t* dst, size_t& out, size_t want) {
    Section& s = *sp;
    const size_t n = s.n_out;
    const size_t to_dst = std::min(n, want - out);
    if (to_dst && !push_segment(sp, 0, to_dst, out)) {
        if (!offload_->fetch(s.ext_token, 0, to_dst, dst + out)) fail("device gunzip: copying a section's text failed");
        fetch_dirty_ = true;
        fetch_keep_.push_back(sp);                                 // (its text must stay where it is until the copy has landed)
    }
    if (to_dst < n) { pend_sec_ = sp; pend_off_ = to_dst; }        // the rest with the next read()
    {
        std::lock_guard<std::mutex> g(sh_->mu);
        sh_->events.push_back(Event{0, s.crc, (uint64_t)n});
    }
    if (&s == run_last_) { window_.assign(run_tail_.begin(), run_tail_.end()); run_last_ = nullptr; }
    out += to_dst;
    total_out += n;
    cur_bit_ = s.end_bit;
    sections_accepted++;
    offloaded_accepted++;
    offloaded_bytes += n;
    resident_bytes += n;
}

// sequential decoding from cur_bit_ with the window known, one buffer-full per call, until a block boundary at or behind
// `until_bit` (what could not be taken from the speculative sections: a gap before a section's start, a section that did
// not chain up, the whole stream when no block start is recognised)
void ParallelGunzip::bridge(uint64_t until_bit, uint8_t* dst, size_t& out, size_t want) {
    if (!bridge_state_) bridge_state_.reset(new BridgeState());
    BridgeState& B = *bridge_state_;
    if (bridge_buf_.empty()) bridge_buf_.resize(WINDOW + BRIDGE_CAP + 512);
    uint8_t* const base = bridge_buf_.data() + WINDOW;
    if (!B.active) {
        const size_t wl = window_.size();
        if (wl) memcpy(base - wl, window_.data(), wl);
        B.inf.reset(data_, size_, cur_bit_);
        B.inf.out = base; B.inf.out_pos = 0; B.inf.out_cap = BRIDGE_CAP; B.inf.hist = wl;
        B.active = true;
        B.until = until_bit;
    }
    const int rc = B.inf.run(B.until);
    const size_t n = B.inf.out_pos;
    if (n) {
        const uint32_t c = crc32_fast(0u, base, n);
        {
            std::lock_guard<std::mutex> g(sh_->mu);
            sh_->events.push_back(Event{0, c, (uint64_t)n});
        }
        emit(base, n, dst, out, want);
        push_window(base, n);
        bridged_bytes += n;
        total_out += n;
        // keep the last 32 KiB as history in front of the buffer
        const size_t keep = std::min<size_t>(WINDOW, B.inf.hist + n);
        memmove(base - keep, base + n - keep, keep);
        B.inf.hist = keep;
        B.inf.out_pos = 0;
    }
    if (rc == GZ_NEED_OUTPUT) return;
    if (rc == GZ_STOPPED) { cur_bit_ = B.inf.bitpos; B.active = false; return; }
    if (rc == GZ_FINAL) {
        uint64_t bit = B.inf.bitpos;
        B.active = false;
        window_.clear();
        if (!member_end(bit)) { done_ = true; return; }
        cur_bit_ = bit;
        return;
    }
    B.active = false;
    fail("corrupt or truncated gzip data");
}

size_t ParallelGunzip::read(uint8_t* dst, size_t want, std::vector<DevSegment>* segs) {
    segs_ = segs;
    struct SegsOff { std::vector<DevSegment>*& p; ~SegsOff() { p = nullptr; } } segs_off{segs_};
    size_t out = 0;
    if (spill_lo_ < spill_.size()) {
        const size_t k = std::min(want, spill_.size() - spill_lo_);
        memcpy(dst, spill_.data() + spill_lo_, k);
        spill_lo_ += k;
        out = k;
        if (spill_lo_ == spill_.size()) { spill_.clear(); spill_lo_ = 0; }
        if (out == want) return out;
    }
    auto finish_fetches = [&] {
        if (fetch_dirty_) { fetch_dirty_ = false; if (!offload_->fetch_wait()) fail("device gunzip: copying a section's text failed"); }
        fetch_keep_.clear();
    };
    if (pend_sec_) {
        // what is left of the resident section the last call ended in
        const size_t k = std::min(want - out, pend_sec_->n_out - pend_off_);
        if (k && !push_segment(pend_sec_, pend_off_, k, out)) {
            if (!offload_->fetch(pend_sec_->ext_token, pend_off_, k, dst + out)) fail("device gunzip: copying a section's text failed");
            fetch_dirty_ = true;
            fetch_keep_.push_back(pend_sec_);
        }
        pend_off_ += k;
        out += k;
        if (pend_off_ == pend_sec_->n_out) { finish_fetches(); pend_sec_.reset(); pend_off_ = 0; }
        if (out == want || bad_) { finish_fetches(); return bad_ ? 0 : out; }
    }
    if (!started_ && !bad_ && !done_) {
        if (size_ == 0) done_ = true;
        else {
            const size_t h = parse_gzip_header(data_, size_, 0);
            if (!h) fail("not a gzip file");
            cur_bit_ = (uint64_t)h * 8;
        }
    }
    auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (out < want && !bad_ && !done_) {
        if (bridge_state_ && bridge_state_->active) { const uint64_t t0 = now_us(); bridge(0, dst, out, want); us_bridge += now_us() - t0; continue; }
        // the lowest section in the queue may be committed next only if every section below it has been created (the device's
        // groups come from the top of the window: there may be a gap below them that nobody has been given yet)
        uint64_t t0 = now_us();
        for (bool need = false;; need = true) {
            top_up(need);
            if (!q_.empty() && q_.begin()->first < lowest_uncreated()) break;
            if (lowest_uncreated() > last_idx_) break;            // nothing left to create
            if (need) std::this_thread::sleep_for(std::chrono::microseconds(50));      // (offload_only: a device lane is about to be free)
        }
        uint64_t t1 = now_us();
        us_top_up += t1 - t0;
        if (q_.empty()) { bridge(UINT64_MAX, dst, out, want); us_bridge += now_us() - t1; continue; }
        std::shared_ptr<Section> f = q_.begin()->second;
        {
            GZ_PROF(4);
            std::unique_lock<std::mutex> lk(f->mu);
            f->cv.wait(lk, [&] { return f->done; });
        }
        t0 = now_us();
        (f->offloaded ? us_wait_device : us_wait_pool) += t0 - t1;
        const bool usable = f->found && !f->error;
        if (usable && f->start_bit == cur_bit_ && f->resident) {
            if (!f->resolved) {
                const bool ok = resolve_run();
                us_resolve += now_us() - t0;
                if (!ok) continue;                  // (failed: bad_ is set; or the run was dropped: the loop bridges the stretch)
                t0 = now_us();
            }
            accept_resident(f, dst, out, want);
            q_.erase(q_.begin());
            us_accept += now_us() - t0;
        } else if (usable && f->start_bit == cur_bit_) {
            GZ_PROF(5);
            accept(*f, dst, out, want);
            q_.erase(q_.begin());
            us_accept += now_us() - t0;
        } else if (usable && f->start_bit > cur_bit_) {
            bridge(f->start_bit, dst, out, want);
            us_bridge += now_us() - t0;
        } else {
            q_.erase(q_.begin());
            sections_discarded++;
        }
    }
    {
        const uint64_t t0 = now_us();
        finish_fetches();
        drain_events(true);
        us_drain += now_us() - t0;
    }
    if (bad_) return 0;
    return out;
}

}  // namespace aqcgz
