// aqc_gz.hpp — the pipe's own gzip codec (host code): what fastq.py:23-24,65-68 gets from Python's gzip module upstream.
//
//   inflate   a table-driven DEFLATE decoder (RFC 1951: 11-bit litlen / 8-bit distance root tables + subtables, 64-bit
//             bit buffer, several literals per refill) in two output flavours: bytes (window known) and 16-bit SYMBOLS, where
//             a value >= 0x8000 stands for "byte j of the 32 KiB window before my start" — so a thread can start in the
//             MIDDLE of a single-member stream, at a block boundary it found by itself, long before that window is known.
//   ParallelGunzip   cuts one gzip stream into sections, decodes them speculatively on the pool in symbol form, and commits
//             them in order: a section counts only if it begins at the very bit its predecessor ended on (decoding is
//             deterministic from a block boundary, so that makes the result exact); its markers are then resolved against the
//             real window.  Anything that does not chain up is decoded again sequentially from the last good bit.
//             Member CRC-32 / ISIZE trailers are verified; a stream that ends early or fails its CRC is an error.
//   deflate   aqc_deflate.cpp: one-pass greedy LZ77 (cost-gated matches) + dynamic Huffman, one block per call.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <vector>

#include "aqc_pool.hpp"

namespace aqcgz {

enum { GZ_OK = 0, GZ_ERR_DATA = -1, GZ_NEED_OUTPUT = 1, GZ_STOPPED = 2, GZ_FINAL = 3, GZ_CONTINUE = 4, GZ_SLOW = 5 };

constexpr int LIT_ROOT = 11, DIST_ROOT = 8;
constexpr int LIT_TABLE = (1 << LIT_ROOT) + 2048, DIST_TABLE = (1 << DIST_ROOT) + 1024;
constexpr uint32_t MARKER = 0x8000u;
constexpr size_t WINDOW = 32768;

// One DEFLATE stream position + the tables of the block it is in.  OutT = uint8_t (bytes) or uint16_t (symbols, see above).
// run() decodes whole blocks and may be resumed after GZ_NEED_OUTPUT with a bigger / emptied output buffer: out[0 .. out_pos)
// is what has been produced, out[-hist .. 0) is readable history (the window, or the marker prefix).
template <typename OutT>
struct Inflater {
    const uint8_t* in = nullptr;
    size_t in_size = 0;
    uint64_t bitpos = 0;          // absolute bit position in `in`
    OutT* out = nullptr;
    size_t out_pos = 0, out_cap = 0;
    size_t hist = 0;              // elements readable before out[0]
    // block state
    int in_block = 0;             // 0 between blocks, 1 stored, 2 huffman
    bool bfinal = false, final_done = false;
    uint32_t stored_left = 0;
    uint64_t blocks = 0;          // blocks completed
    uint32_t lit[LIT_TABLE];
    uint32_t dist[DIST_TABLE];

    void reset(const uint8_t* data, size_t size, uint64_t bit) {
        in = data; in_size = size; bitpos = bit; in_block = 0; bfinal = false; final_done = false; stored_left = 0; blocks = 0;
    }
    // decode until: the final block has ended (GZ_FINAL); a block boundary at or behind stop_bit is reached (GZ_STOPPED);
    // the output is (nearly) full (GZ_NEED_OUTPUT); the data is invalid or ends early (GZ_ERR_DATA)
    int run(uint64_t stop_bit);
    // one transition of run(): a block header read, or a whole block decoded -> GZ_CONTINUE; else what run() would return
    int run_step(uint64_t stop_bit);
    // the rest of the current Huffman block (in_block == 2) -> GZ_OK at its end-of-block code, GZ_NEED_OUTPUT, GZ_ERR_DATA
    int decode_huffman();
    // bookkeeping behind a block's end-of-block code (run_step does it itself; for the callers of decode_pair)
    void end_block() { in_block = 0; blocks++; if (bfinal) final_done = true; }

private:
    int read_header();
};

// Two streams, both inside a Huffman block (in_block == 2), decoded ALTERNATELY symbol group by symbol group until one of them
// leaves its fast loop: returns 0 / 1 = which one, *rc = GZ_OK (its end-of-block code is consumed: call end_block()),
// GZ_ERR_DATA, or GZ_SLOW (input or output headroom used up: decode_huffman() takes it from there).  The other stream stops
// between two symbols, consistent.  One decoder is a single dependency chain that leaves most of a core idle; two independent
// ones in the same instruction window run at 1.4 - 1.6 x the rate of one after the other (tools/ubench/gz_rate.cpp).
template <typename OutT>
int decode_pair(Inflater<OutT>& a, Inflater<OutT>& b, int* rc);

// gzip member header at data[pos]: returns the offset of the deflate data, 0 on a malformed / truncated header
size_t parse_gzip_header(const uint8_t* data, size_t size, size_t pos);

// first bit position in [from_bit, to_bit) at which a non-final dynamic-Huffman block plausibly begins: complete code-length
// / litlen / distance codes, the whole block decodes, its literals are text (FASTQ is) and a sane block header follows.
// UINT64_MAX when none.  False positives are harmless (see ParallelGunzip), they only cost time.
uint64_t find_block_start(const uint8_t* data, size_t size, uint64_t from_bit, uint64_t to_bit);

// whole raw-deflate stream with no history into exactly `cap` bytes (BGZF members): returns bytes written or -1
int64_t inflate_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);
// two such streams decoded alternately on one thread (decode_pair): got[k] = bytes written or -1
void inflate_raw2(const uint8_t* const src[2], const size_t n[2], uint8_t* const dst[2], const size_t cap[2], int64_t got[2]);

uint32_t crc32_fast(uint32_t crc, const uint8_t* p, size_t n);
uint32_t crc32_combine_fast(uint32_t crc1, uint32_t crc2, uint64_t len2);

// Sections decoded somewhere else than on the pool: the GPU (aqc_capi.hip: DeviceInflate, kernels in aqc_gunzip_dev.hpp).  A
// GROUP of consecutive sections is handed over at once; each comes back as what a pool thread would have produced — the block
// boundary it starts at, the one it ends at, its symbols — or as "nothing found".  The consumer's commit rule does not care who
// decoded a section, so exactness does not depend on the device being right, only speed does.
//
// RESIDENT results (round 6): the symbols of a section STAY with the decoder (sym == nullptr, resident == true) — in HBM — and
// the consumer, once it has committed everything in front of a run of chained sections and therefore knows the 32 KiB before
// it, has the decoder resolve the run's markers and compute each section's CRC-32 where the symbols are (resolve()).  The
// host then sees bytes, not symbols: fetch() copies a piece of a resolved section's text wherever the consumer wants it
// (the pipe: straight into the page-locked chunk buffer), and neither the translation nor the checksum costs a CPU cycle.
constexpr size_t NL_PIECE = 65536;      // resolve() counts a resident section's line feeds in pieces of this size, RIGHT-aligned in the section
                                        // (only the first piece is short): piece j of a section of n bytes ends at n - (pieces - 1 - j) * NL_PIECE
// A stretch of a caller's buffer whose bytes are NOT written there because they are text in device memory (a resolved resident
// section, or part of one): ParallelGunzip::read(dst, want, &segments).  The pipe hands such stretches to aqc_frame_mixed, which
// copies them inside the device; PCIe never sees them.
struct DevSegment {
    size_t dst_off = 0, len = 0;        // where in the caller's buffer the bytes belong
    const uint8_t* dev = nullptr;       // the bytes
    int device = -1;                    // ... on this device
    size_t sec_off = 0, sec_len = 0;    // which part of its section this is: [sec_off, sec_off + len) of sec_len bytes
    const uint32_t* piece_nl = nullptr; // line feeds per NL_PIECE piece of the section (see above)
    void* token = nullptr;              // SectionOffload::fetch(token, offset in the section, ...) still works
    class SectionOffload* owner = nullptr;
    std::shared_ptr<void> keep;         // the section: its text stays where it is while somebody holds this
};
struct OffloadResult {
    bool found = false;
    uint64_t start_bit = 0, end_bit = 0;
    const uint16_t* sym = nullptr;      // n_sym symbols, markers relative to the section's start; valid until release(token)
    size_t n_sym = 0;
    void* token = nullptr;
    bool resident = false;              // the symbols are with the decoder: resolve() + fetch() instead of `sym`
};
class SectionOffload {
public:
    virtual ~SectionOffload() {}
    virtual size_t group_bytes() const = 0;          // compressed bytes it likes to take per group
    virtual bool ready() = 0;                        // would submit() be accepted right now?  (never blocks)
    // groups of about `group_bytes` compressed bytes are going to be submitted: set up what that needs (device buffers) in the
    // background; ready() stays false until it exists, so that nobody waits for a group that waits for an allocation
    virtual void prepare(size_t group_bytes) { (void)group_bytes; }
    // the decoder no longer takes work and never will again (as opposed to "busy right now")
    virtual bool gave_up() { return false; }
    // n consecutive sections of data[0, size): search from nominal[k] (exact[k]: the section must start AT it), stop at the first
    // block boundary at or behind stop[k].  done(k, result) is called exactly once per section, from another thread.
    virtual bool submit(const uint8_t* data, size_t size, int n, const uint64_t* nominal, const uint64_t* stop, const uint8_t* exact,
                        std::function<void(int, const OffloadResult&)> done) = 0;
    virtual void release(void* token) = 0;           // the symbols (and the text) of one result are no longer needed
    // Resident results only.  tokens[0, n): consecutive sections of ONE group, each starting at the bit its predecessor ended on;
    // win[0, wlen): the <= 32 KiB of the member's output right before tokens[0]'s first symbol.  Resolves every marker, leaves
    // crc[k] = CRC-32 of section k's bytes and tail[0, *tail_len) = the last <= 32 KiB of the member's output behind the run.
    // Returns 0, GZ_ERR_DATA (a marker points before the member's start: corrupt data) or -2 (the decoder failed: nothing of
    // these sections can be used).
    // piece_nl (may be null): line feeds of every NL_PIECE piece of every section, in order (sum of ceil(bytes / NL_PIECE) entries).
    virtual int resolve(void* const* tokens, int n, const uint8_t* win, size_t wlen, uint32_t* crc, uint8_t* tail, size_t* tail_len, uint32_t* piece_nl) {
        (void)tokens; (void)n; (void)win; (void)wlen; (void)crc; (void)tail; (void)tail_len; (void)piece_nl;
        return -2;
    }
    // where a resolved section's text is (nullptr: not in device memory)
    virtual const uint8_t* text_ptr(void* token, int* device) { (void)token; (void)device; return nullptr; }
    // bytes [off, off + len) of a resolved section -> dst (queued; fetch_wait() returns once every queued copy has landed)
    virtual bool fetch(void* token, size_t off, size_t len, uint8_t* dst) { (void)token; (void)off; (void)len; (void)dst; return false; }
    virtual bool fetch_wait() { return false; }
};
// the device decoder of GPU `device` (aqc_capi.hip); nullptr when it cannot be set up
SectionOffload* make_device_offload(int device, size_t group_bytes);
// kernel / copy microseconds of every device decoder of the process so far: scan, decode, chain + gather, H2D, D2H, groups, sections given, sections found
void device_offload_stats(uint64_t out[8]);
// resident results (round 6): runs resolved on the devices, their sections, microseconds inside resolve(), bytes of text resolved
void device_resolve_stats(uint64_t out[4]);

class ParallelGunzip {
public:
    // data: the whole compressed file (mapped); sections of `section_bytes` compressed bytes, at most `inflight` of them on the
    // pool at a time.  offload: groups of sections go there whenever it is ready (offload_only: the pool only takes what cannot
    // be grouped: the stream's last section).
    ParallelGunzip(const uint8_t* data, size_t size, aqc_host::Pool* pool, int inflight, size_t section_bytes, SectionOffload* offload = nullptr,
                   bool offload_only = false);
    ~ParallelGunzip();
    // segs != nullptr: text that is in device memory (resolved resident sections) is not copied to dst — its place there stays
    // unwritten and is listed in *segs (appended, in order of dst_off) instead
    size_t read(uint8_t* dst, size_t want, std::vector<DevSegment>* segs = nullptr);
    bool failed() const { return bad_; }
    const char* error() const { return err_; }
    // statistics
    uint64_t sections_accepted = 0, sections_discarded = 0, bridged_bytes = 0, total_out = 0;
    uint64_t sections_offloaded = 0, offloaded_accepted = 0, offloaded_bytes = 0;      // handed to the device / committed from it / their text
    uint64_t resident_bytes = 0;                   // of offloaded_bytes: text whose markers and CRC-32 the device resolved (no host translation)
    // where the consumer's wall time inside read() went, microseconds: waiting for a section of the pool / of the device to be
    // decoded, waiting for (and helping with) the translation of what it committed, handing out work, committing, bridging
    uint64_t us_wait_pool = 0, us_wait_device = 0, us_drain = 0, us_top_up = 0, us_accept = 0, us_bridge = 0, us_resolve = 0;

    struct Section;
    struct Shared;

private:
    struct BridgeState {
        Inflater<uint8_t> inf;
        bool active = false;
        uint64_t until = 0;
    };
    void top_up(bool need_front = false);
    void accept(Section& s, uint8_t* dst, size_t& out, size_t want);
    bool resolve_run();
    void accept_resident(const std::shared_ptr<Section>& s, uint8_t* dst, size_t& out, size_t want);
    void bridge(uint64_t until_bit, uint8_t* dst, size_t& out, size_t want);
    void emit(const uint8_t* p, size_t n, uint8_t* dst, size_t& out, size_t want);
    bool member_end(uint64_t& bit);      // trailer + next header at byte-aligned `bit`; false: no further member
    void push_window(const uint8_t* p, size_t n);
    void drain_events(bool wait_all);
    void fail(const char* what);

    const uint8_t* data_;
    size_t size_;
    aqc_host::Pool* pool_;
    int inflight_;
    size_t section_bytes_;
    SectionOffload* offload_ = nullptr;
    bool offload_only_ = false;
    std::shared_ptr<Section> make_section(size_t idx);
    void to_pool(const std::shared_ptr<Section>& s);
    void next_window();
    size_t lowest_uncreated() const { return pool_next_ < dev_hi_ ? pool_next_ : win_hi_; }
    std::shared_ptr<Shared> sh_;
    std::map<size_t, std::shared_ptr<Section>> q_;  // created and not yet committed, by section index
    std::shared_ptr<Section> unlaunched_;          // the last pool section, while it waits for a partner (sections are decoded in pairs)
    // resident sections (their symbols stay with the decoder): the run resolved last — its last section and the window behind it —,
    // the section a read() ended in the middle of, and whether copies are still on their way into the caller's buffer
    uint64_t group_seq_ = 0;
    Section* run_last_ = nullptr;
    std::vector<uint8_t> run_tail_;
    std::shared_ptr<Section> pend_sec_;
    size_t pend_off_ = 0;
    bool fetch_dirty_ = false;
    std::vector<DevSegment>* segs_ = nullptr;      // (for the duration of a read())
    bool push_segment(const std::shared_ptr<Section>& sp, size_t sec_off, size_t len, size_t dst_off);
    std::vector<std::shared_ptr<Section>> fetch_keep_;
    // Section indices: start_idx_ is the section that begins at the stream's known first block, index i > start_idx_ is searched
    // from byte i * section_bytes on, last_idx_ runs to the end of the file.  They are handed out window by window
    // ([win_lo_, win_hi_)): the POOL takes them from the bottom up (pool_next_), the DEVICE in groups from the top down
    // (dev_hi_) — the consumer commits in index order, so it reaches a device group only after everything the pool did in
    // front of it, which is as long as the device can possibly be given; the two meet wherever their speeds put them.
    size_t start_idx_ = 0, last_idx_ = 0, win_lo_ = 0, win_hi_ = 0, pool_next_ = 0, dev_hi_ = 0;
    size_t win_groups_ = 8, keep_fifths_ = 2;        // AQC_GZ_WINDOW / AQC_GZ_KEEP (read when the decoder is made)
    uint64_t start_bit0_ = 0;
    uint64_t cur_bit_ = 0;                         // everything before this bit is decoded and committed
    bool started_ = false, done_ = false, bad_ = false;
    char err_[160] = "";
    std::vector<uint8_t> window_;                  // last <= 32 KiB of the current member's output
    std::vector<uint8_t> spill_;
    size_t spill_lo_ = 0;
    std::vector<uint8_t> bridge_buf_;
    std::unique_ptr<BridgeState> bridge_state_;
    // running CRC-32 / size of the current member, folded in commit order
    uint32_t crc_ = 0;
    uint64_t isize_ = 0;
};

// ---- deflate (aqc_deflate.cpp) -------------------------------------------------------------------------------------------
// one complete raw DEFLATE stream (a single final block, or stored blocks when that is smaller) for src[0, n), n <= 65535 * 4;
// dst must hold deflate_bound(n) bytes; returns the bytes written.  level <= 0: stored.
size_t deflate_bound(size_t n);
size_t deflate_block(const uint8_t* src, size_t n, int level, uint8_t* dst);

// the shared code of the device encoder (aqc_gzdev.hpp: GzCodebookDev has this very layout): bit-reversed code | length << 16
// for all 286 literal/length and 30 distance symbols, and the block header bits, from sampled symbol counts
struct GzCodebook {
    uint32_t lit[286];
    uint32_t dist[30];
    uint32_t hdr[192];
    uint32_t hdr_bits;
    uint32_t pad_[3];
};
bool build_codebook(const uint32_t* lit_freq /* [286] */, const uint32_t* dist_freq /* [30] */, GzCodebook* cb);

}  // namespace aqcgz
