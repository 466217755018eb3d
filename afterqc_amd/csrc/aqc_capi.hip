// aqc_capi.hip — host side of libafterqc_hip.so: the C ABI declared in include/afterqc_hip.h.
//
// One aqc_ctx per GPU.  Each slot owns a HIP stream, device copies of one batch (grow-on-demand,
// sized for 288 GB parts: batches of millions of pairs are the norm) and a result buffer; uploads
// are hipMemcpyAsync on the slot's stream so that slot k+1 uploads while slot k computes.
// Statistics (counters, histograms, QC accumulators, k-mer tables) live in HBM for the lifetime of
// the context and are only copied back on request.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "aqc_kernels.hpp"
#include "aqc_fast.hpp"
#include "aqc_text.hpp"
#include "aqc_gzdev.hpp"
#include "aqc_gunzip_dev.hpp"
#include "aqc_gz.hpp"
#include <zlib.h>
#include <sys/mman.h>
#include <sched.h>
#include <pthread.h>
#include <cctype>

using namespace aqc;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(AQC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Slot {
    hipStream_t stream = nullptr;
    int* status = nullptr;           // first device-side error raised by this slot's kernels (one word per slot: the slots
                                     // of a context may be driven from different host threads); 16 bytes: the word, and at
                                     // byte 8 DevStats::err_key (the earliest record at which upstream's run would have died)
    uint64_t err_record = UINT64_MAX; // ... as the last check_status read it (aqc_error_record)
    bool has_irregular = false;      // some record of the slot has a quality line that is not as long as its sequence line
    DevBuf qlen[2], qview[2];        // quality-line lengths of an uploaded batch (aqc_batch::qlen*), final quality views of LEN_IRR records
    DevBuf seq1, qual1, off1, qoff1, len1, seq2, qual2, off2, qoff2, len2, aux[5], results;
    DevBuf deferred, n_deferred;     // records the lane-per-read kernel hands to the general kernel
    DevBuf off_stage;                // the caller's 64-bit offsets on their way to the 32-bit device form
    // text in / text out (aqc_frame, aqc_format): per file the line table and the name / strand-line descriptors
    DevBuf t_line_end[2], t_tile[2], t_name_off[2], t_name_len[2], t_plus_off[2], t_plus_len[2], t_qual_len[2];
    DevBuf t_scratch;              // FrameMeta[2] + scan totals
    DevBuf f_pos, f_tile, f_plan, f_patch, f_over, f_out[6], f_events[2];
    // AQC_FUSED=1: the verdict kernel placed the slot's records in their streams and copied the whole good ones (aqc_fast.hpp, FUSE)
    DevBuf fz_state, fz_rec[2], fz_misc;      // look-back words per batch; position words per record; ticket | abort | totals[4]
    bool fused = false;                       // ... for the records the slot holds now (aqc_format checks fz_misc's abort word)
    bool formatted_fused = false;             // the last aqc_format took that placement (aqc_format_fused)
    uint64_t n_events[2] = {0, 0};    // aqc_format_spans: events per file
    uint64_t consumed[2] = {0, 0};    // bytes of each file's chunk that the framed records take
    uint64_t f_bytes[6] = {0, 0, 0, 0, 0, 0};
    // gzip members built on the device (aqc_compress)
    DevBuf g_stage, g_sizes, g_offsets, g_total, g_hist, g_code, g_packed[6];
    uint64_t g_bytes[6] = {0, 0, 0, 0, 0, 0};
    bool compressed = false;
    bool framed = false, formatted = false;
    hipEvent_t ev_main = nullptr, ev_qc = nullptr;     // ordering between the slot's stream and the context's QC stream
    // QC kernels of this slot may still run on the QC stream: `gen` counts the aqc_qc_stat calls (bumped AFTER ev_qc is recorded),
    // `synced` how many of them somebody has waited for.  Two threads may look at one slot at a time — the thread that drives
    // it and another thread's aqc_get_qc / aqc_get_kmers, which synchronise every slot (the round-3 advisory: a plain flag that
    // either of them cleared could swallow the other's newer launch).  A waiter only ever marks what it has seen.
    struct QcGen {
        std::atomic<uint64_t> gen{0}, synced{0};
        QcGen() = default;
        QcGen(const QcGen&) {}
        QcGen& operator=(const QcGen&) { return *this; }
        bool pending() const { return synced.load(std::memory_order_acquire) < gen.load(std::memory_order_acquire); }
    } qc;
    aqc_text_chunk last_chunk{};   // what the slot's arenas hold (aqc_reframe)
    uint8_t last_byte[2] = {'\n', '\n'};
    uint32_t max_len = 0;
    uint32_t raw_max_len = 0;      // longest read of the slot (both mates), 0 = unknown
    DevBatch view{};
    uint64_t n = 0;
    bool paired = false, ran = false, used_fast = false, same_arena1 = false, same_arena2 = false;
    hipEvent_t ev[AQC_N_KERNELS][2] = {};
    bool timed[AQC_N_KERNELS] = {};
    // timing region (aqc_timing_reset / aqc_timing_mean): one event pair per launch
    std::vector<hipEvent_t> ring[AQC_N_KERNELS][2];
    int ring_used[AQC_N_KERNELS] = {};
    bool collecting = false;
};

constexpr int RING_CAP = 256;

// record the start/stop event of a launch: the "last launch" pair, or the next ring pair while collecting
static hipEvent_t launch_event(Slot& s, int k, int which) {
    if (s.collecting && s.ring_used[k] < RING_CAP) {
        if ((int)s.ring[k][which].size() <= s.ring_used[k]) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            s.ring[k][which].push_back(e);
        }
        hipEvent_t e = s.ring[k][which][s.ring_used[k]];
        if (which == 1) s.ring_used[k]++;
        return e;
    }
    return s.ev[k][which];
}

// everything queued for the slot has finished: its own stream and, if statRead kernels of this slot were sent to the
// context's QC stream, those too
static hipError_t slot_sync(Slot& s) {
    hipError_t e = hipStreamSynchronize(s.stream);
    const uint64_t g = s.qc.gen.load(std::memory_order_acquire);
    if (e == hipSuccess && s.qc.synced.load(std::memory_order_acquire) < g) {
        e = hipEventSynchronize(s.ev_qc);          // (waits for the event's LATEST record: generation g or a newer one)
        if (e == hipSuccess) {
            uint64_t seen = s.qc.synced.load(std::memory_order_relaxed);
            while (seen < g && !s.qc.synced.compare_exchange_weak(seen, g, std::memory_order_release)) { }
        }
    }
    return e;
}

constexpr uint64_t KMER_CAP = 1ull << 21;
constexpr uint64_t DENSE_CAP = (uint64_t)N_XCD * DENSE_ENTRIES;   // 4^8 pure A/C/G/T k-mers, one copy per XCD

struct QcDev {
    unsigned long long* acc = nullptr;   // [QC_ROWS * QC_COLS]
    KmerTable kt{};
    // k-mer time keys: (epoch << 34 | global record index) * 1024 + position.  The epoch is bumped when a call
    // goes back in the file (statFile's "stat the skipped reads afterwards", qualitycontrol.py:353-355), so the
    // keys order insertions exactly like the reference's sequential dict and merge across GPUs with min().
    unsigned long long last_end = 0, epoch = 0;
};

}  // namespace

struct aqc_ctx {
    bool force_generic = false;
    bool fuse_opt = false;        // AQC_FUSED=1: 2 x <=160 pairs framed on the device take the verdict kernel that also places and copies
    bool qc_inline = false;       // AQC_QC_STREAM=0: statRead kernels on the slot's stream instead of the context's QC stream
    int device = 0;
    int n_slots = 0;
    std::vector<Slot> slots;
    aqc_config cfg{};
    bool has_cfg = false;
    DevBuf circ[5];
    DevBuf kmer_partial;          // per-round u16 count slices of kmer_count_kernel
    DevBuf gz_crc;                // GzCrcTables (aqc_compress)
    hipStream_t qc_stream = nullptr;   // statRead kernels (latency-bound, a few thousand waves) run beside the slots' bandwidth-bound kernels
    std::mutex qc_mu;             // aqc_qc_stat calls of different slots queue up here: they share kmer_partial and the QC stream
    DevCircles circles{};
    unsigned long long *counters = nullptr, *ovl_hist = nullptr, *dist_hist = nullptr;
    QcDev qc[4];
    int n_cu = 256;
    char name[256] = "";
};

struct StatusWords { int status; int pad_; unsigned long long err_key; };
static const StatusWords STATUS_CLEAR{0, 0, ~0ull};
static unsigned long long* err_key_of(const Slot& sl) { return reinterpret_cast<unsigned long long*>(sl.status + 2); }

static int check_status(Slot& sl) {
    StatusWords w{0, 0, ~0ull};
    HIP_TRY(hipMemcpyAsync(&w, sl.status, sizeof(w), hipMemcpyDeviceToHost, sl.stream));
    HIP_TRY(hipStreamSynchronize(sl.stream));
    int st = w.status;
    if (st != 0) {
        (void)hipMemcpyAsync(sl.status, &STATUS_CLEAR, sizeof(STATUS_CLEAR), hipMemcpyHostToDevice, sl.stream);
        (void)hipStreamSynchronize(sl.stream);
        sl.err_record = UINT64_MAX;
        // (a hard device error raised first — a read beyond AQC_MAX_READ_LEN, a device limit — is reported as what it is: the records
        //  in front of a record-tied error are NOT valid then, and aqc_error_record stays unset; round-5 advisory)
        const bool record_kind = st == AQC_ERR_INDEX || st == AQC_ERR_ALPHABET || st == AQC_ERR_ARG;
        if (w.err_key != ~0ull && record_kind) {
            // an exception inside upstream's loop: the run ends at the EARLIEST record that raises, whatever raised first here
            sl.err_record = w.err_key >> 8;
            st = -(int)(w.err_key & 0xffu);
            const char* what = st == AQC_ERR_ALPHABET ? "a base outside the reference's COMP table reached the correction walk (KeyError upstream)"
                             : st == AQC_ERR_INDEX ? "the overlap walk read a quality line beyond its length — the line is shorter than its sequence line (IndexError upstream)"
                             : st == AQC_ERR_ARG ? "a name field the bubble filter converts with int() is not a number (ValueError upstream)"
                                                 : "device-side error";
            return fail(st, "%s; record %llu of the chunk — the run ends there, the records before it are valid (aqc_error_record)", what,
                        (unsigned long long)sl.err_record);
        }
        const char* what = st == AQC_ERR_ALPHABET ? "a base outside the reference's COMP table reached the correction walk (KeyError upstream)"
                         : st == AQC_ERR_READ_TOO_LONG ? "a read (or its quality line) is longer than AQC_MAX_READ_LEN"
                         : st == AQC_ERR_ARG ? "a read shorter than 5 bases reached statRead (IndexError upstream)"
                         : st == AQC_ERR_UNSUPPORTED ? "device limit exceeded (k-mer table full or string longer than 64)"
                                                     : "device-side error";
        return fail(st, "%s", what);
    }
    return 0;
}

#ifndef AQC_FUSE_WPBT
#define AQC_FUSE_WPBT 12
#endif
template <int NW, bool PAIRED, int WPBT, bool BARCODE, bool FUSE = false>
static void launch_fast(aqc_ctx* c, Slot* s, const aqc_config& cfg, const DevStats& st, uint64_t accum_limit, const FuseArgs* fz = nullptr) {
    constexpr uint64_t per_block = (uint64_t)WPBT * FastWaveLds<NW, PAIRED, FUSE>::PPW;
    uint64_t blocks = (s->n + per_block - 1) / per_block;
    // persistent grid: as many workgroups as the LDS footprint lets a CU hold; batches are grid-strided
    const size_t lds = sizeof(FastWaveLds<NW, PAIRED, FUSE>) * WPBT + sizeof(BlockAcc) + 64 + 17 * 16 + (FUSE ? 16 * 8 : 0);
    uint64_t per_cu = (160 * 1024) / lds;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const uint64_t cap = (uint64_t)c->n_cu * per_cu;
    if (blocks > cap) blocks = cap;
    // pairs the fast kernel cannot decide exactly (exotic bytes, very short reads, ...) are queued and
    // finished by the general wave-per-record pipeline right behind it on the same stream
    (void)hipMemsetAsync(s->n_deferred.p, 0, sizeof(unsigned int), s->stream);
    FastArgs K;
    K.fb = s->view; K.cfg = cfg; K.circ = c->circles; K.results = (aqc_result*)s->results.p; K.st = st; K.accum_limit = accum_limit;
    K.deferred = (uint32_t*)s->deferred.p; K.n_deferred = (unsigned int*)s->n_deferred.p;
    K.fz = fz ? *fz : FuseArgs{};
    hipLaunchKernelGGL((fast_filter_overlap_kernel<NW, PAIRED, WPBT, BARCODE, FUSE>), dim3((unsigned)blocks), dim3(WPBT * WAVE), 0, s->stream, K);
}

extern "C" {

int aqc_abi_version(void) { return AQC_ABI_VERSION; }

int aqc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* aqc_last_error(void) { return g_err; }

int aqc_create(int device, int n_slots, aqc_ctx** out) {
    if (!out || n_slots < 1 || n_slots > 16) return fail(AQC_ERR_ARG, "aqc_create: bad arguments");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(AQC_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(AQC_ERR_ARG, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    {
        // host threads waiting for a stream sleep instead of spinning: the pipe keeps half a dozen of them in
        // hipStreamSynchronize, and under a CPU quota every spinning waiter is a core the gzip decoder does not get
        // (AQC_SYNC=spin keeps the runtime's default)
        const char* sy = getenv("AQC_SYNC");
        if (!(sy && !strcmp(sy, "spin")) && hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
    }
    aqc_ctx* c = new aqc_ctx();
    c->device = device;
    c->n_slots = n_slots;
    c->slots.resize(n_slots);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(c->name, sizeof(c->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const char* fg = getenv("AQC_FORCE_GENERIC");
    c->force_generic = fg && fg[0] == '1';
    const char* fu = getenv("AQC_FUSED");
    c->fuse_opt = fu && fu[0] == '1';
    const char* qi = getenv("AQC_QC_STREAM");
    c->qc_inline = qi && qi[0] == '0';
    HIP_TRY(hipFuncSetAttribute((const void*)kmer_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KMER_FUSED_LDS_BYTES));
    HIP_TRY(hipStreamCreateWithFlags(&c->qc_stream, hipStreamNonBlocking));
    for (auto& s : c->slots) {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        for (int k = 0; k < AQC_N_KERNELS; k++)
            for (int j = 0; j < 2; j++) HIP_TRY(hipEventCreate(&s.ev[k][j]));
        HIP_TRY(hipEventCreateWithFlags(&s.ev_main, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.ev_qc, hipEventDisableTiming));
    }
    HIP_TRY(hipMalloc((void**)&c->counters, sizeof(unsigned long long) * (AQC_N_COUNTERS + 16)));   // +16: AQC_PROFILE builds
    HIP_TRY(hipMalloc((void**)&c->ovl_hist, sizeof(unsigned long long) * AQC_QC_COLS));
    HIP_TRY(hipMalloc((void**)&c->dist_hist, sizeof(unsigned long long) * AQC_QC_COLS));
    for (auto& s : c->slots) {
        HIP_TRY(hipMalloc((void**)&s.status, sizeof(StatusWords)));
        HIP_TRY(hipMemcpy(s.status, &STATUS_CLEAR, sizeof(STATUS_CLEAR), hipMemcpyHostToDevice));
    }
    for (int k = 0; k < 4; k++)
        HIP_TRY(hipMalloc((void**)&c->qc[k].acc, sizeof(unsigned long long) * AQC_QC_ROWS * AQC_QC_COLS));
    *out = c;
    return aqc_reset_stats(c);
}

void aqc_destroy(aqc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& s : c->slots) {
        DevBuf* bufs[] = {&s.seq1, &s.qual1, &s.off1, &s.qoff1, &s.len1, &s.seq2, &s.qual2, &s.off2, &s.qoff2, &s.len2,
                          &s.aux[0], &s.aux[1], &s.aux[2], &s.aux[3], &s.aux[4], &s.results,
                          &s.deferred, &s.n_deferred, &s.off_stage, &s.qlen[0], &s.qlen[1], &s.qview[0], &s.qview[1],
                          &s.t_line_end[0], &s.t_line_end[1], &s.t_tile[0], &s.t_tile[1], &s.t_name_off[0], &s.t_name_off[1],
                          &s.t_name_len[0], &s.t_name_len[1], &s.t_plus_off[0], &s.t_plus_off[1], &s.t_plus_len[0], &s.t_plus_len[1],
                          &s.t_qual_len[0], &s.t_qual_len[1], &s.t_scratch, &s.f_pos, &s.f_tile, &s.f_plan, &s.f_patch, &s.fz_state, &s.fz_rec[0], &s.fz_rec[1], &s.fz_misc, &s.f_over, &s.f_events[0], &s.f_events[1], &s.f_out[0], &s.f_out[1], &s.f_out[2],
                          &s.f_out[3], &s.f_out[4], &s.f_out[5], &s.g_stage, &s.g_sizes, &s.g_offsets, &s.g_total, &s.g_hist, &s.g_code,
                          &s.g_packed[0], &s.g_packed[1], &s.g_packed[2], &s.g_packed[3], &s.g_packed[4], &s.g_packed[5]};
        for (DevBuf* b : bufs) b->release();
        for (int k = 0; k < AQC_N_KERNELS; k++)
            for (int j = 0; j < 2; j++) {
                if (s.ev[k][j]) (void)hipEventDestroy(s.ev[k][j]);
                for (hipEvent_t e : s.ring[k][j]) (void)hipEventDestroy(e);
            }
        if (s.status) (void)hipFree(s.status);
        if (s.ev_main) (void)hipEventDestroy(s.ev_main);
        if (s.ev_qc) (void)hipEventDestroy(s.ev_qc);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (c->qc_stream) (void)hipStreamDestroy(c->qc_stream);
    for (auto& b : c->circ) b.release();
    c->kmer_partial.release();
    c->gz_crc.release();
    (void)hipFree(c->counters); (void)hipFree(c->ovl_hist); (void)hipFree(c->dist_hist);
    for (int k = 0; k < 4; k++) {
        (void)hipFree(c->qc[k].acc);
        if (c->qc[k].kt.keys) {
            (void)hipFree(c->qc[k].kt.keys); (void)hipFree(c->qc[k].kt.counts); (void)hipFree(c->qc[k].kt.order);
            (void)hipFree(c->qc[k].kt.dense_count); (void)hipFree(c->qc[k].kt.dense_first); (void)hipFree(c->qc[k].kt.complete);
        }
    }
    delete c;
}

int aqc_device_index(aqc_ctx* c) { return c ? c->device : -1; }

// The NUMA node the GPU hangs off (its PCI function's numa_node in sysfs); -1: unknown / the host has a single node.
int aqc_device_numa_node_of(int device) {
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* p = bus; *p; ++p) *p = (char)tolower((unsigned char)*p);
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (FILE* f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    return node;
}
int aqc_device_numa_node(aqc_ctx* c) { return c ? aqc_device_numa_node_of(c->device) : -1; }

// Bind the calling thread to the CPUs of `node` that it may run on (its current affinity mask intersected with the node's
// cpulist); memory the thread touches first then comes from that node.  Returns the number of CPUs it is bound to, 0 when
// nothing was changed (unknown node, a single-node host, an empty intersection, AQC_PIPE_NUMA=0).
int aqc_bind_thread_to_node(int node) {
    if (node < 0) return 0;
    if (const char* e = getenv("AQC_PIPE_NUMA")) if (e[0] == '0') return 0;
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return 0;
    char list[4096] = "";
    const bool got = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    if (!got) return 0;
    cpu_set_t have, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return 0;
    int n = 0;
    for (char* p = list; *p;) {
        char* q;
        const long a = strtol(p, &q, 10);
        if (q == p) break;
        long b = a;
        if (*q == '-') { p = q + 1; b = strtol(p, &q, 10); }
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu)
            if (CPU_ISSET((int)cpu, &have)) { CPU_SET((int)cpu, &want); ++n; }
        p = *q == ',' ? q + 1 : q;
        if (*q != ',') break;
    }
    if (n == 0 || n == CPU_COUNT(&have)) return 0;           // nothing to choose from
    if (pthread_setaffinity_np(pthread_self(), sizeof(want), &want) != 0) return 0;
    return n;
}

int aqc_device_name(aqc_ctx* c, char* buf, int buflen) {
    if (!c || !buf || buflen <= 0) return fail(AQC_ERR_ARG, "aqc_device_name: bad arguments");
    snprintf(buf, (size_t)buflen, "%s", c->name);
    return 0;
}

int aqc_set_config(aqc_ctx* c, const aqc_config* cfg) {
    if (!c || !cfg) return fail(AQC_ERR_ARG, "aqc_set_config: null argument");
    if (cfg->trim_front < 0 || cfg->trim_tail < 0 || cfg->trim_front2 < 0 || cfg->trim_tail2 < 0)
        return fail(AQC_ERR_ARG, "trim values must be resolved (>= 0) before they reach the device");
    if (cfg->qc_kmer < 1 || cfg->qc_kmer > 8) return fail(AQC_ERR_UNSUPPORTED, "qc_kmer %d outside 1..8", cfg->qc_kmer);
    if (cfg->barcode) {
        if (cfg->barcode_verify_len < 0 || cfg->barcode_verify_len > 32 || cfg->barcode_length < 1 ||
            cfg->barcode_length + 1 + cfg->barcode_verify_len > 62)
            return fail(AQC_ERR_UNSUPPORTED, "barcode_length + verify too long for the device path");
    }
    c->cfg = *cfg;
    c->has_cfg = true;
    return 0;
}

int aqc_set_circles(aqc_ctx* c, const double* cx, const double* cy, const double* r, const int32_t* lane,
                    const int32_t* tile, int32_t n) {
    if (!c || n < 0) return fail(AQC_ERR_ARG, "aqc_set_circles: bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    c->circles = DevCircles{};
    c->circles.n = n;
    if (n == 0) return 0;
    const void* src[5] = {cx, cy, r, lane, tile};
    const size_t sz[5] = {sizeof(double) * n, sizeof(double) * n, sizeof(double) * n, sizeof(int32_t) * n, sizeof(int32_t) * n};
    for (int k = 0; k < 5; k++) {
        if (!src[k]) return fail(AQC_ERR_ARG, "aqc_set_circles: null array");
        if (c->circ[k].reserve(sz[k])) return fail(AQC_ERR_HIP, "hipMalloc failed");
        HIP_TRY(hipMemcpy(c->circ[k].p, src[k], sz[k], hipMemcpyHostToDevice));
    }
    c->circles.cx = (const double*)c->circ[0].p;
    c->circles.cy = (const double*)c->circ[1].p;
    c->circles.cr = (const double*)c->circ[2].p;
    c->circles.lane = (const int32_t*)c->circ[3].p;
    c->circles.tile = (const int32_t*)c->circ[4].p;
    return 0;
}

int aqc_reset_stats(aqc_ctx* c) {
    if (!c) return fail(AQC_ERR_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(c->counters, 0, sizeof(unsigned long long) * (AQC_N_COUNTERS + 16)));
    HIP_TRY(hipMemset(c->ovl_hist, 0, sizeof(unsigned long long) * AQC_QC_COLS));
    HIP_TRY(hipMemset(c->dist_hist, 0, sizeof(unsigned long long) * AQC_QC_COLS));
    for (auto& sl : c->slots) { HIP_TRY(hipMemcpy(sl.status, &STATUS_CLEAR, sizeof(STATUS_CLEAR), hipMemcpyHostToDevice)); sl.err_record = UINT64_MAX; }
    for (int k = 0; k < 4; k++) {
        HIP_TRY(hipMemset(c->qc[k].acc, 0, sizeof(unsigned long long) * AQC_QC_ROWS * AQC_QC_COLS));
        c->qc[k].last_end = 0;
        c->qc[k].epoch = 0;
        if (c->qc[k].kt.keys) {
            HIP_TRY(hipMemset(c->qc[k].kt.keys, 0, sizeof(unsigned long long) * KMER_CAP));
            HIP_TRY(hipMemset(c->qc[k].kt.counts, 0, sizeof(unsigned long long) * KMER_CAP));
            HIP_TRY(hipMemset(c->qc[k].kt.order, 0xff, sizeof(unsigned long long) * KMER_CAP));
            HIP_TRY(hipMemset(c->qc[k].kt.dense_count, 0, sizeof(unsigned int) * DENSE_CAP));
            HIP_TRY(hipMemset(c->qc[k].kt.dense_first, 0xff, sizeof(unsigned long long) * DENSE_CAP));
            HIP_TRY(hipMemset(c->qc[k].kt.complete, 0, sizeof(unsigned int) * (DENSE_ENTRIES / KRED_ENTRIES)));
        }
    }
    HIP_TRY(hipDeviceSynchronize());   // (non-blocking slot streams do not wait for the null stream)
    return 0;
}

// (ARENA_SLACK readable bytes behind every arena: the lane-per-read kernel always loads whole 16-byte chunks, up to
// 256 bytes from the start of a read whatever its length)
constexpr size_t ARENA_SLACK = 1024;
constexpr size_t TEXT_FRONT = 64;

// (`front` readable bytes before the data as well: read 2 is loaded in 16-byte chunks counted from its END, the chunk
// with a read's first bases may begin up to 16 bytes before the read)
static int up(DevBuf& d, const void* src, size_t bytes, hipStream_t st, size_t front = 0) {
    if (d.reserve(front + bytes + ARENA_SLACK)) return fail(AQC_ERR_HIP, "hipMalloc of %zu bytes failed", bytes);
    if (bytes == 0) return 0;
    HIP_TRY(hipMemcpyAsync((uint8_t*)d.p + front, src, bytes, hipMemcpyHostToDevice, st));
    return 0;
}

// 64-bit host offsets -> 32-bit device offsets (through the slot's staging buffer, in stream order)
static int up_offsets(Slot& s, DevBuf& d, const uint64_t* src, uint64_t n) {
    if (d.reserve(sizeof(uint32_t) * (n ? n : 1)) || s.off_stage.reserve(sizeof(uint64_t) * (n ? n : 1)))
        return fail(AQC_ERR_HIP, "hipMalloc failed");
    if (n == 0) return 0;
    HIP_TRY(hipMemcpyAsync(s.off_stage.p, src, sizeof(uint64_t) * n, hipMemcpyHostToDevice, s.stream));
    hipLaunchKernelGGL(narrow_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s.stream, (const uint64_t*)s.off_stage.p,
                       (uint32_t*)d.p, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int fill_slot(aqc_ctx* c, Slot& s, const aqc_batch* b, bool need_qual, bool need_pair) {
    const uint64_t n = b->n;
    if (!b->seq1 || !b->off1 || !b->len1) return fail(AQC_ERR_ARG, "batch: seq1/off1/len1 are required");
    const bool paired = b->seq2 != nullptr;
    if (need_pair && !paired) return fail(AQC_ERR_ARG, "batch: this call needs seq2/off2/len2");
    if (paired && (!b->off2 || !b->len2)) return fail(AQC_ERR_ARG, "batch: off2/len2 missing");
    if (n >= (1ull << 31)) return fail(AQC_ERR_ARG, "batch: more than 2^31 records (split the batch)");
    const uint64_t lim = (1ull << 32) - 4096;      // 32-bit byte offsets on the device, chunk loads may run 288 bytes past a read's start
    if (b->bytes1 >= lim || b->qbytes1 >= lim || b->bytes2 >= lim || b->qbytes2 >= lim) return fail(AQC_ERR_ARG, "batch: an arena must be smaller than 4 GiB (split the batch)");
    // make sure earlier work on this slot has drained before its buffers are overwritten / regrown
    HIP_TRY(slot_sync(s));
    int rc;
    DevBatch v{};
    v.n = n;
    v.first_index = b->first_index;
    if ((rc = up(s.seq1, b->seq1, b->bytes1, s.stream))) return rc;
    v.seq1 = (const uint8_t*)s.seq1.p;
    if (b->qual1 && need_qual) {
        if (b->qual1 == b->seq1) v.qual1 = v.seq1;
        else {
            if ((rc = up(s.qual1, b->qual1, b->qbytes1 ? b->qbytes1 : b->bytes1, s.stream))) return rc;
            v.qual1 = (const uint8_t*)s.qual1.p;
        }
    } else if (need_qual) return fail(AQC_ERR_ARG, "batch: qual1 is required");
    if ((rc = up_offsets(s, s.off1, b->off1, n))) return rc;
    v.off1 = (const uint32_t*)s.off1.p;
    if (b->qoff1) {
        if ((rc = up_offsets(s, s.qoff1, b->qoff1, n))) return rc;
        v.qoff1 = (const uint32_t*)s.qoff1.p;
    }
    if ((rc = up(s.len1, b->len1, sizeof(uint32_t) * n, s.stream))) return rc;
    v.len1 = (const uint32_t*)s.len1.p;
    if (paired) {
        if ((rc = up(s.seq2, b->seq2, b->bytes2, s.stream, TEXT_FRONT))) return rc;
        v.seq2 = (const uint8_t*)s.seq2.p + TEXT_FRONT;
        if (b->qual2 && need_qual) {
            if (b->qual2 == b->seq2) v.qual2 = v.seq2;
            else {
                if ((rc = up(s.qual2, b->qual2, b->qbytes2 ? b->qbytes2 : b->bytes2, s.stream))) return rc;
                v.qual2 = (const uint8_t*)s.qual2.p;
            }
        } else if (need_qual) return fail(AQC_ERR_ARG, "batch: qual2 is required");
        if ((rc = up_offsets(s, s.off2, b->off2, n))) return rc;
        v.off2 = (const uint32_t*)s.off2.p;
        if (b->qoff2) {
            if ((rc = up_offsets(s, s.qoff2, b->qoff2, n))) return rc;
            v.qoff2 = (const uint32_t*)s.qoff2.p;
        }
        if ((rc = up(s.len2, b->len2, sizeof(uint32_t) * n, s.stream))) return rc;
        v.len2 = (const uint32_t*)s.len2.p;
    }
    if (b->aux_ok && b->aux_lane && b->aux_tile && b->aux_x && b->aux_y) {
        const void* src[5] = {b->aux_lane, b->aux_tile, b->aux_x, b->aux_y, b->aux_ok};
        for (int k = 0; k < 5; k++)
            if ((rc = up(s.aux[k], src[k], (k < 4 ? sizeof(int32_t) : 1) * n, s.stream))) return rc;
        v.aux_lane = (const int32_t*)s.aux[0].p;
        v.aux_tile = (const int32_t*)s.aux[1].p;
        v.aux_x = (const int32_t*)s.aux[2].p;
        v.aux_y = (const int32_t*)s.aux[3].p;
        v.aux_ok = (const uint8_t*)s.aux[4].p;
    }
    if (s.results.reserve(sizeof(aqc_result) * (n ? n : 1))) return fail(AQC_ERR_HIP, "hipMalloc failed");
    s.has_irregular = b->qlen1 != nullptr && need_qual;
    if (b->qlen1 && need_qual) {
        // quality strings with lengths of their own: the mates that differ are marked in the device copy of their length words
        if (paired && !b->qlen2) return fail(AQC_ERR_ARG, "batch: qlen1 without qlen2");
        const uint32_t* ql[2] = {b->qlen1, b->qlen2};
        DevBuf* lens[2] = {&s.len1, &s.len2};
        for (int k = 0; k < (paired ? 2 : 1); ++k) {
            if ((rc = up(s.qlen[k], ql[k], sizeof(uint32_t) * n, s.stream))) return rc;
            if (s.qview[k].reserve(sizeof(uint32_t) * (n ? n : 1))) return fail(AQC_ERR_HIP, "hipMalloc failed");
            if (n) hipLaunchKernelGGL(mark_irregular_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s.stream, (uint32_t*)lens[k]->p,
                                      (const uint32_t*)s.qlen[k].p, n);
        }
        HIP_TRY(hipGetLastError());
        v.qlen1 = (const uint32_t*)s.qlen[0].p; v.qview1 = (uint32_t*)s.qview[0].p;
        if (paired) { v.qlen2 = (const uint32_t*)s.qlen[1].p; v.qview2 = (uint32_t*)s.qview[1].p; }
        else { v.qlen2 = v.qlen1; v.qview2 = v.qview1; }
    }
    uint32_t mx = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (b->len1[i] > mx) mx = b->len1[i];
        if (paired && b->len2[i] > mx) mx = b->len2[i];
    }
    s.raw_max_len = mx;
    s.view = v;
    s.n = n;
    s.paired = paired;
    s.ran = false;
    return 0;
}

static int get_slot(aqc_ctx* c, int slot, Slot** out) {
    if (!c) return fail(AQC_ERR_ARG, "null context");
    if (slot < 0 || slot >= c->n_slots) return fail(AQC_ERR_ARG, "slot %d out of range", slot);
    HIP_TRY(hipSetDevice(c->device));
    *out = &c->slots[slot];
    return 0;
}

int aqc_upload(aqc_ctx* c, int slot, const aqc_batch* b) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!b) return fail(AQC_ERR_ARG, "null batch");
    if ((rc = fill_slot(c, *s, b, true, false))) return rc;
    s->framed = s->formatted = false;
    s->max_len = s->raw_max_len;
    return 0;
}

static int grid_for(const aqc_ctx* c, uint64_t n) {
    // persistent grid: enough workgroups to fill every CU several times over, records grid-strided
    uint64_t blocks = (n + WPB - 1) / WPB;
    uint64_t cap = (uint64_t)c->n_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int aqc_run(aqc_ctx* c, int slot, uint64_t accum_limit) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!c->has_cfg) return fail(AQC_ERR_STATE, "aqc_run before aqc_set_config");
    if (c->cfg.paired && !s->paired) return fail(AQC_ERR_STATE, "config says paired but the slot holds single-end records");
    if (c->cfg.debubble && c->circles.n > 0 && !s->view.aux_ok) return fail(AQC_ERR_ARG, "debubble needs the aux_* arrays");
    s->fused = false;
    if (s->n == 0) { s->ran = true; return 0; }
    aqc_config cfg = c->cfg;
    if (!cfg.paired) cfg.no_overlap = 1;
    DevStats st{c->counters, c->ovl_hist, c->dist_hist, s->status, err_key_of(*s)};
    s->err_record = UINT64_MAX;
    if (s->qc.pending()) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_qc, 0));      // (statRead of the previous run still reads the results)
    HIP_TRY(hipEventRecord(launch_event(*s, AQC_K_FILTER_OVERLAP, 0), s->stream));
    // lane-per-pair kernel whenever its preconditions hold; the general wave-per-record kernel otherwise
    const int thr = cfg.qualified_quality_phred + 33;
    // barcodes on that kernel: detectBarcode's three windows must lie in the first 32 bases and the verify sequence must
    // be plain A/C/G/T (2-bit codes); anything else takes the general kernel
    bool barcode_ok = true;
    if (cfg.barcode) {
        barcode_ok = cfg.barcode_verify_len >= 1 && cfg.barcode_length + 1 + cfg.barcode_verify_len <= 31;
        for (int j = 0; j < cfg.barcode_verify_len && barcode_ok; ++j) {
            const uint8_t ch = cfg.barcode_verify[j];
            barcode_ok = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        }
    }
    const bool fast_ok = !c->force_generic && barcode_ok && thr >= 0 && thr <= 127 && s->max_len <= 288 && s->max_len > 0;
    s->used_fast = fast_ok;
    if (!fast_ok) {
        hipLaunchKernelGGL(filter_overlap_kernel, dim3(grid_for(c, s->n)), dim3(BLOCK), 0, s->stream, s->view, cfg, c->circles,
                           (aqc_result*)s->results.p, st, accum_limit);
    } else {
        if (s->deferred.reserve(sizeof(uint32_t) * (s->n + 1)) || s->n_deferred.reserve(sizeof(unsigned int)))
            return fail(AQC_ERR_HIP, "hipMalloc failed");
        // AQC_FUSED=1: pairs of device-framed text whose records are plain four-line text are placed in their output
        // streams by the verdict kernel itself, which also copies the good records that go out as their own bytes
        const bool fuse_ok = c->fuse_opt && s->framed && cfg.paired && !cfg.barcode && s->max_len <= 160 && !s->has_irregular && s->consumed[0] + 16 * s->n < (1ull << 31) && s->consumed[1] + 16 * s->n < (1ull << 31);      // (31-bit stream offsets; a bad record grows by its flag text)
        if (fuse_ok) {
            constexpr uint64_t PPW = FastWaveLds<10, true, true>::PPW;
            const uint64_t n_batches = (s->n + PPW - 1) / PPW;
            if (s->fz_state.reserve(16 * n_batches) || s->fz_rec[0].reserve(4 * s->n) || s->fz_rec[1].reserve(4 * s->n) || s->fz_misc.reserve(64) ||
                s->f_out[0].reserve(s->consumed[0] + 64) || s->f_out[3].reserve(s->consumed[1] + 64))
                return fail(AQC_ERR_HIP, "hipMalloc failed");
            HIP_TRY(hipMemsetAsync(s->fz_state.p, 0, 16 * n_batches, s->stream));
            HIP_TRY(hipMemsetAsync(s->fz_misc.p, 0, 64, s->stream));
            FuseArgs fz{};
            fz.name_off1 = (const uint32_t*)s->t_name_off[0].p; fz.name_off2 = (const uint32_t*)s->t_name_off[1].p;
            fz.out1 = (uint8_t*)s->f_out[0].p; fz.out2 = (uint8_t*)s->f_out[3].p;
            fz.fstate1 = (uint32_t*)s->fz_rec[0].p; fz.fstate2 = (uint32_t*)s->fz_rec[1].p;
            fz.state = (unsigned long long*)s->fz_state.p;
            fz.ticket = (unsigned int*)s->fz_misc.p; fz.abort = (int*)s->fz_misc.p + 1; fz.totals = (unsigned long long*)s->fz_misc.p + 1;
            launch_fast<10, true, AQC_FUSE_WPBT, false, true>(c, s, cfg, st, accum_limit, &fz);
            s->fused = true;
        } else if (s->max_len <= 160) {
            if (cfg.paired) { if (cfg.barcode) launch_fast<10, true, 16, true>(c, s, cfg, st, accum_limit); else launch_fast<10, true, 16, false>(c, s, cfg, st, accum_limit); }
            else { if (cfg.barcode) launch_fast<10, false, 12, true>(c, s, cfg, st, accum_limit); else launch_fast<10, false, 12, false>(c, s, cfg, st, accum_limit); }
        } else if (s->max_len <= 256) {
            if (cfg.paired) { if (cfg.barcode) launch_fast<16, true, 12, true>(c, s, cfg, st, accum_limit); else launch_fast<16, true, 12, false>(c, s, cfg, st, accum_limit); }
            else { if (cfg.barcode) launch_fast<16, false, 11, true>(c, s, cfg, st, accum_limit); else launch_fast<16, false, 11, false>(c, s, cfg, st, accum_limit); }
        } else {
            // 257 .. 288 bases: 2x250 reads that still carry a barcode + verify prefix (BASELINE config 5: 267 bases)
            if (cfg.paired) { if (cfg.barcode) launch_fast<18, true, 12, true>(c, s, cfg, st, accum_limit); else launch_fast<18, true, 12, false>(c, s, cfg, st, accum_limit); }
            else { if (cfg.barcode) launch_fast<18, false, 10, true>(c, s, cfg, st, accum_limit); else launch_fast<18, false, 10, false>(c, s, cfg, st, accum_limit); }
        }
        hipLaunchKernelGGL(filter_overlap_list_kernel, dim3((unsigned)c->n_cu), dim3(BLOCK), 0, s->stream, s->view, cfg, c->circles,
                           (aqc_result*)s->results.p, st, accum_limit, (const uint32_t*)s->deferred.p,
                           (const unsigned int*)s->n_deferred.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(launch_event(*s, AQC_K_FILTER_OVERLAP, 1), s->stream));
    s->timed[AQC_K_FILTER_OVERLAP] = !s->collecting;
    s->ran = true;
    return 0;
}

static int ensure_kmer(aqc_ctx* c, QcDev& q) {
    if (q.kt.keys) return 0;
    HIP_TRY(hipMalloc((void**)&q.kt.keys, sizeof(unsigned long long) * KMER_CAP));
    HIP_TRY(hipMalloc((void**)&q.kt.counts, sizeof(unsigned long long) * KMER_CAP));
    HIP_TRY(hipMalloc((void**)&q.kt.order, sizeof(unsigned long long) * KMER_CAP));
    HIP_TRY(hipMemset(q.kt.keys, 0, sizeof(unsigned long long) * KMER_CAP));
    HIP_TRY(hipMemset(q.kt.counts, 0, sizeof(unsigned long long) * KMER_CAP));
    HIP_TRY(hipMemset(q.kt.order, 0xff, sizeof(unsigned long long) * KMER_CAP));
    q.kt.mask = KMER_CAP - 1;
    HIP_TRY(hipMalloc((void**)&q.kt.dense_count, sizeof(unsigned int) * DENSE_CAP));
    HIP_TRY(hipMalloc((void**)&q.kt.dense_first, sizeof(unsigned long long) * DENSE_CAP));
    HIP_TRY(hipMemset(q.kt.dense_count, 0, sizeof(unsigned int) * DENSE_CAP));
    HIP_TRY(hipMemset(q.kt.dense_first, 0xff, sizeof(unsigned long long) * DENSE_CAP));
    HIP_TRY(hipMalloc((void**)&q.kt.complete, sizeof(unsigned int) * (DENSE_ENTRIES / KRED_ENTRIES)));
    HIP_TRY(hipMemset(q.kt.complete, 0, sizeof(unsigned int) * (DENSE_ENTRIES / KRED_ENTRIES)));
    // the slot streams are non-blocking: make sure the fills have landed before any kernel can touch the tables
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int aqc_qc_stat(aqc_ctx* c, int slot, int which, int mate, uint64_t first, uint64_t count, int post) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (which < 0 || which > 3 || mate < 0 || mate > 1) return fail(AQC_ERR_ARG, "aqc_qc_stat: bad which/mate");
    if (!c->has_cfg) return fail(AQC_ERR_STATE, "aqc_qc_stat before aqc_set_config");
    if (first + count > s->n) return fail(AQC_ERR_ARG, "aqc_qc_stat: range exceeds the slot's %llu records", (unsigned long long)s->n);
    if (mate == 1 && !s->paired) return fail(AQC_ERR_ARG, "aqc_qc_stat: mate 1 of a single-end slot");
    if (post && !s->ran) return fail(AQC_ERR_STATE, "aqc_qc_stat(post) before aqc_run");
    if (count == 0) return 0;
    // one call at a time per context: the count -> reduce pairs below go through ONE slice buffer (kmer_partial) in stream order
    std::lock_guard<std::mutex> qc_lock(c->qc_mu);
    QcDev& q = c->qc[which];
    if ((rc = ensure_kmer(c, q))) return rc;
    // the statRead kernels go to the context's QC stream, behind everything queued on the slot's stream so far (text, results):
    // a few thousand latency-bound waves that overlap with the slot's bandwidth-bound kernels (the formatter) instead of
    // holding them up.  The slot is "in sync" again only when they are done too (slot_sync).
    // (AQC_QC_STREAM=0: on the slot's own stream, one kernel after the other — for profiles: beside the formatter a statRead kernel's
    //  start-to-end time is mostly the wait for free wave slots, e.g. 1.38 ms for a kernel whose waves live 0.06 ms)
    hipStream_t qs = c->qc_inline ? s->stream : c->qc_stream;
    HIP_TRY(hipEventRecord(s->ev_main, s->stream));
    if (!c->qc_inline) HIP_TRY(hipStreamWaitEvent(qs, s->ev_main, 0));
    HIP_TRY(hipEventRecord(launch_event(*s, AQC_K_QC_STAT, 0), qs));
    // LDS sized by the longest read of the slot: many resident workgroups for short reads
    const uint32_t mx = s->raw_max_len ? s->raw_max_len : AQC_MAX_READ_LEN;
    int cols = (int)((mx + 63) / 64 * 64);
    if (cols > AQC_QC_COLS) cols = AQC_QC_COLS;
    const size_t lds = sizeof(unsigned int) * (size_t)(QC_LDS_ROWS + 1) * cols + 16;
    // two 1024-thread workgroups per CU (every wave slot taken) and as few workgroups as that allows: each one ends
    // with ~11 global atomics per cycle; more only when a workgroup's packed counters would pass 4095 reads
    uint64_t blocks = (count + QC_WPB - 1) / QC_WPB;
    if (blocks > (uint64_t)c->n_cu * 2) blocks = (uint64_t)c->n_cu * 2;
    const uint64_t need = (count + (QC_MAX_READS_PER_BLOCK - QC_WPB) - 1) / (QC_MAX_READS_PER_BLOCK - QC_WPB);
    if (blocks < need) blocks = need;
    const unsigned long long g0 = s->view.first_index + first;
    if (g0 < q.last_end) q.epoch++;
    q.last_end = g0 + count;
    const unsigned long long order_base = (q.epoch << 34) | g0;
    // Reads of <= 256 bases: ONE kernel does both halves of statRead (per-cycle rows ride along with the k-mer
    // counting, see kmer_count_kernel); longer reads: the per-cycle kernel runs on its own.
    const uint32_t per_read = mx > (uint32_t)c->cfg.qc_kmer ? mx - (uint32_t)c->cfg.qc_kmer : 1;
    uint32_t rpr_max = 65535u / per_read;
    if (rpr_max < 1) rpr_max = 1;
    const uint64_t max_rounds = 512;                       // 64 MiB of slices at most per launch
    const uint64_t rounds_per_block = (max_rounds + c->n_cu - 1) / c->n_cu;
    const bool fused = cols <= KMER_FUSED_MAX_COLS && rounds_per_block * rpr_max <= (uint64_t)QC_MAX_READS_PER_BLOCK;
    // (fused: the reads whose quality line has a length of its own — a slot that has any: s->has_irregular — get their per-cycle rows
    //  from this kernel too, and only those; their k-mers are counted with everybody else's)
    if (!fused || s->has_irregular)
        hipLaunchKernelGGL(qc_stat_kernel, dim3((unsigned)blocks), dim3(QC_BLOCK), lds, qs, s->view, mate, first, count, post,
                           (const aqc_result*)s->results.p, c->cfg.qc_kmer, q.acc, s->status, cols, fused ? 1 : 0);
    // k-mer dictionary: LDS-resident u16 counters, rounds of <= 65535 k-mers per workgroup, slices reduced afterwards
    {
        uint64_t done = 0;
        while (done < count) {
            uint64_t chunk = count - done;
            if (chunk > max_rounds * rpr_max) chunk = max_rounds * rpr_max;
            // every workgroup the same number of rounds: round the count up to a multiple of the CU count
            uint64_t n_rounds64 = (chunk + rpr_max - 1) / rpr_max;
            if (n_rounds64 > (uint64_t)c->n_cu) {
                n_rounds64 = (n_rounds64 + c->n_cu - 1) / c->n_cu * c->n_cu;
                if (n_rounds64 > max_rounds) n_rounds64 = max_rounds;
            }
            const uint32_t rpr = (uint32_t)((chunk + n_rounds64 - 1) / n_rounds64);
            const uint32_t n_rounds = (uint32_t)((chunk + rpr - 1) / rpr);
            if (c->kmer_partial.reserve((size_t)n_rounds * DENSE_ENTRIES * sizeof(uint16_t))) return fail(AQC_ERR_HIP, "hipMalloc failed");
            unsigned kb = n_rounds < (unsigned)c->n_cu ? n_rounds : (unsigned)c->n_cu;
            hipLaunchKernelGGL(kmer_count_kernel, dim3(kb), dim3(KMER_BLOCK), fused ? KMER_FUSED_LDS_BYTES : KMER_LDS_BYTES, qs, s->view,
                               mate, first + done, chunk, post, (const aqc_result*)s->results.p, c->cfg.qc_kmer, q.kt, order_base + done,
                               (uint16_t*)c->kmer_partial.p, rpr, n_rounds, s->status, fused ? q.acc : (unsigned long long*)nullptr,
                               fused ? cols : 0);
            hipLaunchKernelGGL(kmer_reduce_kernel, dim3(DENSE_ENTRIES / KRED_ENTRIES), dim3(KRED_BLOCK), 0, qs,
                               (const uint16_t*)c->kmer_partial.p, n_rounds, q.kt, c->cfg.qc_kmer);
            done += chunk;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(launch_event(*s, AQC_K_QC_STAT, 1), qs));
    HIP_TRY(hipEventRecord(s->ev_qc, qs));
    s->qc.gen.fetch_add(1, std::memory_order_release);
    s->timed[AQC_K_QC_STAT] = !s->collecting;
    return 0;
}

// ---- text in / text out -----------------------------------------------------------------------------------------
struct FrameExtents { const aqc_text_extent* ext[2]; uint64_t n[2]; uint8_t last[2]; };
static int frame_impl(aqc_ctx* c, int slot, const aqc_text_chunk* ch, aqc_frame_info* info, bool resident, const FrameExtents* fx = nullptr) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!ch || !info || !ch->text1) return fail(AQC_ERR_ARG, "aqc_frame: null argument");
    const bool paired = ch->text2 != nullptr;
    const int nf = paired ? 2 : 1;
    const uint8_t* text[2] = {ch->text1, ch->text2};
    const uint64_t bytes[2] = {ch->bytes1, paired ? ch->bytes2 : 0};
    const int final_[2] = {ch->final1, ch->final2};
    for (int k = 0; k < nf; k++)
        if (bytes[k] >= (1ull << 31) - IDX_TILE) return fail(AQC_ERR_ARG, "aqc_frame: chunks must be < 2 GiB");
    HIP_TRY(slot_sync(*s));
    s->framed = s->formatted = false;
    s->ran = false;
    s->fused = false;
    DevBuf* arena[2] = {&s->seq1, &s->seq2};
    DevBuf* seq_off[2] = {&s->off1, &s->off2};
    DevBuf* qual_off[2] = {&s->qoff1, &s->qoff2};
    DevBuf* seq_len[2] = {&s->len1, &s->len2};
    // scratch: FrameMeta[2] | line totals[2] | tail values[4]
    if (s->t_scratch.reserve(256)) return fail(AQC_ERR_HIP, "hipMalloc failed");
    FrameMeta* d_meta = (FrameMeta*)s->t_scratch.p;
    unsigned long long* d_tot = (unsigned long long*)((uint8_t*)s->t_scratch.p + 64);
    // 1. text to the device; line index in one pass (text_index_kernel): both files in one launch.  The text sits
    //    TEXT_FRONT bytes into its buffer: the writer's 16-byte windows may start a few bytes before a piece's source.
    uint64_t tiles[2] = {0, 0}, cap[2] = {0, 0};
    uint8_t* tbase[2] = {nullptr, nullptr};
    for (int k = 0; k < nf; k++) {
        const size_t slack = IDX_TILE + 64;
        if (arena[k]->reserve(TEXT_FRONT + bytes[k] + slack)) return fail(AQC_ERR_HIP, "hipMalloc of %llu bytes failed", (unsigned long long)bytes[k]);
        tbase[k] = (uint8_t*)arena[k]->p + TEXT_FRONT;
        if (!resident && fx && fx->n[k]) {
            // parts of the chunk are in this device's memory already (aqc_frame_mixed): those move inside HBM, the rest comes up
            uint64_t cur = 0;
            for (uint64_t e = 0; e < fx->n[k]; ++e) {
                const aqc_text_extent& x = fx->ext[k][e];
                if (x.offset < cur || x.offset + x.bytes > bytes[k] || !x.device_text) return fail(AQC_ERR_ARG, "aqc_frame_mixed: extents must be sorted, disjoint and inside the chunk");
                if (x.offset > cur) HIP_TRY(hipMemcpyAsync(tbase[k] + cur, text[k] + cur, x.offset - cur, hipMemcpyHostToDevice, s->stream));
                if (x.bytes) HIP_TRY(hipMemcpyAsync(tbase[k] + x.offset, x.device_text, x.bytes, hipMemcpyDeviceToDevice, s->stream));
                cur = x.offset + x.bytes;
            }
            if (bytes[k] > cur) HIP_TRY(hipMemcpyAsync(tbase[k] + cur, text[k] + cur, bytes[k] - cur, hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipMemsetAsync(tbase[k] + bytes[k], 0, slack, s->stream));
            s->last_byte[k] = bytes[k] ? fx->last[k] : (uint8_t)'\n';
        } else if (!resident) {
            if (bytes[k]) HIP_TRY(hipMemcpyAsync(tbase[k], text[k], bytes[k], hipMemcpyHostToDevice, s->stream));
            HIP_TRY(hipMemsetAsync(tbase[k] + bytes[k], 0, slack, s->stream));
            s->last_byte[k] = bytes[k] ? text[k][bytes[k] - 1] : (uint8_t)'\n';
        }
        tiles[k] = bytes[k] ? (bytes[k] + IDX_TILE - 1) / IDX_TILE : 1;
        // FASTQ lines average ~90 bytes; a chunk with more lines than this guess is indexed again with the exact size
        const uint64_t guess = bytes[k] / 16 + 4096;
        cap[k] = s->t_line_end[k].cap / sizeof(uint32_t) > guess + 2 ? s->t_line_end[k].cap / sizeof(uint32_t) - 2 : guess;
        if (s->t_line_end[k].reserve(sizeof(uint32_t) * (cap[k] + 2))) return fail(AQC_ERR_HIP, "hipMalloc failed");
    }
    const uint64_t all_tiles = tiles[0] + (paired ? tiles[1] : 0);
    if (s->t_tile[0].reserve(sizeof(unsigned long long) * (all_tiles + 1))) return fail(AQC_ERR_HIP, "hipMalloc failed");
    // 2. ... the four lines of every complete group, the lock-step record count, the bytes consumed: all queued behind the index
    //    pass without asking the host for anything — the kernels read the line totals where the index pass left them, their grids
    //    are sized for the most lines the chunk could hold.  ONE copy back (FrameOut), ONE wait per chunk.
    const FrameMeta init{0xffffffffu, 0u, 0xffffffffu, 0u};
    FrameMeta h_meta[2] = {init, init};
    FrameOut fo{};
    FrameOut* d_out = (FrameOut*)((uint8_t*)s->t_scratch.p + 128);
    uint32_t virt[2] = {0, 0};
    for (int k = 0; k < nf; k++)      // an unterminated last line of the file is a line (readline() returns it); it may end in blanks
        if (final_[k] && bytes[k] > 0 && s->last_byte[k] != '\n') virt[k] = (uint32_t)bytes[k] | LINE_WS;
    const bool bubble = c->has_cfg && c->cfg.debubble;
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(hipMemsetAsync(s->t_tile[0].p, 0, sizeof(unsigned long long) * (all_tiles + 1), s->stream));
        IndexFile f[2] = {};
        uint32_t t0 = 0;
        for (int k = 0; k < nf; k++) {
            f[k] = IndexFile{(const uint8_t*)tbase[k], bytes[k], (uint32_t*)s->t_line_end[k].p, cap[k], d_tot + k, t0, (uint32_t)tiles[k]};
            t0 += (uint32_t)tiles[k];
        }
        hipLaunchKernelGGL(text_index_kernel, dim3((unsigned)all_tiles), dim3(TXT_BLOCK), 0, s->stream, f[0], f[1],
                           (unsigned long long*)s->t_tile[0].p, (unsigned int*)((unsigned long long*)s->t_tile[0].p + all_tiles));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(d_meta, h_meta, sizeof(h_meta), hipMemcpyHostToDevice, s->stream));
        uint64_t rec_cap = 0;
        for (int k = 0; k < nf; k++) {
            const uint64_t m = (cap[k] + 1) / 4 + 1;            // records the line table could describe
            rec_cap = std::max(rec_cap, m);
            if (seq_off[k]->reserve(4 * m) || qual_off[k]->reserve(4 * m) || seq_len[k]->reserve(4 * m) || s->t_name_off[k].reserve(4 * m) ||
                s->t_name_len[k].reserve(4 * m) || s->t_plus_off[k].reserve(4 * m) || s->t_plus_len[k].reserve(4 * m) ||
                s->t_qual_len[k].reserve(4 * m))
                return fail(AQC_ERR_HIP, "hipMalloc failed");
            FramedFile ff{(uint32_t*)seq_off[k]->p, (uint32_t*)qual_off[k]->p, (uint32_t*)seq_len[k]->p, (uint32_t*)s->t_name_off[k].p,
                          (uint32_t*)s->t_name_len[k].p, (uint32_t*)s->t_plus_off[k].p, (uint32_t*)s->t_plus_len[k].p,
                          (uint32_t*)s->t_qual_len[k].p};
            hipLaunchKernelGGL(frame_records_kernel, dim3((unsigned)((m + TXT_BLOCK - 1) / TXT_BLOCK)), dim3(TXT_BLOCK), 0, s->stream,
                               (const uint8_t*)tbase[k], (const uint32_t*)s->t_line_end[k].p, (const unsigned long long*)(d_tot + k), virt[k], ff, d_meta + k, (uint64_t)cap[k]);
        }
        hipLaunchKernelGGL(frame_finish_kernel, dim3(1), dim3(1), 0, s->stream, (const unsigned long long*)d_tot, (const FrameMeta*)d_meta,
                           (const uint32_t*)s->t_line_end[0].p, (const uint32_t*)(paired ? s->t_line_end[1].p : s->t_line_end[0].p), (const uint32_t*)s->len1.p,
                           virt[0], virt[1], (unsigned long long)bytes[0], (unsigned long long)bytes[1], nf, (unsigned long long)ch->max_records, d_out,
                           (unsigned long long)cap[0], (unsigned long long)cap[1]);
        if (bubble) {
            // lane / tile / x / y out of the R1 names (preprocesser.py:180-192) for the bubble filter
            for (int k = 0; k < 5; k++)
                if (s->aux[k].reserve((k < 4 ? sizeof(int32_t) : 1) * rec_cap)) return fail(AQC_ERR_HIP, "hipMalloc failed");
            hipLaunchKernelGGL(parse_names_kernel, dim3((unsigned)((rec_cap + TXT_BLOCK - 1) / TXT_BLOCK)), dim3(TXT_BLOCK), 0, s->stream,
                               (const uint8_t*)tbase[0], (const uint32_t*)s->t_name_off[0].p, (const uint32_t*)s->t_name_len[0].p, (const unsigned long long*)&d_out->n,
                               (int32_t*)s->aux[0].p, (int32_t*)s->aux[1].p, (int32_t*)s->aux[2].p, (int32_t*)s->aux[3].p,
                               (uint8_t*)s->aux[4].p);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&fo, d_out, sizeof(fo), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(slot_sync(*s));
        // FASTQ lines average ~90 bytes; a chunk with more lines than the table was sized for (counted, not written) is done again
        bool fits = true;
        for (int k = 0; k < nf; k++) {
            const uint64_t real = fo.lines[k] - (virt[k] ? 1 : 0);
            if (real > cap[k]) {
                fits = false;
                cap[k] = real;
                if (s->t_line_end[k].reserve(sizeof(uint32_t) * (cap[k] + 2))) return fail(AQC_ERR_HIP, "hipMalloc failed");
            }
        }
        if (fits) break;
    }
    // 3. lock-step record count (preprocesser.py:412-429)
    // (a record whose quality line is not as long as its sequence line is a record like any other: fastq.py:37-49 does not look,
    //  and every later stage keeps a view per string — LEN_IRR in aqc_kernels.hpp)
    const uint64_t n = fo.n;
    memset(info, 0, sizeof(*info));
    info->n = n;
    info->avail1 = fo.avail[0];
    info->avail2 = fo.avail[1];
    info->eof1 = (int32_t)fo.eof[0];
    info->eof2 = paired ? (int32_t)fo.eof[1] : 0;
    info->max_len = fo.max_len;
    // 4. slot view: the text IS the arena, every kernel reads the records in place
    DevBatch v{};
    v.n = n;
    v.first_index = ch->first_index;
    v.seq1 = v.qual1 = (const uint8_t*)tbase[0];
    v.off1 = (const uint32_t*)s->off1.p; v.qoff1 = (const uint32_t*)s->qoff1.p; v.len1 = (const uint32_t*)s->len1.p;
    if (paired) {
        v.seq2 = v.qual2 = (const uint8_t*)tbase[1];
        v.off2 = (const uint32_t*)s->off2.p; v.qoff2 = (const uint32_t*)s->qoff2.p; v.len2 = (const uint32_t*)s->len2.p;
    }
    if (s->results.reserve(sizeof(aqc_result) * (n ? n : 1))) return fail(AQC_ERR_HIP, "hipMalloc failed");
    {
        // the quality lines' own lengths (frame_records_kernel) and room for the final quality views of the marked records
        // (written by the verdict kernels for those records only: no traffic for a regular chunk)
        const bool any_irr = fo.first_mismatch[0] < n || (paired && fo.first_mismatch[1] < n);
        s->has_irregular = any_irr;
        for (int k = 0; k < nf; k++)
            if (any_irr && s->qview[k].reserve(sizeof(uint32_t) * (n ? n : 1))) return fail(AQC_ERR_HIP, "hipMalloc failed");
        v.qlen1 = (const uint32_t*)s->t_qual_len[0].p; v.qview1 = (uint32_t*)s->qview[0].p;
        v.qlen2 = paired ? (const uint32_t*)s->t_qual_len[1].p : v.qlen1; v.qview2 = paired ? (uint32_t*)s->qview[1].p : v.qview1;
    }
    if (bubble) {
        v.aux_lane = (const int32_t*)s->aux[0].p; v.aux_tile = (const int32_t*)s->aux[1].p;
        v.aux_x = (const int32_t*)s->aux[2].p; v.aux_y = (const int32_t*)s->aux[3].p; v.aux_ok = (const uint8_t*)s->aux[4].p;
    }
    s->view = v;
    s->n = n;
    s->paired = paired;
    s->raw_max_len = info->max_len;
    s->max_len = info->max_len;
    // 5. bytes consumed by the n records (+ R1's next sequence length for the TOTAL_BASES quirk)
    const uint64_t consumed[2] = {fo.consumed[0], fo.consumed[1]};
    const uint32_t h_next = fo.next_len1;
    info->consumed1 = consumed[0];
    info->consumed2 = consumed[1];
    s->consumed[0] = consumed[0]; s->consumed[1] = consumed[1];
    info->next_len1 = h_next;
    s->framed = true;
    s->last_chunk = *ch;
    return 0;
}

int aqc_frame(aqc_ctx* c, int slot, const aqc_text_chunk* ch, aqc_frame_info* info) { return frame_impl(c, slot, ch, info, false); }

int aqc_frame_mixed(aqc_ctx* c, int slot, const aqc_text_chunk* ch, const aqc_text_extent* ext1, uint64_t n_ext1, uint8_t last1,
                    const aqc_text_extent* ext2, uint64_t n_ext2, uint8_t last2, aqc_frame_info* info) {
    if ((n_ext1 && !ext1) || (n_ext2 && !ext2)) return fail(AQC_ERR_ARG, "aqc_frame_mixed: null extent list");
    const FrameExtents fx{{ext1, ext2}, {n_ext1, ch && ch->text2 ? n_ext2 : 0}, {last1, last2}};
    return frame_impl(c, slot, ch, info, false, &fx);
}

int aqc_reframe(aqc_ctx* c, int slot, aqc_frame_info* info) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->framed) return fail(AQC_ERR_STATE, "aqc_reframe needs a slot filled by aqc_frame");
    const aqc_text_chunk ch = s->last_chunk;
    return frame_impl(c, slot, &ch, info, true);
}

static int format_impl(aqc_ctx* c, int slot, int verdict_slot, uint64_t n, int32_t store_overlap, uint64_t bytes_out[6], bool spans = false) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    const bool plain = verdict_slot != slot;
    Slot* vs = s;
    if (plain && (rc = get_slot(c, verdict_slot, &vs))) return rc;
    if (!bytes_out) return fail(AQC_ERR_ARG, "aqc_format: null argument");
    if (!s->framed) return fail(AQC_ERR_STATE, "aqc_format needs a slot filled by aqc_frame");
    if (!vs->ran) return fail(AQC_ERR_STATE, "aqc_format before aqc_run");
    if (n > s->n || n > vs->n) return fail(AQC_ERR_ARG, "aqc_format: n exceeds the slot's records");
    if (plain) HIP_TRY(slot_sync(*vs));      // the verdicts come from another slot's stream
    FormatView v{};
    v.paired = s->paired ? 1 : 0;
    v.results = (const aqc_result*)vs->results.p;
    v.plain = plain ? 1 : 0;
    v.verdict_paired = vs->paired ? 1 : 0;
    v.barcode = c->cfg.barcode ? 1 : 0;
    v.barcode_length = c->cfg.barcode_length;
    v.store_overlap = (store_overlap && vs->paired) ? 1 : 0;
    v.spans = (spans && !plain) ? 1 : 0;
    v.consumed[0] = (uint32_t)s->consumed[0]; v.consumed[1] = (uint32_t)s->consumed[1];
    v.n_framed = s->n;
    s->n_events[0] = s->n_events[1] = 0;
    const DevBuf* sl[2] = {&s->len1, &s->len2};
    const DevBuf* arena[2] = {&s->seq1, &s->seq2};
    const DevBuf* so[2] = {&s->off1, &s->off2};
    const DevBuf* qo[2] = {&s->qoff1, &s->qoff2};
    for (int k = 0; k < (s->paired ? 2 : 1); k++) {
        v.f[k].text = (const uint8_t*)arena[k]->p + TEXT_FRONT;
        v.f[k].seq_off = (const uint32_t*)so[k]->p;
        v.f[k].qual_off = (const uint32_t*)qo[k]->p;
        v.f[k].seq_len = (const uint32_t*)sl[k]->p;
        v.f[k].name_off = (const uint32_t*)s->t_name_off[k].p;
        v.f[k].name_len = (const uint32_t*)s->t_name_len[k].p;
        v.f[k].plus_off = (const uint32_t*)s->t_plus_off[k].p;
        v.f[k].plus_len = (const uint32_t*)s->t_plus_len[k].p;
        v.f[k].qual_len = (const uint32_t*)s->t_qual_len[k].p;
        v.f[k].qview = (const uint32_t*)s->qview[k].p;
    }
    // streams q = file * 3 + {0 good, 1 bad, 2 overlap}: per-tile byte sums -> tile bases (one launch each), the
    // per-record offsets are formed inside the writer
    const uint64_t n_tiles = n ? (n + FMT_TILE - 1) / FMT_TILE : 1;
    const uint64_t n_super = (n_tiles + FMT_SUPER - 1) / FMT_SUPER;
    bool live[6];
    for (int q = 0; q < 6; q++) live[q] = (q < 3 || s->paired) && (q % 3 != 2 || v.store_overlap);
    unsigned long long h_tot[FMT_STREAMS] = {0, 0, 0, 0, 0, 0, 0, 0};
    // the verdict kernel may have done the placement already (AQC_FUSED=1, all n records of the slot, the two-stream case): its totals
    // stand in for the sums / bases passes — unless it gave the placement up (a deferred pair, a record that is not plain text)
    if (s->fused && !plain && !spans && !v.store_overlap && n == s->n && n > 0) {
        unsigned long long misc[5];
        HIP_TRY(hipMemcpyAsync(misc, s->fz_misc.p, sizeof(misc), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        if ((misc[0] >> 32) == 0) {
            v.fused = 1;
            v.fstate[0] = (const uint32_t*)s->fz_rec[0].p; v.fstate[1] = (const uint32_t*)s->fz_rec[1].p;
            v.fbatch = (const unsigned long long*)s->fz_state.p;
            v.fbatch_shift = 5;
            static_assert(FastWaveLds<10, true, true>::PPW == 32, "fbatch_shift");
            h_tot[0] = misc[1]; h_tot[3] = misc[2]; h_tot[1] = misc[3]; h_tot[4] = misc[4];
        }
    }
    if (!v.fused) {
        s->fused = false;          // (whatever this call writes into the good streams replaces what the verdict kernel left there)
        // f_tile: [FMT_STREAMS x n_tiles] the tiles' prefixes inside their super-tiles | [FMT_STREAMS x n_super] the super-tiles' sums -> bases
        if (s->f_tile.reserve(sizeof(unsigned long long) * FMT_STREAMS * (n_tiles + n_super)) || s->t_scratch.reserve(256))
            return fail(AQC_ERR_HIP, "hipMalloc failed");
        unsigned long long* d_tot = (unsigned long long*)((uint8_t*)s->t_scratch.p + 128);
        unsigned long long* d_super = (unsigned long long*)s->f_tile.p + FMT_STREAMS * n_tiles;
        if (n) hipLaunchKernelGGL(fmt_tile_sums_kernel, dim3((unsigned)n_super), dim3(TXT_BLOCK), 0, s->stream, v, n, n_tiles, n_super, (unsigned long long*)s->f_tile.p, d_super);
        else HIP_TRY(hipMemsetAsync(s->f_tile.p, 0, sizeof(unsigned long long) * FMT_STREAMS * (n_tiles + n_super), s->stream));
        hipLaunchKernelGGL(fmt_tile_bases_kernel, dim3(v.spans ? FMT_STREAMS : 6), dim3(TXT_BLOCK), 0, s->stream, d_super, n_super, d_tot);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_tot, d_tot, sizeof(unsigned long long) * (v.spans ? FMT_STREAMS : 6), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    if (v.spans) {
        for (int f = 0; f < (s->paired ? 2 : 1); ++f) {
            s->n_events[f] = h_tot[FMT_EVENT_STREAM + f];
            if (s->f_events[f].reserve(sizeof(SpanEvent) * (s->n_events[f] + 1))) return fail(AQC_ERR_HIP, "hipMalloc failed");
        }
    }
    FormatOut outs{};
    for (int q = 0; q < 6; q++) {
        s->f_bytes[q] = live[q] ? h_tot[q] : 0;
        bytes_out[q] = s->f_bytes[q];
        if (s->f_out[q].reserve(s->f_bytes[q] + 64)) return fail(AQC_ERR_HIP, "hipMalloc failed");
        outs.p[q] = (uint8_t*)s->f_out[q].p;
    }
    if (n) {
        const uint64_t n_tasks = n * (s->paired ? 2 : 1);
        // 48-byte plans, (sparse) full piece lists for the records that do not fit a plan, and the list of the records the
        // general copy kernel takes (+ its length)
        const uint64_t gen_cap = ((n_tiles + GEN_LISTS - 1) / GEN_LISTS) * FMT_TILE * (s->paired ? 2 : 1);     // worst case: every record
        // plans: one 16-byte word per (record, file), dense; the six words of the records the general kernel takes, in list order
        const uint64_t plan0_bytes = (16 * n_tasks + 255) / 256 * 256;
        if (s->f_plan.reserve(plan0_bytes + 16 * PLAN_Q * gen_cap * GEN_LISTS) || s->f_patch.reserve(16 * n_tasks + 32 * gen_cap * GEN_LISTS) || s->f_over.reserve(sizeof(FmtTask) * n_tasks) ||
            s->f_pos.reserve(4 * gen_cap * GEN_LISTS + 2 * sizeof(unsigned int) * GEN_LISTS + 64))
            return fail(AQC_ERR_HIP, "hipMalloc failed");
        // f_pos: the general kernel's lists | the lengths of those and of the lists of one-piece plans of a spans / fused format;
        // f_patch: the patch words of the dense plan0 | those listed plans (two words each, in list order)
        uint4* d_wplan = (uint4*)((uint8_t*)s->f_patch.p + 16 * n_tasks);
        unsigned int* d_ngen = (unsigned int*)((uint8_t*)s->f_pos.p + 4 * gen_cap * GEN_LISTS);
        unsigned int* d_nwhole = d_ngen + GEN_LISTS;
        const bool sparse = v.spans || v.fused;
        unsigned copy_blocks = (unsigned)((n_tasks + (COPY_BLOCK / 32) * FMT_UNROLL - 1) / ((COPY_BLOCK / 32) * FMT_UNROLL));
#ifdef AQC_COPY_PERSIST
        if (copy_blocks > (unsigned)c->n_cu * AQC_COPY_PERSIST) copy_blocks = (unsigned)c->n_cu * AQC_COPY_PERSIST;
#endif
        for (int pass = 0; pass < (v.store_overlap ? 2 : 1); ++pass) {
            HIP_TRY(hipMemsetAsync(d_ngen, 0, 2 * sizeof(unsigned int) * GEN_LISTS, s->stream));
            // GEN_LISTS x k workgroups; k from the worst case, at most 32 per list
            uint64_t per_list = (gen_cap + GEN_ROUND - 1) / GEN_ROUND;
            if (per_list > 32) per_list = 32;
            if (per_list < 1) per_list = 1;
            // text mode without barcodes, main pass (round 6): place + copy in one kernel, piece lists only for the listed records
            // (AQC_PLACE_COPY=0: the plan / whole-copy pair of rounds 2 - 5, for A/B measurements)
            static const bool place_copy = [] { const char* e = getenv("AQC_PLACE_COPY"); return !(e && e[0] == '0'); }();
            if (place_copy && !sparse && pass == 0 && !v.plain && !v.barcode) {
                const unsigned long long* tb = (const unsigned long long*)s->f_tile.p;
                uint4* const pg = (uint4*)((uint8_t*)s->f_plan.p + plan0_bytes);
                hipLaunchKernelGGL(fmt_place_copy_kernel, dim3((unsigned)n_tiles), dim3(PC_BLOCK), 0, s->stream, v, n, n_tiles, n_super, tb, tb + FMT_STREAMS * n_tiles,
                                   pg, (uint32_t*)s->f_pos.p, d_ngen, gen_cap, outs);
                hipLaunchKernelGGL(fmt_plan_listed_kernel, dim3((unsigned)(GEN_LISTS * per_list)), dim3(FMT_TILE), 0, s->stream, v, pg, (FmtTask*)s->f_over.p,
                                   (const uint32_t*)s->f_pos.p, (const unsigned int*)d_ngen, gen_cap, s->status);
            } else {
            hipLaunchKernelGGL(fmt_plan_kernel, dim3((unsigned)n_tiles), dim3(FMT_TILE), 0, s->stream, v, n, n_tiles, n_super,
                               (const unsigned long long*)s->f_tile.p, (const unsigned long long*)s->f_tile.p + FMT_STREAMS * n_tiles, pass, s->status, (uint4*)s->f_plan.p, (uint4*)s->f_patch.p,
                               (uint4*)((uint8_t*)s->f_plan.p + plan0_bytes), (FmtTask*)s->f_over.p, (uint32_t*)s->f_pos.p, d_ngen, gen_cap, d_wplan, d_nwhole, outs.p[0], outs.p[3],
                               (SpanEvent*)s->f_events[0].p, (SpanEvent*)s->f_events[1].p);
            // (spans / fused mode: what stays in the caller's chunk / what the verdict kernel copied has no plan; the records that are their
            //  own bytes but for the walk's byte patches are still this kernel's)
            // (a barcode run has no one-piece record: fmt_plan_kernel writes no dense plans and nothing walks them)
            if (!sparse) {
                if (!(v.barcode && !v.plain)) hipLaunchKernelGGL(fmt_copy_whole_kernel, dim3(copy_blocks), dim3(COPY_BLOCK), 0, s->stream, v, n_tasks, (const uint4*)s->f_plan.p,
                                                                 (const uint4*)s->f_patch.p, outs);
            } else hipLaunchKernelGGL(fmt_copy_whole_list_kernel, dim3((unsigned)(GEN_LISTS * per_list)), dim3(COPY_BLOCK), 0, s->stream, v, (const uint4*)d_wplan, outs,
                                    (const unsigned int*)d_nwhole, gen_cap);
            }
            hipLaunchKernelGGL(fmt_copy_kernel, dim3((unsigned)(GEN_LISTS * per_list)), dim3(COPY_BLOCK), 0, s->stream, v,
                               (const uint4*)((uint8_t*)s->f_plan.p + plan0_bytes), (const FmtTask*)s->f_over.p, outs, (const uint32_t*)s->f_pos.p,
                               (const unsigned int*)d_ngen, gen_cap);
        }
        HIP_TRY(hipGetLastError());
    }
    s->formatted = true;
    s->formatted_fused = v.fused != 0;
    s->compressed = false;
    return 0;
}

// ---- gzip output on the device (aqc_gzdev.hpp) --------------------------------------------------------------------------------
static int ensure_gz_tables(aqc_ctx* c) {
    if (c->gz_crc.p) return 0;
    GzCrcTables t;
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t v = i;
        for (int k = 0; k < 8; ++k) v = (v >> 1) ^ (0xEDB88320u & (0u - (v & 1u)));
        t.byte_table[i] = v;
    }
    // "advance the CRC register by n zero bytes" is linear: column j is what zlib's crc32_combine makes of the unit vector
    for (int k = 0; k < 8; ++k)
        for (int j = 0; j < 32; ++j) t.shift[k][j] = (uint32_t)crc32_combine((uLong)(1u << j), 0UL, (z_off_t)(GZ_SEG << k));
    if (c->gz_crc.reserve(sizeof(t))) return fail(AQC_ERR_HIP, "hipMalloc failed");
    HIP_TRY(hipMemcpy(c->gz_crc.p, &t, sizeof(t), hipMemcpyHostToDevice));
    return 0;
}

int aqc_compress(aqc_ctx* c, int slot, int32_t level, uint64_t gz_bytes_out[6]) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!gz_bytes_out) return fail(AQC_ERR_ARG, "aqc_compress: null argument");
    if (!s->formatted) return fail(AQC_ERR_STATE, "aqc_compress before aqc_format");
    if (level < 1) return fail(AQC_ERR_UNSUPPORTED, "aqc_compress: level %d (stored output is the host writer's business)", level);
    if ((rc = ensure_gz_tables(c))) return rc;
    static_assert(sizeof(GzCodebookDev) == sizeof(aqcgz::GzCodebook), "host and device codebook layouts must agree");
    GzJob J{};
    uint32_t n_members = 0;
    // members of 64 x 255 bytes, a wave each (round 6: gz_encode_wave_kernel); AQC_GZ_ENCODER=seg: members of 256 x 255 bytes, a
    // thread per 255-byte segment (gz_encode_kernel, rounds 3 - 5)
    static const bool wave_enc = [] { const char* e = getenv("AQC_GZ_ENCODER"); return !(e && e[0] == 's'); }();
    J.member_text = wave_enc ? (uint32_t)GZW_TEXT : (uint32_t)GZ_TEXT;
    J.slot_bytes = wave_enc ? (uint32_t)GZW_SLOT : (uint32_t)GZ_SLOT;
    for (int q = 0; q < 6; ++q) {
        J.text[q] = (const uint8_t*)s->f_out[q].p;
        J.bytes[q] = s->f_bytes[q];
        J.first_block[q] = n_members;
        n_members += (uint32_t)((s->f_bytes[q] + J.member_text - 1) / J.member_text);
        s->g_bytes[q] = 0;
        gz_bytes_out[q] = 0;
    }
    J.first_block[6] = n_members;
    // (`compressed` is set once the streams exist: an error on the way must not let aqc_fetch_gz hand out empty streams)
    if (n_members == 0) { s->compressed = true; return 0; }
    if (s->g_stage.reserve((size_t)n_members * J.slot_bytes) || s->g_sizes.reserve(4 * (size_t)n_members) || s->g_offsets.reserve(8 * (size_t)n_members) ||
        s->g_total.reserve(64) || s->g_hist.reserve(6 * 320 * 4) || s->g_code.reserve(6 * sizeof(GzCodebookDev)))
        return fail(AQC_ERR_HIP, "hipMalloc failed");
    for (int q = 0; q < 6; ++q) {
        const uint64_t nb = J.first_block[q + 1] - J.first_block[q];
        if (s->g_packed[q].reserve(nb * (J.member_text + 31) + 64)) return fail(AQC_ERR_HIP, "hipMalloc failed");
        J.packed[q] = (uint8_t*)s->g_packed[q].p;
    }
    J.stage = (uint8_t*)s->g_stage.p; J.sizes = (uint32_t*)s->g_sizes.p; J.offsets = (uint64_t*)s->g_offsets.p; J.total = (uint64_t*)s->g_total.p;
    J.hist = (uint32_t*)s->g_hist.p; J.code = (const GzCodebookDev*)s->g_code.p; J.crc = (const GzCrcTables*)c->gz_crc.p;
    // 1. symbol counts of a sample of every stream's members
    HIP_TRY(hipMemsetAsync(s->g_hist.p, 0, 6 * 320 * 4, s->stream));
    hipLaunchKernelGGL(gz_hist_kernel, dim3(6 * GZ_SAMPLES), dim3(GZ_THREADS), 0, s->stream, J);
    HIP_TRY(hipGetLastError());
    uint32_t h[6][320];
    HIP_TRY(hipMemcpyAsync(h, s->g_hist.p, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    // 2. one code per stream, built on the host with the routines of its own encoder
    std::vector<aqcgz::GzCodebook> cb(6);
    for (int q = 0; q < 6; ++q)
        if (!aqcgz::build_codebook(h[q], h[q] + 286, &cb[q])) return fail(AQC_ERR_STATE, "aqc_compress: could not build a Huffman code");
    HIP_TRY(hipMemcpyAsync(s->g_code.p, cb.data(), 6 * sizeof(aqcgz::GzCodebook), hipMemcpyHostToDevice, s->stream));
    // 3. members, their places, the contiguous streams
    if (wave_enc) hipLaunchKernelGGL(gz_encode_wave_kernel, dim3(n_members), dim3(WAVE), 0, s->stream, J);
    else hipLaunchKernelGGL(gz_encode_kernel, dim3(n_members), dim3(GZ_THREADS), 0, s->stream, J);
    hipLaunchKernelGGL(gz_offsets_kernel, dim3(6), dim3(GZ_THREADS), 0, s->stream, J);
    hipLaunchKernelGGL(gz_pack_kernel, dim3(n_members), dim3(GZ_THREADS), 0, s->stream, J);
    HIP_TRY(hipGetLastError());
    unsigned long long tot[6];
    HIP_TRY(hipMemcpyAsync(tot, s->g_total.p, sizeof(tot), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));      // (cb and tot live on this stack frame)
    for (int q = 0; q < 6; ++q) {
        s->g_bytes[q] = tot[q];
        gz_bytes_out[q] = tot[q];
    }
    rc = check_status(*s);
    s->compressed = rc == 0;
    return rc;
}

int aqc_fetch_gz(aqc_ctx* c, int slot, int file, int stream, uint8_t* dst, uint64_t cap) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->compressed) return fail(AQC_ERR_STATE, "aqc_fetch_gz before aqc_compress");
    if (file < 0 || file > 1 || stream < 0 || stream > 2) return fail(AQC_ERR_ARG, "aqc_fetch_gz: bad file/stream");
    const int q = file * 3 + stream;
    if (s->g_bytes[q] > cap) return fail(AQC_ERR_ARG, "aqc_fetch_gz: %llu bytes do not fit %llu", (unsigned long long)s->g_bytes[q], (unsigned long long)cap);
    if (s->g_bytes[q]) {
        if (!dst) return fail(AQC_ERR_ARG, "aqc_fetch_gz: null destination");
        HIP_TRY(hipMemcpyAsync(dst, s->g_packed[q].p, s->g_bytes[q], hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return check_status(*s);
}

// ---- gzip input on the device (aqc_gunzip_dev.hpp) ------------------------------------------------------------------------------
// DeviceInflate: the SectionOffload of aqc_gz.hpp.  A group of consecutive sections = one window of the compressed file = one
// pass of scan -> compact -> decode -> chain -> gather.
//
// Round 6.  (1) What a group needs on the device is sized by NEED, not by the worst case: symbols for 6 x its compressed bytes
// (FASTQ expands 3 - 5 x; rounds 4 - 5: 12 x), token entries for 1 per compressed byte + 512 per lane (FASTQ: 0.5 - 0.6 per byte;
// they used to share the symbols' 12 x), one set of decode buffers per decoder instead of one per lane — 2.3 GB for groups of 62 MiB where there were
// two lanes of 9.5 - 10 GB — and they are allocated in the BACKGROUND when the stream announces its group size (prepare()):
// ready() stays false until they exist, so the pool keeps every section until then and nobody waits for a hipMalloc (16 ms per
// GB).  A group that overflows the lean budget comes back short, the host decodes what is missing, and the budgets double for the
// groups after it.  (2) The two lane threads share the decode buffers: one copies its group's compressed bytes out of the file
// mapping into its page-locked stage while the other's kernels run.  (3) RESIDENT results (the default): the sections' symbols
// stay in HBM, in a result set the sections hold until they are dropped; when the consumer arrives with the window before a run
// of them, resolve() turns the symbols into text and computes the CRC-32 of every section there (gzb_windows_kernel,
// gzb_resolve_kernel, gzb_crc_kernel) and fetch() copies text straight to where the consumer wants it.  PCIe carries one byte per
// byte of text instead of two, and the host's 2.2 CPU-seconds per 10 M reads of marker translation + CRC-32 are gone.
// AQC_GZ_RESIDENT=0: the symbols come back into page-locked arenas and the host translates them, as in rounds 4 - 5.
namespace {

std::atomic<uint64_t> g_gzb_stats[8];
std::atomic<uint64_t> g_gzb_resolve_stats[4];       // runs resolved, their sections, microseconds in resolve(), bytes of text resolved

constexpr size_t GZB_SLACK = 4u << 20;              // compressed bytes uploaded behind the last section's stop bit (its last block ends there)

class DeviceInflate : public aqcgz::SectionOffload {
public:
    DeviceInflate(int device, size_t group_bytes) : device_(device), group_bytes_(std::min<size_t>(std::max<size_t>(group_bytes, 1u << 20), 448u << 20)) {
        if (const char* e = getenv("AQC_GZ_RESIDENT")) resident_ = e[0] != '0';
        if (const char* e = getenv("AQC_GZ_RATIO")) ratio_ = (uint32_t)std::max(2, std::min(64, atoi(e)));
        if (const char* e = getenv("AQC_GZ_TOK_RATIO")) tok_ratio_ = (uint32_t)std::max(1, std::min(16, atoi(e)));
    }
    ~DeviceInflate() override {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& l : lanes_) if (l.th.joinable()) l.th.join();
        (void)hipSetDevice(device_);
        const double t0 = now_s();
        if (dev_.stream) (void)hipStreamSynchronize(dev_.stream);
        if (rs_stream_) (void)hipStreamSynchronize(rs_stream_);
        dev_.release();
        for (auto& r : res_) { r.sym.release(); r.text.release(); }
        rs_wins_.release(); rs_tab_.release(); rs_crc_tab_.release();
        if (rs_stream_) (void)hipStreamDestroy(rs_stream_);
        for (auto& l : lanes_) if (l.stage) aqc_host_free(l.stage);
        if (rs_pin_) aqc_host_free(rs_pin_);
        const double t1 = now_s();
        for (auto& a : arenas_) if (a.p) aqc_host_free(a.p);
        if (debug()) fprintf(stderr, "[gz dev %d] tear-down: device buffers + stages %.3f s, arenas %.3f s\n", device_, t1 - t0, now_s() - t1);
    }
    bool start() {
        if (hipSetDevice(device_) != hipSuccess) { (void)hipGetLastError(); return false; }
        // (the lowest stream priority: where a decoder slice and a kernel of the filter compete for the chip, the filter goes first;
        //  the resolve stream, which the consumer WAITS for, gets the highest)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        int prio_dec = prio_lo;
        if (const char* e = getenv("AQC_GZ_PRIO")) { if (e[0] == '0') prio_dec = 0; else if (e[0] == '2') prio_dec = prio_hi; }      // (experiments: 0 normal, 2 highest)
        if (hipStreamCreateWithPriority(&dev_.stream, hipStreamNonBlocking, prio_dec) != hipSuccess) return false;
        for (auto& e : dev_.ev) if (hipEventCreate(&e) != hipSuccess) return false;
        if (hipStreamCreateWithPriority(&rs_stream_, hipStreamNonBlocking, prio_hi) != hipSuccess) return false;
        for (int i = 0; i < N_LANES; ++i) lanes_[i].th = std::thread([this, i] { loop(i); });
        return true;
    }
    size_t group_bytes() const override { return group_bytes_; }
    bool ready() override {
        std::lock_guard<std::mutex> g(mu_);
        if (broken_ || stop_ || !prepared_) return false;
        for (auto& l : lanes_) if (!l.job) return true;
        return false;
    }
    bool gave_up() override {
        std::lock_guard<std::mutex> g(mu_);
        return broken_ || stop_;
    }
    // the stream's groups will be about this big: the decode buffers are set up by a lane thread, ready() is false until they are
    void prepare(size_t group_bytes) override {
        {
            std::lock_guard<std::mutex> g(mu_);
            const size_t want = std::min(group_bytes, group_bytes_);
            if (prepared_ && want <= prepared_for_) return;
            if (want > prepare_want_) prepare_want_ = want;
            // (a decoder that has worked before takes groups at once and grows its buffers on the way, as it always did)
            if (prepared_for_ == 0 && !prepare_busy_) prepared_ = false;
        }
        cv_.notify_all();
    }
    bool submit(const uint8_t* data, size_t size, int n, const uint64_t* nominal, const uint64_t* stop, const uint8_t* exact,
                std::function<void(int, const aqcgz::OffloadResult&)> done) override {
        if (n <= 0) return false;
        std::unique_ptr<Group> gr(new Group());
        gr->data = data; gr->size = size; gr->n = n;
        gr->nominal.assign(nominal, nominal + n); gr->stop.assign(stop, stop + n); gr->exact.assign(exact, exact + n);
        gr->done = std::move(done);
        // one window: from the first section's nominal start to the last one's stop bit (+ slack); bit positions are 32-bit inside it
        const uint64_t byte0 = (nominal[0] >> 3) & ~(uint64_t)15;
        const uint64_t last = (stop[n - 1] >> 3) + 1;
        if (stop[n - 1] == UINT64_MAX || last <= byte0 || last - byte0 + GZB_SLACK >= (500u << 20)) return false;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (broken_ || stop_) return false;
            Lane* pick = nullptr;
            for (auto& l : lanes_) if (!l.job) { pick = &l; break; }
            if (!pick) return false;
            pick->job = std::move(gr);
        }
        cv_.notify_all();
        return true;
    }
    void release(void* token) override {
        Token* t = (Token*)token;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (t->res >= 0) res_[t->res].refs--;
            else arenas_[t->arena].refs--;
        }
        cv_.notify_all();
        delete t;
    }

    // ---- resident results: the consumer has arrived with the window before a run of this decoder's sections ----------------------
    const uint8_t* text_ptr(void* token, int* device) override {
        const Token* t = (const Token*)token;
        if (t->res < 0) return nullptr;
        Res& R = res_[t->res];
        if (t->k < 0 || (size_t)t->k >= R.nsym.size()) return nullptr;
        if (device) *device = device_;
        return (const uint8_t*)R.text.p + R.off[(size_t)t->k];
    }
    int resolve(void* const* tokens, int n, const uint8_t* win, size_t wlen, uint32_t* crc, uint8_t* tail, size_t* tail_len, uint32_t* piece_nl) override {
        if (n <= 0 || wlen > GZB_WINDOW) return -2;
        const double t0 = now_s();
        std::lock_guard<std::mutex> rg(rs_mu_);
        if (hipSetDevice(device_) != hipSuccess) { (void)hipGetLastError(); return -2; }
        const Token* t0k = (const Token*)tokens[0];
        if (t0k->res < 0) return -2;
        Res& R = res_[t0k->res];
        // the run's table: first symbol and length of each section, then the CRC pieces (64 KiB, right-aligned in their section)
        uint32_t n_pieces = 0, max_n = 0;
        uint64_t total = 0;
        for (int k = 0; k < n; ++k) {
            const Token* t = (const Token*)tokens[k];
            if (t->res != t0k->res || t->k < 0 || (size_t)t->k >= R.nsym.size()) return -2;
            const uint32_t ns = R.nsym[(size_t)t->k];
            n_pieces += (ns + GZB_CRC_PIECE - 1u) / GZB_CRC_PIECE;
            max_n = std::max(max_n, ns);
            total += ns;
        }
        // device side of the table: off[n] (u64) | nsym[n] | piece_sec[P] | piece_idx[P] | piece_crc[P] | piece_nl[P] | bad
        const size_t o_nsym = 8ull * n, o_psec = o_nsym + 4ull * n, o_pidx = o_psec + 4ull * n_pieces, o_pcrc = o_pidx + 4ull * n_pieces, o_pnl = o_pcrc + 4ull * n_pieces,
                     o_bad = o_pnl + 4ull * n_pieces;
        const size_t tab_bytes = (o_bad + 4 + 15) & ~(size_t)15;
        const size_t pin_need = tab_bytes + GZB_WINDOW * 2;
        if (rs_pin_cap_ < pin_need) {
            if (rs_pin_) aqc_host_free(rs_pin_);
            rs_pin_cap_ = pin_need + pin_need / 2 + (1u << 20);
            rs_pin_ = (uint8_t*)aqc_host_alloc(rs_pin_cap_);
            if (!rs_pin_) { rs_pin_cap_ = 0; return -2; }
        }
        if (rs_tab_.reserve(tab_bytes) || rs_wins_.reserve((size_t)(n + 1) * GZB_WINDOW)) { (void)hipGetLastError(); return -2; }
        if (!rs_crc_tab_.p) {
            uint32_t tab[GZB_CRC_TAB_WORDS];
            auto advance = [](uint32_t x, uint64_t len) { return (uint32_t)crc32_combine((uLong)x, 0UL, (z_off_t)len); };
            gzb_crc_tables(tab, advance);
            for (int j = 0; j < 32; ++j) adv_piece_[j] = advance(1u << j, GZB_CRC_PIECE);
            if (rs_crc_tab_.reserve(sizeof(tab)) || hipMemcpy(rs_crc_tab_.p, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); rs_crc_tab_.release(); return -2; }
        }
        uint8_t* const pin = rs_pin_;
        uint64_t* const h_off = (uint64_t*)pin;
        uint32_t* const h_nsym = (uint32_t*)(pin + o_nsym);
        uint32_t* const h_psec = (uint32_t*)(pin + o_psec);
        uint32_t* const h_pidx = (uint32_t*)(pin + o_pidx);
        {
            uint32_t p = 0;
            for (int k = 0; k < n; ++k) {
                const Token* t = (const Token*)tokens[k];
                h_off[k] = R.off[(size_t)t->k];
                h_nsym[k] = R.nsym[(size_t)t->k];
                const uint32_t cnt = (h_nsym[k] + GZB_CRC_PIECE - 1u) / GZB_CRC_PIECE;
                for (uint32_t i = 0; i < cnt; ++i, ++p) { h_psec[p] = (uint32_t)k; h_pidx[p] = i; }
            }
            *(uint32_t*)(pin + o_bad) = 0;
        }
        uint8_t* const h_win = pin + tab_bytes;                 // the window before the run, right-aligned; behind it the one behind the run comes back
        memset(h_win, 0, GZB_WINDOW - wlen);
        if (wlen) memcpy(h_win + GZB_WINDOW - wlen, win, wlen);
        GzbResolveJob J{};
        uint8_t* const dtab = (uint8_t*)rs_tab_.p;
        J.sym = (const uint16_t*)R.sym.p; J.text = (uint8_t*)R.text.p; J.wins = (uint8_t*)rs_wins_.p;
        J.off = (const uint64_t*)dtab; J.nsym = (const uint32_t*)(dtab + o_nsym); J.n_run = (uint32_t)n;
        J.valid0 = (uint32_t)(GZB_WINDOW - wlen); J.bad = (uint32_t*)(dtab + o_bad);
        J.piece_sec = (const uint32_t*)(dtab + o_psec); J.piece_idx = (const uint32_t*)(dtab + o_pidx); J.piece_crc = (uint32_t*)(dtab + o_pcrc);
        J.piece_nl = (uint32_t*)(dtab + o_pnl);
        J.n_pieces = n_pieces; J.crc_tab = (const uint32_t*)rs_crc_tab_.p;
        bool ok = hipMemcpyAsync(dtab, pin, tab_bytes, hipMemcpyHostToDevice, rs_stream_) == hipSuccess &&
                  hipMemcpyAsync(rs_wins_.p, h_win, GZB_WINDOW, hipMemcpyHostToDevice, rs_stream_) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(gzb_windows_kernel, dim3(1), dim3(GZB_WIN_THREADS), 0, rs_stream_, J);
            if (max_n) hipLaunchKernelGGL(gzb_resolve_kernel, dim3((max_n + GZB_RES_THREADS * 16 - 1) / (GZB_RES_THREADS * 16), (unsigned)n), dim3(GZB_RES_THREADS), 0, rs_stream_, J);
            if (n_pieces) hipLaunchKernelGGL(gzb_crc_kernel, dim3(n_pieces), dim3(GZB_CRC_THREADS), 0, rs_stream_, J);
            ok = hipGetLastError() == hipSuccess &&
                 hipMemcpyAsync(pin + o_pcrc, dtab + o_pcrc, 8ull * n_pieces + 4, hipMemcpyDeviceToHost, rs_stream_) == hipSuccess &&
                 hipMemcpyAsync(h_win + GZB_WINDOW, (uint8_t*)rs_wins_.p + (size_t)n * GZB_WINDOW, GZB_WINDOW, hipMemcpyDeviceToHost, rs_stream_) == hipSuccess &&
                 hipStreamSynchronize(rs_stream_) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            std::lock_guard<std::mutex> g(mu_);
            broken_ = true;
            return -2;
        }
        if (*(const uint32_t*)(pin + o_bad)) return aqcgz::GZ_ERR_DATA;
        {
            auto advance = [](uint32_t x, uint64_t len) { return (uint32_t)crc32_combine((uLong)x, 0UL, (z_off_t)len); };
            const uint32_t* pc = (const uint32_t*)(pin + o_pcrc);
            for (int k = 0; k < n; ++k) {
                const uint32_t cnt = (h_nsym[k] + GZB_CRC_PIECE - 1u) / GZB_CRC_PIECE;
                crc[k] = gzb_crc_fold(pc, cnt, h_nsym[k], adv_piece_, advance);
                pc += cnt;
            }
            if (piece_nl) memcpy(piece_nl, pin + o_pnl, 4ull * n_pieces);
        }
        const size_t tl = (size_t)std::min<uint64_t>(GZB_WINDOW, wlen + total);
        memcpy(tail, h_win + 2 * GZB_WINDOW - tl, tl);
        *tail_len = tl;
        g_gzb_resolve_stats[0] += 1; g_gzb_resolve_stats[1] += (uint64_t)n; g_gzb_resolve_stats[2] += (uint64_t)((now_s() - t0) * 1e6); g_gzb_resolve_stats[3] += total;
        return 0;
    }
    bool fetch(void* token, size_t off, size_t len, uint8_t* dst) override {
        const Token* t = (const Token*)token;
        if (t->res < 0) return false;
        Res& R = res_[t->res];
        if (t->k < 0 || (size_t)t->k >= R.nsym.size() || off + len > R.nsym[(size_t)t->k]) return false;
        if (!len) return true;
        if (hipSetDevice(device_) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipMemcpyAsync(dst, (const uint8_t*)R.text.p + R.off[(size_t)t->k] + off, len, hipMemcpyDeviceToHost, rs_stream_) != hipSuccess) { (void)hipGetLastError(); return false; }
        return true;
    }
    bool fetch_wait() override {
        if (hipSetDevice(device_) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipStreamSynchronize(rs_stream_) != hipSuccess) { (void)hipGetLastError(); return false; }
        return true;
    }

private:
    // (two lane threads share ONE set of decode buffers: a group's compressed bytes are copied out of the file mapping into the
    //  lane's stage while the other lane's kernels run; the device part of a group takes the set for itself)
    static constexpr int N_LANES = 2, N_ARENAS = 6, N_RES = 12;
    // AQC_GZ_DEBUG=1: what the decoder's set-up and tear-down cost (device buffers, page-locked staging and arenas), on stderr
    static bool debug() { static const bool d = getenv("AQC_GZ_DEBUG") && getenv("AQC_GZ_DEBUG")[0] == '1'; return d; }
    static double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    struct Group {
        const uint8_t* data; size_t size; int n;
        std::vector<uint64_t> nominal, stop;
        std::vector<uint8_t> exact;
        std::function<void(int, const aqcgz::OffloadResult&)> done;
    };
    struct Token { int arena; int res; int k; };
    struct Arena { uint8_t* p = nullptr; size_t cap = 0; int refs = 0; bool filling = false; };
    // a group's symbols and, once resolved, its text (text[i] = the byte of symbol i): kept until the last of its sections is dropped
    struct Res {
        DevBuf sym, text;
        std::vector<uint64_t> off;
        std::vector<uint32_t> nsym;
        int refs = 0;
        bool filling = false;
    };
    struct Lane {
        std::thread th;
        std::unique_ptr<Group> job;
        uint8_t* stage = nullptr;          // page-locked copy of the group's compressed bytes (the file itself is a pageable mapping)
        size_t stage_cap = 0;
    };
    struct DevSet {
        hipStream_t stream = nullptr;
        hipEvent_t ev[7] = {};
        DevBuf comp, tile_cnt, tile_cand, n_cand, c_start, c_end, c_nsym, c_flags, c_symoff, c_symcap, c_tokoff, c_tokcap, blk_sym, tables, blk_tp, c_lanes, l_u32;
        DevBuf s_in, s_out, s_blocks, s_sym, s_off;
        std::vector<DevBuf*> all() {
            return {&comp, &tile_cnt, &tile_cand, &n_cand, &c_start, &c_end, &c_nsym, &c_flags, &c_symoff, &c_symcap, &c_tokoff, &c_tokcap, &blk_sym, &tables, &blk_tp, &c_lanes, &l_u32,
                    &s_in, &s_out, &s_blocks, &s_sym, &s_off};
        }
        size_t bytes() { size_t t = 0; for (DevBuf* x : all()) t += x->cap; return t; }
        void release() {
            for (DevBuf* x : all()) x->release();
            for (auto& e : ev) if (e) (void)hipEventDestroy(e);
            if (stream) (void)hipStreamDestroy(stream);
            stream = nullptr;
        }
    };
    struct Sizes { uint32_t n_tiles, cand_cap, s_symcap; uint64_t blk_sym_cap, blk_tp_cap, s_sym_total; };
    Sizes sizes_for(size_t span, int n, uint64_t sec_max, uint32_t last_bit) const {
        Sizes z;
        z.n_tiles = (uint32_t)(((size_t)(last_bit >> 3) + 1 + GZB_SCAN_TILE - 1) / GZB_SCAN_TILE);
        z.cand_cap = (uint32_t)(span / 4096 + 256);
        z.s_symcap = (uint32_t)std::min<uint64_t>((sec_max * 2 * ratio_ + (2u << 20) + 7) & ~(uint64_t)7, 0xfffffff0u);
        z.blk_sym_cap = gzb_sym_budget(span, ratio_);
        z.blk_tp_cap = gzb_tok_budget(span, tok_ratio_, overlap_tokens_);
        z.s_sym_total = (uint64_t)span * ratio_ + (uint64_t)n * 64 + (1u << 20);
        return z;
    }
    // the decode buffers for a window of `span` compressed bytes in n sections (grow only; dev_mu_ held)
    bool reserve_devset(size_t span, int n, const Sizes& z) {
        DevSet& D = dev_;
        const double t0 = now_s();
        const size_t before = D.bytes();
        if (D.comp.reserve(span + 512) || D.tile_cnt.reserve(4ull * z.n_tiles) || D.tile_cand.reserve(4ull * z.n_tiles * GZB_TILE_CAND) || D.n_cand.reserve(64) ||
            D.c_start.reserve(4ull * z.cand_cap) || D.c_end.reserve(4ull * z.cand_cap) || D.c_nsym.reserve(4ull * z.cand_cap) || D.c_flags.reserve(4ull * z.cand_cap) ||
            D.c_symoff.reserve(8ull * z.cand_cap) || D.c_symcap.reserve(4ull * z.cand_cap) || D.c_tokoff.reserve(8ull * z.cand_cap) || D.c_tokcap.reserve(4ull * z.cand_cap) ||
            D.blk_sym.reserve(2ull * z.blk_sym_cap + 64) || D.blk_tp.reserve(8ull * z.blk_tp_cap + 512) || D.c_lanes.reserve(4ull * z.cand_cap) ||
            D.l_u32.reserve(5ull * 4ull * z.cand_cap * GZB_K) || D.tables.reserve(4ull * z.cand_cap * GZB_TAB_WORDS) || D.s_in.reserve(12ull * n) || D.s_out.reserve(16ull * n) ||
            D.s_off.reserve(8ull * (n + 1)) || D.s_blocks.reserve(12ull * n * GZB_SEC_BLOCKS) || (!resident_ && D.s_sym.reserve(2ull * z.s_sym_total + 64)))
            return false;
        if (debug() && D.bytes() != before)
            fprintf(stderr, "[gz dev %d] decode buffers for %.1f MiB compressed in %d sections (symbols %u x, tokens %u per byte): %.2f GiB (blk_sym %.2f, blk_tp %.2f, tables %.2f, s_sym %.2f), reserve %.3f s\n",
                    device_, span / 1048576.0, n, ratio_, tok_ratio_, D.bytes() / 1073741824.0, D.blk_sym.cap / 1073741824.0, D.blk_tp.cap / 1073741824.0, D.tables.cap / 1073741824.0,
                    D.s_sym.cap / 1073741824.0, now_s() - t0);
        return true;
    }
    bool reserve_stage(Lane& L, size_t span) {
        if (L.stage_cap >= span) return true;
        if (L.stage) aqc_host_free(L.stage);
        L.stage_cap = span + span / 8 + (1u << 20);
        const double t0 = now_s();
        L.stage = (uint8_t*)aqc_host_alloc(L.stage_cap);
        if (debug()) fprintf(stderr, "[gz dev %d] stage: %.0f MiB page-locked in %.3f s\n", device_, L.stage_cap / 1048576.0, now_s() - t0);
        if (!L.stage) { L.stage_cap = 0; return false; }
        return true;
    }

    void loop(int li) {
        Lane& L = lanes_[li];
        (void)hipSetDevice(device_);
        (void)aqc_bind_thread_to_node(aqc_device_numa_node_of(device_));       // (the staging copies and the arenas' first touch happen here)
        for (;;) {
            Group* gr = nullptr;
            size_t prep = 0;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || L.job || (!prepare_busy_ && prepare_want_ > prepared_for_); });
                if (stop_ && !L.job) return;
                if (L.job) gr = L.job.get();
                else { prep = prepare_want_; prepare_busy_ = true; }
            }
            if (!gr) {
                // the stream has announced its groups: the decode buffers, this lane's stage (the other lane makes its own with its
                // first group) and the first result set, before the first group is accepted
                const size_t span = prep + GZB_SLACK + (1u << 20);
                const int n = (int)(prep / (256u << 10)) + 8;
                bool ok;
                {
                    std::lock_guard<std::mutex> dg(dev_mu_);
                    ok = reserve_devset(span, n, sizes_for(span, n, 2u << 20, (uint32_t)std::min<uint64_t>((uint64_t)span * 8, 0xffffffffu)));
                }
                ok = ok && reserve_stage(L, span);
                if (ok && resident_) {
                    // three result sets of a typical group's size (FASTQ expands 3 - 5 x) — the consumer is seldom further behind;
                    // more are made when they are needed, which then costs the lane that needs one 16 ms per GB
                    int got[3] = {-1, -1, -1};
                    for (int& ri : got) ri = take_res((uint64_t)prep * 4);
                    std::lock_guard<std::mutex> g(mu_);
                    for (int ri : got) if (ri >= 0) res_[ri].filling = false;
                }
                {
                    std::lock_guard<std::mutex> g(mu_);
                    prepare_busy_ = false;
                    prepared_for_ = std::max(prepared_for_, prep);
                    prepared_ = true;
                    if (!ok) { (void)hipGetLastError(); broken_ = true; }
                }
                cv_.notify_all();
                continue;
            }
            if (!run_group(L, *gr)) {
                // the device path failed for this group: the sections come back empty, the host decodes that stretch itself
                (void)hipGetLastError();
                aqcgz::OffloadResult none;
                for (int k = 0; k < gr->n; ++k) gr->done(k, none);
                std::lock_guard<std::mutex> g(mu_);
                broken_ = true;
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                L.job.reset();
                prepared_ = true;            // (a decoder that has run a group has its buffers)
                if (prepared_for_ == 0) prepared_for_ = 1;
            }
        }
    }

    // a free arena of at least `need` bytes (waits for one; grows the smallest free one when none is big enough)
    int take_arena(size_t need) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            int best = -1, empty = -1, small = -1;
            for (int i = 0; i < N_ARENAS; ++i) {
                Arena& a = arenas_[i];
                if (a.refs || a.filling) continue;
                if (!a.p) { if (empty < 0) empty = i; continue; }
                if (a.cap >= need) { if (best < 0 || a.cap < arenas_[best].cap) best = i; }
                else if (small < 0) small = i;
            }
            int pick = best >= 0 ? best : (empty >= 0 ? empty : small);
            if (pick >= 0) {
                Arena& a = arenas_[pick];
                a.filling = true;
                if (a.cap < need) {
                    lk.unlock();
                    if (a.p) aqc_host_free(a.p);
                    const size_t want = need + need / 4 + (8u << 20);
                    const double t0 = now_s();
                    a.p = (uint8_t*)aqc_host_alloc(want);
                    if (debug()) fprintf(stderr, "[gz dev %d] arena %d: %.0f MiB page-locked in %.3f s\n", device_, pick, want / 1048576.0, now_s() - t0);
                    a.cap = a.p ? want : 0;
                    lk.lock();
                    if (!a.p) { a.filling = false; return -1; }
                }
                return pick;
            }
            if (stop_) return -1;
            cv_.wait(lk);
        }
    }
    // a free result set for `need` symbols (waits for one; the best fit, else an empty one, else the smallest grows).  The consumer
    // holds at most four groups' worth of sections of a stream (ParallelGunzip::top_up), the chunks on their way to the slots a few
    // more (the pipe's ring: five chunks of ~45 MB), and there are two lanes: twelve never run out.
    int take_res(uint64_t need) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            int best = -1, empty = -1, small = -1;
            for (int i = 0; i < N_RES; ++i) {
                Res& r = res_[i];
                if (r.refs || r.filling) continue;
                if (!r.sym.p) { if (empty < 0) empty = i; continue; }
                if (r.sym.cap >= 2 * need + 64 && r.text.cap >= need + 64) { if (best < 0 || r.sym.cap < res_[best].sym.cap) best = i; }
                else if (small < 0 || r.sym.cap > res_[small].sym.cap) small = i;
            }
            const int pick = best >= 0 ? best : (empty >= 0 ? empty : small);
            if (pick >= 0) {
                Res& r = res_[pick];
                r.filling = true;
                if (r.sym.cap < 2 * need + 64 || r.text.cap < need + 64) {
                    lk.unlock();
                    const double t0 = now_s();
                    const bool bad = r.sym.reserve(2 * need + 64) || r.text.reserve(need + 64);
                    if (debug()) fprintf(stderr, "[gz dev %d] result set %d: %.2f GiB (symbols + text of %.0f M symbols) in %.3f s\n", device_, pick, (r.sym.cap + r.text.cap) / 1073741824.0, need / 1e6, now_s() - t0);
                    lk.lock();
                    if (bad) { (void)hipGetLastError(); r.filling = false; return -1; }
                }
                return pick;
            }
            if (stop_) return -1;
            cv_.wait(lk);
        }
    }

#define GZB_TRY(expr) do { if ((expr) != hipSuccess) return false; } while (0)
    bool run_group(Lane& L, Group& G) {
        const int n = G.n;
        const uint64_t byte0 = (G.nominal[0] >> 3) & ~(uint64_t)15;
        const uint64_t end_byte = std::min<uint64_t>(G.size, (G.stop[n - 1] >> 3) + 1 + GZB_SLACK);
        const size_t span = (size_t)(end_byte - byte0);
        const uint32_t first_bit = (uint32_t)(G.nominal[0] - byte0 * 8), last_bit = (uint32_t)std::min<uint64_t>(G.stop[n - 1] - byte0 * 8, (uint64_t)span * 8);
        uint64_t sec_max = 0;
        for (int k = 0; k < n; ++k) sec_max = std::max<uint64_t>(sec_max, (G.stop[k] - G.nominal[k]) >> 3);
        // the compressed bytes: out of the (pageable, possibly not yet faulted-in) file mapping into page-locked memory with a few
        // threads side by side, then one DMA — a copy straight from the mapping runs at the page-fault rate of one thread.  (Before
        // the decode buffers are taken: the other lane's kernels run meanwhile.)
        if (!reserve_stage(L, span)) return false;
        {
            const int T = span > (8u << 20) ? 4 : 1;
            std::vector<std::thread> th;
            const size_t per = (span + T - 1) / T;
            for (int t = 1; t < T; ++t)
                th.emplace_back([&, t] { const size_t a = std::min(span, t * per), b = std::min(span, a + per); memcpy(L.stage + a, G.data + byte0 + a, b - a); });
            memcpy(L.stage, G.data + byte0, std::min(span, per));
            for (auto& x : th) x.join();
        }
        // section table: nominal, stop, exact (bits relative to the window)
        std::vector<uint32_t> sin(3 * (size_t)n);
        for (int k = 0; k < n; ++k) {
            sin[k] = (uint32_t)(G.nominal[k] - byte0 * 8);
            sin[n + k] = (uint32_t)std::min<uint64_t>(G.stop[k] - byte0 * 8, (uint64_t)span * 8);
            sin[2 * n + k] = G.exact[k];
        }
        std::unique_lock<std::mutex> dg(dev_mu_);
        DevSet& D = dev_;
      for (int attempt = 0;; ++attempt) {
        const Sizes z = sizes_for(span, n, sec_max, last_bit);
        if (!reserve_devset(span, n, z)) return false;
        GZB_TRY(hipEventRecord(D.ev[0], D.stream));
        GZB_TRY(hipMemcpyAsync(D.comp.p, L.stage, span, hipMemcpyHostToDevice, D.stream));
        GZB_TRY(hipMemsetAsync((uint8_t*)D.comp.p + span, 0, 512, D.stream));       // (the lanes' stream windows read up to 200 bytes ahead)
        GZB_TRY(hipMemcpyAsync(D.s_in.p, sin.data(), 12ull * n, hipMemcpyHostToDevice, D.stream));
        GzbJob J{};
        J.comp = (const uint8_t*)D.comp.p; J.comp_bytes = (uint32_t)span; J.scan_byte0 = 0; J.first_bit = first_bit; J.last_bit = last_bit;
        J.n_tiles = z.n_tiles; J.tile_cnt = (uint32_t*)D.tile_cnt.p; J.tile_cand = (uint32_t*)D.tile_cand.p;
        J.cand_cap = z.cand_cap; J.n_cand = (uint32_t*)D.n_cand.p;
        J.c_start = (uint32_t*)D.c_start.p; J.c_end = (uint32_t*)D.c_end.p; J.c_nsym = (uint32_t*)D.c_nsym.p; J.c_flags = (uint32_t*)D.c_flags.p;
        J.c_symoff = (uint64_t*)D.c_symoff.p; J.c_symcap = (uint32_t*)D.c_symcap.p; J.blk_sym = (uint16_t*)D.blk_sym.p; J.blk_sym_cap = z.blk_sym_cap;
        J.c_tokoff = (uint64_t*)D.c_tokoff.p; J.c_tokcap = (uint32_t*)D.c_tokcap.p; J.blk_tp_cap = z.blk_tp_cap; J.tok_ratio = tok_ratio_; J.overlap_tokens = overlap_tokens_;
        J.ratio_cap = ratio_; J.tables = (uint32_t*)D.tables.p; J.blk_tp = (unsigned long long*)D.blk_tp.p;
        J.c_lanes = (uint32_t*)D.c_lanes.p;
        J.l_p = (uint32_t*)D.l_u32.p; J.l_stop = J.l_p + (size_t)z.cand_cap * GZB_K; J.l_start = J.l_stop + (size_t)z.cand_cap * GZB_K;
        J.l_ntok = J.l_start + (size_t)z.cand_cap * GZB_K; J.l_flags = J.l_ntok + (size_t)z.cand_cap * GZB_K;
        {
            static const uint32_t slice = [] { const char* e = getenv("AQC_GZ_SLICE"); return e ? (uint32_t)std::max(16, atoi(e)) : 2048u; }();
            J.slice_tokens = slice;
        }
        J.n_sec = (uint32_t)n; J.s_nominal = (const uint32_t*)D.s_in.p; J.s_stop = J.s_nominal + n; J.s_exact = J.s_nominal + 2 * n;
        J.s_start = (uint32_t*)D.s_out.p; J.s_end = J.s_start + n; J.s_nsym = J.s_start + 2 * n; J.s_nblk = J.s_start + 3 * n;
        J.s_blocks = (uint32_t*)D.s_blocks.p; J.s_off = (uint64_t*)D.s_off.p; J.s_sym = (uint16_t*)D.s_sym.p; J.s_symcap = z.s_symcap;
        // (resident: the result set is taken once the sections' sizes are known, so every section that chained up has its place)
        J.s_sym_total = resident_ ? ~0ull >> 2 : z.s_sym_total;
        GZB_TRY(hipEventRecord(D.ev[1], D.stream));
        hipLaunchKernelGGL(gzb_scan_kernel, dim3(z.n_tiles), dim3(GZB_SCAN_THREADS), 0, D.stream, J);
        hipLaunchKernelGGL(gzb_compact_kernel, dim3(1), dim3(1024), 0, D.stream, J);
        GZB_TRY(hipEventRecord(D.ev[2], D.stream));
        // the decoder in slices (aqc_gunzip_dev.hpp): GZB_K lanes per block, each with its share of it and the overlap: 6 x 2048
        // tokens cover the blocks of zlib (<= 16 K tokens) and of GNU gzip (<= 32 K) with room to spare; a lane that needs more
        // stays unfinished, its block counts as failed, the section ends before it and the host goes on from there
        {
            static const int n_slices = [] { const char* e = getenv("AQC_GZ_SLICES"); return e ? std::max(1, atoi(e)) : 6; }();
            hipLaunchKernelGGL(gzb_tables_kernel, dim3((z.cand_cap + GZB_DEC_THREADS - 1) / GZB_DEC_THREADS), dim3(GZB_DEC_THREADS), 0, D.stream, J);
            const dim3 grid((z.cand_cap * GZB_K + GZB_DEC_THREADS - 1) / GZB_DEC_THREADS);
            for (int sl = 0; sl < n_slices; ++sl) hipLaunchKernelGGL(gzb_decode_kernel, grid, dim3(GZB_DEC_THREADS), 0, D.stream, J);
            // phase 2: a wave per block stitches its lanes' lists together and applies the tokens
            hipLaunchKernelGGL(gzb_expand_kernel, dim3((z.cand_cap + GZB_EXP_WAVES - 1) / GZB_EXP_WAVES), dim3(64 * GZB_EXP_WAVES), 0, D.stream, J);
        }
        GZB_TRY(hipEventRecord(D.ev[3], D.stream));
        hipLaunchKernelGGL(gzb_chain_kernel, dim3((n + 63) / 64), dim3(64), 0, D.stream, J);
        hipLaunchKernelGGL(gzb_place_kernel, dim3(1), dim3(1), 0, D.stream, J);
        GZB_TRY(hipGetLastError());
        // what each section became (start, end, symbols) and where its symbols go
        std::vector<uint32_t> sout(4 * (size_t)n);
        std::vector<uint64_t> soff((size_t)n + 1);
        GZB_TRY(hipMemcpyAsync(sout.data(), D.s_out.p, 16ull * n, hipMemcpyDeviceToHost, D.stream));
        GZB_TRY(hipMemcpyAsync(soff.data(), D.s_off.p, 8ull * (n + 1), hipMemcpyDeviceToHost, D.stream));
        GZB_TRY(hipStreamSynchronize(D.stream));
        int live = 0;
        for (int k = 0; k < n; ++k) if (sout[k] != GZB_NONE && sout[2 * n + k] != 0) ++live;
        // A group that comes back short on the lean budgets (symbols 6 x, one token per compressed byte) is decoded once more with
        // room to spare, and so is every group after it (once per decoder: an input that compresses 6 x and better, or is nearly
        // all literals, is rare — and says so here; what is still missing then is not a matter of space, and the host's)
        if (live < n && !grown_) {
            grown_ = true;
            {
                std::lock_guard<std::mutex> g(mu_);
                ratio_ = std::max(ratio_, 12u);
                tok_ratio_ = std::max(tok_ratio_, 6u);
                overlap_tokens_ = std::max(overlap_tokens_, 2048u);
            }
            if (debug()) fprintf(stderr, "[gz dev %d] %d of %d sections came back empty: symbol space %u x, %u tokens per byte from now on; the group is decoded again\n", device_, n - live, n, ratio_, tok_ratio_);
            if (attempt == 0) continue;
        }
        int ai = -1, ri = -1;
        bool ok = true;
        if (resident_) {
            if (soff[n] && live) {
                ri = take_res(soff[n]);
                if (ri < 0) return false;
                J.s_sym = (uint16_t*)res_[ri].sym.p;
                hipLaunchKernelGGL(gzb_gather_kernel, dim3(n), dim3(GZB_GATHER_THREADS), 0, D.stream, J);
            }
            ok = hipEventRecord(D.ev[4], D.stream) == hipSuccess && hipEventRecord(D.ev[6], D.stream) == hipSuccess && hipEventRecord(D.ev[5], D.stream) == hipSuccess &&
                 hipGetLastError() == hipSuccess && hipStreamSynchronize(D.stream) == hipSuccess;
            if (ri >= 0) {
                std::lock_guard<std::mutex> g(mu_);
                Res& R = res_[ri];
                R.filling = false;
                R.refs = ok ? live : 0;
                R.off.assign(soff.begin(), soff.begin() + n);
                R.nsym.assign(sout.begin() + 2 * n, sout.begin() + 3 * n);
            }
        } else {
            hipLaunchKernelGGL(gzb_gather_kernel, dim3(n), dim3(GZB_GATHER_THREADS), 0, D.stream, J);
            ok = hipEventRecord(D.ev[4], D.stream) == hipSuccess;
            const size_t need = (size_t)soff[n] * 2;
            if (ok && need && live) {
                ai = take_arena(need);
                if (ai < 0) return false;
                ok = hipEventRecord(D.ev[6], D.stream) == hipSuccess &&
                     hipMemcpyAsync(arenas_[ai].p, D.s_sym.p, need, hipMemcpyDeviceToHost, D.stream) == hipSuccess;
            } else ok = ok && hipEventRecord(D.ev[6], D.stream) == hipSuccess;
            ok = ok && hipEventRecord(D.ev[5], D.stream) == hipSuccess && hipStreamSynchronize(D.stream) == hipSuccess;
            if (ai >= 0) {
                std::lock_guard<std::mutex> g(mu_);
                arenas_[ai].filling = false;
                arenas_[ai].refs = ok ? live : 0;
            }
        }
        if (!ok) { cv_.notify_all(); return false; }
        float ms[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&ms[i], D.ev[i], D.ev[i + 1]);
        (void)hipEventElapsedTime(&ms[4], D.ev[6], D.ev[5]);       // (the symbols' copy alone: getting an arena is host time)
        dg.unlock();
        g_gzb_stats[0] += (uint64_t)(ms[1] * 1000); g_gzb_stats[1] += (uint64_t)(ms[2] * 1000); g_gzb_stats[2] += (uint64_t)(ms[3] * 1000);
        g_gzb_stats[3] += (uint64_t)(ms[0] * 1000); g_gzb_stats[4] += (uint64_t)(ms[4] * 1000); g_gzb_stats[5] += 1; g_gzb_stats[6] += (uint64_t)n; g_gzb_stats[7] += (uint64_t)live;
        for (int k = 0; k < n; ++k) {
            aqcgz::OffloadResult r;
            if (sout[k] != GZB_NONE && sout[2 * n + k] != 0) {
                r.found = true;
                r.start_bit = byte0 * 8 + sout[k];
                r.end_bit = byte0 * 8 + sout[n + k];
                r.n_sym = sout[2 * n + k];
                if (resident_) { r.resident = true; r.token = new Token{-1, ri, k}; }
                else { r.sym = (const uint16_t*)arenas_[ai].p + soff[k]; r.token = new Token{ai, -1, k}; }
            }
            G.done(k, r);
        }
        return true;
      }
    }
#undef GZB_TRY

    int device_;
    size_t group_bytes_;
    bool resident_ = true;
    uint32_t ratio_ = 6, tok_ratio_ = 1, overlap_tokens_ = GZB_OVERLAP_TOKENS;
    bool grown_ = false;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false, broken_ = false;
    bool prepared_ = true, prepare_busy_ = false;       // (prepared_: false between prepare() and the moment the buffers it asked for exist)
    size_t prepare_want_ = 0, prepared_for_ = 0;
    Lane lanes_[N_LANES];
    std::mutex dev_mu_;
    DevSet dev_;
    Arena arenas_[N_ARENAS];
    Res res_[N_RES];
    // resolve() / fetch(): the consumer's side
    std::mutex rs_mu_;
    hipStream_t rs_stream_ = nullptr;
    DevBuf rs_wins_, rs_tab_, rs_crc_tab_;
    uint8_t* rs_pin_ = nullptr;
    size_t rs_pin_cap_ = 0;
    uint32_t adv_piece_[32] = {};
};

}  // namespace

}  // extern "C"
namespace aqcgz {
SectionOffload* make_device_offload(int device, size_t group_bytes) {
    std::unique_ptr<DeviceInflate> d(new DeviceInflate(device, group_bytes));
    if (!d->start()) { (void)hipGetLastError(); return nullptr; }
    return d.release();
}
void device_offload_stats(uint64_t out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = g_gzb_stats[i].load();
}
void device_resolve_stats(uint64_t out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = g_gzb_resolve_stats[i].load();
}
}  // namespace aqcgz
extern "C" {

// One gzip file decoded with the device taking every section it can (what the pipe does for a `.gz` input, minus the pool's share
// of the sections): `threads` host threads translate symbols and check CRC-32 / ISIZE, the stream's last section and whatever the
// device does not chain up is decoded on the host.  stats: sections committed from the device / from the host, bytes decoded
// sequentially on the host (bridges), then aqcgz::device_offload_stats()[0..5) of this call (microseconds in the scan + compact,
// decode, chain + gather kernels, H2D, D2H).
int aqc_gunzip_dev(int device, const uint8_t* gz, uint64_t size, uint8_t* out, uint64_t cap, uint64_t* n_out, uint64_t stats[8], int threads,
                   uint64_t section_bytes, uint64_t group_bytes) {
    if (!gz || !out || !n_out || !stats) return fail(AQC_ERR_ARG, "null argument");
    // (one decoder per device and group size for the life of the process: its device buffers and page-locked arenas cost more to
    //  set up than a gigabyte takes to decode)
    static std::mutex cache_mu;
    static std::vector<std::pair<std::pair<int, size_t>, std::unique_ptr<aqcgz::SectionOffload>>> cache;
    const size_t gb = group_bytes ? (size_t)group_bytes : (256u << 20);
    aqcgz::SectionOffload* off = nullptr;
    {
        std::lock_guard<std::mutex> g(cache_mu);
        for (auto& e : cache) if (e.first.first == device && e.first.second == gb) off = e.second.get();
        if (!off) {
            std::unique_ptr<aqcgz::SectionOffload> made(aqcgz::make_device_offload(device, gb));
            if (made) { off = made.get(); cache.emplace_back(std::make_pair(device, gb), std::move(made)); }
        }
    }
    if (!off) return fail(AQC_ERR_HIP, "device gunzip: cannot set up device %d", device);
    uint64_t before[8], after[8];
    aqcgz::device_offload_stats(before);
    aqc_host::Pool pool(threads > 0 ? threads : 0);
    memset(stats, 0, 8 * sizeof(uint64_t));
    uint64_t produced = 0;
    int rc = 0;
    {
        aqcgz::ParallelGunzip pg(gz, (size_t)size, threads > 0 ? &pool : nullptr, std::max(4, 2 * threads), section_bytes ? (size_t)section_bytes : (1u << 20), off, true);
        while (produced < cap) {
            const size_t got = pg.read(out + produced, (size_t)std::min<uint64_t>(cap - produced, 256u << 20));
            if (pg.failed()) { rc = fail(AQC_ERR_ARG, "device gunzip: %s", pg.error()); break; }
            if (!got) break;
            produced += got;
        }
        if (!rc && produced == cap) {
            uint8_t probe;
            if (pg.read(&probe, 1) != 0) rc = fail(AQC_ERR_ARG, "output does not fit");
        }
        stats[0] = pg.offloaded_accepted; stats[1] = pg.sections_accepted - pg.offloaded_accepted; stats[2] = pg.bridged_bytes;
    }
    aqcgz::device_offload_stats(after);
    for (int i = 0; i < 5; ++i) stats[3 + i] = after[i] - before[i];
    *n_out = produced;
    return rc;
}

int aqc_format(aqc_ctx* c, int slot, uint64_t n, int32_t store_overlap, uint64_t bytes_out[6]) {
    return format_impl(c, slot, slot, n, store_overlap, bytes_out);
}

int aqc_format_spans(aqc_ctx* c, int slot, uint64_t n, int32_t store_overlap, uint64_t bytes_out[6], uint64_t n_events[2]) {
    if (!n_events) return fail(AQC_ERR_ARG, "aqc_format_spans: null argument");
    const int rc = format_impl(c, slot, slot, n, store_overlap, bytes_out, true);
    if (rc) return rc;
    n_events[0] = c->slots[slot].n_events[0];
    n_events[1] = c->slots[slot].n_events[1];
    return 0;
}

int aqc_format_fused(aqc_ctx* c, int slot) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->formatted) return fail(AQC_ERR_STATE, "aqc_format_fused before aqc_format");
    return s->formatted_fused ? 1 : 0;
}

int aqc_span_end(aqc_ctx* c, int slot, uint64_t n, uint64_t end[2]) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->framed || !end || n > s->n) return fail(AQC_ERR_ARG, "aqc_span_end: bad arguments");
    for (int f = 0; f < 2; ++f) {
        end[f] = 0;
        if (f == 1 && !s->paired) break;
        if (n == s->n) { end[f] = s->consumed[f]; continue; }
        uint32_t off = 0;               // record n begins where record n - 1 ends
        HIP_TRY(hipMemcpyAsync(&off, (const uint32_t*)s->t_name_off[f].p + n, sizeof(off), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        end[f] = off;
    }
    return 0;
}

int aqc_fetch_span_events(aqc_ctx* c, int slot, int file, aqc_span_event* dst, uint64_t cap) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->formatted) return fail(AQC_ERR_STATE, "aqc_fetch_span_events before aqc_format_spans");
    if (file < 0 || file > 1) return fail(AQC_ERR_ARG, "aqc_fetch_span_events: bad file");
    static_assert(sizeof(aqc_span_event) == sizeof(SpanEvent), "host and device event layouts must agree");
    const uint64_t ne = s->n_events[file];
    if (ne > cap) return fail(AQC_ERR_ARG, "aqc_fetch_span_events: %llu events do not fit %llu", (unsigned long long)ne, (unsigned long long)cap);
    if (ne) {
        if (!dst) return fail(AQC_ERR_ARG, "aqc_fetch_span_events: null destination");
        HIP_TRY(hipMemcpyAsync(dst, s->f_events[file].p, sizeof(SpanEvent) * ne, hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return check_status(*s);
}

int aqc_format_plain(aqc_ctx* c, int slot, int verdict_slot, uint64_t n, int32_t store_overlap, uint64_t bytes_out[6]) {
    if (slot == verdict_slot) return fail(AQC_ERR_ARG, "aqc_format_plain: the verdicts must come from another slot");
    return format_impl(c, slot, verdict_slot, n, store_overlap, bytes_out);
}

int aqc_fetch_streams(aqc_ctx* c, int slot, int32_t gz, uint8_t* const dst[6], const uint64_t cap[6]) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!dst || !cap) return fail(AQC_ERR_ARG, "aqc_fetch_streams: null argument");
    if (!s->formatted) return fail(AQC_ERR_STATE, "aqc_fetch_streams before aqc_format");
    if (gz && !s->compressed) return fail(AQC_ERR_STATE, "aqc_fetch_streams(gz) before aqc_compress");
    for (int q = 0; q < 6; ++q) {
        const uint64_t nb = gz ? s->g_bytes[q] : s->f_bytes[q];
        if (!nb) continue;
        if (!dst[q] || nb > cap[q]) return fail(AQC_ERR_ARG, "aqc_fetch_streams: stream %d (%llu bytes) does not fit", q, (unsigned long long)nb);
        HIP_TRY(hipMemcpyAsync(dst[q], gz ? s->g_packed[q].p : s->f_out[q].p, nb, hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return check_status(*s);
}

int aqc_fetch_text(aqc_ctx* c, int slot, int file, int stream, uint8_t* dst, uint64_t cap) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->formatted) return fail(AQC_ERR_STATE, "aqc_fetch_text before aqc_format");
    if (file < 0 || file > 1 || stream < 0 || stream > 2) return fail(AQC_ERR_ARG, "aqc_fetch_text: bad file/stream");
    const int q = file * 3 + stream;
    if (s->f_bytes[q] > cap) return fail(AQC_ERR_ARG, "aqc_fetch_text: %llu bytes do not fit %llu", (unsigned long long)s->f_bytes[q], (unsigned long long)cap);
    if (s->f_bytes[q]) {
        if (!dst) return fail(AQC_ERR_ARG, "aqc_fetch_text: null destination");
        HIP_TRY(hipMemcpyAsync(dst, s->f_out[q].p, s->f_bytes[q], hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return check_status(*s);
}

// Page-locked host memory.  Not hipHostMalloc: in a fresh process that costs 0.17 s per GiB (4 KiB pages faulted and pinned one by
// one, and calls from several threads serialise), which is as long as the whole 10 M-read job takes.  Anonymous memory on
// transparent huge pages, touched and then registered, is the same memory to the DMA engines (56.7 GB/s H2D either way) for
// 0.04 s per GiB, and threads do it side by side (tools/ubench/pin_rate.cpp).  Portable: the rings are filled by reader threads
// under whichever device is current and DMA-ed from by any context.
namespace {
struct HostRegion { void* user; void* base; size_t map_len; bool registered; };
std::mutex g_host_mu;
std::vector<HostRegion> g_host_regions;
}  // namespace

void* aqc_host_alloc(uint64_t bytes) {
    const size_t HUGE = 2u << 20;
    const size_t len = (((size_t)(bytes ? bytes : 1)) + HUGE - 1) & ~(HUGE - 1);
    void* base = bytes >= (1u << 20) ? mmap(nullptr, len + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0) : MAP_FAILED;
    if (base != MAP_FAILED) {
        uint8_t* p = (uint8_t*)(((uintptr_t)base + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
        (void)madvise(p, len, MADV_HUGEPAGE);
        for (size_t o = 0; o < len; o += 4096) ((volatile uint8_t*)p)[o] = 0;
        if (hipHostRegister(p, len, hipHostRegisterPortable) == hipSuccess) {
            std::lock_guard<std::mutex> g(g_host_mu);
            g_host_regions.push_back(HostRegion{p, base, len + HUGE, true});
            return p;
        }
        (void)hipGetLastError();
        munmap(base, len + HUGE);
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}

void aqc_host_free(void* p) {
    if (!p) return;
    HostRegion r{nullptr, nullptr, 0, false};
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        for (size_t i = 0; i < g_host_regions.size(); ++i)
            if (g_host_regions[i].user == p) { r = g_host_regions[i]; g_host_regions.erase(g_host_regions.begin() + (long)i); break; }
    }
    if (r.registered) {
        (void)hipHostUnregister(r.user);
        munmap(r.base, r.map_len);
    } else (void)hipHostFree(p);
}

int aqc_sync(aqc_ctx* c, int slot) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    HIP_TRY(slot_sync(*s));
    return check_status(*s);
}

int aqc_fetch_results(aqc_ctx* c, int slot, aqc_result* out, uint64_t n) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->ran) return fail(AQC_ERR_STATE, "aqc_fetch_results before aqc_run");
    if (n > s->n) return fail(AQC_ERR_ARG, "aqc_fetch_results: n exceeds the slot's records");
    if (n) HIP_TRY(hipMemcpyAsync(out, s->results.p, sizeof(aqc_result) * n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(slot_sync(*s));
    return check_status(*s);
}

int aqc_fetch_quality_views(aqc_ctx* c, int slot, int mate, uint32_t* out, uint64_t n) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!s->ran) return fail(AQC_ERR_STATE, "aqc_fetch_quality_views before aqc_run");
    if (n > s->n || !out || mate < 0 || mate > 1 || (mate == 1 && !s->paired)) return fail(AQC_ERR_ARG, "aqc_fetch_quality_views: bad arguments");
    if (n == 0) return 0;
    if (s->off_stage.reserve(sizeof(uint32_t) * n)) return fail(AQC_ERR_HIP, "hipMalloc failed");
    hipLaunchKernelGGL(quality_views_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->view, (const aqc_result*)s->results.p, mate,
                       (uint32_t*)s->off_stage.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, s->off_stage.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(slot_sync(*s));
    return 0;
}

int aqc_error_record(aqc_ctx* c, int slot, uint64_t* record) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!record) return fail(AQC_ERR_ARG, "aqc_error_record: null argument");
    *record = s->err_record;
    return 0;
}

int aqc_last_deferred(aqc_ctx* c, int slot, uint32_t* idx, uint64_t cap, uint64_t* n) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!n) return fail(AQC_ERR_ARG, "aqc_last_deferred: null argument");
    if (!s->ran) return fail(AQC_ERR_STATE, "aqc_last_deferred before aqc_run");
    HIP_TRY(slot_sync(*s));
    *n = 0;
    if (!s->used_fast || !s->n_deferred.p) return 0;
    unsigned int m = 0;
    HIP_TRY(hipMemcpy(&m, s->n_deferred.p, sizeof(m), hipMemcpyDeviceToHost));
    *n = m;
    const uint64_t w = m < cap ? m : cap;
    if (idx && w) HIP_TRY(hipMemcpy(idx, s->deferred.p, sizeof(uint32_t) * w, hipMemcpyDeviceToHost));
    return 0;
}

int aqc_kernel_ms(aqc_ctx* c, int slot, float* ms) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    HIP_TRY(slot_sync(*s));
    for (int k = 0; k < AQC_N_KERNELS; k++) {
        ms[k] = 0.f;
        if (s->timed[k]) HIP_TRY(hipEventElapsedTime(&ms[k], s->ev[k][0], s->ev[k][1]));
    }
    return 0;
}

int aqc_timing_reset(aqc_ctx* c, int slot) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    HIP_TRY(slot_sync(*s));
    for (int k = 0; k < AQC_N_KERNELS; k++) { s->ring_used[k] = 0; s->timed[k] = false; }
    s->collecting = true;
    return 0;
}

int aqc_timing_mean(aqc_ctx* c, int slot, float* mean_ms, int32_t* launches) {
    Slot* s;
    int rc = get_slot(c, slot, &s);
    if (rc) return rc;
    if (!mean_ms || !launches) return fail(AQC_ERR_ARG, "null argument");
    HIP_TRY(slot_sync(*s));
    for (int k = 0; k < AQC_N_KERNELS; k++) {
        double sum = 0;
        for (int i = 0; i < s->ring_used[k]; i++) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, s->ring[k][0][i], s->ring[k][1][i]));
            sum += ms;
        }
        launches[k] = s->ring_used[k];
        mean_ms[k] = s->ring_used[k] ? (float)(sum / s->ring_used[k]) : 0.f;
    }
    s->collecting = false;
    return 0;
}

static int sync_all(aqc_ctx* c) {
    HIP_TRY(hipSetDevice(c->device));
    for (auto& s : c->slots) {
        HIP_TRY(slot_sync(s));
        int rc = check_status(s);
        if (rc) return rc;
    }
    return 0;
}

int aqc_get_counters(aqc_ctx* c, int64_t* out) {
    if (!c || !out) return fail(AQC_ERR_ARG, "null argument");
    int rc = sync_all(c);
    if (rc) return rc;
#ifdef AQC_PROFILE
    {
        unsigned long long pr[16];
        (void)hipMemcpy(pr, c->counters + AQC_N_COUNTERS, sizeof(pr), hipMemcpyDeviceToHost);
        unsigned long long tot = 0;
        for (int k = 0; k < 10; k++) tot += pr[k];
        static const char* nm[10] = {"phase1", "normalise", "bubble+len+polyX", "lowq+N", "scan", "verify", "post+walk", "results+counters", "deferred", "-"};
        for (int k = 0; k < 9; k++) fprintf(stderr, "PROF %-18s %6.2f %%\n", nm[k], tot ? 100.0 * pr[k] / tot : 0.0);
        fprintf(stderr, "DEFER alphabet/length %llu  short-partner %llu  adapter-second-scan %llu  walk-anchor %llu\n", pr[10], pr[11], pr[12], pr[13]);
        // load balance of the last fast-kernel launch of slot 0: spread of the waves' end stamps
        {
            Slot& s0 = c->slots[0];
            if (s0.used_fast && s0.n > (1u << 16) && s0.deferred.p) {
                const size_t nw = 4096;
                std::vector<uint32_t> st(2 * nw);
                (void)hipMemcpy(st.data(), (uint32_t*)s0.deferred.p + (s0.n - 2 * nw), sizeof(uint32_t) * 2 * nw, hipMemcpyDeviceToHost);
                // (s_memtime runs on the shader clock and is not synchronised across XCDs: only lifetimes are meaningful)
                std::vector<uint32_t> life;
                double sum_life = 0;
                for (size_t w = 0; w < nw; ++w) { const uint32_t d = st[2 * w + 1] - st[2 * w]; life.push_back(d); sum_life += d; }
                {
                    // by XCD (workgroups are dealt round-robin to the 8 XCDs) and by position in the grid
                    double xs[8] = {0}, qs[4] = {0};
                    for (size_t w = 0; w < nw; ++w) {
                        const size_t gw = nw - 1 - w;          // stamps are stored from the tail backwards
                        xs[(gw / 4) % 8] += life[w];
                        qs[gw * 4 / nw] += life[w];
                    }
                    fprintf(stderr, "WAVES mean lifetime by XCD:");
                    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f", xs[x] / (nw / 8));
                    fprintf(stderr, "  by grid quarter:");
                    for (int q = 0; q < 4; ++q) fprintf(stderr, " %.0f", qs[q] / (nw / 4));
                    fprintf(stderr, "\n");
                }
                std::sort(life.begin(), life.end());
                fprintf(stderr, "WAVES lifetime in shader clocks: min %u p10 %u median %u p90 %u max %u mean %.0f\n", life.front(), life[nw / 10],
                        life[nw / 2], life[nw * 9 / 10], life.back(), sum_life / nw);
            }
        }
        unsigned long long kp[16];
        (void)hipMemcpyFromSymbol(kp, HIP_SYMBOL(g_kprof), sizeof(kp));
        tot = 0;
        for (int k = 0; k < 8; k++) tot += kp[k];
        static const char* kn[8] = {"zero+sync", "descriptors", "front(ws,shfl)", "lds adds", "first-seen", "exotic", "round-end sync", "writeout"};
        for (int k = 0; k < 8; k++) fprintf(stderr, "KPROF %-16s %6.2f %%\n", kn[k], tot ? 100.0 * kp[k] / tot : 0.0);
    }
#endif
    HIP_TRY(hipMemcpy(out, c->counters, sizeof(int64_t) * AQC_N_COUNTERS, hipMemcpyDeviceToHost));
    return 0;
}

int aqc_get_histograms(aqc_ctx* c, int64_t* ovl, int64_t* dist, int32_t n) {
    if (!c || !ovl || !dist || n < 0 || n > AQC_QC_COLS) return fail(AQC_ERR_ARG, "bad argument");
    int rc = sync_all(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ovl, c->ovl_hist, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dist, c->dist_hist, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
    return 0;
}

int aqc_get_qc(aqc_ctx* c, int which, int64_t* out) {
    if (!c || !out || which < 0 || which > 3) return fail(AQC_ERR_ARG, "bad argument");
    int rc = sync_all(c);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, c->qc[which].acc, sizeof(int64_t) * AQC_QC_ROWS * AQC_QC_COLS, hipMemcpyDeviceToHost));
    return 0;
}

int aqc_get_kmers(aqc_ctx* c, int which, uint64_t* keys, int64_t* counts, uint64_t* order, uint64_t cap, uint64_t* n) {
    if (!c || !n || which < 0 || which > 3) return fail(AQC_ERR_ARG, "bad argument");
    int rc = sync_all(c);
    if (rc) return rc;
    *n = 0;
    QcDev& q = c->qc[which];
    if (!q.kt.keys) return 0;
    const uint64_t dcap = cap < KMER_CAP + DENSE_CAP ? cap : KMER_CAP + DENSE_CAP;
    unsigned long long *dk = nullptr, *dc = nullptr, *dord = nullptr, *dn = nullptr;
    HIP_TRY(hipMalloc((void**)&dk, 8 * (dcap + 1)));
    HIP_TRY(hipMalloc((void**)&dc, 8 * (dcap + 1)));
    HIP_TRY(hipMalloc((void**)&dord, 8 * (dcap + 1)));
    HIP_TRY(hipMalloc((void**)&dn, 8));
    HIP_TRY(hipMemset(dn, 0, 8));
    hipLaunchKernelGGL(kmer_compact_kernel, dim3((unsigned)(KMER_CAP / 256)), dim3(256), 0, 0, q.kt, dk, dc, dord,
                       (unsigned long long)dcap, dn);
    hipLaunchKernelGGL(kmer_compact_dense_kernel, dim3((unsigned)(DENSE_ENTRIES / 256)), dim3(256), 0, 0, q.kt, c->cfg.qc_kmer, dk, dc,
                       dord, (unsigned long long)dcap, dn);
    HIP_TRY(hipGetLastError());
    unsigned long long m = 0;
    HIP_TRY(hipMemcpy(&m, dn, 8, hipMemcpyDeviceToHost));
    const uint64_t w = m < dcap ? m : dcap;
    if (w) {
        HIP_TRY(hipMemcpy(keys, dk, 8 * w, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(counts, dc, 8 * w, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(order, dord, 8 * w, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dk); (void)hipFree(dc); (void)hipFree(dord); (void)hipFree(dn);
    *n = m;
    if (m > dcap) return fail(AQC_ERR_ARG, "aqc_get_kmers: %llu entries exceed cap %llu", m, (unsigned long long)dcap);
    return 0;
}

// ---- function seams: run on a scratch slot (the last one) -------------------------------------------
static int seam_prepare(aqc_ctx* c, const aqc_batch* b, bool need_qual, bool need_pair, Slot** out) {
    if (!c || !b) return fail(AQC_ERR_ARG, "null argument");
    Slot* s;
    int rc = get_slot(c, c->n_slots - 1, &s);
    if (rc) return rc;
    if ((rc = fill_slot(c, *s, b, need_qual, need_pair))) return rc;
    for (uint64_t i = 0; i < b->n; i++)
        if (b->len1[i] > AQC_MAX_READ_LEN || (need_pair && b->len2[i] > AQC_MAX_READ_LEN))
            return fail(AQC_ERR_READ_TOO_LONG, "record %llu is longer than %d", (unsigned long long)i, AQC_MAX_READ_LEN);
    *out = s;
    return 0;
}

static int seam_out_bytes(Slot* s, DevBuf& d, void* host, size_t bytes) {
    if (bytes) HIP_TRY(hipMemcpyAsync(host, d.p, bytes, hipMemcpyDeviceToHost, s->stream));
    return 0;
}
#define seam_out(s, d, host, n) seam_out_bytes(s, d, host, sizeof(*(host)) * (n))

int aqc_overlap(aqc_ctx* c, const aqc_batch* b, int32_t* offset, int32_t* overlap_len, int32_t* diff) {
    Slot* s;
    int rc = seam_prepare(c, b, false, true, &s);
    if (rc) return rc;
    const uint64_t n = b->n;
    if (n == 0) return 0;
    DevBuf o[3];
    for (auto& d : o)
        if (d.reserve(4 * n)) return fail(AQC_ERR_HIP, "hipMalloc failed");
    hipLaunchKernelGGL(overlap_seam_kernel, dim3((unsigned)((n + WPB - 1) / WPB)), dim3(BLOCK), 0, s->stream, s->view,
                       (int32_t*)o[0].p, (int32_t*)o[1].p, (int32_t*)o[2].p);
    HIP_TRY(hipGetLastError());
    if ((rc = seam_out(s, o[0], offset, n)) || (rc = seam_out(s, o[1], overlap_len, n)) || (rc = seam_out(s, o[2], diff, n))) return rc;
    HIP_TRY(slot_sync(*s));
    for (auto& d : o) d.release();
    return 0;
}

int aqc_read_stats(aqc_ctx* c, const aqc_batch* b, int32_t max_poly, int32_t mismatch, int32_t qual, uint8_t* polyx,
                   int32_t* low_qual, int32_t* n_count) {
    Slot* s;
    int rc = seam_prepare(c, b, true, false, &s);
    if (rc) return rc;
    const uint64_t n = b->n;
    if (n == 0) return 0;
    DevBuf o[3];
    if (o[0].reserve(n) || o[1].reserve(4 * n) || o[2].reserve(4 * n)) return fail(AQC_ERR_HIP, "hipMalloc failed");
    hipLaunchKernelGGL(read_stats_seam_kernel, dim3((unsigned)((n + WPB - 1) / WPB)), dim3(BLOCK), 0, s->stream, s->view,
                       max_poly, mismatch, qual, (uint8_t*)o[0].p, (int32_t*)o[1].p, (int32_t*)o[2].p);
    HIP_TRY(hipGetLastError());
    if ((rc = seam_out(s, o[0], polyx, n)) || (rc = seam_out(s, o[1], low_qual, n)) || (rc = seam_out(s, o[2], n_count, n))) return rc;
    HIP_TRY(slot_sync(*s));
    for (auto& d : o) d.release();
    return 0;
}

int aqc_edit_distance(aqc_ctx* c, const aqc_batch* b, int32_t* dist) {
    Slot* s;
    int rc = seam_prepare(c, b, false, true, &s);
    if (rc) return rc;
    const uint64_t n = b->n;
    if (n == 0) return 0;
    DevBuf o;
    if (o.reserve(4 * n)) return fail(AQC_ERR_HIP, "hipMalloc failed");
    hipLaunchKernelGGL(edit_distance_seam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, s->view,
                       (int32_t*)o.p, s->status);
    HIP_TRY(hipGetLastError());
    if ((rc = seam_out(s, o, dist, n))) return rc;
    HIP_TRY(slot_sync(*s));
    o.release();
    return check_status(*s);
}

// ---- the reference's existing native seam: libed.so (editdistance/_editdistance.h:16,23, loaded by util.py:16-24) ----
// Same two symbols, same signatures, so that `cdll.LoadLibrary(<this library>)` serves util.editDistance (util.py:70) and
// util.overlap_hm_cpp (util.py:223).  Device-backed like everything else here (one lazily created context on GPU 0, one
// small launch per call); no error channel exists in these signatures, so a failure is printed and the "none" value
// of the interface comes back (0xFFFFFFFF / 0x7FFFFFFF).
static std::mutex g_compat_mu;
static aqc_ctx* g_compat = nullptr;
static DevBuf g_compat_buf[4];

static int compat_prepare(const char* a, size_t la, const char* b, size_t lb, size_t row_bytes) {
    if (!g_compat) {
        int rc = aqc_create(0, 1, &g_compat);
        if (rc) { g_compat = nullptr; return rc; }
    }
    HIP_TRY(hipSetDevice(g_compat->device));
    if (g_compat_buf[0].reserve(la + 16) || g_compat_buf[1].reserve(lb + 16) || g_compat_buf[2].reserve(row_bytes + 16) ||
        g_compat_buf[3].reserve(16))
        return fail(AQC_ERR_HIP, "hipMalloc failed");
    if (la) HIP_TRY(hipMemcpy(g_compat_buf[0].p, a, la, hipMemcpyHostToDevice));
    if (lb) HIP_TRY(hipMemcpy(g_compat_buf[1].p, b, lb, hipMemcpyHostToDevice));
    return 0;
}

unsigned int edit_distance(const char* a, const unsigned int asize, const char* b, const unsigned int bsize) {
    if (asize == 0) return bsize;                    // (_editdistance.cpp:101-102)
    if (bsize == 0) return asize;
    std::lock_guard<std::mutex> g(g_compat_mu);
    int out = -1;
    if (compat_prepare(a, asize, b, bsize, sizeof(int) * ((size_t)bsize + 1)) == 0) {
        hipLaunchKernelGGL(edit_distance_any_kernel, dim3(1), dim3(WAVE), 0, 0, (const uint8_t*)g_compat_buf[0].p, (int)asize,
                           (const uint8_t*)g_compat_buf[1].p, (int)bsize, (int*)g_compat_buf[2].p, (int*)g_compat_buf[3].p);
        if (hipMemcpy(&out, g_compat_buf[3].p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) out = -1;
    }
    if (out < 0) fprintf(stderr, "libafterqc_hip: edit_distance failed: %s\n", g_err);
    return (unsigned int)out;
}

int seek_overlap(const char* r1, const int len1, const char* r2, const int len2, const int limit_distance,
                 const int complete_compare_require, const int overlap_require) {
    std::lock_guard<std::mutex> g(g_compat_mu);
    int out = 0x7FFFFFFF;
    bool ok = len1 >= 0 && len2 >= 0 && compat_prepare(r1, (size_t)len1, r2, (size_t)len2, 0) == 0;
    if (ok) {
        hipLaunchKernelGGL(seek_overlap_kernel, dim3(1), dim3(WAVE), 0, 0, (const uint8_t*)g_compat_buf[0].p, len1,
                           (const uint8_t*)g_compat_buf[1].p, len2, limit_distance, complete_compare_require, overlap_require,
                           (int*)g_compat_buf[3].p);
        ok = hipMemcpy(&out, g_compat_buf[3].p, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) { fprintf(stderr, "libafterqc_hip: seek_overlap failed: %s\n", g_err); out = 0x7FFFFFFF; }
    return out;
}

}  // extern "C"
