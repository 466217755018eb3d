/*
 * afterqc_hip.h — C ABI of the MI355X-native AfterQC hot path (libafterqc_hip.so).
 *
 * This is the drop-in boundary for ONE path of OpenGene/AfterQC: the per-read loop of
 * preprocesser.py:411-631 (filter / trim / R1xR2 overlap + correction) and the per-cycle QC
 * accumulators of qualitycontrol.py:73-122.  Everything crossing it is a plain pointer + size;
 * no Python / torch type appears in a signature.  The reference has no batch FFI of its own (its
 * only native seam is editdistance/_editdistance.h:16,23), so each entry point below cites the
 * reference *function(s)* whose work it replaces; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative AQC_ERR_* code; aqc_last_error() returns a
 *     thread-local human readable message for the last failure;
 *   - one aqc_ctx per GPU, used from one host thread at a time; contexts are independent;
 *   - host buffers are borrowed for the duration of the call only (aqc_upload / aqc_frame DMA them with hipMemcpyAsync on
 *     the slot's stream straight from the caller's pointer: pass page-locked memory from aqc_host_alloc for full-rate,
 *     truly asynchronous copies; pageable memory works but is staged by the runtime);
 *   - a "record" is one read (single-end) or one read pair; reads are byte strings exactly as in the
 *     FASTQ file (no recoding), addressed by (offset, length) into a packed arena (SoA);
 *   - all integer results are bit-exact restatements of the reference's arithmetic.
 */
#ifndef AFTERQC_HIP_H
#define AFTERQC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQC_ABI_VERSION 3

/* longest read the reference's QC can hold (qualitycontrol.py:23 MAX_LEN = 1000) */
#define AQC_MAX_READ_LEN 1000
#define AQC_QC_COLS 1024 /* padded MAX_LEN */

/* error codes */
#define AQC_OK 0
#define AQC_ERR_HIP -1          /* a HIP runtime call failed */
#define AQC_ERR_ARG -2          /* bad argument */
#define AQC_ERR_READ_TOO_LONG -3 /* a read is longer than AQC_MAX_READ_LEN */
#define AQC_ERR_NO_DEVICE -4    /* no gfx950 device visible */
#define AQC_ERR_STATE -5        /* call sequence error (e.g. results of a slot that never ran) */
#define AQC_ERR_ALPHABET -6     /* a byte the reference would raise KeyError on (util.py:27,36-37) */
#define AQC_ERR_UNSUPPORTED -7  /* option value outside what the device path implements */
#define AQC_ERR_INDEX -8        /* the overlap walk indexed a quality string outside its length (IndexError upstream,
                                   preprocesser.py:566-567: a quality line shorter than the overlap it is read in) */

/* record verdicts == the reference's flag strings (preprocesser.py:436-614); AQC_GOOD = written to good/ */
enum aqc_flag {
    AQC_GOOD = 0,
    AQC_BADBCD1 = 1,
    AQC_BADBCD2 = 2,
    AQC_BADTRIM1 = 3,
    AQC_BADTRIM2 = 4,
    AQC_BADBBL = 5,
    AQC_BADLEN = 6,
    AQC_BADPOL = 7,
    AQC_BADLQC = 8,
    AQC_BADNCT = 9,
    AQC_BADDIFF = 10,
    AQC_BADMISMATCH = 11,
    AQC_N_FLAGS = 12
};

/* edit kinds recorded by the correction walk (preprocesser.py:563-598) */
#define AQC_EDIT_FIX_R2 1 /* R2 base := complement(b1), R2 qual := q1   (preprocesser.py:575-576) */
#define AQC_EDIT_FIX_R1 2 /* R1 base := b2, R1 qual := q2               (preprocesser.py:583-584) */
#define AQC_EDIT_MASK 3   /* both quals := '!'                           (preprocesser.py:590-592) */

#pragma pack(push, 1)
/* one edit: `o` is the walk index of preprocesser.py:563; with the record's final len1/len2 and
 * overlap_len the touched positions are R1[len1 - overlap_len + o] and R2[len2 - 1 - o]
 * (indices into the FINAL, i.e. trimmed / adapter-cut, reads). */
typedef struct aqc_edit {
    uint16_t o;
    uint8_t kind; /* AQC_EDIT_* */
    uint8_t base; /* new base for FIX_*, unused for MASK */
    uint8_t qual; /* new quality character for FIX_*, '!' for MASK */
} aqc_edit;

/* 32-byte per-record result: everything the writer needs to emit the good/bad/overlap records
 * (preprocesser.py:206-232) without touching the sequence bytes again except to apply <=3 edits. */
typedef struct aqc_result {
    uint8_t flag;       /* enum aqc_flag */
    uint8_t n_edits;    /* 0..3, applied in order even when flag == AQC_BADMISMATCH */
    uint16_t start1;    /* final R1 = original R1[start1 : start1 + len1] */
    uint16_t len1;
    uint16_t start2;    /* final R2 (paired only) */
    uint16_t len2;
    int16_t offset;     /* final util.overlap() offset   (util.py:184,207,212) */
    uint16_t overlap_len; /* final overlap_len */
    uint16_t distance;  /* final diff */
    aqc_edit edits[3];
    uint8_t barcode;    /* low nibble: barcode length found in R1, high nibble: in R2, each coded
                           0 none, 1 barcode_length-1, 2 barcode_length, 3 barcode_length+1
                           (barcodeprocesser.py:19-32) */
} aqc_result;
#pragma pack(pop)

/* options of after.py:17-92 that the per-read loop consults, after the post-processing of
 * after.py:196-221 and the auto-trim of preprocesser.py:261-280 */
typedef struct aqc_config {
    int32_t paired;                  /* read2_file given */
    int32_t count_r2_bases;          /* index2 file present -> R2 bases enter total/good bases (preprocesser.py:426-431,622-623) */
    int32_t trim_front, trim_tail;   /* R1 (preprocesser.py:455-456) */
    int32_t trim_front2, trim_tail2; /* R2 (preprocesser.py:462) */
    int32_t seq_len_req;             /* -s */
    int32_t poly_size_limit;         /* -p */
    int32_t allow_mismatch_in_poly;  /* -a */
    int32_t qualified_quality_phred; /* -q */
    int32_t unqualified_base_limit;  /* -u */
    int32_t n_base_limit;            /* -n */
    int32_t no_overlap;              /* --no_overlap */
    int32_t no_correction;           /* --no_correction */
    int32_t mask_mismatch;           /* --mask_mismatch */
    int32_t barcode;                 /* options.barcode after after.py:215-221 */
    int32_t barcode_length;          /* --barcode_length */
    int32_t barcode_verify_len;
    uint8_t barcode_verify[32];      /* --barcode_verify */
    int32_t debubble;                /* --debubble */
    int32_t qc_kmer;                 /* --qc_kmer (1..8 on the device path) */
} aqc_config;

/* one batch of records in host memory, packed SoA.  Read i's bases are seq1[off1[i] .. +len1[i]) and
 * its qualities qual1[qoff1[i] .. +len1[i]); qoff == NULL means "same offsets as the bases".  The
 * two arenas may be one and the same buffer — e.g. the raw FASTQ text chunk itself, addressed in
 * place (zero-copy framing).  Arenas must be followed by >= 64 readable bytes of padding
 * (bytes1/bytes2 include it).  seq2/qual2/off2/len2 are NULL for single-end input.  aux_* (may be NULL unless cfg.debubble)
 * carry the integers the reference parses out of the R1 name (preprocesser.py:180-192);
 * aux_ok[i] == 0 means the name did not match the pattern (-> not in a bubble). */
typedef struct aqc_batch {
    uint64_t n;                 /* records (< 2^31 per batch) */
    uint64_t first_index;       /* 0-based global index of record 0 (TOTAL_READS - 1 of preprocesser.py:433) */
    const uint8_t* seq1;
    const uint8_t* qual1;
    const uint64_t* off1;
    const uint64_t* qoff1;      /* NULL -> off1 */
    const uint32_t* len1;
    uint64_t bytes1;            /* size of the seq1 arena (and of qual1 unless qbytes1 != 0) */
    uint64_t qbytes1;           /* size of the qual1 arena, 0 -> bytes1 */
    const uint8_t* seq2;
    const uint8_t* qual2;
    const uint64_t* off2;
    const uint64_t* qoff2;
    const uint32_t* len2;
    uint64_t bytes2;
    uint64_t qbytes2;
    const int32_t* aux_lane;
    const int32_t* aux_tile;
    const int32_t* aux_x;
    const int32_t* aux_y;
    const uint8_t* aux_ok;
    /* lengths of the QUALITY strings where they are not the reads' (NULL: qlen == len for every record).  The reference never
     * compares the two lines of a record (fastq.py:37-49): each string is sliced, counted and indexed by its own length
     * (preprocesser.py:19-28,61-76,565-568, qualitycontrol.py:81-88), and so it is here; aqc_fetch_quality_views returns the
     * slices of the quality strings that go with start1/len1, start2/len2 of the result records. */
    const uint32_t* qlen1;
    const uint32_t* qlen2;
} aqc_batch;

/* scalar counters of preprocesser.py:378-409 (+ the 12-cell error matrix, init_error_matrix :125-132) */
enum aqc_counter {
    AQC_C_TOTAL_READS = 0,
    AQC_C_TOTAL_BASES,
    AQC_C_GOOD_READS,
    AQC_C_GOOD_BASES,
    AQC_C_FLAG0, /* AQC_C_FLAG0 + flag = number of records with that verdict (AQC_GOOD..AQC_BADMISMATCH) */
    AQC_C_READ_CORRECTED = AQC_C_FLAG0 + AQC_N_FLAGS,
    AQC_C_BASE_CORRECTED,
    AQC_C_BASE_SKIPPED_CORRECTION,
    AQC_C_BASE_ZERO_QUAL_MASKED,
    AQC_C_OVERLAPPED,
    AQC_C_OVERLAP_LEN_SUM,
    AQC_C_OVERLAP_BASE_SUM,
    AQC_C_OVERLAP_BASE_ERR,
    AQC_C_TRIMMED_ADAPTER_BASE,
    AQC_C_TRIMMED_ADAPTER_READ,
    AQC_C_ERR_MATRIX0, /* 16 cells [correct][error] over A,T,C,G (diagonal unused) */
    AQC_N_COUNTERS = AQC_C_ERR_MATRIX0 + 16
};

/* rows of one QualityControl accumulator block (qualitycontrol.py:33-57), each AQC_QC_COLS int64 */
enum aqc_qc_row {
    AQC_QC_TOTAL_NUM = 0,
    AQC_QC_TOTAL_QUAL,
    AQC_QC_BASE_COUNT_A, AQC_QC_BASE_COUNT_T, AQC_QC_BASE_COUNT_C, AQC_QC_BASE_COUNT_G,
    AQC_QC_BASE_QUAL_A, AQC_QC_BASE_QUAL_T, AQC_QC_BASE_QUAL_C, AQC_QC_BASE_QUAL_G,
    AQC_QC_DISCONTINUITY,
    AQC_QC_GC_HIST,
    AQC_QC_SCALARS, /* [0] totalKmer, [1] reads stat'd */
    AQC_QC_ROWS
};
/* which QualityControl object (preprocesser.py:247-254) */
#define AQC_QC_R1_PRE 0
#define AQC_QC_R2_PRE 1
#define AQC_QC_R1_POST 2
#define AQC_QC_R2_POST 3

/* kernels whose last launch duration aqc_kernel_ms() reports (HIP events on the slot's stream) */
enum aqc_kernel_id { AQC_K_FILTER_OVERLAP = 0, AQC_K_QC_STAT = 1, AQC_N_KERNELS = 2 };

typedef struct aqc_ctx aqc_ctx;

/* ---- lifetime ------------------------------------------------------------------------------- */
int aqc_abi_version(void);
int aqc_device_count(void);
const char* aqc_last_error(void);
/* one context per GPU; n_slots >= 1 double/triple-buffer slots, each with its own HIP stream */
int aqc_create(int device, int n_slots, aqc_ctx** out);
void aqc_destroy(aqc_ctx* ctx);
int aqc_device_name(aqc_ctx* ctx, char* buf, int buflen);
int aqc_device_index(aqc_ctx* ctx); /* the HIP device the context lives on */
/* The NUMA node the context's GPU hangs off (/sys/bus/pci/devices/<bus id>/numa_node; -1: unknown or a single-node host), and a
 * helper that binds the CALLING thread to that node's CPUs (intersected with the CPUs it may use; returns how many, 0 = left
 * alone; AQC_PIPE_NUMA=0 switches it off).  aqc_pipe_run binds every slot worker to the node of its context's GPU: the
 * page-locked output buffers it touches first and the copies it drives then stay on the GPU's side of the socket link. */
int aqc_device_numa_node(aqc_ctx* ctx);
int aqc_device_numa_node_of(int device);
int aqc_bind_thread_to_node(int node);

/* ---- configuration ---------------------------------------------------------------------------- */
int aqc_set_config(aqc_ctx* ctx, const aqc_config* cfg);
/* circles.csv rows (preprocesser.py:157-174): float64 x, y, radius; int lane, tile */
int aqc_set_circles(aqc_ctx* ctx, const double* cx, const double* cy, const double* radius,
                    const int32_t* lane, const int32_t* tile, int32_t n);
/* zero the counters, histograms, QC accumulators and k-mer tables of the context */
int aqc_reset_stats(aqc_ctx* ctx);

/* ---- the hot path: preprocesser.py:411-631 over a batch --------------------------------------- */
/* copy a batch into slot `slot` (async on the slot's stream; see the note on host buffers above) */
int aqc_upload(aqc_ctx* ctx, int slot, const aqc_batch* batch);
/* run filter / trim / overlap / correction over the slot's records (async).  Records with batch
 * index >= accum_limit still get a result but do not enter counters / histograms (used for the
 * --qc_only early break, preprocesser.py:630-631).  Pass UINT64_MAX for "all". */
int aqc_run(aqc_ctx* ctx, int slot, uint64_t accum_limit);
/* QualityControl.statRead (qualitycontrol.py:73-122) over records [first, first+count) of the slot
 * into accumulator `which`, reading mate 0 (seq1/qual1) or mate 1 (seq2/qual2) of each record.
 * post != 0: stat the FINAL reads (trim + edits from the slot's results applied) and only records
 * whose verdict is AQC_GOOD (preprocesser.py:624-627).  Calls for different slots of one context may come from different
 * threads (they queue up inside); the k-mer dictionary's insertion order follows the records' global indices, so calls
 * into ONE accumulator must still be issued in record order. */
int aqc_qc_stat(aqc_ctx* ctx, int slot, int which, int mate, uint64_t first, uint64_t count, int post);
/* diagnostics: the records of the slot's last aqc_run that the lane-per-read kernel did NOT decide itself but handed to the
 * general wave-per-record kernel (bytes outside A C G T N, reads under 16 bases, the second-scan corner of the adapter
 * cut, ...).  *n = their number; up to `cap` record indices go to idx (may be NULL).  Results are identical either way. */
int aqc_last_deferred(aqc_ctx* ctx, int slot, uint32_t* idx, uint64_t cap, uint64_t* n);
/* wait for the slot and copy its n result records to `out` */
int aqc_fetch_results(aqc_ctx* ctx, int slot, aqc_result* out, uint64_t n);
int aqc_sync(aqc_ctx* ctx, int slot);
/* The slice of mate's (0 / 1) QUALITY string that the final read of each record keeps: out[i] = start | length << 16.  It is
 * (start, len) of the result record unless the record's quality line is not as long as its sequence line — then every slice
 * of preprocesser.py:19-28,521-524 / barcodeprocesser.py:42-43,66-69 was taken of a string of another length, and the walk's
 * edits (aqc_edit.o) touched the quality string at length - overlap_len + o (mate 0; a negative index wraps the python way) /
 * length - 1 - o (mate 1) of THIS slice. */
int aqc_fetch_quality_views(aqc_ctx* ctx, int slot, int mate, uint32_t* out, uint64_t n);
/* An exception INSIDE the reference's loop ends its run at that record, everything before it having been written: KeyError of
 * util.complement / the error matrix (AQC_ERR_ALPHABET), IndexError of the walk on a short quality line (AQC_ERR_INDEX), int() of
 * a name field for the bubble filter (AQC_ERR_ARG).  After a call on the slot returned one of those, *record is the index in
 * the slot of the EARLIEST record that raises (UINT64_MAX: the error is not tied to a record).  The results of the records
 * before it are valid: aqc_format(ctx, slot, *record, ...) builds exactly what upstream had written when it died. */
int aqc_error_record(aqc_ctx* ctx, int slot, uint64_t* record);
/* duration in ms of the last launch of each kernel on this slot (valid after aqc_sync) */
int aqc_kernel_ms(aqc_ctx* ctx, int slot, float* ms /* [AQC_N_KERNELS] */);
/* HIP-event timing over a region: aqc_timing_reset() starts collecting one event pair per kernel
 * launch on the slot's stream (up to 256 launches per kernel); aqc_timing_mean() waits for the slot
 * and returns the mean launch duration and the number of launches per kernel since the reset. */
int aqc_timing_reset(aqc_ctx* ctx, int slot);
int aqc_timing_mean(aqc_ctx* ctx, int slot, float* mean_ms /* [AQC_N_KERNELS] */, int32_t* launches /* [AQC_N_KERNELS] */);

/* ---- accumulated statistics (host-side merge across GPUs is a plain integer sum) -------------- */
int aqc_get_counters(aqc_ctx* ctx, int64_t* out /* [AQC_N_COUNTERS] */);
/* overlap_histgram / distance_histgram of preprocesser.py:256-258,517,536; n <= AQC_QC_COLS entries each */
int aqc_get_histograms(aqc_ctx* ctx, int64_t* overlap_hist, int64_t* distance_hist, int32_t n);
int aqc_get_qc(aqc_ctx* ctx, int which, int64_t* out /* [AQC_QC_ROWS * AQC_QC_COLS] */);
/* k-mer dictionary of QualityControl `which` (qualitycontrol.py:113-122): keys as 8 raw bytes
 * (zero padded above qc_kmer), counts, and the dict insertion rank (ties in sortKmer keep it).
 * Returns the number of entries through *n (<= cap). */
int aqc_get_kmers(aqc_ctx* ctx, int which, uint64_t* keys, int64_t* counts, uint64_t* order,
                  uint64_t cap, uint64_t* n);

/* ---- text in / text out: FASTQ framing and output formatting on the device --------------------- */
/* One chunk of raw FASTQ text per input file, exactly as read from the (decompressed) file.  aqc_frame
 * replaces fastq.Reader.nextRead (fastq.py:37-49) for every record of the chunk: a record is 4 lines, each
 * readline().rstrip(); a line that is empty after stripping ends that file (eof); a trailing partial record
 * stays in the chunk (bytes >= consumed) and must be presented again at the head of the next chunk unless
 * `final` (nothing follows: an unterminated last line counts as a line, a partial group is dropped).
 * Mate files are read in lock step (preprocesser.py:412-429): n = min(avail1, avail2, max_records).
 * After aqc_frame the slot holds the n records exactly as after aqc_upload (the text is the arena), so
 * aqc_run / aqc_qc_stat / aqc_fetch_results apply.  With cfg.debubble (aqc_set_config before aqc_frame) the
 * lane / tile / x / y of preprocesser.py:180-192 are parsed out of the R1 names on the device.  Chunks must be < 2 GiB. */
typedef struct aqc_text_chunk {
    const uint8_t* text1;
    uint64_t bytes1;
    const uint8_t* text2;   /* NULL: single-end */
    uint64_t bytes2;
    int32_t final1, final2; /* no more bytes will follow this chunk in that file */
    uint64_t max_records;   /* cap on n (UINT64_MAX: none) */
    uint64_t first_index;   /* 0-based global index of the chunk's first record */
} aqc_text_chunk;

typedef struct aqc_frame_info {
    uint64_t n;             /* records now in the slot */
    uint64_t avail1, avail2;       /* complete records each chunk held before its first empty line */
    uint64_t consumed1, consumed2; /* bytes of each chunk that belong to records [0, n) */
    int32_t eof1, eof2;     /* an empty line ended that file inside this chunk (fastq.py:44-47) */
    uint32_t max_len;       /* longest sequence line framed */
    uint32_t next_len1;     /* bases of R1's record n when avail1 > n, else 0: the record the reference has
                               already added to TOTAL_BASES when a shorter mate file ends the loop
                               (preprocesser.py:416-421) */
} aqc_frame_info;

int aqc_frame(aqc_ctx* ctx, int slot, const aqc_text_chunk* chunk, aqc_frame_info* info);
/* aqc_frame for a chunk PARTS of which are already in this GPU's memory (round 6: a `.gz` input decoded on the device, whose text
 * never left HBM — gzip.open(...).readline() of fastq.py:23-24,37-49 without the text crossing PCIe twice).  ext1 / ext2: stretches
 * [offset, offset + bytes) of the chunk of file 1 / 2, sorted by offset, not overlapping, whose bytes are at `device_text` (memory
 * of ctx's device; the caller keeps it valid until the call returns); text1 / text2 hold every OTHER byte of the chunk at its own
 * offset (what they hold inside a stretch is ignored).  last1 / last2: the chunk's last byte of each file (the framing needs it on
 * the host: an unterminated last line of a final chunk is a line).  Everything else as aqc_frame. */
typedef struct aqc_text_extent {
    uint64_t offset, bytes;
    const void* device_text;
} aqc_text_extent;
int aqc_frame_mixed(aqc_ctx* ctx, int slot, const aqc_text_chunk* chunk, const aqc_text_extent* ext1, uint64_t n_ext1, uint8_t last1,
                    const aqc_text_extent* ext2, uint64_t n_ext2, uint8_t last2, aqc_frame_info* info);
/* the same framing again over the text the slot already holds in HBM (no host copy): line index + record framing of the
 * chunk of the last aqc_frame.  For measurements with the input resident in HBM (bench.py). */
int aqc_reframe(aqc_ctx* ctx, int slot, aqc_frame_info* info);
/* seqFilter.writeReads (preprocesser.py:206-232) + fastq.Writer.writeLines (fastq.py:87-93) for records
 * [0, n) of a framed slot after aqc_run: builds, in record order, the text of the good and of the bad output
 * of each file (name / bases / strand line / qualities + "\n"; bad names "@" + FLAG + name[1:]; trimmed
 * slices with the walk's edits applied; with cfg.barcode the names carry the moved barcode,
 * barcodeprocesser.py:34-45).  store_overlap != 0 (--store_overlap, pairs only) adds the third stream: name / last
 * overlap_len bases / strand line / last overlap_len qualities of every good pair with overlap_len > 30 whose
 * mismatches were all corrected (getOverlap, preprocesser.py:78-84,614-616).
 * bytes_out[file * 3 + stream], stream 0 good / 1 bad / 2 overlap.  Index files are not handled here (host side). */
int aqc_format(aqc_ctx* ctx, int slot, uint64_t n, int32_t store_overlap, uint64_t bytes_out[6]);
/* aqc_format WITHOUT the copies nobody needs.  A good record that is written as its own bytes — not trimmed, not renamed, no edit of
 * the walk in it, every line followed directly by its '\n': the bulk of a run without trimming — already stands in the chunk the
 * caller handed to aqc_frame; seqFilter.writeReads (preprocesser.py:206-232) would write those very bytes.  aqc_format_spans leaves
 * such records OUT of the good streams (stream 0 of a file then holds only the good records that had to be rebuilt: trimmed,
 * corrected, renamed) and makes an EVENT per file for every other record of [0, n), in record order: where the record stood in
 * the chunk and what it gives to stream 0.  The good output of a file is then, in order:
 *     chunk bytes [0, e0.in_start) | e0.out_len bytes of stream 0 | chunk bytes [e0.in_start + e0.in_len, e1.in_start) | e1.out_len
 *     bytes of stream 0 | ... | chunk bytes from behind the last event up to the end of record n - 1 (consumed1 / consumed2 of
 *     aqc_frame_info when n is the slot's record count)
 * — byte for byte what aqc_format + aqc_fetch_text(file, 0) hand out.  The bad and overlap streams are as with aqc_format.
 * aqc_pipe_run writes plain-text outputs this way (writev from its page-locked input buffers): PCIe carries the good records one
 * way only.  n_events[file] events; fetch them with aqc_fetch_span_events, the streams with aqc_fetch_streams / aqc_fetch_text. */
typedef struct aqc_span_event {
    uint32_t in_start; /* the record's first byte in its chunk */
    uint32_t in_len;   /* the chunk bytes it takes (up to the next record's first byte) */
    uint32_t out_len;  /* the bytes it contributes to stream 0, in order (0: a bad record) */
} aqc_span_event;
int aqc_format_spans(aqc_ctx* ctx, int slot, uint64_t n, int32_t store_overlap, uint64_t bytes_out[6], uint64_t n_events[2]);
int aqc_fetch_span_events(aqc_ctx* ctx, int slot, int file, aqc_span_event* dst, uint64_t cap);
/* the chunk bytes of each file up to the end of record n - 1 of a framed slot (n == the slot's record count: consumed1 / consumed2):
 * where the last piece of an aqc_format_spans output ends when only the first n records are written */
int aqc_span_end(aqc_ctx* ctx, int slot, uint64_t n, uint64_t end[2]);
/* AQC_FUSED=1 in the environment of aqc_create (opt-in): for 2 x <= 160 pairs framed by aqc_frame, without barcodes,
 * aqc_run's verdict kernel also places every record in its output stream and copies the good
 * records that go out as their own bytes; aqc_format(n = all records, store_overlap = 0) then only rebuilds the rest.  Same bytes
 * either way (seqFilter.writeReads, preprocesser.py:206-232).  1: the slot's last aqc_format took that placement, 0: it did not
 * (not eligible, or the kernel gave the placement up: a deferred pair, CR LF / blank-padded lines), < 0: error. */
int aqc_format_fused(aqc_ctx* ctx, int slot);
/* index files (-7 / -5, preprocesser.py:222-232): the n records framed into `slot` are written WHOLE, routed and (when
 * bad) renamed by the verdicts of the read records of `verdict_slot` (same chunk, same n); the overlap stream takes
 * the whole record wherever the reads' overlap record is written (preprocesser.py:616). */
int aqc_format_plain(aqc_ctx* ctx, int slot, int verdict_slot, uint64_t n, int32_t store_overlap, uint64_t bytes_out[6]);
/* copy one formatted stream (file 0/1, stream 0 good / 1 bad / 2 overlap) to host memory and wait for it */
int aqc_fetch_text(aqc_ctx* ctx, int slot, int file, int stream, uint8_t* dst, uint64_t cap);
/* all six streams of the slot with ONE wait (what the pipe's slot workers call): the text of aqc_format (gz == 0) or the
 * compressed streams of aqc_compress (gz != 0) into dst[file * 3 + stream] (cap[...] bytes each; an empty stream may have a
 * NULL destination). */
int aqc_fetch_streams(aqc_ctx* ctx, int slot, int32_t gz, uint8_t* const dst[6], const uint64_t cap[6]);
/* gzip output built on the device (fastq.Writer with a ".gz" name, fastq.py:65-68; --compression, after.py:91-92): the six
 * formatted streams of the slot become BGZF-compatible gzip members in HBM (<= 0xff00 bytes of text each: one dynamic-Huffman
 * block; matches = runs and "same column, four lines up"; one shared code per stream and call, built by the host from sampled
 * symbol counts; CRC-32 on the device).  gz_bytes_out[file * 3 + stream] = compressed bytes; aqc_fetch_gz copies one
 * compressed stream to host memory and waits for it.  Streams concatenate into valid .gz files; what they decompress to is
 * byte for byte what aqc_fetch_text hands out.  level >= 1 (stored output, level 0, stays with the host writer). */
int aqc_compress(aqc_ctx* ctx, int slot, int32_t level, uint64_t gz_bytes_out[6]);
int aqc_fetch_gz(aqc_ctx* ctx, int slot, int file, int stream, uint8_t* dst, uint64_t cap);
/* gzip INPUT decoded with the device's help (fastq.py:23-24: gzip.open(name, "r"); csrc/aqc_gunzip_dev.hpp): the gzip file at
 * gz[0, size) into out.  Every section the stream can be cut into goes to the GPU `device` in groups of `group_bytes` compressed
 * bytes (0: 256 MiB, and never more than a quarter of the file; sections of `section_bytes`, 0: 1 MiB) — a LANE per deflate block, the blocks found by scanning every bit
 * position — while `threads` host threads resolve the markers and check each member's CRC-32 / ISIZE; what the device does
 * not chain up (final / fixed-Huffman blocks, the stream's last section) is decoded on the host, so ANY valid gzip file
 * comes out exactly.  It is what a `.gz` input of aqc_pipe_run goes through, minus the host pool's share of the sections.
 * stats: sections committed from the device / from the host, bytes decoded sequentially on the host, microseconds in the
 * scan + compact / decode / chain + gather kernels and in the H2D / D2H copies. */
int aqc_gunzip_dev(int device, const uint8_t* gz, uint64_t size, uint8_t* out, uint64_t cap, uint64_t* n_out, uint64_t stats[8], int threads,
                   uint64_t section_bytes, uint64_t group_bytes);
/* page-locked host memory for text chunks and fetched streams (hipHostMalloc): full-rate DMA */
void* aqc_host_alloc(uint64_t bytes);
void aqc_host_free(void* p);

/* ---- function seams (same device code as the hot path, one result per input) ------------------ */
/* util.overlap(r1, r2) (util.py:88-89,158-212) for n pairs -> offset / overlap_len / diff */
int aqc_overlap(aqc_ctx* ctx, const aqc_batch* pairs, int32_t* offset, int32_t* overlap_len, int32_t* diff);
/* hasPolyX(seq, maxPoly, mismatch) (preprocesser.py:30-51) on seq1 -> the byte found or 0 for None;
 * lowQualityNum(read, qual) (:61-68) on qual1; nNumber(read) (:70-76) on seq1 */
int aqc_read_stats(aqc_ctx* ctx, const aqc_batch* reads, int32_t max_poly, int32_t mismatch, int32_t qual,
                   uint8_t* polyx, int32_t* low_qual, int32_t* n_count);
/* util.editDistance(a, b) (util.py:65-83; native twin editdistance/_editdistance.h:16) for n string pairs
 * given as seq1 (a) / seq2 (b) of a batch; strings up to 64 bytes on the device path */
int aqc_edit_distance(aqc_ctx* ctx, const aqc_batch* pairs, int32_t* dist);

/* ---- whole-input pipeline: the byte path of seqFilter.run's main loop (preprocesser.py:411-631) --------------------- */
/* Reads one input (a read file, or a pair of mate files; plain, .gz, or text already in host memory), cuts it into chunks
 * of exactly chunk_records records, deals chunk i to context i % n_ctx (first_index = i * chunk_records, so the sampling
 * rules of preprocesser.py:624 / qualitycontrol.py:343-350 stay exact; no collective), runs aqc_frame ->
 * aqc_run -> aqc_qc_stat(post, first qc_sample records) -> aqc_format -> aqc_fetch_text per chunk on side threads and
 * writes the good / bad / overlap streams in chunk order (fastq.Writer, fastq.py:63-93; .gz output as independent
 * BGZF-compatible members).  The contexts must be configured (aqc_set_config / aqc_set_circles) by the caller, who also
 * merges their statistics afterwards (plain sums).  While a pipe runs, its contexts belong to it.
 * End of input as upstream has it (fastq.py:37-49, preprocesser.py:412-429; round 6: inside the pipe): a line that is empty after
 * rstrip() ends its file there, a trailing partial record is dropped, the first reader to run dry ends the loop — R1 is read first,
 * so when a shorter R2 ends it, R1's next record has already been counted into TOTAL_BASES (result->extra_bases).  The chunk in which
 * the input ends is the run's last; chunks behind it are never run.  result->anomaly is left for shapes the chunking itself cannot
 * describe (a chunk without its records although nothing ended there); callers then rerun the input through the per-chunk calls. */
typedef struct aqc_pipe aqc_pipe;

typedef struct aqc_pipe_io {
    const char* in_path[2];        /* read 1 / read 2 file (NULL: in_mem, or single-end for index 1) */
    const uint8_t* in_mem[2];      /* alternative to a path: the FASTQ text in host memory (page-locked memory from
                                      aqc_host_alloc is used in place, zero copy) */
    uint64_t in_mem_bytes[2];
    int32_t gzip_in[2];            /* 1: the file is a gzip stream (fastq.py:23-24); 2: a bzip2 file (fastq.py:25-26; decoded by libbz2, loaded at run
                                      time, on the pipe's own threads: every stream of the file, streams in parallel) */
    const char* out_path[2][3];    /* per input: good / bad / overlap output file, NULL = that stream is dropped */
    int32_t gzip_out;              /* write .gz (preprocesser.py:318-321) */
    int32_t gzip_level;            /* --compression */
} aqc_pipe_io;

typedef struct aqc_pipe_opts {
    uint64_t chunk_records;        /* records per chunk (0: 131072) */
    int64_t qc_sample;             /* --qc_sample: post-filter QC while TOTAL_READS < qc_sample (<= 0: every record) */
    int32_t store_overlap;         /* --store_overlap */
    int32_t no_output;             /* 1: verdicts and statistics only, no text is formatted or fetched (--qc_only style) */
    uint64_t chunk_index0;         /* this input holds chunks chunk_index0, chunk_index0 + stride, ... of a larger one: */
    uint64_t chunk_index_stride;   /*   first_index = (chunk_index0 + i * stride) * chunk_records   (0 = 1) */
} aqc_pipe_opts;

typedef struct aqc_pipe_result {
    uint64_t records;              /* records processed */
    uint64_t chunks;
    uint64_t bytes_out[6];         /* [file * 3 + stream] */
    int32_t anomaly;               /* the input is not of a shape the pipe takes: outputs and statistics are incomplete */
    int32_t fused_chunks;          /* chunks whose records the verdict kernel placed itself (contexts created with AQC_FUSED=1) */
    uint64_t extra_bases;          /* bases of the R1 record upstream had already read (and counted into TOTAL_BASES, preprocesser.py:416-421)
                                      when a shorter mate file ended its loop; 0 otherwise */
    double seconds;
    /* where the time went, summed over the threads of each kind (seconds): reader file reads / newline counts / waits for a
     * free input buffer; slot workers in aqc_frame (upload + framing) / run + QC + format / waits for an output buffer set /
     * aqc_fetch_text (download); writer commits */
    double t_read, t_count, t_wait_ring, t_frame, t_kernels, t_wait_set, t_fetch, t_write;
} aqc_pipe_result;

int aqc_pipe_create(aqc_ctx** ctxs, int32_t n_ctx, int32_t slots_per_ctx, int32_t io_threads, aqc_pipe** out);
void aqc_pipe_destroy(aqc_pipe* p);
int aqc_pipe_run(aqc_pipe* p, const aqc_pipe_io* io, const aqc_pipe_opts* opts, aqc_pipe_result* result);
const char* aqc_pipe_last_error(void);
/* a read file as a byte stream (fastq.Reader's `self.__file`, fastq.py:23-28), served by the pipe's readers: parallel pread
 * for plain files; for .gz (gzip.open upstream, fastq.py:23-24) the pipe's own decoder: BGZF members inflated independently,
 * any other gzip data — one big member included — by speculative sections on many threads, committed in order (exact:
 * csrc/aqc_gz.hpp), member CRC-32 / length verified, a truncated or corrupt file is an error.
 * aqc_source_read fills dst with the next `want` decompressed bytes and returns their number (< want only at the end of
 * the stream, -1 on a read / format error: aqc_source_error says which). */
typedef struct aqc_source aqc_source;
aqc_source* aqc_source_open(const char* path, int32_t gzip, int32_t io_threads);      /* gzip: 0 plain, 1 gzip, 2 bzip2 */
/* gz_section_bytes: compressed bytes per speculative section (0: chosen from the file size; AQC_GZ_SECTION overrides) */
aqc_source* aqc_source_open2(const char* path, int32_t gzip, int32_t io_threads, uint64_t gz_section_bytes);
int64_t aqc_source_read(aqc_source* s, uint8_t* dst, uint64_t want);
const char* aqc_source_error(aqc_source* s);
/* diagnostics of the parallel gunzip: sections accepted, sections discarded, bytes decoded sequentially instead, bytes out */
int aqc_source_gz_stats(aqc_source* s, uint64_t out[4]);
/* gzip inputs of every aqc_pipe_run of the process so far: sections committed, of them decoded on a GPU (aqc_gunzip_dev.hpp),
 * bytes of text, of them from sections decoded on a GPU.  AQC_GZ_DEVICE_IN=0 keeps gzip input on the host pool. */
int aqc_gz_input_stats(uint64_t out[4]);
void aqc_source_close(aqc_source* s);
/* the codec's pieces on their own (host only; the CPU tests pin them against zlib): one raw DEFLATE stream for src[0, n)
 * (dst must hold n + n / 1000 + 400 bytes; level <= 0 stores), its inverse into exactly `cap` bytes (-1: invalid data or a
 * different length), and the CRC-32 of gzip (crc = 0 to start) */
int aqc_gz_deflate_block(const uint8_t* src, uint64_t n, int32_t level, uint8_t* dst, uint64_t cap, uint64_t* out_n);
int64_t aqc_gz_inflate_raw(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap);
uint32_t aqc_gz_crc32(uint32_t crc, const uint8_t* p, uint64_t n);

/* host-only pieces of the pipe, callable without a GPU (the CPU tests use them):
 * the newline counter the chunk boundaries are found with; BGZF-style gzip members as the pipe's writer makes them (dst must
 * hold n + n / 200 + 64 bytes per 64 KiB block); and the reader half alone — cuts input `file_index` of `io` into chunks of
 * chunk_records records exactly as aqc_pipe_run does (file / gzip / BGZF / memory sources, carry-over, growth) and reports
 * each chunk's bytes and line count plus the CRC-32 of the concatenated (decompressed) chunks */
uint64_t aqc_host_count_newlines(const uint8_t* p, uint64_t n);
int aqc_bgzf_compress(const uint8_t* src, uint64_t n, int32_t level, uint8_t* dst, uint64_t cap, uint64_t* out_n);
int aqc_pipe_split(const aqc_pipe_io* io, int32_t file_index, uint64_t chunk_records, int32_t io_threads, uint64_t* bytes,
                   uint64_t* lines, uint64_t cap, uint64_t* n_chunks, uint32_t* crc);

/* ---- the reference's EXISTING native seam (libed.so), for ABI compatibility ------------------------ */
/* editdistance/_editdistance.h:16 — Levenshtein distance; util.editDistance binds it at util.py:70 */
unsigned int edit_distance(const char* a, const unsigned int asize, const char* b, const unsigned int bsize);
/* editdistance/_editdistance.h:23 — bound by util.overlap_hm_cpp (util.py:223).  r2 is ALREADY reverse-complemented.
 * Implemented with the semantics of the live scan util.overlap_hm (util.py:158-212; the vendored C++ differs from it, see
 * SURVEY.md App. B-7): returns (offset << 8) + diff of the first accepted offset, 0x7FFFFFFF for none.  With
 * (limit_distance, complete_compare_require, overlap_require) = (3, 50, 30) it is util.overlap. */
int seek_overlap(const char* r1, const int len1, const char* r2_revcomp, const int len2, const int limit_distance,
                 const int complete_compare_require, const int overlap_require);

#ifdef __cplusplus
}
#endif
#endif /* AFTERQC_HIP_H */
