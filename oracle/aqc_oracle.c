/*
 * aqc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C scalar restatement of the AfterQC per-read hot path, written to be read side by side
 * with the reference: every function cites the reference file:line it follows and keeps the
 * reference's control flow (including its quirks) instead of any closed form.
 * It is the checker for the HIP path: only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py may load it.  The product (afterqc_amd/) never imports anything from oracle/.
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here against the golden
 * vectors produced by the real reference (tests/golden/make_golden.py), and the end-to-end
 * counters / output digests of 36 reference runs.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC) -> oracle/liboracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/afterqc_hip.h"

#define ORC_MAXLEN (AQC_MAX_READ_LEN + 64)

/* util.py:27 COMP (the '\n' entry cannot occur inside a stripped line) */
static int comp_of(int c) {
    switch (c) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'a': return 't';
        case 't': return 'a';
        case 'c': return 'g';
        case 'g': return 'c';
        case 'N': return 'N';
        default: return -1; /* KeyError in util.complement (util.py:36-37) */
    }
}

/* util.py:42-51 reverseComplement: unknown -> 'N' */
void orc_reverse_complement(const uint8_t* s, int n, uint8_t* out) {
    for (int i = 0; i < n; i++) {
        int c = comp_of(s[n - i - 1]);
        out[i] = (uint8_t)(c < 0 ? 'N' : c);
    }
}

/* util.py:158-212 overlap_hm, loops kept literal (the Python variable `i` survives the for loop) */
void orc_overlap_hm(const uint8_t* r1, int len1, const uint8_t* r2, int len2, int* o_offset, int* o_len, int* o_diff) {
    uint8_t rr2[ORC_MAXLEN];
    orc_reverse_complement(r2, len2, rr2);
    const int limit_distance = 3, overlap_require = 30, complete_compare_require = 50;
    int overlap_len = 0, offset = 0, diff = 0, i = 0;
    /* forward, util.py:172-186 */
    while (offset < len1 - overlap_require) {
        overlap_len = (len1 - offset < len2) ? len1 - offset : len2;
        diff = 0;
        for (int k = 0; k < overlap_len; k++) {
            i = k;
            if (r1[offset + k] != rr2[k]) {
                diff += 1;
                if (diff >= limit_distance && k < complete_compare_require) break;
            }
        }
        if (diff < limit_distance || (diff >= limit_distance && i > complete_compare_require)) {
            *o_offset = offset; *o_len = overlap_len; *o_diff = diff;
            return;
        }
        offset += 1;
    }
    /* reverse, util.py:194-209 */
    offset = 0;
    while (offset > -(len2 - overlap_require)) {
        int a = -offset;
        overlap_len = (len1 < len2 - a) ? len1 : len2 - a;
        diff = 0;
        for (int k = 0; k < overlap_len; k++) {
            i = k;
            if (r1[k] != rr2[a + k]) {
                diff += 1;
                if (diff >= limit_distance && k < complete_compare_require) break;
            }
        }
        if (diff < limit_distance || (diff >= limit_distance && i > complete_compare_require)) {
            *o_offset = offset; *o_len = overlap_len; *o_diff = diff;
            return;
        }
        offset -= 1;
    }
    *o_offset = 0; *o_len = 0; *o_diff = 0; /* util.py:212 */
}

/* preprocesser.py:30-51 hasPolyX -> the byte, or 0 for None */
int orc_has_polyx(const uint8_t* seq, int len, int maxPoly, int mismatch) {
    if (len < maxPoly) return 0;
    static const char polyArray[9] = {'A', 'T', 'C', 'G', 'a', 't', 'c', 'g', 'N'};
    int polyCount[9] = {0};
    for (int x = 0; x < len; x++) {
        int f = -1;
        for (int k = 0; k < 9; k++) if (seq[x] == (uint8_t)polyArray[k]) f = k;
        if (f < 0) return 0;
        if (x >= maxPoly) {
            int t = -1;
            for (int k = 0; k < 9; k++) if (seq[x - maxPoly] == (uint8_t)polyArray[k]) t = k;
            polyCount[t] -= 1;
        }
        polyCount[f] += 1;
        if (polyCount[f] >= maxPoly - mismatch) return seq[x];
    }
    return 0;
}

/* preprocesser.py:61-68 */
int orc_low_quality_num(const uint8_t* qual, int len, int q) {
    q += 33;
    int n = 0;
    for (int i = 0; i < len; i++) if ((int)qual[i] < q) n++;
    return n;
}

/* preprocesser.py:70-76 */
int orc_n_number(const uint8_t* seq, int len) {
    int n = 0;
    for (int i = 0; i < len; i++) if (seq[i] == 'N') n++;
    return n;
}

/* util.py:72-83 (the DP fallback of editDistance); also the semantics of
 * editdistance/_editdistance.cpp:100-126 for non-empty inputs */
int orc_edit_distance(const uint8_t* a, int la, const uint8_t* b, int lb) {
    int m = la + 1, n = lb + 1;
    int* tbl = (int*)malloc(sizeof(int) * (size_t)m * (size_t)n);
    for (int i = 0; i < m; i++) tbl[i * n] = i;
    for (int j = 0; j < n; j++) tbl[j] = j;
    for (int i = 1; i < m; i++)
        for (int j = 1; j < n; j++) {
            int cost = a[i - 1] == b[j - 1] ? 0 : 1;
            int v = tbl[i * n + j - 1] + 1;
            if (tbl[(i - 1) * n + j] + 1 < v) v = tbl[(i - 1) * n + j] + 1;
            if (tbl[(i - 1) * n + j - 1] + cost < v) v = tbl[(i - 1) * n + j - 1] + cost;
            tbl[i * n + j] = v;
        }
    int r = tbl[(m - 1) * n + (n - 1)];
    free(tbl);
    return r;
}

/* barcodeprocesser.py:9-14 */
static int diff_number(const uint8_t* a, const uint8_t* b, int n) {
    int d = 0;
    for (int i = 0; i < n; i++) if (a[i] != b[i]) d++;
    return d;
}

/* barcodeprocesser.py:19-32 (a window that runs off the end of seq is shorter than verify in
 * Python: diffNumber then indexes past the slice -> cannot happen because len > verifyLen+barcodeLen+1) */
int orc_detect_barcode(const uint8_t* seq, int len, int barcodeLen, const uint8_t* verify, int verifyLen) {
    if (len <= verifyLen + barcodeLen + 1) return 0;
    if (diff_number(seq + barcodeLen, verify, verifyLen) <= 1) return barcodeLen;
    if (diff_number(seq + barcodeLen - 1, verify, verifyLen) == 0) return barcodeLen - 1;
    if (diff_number(seq + barcodeLen + 1, verify, verifyLen) == 0) return barcodeLen + 1;
    return 0;
}

/* barcodeprocesser.py:47-75 cleanBarcodeTail -> number of bases cut from both tails.
 * readStart1/2 = seq[0:barcodeLen] + verify (barcodeprocesser.py:78-79). */
int orc_clean_barcode_tail(const uint8_t* s1, int r1len, const uint8_t* s2, int r2len, const uint8_t* readStart1,
                           int n1, const uint8_t* readStart2, int n2) {
    uint8_t reverse1[128], reverse2[128];
    orc_reverse_complement(readStart1, n1, reverse1);
    orc_reverse_complement(readStart2, n2, reverse2);
    int barcodeStringLen = n1 < n2 ? n1 : n2;
    for (int i = 0; i < barcodeStringLen; i++) {
        int compLen = barcodeStringLen - i;
        if (compLen >= r1len || compLen >= r2len) continue;
        /* read1[1][-compLen:] vs reverse2[i:]  (reverse2 has n2 chars) */
        int d1 = orc_edit_distance(s1 + r1len - compLen, compLen, reverse2 + i, n2 - i);
        int d2 = orc_edit_distance(s2 + r2len - compLen, compLen, reverse1 + i, n1 - i);
        /* threshold = compLen/5: float in py3, int floor in py2; both compare identically with ints */
        if (d1 * 5 <= compLen && d2 * 5 <= compLen) return compLen;
    }
    return 0;
}

/* preprocesser.py:176-204 isInBubble, given the integers parsed from the name */
int orc_in_bubble(int lane, int tile, int x, int y, const double* cx, const double* cy, const double* cr,
                  const int32_t* clane, const int32_t* ctile, int n) {
    for (int i = 0; i < n; i++) {
        if (ctile[i] != tile) continue;
        if (clane[i] == lane) {
            double dx = cx[i] - (double)x, dy = cy[i] - (double)y;
            double lhs = dx * dx;
            double t = dy * dy;
            lhs = lhs + t;
            if (lhs < cr[i] * cr[i]) return 1;
        }
    }
    return 0;
}

/* base index in ALL_BASES = ("A","T","C","G") (qualitycontrol.py:24), -1 if not one of them */
static int base_idx(int c) {
    switch (c) {
        case 'A': return 0;
        case 'T': return 1;
        case 'C': return 2;
        case 'G': return 3;
        default: return -1;
    }
}

typedef struct orc_circles {
    const double *cx, *cy, *cr;
    const int32_t *lane, *tile;
    int32_t n;
} orc_circles;

/*
 * One record through preprocesser.py:436-631.  Returns the verdict; fills the result record and
 * (if accumulate) the counters / histograms.  Strings are copied so edits can be applied in place
 * exactly like the reference does.
 */
static int process_record(const aqc_config* cfg, const orc_circles* circ, const uint8_t* s1o, const uint8_t* q1o,
                          int l1o, const uint8_t* s2o, const uint8_t* q2o, int l2o, int aux_ok, int lane, int tile,
                          int x, int y, aqc_result* res, int64_t* C, int64_t* ovl_hist, int64_t* dist_hist,
                          int* status) {
    uint8_t s1[ORC_MAXLEN], q1[ORC_MAXLEN], s2[ORC_MAXLEN], q2[ORC_MAXLEN];
    int paired = cfg->paired;
    int len1 = l1o, len2 = paired ? l2o : 0;
    int start1 = 0, start2 = 0;
    memcpy(s1, s1o, (size_t)l1o); memcpy(q1, q1o, (size_t)l1o);
    if (paired) { memcpy(s2, s2o, (size_t)l2o); memcpy(q2, q2o, (size_t)l2o); }
    memset(res, 0, sizeof(*res));

    /* preprocesser.py:416,431 */
    C[AQC_C_TOTAL_BASES] += l1o;
    if (paired && cfg->count_r2_bases) C[AQC_C_TOTAL_BASES] += l2o;
    C[AQC_C_TOTAL_READS] += 1;

#define FINISH(FLAG)                                                              \
    do {                                                                          \
        res->flag = (uint8_t)(FLAG);                                              \
        res->start1 = (uint16_t)start1; res->len1 = (uint16_t)len1;               \
        res->start2 = (uint16_t)start2; res->len2 = (uint16_t)len2;               \
        C[AQC_C_FLAG0 + (FLAG)] += 1;                                             \
        return (FLAG);                                                            \
    } while (0)

    /* barcode, preprocesser.py:436-452 */
    if (cfg->barcode) {
        int bl = cfg->barcode_length, vl = cfg->barcode_verify_len;
        int b1 = orc_detect_barcode(s1, len1, bl, cfg->barcode_verify, vl);
        if (b1 == 0) FINISH(AQC_BADBCD1);
        res->barcode = (uint8_t)(b1 - bl + 2);
        if (!paired) {
            /* moveBarcodeToName, barcodeprocesser.py:34-45 (the name itself is rewritten by the host).
             * Single-end passes the DESIGN length, not the detected one (preprocesser.py:444). */
            int rm = vl + bl;
            start1 += rm; len1 = len1 - rm > 0 ? len1 - rm : 0;
            memmove(s1, s1 + rm, (size_t)len1); memmove(q1, q1 + rm, (size_t)len1);
        } else {
            int b2 = orc_detect_barcode(s2, len2, bl, cfg->barcode_verify, vl);
            if (b2 == 0) FINISH(AQC_BADBCD2);
            res->barcode |= (uint8_t)((b2 - bl + 2) << 4);
            /* moveAndTrimPair, barcodeprocesser.py:77-82 */
            uint8_t rs1[64], rs2[64];
            memcpy(rs1, s1, (size_t)b1); memcpy(rs1 + b1, cfg->barcode_verify, (size_t)vl);
            memcpy(rs2, s2, (size_t)b2); memcpy(rs2 + b2, cfg->barcode_verify, (size_t)vl);
            int rm1 = vl + b1, rm2 = vl + b2;
            start1 += rm1; len1 = len1 - rm1 > 0 ? len1 - rm1 : 0;
            memmove(s1, s1 + rm1, (size_t)len1); memmove(q1, q1 + rm1, (size_t)len1);
            start2 += rm2; len2 = len2 - rm2 > 0 ? len2 - rm2 : 0;
            memmove(s2, s2 + rm2, (size_t)len2); memmove(q2, q2 + rm2, (size_t)len2);
            int cut = orc_clean_barcode_tail(s1, len1, s2, len2, rs1, b1 + vl, rs2, b2 + vl);
            len1 -= cut; len2 -= cut;
        }
    }

    /* trim, preprocesser.py:455-466 with Python slice semantics of trim() :19-28 */
    if (cfg->trim_front > 0 || cfg->trim_tail > 0) {
        int f = cfg->trim_front, t = cfg->trim_tail;
        int end = t > 0 ? (len1 - t > 0 ? len1 - t : 0) : len1;
        int st = f < len1 ? f : len1; /* f >= 0 here (auto-trim resolved by the host) */
        int nl = end - st > 0 ? end - st : 0;
        memmove(s1, s1 + st, (size_t)nl); memmove(q1, q1 + st, (size_t)nl);
        start1 += st; len1 = nl;
        if (len1 < 5) FINISH(AQC_BADTRIM1);
        if (paired) {
            f = cfg->trim_front2; t = cfg->trim_tail2;
            end = t > 0 ? (len2 - t > 0 ? len2 - t : 0) : len2;
            st = f < len2 ? f : len2;
            nl = end - st > 0 ? end - st : 0;
            memmove(s2, s2 + st, (size_t)nl); memmove(q2, q2 + st, (size_t)nl);
            start2 += st; len2 = nl;
            if (len2 < 5) FINISH(AQC_BADTRIM2);
        }
    }

    /* bubble, preprocesser.py:469-473 */
    if (cfg->debubble) {
        if (aux_ok && orc_in_bubble(lane, tile, x, y, circ->cx, circ->cy, circ->cr, circ->lane, circ->tile, circ->n))
            FINISH(AQC_BADBBL);
    }

    /* length, preprocesser.py:476-479 (R1 only) */
    if (len1 < cfg->seq_len_req) FINISH(AQC_BADLEN);

    /* polyX, preprocesser.py:482-490 */
    if (cfg->poly_size_limit > 0) {
        int p1 = orc_has_polyx(s1, len1, cfg->poly_size_limit, cfg->allow_mismatch_in_poly);
        int p2 = 0;
        if (paired) p2 = orc_has_polyx(s2, len2, cfg->poly_size_limit, cfg->allow_mismatch_in_poly);
        if (p1 != 0 || p2 != 0) FINISH(AQC_BADPOL);
    }

    /* low quality, preprocesser.py:493-501: only lowQual1 is tested (upstream quirk) */
    if (cfg->unqualified_base_limit > 0) {
        int lq1 = orc_low_quality_num(q1, len1, cfg->qualified_quality_phred);
        if (lq1 > cfg->unqualified_base_limit || lq1 > cfg->unqualified_base_limit) FINISH(AQC_BADLQC);
    }

    /* N, preprocesser.py:504-512 */
    if (cfg->n_base_limit > 0) {
        int n1 = orc_n_number(s1, len1), n2 = 0;
        if (paired) n2 = orc_n_number(s2, len2);
        if (n1 > cfg->n_base_limit || n2 > cfg->n_base_limit) FINISH(AQC_BADNCT);
    }

    /* overlap + correction, preprocesser.py:515-617 */
    if (paired && !cfg->no_overlap) {
        int offset, overlap_len, distance;
        orc_overlap_hm(s1, len1, s2, len2, &offset, &overlap_len, &distance);
        if (ovl_hist) ovl_hist[overlap_len] += 1;
        if (offset < 0 && overlap_len > 30) {
            len1 = overlap_len; /* r1[1][0:overlap_len] etc.: overlap_len <= len1, len2 here */
            len2 = overlap_len;
            C[AQC_C_TRIMMED_ADAPTER_BASE] += 2 * (offset < 0 ? -offset : offset);
            C[AQC_C_TRIMMED_ADAPTER_READ] += 1;
            if (len1 < cfg->seq_len_req) FINISH(AQC_BADLEN);
            orc_overlap_hm(s1, len1, s2, len2, &offset, &overlap_len, &distance);
        }
        res->offset = (int16_t)offset; res->overlap_len = (uint16_t)overlap_len; res->distance = (uint16_t)distance;
        if (dist_hist) dist_hist[distance] += 1;
        if (distance > 3) FINISH(AQC_BADDIFF);
        if (overlap_len > 30) {
            C[AQC_C_OVERLAPPED] += 1;
            C[AQC_C_OVERLAP_LEN_SUM] += overlap_len;
            C[AQC_C_OVERLAP_BASE_SUM] += overlap_len * 2;
            C[AQC_C_OVERLAP_BASE_ERR] += distance;
            int corrected = 0, zero_qual_masked = 0, skipped_mismatch = 0;
            if (distance > 0) {
                int64_t err_mtx[16] = {0};
                for (int o = 0; o < overlap_len; o++) {
                    int p1 = len1 - overlap_len + o, p2 = len2 - o - 1;
                    int b1 = s1[p1];
                    int b2 = comp_of(s2[p2]);
                    if (b2 < 0) { *status = AQC_ERR_ALPHABET; FINISH(AQC_BADMISMATCH); }
                    int qa = q1[p1], qb = q2[p2];
                    if (b1 != b2) {
                        int this_is_corrected = 0;
                        if (qa - 33 >= 30 && qb - 33 <= 14) {
                            if (b1 != 'N' && b2 != 'N') {
                                int cb1 = comp_of(b1), cb2 = comp_of(b2);
                                if (cb1 < 0 || base_idx(cb1) < 0 || base_idx(cb2) < 0) { *status = AQC_ERR_ALPHABET; FINISH(AQC_BADMISMATCH); }
                                err_mtx[base_idx(cb1) * 4 + base_idx(cb2)] += 1;
                            }
                            if (!cfg->no_correction) {
                                int cb1 = comp_of(b1);
                                if (cb1 < 0) { *status = AQC_ERR_ALPHABET; FINISH(AQC_BADMISMATCH); }
                                s2[p2] = (uint8_t)cb1; q2[p2] = (uint8_t)qa;
                                res->edits[corrected + zero_qual_masked] = (aqc_edit){(uint16_t)o, AQC_EDIT_FIX_R2, (uint8_t)cb1, (uint8_t)qa};
                                corrected += 1; this_is_corrected = 1;
                            }
                        } else if (qb - 33 >= 30 && qa - 33 <= 14) {
                            if (b1 != 'N' && b2 != 'N') {
                                if (base_idx(b2) < 0 || base_idx(b1) < 0) { *status = AQC_ERR_ALPHABET; FINISH(AQC_BADMISMATCH); }
                                err_mtx[base_idx(b2) * 4 + base_idx(b1)] += 1;
                            }
                            if (!cfg->no_correction) {
                                s1[p1] = (uint8_t)b2; q1[p1] = (uint8_t)qb;
                                res->edits[corrected + zero_qual_masked] = (aqc_edit){(uint16_t)o, AQC_EDIT_FIX_R1, (uint8_t)b2, (uint8_t)qb};
                                corrected += 1; this_is_corrected = 1;
                            }
                        }
                        if (!this_is_corrected) {
                            if (cfg->mask_mismatch) {
                                q2[p2] = '!'; q1[p1] = '!';
                                res->edits[corrected + zero_qual_masked] = (aqc_edit){(uint16_t)o, AQC_EDIT_MASK, 0, '!'};
                                zero_qual_masked += 1;
                            } else {
                                skipped_mismatch += 1;
                            }
                        }
                        res->n_edits = (uint8_t)(corrected + zero_qual_masked);
                        if (corrected + zero_qual_masked + skipped_mismatch >= distance) break;
                    }
                }
                if (corrected + zero_qual_masked + skipped_mismatch == distance) {
                    for (int k = 0; k < 16; k++) C[AQC_C_ERR_MATRIX0 + k] += err_mtx[k];
                    if (corrected > 0) C[AQC_C_READ_CORRECTED] += 1;
                    C[AQC_C_BASE_CORRECTED] += corrected;
                    C[AQC_C_BASE_ZERO_QUAL_MASKED] += zero_qual_masked * 2;
                    C[AQC_C_BASE_SKIPPED_CORRECTION] += skipped_mismatch * 2;
                } else {
                    FINISH(AQC_BADMISMATCH);
                }
            }
        }
    }

    /* good, preprocesser.py:620-629 */
    C[AQC_C_GOOD_BASES] += len1;
    if (paired && cfg->count_r2_bases) C[AQC_C_GOOD_BASES] += len2;
    C[AQC_C_GOOD_READS] += 1;
    FINISH(AQC_GOOD);
#undef FINISH
}

/* The batch twin of aqc_upload + aqc_run + aqc_fetch_results + aqc_get_counters/histograms. */
int orc_process_batch(const aqc_config* cfg, const aqc_batch* b, const double* cx, const double* cy, const double* cr,
                      const int32_t* clane, const int32_t* ctile, int32_t ncircles, aqc_result* results,
                      int64_t* counters, int64_t* ovl_hist, int64_t* dist_hist, uint64_t accum_limit) {
    orc_circles circ = {cx, cy, cr, clane, ctile, ncircles};
    int status = 0;
    int64_t scratch[AQC_N_COUNTERS];
    for (uint64_t i = 0; i < b->n; i++) {
        if (b->len1[i] > AQC_MAX_READ_LEN || (cfg->paired && b->len2[i] > AQC_MAX_READ_LEN)) return AQC_ERR_READ_TOO_LONG;
        int acc = i < accum_limit;
        int64_t* C = counters;
        if (!acc) { memset(scratch, 0, sizeof(scratch)); C = scratch; }
        const uint64_t* qo1 = b->qoff1 ? b->qoff1 : b->off1;
        const uint64_t* qo2 = b->qoff2 ? b->qoff2 : b->off2;
        process_record(cfg, &circ, b->seq1 + b->off1[i], b->qual1 + qo1[i], (int)b->len1[i],
                       cfg->paired ? b->seq2 + b->off2[i] : 0, cfg->paired ? b->qual2 + qo2[i] : 0,
                       cfg->paired ? (int)b->len2[i] : 0, b->aux_ok ? b->aux_ok[i] : 0, b->aux_lane ? b->aux_lane[i] : 0,
                       b->aux_tile ? b->aux_tile[i] : 0, b->aux_x ? b->aux_x[i] : 0, b->aux_y ? b->aux_y[i] : 0,
                       &results[i], C, acc ? ovl_hist : 0, acc ? dist_hist : 0, &status);
    }
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * QualityControl.statRead (qualitycontrol.py:73-122) with an insertion-ordered k-mer dictionary.
 * ------------------------------------------------------------------------------------------------ */
typedef struct orc_kmer_entry {
    uint8_t key[16];
    int64_t count;
    uint64_t order;
    int used;
} orc_kmer_entry;

typedef struct orc_qc {
    int64_t acc[AQC_QC_ROWS * AQC_QC_COLS];
    int kmer_len;
    orc_kmer_entry* tab;
    uint64_t cap, n;
} orc_qc;

orc_qc* orc_qc_new(int kmer_len) {
    orc_qc* q = (orc_qc*)calloc(1, sizeof(orc_qc));
    q->kmer_len = kmer_len;
    q->cap = 1u << 16;
    q->tab = (orc_kmer_entry*)calloc(q->cap, sizeof(orc_kmer_entry));
    return q;
}

void orc_qc_free(orc_qc* q) {
    if (q) { free(q->tab); free(q); }
}

static uint64_t kmer_hash(const uint8_t* k, int n) {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < n; i++) { h ^= k[i]; h *= 1099511628211ull; }
    return h;
}

static orc_kmer_entry* kmer_find(orc_qc* q, const uint8_t* k, int insert);

static void kmer_grow(orc_qc* q) {
    orc_kmer_entry* old = q->tab;
    uint64_t oc = q->cap;
    q->cap *= 2;
    q->tab = (orc_kmer_entry*)calloc(q->cap, sizeof(orc_kmer_entry));
    q->n = 0;
    for (uint64_t i = 0; i < oc; i++)
        if (old[i].used) {
            orc_kmer_entry* e = kmer_find(q, old[i].key, 1);
            e->count = old[i].count; e->order = old[i].order;
        }
    free(old);
}

static orc_kmer_entry* kmer_find(orc_qc* q, const uint8_t* k, int insert) {
    uint64_t m = q->cap - 1, h = kmer_hash(k, q->kmer_len) & m;
    while (q->tab[h].used) {
        if (memcmp(q->tab[h].key, k, (size_t)q->kmer_len) == 0) return &q->tab[h];
        h = (h + 1) & m;
    }
    if (!insert) return 0;
    q->tab[h].used = 1;
    memcpy(q->tab[h].key, k, (size_t)q->kmer_len);
    q->tab[h].count = 0;
    q->tab[h].order = 0;
    q->n++;
    return &q->tab[h];
}

/* t0: global scan time of base 0 of this read (monotonic over the reference's sequential order); entries
 * remember 2*(t0+i) when first scanned / 2*(t0+i)+1 when first inserted as a reverse complement, which is the
 * dict insertion order and — unlike a local rank — can be merged across shards with min(). */
int orc_qc_stat_read(orc_qc* qc, const uint8_t* seq, const uint8_t* qual, int seqlen, uint64_t t0) {
    if (seqlen > AQC_MAX_READ_LEN) return AQC_ERR_READ_TOO_LONG;
    if (seqlen < 5 && seqlen > 0) return AQC_ERR_ARG; /* seq[j+1] IndexError upstream (qualitycontrol.py:106-107) */
    int64_t* A = qc->acc;
    int gc = 0;
    for (int i = 0; i < seqlen; i++) {
        A[AQC_QC_TOTAL_NUM * AQC_QC_COLS + i] += 1;
        int qnum = (int)qual[i] - 33;
        A[AQC_QC_TOTAL_QUAL * AQC_QC_COLS + i] += qnum;
        int b = seq[i];
        if (b == 'G' || b == 'C') gc += 1;
        int bi = base_idx(b);
        if (bi >= 0) {
            A[(AQC_QC_BASE_COUNT_A + bi) * AQC_QC_COLS + i] += 1;
            A[(AQC_QC_BASE_QUAL_A + bi) * AQC_QC_COLS + i] += qnum;
        }
        int left = i - 2, right = i + 3;
        if (left < 0) { left = 0; right = 5; }
        else if (right >= seqlen) { right = seqlen; left = seqlen - 5; }
        int disc = 0;
        for (int j = left; j < right - 1; j++) if (seq[j] != seq[j + 1]) disc += 1;
        A[AQC_QC_DISCONTINUITY * AQC_QC_COLS + i] += disc;
    }
    A[AQC_QC_GC_HIST * AQC_QC_COLS + gc] += 1;
    A[AQC_QC_SCALARS * AQC_QC_COLS + 1] += 1;
    int k = qc->kmer_len;
    for (int i = 0; i < seqlen - k; i++) {
        A[AQC_QC_SCALARS * AQC_QC_COLS + 0] += 1; /* totalKmer */
        if (qc->n * 2 + 4 > qc->cap) kmer_grow(qc);
        orc_kmer_entry* e = kmer_find(qc, seq + i, 0);
        if (e) {
            e->count += 1;
        } else {
            e = kmer_find(qc, seq + i, 1);
            e->count = 1;
            e->order = 2 * (t0 + (uint64_t)i);
            uint8_t rc[16];
            orc_reverse_complement(seq + i, k, rc);
            if (!kmer_find(qc, rc, 0)) {
                orc_kmer_entry* r = kmer_find(qc, rc, 1);
                r->count = 0;
                r->order = 2 * (t0 + (uint64_t)i) + 1;
            }
        }
    }
    return 0;
}

void orc_qc_get(const orc_qc* qc, int64_t* out) { memcpy(out, qc->acc, sizeof(qc->acc)); }

uint64_t orc_qc_kmer_count(const orc_qc* qc) { return qc->n; }

/* dump the dictionary sorted by insertion rank; keys are kmer_len bytes each (packed) */
static int cmp_order(const void* a, const void* b) {
    const orc_kmer_entry *x = (const orc_kmer_entry*)a, *y = (const orc_kmer_entry*)b;
    return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

uint64_t orc_qc_get_kmers(const orc_qc* qc, uint8_t* keys, int64_t* counts, uint64_t* orders, uint64_t cap) {
    orc_kmer_entry* tmp = (orc_kmer_entry*)malloc(sizeof(orc_kmer_entry) * (qc->n + 1));
    uint64_t m = 0;
    for (uint64_t i = 0; i < qc->cap; i++) if (qc->tab[i].used) tmp[m++] = qc->tab[i];
    qsort(tmp, m, sizeof(orc_kmer_entry), cmp_order);
    uint64_t w = m < cap ? m : cap;
    for (uint64_t i = 0; i < w; i++) {
        memcpy(keys + i * (uint64_t)qc->kmer_len, tmp[i].key, (size_t)qc->kmer_len);
        counts[i] = tmp[i].count;
        if (orders) orders[i] = tmp[i].order;
    }
    free(tmp);
    return m;
}
