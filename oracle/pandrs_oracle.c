/*
 * pandrs_oracle.c — CPU restatement of the reference's groupby-aggregate / hash-join path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pandrs_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * or as the timed CPU baseline — never as the product path.
 *
 * The reference (cool-japan/pandrs) is Rust and cannot be built here (no cargo/rustc), so
 * this is a restatement that follows the cited source lines; it is pinned against the
 * reference's own known-answer tests in tests/golden/ (see tests/test_oracle_golden.py).
 *
 * Two groupby implementations, cross-checked against each other in tests/:
 *   oracle_groupby_agg        typed keys, sort-based grouping, folds in ascending row order —
 *                             same results as the reference, output sorted by key.
 *   oracle_groupby_agg_ref    "faithful" shape of the reference algorithm: per-row key
 *                             stringification, HashMap<Vec<String>,Vec<usize>> (SipHash-1-3),
 *                             per-group gather + fold.  This is what bench.py times as the
 *                             CPU baseline ("port").
 *
 * Reference lines followed:
 *   key formation, null => own group        src/optimized/split_dataframe/group/grouping.rs:62-104
 *   fold semantics per (dtype, op)          src/optimized/split_dataframe/group/aggregation.rs:500-754
 *   variance / std                          src/optimized/split_dataframe/group/aggregation.rs:875-903
 *   LazyFrame inline copy                   src/optimized/lazy.rs:186-404
 *   join indices                            src/optimized/split_dataframe/join.rs:106-224
 *   join gathers (fill 0 / 0.0 / false)     src/optimized/split_dataframe/join.rs:296-357
 *   null bitmask (LSB first, 1 = null)      src/core/column.rs:163-177
 *   whole-column reductions                 src/optimized/jit/simd.rs:9-112, :116-199, :290-333
 *                                           src/optimized/jit/parallel.rs:71-102 (Kahan)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pandrs_hip.h" /* enums + column/agg structs only */

typedef pandrs_hip_column ocol;
typedef pandrs_hip_agg_spec oagg;

/* ---------------------------------------------------------------- helpers */

static inline int is_null(const uint8_t *mask, int64_t i) {
    /* src/column/int64_column.rs:108-126 — bit set => None */
    return mask && ((mask[i >> 3] >> (i & 7)) & 1);
}

static const uint64_t CANON_NAN = 0x7FF8000000000000ull;

/* 8-byte key cell: i64 value / f64 bits (NaNs collapsed: val.to_string() == "NaN" for every
 * NaN, grouping.rs:79) / zero-extended u32 code / bool bit. */
static inline uint64_t key_cell(const ocol *c, int64_t i) {
    switch (c->dtype) {
    case PANDRS_HIP_I64: return (uint64_t)((const int64_t *)c->data)[i];
    case PANDRS_HIP_F64: {
        uint64_t b; memcpy(&b, (const double *)c->data + i, 8);
        if ((b & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) b = CANON_NAN;
        return b;
    }
    case PANDRS_HIP_U32CODE: return ((const uint32_t *)c->data)[i];
    case PANDRS_HIP_BOOLBITS: return (((const uint8_t *)c->data)[i >> 3] >> (i & 7)) & 1;
    case PANDRS_HIP_CELL64: return ((const uint64_t *)c->data)[i];   /* already a cell (multi-GPU shuffle output) */
    }
    return 0;
}

/* total order used only to make the oracle's OUTPUT order deterministic (the reference's is
 * HashMap order): nulls last; i64 signed; f64 by IEEE total order of the cell; codes/bools
 * unsigned. */
static inline uint64_t sortable(int dtype, uint64_t cell) {
    if (dtype == PANDRS_HIP_I64) return cell ^ 0x8000000000000000ull;
    if (dtype == PANDRS_HIP_F64) return (cell >> 63) ? ~cell : (cell | 0x8000000000000000ull);
    return cell;
}

typedef struct {
    const ocol *keys; int n_keys;
} sort_ctx;

static int cmp_rows(const void *pa, const void *pb, void *vctx) {
    const sort_ctx *c = (const sort_ctx *)vctx;
    int64_t a = *(const int64_t *)pa, b = *(const int64_t *)pb;
    for (int k = 0; k < c->n_keys; k++) {
        int na = is_null(c->keys[k].null_mask, a), nb = is_null(c->keys[k].null_mask, b);
        if (na != nb) return na - nb;
        if (na) continue;
        uint64_t ka = sortable(c->keys[k].dtype, key_cell(&c->keys[k], a));
        uint64_t kb = sortable(c->keys[k].dtype, key_cell(&c->keys[k], b));
        if (ka != kb) return ka < kb ? -1 : 1;
    }
    return a < b ? -1 : (a > b ? 1 : 0); /* ascending row order inside a group */
}

static int same_group(const sort_ctx *c, int64_t a, int64_t b) {
    for (int k = 0; k < c->n_keys; k++) {
        int na = is_null(c->keys[k].null_mask, a), nb = is_null(c->keys[k].null_mask, b);
        if (na != nb) return 0;
        if (na) continue;
        if (key_cell(&c->keys[k], a) != key_cell(&c->keys[k], b)) return 0;
    }
    return 1;
}

/* aggregation.rs:881-903 */
static double variance_of(const double *v, int64_t n) {
    if (n == 0) return 0.0;
    double nn = (double)n, s = 0.0;
    for (int64_t i = 0; i < n; i++) s += v[i];
    double mean = s / nn, ss = 0.0;
    for (int64_t i = 0; i < n; i++) { double d = v[i] - mean; ss += d * d; }
    return n > 1 ? ss / (nn - 1.0) : 0.0;
}

static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static int cmp_f64_partial(const void *a, const void *b) {
    /* aggregation.rs:714: partial_cmp().unwrap_or(Equal) */
    double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* Rust f64::min / f64::max ignore a NaN operand (aggregation.rs:653, :666).  Ties between
 * +0.0 and -0.0 are unspecified in the reference (llvm.minnum); the oracle and the engine both
 * resolve them by IEEE total order (min -> -0.0, max -> +0.0). */
static inline double rust_min(double a, double b) {
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    if (a == b) return signbit(a) ? a : b;
    return a < b ? a : b;
}
static inline double rust_max(double a, double b) {
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    if (a == b) return signbit(a) ? b : a;
    return a > b ? a : b;
}

/* GroupBy::calculate_aggregation, aggregation.rs:500-754.  rows = ascending row indices of
 * one group.  Returns 0 ok, else a pandrs_hip_status. */
static int fold_group(const ocol *col, int op, const int64_t *rows, int64_t n, double *out,
                      double *scratch /* n doubles */) {
    if (op == PANDRS_HIP_AGG_COUNT) { *out = (double)n; return 0; }   /* :743 counts nulls too */
    if (op == PANDRS_HIP_AGG_CUSTOM) return PANDRS_HIP_ERR_OPERATION_FAILED; /* :744 */
    if (col->dtype == PANDRS_HIP_I64) {
        const int64_t *d = (const int64_t *)col->data; const uint8_t *m = col->null_mask;
        switch (op) {
        case PANDRS_HIP_AGG_SUM: {           /* :507-515, wrapping i64 (release build) */
            uint64_t s = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) s += (uint64_t)d[rows[i]];
            *out = (double)(int64_t)s; return 0; }
        case PANDRS_HIP_AGG_MEAN: {          /* :516-530 */
            uint64_t s = 0; int64_t c = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) { s += (uint64_t)d[rows[i]]; c++; }
            *out = c > 0 ? (double)(int64_t)s / (double)c : 0.0; return 0; }
        case PANDRS_HIP_AGG_MIN: {           /* :531-543 — sentinel unchanged => 0.0 */
            int64_t v = INT64_MAX;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i]) && d[rows[i]] < v) v = d[rows[i]];
            *out = v == INT64_MAX ? 0.0 : (double)v; return 0; }
        case PANDRS_HIP_AGG_MAX: {           /* :544-556 */
            int64_t v = INT64_MIN;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i]) && d[rows[i]] > v) v = d[rows[i]];
            *out = v == INT64_MIN ? 0.0 : (double)v; return 0; }
        case PANDRS_HIP_AGG_STD: case PANDRS_HIP_AGG_VAR: {   /* :557-584 */
            int64_t c = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) scratch[c++] = (double)d[rows[i]];
            double var = c ? variance_of(scratch, c) : 0.0;
            *out = op == PANDRS_HIP_AGG_STD ? sqrt(var) : var; return 0; }
        case PANDRS_HIP_AGG_MEDIAN: {        /* :585-604 — the add of the two middles is in i64 */
            int64_t *iv = (int64_t *)scratch; int64_t c = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) iv[c++] = d[rows[i]];
            if (!c) { *out = 0.0; return 0; }
            qsort(iv, (size_t)c, 8, cmp_i64);
            int64_t mid = c / 2;
            *out = (c % 2 == 0) ? (double)(int64_t)((uint64_t)iv[mid - 1] + (uint64_t)iv[mid]) / 2.0
                                : (double)iv[mid];
            return 0; }
        case PANDRS_HIP_AGG_FIRST:           /* :605-614 */
            *out = (n && !is_null(m, rows[0])) ? (double)d[rows[0]] : 0.0; return 0;
        case PANDRS_HIP_AGG_LAST:            /* :615-624 */
            *out = (n && !is_null(m, rows[n - 1])) ? (double)d[rows[n - 1]] : 0.0; return 0;
        case PANDRS_HIP_AGG_NUNIQUE: {       /* legacy AggFunc::Nunique, src/dataframe/groupby.rs:514-519 (values as f64
                                              * there; distinct i64 stay distinct here — exact for |x| < 2^53) */
            int64_t *iv = (int64_t *)scratch; int64_t c = 0, u = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) iv[c++] = d[rows[i]];
            qsort(iv, (size_t)c, 8, cmp_i64);
            for (int64_t i = 0; i < c; i++) if (i == 0 || iv[i] != iv[i - 1]) u++;
            *out = (double)u; return 0; }    /* :467 — no values => 0.0 */
        }
    } else if (col->dtype == PANDRS_HIP_F64) {
        const double *d = (const double *)col->data; const uint8_t *m = col->null_mask;
        switch (op) {
        case PANDRS_HIP_AGG_SUM: {           /* :625-633 sequential, ascending rows */
            double s = 0.0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) s += d[rows[i]];
            *out = s; return 0; }
        case PANDRS_HIP_AGG_MEAN: {          /* :634-648 */
            double s = 0.0; int64_t c = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) { s += d[rows[i]]; c++; }
            *out = c > 0 ? s / (double)c : 0.0; return 0; }
        case PANDRS_HIP_AGG_MIN: {           /* :649-661 */
            double v = INFINITY;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) v = rust_min(v, d[rows[i]]);
            *out = v == INFINITY ? 0.0 : v; return 0; }
        case PANDRS_HIP_AGG_MAX: {           /* :662-674 */
            double v = -INFINITY;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) v = rust_max(v, d[rows[i]]);
            *out = v == -INFINITY ? 0.0 : v; return 0; }
        case PANDRS_HIP_AGG_STD: case PANDRS_HIP_AGG_VAR: {   /* :675-702 */
            int64_t c = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) scratch[c++] = d[rows[i]];
            double var = c ? variance_of(scratch, c) : 0.0;
            *out = op == PANDRS_HIP_AGG_STD ? sqrt(var) : var; return 0; }
        case PANDRS_HIP_AGG_MEDIAN: {        /* :703-722 */
            int64_t c = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) scratch[c++] = d[rows[i]];
            if (!c) { *out = 0.0; return 0; }
            qsort(scratch, (size_t)c, 8, cmp_f64_partial);
            int64_t mid = c / 2;
            *out = (c % 2 == 0) ? (scratch[mid - 1] + scratch[mid]) / 2.0 : scratch[mid];
            return 0; }
        case PANDRS_HIP_AGG_FIRST:           /* :723-732 */
            *out = (n && !is_null(m, rows[0])) ? d[rows[0]] : 0.0; return 0;
        case PANDRS_HIP_AGG_LAST:            /* :733-742 */
            *out = (n && !is_null(m, rows[n - 1])) ? d[rows[n - 1]] : 0.0; return 0;
        case PANDRS_HIP_AGG_NUNIQUE: {       /* src/dataframe/groupby.rs:514-519: sort_by(partial_cmp), Vec::dedup (==), len.
                                              * NaNs: the reference's sort leaves their place unspecified; here they are
                                              * set aside first and every NaN counts as its own value (NaN != NaN). */
            int64_t c = 0, nans = 0, u = 0;
            for (int64_t i = 0; i < n; i++) if (!is_null(m, rows[i])) {
                if (isnan(d[rows[i]])) nans++; else scratch[c++] = d[rows[i]];
            }
            qsort(scratch, (size_t)c, 8, cmp_f64_partial);
            for (int64_t i = 0; i < c; i++) if (i == 0 || scratch[i] != scratch[i - 1]) u++;
            *out = (double)(u + nans); return 0; }
        }
    }
    return PANDRS_HIP_ERR_OPERATION_FAILED;  /* :748 — String/Bool x numeric op */
}

void oracle_free(void *p) { free(p); }

/* ---------------------------------------------------------------- typed groupby
 * Outputs are malloc'd here; release with oracle_free.
 *   out_keys      [n_keys * n_groups] cells, key-major
 *   out_key_null  [n_keys * n_groups] bytes
 *   out_aggs      [n_aggs * n_groups] doubles, agg-major
 * Output order: ascending by key (nulls last). */
int oracle_groupby_agg(const ocol *keys, int n_keys, int64_t n_rows,
                       const ocol *vals, int n_vals, const oagg *aggs, int n_aggs,
                       int64_t *out_n_groups, uint64_t **out_keys, uint8_t **out_key_null,
                       double **out_aggs) {
    *out_n_groups = 0; *out_keys = NULL; *out_key_null = NULL; *out_aggs = NULL;
    for (int a = 0; a < n_aggs; a++)
        if (aggs[a].col < 0 || aggs[a].col >= n_vals) return PANDRS_HIP_ERR_INVALID_ARGUMENT;
    int64_t *rows = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_rows ? n_rows : 1));
    for (int64_t i = 0; i < n_rows; i++) rows[i] = i;
    sort_ctx sc = { keys, n_keys };
    qsort_r(rows, (size_t)n_rows, sizeof(int64_t), cmp_rows, &sc);
    int64_t g = 0;
    for (int64_t i = 0; i < n_rows; i++) if (i == 0 || !same_group(&sc, rows[i - 1], rows[i])) g++;
    size_t gg = (size_t)(g ? g : 1);
    uint64_t *ok = (uint64_t *)calloc(gg * (size_t)(n_keys ? n_keys : 1), 8);
    uint8_t *on = (uint8_t *)calloc(gg * (size_t)(n_keys ? n_keys : 1), 1);
    double *oa = (double *)calloc(gg * (size_t)(n_aggs ? n_aggs : 1), 8);
    double *scratch = (double *)malloc(8 * (size_t)(n_rows ? n_rows : 1));
    int rc = 0; int64_t gi = 0;
    for (int64_t i = 0; i < n_rows && !rc;) {
        int64_t j = i + 1;
        while (j < n_rows && same_group(&sc, rows[i], rows[j])) j++;
        for (int k = 0; k < n_keys; k++) {
            int nu = is_null(keys[k].null_mask, rows[i]);
            on[(size_t)k * (size_t)g + (size_t)gi] = (uint8_t)nu;
            ok[(size_t)k * (size_t)g + (size_t)gi] = nu ? 0 : key_cell(&keys[k], rows[i]);
        }
        for (int a = 0; a < n_aggs && !rc; a++)
            rc = fold_group(&vals[aggs[a].col], aggs[a].op, rows + i, j - i,
                            &oa[(size_t)a * (size_t)g + (size_t)gi], scratch);
        gi++; i = j;
    }
    free(scratch); free(rows);
    if (rc) { free(ok); free(on); free(oa); return rc; }
    *out_n_groups = g; *out_keys = ok; *out_key_null = on; *out_aggs = oa;
    return 0;
}

/* ---------------------------------------------------------------- faithful groupby
 * Shape of LazyFrame::execute's Aggregate arm (lazy.rs:186-404) / group_by + aggregate
 * (grouping.rs:60-104, aggregation.rs:792-808): per row build Vec<String>, hash with the
 * std HashMap hasher (SipHash-1-3), push the row index onto the group's Vec<usize>; then for
 * each group gather-and-fold every aggregate.  `pools[k]` supplies the strings for U32CODE key
 * columns (may be NULL: the decimal code is used).  Output order = table iteration order. */

#define ROTL(x, b) (uint64_t)(((x) << (b)) | ((x) >> (64 - (b))))
#define SIPROUND do { v0 += v1; v1 = ROTL(v1, 13); v1 ^= v0; v0 = ROTL(v0, 32); \
    v2 += v3; v3 = ROTL(v3, 16); v3 ^= v2; v0 += v3; v3 = ROTL(v3, 21); v3 ^= v0; \
    v2 += v1; v1 = ROTL(v1, 17); v1 ^= v2; v2 = ROTL(v2, 32); } while (0)

static uint64_t siphash13(const uint8_t *in, size_t len) {
    uint64_t k0 = 0x0706050403020100ull, k1 = 0x0f0e0d0c0b0a0908ull;
    uint64_t v0 = 0x736f6d6570736575ull ^ k0, v1 = 0x646f72616e646f6dull ^ k1;
    uint64_t v2 = 0x6c7967656e657261ull ^ k0, v3 = 0x7465646279746573ull ^ k1;
    const uint8_t *end = in + len - (len % 8);
    uint64_t b = ((uint64_t)len) << 56, m;
    for (; in != end; in += 8) { memcpy(&m, in, 8); v3 ^= m; SIPROUND; v0 ^= m; }
    switch (len & 7) {
    case 7: b |= ((uint64_t)in[6]) << 48; /* fallthrough */
    case 6: b |= ((uint64_t)in[5]) << 40; /* fallthrough */
    case 5: b |= ((uint64_t)in[4]) << 32; /* fallthrough */
    case 4: b |= ((uint64_t)in[3]) << 24; /* fallthrough */
    case 3: b |= ((uint64_t)in[2]) << 16; /* fallthrough */
    case 2: b |= ((uint64_t)in[1]) << 8;  /* fallthrough */
    case 1: b |= ((uint64_t)in[0]); break;
    case 0: break;
    }
    v3 ^= b; SIPROUND; v0 ^= b; v2 ^= 0xff; SIPROUND; SIPROUND; SIPROUND;
    return v0 ^ v1 ^ v2 ^ v3;
}

typedef struct {
    char *key; uint32_t key_len; uint64_t hash;
    int64_t *rows; int64_t n, cap;
    int64_t first_row;
} sgroup;

typedef struct {
    sgroup *groups; int64_t n_groups, cap_groups;
    int64_t *slots; int64_t n_slots; /* open addressing over group indices, -1 empty */
} stable;

static void stable_grow(stable *t) {
    int64_t ns = t->n_slots * 2;
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)ns);
    for (int64_t i = 0; i < ns; i++) s[i] = -1;
    for (int64_t g = 0; g < t->n_groups; g++) {
        int64_t p = (int64_t)(t->groups[g].hash & (uint64_t)(ns - 1));
        while (s[p] >= 0) p = (p + 1) & (ns - 1);
        s[p] = g;
    }
    free(t->slots); t->slots = s; t->n_slots = ns;
}

static int format_key_part(char *buf, size_t cap, const ocol *c, int64_t i,
                           const char *const *pool) {
    if (is_null(c->null_mask, i)) return snprintf(buf, cap, "NULL");       /* grouping.rs:74 */
    switch (c->dtype) {
    case PANDRS_HIP_I64: return snprintf(buf, cap, "%lld", (long long)((const int64_t *)c->data)[i]);
    case PANDRS_HIP_F64: {
        double v = ((const double *)c->data)[i];
        if (isnan(v)) return snprintf(buf, cap, "NaN");
        return snprintf(buf, cap, "%.17g", v);  /* injective like Rust's shortest repr */
    }
    case PANDRS_HIP_U32CODE: {
        uint32_t code = ((const uint32_t *)c->data)[i];
        if (pool) return snprintf(buf, cap, "%s", pool[code]);
        return snprintf(buf, cap, "%u", code);
    }
    case PANDRS_HIP_BOOLBITS:
        return snprintf(buf, cap, "%s",
                        ((((const uint8_t *)c->data)[i >> 3] >> (i & 7)) & 1) ? "true" : "false");
    }
    return 0;
}

/* Phase 1 of the reference's grouping over the rows [row_lo, row_hi): one String per key part, a
 * SipHash-1-3 of the Vec<String>, a HashMap probe and a Vec::push per row (lazy.rs:195-236,
 * grouping.rs:62-104).  Returns the largest group or -1 on error. */
static int64_t build_string_groups(const ocol *keys, int n_keys, const char *const *const *pools,
                                   int64_t row_lo, int64_t row_hi, stable *t) {
    t->n_groups = 0; t->cap_groups = 1024; t->n_slots = 2048;
    t->groups = (sgroup *)malloc(sizeof(sgroup) * (size_t)t->cap_groups);
    t->slots = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->n_slots);
    for (int64_t i = 0; i < t->n_slots; i++) t->slots[i] = -1;
    char buf[1024];
    int64_t max_group = 0;
    for (int64_t row = row_lo; row < row_hi; row++) {
        /* Vec<String>: each part followed by 0xFF as <str as Hash>::hash writes it; a length
         * prefix as <[T] as Hash> does (lazy.rs:195-236). */
        size_t len = 0;
        uint64_t nk = (uint64_t)n_keys; memcpy(buf, &nk, 8); len = 8;
        for (int k = 0; k < n_keys; k++) {
            char *part = (char *)malloc(64);               /* val.to_string(): one heap String per part */
            int w = format_key_part(part, 64, &keys[k], row, pools ? pools[k] : NULL);
            if (w >= 64) { free(part); part = (char *)malloc((size_t)w + 1);
                           format_key_part(part, (size_t)w + 1, &keys[k], row, pools ? pools[k] : NULL); }
            if (len + (size_t)w + 1 > sizeof(buf)) { free(part); return -1; }
            memcpy(buf + len, part, (size_t)w); len += (size_t)w; buf[len++] = (char)0xFF;
            free(part);
        }
        uint64_t h = siphash13((const uint8_t *)buf, len);
        int64_t p = (int64_t)(h & (uint64_t)(t->n_slots - 1)), g = -1;
        while (t->slots[p] >= 0) {
            sgroup *c = &t->groups[t->slots[p]];
            if (c->hash == h && c->key_len == len && memcmp(c->key, buf, len) == 0) { g = t->slots[p]; break; }
            p = (p + 1) & (t->n_slots - 1);
        }
        if (g < 0) {
            if (t->n_groups == t->cap_groups) {
                t->cap_groups *= 2;
                t->groups = (sgroup *)realloc(t->groups, sizeof(sgroup) * (size_t)t->cap_groups);
            }
            g = t->n_groups++;
            sgroup *c = &t->groups[g];
            c->key = (char *)malloc(len); memcpy(c->key, buf, len); c->key_len = (uint32_t)len;
            c->hash = h; c->rows = NULL; c->n = 0; c->cap = 0; c->first_row = row;
            t->slots[p] = g;
            if (t->n_groups * 2 > t->n_slots) stable_grow(t);
        }
        sgroup *c = &t->groups[g];
        if (c->n == c->cap) {                                 /* Vec::push growth */
            c->cap = c->cap ? c->cap * 2 : 4;
            c->rows = (int64_t *)realloc(c->rows, sizeof(int64_t) * (size_t)c->cap);
        }
        c->rows[c->n++] = row;
        if (c->n > max_group) max_group = c->n;
    }
    return max_group;
}

/* Phase 2: per group, the reference's gather + fold for every requested aggregate
 * (aggregation.rs:500-754); `parallel` folds the groups on all cores like par_aggregate (:22-182). */
static int fold_string_groups(const stable *t, const ocol *keys, int n_keys, const ocol *vals,
                              const oagg *aggs, int n_aggs, int64_t max_group, int n_threads,
                              int64_t *out_n_groups, uint64_t **out_keys, uint8_t **out_key_null, double **out_aggs) {
    int64_t g = t->n_groups; size_t gg = (size_t)(g ? g : 1);
    uint64_t *ok = (uint64_t *)calloc(gg * (size_t)(n_keys ? n_keys : 1), 8);
    uint8_t *on = (uint8_t *)calloc(gg * (size_t)(n_keys ? n_keys : 1), 1);
    double *oa = (double *)calloc(gg * (size_t)(n_aggs ? n_aggs : 1), 8);
    int rc = 0;
#pragma omp parallel num_threads(n_threads > 1 ? n_threads : 1)
    {
        double *scratch = (double *)malloc(8 * (size_t)(max_group ? max_group : 1));
#pragma omp for schedule(dynamic, 256)
        for (int64_t gi = 0; gi < g; gi++) {
            const sgroup *c = &t->groups[gi];
            for (int k = 0; k < n_keys; k++) {
                int nu = is_null(keys[k].null_mask, c->first_row);
                on[(size_t)k * (size_t)g + (size_t)gi] = (uint8_t)nu;
                ok[(size_t)k * (size_t)g + (size_t)gi] = nu ? 0 : key_cell(&keys[k], c->first_row);
            }
            for (int a = 0; a < n_aggs; a++) {
                int r = fold_group(&vals[aggs[a].col], aggs[a].op, c->rows, c->n,
                                   &oa[(size_t)a * (size_t)g + (size_t)gi], scratch);
                if (r) {
#pragma omp atomic write
                    rc = r;
                }
            }
        }
        free(scratch);
    }
    if (rc) { free(ok); free(on); free(oa); return rc; }
    *out_n_groups = g; *out_keys = ok; *out_key_null = on; *out_aggs = oa;
    return 0;
}

static void free_string_groups(stable *t) {
    for (int64_t gi = 0; gi < t->n_groups; gi++) { free(t->groups[gi].key); free(t->groups[gi].rows); }
    free(t->groups); free(t->slots);
}

int oracle_groupby_agg_ref(const ocol *keys, int n_keys, const char *const *const *pools,
                           int64_t n_rows, const ocol *vals, int n_vals,
                           const oagg *aggs, int n_aggs,
                           int64_t *out_n_groups, uint64_t **out_keys, uint8_t **out_key_null,
                           double **out_aggs) {
    *out_n_groups = 0; *out_keys = NULL; *out_key_null = NULL; *out_aggs = NULL;
    for (int a = 0; a < n_aggs; a++)
        if (aggs[a].col < 0 || aggs[a].col >= n_vals) return PANDRS_HIP_ERR_INVALID_ARGUMENT;
    stable t;
    int64_t max_group = build_string_groups(keys, n_keys, pools, 0, n_rows, &t);
    if (max_group < 0) { free_string_groups(&t); return PANDRS_HIP_ERR_INVALID_ARGUMENT; }
    int rc = fold_string_groups(&t, keys, n_keys, vals, aggs, n_aggs, max_group, 1, out_n_groups, out_keys, out_key_null, out_aggs);
    free_string_groups(&t);
    return rc;
}

/* The reference's PARALLEL shape: par_groupby's chunk-map on all cores, then its
 * serial merge of the chunk maps in chunk order (grouping.rs:203-280: the row lists stay ascending),
 * then par_aggregate's parallel per-group folds (aggregation.rs:22-182, without its row race).
 * Same results as oracle_groupby_agg_ref up to group order. */
int oracle_groupby_agg_ref_mt(const ocol *keys, int n_keys, const char *const *const *pools,
                              int64_t n_rows, const ocol *vals, int n_vals,
                              const oagg *aggs, int n_aggs, int n_threads,
                              int64_t *out_n_groups, uint64_t **out_keys, uint8_t **out_key_null,
                              double **out_aggs) {
    *out_n_groups = 0; *out_keys = NULL; *out_key_null = NULL; *out_aggs = NULL;
    for (int a = 0; a < n_aggs; a++)
        if (aggs[a].col < 0 || aggs[a].col >= n_vals) return PANDRS_HIP_ERR_INVALID_ARGUMENT;
    if (n_threads < 1) n_threads = 1;
    stable *parts = (stable *)calloc((size_t)n_threads, sizeof(stable));
    int failed = 0;
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (int t = 0; t < n_threads; t++) {
        const int64_t lo = n_rows * t / n_threads, hi = n_rows * (t + 1) / n_threads;
        if (build_string_groups(keys, n_keys, pools, lo, hi, &parts[t]) < 0) {
#pragma omp atomic write
            failed = 1;
        }
    }
    /* serial merge into the first chunk's map, chunk by chunk */
    stable *m = &parts[0];
    int64_t max_group = 0;
    for (int64_t gi = 0; gi < m->n_groups; gi++) if (m->groups[gi].n > max_group) max_group = m->groups[gi].n;
    for (int t = 1; t < n_threads && !failed; t++) {
        stable *s = &parts[t];
        for (int64_t gi = 0; gi < s->n_groups; gi++) {
            sgroup *src = &s->groups[gi];
            int64_t p = (int64_t)(src->hash & (uint64_t)(m->n_slots - 1)), g = -1;
            while (m->slots[p] >= 0) {
                sgroup *c = &m->groups[m->slots[p]];
                if (c->hash == src->hash && c->key_len == src->key_len && memcmp(c->key, src->key, src->key_len) == 0) { g = m->slots[p]; break; }
                p = (p + 1) & (m->n_slots - 1);
            }
            if (g < 0) {                                     /* the whole group moves over */
                if (m->n_groups == m->cap_groups) {
                    m->cap_groups *= 2;
                    m->groups = (sgroup *)realloc(m->groups, sizeof(sgroup) * (size_t)m->cap_groups);
                }
                g = m->n_groups++;
                m->groups[g] = *src;
                src->key = NULL; src->rows = NULL;
                m->slots[p] = g;
                if (m->n_groups * 2 > m->n_slots) stable_grow(m);
            } else {                                         /* extend(): the later chunk's rows come after */
                sgroup *c = &m->groups[g];
                if (c->n + src->n > c->cap) {
                    c->cap = (c->n + src->n) * 2;
                    c->rows = (int64_t *)realloc(c->rows, sizeof(int64_t) * (size_t)c->cap);
                }
                memcpy(c->rows + c->n, src->rows, sizeof(int64_t) * (size_t)src->n);
                c->n += src->n;
            }
            if (m->groups[g].n > max_group) max_group = m->groups[g].n;
        }
    }
    int rc = failed ? PANDRS_HIP_ERR_INVALID_ARGUMENT
                    : fold_string_groups(m, keys, n_keys, vals, aggs, n_aggs, max_group, n_threads, out_n_groups, out_keys, out_key_null, out_aggs);
    for (int t = 0; t < n_threads; t++) free_string_groups(&parts[t]);
    free(parts);
    return rc;
}

/* ---------------------------------------------------------------- join
 * join_impl up to join_indices (join.rs:106-224).  Typed: right rows sorted by (key,row), each
 * left row binary-searches its run, so matches come out in ascending right-row order exactly as
 * the reference's per-key Vec<usize> (join.rs:114, :156-158).  -1 = None. */
typedef struct { uint64_t key; int64_t row; } krow;
static int cmp_krow(const void *a, const void *b) {
    const krow *x = (const krow *)a, *y = (const krow *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row ? 1 : 0);
}

int oracle_join_indices(const ocol *lkey, int64_t n_left, const ocol *rkey, int64_t n_right,
                        int how, int64_t *out_n, int64_t **out_left, int64_t **out_right) {
    *out_n = 0; *out_left = NULL; *out_right = NULL;
    if (lkey->dtype != rkey->dtype) return PANDRS_HIP_ERR_TYPE_MISMATCH;     /* join.rs:98-104 */
    krow *r = (krow *)malloc(sizeof(krow) * (size_t)(n_right ? n_right : 1));
    int64_t nr = 0;
    for (int64_t i = 0; i < n_right; i++)
        if (!is_null(rkey->null_mask, i)) { r[nr].key = key_cell(rkey, i); r[nr].row = i; nr++; } /* :112 */
    qsort(r, (size_t)nr, sizeof(krow), cmp_krow);
    int64_t cap = n_left + n_right + 16, n = 0;
    int64_t *ol = (int64_t *)malloc(8 * (size_t)cap), *orr = (int64_t *)malloc(8 * (size_t)cap);
    uint8_t *matched = (uint8_t *)calloc((size_t)(n_right ? n_right : 1), 1);
#define PUSH(L, R) do { if (n == cap) { cap *= 2; ol = (int64_t *)realloc(ol, 8 * (size_t)cap); \
        orr = (int64_t *)realloc(orr, 8 * (size_t)cap); } ol[n] = (L); orr[n] = (R); n++; } while (0)
    for (int64_t i = 0; i < n_left; i++) {
        if (is_null(lkey->null_mask, i)) continue;            /* :152 — dropped even for left/outer */
        uint64_t k = key_cell(lkey, i);
        int64_t lo = 0, hi = nr;
        while (lo < hi) { int64_t mid = (lo + hi) / 2; if (r[mid].key < k) lo = mid + 1; else hi = mid; }
        if (lo < nr && r[lo].key == k) {
            for (int64_t j = lo; j < nr && r[j].key == k; j++) { PUSH(i, r[j].row); matched[r[j].row] = 1; }
        } else if (how == PANDRS_HIP_JOIN_LEFT || how == PANDRS_HIP_JOIN_OUTER) {
            PUSH(i, -1);                                       /* :159-162 */
        }
    }
    if (how == PANDRS_HIP_JOIN_RIGHT || how == PANDRS_HIP_JOIN_OUTER)      /* :211-224, nulls included */
        for (int64_t i = 0; i < n_right; i++) if (!matched[i]) PUSH(-1, i);
#undef PUSH
    free(r); free(matched);
    *out_n = n; *out_left = ol; *out_right = orr;
    return 0;
}

/* The same pairs in the reference's own SHAPE (join.rs:107-224): the right side goes into a
 * HashMap<String, Vec<usize>> — per non-null row one heap String (val.to_string()), one SipHash-1-3 of its bytes
 * (+ 0xFF, <str as Hash>), a probe and a Vec::push (:107-142) — then every non-null left row formats ITS key,
 * hashes it and looks it up (:146-208); unmatched right rows are appended by a scan over a matched-set (:211-224,
 * a HashSet<usize> there, a byte map here).  Serial, like the reference.  This is what bench.py times as the join's
 * cpu_baseline; tests/test_oracle_golden.py checks it pair for pair against oracle_join_indices. */
int oracle_join_indices_ref(const ocol *lkey, int64_t n_left, const ocol *rkey, int64_t n_right,
                            int how, int64_t *out_n, int64_t **out_left, int64_t **out_right) {
    *out_n = 0; *out_left = NULL; *out_right = NULL;
    if (lkey->dtype != rkey->dtype) return PANDRS_HIP_ERR_TYPE_MISMATCH;     /* join.rs:98-104 */
    stable t;
    t.n_groups = 0; t.cap_groups = 1024; t.n_slots = 2048;
    t.groups = (sgroup *)malloc(sizeof(sgroup) * (size_t)t.cap_groups);
    t.slots = (int64_t *)malloc(sizeof(int64_t) * (size_t)t.n_slots);
    for (int64_t i = 0; i < t.n_slots; i++) t.slots[i] = -1;
    char buf[96];
    for (int64_t row = 0; row < n_right; row++) {
        if (is_null(rkey->null_mask, row)) continue;                        /* :112 `if let Ok(Some(val))` */
        char *part = (char *)malloc(64);                                     /* val.to_string() */
        int w = format_key_part(part, 64, rkey, row, NULL);
        memcpy(buf, part, (size_t)w); buf[w] = (char)0xFF; free(part);
        const size_t len = (size_t)w + 1;
        uint64_t h = siphash13((const uint8_t *)buf, len);
        int64_t p = (int64_t)(h & (uint64_t)(t.n_slots - 1)), g = -1;
        while (t.slots[p] >= 0) {
            sgroup *c = &t.groups[t.slots[p]];
            if (c->hash == h && c->key_len == len && memcmp(c->key, buf, len) == 0) { g = t.slots[p]; break; }
            p = (p + 1) & (t.n_slots - 1);
        }
        if (g < 0) {
            if (t.n_groups == t.cap_groups) { t.cap_groups *= 2; t.groups = (sgroup *)realloc(t.groups, sizeof(sgroup) * (size_t)t.cap_groups); }
            g = t.n_groups++;
            sgroup *c = &t.groups[g];
            c->key = (char *)malloc(len); memcpy(c->key, buf, len); c->key_len = (uint32_t)len;
            c->hash = h; c->rows = NULL; c->n = 0; c->cap = 0; c->first_row = row;
            t.slots[p] = g;
            if (t.n_groups * 2 > t.n_slots) stable_grow(&t);
        }
        sgroup *c = &t.groups[g];
        if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 4; c->rows = (int64_t *)realloc(c->rows, sizeof(int64_t) * (size_t)c->cap); }
        c->rows[c->n++] = row;
    }
    int64_t cap = n_left + n_right + 16, n = 0;
    int64_t *ol = (int64_t *)malloc(8 * (size_t)cap), *orr = (int64_t *)malloc(8 * (size_t)cap);
    uint8_t *matched = (uint8_t *)calloc((size_t)(n_right ? n_right : 1), 1);
#define PUSH(L, R) do { if (n == cap) { cap *= 2; ol = (int64_t *)realloc(ol, 8 * (size_t)cap); \
        orr = (int64_t *)realloc(orr, 8 * (size_t)cap); } ol[n] = (L); orr[n] = (R); n++; } while (0)
    for (int64_t row = 0; row < n_left; row++) {
        if (is_null(lkey->null_mask, row)) continue;                        /* :152 */
        char *part = (char *)malloc(64);
        int w = format_key_part(part, 64, lkey, row, NULL);
        memcpy(buf, part, (size_t)w); buf[w] = (char)0xFF; free(part);
        const size_t len = (size_t)w + 1;
        uint64_t h = siphash13((const uint8_t *)buf, len);
        int64_t p = (int64_t)(h & (uint64_t)(t.n_slots - 1)), g = -1;
        while (t.slots[p] >= 0) {
            sgroup *c = &t.groups[t.slots[p]];
            if (c->hash == h && c->key_len == len && memcmp(c->key, buf, len) == 0) { g = t.slots[p]; break; }
            p = (p + 1) & (t.n_slots - 1);
        }
        if (g >= 0) {
            const sgroup *c = &t.groups[g];
            for (int64_t j = 0; j < c->n; j++) { PUSH(row, c->rows[j]); matched[c->rows[j]] = 1; }   /* :156-158 */
        } else if (how == PANDRS_HIP_JOIN_LEFT || how == PANDRS_HIP_JOIN_OUTER) {
            PUSH(row, -1);                                                   /* :159-162 */
        }
    }
    if (how == PANDRS_HIP_JOIN_RIGHT || how == PANDRS_HIP_JOIN_OUTER)
        for (int64_t i = 0; i < n_right; i++) if (!matched[i]) PUSH(-1, i);  /* :211-224 */
#undef PUSH
    free(matched); free_string_groups(&t);
    *out_n = n; *out_left = ol; *out_right = orr;
    return 0;
}

/* join.rs:296-357 gathers: None index or null source => fill */
void oracle_gather_i64(const int64_t *src, const uint8_t *mask, const int64_t *idx, int64_t n,
                       int64_t fill, int64_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (idx[i] >= 0 && !is_null(mask, idx[i])) ? src[idx[i]] : fill;
}
void oracle_gather_f64(const double *src, const uint8_t *mask, const int64_t *idx, int64_t n,
                       double fill, double *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (idx[i] >= 0 && !is_null(mask, idx[i])) ? src[idx[i]] : fill;
}
void oracle_gather_u32(const uint32_t *src, const uint8_t *mask, const int64_t *idx, int64_t n,
                       uint32_t fill, uint32_t *out) {
    for (int64_t i = 0; i < n; i++) out[i] = (idx[i] >= 0 && !is_null(mask, idx[i])) ? src[idx[i]] : fill;
}
void oracle_gather_bool(const uint8_t *bits, const uint8_t *mask, const int64_t *idx, int64_t n,
                        uint8_t fill, uint8_t *out) {
    for (int64_t i = 0; i < n; i++)
        out[i] = (idx[i] >= 0 && !is_null(mask, idx[i])) ? ((bits[idx[i] >> 3] >> (idx[i] & 7)) & 1) : fill;
}

/* ---------------------------------------------------------------- K1 whole-column reductions
 * out[0..3] = sum, mean, min, max.  Follows Int64Column::{sum,mean,min,max}
 * (src/column/int64_column.rs:129-241) / the null-skipping column folds; the lane-ordered SIMD
 * variants (simd.rs:116-199) differ only in f64 summation order, which tests allow 1e-9 for.
 * Empty (or all-null): sum 0, mean 0, min/max = +inf/-inf (f64) or i64::MAX/MIN (simd.rs:9-112). */
int oracle_reduce_column(const ocol *c, int64_t n, double out[4], int64_t *out_count) {
    int64_t cnt = 0;
    if (c->dtype == PANDRS_HIP_F64) {
        const double *d = (const double *)c->data; double s = 0, mn = INFINITY, mx = -INFINITY;
        for (int64_t i = 0; i < n; i++) if (!is_null(c->null_mask, i)) {
            s += d[i]; mn = rust_min(mn, d[i]); mx = rust_max(mx, d[i]); cnt++; }
        out[0] = s; out[1] = cnt ? s / (double)cnt : 0.0; out[2] = mn; out[3] = mx;
    } else if (c->dtype == PANDRS_HIP_I64) {
        const int64_t *d = (const int64_t *)c->data; uint64_t s = 0; int64_t mn = INT64_MAX, mx = INT64_MIN;
        for (int64_t i = 0; i < n; i++) if (!is_null(c->null_mask, i)) {
            s += (uint64_t)d[i]; if (d[i] < mn) mn = d[i]; if (d[i] > mx) mx = d[i]; cnt++; }
        out[0] = (double)(int64_t)s; out[1] = cnt ? (double)(int64_t)s / (double)cnt : 0.0;
        out[2] = (double)mn; out[3] = (double)mx;
    } else return PANDRS_HIP_ERR_OPERATION_FAILED;
    *out_count = cnt;
    return 0;
}

/* ---- K1, restated family by family ------------------------------------
 * (A) OptimizedDataFrame::{sum,mean,max,min}      src/optimized/split_dataframe/aggregate.rs:21-215
 *     non-null values collected as f64 (`v as f64` for Int64), empty => sum 0.0 / the others Err(Error::Empty);
 *     sum = parallel_sum_f64: Kahan per chunk + Kahan over the chunk sums (src/optimized/jit/parallel.rs:71-102;
 *     restated with one chunk: the chunking depends on the thread pool, the result only in the last bits);
 *     min / max = fold(+-inf, f64::min / f64::max): NaN operands dropped, infinities kept.
 * (B) Float64Column::{sum,mean,min,max}           src/column/float64_column.rs:100-199
 *     Int64Column::{sum,mean,min,max}             src/column/int64_column.rs:100-199
 *     empty data => None; mean None when no non-null value; f64 min / max skip NON-FINITE values (is_finite) and are
 *     None when none is left; Int64 sum is i64 (wrapping in release builds), mean = sum as f64 / count as f64.
 * (C) simd_{sum,mean,min,max}_{f64,i64}           src/optimized/jit/simd.rs:9-112   (no null masks: whole slices)
 *     simd_mean_i64 = simd_sum_i64 / len as i64 (INTEGER division, :77-82); empty => 0 / 0.0 and the fold identities. */
typedef struct oracle_k1 {
    int32_t a_empty; double a_sum, a_mean, a_min, a_max;
    int32_t b_data_empty, b_mean_none, b_minmax_none;
    double b_sum_f64, b_mean, b_min, b_max; int64_t b_sum_i64, b_min_i64, b_max_i64;
    double c_sum_f64, c_mean_f64, c_min_f64, c_max_f64; int64_t c_sum_i64, c_mean_i64, c_min_i64, c_max_i64;
} oracle_k1;

static double kahan_sum(const double *v, int64_t n) {            /* parallel.rs:76-85 */
    double sum = 0.0, c = 0.0;
    for (int64_t i = 0; i < n; i++) { double y = v[i] - c, t = sum + y; c = (t - sum) - y; sum = t; }
    return sum;
}

int oracle_k1_stats(const ocol *c, int64_t n, oracle_k1 *o) {
    if (c->dtype != PANDRS_HIP_F64 && c->dtype != PANDRS_HIP_I64) return PANDRS_HIP_ERR_TYPE_MISMATCH;   /* Error::Type */
    memset(o, 0, sizeof *o);
    const int f64 = c->dtype == PANDRS_HIP_F64;
    const double *d = (const double *)c->data; const int64_t *q = (const int64_t *)c->data;
    /* (A) */
    double *vals = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) if (!is_null(c->null_mask, i)) vals[m++] = f64 ? d[i] : (double)q[i];
    o->a_empty = m == 0;
    o->a_sum = m ? kahan_sum(vals, m) : 0.0;
    o->a_mean = m ? o->a_sum / (double)m : 0.0;          /* parallel_mean_f64_value: Kahan sum / count */
    o->a_min = INFINITY; o->a_max = -INFINITY;
    for (int64_t i = 0; i < m; i++) { o->a_min = rust_min(o->a_min, vals[i]); o->a_max = rust_max(o->a_max, vals[i]); }
    free(vals);
    /* (B) */
    o->b_data_empty = n == 0;
    int64_t cnt = 0, fin = 0; double bs = 0.0; uint64_t bi = 0;
    o->b_min = INFINITY; o->b_max = -INFINITY; o->b_min_i64 = INT64_MAX; o->b_max_i64 = INT64_MIN;
    for (int64_t i = 0; i < n; i++) if (!is_null(c->null_mask, i)) {
        cnt++;
        if (f64) {
            bs += d[i];
            if (isfinite(d[i])) { fin++; o->b_min = rust_min(o->b_min, d[i]); o->b_max = rust_max(o->b_max, d[i]); }
        } else {
            bi += (uint64_t)q[i]; fin++;
            if (q[i] < o->b_min_i64) o->b_min_i64 = q[i];
            if (q[i] > o->b_max_i64) o->b_max_i64 = q[i];
        }
    }
    o->b_sum_f64 = bs; o->b_sum_i64 = (int64_t)bi;
    o->b_mean_none = n == 0 || cnt == 0;
    o->b_mean = cnt ? (f64 ? bs : (double)(int64_t)bi) / (double)cnt : 0.0;
    o->b_minmax_none = n == 0 || fin == 0;
    if (!f64) { o->b_min = (double)o->b_min_i64; o->b_max = (double)o->b_max_i64; }
    /* (C): the whole slice, masks do not exist at this level */
    if (f64) {
        double s = 0.0; o->c_min_f64 = INFINITY; o->c_max_f64 = -INFINITY;
        for (int64_t i = 0; i < n; i++) { s += d[i]; o->c_min_f64 = rust_min(o->c_min_f64, d[i]); o->c_max_f64 = rust_max(o->c_max_f64, d[i]); }
        o->c_sum_f64 = s; o->c_mean_f64 = n ? s / (double)n : 0.0;
    } else {
        uint64_t s = 0; o->c_min_i64 = INT64_MAX; o->c_max_i64 = INT64_MIN;
        for (int64_t i = 0; i < n; i++) { s += (uint64_t)q[i]; if (q[i] < o->c_min_i64) o->c_min_i64 = q[i]; if (q[i] > o->c_max_i64) o->c_max_i64 = q[i]; }
        o->c_sum_i64 = (int64_t)s;
        /* Rust's i64 `/` truncates toward zero, like C's; i64::MIN / -1 cannot occur (len > 0) */
        o->c_mean_i64 = n ? (int64_t)s / n : 0;
    }
    return 0;
}

/* Fused C5 shape: inner_join (join.rs:32) then group_by(g).aggregate([(v, Sum)]). */
int oracle_join_groupby_sum(const ocol *lkey, const ocol *lval, int64_t n_left,
                            const ocol *rkey, const ocol *rgroup, int64_t n_right,
                            int64_t *out_n_groups, uint64_t **out_keys, uint8_t **out_key_null,
                            double **out_aggs) {
    int64_t n, *li, *ri;
    int rc = oracle_join_indices(lkey, n_left, rkey, n_right, PANDRS_HIP_JOIN_INNER, &n, &li, &ri);
    if (rc) return rc;
    size_t nn = (size_t)(n ? n : 1);
    /* gathered columns: misses/nulls filled with 0 / 0.0 and NOT null (join.rs:304-307) */
    double *v = (double *)malloc(8 * nn); int64_t *vi = (int64_t *)malloc(8 * nn);
    uint64_t *g = (uint64_t *)malloc(8 * nn);
    ocol gv = { NULL, NULL, lval->dtype, 0 }, gg = { g, NULL, rgroup->dtype, 0 };
    if (lval->dtype == PANDRS_HIP_F64) { oracle_gather_f64((const double *)lval->data, lval->null_mask, li, n, 0.0, v); gv.data = v; }
    else { oracle_gather_i64((const int64_t *)lval->data, lval->null_mask, li, n, 0, vi); gv.data = vi; }
    if (rgroup->dtype == PANDRS_HIP_U32CODE) {
        oracle_gather_u32((const uint32_t *)rgroup->data, rgroup->null_mask, ri, n, 0, (uint32_t *)g);
    } else {
        oracle_gather_i64((const int64_t *)rgroup->data, rgroup->null_mask, ri, n, 0, (int64_t *)g);
    }
    oagg a = { 0, PANDRS_HIP_AGG_SUM };
    rc = oracle_groupby_agg(&gg, 1, n, &gv, 1, &a, 1, out_n_groups, out_keys, out_key_null, out_aggs);
    free(v); free(vi); free(g); free(li); free(ri);
    return rc;
}

/* ---------------------------------------------------------------- fair typed CPU baseline
 * "a fair typed CPU baseline (i64 open-addressing hash, all cores) so the
 * speed-up is not just 'removed the strings'".  NOT the reference's algorithm (that is
 * oracle_groupby_agg_ref above) — it is what a good CPU implementation of the same operator does:
 * one i64 key, f64 value columns, {count, sum, min, max} per column kept per group (mean = sum /
 * count).  Every thread owns the keys whose hash falls in its slice, scans the key column, and
 * aggregates its rows in a private open-addressing table: no locks, no merge.  Used only by
 * bench.py's cpu_baseline leg and checked against oracle_groupby_agg in tests/. */
#include <omp.h>
typedef struct { uint64_t key; int64_t n; int used; } tslot;
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33; return x;
}
int oracle_groupby_typed_mt(const int64_t *keys, int64_t n_rows, const double *const *vals, int n_vals,
                            int n_threads, int64_t *out_n_groups, uint64_t **out_keys,
                            double **out_stats /* [n_groups][1 + 3 * n_vals]: count, then sum,min,max per column */) {
    if (n_threads < 1) n_threads = 1;
    const int W = 1 + 3 * n_vals;
    uint64_t **tk = (uint64_t **)calloc((size_t)n_threads, sizeof(*tk));
    double **ts = (double **)calloc((size_t)n_threads, sizeof(*ts));
    int64_t *tn = (int64_t *)calloc((size_t)n_threads, sizeof(*tn));
    int failed = 0;
    /* phase 1: the owner thread of every row (hash once per row, all cores) */
    uint8_t *owner = (uint8_t *)malloc((size_t)(n_rows ? n_rows : 1));
    if (n_threads > 255) n_threads = 255;
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int64_t i = 0; i < n_rows; i++) owner[i] = (uint8_t)((mix64((uint64_t)keys[i]) >> 40) % (uint64_t)n_threads);
    /* phase 2: every thread aggregates the rows it owns in a private table */
#pragma omp parallel num_threads(n_threads)
    {
        const int t = omp_get_thread_num();
        int64_t cap = 1 << 12, ng = 0;
        tslot *slots = (tslot *)calloc((size_t)cap, sizeof(tslot));
        int64_t *slot_group = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
        int64_t gcap = 1 << 11;
        uint64_t *gk = (uint64_t *)malloc(8 * (size_t)gcap);
        double *gs = (double *)malloc(8 * (size_t)gcap * (size_t)W);
        for (int64_t i = 0; i < n_rows && slots && gk && gs; i++) {
            if (owner[i] != (uint8_t)t) continue;
            const uint64_t k = (uint64_t)keys[i], h = mix64(k);
            int64_t p = (int64_t)(h & (uint64_t)(cap - 1));
            while (slots[p].used && slots[p].key != k) p = (p + 1) & (cap - 1);
            int64_t g;
            if (!slots[p].used) {
                if (ng == gcap) {
                    gcap *= 2;
                    gk = (uint64_t *)realloc(gk, 8 * (size_t)gcap);
                    gs = (double *)realloc(gs, 8 * (size_t)gcap * (size_t)W);
                }
                g = ng++;
                slots[p].used = 1; slots[p].key = k; slot_group[p] = g;
                gk[g] = k;
                double *st = gs + (size_t)g * (size_t)W;
                st[0] = 0.0;
                for (int c = 0; c < n_vals; c++) { st[1 + 3 * c] = 0.0; st[2 + 3 * c] = INFINITY; st[3 + 3 * c] = -INFINITY; }
                if (ng * 2 > cap) {                          /* grow + rehash */
                    int64_t ncap = cap * 2;
                    tslot *ns = (tslot *)calloc((size_t)ncap, sizeof(tslot));
                    int64_t *nsg = (int64_t *)malloc(sizeof(int64_t) * (size_t)ncap);
                    for (int64_t q = 0; q < cap; q++) if (slots[q].used) {
                        int64_t r = (int64_t)(mix64(slots[q].key) & (uint64_t)(ncap - 1));
                        while (ns[r].used) r = (r + 1) & (ncap - 1);
                        ns[r] = slots[q]; nsg[r] = slot_group[q];
                    }
                    free(slots); free(slot_group); slots = ns; slot_group = nsg; cap = ncap;
                }
            } else {
                g = slot_group[p];
            }
            double *st = gs + (size_t)g * (size_t)W;
            st[0] += 1.0;
            for (int c = 0; c < n_vals; c++) {
                const double v = vals[c][i];
                st[1 + 3 * c] += v;
                if (v < st[2 + 3 * c]) st[2 + 3 * c] = v;
                if (v > st[3 + 3 * c]) st[3 + 3 * c] = v;
            }
        }
        if (!slots || !gk || !gs) {
#pragma omp atomic write
            failed = 1;
        }
        free(slots); free(slot_group);
        tk[t] = gk; ts[t] = gs; tn[t] = ng;
    }
    int64_t G = 0;
    for (int t = 0; t < n_threads; t++) G += tn[t];
    uint64_t *ok = (uint64_t *)malloc(8 * (size_t)(G ? G : 1));
    double *os = (double *)malloc(8 * (size_t)(G ? G : 1) * (size_t)W);
    int64_t o = 0;
    for (int t = 0; t < n_threads; t++) {
        if (tn[t] && tk[t] && ts[t]) {
            memcpy(ok + o, tk[t], 8 * (size_t)tn[t]);
            memcpy(os + (size_t)o * (size_t)W, ts[t], 8 * (size_t)tn[t] * (size_t)W);
            o += tn[t];
        }
        free(tk[t]); free(ts[t]);
    }
    free(tk); free(ts); free(tn); free(owner);
    if (failed) { free(ok); free(os); return PANDRS_HIP_ERR_OUT_OF_MEMORY; }
    *out_n_groups = G; *out_keys = ok; *out_stats = os;
    return 0;
}
