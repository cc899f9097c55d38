/*
 * groupby_c_abi.c — the drop-in boundary exercised from plain C: only include/pandrs_hip.h and
 * libpandrs_hip.so, no Python, no torch.  Plays the role of the reference's
 * examples/optimized_groupby_example.rs (same data: values [10,20,15,30,25,15] by category
 * [A,B,A,C,B,A]) and of a 4-row join (tests/optimized_join_test.rs:6-41).
 *
 *   gcc -O2 -Iinclude examples/groupby_c_abi.c -Lpandrs_amd -lpandrs_hip -Wl,-rpath,$PWD/pandrs_amd -o /tmp/groupby_c_abi
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pandrs_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int32_t st__ = (call);                                                   \
        if (st__ != PANDRS_HIP_OK) {                                             \
            fprintf(stderr, "%s -> status %d: %s\n", #call, st__, pandrs_hip_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void) {
    pandrs_hip_ctx *ctx = NULL;
    CHECK(pandrs_hip_init(NULL));
    CHECK(pandrs_hip_ctx_create(0, &ctx));

    /* string-pool codes of the category column: A=0, B=1, C=2 (equal string <=> equal code) */
    const char *pool[] = {"A", "B", "C"};
    uint32_t category[6] = {0, 1, 0, 2, 1, 0};
    int64_t values[6] = {10, 20, 15, 30, 25, 15};
    pandrs_hip_column key = {category, NULL, PANDRS_HIP_U32CODE, 0};
    pandrs_hip_column val = {values, NULL, PANDRS_HIP_I64, 0};
    pandrs_hip_agg_spec aggs[5] = {{0, PANDRS_HIP_AGG_COUNT}, {0, PANDRS_HIP_AGG_SUM}, {0, PANDRS_HIP_AGG_MEAN},
                                   {0, PANDRS_HIP_AGG_MIN}, {0, PANDRS_HIP_AGG_MAX}};
    int64_t n_groups = 0;
    CHECK(pandrs_hip_groupby_agg(ctx, PANDRS_HIP_MEM_HOST, &key, 1, 6, &val, 1, aggs, 5, &n_groups));
    uint64_t *gk = malloc(sizeof(uint64_t) * (size_t)n_groups);
    uint8_t *gn = malloc((size_t)n_groups);
    double *ga[5];
    for (int a = 0; a < 5; a++) ga[a] = malloc(sizeof(double) * (size_t)n_groups);
    uint64_t *keys_out[1] = {gk};
    uint8_t *null_out[1] = {gn};
    CHECK(pandrs_hip_groupby_fetch(ctx, PANDRS_HIP_MEM_HOST, keys_out, null_out, ga));
    printf("groups: %lld\n", (long long)n_groups);
    int ok = n_groups == 3;
    for (int64_t g = 0; g < n_groups; g++) {
        const char *name = gn[g] ? "NULL" : pool[gk[g]];
        printf("  %s: count %.0f sum %.0f mean %.6f min %.0f max %.0f\n", name, ga[0][g], ga[1][g], ga[2][g], ga[3][g], ga[4][g]);
        if (!strcmp(name, "A")) ok &= ga[0][g] == 3 && ga[1][g] == 40 && ga[3][g] == 10 && ga[4][g] == 15;
        if (!strcmp(name, "B")) ok &= ga[0][g] == 2 && ga[1][g] == 45 && ga[2][g] == 22.5;
        if (!strcmp(name, "C")) ok &= ga[0][g] == 1 && ga[1][g] == 30;
    }

    /* inner / outer join of ids [1,2,3,4] with [1,2,5,6] */
    int64_t lid[4] = {1, 2, 3, 4}, rid[4] = {1, 2, 5, 6};
    pandrs_hip_column lk = {lid, NULL, PANDRS_HIP_I64, 0}, rk = {rid, NULL, PANDRS_HIP_I64, 0};
    int64_t n_rows = 0, li[8], ri[8];
    CHECK(pandrs_hip_join_indices(ctx, PANDRS_HIP_MEM_HOST, &lk, 4, &rk, 4, PANDRS_HIP_JOIN_OUTER, &n_rows));
    CHECK(pandrs_hip_join_fetch(ctx, PANDRS_HIP_MEM_HOST, li, ri));
    printf("outer join rows: %lld\n", (long long)n_rows);
    for (int64_t i = 0; i < n_rows; i++) printf("  (%lld, %lld)\n", (long long)li[i], (long long)ri[i]);
    const int64_t want_l[6] = {0, 1, 2, 3, -1, -1}, want_r[6] = {0, 1, -1, -1, 2, 3};
    ok &= n_rows == 6 && !memcmp(li, want_l, sizeof want_l) && !memcmp(ri, want_r, sizeof want_r);

    /* error path: dtype mismatch maps to ColumnTypeMismatch (join.rs:98-104) */
    double rf[1] = {1.0};
    pandrs_hip_column rkf = {rf, NULL, PANDRS_HIP_F64, 0};
    int32_t st = pandrs_hip_join_indices(ctx, PANDRS_HIP_MEM_HOST, &lk, 4, &rkf, 1, PANDRS_HIP_JOIN_INNER, &n_rows);
    printf("type mismatch -> status %d (%s)\n", st, pandrs_hip_last_error());
    ok &= st == PANDRS_HIP_ERR_TYPE_MISMATCH;

    pandrs_hip_ctx_destroy(ctx);
    puts(ok ? "C ABI example: OK" : "C ABI example: MISMATCH");
    return ok ? 0 : 2;
}
