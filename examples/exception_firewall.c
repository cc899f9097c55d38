/* The C ABI never lets a C++ exception out (include/pandrs_hip.h, "Conventions"; the reference's errors are
 * Result<T, pandrs::Error>, src/core/error.rs:6): this program makes the host code of an entry point throw —
 * std::bad_alloc, a std::exception, a foreign exception, and a real oversized std::vector::resize — and must see
 * status codes, not a crash.  Plain C, links libpandrs_hip.so only, needs no GPU. */
#include <stdio.h>
#include <string.h>

#include "pandrs_hip.h"

int main(void) {
    const struct { int64_t what; int32_t want; const char *needle; } cases[] = {
        {1, PANDRS_HIP_ERR_OUT_OF_MEMORY, "bad_alloc"},
        {2, PANDRS_HIP_ERR_COMPUTATION, "C++ exception"},
        {3, PANDRS_HIP_ERR_COMPUTATION, "unknown C++ exception"},
        {4, -1, ""},      /* length_error (-> COMPUTATION) or bad_alloc (-> OUT_OF_MEMORY), by the C++ library's choice */
    };
    int failed = 0;
    for (unsigned i = 0; i < sizeof cases / sizeof cases[0]; i++) {
        const int32_t st = pandrs_hip_ctx_set_option(NULL, "test_throw", cases[i].what);
        const char *msg = pandrs_hip_last_error();
        const int ok = cases[i].want >= 0 ? (st == cases[i].want && strstr(msg, cases[i].needle) != NULL)
                                          : (st == PANDRS_HIP_ERR_OUT_OF_MEMORY || st == PANDRS_HIP_ERR_COMPUTATION);
        printf("test_throw=%lld -> status %d (%s)%s\n", (long long)cases[i].what, st, msg, ok ? "" : "   UNEXPECTED");
        failed += !ok;
    }
    if (pandrs_hip_ctx_set_option(NULL, "test_throw", 0) != PANDRS_HIP_OK) failed++;
    if (pandrs_hip_abi_version() != PANDRS_HIP_ABI_VERSION) failed++;       /* the library is still usable */
    printf(failed ? "exception firewall: FAILED\n" : "exception firewall: OK\n");
    return failed ? 2 : 0;
}
