// l2_probe.hip — random 16-byte table lookups: how fast when the table region a workgroup probes fits its XCD's L2?
// P table regions of S bytes; the workgroups of group g = blockIdx % 8 (one XCD under round-robin dispatch) walk the
// regions p = g, g + 8, ... together, every thread doing LOOKUPS random 16-B loads (U in flight) per region.
// Compared with: the same loads spread over the whole table (no locality).   GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int U, bool LOCAL>
__global__ __launch_bounds__(1024) void probe_kernel(const uint4 *table, uint32_t P, uint32_t entries_per_region, uint32_t lookups, u64 *out) {
    const uint32_t g = blockIdx.x & 7, rank = blockIdx.x >> 3, per_group = gridDim.x >> 3;
    u64 acc = 0;
    const uint32_t total_entries = P * entries_per_region;
    for (uint32_t p = g; p < P; p += 8) {
        const uint4 *reg = table + (size_t)p * entries_per_region;
        uint32_t s = mix(p * 7919u + rank * 1024u + threadIdx.x + 1u);
        for (uint32_t i = 0; i < lookups; i += U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                s = s * 1664525u + 1013904223u;
                const uint32_t r = mix(s);
                v[u] = LOCAL ? reg[r % entries_per_region] : table[r % total_entries];
            }
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u].x + v[u].z;
        }
    }
    (void)per_group;
    if (acc == 0x123456789ull) out[0] = acc;
}

// second experiment (argv: "visits"): short visits — every region gets only `lookups` x 1024 x (workgroups per group) probes
// before the group moves on, as a probe over many small partitions does; with a 16 B/lookup input stream beside it or not
template <bool STREAM>
__global__ __launch_bounds__(1024) void visit_kernel(const uint4 *table, uint32_t P, uint32_t entries_per_region, uint32_t lookups,
                                                     const uint4 *stream, u64 *out) {
    const uint32_t g = blockIdx.x & 7, rank = blockIdx.x >> 3, per_group = gridDim.x >> 3;
    u64 acc = 0;
    for (uint32_t p = g; p < P; p += 8) {
        const uint4 *reg = table + (size_t)p * entries_per_region;
        const uint4 *src = stream + ((size_t)(p >> 3) * per_group + rank) * 1024 * lookups;
        for (uint32_t i = 0; i < lookups; i += 8) {
            uint32_t r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (STREAM) { const u64 *q = reinterpret_cast<const u64 *>(&src[(size_t)(i + u) * 1024 + threadIdx.x]); const u64 x0 = __builtin_nontemporal_load(q), x1 = __builtin_nontemporal_load(q + 1); r[u] = mix((uint32_t)(x0 ^ x1) ^ (p * 7919u) ^ (threadIdx.x * 2654435761u + i + u)); }
                else r[u] = mix((p * 7919u + rank * 1024u + threadIdx.x) * 2654435761u + i + u);
            }
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = reg[r[u] % entries_per_region];
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u].x + v[u].z;
        }
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

int visits_main() {
    u64 *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t P = 2048, kb = 1024, epr = kb * 1024 / 16;
    uint4 *table; CK(hipMalloc(&table, (size_t)P * epr * 16));
    CK(hipMemset(table, 1, (size_t)P * epr * 16));
    for (uint32_t lookups : {8u, 16u, 64u, 256u}) {
        const uint32_t grid = 256;
        const size_t n_stream = (size_t)(P / 8) * grid * 1024 * lookups;          // uint4 per lookup
        uint4 *stream; CK(hipMalloc(&stream, n_stream * 16));
        CK(hipMemset(stream, 3, n_stream * 16));
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9;
            for (int it = 0; it < 3; it++) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL((visit_kernel<false>), dim3(grid), dim3(1024), 0, 0, table, P, epr, lookups, stream, out);
                else hipLaunchKernelGGL((visit_kernel<true>), dim3(grid), dim3(1024), 0, 0, table, P, epr, lookups, stream, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("1 MB regions, %4u lookups per thread and visit (%6.1f per 128-B line), %-9s  %.3f ms  %.1f G lookups/s\n", lookups,
                   (double)lookups * 1024 * (grid / 8) / (kb * 8.0), mode == 0 ? "no stream" : "stream", best, (double)n_stream / best / 1e6);
        }
        CK(hipFree(stream));
    }
    return 0;
}


// third experiment (argv: "bucket"): a lookup that needs its whole 64-byte bucket (4 entries of 16 B) out of a 128 MB table:
// (a) every lane loads its own bucket with four 16-byte loads (four L2 requests per lookup);
// (b) a QUAD of lanes serves its four lookups one after the other, lane j loading entry j (one coalesced 64-byte request per lookup).
template <bool QUAD>
__global__ __launch_bounds__(256) void bucket_kernel(const uint4 *table, uint32_t n_buckets, uint32_t lookups, u64 *out) {
    u64 acc = 0;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t s = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    for (uint32_t i = 0; i < lookups; i += 4) {
        uint32_t b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { s = s * 1664525u + 1013904223u; b[u] = mix(s) % n_buckets; }
        uint4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (QUAD) {
                    const uint32_t bq = __shfl(b[u], (lane & ~3u) + q, 64);      // the bucket of lane q of this quad
                    v[u][q] = table[(size_t)bq * 4 + (lane & 3u)];
                } else v[u][q] = table[(size_t)b[u] * 4 + q];
            }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc += v[u][q].x + v[u][q].z;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
int bucket_main() {
    u64 *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t n_buckets = 2u << 20;                      // 128 MB
    uint4 *table; CK(hipMalloc(&table, (size_t)n_buckets * 64));
    CK(hipMemset(table, 1, (size_t)n_buckets * 64));
    const uint32_t grid = 256 * 8, lookups = 96;
    for (int mode = 0; mode < 2; mode++) {
        float best = 1e9;
        for (int it = 0; it < 3; it++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL((bucket_kernel<false>), dim3(grid), dim3(256), 0, 0, table, n_buckets, lookups, out);
            else hipLaunchKernelGGL((bucket_kernel<true>), dim3(grid), dim3(256), 0, 0, table, n_buckets, lookups, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("64-byte buckets out of 128 MB, %s: %.3f ms  %.1f G lookups/s\n", mode == 0 ? "4 loads per lane      " : "quad of lanes per lookup",
               best, (double)grid * 256 * lookups / best / 1e6);
    }
    return 0;
}

// fourth experiment (argv: "lbucket"): the same two bucket forms against L2-sized regions (the XCD group walks the regions together)
template <bool QUAD>
__global__ __launch_bounds__(256) void local_bucket_kernel(const uint4 *table, uint32_t P, uint32_t buckets_per_region, uint32_t lookups, u64 *out) {
    const uint32_t g = blockIdx.x & 7, lane = threadIdx.x & 63;
    u64 acc = 0;
    for (uint32_t p = g; p < P; p += 8) {
        const uint4 *reg = table + (size_t)p * buckets_per_region * 4;
        uint32_t s = mix(p * 7919u + (blockIdx.x >> 3) * 256u + threadIdx.x + 1u);
        for (uint32_t i = 0; i < lookups; i += 4) {
            uint32_t b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { s = s * 1664525u + 1013904223u; b[u] = mix(s) % buckets_per_region; }
            uint4 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (QUAD) {
                        const uint32_t bq = __shfl(b[u], (lane & ~3u) + q, 64);
                        v[u][q] = reg[(size_t)bq * 4 + (lane & 3u)];
                    } else v[u][q] = reg[(size_t)b[u] * 4 + q];
                }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc += v[u][q].x + v[u][q].z;
        }
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
int local_bucket_main() {
    u64 *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t P = 1024, bpr = 1024 * 1024 / 64;          // 1 MB regions
    uint4 *table; CK(hipMalloc(&table, (size_t)P * bpr * 64));
    CK(hipMemset(table, 1, (size_t)P * bpr * 64));
    const uint32_t grid = 256 * 4, lookups = 16;
    for (int mode = 0; mode < 2; mode++) {
        float best = 1e9;
        for (int it = 0; it < 3; it++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL((local_bucket_kernel<false>), dim3(grid), dim3(256), 0, 0, table, P, bpr, lookups, out);
            else hipLaunchKernelGGL((local_bucket_kernel<true>), dim3(grid), dim3(256), 0, 0, table, P, bpr, lookups, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("64-byte buckets out of 1 MB regions (L2), %s: %.3f ms  %.1f G lookups/s\n", mode == 0 ? "4 loads per lane      " : "quad of lanes per lookup",
               best, (double)(P / 8) * grid * 256 * lookups / best / 1e6);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'l') return local_bucket_main();
    if (argc > 1 && argv[1][0] == 'b') return bucket_main();
    if (argc > 1) return visits_main();
    const uint32_t P = 1024;
    u64 *out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t kb : {256u, 1024u, 1536u, 2048u, 3072u, 6144u, 16384u}) {
        const uint32_t epr = kb * 1024 / 16;
        uint4 *table; CK(hipMalloc(&table, (size_t)P * epr * 16));
        CK(hipMemset(table, 1, (size_t)P * epr * 16));
        for (int wg_per_cu = 1; wg_per_cu <= 2; wg_per_cu++) {
            const uint32_t grid = 256 * wg_per_cu, lookups = 64 / wg_per_cu;   // per thread per region
            for (int mode = 0; mode < 2; mode++) {
                float best = 1e9;
                for (int it = 0; it < 3; it++) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL((probe_kernel<8, true>), dim3(grid), dim3(1024), 0, 0, table, P, epr, lookups, out);
                    else hipLaunchKernelGGL((probe_kernel<8, false>), dim3(grid), dim3(1024), 0, 0, table, P, epr, lookups, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                const double n = (double)(P / 8) * grid * 1024.0 * lookups;
                printf("region %5u KB  %d WG/CU  %-6s  %.3f ms  %.1f G lookups/s\n", kb, wg_per_cu, mode == 0 ? "local" : "global", best, n / best / 1e6);
            }
        }
        CK(hipFree(table));
    }
    return 0;
}
