// Do narrower-than-agent scope atomics on a small table execute in the XCD's L2?  (C5's post-join sum has 1.6 MB of state.)
// N random adds into a table of G 8-byte cells, one table per XCC id (HW_REG_XCC_ID), by scope and type.
// build: hipcc --offload-arch=gfx950 -O3 -o experiments/ubench/l2_atomic experiments/ubench/l2_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int SCOPE, bool F64, bool PER_XCC>
__global__ __launch_bounds__(256) void adds(unsigned long long *tab, uint32_t G, uint32_t per_thread, unsigned long long *xcc_seen) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID[3:0]
    unsigned long long *t = tab + (PER_XCC ? (size_t)xcc * G : 0);
    uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) atomicAdd(&xcc_seen[xcc], 1ull);
    for (uint32_t i = 0; i < per_thread; i++) {
        x = mix(x + 0x9e3779b9u * (i + 1));
        const uint32_t g = (uint32_t)(((uint64_t)x * G) >> 32);
        if (F64) __hip_atomic_fetch_add(reinterpret_cast<double *>(t + g), 1.0, __ATOMIC_RELAXED, SCOPE);
        else __hip_atomic_fetch_add(t + g, 1ull, __ATOMIC_RELAXED, SCOPE);
    }
}
template <int SCOPE, bool F64, bool PER_XCC>
int run(const char *name, unsigned long long *tab, uint32_t G, unsigned long long *seen) {
    const uint32_t blocks = 256 * 8, per_thread = 256;
    const double n = (double)blocks * 256 * per_thread;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(tab, 0, (size_t)8 * G * 8)); CK(hipMemset(seen, 0, 64));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((adds<SCOPE, F64, PER_XCC>), dim3(blocks), dim3(256), 0, 0, tab, G, per_thread, seen);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h((size_t)8 * G); CK(hipMemcpy(h.data(), tab, h.size() * 8, hipMemcpyDeviceToHost));
    double total = 0;
    for (size_t i = 0; i < h.size(); i++) { if (F64) { double d; memcpy(&d, &h[i], 8); total += d; } else total += (double)h[i]; }
    unsigned long long hs[8]; CK(hipMemcpy(hs, seen, 64, hipMemcpyDeviceToHost));
    printf("%-44s G %7u  %8.3f ms  %7.1f G adds/s  sum %s (%.0f of %.0f)  blocks per xcc %llu %llu %llu %llu %llu %llu %llu %llu\n", name, G, best, n / best / 1e6,
           total == n ? "ok" : "LOST", total, n, hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]);
    return 0;
}
int main() {
    unsigned long long *tab, *seen;
    const uint32_t GMAX = 1u << 22;
    CK(hipMalloc(&tab, (size_t)8 * GMAX * 8)); CK(hipMalloc(&seen, 64));
    for (uint32_t G : {1000u, 100000u, 1000000u}) {
        run<__HIP_MEMORY_SCOPE_AGENT, false, false>("u64 agent, one table", tab, G, seen);
        run<__HIP_MEMORY_SCOPE_AGENT, false, true>("u64 agent, table per XCC", tab, G, seen);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, false, true>("u64 workgroup scope, table per XCC", tab, G, seen);
        run<__HIP_MEMORY_SCOPE_WAVEFRONT, false, true>("u64 wavefront scope, table per XCC", tab, G, seen);
        run<__HIP_MEMORY_SCOPE_AGENT, true, true>("f64 agent, table per XCC", tab, G, seen);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, true, true>("f64 workgroup scope, table per XCC", tab, G, seen);
        run<__HIP_MEMORY_SCOPE_WAVEFRONT, true, true>("f64 wavefront scope, table per XCC", tab, G, seen);
    }
    return 0;
}
