// fresh_read.hip — the first read of a buffer after a kernel wrote it runs ~25 % slower than a second read
// (stream_formats.hip).  Does the load / store flavour change that?   GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;
constexpr int R = 5;

template <int LK>
__device__ __forceinline__ u64 ld(const u64 *p) {
    if (LK == 0) return __builtin_nontemporal_load(p);
    if (LK == 1) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int LK>
__global__ __launch_bounds__(1024) void read_kernel(const u64 *base, size_t n_rows, u64 *out) {
    const size_t T = 1024, per_wg = (n_rows / gridDim.x) & ~size_t(4095);
    const size_t beg = per_wg * blockIdx.x, end = beg + per_wg;
    u64 acc = 0;
    for (size_t i = beg + threadIdx.x; i + T < end; i += 2 * T) {
        u64 v[2][R];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int c = 0; c < R; c++) v[u][c] = ld<LK>(base + (size_t)c * n_rows + i + u * T);
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int c = 0; c < R; c++) acc += v[u][c];
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
// SK 0 plain, 1 nt, 2 sc1 (agent-scope relaxed atomic store: write-through)
template <int SK>
__global__ void fill_kernel(u64 *p, size_t n, u64 salt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        u64 x = (i + salt) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        if (SK == 0) p[i] = x;
        else if (SK == 1) __builtin_nontemporal_store(x, p + i);
        else __hip_atomic_store(p + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    const size_t n = 100000000;
    u64 *b, *dummy; CK(hipMalloc(&b, n * R * 8)); CK(hipMalloc(&dummy, 4096));
    hipEvent_t a1, a2; CK(hipEventCreate(&a1)); CK(hipEventCreate(&a2));
    const char *lname[3] = {"nt load", "plain load", "sc1 load"}, *sname[3] = {"plain store", "nt store", "sc1 store"};
    for (int sk = 0; sk < 3; sk++)
        for (int lk = 0; lk < 3; lk++) {
            float first = 0, second = 0, wr = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(a1));
                if (sk == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(4096), dim3(256), 0, 0, b, n * R, (u64)rep);
                else if (sk == 1) hipLaunchKernelGGL(fill_kernel<1>, dim3(4096), dim3(256), 0, 0, b, n * R, (u64)rep);
                else hipLaunchKernelGGL(fill_kernel<2>, dim3(4096), dim3(256), 0, 0, b, n * R, (u64)rep);
                CK(hipEventRecord(a2)); CK(hipEventSynchronize(a2)); CK(hipEventElapsedTime(&wr, a1, a2));
                for (int pass = 0; pass < 2; pass++) {
                    CK(hipEventRecord(a1));
                    if (lk == 0) hipLaunchKernelGGL(read_kernel<0>, dim3(256), dim3(1024), 0, 0, b, n, dummy);
                    else if (lk == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(256), dim3(1024), 0, 0, b, n, dummy);
                    else hipLaunchKernelGGL(read_kernel<2>, dim3(256), dim3(1024), 0, 0, b, n, dummy);
                    CK(hipEventRecord(a2)); CK(hipEventSynchronize(a2));
                    float m; CK(hipEventElapsedTime(&m, a1, a2)); (pass ? second : first) = m;
                }
            }
            printf("%-12s (%.3f ms) then %-10s: first read %.3f ms, second read %.3f ms\n", sname[sk], wr, lname[lk], first, second);
        }
    return 0;
}
