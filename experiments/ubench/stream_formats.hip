// stream_formats.hip — which row format can one workgroup-per-CU kernel read fastest from HBM?
// (input-format decision for the aggregate kernel; GPU box only, not part of the product)
//   hipcc -O3 --offload-arch=gfx950 stream_formats.hip -o stream_formats && ./stream_formats
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int R = 5;                 // 8-byte words per row (key + 4 values)
typedef unsigned long long u64;
struct alignas(16) u64x2 { u64 x, y; };

__device__ __forceinline__ u64 ldnt(const u64 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ u64x2 ldnt2(const u64x2 *p) {
    u64x2 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); return r;
}

// mode 0: SoA, 8 B per lane per column, rows i and i + T (the round-1 aggregate's pattern)
// mode 1: SoA, 16 B per lane per column (rows 2t, 2t+1)
// mode 2: AoS 40-B records, 5 x 8-B loads per row, rows i and i + T
// mode 3: AoS, adjacent row pair per lane = 80 B = 5 x 16-B loads
// mode 4: AoS, wave-contiguous 16-B loads (lane l takes bytes [16 l, 16 l + 16) of each KiB; rows are NOT lane-aligned: bandwidth ceiling only)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(1024) void read_kernel(const u64 *base, size_t n_rows, u64 *out, size_t misalign = 0) {
    const size_t T = 1024;
    const size_t per_wg = (n_rows / gridDim.x) & ~size_t(4095);
    const size_t beg = per_wg * blockIdx.x + misalign, end = beg + per_wg - 4096;
    u64 acc = 0;
    if (MODE == 0) {
        for (size_t i = beg + threadIdx.x; i + (UNROLL - 1) * T < end; i += UNROLL * T) {
            u64 v[UNROLL][R];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) v[u][c] = ldnt(base + (size_t)c * n_rows + i + u * T);
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) acc += v[u][c];
        }
    } else if (MODE == 1) {
        for (size_t i = beg + 2 * threadIdx.x; i + (UNROLL - 1) * 2 * T < end; i += UNROLL * 2 * T) {
            u64x2 v[UNROLL][R];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) v[u][c] = ldnt2(reinterpret_cast<const u64x2 *>(base + (size_t)c * n_rows + i + u * 2 * T));
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) acc += v[u][c].x + v[u][c].y;
        }
    } else if (MODE == 2) {
        for (size_t i = beg + threadIdx.x; i + (UNROLL - 1) * T < end; i += UNROLL * T) {
            u64 v[UNROLL][R];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) v[u][c] = ldnt(base + (i + u * T) * R + c);
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) acc += v[u][c];
        }
    } else if (MODE == 3) {
        for (size_t i = beg + 2 * threadIdx.x; i + (UNROLL - 1) * 2 * T < end; i += UNROLL * 2 * T) {
            u64x2 v[UNROLL][R];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) v[u][c] = ldnt2(reinterpret_cast<const u64x2 *>(base + (i + u * 2 * T) * R) + c);
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int c = 0; c < R; c++) acc += v[u][c].x + v[u][c].y;
        }
    } else {
        const u64x2 *b2 = reinterpret_cast<const u64x2 *>(base + beg * R);
        const size_t n16 = per_wg * R / 2;
        for (size_t i = threadIdx.x; i + (UNROLL * R - 1) * T < n16; i += UNROLL * R * T) {
            u64x2 v[UNROLL * R];
#pragma unroll
            for (int u = 0; u < UNROLL * R; u++) v[u] = ldnt2(b2 + i + u * T);
#pragma unroll
            for (int u = 0; u < UNROLL * R; u++) acc += v[u].x + v[u].y;
        }
    }
    if (acc == 0x123456789ull) out[0] = acc;    // keep the loads alive
}

__global__ __launch_bounds__(256) void copy16_kernel(const u64x2 *in, u64x2 *out, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u64x2 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        out[i] = a; out[i + stride] = b; out[i + 2 * stride] = c; out[i + 3 * stride] = d;
    }
}

template <int MODE, int UNROLL>
static int run(const char *name, const u64 *buf, size_t n_rows, u64 *out, int grid, size_t misalign = 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 5; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((read_kernel<MODE, UNROLL>), dim3(grid), dim3(1024), 0, 0, buf, n_rows, out, misalign);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const size_t per_wg = (n_rows / grid) & ~size_t(4095);
    printf("%-44s grid %4d unroll %d misalign %2zu rows  %.3f ms  %.2f TB/s\n", name, grid, UNROLL, misalign, best, (per_wg - 4096) * grid * R * 8 / best / 1e9);
    return 0;
}

__global__ void fill_random(u64 *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { u64 x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; p[i] = x; }
}

int main(int argc, char **argv) {
    const size_t n = 100000000;
    u64 *buf, *out; CK(hipMalloc(&buf, n * R * 8)); CK(hipMalloc(&out, n * R * 8));
    const bool rnd = argc > 1 && argv[1][0] == 'r';
    if (rnd) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, buf, n * R); CK(hipDeviceSynchronize()); printf("random data\n"); }
    else { CK(hipMemset(buf, 1, n * R * 8)); printf("constant data\n"); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, 0, (const u64x2 *)buf, (u64x2 *)out, n * R / 2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 16 B/lane 4 GB -> 4 GB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * n * R * 8 / ms / 1e9);
    }
    {   // does a read stream run slower on a buffer that a kernel has just written?
        u64 *third; CK(hipMalloc(&third, n * R * 8));
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, third, n * R); CK(hipDeviceSynchronize());
        hipEvent_t a1, a2; CK(hipEventCreate(&a1)); CK(hipEventCreate(&a2));
        auto rd = [&](const char *what, const u64 *p) {
            CK(hipEventRecord(a1));
            hipLaunchKernelGGL((read_kernel<0, 2>), dim3(256), dim3(1024), 0, 0, p, n, third + n * R - 8, (size_t)0);
            CK(hipEventRecord(a2)); CK(hipEventSynchronize(a2));
            float m; CK(hipEventElapsedTime(&m, a1, a2)); printf("  read %-28s %.3f ms\n", what, m); return 0;
        };
        for (int it = 0; it < 2; it++) {
            printf("copy A -> B\n");
            hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, 0, (const u64x2 *)buf, (u64x2 *)out, n * R / 2);
            rd("C (untouched)", third); rd("B (just written)", out); rd("A (copy source)", buf); rd("B again", out); rd("B again", out); rd("C", third);
            printf("fill_random B (8 B stores, every line written whole by one wave)\n");
            hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, out, n * R);
            rd("B (just written)", out); rd("B again", out); rd("C", third);
            printf("hipMemsetAsync B\n");
            CK(hipMemsetAsync(out, 7, n * R * 8, 0));
            rd("B (just written)", out); rd("B again", out);
        }
    }
    for (size_t mis : {0, 1, 5, 8, 16}) run<0, 2>("SoA 8 B/lane (rows i, i+T)", buf, n, out, 256, mis);
    for (int grid : {256}) {
        run<0, 2>("SoA 8 B/lane (rows i, i+T)", buf, n, out, grid);
        run<0, 4>("SoA 8 B/lane (rows i, i+T)", buf, n, out, grid);
        run<1, 1>("SoA 16 B/lane (rows 2t, 2t+1)", buf, n, out, grid);
        run<1, 2>("SoA 16 B/lane (rows 2t, 2t+1)", buf, n, out, grid);
        run<2, 2>("AoS 40 B records, 5 x 8 B per row", buf, n, out, grid);
        run<2, 4>("AoS 40 B records, 5 x 8 B per row", buf, n, out, grid);
        run<3, 1>("AoS row pair, 5 x 16 B per lane", buf, n, out, grid);
        run<3, 2>("AoS row pair, 5 x 16 B per lane", buf, n, out, grid);
        run<4, 1>("AoS wave-contiguous 16 B (ceiling)", buf, n, out, grid);
        run<4, 2>("AoS wave-contiguous 16 B (ceiling)", buf, n, out, grid);
    }
    return 0;
}
