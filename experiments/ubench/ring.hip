// ring.hip — the gate for the "LDS-resident tables fed through an on-chip ring" design (VERDICT r2, next #1a).
// Question: if a 256-way scatter writes 16-byte {key, value} records into a SMALL ring that is reused all call
// long (so it can stay in the 256 MiB Infinity Cache) and workgroups on other XCDs read every record once, is
// the round trip faster per byte than the same traffic through a buffer as large as the input (HBM)?
//   T0  floors on this box: read 1.6 GB of (key, value) columns; scatter them 256-way into a 1.6 GB buffer and
//       read that back (two launches: what the radix path pays today at its best)
//   T1  launch-per-chunk: producer launch + consumer launch per chunk, ring of 16/32/64/128 MB reused
//       against a fresh region of a large buffer per chunk
//   T2  ONE persistent launch: every workgroup produces a tile of each chunk into ring slot c % S with sc1
//       (write-through) 16-byte stores, signals a per-chunk counter, and folds ITS partition of chunk c - LAG
//       with sc1 loads after polling that chunk's counter (Guideline 16 R1, counter form).  Spins are bounded.
// Every variant checks a checksum (wrapping u64 sum of all keys and value bits read by the consumers).
//   hipcc -O3 --offload-arch=gfx950 ring.hip -o ring && ./ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
constexpr int AUX_SC1 = 16;

template <bool SC1> __device__ __forceinline__ void st16(rsrc_t rs, void *base, u32 off, u64 a, u64 b) {
    if (SC1) {
        u32x4 v = {(u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32)};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX_SC1);
    } else {
        u32x4 v = {(u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32)};
        *reinterpret_cast<u32x4 *>(reinterpret_cast<char *>(base) + off) = v;
    }
}
template <bool SC1> __device__ __forceinline__ u32x4 ld16(rsrc_t rs, const void *base, u32 off) {
    if (SC1) return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX_SC1);
    return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(base) + off);
}
__device__ __forceinline__ u64 fold4(u32x4 v) { return ((u64)v.y << 32 | v.x) + ((u64)v.w << 32 | v.z); }

// ---------------------------------------------------------------- T0 / T1 kernels
// producer: tile of TILE rows per workgroup; sorted position s of the tile goes to partition s / RUN (RUN = TILE / NP
// rows per (tile, partition) run); the region of partition p holds cap rows, tile k's run sits at k * RUN.
template <int BLOCK, int TILE, bool SC1>
__global__ __launch_bounds__(BLOCK) void produce_kernel(const u64 *keys, const u64 *vals, size_t row0, int np, u32 cap, char *dst,
                                                        u32 dst_bytes, int read_input) {
    const int run = TILE / np;
    const size_t base = row0 + (size_t)blockIdx.x * TILE;
    const rsrc_t rs = make_rsrc(dst, dst_bytes);
    constexpr int R = TILE / BLOCK;
    u64 k[R], v[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int s = j * BLOCK + threadIdx.x;
        if (read_input) { k[j] = __builtin_nontemporal_load(keys + base + s); v[j] = __builtin_nontemporal_load(vals + base + s); }
        else { k[j] = base + s; v[j] = s; }
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int s = j * BLOCK + threadIdx.x;
        const u32 p = s / run, r = s % run;
        const u32 row = p * cap + blockIdx.x * run + r;
        st16<SC1>(rs, dst, row * 16u, k[j], v[j]);
    }
}
template <int BLOCK, bool SC1>
__global__ __launch_bounds__(BLOCK) void consume_kernel(const char *src, u32 src_bytes, u32 cap, u32 rows, u64 *out) {
    const rsrc_t rs = make_rsrc(src, src_bytes);
    u64 acc = 0;
    const u32 beg = blockIdx.x * cap;
    for (u32 i = threadIdx.x; i < rows; i += 4 * BLOCK) {
        u32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const u32 r = i + u * BLOCK; a[u] = r < rows ? ld16<SC1>(rs, src, (beg + r) * 16u) : u32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += fold4(a[u]);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095), acc);
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void read_kernel(const u64 *keys, const u64 *vals, size_t n, u64 *out) {
    u64 acc = 0;
    const size_t per = n / gridDim.x, beg = per * blockIdx.x;
    for (size_t i = threadIdx.x; i < per; i += 4 * BLOCK) {
        u64 a[8];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t r = i + (size_t)u * BLOCK;
            a[2 * u] = r < per ? __builtin_nontemporal_load(keys + beg + r) : 0;
            a[2 * u + 1] = r < per ? __builtin_nontemporal_load(vals + beg + r) : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += a[u];
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095), acc);
}

// ---------------------------------------------------------------- T2 persistent kernel
struct Sync {            // zeroed before every launch
    u32 timeout;         // set by a spin that gave up: every other spin then gives up too
    u32 pad[31];
    u32 done[4096];      // arrivals of producers per chunk
    u32 consumed[4096];  // arrivals of consumers per chunk
};
__device__ __forceinline__ bool wait_ge(u32 *word, u32 target, u32 *tmo) {
    // ONE lane polls, relaxed agent-scope loads (sc1), bounded
    for (u32 spins = 0;; spins++) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if ((spins & 63) == 63 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (spins > 4000000u) { __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}
template <int BLOCK, int TILE, bool PLAIN_ACQ>
__global__ __launch_bounds__(BLOCK) void persistent_kernel(const u64 *keys, const u64 *vals, int n_chunks, int slots, int lag, char *ring,
                                                           u32 ring_bytes, Sync *sy, u64 *out) {
    __shared__ int ok_s;
    const int G = gridDim.x, b = blockIdx.x;
    const int run = TILE / G;                      // rows per (tile, partition)
    const u32 slot_rows = (u32)G * TILE;
    const rsrc_t rs = make_rsrc(ring, ring_bytes);
    constexpr int R = TILE / BLOCK;
    u64 acc = 0;
    bool alive = true;
    for (int c = 0; c < n_chunks + lag && alive; c++) {
        if (c < n_chunks) {
            if (c >= slots) {                      // the slot must have been drained by every consumer
                if (threadIdx.x == 0) ok_s = wait_ge(&sy->consumed[c - slots], (u32)G, &sy->timeout);
                __syncthreads();
                if (!ok_s) { alive = false; break; }
            }
            const size_t base = ((size_t)c * G + b) * TILE;
            u64 k[R], v[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int s = j * BLOCK + threadIdx.x;
                k[j] = __builtin_nontemporal_load(keys + base + s);
                v[j] = __builtin_nontemporal_load(vals + base + s);
            }
            const u32 slot0 = (u32)(c % slots) * slot_rows;
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int s = j * BLOCK + threadIdx.x;
                const u32 p = s / run, r = s % run;
                const u32 row = slot0 + p * TILE + b * run + r;
                if (PLAIN_ACQ) st16<false>(rs, ring, row * 16u, k[j], v[j]); else st16<true>(rs, ring, row * 16u, k[j], v[j]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
            __syncthreads();
            if (threadIdx.x == 0) {
                if (PLAIN_ACQ) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_fetch_add(&sy->done[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const int cc = c - lag;
        if (cc >= 0) {
            if (threadIdx.x == 0) {
                ok_s = wait_ge(&sy->done[cc], (u32)G, &sy->timeout);
                if (PLAIN_ACQ) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            }
            __syncthreads();
            if (!ok_s) { alive = false; break; }
            const u32 seg = (u32)(cc % slots) * slot_rows + (u32)b * TILE;
            u32x4 a[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                const u32 r = j * BLOCK + threadIdx.x;
                a[j] = PLAIN_ACQ ? ld16<false>(rs, ring, (seg + r) * 16u) : ld16<true>(rs, ring, (seg + r) * 16u);
            }
#pragma unroll
            for (int j = 0; j < R; j++) acc += fold4(a[j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&sy->consumed[cc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095), acc);
}

// ---------------------------------------------------------------- T3: wave-specialised persistent kernel
// 16 waves per workgroup, one workgroup per CU: waves 0-7 produce (input -> ring, sc1 stores, next tile's loads in flight
// under this tile's stores), waves 8-15 consume (ring -> registers, sc1 loads).  Both halves run the same two barriers per
// phase, so the workgroup barrier is the only intra-workgroup synchronisation.
template <int TILE, bool PREFETCH>
__global__ __launch_bounds__(1024) void specialised_kernel(const u64 *keys, const u64 *vals, int n_chunks, int slots, int lag, char *ring,
                                                           u32 ring_bytes, Sync *sy, u64 *out) {
    __shared__ int ok_s[2];
    constexpr int HALF = 512;
    const int G = gridDim.x, b = blockIdx.x;
    const bool producer = threadIdx.x < HALF;
    const int t = threadIdx.x & (HALF - 1);
    const int run = TILE / G;
    const u32 slot_rows = (u32)G * TILE;
    const rsrc_t rs = make_rsrc(ring, ring_bytes);
    constexpr int R = TILE / HALF;
    u64 acc = 0;
    u64 k[R], v[R], k2[R], v2[R];
    if (threadIdx.x < 2) ok_s[threadIdx.x] = 1;
    if (producer && PREFETCH) {
        const size_t base = (size_t)b * TILE;
#pragma unroll
        for (int j = 0; j < R; j++) { k[j] = __builtin_nontemporal_load(keys + base + j * HALF + t); v[j] = __builtin_nontemporal_load(vals + base + j * HALF + t); }
    }
    __syncthreads();
    for (int c = 0; c < n_chunks + lag; c++) {
        const int cc = c - lag;
        if (producer) {
            if (c < n_chunks && c >= slots && t == 0) ok_s[0] = wait_ge(&sy->consumed[c - slots], (u32)G, &sy->timeout);
        } else {
            if (cc >= 0 && t == 0) ok_s[1] = wait_ge(&sy->done[cc], (u32)G, &sy->timeout);
        }
        __syncthreads();
        if (!ok_s[0] || !ok_s[1]) break;
        if (producer) {
            if (c < n_chunks) {
                if (PREFETCH) {
                    if (c + 1 < n_chunks) {
                        const size_t base = ((size_t)(c + 1) * G + b) * TILE;
#pragma unroll
                        for (int j = 0; j < R; j++) { k2[j] = __builtin_nontemporal_load(keys + base + j * HALF + t); v2[j] = __builtin_nontemporal_load(vals + base + j * HALF + t); }
                    }
                } else {
                    const size_t base = ((size_t)c * G + b) * TILE;
#pragma unroll
                    for (int j = 0; j < R; j++) { k[j] = __builtin_nontemporal_load(keys + base + j * HALF + t); v[j] = __builtin_nontemporal_load(vals + base + j * HALF + t); }
                }
                const u32 slot0 = (u32)(c % slots) * slot_rows;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const int s = j * HALF + t;
                    const u32 p = s / run, r = s % run;
                    st16<true>(rs, ring, (slot0 + p * TILE + b * run + r) * 16u, k[j], v[j]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (PREFETCH) {
#pragma unroll
                    for (int j = 0; j < R; j++) { k[j] = k2[j]; v[j] = v2[j]; }
                }
            }
        } else if (cc >= 0) {
            const u32 seg = (u32)(cc % slots) * slot_rows + (u32)b * TILE;
            u32x4 a[R];
#pragma unroll
            for (int j = 0; j < R; j++) a[j] = ld16<true>(rs, ring, (seg + j * HALF + t) * 16u);
#pragma unroll
            for (int j = 0; j < R; j++) acc += fold4(a[j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (t == 0) {
            if (producer) { if (c < n_chunks) __hip_atomic_fetch_add(&sy->done[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else if (cc >= 0) __hip_atomic_fetch_add(&sy->consumed[cc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + ((blockIdx.x * 16 + (threadIdx.x >> 6)) & 4095), acc);
}

// ---------------------------------------------------------------- T4: does the Infinity Cache absorb writes?
// every workgroup streams 16-byte stores over ITS slice of a region, sweeps times; total bytes written are the same
template <bool SC1>
__global__ __launch_bounds__(1024) void sweep_write_kernel(char *dst, size_t region_bytes, int sweeps) {
    const size_t per = region_bytes / gridDim.x;
    char *mine = dst + per * blockIdx.x;
    for (int s = 0; s < sweeps; s++)
        for (size_t off = (size_t)threadIdx.x * 16; off < per; off += 1024 * 16) {
            u32x4 v = {(u32)off, (u32)s, 1u, 2u};
            if (SC1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(mine + off));
            else *reinterpret_cast<u32x4 *>(mine + off) = v;
        }
}
__global__ __launch_bounds__(1024) void sweep_read_kernel(const char *src, size_t region_bytes, int sweeps, u64 *out) {
    const size_t per = region_bytes / gridDim.x;
    const char *mine = src + per * ((blockIdx.x + 3) % gridDim.x);      // another workgroup's slice (another XCD's)
    u64 acc = 0;
    for (int s = 0; s < sweeps; s++)
        for (size_t off = (size_t)threadIdx.x * 16; off < per; off += 4 * 1024 * 16) {
            u32x4 a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = off + u * 16384 < per ? *reinterpret_cast<const u32x4 *>(mine + off + u * 16384) : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) acc += fold4(a[u]);
        }
    if (acc == 0x123456789ull) out[0] = acc;
}

// ---------------------------------------------------------------- host
static float time_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

int main(int argc, char **argv) {
    const size_t N = 100u << 20;                   // 104.9 M rows = 1.68 GB of (key, value)
    u64 *keys, *vals, *out;
    char *big;
    const size_t big_bytes = (size_t)4 << 30;
    CK(hipMalloc(&keys, N * 8)); CK(hipMalloc(&vals, N * 8)); CK(hipMalloc(&out, 4096 * 8)); CK(hipMalloc(&big, big_bytes));
    Sync *sy; CK(hipMalloc(&sy, sizeof(Sync)));
    {   // fill: key = i * odd constant, value = i
        std::vector<u64> h(N);
        for (size_t i = 0; i < N; i++) h[i] = i * 0x9E3779B97F4A7C15ull;
        CK(hipMemcpy(keys, h.data(), N * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < N; i++) h[i] = i;
        CK(hipMemcpy(vals, h.data(), N * 8, hipMemcpyHostToDevice));
    }
    auto checksum = [&](size_t rows) { u64 s = 0; for (size_t i = 0; i < rows; i++) s += i * 0x9E3779B97F4A7C15ull + i; return s; };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto get_out = [&]() { std::vector<u64> h(4096); CK(hipMemcpy(h.data(), out, 4096 * 8, hipMemcpyDeviceToHost)); u64 t = 0; for (u64 x : h) t += x; return t; };
    const double GB = (double)N * 16 / 1e9;

    if (argc > 1 && argv[1][0] == '4') {
        const size_t total = (size_t)2 << 30;
        for (int mb : {16, 64, 128, 192, 512, 2048}) {
            const size_t region = (size_t)mb << 20; const int sweeps = total / region;
            for (int mode = 0; mode < 3; mode++) {
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) sweep_write_kernel<false><<<1024, 1024>>>(big, region, sweeps);
                    else if (mode == 1) sweep_write_kernel<true><<<1024, 1024>>>(big, region, sweeps);
                    else sweep_read_kernel<<<1024, 1024>>>(big, region, sweeps, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    best = std::min(best, time_ms(e0, e1));
                }
                printf("T4 %s 2 GB as %4d MB region x %3d sweeps: %.3f ms  %.2f TB/s\n", mode == 0 ? "plain stores" : mode == 1 ? "nt stores   " : "loads       ", mb, sweeps, best, 2.147 / best);
            }
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == '5') {
        // launch-per-chunk, phases separated: produce-only, consume-only, both; ring vs fresh; 128 / 64 MB chunks
        constexpr int TILE = 8192, NP = 256;
        for (int ring_mb : {64, 128})
        for (int what = 0; what < 3; what++)
        for (int fresh = 0; fresh < 2; fresh++)
        for (int read_input = 0; read_input < 2; read_input++) {
            if (what == 1 && read_input) continue;
            const size_t chunk_rows = (size_t)ring_mb << 20 >> 4;
            const int tiles = chunk_rows / TILE; const u32 cap = chunk_rows / NP;
            const int n_chunks = N / chunk_rows;
            const size_t span = fresh ? big_bytes / ((size_t)ring_mb << 20) : 1;
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                for (int c = 0; c < n_chunks; c++) {
                    char *dst = big + (size_t)(c % span) * ((size_t)ring_mb << 20);
                    if (what != 1) produce_kernel<1024, TILE, false><<<tiles, 1024>>>(keys, vals, c * chunk_rows, NP, cap, dst, (u32)(chunk_rows * 16), read_input);
                    if (what != 0) consume_kernel<1024, false><<<4 * NP, 1024>>>(dst, (u32)(chunk_rows * 16), cap / 4, cap / 4, out);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                best = std::min(best, time_ms(e0, e1));
            }
            printf("T5 chunk %3d MB x %2d, %s, %s, %s: %.3f ms  (%.1f us per chunk)\n", ring_mb, n_chunks, what == 0 ? "produce only" : what == 1 ? "consume only" : "produce+consume",
                   fresh ? "FRESH" : "RING ", read_input ? "input from HBM" : "no input read ", best, best * 1000 / n_chunks);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == '6') {
        const size_t total = (size_t)2 << 30;
        for (int mb : {64, 128, 192, 2048}) {
            const size_t region = (size_t)mb << 20; const int sweeps = total / region;
            for (int mode = 0; mode < 4; mode++) {
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) sweep_write_kernel<false><<<1024, 1024>>>(big, region, sweeps);
                    else if (mode == 1) for (int s2 = 0; s2 < sweeps; s2++) sweep_write_kernel<false><<<1024, 1024>>>(big, region, 1);
                    else if (mode == 2) sweep_read_kernel<<<1024, 1024>>>(big, region, sweeps, out);
                    else for (int s2 = 0; s2 < sweeps; s2++) sweep_read_kernel<<<1024, 1024>>>(big, region, 1, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    best = std::min(best, time_ms(e0, e1));
                }
                printf("T6 %s 2 GB as %4d MB region x %3d sweeps: %.3f ms  %.2f TB/s\n", mode == 0 ? "stores, one launch      " : mode == 1 ? "stores, launch per sweep" : mode == 2 ? "loads, one launch       " : "loads, launch per sweep ", mb, sweeps, best, 2.147 / best);
            }
        }
        return 0;
    }
    // ---- T0a: read floor
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(out, 0, 4096 * 8));
        CK(hipEventRecord(e0));
        read_kernel<1024><<<1024, 1024>>>(keys, vals, N, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("T0a read %.2f GB of columns: %.3f ms  %.2f TB/s  %s\n", GB, time_ms(e0, e1), GB / time_ms(e0, e1), get_out() == checksum(N) ? "ok" : "CHECKSUM MISMATCH");
    }
    // ---- T0b: whole-input scatter into a 1.68 GB buffer + read back (two launches)
    {
        constexpr int TILE = 8192, NP = 256;
        const int tiles = N / TILE; const u32 cap = N / NP;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(out, 0, 4096 * 8));
            hipEvent_t em; CK(hipEventCreate(&em));
            CK(hipEventRecord(e0));
            produce_kernel<1024, TILE, false><<<tiles, 1024>>>(keys, vals, 0, NP, cap, big, (u32)(N * 16), 1);
            CK(hipEventRecord(em));
            consume_kernel<1024, false><<<NP, 1024>>>(big, (u32)(N * 16), cap, cap, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            printf("T0b scatter 256-way into 1.68 GB + read back: %.3f + %.3f = %.3f ms  %s\n", time_ms(e0, em), time_ms(em, e1), time_ms(e0, e1),
                   get_out() == checksum(N) ? "ok" : "CHECKSUM MISMATCH");
        }
    }
    // ---- T1: launch per chunk, ring vs fresh regions
    for (int sc1 = 0; sc1 < 2; sc1++)
    for (int ring_mb : {16, 32, 64, 128}) {
        for (int fresh = 0; fresh < 2; fresh++) {
            for (int read_input = 1; read_input >= 0; read_input--) {
                if (sc1 && (!read_input)) continue;
                constexpr int TILE = 8192, NP = 256;
                const size_t chunk_rows = (size_t)ring_mb << 20 >> 4;
                const int tiles = chunk_rows / TILE; const u32 cap = chunk_rows / NP;
                const int n_chunks = N / chunk_rows;
                const size_t span = fresh ? big_bytes / ((size_t)ring_mb << 20) : 1;
                float best = 1e9;
                bool ok = true;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(out, 0, 4096 * 8));
                    CK(hipEventRecord(e0));
                    for (int c = 0; c < n_chunks; c++) {
                        char *dst = big + (size_t)(c % span) * ((size_t)ring_mb << 20);
                        if (sc1) {
                            produce_kernel<1024, TILE, true><<<tiles, 1024>>>(keys, vals, c * chunk_rows, NP, cap, dst, (u32)(chunk_rows * 16), read_input);
                            consume_kernel<1024, true><<<NP, 1024>>>(dst, (u32)(chunk_rows * 16), cap, cap, out);
                        } else {
                            produce_kernel<1024, TILE, false><<<tiles, 1024>>>(keys, vals, c * chunk_rows, NP, cap, dst, (u32)(chunk_rows * 16), read_input);
                            consume_kernel<1024, false><<<NP, 1024>>>(dst, (u32)(chunk_rows * 16), cap, cap, out);
                        }
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    best = std::min(best, time_ms(e0, e1));
                    if (read_input && get_out() != checksum((size_t)n_chunks * chunk_rows)) ok = false;
                }
                printf("T1 %s chunk %3d MB x %3d launches-pairs, %s, %s: %.3f ms per %.2f GB  (%.2f TB/s of input)  %s\n", sc1 ? "sc1  " : "plain", ring_mb, n_chunks,
                       fresh ? "FRESH 4 GB region" : "RING reused      ", read_input ? "input from HBM" : "no input read ", best, GB, GB / best, ok ? "ok" : "CHECKSUM MISMATCH");
            }
        }
    }
    // ---- T2: one persistent launch
    auto run_t2 = [&](auto kernel, const char *name, int G, int tile, int slots, int lag, bool fresh) {
        const size_t chunk_rows = (size_t)G * tile;
        const int n_chunks = std::min<size_t>(N / chunk_rows, 4000);
        const int eff_slots = fresh ? std::min<size_t>(n_chunks, (big_bytes - 1) / (chunk_rows * 16)) : slots;
        const size_t ring_bytes = (size_t)eff_slots * chunk_rows * 16;
        if (ring_bytes > big_bytes || ring_bytes >= ((size_t)1 << 32)) { printf("T2 %s skipped (ring %zu MB)\n", name, ring_bytes >> 20); return; }
        float best = 1e9; bool ok = true; u32 tmo = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemsetAsync(sy, 0, sizeof(Sync)));
            CK(hipMemsetAsync(out, 0, 4096 * 8));
            CK(hipEventRecord(e0));
            kernel<<<G, 1024>>>(keys, vals, n_chunks, eff_slots, lag, big, (u32)ring_bytes, sy, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            best = std::min(best, time_ms(e0, e1));
            CK(hipMemcpy(&tmo, &sy->timeout, 4, hipMemcpyDeviceToHost));
            if (tmo || get_out() != checksum((size_t)n_chunks * chunk_rows)) ok = false;
        }
        const double gb = (double)n_chunks * chunk_rows * 16 / 1e9;
        printf("T2 %s G %d tile %d slots %2d (%4zu MB ring%s) lag %d: %.3f ms per %.2f GB  (%.2f TB/s of input)  %s%s\n", name, G, tile, eff_slots, ring_bytes >> 20,
               fresh ? ", FRESH" : "", lag, best, gb, gb / best, ok ? "ok" : "CHECKSUM MISMATCH", tmo ? " TIMEOUT" : "");
        fflush(stdout);
    };
    for (int fresh = 0; fresh < 2; fresh++) {
        for (int slots : {2, 4, 8})
            for (int lag : {1, 2, 3}) {
                if (lag >= slots && !fresh) continue;
                if (fresh && slots != 4) continue;
                run_t2(persistent_kernel<1024, 8192, false>, "sc1   1024x8192", 256, 8192, slots, lag, fresh);
                run_t2(persistent_kernel<1024, 4096, false>, "sc1   1024x4096", 256, 4096, slots, lag, fresh);
            }
        run_t2(persistent_kernel<1024, 8192, true>, "plain+fence 8192", 256, 8192, 4, 2, fresh);
        run_t2(persistent_kernel<1024, 4096, true>, "plain+fence 4096", 256, 4096, 4, 2, fresh);
    }
    for (int fresh = 0; fresh < 2; fresh++)
        for (int slots : {3, 4, 8, 16})
            for (int lag : {1, 2, 4}) {
                if (lag >= slots && !fresh) continue;
                if (fresh && slots != 4) continue;
                run_t2(specialised_kernel<4096, true>, "T3 halves 4096 prefetch", 256, 4096, slots, lag, fresh);
                run_t2(specialised_kernel<8192, false>, "T3 halves 8192         ", 256, 8192, slots, lag, fresh);
                run_t2(specialised_kernel<4096, false>, "T3 halves 4096         ", 256, 4096, slots, lag, fresh);
            }
    return 0;
}
