// device_utils.hpp — device-side helpers shared by the groupby / join kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pandrs {

constexpr uint64_t EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;  // LDS/global table sentinel
constexpr uint64_t CANON_NAN = 0x7FF8000000000000ull;
constexpr int DT_CELL = 4;  // internal: already-normalised 8-byte key cells + byte null flags

// Key source.  For reference dtypes `null_bits` is the LSB-first bitmap (1 = null,
// reference src/core/column.rs:163-177); for DT_CELL `null_bytes` has one byte per row.
struct KeyDesc {
    const void *data;
    const uint8_t *null_bits;
    const uint8_t *null_bytes;
    int dtype;
};

__device__ __forceinline__ bool bit_at(const uint8_t *bits, int64_t i) {
    return (bits[i >> 3] >> (i & 7)) & 1;
}

__device__ __forceinline__ bool key_is_null(const KeyDesc &k, int64_t i) {
    if (k.null_bits) return bit_at(k.null_bits, i);
    if (k.null_bytes) return k.null_bytes[i] != 0;
    return false;
}

// 8-byte key cell: i64 value / f64 bits with all NaNs collapsed (the reference groups on
// val.to_string(), so every NaN is "NaN" and 0.0 != -0.0; grouping.rs:79) / zero-extended
// string-pool code / bool bit.
__device__ __forceinline__ uint64_t key_cell(const KeyDesc &k, int64_t i) {
    switch (k.dtype) {
    case PANDRS_HIP_I64:
    case DT_CELL:
        return reinterpret_cast<const uint64_t *>(k.data)[i];
    case PANDRS_HIP_F64: {
        uint64_t b = reinterpret_cast<const uint64_t *>(k.data)[i];
        return ((b & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) ? CANON_NAN : b;
    }
    case PANDRS_HIP_U32CODE:
        return reinterpret_cast<const uint32_t *>(k.data)[i];
    default:
        return bit_at(reinterpret_cast<const uint8_t *>(k.data), i) ? 1ull : 0ull;
    }
}

// Row of the estimate's sample: one row per `stride` rows, at a pseudo-random offset inside its stride.  A FIXED stride aliases with
// periodic row layouts — every entity 3 rows, 24 rows per day: with 3 | stride every sampled row is the first of its run, its
// neighbour always equal, and the adjacent-pair statistic read "0 of 262144 pairs differ": the run bound collapsed the estimate to the
// sample's own distinct count (230 K for 1 M groups) and the call paid for it with overflowing tables (49 ms).
__device__ __forceinline__ int64_t sample_row(int64_t s, int64_t stride) {
    uint32_t h = (uint32_t)s * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
    return s * stride + (int64_t)(((uint64_t)h * (uint64_t)stride) >> 32);
}

// 32-bit mix of a 64-bit cell: three 32-bit multiplies.  High bits pick the radix partition
// (mulhi), a re-multiplied copy picks the LDS slot, so the two are decorrelated.
__device__ __forceinline__ uint32_t hash32(uint64_t k, uint32_t seed) {
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t x = (lo ^ seed) * 0x9E3779B1u + hi * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 13;
    return x;
}
__device__ __forceinline__ uint32_t part_of(uint32_t h, uint32_t P) { return __umulhi(h, P); }
__device__ __forceinline__ uint32_t slot_of(uint32_t h, uint32_t T) {
    return __umulhi(h * 0x9E3779B1u + 0x7F4A7C15u, T);
}
// owner rank of a key for the multi-GPU exchange: independent of the partition hash
__device__ __forceinline__ uint32_t owner_of(uint64_t k, uint32_t n_ranks) {
    return __umulhi(hash32(k, 0x5BD1E995u) * 0xC2B2AE35u, n_ranks);
}

// order-preserving u64 encodings so f64 / i64 min-max run on native ds_min_u64 / ds_max_u64.
__device__ __forceinline__ uint64_t enc_f64(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(uint64_t e) {
    uint64_t b = (e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ uint64_t enc_i64(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
__device__ __forceinline__ int64_t dec_i64(uint64_t e) { return (int64_t)(e ^ 0x8000000000000000ull); }

// Workgroup barrier that orders LDS traffic only.  hipcc's __syncthreads() also drains the
// vector-memory counter (s_waitcnt vmcnt(0)), i.e. every wave would wait for its in-flight global
// stores and prefetch loads at each barrier; the kernels here exchange data between waves through
// LDS alone, so lgkmcnt(0) + s_barrier is sufficient and keeps HBM traffic in flight across barriers.
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- block-wide exclusive scan of one value per thread (THREADS multiple of 64, <= 1024) -------
// `wave_tot` = 17-entry LDS scratch.  Returns the exclusive prefix; *total gets the block sum.
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot,
                                                         uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = THREADS / 64;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    block_sync_lds();
    if (wave == 0) {
        uint32_t w = lane < NW ? wave_tot[lane] : 0u;
        uint32_t winc = w;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            uint32_t t = __shfl_up(winc, d, 64);
            if (lane >= d) winc += t;
        }
        if (lane < NW) wave_tot[lane] = winc - w;  // exclusive wave base
        if (lane == NW - 1) wave_tot[16] = winc;
    }
    block_sync_lds();
    uint32_t base = wave_tot[wave];
    if (total) *total = wave_tot[16];
    uint32_t r = base + inc - v;
    block_sync_lds();  // wave_tot reusable after return
    return r;
}

// Length of the run of equal keys that starts at i in a sorted array of n keys: gallop, then
// binary search — O(log run) reads, whatever the run length.
__device__ __forceinline__ uint32_t sorted_run_length(const uint64_t *keys, uint32_t i, uint32_t n) {
    const uint64_t k = keys[i];
    uint32_t lo = i, step = 1;
    while (lo + step < n && keys[lo + step] == k) { lo += step; step <<= 1; }
    uint32_t hi = lo + step < n ? lo + step : n;        // keys[hi] != k, or hi == n
    while (lo + 1 < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] == k) lo = mid; else hi = mid;
    }
    return lo + 1 - i;
}

}  // namespace pandrs
