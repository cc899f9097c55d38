// reduce.hip — whole-column reductions and the join's column gathers.
//
// Replaces simd_{sum,mean,min,max}_{f64,i64} (reference src/optimized/jit/simd.rs:9-112) and the
// null-skipping Int64Column/Float64Column::{sum,mean,min,max} (src/column/int64_column.rs:129-241);
// gathers replace the per-element `col.get(idx)` loops of join_impl
// (src/optimized/split_dataframe/join.rs:296-357, :475-552).
// Both are pure HBM streams: 16-byte loads, wave shuffles, one partial per workgroup.
#include "common.hpp"
#include "device_utils.hpp"

#include <cmath>
#include <limits>

namespace pandrs {

constexpr int RD_THREADS = 512;
constexpr int RD_BLOCKS = 512;        // partials: 512 x 80 B = 40 KB, inside the 64 KB pinned read-back block
constexpr int RD_VEC = 4;             // 16-byte loads in flight per lane and step (8 values)

struct RedPartial {
    double fsum, fsq;      // sum of the values as f64 ((double)v for i64) and of their squares
    uint64_t isum;         // i64: wrapping integer sum
    double mn, mx;         // min / max over the non-null values, NaN operands ignored (f64::min / f64::max); i64: as bits
    double fmn, fmx;       // f64: over the FINITE values only (Float64Column::min/max, float64_column.rs:147-199)
    uint64_t cnt, cnt_fin; // non-null values; f64: finite ones among them
    uint64_t pad;
};

struct RedAcc {
    double fs = 0.0, fq = 0.0;
    uint64_t is = 0, cnt = 0, cfin = 0;
    double mn, mx, fmn, fmx;
    int64_t imn, imx;
};

template <bool IS_F64>
__device__ __forceinline__ void red_take(RedAcc &a, uint64_t b) {
    a.cnt++;
    if (IS_F64) {
        const double d = __longlong_as_double((long long)b);
        a.fs += d; a.fq = fma(d, d, a.fq);
        a.mn = fmin(a.mn, d); a.mx = fmax(a.mx, d);                       // v_min_f64 / v_max_f64: a NaN operand is dropped
        const bool fin = (b & 0x7FF0000000000000ull) != 0x7FF0000000000000ull;
        a.cfin += fin ? 1 : 0;
        a.fmn = fmin(a.fmn, fin ? d : __longlong_as_double(0x7FF0000000000000ll));
        a.fmx = fmax(a.fmx, fin ? d : __longlong_as_double((long long)0xFFF0000000000000ull));
    } else {
        const int64_t v = (int64_t)b;
        const double d = (double)v;
        a.is += b; a.fs += d; a.fq = fma(d, d, a.fq);
        a.imn = v < a.imn ? v : a.imn; a.imx = v > a.imx ? v : a.imx;
    }
}

// One HBM stream: every lane keeps RD_VEC 16-byte non-temporal loads in flight (the round-1 kernel issued single
// 8-byte loads from 2048 x 256 threads and ran at 2.6 TB/s).  HAS_NULLS: the bitmap byte of an 8-row group rides along.
template <bool IS_F64, bool HAS_NULLS>
__global__ __launch_bounds__(RD_THREADS) void reduce_kernel(const uint64_t *data0, const uint8_t *null_bits,
                                                            int64_t n0, RedPartial *partials) {
    // a column that starts 8 bytes off a 16-byte boundary (a sliced Arc<[T]>): row 0 is taken on its own
    const int64_t head = (n0 > 0 && (reinterpret_cast<uintptr_t>(data0) & 8)) ? 1 : 0;
    const uint64_t *data = data0 + head;
    const int64_t n = n0 - head;
    RedAcc a;
    a.mn = a.fmn = __longlong_as_double(0x7FF0000000000000ll);
    a.mx = a.fmx = __longlong_as_double((long long)0xFFF0000000000000ull);
    a.imn = INT64_MAX; a.imx = INT64_MIN;
    struct alignas(16) U2 { uint64_t x, y; };
    const int64_t n2 = n >> 1;                                   // 16-byte pairs
    const U2 *d2 = reinterpret_cast<const U2 *>(data);           // column bases are >= 16-byte aligned (Arc<[T]> / arena)
    const int64_t stride = (int64_t)gridDim.x * RD_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * RD_THREADS + threadIdx.x; i < n2; i += stride * RD_VEC) {
        U2 v[RD_VEC];
        uint32_t nb[RD_VEC];
#pragma unroll
        for (int u = 0; u < RD_VEC; u++) {
            const int64_t j = i + u * stride;
            const int64_t jc = j < n2 ? j : n2 - 1;
            v[u].x = __builtin_nontemporal_load(&d2[jc].x);
            v[u].y = __builtin_nontemporal_load(&d2[jc].y);
            nb[u] = !HAS_NULLS ? 0u : head ? (uint32_t)bit_at(null_bits, 2 * jc + 1) | (uint32_t)bit_at(null_bits, 2 * jc + 2) << 1
                                          : (uint32_t)(null_bits[jc >> 2] >> ((jc & 3) * 2)) & 3u;      // rows 2 jc, 2 jc + 1 (+ head)
        }
#pragma unroll
        for (int u = 0; u < RD_VEC; u++) {
            if (i + u * stride < n2) {
                if (!(nb[u] & 1)) red_take<IS_F64>(a, v[u].x);
                if (!(nb[u] & 2)) red_take<IS_F64>(a, v[u].y);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n & 1) {                                             // odd tail
            const int64_t j = n - 1;
            if (!(HAS_NULLS && bit_at(null_bits, j + head))) red_take<IS_F64>(a, data[j]);
        }
        if (head && !(HAS_NULLS && bit_at(null_bits, 0))) red_take<IS_F64>(a, data0[0]);
    }
    if (!IS_F64) { a.mn = __longlong_as_double(a.imn); a.mx = __longlong_as_double(a.imx); }
    auto lo = [](double x, double y, bool f64) { return f64 ? fmin(x, y) : ((int64_t)__double_as_longlong(x) < (int64_t)__double_as_longlong(y) ? x : y); };
    auto hi = [](double x, double y, bool f64) { return f64 ? fmax(x, y) : ((int64_t)__double_as_longlong(x) > (int64_t)__double_as_longlong(y) ? x : y); };
    for (int d = 32; d >= 1; d >>= 1) {
        a.fs += __shfl_down(a.fs, d, 64);
        a.fq += __shfl_down(a.fq, d, 64);
        a.is += __shfl_down(a.is, d, 64);
        a.cnt += __shfl_down(a.cnt, d, 64);
        a.cfin += __shfl_down(a.cfin, d, 64);
        a.mn = lo(a.mn, __shfl_down(a.mn, d, 64), IS_F64); a.mx = hi(a.mx, __shfl_down(a.mx, d, 64), IS_F64);
        a.fmn = fmin(a.fmn, __shfl_down(a.fmn, d, 64)); a.fmx = fmax(a.fmx, __shfl_down(a.fmx, d, 64));
    }
    __shared__ RedPartial sh[RD_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = RedPartial{a.fs, a.fq, a.is, a.mn, a.mx, a.fmn, a.fmx, a.cnt, a.cfin, 0};
    __syncthreads();
    if (threadIdx.x == 0) {
        RedPartial r = sh[0];
        for (int j = 1; j < RD_THREADS / 64; j++) {
            r.fsum += sh[j].fsum; r.fsq += sh[j].fsq; r.isum += sh[j].isum; r.cnt += sh[j].cnt; r.cnt_fin += sh[j].cnt_fin;
            r.mn = lo(r.mn, sh[j].mn, IS_F64); r.mx = hi(r.mx, sh[j].mx, IS_F64);
            r.fmn = fmin(r.fmn, sh[j].fmn); r.fmx = fmax(r.fmx, sh[j].fmx);
        }
        partials[blockIdx.x] = r;
    }
}

// All the statistics K1's three families of reference functions need (see pandrs_hip_reduce_stats in the header).
int32_t reduce_stats_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *col, int64_t n,
                           pandrs_hip_column_stats *st) {
    if (!c || !col || !st || n < 0) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "reduce: bad arguments");
    if (col->dtype != PANDRS_HIP_I64 && col->dtype != PANDRS_HIP_F64)
        return fail(PANDRS_HIP_ERR_TYPE_MISMATCH, "reduce: dtype %d is not numeric", col->dtype);       // Error::Type (aggregate.rs:57)
    const bool f64 = col->dtype == PANDRS_HIP_F64;
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    int blocks = (int)std::min<int64_t>(RD_BLOCKS, std::max<int64_t>(1, (n / 2 + RD_THREADS * RD_VEC - 1) / (RD_THREADS * RD_VEC)));
    RedPartial *h = reinterpret_cast<RedPartial *>(c->pinned);
    size_t need = Arena::padded(sizeof(RedPartial) * blocks) + (mem_space == PANDRS_HIP_MEM_HOST ? size_t(n) * 8 + (n + 7) / 8 + 4096 : 0) + 4096;
    ST_TRY(c->work.ensure(need, c->stream));
    RedPartial *dp = c->work.take<RedPartial>(blocks);
    const uint64_t *data = (const uint64_t *)col->data;
    const uint8_t *mask = col->null_mask;
    if (mem_space == PANDRS_HIP_MEM_HOST && n > 0) {
        uint64_t *dd = c->work.take<uint64_t>(n);
        HIP_TRY(hipMemcpyAsync(dd, col->data, size_t(n) * 8, hipMemcpyHostToDevice, c->stream));
        data = dd;
        if (mask) {
            uint8_t *dm = c->work.take<uint8_t>((n + 7) / 8);
            HIP_TRY(hipMemcpyAsync(dm, mask, (n + 7) / 8, hipMemcpyHostToDevice, c->stream));
            mask = dm;
        }
    }
    if (n > 0 && (reinterpret_cast<uintptr_t>(data) & 7))
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "reduce: the column must be 8-byte aligned");
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_OTHER);
        if (f64 && mask) hipLaunchKernelGGL((reduce_kernel<true, true>), dim3(blocks), dim3(RD_THREADS), 0, c->stream, data, mask, n, dp);
        else if (f64) hipLaunchKernelGGL((reduce_kernel<true, false>), dim3(blocks), dim3(RD_THREADS), 0, c->stream, data, mask, n, dp);
        else if (mask) hipLaunchKernelGGL((reduce_kernel<false, true>), dim3(blocks), dim3(RD_THREADS), 0, c->stream, data, mask, n, dp);
        else hipLaunchKernelGGL((reduce_kernel<false, false>), dim3(blocks), dim3(RD_THREADS), 0, c->stream, data, mask, n, dp);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(h, dp, sizeof(RedPartial) * blocks, hipMemcpyDeviceToHost, c->stream));
    c->timings.algorithmic_bytes = n * 8 + (col->null_mask ? n / 8 : 0);
    ST_TRY(timings_end(c));
    // host combine of <= 512 partials, in block order (deterministic)
    double fs = 0.0, fq = 0.0; uint64_t is = 0, cnt = 0, cfin = 0;
    const double PINF = std::numeric_limits<double>::infinity();
    double mn = PINF, mx = -PINF, fmn = PINF, fmx = -PINF;
    int64_t imn = INT64_MAX, imx = INT64_MIN;
    for (int b = 0; b < blocks; b++) {
        fs += h[b].fsum; fq += h[b].fsq; is += h[b].isum; cnt += h[b].cnt; cfin += h[b].cnt_fin;
        if (f64) {
            mn = std::fmin(mn, h[b].mn); mx = std::fmax(mx, h[b].mx);
            // +0.0 / -0.0: std::fmin may keep either; the kernel's v_min_f64 orders -0 < +0, do the same here
            if (h[b].mn == 0.0 && mn == 0.0 && std::signbit(h[b].mn)) mn = h[b].mn;
            if (h[b].mx == 0.0 && mx == 0.0 && !std::signbit(h[b].mx)) mx = h[b].mx;
            fmn = std::fmin(fmn, h[b].fmn); fmx = std::fmax(fmx, h[b].fmx);
            if (h[b].fmn == 0.0 && fmn == 0.0 && std::signbit(h[b].fmn)) fmn = h[b].fmn;
            if (h[b].fmx == 0.0 && fmx == 0.0 && !std::signbit(h[b].fmx)) fmx = h[b].fmx;
        } else {
            int64_t a, b2; std::memcpy(&a, &h[b].mn, 8); std::memcpy(&b2, &h[b].mx, 8);
            imn = std::min(imn, a); imx = std::max(imx, b2);
        }
    }
    *st = pandrs_hip_column_stats{};
    st->count = (int64_t)cnt; st->sum_f64 = fs; st->sum_sq = fq;
    if (f64) {
        st->count_finite = (int64_t)cfin; st->min = mn; st->max = mx; st->min_finite = fmn; st->max_finite = fmx;
    } else {
        st->sum_i64 = (int64_t)is; st->count_finite = (int64_t)cnt;
        st->min_i64 = imn; st->max_i64 = imx;
        st->min = st->min_finite = (double)imn; st->max = st->max_finite = (double)imx;
    }
    return 0;
}

int32_t reduce_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *col, int64_t n,
                     double out[4], int64_t *out_count, double *out_sumsq) {
    if (!out || !out_count) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "reduce: bad arguments");
    pandrs_hip_column_stats st;
    if (col && col->dtype != PANDRS_HIP_I64 && col->dtype != PANDRS_HIP_F64)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "reduce: dtype %d is not numeric", col->dtype);
    ST_TRY(reduce_stats_entry(c, mem_space, col, n, &st));
    const bool f64 = col->dtype == PANDRS_HIP_F64;
    out[0] = f64 ? st.sum_f64 : (double)st.sum_i64;
    out[1] = st.count ? out[0] / (double)st.count : 0.0;
    out[2] = st.min; out[3] = st.max;
    *out_count = st.count;
    if (out_sumsq) *out_sumsq = st.sum_sq;
    return 0;
}

// ---- gathers --------------------------------------------------------------------------------------
// kind 0: 8-byte elements, 1: 4-byte, 2: bit-packed source -> byte per row
template <int KIND>
__global__ void gather_kernel(const void *src, const uint8_t *null_bits, const int64_t *idx, int64_t n,
                              uint64_t fill, void *out, int64_t n_src, const int64_t *only_where_negative) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (only_where_negative && only_where_negative[i] >= 0) return;     // (the join-key column: rows that have a left row keep it)
    int64_t j = idx[i];
    bool take = j >= 0 && (n_src < 0 || j < n_src) && !(null_bits && bit_at(null_bits, j));   // n_src < 0: length not known to the ABI
    if (KIND == 0) reinterpret_cast<uint64_t *>(out)[i] = take ? reinterpret_cast<const uint64_t *>(src)[j] : fill;
    else if (KIND == 1) reinterpret_cast<uint32_t *>(out)[i] = take ? reinterpret_cast<const uint32_t *>(src)[j] : (uint32_t)fill;
    else reinterpret_cast<uint8_t *>(out)[i] = take ? (uint8_t)bit_at(reinterpret_cast<const uint8_t *>(src), j) : (uint8_t)fill;
}

// the launch itself; the caller holds the context's mutex
static int32_t gather_locked(pandrs_hip_ctx *c, int kind, const void *src, const uint8_t *mask, const int64_t *idx, int64_t n,
                             uint64_t fill_bits, void *out, int64_t n_src, const int64_t *only_where_negative);

int32_t gather_entry(pandrs_hip_ctx *c, int32_t mem_space, int kind, const void *src, const uint8_t *mask,
                     const int64_t *idx, int64_t n, uint64_t fill_bits, void *out, int64_t n_src, const int64_t *only_where_negative) {
    if (!c || n < 0 || (n && (!idx || !out))) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "gather: bad arguments");
    if (n == 0) return 0;
    if (mem_space == PANDRS_HIP_MEM_HOST)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT,
                    "gather takes device pointers only (the source length is not part of the ABI); "
                    "stage columns with your allocator and pass PANDRS_HIP_MEM_DEVICE");
    std::lock_guard<std::mutex> lock(c->mu);
    return gather_locked(c, kind, src, mask, idx, n, fill_bits, out, n_src, only_where_negative);
}

static int32_t gather_locked(pandrs_hip_ctx *c, int kind, const void *src, const uint8_t *mask, const int64_t *idx, int64_t n,
                             uint64_t fill_bits, void *out, int64_t n_src, const int64_t *only_where_negative) {
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_GATHER);
        dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (kind == 0) hipLaunchKernelGGL(gather_kernel<0>, grid, block, 0, c->stream, src, mask, idx, n, fill_bits, out, n_src, only_where_negative);
        else if (kind == 1) hipLaunchKernelGGL(gather_kernel<1>, grid, block, 0, c->stream, src, mask, idx, n, fill_bits, out, n_src, only_where_negative);
        else hipLaunchKernelGGL(gather_kernel<2>, grid, block, 0, c->stream, src, mask, idx, n, fill_bits, out, n_src, only_where_negative);
        HIP_TRY(hipGetLastError());
    }
    int64_t esz = kind == 0 ? 8 : (kind == 1 ? 4 : 1);
    c->timings.algorithmic_bytes = n * (8 + 2 * esz);
    ST_TRY(timings_end(c));
    return 0;
}

// Column gather with the source length in the signature, so that host-resident columns can be staged:
// the form a host-side shim uses (pandrs_hip_gather_column).
int32_t gather_column_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *src, int64_t n_src,
                            const int64_t *idx, int64_t n, uint64_t fill_bits, void *out) {
    if (!c || !src || n_src < 0 || n < 0 || (n && (!idx || !out)))
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "gather_column: bad arguments");
    if (src->dtype < PANDRS_HIP_I64 || src->dtype > PANDRS_HIP_BOOLBITS)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "gather_column: bad dtype %d", src->dtype);
    const int kind = src->dtype == PANDRS_HIP_U32CODE ? 1 : (src->dtype == PANDRS_HIP_BOOLBITS ? 2 : 0);
    if (n == 0) return 0;
    if (mem_space == PANDRS_HIP_MEM_DEVICE)
        return gather_entry(c, mem_space, kind, src->data, src->null_mask, idx, n, fill_bits, out, n_src);
    const size_t esz = kind == 0 ? 8 : (kind == 1 ? 4 : 1);
    const void *d_src = nullptr; const uint8_t *d_mask = nullptr; int64_t *d_idx = nullptr; void *d_out = nullptr;
    std::lock_guard<std::mutex> lock(c->mu);          // one critical section: the staged buffers live in the context
    {
        HIP_TRY(hipSetDevice(c->device));
        const size_t sbytes = dtype_bytes(src->dtype, n_src), mbytes = (size_t)(n_src + 7) / 8;
        ST_TRY(c->staging.ensure(sbytes + mbytes + size_t(n) * (8 + esz) + 4096, c->stream));
        if (n_src > 0 && src->data) {
            void *p = c->staging.take<uint8_t>(sbytes + 16);
            if (!p) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
            HIP_TRY(hipMemcpyAsync(p, src->data, sbytes, hipMemcpyHostToDevice, c->stream));
            d_src = p;
            if (src->null_mask) {
                uint8_t *m = c->staging.take<uint8_t>(mbytes + 16);
                if (!m) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
                HIP_TRY(hipMemcpyAsync(m, src->null_mask, mbytes, hipMemcpyHostToDevice, c->stream));
                d_mask = m;
            }
        }
        d_idx = c->staging.take<int64_t>(n);
        d_out = c->staging.take<uint8_t>(size_t(n) * esz + 16);
        if (!d_idx || !d_out) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
        HIP_TRY(hipMemcpyAsync(d_idx, idx, size_t(n) * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (!d_src) {                                   // empty source: every row takes the fill value
        for (int64_t i = 0; i < n; i++) {
            if (kind == 0) reinterpret_cast<uint64_t *>(out)[i] = fill_bits;
            else if (kind == 1) reinterpret_cast<uint32_t *>(out)[i] = (uint32_t)fill_bits;
            else reinterpret_cast<uint8_t *>(out)[i] = (uint8_t)fill_bits;
        }
        return 0;
    }
    ST_TRY(gather_locked(c, kind, d_src, d_mask, d_idx, n, fill_bits, d_out, n_src, nullptr));
    HIP_TRY(hipMemcpyAsync(out, d_out, size_t(n) * esz, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// One output column of a join, gathered through the RETAINED pairs of the context's last pandrs_hip_join_indices (side 0: the
// left row of every pair, side 1: the right row) — join.rs:286-552 without the pairs ever leaving HBM.  The source column may be
// resident (PANDRS_HIP_MEM_DEVICE) or a host column (staged); the output goes to either memory space.
// `key_right` (the join-key column, join.rs:364-470): rows whose pair has no left row take the RIGHT key column's value at the
// pair's right row instead of the fill value (`src` is then the left key column, side 0).
int32_t join_gather_entry(pandrs_hip_ctx *c, int32_t src_mem_space, const pandrs_hip_column *src, int64_t n_src, int32_t side,
                          uint64_t fill_bits, int32_t out_mem_space, void *out, const pandrs_hip_column *key_right, int64_t n_right) {
    if (!c || !src || n_src < 0 || (side != 0 && side != 1) || (key_right && (side != 0 || n_right < 0)))
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join_gather: bad arguments");
    if (src->dtype < PANDRS_HIP_I64 || src->dtype > PANDRS_HIP_BOOLBITS)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join_gather: bad dtype %d", src->dtype);
    if (key_right && key_right->dtype != src->dtype)
        return fail(PANDRS_HIP_ERR_TYPE_MISMATCH, "join_gather_key: the key columns have dtypes %d and %d", src->dtype, key_right->dtype);
    const int kind = src->dtype == PANDRS_HIP_U32CODE ? 1 : (src->dtype == PANDRS_HIP_BOOLBITS ? 2 : 0);
    const size_t esz = kind == 0 ? 8 : (kind == 1 ? 4 : 1);
    struct Staged { const void *data; const uint8_t *mask; int64_t n; };
    Staged s0{src->data, src->null_mask, n_src}, s1{key_right ? key_right->data : nullptr, key_right ? key_right->null_mask : nullptr, n_right};
    const int64_t *d_left = nullptr, *d_right = nullptr;
    void *d_out = out;
    int64_t n = 0;
    // ONE critical section for the whole call: the retained pairs, the staged source and the staged output all live in the context,
    // and a concurrent call on the same context between two steps could replace any of them
    std::lock_guard<std::mutex> lock(c->mu);
    {
        if (!c->jn.valid) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no join result retained in this context");
        n = c->jn.n_rows;
        if (n == 0) return 0;
        if (!out) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join_gather: null output");
        d_left = c->jn.left_idx; d_right = c->jn.right_idx;
        HIP_TRY(hipSetDevice(c->device));
        const bool stage_src = src_mem_space == PANDRS_HIP_MEM_HOST, stage_out = out_mem_space == PANDRS_HIP_MEM_HOST;
        auto col_bytes = [&](const pandrs_hip_column *col, int64_t rows) { return Arena::padded(dtype_bytes(col->dtype, rows) + 16) + Arena::padded((size_t)(rows + 7) / 8 + 16); };
        size_t need = 4096;
        if (stage_src) need += col_bytes(src, n_src) + (key_right ? col_bytes(key_right, n_right) : 0);
        if (stage_out) need += Arena::padded(size_t(n) * esz + 16);
        if (stage_src || stage_out) ST_TRY(c->staging.ensure(need, c->stream));
        auto stage = [&](const pandrs_hip_column *col, Staged &st) -> int32_t {
            st.data = nullptr; st.mask = nullptr;
            if (st.n <= 0 || !col->data) return 0;
            const size_t sbytes = dtype_bytes(col->dtype, st.n), mbytes = (size_t)(st.n + 7) / 8;
            void *p = c->staging.take<uint8_t>(sbytes + 16);
            if (!p) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
            HIP_TRY(hipMemcpyAsync(p, col->data, sbytes, hipMemcpyHostToDevice, c->stream));
            st.data = p;
            if (col->null_mask) {
                uint8_t *m = c->staging.take<uint8_t>(mbytes + 16);
                if (!m) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
                HIP_TRY(hipMemcpyAsync(m, col->null_mask, mbytes, hipMemcpyHostToDevice, c->stream));
                st.mask = m;
            }
            return 0;
        };
        if (stage_src) {
            ST_TRY(stage(src, s0));
            if (key_right) ST_TRY(stage(key_right, s1));
        }
        if (stage_out) {
            d_out = c->staging.take<uint8_t>(size_t(n) * esz + 16);
            if (!d_out) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
        }
    }
    // a source without rows is never dereferenced: the kernel's bounds test (n_src = 0) sends every row to the fill value
    ST_TRY(gather_locked(c, kind, s0.data ? s0.data : (const void *)d_left, s0.mask, side ? d_right : d_left, n, fill_bits, d_out,
                         s0.data ? s0.n : 0, nullptr));
    if (key_right)
        ST_TRY(gather_locked(c, kind, s1.data ? s1.data : (const void *)d_left, s1.mask, d_right, n, fill_bits, d_out, s1.data ? s1.n : 0,
                             /*only_where_negative=*/d_left));
    if (out_mem_space == PANDRS_HIP_MEM_HOST) HIP_TRY(hipMemcpyAsync(out, d_out, size_t(n) * esz, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace pandrs
