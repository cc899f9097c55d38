// reduce.hip — whole-column reductions and the join's column gathers.
//
// Replaces simd_{sum,mean,min,max}_{f64,i64} (reference src/optimized/jit/simd.rs:9-112) and the
// null-skipping Int64Column/Float64Column::{sum,mean,min,max} (src/column/int64_column.rs:129-241);
// gathers replace the per-element `col.get(idx)` loops of join_impl
// (src/optimized/split_dataframe/join.rs:296-357, :475-552).
// Both are pure HBM streams: 16-byte loads, wave shuffles, one partial per workgroup.
#include "common.hpp"
#include "device_utils.hpp"

namespace pandrs {

constexpr int RD_THREADS = 256;
constexpr int RD_BLOCKS = 2048;

struct RedPartial {
    double fsum, fsq;   // sum and sum of squares (as f64)
    uint64_t isum;
    uint64_t mn, mx;   // order-preserving encodings
    uint64_t cnt;
};

template <bool IS_F64>
__global__ __launch_bounds__(RD_THREADS) void reduce_kernel(const uint64_t *data, const uint8_t *null_bits,
                                                            int64_t n, RedPartial *partials) {
    double fs = 0.0, fq = 0.0;
    uint64_t is = 0, cnt = 0;
    uint64_t mn = IS_F64 ? enc_f64(__longlong_as_double(0x7FF0000000000000ll)) : enc_i64(INT64_MAX);
    uint64_t mx = IS_F64 ? enc_f64(__longlong_as_double((long long)0xFFF0000000000000ull)) : enc_i64(INT64_MIN);
    for (int64_t i = (int64_t)blockIdx.x * RD_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * RD_THREADS) {
        if (null_bits && bit_at(null_bits, i)) continue;
        uint64_t b = data[i];
        cnt++;
        if (IS_F64) {
            double d = __longlong_as_double((long long)b);
            fs += d; fq += d * d;
            if (d == d) { uint64_t e = enc_f64(d); mn = e < mn ? e : mn; mx = e > mx ? e : mx; }
        } else {
            is += b;
            const double d = (double)(int64_t)b; fq += d * d;
            uint64_t e = enc_i64((int64_t)b);
            mn = e < mn ? e : mn; mx = e > mx ? e : mx;
        }
    }
    // wave reduce
    for (int d = 32; d >= 1; d >>= 1) {
        fs += __shfl_down(fs, d, 64);
        fq += __shfl_down(fq, d, 64);
        is += __shfl_down(is, d, 64);
        cnt += __shfl_down(cnt, d, 64);
        uint64_t a = __shfl_down(mn, d, 64), b2 = __shfl_down(mx, d, 64);
        mn = a < mn ? a : mn; mx = b2 > mx ? b2 : mx;
    }
    __shared__ RedPartial sh[RD_THREADS / 64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = RedPartial{fs, fq, is, mn, mx, cnt};
    __syncthreads();
    if (threadIdx.x == 0) {
        RedPartial r = sh[0];
        for (int j = 1; j < RD_THREADS / 64; j++) {
            r.fsum += sh[j].fsum; r.fsq += sh[j].fsq; r.isum += sh[j].isum; r.cnt += sh[j].cnt;
            r.mn = sh[j].mn < r.mn ? sh[j].mn : r.mn; r.mx = sh[j].mx > r.mx ? sh[j].mx : r.mx;
        }
        partials[blockIdx.x] = r;
    }
}

int32_t reduce_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *col, int64_t n,
                     double out[4], int64_t *out_count, double *out_sumsq) {
    if (!c || !col || !out || !out_count || n < 0) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "reduce: bad arguments");
    if (col->dtype != PANDRS_HIP_I64 && col->dtype != PANDRS_HIP_F64)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "reduce: dtype %d is not numeric", col->dtype);
    const bool f64 = col->dtype == PANDRS_HIP_F64;
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    int blocks = (int)std::min<int64_t>(RD_BLOCKS, std::max<int64_t>(1, (n + RD_THREADS - 1) / RD_THREADS));
    RedPartial *h = reinterpret_cast<RedPartial *>(c->pinned);   // 2048 * 40 B = 80 KB > 64 KB pinned: use blocks <= 1024
    blocks = std::min(blocks, 1024);
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
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_OTHER);
        if (f64) hipLaunchKernelGGL(reduce_kernel<true>, dim3(blocks), dim3(RD_THREADS), 0, c->stream, data, mask, n, dp);
        else hipLaunchKernelGGL(reduce_kernel<false>, dim3(blocks), dim3(RD_THREADS), 0, c->stream, data, mask, n, dp);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(h, dp, sizeof(RedPartial) * blocks, hipMemcpyDeviceToHost, c->stream));
    c->timings.algorithmic_bytes = n * 8 + (col->null_mask ? n / 8 : 0);
    ST_TRY(timings_end(c));
    // host combine of <= 1024 partials, in block order (deterministic)
    double fs = 0.0, fq = 0.0; uint64_t is = 0, cnt = 0;
    uint64_t mn = ~0ull, mx = 0;
    for (int b = 0; b < blocks; b++) {
        fs += h[b].fsum; fq += h[b].fsq; is += h[b].isum; cnt += h[b].cnt;
        mn = h[b].mn < mn ? h[b].mn : mn; mx = h[b].mx > mx ? h[b].mx : mx;
    }
    auto dec_f = [](uint64_t e) { uint64_t b = (e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e; double d; std::memcpy(&d, &b, 8); return d; };
    if (f64) {
        out[0] = fs; out[1] = cnt ? fs / (double)cnt : 0.0;
        out[2] = dec_f(mn); out[3] = dec_f(mx);
    } else {
        out[0] = (double)(int64_t)is; out[1] = cnt ? (double)(int64_t)is / (double)cnt : 0.0;
        out[2] = (double)(int64_t)(mn ^ 0x8000000000000000ull); out[3] = (double)(int64_t)(mx ^ 0x8000000000000000ull);
    }
    *out_count = (int64_t)cnt;
    if (out_sumsq) *out_sumsq = fq;
    return 0;
}

// ---- gathers --------------------------------------------------------------------------------------
// kind 0: 8-byte elements, 1: 4-byte, 2: bit-packed source -> byte per row
template <int KIND>
__global__ void gather_kernel(const void *src, const uint8_t *null_bits, const int64_t *idx, int64_t n,
                              uint64_t fill, void *out, int64_t n_src) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t j = idx[i];
    bool take = j >= 0 && (n_src < 0 || j < n_src) && !(null_bits && bit_at(null_bits, j));   // n_src < 0: length not known to the ABI
    if (KIND == 0) reinterpret_cast<uint64_t *>(out)[i] = take ? reinterpret_cast<const uint64_t *>(src)[j] : fill;
    else if (KIND == 1) reinterpret_cast<uint32_t *>(out)[i] = take ? reinterpret_cast<const uint32_t *>(src)[j] : (uint32_t)fill;
    else reinterpret_cast<uint8_t *>(out)[i] = take ? (uint8_t)bit_at(reinterpret_cast<const uint8_t *>(src), j) : (uint8_t)fill;
}

int32_t gather_entry(pandrs_hip_ctx *c, int32_t mem_space, int kind, const void *src, const uint8_t *mask,
                     const int64_t *idx, int64_t n, uint64_t fill_bits, void *out, int64_t n_src) {
    if (!c || n < 0 || (n && (!idx || !out))) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "gather: bad arguments");
    if (n == 0) return 0;
    if (mem_space == PANDRS_HIP_MEM_HOST)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT,
                    "gather takes device pointers only (the source length is not part of the ABI); "
                    "stage columns with your allocator and pass PANDRS_HIP_MEM_DEVICE");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_GATHER);
        dim3 grid((unsigned)((n + 255) / 256)), block(256);
        if (kind == 0) hipLaunchKernelGGL(gather_kernel<0>, grid, block, 0, c->stream, src, mask, idx, n, fill_bits, out, n_src);
        else if (kind == 1) hipLaunchKernelGGL(gather_kernel<1>, grid, block, 0, c->stream, src, mask, idx, n, fill_bits, out, n_src);
        else hipLaunchKernelGGL(gather_kernel<2>, grid, block, 0, c->stream, src, mask, idx, n, fill_bits, out, n_src);
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
    {
        std::lock_guard<std::mutex> lock(c->mu);
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
    ST_TRY(gather_entry(c, PANDRS_HIP_MEM_DEVICE, kind, d_src, d_mask, d_idx, n, fill_bits, d_out, n_src));
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipMemcpyAsync(out, d_out, size_t(n) * esz, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace pandrs
