// segsort.hip — segmented sort of radix partitions of ANY size (gfx950, wave64).
//
// After a radix partition every key's rows sit in one partition, but nothing orders them.  The
// reference's HashMap<key, Vec<usize>> gives two orders for free that device paths must rebuild:
// a key's build rows ascending by row (join.rs:114, :156-158) and, for Median, a group's values
// ascending (aggregation.rs:585-604, :703-722).  Both are "sort every partition by (key, payload)".
//
//   1. a device-built task list cuts every partition into 8192-element tiles;
//   2. chunk sort: one workgroup per tile, bitonic sort in LDS (the whole job for partitions that
//      fit one tile — the common case, one read and one write of the data);
//   3. merge passes, only for partitions with more than one tile: run length doubles per pass; one
//      workgroup per OUTPUT tile finds its two input windows by a merge-path search on the tile's
//      diagonals (global memory, O(log n) reads), stages them in LDS, and every thread merges 8
//      outputs after its own merge-path search in LDS.  Ping-pong between the array and a scratch
//      copy; an odd number of passes ends with a copy back.
// Every pass streams the multi-tile partitions once: HBM-bound, no atomics.
#include "engine.hpp"

namespace pandrs {

constexpr int SS_THREADS = 1024;
constexpr int SS_EPT = SS_TILE / SS_THREADS;     // outputs per thread in a merge pass

// composite order: key, then payload
template <typename PT>
__device__ __forceinline__ bool pair_le(uint64_t ka, PT pa, uint64_t kb, PT pb) {
    return ka < kb || (ka == kb && pa <= pb);
}

// One workgroup: exclusive scan of the partitions' tile counts, one task per tile.
// counters[0] = tasks written, counters[1] = most tiles in one partition.
__global__ __launch_bounds__(SS_THREADS) void build_sort_tasks_kernel(const uint32_t *offsets, uint32_t NB, uint32_t n_parts, const uint8_t *only,
                                                                      SortTask *tasks, uint32_t max_tasks, uint32_t *counters) {
    __shared__ uint32_t wt[17];
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_parts; base += SS_THREADS) {
        const uint32_t p = base + threadIdx.x;
        uint32_t beg = 0, end = 0;
        if (p < n_parts && (!only || only[p])) { beg = offsets[(size_t)p * NB]; end = offsets[(size_t)(p + 1) * NB]; }
        const uint32_t nt = (end - beg + SS_TILE - 1) / SS_TILE;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan<SS_THREADS>(nt, wt, &tot);
        if (nt) atomicMax(&s_max, nt);
        for (uint32_t t = 0; t < nt; t++)
            if (carry + ex + t < max_tasks) tasks[carry + ex + t] = SortTask{beg, end, t, nt > 1 ? 1u : 0u};
        carry += tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) { counters[0] = min(carry, max_tasks); counters[1] = s_max; }
}

// LDS: sk[SS_TILE] u64 | sp[SS_TILE] PT.  ENC: 0 payload as is, 1 f64 -> order-preserving u64, 2 i64 -> u64.
template <typename PT, int ENC>
__global__ __launch_bounds__(SS_THREADS) void chunk_sort_kernel(const SortTask *tasks, const uint32_t *counters,
                                                                uint64_t *keys, PT *pay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x >= counters[0]) return;
    const SortTask t = tasks[blockIdx.x];
    const uint32_t beg = t.pbeg + t.tile * SS_TILE, end = min(beg + SS_TILE, t.pend), n = end - beg;
    uint64_t *sk = reinterpret_cast<uint64_t *>(smem);
    PT *sp = reinterpret_cast<PT *>(sk + SS_TILE);
    uint32_t n2 = 64;
    while (n2 < n) n2 <<= 1;
    for (uint32_t i = threadIdx.x; i < n2; i += SS_THREADS) {
        uint64_t k = ~0ull;
        PT p = (PT)~(PT)0;
        if (i < n) {
            k = keys[beg + i];
            p = pay[beg + i];
            if (ENC == 1) p = (PT)enc_f64(__longlong_as_double((long long)p));
            if (ENC == 2) p = (PT)enc_i64((int64_t)p);
        }
        sk[i] = k; sp[i] = p;
    }
    __syncthreads();
    // every thread owns SS_CPT compare-exchanges per step (indexed by CE, so no thread idles on the upper half of
    // a pair) and issues all of their LDS reads before the first compare: the step is LDS-latency bound otherwise
    constexpr int SS_CPT = SS_TILE / 2 / SS_THREADS;
    for (uint32_t k = 2, lk = 1; k <= n2; k <<= 1, lk++) {
        for (int lj = (int)lk - 1; lj >= 0; lj--) {
            const uint32_t j = 1u << lj;
            uint32_t l[SS_CPT];
            uint64_t ka[SS_CPT], kb[SS_CPT];
            PT pa[SS_CPT], pb[SS_CPT];
#pragma unroll
            for (int q = 0; q < SS_CPT; q++) {
                const uint32_t c = min((uint32_t)(q * SS_THREADS) + threadIdx.x, (n2 >> 1) - 1);
                l[q] = ((c >> lj) << (lj + 1)) | (c & (j - 1));
                ka[q] = sk[l[q]]; kb[q] = sk[l[q] + j];
                pa[q] = sp[l[q]]; pb[q] = sp[l[q] + j];
            }
#pragma unroll
            for (int q = 0; q < SS_CPT; q++) {
                if ((uint32_t)(q * SS_THREADS) + threadIdx.x >= (n2 >> 1)) continue;
                const bool gt = !pair_le<PT>(ka[q], pa[q], kb[q], pb[q]);
                const bool up = (l[q] & k) == 0;
                if (gt == up) { sk[l[q]] = kb[q]; sk[l[q] + j] = ka[q]; sp[l[q]] = pb[q]; sp[l[q] + j] = pa[q]; }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += SS_THREADS) { keys[beg + i] = sk[i]; pay[beg + i] = sp[i]; }
}

// merge path: how many of the first d merged elements come from A (ties go to A)
template <typename PT, typename KP, typename PP>
__device__ __forceinline__ uint32_t merge_path(KP ak, PP ap, uint32_t la, KP bk, PP bp, uint32_t lb, uint32_t d) {
    uint32_t lo = d > lb ? d - lb : 0u, hi = min(d, la);
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pair_le<PT>(ak[mid], ap[mid], bk[d - 1 - mid], bp[d - 1 - mid])) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One workgroup per output tile of a multi-tile partition; `run` = sorted run length entering the pass.
template <typename PT>
__global__ __launch_bounds__(SS_THREADS) void merge_pass_kernel(const SortTask *tasks, const uint32_t *counters,
                                                                const uint64_t *src_k, const PT *src_p,
                                                                uint64_t *dst_k, PT *dst_p, uint32_t run) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_split[2];
    if (blockIdx.x >= counters[0]) return;
    const SortTask t = tasks[blockIdx.x];
    if (!t.multi) return;
    const uint32_t np = t.pend - t.pbeg;
    const uint32_t o0 = t.tile * SS_TILE, o1 = min(o0 + SS_TILE, np);
    const uint64_t pair = (uint64_t)o0 / (2ull * run);
    const uint32_t abeg = (uint32_t)min<uint64_t>(pair * 2ull * run, np);
    const uint32_t aend = (uint32_t)min<uint64_t>((uint64_t)abeg + run, np);
    const uint32_t bend = (uint32_t)min<uint64_t>((uint64_t)abeg + 2ull * run, np);
    const uint64_t *ak = src_k + t.pbeg + abeg, *bk = src_k + t.pbeg + aend;
    const PT *ap = src_p + t.pbeg + abeg, *bp = src_p + t.pbeg + aend;
    const uint32_t la = aend - abeg, lb = bend - aend;
    const uint32_t d0 = o0 - abeg, d1 = o1 - abeg;
    if (threadIdx.x == 0) s_split[0] = merge_path<PT>(ak, ap, la, bk, bp, lb, d0);
    if (threadIdx.x == 64) s_split[1] = merge_path<PT>(ak, ap, la, bk, bp, lb, d1);
    __syncthreads();
    const uint32_t a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
    const uint32_t na = a1 - a0, nb = b1 - b0;
    uint64_t *lk = reinterpret_cast<uint64_t *>(smem);
    PT *lp = reinterpret_cast<PT *>(lk + SS_TILE);
    for (uint32_t i = threadIdx.x; i < na; i += SS_THREADS) { lk[i] = ak[a0 + i]; lp[i] = ap[a0 + i]; }
    for (uint32_t i = threadIdx.x; i < nb; i += SS_THREADS) { lk[na + i] = bk[b0 + i]; lp[na + i] = bp[b0 + i]; }
    __syncthreads();
    const uint32_t nout = na + nb;
    const uint32_t e0 = min((uint32_t)threadIdx.x * SS_EPT, nout), e1 = min(e0 + SS_EPT, nout);
    uint32_t i = merge_path<PT>(lk, lp, na, lk + na, lp + na, nb, e0), j = e0 - i;
    uint64_t *ok = dst_k + t.pbeg + o0;
    PT *op = dst_p + t.pbeg + o0;
    for (uint32_t e = e0; e < e1; e++) {
        const bool take_a = j >= nb || (i < na && pair_le<PT>(lk[i], lp[i], lk[na + j], lp[na + j]));
        const uint32_t s = take_a ? i : na + j;
        ok[e] = lk[s]; op[e] = lp[s];
        if (take_a) i++; else j++;
    }
}

template <typename PT>
__global__ __launch_bounds__(256) void copy_back_kernel(const SortTask *tasks, const uint32_t *counters,
                                                        const uint64_t *src_k, const PT *src_p, uint64_t *dst_k, PT *dst_p) {
    if (blockIdx.x >= counters[0]) return;
    const SortTask t = tasks[blockIdx.x];
    if (!t.multi) return;
    const uint32_t beg = t.pbeg + t.tile * SS_TILE, end = min(beg + SS_TILE, t.pend);
    for (uint32_t i = beg + threadIdx.x; i < end; i += 256) { dst_k[i] = src_k[i]; dst_p[i] = src_p[i]; }
}

size_t segsort_workspace_bytes(int64_t n_rows, uint32_t n_parts, size_t pay_bytes) {
    const size_t max_tasks = (size_t)n_rows / SS_TILE + n_parts + 8;
    return Arena::padded(max_tasks * sizeof(SortTask)) + Arena::padded(256)
         + Arena::padded(size_t(n_rows + 1) * 8) + Arena::padded(size_t(n_rows + 1) * pay_bytes) + 4096;
}

// Sorts partitions [0, n_parts) of (keys, pay), described by `offsets` (PartInfo layout), in place.
// Workspace comes from c->work (not reset here).  Synchronises the stream once (after the task list is built).
template <typename PT>
static int32_t segsort_impl(pandrs_hip_ctx *c, uint64_t *keys, PT *pay, const uint32_t *offsets, uint32_t NB,
                            uint32_t n_parts, int64_t n_rows, int enc, const uint8_t *only = nullptr, SortTiles *tiles = nullptr) {
    if (n_rows <= 0 || n_parts == 0) return 0;
    const uint32_t max_tasks = (uint32_t)((size_t)n_rows / SS_TILE + n_parts + 8);
    SortTask *tasks = c->work.take<SortTask>(max_tasks);
    uint32_t *counters = c->work.take<uint32_t>(64);
    if (!tasks || !counters) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (segmented sort)");
    if (tiles) { tiles->tasks = tasks; tiles->counters = counters; tiles->max_tasks = max_tasks; }
    hipLaunchKernelGGL(build_sort_tasks_kernel, dim3(1), dim3(SS_THREADS), 0, c->stream, offsets, NB, n_parts, only, tasks, max_tasks, counters);
    const size_t lds = (size_t)SS_TILE * (8 + sizeof(PT)) + 64;
    // the task count first (the one host sync of this function): grids are sized exactly, and a call with
    // nothing to sort (every partition filtered out) launches nothing more
    uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
    HIP_TRY(hipMemcpyAsync(h, counters, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t n_tasks = h[0], max_tiles = h[1];
    if (tiles) tiles->max_tasks = n_tasks;
    if (n_tasks == 0) return 0;
    auto launch_chunk = [&](auto kernel) -> int32_t {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(n_tasks), dim3(SS_THREADS), lds, c->stream, tasks, counters, keys, pay);
        return 0;
    };
    if (enc == 1) ST_TRY(launch_chunk(chunk_sort_kernel<PT, 1>));
    else if (enc == 2) ST_TRY(launch_chunk(chunk_sort_kernel<PT, 2>));
    else ST_TRY(launch_chunk(chunk_sort_kernel<PT, 0>));
    HIP_TRY(hipGetLastError());
    if (max_tiles <= 1) return 0;
    uint64_t *tk = c->work.take<uint64_t>(n_rows + 1);
    PT *tp = c->work.take<PT>(n_rows + 1);
    if (!tk || !tp) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (segmented sort scratch)");
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(merge_pass_kernel<PT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint64_t *sk = keys, *dk = tk;
    PT *sp = pay, *dp = tp;
    for (uint64_t run = SS_TILE; run < (uint64_t)max_tiles * SS_TILE; run <<= 1) {
        hipLaunchKernelGGL(merge_pass_kernel<PT>, dim3(n_tasks), dim3(SS_THREADS), lds, c->stream, tasks, counters, sk, sp, dk, dp, (uint32_t)run);
        std::swap(sk, dk); std::swap(sp, dp);
    }
    if (sk != keys)
        hipLaunchKernelGGL(copy_back_kernel<PT>, dim3(n_tasks), dim3(256), 0, c->stream, tasks, counters, sk, sp, keys, pay);
    HIP_TRY(hipGetLastError());
    return 0;
}

int32_t segmented_sort_u32(pandrs_hip_ctx *c, uint64_t *keys, uint32_t *pay, const uint32_t *offsets, uint32_t NB,
                           uint32_t n_parts, int64_t n_rows, const uint8_t *only) {
    return segsort_impl<uint32_t>(c, keys, pay, offsets, NB, n_parts, n_rows, 0, only);
}
int32_t segmented_sort_u64(pandrs_hip_ctx *c, uint64_t *keys, uint64_t *pay, const uint32_t *offsets, uint32_t NB,
                           uint32_t n_parts, int64_t n_rows, int enc, const uint8_t *only, SortTiles *tiles) {
    return segsort_impl<uint64_t>(c, keys, pay, offsets, NB, n_parts, n_rows, enc, only, tiles);
}

}  // namespace pandrs
