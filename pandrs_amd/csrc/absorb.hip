// absorb.hip — hot-key absorb-and-spill in front of the radix path (gfx950, wave64).
//
// Skewed key columns (the reference's own benches draw 80 % of the rows from 20 % of the keys,
// benches/enhanced_comprehensive_benchmark.rs:53-59; BASELINE config 3) put most rows on a few keys.  Moving every
// row through the radix partition (read + write + read: 3 x the input over the fabric) to fold it into a group that
// could have stayed on the chip is the waste this pass removes:
//
//   absorb_kernel      ONE pass over the ORIGINAL columns.  Every workgroup (one per CU) owns a contiguous row range
//                      and an LDS table that takes the first keys it sees — under skew the hot ones — with a probe of
//                      at most two 4-key buckets.  A row whose key finds neither itself nor a free slot there is
//                      SPILLED straight into its radix partition: region (workgroup w, partition p) of the spill
//                      columns, appended at an LDS cursor (no global atomics; a workgroup's few open lines complete
//                      in its XCD's L2).  At the end the table's groups leave as partial records, like the direct path's.
//   build_spill_tables_kernel  the work list of the lean aggregate (aggregate2.hip) over those regions: one LDS table
//                      per (partition, group of consecutive workgroups), fed by the group's non-empty regions.
//   aggregate2 then folds the spilled rows — no estimate, no histogram, no scatter, no host round trip in between — and
//   appends ITS groups as partial records behind the absorbed ones (same counter); run_engine (groupby.hip) merges once.
//
// A key may be absorbed by one workgroup and spilled by another (or by the same one after losing a CAS race): the
// final merge adds the partial states of equal keys whatever produced them, so only the totals matter.
// Semantics are the aggregate kernels' (aggregation.rs:500-754): NULL key = its own group, NaN ignored by min / max,
// null values skipped, group size counts every row.
#include "aggregate.hpp"

namespace pandrs {

constexpr int AB_THREADS = 1024;
constexpr uint32_t AB_NONE = 0xFFFFFFFFu;

// One bucket of four keys: the slot of k, a slot claimed for k, or AB_NONE.  No loop: at most two claims are tried
// (a lost race for the first free slot moves on to the next free one the snapshot showed).
__device__ __forceinline__ uint32_t absorb_try_bucket(uint64_t k, uint64_t *keys, uint32_t bk, bool claim = true) {
    const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(keys + 4 * bk);
    const ulonglong2 lo = bp[0], hi = bp[1];
    const uint64_t c4[4] = {lo.x, lo.y, hi.x, hi.y};
    int hit = -1, e0 = -1, e1 = -1;
#pragma unroll
    for (int q = 3; q >= 0; q--) {
        if (c4[q] == k) hit = q;
        if (c4[q] == EMPTY_KEY) { e1 = e0; e0 = q; }
    }
    if (hit >= 0) return 4 * bk + hit;
    if (claim && e0 >= 0) {
        const uint64_t old = atomicCAS((unsigned long long *)&keys[4 * bk + e0], EMPTY_KEY, k);
        if (old == EMPTY_KEY || old == k) return 4 * bk + e0;
        if (e1 >= 0) {
            const uint64_t old1 = atomicCAS((unsigned long long *)&keys[4 * bk + e1], EMPTY_KEY, k);
            if (old1 == EMPTY_KEY || old1 == k) return 4 * bk + e1;
        }
    }
    return AB_NONE;
}

// PROFILE: bit0 the sources carry null bitmaps, bits1-3 ops present (add, min, max), bit4 kind (0 f64, 1 i64) —
// the uniform profiles of aggregate_kernel.  LDS: keys[T1] u64 | states[S][T1] u64 | gsz[T1] u32 | misc[32] u32,
// T1 = T + 2 (slot T: the key equal to the table sentinel, slot T + 1: the NULL key; both always absorbed).
template <int NSRC, int PROFILE>
__global__ __launch_bounds__(AB_THREADS) void absorb_kernel(AbsorbArgs a) {
    constexpr bool HAS_V = (PROFILE & 1) != 0, OP_ADD = ((PROFILE >> 1) & 1) != 0;
    constexpr bool OP_MIN = ((PROFILE >> 2) & 1) != 0, OP_MAX = ((PROFILE >> 3) & 1) != 0;
    constexpr int KIND = (PROFILE >> 4) & 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t T = a.T, T1 = T + 2, tid = threadIdx.x;
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *st = keys + T1;
    uint32_t *gsz = reinterpret_cast<uint32_t *>(st + (size_t)a.n_lds_states * T1);
    uint32_t *misc = gsz + ((T1 + 3) & ~3u);
    // misc[0..16] scan scratch, [20] a spill region overflowed, [21] sentinel key seen, [22] output base, [23] NULL key seen,
    // [32 + p] spill cursor of partition p (this workgroup's regions)
    const uint32_t w = blockIdx.x;
    const uint32_t beg = min(w * a.chunk, a.n_rows), end = min(beg + a.chunk, a.n_rows);
    const uint32_t PS = a.spill_P, cap_wp = a.spill_cap;

    for (uint32_t s = tid; s < T1; s += AB_THREADS) { keys[s] = (a.hot_image && s < T) ? a.hot_image[s] : EMPTY_KEY; gsz[s] = 0; }
    for (int k = 0; k < a.n_lds_states; k++) {
        const uint64_t idv = state_identity(a.lds_kind[k]);
        uint64_t *dst = st + (size_t)k * T1;
        for (uint32_t s = tid; s < T1; s += AB_THREADS) dst[s] = idv;
    }
    if (tid < 32 + 64) misc[tid] = 0;
    __syncthreads();

    const uint32_t NBK = T >> 2;
    const bool claim = a.image_only == 0;
    auto fetch = [&](uint32_t i0, uint64_t (&k2)[2], uint64_t (&v)[2][NSRC], bool (&ok)[2][NSRC], bool (&kn)[2]) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t i = min(i0 + h * AB_THREADS, end - 1);
            kn[h] = key_is_null(a.key, i);
            k2[h] = key_cell(a.key, i);
#pragma unroll
            for (int c = 0; c < NSRC; c++) {
                v[h][c] = __builtin_nontemporal_load(a.vals[c] + i);
                ok[h][c] = HAS_V ? !bit_at(a.null_bits[c], i) : true;
            }
        }
    };
    if (beg < end) {
        uint64_t k2n[2], vn[2][NSRC];
        bool okn[2][NSRC], knn[2];
        fetch(min(beg + tid, end - 1), k2n, vn, okn, knn);
        // every wave runs the same number of iterations (the ballots below need all its lanes)
        const uint32_t n_iter = (end - beg + 2 * AB_THREADS - 1) / (2 * AB_THREADS);
        for (uint32_t it = 0; it < n_iter; it++) {
            const uint32_t i0 = beg + tid + it * 2 * AB_THREADS;
            uint64_t k2[2], v[2][NSRC];
            bool ok[2][NSRC], kn[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                k2[h] = k2n[h]; kn[h] = knn[h];
#pragma unroll
                for (int c = 0; c < NSRC; c++) { v[h][c] = vn[h][c]; ok[h][c] = okn[h][c]; }
            }
            if (it + 1 < n_iter) fetch(min(i0 + 2 * AB_THREADS, end - 1), k2n, vn, okn, knn);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool active = i0 + h * AB_THREADS < end;
                const uint64_t k = k2[h];
                uint32_t slot = AB_NONE;
                if (active) {
                    if (kn[h]) { slot = T + 1; misc[23] = 1; }
                    else if (k == EMPTY_KEY) { slot = T; misc[21] = 1; }
                    else {
                        const uint32_t b0 = slot_of(hash32(k, a.seed), NBK);
                        slot = absorb_try_bucket(k, keys, b0, claim);
                        if (slot == AB_NONE) slot = absorb_try_bucket(k, keys, b0 + 1 == NBK ? 0 : b0 + 1, claim);
                    }
                }
                // many lanes of the wave in ONE slot (a dominant key: 64 lanes adding to the same LDS words serialise): the lanes in the
                // first placed lane's slot fold on the VALU and one lane applies the totals (as aggregate2's wave_fold), the others go on below
                bool folded = false;
                if (a.fold) {                                  // (uniform; off for ordinary inputs: the check alone cost C3 3 %)
                    const unsigned long long placed = __ballot(slot != AB_NONE);
                    if (placed) {                              // wave-uniform
                        // (v_readlane with a scalar lane index: __shfl would be a ds_bpermute, i.e. one more op on the LDS pipe this kernel is bound by)
                        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)slot, __builtin_amdgcn_readfirstlane(__ffsll((long long)placed) - 1));
                        const bool same = slot == s0;
                        const unsigned long long samew = __ballot(same);
                        if (__popcll(samew) >= 40) {
                            const bool l0 = (tid & 63) == 0;
                            if (l0) atomicAdd(&gsz[s0], (uint32_t)__popcll(samew));
#pragma unroll
                            for (int c = 0; c < NSRC; c++) {
                                const bool valid = same && ok[h][c];
                                if (HAS_V && a.st_nn[c] >= 0) {
                                    const unsigned long long vm = __ballot(valid);
                                    if (l0 && vm) atomicAdd((unsigned long long *)&st[(size_t)a.st_nn[c] * T1 + s0], (unsigned long long)__popcll(vm));
                                }
                                if (OP_ADD) {
                                    const uint64_t tot = wave_reduce64<KIND == 0 ? 0 : 1>(valid ? v[h][c] : 0ull, 0ull);
                                    if (l0) {
                                        if (KIND == 0) atomicAdd(reinterpret_cast<double *>(&st[(size_t)a.st_add[c] * T1 + s0]), __longlong_as_double((long long)tot));
                                        else atomicAdd((unsigned long long *)&st[(size_t)a.st_add[c] * T1 + s0], (unsigned long long)tot);
                                    }
                                }
                                if (OP_MIN || OP_MAX) {
                                    bool cmp = valid;
                                    if (KIND == 0) { const double dv = __longlong_as_double((long long)v[h][c]); cmp = cmp && dv == dv; }
                                    const uint64_t e = KIND == 0 ? enc_f64(__longlong_as_double((long long)v[h][c])) : enc_i64((int64_t)v[h][c]);
                                    if (OP_MIN) {
                                        const uint64_t mn = wave_reduce64<2>(cmp ? e : ~0ull, ~0ull);
                                        if (l0 && mn != ~0ull) atomicMin((unsigned long long *)&st[(size_t)a.st_min[c] * T1 + s0], (unsigned long long)mn);
                                    }
                                    if (OP_MAX) {
                                        const uint64_t mx = wave_reduce64<3>(cmp ? e : 0ull, 0ull);
                                        if (l0 && mx != 0ull) atomicMax((unsigned long long *)&st[(size_t)a.st_max[c] * T1 + s0], (unsigned long long)mx);
                                    }
                                }
                            }
                            folded = same;
                        }
                    }
                }
                if (slot != AB_NONE && !folded) {
                    atomicAdd(&gsz[slot], 1u);
                    uint64_t enc[NSRC], cur_mn[NSRC], cur_mx[NSRC];
#pragma unroll
                    for (int c = 0; c < NSRC; c++) {
                        enc[c] = KIND == 0 ? enc_f64(__longlong_as_double((long long)v[h][c])) : enc_i64((int64_t)v[h][c]);
                        cur_mn[c] = OP_MIN ? st[(size_t)a.st_min[c] * T1 + slot] : 0ull;
                        cur_mx[c] = OP_MAX ? st[(size_t)a.st_max[c] * T1 + slot] : ~0ull;
                    }
#pragma unroll
                    for (int c = 0; c < NSRC; c++) {
                        if (!ok[h][c]) continue;
                        const uint64_t x = v[h][c];
                        if (HAS_V && a.st_nn[c] >= 0) atomicAdd((unsigned long long *)&st[(size_t)a.st_nn[c] * T1 + slot], 1ull);
                        bool cmp = true;
                        if (KIND == 0) {
                            const double d = __longlong_as_double((long long)x);
                            if (OP_ADD) atomicAdd(reinterpret_cast<double *>(&st[(size_t)a.st_add[c] * T1 + slot]), d);
                            cmp = d == d;          // Rust f64::min / max ignore NaN operands (aggregation.rs:653, :666)
                        } else if (OP_ADD) {
                            atomicAdd((unsigned long long *)&st[(size_t)a.st_add[c] * T1 + slot], x);
                        }
                        if (OP_MIN && cmp && enc[c] < cur_mn[c]) atomicMin((unsigned long long *)&st[(size_t)a.st_min[c] * T1 + slot], enc[c]);
                        if (OP_MAX && cmp && enc[c] > cur_mx[c]) atomicMax((unsigned long long *)&st[(size_t)a.st_max[c] * T1 + slot], enc[c]);
                    }
                }
                // ---- spill: straight into the row's radix partition, at this workgroup's cursor for it ----
                const bool spill = active && slot == AB_NONE;
                if (PS == 1) {
                    // one region per workgroup (the compact spill of a long tail: compact_spill_kernel closes the gaps afterwards):
                    // one LDS atomic per wave and batch instead of one per spilled lane on the same word
                    const unsigned long long sm = __ballot(spill);
                    if (sm) {                                  // wave-uniform
                        uint32_t base = 0;
                        if ((tid & 63) == 0) base = atomicAdd(&misc[32], (uint32_t)__popcll(sm));
                        base = __shfl(base, 0, 64);
                        const uint32_t at = base + (uint32_t)__popcll(sm & ((1ull << (tid & 63)) - 1ull));
                        if (spill) {
                            if (at < cap_wp) {
                                const size_t pos = (size_t)w * cap_wp + at;
                                a.sp_keys[pos] = k;
#pragma unroll
                                for (int c = 0; c < NSRC; c++) {
                                    a.sp_vals[c][pos] = v[h][c];
                                    if (HAS_V) a.sp_valid[c][pos] = ok[h][c] ? 1 : 0;
                                }
                            } else misc[20] = 1;
                        }
                    }
                } else if (spill) {
                    const uint32_t p = part_of(hash32(k, a.seed), PS);
                    const uint32_t at = atomicAdd(&misc[32 + p], 1u);
                    if (at < cap_wp) {
                        const size_t pos = ((size_t)w * PS + p) * cap_wp + at;
                        a.sp_keys[pos] = k;
#pragma unroll
                        for (int c = 0; c < NSRC; c++) {
                            a.sp_vals[c][pos] = v[h][c];
                            if (HAS_V) a.sp_valid[c][pos] = ok[h][c] ? 1 : 0;
                        }
                    } else misc[20] = 1;
                }
            }
        }
    }
    __syncthreads();
    // ---- the table's groups leave as partial records (keys, null flag, group size, states in ABI order) ----
    const bool sentinel = misc[21] != 0, nullseen = misc[23] != 0;
    auto occupied = [&](uint32_t s) { return s < T ? keys[s] != EMPTY_KEY : (s == T ? sentinel : nullseen); };
    for (uint32_t p = tid; p < PS; p += AB_THREADS) a.sp_count[w * PS + p] = min(misc[32 + p], cap_wp);
    uint32_t mine = 0;
    for (uint32_t s = tid; s < T1; s += AB_THREADS) mine += occupied(s) ? 1u : 0u;
    uint32_t total;
    block_exclusive_scan<AB_THREADS>(mine, misc, &total);
    if (tid == 0) {
        if (misc[20]) a.counters[1] = 1;
        misc[22] = atomicAdd(&a.counters[2], total);          // the counter aggregate2 appends its partial records at
        uint32_t spilled = 0;
        for (uint32_t p = 0; p < PS; p++) spilled += min(misc[32 + p], cap_wp);
        if (spilled) atomicAdd(&a.counters[3], spilled);
    }
    __syncthreads();
    uint32_t run = misc[22];
    __syncthreads();
    for (uint32_t sbase = 0; sbase < T1; sbase += AB_THREADS) {
        const uint32_t s = sbase + tid;
        const bool occ = s < T1 && occupied(s);
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan<AB_THREADS>(occ ? 1u : 0u, misc, &tot);
        if (occ) {
            const size_t pos = run + ex;
            const bool isnull = s == T + 1;
            a.out_keys[pos] = isnull ? 0ull : (s < T ? keys[s] : EMPTY_KEY);
            a.out_null[pos] = isnull ? 1 : 0;
            a.out_states[pos] = gsz[s];
            for (int k = 0; k < a.n_lds_states; k++)
                a.out_states[(size_t)(a.lds_abs[k] + 1) * a.cap + pos] = state_natural(a.lds_kind[k], st[(size_t)k * T1 + s]);
        }
        run += tot;
    }
}

// Compact spill: the workgroups' regions (region w = rows [w * cap_wp, w * cap_wp + sp_count[w])) copied back to back — the input of
// an ordinary engine run over the rows the absorb tables did not take.  blockIdx.y = workgroup region, blockIdx.x strides over its rows.
struct CompactArgs {
    const uint32_t *sp_count; uint32_t n_wg, cap_wp; int n_src, has_v;
    const uint64_t *src_keys; const uint64_t *src_vals[MAX_ABS_SRC]; const uint8_t *src_valid[MAX_ABS_SRC];
    uint64_t *dst_keys; uint64_t *dst_vals[MAX_ABS_SRC]; uint8_t *dst_valid[MAX_ABS_SRC];
    uint32_t *total;                                 // [0] = rows in all
};
__global__ __launch_bounds__(256) void compact_spill_kernel(CompactArgs a) {
    __shared__ uint32_t s_off;
    const uint32_t w = blockIdx.y;
    if (threadIdx.x < 64) {                          // rows of the regions before w (n_wg <= 1024: 16 per lane)
        uint32_t part = 0;
        for (uint32_t r = threadIdx.x; r < w; r += 64) part += a.sp_count[r];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
        if (threadIdx.x == 0) s_off = part;
    }
    __syncthreads();
    const uint32_t n = a.sp_count[w], off = s_off;
    if (w == a.n_wg - 1 && blockIdx.x == 0 && threadIdx.x == 0) a.total[0] = off + n;
    const size_t base = (size_t)w * a.cap_wp;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        a.dst_keys[off + i] = a.src_keys[base + i];
        for (int c = 0; c < a.n_src; c++) {
            a.dst_vals[c][off + i] = a.src_vals[c][base + i];
            if (a.has_v) a.dst_valid[c][off + i] = a.src_valid[c][base + i];
        }
    }
}
void launch_compact_spill(pandrs_hip_ctx *c, const AbsorbArgs &a, uint32_t n_wg, uint64_t *dst_keys, uint64_t *const *dst_vals, uint8_t *const *dst_valid,
                          bool has_v, uint32_t *total) {
    CompactArgs ca{};
    ca.sp_count = a.sp_count; ca.n_wg = n_wg; ca.cap_wp = a.spill_cap; ca.n_src = a.n_src; ca.has_v = has_v ? 1 : 0;
    ca.src_keys = a.sp_keys; ca.dst_keys = dst_keys; ca.total = total;
    for (int s2 = 0; s2 < a.n_src; s2++) { ca.src_vals[s2] = a.sp_vals[s2]; ca.src_valid[s2] = a.sp_valid[s2]; ca.dst_vals[s2] = dst_vals[s2]; ca.dst_valid[s2] = dst_valid[s2]; }
    hipLaunchKernelGGL(compact_spill_kernel, dim3(8, n_wg), dim3(256), 0, c->stream, ca);
}

// The lean aggregate's work list over the spill regions: table (p, j) = partition p, workgroups [j * wpt, (j + 1) * wpt);
// its tasks = those workgroups' non-empty regions (p).  Every table is `multi`: its groups leave as partial records.
// One workgroup; counts[0] = tasks, counts[1] = tables.
__global__ __launch_bounds__(1024) void build_spill_tables_kernel(const uint32_t *sp_count, uint32_t n_wg, uint32_t PS, uint32_t cap_wp,
                                                                  uint32_t wpt, AggTask *tasks, AggTable *tables, uint32_t *counts) {
    __shared__ uint32_t wt[17];
    const uint32_t tpp = (n_wg + wpt - 1) / wpt, n_cand = PS * tpp, t = threadIdx.x;     // n_cand <= 1024 (host)
    uint32_t n_task = 0;
    const uint32_t p = t / tpp, j = t % tpp;
    if (t < n_cand)
        for (uint32_t w = j * wpt; w < min((j + 1) * wpt, n_wg); w++) n_task += sp_count[w * PS + p] ? 1u : 0u;
    uint32_t tot_task, tot_tab;
    const uint32_t ex_task = block_exclusive_scan<1024>(n_task, wt, &tot_task);
    const uint32_t ex_tab = block_exclusive_scan<1024>(n_task ? 1u : 0u, wt, &tot_tab);
    if (n_task) {
        uint32_t ti = ex_task;
        for (uint32_t w = j * wpt; w < min((j + 1) * wpt, n_wg); w++) {
            const uint32_t n = sp_count[w * PS + p];
            if (!n) continue;
            const uint32_t base = (w * PS + p) * cap_wp;
            tasks[ti++] = AggTask{p, base, base + n, 1u};
        }
        tables[ex_tab] = AggTable{ex_task, n_task, p, 1u};
    }
    if (t == 0) { counts[0] = tot_task; counts[1] = tot_tab; }
}

template <int NSRC>
static bool launch_absorb_profile(pandrs_hip_ctx *c, const AbsorbArgs &a, int profile, size_t lds, uint32_t grid) {
#define ABSORB_CASE(P)                                                                                                        \
    case P:                                                                                                                   \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(absorb_kernel<NSRC, P>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess) return false;                                                        \
        hipLaunchKernelGGL((absorb_kernel<NSRC, P>), dim3(grid), dim3(AB_THREADS), lds, c->stream, a);                        \
        return true;
    switch (profile) {
        ABSORB_CASE(2) ABSORB_CASE(3)                 // f64 sum (+ null masks)
        ABSORB_CASE(14) ABSORB_CASE(15)               // f64 sum + min + max
        ABSORB_CASE(12) ABSORB_CASE(13)               // f64 min + max
        ABSORB_CASE(18) ABSORB_CASE(19)               // i64 sum
        ABSORB_CASE(30) ABSORB_CASE(31)               // i64 sum + min + max
    default: return false;
    }
#undef ABSORB_CASE
}

bool absorb_has(int n_src, int profile) {
    if (n_src < 1 || n_src > 4) return false;
    switch (profile) { case 2: case 3: case 14: case 15: case 12: case 13: case 18: case 19: case 30: case 31: return true; default: return false; }
}

bool launch_absorb(pandrs_hip_ctx *c, const AbsorbArgs &a, int n_src, int profile, size_t lds, uint32_t grid) {
    switch (n_src) {
    case 1: return launch_absorb_profile<1>(c, a, profile, lds, grid);
    case 2: return launch_absorb_profile<2>(c, a, profile, lds, grid);
    case 3: return launch_absorb_profile<3>(c, a, profile, lds, grid);
    case 4: return launch_absorb_profile<4>(c, a, profile, lds, grid);
    default: return false;
    }
}

void launch_build_spill_tables(pandrs_hip_ctx *c, const uint32_t *sp_count, uint32_t n_wg, uint32_t PS, uint32_t cap_wp, uint32_t wpt,
                               AggTask *tasks, AggTable *tables, uint32_t *counts) {
    hipLaunchKernelGGL(build_spill_tables_kernel, dim3(1), dim3(1024), 0, c->stream, sp_count, n_wg, PS, cap_wp, wpt, tasks, tables, counts);
}

}  // namespace pandrs
