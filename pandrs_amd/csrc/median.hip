// median.hip — AggregateOp::Median on the device (gfx950, wave64).
//
// Reference: GroupBy::calculate_aggregation, Int64 arm aggregation.rs:585-604 and Float64 arm
// :703-722 — collect the group's non-null values, sort, take the middle (odd count) or the mean of
// the two middles (even count; for Int64 the two are ADDED IN i64, then `as f64 / 2.0`); a group
// without non-null values gives 0.0.
//
// Device plan, one pass per Median column, after the engine has produced the groups:
//   1. rows with a null value are dropped by a stream compaction (only when the column has a mask);
//   2. (key cell, value) pairs are radix-partitioned on the key hash, null keys to their own
//      partition (one group, grouping.rs:74);
//   3. every partition is sorted by (key, order-preserving value code) — segsort.hip, any size;
//   4. a run of equal keys is a group in ascending value order: its median goes into a global
//      open-addressing table key -> median;
//   5. the engine's groups look their median up (miss = no non-null value = 0.0).
// f64 values are ordered by their IEEE total order (the reference's partial_cmp sort leaves the
// position of NaNs unspecified; -0.0 sorts before +0.0, which compare equal anyway).
#include "engine.hpp"

#include <algorithm>
#include <cmath>

namespace pandrs {

constexpr int MC_THREADS = 256, MC_RPT = 8;

// keeps the rows whose value is not null: key cell, key-null byte, raw 8-byte value (order arbitrary)
__global__ __launch_bounds__(MC_THREADS) void compact_valid_kernel(KeyDesc key, const uint64_t *vals, const uint8_t *vnull,
                                                                   int64_t n, uint64_t *out_cell, uint8_t *out_knull,
                                                                   uint64_t *out_val, unsigned long long *cursor) {
    __shared__ uint32_t wt[17];
    __shared__ unsigned long long s_base;
    const int64_t base = (int64_t)blockIdx.x * (MC_THREADS * MC_RPT) + threadIdx.x;
    bool keep[MC_RPT];
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < MC_RPT; r++) {
        const int64_t i = base + (int64_t)r * MC_THREADS;
        keep[r] = i < n && !bit_at(vnull, i);
        mine += keep[r] ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<MC_THREADS>(mine, wt, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
    __syncthreads();
    uint64_t pos = s_base + ex;
#pragma unroll
    for (int r = 0; r < MC_RPT; r++) {
        if (!keep[r]) continue;
        const int64_t i = base + (int64_t)r * MC_THREADS;
        out_cell[pos] = key_cell(key, i);
        out_knull[pos] = key_is_null(key, i) ? 1 : 0;
        out_val[pos] = vals[i];
        pos++;
    }
}

__global__ void fill_range_kernel(uint64_t *a, const uint32_t *beg, const uint32_t *end, uint64_t v) {
    const uint32_t b = *beg, e = *end;
    for (uint32_t i = b + blockIdx.x * blockDim.x + threadIdx.x; i < e; i += gridDim.x * blockDim.x) a[i] = v;
}

struct __attribute__((aligned(16))) MedianEntry {
    uint64_t key;
    double median;
};

// keys / vals: the sorted partitions; rows [0, *null_beg) have non-null keys (equal keys adjacent,
// a key lives in one partition), rows [*null_beg, n) are the NULL-key group (cells zeroed).
// kind 0: vals are enc_f64 codes, 1: enc_i64 codes.
__global__ void median_runs_kernel(const uint64_t *keys, const uint64_t *vals, const uint32_t *null_beg, uint32_t n, int kind,
                                   MedianEntry *table, uint32_t table_mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t nb = *null_beg;
    const bool null_grp = i >= nb;
    const uint32_t seg_beg = null_grp ? nb : 0u, seg_end = null_grp ? n : nb;
    const uint64_t k = keys[i];
    if (i > seg_beg && keys[i - 1] == k) return;
    const uint32_t m = sorted_run_length(keys, i, seg_end), mid = m >> 1;
    double med;
    if (kind == 0) {
        const double hi = dec_f64(vals[i + mid]);
        med = (m & 1) ? hi : (dec_f64(vals[i + mid - 1]) + hi) / 2.0;           // aggregation.rs:715-719
    } else {
        const int64_t hi = dec_i64(vals[i + mid]);
        med = (m & 1) ? (double)hi                                               // aggregation.rs:597-601: the add is in i64
                      : (double)(int64_t)((uint64_t)dec_i64(vals[i + mid - 1]) + (uint64_t)hi) / 2.0;
    }
    if (null_grp) { table[table_mask + 2].median = med; return; }
    if (k == EMPTY_KEY) { table[table_mask + 1].median = med; return; }
    uint32_t slot = hash32(k, 0x2545F491u) & table_mask;
    for (uint32_t probes = 0; probes <= table_mask; probes++) {      // bounded: never spin on a full table
        uint64_t old = atomicCAS((unsigned long long *)&table[slot].key, EMPTY_KEY, k);
        if (old == EMPTY_KEY) { table[slot].median = med; break; }
        slot = (slot + 1) & table_mask;
    }
}

// ---- Nunique (legacy AggFunc::Nunique, src/dataframe/groupby.rs:514-519): after the same sort a group's
// distinct values are the positions whose value differs from the predecessor's (`==` on the decoded
// values: -0.0 == 0.0, NaN != NaN, as Vec::dedup sees them).  Lanes of a wave that share a key form a
// segment (the rows are sorted, so a key's lanes are contiguous); the segment's first lane adds the
// segment's count of new values to the key's table entry — one global atomic per key and wave, not per
// row, so one huge group costs n/64 same-address atomics instead of n.
__global__ __launch_bounds__(256) void nunique_runs_kernel(const uint64_t *keys, const uint64_t *vals, const uint32_t *null_beg,
                                                           uint32_t n, int kind, MedianEntry *table, uint32_t table_mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool valid = i < n;
    const uint32_t nb = *null_beg;
    const uint64_t k = valid ? keys[i] : 0ull;
    const bool null_grp = valid && i >= nb;
    const bool run_head = valid && (i == 0 || i == nb || keys[i - 1] != k);
    bool newv = run_head;
    if (valid && !run_head) {
        const uint64_t a = vals[i - 1], b = vals[i];
        newv = kind == 0 ? dec_f64(a) != dec_f64(b) : a != b;
    }
    const bool seg_head = valid && (lane == 0 || run_head);
    const unsigned long long H = __ballot(seg_head), NV = __ballot(newv);
    if (!seg_head) return;
    const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1)) << (lane + 1);   // heads after this lane
    const unsigned long long seg = (above ? ((1ull << __builtin_ctzll(above)) - 1) : ~0ull) & ~((1ull << lane) - 1);
    const unsigned long long cnt = (unsigned long long)__builtin_popcountll(NV & seg);
    if (cnt == 0) return;
    unsigned long long *counter = nullptr;
    if (null_grp) counter = reinterpret_cast<unsigned long long *>(&table[table_mask + 2].median);
    else if (k == EMPTY_KEY) counter = reinterpret_cast<unsigned long long *>(&table[table_mask + 1].median);
    else {
        uint32_t slot = hash32(k, 0x2545F491u) & table_mask;
        for (uint32_t probes = 0; probes <= table_mask; probes++) {      // bounded: never spin on a full table
            const uint64_t old = atomicCAS((unsigned long long *)&table[slot].key, EMPTY_KEY, k);
            if (old == EMPTY_KEY || old == k) { counter = reinterpret_cast<unsigned long long *>(&table[slot].median); break; }
            slot = (slot + 1) & table_mask;
        }
    }
    if (counter) atomicAdd(counter, cnt);
}

__global__ void clear_table_values_kernel(MedianEntry *table, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) table[i].median = 0.0;        // all-zero bits: also the u64 counter 0
}

// mode 0: the entry holds the median; 1: a u64 count of distinct values in the same 8 bytes
__global__ void median_lookup_kernel(const uint64_t *gkeys, const uint8_t *gnull, int64_t n_groups,
                                     const MedianEntry *table, uint32_t table_mask, double *out, int mode) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_groups) return;
    const uint64_t k = gkeys[j];
    double med = 0.0;                                   // no non-null value in the group (aggregation.rs:592, :710)
    if (gnull[j]) med = table[table_mask + 2].median;
    else if (k == EMPTY_KEY) med = table[table_mask + 1].median;
    else {
        uint32_t slot = hash32(k, 0x2545F491u) & table_mask;
        for (uint32_t probes = 0; probes <= table_mask; probes++) {
            const MedianEntry e = table[slot];
            if (e.key == k) { med = e.median; break; }
            if (e.key == EMPTY_KEY) break;
            slot = (slot + 1) & table_mask;
        }
    }
    out[j] = mode ? (double)(unsigned long long)__double_as_longlong(med) : med;
}

// Fills aggregate `fin_index` of the retained groupby result (c->gb) with the groups' medians of
// one value column.  `key` is the engine's key source (original column or packed cells), `kind`
// 0 = f64, 1 = i64.  Uses c->work from scratch (the engine is done with it).
int32_t median_pass(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, const void *vdata, const uint8_t *vnull,
                    int kind, int fin_index, int mode) {
    GroupbyResult &res = c->gb;
    const int64_t G = res.n_groups;
    if (G <= 0) return 0;
    if (n_rows >= (int64_t(1) << 32) - 16384)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "median: more than 2^32 rows per call");
    PhaseTimer pt(c, PANDRS_HIP_PHASE_OTHER);
    c->quiet++;
    struct Unquiet { pandrs_hip_ctx *c; ~Unquiet() { c->quiet--; } } unq{c};
    uint32_t cap_tab = 64;
    while ((double)cap_tab < 1.5 * (double)G) cap_tab <<= 1;
    const size_t ws = engine_workspace_bytes(n_rows, 4, 1) + segsort_workspace_bytes(n_rows, P_MAX + 2, 8)
                    + Arena::padded(size_t(cap_tab + 4) * 16) + (1 << 20);
    ST_TRY(c->work.ensure(ws, c->stream));
    uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
    KeyDesc kd = key;
    const uint64_t *vals = reinterpret_cast<const uint64_t *>(vdata);
    int64_t nv = n_rows;
    if (vnull && n_rows > 0) {
        uint64_t *cc = c->work.take<uint64_t>(n_rows + 1), *cv = c->work.take<uint64_t>(n_rows + 1);
        uint8_t *cn = c->work.take<uint8_t>(n_rows + 16);
        unsigned long long *cur = c->work.take<unsigned long long>(8);
        if (!cc || !cv || !cn || !cur) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (median)");
        HIP_TRY(hipMemsetAsync(cur, 0, 64, c->stream));
        hipLaunchKernelGGL(compact_valid_kernel, dim3((unsigned)((n_rows + MC_THREADS * MC_RPT - 1) / (MC_THREADS * MC_RPT))),
                           dim3(MC_THREADS), 0, c->stream, key, vals, vnull, n_rows, cc, cn, cv, cur);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, cur, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        nv = (int64_t)((uint64_t)h[0] | ((uint64_t)h[1] << 32));
        kd = KeyDesc{cc, nullptr, cn, DT_CELL};
        vals = cv;
    }
    MedianEntry *table = c->work.take<MedianEntry>((size_t)cap_tab + 4);
    if (!table) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (median)");
    HIP_TRY(hipMemsetAsync(table, 0xFF, size_t(cap_tab) * 16, c->stream));
    HIP_TRY(hipMemsetAsync(&table[cap_tab], 0, 64, c->stream));    // [cap] the key ~0's entry, [cap+1] the NULL group's: 0.0 until a run fills them
    if (mode == 1) {                                               // counters start at 0 (the memset left all-ones)
        hipLaunchKernelGGL(clear_table_values_kernel, dim3((cap_tab + 255) / 256), dim3(256), 0, c->stream, table, cap_tab);
        HIP_TRY(hipGetLastError());
    }
    if (nv > 0) {
        uint64_t *pk = c->work.take<uint64_t>(nv + 1), *pv = c->work.take<uint64_t>(nv + 1);
        if (!pk || !pv) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (median)");
        int64_t P = std::max<int64_t>(1, (int64_t)std::ceil((double)nv / 4900.0));
        P = std::min<int64_t>(P, P_MAX);
        PartInfo part{};
        ScatterArgs sa{};
        sa.key = kd; sa.pkeys = pk; sa.n_rows = nv; sa.P = (uint32_t)P; sa.seed = 0x3C6EF372u;
        sa.mv[sa.n_move++] = MoveDesc{vals, pv, 0, 0};
        ST_TRY(radix_partition(c, sa, &part, PANDRS_HIP_PHASE_OTHER, PANDRS_HIP_PHASE_OTHER, PANDRS_HIP_PHASE_OTHER));
        const uint32_t *null_beg = part.offsets + (size_t)P * part.NB, *null_end = part.offsets + (size_t)(P + 1) * part.NB;
        hipLaunchKernelGGL(fill_range_kernel, dim3(256), dim3(256), 0, c->stream, pk, null_beg, null_end, 0ull);
        ST_TRY(segmented_sort_u64(c, pk, pv, part.offsets, part.NB, (uint32_t)P + 1, nv, kind == 0 ? 1 : 2));
        if (mode == 1)
            hipLaunchKernelGGL(nunique_runs_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, c->stream,
                               pk, pv, null_beg, (uint32_t)nv, kind, table, cap_tab - 1);
        else
            hipLaunchKernelGGL(median_runs_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, c->stream,
                               pk, pv, null_beg, (uint32_t)nv, kind, table, cap_tab - 1);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(median_lookup_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, c->stream,
                       res.keys, res.key_null, G, table, cap_tab - 1, res.aggs + (size_t)fin_index * res.cap, mode);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pandrs
