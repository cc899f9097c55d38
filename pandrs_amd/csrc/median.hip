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
//   3. FAST PATH (group_sort_kernel): a partition that fits LDS is grouped by key with an LDS hash table and
//      every group's values are sorted on their own, one wave per batch of groups — the result goes
//      straight into a global open-addressing table key -> median; partitions that do not fit are
//      flagged and take the GENERAL PATH: sorted by (key, order-preserving value code) — segsort.hip, any
//      size — where a run of equal keys is a group in ascending value order;
//   4. either way the group's median (or distinct count) lands in the global table key -> value;
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

// General path, after the segmented sort.  One workgroup per sorted tile (SortTask): equal keys are adjacent
// inside the tile's partition [pbeg, pend); the NULL-key partition (pbeg == *null_beg, cells zeroed) is one group.
// kind 0: vals are enc_f64 codes, 1: enc_i64 codes.
constexpr int MR_THREADS = 1024;
__global__ __launch_bounds__(MR_THREADS) void median_runs_kernel(const SortTask *tasks, const uint32_t *counters,
                                                                 const uint64_t *keys, const uint64_t *vals, const uint32_t *null_beg,
                                                                 int kind, MedianEntry *table, uint32_t table_mask) {
    if (blockIdx.x >= counters[0]) return;
    const SortTask t = tasks[blockIdx.x];
    const uint32_t tbeg = t.pbeg + t.tile * SS_TILE, tend = min(tbeg + SS_TILE, t.pend);
    const bool null_grp = t.pbeg >= *null_beg;
    for (uint32_t i = tbeg + threadIdx.x; i < tend; i += MR_THREADS) {
        const uint64_t k = keys[i];
        if (i > t.pbeg && keys[i - 1] == k) continue;
        const uint32_t m = sorted_run_length(keys, i, t.pend), mid = m >> 1;
        double med;
        if (kind == 0) {
            const double hi = dec_f64(vals[i + mid]);
            med = (m & 1) ? hi : (dec_f64(vals[i + mid - 1]) + hi) / 2.0;           // aggregation.rs:715-719
        } else {
            const int64_t hi = dec_i64(vals[i + mid]);
            med = (m & 1) ? (double)hi                                               // aggregation.rs:597-601: the add is in i64
                          : (double)(int64_t)((uint64_t)dec_i64(vals[i + mid - 1]) + (uint64_t)hi) / 2.0;
        }
        if (null_grp) { table[table_mask + 2].median = med; continue; }
        if (k == EMPTY_KEY) { table[table_mask + 1].median = med; continue; }
        uint32_t slot = hash32(k, 0x2545F491u) & table_mask;
        for (uint32_t probes = 0; probes <= table_mask; probes++) {      // bounded: never spin on a full table
            uint64_t old = atomicCAS((unsigned long long *)&table[slot].key, EMPTY_KEY, k);
            if (old == EMPTY_KEY) { table[slot].median = med; break; }
            slot = (slot + 1) & table_mask;
        }
    }
}

// ---- Nunique (legacy AggFunc::Nunique, src/dataframe/groupby.rs:514-519): after the same sort a group's
// distinct values are the positions whose value differs from the predecessor's (`==` on the decoded
// values: -0.0 == 0.0, NaN != NaN, as Vec::dedup sees them).  Lanes of a wave that share a key form a
// segment (the rows are sorted, so a key's lanes are contiguous); the segment's first lane adds the
// segment's count of new values to the key's table entry — one global atomic per key and wave, not per
// row, so one huge group costs n/64 same-address atomics instead of n.
__global__ __launch_bounds__(MR_THREADS) void nunique_runs_kernel(const SortTask *tasks, const uint32_t *counters,
                                                                  const uint64_t *keys, const uint64_t *vals, const uint32_t *null_beg,
                                                                  int kind, MedianEntry *table, uint32_t table_mask) {
    if (blockIdx.x >= counters[0]) return;
    const SortTask t = tasks[blockIdx.x];
    const uint32_t tbeg = t.pbeg + t.tile * SS_TILE, tend = min(tbeg + SS_TILE, t.pend);
    const bool null_grp = t.pbeg >= *null_beg;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t base = tbeg; base < tend; base += MR_THREADS) {         // uniform trip count: the ballots need every lane
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < tend;
        const uint64_t k = valid ? keys[i] : 0ull;
        const bool run_head = valid && (i == t.pbeg || keys[i - 1] != k);
        bool newv = run_head;
        if (valid && !run_head) {
            const uint64_t a = vals[i - 1], b = vals[i];
            newv = kind == 0 ? dec_f64(a) != dec_f64(b) : a != b;
        }
        const bool seg_head = valid && (lane == 0 || run_head);
        const unsigned long long H = __ballot(seg_head), NV = __ballot(newv);
        if (!seg_head) continue;
        const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1)) << (lane + 1);   // heads after this lane
        const unsigned long long seg = (above ? ((1ull << __builtin_ctzll(above)) - 1) : ~0ull) & ~((1ull << lane) - 1);
        const unsigned long long cnt = (unsigned long long)__builtin_popcountll(NV & seg);
        if (cnt == 0) continue;
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

// ---- fast path: one workgroup per partition, everything in LDS -----------------------------------
// A partition of <= GS_CAP rows with <= GS_KEYS distinct keys never needs the general sort: its rows
// are grouped by key with an LDS hash table (count per key -> scan -> place the value codes into the
// key's run), every run is sorted on its own — runs are short (a group's share of the rows), so a
// wave sorts a run with a bitonic network over LDS without block barriers; runs of >= GS_BIG values
// are sorted by the whole workgroup — and the median / distinct count goes straight into the global
// key table.  One read of the partitioned pairs, nothing written back.  The network is the
// "always ascending" bitonic variant (first step of a merge mirrors the partner, i ^ (k - 1)), so an
// arbitrary run length needs no padding: a compare-exchange whose upper index is past the end is a no-op.
// Partitions that do not fit (a hot key, a huge NULL group, nearly unique keys) are flagged in
// `only[]` and left to the general path below.
constexpr int GS_THREADS = 1024;
constexpr uint32_t GS_CAP = 15360, GS_SLOTS = 2048, GS_KEYS = 1536, GS_BIG = 1024;
constexpr int GS_RPT = GS_CAP / GS_THREADS;
constexpr uint32_t GS_PROBES = 192;
constexpr size_t gs_lds_bytes(size_t value_bytes) {
    return size_t(GS_CAP) * value_bytes + size_t(GS_SLOTS + 2) * 8 + 2 * size_t(GS_SLOTS + 4) * 4 + 64 * 4 + 32 * 4 + 72 * 4 + (GS_SLOTS + 16) * 2;
}
constexpr size_t GS_LDS = gs_lds_bytes(8);

struct GroupSortArgs {
    const uint64_t *pkeys;           // partitioned key cells
    const void *pvals;               // partitioned payload: raw 8-byte values (reduce modes) or u32 row indices (REORDER)
    uint64_t *out_keys;              // REORDER: the partition rewritten in place, grouped by key (equal keys
    uint32_t *out_rows;              //          adjacent, any order between keys), rows ascending inside a key
    const uint32_t *offsets;
    uint32_t NB, P;
    int kind, mode;                  // kind 0 f64 / 1 i64; mode 0 median / 1 distinct count
    MedianEntry *table;
    uint32_t table_mask;
    uint8_t *only;                   // [P + 1] out: 1 = this partition is left to the general path
};

// compare-exchange c of one step of the ascending bitonic network over v[0, m): distance 1 << lj; `mask` is
// the partner distance as an xor mask — 2j - 1 for the first step of a merge (the mirrored partner), j after
template <typename VT>
__device__ __forceinline__ void gs_cex(VT *v, uint32_t m, uint32_t lj, uint32_t mask, uint32_t c) {
    const uint32_t l = ((c >> lj) << (lj + 1)) | (c & ((1u << lj) - 1)), r = l ^ mask;
    const VT x = v[l], y = v[min(r, m - 1)];
    if (r < m && x > y) { v[l] = y; v[r] = x; }
}

__device__ __forceinline__ void gs_publish(const GroupSortArgs &a, bool null_part, uint64_t key, double value) {
    if (null_part) { a.table[a.table_mask + 2].median = value; return; }
    if (key == EMPTY_KEY) { a.table[a.table_mask + 1].median = value; return; }
    uint32_t slot = hash32(key, 0x2545F491u) & a.table_mask;
    for (uint32_t probes = 0; probes <= a.table_mask; probes++) {        // bounded: never spin on a full table
        const uint64_t old = atomicCAS((unsigned long long *)&a.table[slot].key, EMPTY_KEY, key);
        if (old == EMPTY_KEY) { a.table[slot].median = value; return; }
        slot = (slot + 1) & a.table_mask;
    }
}

// VT = uint64_t, REORDER = false: Median / Nunique as described above.  VT = uint32_t, REORDER = true: the payload
// is the row index; the sorted runs are written back over the partition (key cell of the run, ascending rows) —
// group_by's own result (entries.hip, build_sorted_groups) without the general sort.
template <typename VT, bool REORDER>
__global__ __launch_bounds__(GS_THREADS) void group_sort_kernel(GroupSortArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t p = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint32_t beg = a.offsets[(size_t)p * a.NB], end = a.offsets[(size_t)(p + 1) * a.NB], n = end - beg;
    if (n == 0) return;
    if (n > GS_CAP) { if (tid == 0) a.only[p] = 1; return; }
    const bool null_part = p == a.P;
    VT *lv = reinterpret_cast<VT *>(smem);
    uint64_t *hk = reinterpret_cast<uint64_t *>(lv + GS_CAP);     // [GS_SLOTS + 1]: the last entry is the key ~0's
    uint32_t *hc = reinterpret_cast<uint32_t *>(hk + GS_SLOTS + 2);   // rows per key
    uint32_t *hs = hc + GS_SLOTS + 4;                             // run start, then (after placement) run end
    uint32_t *big = hs + GS_SLOTS + 4;                            // slots of the runs sorted by the whole workgroup
    uint32_t *wt = big + 64;                                      // scan scratch; [20] overflow, [21] distinct keys, [22] big runs
    uint32_t *occ = wt + 32;                                      // one bit per slot: the slot holds a run
    uint16_t *runs = reinterpret_cast<uint16_t *>(occ + 72);      // the occupied slots, any order: wt[23] of them; wt[24] = next to take
    for (uint32_t s = tid; s < GS_SLOTS + 1; s += GS_THREADS) { hk[s] = EMPTY_KEY; hc[s] = 0; }
    if (tid < 32) wt[tid] = 0;
    if (tid < 72) occ[tid] = 0;
    __syncthreads();
    // 1. count the rows of every key; a thread keeps the slots of its rows.  All of the thread's key and
    // value loads are issued up front (one HBM latency per partition, not one per row).
    uint64_t kreg[GS_RPT];
    VT xreg[GS_RPT];
#pragma unroll
    for (int r = 0; r < GS_RPT; r++) {
        const uint32_t i = min((uint32_t)(r * GS_THREADS) + tid, n - 1);
        kreg[r] = a.pkeys[beg + i];
        xreg[r] = reinterpret_cast<const VT *>(a.pvals)[beg + i];
    }
    uint32_t sl[GS_RPT];
#pragma unroll
    for (int r = 0; r < GS_RPT; r++) {
        const uint32_t i = r * GS_THREADS + tid;
        sl[r] = 0;
        if (i >= n) continue;
        uint32_t s = 0;
        if (!null_part && !wt[20]) {
            const uint64_t k = kreg[r];
            s = GS_SLOTS;
            if (k != EMPTY_KEY) {
                s = hash32(k, 0x68E31DA4u) & (GS_SLOTS - 1);
                // (probe chains are short at <= 75 % load; a long one means the partition has too many keys: give
                // up at once — and stop every other thread's search — instead of walking a full table per row)
                uint32_t probes = 0;
                for (; probes < GS_PROBES; probes++) {
                    const uint64_t old = atomicCAS((unsigned long long *)&hk[s], EMPTY_KEY, k);
                    if (old == EMPTY_KEY) { if (atomicAdd(&wt[21], 1u) >= GS_KEYS) wt[20] = 1; break; }
                    if (old == k) break;
                    if ((probes & 7) == 7 && wt[20]) { probes = GS_PROBES; break; }
                    s = (s + 1) & (GS_SLOTS - 1);
                }
                if (probes >= GS_PROBES) { wt[20] = 1; s = 0; }
            }
        }
        sl[r] = s;
        atomicAdd(&hc[s], 1u);
    }
    __syncthreads();
    if (wt[20]) { if (tid == 0) a.only[p] = 1; return; }          // too many distinct keys for the table
    // 2. exclusive scan of the counts = start of every run
    {
        const uint32_t c0 = hc[2 * tid], c1 = hc[2 * tid + 1];
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan<GS_THREADS>(c0 + c1, wt, &tot);
        hs[2 * tid] = ex; hs[2 * tid + 1] = ex + c0;
        if (tid == GS_THREADS - 1) hs[GS_SLOTS] = ex + c0 + c1;
        if (c0) { atomicOr(&occ[(2 * tid) >> 5], 1u << ((2 * tid) & 31)); runs[atomicAdd(&wt[23], 1u)] = (uint16_t)(2 * tid); }
        if (c1) { atomicOr(&occ[(2 * tid + 1) >> 5], 1u << ((2 * tid + 1) & 31)); runs[atomicAdd(&wt[23], 1u)] = (uint16_t)(2 * tid + 1); }
        if (tid == 0 && hc[GS_SLOTS]) { atomicOr(&occ[GS_SLOTS >> 5], 1u << (GS_SLOTS & 31)); runs[atomicAdd(&wt[23], 1u)] = (uint16_t)GS_SLOTS; }
        if (c0 >= GS_BIG) big[atomicAdd(&wt[22], 1u)] = 2 * tid;
        if (c1 >= GS_BIG) big[atomicAdd(&wt[22], 1u)] = 2 * tid + 1;
        if (tid == 0 && hc[GS_SLOTS] >= GS_BIG) big[atomicAdd(&wt[22], 1u)] = GS_SLOTS;
    }
    __syncthreads();
    // 3. place the order-preserving value codes into their key's run
#pragma unroll
    for (int r = 0; r < GS_RPT; r++) {
        const uint32_t i = r * GS_THREADS + tid;
        if (i >= n) continue;
        const VT x = xreg[r];
        const uint32_t pos = atomicAdd(&hs[sl[r]], 1u);
        if constexpr (REORDER) lv[pos] = x;
        else lv[pos] = a.kind == 0 ? enc_f64(__longlong_as_double((long long)x)) : enc_i64((int64_t)x);
    }
    __syncthreads();
    // 4a. the few long runs: the whole workgroup sorts each
    const uint32_t n_big = wt[22];
    for (uint32_t b = 0; b < n_big; b++) {
        const uint32_t s = big[b], m = hc[s];
        VT *v = lv + (hs[s] - m);
        uint32_t lm = 1;
        while ((1u << lm) < m) lm++;
        for (uint32_t lk = 1; lk <= lm; lk++) {
            for (int lj = (int)lk - 1; lj >= 0; lj--) {
                const uint32_t mask = lj + 1 == (int)lk ? (2u << lj) - 1 : 1u << lj;
                for (uint32_t c = tid; c < (1u << (lm - 1)); c += GS_THREADS) gs_cex(v, m, (uint32_t)lj, mask, c);
                __syncthreads();
            }
        }
    }
    // 4b. the waves take batches of GS_ILP runs from the shared list (an LDS ticket, so long and short runs
    // balance out) and sort the short ones — a bitonic network over LDS without workgroup barriers —
    // GS_ILP runs side by side — one step of a run is two dependent LDS round trips and
    // nothing else, so four independent runs in flight are what keeps the wave busy — and leaves each run's
    // result in the slot's (count, cursor) words.  The global table is filled afterwards by all threads at
    // once: a wave publishing run after run would wait for one device-scope atomic round trip per run.
    constexpr int GS_ILP = 4;
    const uint32_t n_runs = wt[23];
    {
        for (;;) {                                        // waves take batches of GS_ILP runs from the shared list
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(&wt[24], (uint32_t)GS_ILP);
            first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
            if (first >= n_runs) break;
            const uint32_t my_slot = first + lane < n_runs && lane < GS_ILP ? (uint32_t)runs[first + lane] : 0u;
            const uint32_t my_m = first + lane < n_runs && lane < GS_ILP ? hc[my_slot] : 0u;
            const uint32_t my_end = my_m ? hs[my_slot] : 0u;
            uint32_t rm[GS_ILP], rbase[GS_ILP], rslot[GS_ILP], lm = 0;
#pragma unroll
            for (int r = 0; r < GS_ILP; r++) {
                rm[r] = (uint32_t)__builtin_amdgcn_readlane((int)my_m, r);
                rbase[r] = rm[r] ? (uint32_t)__builtin_amdgcn_readlane((int)my_end, r) - rm[r] : 0u;
                rslot[r] = (uint32_t)__builtin_amdgcn_readlane((int)my_slot, r);
                if (rm[r] >= 2 && rm[r] < GS_BIG) { uint32_t l2 = 1; while ((1u << l2) < rm[r]) l2++; lm = max(lm, l2); }
            }
            for (uint32_t lk = 1; lk <= lm; lk++) {
                for (int lj = (int)lk - 1; lj >= 0; lj--) {
                    const uint32_t mask = lj + 1 == (int)lk ? (2u << lj) - 1 : 1u << lj;
                    for (uint32_t c = lane; c < (1u << (lm - 1)); c += 64) {
                        const uint32_t l = ((c >> lj) << (lj + 1)) | (c & ((1u << lj) - 1)), rr = l ^ mask;
                        VT x[GS_ILP], y[GS_ILP];
                        bool live[GS_ILP];
#pragma unroll
                        for (int r = 0; r < GS_ILP; r++) {          // all loads first: independent, in flight together
                            live[r] = rr < rm[r] && rm[r] < GS_BIG;  // (a run beyond its own size: the extra steps are no-ops)
                            const uint32_t lc = live[r] ? l : 0u, rc = live[r] ? rr : 0u;
                            x[r] = lv[rbase[r] + lc]; y[r] = lv[rbase[r] + rc];
                        }
#pragma unroll
                        for (int r = 0; r < GS_ILP; r++)
                            if (live[r] && x[r] > y[r]) { lv[rbase[r] + l] = y[r]; lv[rbase[r] + rr] = x[r]; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the wave's own LDS traffic, in order
                    __builtin_amdgcn_wave_barrier();
                }
            }
#pragma unroll
            for (int r = 0; r < GS_ILP; r++) {
                const uint32_t m = rm[r];
                if (m == 0) continue;
                const VT *v = lv + rbase[r];
                if constexpr (REORDER) {
                    const uint64_t key = null_part ? 0ull : (rslot[r] == GS_SLOTS ? EMPTY_KEY : hk[rslot[r]]);
                    for (uint32_t j = lane; j < m; j += 64) {
                        a.out_keys[beg + rbase[r] + j] = key;
                        a.out_rows[beg + rbase[r] + j] = (uint32_t)v[j];
                    }
                    continue;
                }
                double out;
                if (a.mode == 0) {
                    const uint32_t mid = m >> 1;
                    if (a.kind == 0) {
                        const double hi = dec_f64(v[mid]);
                        out = (m & 1) ? hi : (dec_f64(v[mid - 1]) + hi) / 2.0;                      // aggregation.rs:715-719
                    } else {
                        const int64_t hi = dec_i64(v[mid]);
                        out = (m & 1) ? (double)hi                                                  // aggregation.rs:597-601: the add is in i64
                                      : (double)(int64_t)((uint64_t)dec_i64(v[mid - 1]) + (uint64_t)hi) / 2.0;
                    }
                } else {
                    uint32_t cnt = 0;
                    for (uint32_t j = lane; j < m; j += 64) {
                        bool nv = j == 0;
                        if (j > 0) { const uint64_t x = v[j - 1], y = v[j]; nv = a.kind == 0 ? dec_f64(x) != dec_f64(y) : x != y; }
                        cnt += nv ? 1u : 0u;
                    }
                    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
                    out = __longlong_as_double((long long)(unsigned long long)cnt);     // a u64 count in the entry's 8 bytes
                }
                if (lane == 0) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(out);
                    hc[rslot[r]] = (uint32_t)bits; hs[rslot[r]] = (uint32_t)(bits >> 32);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (REORDER) return;
    __syncthreads();
    for (uint32_t s = tid; s < GS_SLOTS + 1; s += GS_THREADS) {
        if (!((occ[s >> 5] >> (s & 31)) & 1u)) continue;
        const double out = __longlong_as_double((long long)((unsigned long long)hc[s] | ((unsigned long long)hs[s] << 32)));
        gs_publish(a, null_part, s == GS_SLOTS ? EMPTY_KEY : hk[s], out);
    }
}

// ---- flagged partitions that hold ONE key (a category of a low-cardinality column, the NULL group): the median
// by radix SELECT instead of a sort.  One workgroup per partition: six passes over the partition's values, each
// a histogram of the next 11 (last: 9) bits of the order-preserving codes that still match the prefix found so
// far, narrow the rank down to one code; for an even count the lower middle is selected the same way.  A wave
// whose lanes all hit one bin (constant or clustered columns) adds its population count once — 64 same-address
// LDS atomics would serialise.  The partition's flag is cleared; partitions with several keys keep it.
constexpr int SK_THREADS = 1024, SK_BINS = 2048;
constexpr uint32_t SK_MAX_ROWS = 1u << 16;
__global__ __launch_bounds__(SK_THREADS) void single_key_median_kernel(GroupSortArgs a) {
    __shared__ uint32_t hist[SK_BINS];
    __shared__ uint32_t wt[20];
    __shared__ uint32_t s_bin, s_less;
    const uint32_t p = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (!a.only[p]) return;
    const uint32_t beg = a.offsets[(size_t)p * a.NB], end = a.offsets[(size_t)(p + 1) * a.NB], n = end - beg;
    // one workgroup streams the partition 7-13 times: only for partitions a single tile of the chip-wide selection
    // below would cover anyway (100 M rows in ONE group measured 707 ms this way, 3.6 ms spread over the chip)
    if (n == 0 || n > SK_MAX_ROWS) return;
    const bool null_part = p == a.P;
    const uint64_t k0 = a.pkeys[beg];
    int diff = 0;
    if (!null_part)
        for (uint32_t i = tid; i < n; i += SK_THREADS) diff |= a.pkeys[beg + i] != k0 ? 1 : 0;
    if (__syncthreads_or(diff)) return;                              // several keys: the general path sorts it
    const uint64_t *vals = reinterpret_cast<const uint64_t *>(a.pvals) + beg;
    auto code_of = [&](uint32_t i) -> uint64_t {
        const uint64_t x = vals[i];
        return a.kind == 0 ? enc_f64(__longlong_as_double((long long)x)) : enc_i64((int64_t)x);
    };
    auto select = [&](uint32_t rank) -> uint64_t {                   // the code of the element of that rank (0-based)
        uint64_t prefix = 0;
        int shift = 64;
        for (int pass = 0; pass < 6; pass++) {
            const int bits = pass < 5 ? 11 : 9;
            shift -= bits;
            for (uint32_t b = tid; b < SK_BINS; b += SK_THREADS) hist[b] = 0;
            __syncthreads();
            for (uint32_t i0 = 0; i0 < n; i0 += SK_THREADS) {        // uniform trip count: the ballots need every lane
                const uint32_t i = i0 + tid;
                bool in = i < n;
                uint32_t bin = 0;
                if (in) {
                    const uint64_t c = code_of(i);
                    in = pass == 0 || (c >> (shift + bits)) == (prefix >> (shift + bits));
                    bin = (uint32_t)(c >> shift) & ((1u << bits) - 1);
                }
                const unsigned long long act = __ballot(in);
                if (!act) continue;
                const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(act));
                if (__ballot(in && bin == first) == act) {
                    if (lane == (uint32_t)__builtin_ctzll(act)) atomicAdd(&hist[first], (uint32_t)__builtin_popcountll(act));
                } else if (in) atomicAdd(&hist[bin], 1u);
            }
            __syncthreads();
            const uint32_t c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
            uint32_t tot;
            const uint32_t ex = block_exclusive_scan<SK_THREADS>(c0 + c1, wt, &tot);
            if (rank >= ex && rank < ex + c0) { s_bin = 2 * tid; s_less = ex; }
            else if (rank >= ex + c0 && rank < ex + c0 + c1) { s_bin = 2 * tid + 1; s_less = ex + c0; }
            __syncthreads();
            prefix |= (uint64_t)s_bin << shift;
            rank -= s_less;
            __syncthreads();
        }
        return prefix;
    };
    const uint32_t mid = n >> 1;
    const uint64_t hi_c = select(mid);
    const uint64_t lo_c = (n & 1) ? hi_c : select(mid - 1);
    if (tid == 0) {
        double med;
        if (a.kind == 0) {
            const double hi = dec_f64(hi_c);
            med = (n & 1) ? hi : (dec_f64(lo_c) + hi) / 2.0;                               // aggregation.rs:715-719
        } else {
            const int64_t hi = dec_i64(hi_c);
            med = (n & 1) ? (double)hi                                                     // aggregation.rs:597-601: the add is in i64
                          : (double)(int64_t)((uint64_t)dec_i64(lo_c) + (uint64_t)hi) / 2.0;
        }
        gs_publish(a, null_part, k0, med);
        a.only[p] = 0;
    }
}

// ---- the same selection for single-key partitions of MORE than SK_MAX_ROWS rows, spread over the chip ---------
// (one group holding a large share of 100 M rows: one workgroup would stream it for 700 ms).  The flagged big
// partitions are cut into tiles of SEL_TILE rows; per pass every tile's workgroup histograms its rows that still
// match the partition's prefix (LDS, then its non-empty bins into the partition's global histogram), and one
// workgroup per partition picks the bin holding the wanted rank.  Six (histogram, pick) launches find the upper
// middle; the lower middle of an even count is the same code when an equal element ranks below it (the final
// rank inside the code's run is >= 1), otherwise the largest code below — one more pass (atomicMax).  No host
// round trip: every kernel is launched for the worst-case grid and exits on the device-side counts.
constexpr uint32_t SEL_TILE = 1u << 16;
struct SelTask { uint32_t pbeg, pend, tile, idx; };
struct SelState {
    unsigned long long prefix, hi, key;
    unsigned long long region, distinct;   // Nunique: first entry of the partition's hash-set region; values counted so far
    uint32_t rank, n, active, multi, part, need_lo, region_mask, pad1;
};
struct SelArgs {
    const uint64_t *pkeys, *pvals;
    const uint32_t *offsets;
    uint32_t NB, P;
    int kind, pass;
    MedianEntry *table;
    uint32_t table_mask;
    uint8_t *only;
    SelTask *tasks;
    uint32_t *counters;                 // [0] tasks, [1] big partitions
    SelState *st;
    uint32_t *hist;                     // [max_big][SK_BINS], zero between passes
    unsigned long long *maxbelow;       // [max_big]
    uint32_t max_tasks, max_big;
    uint32_t min_rows;                  // a flagged partition takes part when it has MORE rows than this
    unsigned long long *sets;           // Nunique: the hash sets of all partitions (EMPTY_KEY-filled), `sets_cap` entries
    unsigned long long sets_cap;
};
__device__ __forceinline__ uint64_t sel_code(const SelArgs &a, uint32_t row) {
    const uint64_t x = a.pvals[row];
    return a.kind == 0 ? enc_f64(__longlong_as_double((long long)x)) : enc_i64((int64_t)x);
}
// one workgroup: the big flagged partitions get a dense index, a state and their tile tasks
__global__ __launch_bounds__(SK_THREADS) void sel_init_kernel(SelArgs a) {
    __shared__ uint32_t wt[17];
    uint32_t task_carry = 0, big_carry = 0, unit_carry = 0;
    for (uint32_t base = 0; base < a.P + 1; base += SK_THREADS) {
        const uint32_t p = base + threadIdx.x;
        uint32_t beg = 0, end = 0;
        bool big = false;
        if (p <= a.P && a.only[p]) {
            beg = a.offsets[(size_t)p * a.NB]; end = a.offsets[(size_t)(p + 1) * a.NB];
            big = end - beg > a.min_rows;
        }
        const uint32_t nt = big ? (end - beg + SEL_TILE - 1) / SEL_TILE : 0u;
        uint32_t tot_t, tot_b;
        const uint32_t ex_t = block_exclusive_scan<SK_THREADS>(nt, wt, &tot_t);
        __syncthreads();
        const uint32_t ex_b = block_exclusive_scan<SK_THREADS>(big ? 1u : 0u, wt, &tot_b);
        __syncthreads();
        // Nunique: a hash-set region of 2^k >= 1.5 n entries (k >= 10), handed out in units of 1024 entries
        uint32_t units = 0;
        if (big && a.sets) { uint64_t sz = 1024; while (sz * 2 < 3ull * (end - beg)) sz <<= 1; units = (uint32_t)(sz >> 10); }
        uint32_t tot_u;
        const uint32_t ex_u = block_exclusive_scan<SK_THREADS>(units, wt, &tot_u);
        __syncthreads();
        const uint32_t idx = big_carry + ex_b;
        if (big && idx < a.max_big) {
            SelState s{};
            s.n = end - beg; s.rank = s.n >> 1; s.active = 1; s.part = p; s.key = a.pkeys[beg];
            s.region = ((unsigned long long)unit_carry + ex_u) << 10; s.region_mask = (units << 10) - 1;
            if (a.sets && (units >= (1u << 21) || s.region + ((unsigned long long)units << 10) > a.sets_cap)) s.multi = 1;   // (a > 2^31-entry set: leave it to the sort)
            a.st[idx] = s;
            for (uint32_t t = 0; t < nt; t++)
                if (task_carry + ex_t + t < a.max_tasks) a.tasks[task_carry + ex_t + t] = SelTask{beg, end, t, idx};
        }
        task_carry += tot_t; big_carry += tot_b; unit_carry += tot_u;
    }
    if (threadIdx.x == 0) { a.counters[0] = min(task_carry, a.max_tasks); a.counters[1] = min(big_carry, a.max_big); }
}
// a tile whose keys are not all the partition's first key marks the partition (the NULL group is one group anyway)
__global__ __launch_bounds__(SK_THREADS) void sel_multi_kernel(SelArgs a) {
    if (blockIdx.x >= a.counters[0]) return;
    const SelTask t = a.tasks[blockIdx.x];
    if (a.st[t.idx].part == a.P) return;
    const uint64_t k0 = a.st[t.idx].key;
    const uint32_t b = t.pbeg + t.tile * SEL_TILE, e = min(b + SEL_TILE, t.pend);
    int diff = 0;
    for (uint32_t i = b + threadIdx.x; i < e; i += SK_THREADS) diff |= a.pkeys[i] != k0 ? 1 : 0;
    if (__syncthreads_or(diff) && threadIdx.x == 0) a.st[t.idx].multi = 1;
}
__global__ __launch_bounds__(SK_THREADS) void sel_hist_kernel(SelArgs a) {
    __shared__ uint32_t hist[SK_BINS];
    if (blockIdx.x >= a.counters[0]) return;
    const SelTask t = a.tasks[blockIdx.x];
    const SelState S = a.st[t.idx];
    if (S.multi || !S.active) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const int bits = a.pass < 5 ? 11 : 9, shift = 64 - 11 * a.pass - bits;
    for (uint32_t bb = tid; bb < SK_BINS; bb += SK_THREADS) hist[bb] = 0;
    __syncthreads();
    const uint32_t b = t.pbeg + t.tile * SEL_TILE, e = min(b + SEL_TILE, t.pend);
    for (uint32_t i0 = b; i0 < e; i0 += SK_THREADS) {                // uniform trip count: the ballots need every lane
        const uint32_t i = i0 + tid;
        bool in = i < e;
        uint32_t bin = 0;
        if (in) {
            const uint64_t c = sel_code(a, i);
            in = a.pass == 0 || (c >> (shift + bits)) == (S.prefix >> (shift + bits));
            bin = (uint32_t)(c >> shift) & ((1u << bits) - 1);
        }
        const unsigned long long act = __ballot(in);
        if (!act) continue;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(act));
        if (__ballot(in && bin == first) == act) {
            if (lane == (uint32_t)__builtin_ctzll(act)) atomicAdd(&hist[first], (uint32_t)__builtin_popcountll(act));
        } else if (in) atomicAdd(&hist[bin], 1u);
    }
    __syncthreads();
    uint32_t *g = a.hist + (size_t)t.idx * SK_BINS;
    for (uint32_t bb = tid; bb < SK_BINS; bb += SK_THREADS)
        if (hist[bb]) atomicAdd(&g[bb], hist[bb]);
}
__global__ __launch_bounds__(SK_THREADS) void sel_pick_kernel(SelArgs a) {
    __shared__ uint32_t wt[17];
    __shared__ uint32_t s_bin, s_less;
    const uint32_t idx = blockIdx.x, tid = threadIdx.x;
    if (idx >= a.counters[1]) return;
    SelState S = a.st[idx];
    if (S.multi || !S.active) return;
    const int bits = a.pass < 5 ? 11 : 9, shift = 64 - 11 * a.pass - bits;
    uint32_t *g = a.hist + (size_t)idx * SK_BINS;
    const uint32_t c0 = g[2 * tid], c1 = g[2 * tid + 1];
    g[2 * tid] = 0; g[2 * tid + 1] = 0;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan<SK_THREADS>(c0 + c1, wt, &tot);
    if (S.rank >= ex && S.rank < ex + c0) { s_bin = 2 * tid; s_less = ex; }
    else if (S.rank >= ex + c0 && S.rank < ex + c0 + c1) { s_bin = 2 * tid + 1; s_less = ex + c0; }
    __syncthreads();
    if (tid == 0) {
        S.prefix |= (unsigned long long)s_bin << shift;
        S.rank -= s_less;
        if (a.pass == 5) {
            S.hi = S.prefix;
            S.active = 0;
            S.need_lo = (!(S.n & 1) && S.rank == 0) ? 1u : 0u;       // even count and no equal element ranks below the upper middle
        }
        a.st[idx] = S;
    }
}
__global__ __launch_bounds__(SK_THREADS) void sel_maxbelow_kernel(SelArgs a) {
    if (blockIdx.x >= a.counters[0]) return;
    const SelTask t = a.tasks[blockIdx.x];
    const SelState S = a.st[t.idx];
    if (S.multi || !S.need_lo) return;
    const uint32_t b = t.pbeg + t.tile * SEL_TILE, e = min(b + SEL_TILE, t.pend);
    unsigned long long m = 0;
    for (uint32_t i = b + threadIdx.x; i < e; i += SK_THREADS) {
        const unsigned long long c = sel_code(a, i);
        if (c < S.hi && c > m) m = c;
    }
    for (int o = 32; o >= 1; o >>= 1) { const unsigned long long q = __shfl_xor(m, o, 64); m = q > m ? q : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&a.maxbelow[t.idx], m);
}
__global__ void sel_finish_kernel(SelArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.counters[1]) return;
    const SelState S = a.st[idx];
    if (S.multi) return;                                              // several keys: stays flagged for the general path
    const unsigned long long lo_c = S.need_lo ? a.maxbelow[idx] : S.hi;
    double med;
    if (a.kind == 0) {
        const double hi = dec_f64(S.hi);
        med = (S.n & 1) ? hi : (dec_f64(lo_c) + hi) / 2.0;                                 // aggregation.rs:715-719
    } else {
        const int64_t hi = dec_i64(S.hi);
        med = (S.n & 1) ? (double)hi                                                       // aggregation.rs:597-601: the add is in i64
                        : (double)(int64_t)((uint64_t)dec_i64(lo_c) + (uint64_t)hi) / 2.0;
    }
    GroupSortArgs ga{};
    ga.table = a.table; ga.table_mask = a.table_mask;
    gs_publish(ga, S.part == a.P, S.key, med);
    a.only[S.part] = 0;
}

// ---- Nunique of the flagged single-key partitions (few huge groups, the hot key of a skewed column): the number of
// distinct values WITHOUT a sort — every tile inserts its rows' codes into the partition's open-addressing hash
// set in global memory (a plain load first: a duplicate, the common case, costs no atomic; an empty slot is
// claimed with one CAS) and counts the claims.  `==` semantics of Vec::dedup: -0.0 joins 0.0, every NaN counts.
__global__ __launch_bounds__(SK_THREADS) void distinct_insert_kernel(SelArgs a) {
    if (blockIdx.x >= a.counters[0]) return;
    const SelTask t = a.tasks[blockIdx.x];
    const SelState S = a.st[t.idx];
    if (S.multi) return;
    unsigned long long *set = a.sets + S.region;
    const uint32_t b = t.pbeg + t.tile * SEL_TILE, e = min(b + SEL_TILE, t.pend);
    const unsigned long long zero_code = enc_f64(0.0);
    unsigned long long mine = 0;
    for (uint32_t i = b + threadIdx.x; i < e; i += SK_THREADS) {
        unsigned long long c;
        if (a.kind == 0) {
            const double x = __longlong_as_double((long long)a.pvals[i]);
            if (x != x) { mine++; continue; }                        // NaN != NaN: each its own value
            c = x == 0.0 ? zero_code : enc_f64(x);
        } else c = enc_i64((int64_t)a.pvals[i]);
        if (c == EMPTY_KEY) { atomicOr(&a.st[t.idx].need_lo, 1u); continue; }      // i64::MAX's code is the empty marker: a flag instead
        uint32_t slot = hash32(c, 0x51ED270Bu) & S.region_mask;
        for (uint32_t probes = 0; probes <= S.region_mask; probes++) {
            unsigned long long cur = set[slot];
            if (cur == c) break;
            if (cur == EMPTY_KEY) {
                cur = atomicCAS(&set[slot], EMPTY_KEY, c);
                if (cur == EMPTY_KEY) { mine++; break; }
                if (cur == c) break;
            }
            slot = (slot + 1) & S.region_mask;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&a.st[t.idx].distinct, mine);
}
__global__ void distinct_finish_kernel(SelArgs a) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.counters[1]) return;
    const SelState S = a.st[idx];
    if (S.multi) return;
    GroupSortArgs ga{};
    ga.table = a.table; ga.table_mask = a.table_mask;
    gs_publish(ga, S.part == a.P, S.key, __longlong_as_double((long long)(S.distinct + (S.need_lo ? 1ull : 0ull))));   // a u64 count in the entry's 8 bytes
    a.only[S.part] = 0;
}

// Everything after the null compaction, for one set of (key cell, value) pairs; `depth` 1 = the rows of one
// hot partition of the level above, partitioned again with another hash `seed`.
constexpr uint32_t HOT_MIN_ROWS = 1u << 22, HOT_MAX = 8;     // (a nested level costs ~0.3 ms of launches and syncs: only where the sort would cost more)
// the flagged non-NULL partitions of more than HOT_MIN_ROWS rows: out[0] = how many, then (partition, beg, end) triples
__global__ __launch_bounds__(SK_THREADS) void list_hot_partitions_kernel(const uint8_t *only, const uint32_t *offsets, uint32_t NB,
                                                                         uint32_t P, uint32_t *out) {
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < P; p += SK_THREADS) {
        if (!only[p]) continue;
        const uint32_t beg = offsets[(size_t)p * NB], end = offsets[(size_t)(p + 1) * NB];
        if (end - beg <= HOT_MIN_ROWS) continue;
        const uint32_t i = atomicAdd(&s_n, 1u);
        if (i < HOT_MAX) { out[1 + 3 * i] = p; out[2 + 3 * i] = beg; out[3 + 3 * i] = end; }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = s_n;
}
static int32_t median_pairs(pandrs_hip_ctx *c, const KeyDesc &kd, const uint64_t *vals, int64_t nv, int64_t G, int kind, int mode,
                            MedianEntry *table, uint32_t cap_tab, uint32_t seed, int depth) {
    uint64_t *pk = c->work.take<uint64_t>(nv + 1), *pv = c->work.take<uint64_t>(nv + 1);
    if (!pk || !pv) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (median)");
    // Fast path first: partitions of ~9 K rows (they must fit GS_CAP with room for an uneven hash
    // split) with at most GS_KEYS distinct keys each are finished in LDS by group_sort_kernel; what it
    // flags in `only` (oversized partitions: a hot key, a large NULL group) goes through the general
    // sort below.  With nearly unique keys (more than ~1 K groups per partition even at P_MAX) the
    // fast path cannot apply anywhere and is skipped.
    const bool fast = !c->opt.median_generic;
    int64_t P = std::max<int64_t>(1, (int64_t)std::ceil((double)nv / (fast ? 9000.0 : 4900.0)));
    if (fast) P = std::max<int64_t>(P, (int64_t)std::ceil((double)G / 1000.0));
    if (fast && depth > 0) P = P_MAX;        // the finest split: the hot key should end up ALONE in its sub-partition (selection path)
    P = std::min<int64_t>(P, P_MAX);
    const bool use_fast = fast && (depth > 0 || (double)G / (double)P <= 1200.0);
    if (!use_fast) P = std::min<int64_t>(std::max<int64_t>(1, (int64_t)std::ceil((double)nv / 4900.0)), P_MAX);
    PartInfo part{};
    ScatterArgs sa{};
    sa.key = kd; sa.pkeys = pk; sa.n_rows = nv; sa.P = (uint32_t)P; sa.seed = seed; sa.allow_two_pass = 1;
    sa.mv[sa.n_move++] = MoveDesc{vals, pv, 0, 0};
    ST_TRY(radix_partition(c, sa, &part, PANDRS_HIP_PHASE_OTHER, PANDRS_HIP_PHASE_OTHER, PANDRS_HIP_PHASE_OTHER));
    const uint32_t *null_beg = part.offsets + (size_t)P * part.NB, *null_end = part.offsets + (size_t)(P + 1) * part.NB;
    uint8_t *only = nullptr;
    if (use_fast) {
        only = c->work.take<uint8_t>((size_t)P + 16);
        if (!only) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (median)");
        HIP_TRY(hipMemsetAsync(only, 0, (size_t)P + 16, c->stream));
        GroupSortArgs ga{};
        ga.pkeys = pk; ga.pvals = pv; ga.offsets = part.offsets; ga.NB = part.NB; ga.P = (uint32_t)P;
        ga.kind = kind; ga.mode = mode; ga.table = table; ga.table_mask = cap_tab - 1; ga.only = only;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(group_sort_kernel<uint64_t, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)GS_LDS));
        hipLaunchKernelGGL((group_sort_kernel<uint64_t, false>), dim3((unsigned)P + 1), dim3(GS_THREADS), GS_LDS, c->stream, ga);
        if (mode == 0) {    // Median of a flagged partition that holds one key: selection, no sort
            hipLaunchKernelGGL(single_key_median_kernel, dim3((unsigned)P + 1), dim3(SK_THREADS), 0, c->stream, ga);
            if ((uint64_t)nv > SK_MAX_ROWS) {       // ... and of the ones too large for one workgroup
                SelArgs sa2{};
                sa2.pkeys = pk; sa2.pvals = pv; sa2.offsets = part.offsets; sa2.NB = part.NB; sa2.P = (uint32_t)P;
                sa2.kind = kind; sa2.table = table; sa2.table_mask = cap_tab - 1; sa2.only = only; sa2.min_rows = SK_MAX_ROWS;
                sa2.max_big = (uint32_t)((uint64_t)nv / SK_MAX_ROWS + 2);
                sa2.max_tasks = (uint32_t)((uint64_t)nv / SEL_TILE + sa2.max_big + 2);
                sa2.tasks = c->work.take<SelTask>(sa2.max_tasks);
                sa2.counters = c->work.take<uint32_t>(64);
                sa2.st = c->work.take<SelState>(sa2.max_big);
                sa2.hist = c->work.take<uint32_t>((size_t)sa2.max_big * SK_BINS);
                sa2.maxbelow = c->work.take<unsigned long long>(sa2.max_big);
                if (!sa2.tasks || !sa2.counters || !sa2.st || !sa2.hist || !sa2.maxbelow)
                    return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (median selection)");
                HIP_TRY(hipMemsetAsync(sa2.counters, 0, 256, c->stream));
                HIP_TRY(hipMemsetAsync(sa2.hist, 0, (size_t)sa2.max_big * SK_BINS * 4, c->stream));
                HIP_TRY(hipMemsetAsync(sa2.maxbelow, 0, (size_t)sa2.max_big * 8, c->stream));
                hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(SK_THREADS), 0, c->stream, sa2);
                hipLaunchKernelGGL(sel_multi_kernel, dim3(sa2.max_tasks), dim3(SK_THREADS), 0, c->stream, sa2);
                for (int pass = 0; pass < 6; pass++) {
                    sa2.pass = pass;
                    hipLaunchKernelGGL(sel_hist_kernel, dim3(sa2.max_tasks), dim3(SK_THREADS), 0, c->stream, sa2);
                    hipLaunchKernelGGL(sel_pick_kernel, dim3(sa2.max_big), dim3(SK_THREADS), 0, c->stream, sa2);
                }
                hipLaunchKernelGGL(sel_maxbelow_kernel, dim3(sa2.max_tasks), dim3(SK_THREADS), 0, c->stream, sa2);
                hipLaunchKernelGGL(sel_finish_kernel, dim3((sa2.max_big + 255) / 256), dim3(256), 0, c->stream, sa2);
            }
        }
        if (mode == 1 && (uint64_t)nv > GS_CAP) {     // Nunique of the flagged partitions that hold one key: hash sets, no sort
            SelArgs sa2{};
            sa2.pkeys = pk; sa2.pvals = pv; sa2.offsets = part.offsets; sa2.NB = part.NB; sa2.P = (uint32_t)P;
            sa2.kind = kind; sa2.table = table; sa2.table_mask = cap_tab - 1; sa2.only = only; sa2.min_rows = GS_CAP;
            sa2.max_big = (uint32_t)std::min<uint64_t>((uint64_t)P + 1, (uint64_t)nv / GS_CAP + 2);
            sa2.max_tasks = (uint32_t)((uint64_t)nv / SEL_TILE + sa2.max_big + 2);
            sa2.sets_cap = 3ull * (uint64_t)nv + 1024ull * sa2.max_big;
            sa2.tasks = c->work.take<SelTask>(sa2.max_tasks);
            sa2.counters = c->work.take<uint32_t>(64);
            sa2.st = c->work.take<SelState>(sa2.max_big);
            sa2.sets = c->work.take<unsigned long long>(sa2.sets_cap);
            if (!sa2.tasks || !sa2.counters || !sa2.st || !sa2.sets)
                return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (distinct count)");
            HIP_TRY(hipMemsetAsync(sa2.counters, 0, 256, c->stream));
            HIP_TRY(hipMemsetAsync(sa2.sets, 0xFF, sa2.sets_cap * 8, c->stream));
            hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(SK_THREADS), 0, c->stream, sa2);
            hipLaunchKernelGGL(sel_multi_kernel, dim3(sa2.max_tasks), dim3(SK_THREADS), 0, c->stream, sa2);
            hipLaunchKernelGGL(distinct_insert_kernel, dim3(sa2.max_tasks), dim3(SK_THREADS), 0, c->stream, sa2);
            hipLaunchKernelGGL(distinct_finish_kernel, dim3((sa2.max_big + 255) / 256), dim3(256), 0, c->stream, sa2);
        }
        HIP_TRY(hipGetLastError());
    }
    // A big partition that is STILL flagged holds several keys — typically a hot key next to the ~100 ordinary
    // keys of its hash partition.  Sorting it whole is what made skewed inputs slow (one key on 90 % of 100 M rows:
    // 31 ms); instead its rows are partitioned once more with an independent hash, which leaves the hot key (almost
    // surely) alone in one sub-partition — the selection path — and the ordinary keys in LDS-sized ones.
    if (use_fast && depth == 0 && (uint64_t)nv > HOT_MIN_ROWS) {
        uint32_t *hot = c->work.take<uint32_t>(64);
        if (!hot) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (median)");
        hipLaunchKernelGGL(list_hot_partitions_kernel, dim3(1), dim3(SK_THREADS), 0, c->stream, only, part.offsets, part.NB, (uint32_t)P, hot);
        HIP_TRY(hipGetLastError());
        uint32_t *hh = reinterpret_cast<uint32_t *>(c->pinned);
        HIP_TRY(hipMemcpyAsync(hh, hot, 4 * (1 + 3 * HOT_MAX), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const uint32_t n_hot = std::min<uint32_t>(hh[0], HOT_MAX);
        uint32_t list[3 * HOT_MAX];
        for (uint32_t i = 0; i < 3 * n_hot; i++) list[i] = hh[1 + i];          // (the pinned buffer is reused by the nested calls)
        for (uint32_t i = 0; i < n_hot; i++) {
            const uint32_t pid = list[3 * i], beg = list[3 * i + 1], end = list[3 * i + 2];
            const KeyDesc sub{pk + beg, nullptr, nullptr, DT_CELL};
            ST_TRY(median_pairs(c, sub, pv + beg, (int64_t)(end - beg), G, kind, mode, table, cap_tab, 0x7F4A7C15u, 1));
            HIP_TRY(hipMemsetAsync(only + pid, 0, 1, c->stream));               // done: the general path below skips it
        }
    }
    // general path (every partition, or only the flagged ones): sort by (key, value code), walk the runs
    hipLaunchKernelGGL(fill_range_kernel, dim3(256), dim3(256), 0, c->stream, pk, null_beg, null_end, 0ull);
    SortTiles tiles;
    ST_TRY(segmented_sort_u64(c, pk, pv, part.offsets, part.NB, (uint32_t)P + 1, nv, kind == 0 ? 1 : 2, only, &tiles));
    if (tiles.max_tasks == 0) { /* every partition was finished by the fast path */ }
    else if (mode == 1)
        hipLaunchKernelGGL(nunique_runs_kernel, dim3(tiles.max_tasks), dim3(MR_THREADS), 0, c->stream,
                           tiles.tasks, tiles.counters, pk, pv, null_beg, kind, table, cap_tab - 1);
    else
        hipLaunchKernelGGL(median_runs_kernel, dim3(tiles.max_tasks), dim3(MR_THREADS), 0, c->stream,
                           tiles.tasks, tiles.counters, pk, pv, null_beg, kind, table, cap_tab - 1);
    HIP_TRY(hipGetLastError());
    return 0;
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
    const size_t ws = engine_workspace_bytes(n_rows, 4, 1) + two_pass_workspace_bytes(n_rows, 1, 1) + segsort_workspace_bytes(n_rows, P_MAX + 2, 8)
                    + Arena::padded(size_t(cap_tab + 4) * 16) + (1 << 20)
                    + (size_t(n_rows) / SEL_TILE + size_t(n_rows) / SK_MAX_ROWS + 8) * (sizeof(SelTask) + sizeof(SelState) + SK_BINS * 4 + 16) + 8192;
    const size_t ws_all = 2 * ws + 2 * Arena::padded(size_t(n_rows + 1) * 8)       // + one nested level (hot partitions)
                        + (mode == 1 ? 2 * (Arena::padded((3 * size_t(n_rows) + 1024 * size_t(P_MAX + 2)) * 8) + size_t(P_MAX + 4) * sizeof(SelState)) : 0);   // Nunique: hash sets
    ST_TRY(c->work.ensure(ws_all, c->stream));
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
    if (nv > 0) ST_TRY(median_pairs(c, kd, vals, nv, G, kind, mode, table, cap_tab, 0x3C6EF372u, 0));
    hipLaunchKernelGGL(median_lookup_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, c->stream,
                       res.keys, res.key_null, G, table, cap_tab - 1, res.aggs + (size_t)fin_index * res.cap, mode);
    HIP_TRY(hipGetLastError());
    return 0;
}

// group_by's own result (entries.hip): rewrites every partition that fits LDS in place, grouped by key with the
// rows of a key ascending; `only[p]` (zeroed here) is set for the partitions left to the general sort.
int32_t group_order_partitions(pandrs_hip_ctx *c, uint64_t *pk, uint32_t *prow, const uint32_t *offsets, uint32_t NB,
                               uint32_t P, uint8_t *only) {
    HIP_TRY(hipMemsetAsync(only, 0, (size_t)P + 16, c->stream));
    GroupSortArgs ga{};
    ga.pkeys = pk; ga.pvals = prow; ga.out_keys = pk; ga.out_rows = prow;
    ga.offsets = offsets; ga.NB = NB; ga.P = P; ga.only = only;
    const size_t lds = gs_lds_bytes(4);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(group_sort_kernel<uint32_t, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((group_sort_kernel<uint32_t, true>), dim3(P + 1), dim3(GS_THREADS), lds, c->stream, ga);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pandrs
