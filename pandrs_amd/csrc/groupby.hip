// groupby.hip — LDS hash aggregate + the engine around it (gfx950, wave64); the radix partitioner it
// drives is partition.hip, the host entry points are entries.hip.
//
// Replaces the reference's group_by + aggregate hot loops
//   src/optimized/split_dataframe/group/grouping.rs:62-104   (row -> group)
//   src/optimized/split_dataframe/group/aggregation.rs:500-754 (per-group fold)
//   src/optimized/lazy.rs:186-404                             (inline copy)
// which stringify every key and gather per group on one CPU thread.
//
// Pipeline (all kernels on the context's stream, inputs/outputs resident in HBM):
//   estimate  sampled distinct count -> radix fan-out P so a partition's groups fit one LDS table
//   histogram per-workgroup counts of hash-partition ids (keys only, coalesced 8-byte scan)
//   scan      exclusive scan, partition-major, -> exact write cursor per (partition, workgroup)
//   scatter   tile-local counting sort in LDS, then each column is staged through LDS so a
//             partition's rows leave the CU as contiguous runs (coalesced HBM writes)
//   aggregate one workgroup per partition: open-addressing key table + per-state arrays in LDS,
//             native ds_cmpst_b64 / ds_add_f64 / ds_min_u64 / ds_max_u64 / ds_add_u64, then
//             ballot + prefix-sum compaction of occupied slots into dense output rows, with the
//             reference's finalisation rules applied in registers.
// HBM traffic: keys once (histogram) + all columns read/written once (scatter) + read once
// (aggregate) + outputs.  No MFMA: the path is integer/byte work bounded by HBM.
#include "aggregate.hpp"

#include <algorithm>
#include <cmath>

namespace pandrs {

// ------------------------------------------------------------------------------------ aggregate
// one workgroup: partition sizes -> task list (a partition of more than over_rows rows becomes
// ceil(size / slice_rows) tasks flagged `multi`; empty partitions get no task).  over_rows >= slice_rows: a partition
// is cut only when it is far above the average, and then into pieces the size of an average partition — a piece is one
// workgroup's job, and a job four times the usual one is the kernel's tail.
__global__ __launch_bounds__(1024) void build_tasks_kernel(const uint32_t *offsets, uint32_t NB, uint32_t P1, uint32_t over_rows,
                                                           uint32_t slice_rows, AggTask *tasks, uint32_t *n_tasks,
                                                           uint32_t max_tasks) {
    __shared__ uint32_t wt[17];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < P1; base += 1024) {
        uint32_t p = base + threadIdx.x, beg = 0, end = 0, ns = 0;
        if (p < P1) {
            beg = offsets[(size_t)p * NB]; end = offsets[(size_t)(p + 1) * NB];
            ns = end - beg > over_rows ? (end - beg + slice_rows - 1) / slice_rows : (end > beg ? 1u : 0u);
        }
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<1024>(ns, wt, &tot) + carry;
        for (uint32_t q = 0; q < ns; q++) {
            if (ex + q < max_tasks) {
                uint32_t b = beg + q * slice_rows;
                tasks[ex + q] = AggTask{p, b, ns > 1 ? min(b + slice_rows, end) : end, ns > 1 ? 1u : 0u};
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_tasks = min(carry, max_tasks);
}

// aggregate2's work list: tables (one per partition; an oversized partition is cut into tables of <= slice_rows
// rows flagged `multi`) and, per table, the row ranges that feed it.  One workgroup; counts[0] = tasks, [1] = tables.
// `order` (with `trows`, scratch of the same length): the tables' indices by falling size, in 64 classes of an eighth of the average
// table — aggregate2's workgroups draw tables from a ticket counter in this order, so the large ones (partitions holding hot keys,
// not far enough above the average to be cut) start first and the small ones fill the gaps behind them.
__global__ __launch_bounds__(1024) void build_tables_kernel(SegSource ss, uint32_t P1, uint32_t over_rows, uint32_t slice_rows_in, AggTask *tasks,
                                                            AggTable *tables, uint32_t *counts, uint32_t max_tasks,
                                                            uint32_t max_tables, uint32_t *order, uint32_t *trows) {
    __shared__ uint32_t wt[17];
    __shared__ uint32_t carry[2];
    __shared__ uint32_t cls_n[64];
    __shared__ unsigned long long all_rows;
    if (threadIdx.x < 2) carry[threadIdx.x] = 0;
    if (threadIdx.x < 64) cls_n[threadIdx.x] = 0;
    if (threadIdx.x == 0) all_rows = 0;
    __syncthreads();
    for (uint32_t base = 0; base < P1; base += 1024) {
        const uint32_t p = base + threadIdx.x;
        uint32_t sb[8], se[8], n_seg = 0;
        uint64_t R = 0;
        if (p < P1) {
            if (ss.gbeg) {
                n_seg = 8;
                uint32_t cu[8], en[8];
#pragma unroll
                for (int g = 0; g < 8; g++) { sb[g] = ss.gbeg[p * 8 + g]; cu[g] = ss.gcur[p * 8 + g]; en[g] = ss.gend[p * 8 + g]; }      // 24 loads in flight
#pragma unroll
                for (int g = 0; g < 8; g++) se[g] = max(min(cu[g], en[g]), sb[g]);
            } else {
                n_seg = 1;
                sb[0] = ss.offsets[(size_t)p * ss.NB]; se[0] = ss.offsets[(size_t)(p + 1) * ss.NB];
            }
            for (uint32_t g = 0; g < n_seg; g++) R += se[g] - sb[g];
        }
        const uint32_t n_tab = R > over_rows ? (uint32_t)((R + slice_rows_in - 1) / slice_rows_in) : (R ? 1u : 0u);
        const uint64_t slice_rows = n_tab > 1 ? (uint64_t)slice_rows_in : max(R, (uint64_t)1);      // (one table: one virtual range)
        // tasks = non-empty intersections of the segments with the tables' virtual row ranges
        uint32_t n_task = 0;
        {
            uint64_t v0 = 0;
            for (uint32_t g = 0; g < n_seg; g++) {
                const uint64_t len = se[g] - sb[g];
                if (len) n_task += (uint32_t)((v0 + len - 1) / slice_rows - v0 / slice_rows) + 1u;
                v0 += len;
            }
        }
        uint32_t tot_tab, tot_task;
        const uint32_t ex_tab = block_exclusive_scan<1024>(n_tab, wt, &tot_tab) + carry[1];
        const uint32_t ex_task = block_exclusive_scan<1024>(n_task, wt, &tot_task) + carry[0];
        if (n_tab && ex_tab + n_tab <= max_tables && ex_task + n_task <= max_tasks) {
            uint32_t ti = ex_task, tb = ex_tab, g = 0;
            uint64_t v0 = 0;                       // virtual row where segment g starts
            for (uint32_t j = 0; j < n_tab; j++) {
                const uint64_t lo = (uint64_t)j * slice_rows, hi = min((uint64_t)(j + 1) * slice_rows, (uint64_t)R);
                const uint32_t first = ti;
                while (g < n_seg) {
                    const uint64_t len = se[g] - sb[g];
                    const uint64_t a0 = max(lo, v0), a1 = min(hi, v0 + len);
                    if (a1 > a0) tasks[ti++] = AggTask{p, sb[g] + (uint32_t)(a0 - v0), sb[g] + (uint32_t)(a1 - v0), n_tab > 1 ? 1u : 0u};
                    if (v0 + len > hi) break;      // the segment continues in the next table
                    v0 += len; g++;
                }
                if (trows) trows[tb] = (uint32_t)(hi - lo);
                tables[tb++] = AggTable{first, ti - first, p, n_tab > 1 ? 1u : 0u};
            }
            if (trows) atomicAdd(&all_rows, (unsigned long long)R);
        }
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += tot_task; carry[1] += tot_tab; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[0] = min(carry[0], max_tasks); counts[1] = min(carry[1], max_tables);
        counts[2] = (carry[0] > max_tasks || carry[1] > max_tables) ? 1u : 0u;     // cannot happen with the caller's bounds
    }
    if (!order) return;
    // counting sort of the table indices by size class (class 0 = the largest); the order inside a class is whatever the cursors give
    const uint32_t n_tables = min(carry[1], max_tables);
    const unsigned long long unit = max(all_rows / (8ull * max(n_tables, 1u)), 1ull);       // an eighth of the average table
    auto cls_of = [&](uint32_t t) { return 63u - (uint32_t)min(((unsigned long long)trows[t] + unit / 2) / unit, 63ull); };   // (rounded: near-equal tables share a class)
    for (uint32_t t = threadIdx.x; t < n_tables; t += 1024) atomicAdd(&cls_n[cls_of(t)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 64; k++) { const uint32_t n = cls_n[k]; cls_n[k] = run; run += n; }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_tables; t += 1024) order[atomicAdd(&cls_n[cls_of(t)], 1u)] = t;
}

// The reference's finalisation of one aggregate from the group's states
// (aggregation.rs:507-556, :625-674, :743).
__device__ __forceinline__ double finalize(const FinDev &f, const uint64_t *st, uint32_t stride,
                                           uint32_t slot, uint64_t gsize) {
    auto cell = [&](int8_t s) { return st[(size_t)s * stride + slot]; };
    switch (f.op) {
    case PANDRS_HIP_AGG_COUNT: return (double)gsize;
    case PANDRS_HIP_AGG_SUM:
        return f.kind == 0 ? __longlong_as_double((long long)cell(f.st_add))
                           : (double)(int64_t)cell(f.st_add);
    case PANDRS_HIP_AGG_MEAN: {
        uint64_t nn = f.st_nn >= 0 ? cell(f.st_nn) : gsize;
        if (nn == 0) return 0.0;
        double s = f.kind == 0 ? __longlong_as_double((long long)cell(f.st_add))
                               : (double)(int64_t)cell(f.st_add);
        return s / (double)nn;
    }
    case PANDRS_HIP_AGG_MIN:
        if (f.kind == 0) {
            double v = dec_f64(cell(f.st_min));
            return v == __longlong_as_double(0x7FF0000000000000ll) ? 0.0 : v;
        } else {
            int64_t v = dec_i64(cell(f.st_min));
            return v == INT64_MAX ? 0.0 : (double)v;
        }
    case PANDRS_HIP_AGG_MAX:
        if (f.kind == 0) {
            double v = dec_f64(cell(f.st_max));
            return v == __longlong_as_double((long long)0xFFF0000000000000ull) ? 0.0 : v;
        } else {
            int64_t v = dec_i64(cell(f.st_max));
            return v == INT64_MIN ? 0.0 : (double)v;
        }
    case PANDRS_HIP_AGG_STD:
    case PANDRS_HIP_AGG_VAR: {      // aggregation.rs:881-903: Bessel, n <= 1 => 0.0, empty => 0.0
        uint64_t nn = f.st_nn >= 0 ? cell(f.st_nn) : gsize;
        double var = nn > 1 ? __longlong_as_double((long long)cell(f.st_ssq)) / ((double)nn - 1.0) : 0.0;
        return f.op == PANDRS_HIP_AGG_STD ? sqrt(var) : var;
    }
    case PANDRS_HIP_AGG_FIRST:
    case PANDRS_HIP_AGG_LAST: {     // aggregation.rs:605-624, :723-742: value at the first / last row, null => 0.0
        int64_t row = dec_i64(cell(f.op == PANDRS_HIP_AGG_FIRST ? f.rowsrc_min : f.rowsrc_max));
        if (f.col_null && bit_at(f.col_null, row)) return 0.0;
        return f.kind == 0 ? reinterpret_cast<const double *>(f.col_data)[row]
                           : (double)reinterpret_cast<const int64_t *>(f.col_data)[row];
    }
    }
    return 0.0;
}

// LDS: keys[T+1] | gsize[T+1] | states[round_states][T+1] | posmap[T+1] (u32) | misc[32]
// One workgroup per radix partition.  The aggregated columns are processed in `n_rounds` passes
// over the partition (key table and group sizes persist, the state arrays are reused), which
// keeps the bytes per slot small => more slots per table => fewer, longer radix partitions.
// NSRC    compile-time bound on sources per round (0 = runtime)
// PROFILE -1 = every per-source property is read from the descriptors at run time;
//         >= 0 = all sources share (kind, ops, validity): bit0 has validity bytes,
//                bits1-3 ops present (add, min, max), bit4 kind (0 f64, 1 i64); raw rows, one round.
//         The uniform profiles remove ~300 scalar branches per row pair from the hot loop.
//         RUNS (uniform profiles only): the rows arrive CLUSTERED by key (input sorted or grouped by key —
//         the radix partition keeps neighbouring rows together), so with the usual row-per-lane layout
//         the lanes of a wave hold the same few keys and their LDS atomics serialise on the same
//         addresses.  Here every thread takes AG_RUN CONSECUTIVE rows, folds equal neighbours in
//         registers and touches the table once per run.
//         MERGE (NSRC 0, PROFILE -1): the rows are partial records (merge of partials, multi-GPU exchange):
//         every source feeds exactly ONE state and carries no validity, so the loop is one record per
//         thread with a runtime loop over the states — no per-source register arrays (the generic loop
//         keeps 2 x 16 values + flags live and spills at 12+ states).
template <int NSRC, int PROFILE, bool RUNS = false, bool MERGE = false>
__global__ __launch_bounds__(AG_THREADS) void aggregate_kernel(AggArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NS = NSRC > 0 ? NSRC : MAX_SRC;
    constexpr bool GEN = PROFILE < 0;
    auto f_kind = [](const SrcDev &sd) -> int { return GEN ? sd.kind : ((PROFILE >> 4) & 1); };
    auto f_add = [](const SrcDev &sd) -> bool { return GEN ? sd.st_add >= 0 : ((PROFILE >> 1) & 1) != 0; };
    auto f_min = [](const SrcDev &sd) -> bool { return GEN ? sd.st_min >= 0 : ((PROFILE >> 2) & 1) != 0; };
    auto f_max = [](const SrcDev &sd) -> bool { return GEN ? sd.st_max >= 0 : ((PROFILE >> 3) & 1) != 0; };
    auto f_valid = [](const SrcDev &sd) -> bool { return GEN ? sd.valid != nullptr : (PROFILE & 1) != 0; };
    auto f_nn = [](const SrcDev &sd) -> bool { return (GEN || (PROFILE & 1)) ? sd.st_nn >= 0 : false; };
    const uint32_t T = a.T, T1 = T + 2, tid = threadIdx.x;   // slot T: sentinel-valued key, slot T+1: NULL key (direct mode)
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *gsz = keys + T1;
    uint64_t *st = gsz + T1;
    uint32_t *posmap = reinterpret_cast<uint32_t *>(st + (size_t)a.round_states * T1);
    uint32_t *misc = posmap + ((T1 + 1) & ~1u);
    // misc[0..16] wave totals, [20] overflow, [21] sentinel-key-present, [22] output base, [23] NULL-key-present
    uint32_t p = blockIdx.x, beg, end;
    const bool direct = a.direct != 0;
    bool multi = false;
    if (direct) {
        beg = min(p * a.d_chunk, a.d_rows); end = min(beg + a.d_chunk, a.d_rows);
    } else if (a.tasks) {
        if (blockIdx.x >= *a.n_tasks) return;
        const AggTask t = a.tasks[blockIdx.x];
        p = t.part; beg = t.beg; end = t.end; multi = t.multi != 0;
    } else {
        beg = a.offsets[(size_t)p * a.NB]; end = a.offsets[(size_t)(p + 1) * a.NB];
    }
    if (beg >= end) return;
    // a slice of an oversized partition emits partial records into the side buffers
    uint64_t *const o_keys = multi ? a.side_keys : a.out_keys;
    uint8_t *const o_null = multi ? a.side_null : a.out_null;
    uint64_t *const o_states = multi ? a.side_states : a.out_states;
    const size_t o_cap = multi ? a.side_cap : a.cap;
    const bool emit_partials = a.partials != 0 || multi;

    for (uint32_t s = tid; s < T1; s += AG_THREADS) { keys[s] = EMPTY_KEY; gsz[s] = 0; }
    if (tid < 32) misc[tid] = 0;

    for (int round = 0; round < a.n_rounds; round++) {
        const int s0 = a.round_src_begin[round], nsrc = a.round_src_begin[round + 1] - s0;
        for (int k = 0; k < a.n_states; k++) {
            if (a.st_round[k] != round) continue;
            uint64_t idv = state_identity(a.kinds[k]);
            uint64_t *dst = st + (size_t)a.st_lds[k] * T1;
            for (uint32_t s = tid; s < T1; s += AG_THREADS) dst[s] = idv;
        }
        __syncthreads();

        auto find_slot = [&](uint64_t k) -> uint32_t {
            if (k == EMPTY_KEY) { misc[21] = 1; return T; }
            const uint32_t NBK = T >> 2;
            uint32_t bk = slot_of(hash32(k, a.seed), NBK), probe = 0;
            while (probe < NBK) {
                const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(keys + 4 * bk);
                const ulonglong2 lo = bp[0], hi = bp[1];
                const uint64_t c4[4] = {lo.x, lo.y, hi.x, hi.y};
                int hit = -1, emp = -1;
#pragma unroll
                for (int q = 3; q >= 0; q--) {
                    if (c4[q] == k) hit = q;
                    if (c4[q] == EMPTY_KEY) emp = q;
                }
                if (hit >= 0) return 4 * bk + hit;
                if (emp >= 0) {
                    const uint64_t old = atomicCAS((unsigned long long *)&keys[4 * bk + emp], EMPTY_KEY, k);
                    if (old == EMPTY_KEY || old == k) return 4 * bk + emp;
                    continue;
                }
                bk = bk + 1 == NBK ? 0 : bk + 1;
                probe++;
            }
            misc[20] = 1;                       // table full: host retries with more partitions
            return T + 2;
        };
        if constexpr (MERGE) {
            for (uint32_t i = beg + tid; i < end; i += AG_THREADS) {
                if (*reinterpret_cast<volatile uint32_t *>(&misc[20])) break;      // a table that is full stays full: the attempt is lost, stop walking it (every further unplaced row probed all of its buckets: 44 ms for 100 M rows)
                const uint32_t slot = find_slot(__builtin_nontemporal_load(a.pkeys + i));
                if (slot > T) continue;
                if (round == 0) atomicAdd((unsigned long long *)&gsz[slot], (unsigned long long)a.pgsize[i]);
                for (int c = 0; c < nsrc; c++) {
                    const SrcDev &sd = a.src[s0 + c];
                    const uint64_t x = __builtin_nontemporal_load(sd.vals + i);
                    if (sd.st_add >= 0) {
                        if (sd.kind == 0) atomicAdd(reinterpret_cast<double *>(&st[(size_t)sd.st_add * T1 + slot]), __longlong_as_double((long long)x));
                        else atomicAdd((unsigned long long *)&st[(size_t)sd.st_add * T1 + slot], x);
                    } else if (sd.st_nn >= 0) {
                        atomicAdd((unsigned long long *)&st[(size_t)sd.st_nn * T1 + slot], x);
                    } else {
                        // partial extremes are never NaN (NaN operands were ignored when they were folded)
                        const uint64_t e = sd.kind == 0 ? enc_f64(__longlong_as_double((long long)x)) : enc_i64((int64_t)x);
                        if (sd.st_min >= 0) {
                            if (e < st[(size_t)sd.st_min * T1 + slot]) atomicMin((unsigned long long *)&st[(size_t)sd.st_min * T1 + slot], e);
                        } else if (sd.st_max >= 0) {
                            if (e > st[(size_t)sd.st_max * T1 + slot]) atomicMax((unsigned long long *)&st[(size_t)sd.st_max * T1 + slot], e);
                        }
                    }
                }
            }
        } else if constexpr (RUNS) {
            constexpr int AG_RUN = 4;
            for (uint32_t base = beg + tid * AG_RUN; base < end; base += AG_THREADS * AG_RUN) {
                if (*reinterpret_cast<volatile uint32_t *>(&misc[20])) break;
                uint64_t rk[AG_RUN], rv[AG_RUN][NS];
                bool rok[AG_RUN][NS];
#pragma unroll
                for (int j = 0; j < AG_RUN; j++) {
                    const uint32_t i = min(base + j, end - 1);
                    rk[j] = __builtin_nontemporal_load(a.pkeys + i);
#pragma unroll
                    for (int c = 0; c < NS; c++) {
                        if (c < nsrc) {
                            const SrcDev &sd = a.src[s0 + c];
                            rv[j][c] = __builtin_nontemporal_load(sd.vals + i);
                            rok[j][c] = f_valid(sd) ? sd.valid[i] != 0 : true;
                        }
                    }
                }
                const int n_here = (int)min((uint32_t)AG_RUN, end - base);
                uint64_t cur = rk[0], r_gs = 0, r_sum[NS], r_min[NS], r_max[NS];
                uint32_t r_nn[NS];
#pragma unroll
                for (int c = 0; c < NS; c++) { r_sum[c] = 0; r_nn[c] = 0; r_min[c] = ~0ull; r_max[c] = 0ull; }
#pragma unroll
                for (int j = 0; j <= AG_RUN; j++) {
                    const bool more = j < n_here;
                    if (j > 0 && (!more || rk[j < AG_RUN ? j : 0] != cur)) {
                        // ---- the run ends: one table update for all its rows ----
                        const uint32_t slot = find_slot(cur);
                        if (slot <= T) {
                            if (round == 0) atomicAdd((unsigned long long *)&gsz[slot], r_gs);
#pragma unroll
                            for (int c = 0; c < NS; c++) {
                                if (c < nsrc) {
                                    const SrcDev &sd = a.src[s0 + c];
                                    if (r_nn[c]) {
                                        if (f_nn(sd)) atomicAdd((unsigned long long *)&st[(size_t)sd.st_nn * T1 + slot], (unsigned long long)r_nn[c]);
                                        if (f_add(sd)) {
                                            if (f_kind(sd) == 0) atomicAdd(reinterpret_cast<double *>(&st[(size_t)sd.st_add * T1 + slot]), __longlong_as_double((long long)r_sum[c]));
                                            else atomicAdd((unsigned long long *)&st[(size_t)sd.st_add * T1 + slot], r_sum[c]);
                                        }
                                    }
                                    if (f_min(sd) && r_min[c] < st[(size_t)sd.st_min * T1 + slot])
                                        atomicMin((unsigned long long *)&st[(size_t)sd.st_min * T1 + slot], r_min[c]);
                                    if (f_max(sd) && r_max[c] > st[(size_t)sd.st_max * T1 + slot])
                                        atomicMax((unsigned long long *)&st[(size_t)sd.st_max * T1 + slot], r_max[c]);
                                }
                            }
                        }
                        r_gs = 0;
#pragma unroll
                        for (int c = 0; c < NS; c++) { r_sum[c] = 0; r_nn[c] = 0; r_min[c] = ~0ull; r_max[c] = 0ull; }
                    }
                    if (!more || j == AG_RUN) break;
                    cur = rk[j];
                    r_gs += 1;
#pragma unroll
                    for (int c = 0; c < NS; c++) {
                        if (c < nsrc && rok[j][c]) {
                            const SrcDev &sd = a.src[s0 + c];
                            const uint64_t x = rv[j][c];
                            bool cmp = true;
                            uint64_t e;
                            if (f_kind(sd) == 0) {
                                const double dd = __longlong_as_double((long long)x);
                                cmp = dd == dd; e = enc_f64(dd);
                                r_sum[c] = r_nn[c] ? (uint64_t)__double_as_longlong(__longlong_as_double((long long)r_sum[c]) + dd) : x;
                            } else {
                                e = enc_i64((int64_t)x);
                                r_sum[c] += x;
                            }
                            r_nn[c]++;
                            if (cmp) { if (e < r_min[c]) r_min[c] = e; if (e > r_max[c]) r_max[c] = e; }
                        }
                    }
                }
            }
        } else {
        // Software pipeline: two rows per thread are processed while the NEXT two rows' global loads
        // are already in flight (indices are clamped to the partition's last row, so the prefetch
        // loads are unconditional; a clamped row is simply never processed).
        auto fetch = [&](uint32_t i0, uint64_t (&k2)[2], uint64_t (&v)[2][NS], uint64_t (&gs)[2], bool (&ok)[2][NS], bool (&kn)[2]) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t i = min(i0 + h * AG_THREADS, end - 1);
                if (direct) {
                    kn[h] = key_is_null(a.dkey, i);
                    k2[h] = key_cell(a.dkey, i);
                } else {
                    kn[h] = false;
                    k2[h] = __builtin_nontemporal_load(a.pkeys + i);
                }
                gs[h] = (GEN && round == 0 && a.pgsize) ? (uint64_t)a.pgsize[i] : 1ull;
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    if (c < nsrc) {
                        const SrcDev &sd = a.src[s0 + c];
                        v[h][c] = __builtin_nontemporal_load(sd.vals + i);
                        ok[h][c] = f_valid(sd) ? (direct ? !bit_at(sd.valid, i) : sd.valid[i] != 0) : true;
                    }
                }
            }
        };
        uint64_t k2n[2], vn[2][NS], gsn[2];
        bool okn[2][NS], knn[2];
        if (beg + tid < end) fetch(beg + tid, k2n, vn, gsn, okn, knn);
        for (uint32_t i0 = beg + tid; i0 < end; i0 += 2 * AG_THREADS) {
            if (*reinterpret_cast<volatile uint32_t *>(&misc[20])) break;
            const bool has1 = i0 + AG_THREADS < end;
            uint64_t k2[2], v[2][NS], gs[2];
            bool ok[2][NS], kn[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                k2[h] = k2n[h]; gs[h] = gsn[h]; kn[h] = knn[h];
#pragma unroll
                for (int c = 0; c < NS; c++) { v[h][c] = vn[h][c]; ok[h][c] = okn[h][c]; }
            }
            if (i0 + 2 * AG_THREADS < end) fetch(i0 + 2 * AG_THREADS, k2n, vn, gsn, okn, knn);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (h && !has1) break;
                const uint64_t k = k2[h];
                uint32_t slot;
                if (kn[h]) {
                    slot = T + 1;               // NULL key: its own group (grouping.rs:74)
                    misc[23] = 1;
                } else if (k == EMPTY_KEY) {
                    slot = T;
                    misc[21] = 1;
                } else {
                    // 4-key buckets (32 B, two ds_read_b128): one LDS round trip tests four
                    // slots, so a wave's longest probe chain stays ~1-3 trips even at 70-80 % load.
                    const uint32_t NBK = T >> 2;
                    uint32_t bk = slot_of(hash32(k, a.seed), NBK);
                    uint32_t probe = 0;
                    slot = T;
                    while (probe < NBK) {
                        const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(keys + 4 * bk);
                        ulonglong2 lo = bp[0], hi = bp[1];
                        uint64_t c4[4] = {lo.x, lo.y, hi.x, hi.y};
                        int hit = -1, emp = -1;
#pragma unroll
                        for (int q = 3; q >= 0; q--) {
                            if (c4[q] == k) hit = q;
                            if (c4[q] == EMPTY_KEY) emp = q;
                        }
                        if (hit >= 0) { slot = 4 * bk + hit; break; }
                        if (emp >= 0) {
                            if (round > 0) break;   // cannot happen: the key was inserted in round 0
                            uint64_t old = atomicCAS((unsigned long long *)&keys[4 * bk + emp], EMPTY_KEY, k);
                            if (old == EMPTY_KEY || old == k) { slot = 4 * bk + emp; break; }
                            continue;               // lost the race for that slot: re-read this bucket
                        }
                        bk = bk + 1 == NBK ? 0 : bk + 1;
                        probe++;
                    }
                    if (slot == T) { misc[20] = 1; continue; }   // table full: host retries with more partitions
                }
                if (round == 0) atomicAdd((unsigned long long *)&gsz[slot], gs[h]);
                // min/max: all current states are read first (back-to-back ds_read_b64, one wait);
                // an LDS atomic is issued only where the row improves the state.  Skipping on a
                // stale read is safe: states only move towards the extreme.
                uint64_t enc[NS], cur_mn[NS], cur_mx[NS];
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    if (c < nsrc) {
                        const SrcDev &sd = a.src[s0 + c];
                        enc[c] = f_kind(sd) == 0 ? enc_f64(__longlong_as_double((long long)v[h][c])) : enc_i64((int64_t)v[h][c]);
                        cur_mn[c] = f_min(sd) ? st[(size_t)sd.st_min * T1 + slot] : 0ull;
                        cur_mx[c] = f_max(sd) ? st[(size_t)sd.st_max * T1 + slot] : ~0ull;
                    }
                }
#pragma unroll
                for (int c = 0; c < NS; c++) {
                    if (c < nsrc && ok[h][c]) {
                        const SrcDev &sd = a.src[s0 + c];
                        const uint64_t x = v[h][c];
                        if (f_nn(sd)) atomicAdd((unsigned long long *)&st[(size_t)sd.st_nn * T1 + slot], 1ull);
                        bool cmp = true;
                        if (f_kind(sd) == 0) {
                            double d = __longlong_as_double((long long)x);
                            if (f_add(sd)) atomicAdd(reinterpret_cast<double *>(&st[(size_t)sd.st_add * T1 + slot]), d);
                            cmp = d == d;   // Rust f64::min/max ignore NaN operands (aggregation.rs:653,:666)
                        } else {
                            if (f_add(sd)) atomicAdd((unsigned long long *)&st[(size_t)sd.st_add * T1 + slot], x);
                            if (GEN && sd.st_fadd >= 0)
                                atomicAdd(reinterpret_cast<double *>(&st[(size_t)sd.st_fadd * T1 + slot]), (double)(int64_t)x);
                        }
                        if (cmp && f_min(sd) && enc[c] < cur_mn[c])
                            atomicMin((unsigned long long *)&st[(size_t)sd.st_min * T1 + slot], enc[c]);
                        if (cmp && f_max(sd) && enc[c] > cur_mx[c])
                            atomicMax((unsigned long long *)&st[(size_t)sd.st_max * T1 + slot], enc[c]);
                    }
                }
            }
        }
        }   // !RUNS
        __syncthreads();
        if (misc[20]) { if (tid == 0) a.counters[1] = 1; return; }
        const bool sentinel = misc[21] != 0, nullseen = misc[23] != 0;
        const bool null_part = !direct && p == a.P;
        auto occupied = [&](uint32_t s) { return s < T ? keys[s] != EMPTY_KEY : (s == T ? sentinel : nullseen); };

        if (round == 0) {
            // ---- compaction: occupied slots -> dense output rows (ballot + prefix sum) ----
            uint32_t mine = 0;
            for (uint32_t s = tid; s < T1; s += AG_THREADS)
                mine += occupied(s) ? 1u : 0u;
            uint32_t total;
            block_exclusive_scan<AG_THREADS>(mine, misc, &total);
            if (direct && total > a.d_task_cap) { if (tid == 0) a.counters[1] = 1; return; }   // estimate was too low
            if (tid == 0) misc[22] = atomicAdd(&a.counters[multi ? 2 : 0], total);
            __syncthreads();
            uint32_t run = misc[22];
            __syncthreads();
            for (uint32_t sbase = 0; sbase < T1; sbase += AG_THREADS) {
                uint32_t s = sbase + tid;
                bool occ = s < T1 && occupied(s);
                uint32_t tot;
                uint32_t ex = block_exclusive_scan<AG_THREADS>(occ ? 1u : 0u, misc, &tot);
                if (occ) {
                    uint32_t pos = run + ex;
                    posmap[s] = pos;
                    const bool isnull = null_part || s == T + 1;
                    o_keys[pos] = isnull ? 0ull : (s < T ? keys[s] : EMPTY_KEY);
                    o_null[pos] = isnull ? 1 : 0;
                    if (emit_partials) o_states[pos] = gsz[s];
                }
                run += tot;
            }
            __syncthreads();
        }
        // ---- second pass (Std / Var): sum of squared deviations from the group mean, the
        // reference's two-pass variance (aggregation.rs:881-903).  Sums and counts of this (only)
        // round are still in LDS; the partition's rows are streamed once more.
        if (GEN && a.second_pass) {
            const bool sentinel2 = misc[21] != 0;
            (void)sentinel2;
            for (uint32_t i = beg + tid; i < end; i += AG_THREADS) {
                const uint64_t k = a.pkeys[i];
                uint32_t slot = T;
                if (k != EMPTY_KEY) {
                    const uint32_t NBK = T >> 2;
                    uint32_t bk = slot_of(hash32(k, a.seed), NBK);
                    for (uint32_t probe = 0; probe < NBK && slot == T; probe++) {
                        const uint64_t *b4 = keys + 4 * bk;
#pragma unroll
                        for (int q = 0; q < 4; q++) if (b4[q] == k) slot = 4 * bk + q;
                        bk = bk + 1 == NBK ? 0 : bk + 1;
                    }
                    if (slot == T) continue;        // unreachable: every key was inserted above
                }
                const uint64_t g = gsz[slot];
                if (a.n_mvar > 0) {                 // partial records (aggregate.hpp, MergeVar): n_i (m_i - m)^2
                    for (int v = 0; v < a.n_mvar; v++) {
                        const MergeVar &mv = a.mvar[v];
                        const uint64_t nn_i = mv.nn_col ? mv.nn_col[i] : (uint64_t)a.pgsize[i];
                        if (nn_i == 0) continue;
                        const double sum = __longlong_as_double((long long)st[(size_t)mv.sum * T1 + slot]);
                        const uint64_t nn = mv.nn >= 0 ? st[(size_t)mv.nn * T1 + slot] : g;
                        const double dlt = __longlong_as_double((long long)mv.sum_col[i]) / (double)nn_i - sum / (double)nn;
                        atomicAdd(reinterpret_cast<double *>(&st[(size_t)mv.ssq * T1 + slot]), (double)nn_i * dlt * dlt);
                    }
                    continue;
                }
                for (int c = 0; c < nsrc; c++) {
                    const SrcDev &sd = a.src[s0 + c];
                    if (sd.st_ssq < 0) continue;
                    if (sd.valid && sd.valid[i] == 0) continue;
                    const uint64_t x = sd.vals[i];
                    const double xv = sd.kind == 0 ? __longlong_as_double((long long)x) : (double)(int64_t)x;
                    const int8_t ssum = sd.kind == 0 ? sd.st_add : sd.st_fadd;
                    const double sum = __longlong_as_double((long long)st[(size_t)ssum * T1 + slot]);
                    const uint64_t nn = sd.st_nn >= 0 ? st[(size_t)sd.st_nn * T1 + slot] : g;
                    const double dlt = xv - sum / (double)nn;
                    atomicAdd(reinterpret_cast<double *>(&st[(size_t)sd.st_ssq * T1 + slot]), dlt * dlt);
                }
            }
            __syncthreads();
        }
        // ---- this round's outputs ----
        for (uint32_t s = tid; s < T1; s += AG_THREADS) {
            if (!occupied(s)) continue;
            const size_t pos = posmap[s];
            if (emit_partials) {
                for (int k = 0; k < a.n_states; k++)
                    if (a.st_round[k] == round)
                        o_states[(size_t)(k + 1) * o_cap + pos] =
                            state_natural(a.kinds[k], st[(size_t)a.st_lds[k] * T1 + s]);
            } else {
                const uint64_t g = gsz[s];
                for (int f = 0; f < a.n_fin; f++)
                    if (a.fin[f].round == round)
                        a.out_aggs[(size_t)f * a.cap + pos] = finalize(a.fin[f], st, T1, s, g);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------ host side

// Builds the state layout from (value dtypes, has-null flags, aggregate specs).  The layout is a
// pure function of these, so shards on different GPUs produce mergeable partials.
int32_t build_plan(const int32_t *val_dtypes, const uint8_t *val_has_nulls, int n_vals,
                          const pandrs_hip_agg_spec *aggs, int n_aggs, Plan &pl) {
    if (n_aggs > MAX_AGGS) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "more than %d aggregates", MAX_AGGS);
    if (n_vals > 1024) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "too many value columns");
    int src_of[1024];
    for (int i = 0; i < n_vals; i++) src_of[i] = -1;
    bool too_many = false;
    auto new_state = [&](int8_t kind) -> int8_t {
        if (pl.n_states >= MAX_STATES) { too_many = true; return -1; }
        pl.kinds[pl.n_states] = kind;
        return (int8_t)pl.n_states++;
    };
    for (int a = 0; a < n_aggs; a++) {
        int c = aggs[a].col, op = aggs[a].op;
        if (c < 0 || c >= n_vals) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "aggregate %d: column %d out of range", a, c);
        if (op < 0 || op > PANDRS_HIP_AGG_NUNIQUE) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bad op %d", op);
        pl.fin_op[a] = (int8_t)op; pl.fin_kind[a] = 0; pl.fin_src[a] = -1;
        if (op == PANDRS_HIP_AGG_COUNT) continue;   // any dtype (aggregation.rs:743)
        if (op == PANDRS_HIP_AGG_CUSTOM)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED,
                        "Custom aggregation requires a custom function, use aggregate_custom instead");
        int dt = val_dtypes[c];
        if (dt != PANDRS_HIP_I64 && dt != PANDRS_HIP_F64)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED,
                        "Aggregation operation %d is not supported for column type %d", op, dt);
        if (is_sorted_pass_op(op)) {            // Median / Nunique: no engine state, filled by median_pass after the run
            pl.fin_kind[a] = dt == PANDRS_HIP_F64 ? 0 : 1;
            pl.has_median = true;
            continue;
        }
        int s = src_of[c];
        if (s < 0) {
            if (pl.n_src >= MAX_SRC) return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "more than %d aggregated columns", MAX_SRC);
            s = src_of[c] = pl.n_src++;
            pl.src_col[s] = c;
            pl.src_kind[s] = dt == PANDRS_HIP_F64 ? 0 : 1;
            pl.st_add[s] = pl.st_min[s] = pl.st_max[s] = pl.st_nn[s] = pl.st_fadd[s] = pl.st_ssq[s] = -1;
        }
        const bool f64 = dt == PANDRS_HIP_F64;
        pl.fin_kind[a] = f64 ? 0 : 1;
        pl.fin_src[a] = s;
        if ((op == PANDRS_HIP_AGG_SUM || op == PANDRS_HIP_AGG_MEAN) && pl.st_add[s] < 0)
            pl.st_add[s] = new_state(f64 ? SK_ADD_F64 : SK_ADD_I64);
        if (op == PANDRS_HIP_AGG_MEAN && val_has_nulls[c] && pl.st_nn[s] < 0)
            pl.st_nn[s] = new_state(SK_ADD_I64);
        if (op == PANDRS_HIP_AGG_MIN && pl.st_min[s] < 0)
            pl.st_min[s] = new_state(f64 ? SK_MIN_F64 : SK_MIN_I64);
        if (op == PANDRS_HIP_AGG_MAX && pl.st_max[s] < 0)
            pl.st_max[s] = new_state(f64 ? SK_MAX_F64 : SK_MAX_I64);
        if (op == PANDRS_HIP_AGG_STD || op == PANDRS_HIP_AGG_VAR) {
            // two-pass variance over the non-null values as f64 (aggregation.rs:557-584, :675-702)
            if (f64 && pl.st_add[s] < 0) pl.st_add[s] = new_state(SK_ADD_F64);
            if (!f64 && pl.st_fadd[s] < 0) pl.st_fadd[s] = new_state(SK_ADD_F64);
            if (val_has_nulls[c] && pl.st_nn[s] < 0) pl.st_nn[s] = new_state(SK_ADD_I64);
            if (pl.st_ssq[s] < 0) pl.st_ssq[s] = new_state(SK_ADD_F64);
            pl.needs_second_pass = true; pl.mergeable = false;
        }
        if (op == PANDRS_HIP_AGG_FIRST || op == PANDRS_HIP_AGG_LAST) {
            if (pl.st_firstrow < 0) { pl.st_firstrow = new_state(SK_MIN_I64); pl.st_lastrow = new_state(SK_MAX_I64); }
            pl.mergeable = false;
        }
        if (too_many) return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "more than %d aggregate states", MAX_STATES);
    }
    pl.n_fin = n_aggs;
    return 0;
}

template <typename K>
static int32_t set_max_lds(K kernel, int bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

template <int NSRC, int PROFILE>
static void launch_aggregate_one(pandrs_hip_ctx *c, const AggArgs &a, size_t lds) {
    if (PROFILE >= 0 && c->clustered_rows && !a.direct) {       // rows clustered by key: fold runs inside the wave first
        (void)set_max_lds(aggregate_kernel<NSRC, PROFILE, (PROFILE >= 0)>, (int)lds);
        hipLaunchKernelGGL((aggregate_kernel<NSRC, PROFILE, (PROFILE >= 0)>), dim3(a.launch_grid), dim3(AG_THREADS), lds, c->stream, a);
        return;
    }
    (void)set_max_lds(aggregate_kernel<NSRC, PROFILE>, (int)lds);
    hipLaunchKernelGGL((aggregate_kernel<NSRC, PROFILE>), dim3(a.launch_grid), dim3(AG_THREADS), lds, c->stream, a);
}
template <int NSRC>
static bool launch_aggregate_profile(pandrs_hip_ctx *c, const AggArgs &a, int profile, size_t lds) {
    // instantiated uniform profiles: {f64, i64} x {sum only, sum+min+max} x {no validity, validity}
    switch (profile) {
#define PROF(K, OPS, V) case ((K) << 4 | (OPS) << 1 | (V)): launch_aggregate_one<NSRC, ((K) << 4 | (OPS) << 1 | (V))>(c, a, lds); return true;
        PROF(0, 1, 0) PROF(0, 1, 1) PROF(0, 7, 0) PROF(0, 7, 1) PROF(0, 6, 0) PROF(0, 6, 1)
        PROF(0, 2, 0) PROF(0, 4, 0) PROF(0, 3, 0) PROF(0, 5, 0)
        PROF(1, 1, 0) PROF(1, 1, 1) PROF(1, 7, 0) PROF(1, 7, 1) PROF(1, 6, 0)
#undef PROF
    default: return false;
    }
}
static void launch_aggregate(pandrs_hip_ctx *c, const AggArgs &a, int max_src_per_round, int profile, size_t lds) {
    if (profile >= 0) {
        bool done = false;
        switch (max_src_per_round) {
        case 1: done = launch_aggregate_profile<1>(c, a, profile, lds); break;
        case 2: done = launch_aggregate_profile<2>(c, a, profile, lds); break;
        case 4: done = launch_aggregate_profile<4>(c, a, profile, lds); break;
        }
        if (done) return;
    }
    if (a.pgsize && !a.direct && !a.second_pass && (max_src_per_round > 4 || c->opt.generic_aggregate < 0)) {
        // partial records with many states: the lean one-record-per-thread loop
        (void)set_max_lds(aggregate_kernel<0, -1, false, true>, (int)lds);
        hipLaunchKernelGGL((aggregate_kernel<0, -1, false, true>), dim3(a.launch_grid), dim3(AG_THREADS), lds, c->stream, a);
        return;
    }
    switch (max_src_per_round) {
    case 0:     // count-only: no value sources
    case 1: launch_aggregate_one<1, -1>(c, a, lds); break;
    case 2: launch_aggregate_one<2, -1>(c, a, lds); break;
    case 3: launch_aggregate_one<3, -1>(c, a, lds); break;
    case 4: launch_aggregate_one<4, -1>(c, a, lds); break;
    default: launch_aggregate_one<0, -1>(c, a, lds);
    }
}

// copies r2's groups (keys row 0, null flags, aggregates or states) behind res's; `cap` = res's stride.
// One launch for all columns (blockIdx.y = column; the last one is the null-flag byte column).
struct AppendArgs {
    uint64_t *dst[MAX_AGGS + 2];
    const uint64_t *src[MAX_AGGS + 2];
    uint8_t *dst_null;
    const uint8_t *src_null;
    uint32_t n, n_cols;
};
__global__ __launch_bounds__(256) void append_groups_kernel(AppendArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    if (blockIdx.y == a.n_cols) a.dst_null[i] = a.src_null[i];
    else a.dst[blockIdx.y][i] = a.src[blockIdx.y][i];
}
static int32_t append_groups(pandrs_hip_ctx *c, GroupbyResult &res, size_t cap, const GroupbyResult &r2,
                             bool partials, const Plan &pl, int n_aggs) {
    const size_t g2 = (size_t)r2.n_groups, at = (size_t)res.n_groups;
    if (g2 == 0) return 0;
    if (at + g2 > cap) return fail(PANDRS_HIP_ERR_COMPUTATION, "nested result has more groups than reserved");
    AppendArgs a{};
    uint32_t nc = 0;
    a.dst[nc] = res.keys + at; a.src[nc++] = r2.keys;
    if (partials) {
        for (size_t k = 0; k < 1 + (size_t)pl.n_states; k++) {
            a.dst[nc] = res.states + k * cap + at; a.src[nc++] = r2.states + k * (size_t)r2.cap;
        }
    } else {
        for (int f = 0; f < n_aggs; f++) {
            a.dst[nc] = reinterpret_cast<uint64_t *>(res.aggs + (size_t)f * cap + at);
            a.src[nc++] = reinterpret_cast<const uint64_t *>(r2.aggs + (size_t)f * (size_t)r2.cap);
        }
    }
    a.dst_null = res.key_null + at; a.src_null = r2.key_null;
    a.n = (uint32_t)g2; a.n_cols = nc;
    hipLaunchKernelGGL(append_groups_kernel, dim3((unsigned)((g2 + 255) / 256), nc + 1), dim3(256), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    res.n_groups += (int64_t)g2;
    return 0;
}

struct EngSrc;
static int32_t run_two_level(pandrs_hip_ctx *c, const RowSource &rs, const Plan &pl, bool merge, bool partials,
                             int n_aggs, int key_dtype, int n_keys_out, int res_slot, int64_t est, int64_t groups_per_run);

// One engine source = one partitioned 8-byte column feeding 1..4 states.
struct EngSrc {
    const void *data = nullptr;        // un-partitioned input column
    const uint8_t *null_bits = nullptr;
    const uint8_t *valid_bytes = nullptr;
    int8_t kind = 0;
    int8_t st_add = -1, st_min = -1, st_max = -1, st_nn = -1, st_fadd = -1, st_ssq = -1;   // absolute state ids
    bool rowidx = false;               // synthetic source: the original row index (First / Last)
    int n_states() const { return (st_add >= 0) + (st_min >= 0) + (st_max >= 0) + (st_nn >= 0) + (st_fadd >= 0) + (st_ssq >= 0); }
};

int64_t lean_table_slots(const pandrs_hip_ctx *c, int round_states) {
    const size_t slot_bytes = 13 + 8 * (size_t)round_states;
    int64_t T = (int64_t)(((size_t)c->lds_bytes - 512 - 192 - AGG2_LDS_EXTRA) / slot_bytes) - 3;
    return std::min<int64_t>(T, 32768) & ~int64_t(15);
}

constexpr int32_t SMALL_NOT_TAKEN = -1000;
constexpr int64_t SMALL_MAX_ROWS = int64_t(1) << 21;
constexpr int SMALL_MAX_STATES = 8;

// Small calls: aggregate2's SMALL mode folds row chunks of the ORIGINAL columns into LDS tables and flushes their
// groups into a context-owned global table (<= 16 K groups); small_output_kernel turns that table into result rows and
// re-arms it.  Returns SMALL_NOT_TAKEN when the call does not qualify or the table overflowed (the caller goes on with
// the general path; nothing is left behind).
int32_t run_small(pandrs_hip_ctx *c, const RowSource &rs, const Plan &pl, const std::vector<EngSrc> &srcs, int n_aggs,
                  int n_keys_out, GroupbyResult &res, Arena &rarena) {
    const int64_t N = rs.n_rows;
    const int n_src = (int)srcs.size();
    // a call that outgrew the tables paid for the attempt: sit out a growing number of the following calls
    if (c->small_skip > 0) { c->small_skip--; return SMALL_NOT_TAKEN; }
    if (c->opt.no_small || N > SMALL_MAX_ROWS || n_src < 1 || !pl.mergeable || pl.needs_second_pass || c->opt.generic_aggregate ||
        c->opt.agg_v1 || c->opt.deterministic)
        return SMALL_NOT_TAKEN;
    auto prof_of = [](const EngSrc &e) {
        int ops = (e.st_add >= 0 ? 1 : 0) | (e.st_min >= 0 ? 2 : 0) | (e.st_max >= 0 ? 4 : 0);
        return (e.kind << 4) | (ops << 1) | (e.null_bits ? 1 : 0);
    };
    const int profile = prof_of(srcs[0]);
    int round_states = 0;
    for (auto &e : srcs) {
        if (e.rowidx || e.valid_bytes || !e.data || prof_of(e) != profile) return SMALL_NOT_TAKEN;
        round_states += e.n_states();
    }
    if (!aggregate2_small_has(n_src, profile) || round_states > SMALL_MAX_STATES) return SMALL_NOT_TAKEN;

    const uint32_t S = SMALL_G_SLOTS;
    const size_t GS = (size_t)S + 2;
    if (!c->small_table) {
        const size_t bytes = GS * 8 + GS * 8 * SMALL_MAX_STATES + GS * 4 + 64 * 4;
        HIP_TRY(hipMalloc(&c->small_table, bytes));
        HIP_TRY(hipMemsetAsync(c->small_table, 0, bytes, c->stream));
        HIP_TRY(hipMemsetAsync(c->small_table, 0xFF, GS * 8, c->stream));       // keys: EMPTY_KEY
    }
    uint64_t *g_keys = reinterpret_cast<uint64_t *>(c->small_table);
    uint64_t *g_states = g_keys + GS;
    uint32_t *g_cnt = reinterpret_cast<uint32_t *>(g_states + GS * SMALL_MAX_STATES);
    uint32_t *counters = g_cnt + GS;

    const size_t slot_bytes = 13 + 8 * (size_t)round_states;
    int64_t T = (int64_t)(((size_t)c->lds_bytes - 512 - 192 - AGG2_LDS_EXTRA) / slot_bytes) - 3;
    T = std::min<int64_t>(T, 32768) & ~int64_t(15);
    if (T < 64) return SMALL_NOT_TAKEN;

    const size_t cap = (size_t)std::min<int64_t>(N, (int64_t)GS);
    ST_TRY(rarena.ensure((size_t)n_keys_out * (Arena::padded(cap * 8) + Arena::padded(cap)) +
                         std::max<size_t>(n_aggs, 1) * Arena::padded(cap * 8 + 256) + 8192, c->stream));
    res.cap = (int64_t)cap;
    res.keys = rarena.take<uint64_t>(cap * n_keys_out);
    res.key_null = rarena.take<uint8_t>(cap * n_keys_out);
    res.aggs = rarena.take<double>(cap * std::max<size_t>(n_aggs, 1) + 32);
    if (!res.keys || !res.key_null || !res.aggs) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "result arena too small");

    AggArgs aa{};
    int8_t st_lds[MAX_STATES];
    for (int k = 0; k < MAX_STATES; k++) st_lds[k] = -1;
    const int mm = ((profile >> 2) & 1) + ((profile >> 3) & 1);
    const int mbase = ((profile >> 1) & 1) ? n_src : 0;
    int next_nn = mbase + n_src * mm;
    for (int s = 0; s < n_src; s++) {
        const EngSrc &e = srcs[s];
        SrcDev &sd = aa.src[s];
        sd = SrcDev{static_cast<const uint64_t *>(e.data), e.null_bits, e.kind, -1, -1, -1, -1, -1, -1, {0}};     // SMALL: valid = the column's null bitmap
        auto put = [&](int8_t abs_id, int8_t &lds_id, int at) {
            if (abs_id < 0) return;
            st_lds[abs_id] = (int8_t)at; lds_id = (int8_t)at;
        };
        put(e.st_add, sd.st_add, s);
        put(e.st_min, sd.st_min, mbase + s * mm);
        put(e.st_max, sd.st_max, mbase + s * mm + mm - 1);
        if (e.st_nn >= 0) put(e.st_nn, sd.st_nn, next_nn++);
    }
    aa.dkey = rs.key; aa.s_rows = (uint32_t)N;
    // rows per workgroup: every workgroup flushes its groups with global atomics, so more than ~64 of them cost more in
    // the flush than they gain in the stream (experiments/c1_chunks.py: 1 M rows / 1 K groups 59 us at 16 K rows per
    // workgroup, 79 us at 4 K; 100 K rows / 100 groups 40 us at 4 K, 46 us at 16 K)
    const int64_t chunk = c->opt.small_chunk > 0 ? c->opt.small_chunk : std::max<int64_t>(4096, (N + 63) / 64);
    aa.s_chunk = (uint32_t)((chunk + 1023) / 1024 * 1024);
    aa.g_slots = S; aa.g_keys = g_keys; aa.g_cnt = g_cnt; aa.g_states = g_states;
    aa.T = (uint32_t)T; aa.seed = 0x9E3779B9u; aa.n_src = n_src; aa.n_states = pl.n_states;
    aa.n_fin = n_aggs; aa.partials = 0; aa.n_rounds = 1; aa.round_states = round_states; aa.second_pass = 0;
    std::memcpy(aa.kinds, pl.kinds, sizeof aa.kinds);
    for (int f = 0; f < n_aggs; f++) {
        FinDev &fd = aa.fin[f];
        fd = FinDev{};
        fd.op = pl.fin_op[f]; fd.kind = pl.fin_kind[f];
        fd.st_add = fd.st_nn = fd.st_min = fd.st_max = fd.st_ssq = fd.st_fadd = fd.rowsrc_min = fd.rowsrc_max = -1;
        const int s = pl.fin_src[f];
        if (s < 0) { aa.s_need_cnt = 1; continue; }
        auto lds_of = [&](int8_t abs_id) -> int8_t { return abs_id < 0 ? (int8_t)-1 : st_lds[abs_id]; };
        fd.st_add = lds_of(pl.st_add[s]); fd.st_nn = lds_of(pl.st_nn[s]);
        fd.st_min = lds_of(pl.st_min[s]); fd.st_max = lds_of(pl.st_max[s]);
        if (fd.op == PANDRS_HIP_AGG_COUNT || (fd.op == PANDRS_HIP_AGG_MEAN && fd.st_nn < 0)) aa.s_need_cnt = 1;
        if (fd.op != PANDRS_HIP_AGG_COUNT && fd.op != PANDRS_HIP_AGG_SUM && fd.op != PANDRS_HIP_AGG_MEAN &&
            fd.op != PANDRS_HIP_AGG_MIN && fd.op != PANDRS_HIP_AGG_MAX)
            return SMALL_NOT_TAKEN;
    }
    aa.out_keys = res.keys; aa.out_null = res.key_null; aa.out_aggs = res.aggs; aa.cap = cap; aa.counters = counters;
    volatile uint32_t *hp = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1040;
    hp[4] = 0;
    aa.host_out = nullptr;                       // the fold publishes nothing: the output kernel does
    const uint32_t n_tables = (aa.s_rows + aa.s_chunk - 1) / aa.s_chunk;
    aa.launch_grid = n_tables;
    const size_t lds = (size_t)(T + 3) * slot_bytes + 192 + AGG2_LDS_EXTRA;
    // (no phase events: each one is a barrier packet between two 10-30 us kernels)
    if (!launch_aggregate2_small(c, aa, n_src, profile, lds, std::min<uint32_t>((uint32_t)c->n_cu, n_tables)))
        return SMALL_NOT_TAKEN;
    aa.host_out = const_cast<uint32_t *>(hp);
    launch_small_output(c, aa, n_src, profile);
    HIP_TRY(hipGetLastError());
    for (int spin = 0; spin < 4000000 && hp[4] != 1; spin++) __builtin_ia32_pause();
    if (hp[4] != 1) HIP_TRY(hipStreamSynchronize(c->stream));
    if (hp[4] != 1) return fail(PANDRS_HIP_ERR_COMPUTATION, "small path: no completion record");
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    c->timings.n_partitions = 0; c->timings.table_slots = T; c->timings.retries = 0; c->timings.estimated_groups = hp[0];
    if (hp[1]) {                                // more groups than the tables hold: the global table is armed again, nothing kept
        c->small_backoff = std::min(c->small_backoff ? c->small_backoff * 2 : 4, 256);
        c->small_skip = c->small_backoff;
        return SMALL_NOT_TAKEN;
    }
    c->small_backoff = 0;
    res.n_groups = hp[0]; res.valid = true;
    c->timings_lazy = true;
    return 0;
}

// Core: groups rs by key and reduces the plan's states.  Result retained in c->gb.
// ---- hot-key absorb-and-spill (kernels: absorb.hip) ------------------------------------------------------------------
// One pass over the ORIGINAL columns folds the rows of the keys that found a slot in a workgroup's LDS table; the other
// rows are spilled straight into P_s radix partitions (per-workgroup regions), the lean aggregate folds those regions
// and appends its groups as partial records behind the absorbed ones, and one merge finishes.  No host round trip
// between the absorb pass and the spill aggregate.  ABSORB_NOT_TAKEN: the caller continues with the ordinary path.
constexpr int32_t ABSORB_NOT_TAKEN = -1001;
static int64_t absorb_table_slots(const pandrs_hip_ctx *c, int lds_states) {
    return ((int64_t)(((size_t)c->lds_bytes - 512 - 640) / (12 + 8 * (size_t)lds_states)) - 2) & ~int64_t(3);
}
static int32_t run_absorb(pandrs_hip_ctx *c, const RowSource &rs, const Plan &pl, const std::vector<EngSrc> &srcs, int profile, int64_t T,
                          int64_t est, bool partials, int n_aggs, int key_dtype, int n_keys_out, int res_slot, const uint64_t *hot_image) {
    const int64_t N = rs.n_rows;
    const int n_src = (int)srcs.size();
    const bool has_v = (profile & 1) != 0;
    const uint32_t n_wg = (uint32_t)std::max(c->n_cu, 1);
    const uint32_t chunk = (uint32_t)(((N + n_wg - 1) / n_wg + 15) & ~int64_t(15));
    // the spill side: the lean aggregate's table geometry decides the fan-out of the spilled rows
    int round_states = 0;
    for (auto &e : srcs) round_states += (e.st_add >= 0) + (e.st_min >= 0) + (e.st_max >= 0) + (e.st_nn >= 0);
    const int64_t T2 = lean_table_slots(c, round_states);
    if (T2 < 64) return ABSORB_NOT_TAKEN;
    int64_t PS = 1;
    while ((double)est / (double)PS > (double)T2 * 0.60 && PS < 64) PS *= 2;
    // more groups behind the hot keys than 64 spill partitions of LDS tables hold (a long tail): COMPACT spill — the rows the tables
    // do not take are appended to one buffer, the ordinary engine groups them (partial states), and one merge joins both halves
    const bool compact = (double)est / (double)PS > (double)T2 * 0.80;
    if (compact) PS = 1;                          // one region per workgroup, sized for all of its rows; closed up by compact_spill_kernel
    if (compact && (res_slot != 0 || !hot_image || N >= (int64_t(1) << 32) - (int64_t(1) << 22))) return ABSORB_NOT_TAKEN;
    const uint32_t cap_wp = compact ? (uint32_t)((chunk + 15) & ~15u) : (uint32_t)((((int64_t)chunk / PS) * 2 + 256 + 15) & ~int64_t(15));
    const size_t region_rows = (size_t)n_wg * (size_t)PS * cap_wp;
    if (region_rows >= (size_t(1) << 32) - (size_t(1) << 20)) return ABSORB_NOT_TAKEN;
    // tables over the regions: one per CU (two measured 3 % slower: a table's fixed cost), each fed by `wpt` consecutive workgroups' regions of one partition
    uint32_t tpp = std::max<uint32_t>(1, std::min<uint32_t>(n_wg, n_wg / (uint32_t)PS));
    while ((uint64_t)tpp * (uint64_t)PS > 1024) tpp--;
    const uint32_t wpt = (n_wg + tpp - 1) / tpp;
    const uint32_t max_tables = (uint32_t)PS * ((n_wg + wpt - 1) / wpt), max_tasks = (uint32_t)PS * n_wg;
    const size_t n_state = 1 + (size_t)pl.n_states;
    const size_t dcap = (size_t)n_wg * (size_t)(T + 2) + (size_t)max_tables * (size_t)(T2 + 2);
    ST_TRY(c->temp.ensure(Arena::padded(dcap * 8) + Arena::padded(dcap) + n_state * Arena::padded(dcap * 8 + 256) + Arena::padded((size_t)max_tasks * 16 + 256) +
                          Arena::padded((size_t)max_tables * 16 + 256) + Arena::padded((size_t)n_wg * PS * 4 + 256) + 65536, c->stream));
    uint64_t *rk = c->temp.take<uint64_t>(dcap);
    uint8_t *rn = c->temp.take<uint8_t>(dcap);
    uint64_t *rst = c->temp.take<uint64_t>(dcap * n_state + 32);
    uint32_t *counters = c->temp.take<uint32_t>(64);
    uint32_t *sp_count = c->temp.take<uint32_t>((size_t)n_wg * PS + 16);
    AggTask *tasks = c->temp.take<AggTask>(max_tasks + 8);
    AggTable *tables = c->temp.take<AggTable>(max_tables + 8);
    uint32_t *n_tasks = c->temp.take<uint32_t>(64);
    if (!rk || !rn || !rst || !counters || !sp_count || !tasks || !tables || !n_tasks) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "temp arena too small (absorb)");
    ST_TRY(c->absorb.ensure((compact ? 2 : 1) * ((1 + (size_t)n_src) * Arena::padded(region_rows * 8 + 256) + (has_v ? (size_t)n_src * Arena::padded(region_rows + 256) : 0)) + (1 << 20), c->stream));
    AbsorbArgs a{};
    a.key = rs.key; a.n_rows = (uint32_t)N; a.chunk = chunk; a.T = (uint32_t)T; a.seed = ABSORB_SEED; a.n_src = n_src; a.hot_image = hot_image; a.image_only = compact ? 1 : 0;
    a.fold = c->est_near_same > 0.3 ? 1 : 0;       // (a key with > 55 % of the rows, or long runs: 40 of a wave's 64 lanes in one slot is likely)
    a.spill_P = (uint32_t)PS; a.spill_cap = cap_wp;
    a.sp_keys = c->absorb.take<uint64_t>(region_rows + 16);
    AggArgs aa{};
    int next = 0;
    // the lean aggregate's fixed LDS state order (run_engine): adds of source 0..n-1, per source its min-type states, then the counts
    const int v2_mm = ((profile >> 2) & 1) + ((profile >> 3) & 1), v2_mbase = ((profile >> 1) & 1) ? n_src : 0;
    int v2_next_nn = v2_mbase + n_src * v2_mm;
    for (int k = 0; k < MAX_STATES; k++) { aa.st_round[k] = -1; aa.st_lds[k] = -1; }
    for (int s = 0; s < n_src; s++) {
        const EngSrc &e = srcs[s];
        a.vals[s] = reinterpret_cast<const uint64_t *>(e.data);
        a.null_bits[s] = e.null_bits;
        a.sp_vals[s] = c->absorb.take<uint64_t>(region_rows + 16);
        a.sp_valid[s] = has_v ? c->absorb.take<uint8_t>(region_rows + 16) : nullptr;
        if (!a.sp_vals[s] || (has_v && !a.sp_valid[s])) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "absorb arena too small");
        auto place = [&](int8_t abs_id, int8_t &lds_id) {
            lds_id = -1;
            if (abs_id < 0) return;
            lds_id = (int8_t)next; a.lds_abs[next] = abs_id; a.lds_kind[next] = pl.kinds[abs_id]; next++;
        };
        place(e.st_add, a.st_add[s]); place(e.st_min, a.st_min[s]); place(e.st_max, a.st_max[s]); place(e.st_nn, a.st_nn[s]);
        SrcDev &sd = aa.src[s];
        sd = SrcDev{a.sp_vals[s], a.sp_valid[s], e.kind, -1, -1, -1, -1, -1, -1, {0}};
        auto put = [&](int8_t abs_id, int8_t &lds_id, int at) {
            if (abs_id < 0) return;
            aa.st_round[abs_id] = 0; aa.st_lds[abs_id] = (int8_t)at; lds_id = (int8_t)at;
        };
        put(e.st_add, sd.st_add, s);
        put(e.st_min, sd.st_min, v2_mbase + s * v2_mm);
        put(e.st_max, sd.st_max, v2_mbase + s * v2_mm + v2_mm - 1);
        if (e.st_nn >= 0) put(e.st_nn, sd.st_nn, v2_next_nn++);
    }
    if (!a.sp_keys) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "absorb arena too small");
    a.n_lds_states = next;
    a.out_keys = rk; a.out_null = rn; a.out_states = rst; a.cap = dcap;
    a.counters = counters; a.sp_count = sp_count;
    HIP_TRY(hipMemsetAsync(counters, 0, 256, c->stream));
    const size_t lds = (size_t)(T + 2) * (12 + 8 * (size_t)next) + 16 + 96 * 4 + 64;
    aa.pkeys = a.sp_keys; aa.P = (uint32_t)PS; aa.T = (uint32_t)T2; aa.seed = a.seed; aa.n_src = n_src; aa.n_states = pl.n_states;
    aa.n_fin = 0; aa.partials = 1; aa.n_rounds = 1; aa.round_states = round_states; aa.second_pass = 0;
    aa.round_src_begin[0] = 0; aa.round_src_begin[1] = (int8_t)n_src;
    std::memcpy(aa.kinds, pl.kinds, sizeof aa.kinds);
    aa.out_keys = rk; aa.out_null = rn; aa.out_states = rst; aa.cap = dcap;
    aa.side_keys = rk; aa.side_null = rn; aa.side_states = rst; aa.side_cap = dcap;        // every table is `multi`: side = the record buffer
    aa.counters = counters; aa.tasks = tasks; aa.tables = tables; aa.n_tasks = n_tasks; aa.launch_grid = max_tables;
    volatile uint32_t *hp = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1040;
    hp[4] = 0;
    aa.host_out = const_cast<uint32_t *>(hp); aa.scatter_flags = nullptr;
    const size_t lds2 = (size_t)(T2 + 3) * (13 + 8 * (size_t)round_states) + 192 + AGG2_LDS_EXTRA;
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_AGGREGATE);
        if (!launch_absorb(c, a, n_src, profile, lds, n_wg)) { (void)hipGetLastError(); return ABSORB_NOT_TAKEN; }
        if (!compact) {
            launch_build_spill_tables(c, sp_count, n_wg, (uint32_t)PS, cap_wp, wpt, tasks, tables, n_tasks);
            if (!launch_aggregate2(c, aa, n_src, profile, lds2, (uint32_t)std::min<int64_t>(c->n_cu, max_tables)))
                return fail(PANDRS_HIP_ERR_COMPUTATION, "absorb: the lean aggregate has no instantiation for this profile");
        }
        HIP_TRY(hipGetLastError());
    }
    uint32_t h4[4] = {0, 0, 0, 0};
    int64_t n_compact = 0;
    if (compact) {
        // the spilled rows, closed up, through the ordinary engine (partial states), appended behind the absorbed records
        uint64_t *ck = c->absorb.take<uint64_t>(region_rows + 16), *cv[MAX_ABS_SRC]{};
        uint8_t *cvalid[MAX_ABS_SRC]{};
        for (int s2 = 0; s2 < n_src; s2++) {
            cv[s2] = c->absorb.take<uint64_t>(region_rows + 16);
            cvalid[s2] = has_v ? c->absorb.take<uint8_t>(region_rows + 16) : nullptr;
            if (!cv[s2] || (has_v && !cvalid[s2])) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "absorb arena too small (compact spill)");
        }
        if (!ck) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "absorb arena too small (compact spill)");
        launch_compact_spill(c, a, n_wg, ck, cv, cvalid, has_v, counters + 4);
        HIP_TRY(hipGetLastError());
        uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
        HIP_TRY(hipMemcpyAsync(h, counters, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h[1]) return ABSORB_NOT_TAKEN;
        const int64_t n_abs = h[2];
        n_compact = h[4];
        h4[2] = (uint32_t)n_abs;
        if (n_compact > 0) {
            // The tables of this mode take no keys beyond the image (AbsorbArgs::image_only): every workgroup absorbs the SAME keys, so
            // the spilled rows' keys are disjoint from the absorbed ones.  (1) group the spilled rows with the ordinary engine AS THE
            // RESULT of this call, with room reserved behind its groups; (2) merge the absorbed records (a few hundred thousand) in a
            // nested run; (3) append (2)'s groups.  The order matters: run (1) may itself nest into slot res_slot + 1 (a slice merge of
            // an oversized partition, the overflow run of an underestimated tail), so slot res_slot + 1 must not hold anything yet.
            // The absorbed records live in c->temp, which no run below touches (no_direct, no_absorb).
            // (Merging the spilled rows' groups as partial records with the absorbed ones cost 11 ms for a 16 M-group tail.)
            Options saved = c->opt;
            pandrs_hip_timings tsave = c->timings;
            c->opt.no_direct = 1; c->opt.no_absorb = 1; c->opt.partitions = 0;
            c->quiet++;
            const int64_t hot_bound = std::min<int64_t>((int64_t)T + 2, n_abs);     // image keys + the NULL key + the sentinel-valued key
            int32_t st2;
            {
                RowSource sp;
                sp.n_rows = n_compact;
                sp.key = KeyDesc{ck, nullptr, nullptr, DT_CELL};          // (NULL keys and the sentinel-valued key are always absorbed)
                for (int s2 = 0; s2 < n_src; s2++) {
                    sp.val_data[s2] = cv[s2];
                    sp.val_null_bits[s2] = nullptr;
                    sp.val_valid_bytes[s2] = has_v ? cvalid[s2] : nullptr;
                }
                c->opt.groups_hint = std::max<int64_t>(c->opt.tail_groups_hint, 0);   // (0:) its own estimate: the tail's cardinality is what the first one could not see
                c->reserve_groups = hot_bound;
                st2 = run_engine(c, sp, pl, /*merge=*/false, partials, n_aggs, key_dtype, n_keys_out, res_slot);
                c->reserve_groups = 0;
            }
            GroupbyResult &r2 = c->gb2;
            int64_t n_hot = 0;
            if (!st2 && n_abs > 0) {
                RowSource ms;
                ms.n_rows = n_abs;
                ms.key = KeyDesc{rk, nullptr, rn, DT_CELL};
                ms.merge_states = rst;
                ms.merge_stride = dcap;
                c->opt.groups_hint = std::max<int64_t>(hot_bound, 1);
                st2 = run_engine(c, ms, pl, /*merge=*/true, partials, n_aggs, key_dtype, 1, res_slot + 1);
                if (!st2) n_hot = r2.n_groups;
            }
            c->quiet--;
            c->opt = saved;
            pandrs_hip_timings tnested = c->timings;
            c->timings = tsave;
            if (st2) {
                // the tail alone holds more groups than one radix level takes (a nested run cannot go two-level): the ordinary path answers
                if (c->capacity_exceeded) { c->capacity_exceeded = false; return ABSORB_NOT_TAKEN; }
                return st2;
            }
            GroupbyResult &res = c->gb;
            if (n_hot > 0) {
                ST_TRY(append_groups(c, res, (size_t)res.cap, r2, partials, pl, n_aggs));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            (void)tnested;
            c->timings.estimated_groups = est;
            c->timings.absorbed_rows = N - n_compact;
            c->timings.n_partitions = -1;
            c->timings.table_slots = T;
            c->timings.retries = 0;
            return 0;
        }
    } else {
    for (int spin = 0; spin < 4000000 && hp[4] != 1; spin++) __builtin_ia32_pause();
    if (hp[4] != 1) HIP_TRY(hipStreamSynchronize(c->stream));
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (hp[4] == 1) { for (int i = 0; i < 3; i++) h4[i] = hp[i]; }
    else {
        uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
        HIP_TRY(hipMemcpyAsync(h, counters, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 3; i++) h4[i] = h[i];
    }
    }
    if (h4[1]) return ABSORB_NOT_TAKEN;            // a spill region or a spill table overflowed: the ordinary path answers
    uint32_t *hs = reinterpret_cast<uint32_t *>(c->pinned) + 1100;                           // own corner: the merge below reads its counters at +0
    HIP_TRY(hipMemcpyAsync(hs, counters + (compact ? 4 : 3), 4, hipMemcpyDeviceToHost, c->stream));   // rows spilled: reported, not needed to continue
    // one merge of everything (the direct path's last step)
    RowSource ms;
    ms.n_rows = h4[2];
    ms.key = KeyDesc{rk, nullptr, rn, DT_CELL};
    ms.merge_states = rst;
    ms.merge_stride = dcap;
    Options saved = c->opt;
    c->opt.no_direct = 1; c->opt.no_absorb = 1;
    c->opt.groups_hint = std::max<int64_t>(std::min<int64_t>(est, ms.n_rows), 1);
    pandrs_hip_timings tsave = c->timings;
    c->quiet++;
    const int32_t st = run_engine(c, ms, pl, /*merge=*/true, partials, n_aggs, key_dtype, n_keys_out, res_slot);
    c->quiet--;
    c->opt = saved;
    c->timings = tsave;
    c->timings.estimated_groups = est;
    c->timings.absorbed_rows = N - (int64_t)hs[0];         // (the merge synchronised the stream)
    c->timings.n_partitions = hs[0] ? (compact ? -1 : PS) : 0;   // nothing spilled: no radix partition took part (the few-groups case); -1: compact spill
    c->timings.table_slots = T;
    c->timings.retries = 0;
    // (the merge is a nested run: it cannot go two-level.  More groups among the records than one radix level takes — only with a
    // tiny test level, p_max — hands the call back to the ordinary path instead of failing it)
    if (st && c->capacity_exceeded) { c->capacity_exceeded = false; return ABSORB_NOT_TAKEN; }
    return st;
}

// ---- rows clustered by key (sorted input, input grouped by key, time-ordered keys) ------------------------------------------------
// No partition at all: ONE pass over the ORIGINAL columns.  The rows are cut into chunks short enough that a chunk's runs fit one LDS
// table; clustered_kernel (clustered.hip; any key dtype, null bitmaps in place) folds every thread's consecutive rows in registers and
// touches the table once per run; a chunk's groups leave as partial records, and one merge of the records — groups + chunk
// boundaries + keys that come back in a later run — is the result.  100 M sorted rows, 1 M groups, 12 states: 4.9 ms (scatter + the
// older kernel's RUNS instantiation) -> see experiments/clustered_check.py.  CLUSTERED_NOT_TAKEN: the caller goes on as before
// (a chunk with more runs than the sample promised fills its table; the record buffer is sized for twice the promised runs).
constexpr int32_t CLUSTERED_NOT_TAKEN = -1002;
static int32_t run_clustered(pandrs_hip_ctx *c, const RowSource &rs, const Plan &pl, const std::vector<EngSrc> &srcs, int64_t est, bool partials,
                             int n_aggs, int key_dtype, int n_keys_out, int res_slot) {
    const int64_t N = rs.n_rows;
    const int n_src = (int)srcs.size();
    if (n_src < 1 || n_src > 4 || N >= (int64_t(1) << 32) - (int64_t(1) << 22)) return CLUSTERED_NOT_TAKEN;
    auto prof_of = [](const EngSrc &e) {
        int ops = (e.st_add >= 0 ? 1 : 0) | (e.st_min >= 0 ? 2 : 0) | (e.st_max >= 0 ? 4 : 0);
        return (e.kind << 4) | (ops << 1) | (e.null_bits ? 1 : 0);
    };
    const int profile = prof_of(srcs[0]);
    int round_states = 0;
    for (auto &e : srcs) {
        if (e.rowidx || e.valid_bytes || !e.data || e.st_fadd >= 0 || e.st_ssq >= 0 || prof_of(e) != profile) return CLUSTERED_NOT_TAKEN;
        round_states += e.n_states();
    }
    if (!clustered_has(n_src, profile)) return CLUSTERED_NOT_TAKEN;
    const int64_t T = lean_table_slots(c, round_states);
    if (T < 256) return CLUSTERED_NOT_TAKEN;
    // runs per row from the estimate's adjacent pairs; only long runs pay (a record per run goes through the merge)
    const double runs_per_row = std::max(1.0 - c->est_near_same, 1e-7);
    // (measured up to one run per 8 rows — experiments/clustered_check.py: 100 M rows in runs of 8, 12 states 4.7 -> 4.2 ms, one sum 1.7 -> 0.9)
    if (runs_per_row > (c->opt.clustered_max_runs_pct > 0 ? c->opt.clustered_max_runs_pct / 100.0 : 0.13)) return CLUSTERED_NOT_TAKEN;
    // a chunk's runs fill a third of its table (the sample's figure is an average), and there are chunks enough for every CU four times over
    int64_t chunk = (int64_t)((double)T * 0.33 / runs_per_row);
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(N / (4 * (int64_t)std::max(c->n_cu, 1)), 16384));
    if (c->opt.clustered_chunk > 0) chunk = c->opt.clustered_chunk;
    chunk = std::max<int64_t>((chunk + 1023) / 1024 * 1024, 4096);
    const int64_t n_tables = (N + chunk - 1) / chunk;
    const int64_t runs = (int64_t)((double)N * runs_per_row);
    const size_t dcap = (size_t)std::min<int64_t>(n_tables * (T + 2), 2 * runs + 2 * n_tables + 65536);
    const size_t n_state = 1 + (size_t)pl.n_states;
    // the records are written once and read once more by the merge: when they would be more than 0.6 of the input's bytes (short runs and
    // many states: 12 states at one run per 8 rows measured 4.1 ms against 3.9 for the lean kernel behind the exact partition), or the
    // buffer cannot be had (memory limit), the ordinary path answers
    const size_t rec_bytes = Arena::padded(dcap * 8) + Arena::padded(dcap) + n_state * Arena::padded(dcap * 8 + 256) + 65536;
    // (0.6: sorted rows of 10 M groups, 10 rows each, 12 states — records 0.54 of the input — still win here, 6.8 ms against 9.0)
    if ((double)rec_bytes > 0.6 * (double)N * (double)(8 + 8 * n_src) + (double)(64 << 20)) return CLUSTERED_NOT_TAKEN;
    if (c->temp.ensure(rec_bytes, c->stream) != 0) { (void)hipGetLastError(); return CLUSTERED_NOT_TAKEN; }
    uint64_t *rk = c->temp.take<uint64_t>(dcap);
    uint8_t *rn = c->temp.take<uint8_t>(dcap);
    uint64_t *rst = c->temp.take<uint64_t>(dcap * n_state + 32);
    uint32_t *counters = c->temp.take<uint32_t>(64);
    if (!rk || !rn || !rst || !counters) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "temp arena too small (clustered rows)");
    HIP_TRY(hipMemsetAsync(counters, 0, 256, c->stream));
    AggArgs aa{};
    for (int k = 0; k < MAX_STATES; k++) { aa.st_round[k] = -1; aa.st_lds[k] = -1; }
    // the lean aggregate's fixed LDS state order: adds of source 0..n-1, per source its min-type states, then the counts
    const int mm = ((profile >> 2) & 1) + ((profile >> 3) & 1), mbase = ((profile >> 1) & 1) ? n_src : 0;
    int next_nn = mbase + n_src * mm;
    for (int s = 0; s < n_src; s++) {
        const EngSrc &e = srcs[s];
        SrcDev &sd = aa.src[s];
        sd = SrcDev{static_cast<const uint64_t *>(e.data), e.null_bits, e.kind, -1, -1, -1, -1, -1, -1, {0}};     // SMALL: valid = the column's null bitmap
        auto put = [&](int8_t abs_id, int8_t &lds_id, int at) {
            if (abs_id < 0) return;
            aa.st_round[abs_id] = 0; aa.st_lds[abs_id] = (int8_t)at; lds_id = (int8_t)at;
        };
        put(e.st_add, sd.st_add, s);
        put(e.st_min, sd.st_min, mbase + s * mm);
        put(e.st_max, sd.st_max, mbase + s * mm + mm - 1);
        if (e.st_nn >= 0) put(e.st_nn, sd.st_nn, next_nn++);
    }
    aa.dkey = rs.key; aa.s_rows = (uint32_t)N; aa.s_chunk = (uint32_t)chunk;
    {
        uintptr_t bits = reinterpret_cast<uintptr_t>(rs.key.data);
        for (auto &e : srcs) bits |= reinterpret_cast<uintptr_t>(e.data);
        aa.s_vec = (bits & 15) == 0 ? 1u : 0u;
    }
    aa.T = (uint32_t)T; aa.seed = 0x9E3779B9u; aa.n_src = n_src; aa.n_states = pl.n_states;
    aa.n_fin = 0; aa.partials = 1; aa.n_rounds = 1; aa.round_states = round_states; aa.second_pass = 0;
    aa.round_src_begin[0] = 0; aa.round_src_begin[1] = (int8_t)n_src;
    std::memcpy(aa.kinds, pl.kinds, sizeof aa.kinds);
    aa.out_keys = rk; aa.out_null = rn; aa.out_states = rst; aa.cap = dcap;
    aa.side_keys = rk; aa.side_null = rn; aa.side_states = rst; aa.side_cap = dcap;
    aa.counters = counters; aa.launch_grid = (uint32_t)n_tables;
    volatile uint32_t *hp = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1040;
    hp[4] = 0;
    aa.host_out = const_cast<uint32_t *>(hp); aa.scatter_flags = nullptr;
    const size_t lds = (size_t)(T + 3) * (13 + 8 * (size_t)round_states) + 192 + 64;
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_AGGREGATE);
        if (!launch_clustered(c, aa, n_src, profile, lds, (uint32_t)std::min<int64_t>(c->n_cu, n_tables))) return CLUSTERED_NOT_TAKEN;
        HIP_TRY(hipGetLastError());
    }
    for (int spin = 0; spin < 20000000 && hp[4] != 1; spin++) __builtin_ia32_pause();
    if (hp[4] != 1) HIP_TRY(hipStreamSynchronize(c->stream));
    if (hp[4] != 1) return fail(PANDRS_HIP_ERR_COMPUTATION, "clustered rows: no completion record");
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const uint32_t failed = hp[1], n_rec = hp[2];
    if (failed) return CLUSTERED_NOT_TAKEN;          // a chunk held more runs than its table takes, or the record buffer ran out
    RowSource ms;
    ms.n_rows = n_rec;
    ms.key = KeyDesc{rk, nullptr, rn, DT_CELL};
    ms.merge_states = rst;
    ms.merge_stride = dcap;
    Options saved = c->opt;
    c->opt.no_direct = 1; c->opt.no_absorb = 1;
    c->opt.groups_hint = std::max<int64_t>(std::min<int64_t>(est, ms.n_rows), 1);
    pandrs_hip_timings tsave = c->timings;
    c->quiet++;
    const int32_t st = run_engine(c, ms, pl, /*merge=*/true, partials, n_aggs, key_dtype, n_keys_out, res_slot);
    c->quiet--;
    c->opt = saved;
    c->timings = tsave;
    c->timings.estimated_groups = est;
    c->timings.n_partitions = -2;                    // (reported: no radix partition of the rows; -2 = the clustered-rows pass)
    c->timings.table_slots = T;
    c->timings.retries = 0;
    if (st && c->capacity_exceeded) { c->capacity_exceeded = false; return CLUSTERED_NOT_TAKEN; }
    return st;
}

int32_t run_engine(pandrs_hip_ctx *c, const RowSource &rs, const Plan &pl, bool merge,
                          bool partials, int n_aggs, int key_dtype, int n_keys_out, int res_slot) {
    if (res_slot < 0 || res_slot > 2) return fail(PANDRS_HIP_ERR_COMPUTATION, "engine nesting too deep");
    GroupbyResult &res = res_slot == 0 ? c->gb : (res_slot == 1 ? c->gb2 : c->gb3);
    Arena &rarena = res_slot == 0 ? c->result : (res_slot == 1 ? c->result2 : c->result3);
    res = GroupbyResult{};
    res.n_keys = n_keys_out; res.n_aggs = n_aggs; res.n_state = 1 + pl.n_states; res.partials = partials;
    res.key_dtype = key_dtype;
    const int64_t N = rs.n_rows;
    if (N == 0) { res.valid = true; return 0; }
    if (N >= (int64_t(1) << 32) - SC_TILE_MAX)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "n_rows %lld exceeds the 2^32 per-call limit", (long long)N);

    // ---- engine sources.  merge mode: every partial state column is its own single-op source.
    std::vector<EngSrc> srcs;
    if (merge) {
        for (int s = 0; s < pl.n_states; s++) {
            EngSrc e;
            e.data = rs.merge_cols[s] ? rs.merge_cols[s] : rs.merge_states + (size_t)(s + 1) * rs.merge_stride;
            switch (pl.kinds[s]) {
            case SK_ADD_F64: e.kind = 0; e.st_add = (int8_t)s; break;
            case SK_ADD_I64: e.kind = 1; e.st_add = (int8_t)s; break;
            case SK_MIN_F64: e.kind = 0; e.st_min = (int8_t)s; break;
            case SK_MAX_F64: e.kind = 0; e.st_max = (int8_t)s; break;
            case SK_MIN_I64: e.kind = 1; e.st_min = (int8_t)s; break;
            case SK_MAX_I64: e.kind = 1; e.st_max = (int8_t)s; break;
            }
            srcs.push_back(e);
        }
    } else {
        for (int s = 0; s < pl.n_src; s++) {
            EngSrc e;
            e.data = rs.val_data[s]; e.null_bits = rs.val_null_bits[s]; e.valid_bytes = rs.val_valid_bytes[s]; e.kind = pl.src_kind[s];
            e.st_add = pl.st_add[s]; e.st_min = pl.st_min[s]; e.st_max = pl.st_max[s]; e.st_nn = pl.st_nn[s];
            e.st_fadd = pl.st_fadd[s]; e.st_ssq = pl.st_ssq[s];
            srcs.push_back(e);
        }
        if (pl.st_firstrow >= 0) {
            EngSrc e;
            e.kind = 1; e.rowidx = true; e.st_min = pl.st_firstrow; e.st_max = pl.st_lastrow;
            e.data = rs.row_index;             // nullptr: the row's own index
            srcs.push_back(e);
        }
    }
    // (Inside one call, for the row slices of an oversized partition, these merge after all.  First / Last: min / max of the row
    // index merge like any other state and the value is looked up behind the merge — across shards a row index means nothing.
    // Std / Var: every slice runs the two passes over its own rows, the merge adds the between-slice term — aggregate.hpp, MergeVar.)
    const bool nested_slice_merge = merge && !partials && c->quiet > 0;
    if ((partials || merge) && !pl.mergeable && !nested_slice_merge)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED,
                    "Std/Var/Median/First/Last partial states are not mergeable across shards yet");
    const int n_src = (int)srcs.size();
    // (merges of more than 16 states: the record loop of aggregate_kernel<.., MERGE> walks the states at run time; the catch-all instantiation a
    // Std / Var merge needs keeps register arrays for 16 sources)
    if (n_src > (merge ? MAX_MERGE_SRC : MAX_SRC) || (merge && n_src > MAX_SRC && pl.needs_second_pass))
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "too many states to merge (%d)", n_src);

    // ---- small calls (launch-bound: the reference's 1 M-row case): two launches, no estimate, no partition
    if (!merge && !partials && res_slot == 0 && !c->quiet && c->opt.groups_hint <= 0 && !rs.pre) {
        int32_t st = run_small(c, rs, pl, srcs, n_aggs, n_keys_out, res, rarena);
        if (st != SMALL_NOT_TAKEN) return st;
    }

    // ---- workspace upper bound so that one ensure() covers the whole call (incl. retries)
    // the capacity layout of radix_partition_sampled over-allocates the partitioned columns by <= 25 %
    size_t ws = rs.pre ? engine_workspace_bytes(0, 0, 0) + (size_t(64) << 20)            // the partitioned columns are the producer's
                       : engine_workspace_bytes(N + N / 4 + 131072, 1 + n_src + (merge ? 1 : 0), n_src);
    ST_TRY(c->work.ensure(ws, c->stream));
    int64_t est = rs.pre ? std::max<int64_t>(rs.pre->est_groups, 1) : c->opt.groups_hint;
    // hot-key absorb-and-spill (absorb.hip) is decided from the estimate's own sample: its key table is kept for one
    // more look (how many rows do the most frequent keys hold?) when the call could take that path at all
    int absorb_profile = -1;
    {
        auto prof_of = [](const EngSrc &e) {
            int ops = (e.st_add >= 0 ? 1 : 0) | (e.st_min >= 0 ? 2 : 0) | (e.st_max >= 0 ? 4 : 0);
            return (e.kind << 4) | (ops << 1) | (e.null_bits ? 1 : 0);
        };
        bool ok = !merge && !rs.pre && pl.mergeable && c->opt.no_absorb <= 0 && res_slot == 0 && !c->quiet &&
                  N >= (int64_t(1) << (c->opt.no_absorb < 0 ? 16 : 22)) &&      // (no_absorb = -1, tests / fuzz: small inputs too)
                  n_src >= 1 && n_src <= MAX_ABS_SRC && c->opt.partitions <= 0 && !c->opt.generic_aggregate && !c->opt.deterministic;
        for (auto &e : srcs) ok = ok && !e.valid_bytes && !e.rowidx && e.st_fadd < 0 && e.st_ssq < 0;
        if (ok) {
            absorb_profile = prof_of(srcs[0]);
            for (int s = 1; s < n_src; s++) if (prof_of(srcs[s]) != absorb_profile) absorb_profile = -1;
            if (absorb_profile >= 0 && (!absorb_has(n_src, absorb_profile) || !aggregate2_has(n_src, absorb_profile))) absorb_profile = -1;
        }
    }
    if (est <= 0) ST_TRY(estimate_groups(c, rs.key, N, &est, /*keep_table=*/absorb_profile >= 0));
    else { c->clustered_rows = false; c->clumped_rows = false; c->est_near_same = 0.0; c->est_far_same = 0.0; c->est_far_equal = 0; }          // no sample taken: nothing known about the row order
    c->timings.estimated_groups = est;
    int64_t T_abs = 0;
    bool do_absorb = false;
    const uint64_t *hot_image = nullptr;
    if (c->est_kept) {
        int lds_states = 0;
        for (auto &e : srcs) lds_states += (e.st_add >= 0) + (e.st_min >= 0) + (e.st_max >= 0) + (e.st_nn >= 0);
        T_abs = absorb_table_slots(c, lds_states);
        int total_states = 0;
        for (auto &e : srcs) total_states += e.n_states();
        const int64_t Td = std::min<int64_t>((int64_t)(((size_t)c->lds_bytes - 512 - 192) / (20 + 8 * (size_t)total_states)) - 3, 32768) & ~int64_t(3);
        const bool direct_would = Td >= 64 && est * 2 <= Td && !c->opt.no_direct && (N >= (int64_t(1) << 22) || c->opt.no_direct < 0);
        if (T_abs >= 256 && !c->clustered_rows && !c->opt.no_direct && est * 2 <= T_abs) {
            // every group fits a workgroup's table with room to spare: the absorb pass IS the few-groups direct path (nothing spills),
            // with the leaner kernel (100 M rows, 1 K groups: 0.59 -> 0.45 ms for one sum, 0.89 -> 0.70 for six aggregates)
            estimate_release(c);
            do_absorb = true;
        } else if (T_abs >= 256 && !c->clustered_rows && !direct_would && (est <= 64 * T_abs || (c->est_repeat_share >= 0.5 && res_slot == 0)) &&
                   N >= (int64_t(1) << (c->opt.no_absorb < 0 ? 16 : 24))) {
            // not where the direct path answers, not where the table is a drop in the ocean — unless the sample itself shows a hot set
            // (half its rows on keys sighted three times or more: a long tail behind hot keys, compact spill): absorb when the most
            // frequent keys — as many as the table takes — hold most of the rows
            double share = 0.0;
            // (the same bar behind a long tail — the compact spill.  0.65 kept 95 % of C2's rows on 2 K keys, of which the table holds 1.2 K, on the
            // radix path (4.6 instead of 5.4 ms) but sent a join's pair groupby with 60 % of the pairs in 16 groups + 4 M singleton groups there too: 20 ms instead of 4.9)
            const double min_share = c->opt.no_absorb < 0 ? 0.0 : 0.60;
            ST_TRY(estimate_coverage(c, rs.key, N, (int64_t)((double)T_abs * 0.80), &share, c->opt.no_hot_image ? 0 : T_abs, ABSORB_SEED, min_share, &hot_image));
            do_absorb = share >= min_share;        // (no_absorb = -1, tests: whenever it is possible)
        } else estimate_release(c);
    }
    if (do_absorb) {
        const int32_t st = run_absorb(c, rs, pl, srcs, absorb_profile, T_abs, est, partials, n_aggs, key_dtype, n_keys_out, res_slot, hot_image);
        if (st != ABSORB_NOT_TAKEN) return st;
    }
    if (c->clustered_rows && !merge && !rs.pre && pl.mergeable && !pl.needs_second_pass && res_slot == 0 && !c->quiet && !c->opt.no_clustered &&
        !c->opt.generic_aggregate && !c->opt.agg_v1 && !c->opt.deterministic && c->opt.partitions <= 0 && N >= (int64_t(1) << 20)) {
        const int32_t st = run_clustered(c, rs, pl, srcs, est, partials, n_aggs, key_dtype, n_keys_out, res_slot);
        if (st != CLUSTERED_NOT_TAKEN) return st;
    }

    // ---- low-cardinality direct path: when every group fits one LDS table with room to spare,
    // skip the radix partition altogether.  Each workgroup pre-aggregates a contiguous row range
    // of the ORIGINAL columns (one HBM pass), emits its groups as partial records, and the few
    // records (<= tasks x G) are merged by the normal engine.  Also the cure for one-hot-key
    // inputs (bool keys, a dominant key), where a radix partition would put all rows on one CU.
    // (below a few million rows the whole call is launch-bound and the two-stage direct path loses)
    bool has_valid_bytes = false;
    for (auto &e : srcs) has_valid_bytes |= e.valid_bytes != nullptr;
    // (pl.n_states <= MAX_MERGE_SRC: the records' merge takes every state as a source of its own — 8 columns x sum / min / max = 24 states
    // used to FAIL the call here, "too many states to merge", when a merge took 16)
    if (!merge && !rs.pre && pl.mergeable && !c->opt.no_direct && !has_valid_bytes && n_src <= MAX_SRC && pl.n_states <= MAX_MERGE_SRC &&
        (N >= (int64_t(1) << 22) || c->opt.no_direct < 0)) {
        int total_states = 0;
        for (auto &e : srcs) total_states += e.n_states();
        const size_t sb = 20 + 8 * (size_t)total_states;
        int64_t Td = (int64_t)(((size_t)c->lds_bytes - 512 - 192) / sb) - 3;
        Td = std::min<int64_t>(Td, 32768) & ~int64_t(3);
        if (Td >= 64 && est * 2 <= Td) {
            const uint32_t n_tasks = (uint32_t)std::min<int64_t>(std::max<int64_t>(N / 65536, 1), 1024);
            const uint32_t chunk = (uint32_t)((N + n_tasks - 1) / n_tasks);
            const size_t dcap = (size_t)n_tasks * (size_t)std::min<int64_t>(Td + 2, std::max<int64_t>(est * 4, 64) + 2);
            const size_t n_state = 1 + (size_t)pl.n_states;
            ST_TRY(c->temp.ensure(Arena::padded(dcap * 8) + Arena::padded(dcap) + n_state * Arena::padded(dcap * 8 + 256) + 8192, c->stream));
            uint64_t *rk = c->temp.take<uint64_t>(dcap);
            uint8_t *rn = c->temp.take<uint8_t>(dcap);
            uint64_t *rst = c->temp.take<uint64_t>(dcap * n_state + 32);
            uint32_t *counters = c->temp.take<uint32_t>(64);
            if (!rk || !rn || !rst || !counters) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "temp arena too small");
            HIP_TRY(hipMemsetAsync(counters, 0, 256, c->stream));
            AggArgs aa{};
            aa.direct = 1; aa.dkey = rs.key; aa.d_rows = (uint32_t)N; aa.d_chunk = chunk; aa.launch_grid = n_tasks;
            aa.T = (uint32_t)Td; aa.seed = 0x9E3779B9u; aa.n_src = n_src; aa.n_states = pl.n_states;
            aa.n_fin = 0; aa.partials = 1; aa.n_rounds = 1; aa.round_states = total_states; aa.second_pass = 0;
            aa.round_src_begin[0] = 0; aa.round_src_begin[1] = (int8_t)n_src;
            std::memcpy(aa.kinds, pl.kinds, sizeof aa.kinds);
            int next = 0;
            for (int k = 0; k < MAX_STATES; k++) { aa.st_round[k] = -1; aa.st_lds[k] = -1; }
            for (int sidx = 0; sidx < n_src; sidx++) {
                EngSrc &e = srcs[sidx];
                SrcDev &sd = aa.src[sidx];
                sd = SrcDev{reinterpret_cast<const uint64_t *>(e.data), e.null_bits, e.kind, -1, -1, -1, -1, -1, -1, {0}};
                auto place = [&](int8_t abs_id, int8_t &lds_id) {
                    if (abs_id < 0) return;
                    aa.st_round[abs_id] = 0; aa.st_lds[abs_id] = (int8_t)next; lds_id = (int8_t)next; next++;
                };
                place(e.st_add, sd.st_add); place(e.st_min, sd.st_min); place(e.st_max, sd.st_max); place(e.st_nn, sd.st_nn);
            }
            // capacity per task is bounded by dcap / n_tasks records: a task that finds more groups than that
            // (the estimate was too low) must not write past the buffer -> it raises the overflow flag instead
            aa.out_keys = rk; aa.out_null = rn; aa.out_states = rst; aa.cap = dcap; aa.counters = counters;
            aa.d_task_cap = (uint32_t)(dcap / n_tasks);
            int profile = -1;
            if (n_src > 0 && !c->opt.generic_aggregate) {
                auto prof_of = [](const EngSrc &e) {
                    int ops = (e.st_add >= 0 ? 1 : 0) | (e.st_min >= 0 ? 2 : 0) | (e.st_max >= 0 ? 4 : 0);
                    return (e.kind << 4) | (ops << 1) | (e.null_bits ? 1 : 0);
                };
                profile = prof_of(srcs[0]);
                for (int sidx = 1; sidx < n_src; sidx++) if (prof_of(srcs[sidx]) != profile) profile = -1;
            }
            {
                PhaseTimer pt(c, PANDRS_HIP_PHASE_AGGREGATE);
                launch_aggregate(c, aa, n_src, profile, (size_t)(Td + 3) * sb + 192);
                HIP_TRY(hipGetLastError());
            }
            uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
            HIP_TRY(hipMemcpyAsync(h, counters, 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (h[1] == 0) {
                c->timings.n_partitions = 0; c->timings.table_slots = Td;
                RowSource ms;
                ms.n_rows = h[0];
                ms.key = KeyDesc{rk, nullptr, rn, DT_CELL};
                ms.merge_states = rst;
                ms.merge_stride = dcap;
                Options saved = c->opt;
                c->opt.no_direct = 1;                 // the merge input is tiny; never recurse
                c->opt.groups_hint = std::max<int64_t>(est, 1);   // cardinality is known: no second estimate
                pandrs_hip_timings tsave = c->timings;
                c->quiet++;
                int32_t st = run_engine(c, ms, pl, /*merge=*/true, partials, n_aggs, key_dtype, n_keys_out, res_slot);
                c->quiet--;
                c->opt = saved;
                c->timings = tsave;
                c->timings.estimated_groups = est;
                return st;
            }
            // a task overflowed its table or its record budget: fall through to the partitioned path
        }
    }

    // ---- rounds x table geometry x fan-out.  Fewer sources per round => fewer bytes per slot =>
    // more slots per LDS table => fewer radix partitions (cheaper scatter), at the price of
    // re-reading the partition's keys once per extra round.
    const size_t lds_budget = (size_t)c->lds_bytes - 512;
    const double LOAD = c->opt.load_pct > 0 ? c->opt.load_pct / 100.0 : 0.70;
    int spr = n_src > 0 ? n_src : 1;           // sources per round
    int n_rounds = 1, round_states = 0, max_spr = 0;
    int64_t T = 0, P = 0, auto_slice_rows = 0;
    int8_t round_begin[MAX_ROUNDS + 1];
    // uniform profile: raw rows, every source the same kind / ops / validity (one kernel instantiation)
    auto prof_of = [](const EngSrc &e) {
        int ops = (e.st_add >= 0 ? 1 : 0) | (e.st_min >= 0 ? 2 : 0) | (e.st_max >= 0 ? 4 : 0);
        return (e.kind << 4) | (ops << 1) | ((e.null_bits || e.valid_bytes) ? 1 : 0);
    };
    int uni_profile = -1;
    if (!merge && n_src > 0 && !c->opt.generic_aggregate && pl.mergeable) {
        uni_profile = prof_of(srcs[0]);
        for (int s = 1; s < n_src; s++) if (prof_of(srcs[s]) != uni_profile) uni_profile = -1;
    }
    // Columns of MIXED kinds / op sets (f64 and i64 side by side, sums on some and min / max on others): no uniform profile, but the lean
    // kernel's rounds hand on nothing that depends on the profile (keys, tags, output positions) — so the sources are ordered by profile
    // and every round takes up to 4 sources of ONE profile, with that profile's instantiation.  (Before: the older kernel; 4 f64 + 4 i64
    // columns x sum, 50 M rows, 3.8 ms in its rounds of 4, 8-13 ms with a dominant key — experiments/mixed_wide.py, wide_hot.py.)
    int round_prof[MAX_ROUNDS];
    for (int r = 0; r < MAX_ROUNDS; r++) round_prof[r] = uni_profile;
    bool grouped = false;
    if (uni_profile < 0 && !merge && !rs.pre && !partials && n_src >= 2 && !c->opt.generic_aggregate && pl.mergeable && !pl.needs_second_pass &&
        !c->opt.no_lean_rounds && !c->opt.no_profile_rounds && !c->opt.agg_v1 && c->opt.src_per_round <= 0 && c->opt.partitions <= 0 &&
        (!c->clustered_rows || 1.0 - c->est_near_same > 0.09)) {
        bool all_ok = true;
        for (auto &e : srcs)
            all_ok = all_ok && !e.rowidx && e.data && e.st_fadd < 0 && e.st_ssq < 0 && !(e.null_bits && e.valid_bytes) && aggregate2_has(1, prof_of(e));
        if (all_ok) {
            std::stable_sort(srcs.begin(), srcs.end(), [&](const EngSrc &x, const EngSrc &y) { return prof_of(x) < prof_of(y); });
            grouped = true;
        }
    }
    // the lean persistent kernel (aggregate2.hip): one round of a uniform profile over unclustered raw rows
    // rows clustered in SHORT runs (fewer than ~11 rows: what the one-pass path above does not take or handed back): 3-10 lanes of a wave
    // on one slot cost the lean kernel less than the older kernel's RUNS instantiation costs everywhere else (runs of 4, C2's shape:
    // 3.5 ms lean against 4.9; long runs the other way round: sorted 6.0 against 4.9) — behind the EXACT partition: the sampled region
    // plan assumes random row order
    const bool short_runs = c->clustered_rows && 1.0 - c->est_near_same > 0.09;
    // ROUNDS of the lean kernel (round 4, late): more than 4 uniform columns, or 4 whose states need more tables than the scatter can
    // feed, are folded 4 (2, 1) columns at a time by one launch each over the same partitions; launch 0 leaves every table's keys and
    // output positions behind and the later launches start from them (aggregate.hpp, snap_*).  8 columns x sum / mean / min / max over
    // 1 M groups, 50 M rows: 9.2 ms (the older kernel, one round of 24 states in 764-slot tables) -> see experiments/cliff_hunt.py.
    // Not for partial states, merges and pre-partitioned rows.
    // A DOMINANT key (rows far apart share their key far more often than `est` equally likely keys would: a key with >= ~1 % of the rows)
    // needs its partition cut into row slices.  The lean kernel's rounds cut it (a piece's partial record is filled in round by round at
    // launch 0's position, one merge of up to 39 states at the end); the OLDER kernel's rounds do not — half the rows on one key, 4 f64 +
    // 4 i64 columns x sum, 50 M rows: 248 ms in its rounds (one workgroup walks 25 M rows), 8.2 in one round with slices — so `rounds_ok`
    // keeps it to one round then (experiments/wide_hot.py; what still gets there: Std / Var, First / Last, forced plans).
    const bool dominant_key = c->est_far_equal >= 8 && c->est_far_same * (double)est > 20.0;
    const bool rounds_ok = !(dominant_key && pl.n_states <= MAX_MERGE_SRC) || c->opt.src_per_round > 0;
    const bool lean_rounds_ok = !partials && !merge && !rs.pre && !c->opt.no_lean_rounds;            // (slices work inside the lean kernel's rounds)
    // (long runs too when there are more than 4 columns: the one-pass path takes at most 4, and the burst kernel in rounds behind the exact
    // partition — sorted rows, 8 columns x 4 aggregates, 50 M rows: 3.3 ms — beats the older kernel's 24 states in one table: 9.6)
    const bool wide_clustered = c->clustered_rows && n_src > 4 && clustered_has(4, uni_profile) && !c->opt.no_burst_kernel;
    const bool v2_ok = grouped ? lean_rounds_ok
                     : (uni_profile >= 0 && !pl.needs_second_pass && (!c->clustered_rows || short_runs || wide_clustered) && !c->opt.agg_v1 &&
                        aggregate2_has(std::min(n_src, 4), uni_profile) && (n_src <= 4 || lean_rounds_ok));
    if (v2_ok && n_src > 4) spr = 4;
    // the older kernel with more than 4 sources in a round is its catch-all instantiation (register arrays for 16 sources: it spills):
    // 4 f64 + 4 i64 columns x sum, 50 M rows, 8.3 ms in one round, 3.8 in two (experiments/mixed_wide.py).  Merges have their own loop.
    if (!v2_ok && !merge && n_src > 4 && !pl.needs_second_pass && rounds_ok) spr = 4;
    bool use_v2 = false;
    // rounds only when one round would need more partitions than this.  The older kernel's rounds are dear (3072); the lean kernel's cost
    // one more pass over the key column per round, which a fan-out beyond ~4 K costs the scatter too (experiments/p_target_sweep.py,
    // C2's 12 states, 100 M rows: 5 M uniform groups one round at P = 5120 5.43 ms, two rounds at 2816 4.82; 7 M: 6.05 / 5.37; 4 M: a tie;
    // before the lean kernel had rounds, one lean round up to 8192 was the better plan: 5.9 and 7.3 ms with the older kernel's rounds)
    const int64_t P_TARGET = c->opt.p_target > 0 ? c->opt.p_target : (v2_ok ? (lean_rounds_ok ? 4096 : P_MAX) : 3072);
    if (grouped && v2_ok) {
        // one round per run of up to 4 sources of one profile
        n_rounds = 0; round_states = 0; max_spr = 0;
        for (int b = 0; b < n_src;) {
            const int p = prof_of(srcs[b]);
            int e2 = b, ns = 0;
            while (e2 < n_src && e2 - b < 4 && prof_of(srcs[e2]) == p) { ns += srcs[e2].n_states(); e2++; }
            round_begin[n_rounds] = (int8_t)b; round_prof[n_rounds] = p; n_rounds++; round_begin[n_rounds] = (int8_t)e2;
            round_states = std::max(round_states, ns); max_spr = std::max(max_spr, e2 - b);
            b = e2;
        }
        use_v2 = true;
        const size_t sb = 13 + 8 * (size_t)round_states;
        T = (int64_t)((lds_budget - 192 - AGG2_LDS_EXTRA) / sb) - 3;
        T = std::min<int64_t>(T, 32768) & ~int64_t(15);
        P = (int64_t)std::ceil((double)est / ((double)T * (c->opt.load_pct > 0 ? LOAD : 0.6)));
    } else
    for (;; spr = (spr + 1) / 2) {
        if (c->opt.src_per_round > 0 && !pl.needs_second_pass) spr = (int)std::min<int64_t>(c->opt.src_per_round, std::max(n_src, 1));
        n_rounds = n_src ? (n_src + spr - 1) / spr : 1;
        if (n_rounds > MAX_ROUNDS) { spr = (n_src + MAX_ROUNDS - 1) / MAX_ROUNDS; n_rounds = (n_src + spr - 1) / spr; }
        round_states = 0; max_spr = 0;
        for (int r = 0; r < n_rounds; r++) {
            int b0 = r * spr, b1 = std::min(n_src, b0 + spr), ns = 0;
            round_begin[r] = (int8_t)b0; round_begin[r + 1] = (int8_t)b1;
            for (int s = b0; s < b1; s++) ns += srcs[s].n_states();
            round_states = std::max(round_states, ns);
            max_spr = std::max(max_spr, b1 - b0);
        }
        if (n_src == 0) { round_begin[0] = round_begin[1] = 0; }
        use_v2 = v2_ok && max_spr <= 4 && (n_rounds == 1 || lean_rounds_ok);
        const size_t slot_bytes = (use_v2 ? 13 : 20) + 8 * (size_t)round_states;    // aggregate2: u32 group sizes, one tag byte, no position map
        T = (int64_t)((lds_budget - 192 - (use_v2 ? AGG2_LDS_EXTRA : 0)) / slot_bytes) - 3;
        T = std::min<int64_t>(T, 32768) & (use_v2 ? ~int64_t(15) : ~int64_t(3));   // 16-slot groups / 4-key buckets
        // (the lean kernel's rounds have no overflow run behind a full table: planned at load 0.6, an estimate 35 % too low still fits —
        // at 0.7 one that was 25 % too low cost the attempt: 6 M groups estimated as 4.5 M, 10.9 ms against 6.8 for the single round)
        P = (int64_t)std::ceil((double)est / ((double)T * ((use_v2 && n_rounds > 1 && c->opt.load_pct <= 0) ? 0.6 : LOAD)));
        if (c->opt.src_per_round > 0 || spr <= 1 || P <= P_TARGET || pl.needs_second_pass) break;
    }
    if (T < 64) return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "too many aggregate states for one LDS table");
    const size_t slot_bytes = (use_v2 ? 13 : 20) + 8 * (size_t)round_states;
    if (rs.pre) {
        P = rs.pre->part.P;
        // few large partitions: cut them into ~512 row slices for the chip's workgroups (partial records merged below)
        if (P < 256) auto_slice_rows = std::max<int64_t>(N / 512, 65536);
    } else if (c->opt.partitions > 0) P = c->opt.partitions;
    else {
        // enough workgroups to fill 256 CUs (a partial record carries every state: fewer per workgroup)
        int64_t p_par = std::min<int64_t>(512, N / (merge ? 2048 : 16384));
        // Mid cardinalities (too many groups for the direct path, far fewer than 512 tables hold): a few
        // large partitions cut into ~512 row slices beat 512 small ones - the scatter fans out less and
        // an LDS table with more distinct groups sees fewer same-address atomics (hot keys) - as long as
        // the slices' partial records (slices per partition x groups) stay cheap to merge.
        if (!merge && !c->opt.no_slice && c->opt.slice_rows <= 0 && pl.mergeable && n_rounds == 1 &&
            N >= (int64_t(1) << 24) && P < p_par && pl.n_states >= 4 && pl.n_states <= MAX_MERGE_SRC) {   // (1-2 states: measured 5 % slower; > 16: the slices' records could not be merged)
            int64_t p_rec = 16;
            while (p_rec * 262144 < p_par * est) p_rec *= 2;
            if (std::max(P, p_rec) <= 64) { P = std::max(P, p_rec); auto_slice_rows = N / p_par; p_par = 1; }
        }
        P = std::max<int64_t>(std::max<int64_t>(P, p_par), 1);
        if (use_v2 && P > c->n_cu && auto_slice_rows == 0) {
            // one persistent workgroup per CU walks P near-equal partitions: a multiple of the CU count has no
            // ragged last round (1152 partitions on 256 CUs = 4.5 rounds, paid as 5).  Round down while the
            // table load stays <= 0.80, else up.
            const int64_t down = P / c->n_cu * c->n_cu, up = down + c->n_cu;
            P = (double)est / ((double)down * (double)T) <= 0.80 ? down : up;
        } else if (P > 256) P = (P + 127) / 128 * 128;
    }
    const int64_t P_LIMIT = c->opt.p_max > 0 ? std::min<int64_t>(c->opt.p_max, P_MAX) : P_MAX;
    if (rs.pre && (P > P_LIMIT || !v2_ok)) return fail(PANDRS_HIP_ERR_COMPUTATION, "pre-partitioned rows: unsupported plan or fan-out");
    if (P > P_LIMIT && c->opt.partitions <= 0 && res_slot == 0 && !c->quiet)
        // more groups than one radix level can hold (P_LIMIT tables of T slots): split by an
        // independent hash into super-partitions and run the engine on each
        return run_two_level(c, rs, pl, merge, partials, n_aggs, key_dtype, n_keys_out, res_slot, est,
                             (int64_t)((double)P_LIMIT * 0.6 * (double)T * LOAD));
    P = std::min<int64_t>(std::max<int64_t>(P, 1), P_LIMIT);

    // (a nested merge — the records of an oversized partition's pieces, of the direct path's chunks — hashes with a seed of its own:
    // the pieces' keys all come from a few of the parent's partitions, i.e. from a few RANGES of the parent's hash, and
    // part_of() maps a range of the same hash onto a handful of the merge's partitions, which then overflow and cost a retry)
    const uint32_t seed = (merge && res_slot > 0) ? 0x68E31DA5u : 0x9E3779B9u;
    bool sampled_failed = false;       // a capacity-mode run overflowed a region: repeat with the exact histogram
    bool polled = false;
    uint32_t ov_cap = 0;
    for (int attempt = 0;; attempt++) {
        c->work.off = 0;
        static const bool trace = std::getenv("PANDRS_HIP_ENGINE_TRACE") != nullptr;       // one line per attempt on stderr (a diagnostic)
        if (trace) fprintf(stderr, "[engine] slot %d merge %d partials %d rows %lld est %lld P %lld T %lld attempt %d lean %d profile %d clustered %d pre %d\n", res_slot, (int)merge, (int)partials,
                           (long long)N, (long long)est, (long long)P, (long long)T, attempt, (int)use_v2, uni_profile, (int)c->clustered_rows, rs.pre ? 1 : 0);
        c->timings.n_partitions = P; c->timings.table_slots = T; c->timings.retries = attempt;
        const uint32_t P1 = (uint32_t)P + 1;
        // capacity mode (no histogram pass): aggregate2 only (it walks a partition's 8 row ranges), unclustered rows
        const bool sampled = rs.pre ? true
                           : use_v2 && !sampled_failed && !c->clustered_rows && !c->clumped_rows && !c->opt.exact_partition && c->opt.shared_cursors &&
                             c->opt.scatter_threads != 512 && c->opt.scatter_staged && sampled_partition_ok(N, P);
        if (rs.pre && (!use_v2 || sampled_failed || attempt > 0))      // a full table or a dropped run: the producer must start over
            return fail(PANDRS_HIP_ERR_COMPUTATION, "pre-partitioned rows: a partition did not fit");
        const size_t NP = rs.pre ? 0 : sampled ? (size_t)sampled_partition_rows(N, P) : (size_t)N;     // rows of the partitioned columns
        uint32_t *counters = c->work.take<uint32_t>(64);
        uint64_t *pkeys = rs.pre ? const_cast<uint64_t *>(rs.pre->pkeys) : c->work.take<uint64_t>(NP);
        if (!counters || !pkeys) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small");
        HIP_TRY(hipMemsetAsync(counters, 0, 64 * 4, c->stream));

        ScatterArgs sa{};
        sa.key = rs.key; sa.pkeys = pkeys; sa.n_rows = N; sa.P = (uint32_t)P; sa.seed = seed;
        AggArgs aa{};
        int64_t *pgsize = nullptr;
        if (merge) {
            pgsize = c->work.take<int64_t>(N);
            if (!pgsize) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small");
            sa.mv[sa.n_move++] = MoveDesc{rs.merge_gsize ? (const void *)rs.merge_gsize : (const void *)rs.merge_states, pgsize, 0, 0};
        }
        // LDS index of every state inside its round
        int8_t st_round[MAX_STATES], st_lds[MAX_STATES];
        for (int k = 0; k < MAX_STATES; k++) { st_round[k] = -1; st_lds[k] = -1; }
        // aggregate2's fixed LDS state order: the adds of source 0..n-1, then per source its min-type states
        // (min, ~max), then the non-null counts
        for (int r = 0; r < n_rounds; r++) {
            int next = 0;
            const int n_r = round_begin[r + 1] - round_begin[r];          // (the order holds inside every round of the lean kernel)
            const int rp = round_prof[r];                                   // (rounds grouped by profile: each round its own)
            const int v2_mm = use_v2 ? ((rp >> 2) & 1) + ((rp >> 3) & 1) : 0;
            const int v2_mbase = use_v2 && ((rp >> 1) & 1) ? n_r : 0;
            int v2_next_nn = v2_mbase + n_r * v2_mm;
            for (int s = round_begin[r]; s < round_begin[r + 1]; s++) {
                EngSrc &e = srcs[s];
                uint64_t *pv = rs.pre ? const_cast<uint64_t *>(rs.pre->pvals[s]) : c->work.take<uint64_t>(NP);
                if (!pv) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small");
                sa.mv[sa.n_move++] = MoveDesc{e.data, pv, (e.rowidx && !e.data) ? 5 : 0, 0};
                uint8_t *pvalid = nullptr;
                if (e.null_bits || e.valid_bytes) {
                    pvalid = c->work.take<uint8_t>(NP);
                    if (!pvalid) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small");
                    sa.mv[sa.n_move++] = e.null_bits ? MoveDesc{e.null_bits, pvalid, 1, 0} : MoveDesc{e.valid_bytes, pvalid, 2, 0};
                }
                SrcDev &sd = aa.src[s];
                sd = SrcDev{pv, pvalid, e.kind, -1, -1, -1, -1, -1, -1, {0}};
                auto place = [&](int8_t abs_id, int8_t &lds_id) {
                    if (abs_id < 0) return;
                    st_round[abs_id] = (int8_t)r; st_lds[abs_id] = (int8_t)next; lds_id = (int8_t)next; next++;
                };
                if (use_v2) {
                    auto put = [&](int8_t abs_id, int8_t &lds_id, int at) {
                        if (abs_id < 0) return;
                        st_round[abs_id] = (int8_t)r; st_lds[abs_id] = (int8_t)at; lds_id = (int8_t)at;
                    };
                    const int sl = s - round_begin[r];
                    put(e.st_add, sd.st_add, sl);
                    put(e.st_min, sd.st_min, v2_mbase + sl * v2_mm);
                    put(e.st_max, sd.st_max, v2_mbase + sl * v2_mm + v2_mm - 1);
                    if (e.st_nn >= 0) put(e.st_nn, sd.st_nn, v2_next_nn++);
                    continue;
                }
                place(e.st_add, sd.st_add); place(e.st_min, sd.st_min); place(e.st_max, sd.st_max); place(e.st_nn, sd.st_nn);
                place(e.st_fadd, sd.st_fadd); place(e.st_ssq, sd.st_ssq);
            }
        }
        if (merge && pl.needs_second_pass) {
            if (n_rounds != 1) return fail(PANDRS_HIP_ERR_COMPUTATION, "Std / Var records merge in one round only");
            for (int s2 = 0; s2 < pl.n_src; s2++) {
                if (pl.st_ssq[s2] < 0) continue;
                if (aa.n_mvar >= MAX_MERGE_VAR) return fail(PANDRS_HIP_ERR_COMPUTATION, "too many Std / Var columns to merge");
                const int8_t ssum = pl.src_kind[s2] == 0 ? pl.st_add[s2] : pl.st_fadd[s2], snn = pl.st_nn[s2];
                MergeVar &mv = aa.mvar[aa.n_mvar++];
                mv.ssq = st_lds[pl.st_ssq[s2]]; mv.sum = st_lds[ssum]; mv.nn = snn >= 0 ? st_lds[snn] : (int8_t)-1;
                mv.sum_col = aa.src[ssum].vals; mv.nn_col = snn >= 0 ? aa.src[snn].vals : nullptr;
            }
        }
        PartInfo part;
        if (rs.pre) part = rs.pre->part;
        else if (sampled) ST_TRY(radix_partition_sampled(c, sa, &part, PANDRS_HIP_PHASE_HISTOGRAM, PANDRS_HIP_PHASE_SCATTER));
        else ST_TRY(radix_partition(c, sa, &part, PANDRS_HIP_PHASE_HISTOGRAM, PANDRS_HIP_PHASE_SCAN, PANDRS_HIP_PHASE_SCATTER));
        const uint32_t NB = part.NB;
        uint32_t *offsets = part.offsets;

        // ---- aggregate
        // oversized partitions (a hot key, heavy skew) are cut into row slices for separate workgroups
        int n_var_src = 0;
        for (int s2 = 0; s2 < pl.n_src; s2++) n_var_src += pl.st_ssq[s2] >= 0;
        const bool slicing = !c->opt.no_slice && (pl.mergeable || (!partials && !merge && n_var_src <= MAX_MERGE_VAR)) && (n_rounds == 1 || (use_v2 && !partials)) && pl.n_states <= MAX_MERGE_SRC;      // (the lean kernel's rounds fill a piece's record round by round)
        const int64_t slice_rows = c->opt.slice_rows > 0 ? c->opt.slice_rows
                                 : auto_slice_rows > 0 ? auto_slice_rows
                                                       : std::max<int64_t>(int64_t(1) << 18, (c->opt.wide_slices ? 4 : (c->opt.slice_over > 0 ? c->opt.slice_over : 2)) * (N / std::max<int64_t>(P, 1)));
        // (cut when far above the average, into pieces of the average size: build_tasks_kernel.  A forced or mid-cardinality
        // slice length is both at once.)
        const int64_t piece_rows = (c->opt.slice_rows > 0 || auto_slice_rows > 0 || c->opt.wide_slices) ? slice_rows
                                 : std::min<int64_t>(slice_rows, std::max<int64_t>(int64_t(1) << 16, N / std::max<int64_t>(P, 1)));
        const int64_t max_slices = slicing ? N / piece_rows + N / slice_rows + 2 : 0;       // slices of multi-slice partitions
        const size_t side_cap = (size_t)max_slices * (size_t)(T + 2);
        // (rows that full tables hand to an overflow run may bring up to one group each: room for them, bounded by what is likely)
        const bool want_ov = use_v2 && n_rounds == 1 && !merge && res_slot == 0 && !c->opt.no_overflow_run && n_src >= 1 && n_src <= 4 && N >= (int64_t(1) << 16);
        const int64_t ov_rows = want_ov ? std::min<int64_t>(std::max<int64_t>(N / 4, 65536), int64_t(1) << 30) : 0;
        size_t cap = (size_t)std::min<int64_t>(N, (int64_t)P1 * (T + 2) + std::min<int64_t>(ov_rows, std::max<int64_t>(4 * est, int64_t(1) << 20))) + (slicing ? side_cap : 0) +
                     (res_slot == 0 ? (size_t)c->reserve_groups : 0);       // (+ groups a caller will append: the absorb pass's compact spill)
        size_t out_cols = partials ? (size_t)(1 + pl.n_states) : (size_t)n_aggs;
        ST_TRY(rarena.ensure((size_t)n_keys_out * (Arena::padded(cap * 8) + Arena::padded(cap)) + std::max<size_t>(out_cols, 1) * Arena::padded(cap * 8 + 256) + 8192, c->stream));
        res.cap = (int64_t)cap;
        res.keys = rarena.take<uint64_t>(cap * n_keys_out);      // [n_keys][cap]; row 0 holds the engine's cell
        res.key_null = rarena.take<uint8_t>(cap * n_keys_out);
        if (partials) res.states = rarena.take<uint64_t>(cap * out_cols + 32);
        else res.aggs = rarena.take<double>(cap * std::max<size_t>(out_cols, 1) + 32);
        if (!res.keys || !res.key_null || (!res.states && !res.aggs))
            return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "result arena too small");
        aa.pkeys = pkeys; aa.offsets = offsets; aa.pgsize = pgsize; aa.NB = NB; aa.P = (uint32_t)P;
        aa.T = (uint32_t)T; aa.seed = seed; aa.n_src = n_src; aa.n_states = pl.n_states;
        aa.n_fin = partials ? 0 : n_aggs; aa.partials = partials ? 1 : 0;
        aa.n_rounds = n_rounds; aa.round_states = round_states; aa.second_pass = pl.needs_second_pass ? 1 : 0;
        std::memcpy(aa.round_src_begin, round_begin, sizeof aa.round_src_begin);
        std::memcpy(aa.kinds, pl.kinds, sizeof aa.kinds);
        std::memcpy(aa.st_round, st_round, sizeof st_round);
        std::memcpy(aa.st_lds, st_lds, sizeof st_lds);
        for (int f = 0; f < n_aggs && !partials; f++) {
            FinDev &fd = aa.fin[f];
            fd = FinDev{};
            fd.op = pl.fin_op[f]; fd.kind = pl.fin_kind[f];
            fd.st_add = fd.st_nn = fd.st_min = fd.st_max = fd.st_ssq = fd.st_fadd = fd.rowsrc_min = fd.rowsrc_max = -1;
            int s = pl.fin_src[f];
            if (s < 0) continue;                                   // COUNT: round 0, group size only
            auto lds_of = [&](int8_t abs_id) -> int8_t { return abs_id < 0 ? (int8_t)-1 : st_lds[abs_id]; };
            fd.st_add = lds_of(pl.st_add[s]); fd.st_nn = lds_of(pl.st_nn[s]);
            fd.st_min = lds_of(pl.st_min[s]); fd.st_max = lds_of(pl.st_max[s]);
            fd.st_ssq = lds_of(pl.st_ssq[s]); fd.st_fadd = lds_of(pl.st_fadd[s]);
            int8_t any = pl.st_add[s] >= 0 ? pl.st_add[s] : (pl.st_min[s] >= 0 ? pl.st_min[s] : (pl.st_max[s] >= 0 ? pl.st_max[s] : (pl.st_ssq[s] >= 0 ? pl.st_ssq[s] : (int8_t)-1)));
            fd.round = any >= 0 ? st_round[any] : 0;
            if (fd.op == PANDRS_HIP_AGG_FIRST || fd.op == PANDRS_HIP_AGG_LAST) {
                fd.rowsrc_min = st_lds[pl.st_firstrow]; fd.rowsrc_max = st_lds[pl.st_lastrow];
                fd.round = st_round[pl.st_firstrow];
                fd.col_data = rs.fin_data[s] ? rs.fin_data[s] : rs.val_data[s];
                fd.col_null = rs.fin_data[s] ? rs.fin_null_bits[s] : rs.val_null_bits[s];
            }
        }
        aa.out_keys = res.keys; aa.out_null = res.key_null; aa.out_aggs = res.aggs;
        aa.out_states = res.states; aa.cap = cap; aa.counters = counters; aa.launch_grid = (uint32_t)P + 1;
        if (slicing || use_v2) {
            const uint32_t max_tables = (uint32_t)(P1 + max_slices);
            const uint32_t max_tasks = use_v2 ? 8 * P1 + max_tables : max_tables;
            const size_t n_state_all = 1 + (size_t)pl.n_states;
            const bool lean_rounds = use_v2 && n_rounds > 1;
            const size_t snap_slots = lean_rounds ? (size_t)max_tables * (size_t)(T + 2) : 0;      // every table's key snapshot (launch 0 -> the later rounds)
            ST_TRY(c->side.ensure(Arena::padded(side_cap * 8) + Arena::padded(side_cap) + n_state_all * Arena::padded(side_cap * 8 + 256) + 8192 +
                                  Arena::padded(snap_slots * 8 + 256) + Arena::padded(snap_slots * 4 + 256) + Arena::padded(snap_slots + 256), c->stream));
            aa.side_keys = c->side.take<uint64_t>(side_cap);
            aa.side_null = c->side.take<uint8_t>(side_cap);
            aa.side_states = c->side.take<uint64_t>(side_cap * n_state_all + 32);
            aa.side_cap = side_cap;
            if (lean_rounds) {
                aa.snap_keys = c->side.take<uint64_t>(snap_slots + 16);
                aa.snap_pos = c->side.take<uint32_t>(snap_slots + 16);
                aa.snap_ctrl = c->side.take<uint8_t>(snap_slots + 16);
                if (!aa.snap_keys || !aa.snap_pos || !aa.snap_ctrl) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "side arena too small (rounds)");
            }
            AggTask *tasks = c->work.take<AggTask>(max_tasks + 8);
            AggTable *tables = c->work.take<AggTable>(max_tables + 8);
            uint32_t *n_tasks = c->work.take<uint32_t>(64);
            uint32_t *order = use_v2 && !c->opt.no_table_order ? c->work.take<uint32_t>(2 * (size_t)max_tables + 16) : nullptr;
            if (!aa.side_keys || !aa.side_null || !aa.side_states || !tasks || !tables || !n_tasks || (use_v2 && !c->opt.no_table_order && !order))
                return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (slices)");
            const uint32_t srows = slicing ? (uint32_t)std::min<int64_t>(slice_rows, 0xFFFFFFFFll) : 0xFFFFFFFFu;
            const uint32_t prows = slicing ? (uint32_t)std::min<int64_t>(piece_rows, 0xFFFFFFFFll) : 0xFFFFFFFFu;
            if (use_v2) {
                const SegSource ss{offsets, NB, part.gbeg, part.gcur, part.gend};
                hipLaunchKernelGGL(build_tables_kernel, dim3(1), dim3(1024), 0, c->stream, ss, P1, srows, prows, tasks, tables, n_tasks,
                                   max_tasks, max_tables, order, order ? order + max_tables + 8 : nullptr);
                aa.tables = tables; aa.order = order; aa.launch_grid = max_tables;
            } else {
                hipLaunchKernelGGL(build_tasks_kernel, dim3(1), dim3(1024), 0, c->stream, offsets, NB, P1, srows, prows, tasks, n_tasks, max_tasks);
                aa.launch_grid = max_tasks;
            }
            aa.tasks = tasks; aa.n_tasks = n_tasks;
        }
        {
            PhaseTimer pt(c, PANDRS_HIP_PHASE_AGGREGATE);
            size_t lds = (size_t)(T + 3) * slot_bytes + 192 + (use_v2 ? AGG2_LDS_EXTRA : 0);
            const int profile = (n_rounds == 1 || use_v2) ? uni_profile : -1;
            volatile uint32_t *hp = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1040;      // aggregate2's own corner
            if (use_v2) { hp[4] = 0; aa.host_out = const_cast<uint32_t *>(hp); aa.scatter_flags = sampled ? part.flags : nullptr; }
            // a full table's unplaced rows go to a buffer and are grouped in a run of their own (AggArgs::ov_keys): an estimate that
            // was too low costs one small extra run instead of the whole call again
            ov_cap = 0;
            if (want_ov && c->overflow.ensure((size_t)(1 + n_src) * Arena::padded((size_t)ov_rows * 8 + 256) + (size_t)n_src * Arena::padded((size_t)ov_rows + 256) + 4096, c->stream) == 0) {
                // (a configured memory limit that leaves no room for the buffer: the call goes on without it — a full table then fails the attempt as before)
                ov_cap = (uint32_t)ov_rows;
                aa.ov_keys = c->overflow.take<uint64_t>(ov_cap);
                for (int s2 = 0; s2 < n_src; s2++) {
                    aa.ov_vals[s2] = c->overflow.take<uint64_t>(ov_cap);
                    aa.ov_valid[s2] = c->overflow.take<uint8_t>(ov_cap);
                    if (!aa.ov_vals[s2] || !aa.ov_valid[s2]) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "overflow arena too small");
                }
                if (!aa.ov_keys) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "overflow arena too small");
                aa.ov_cap = ov_cap;
            } else { aa.ov_keys = nullptr; aa.ov_cap = 0; }
            // keys that arrive in bursts inside their partition (rows clustered in short runs, keys local in position — both behind the exact
            // partition): every thread folds 8 consecutive rows of the partition in registers (clustered.hip, PARTS) instead of the lean
            // kernel's row per lane, whose fast path never gets going when a key is new as its burst arrives
            bool bursts = use_v2 && !sampled && !rs.pre && (c->clumped_rows || c->clustered_rows) && !c->opt.no_burst_kernel;
            for (int r = 0; r < n_rounds && bursts; r++) bursts = clustered_has(round_begin[r + 1] - round_begin[r], round_prof[r]);
            if (use_v2 && n_rounds > 1) {
                // one launch per round over the same tables; only the last one publishes (host_out), the others re-arm the launch counters
                uint32_t *const publish = aa.host_out;
                polled = true;
                for (int r = 0; r < n_rounds && polled; r++) {
                    aa.cur_round = r; aa.src_base = round_begin[r];
                    aa.host_out = r + 1 == n_rounds ? publish : nullptr;
                    polled = bursts ? launch_clustered_parts(c, aa, round_begin[r + 1] - round_begin[r], round_prof[r], lds, (uint32_t)std::min<int64_t>(c->n_cu, aa.launch_grid))
                                    : launch_aggregate2(c, aa, round_begin[r + 1] - round_begin[r], round_prof[r], lds, (uint32_t)std::min<int64_t>(c->n_cu, aa.launch_grid));
                }
                if (!polled) return fail(PANDRS_HIP_ERR_COMPUTATION, "lean aggregate: no instantiation for a round of this profile");
            } else {
            polled = bursts ? launch_clustered_parts(c, aa, n_src, profile, lds, (uint32_t)std::min<int64_t>(c->n_cu, aa.launch_grid))
                            : (use_v2 && launch_aggregate2(c, aa, n_src, profile, lds, (uint32_t)std::min<int64_t>(c->n_cu, aa.launch_grid)));
            if (!polled) launch_aggregate(c, aa, max_spr, profile, lds);
            }
            HIP_TRY(hipGetLastError());
        }
        uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
        bool have = false;
        if (polled) {        // aggregate2's last workgroup wrote counters + scatter flag to pinned memory: poll, then fall back
            volatile uint32_t *hp = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1040;
            for (int spin = 0; spin < 4000000 && hp[4] != 1; spin++) __builtin_ia32_pause();
            if (hp[4] == 1) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                for (int i = 0; i < 4; i++) h[i] = hp[i];
                h[6] = hp[6];
                if (c->opt.agg_ablate == 8) fprintf(stderr, "aggregate2: %u rows took the retry queue\n", hp[5]);
                have = true;
            }
        }
        if (!have) {
            if (sampled)         // the scatter's overflow flag rides on the same read-back
                HIP_TRY(hipMemcpyAsync(counters + 3, part.flags, 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(h, counters, 32, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        if (sampled && h[3]) {           // a region's sampled capacity was too small (skew the sample did not show)
            sampled_failed = true;
            attempt--;
            continue;
        }
        if (h[1] == 0) {
            res.n_groups = h[0]; res.valid = true;
            const int64_t n_side = h[2];
            const int64_t n_ov = ov_cap ? (int64_t)h[6] : 0;       // (read NOW: the nested runs below reuse the pinned words)
            if (n_side > 0) {
                // merge the partial records of the sliced partitions and append their groups
                RowSource ms;
                ms.n_rows = n_side;
                ms.key = KeyDesc{aa.side_keys, nullptr, aa.side_null, DT_CELL};
                ms.merge_states = aa.side_states;
                ms.merge_stride = side_cap;
                for (int s2 = 0; s2 < pl.n_src; s2++) {        // First / Last finish with a look-up in the original column
                    ms.fin_data[s2] = rs.fin_data[s2] ? rs.fin_data[s2] : rs.val_data[s2];
                    ms.fin_null_bits[s2] = rs.fin_data[s2] ? rs.fin_null_bits[s2] : rs.val_null_bits[s2];
                }
                Options saved = c->opt;
                c->opt.no_slice = 1; c->opt.no_direct = 1; c->opt.partitions = 0;
                c->opt.groups_hint = std::max<int64_t>(std::min<int64_t>(n_side, est), 1);   // an upper bound: no second estimate
                pandrs_hip_timings tsave = c->timings;
                c->quiet++;
                int32_t st = run_engine(c, ms, pl, /*merge=*/true, partials, n_aggs, key_dtype, 1, res_slot + 1);
                c->quiet--;
                c->opt = saved;
                c->timings = tsave;
                if (st) return st;
                GroupbyResult &r2 = res_slot == 0 ? c->gb2 : c->gb3;
                ST_TRY(append_groups(c, res, cap, r2, partials, pl, n_aggs));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            if (n_ov > 0) {
                // the rows the full tables could not place: their keys are in no table, so their groups are simply appended
                RowSource os;
                os.n_rows = n_ov;
                os.key = KeyDesc{aa.ov_keys, nullptr, nullptr, DT_CELL};
                for (int s2 = 0; s2 < n_src; s2++) {
                    os.val_data[s2] = aa.ov_vals[s2];
                    os.val_null_bits[s2] = nullptr;
                    os.val_valid_bytes[s2] = (uni_profile & 1) ? aa.ov_valid[s2] : nullptr;
                }
                Options saved = c->opt;
                c->opt.partitions = 0; c->opt.no_absorb = 1;
                c->opt.groups_hint = n_ov <= 4096 ? n_ov : 0;       // (a handful of rows: every row its own group at worst — no sample, no round trip for it)
                pandrs_hip_timings tsave = c->timings;
                c->quiet++;
                int32_t st = run_engine(c, os, pl, /*merge=*/false, partials, n_aggs, key_dtype, 1, res_slot + 1);
                c->quiet--;
                c->opt = saved;
                c->timings = tsave;
                // the nested run cannot leave one radix level (it is not the call's own run): when the unplaced rows alone hold more
                // groups than one level takes, THIS attempt has failed — more partitions, or the two-level path, below — and the call
                // must not fail with the nested run's error (fuzz, round 4: p_max = 24 with 900 K groups)
                if (st && !c->capacity_exceeded) return st;
                GroupbyResult &r2 = res_slot == 0 ? c->gb2 : c->gb3;
                if (st) c->capacity_exceeded = false;
                else if ((size_t)res.n_groups + (size_t)r2.n_groups <= cap) {
                    ST_TRY(append_groups(c, res, cap, r2, partials, pl, n_aggs));
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    c->timings.retries = attempt + 100;  // (100 + attempt: an overflow run answered)
                    return 0;
                }
                res.valid = false;                       // more new groups than the result has room for: the attempt failed after all
                h[1] = 1;
            } else return 0;
        }
        if (h[1] == 0) return 0;
        if (P >= P_LIMIT) {
            if (c->opt.partitions <= 0 && res_slot == 0 && !c->quiet)      // the estimate was far too low: two-level with a safe bound
                return run_two_level(c, rs, pl, merge, partials, n_aggs, key_dtype, n_keys_out, res_slot, N,
                                     (int64_t)((double)P_LIMIT * 0.6 * (double)T * LOAD));
            c->capacity_exceeded = true;
            return fail(PANDRS_HIP_ERR_COMPUTATION,
                        "group cardinality exceeds the radix capacity (%lld partitions x %lld slots)",
                        (long long)P_LIMIT, (long long)T);
        }
        P = std::min<int64_t>(P * 4, P_LIMIT);
    }
}

// ---- two-level path: more groups than P_MAX LDS tables can hold --------------------------------------
// One extra radix pass with an INDEPENDENT hash splits the rows into K super-partitions (disjoint
// key sets); the engine then runs on each super-partition in turn and the group lists are
// concatenated.  Costs one more read + write of every column, only for inputs like a nearly
// unique key column.
static int32_t run_two_level(pandrs_hip_ctx *c, const RowSource &rs, const Plan &pl, bool merge, bool partials,
                             int n_aggs, int key_dtype, int n_keys_out, int res_slot, int64_t est,
                             int64_t groups_per_run) {
    const int64_t N = rs.n_rows;
    int64_t K = (int64_t)std::ceil((double)std::max<int64_t>(est, 1) / (double)std::max<int64_t>(groups_per_run, 1));
    K = std::min<int64_t>(std::max<int64_t>(K, 2), P_MAX);
  for (;; K = std::min<int64_t>(K * 4, P_MAX)) {        // a sub-run that still overflows => finer super-partitions
    GroupbyResult &res = c->gb;
    Arena &rarena = c->result;

    // ---- columns that travel: key cells, every plan source (+ validity bytes), row index, merge columns
    const int n_src = merge ? 0 : pl.n_src;
    const int n_mcols = merge ? pl.n_states : 0;
    size_t need = Arena::padded(size_t(N) * 8) * (size_t)(1 + n_src + n_mcols + (merge ? 1 : 0) + (pl.st_firstrow >= 0 ? 1 : 0))
                + Arena::padded(size_t(N)) * (size_t)(n_src + 1) + (1 << 20);
    ST_TRY(c->super.ensure(need, c->stream));
    ST_TRY(c->work.ensure(engine_workspace_bytes(N, 1 + n_src + n_mcols + 2, n_src), c->stream));
    ScatterArgs sa{};
    uint64_t *spk = c->super.take<uint64_t>(N);
    if (!spk) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "super arena too small");
    sa.key = rs.key; sa.pkeys = spk; sa.n_rows = N; sa.P = (uint32_t)K; sa.seed = 0xC2B2AE3Du;
    uint64_t *spv[MAX_SRC]{}; uint8_t *spvalid[MAX_SRC]{};
    for (int s2 = 0; s2 < n_src; s2++) {
        spv[s2] = c->super.take<uint64_t>(N);
        if (!spv[s2]) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "super arena too small");
        sa.mv[sa.n_move++] = MoveDesc{rs.val_data[s2], spv[s2], 0, 0};
        if (rs.val_null_bits[s2] || rs.val_valid_bytes[s2]) {
            spvalid[s2] = c->super.take<uint8_t>(N);
            if (!spvalid[s2]) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "super arena too small");
            sa.mv[sa.n_move++] = rs.val_null_bits[s2] ? MoveDesc{rs.val_null_bits[s2], spvalid[s2], 1, 0}
                                                      : MoveDesc{rs.val_valid_bytes[s2], spvalid[s2], 2, 0};
        }
    }
    uint64_t *sprow = nullptr;
    if (!merge && pl.st_firstrow >= 0) {
        sprow = c->super.take<uint64_t>(N);
        if (!sprow) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "super arena too small");
        sa.mv[sa.n_move++] = rs.row_index ? MoveDesc{rs.row_index, sprow, 0, 0} : MoveDesc{nullptr, sprow, 5, 0};
    }
    int64_t *spg = nullptr; uint64_t *spm[MAX_STATES]{};
    if (merge) {
        spg = c->super.take<int64_t>(N);
        if (!spg) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "super arena too small");
        sa.mv[sa.n_move++] = MoveDesc{rs.merge_gsize ? (const void *)rs.merge_gsize : (const void *)rs.merge_states, spg, 0, 0};
        for (int k = 0; k < n_mcols; k++) {
            spm[k] = c->super.take<uint64_t>(N);
            if (!spm[k]) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "super arena too small");
            const uint64_t *src = rs.merge_cols[k] ? rs.merge_cols[k] : rs.merge_states + (size_t)(k + 1) * rs.merge_stride;
            sa.mv[sa.n_move++] = MoveDesc{src, spm[k], 0, 0};
        }
    }
    uint8_t *ones = c->super.take<uint8_t>(N);           // null flags of the NULL super-partition's rows
    uint32_t *d_off = c->super.take<uint32_t>(K + 8);
    if (!ones || !d_off) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "super arena too small");
    c->work.off = 0;
    PartInfo part;
    ST_TRY(radix_partition(c, sa, &part, PANDRS_HIP_PHASE_HISTOGRAM, PANDRS_HIP_PHASE_SCAN, PANDRS_HIP_PHASE_SCATTER));
    gather_part_offsets(c, part.offsets, part.NB, (uint32_t)K + 2, d_off);
    std::vector<uint32_t> off(K + 2);
    HIP_TRY(hipMemcpyAsync(off.data(), d_off, (K + 2) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(ones, 1, size_t(N), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));

    // ---- result arrays for the concatenation (every row could be its own group)
    const size_t cap = (size_t)N + 8 + (res_slot == 0 ? (size_t)c->reserve_groups : 0);
    const size_t out_cols = partials ? (size_t)(1 + pl.n_states) : (size_t)std::max(n_aggs, 1);
    res = GroupbyResult{};
    res.n_keys = n_keys_out; res.n_aggs = n_aggs; res.n_state = 1 + pl.n_states; res.partials = partials;
    res.key_dtype = key_dtype;
    ST_TRY(rarena.ensure((size_t)n_keys_out * (Arena::padded(cap * 8) + Arena::padded(cap)) + out_cols * Arena::padded(cap * 8 + 256) + 8192, c->stream));
    res.cap = (int64_t)cap;
    res.keys = rarena.take<uint64_t>(cap * n_keys_out);
    res.key_null = rarena.take<uint8_t>(cap * n_keys_out);
    if (partials) res.states = rarena.take<uint64_t>(cap * out_cols + 32);
    else res.aggs = rarena.take<double>(cap * out_cols + 32);
    if (!res.keys || !res.key_null || (!res.states && !res.aggs)) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "result arena too small");

    Options saved = c->opt;
    pandrs_hip_timings tsave = c->timings;
    c->opt.no_direct = 1; c->opt.groups_hint = 0; c->opt.partitions = 0;
    int32_t st = 0;
    c->capacity_exceeded = false;
    for (int64_t sp = 0; sp <= K && !st; sp++) {          // sp == K: the NULL-key rows
        const uint32_t b = off[sp], e2 = off[sp + 1];
        if (b == e2) continue;
        RowSource sub;
        sub.n_rows = (int64_t)e2 - b;
        sub.key = KeyDesc{spk + b, nullptr, sp == K ? ones : nullptr, DT_CELL};
        for (int s2 = 0; s2 < n_src; s2++) {
            sub.val_data[s2] = spv[s2] + b;
            sub.val_valid_bytes[s2] = spvalid[s2] ? spvalid[s2] + b : nullptr;
            sub.fin_data[s2] = rs.fin_data[s2] ? rs.fin_data[s2] : rs.val_data[s2];
            sub.fin_null_bits[s2] = rs.fin_data[s2] ? rs.fin_null_bits[s2] : rs.val_null_bits[s2];
        }
        sub.row_index = sprow ? sprow + b : nullptr;
        if (merge) {
            sub.merge_gsize = spg + b;
            for (int k = 0; k < n_mcols; k++) sub.merge_cols[k] = spm[k] + b;
            sub.merge_states = spm[0];          // non-null marker only; merge_cols carry the data
        }
        c->quiet++;
        st = run_engine(c, sub, pl, merge, partials, n_aggs, key_dtype, 1, res_slot + 1);
        c->quiet--;
        if (!st) st = append_groups(c, res, cap, c->gb2, partials, pl, n_aggs);
        if (!st) HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->opt = saved;
    c->timings = tsave;
    c->timings.n_partitions = K; c->timings.retries = 0; c->timings.estimated_groups = est;
    if (st && c->capacity_exceeded && K < P_MAX) { c->capacity_exceeded = false; continue; }
    if (st) return st;
    res.valid = true;
    return 0;
  }
}

}  // namespace pandrs
