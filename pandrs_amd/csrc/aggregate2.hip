// aggregate2.hip — the lean persistent LDS hash aggregate for the common case: raw partitioned rows,
// one round, every aggregated column of the same kind with the same ops (a "uniform profile").
//
// Same job as aggregate_kernel (groupby.hip) — the per-group fold of
// src/optimized/split_dataframe/group/aggregation.rs:500-754 on one radix partition held in LDS —
// rebuilt around what the round-1 profile showed (profiles/r01_final3_lds_counters.json: 29 LDS
// instructions per 64 rows, the LDS pipe 53 % busy, one 157 KB workgroup per CU so nothing overlaps
// a workgroup's prologue / epilogue):
//   * persistent workgroups: one per CU walks the task list (task = partition or row slice); the first
//     row batch of the NEXT task is already in flight (registers) while the current task's table is
//     compacted and written out, so the HBM stream does not stop between partitions;
//   * max states are kept NEGATED (~enc), so min and max updates are the same ds_min_u64 and a wave
//     retires its few (lane, state) updates in 2-3 dense instructions instead of 8 sparse ones
//     (per row a state improves with probability ~1/k at the group's k-th row, but with 64 lanes
//     nearly every one of the 8 conditional atomics used to issue);
//   * group sizes are u32 and there is no position map: 12 + 8 x states bytes per slot instead of
//     20 + 8 x states => more slots per table => fewer radix partitions for the scatter;
//   * one block scan for the compaction instead of three; per-source pointers live in SGPRs.
#include "aggregate.hpp"
#include "swiss.hpp"
#include <atomic>

namespace pandrs {

namespace {

template <int NSRC, int PROFILE, int ABLATE, int DEPTH, bool SMALL = false>
__global__ __launch_bounds__(AG_THREADS) void aggregate2_kernel(AggArgs a) {
    constexpr bool HAS_V = (PROFILE & 1) != 0, OP_ADD = ((PROFILE >> 1) & 1) != 0;
    constexpr bool OP_MIN = ((PROFILE >> 2) & 1) != 0, OP_MAX = ((PROFILE >> 3) & 1) != 0;
    constexpr int KIND = (PROFILE >> 4) & 1;                 // 0 f64, 1 i64
    constexpr int MM = (OP_MIN ? 1 : 0) + (OP_MAX ? 1 : 0);  // min-type states per source
    constexpr uint64_t M_IDENT = KIND == 0 ? 0xFFF0000000000000ull : ~0ull;   // enc(+inf) = ~enc(-inf); enc(MAX) = ~enc(MIN)
    constexpr uint32_t QCAP = 128;                           // per-wave retry queue (row indices)
    constexpr bool NT = ABLATE == 7;                         // experiments: non-temporal row loads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t T = a.T, T1 = T + (SMALL ? 3 : 2), tid = threadIdx.x;   // slot T: the key equal to the table sentinel; T + 1: the NULL key (SMALL)
    // LDS: keys[T1] | states[round_states][T1] | gsz[T1] (u32) | ctrl[T] (u8) | misc[40] | queue[16][QCAP]   (T multiple of 16)
    // state order (fixed by run_engine for this kernel): adds of source 0..n-1, then per source its
    // min-type states (min, ~max), then the non-null counts
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *st = keys + T1;
    uint32_t *gsz = reinterpret_cast<uint32_t *>(st + (size_t)a.round_states * T1);
    // [T] slot tags, read as 16-byte groups (ds_read_b128): the start is rounded up to 16 bytes — in SMALL mode (T1 = T + 3)
    // an odd 1 + round_states leaves gsz 8 bytes off (the LDS budget keeps ~90 spare bytes for this)
    uint8_t *ctrl = smem + ((static_cast<uint32_t>(reinterpret_cast<unsigned char *>(gsz + ((T1 + 3) & ~3u)) - smem) + 15u) & ~15u);
    uint32_t *misc = reinterpret_cast<uint32_t *>(ctrl + T);
    uint32_t *queue = misc + 40 + (tid >> 6) * QCAP;
    // misc[0..16] scan scratch, [20] overflow, [21] sentinel-key-present, [22] output base
    constexpr uint32_t m_base = OP_ADD ? (uint32_t)NSRC : 0u;
    const uint64_t *vals[NSRC];
    const uint8_t *valid[NSRC];
    int nn_idx[NSRC];
#pragma unroll
    for (int c = 0; c < NSRC; c++) {
        const SrcDev &sd = a.src[(SMALL ? 0 : a.src_base) + c];
        vals[c] = sd.vals; valid[c] = sd.valid; nn_idx[c] = HAS_V ? sd.st_nn : -1;
    }
    const bool resume = !SMALL && a.snap_keys && a.cur_round > 0;      // a later round: tables start from launch 0's snapshot
    const uint64_t *pkeys = a.pkeys;
    const uint32_t seed = a.seed, NG = T >> 4, lane = tid & 63;

    // every workgroup ends here: the last one publishes the call's counters to the host
    auto finish = [&]() {
        __syncthreads();
        if (tid == 0 && (a.host_out || (!SMALL && a.snap_keys))) {
            __threadfence();
            if (atomicAdd(&a.counters[8], 1u) == gridDim.x - 1) {
                __threadfence();
                if (a.host_out) {
                    for (int i = 0; i < 3; i++) a.host_out[i] = __hip_atomic_load(&a.counters[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.host_out[6] = __hip_atomic_load(&a.counters[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ABLATE == 8) a.host_out[5] = __hip_atomic_load(&a.counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.host_out[3] = a.scatter_flags ? __hip_atomic_load(a.scatter_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    __hip_atomic_store(&a.host_out[4], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    // rounds: every launch but the last publishes nothing and re-arms the per-launch counters (workgroups done, table tickets)
                    a.counters[8] = 0; a.counters[10] = 0;
                    __threadfence();
                }
            }
        }
    };
    // SMALL: table b = rows [b * s_chunk, (b + 1) * s_chunk) of the original columns; no task list in memory
    auto small_task = [&](uint32_t b) { return AggTask{0u, (uint32_t)min(b * a.s_chunk, a.s_rows), (uint32_t)min((b + 1) * a.s_chunk, a.s_rows), 0u}; };
    auto get_table = [&](uint32_t b) { return SMALL ? AggTable{b, 1u, 0u, 0u} : a.tables[a.order ? a.order[b] : b]; };
    auto get_task = [&](uint32_t i) { return SMALL ? small_task(i) : a.tasks[i]; };
    const uint32_t n_tables = SMALL ? (a.s_rows + a.s_chunk - 1) / a.s_chunk : a.n_tasks[1];
    uint32_t tb = blockIdx.x;
    if (tb >= n_tables) { finish(); return; }
    // the capacity-mode scatter dropped rows (its sampled regions were too small: rows clumped by position in a way the estimate's adjacent
    // pairs do not show): the host repeats the call with the exact histogram whatever happens here — do not aggregate what is incomplete
    if (!SMALL && a.scatter_flags && __hip_atomic_load(a.scatter_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { finish(); return; }
    // (rounds: a table of launch 0 filled up — its snapshot is incomplete, the attempt is lost)
    if (resume && __hip_atomic_load(&a.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { finish(); return; }
    auto table_id = [&](uint32_t b) { return (SMALL || !a.order) ? b : a.order[b]; };
    AggTable tab = get_table(tb);
    uint32_t ti = tab.task_beg;
    AggTask cur = get_task(ti);

    // Register ring of DEPTH row slots per thread (one row per slot): slot d holds batch (pit + d) of the
    // current task while the loads of the following batches are in flight.  Batches start at a 128-byte
    // boundary (16 rows: a wave's 512-byte loads cover whole lines; misaligned streams measured 15 %
    // slower, experiments/ubench/stream_formats.hip) and a task's batch count is padded to a multiple of
    // DEPTH so that slot numbers are compile-time constants across task boundaries.  Plain loads: the rows were written
    // by the scatter just before, and the FIRST read of freshly written lines is faster with plain than with nt loads
    // (experiments/ubench/fresh_read.hip; this kernel on C2: 1.08 ms plain, 1.27 ms nt).  The PMC read counter shows
    // 5.7 GB for the 4.0 GB of rows with plain loads and exactly 4.0 GB with nt loads (ABLATE 7 = the nt variant).
    uint64_t rk[DEPTH], rv[DEPTH][NSRC];
    uint32_t rok[DEPTH];
    auto n_batches = [](const AggTask &tk) {
        const uint32_t b0 = tk.beg & ~15u;
        return ((tk.end - b0 + AG_THREADS - 1) / AG_THREADS + DEPTH - 1) / DEPTH * DEPTH;
    };
    auto fetch = [&](int d, const AggTask &tk, uint32_t batch) {
        uint32_t i = (tk.beg & ~15u) + batch * AG_THREADS + tid;
        i = min(max(i, tk.beg), tk.end - 1);          // rows outside [beg, end) are loaded (in bounds) but never processed
        if (ABLATE == 6) i = tk.beg + ((i - tk.beg) & 2047u);     // experiments: every load hits L2 (compute time alone)
        rk[d] = SMALL ? key_cell(a.dkey, i) : (NT ? __builtin_nontemporal_load(pkeys + i) : pkeys[i]);
        rok[d] = 0x7FFFFFFFu;                            // bit c: value c is valid; bit 31: the key is NULL (SMALL)
        if (SMALL && key_is_null(a.dkey, i)) rok[d] |= 0x80000000u;
#pragma unroll
        for (int c = 0; c < NSRC; c++) {
            rv[d][c] = NT ? __builtin_nontemporal_load(vals[c] + i) : vals[c][i];
            if (HAS_V && (SMALL ? bit_at(valid[c], i) : valid[c][i] == 0)) rok[d] &= ~(1u << c);
        }
    };

    // ---- the fold of one row into its group's states (aggregation.rs:625-674) ----
    auto update = [&](uint32_t slot, const uint64_t (&v)[NSRC], uint32_t okm) {
        atomicAdd(&gsz[slot], 1u);
        if (ABLATE == 2) return;
        // current min-type states first (back-to-back ds_read_b64, one wait), then the adds
        uint64_t cm[NSRC][MM > 0 ? MM : 1];
        if (MM > 0 && ABLATE != 1) {
#pragma unroll
            for (int c = 0; c < NSRC; c++)
#pragma unroll
                for (int j = 0; j < MM; j++) cm[c][j] = st[(size_t)(m_base + c * MM + j) * T1 + slot];
        }
#pragma unroll
        for (int c = 0; c < NSRC; c++) {
            if (!HAS_V || ((okm >> c) & 1)) {
                if (OP_ADD) {
                    if (KIND == 0) atomicAdd(reinterpret_cast<double *>(&st[(size_t)c * T1 + slot]), __longlong_as_double((long long)v[c]));
                    else atomicAdd((unsigned long long *)&st[(size_t)c * T1 + slot], (unsigned long long)v[c]);
                }
                if (HAS_V && nn_idx[c] >= 0) atomicAdd((unsigned long long *)&st[(size_t)nn_idx[c] * T1 + slot], 1ull);
            }
        }
        if (MM > 0 && ABLATE != 1) {
            // which (source, min/max) states does this row improve?  Skipping on a stale read is safe: states
            // only move towards the extreme.  Rust's f64::min/max ignore NaN operands (aggregation.rs:653, :666).
            uint32_t upd = 0;
            uint64_t e[NSRC];
#pragma unroll
            for (int c = 0; c < NSRC; c++) {
                bool cmp = !HAS_V || ((okm >> c) & 1) != 0;
                if (KIND == 0) { const double d = __longlong_as_double((long long)v[c]); cmp = cmp && d == d; }
                e[c] = enc_val<KIND>(v[c]);
                if (OP_MIN) upd |= (cmp && e[c] < cm[c][0]) ? 1u << (c * MM) : 0u;
                if (OP_MAX) upd |= (cmp && e[c] > ~cm[c][MM - 1]) ? 1u << (c * MM + MM - 1) : 0u;
            }
            // one ds_min_u64 per pending update and lane: lanes hold different states in the same
            // instruction, so the wave needs max-popcount iterations, not one per state
            while (upd) {
                const uint32_t q = (uint32_t)__ffs((int)upd) - 1u;
                upd &= upd - 1;
                const uint32_t c_ = MM == 2 ? q >> 1 : q;
                uint64_t ee = e[0];
#pragma unroll
                for (int c = 1; c < NSRC; c++) ee = c_ == (uint32_t)c ? e[c] : ee;
                const uint64_t neg = OP_MIN ? (MM == 2 ? 0ull - (uint64_t)(q & 1) : 0ull) : ~0ull;
                atomicMin((unsigned long long *)&st[(size_t)(m_base + q) * T1 + slot], ee ^ neg);
            }
        }
    };
    // ---- many placed rows of the wave in ONE slot (see the call site; `ok` marks them): VALU-only wave reductions, then lane 0 applies the totals —
    // same rules as update(): nulls skipped, NaN ignored by min / max, sums re-associate (as everywhere)
    auto wave_fold = [&](uint32_t slot, unsigned long long okw, bool ok, const uint64_t (&v)[NSRC], uint32_t okm) {
        if (lane == 0) atomicAdd(&gsz[slot], (uint32_t)__popcll(okw));
        if (ABLATE == 2) return;
#pragma unroll
        for (int c = 0; c < NSRC; c++) {
            const bool valid = ok && (!HAS_V || ((okm >> c) & 1));
            if (HAS_V && nn_idx[c] >= 0) {
                const unsigned long long vm = __ballot(valid);
                if (lane == 0 && vm) atomicAdd((unsigned long long *)&st[(size_t)nn_idx[c] * T1 + slot], (unsigned long long)__popcll(vm));
            }
            if (OP_ADD) {
                const uint64_t tot = wave_reduce64<KIND == 0 ? 0 : 1>(valid ? v[c] : 0ull, 0ull);       // (+0.0 has the bits of 0)
                if (lane == 0) {
                    if (KIND == 0) atomicAdd(reinterpret_cast<double *>(&st[(size_t)c * T1 + slot]), __longlong_as_double((long long)tot));
                    else atomicAdd((unsigned long long *)&st[(size_t)c * T1 + slot], (unsigned long long)tot);
                }
            }
            if (MM > 0 && ABLATE != 1) {
                bool cmp = valid;
                if (KIND == 0) { const double dv = __longlong_as_double((long long)v[c]); cmp = cmp && dv == dv; }
                const uint64_t e = enc_val<KIND>(v[c]);
                if (OP_MIN) {
                    const uint64_t mn = wave_reduce64<2>(cmp ? e : ~0ull, ~0ull);
                    if (lane == 0 && mn != ~0ull) atomicMin((unsigned long long *)&st[(size_t)(m_base + c * MM) * T1 + slot], (unsigned long long)mn);
                }
                if (OP_MAX) {
                    const uint64_t mx = wave_reduce64<3>(cmp ? e : 0ull, 0ull);      // max of the codes; stored negated like every max state
                    if (lane == 0 && mx != 0ull) atomicMin((unsigned long long *)&st[(size_t)(m_base + c * MM + MM - 1) * T1 + slot], (unsigned long long)~mx);
                }
            }
        }
    };
    // ---- rows the fast path could not place (first sight of a key, a key outside its home group, a tag
    // collision, the sentinel-valued key): the wave keeps their row indices and handles 64 at a time, one
    // per lane, with the general probe — so the rare path costs per ROW, not per wave that contains one.
    uint32_t qn = 0;                                   // wave-uniform
    bool cur_multi = false;                            // the table being filled is a slice of an oversized partition
    auto drain = [&](uint32_t n_take) {                // n_take <= 64 entries from the top of the wave's queue
        qn -= n_take;
        if (!SMALL && a.snap_keys && *reinterpret_cast<volatile uint32_t *>(&misc[20])) return;       // (rounds: the table is full, the attempt is lost)
        if (ABLATE == 8 && lane == 0) atomicAdd(&a.counters[5], n_take);       // experiments: rows that took the retry queue
        bool ovf = false, placed = false;
        uint64_t ov_k = 0, ov_v[NSRC];
        uint32_t ov_okm = 0, pslot = 0;
#pragma unroll
        for (int c = 0; c < NSRC; c++) ov_v[c] = 0;
        if (lane < n_take) {
            const uint32_t i = queue[qn + lane];
            const uint64_t k = SMALL ? key_cell(a.dkey, i) : pkeys[i];
            const bool knull = SMALL && key_is_null(a.dkey, i);
            uint64_t v[NSRC];
            uint32_t okm = 0x7FFFFFFFu;
#pragma unroll
            for (int c = 0; c < NSRC; c++) {
                // the queued rows' values are re-read 64+ rows after the ring loaded them: a third of these lines have left L2 by then
                // (C2: 7.5 % of the rows take the queue; PMC 5.7 GB fetched for 4.0 GB of rows; ABLATE 9 = no re-read: 1.14 -> 0.99 ms).
                // Carrying the values in a per-wave global ring instead cost the same time in stores (5 per batch) as it saved in gathers.
                v[c] = ABLATE == 9 ? k : vals[c][i];
                if (HAS_V && (SMALL ? bit_at(valid[c], i) : valid[c][i] == 0)) okm &= ~(1u << c);
            }
            uint32_t slot = T;
            if (knull) { slot = T + 1; misc[23] = 1; }           // NULL key: its own group (grouping.rs:74)
            else if (k == EMPTY_KEY) misc[21] = 1;
            // (at most 16 groups = a 256-slot window — 8 overflowed a handful of rows of uniform C2 at load 0.69 —: inserts and lookups obey the same bound, so a key is either inside its window or —
            // consistently, slots never free up — handed to the overflow path; a FULL table is no longer walked end to end per unplaced row)
            // (rounds: no overflow run behind a full table — the whole table is the window, and the attempt ends when it is really full)
            else slot = swiss_find(k, keys, ctrl, T, seed, SMALL ? 8u : (a.snap_keys ? 0xFFFFFFFFu : 16u));
            if (slot <= T + 1) { placed = true; pslot = slot; }
            else ovf = true;                            // table full
            ov_k = k; ov_okm = okm;
#pragma unroll
            for (int c = 0; c < NSRC; c++) ov_v[c] = v[c];
        }
        // A key's FIRST rows arrive here 64 at a time when its rows lie together inside the partition (nearly sorted input, a key of the
        // absorb pass's spill): 64 lanes on the new slot would serialise on every state (100 rows per key shuffled within +-50 rows, C2's
        // shape: aggregate 3.6 ms against 1.1 in random order).  Same remedy as in the fast path: the lanes in the first placed lane's slot
        // fold on the VALU when they are many.
        {
            const unsigned long long pw = __ballot(placed);
            if (pw) {
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)pslot, __builtin_amdgcn_readfirstlane(__ffsll((long long)pw) - 1));
                const bool same = placed && pslot == s0;
                const unsigned long long samew = __ballot(same);
                if ((uint32_t)__popcll(samew) >= (cur_multi ? a.fold_min_multi : a.fold_min)) {
                    wave_fold(s0, samew, same, ov_v, ov_okm);
                    placed = placed && !same;
                }
            }
            if (placed) update(pslot, ov_v, ov_okm);
        }
        // rows of a full table: appended for a run of their own (see AggArgs::ov_keys), or — no buffer, a `multi` table, the
        // buffer full — the overflow flag and the host retries with more partitions
        const unsigned long long om = __ballot(ovf);
        if (om) {                                       // wave-uniform, rare
            if (SMALL || !a.ov_keys || cur_multi) { if (ovf) misc[20] = 1; }
            else {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&a.counters[6], (uint32_t)__popcll(om));
                base = __shfl(base, 0, 64);
                const uint32_t pos = base + (uint32_t)__popcll(om & ((1ull << lane) - 1ull));
                if (ovf) {
                    if (pos < a.ov_cap) {
                        a.ov_keys[pos] = ov_k;
#pragma unroll
                        for (int c = 0; c < NSRC; c++) {
                            a.ov_vals[c][pos] = ov_v[c];
                            if (HAS_V) a.ov_valid[c][pos] = (uint8_t)((ov_okm >> c) & 1u);
                        }
                    } else misc[20] = 1;
                }
            }
        }
    };

#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch(d, cur, d);

    for (;;) {                                         // one LDS table per iteration
        // the NEXT table is drawn from a ticket counter now (its first rows fly under this table's last ones): the first gridDim.x tables
        // are the workgroups' own, the others go to whoever comes first — partitions differ in size under skewed keys, and a fixed
        // round-robin's slowest workgroup (the one whose four partitions each hold a hot key) was the kernel's time
        if (tid == 0) misc[33] = gridDim.x + atomicAdd(&a.counters[10], 1u);
        const bool multi = tab.multi != 0;
        const uint32_t t_end = tab.task_beg + tab.n_tasks;
        cur_multi = multi;

        const size_t snap = resume || (!SMALL && a.snap_keys) ? (size_t)table_id(tb) : 0;
        if (resume) {           // a later round: launch 0's key table — every key of this table has its slot already
            const uint64_t *sk = a.snap_keys + snap * (T + 2);
            const uint32_t *sc = reinterpret_cast<const uint32_t *>(a.snap_ctrl + snap * T);
            for (uint32_t s = tid; s < T1; s += AG_THREADS) { keys[s] = sk[s]; gsz[s] = 0; }
            for (uint32_t s = tid; s < (T >> 2); s += AG_THREADS) reinterpret_cast<uint32_t *>(ctrl)[s] = sc[s];
        } else {
            for (uint32_t s = tid; s < T1; s += AG_THREADS) { keys[s] = EMPTY_KEY; gsz[s] = 0; }
            for (uint32_t s = tid; s < (T >> 2); s += AG_THREADS) reinterpret_cast<uint32_t *>(ctrl)[s] = 0;
        }
        for (int k = 0; k < a.round_states; k++) {
            const uint64_t idv = ((uint32_t)k >= m_base && (uint32_t)k < m_base + (uint32_t)(NSRC * MM)) ? M_IDENT : 0ull;
            uint64_t *dst = st + (size_t)k * T1;
            for (uint32_t s = tid; s < T1; s += AG_THREADS) dst[s] = idv;
        }
        if (tid < 32) misc[tid] = 0;
        __syncthreads();
        if (resume && tid == 0) misc[21] = a.snap_pos[snap * (T + 2) + T + 1];      // the sentinel-valued key was seen in launch 0
        const uint32_t tbn = misc[33];
        const bool have_next_tab = tbn < n_tables;
        const AggTable ntab = get_table(have_next_tab ? tbn : tb);

      for (;;) {                                       // the row ranges that feed this table
        const bool more_in_tab = ti + 1 < t_end;
        const bool have_next = more_in_tab || have_next_tab;
        const AggTask nxt = get_task(more_in_tab ? ti + 1 : (have_next_tab ? ntab.task_beg : ti));
        const uint32_t beg = cur.beg, end = cur.end, beg0 = cur.beg & ~15u;
        const uint32_t n_it = n_batches(cur);
        for (uint32_t pit = 0; pit < n_it; pit += DEPTH) {
            const bool last = pit + DEPTH >= n_it;
            // SMALL: a full table is an expected outcome (the caller could not know the group count): stop feeding it,
            // a probe of a full table walks every group
            if (SMALL && *reinterpret_cast<volatile uint32_t *>(&misc[20])) { qn = 0; break; }
#pragma unroll
            for (int h = 0; h < DEPTH; h++) {
                const uint32_t row = beg0 + (pit + h) * AG_THREADS + tid;
                const uint64_t k = rk[h];
                uint64_t v[NSRC];
                const uint32_t okm = rok[h];
#pragma unroll
                for (int c = 0; c < NSRC; c++) v[c] = rv[h][c];
                // refill this slot: DEPTH batches ahead, or the next task's batch h (its loads fly under this task's epilogue)
                if (!last) fetch(h, cur, pit + h + DEPTH);
                else if (have_next) fetch(h, nxt, h);
                const bool act = row >= beg && row < end;
                if (ABLATE >= 3 && ABLATE < 6) {          // experiments: keep every load alive without using it
                    uint64_t x = k;
#pragma unroll
                    for (int c = 0; c < NSRC; c++) x ^= v[c];
                    if (x == 0x1234567ull) misc[30] = 1;
                    continue;
                }
                // ---- fast path, branch-free: home group's 16 tags (one ds_read_b128), SWAR match, first
                // candidate's key verified (one ds_read_b64)
                const uint32_t hsh = hash32(k, seed);
                const uint32_t g = slot_of(hsh, NG);
                const uint32_t tag4 = max(hsh & 0xFFu, 1u) * 0x01010101u;
                const uint4 cw = *reinterpret_cast<const uint4 *>(ctrl + 16 * g);
                const uint32_t x0 = cw.x ^ tag4, x1 = cw.y ^ tag4, x2 = cw.z ^ tag4, x3 = cw.w ^ tag4;
                const uint32_t c0 = (x0 - 0x01010101u) & ~x0 & 0x80808080u, c1 = (x1 - 0x01010101u) & ~x1 & 0x80808080u;
                const uint32_t c2 = (x2 - 0x01010101u) & ~x2 & 0x80808080u, c3 = (x3 - 0x01010101u) & ~x3 & 0x80808080u;
                uint32_t sel = c3, off = 12;
                sel = c2 ? c2 : sel; off = c2 ? 8u : off;
                sel = c1 ? c1 : sel; off = c1 ? 4u : off;
                sel = c0 ? c0 : sel; off = c0 ? 0u : off;
                uint32_t idx = 16 * g + off + (((uint32_t)__ffs((int)sel) - 1u) >> 3);
                idx = sel ? idx : T;                       // no candidate: slot T never holds a real key
                const bool ok = act && keys[idx] == k && k != EMPTY_KEY && !(SMALL && (okm >> 31));
                const unsigned long long miss = __ballot(act && !ok);
                if (miss) {                                // wave-uniform
                    if (act && !ok) queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(miss >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)miss, 0u))] = row;
                    qn += (uint32_t)__popcll(miss);
                    if (qn >= 64) drain(64);
                }
                // a slice of an oversized partition is mostly ONE key (a hot key, the NULL group): 64 lanes adding to the same
                // LDS words serialise (13 atomics x 64 lanes per batch for C2's profile: ~830 LDS cycles per 64 rows).  When every
                // placed row of the wave sits in the same slot, the wave folds its rows on the VALU first and one lane updates the table.
                // The same holds, less extremely, for a hot key that shares an ordinary table with other keys (a key holding 2-3 x the
                // average partition's rows is not cut out: 250 K rows x 13 same-address atomics per table, 6.6 ms for C2's shape with
                // half the rows on 200 keys): the lanes in the FIRST placed lane's slot fold when they are many, the others update as usual.
                {
                    const unsigned long long okw = __ballot(ok);
                    if (okw) {
                        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)idx, __builtin_amdgcn_readfirstlane(__ffsll((long long)okw) - 1));   // (not __shfl: that is a ds_bpermute, LDS pipe)
                        const bool same = ok && idx == s0;
                        const unsigned long long samew = __ballot(same);
                        if ((uint32_t)__popcll(samew) >= (cur_multi ? a.fold_min_multi : a.fold_min)) {
                            wave_fold(s0, samew, same, v, okm);
                            if (ok && !same) update(idx, v, okm);
                            continue;
                        }
                    }
                }
                if (ok) update(idx, v, okm);
            }
        }
        if (!more_in_tab) { cur = nxt; break; }         // nxt = the next table's first range (already in flight)
        ti++; cur = nxt;
      }
        if (qn) drain(qn);
        __syncthreads();
        if (misc[20]) { if (tid == 0) a.counters[1] = 1; finish(); return; }
        if (ABLATE == 4) { if (!have_next_tab) break; tb = tbn; tab = ntab; ti = tab.task_beg; continue; }   // experiments: no compaction / outputs
        if (SMALL) {
            // flush: every group of this table joins the context's global table (device-scope atomics: a few thousand per
            // workgroup).  Sums add; min states go in as ~enc and max states as enc, both through atomicMax, so an
            // all-zero global table is an armed one whatever the next call aggregates.
            const bool sentinel = misc[21] != 0, nullseen = misc[23] != 0;
            const size_t GS = (size_t)a.g_slots + 2;
            for (uint32_t s = tid; s < T1; s += AG_THREADS) {
                const bool occ = s < T ? keys[s] != EMPTY_KEY : (s == T ? sentinel : nullseen);
                if (!occ) continue;
                uint32_t gs = s == T ? a.g_slots : a.g_slots + 1;
                if (s < T) {
                    const uint64_t k = keys[s];
                    uint32_t slot = hash32(k, 0x2545F491u) & (a.g_slots - 1), probes = 0;
                    gs = 0xFFFFFFFFu;
                    for (; probes < 1024; probes++) {
                        const uint64_t old = atomicCAS((unsigned long long *)&a.g_keys[slot], EMPTY_KEY, k);
                        if (old == EMPTY_KEY || old == k) { gs = slot; break; }
                        slot = (slot + 1) & (a.g_slots - 1);
                    }
                    if (gs == 0xFFFFFFFFu) { a.counters[1] = 1; continue; }    // too many groups for the small path: the host falls back
                }
                if (a.s_need_cnt || s >= T) atomicAdd(&a.g_cnt[gs], gsz[s]);       // (slots T, T + 1 are occupied iff their size > 0)
                for (int k = 0; k < a.round_states; k++) {
                    const uint64_t cell = st[(size_t)k * T1 + s];
                    unsigned long long *g = reinterpret_cast<unsigned long long *>(&a.g_states[(size_t)k * GS + gs]);
                    if ((uint32_t)k < m_base) {
                        if (KIND == 0) atomicAdd(reinterpret_cast<double *>(g), __longlong_as_double((long long)cell));
                        else atomicAdd(g, (unsigned long long)cell);
                    } else if ((uint32_t)k < m_base + (uint32_t)(NSRC * MM)) {
                        if (cell != M_IDENT) atomicMax(g, (unsigned long long)~cell);
                    } else atomicAdd(g, (unsigned long long)cell);
                }
            }
            if (!have_next_tab) break;
            __syncthreads();
            tb = tbn; tab = ntab; ti = tab.task_beg;
            continue;
        }

        // ---- compaction + outputs: every thread owns a contiguous run of slots, ONE block scan ----
        uint64_t *const o_keys = multi ? a.side_keys : a.out_keys;
        uint8_t *const o_null = multi ? a.side_null : a.out_null;
        uint64_t *const o_states = multi ? a.side_states : a.out_states;
        const size_t o_cap = multi ? a.side_cap : a.cap;
        const bool emit_partials = a.partials != 0 || multi;
        const bool sentinel = misc[21] != 0;
        const bool null_part = tab.part == a.P;
        const uint32_t spt = (T1 + AG_THREADS - 1) / AG_THREADS;
        const uint32_t s_beg = min(tid * spt, T1), s_end = min(s_beg + spt, T1);
        auto occupied = [&](uint32_t s) { return s < T ? keys[s] != EMPTY_KEY : (s == T && sentinel); };
        const bool rounds = !SMALL && a.snap_keys != nullptr;
        size_t pos = 0;
        if (!resume) {
            uint32_t mine = 0;
            for (uint32_t s = s_beg; s < s_end; s++) mine += occupied(s) ? 1u : 0u;
            uint32_t total;
            const uint32_t ex = block_exclusive_scan<AG_THREADS>(mine, misc, &total);
            if (tid == 0) misc[22] = atomicAdd(&a.counters[multi ? 2 : 0], total);
            __syncthreads();
            pos = (size_t)misc[22] + ex;
        }
        if (rounds && !resume) {            // launch 0 of several: the key table and every slot's output position for the launches to come
            uint64_t *sk = a.snap_keys + snap * (T + 2);
            uint32_t *sc = reinterpret_cast<uint32_t *>(a.snap_ctrl + snap * T);
            for (uint32_t s = tid; s < T1; s += AG_THREADS) sk[s] = keys[s];
            for (uint32_t s = tid; s < (T >> 2); s += AG_THREADS) sc[s] = reinterpret_cast<const uint32_t *>(ctrl)[s];
            if (tid == 0) a.snap_pos[snap * (T + 2) + T + 1] = sentinel ? 1u : 0u;
        }
        for (uint32_t s = s_beg; s < s_end; s++) {
            if (rounds && s <= T) {
                uint32_t *sp = a.snap_pos + snap * (T + 2) + s;
                if (resume) { if (!occupied(s)) continue; pos = *sp; }
                else *sp = occupied(s) ? (uint32_t)pos : 0xFFFFFFFFu;
            }
            if (!occupied(s)) continue;
            if (!resume) {
                o_keys[pos] = null_part ? 0ull : (s < T ? keys[s] : EMPTY_KEY);
                o_null[pos] = null_part ? 1 : 0;
            }
            const uint64_t g = gsz[s];
            if (emit_partials) {
                if (!resume) o_states[pos] = g;
                for (int k = 0; k < a.n_states; k++) {
                    if (rounds && a.st_round[k] != a.cur_round) continue;       // (a piece's record is filled in round by round, at launch 0's position)
                    const uint32_t l = (uint32_t)a.st_lds[k];
                    uint64_t cell = st[(size_t)l * T1 + s];
                    const int8_t kd = a.kinds[k];
                    if (kd == SK_MAX_F64 || kd == SK_MAX_I64) cell = ~cell;
                    o_states[(size_t)(k + 1) * o_cap + pos] = state_natural(kd, cell);
                }
            } else {
                for (int f = 0; f < a.n_fin; f++) {
                    const FinDev &fd = a.fin[f];
                    if (rounds && fd.round != a.cur_round) continue;        // this output's states live in another launch's tables
                    auto cell = [&](int8_t l) { return st[(size_t)l * T1 + s]; };
                    double r = 0.0;
                    switch (fd.op) {
                    case PANDRS_HIP_AGG_COUNT: r = (double)g; break;
                    case PANDRS_HIP_AGG_SUM:
                        r = KIND == 0 ? __longlong_as_double((long long)cell(fd.st_add)) : (double)(int64_t)cell(fd.st_add);
                        break;
                    case PANDRS_HIP_AGG_MEAN: {
                        const uint64_t nn = fd.st_nn >= 0 ? cell(fd.st_nn) : g;
                        const double sum = KIND == 0 ? __longlong_as_double((long long)cell(fd.st_add)) : (double)(int64_t)cell(fd.st_add);
                        r = nn ? sum / (double)nn : 0.0;
                        break;
                    }
                    case PANDRS_HIP_AGG_MIN:
                    case PANDRS_HIP_AGG_MAX: {
                        // untouched identity => the reference's sentinel rule: 0.0 (aggregation.rs:640-674)
                        const bool mx = fd.op == PANDRS_HIP_AGG_MAX;
                        const uint64_t raw = cell(mx ? fd.st_max : fd.st_min);
                        if (raw != M_IDENT) {
                            const uint64_t ce = mx ? ~raw : raw;
                            r = KIND == 0 ? dec_f64(ce) : (double)dec_i64(ce);
                        }
                        break;
                    }
                    }
                    a.out_aggs[(size_t)f * a.cap + pos] = r;
                }
            }
            pos++;
        }
        if (!have_next_tab) break;
        __syncthreads();            // the table is re-initialised next
        tb = tbn; tab = ntab; ti = tab.task_beg;
    }
    finish();
}

template <int NSRC, int PROFILE, int ABLATE = 0, int DEPTH = 0>
void launch_one(pandrs_hip_ctx *c, const AggArgs &a, size_t lds, uint32_t grid) {
    // rows in flight per thread: as many as the register budget of a 1024-thread workgroup (128 VGPRs) allows
    constexpr int D = DEPTH > 0 ? DEPTH : (NSRC <= 2 ? 4 : 2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(aggregate2_kernel<NSRC, PROFILE, ABLATE, D>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((aggregate2_kernel<NSRC, PROFILE, ABLATE, D>), dim3(grid), dim3(AG_THREADS), lds, c->stream, a);
}

template <int NSRC>
bool launch_profile(pandrs_hip_ctx *c, const AggArgs &a, int profile, size_t lds, uint32_t grid) {
    switch (profile) {
#define PROF(K, OPS, V) case ((K) << 4 | (OPS) << 1 | (V)): launch_one<NSRC, ((K) << 4 | (OPS) << 1 | (V))>(c, a, lds, grid); return true;
        PROF(0, 1, 0) PROF(0, 1, 1) PROF(0, 7, 0) PROF(0, 7, 1) PROF(0, 6, 0) PROF(0, 6, 1)
        PROF(0, 2, 0) PROF(0, 4, 0) PROF(0, 3, 0) PROF(0, 5, 0)
        PROF(1, 1, 0) PROF(1, 1, 1) PROF(1, 7, 0) PROF(1, 7, 1) PROF(1, 6, 0)
#undef PROF
    default: return false;
    }
}

// ---- the small path's second launch: global table -> result rows (any order), finalised like the compaction above;
// every visited slot is cleared, so the table is armed again when the kernel ends.  Last workgroup publishes the counters.
__global__ __launch_bounds__(256) void small_output_kernel(AggArgs a, int n_src, int profile) {
    const bool op_add = (profile >> 1) & 1, op_min = (profile >> 2) & 1, op_max = (profile >> 3) & 1;
    const int kind = (profile >> 4) & 1, mm = (op_min ? 1 : 0) + (op_max ? 1 : 0);
    (void)op_add;
    const uint32_t S = a.g_slots, s = blockIdx.x * 256 + threadIdx.x;
    const size_t GS = (size_t)S + 2;
    if (s < S + 2) {
        const uint32_t cnt = a.g_cnt[s];
        const uint64_t key = a.g_keys[s];
        uint64_t cells[8];                       // all loads first: one memory round trip, not one per dependent step
#pragma unroll
        for (int k = 0; k < 8; k++) cells[k] = k < a.round_states ? a.g_states[(size_t)k * GS + s] : 0ull;
        const bool occ = s < S ? key != EMPTY_KEY : cnt > 0;
        if (occ) {
            const uint32_t pos = atomicAdd(&a.counters[0], 1u);
            if (pos < a.cap) {
                a.out_keys[pos] = s == S + 1 ? 0ull : (s == S ? EMPTY_KEY : key);
                a.out_null[pos] = s == S + 1 ? 1 : 0;
                auto cell = [&](int8_t k) {
                    uint64_t r = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) r = j == k ? cells[j] : r;
                    return r;
                };
                for (int f = 0; f < a.n_fin; f++) {
                    const FinDev &fd = a.fin[f];
                    double r = 0.0;
                    switch (fd.op) {
                    case PANDRS_HIP_AGG_COUNT: r = (double)cnt; break;
                    case PANDRS_HIP_AGG_SUM:
                        r = kind == 0 ? __longlong_as_double((long long)cell(fd.st_add)) : (double)(int64_t)cell(fd.st_add);
                        break;
                    case PANDRS_HIP_AGG_MEAN: {
                        const uint64_t nn = fd.st_nn >= 0 ? cell(fd.st_nn) : (uint64_t)cnt;
                        const double sum = kind == 0 ? __longlong_as_double((long long)cell(fd.st_add)) : (double)(int64_t)cell(fd.st_add);
                        r = nn ? sum / (double)nn : 0.0;
                        break;
                    }
                    case PANDRS_HIP_AGG_MIN: {          // stored as ~enc; 0 = no value; the reference's sentinel rule: +inf / i64::MAX => 0.0
                        const uint64_t raw = cell(fd.st_min);
                        if (raw) {
                            const uint64_t e = ~raw;
                            if (kind == 0) { const double v = dec_f64(e); r = v == __longlong_as_double(0x7FF0000000000000ll) ? 0.0 : v; }
                            else { const int64_t v = dec_i64(e); r = v == INT64_MAX ? 0.0 : (double)v; }
                        }
                        break;
                    }
                    case PANDRS_HIP_AGG_MAX: {          // stored as enc
                        const uint64_t raw = cell(fd.st_max);
                        if (raw) {
                            if (kind == 0) { const double v = dec_f64(raw); r = v == __longlong_as_double((long long)0xFFF0000000000000ull) ? 0.0 : v; }
                            else { const int64_t v = dec_i64(raw); r = v == INT64_MIN ? 0.0 : (double)v; }
                        }
                        break;
                    }
                    }
                    a.out_aggs[(size_t)f * a.cap + pos] = r;
                }
            } else a.counters[1] = 1;
            // re-arm the slot
            a.g_keys[s] = EMPTY_KEY; a.g_cnt[s] = 0;
            for (int k = 0; k < a.round_states; k++) a.g_states[(size_t)k * GS + s] = 0ull;
        }
    }
    (void)n_src; (void)mm;
    __syncthreads();
    if (threadIdx.x == 0 && a.host_out) {
        __threadfence();
        if (atomicAdd(&a.counters[9], 1u) == gridDim.x - 1) {
            __threadfence();
            for (int i = 0; i < 3; i++) a.host_out[i] = __hip_atomic_load(&a.counters[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.host_out[3] = 0;
            for (int i = 0; i < 16; i++) a.counters[i] = 0;            // the counter block is armed for the next call too
            __hip_atomic_store(&a.host_out[4], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int NSRC, int PROFILE>
void launch_small_one(pandrs_hip_ctx *c, const AggArgs &a, size_t lds, uint32_t grid) {
    constexpr int D = NSRC <= 2 ? 4 : 2;
    static std::atomic<uint64_t> attr_done{0};             // per device: the attribute call costs a driver round trip
    const uint64_t bit = 1ull << (c->device & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(aggregate2_kernel<NSRC, PROFILE, 0, D, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(c->lds_bytes));
        attr_done.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((aggregate2_kernel<NSRC, PROFILE, 0, D, true>), dim3(grid), dim3(AG_THREADS), lds, c->stream, a);
}
template <int NSRC>
bool launch_small_profile(pandrs_hip_ctx *c, const AggArgs &a, int profile, size_t lds, uint32_t grid) {
    switch (profile) {      // the common small-frame shapes: {f64, i64} x {sum, sum+min+max, min+max} (+ null masks for f64)
#define PROF(K, OPS, V) case ((K) << 4 | (OPS) << 1 | (V)): launch_small_one<NSRC, ((K) << 4 | (OPS) << 1 | (V))>(c, a, lds, grid); return true;
        PROF(0, 1, 0) PROF(0, 1, 1) PROF(0, 7, 0) PROF(0, 7, 1) PROF(0, 6, 0) PROF(1, 1, 0) PROF(1, 7, 0)
#undef PROF
    default: return false;
    }
}

}  // namespace

bool aggregate2_small_has(int n_src, int profile) {
    if (n_src < 1 || n_src > 2) return false;
    switch (profile) { case 2: case 3: case 14: case 15: case 12: case 18: case 30: return true; default: return false; }
}
// lanes of a wave in one slot from which they are folded on the VALU instead of serialising on the slot's LDS words
// (experiments/fold_cut_sweep.py: two keys sharing a wave 25 + 25 fold slower than they serialise — 80 % of the rows on 2000 keys 5.1 ms with 16, 4.7 with 32..48)
constexpr uint32_t FOLD_MIN = 40, FOLD_MIN_MULTI = 8;
bool launch_aggregate2_small(pandrs_hip_ctx *c, const AggArgs &a_in, int n_src, int profile, size_t lds, uint32_t grid) {
    AggArgs a = a_in;
    a.fold_min = a.fold_min_multi = 65;        // (never: small tables see few rows per key)
    switch (n_src) {
    case 1: return launch_small_profile<1>(c, a, profile, lds, grid);
    case 2: return launch_small_profile<2>(c, a, profile, lds, grid);
    default: return false;
    }
}
void launch_small_output(pandrs_hip_ctx *c, const AggArgs &a, int n_src, int profile) {
    hipLaunchKernelGGL(small_output_kernel, dim3((a.g_slots + 2 + 255) / 256), dim3(256), 0, c->stream, a, n_src, profile);
}

namespace {
}  // namespace

bool aggregate2_has(int n_src, int profile) {
    if (n_src < 1 || n_src > 4) return false;
    switch (profile) {
    case 2: case 3: case 14: case 15: case 12: case 13: case 4: case 8: case 6: case 10:
    case 18: case 19: case 30: case 31: case 28: return true;
    default: return false;
    }
}

bool launch_aggregate2(pandrs_hip_ctx *c, const AggArgs &a_in, int n_src, int profile, size_t lds, uint32_t grid) {
    AggArgs a = a_in;
    a.fold_min = c->opt.fold_min > 0 ? (uint32_t)c->opt.fold_min : FOLD_MIN;
    a.fold_min_multi = c->opt.fold_min_multi > 0 ? (uint32_t)c->opt.fold_min_multi : FOLD_MIN_MULTI;
    if ((c->opt.agg_depth > 0 || c->opt.agg_ablate > 0) && profile == 14 && (n_src == 4 || n_src == 2)) {
        // experiments only: what the kernel's time is made of (ablate: 1 no min / max work, 2 key lookup + group
        // size only, 3 the HBM stream alone, 4 ... without epilogue, 6 full work on L2-resident rows) and the ring depth
        const int ab = (int)c->opt.agg_ablate, dp = (int)c->opt.agg_depth;
#define EXP(N, A, D) if (n_src == N && ab == A && dp == D) { launch_one<N, 14, A, D>(c, a, lds, grid); return true; }
        EXP(4, 7, 0) EXP(4, 0, 2) EXP(4, 0, 3) EXP(4, 0, 5) EXP(4, 1, 0) EXP(4, 2, 0) EXP(4, 3, 0) EXP(4, 4, 0) EXP(4, 6, 0) EXP(4, 8, 0) EXP(4, 9, 0)
        EXP(2, 0, 2) EXP(2, 0, 3) EXP(2, 0, 6) EXP(2, 1, 0) EXP(2, 2, 0) EXP(2, 3, 0) EXP(2, 6, 0)
#undef EXP
    }
    switch (n_src) {
    case 1: return launch_profile<1>(c, a, profile, lds, grid);
    case 2: return launch_profile<2>(c, a, profile, lds, grid);
    case 3: return launch_profile<3>(c, a, profile, lds, grid);
    case 4: return launch_profile<4>(c, a, profile, lds, grid);
    default: return false;
    }
}

}  // namespace pandrs
