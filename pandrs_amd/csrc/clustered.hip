// clustered.hip — rows CLUSTERED by key (sorted input, input grouped by key, time-ordered keys): the per-group fold of
// src/optimized/split_dataframe/group/aggregation.rs:500-754 in ONE pass over the ORIGINAL columns, no radix partition.
//
// A hash partition is the wrong tool for such rows: the scatter's tiles each hold a handful of partitions (its sampled
// region plan assumes random order), and whichever kernel aggregates them finds the 64 lanes of a wave on one LDS slot.
// Here the row order is used instead:
//   * the rows are cut into chunks short enough that a chunk's runs of equal keys fit one LDS table (the host sizes the
//     chunk from the estimate's adjacent-pair statistic); one persistent workgroup per CU draws chunks from a ticket counter;
//   * every THREAD takes CL_R CONSECUTIVE rows and folds equal neighbours in registers — plain per-lane arithmetic, no
//     cross-lane traffic — and goes to the table once per run it ends (Swiss-table probe + one LDS atomic per state);
//   * a chunk's groups leave as partial records (key cell, null flag, group size, states in ABI order) at a device
//     counter; one merge of the records — groups + chunk boundaries + keys that come back in a later run — is the result
//     (groupby.hip, run_clustered).
// Same uniform profiles, LDS layout and state encodings as the lean aggregate (aggregate2.hip), whose Swiss table this shares.
//
// PARTS: the same fold over the rows of radix PARTITIONS instead of chunks of the original columns — for keys that arrive in BURSTS
// inside their partition (keys local in position: nearly sorted input; short runs).  There the lean aggregate's fast path never gets
// going: a key is new when its burst arrives, its first 64 rows take the retry queue, and the rest of a 100-row burst is one more wave.
// Drop-in for aggregate2_kernel behind the exact partition: same work list (tables / tasks / order), same outputs (final aggregates or
// partial records at counters[0], partial records of an oversized partition's pieces at side_* / counters[2]).
#include "aggregate.hpp"
#include "swiss.hpp"
#include <atomic>

namespace pandrs {

namespace {

constexpr int CL_R = 8;        // consecutive rows per thread: 32 bytes of every column (two 16-byte loads when the columns are 16-byte aligned)

template <int NSRC, int PROFILE, bool PARTS>
__global__ __launch_bounds__(AG_THREADS) void clustered_kernel(AggArgs a) {
    constexpr bool HAS_V = (PROFILE & 1) != 0, OP_ADD = ((PROFILE >> 1) & 1) != 0;
    constexpr bool OP_MIN = ((PROFILE >> 2) & 1) != 0, OP_MAX = ((PROFILE >> 3) & 1) != 0;
    constexpr int KIND = (PROFILE >> 4) & 1;                 // 0 f64, 1 i64
    constexpr int MM = (OP_MIN ? 1 : 0) + (OP_MAX ? 1 : 0);
    constexpr uint64_t M_IDENT = KIND == 0 ? 0xFFF0000000000000ull : ~0ull;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t T = a.T, T1 = T + 3, tid = threadIdx.x;    // slot T: the key equal to the table sentinel; T + 1: the NULL key
    // LDS as in aggregate2's SMALL mode: keys[T1] | states[round_states][T1] | gsz[T1] (u32) | ctrl[T] (u8, 16-byte aligned) | misc[40]
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint64_t *st = keys + T1;
    uint32_t *gsz = reinterpret_cast<uint32_t *>(st + (size_t)a.round_states * T1);
    uint8_t *ctrl = smem + ((static_cast<uint32_t>(reinterpret_cast<unsigned char *>(gsz + ((T1 + 3) & ~3u)) - smem) + 15u) & ~15u);
    uint32_t *misc = reinterpret_cast<uint32_t *>(ctrl + T);
    // misc[0..16] scan scratch, [20] table full, [21] sentinel-valued key present, [22] output base, [23] NULL key present, [33] next chunk
    constexpr uint32_t m_base = OP_ADD ? (uint32_t)NSRC : 0u;
    const uint64_t *vals[NSRC];
    const uint8_t *valid[NSRC];
    int nn_idx[NSRC];
#pragma unroll
    for (int c = 0; c < NSRC; c++) {
        const SrcDev &sd = a.src[(PARTS ? a.src_base : 0) + c];       // (PARTS in rounds: this launch's columns)
        vals[c] = sd.vals; valid[c] = sd.valid; nn_idx[c] = HAS_V ? sd.st_nn : -1;
    }
    const bool rounds = PARTS && a.snap_keys != nullptr, resume = rounds && a.cur_round > 0;      // as in aggregate2_kernel (aggregate.hpp, snap_*)
    const uint32_t seed = a.seed;
    const bool key8 = a.dkey.dtype == PANDRS_HIP_I64 || a.dkey.dtype == DT_CELL || a.dkey.dtype == PANDRS_HIP_F64;
    const bool key_nulls = a.dkey.null_bits != nullptr || a.dkey.null_bytes != nullptr;

    // every workgroup ends here: the last one publishes the call's counters to the host
    auto finish = [&]() {
        __syncthreads();
        if (tid == 0 && (a.host_out || rounds)) {
            __threadfence();
            if (atomicAdd(&a.counters[8], 1u) == gridDim.x - 1) {
                __threadfence();
                if (a.host_out) {
                    for (int i = 0; i < 3; i++) a.host_out[i] = __hip_atomic_load(&a.counters[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.host_out[3] = 0; a.host_out[6] = 0;
                    __hip_atomic_store(&a.host_out[4], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                } else { a.counters[8] = 0; a.counters[10] = 0; __threadfence(); }       // rounds: every launch but the last re-arms the launch counters
            }
        }
    };
    const uint32_t n_chunks = PARTS ? a.n_tasks[1] : (a.s_rows + a.s_chunk - 1) / a.s_chunk;       // PARTS: the work list's tables
    uint32_t cb = blockIdx.x;
    if (cb >= n_chunks) { finish(); return; }
    if (resume && __hip_atomic_load(&a.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { finish(); return; }      // launch 0 lost the attempt

    // one finished run of a thread -> its group's states: find or claim the key's slot, one LDS atomic per state
    // (the fold of aggregation.rs:625-674 with the run's totals in place of one row)
    auto flush = [&](uint64_t k, bool knull, uint32_t cnt, const uint32_t (&nn)[NSRC], const uint64_t (&add)[NSRC], const uint64_t (&mn)[NSRC],
                     const uint64_t (&mx)[NSRC]) {
        uint32_t slot = T;
        if (knull) { slot = T + 1; misc[23] = 1; }               // NULL key: its own group (grouping.rs:74)
        else if (k == EMPTY_KEY) misc[21] = 1;
        // (chunks: a probe chain of 8 groups means the table is nearly full — the sample's runs per row were off, the host falls back;
        // PARTS: the tables are planned at load 0.7 and there is no overflow run behind them: the whole table is the window)
        else slot = swiss_find(k, keys, ctrl, T, seed, PARTS ? 0xFFFFFFFFu : 8u);
        if (slot > T + 1) { misc[20] = 1; return; }              // table full: the host takes the ordinary path
        atomicAdd(&gsz[slot], cnt);
#pragma unroll
        for (int c = 0; c < NSRC; c++) {
            if (OP_ADD) {
                if (KIND == 0) atomicAdd(reinterpret_cast<double *>(&st[(size_t)c * T1 + slot]), __longlong_as_double((long long)add[c]));
                else atomicAdd((unsigned long long *)&st[(size_t)c * T1 + slot], (unsigned long long)add[c]);
            }
            if (HAS_V && nn_idx[c] >= 0 && nn[c]) atomicAdd((unsigned long long *)&st[(size_t)nn_idx[c] * T1 + slot], (unsigned long long)nn[c]);
            // min states hold enc, max states ~enc (both through ds_min_u64); ~0 / 0 = the run had no comparable value
            if (OP_MIN && mn[c] != ~0ull) atomicMin((unsigned long long *)&st[(size_t)(m_base + c * MM) * T1 + slot], (unsigned long long)mn[c]);
            if (OP_MAX && mx[c] != 0ull) atomicMin((unsigned long long *)&st[(size_t)(m_base + c * MM + MM - 1) * T1 + slot], (unsigned long long)~mx[c]);
        }
    };

    for (;;) {                                         // one LDS table per chunk
        if (tid == 0) misc[33] = gridDim.x + atomicAdd(&a.counters[10], 1u);       // the next chunk: whoever comes first
        const size_t snap = rounds ? (size_t)(a.order ? a.order[cb] : cb) : 0;
        if (resume) {           // a later round: launch 0's key table — every key of this table has its slot already
            const uint64_t *sk = a.snap_keys + snap * (T + 2);
            const uint32_t *sc = reinterpret_cast<const uint32_t *>(a.snap_ctrl + snap * T);
            for (uint32_t s = tid; s < T1; s += AG_THREADS) { keys[s] = s < T + 2 ? sk[s] : EMPTY_KEY; gsz[s] = 0; }
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
        const uint32_t cbn = misc[33];
        const AggTable tab = PARTS ? a.tables[a.order ? a.order[cb] : cb] : AggTable{0u, 1u, 0u, 0u};
        const bool multi = PARTS && tab.multi != 0, part_null = PARTS && tab.part == a.P;
      for (uint32_t ti = 0; ti < tab.n_tasks; ti++) {          // the row ranges that feed this table (chunks: one)
        uint32_t r_beg, r_end;
        if (PARTS) { const AggTask tk = a.tasks[tab.task_beg + ti]; r_beg = tk.beg; r_end = tk.end; }
        else { r_beg = cb * a.s_chunk; r_end = min(r_beg + a.s_chunk, a.s_rows); }

        // tiles of AG_THREADS x CL_R rows; no barrier inside a chunk: the waves drift apart and hide each other's loads
        // (PARTS: a range starts anywhere — the tiles start at the 64-byte boundary below it and the rows in front are masked)
        for (uint32_t t0 = PARTS ? (r_beg & ~(uint32_t)(CL_R - 1)) : r_beg; t0 < r_end; t0 += AG_THREADS * CL_R) {
            if (*reinterpret_cast<volatile uint32_t *>(&misc[20])) break;
            const uint32_t i0 = t0 + tid * CL_R;
            if (i0 >= r_end) continue;
            const uint32_t nrow = min((uint32_t)CL_R, r_end - i0);
            const uint32_t lo = (PARTS && r_beg > i0) ? r_beg - i0 : 0u;         // the thread's rows [lo, nrow) belong to the range
            if (lo >= nrow) continue;
            uint64_t k[CL_R], v[CL_R][NSRC];
            uint32_t okm[CL_R];              // bit c: value c is valid; bit 31: the key is NULL
            if (PARTS || (a.s_vec && nrow == CL_R)) {
                // (s_chunk is a multiple of 1024 and the host checked the columns' 16-byte alignment: i0 is a multiple of 4)
                if (key8) {
                    const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint64_t *>(a.dkey.data) + i0);
#pragma unroll
                    for (int j = 0; j < CL_R / 2; j++) {
                        const uint4 x = p[j];
                        k[2 * j] = ((uint64_t)x.y << 32) | x.x; k[2 * j + 1] = ((uint64_t)x.w << 32) | x.z;
                    }
                    if (a.dkey.dtype == PANDRS_HIP_F64) {
#pragma unroll
                        for (int r = 0; r < CL_R; r++) k[r] = ((k[r] & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) ? CANON_NAN : k[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < CL_R; r++) k[r] = key_cell(a.dkey, i0 + r);
                }
#pragma unroll
                for (int c = 0; c < NSRC; c++) {
                    const uint4 *p = reinterpret_cast<const uint4 *>(vals[c] + i0);
#pragma unroll
                    for (int j = 0; j < CL_R / 2; j++) {
                        const uint4 x = p[j];
                        v[2 * j][c] = ((uint64_t)x.y << 32) | x.x; v[2 * j + 1][c] = ((uint64_t)x.w << 32) | x.z;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < CL_R; r++) {
                    const uint32_t i = min(i0 + r, r_end - 1);        // rows past the end are loaded (in bounds) and never folded
                    k[r] = key_cell(a.dkey, i);
#pragma unroll
                    for (int c = 0; c < NSRC; c++) v[r][c] = vals[c][i];
                }
            }
#pragma unroll
            for (int r = 0; r < CL_R; r++) {
                const uint32_t i = min(max(i0 + r, r_beg), r_end - 1);
                okm[r] = 0x7FFFFFFFu;
                if (PARTS ? part_null : (key_nulls && key_is_null(a.dkey, i))) okm[r] |= 0x80000000u;
                if (HAS_V) {
#pragma unroll
                    for (int c = 0; c < NSRC; c++) if (PARTS ? valid[c][i] == 0 : bit_at(valid[c], i)) okm[r] &= ~(1u << c);       // PARTS: validity bytes, 1 = valid
                }
            }
            // ---- the thread's rows in order: equal neighbours fold in registers, a run that ends goes to the table
            uint64_t ck = 0, add[NSRC], mn[NSRC], mx[NSRC];
            uint32_t cnt = 0, nn[NSRC];
            bool cnull = false;
#pragma unroll
            for (int c = 0; c < NSRC; c++) { add[c] = 0; mn[c] = ~0ull; mx[c] = 0; nn[c] = 0; }
#pragma unroll
            for (int r = 0; r <= CL_R; r++) {
                const bool in = r < CL_R && (uint32_t)r < nrow && (uint32_t)r >= lo;
                const bool rnull = in && (okm[r < CL_R ? r : 0] >> 31) != 0;
                const uint64_t rk = k[r < CL_R ? r : 0];
                const bool same = in && cnt > 0 && (rnull ? cnull : (!cnull && rk == ck));
                if (cnt > 0 && !same) {                       // the run ends here (or the thread's rows do)
                    flush(ck, cnull, cnt, nn, add, mn, mx);
                    cnt = 0;
#pragma unroll
                    for (int c = 0; c < NSRC; c++) { add[c] = 0; mn[c] = ~0ull; mx[c] = 0; nn[c] = 0; }
                }
                if (!in) continue;
                ck = rk; cnull = rnull; cnt++;
                const uint32_t om = okm[r < CL_R ? r : 0];
#pragma unroll
                for (int c = 0; c < NSRC; c++) {
                    const uint64_t x = v[r < CL_R ? r : 0][c];
                    const bool ok = !HAS_V || ((om >> c) & 1);
                    if (!ok) continue;
                    nn[c]++;
                    if (OP_ADD) {
                        if (KIND == 0) add[c] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)add[c]) + __longlong_as_double((long long)x));
                        else add[c] += x;
                    }
                    if (MM > 0) {
                        // Rust's f64::min / max ignore NaN operands (aggregation.rs:653, :666)
                        bool cmp = true;
                        if (KIND == 0) { const double d = __longlong_as_double((long long)x); cmp = d == d; }
                        if (cmp) {
                            const uint64_t e = enc_val<KIND>(x);
                            if (OP_MIN) mn[c] = e < mn[c] ? e : mn[c];
                            if (OP_MAX) mx[c] = e > mx[c] ? e : mx[c];
                        }
                    }
                }
            }
        }
      }
        __syncthreads();
        if (misc[20]) { if (tid == 0) a.counters[1] = 1; finish(); return; }

        // ---- compaction (one block scan, every thread a run of slots).  Chunks: the groups as partial records at counters[2].
        // PARTS: as aggregate2_kernel — final aggregates or partial records at counters[0]; a piece of an oversized partition: records at counters[2]
        const bool sentinel = misc[21] != 0, nullseen = misc[23] != 0;
        const uint32_t spt = (T1 + AG_THREADS - 1) / AG_THREADS;
        const uint32_t s_beg = min(tid * spt, T1), s_end = min(s_beg + spt, T1);
        auto occupied = [&](uint32_t s) { return s < T ? keys[s] != EMPTY_KEY : (s == T ? sentinel : (s == T + 1 && nullseen)); };
        const bool to_out = PARTS && !multi;
        uint64_t *const o_keys = to_out ? a.out_keys : a.side_keys;
        uint8_t *const o_null = to_out ? a.out_null : a.side_null;
        uint64_t *const o_states = to_out ? a.out_states : a.side_states;
        const size_t o_cap = to_out ? a.cap : a.side_cap;
        const bool emit_partials = !PARTS || a.partials != 0 || multi;
        size_t pos = 0;
        if (!resume) {
            uint32_t mine = 0;
            for (uint32_t s = s_beg; s < s_end; s++) mine += occupied(s) ? 1u : 0u;
            uint32_t total;
            const uint32_t ex = block_exclusive_scan<AG_THREADS>(mine, misc, &total);
            if (tid == 0) misc[22] = atomicAdd(&a.counters[to_out ? 0 : 2], total);
            __syncthreads();
            // (chunks: the record buffer is sized for the runs the sample promised, not for every chunk's full table)
            if ((size_t)misc[22] + total > o_cap) { if (tid == 0) a.counters[1] = 1; finish(); return; }
            pos = (size_t)misc[22] + ex;
        }
        if (rounds && !resume) {            // launch 0 of several: the key table for the launches to come (the positions follow below)
            uint64_t *sk = a.snap_keys + snap * (T + 2);
            uint32_t *sc = reinterpret_cast<uint32_t *>(a.snap_ctrl + snap * T);
            for (uint32_t s = tid; s < T + 2; s += AG_THREADS) sk[s] = keys[s];
            for (uint32_t s = tid; s < (T >> 2); s += AG_THREADS) sc[s] = reinterpret_cast<const uint32_t *>(ctrl)[s];
        }
        for (uint32_t s = s_beg; s < s_end; s++) {
            if (rounds && s < T + 2) {
                uint32_t *sp = a.snap_pos + snap * (T + 2) + s;
                if (resume) { const uint32_t p = *sp; if (p == 0xFFFFFFFFu) continue; pos = p; }
                else *sp = occupied(s) ? (uint32_t)pos : 0xFFFFFFFFu;
            }
            if (resume ? s >= T + 2 : !occupied(s)) continue;
            if (!resume) {
                o_keys[pos] = s < T ? keys[s] : (s == T ? EMPTY_KEY : 0ull);
                o_null[pos] = s == T + 1 ? 1 : 0;
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
                // the reference's finalisation (aggregation.rs:507-556, :625-674, :743), as in aggregate2_kernel
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
        if (cbn >= n_chunks) break;
        __syncthreads();            // the table is re-initialised next
        cb = cbn;
    }
    finish();
}

template <int NSRC, int PROFILE, bool PARTS>
void launch_one(pandrs_hip_ctx *c, const AggArgs &a, size_t lds, uint32_t grid) {
    static std::atomic<uint64_t> attr_done{0};             // per device: the attribute call costs a driver round trip
    const uint64_t bit = 1ull << (c->device & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(clustered_kernel<NSRC, PROFILE, PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(c->lds_bytes));
        attr_done.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((clustered_kernel<NSRC, PROFILE, PARTS>), dim3(grid), dim3(AG_THREADS), lds, c->stream, a);
}

template <int NSRC, bool PARTS>
bool launch_profile(pandrs_hip_ctx *c, const AggArgs &a, int profile, size_t lds, uint32_t grid) {
    switch (profile) {      // {f64, i64} x {sum, sum+min+max} (+ null masks for f64), f64 min+max, min alone, max alone
#define PROF(K, OPS, V) case ((K) << 4 | (OPS) << 1 | (V)): launch_one<NSRC, ((K) << 4 | (OPS) << 1 | (V)), PARTS>(c, a, lds, grid); return true;
        PROF(0, 1, 0) PROF(0, 1, 1) PROF(0, 7, 0) PROF(0, 7, 1) PROF(0, 6, 0) PROF(0, 2, 0) PROF(0, 4, 0) PROF(1, 1, 0) PROF(1, 7, 0)
#undef PROF
    default: return false;
    }
}

}  // namespace

bool clustered_has(int n_src, int profile) {
    if (n_src < 1 || n_src > 4) return false;
    switch (profile) { case 2: case 3: case 14: case 15: case 12: case 4: case 8: case 18: case 30: return true; default: return false; }
}

bool launch_clustered(pandrs_hip_ctx *c, const AggArgs &a, int n_src, int profile, size_t lds, uint32_t grid) {
    switch (n_src) {
    case 1: return launch_profile<1, false>(c, a, profile, lds, grid);
    case 2: return launch_profile<2, false>(c, a, profile, lds, grid);
    case 3: return launch_profile<3, false>(c, a, profile, lds, grid);
    case 4: return launch_profile<4, false>(c, a, profile, lds, grid);
    default: return false;
    }
}

// the same fold over the rows of radix partitions (the lean aggregate's work list and outputs): keys that arrive in bursts
bool launch_clustered_parts(pandrs_hip_ctx *c, const AggArgs &a_in, int n_src, int profile, size_t lds, uint32_t grid) {
    AggArgs a = a_in;
    a.dkey = KeyDesc{a.pkeys, nullptr, nullptr, DT_CELL};
    switch (n_src) {
    case 1: return launch_profile<1, true>(c, a, profile, lds, grid);
    case 2: return launch_profile<2, true>(c, a, profile, lds, grid);
    case 3: return launch_profile<3, true>(c, a, profile, lds, grid);
    case 4: return launch_profile<4, true>(c, a, profile, lds, grid);
    default: return false;
    }
}

}  // namespace pandrs
