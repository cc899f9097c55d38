// common.hpp — context, workspace arenas, error plumbing for libpandrs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pandrs_hip.h"

namespace pandrs {

// ---- thread-local error string (pandrs_hip_last_error) ---------------------------------------
// A fixed buffer, not a std::string: recording an error must not allocate (the error may BE an allocation failure).
struct ErrorText { char text[512]; };
inline ErrorText &last_error() {
    thread_local ErrorText e{};
    return e;
}
inline int32_t fail(int32_t status, const char *fmt, ...) noexcept {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error().text, sizeof last_error().text, fmt, ap);
    va_end(ap);
    return status;
}

// ---- the exception firewall of the C ABI -------------------------------------------------------
// "Nothing panics / aborts across the ABI" (include/pandrs_hip.h; the reference returns Result<T, pandrs::Error>,
// src/core/error.rs:6).  The host code behind the entry points uses std::vector / std::mutex / new; an exception that
// left an extern "C" frame would unwind into the caller's language (Rust: undefined behaviour or an abort).  Every
// entry point is a function-try-block whose handler calls this: std::bad_alloc -> PANDRS_HIP_ERR_OUT_OF_MEMORY,
// anything else -> PANDRS_HIP_ERR_COMPUTATION, with pandrs_hip_last_error() set.
inline int32_t on_exception(const char *entry) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "%s: host allocation failed (std::bad_alloc)", entry);
    } catch (const std::exception &e) {
        return fail(PANDRS_HIP_ERR_COMPUTATION, "%s: C++ exception: %s", entry, e.what());
    } catch (...) {
        return fail(PANDRS_HIP_ERR_COMPUTATION, "%s: unknown C++ exception", entry);
    }
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::pandrs::fail(e__ == hipErrorOutOfMemory ? PANDRS_HIP_ERR_OUT_OF_MEMORY \
                                                             : PANDRS_HIP_ERR_COMPUTATION, \
                                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                  __FILE__, __LINE__);                                    \
    } while (0)
#define ST_TRY(expr)               \
    do {                           \
        int32_t s__ = (expr);      \
        if (s__) return s__;       \
    } while (0)

// GpuConfig.memory_limit (src/gpu/mod.rs:22): upper bound for any single arena of the library; 0 = none
inline size_t &arena_limit() {
    static size_t limit = 0;
    return limit;
}

// pandrs_hip_config as passed to pandrs_hip_init (capi.hip); thresholds apply only once a config was passed explicitly
int64_t config_min_size_threshold();
bool config_use_pinned_memory();
inline int32_t below_threshold(int64_t n_rows) {
    const int64_t t = config_min_size_threshold();
    if (t > 0 && n_rows < t)
        return fail(PANDRS_HIP_ERR_BELOW_THRESHOLD, "%lld rows is below pandrs_hip_config.min_size_threshold (%lld): keep the CPU path",
                    (long long)n_rows, (long long)t);
    return 0;
}

// device allocations made by the library since it was loaded (arena growths, resident columns, one-off tables):
// a steady-state call makes none — pandrs_hip_alloc_events lets tests assert it
inline std::atomic<int64_t> &alloc_events() {
    static std::atomic<int64_t> n{0};
    return n;
}

// ---- bump arena over one hipMalloc block -------------------------------------------------------
// Sized for 288 GB of HBM: one big block per purpose, grown (never shrunk) between calls, so the
// steady state performs no hipMalloc/hipFree inside a timed call.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;

    int32_t ensure(size_t bytes, hipStream_t stream) {
        off = 0;
        if (bytes <= cap) return 0;
        if (base) {
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipFree(base));
            base = nullptr;
            cap = 0;
        }
        size_t want = bytes + (bytes >> 3) + (1u << 20);
        if (arena_limit() && want > arena_limit())
            return ::pandrs::fail(PANDRS_HIP_ERR_OUT_OF_MEMORY,
                                  "workspace of %zu bytes exceeds pandrs_hip_config.memory_limit (%zu)", want, arena_limit());
        HIP_TRY(hipMalloc((void **)&base, want));
        alloc_events()++;
        cap = want;
        return 0;
    }
    template <typename T>
    T *take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        if (off + bytes > cap) return nullptr;
        T *p = reinterpret_cast<T *>(base + off);
        off += bytes;
        return p;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
    void release() {
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = off = 0;
    }
};

// ---- retained results --------------------------------------------------------------------------
struct GroupbyResult {
    bool valid = false;
    bool partials = false;      // states retained instead of finalised aggregates
    int64_t n_groups = 0;
    int64_t cap = 0;            // row capacity (stride) of the arrays below
    int n_keys = 0, n_aggs = 0, n_state = 0;
    int key_dtype = 0;
    uint64_t *keys = nullptr;   // [n_keys][cap]
    uint8_t *key_null = nullptr;// [n_keys][cap]
    double *aggs = nullptr;     // [n_aggs][cap]
    uint64_t *states = nullptr; // [n_state][cap]  (partials)
};

// group_by's row -> group assignment in CSR form: group g = rows[offsets[g] .. offsets[g+1]), ascending
struct GroupsResult {
    bool valid = false;
    int64_t n_groups = 0, n_rows = 0, cap = 0;
    int n_keys = 0;
    uint64_t *keys = nullptr;    // [n_keys][cap]
    uint8_t *key_null = nullptr; // [n_keys][cap]
    int64_t *offsets = nullptr;  // [n_groups + 1]
    int64_t *rows = nullptr;     // [n_rows]
};

// rows of one shard bucketed by owner rank (pandrs_hip_shuffle_split)
struct ShuffleResult {
    bool valid = false;
    int64_t n_rows = 0;
    int n_payload = 0;
    uint64_t *cells = nullptr;
    uint8_t *key_null = nullptr;
    uint64_t *pay[16]{};
    uint8_t *pay_null[16]{};     // nullptr for payloads without a mask
};

struct JoinResult {
    bool valid = false;
    int64_t n_rows = 0;
    int64_t *left_idx = nullptr, *right_idx = nullptr;
};

struct Options {
    int64_t groups_hint = 0;     // 0 = estimate from a sample
    int64_t scatter_staged = 1;  // stage columns through LDS for coalesced partition writes
    int64_t partitions = 0;      // 0 = auto
    int64_t scatter_threads = 1024;  // 1024: one 8192-row tile per CU; 512: two 4096-row tiles per CU
    int64_t src_per_round = 0;       // 0 = auto; aggregated columns per pass over a partition
    int64_t shared_cursors = 1;      // scatter: per-(partition, XCD group) shared write cursors
    int64_t p_max = 0;               // 0 = default (4096); lower values force the two-level path (testing)
    int64_t no_slice = 0;            // 1 = never split oversized partitions across workgroups
    int64_t slice_rows = 0;          // 0 = auto; rows per slice of an oversized partition
    int64_t no_direct = 0;           // 1 = never take the partition-free low-cardinality path
    int64_t fold_min = 0, fold_min_multi = 0;   // experiments: lanes in one slot from which a wave folds them (0 = default 40 / 8; 65 = never)
    int64_t slice_over = 0;          // experiments: a partition is cut when it holds more than this many average partitions' rows (0 = default 2)
    int64_t wide_slices = 0;         // 1 = an oversized partition is cut into pieces as long as the cutting threshold (4 x the average partition) instead of average-sized ones (A/B)
    int64_t sorted_dictionary = 0;   // 1 = a too-wide column of a composite key is dictionary-encoded by ordering its rows (the path that takes any cardinality) instead of a hashed look-up
    int64_t no_overflow_run = 0;     // 1 = a full LDS table fails the attempt (retry with 4 x the fan-out) instead of handing its unplaced rows to a run of their own
    int64_t tail_groups_hint = 0;    // tests: the group estimate handed to the compact spill's tail run (0 = its own estimate); a low value makes its tables overflow
    int64_t no_census = 0;           // 1 = the group estimate never takes its second stage (the hash-slice census of a tenth of the rows when the sample shows an unresolved tail)
    int64_t no_chao = 0;             // 1 = the group estimate is the uniform-occupancy model alone (no Chao1 term from the sample's singletons / doubletons)
    int64_t no_absorb = 0;           // 1 = never run the hot-key absorb-and-spill pass in front of the radix path
    int64_t no_hot_image = 0;        // 1 = the absorb tables start empty (first come, first served) instead of from the sample's hot keys
    int64_t generic_aggregate = 0;   // 1 = force the descriptor-driven aggregate kernel (testing)
    int64_t median_generic = 0;      // 1 = Median / Nunique: skip the LDS group-sort fast path (testing)
    int64_t load_pct = 0;            // 0 = default LDS table load factor (percent)
    int64_t no_clustered = 0;        // 1 = never the one-pass path for rows clustered by key (groupby.hip, run_clustered)
    int64_t clustered_chunk = 0;     // experiments / tests: rows per chunk there (0 = from the sample's runs per row)
    int64_t clustered_max_runs_pct = 0;   // experiments / tests: the path is taken up to this many runs per 100 rows (0 = default 13)
    int64_t no_profile_rounds = 0;   // 1 = columns without a uniform profile always go to the older kernel (before: the only choice)
    int64_t no_burst_kernel = 0;     // 1 = partitions whose keys arrive in bursts still go to the lean kernel (row per lane)
    int64_t no_window_bound = 0;     // 1 = the estimate never looks at windows of consecutive rows (keys local in position)
    int64_t no_lean_rounds = 0;      // 1 = more than one round always means the older kernel (before: the only choice)
    int64_t no_table_order = 0;      // experiments: aggregate2 draws its tables in partition order instead of largest first
    int64_t p_target = 0;            // 0 = default fan-out target for the rounds heuristic
    int64_t no_runs = 0;             // 1 = never use the run-folding aggregate kernels
    int64_t join_one_pass = 0;       // 1 = probe with the single-pass (decoupled look-back) kernel instead of lookup / scan / emit
    int64_t join_no_pairpart = 0;    // 1 = the L2-region path emits its pairs at one cursor and lets the engine partition them; 2 = tests: undersized regions
    int64_t scatter_wide = 0;        // 1 = the scatter's wide tile (16 K rows, two staging halves) whenever it fits, -1 = never; 0 = at fan-outs >= 1024 (see scatter_wide_ok)
    int64_t two_pass_min_p = 0;      // experiments: the fan-out from which the two passes are taken (default 6144)
    int64_t two_pass = 0;            // -1 = the exact radix partition never takes two passes (64 buckets, then the rest) at fan-outs >= 6144
    int64_t join_pair_p = 0;         // the L2-region probe's minimum pair fan-out (default 256; fewer = longer runs, the engine slices the partitions)
    int64_t join_no_l2 = 0;          // 1 = the fused join never takes the L2-resident-table path; -1 = always tries it (testing)
    int64_t join_generic = 0;        // 1 = always sort the join build side with the general segmented sort (testing)
    int64_t exact_partition = 0;     // 1 = always run the exact histogram (never the sampled-capacity partition)
    int64_t deterministic = 0;       // 1 = f64 Sum / Mean and Std / Var folded in ascending row order per group (bit-identical to the reference's fold)
    int64_t small_chunk = 0;         // experiments: rows per workgroup of the small path (0 = auto)
    int64_t no_small = 0;            // 1 = never take the two-launch small-call path (groupby.hip run_small)
    int64_t agg_v1 = 0;              // 1 = never use the lean persistent aggregate kernel (aggregate2.hip)
    int64_t agg_depth = 0;           // experiments: register-ring depth of aggregate2 (C2 profile)
    int64_t agg_ablate = 0;          // experiments: 1 no min/max, 2 lookup only, 3 stream only (C2 profile of aggregate2)
};

}  // namespace pandrs

struct pandrs_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // one arena per lifetime class: buffers that must survive a nested engine run never share an
    // arena with what that run allocates (work: per-run scratch; temp: direct-path records;
    // side: slice records; super: two-level columns; packed: multi-key cells and dictionaries;
    // pairs: fused-join pairs; groups: retained group index (CSR); shuf: retained shuffle buckets)
    pandrs::Arena work, result, staging, temp, result2, result3, side, super, packed, pairs, groups, shuf, absorb, overflow;
    pandrs::Options opt;
    pandrs_hip_timings timings{};
    pandrs::GroupbyResult gb, gb2, gb3;   // gb2 / gb3: nested results (slice merges, two-level sub-runs)
    pandrs::JoinResult jn;
    pandrs::GroupsResult gr;
    pandrs::ShuffleResult sh;
    // phase timing: pairs of events
    hipEvent_t ev_begin[PANDRS_HIP_MAX_PHASES]{}, ev_end[PANDRS_HIP_MAX_PHASES]{};
    bool ev_used[PANDRS_HIP_MAX_PHASES]{};
    hipEvent_t ev_call_begin = nullptr, ev_call_end = nullptr;
    void *pinned = nullptr;      // small pinned host block for readbacks
    bool timings_lazy = false, timings_pending = false;   // see timings_resolve
    bool pair_fallback = false;       // fused join: the partitioned pair output overflowed a region, the plain emission answered
    int small_skip = 0, small_backoff = 0;     // run_small's back-off after a call that did not fit
    void *small_table = nullptr;      // the small path's armed global table (groupby.hip run_small)
    uint64_t *est_table = nullptr;    // estimate_groups' armed hash table + counters (own allocation)
    uint64_t *census_table = nullptr; // the estimate's second stage (hash-slice census): its own armed table + counters
    int timings_census = 0;           // the last estimate took its figure from the census
    int64_t reserve_groups = 0;       // result rows the next res_slot-0 engine run keeps free behind its groups (a caller appends there)
    double est_far_same = 0.0;        // ... and of rows 32 sample strides apart (a dominant key: equal keys in any row order)
    int64_t est_far_equal = 0;        // the count behind it
    double est_near_same = 0.0;       // of the last estimate's sample: share of adjacent row pairs with equal keys (a dominant key or clustered rows)
    double est_repeat_share = 0.0;    // of the last estimate's sample: rows on keys sighted >= 3 times (a hot set shows here whatever the tail's length)
    bool est_kept = false;            // the table still holds the last estimate's keys (estimate_coverage / estimate_release pending)
    bool clustered_rows = false;      // last estimate: most adjacent rows share their key (sorted / grouped input)
    bool clumped_rows = false;        // last estimate: keys are local in position without neighbours being equal (nearly sorted input): no sampled region plan
    bool capacity_exceeded = false;   // set when a run needed more radix partitions than allowed
    // resident columns (pandrs_hip_column_upload): device data pointer -> {allocation, bytes}; freed by _release / ctx_destroy
    struct Resident { void *base; size_t bytes; };
    std::unordered_map<const void *, Resident> resident;
    size_t resident_bytes = 0;
    int quiet = 0;               // > 0: nested engine runs (slice / direct merges) do not record phase events
    int lds_bytes = 0;           // usable LDS per workgroup
    int n_cu = 0;
};

namespace pandrs {

struct PhaseTimer {
    pandrs_hip_ctx *c;
    int phase;
    PhaseTimer(pandrs_hip_ctx *ctx, int ph) : c(ctx), phase(ph) {
        if (c->quiet) return;
        if (!c->ev_used[phase]) {
            (void)hipEventRecord(c->ev_begin[phase], c->stream);
            c->ev_used[phase] = true;
        }
    }
    ~PhaseTimer() { if (!c->quiet) (void)hipEventRecord(c->ev_end[phase], c->stream); }
};

inline void timings_begin(pandrs_hip_ctx *c) {
    std::memset(&c->timings, 0, sizeof c->timings);
    for (int i = 0; i < PANDRS_HIP_MAX_PHASES; i++) c->ev_used[i] = false;
    c->timings_lazy = c->timings_pending = false;
    (void)hipEventRecord(c->ev_call_begin, c->stream);
}
// Small calls have already seen their completion record in pinned memory: their event times are resolved only when
// somebody asks (pandrs_hip_get_timings), which saves the call a second host-device round trip.
inline int32_t timings_resolve(pandrs_hip_ctx *c) {
    c->timings_pending = false;
    HIP_TRY(hipEventSynchronize(c->ev_call_end));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_call_begin, c->ev_call_end));
    c->timings.total_ms = ms;
    for (int i = 0; i < PANDRS_HIP_MAX_PHASES; i++) {
        if (!c->ev_used[i]) continue;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_begin[i], c->ev_end[i]));
        c->timings.phase_ms[i] = ms;
    }
    return 0;
}
inline int32_t timings_end(pandrs_hip_ctx *c) {
    HIP_TRY(hipEventRecord(c->ev_call_end, c->stream));
    if (c->timings_lazy) { c->timings_lazy = false; c->timings_pending = true; return 0; }
    return timings_resolve(c);
}

// entry points implemented across the .hip files
int32_t groupby_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *keys,
                      int32_t n_keys, int64_t n_rows, const pandrs_hip_column *vals, int32_t n_vals,
                      const pandrs_hip_agg_spec *aggs, int32_t n_aggs, bool partials,
                      int64_t *out_n_groups, int32_t *out_n_state);
int32_t groupby_indices_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *keys, int32_t n_keys,
                              int64_t n_rows, int64_t *out_n_groups);
int32_t shuffle_split_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *key,
                            const pandrs_hip_column *payload, int32_t n_payload, int64_t n_rows, int32_t n_ranks,
                            int32_t drop_null_keys, int64_t *out_counts, int64_t *out_n_rows);
int32_t key_hash_cells_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *keys, int32_t n_keys,
                             int64_t n_rows, uint64_t *out_cells);
int32_t bytes_to_bitmap_entry(pandrs_hip_ctx *c, int32_t mem_space, const uint8_t *bytes, int64_t n, uint8_t *out);
int32_t groupby_merge_entry(pandrs_hip_ctx *c, int32_t mem_space, int32_t key_dtype,
                            const uint64_t *records,
                            int64_t n_rows, const int32_t *val_dtypes, int32_t n_vals,
                            const uint8_t *val_has_nulls, const pandrs_hip_agg_spec *aggs,
                            int32_t n_aggs, int64_t *out_n_groups);
int32_t partials_split_entry(pandrs_hip_ctx *c, int32_t mem_space, int32_t n_ranks,
                             uint64_t *out_records, int64_t *out_counts);
// (in-library exchange, dist.hip: the partial records as one block per owner, no host round trip; the merge of received blocks)
int32_t partials_split_blocks_entry(pandrs_hip_ctx *c, int32_t n_ranks, uint64_t *out, int64_t *count_row);
int32_t groupby_merge_blocks_entry(pandrs_hip_ctx *c, int32_t key_dtype, const uint64_t *blocks, const int64_t *roff, int32_t n_src,
                                   const int32_t *val_dtypes, int32_t n_vals, const uint8_t *val_has_nulls,
                                   const pandrs_hip_agg_spec *aggs, int32_t n_aggs, int64_t *out_n_groups);
int32_t join_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *lk, int64_t nl,
                   const pandrs_hip_column *rk, int64_t nr, int32_t how, int64_t *out_n);
int32_t join_groupby_sum_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *lk,
                               const pandrs_hip_column *lv, int64_t nl,
                               const pandrs_hip_column *rk, const pandrs_hip_column *rg,
                               int64_t nr, int64_t *out_n_groups);
int32_t gather_entry(pandrs_hip_ctx *c, int32_t mem_space, int kind, const void *src,
                     const uint8_t *mask, const int64_t *idx, int64_t n, uint64_t fill_bits,
                     void *out, int64_t n_src = -1, const int64_t *only_where_negative = nullptr);
int32_t join_gather_entry(pandrs_hip_ctx *c, int32_t src_mem_space, const pandrs_hip_column *src, int64_t n_src, int32_t side,
                          uint64_t fill_bits, int32_t out_mem_space, void *out, const pandrs_hip_column *key_right = nullptr, int64_t n_right = 0);
int32_t gather_column_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *src, int64_t n_src,
                            const int64_t *idx, int64_t n, uint64_t fill_bits, void *out);
int32_t reduce_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *col, int64_t n,
                     double out[4], int64_t *out_count, double *out_sumsq = nullptr);
int32_t reduce_stats_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *col, int64_t n,
                           pandrs_hip_column_stats *st);

inline size_t dtype_bytes(int dtype, int64_t n) {
    switch (dtype) {
    case PANDRS_HIP_I64: case PANDRS_HIP_F64: case PANDRS_HIP_CELL64: return size_t(n) * 8;
    case PANDRS_HIP_U32CODE: return size_t(n) * 4;
    default: return size_t((n + 7) / 8);
    }
}

}  // namespace pandrs
