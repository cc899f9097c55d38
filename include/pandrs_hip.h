/*
 * pandrs_hip.h — C ABI of libpandrs_hip.so, the MI355X (gfx950) groupby-aggregate /
 * hash-join engine that sits behind PandRS's GroupBy / agg() / join API.
 *
 * The reference (cool-japan/pandrs) exposes NO FFI for this path: the
 * seams a maintainer would bind are Rust methods.  Every entry point below names the
 * reference interface it replaces (file:line into the reference tree).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - every function returns a pandrs_hip_status (0 = ok); on failure a thread-local
 *     message is available from pandrs_hip_last_error().  Nothing panics/aborts
 *     across the ABI (reference: Result<T, pandrs::Error>, src/core/error.rs:6): every
 *     entry point catches C++ exceptions of its host code (std::bad_alloc ->
 *     PANDRS_HIP_ERR_OUT_OF_MEMORY, anything else -> PANDRS_HIP_ERR_COMPUTATION).
 *   - `mem_space` says where the caller's column / output pointers live:
 *     PANDRS_HIP_MEM_HOST (library stages H2D/D2H itself) or PANDRS_HIP_MEM_DEVICE
 *     (pointers are HBM addresses on the context's device; nothing crosses PCIe).
 *     Device inputs must be COMPLETE when a call starts: the context's stream is
 *     non-blocking, so a caller that produced them on another stream synchronises that
 *     stream first.  Device outputs are complete when the call returns.
 *   - caller pointers are never retained past return (reference columns are Arc<[T]>
 *     borrowed for the call, src/column/int64_column.rs:52-56).
 *   - a context owns one HIP stream and a workspace arena; calls on ONE context are
 *     serialised by an internal mutex, distinct contexts run concurrently
 *     (reference re-entrancy requirement: tests/concurrency_test.rs:351-398).
 *   - null masks are LSB-first bitmaps, bit = 1 => null (src/core/column.rs:163-177),
 *     may be NULL (= no nulls).
 */
#ifndef PANDRS_HIP_H
#define PANDRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PANDRS_HIP_ABI_VERSION 1

/* status codes; the names map onto pandrs::Error variants (src/core/error.rs) */
typedef enum pandrs_hip_status {
    PANDRS_HIP_OK = 0,
    PANDRS_HIP_ERR_INVALID_ARGUMENT = 1,   /* Error::InvalidInput / ColumnNotFound at the shim */
    PANDRS_HIP_ERR_TYPE_MISMATCH = 2,      /* Error::ColumnTypeMismatch (join.rs:98-104) */
    PANDRS_HIP_ERR_OPERATION_FAILED = 3,   /* Error::OperationFailed (aggregation.rs:744-752, lazy.rs:377-382) */
    PANDRS_HIP_ERR_COMPUTATION = 4,        /* Error::Computation(String) — device failures (src/gpu/mod.rs:206-210) */
    PANDRS_HIP_ERR_OUT_OF_MEMORY = 5,
    PANDRS_HIP_ERR_NOT_INITIALIZED = 6,
    PANDRS_HIP_ERR_BELOW_THRESHOLD = 7     /* fewer rows than GpuConfig.min_size_threshold: the caller keeps its CPU path
                                              (src/optimized/split_dataframe/gpu.rs:30-32); nothing was computed */
} pandrs_hip_status;

/* column element types.  Layouts are the reference's own:
 *   I64      Int64Column.data   Arc<[i64]>  (src/column/int64_column.rs:52-56)
 *   F64      Float64Column.data Arc<[f64]>  (src/column/float64_column.rs:9-13)
 *   U32CODE  StringColumn.indices Arc<[u32]> — global string-pool codes, equal string
 *            <=> equal code (src/column/string_column.rs:26-32, string_pool.rs:28-53)
 *   BOOLBITS BooleanColumn.data BitMask, LSB-first packed (src/column/boolean_column.rs:10-15)
 *   CELL64   (no reference counterpart; KEY columns only) already-normalised 8-byte key cells — the form
 *            group keys are returned in and pandrs_hip_shuffle_fetch delivers: i64 value / canonical
 *            f64 bits / zero-extended code / bool bit.  Equal cells <=> equal keys. */
typedef enum pandrs_hip_dtype {
    PANDRS_HIP_I64 = 0,
    PANDRS_HIP_F64 = 1,
    PANDRS_HIP_U32CODE = 2,
    PANDRS_HIP_BOOLBITS = 3,
    PANDRS_HIP_CELL64 = 4
} pandrs_hip_dtype;

/* AggregateOp, same order as src/optimized/split_dataframe/group/types.rs:11-34 */
typedef enum pandrs_hip_agg_op {
    PANDRS_HIP_AGG_SUM = 0,
    PANDRS_HIP_AGG_MEAN = 1,
    PANDRS_HIP_AGG_MIN = 2,
    PANDRS_HIP_AGG_MAX = 3,
    PANDRS_HIP_AGG_COUNT = 4,
    PANDRS_HIP_AGG_STD = 5,
    PANDRS_HIP_AGG_VAR = 6,
    PANDRS_HIP_AGG_MEDIAN = 7,
    PANDRS_HIP_AGG_FIRST = 8,
    PANDRS_HIP_AGG_LAST = 9,
    PANDRS_HIP_AGG_CUSTOM = 10,  /* always PANDRS_HIP_ERR_OPERATION_FAILED (aggregation.rs:744) */
    /* beyond AggregateOp: the legacy frame's AggFunc::Nunique (src/dataframe/groupby.rs:514-519, SURVEY.md
     * 8f item 2) — the number of distinct non-null values of the group (sort + dedup: values equal under
     * `==`, so -0.0 and 0.0 are one value and every NaN is its own), 0.0 for a group without values (:467). */
    PANDRS_HIP_AGG_NUNIQUE = 11
} pandrs_hip_agg_op;

/* JoinType, src/optimized/split_dataframe/join.rs:11-20 */
typedef enum pandrs_hip_join_type {
    PANDRS_HIP_JOIN_INNER = 0,
    PANDRS_HIP_JOIN_LEFT = 1,
    PANDRS_HIP_JOIN_RIGHT = 2,
    PANDRS_HIP_JOIN_OUTER = 3
} pandrs_hip_join_type;

typedef enum pandrs_hip_mem_space {
    PANDRS_HIP_MEM_HOST = 0,
    PANDRS_HIP_MEM_DEVICE = 1
} pandrs_hip_mem_space;

/* Mirrors GpuConfig (src/gpu/mod.rs:18-44). */
typedef struct pandrs_hip_config {
    int32_t enabled;             /* GpuConfig.enabled */
    int32_t device_id;           /* GpuConfig.device_id */
    int64_t memory_limit;        /* GpuConfig.memory_limit, bytes; 0 = no limit */
    int32_t fallback_to_cpu;     /* GpuConfig.fallback_to_cpu — honoured by the CALLER (shim keeps the
                                    reference CPU path); this library never computes on the CPU */
    int32_t use_pinned_memory;   /* GpuConfig.use_pinned_memory: host columns of a MEM_HOST call are page-locked
                                    (hipHostRegister) for the duration of the call, so their H2D copies are DMA */
    int64_t min_size_threshold;  /* GpuConfig.min_size_threshold (src/gpu/mod.rs:41).  Honoured once a config has been
                                    passed to pandrs_hip_init: the frame-level entry points (groupby_agg, groupby_indices,
                                    join_indices, join_groupby_sum, reduce_*) return PANDRS_HIP_ERR_BELOW_THRESHOLD for
                                    fewer rows.  Without an explicit config the threshold is 0: this library has no CPU
                                    path of its own to prefer. */
} pandrs_hip_config;

/* One typed column view.  `data` element type per `dtype`; `null_mask` may be NULL. */
typedef struct pandrs_hip_column {
    const void *data;
    const uint8_t *null_mask;
    int32_t dtype;      /* pandrs_hip_dtype */
    int32_t reserved;
} pandrs_hip_column;

/* One requested aggregate: (index into vals[], op).  Replaces the
 * (String column, AggregateOp, String alias) triples of GroupBy::aggregate
 * (aggregation.rs:763-767); alias naming stays on the host side. */
typedef struct pandrs_hip_agg_spec {
    int32_t col;
    int32_t op;         /* pandrs_hip_agg_op */
} pandrs_hip_agg_spec;

/* Per-call measurements, for the bench (SURVEY.md §8b "pandrs_hip_get_timings"). */
#define PANDRS_HIP_MAX_PHASES 12
typedef struct pandrs_hip_timings {
    double total_ms;                          /* hipEvent time, whole call on the ctx stream */
    double phase_ms[PANDRS_HIP_MAX_PHASES];   /* per phase, see PANDRS_HIP_PHASE_*: from the phase's first launch to its last (phases
                                                 of one call may overlap); small calls (the two-launch path) record none */
    int64_t algorithmic_bytes;                /* SURVEY.md §8d formula for this call */
    int64_t n_partitions;                     /* radix fan-out chosen (0: the small-call path; -1: hot-key absorb pass with a COMPACT spill —
                                                 the rest of the rows went through a run of their own; -2: rows clustered by key, one pass
                                                 over the original columns and no partition at all; fused join: the probe side's fan-out,
                                                 0 = general fallback) */
    int64_t table_slots;                      /* LDS hash-table slots per partition */
    int64_t retries;                          /* overflow retries taken; 100 + retries: full LDS tables handed their unplaced rows to a
                                                 run of their own instead (groupby; the estimate was too low).  (fused join: 1 = the partitioned pair output overflowed a
                                                 region and the one-cursor emission answered; 2 = the groupby engine refused the
                                                 pre-partitioned pairs — a full LDS table — and the whole call was repeated with
                                                 the one-cursor emission) */
    int64_t estimated_groups;
    int64_t absorbed_rows;                    /* rows folded by the hot-key absorb pass in front of the radix path (0: it did not run) */
} pandrs_hip_timings;

enum {
    PANDRS_HIP_PHASE_STAGE_IN = 0,    /* H2D staging (host mem_space only) */
    PANDRS_HIP_PHASE_ESTIMATE = 1,    /* sampled cardinality estimate */
    PANDRS_HIP_PHASE_HISTOGRAM = 2,   /* radix histogram */
    PANDRS_HIP_PHASE_SCAN = 3,        /* exclusive scan of bucket counts */
    PANDRS_HIP_PHASE_SCATTER = 4,     /* LDS-staged radix scatter */
    PANDRS_HIP_PHASE_AGGREGATE = 5,   /* per-partition LDS hash aggregate + compaction */
    PANDRS_HIP_PHASE_BUILD = 6,       /* join: build side */
    PANDRS_HIP_PHASE_PROBE = 7,       /* join: probe count + write */
    PANDRS_HIP_PHASE_GATHER = 8,
    PANDRS_HIP_PHASE_OTHER = 9,
    PANDRS_HIP_PHASE_PREPARTITION = 10  /* first pass (64 buckets) of a two-pass radix partition: fan-outs >= 6144 */
};

typedef struct pandrs_hip_ctx pandrs_hip_ctx;

/* ---- library lifetime ----------------------------------------------------------------
 * Replaces init_gpu / get_gpu_manager (src/gpu/mod.rs:214-282).  Idempotent. */
int32_t pandrs_hip_abi_version(void);
int32_t pandrs_hip_init(const pandrs_hip_config *cfg /* NULL = defaults */);
int32_t pandrs_hip_shutdown(void);
int32_t pandrs_hip_device_count(int32_t *out_count);
const char *pandrs_hip_last_error(void);

/* ---- contexts (one stream + workspace per context) ------------------------------------ */
int32_t pandrs_hip_ctx_create(int32_t device_id, pandrs_hip_ctx **out_ctx);
int32_t pandrs_hip_ctx_destroy(pandrs_hip_ctx *ctx);
int32_t pandrs_hip_ctx_synchronize(pandrs_hip_ctx *ctx);
/* Pre-size the workspace arena (bytes) so the first timed call does not pay hipMalloc. */
int32_t pandrs_hip_ctx_reserve(pandrs_hip_ctx *ctx, int64_t workspace_bytes);
/* Number of device allocations (hipMalloc) the library has made in this process: workspace arenas grow but never
 * shrink, so a repeated call of the same shape adds none — tests assert that on the steady state. */
int32_t pandrs_hip_alloc_events(int64_t *out_device_allocations);
/* Tuning / testing knobs — EVERY name the library accepts (all default to 0 = automatic unless noted; the tests and
 * experiments/ use them to force individual code paths; tests/test_abi.py checks this list against capi.hip):
 *  planning
 *   "groups_hint"       expected number of groups (skips the sampled estimate)
 *   "partitions"        force the radix fan-out P
 *   "p_max"             lower the fan-out cap (forces the two-level path above it)
 *   "p_target"          rounds heuristic: take several aggregate rounds only above this fan-out (default: 8192 where the lean kernel answers, else 3072)
 *   "src_per_round"     force the number of value columns folded per aggregate round
 *   "load_pct"          LDS table load factor in percent (default 70)
 *  partition (scatter) pass
 *   "scatter_staged"    1 (default) = stage columns through LDS and write contiguous per-partition runs; 0 = direct stores
 *   "scatter_threads"   1024 (default) or 512 threads per scatter workgroup
 *   "shared_cursors"    1 (default) = one write cursor per (partition, XCD group); 0 = private cursors per workgroup
 *   "exact_partition"   1 = always the exact histogram + scan layout, never the sampled-capacity layout
 *  aggregate pass
 *   "agg_v1"            1 = never the lean persistent aggregate (aggregate2.hip); the round-1 kernel answers
 *   "generic_aggregate" 1 = the descriptor-driven generic instantiation of the round-1 kernel
 *   "agg_depth"         register-ring depth of the lean aggregate (2..4; default 3)
 *   "agg_ablate"        experiments only: switch parts of the lean aggregate off (see experiments/agg2_ablate.py)
 *   "no_runs"           1 = never the clustered-rows (RUNS) instantiation
 *   "no_direct"         1 = never the few-groups direct path (-1 = allow it below 4 M rows too)
 *   "test_throw"        tests of the exception firewall (ctx may be NULL): 1 = the entry point's host code throws std::bad_alloc
 *                       (-> PANDRS_HIP_ERR_OUT_OF_MEMORY), 2 = std::out_of_range, 3 = a non-std exception, 4 = an oversized
 *                       std::vector::resize (2 - 4 -> PANDRS_HIP_ERR_COMPUTATION, or OUT_OF_MEMORY for bad_alloc); never a crash
 *   "no_census"         1 = the group estimate never takes its second stage (a hash-slice census of a tenth of the rows, run when the
 *                       strided sample shows singletons its repeating keys cannot explain: a long tail behind a broad hot class)
 *   "tail_groups_hint"  tests: the group estimate handed to the tail run of the absorb pass's compact spill (0 = its own sample)
 *   "no_overflow_run"   1 = a full LDS table fails the attempt (the call is retried with 4 x the fan-out) instead of handing the rows it
 *                       could not place to a run of their own, whose groups are appended
 *   "sorted_dictionary" 1 = a column of a composite key that is too wide for its share of the 64-bit cell gets its dictionary codes by
 *                       ordering the rows (any cardinality) instead of listing its distinct cells and a hashed look-up per row
 *   "no_chao"           1 = the sampled group estimate is the uniform-occupancy model alone (no Chao1 term: tests, A/B)
 *   "no_absorb"         1 = never the hot-key absorb-and-spill pass in front of the radix path (-1 = whenever it is possible: tests)
 *   "no_hot_image"      1 = the absorb tables start empty (first come, first served) instead of from the sample's most frequent keys
 *   "no_slice"          1 = never cut oversized partitions into row slices; "slice_rows" forces the slice length
 *   "wide_slices"       1 = the pieces of an oversized partition are as long as the cutting threshold (4 x the average partition) instead of
 *                       average-sized (A/B: a piece is one workgroup's job, long pieces are the aggregate pass's tail)
 *   "slice_over"        experiments: a partition is cut when it holds more than this many average partitions' rows (default 2)
 *   "no_clustered"      1 = never the one-pass path for rows clustered by key (sorted input, input grouped by key); "clustered_chunk"
 *                       rows per chunk there, "clustered_max_runs_pct" runs per 100 rows up to which it is taken (default 13)
 *   "no_profile_rounds" 1 = aggregated columns of mixed kinds / op sets go to the older kernel (default: ordered by profile, the lean kernel
 *                       takes up to 4 columns of one profile per round)
 *   "no_burst_kernel"   1 = partitions whose keys arrive in bursts (short runs, keys local in position) are aggregated row per lane by the
 *                       lean kernel instead of 8 consecutive rows per thread
 *   "no_window_bound"   1 = the group estimate never counts distinct keys in windows of consecutive rows (its bound for keys that are
 *                       local in position: nearly sorted input)
 *   "no_lean_rounds"    1 = several aggregate rounds always run the older kernel (default: uniform profiles run the lean kernel once per
 *                       round of <= 4 columns over the same partitions)
 *   "no_table_order"    experiments: 1 = the lean aggregate draws its tables in partition order instead of largest first
 *   "fold_min", "fold_min_multi"  experiments: lanes of a wave in one table slot from which the lean aggregate folds them on the VALU
 *                       (ordinary tables: default 40; pieces of an oversized partition: default 8; 65 = never)
 *   "no_small"          1 = never the two-launch path for calls of <= 2 M rows; "small_chunk" rows per workgroup there
 *   "deterministic"     1 = f64 Sum / Mean / Std / Var re-folded in ascending row order (bit-identical to the
 *                       reference's sequential fold; about 3 x the default time)
 *  join
 *   "join_generic"      1 = always build the join with the general segmented sort
 *   "join_one_pass"     1 = single-pass probe (decoupled look-back) instead of count + emit
 *   "join_no_l2"        1 = fused join->groupby never takes the L2-region path for large build sides
 *   "scatter_wide"      the scatter's wide tile (16 K rows per workgroup, staged in two halves): 1 = whenever its LDS fits, -1 = never;
 *                       default: at fan-outs >= 1024 with two or more 8-byte value columns (>= 1536 with one)
 *   "two_pass"          -1 = the exact radix partition never takes two passes (64-127 buckets, then the rest) at fan-outs >= 6144;
 *                       "two_pass_min_p" = another threshold (tests: also lifts the 4 M-row minimum)
 *   "join_pair_p"       the L2-region probe's minimum pair fan-out (0 = the default)
 *   "join_no_pairpart"  1 = the L2-region probe emits its pairs through one cursor instead of pre-partitioned
 *  Median / Nunique
 *   "median_generic"    1 = always the general segmented-sort pass, never the LDS group-sort path
 * Unknown names are rejected with PANDRS_HIP_ERR_INVALID_ARGUMENT.
 * Diagnostics (environment, read once): PANDRS_HIP_ENGINE_TRACE=1 prints one line per engine attempt (rows, estimate, fan-out, table
 * slots, kernel family, nesting level) on stderr; PANDRS_HIP_DIST_TRACE=1 the wall time of every stage of a distributed groupby. */
int32_t pandrs_hip_ctx_set_option(pandrs_hip_ctx *ctx, const char *name, int64_t value);
int32_t pandrs_hip_get_timings(pandrs_hip_ctx *ctx, pandrs_hip_timings *out);

/* ---- resident columns --------------------------------------------------------------------------
 * The reference's columns are immutable Arc<[T]> buffers (src/column/int64_column.rs:10, float64_column.rs:9,
 * string_column.rs:26) that every operator of the public frame Arc-clones into the split frame
 * (src/optimized/dataframe/transformations.rs:524-577 aggregate, :628-694 join): nothing ever writes to them.  A shim
 * therefore uploads a column ONCE and serves every later aggregate / join on it from HBM:
 *   pandrs_hip_column_upload copies the host column (data + null bitmap, if any) into one device allocation owned by
 *     the context and fills *out_device_col with the device pointers (same dtype); returns after the copy has
 *     completed, so the host buffers may be dropped.  Use the descriptor with PANDRS_HIP_MEM_DEVICE in any entry
 *     point, as often as wanted.
 *   pandrs_hip_column_release frees it (waits for work in flight on the context's stream first); columns still
 *     resident when the context is destroyed are freed with it.  Releasing a descriptor twice, or one that did not
 *     come from column_upload on this context, is PANDRS_HIP_ERR_INVALID_ARGUMENT.
 *   pandrs_hip_resident_bytes reports what the context currently holds (counted against
 *     pandrs_hip_config.memory_limit).
 * The Rust shim keys its handles by the Arc's data pointer and keeps a Weak beside each (integration/rust/
 * hip_shim.rs `ResidentCache`): a dropped column can neither be served stale nor have its address reused while cached. */
int32_t pandrs_hip_column_upload(pandrs_hip_ctx *ctx, const pandrs_hip_column *host_col, int64_t n_rows,
                                 pandrs_hip_column *out_device_col);
int32_t pandrs_hip_column_release(pandrs_hip_ctx *ctx, const pandrs_hip_column *device_col);
int32_t pandrs_hip_resident_bytes(pandrs_hip_ctx *ctx, int64_t *out_bytes, int64_t *out_columns);

/* ---- groupby-aggregate ------------------------------------------------------------------
 * Replaces OptimizedDataFrame::group_by(..)?.aggregate(..)
 *   (src/optimized/split_dataframe/group/grouping.rs:22-115 +
 *    src/optimized/split_dataframe/group/aggregation.rs:500-871)
 * and the inline copy in LazyFrame::execute (src/optimized/lazy.rs:186-404).
 *
 * Semantics (bit-for-bit the reference's fold rules):
 *   - a null key is its own group ("NULL", grouping.rs:74); f64 keys: all NaNs one group,
 *     0.0 != -0.0 (string equality of val.to_string()).
 *   - every aggregate is an f64; I64 Sum wraps in i64 then casts; Mean of nothing = 0.0;
 *     Min/Max whose sentinel is unchanged = 0.0; Count = group size INCLUDING nulls;
 *     Std/Var = two-pass, Bessel; First/Last = value at first/last row, null => 0.0;
 *     Median = middle of the sorted non-null values, even counts average the two middles
 *     (Int64: added in i64 first), no non-null value => 0.0 (aggregation.rs:585-604, :703-722).
 *     Nunique (not an AggregateOp: the legacy frame's AggFunc::Nunique) = number of distinct non-null
 *     values under `==` (-0.0 joins 0.0, every NaN counts), none => 0.0 (src/dataframe/groupby.rs:514-519).
 *   - group order in the output is unspecified (reference: HashMap order).
 *   - value dtypes: I64 / F64 for numeric ops; Count accepts any dtype; anything else =>
 *     PANDRS_HIP_ERR_OPERATION_FAILED (aggregation.rs:748).
 *
 * Two steps so that outputs are caller-allocated after a size query:
 *   1. pandrs_hip_groupby_agg computes on the device and keeps the result in the ctx,
 *      returning n_groups;
 *   2. pandrs_hip_groupby_fetch copies it out and may be called repeatedly until the next
 *      compute call on the same ctx.
 * out_keys[k]     : n_groups 8-byte cells — i64 value / f64 bits / zero-extended u32 code /
 *                   0-1 for bool.  Stringification stays host-side (aggregation.rs:856-860).
 * out_key_null[k] : n_groups bytes, 1 = the NULL group (may be NULL pointer to skip).
 * out_aggs[a]     : n_groups doubles per requested aggregate, in request order.
 */
int32_t pandrs_hip_groupby_agg(pandrs_hip_ctx *ctx, int32_t mem_space,
                               const pandrs_hip_column *keys, int32_t n_keys,
                               int64_t n_rows,
                               const pandrs_hip_column *vals, int32_t n_vals,
                               const pandrs_hip_agg_spec *aggs, int32_t n_aggs,
                               int64_t *out_n_groups);
int32_t pandrs_hip_groupby_fetch(pandrs_hip_ctx *ctx, int32_t mem_space,
                                 uint64_t *const *out_keys, uint8_t *const *out_key_null,
                                 double *const *out_aggs);

/* ---- mergeable partial aggregates (multi-GPU) -----------------------------
 * The reference has no distributed path for groupby (src/distributed is an in-process DataFusion
 * wrapper); these three calls are the pieces of the row-range-sharded plan:
 *   pandrs_hip_groupby_partials : same inputs as groupby_agg, but keeps the un-finalised
 *     per-group states (group size, then per value column: sum, non-null count, min, max as the
 *     requested ops need) so that partials from several row-range shards can be merged.  The
 *     state layout is a pure function of (value dtypes, has-null flags, agg specs): every rank
 *     computes the same layout.  *out_n_state = 8-byte state cells per group (incl. group size).
 *   pandrs_hip_partials_split : buckets the retained partial rows by owner rank
 *     (hash(key) mod n_ranks; the NULL group goes to rank 0) and writes them rank-contiguous as
 *     packed records of W = 2 + n_state 8-byte cells: [key cell, key_null (0/1), states...],
 *     ready for ONE all-to-all; out_counts[r] = records destined to rank r (host array).
 *   pandrs_hip_groupby_merge : consumes concatenated packed records (from all peers), merges
 *     equal keys and finalises the aggregates exactly like groupby_agg; fetch with
 *     pandrs_hip_groupby_fetch.  val_dtypes / val_has_nulls / aggs must equal the producers'. */
int32_t pandrs_hip_groupby_partials(pandrs_hip_ctx *ctx, int32_t mem_space,
                                    const pandrs_hip_column *keys, int32_t n_keys,
                                    int64_t n_rows,
                                    const pandrs_hip_column *vals, int32_t n_vals,
                                    const pandrs_hip_agg_spec *aggs, int32_t n_aggs,
                                    int64_t *out_n_groups, int32_t *out_n_state);
int32_t pandrs_hip_partials_split(pandrs_hip_ctx *ctx, int32_t mem_space, int32_t n_ranks,
                                  uint64_t *out_records /* [n_groups][2 + n_state] */,
                                  int64_t *out_counts /* host, n_ranks */);
int32_t pandrs_hip_groupby_merge(pandrs_hip_ctx *ctx, int32_t mem_space, int32_t key_dtype,
                                 const uint64_t *records /* [n_rows][2 + n_state] */,
                                 int64_t n_rows,
                                 const int32_t *val_dtypes, int32_t n_vals,
                                 const uint8_t *val_has_nulls,
                                 const pandrs_hip_agg_spec *aggs, int32_t n_aggs,
                                 int64_t *out_n_groups);

/* ---- group_by's row -> group assignment ----------------------------------
 * Replaces the body of OptimizedDataFrame::group_by (src/optimized/split_dataframe/group/
 * grouping.rs:22-115), which builds HashMap<Vec<String>, Vec<usize>>: per group, the ascending list
 * of its row indices (:98-103).  GroupBy.groups is a pub field (group/types.rs:52) that filter /
 * transform / aggregate_custom (group/operations.rs:51-435, aggregation.rs:391-497) and
 * par_groupby's sub-frame gathers (grouping.rs:286-328) read.  Device form = CSR:
 * group g has key cells out_keys[k][g] (+ null flags) and rows out_rows[out_offsets[g] ..
 * out_offsets[g+1]), ascending.  Null keys form one group per distinct combination, like the
 * reference's "NULL" strings (grouping.rs:74).  Group order is unspecified (HashMap order there). */
int32_t pandrs_hip_groupby_indices(pandrs_hip_ctx *ctx, int32_t mem_space,
                                   const pandrs_hip_column *keys, int32_t n_keys, int64_t n_rows,
                                   int64_t *out_n_groups);
/* out_keys[k] / out_key_null[k]: n_groups entries each; out_offsets: n_groups + 1; out_rows: n_rows.
 * Any pointer may be NULL to skip that output. */
int32_t pandrs_hip_groupby_indices_fetch(pandrs_hip_ctx *ctx, int32_t mem_space,
                                         uint64_t *const *out_keys, uint8_t *const *out_key_null,
                                         int64_t *out_offsets, int64_t *out_rows);

/* ---- multi-GPU: row shuffle by key owner ("radix all-to-all ... on key") -------------
 * The general exchange for what pre-aggregated partials cannot express (Std/Var/Median, or both
 * sides of a join): every row goes to the rank that owns its key, owner = f(key cell) mod n_ranks,
 * the same function on every rank and for every column that shares the key.  This call buckets ONE
 * shard's rows by owner and keeps them rank-contiguous in the context: key cells, one null byte per
 * row, and every payload column (I64 / F64 as is, U32CODE zero-extended to 8 bytes) with one null
 * byte per row for masked payloads.  out_counts[r] = rows for rank r.  Rows with a NULL key go to
 * the last rank (they form one group, grouping.rs:74) or are dropped when drop_null_keys != 0
 * (they never match in a join, join.rs:112, :152).  After the all-to-all the receiver turns the null
 * bytes into bitmaps (pandrs_hip_bytes_to_bitmap) and calls the ordinary entry points with key
 * dtype PANDRS_HIP_CELL64. */
int32_t pandrs_hip_shuffle_split(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *key,
                                 const pandrs_hip_column *payload, int32_t n_payload, int64_t n_rows,
                                 int32_t n_ranks, int32_t drop_null_keys, int64_t *out_counts,
                                 int64_t *out_n_rows);
/* One 64-bit hash cell per row of a COMPOSITE key (n_keys columns, nulls included): equal key tuples
 * get equal cells on every rank, so the cells can serve as the shuffle key (dtype CELL64) of a
 * multi-key groupby while the key columns themselves travel as payload.  Collisions only co-locate
 * different tuples on one rank; the receiving groupby still separates them. */
int32_t pandrs_hip_key_hash_cells(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *keys,
                                  int32_t n_keys, int64_t n_rows, uint64_t *out_cells);
/* out_cells / out_key_null: out_n_rows entries; out_payload[c] (8 bytes per row) / out_payload_null[c]
 * (1 byte per row; ignored for payloads without a mask).  Any pointer may be NULL to skip it. */
int32_t pandrs_hip_shuffle_fetch(pandrs_hip_ctx *ctx, int32_t mem_space, uint64_t *out_cells,
                                 uint8_t *out_key_null, uint64_t *const *out_payload,
                                 uint8_t *const *out_payload_null);
/* one byte per row (non-zero = set) -> LSB-first bitmap of (n + 7) / 8 bytes (src/core/column.rs:163-177) */
int32_t pandrs_hip_bytes_to_bitmap(pandrs_hip_ctx *ctx, int32_t mem_space, const uint8_t *bytes, int64_t n,
                                   uint8_t *out_bitmap);

/* ---- hash join ------------------------------------------------------------------------------
 * Replaces OptimizedDataFrame::join_impl (src/optimized/split_dataframe/join.rs:76-555) up to
 * the join_indices vector (:146-224); the column gathers (:286-552) are pandrs_hip_gather_*.
 * Order contract = the reference's: left rows ascending; for one left row its matches in
 * ascending right-row order; left/outer misses in place; right/outer unmatched right rows
 * appended ascending.  Null keys never match and null LEFT keys are dropped even for
 * left/outer (join.rs:152).  -1 marks the missing side.  dtype mismatch between the two key
 * columns => PANDRS_HIP_ERR_TYPE_MISMATCH (join.rs:98-104). */
int32_t pandrs_hip_join_indices(pandrs_hip_ctx *ctx, int32_t mem_space,
                                const pandrs_hip_column *left_key, int64_t n_left,
                                const pandrs_hip_column *right_key, int64_t n_right,
                                int32_t how, int64_t *out_n_rows);
int32_t pandrs_hip_join_fetch(pandrs_hip_ctx *ctx, int32_t mem_space,
                              int64_t *out_left_idx, int64_t *out_right_idx);

/* out[i] = idx[i] >= 0 && !null(src, idx[i]) ? src[idx[i]] : fill
 * (join.rs:296-357: misses and nulls become 0 / 0.0 / "" / false, not nulls). */
int32_t pandrs_hip_gather_i64(pandrs_hip_ctx *ctx, int32_t mem_space, const int64_t *src,
                              const uint8_t *src_null_mask, const int64_t *idx, int64_t n,
                              int64_t fill, int64_t *out);
int32_t pandrs_hip_gather_f64(pandrs_hip_ctx *ctx, int32_t mem_space, const double *src,
                              const uint8_t *src_null_mask, const int64_t *idx, int64_t n,
                              double fill, double *out);
int32_t pandrs_hip_gather_u32(pandrs_hip_ctx *ctx, int32_t mem_space, const uint32_t *src,
                              const uint8_t *src_null_mask, const int64_t *idx, int64_t n,
                              uint32_t fill, uint32_t *out);
/* The same gather with the source length in the signature, for either memory space (host columns are
 * staged): src->dtype selects the element — out holds 8 bytes per row for I64 / F64 (fill_bits = the
 * fill value's bit pattern), 4 for U32CODE, and one 0/1 byte per row for BOOLBITS. */
int32_t pandrs_hip_gather_column(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *src,
                                 int64_t n_src, const int64_t *idx, int64_t n, uint64_t fill_bits, void *out);
/* One output column of the join whose pairs this context still holds (the last pandrs_hip_join_indices):
 * out[i] = the gather above with idx = the pairs' left rows (side 0) or right rows (side 1).  This is
 * join_impl's column assembly (join.rs:286-552) with the index pairs never leaving HBM: a shim fetches the
 * joined COLUMNS, not 16 bytes of indices per output row.  `src` lives in src_mem_space (a resident column
 * descriptor from pandrs_hip_column_upload, or a host column, which is staged); `out` (element sizes as for
 * pandrs_hip_gather_column, pandrs_hip_join_indices' *out_n_rows elements) lives in out_mem_space. */
int32_t pandrs_hip_join_gather(pandrs_hip_ctx *ctx, int32_t src_mem_space, const pandrs_hip_column *src,
                               int64_t n_src, int32_t side, uint64_t fill_bits, int32_t out_mem_space, void *out);
/* The join-key column of that frame (join.rs:364-470): the LEFT key's value where the pair has a left row (a null
 * there becomes the fill value), else the RIGHT key's value at the pair's right row (right / outer joins). */
int32_t pandrs_hip_join_gather_key(pandrs_hip_ctx *ctx, int32_t src_mem_space, const pandrs_hip_column *left_key,
                                   int64_t n_left, const pandrs_hip_column *right_key, int64_t n_right,
                                   uint64_t fill_bits, int32_t out_mem_space, void *out);
/* bit-packed source (BooleanColumn), byte-per-row output */
int32_t pandrs_hip_gather_bool(pandrs_hip_ctx *ctx, int32_t mem_space, const uint8_t *src_bits,
                               const uint8_t *src_null_mask, const int64_t *idx, int64_t n,
                               uint8_t fill, uint8_t *out);

/* Fused inner join -> groupby(right payload g).sum(left payload v)  (BASELINE config 5):
 * never materialises the join rows.  Equivalent to inner_join (join.rs:32) followed by
 * group_by(g).aggregate([(v, Sum)]) (aggregation.rs:763).  Fetch with groupby_fetch. */
int32_t pandrs_hip_join_groupby_sum(pandrs_hip_ctx *ctx, int32_t mem_space,
                                    const pandrs_hip_column *left_key,
                                    const pandrs_hip_column *left_val, int64_t n_left,
                                    const pandrs_hip_column *right_key,
                                    const pandrs_hip_column *right_group, int64_t n_right,
                                    int64_t *out_n_groups);

/* ---- whole-column reductions ----------------------------------------------
 * Replaces simd_{sum,mean,min,max}_{f64,i64} (src/optimized/jit/simd.rs:9-112) and
 * Int64Column/Float64Column::{sum,mean,min,max}.  out[0..3] = sum, mean, min, max as f64;
 * out_count = number of non-null elements.  Empty input: sum 0, mean 0 (simd.rs), min/max
 * +inf/-inf for f64 and i64::MAX/MIN (as f64) for i64. */
int32_t pandrs_hip_reduce_column(pandrs_hip_ctx *ctx, int32_t mem_space,
                                 const pandrs_hip_column *col, int64_t n,
                                 double out[4], int64_t *out_count);

/* (sum, sum of squares, count) of the non-null values as f64 — the accumulator triple of
 * parallel_std_f64 / parallel_var_f64 (src/optimized/jit/parallel.rs:190-250), from which the caller
 * derives the reference's POPULATION variance  max(sum_sq / n - mean^2, 0)  (:222-233; n <= 1 => 0). */
int32_t pandrs_hip_reduce_moments(pandrs_hip_ctx *ctx, int32_t mem_space,
                                  const pandrs_hip_column *col, int64_t n,
                                  double *out_sum, double *out_sum_sq, int64_t *out_count);

/* ---- multi-GPU: the exchange inside the library (no reference counterpart: ---------------------
 * src/gpu/multi_gpu.rs:326-360 splits rows on the host, src/distributed is an in-process DataFusion wrapper).
 * One process per GPU; every rank calls the same entry point with its own row-range shard.  RCCL is opened at run
 * time (librccl.so.1), so single-GPU users do not need it.
 *
 * pandrs_hip_comm wraps an ncclComm_t: rank 0 calls comm_unique_id, the host distributes the 128 bytes (any
 * channel), every rank calls comm_init; or comm_adopt takes an ncclComm_t the host already owns. */
typedef struct pandrs_hip_comm pandrs_hip_comm;
int32_t pandrs_hip_comm_unique_id(char out_id[128]);
int32_t pandrs_hip_comm_init(pandrs_hip_ctx *ctx, const char id[128], int32_t rank, int32_t world, pandrs_hip_comm **out_comm);
int32_t pandrs_hip_comm_adopt(void *nccl_comm, int32_t rank, int32_t world, pandrs_hip_comm **out_comm);
/* Any other fabric (and the multi-rank tests on one GPU): the collectives the exchange needs, as host callbacks over
 * HOST buffers — the library stages its device buffers through them.  Every callback returns 0 on success and is
 * called by every rank in the same order:
 *   all_gather          every rank contributes `bytes` bytes; recv holds world * bytes, rank-major
 *   all_reduce_max_i64  element-wise maximum over the ranks of n int64 values, in place
 *   all_to_all_v        send_bytes[p] bytes at send + send_off[p] go to rank p; recv_bytes[p] bytes from rank p land at
 *                       recv + recv_off[p] (the counts were agreed by an all_gather before) */
typedef struct pandrs_hip_transport {
    void *user;
    int32_t (*all_gather)(void *user, const void *send, void *recv, int64_t bytes);
    int32_t (*all_reduce_max_i64)(void *user, int64_t *vals, int32_t n);
    int32_t (*all_to_all_v)(void *user, const void *send, const int64_t *send_bytes, const int64_t *send_off,
                            void *recv, const int64_t *recv_bytes, const int64_t *recv_off);
} pandrs_hip_transport;
int32_t pandrs_hip_comm_adopt_transport(const pandrs_hip_transport *transport, int32_t rank, int32_t world, pandrs_hip_comm **out_comm);
int32_t pandrs_hip_comm_destroy(pandrs_hip_comm *comm);

/* Row-range-sharded group_by(..).aggregate(..) (aggregation.rs:763) over all ranks' rows: local partial states ->
 * owner split (one block of records per owner, a small column store; no host round trip) -> count exchange straight from the
 * device counts -> ONE grouped ncclSend / ncclRecv all-to-all, a block per peer, on the context's stream -> merge.  The result (fetch with groupby_fetch) holds the groups this rank owns; the ranks' key sets are
 * disjoint.  That is the path for Sum / Mean / Min / Max / Count over one key column.  Anything else except First / Last
 * (Std / Var / Median / Nunique, composite keys of up to 8 columns) takes the row shuffle inside the same call: every row
 * goes to the owner of its key (the radix partitioner with P = world; a composite key on a hash cell of the tuple), one count
 * exchange, one grouped all-to-all of all columns, and the owner runs the ordinary groupby on what it received.
 * Null-mask presence may differ between ranks: the layout (which value columns carry a non-null count) is agreed ON the
 * count exchange — every rank plans with the layout of the previous call of this shape plus its own masks, and the ranks
 * repeat their local phase once when they find they disagreed — so a steady-state call makes no collective of its own
 * for it (at most 62 value columns on this path).  A rank whose local phase fails still joins the count exchange with its
 * status: EVERY rank then returns an error (nobody is left blocked in a collective).  A failure that is rank-local AFTER
 * the count exchange (the receive buffer cannot be allocated) aborts the communicator (ncclCommAbort) so that the peers
 * return instead of blocking; the communicator then refuses further calls (PANDRS_HIP_ERR_NOT_INITIALIZED).  The exchange
 * buffers live in the communicator and are only ever grown. */
int32_t pandrs_hip_dist_groupby_agg(pandrs_hip_ctx *ctx, pandrs_hip_comm *comm, int32_t mem_space,
                                    const pandrs_hip_column *keys, int32_t n_keys, int64_t n_rows,
                                    const pandrs_hip_column *vals, int32_t n_vals,
                                    const pandrs_hip_agg_spec *aggs, int32_t n_aggs, int64_t *out_n_groups);

/* BASELINE config 5 across ranks: inner_join (join.rs:32) of row-range-sharded sides + group_by(g).sum(v).  The
 * build (right) side is all-gathered, the probe side stays on its GPU, the local fused result goes through the
 * groupby exchange.  Device-resident shards, 8-byte build-side columns. */
int32_t pandrs_hip_dist_join_groupby_sum(pandrs_hip_ctx *ctx, pandrs_hip_comm *comm, int32_t mem_space,
                                         const pandrs_hip_column *left_key, const pandrs_hip_column *left_val, int64_t n_left,
                                         const pandrs_hip_column *right_key, const pandrs_hip_column *right_group, int64_t n_right,
                                         int64_t *out_n_groups);

/* Everything K1's three families of reference functions need, from ONE pass over the column:
 *  (A) OptimizedDataFrame::{sum,mean,min,max}  (src/optimized/split_dataframe/aggregate.rs:21-215): values as f64
 *      ((v as f64) for Int64), sum = sum_f64 (0.0 when count == 0), mean = sum_f64 / count and min / max = `min` /
 *      `max` (fold with f64::min / f64::max from +-inf: NaN operands are ignored, infinities are not), each
 *      Err(Error::Empty) when count == 0;
 *  (B) Float64Column::{sum,mean,min,max} (src/column/float64_column.rs:100-199) and Int64Column's
 *      (src/column/int64_column.rs:100-199): min / max skip NON-FINITE values -> `min_finite` / `max_finite`, None when
 *      count_finite == 0; Int64Column::sum is the wrapping i64 sum `sum_i64`, its mean sum_i64 as f64 / count;
 *  (C) simd_{sum,mean,min,max}_{f64,i64} (src/optimized/jit/simd.rs:9-112): simd_mean_i64 is the INTEGER division
 *      sum_i64 / count (:77-82), empties give 0 / 0.0 and the fold identities (+-inf, i64::MAX / MIN).
 * f64 sums are accumulated pairwise on the device (the reference: Kahan per chunk + Kahan combine,
 * src/optimized/jit/parallel.rs:71-102); they agree to 1e-9 relative, not bit for bit.
 * +0.0 / -0.0 ties: -0.0 < +0.0 here; f64::min leaves the tie unspecified. */
typedef struct pandrs_hip_column_stats {
    int64_t count;          /* non-null values */
    int64_t count_finite;   /* f64: finite ones among them; i64: = count */
    double sum_f64;         /* sum of the values as f64 */
    double sum_sq;          /* sum of their squares as f64 */
    int64_t sum_i64;        /* i64 columns: wrapping integer sum; 0 for f64 columns */
    int64_t min_i64, max_i64; /* i64 columns: exact extremes (i64::MAX / MIN when count == 0); 0 for f64 columns */
    double min, max;        /* NaN-ignoring extremes over the non-null values; +inf / -inf when there is none; i64: as f64 */
    double min_finite, max_finite; /* the same over finite values only */
} pandrs_hip_column_stats;

int32_t pandrs_hip_reduce_stats(pandrs_hip_ctx *ctx, int32_t mem_space,
                                const pandrs_hip_column *col, int64_t n, pandrs_hip_column_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* PANDRS_HIP_H */
