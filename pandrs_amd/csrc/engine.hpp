// engine.hpp — declarations shared by groupby.hip and join.hip (radix partitioner + aggregate engine).
#pragma once
#include "common.hpp"
#include "device_utils.hpp"

namespace pandrs {

constexpr int MAX_SRC = 16;        // aggregated columns of a call (and sources per round of the older kernel's catch-all instantiation)
constexpr int MAX_MERGE_SRC = 39;  // states a merge of partial records takes (every state is a source of its own; + the group size <= MAX_MOVE)
constexpr int MAX_STATES = 40;
constexpr int MAX_AGGS = 64;
constexpr int MAX_MOVE = 40;
constexpr int P_MAX = 8192;      // bounded by the scatter's LDS counters (2 x 4 B x (P+1) beside the 80 KB tile buffers)

constexpr int HI_THREADS = 1024;   // histogram workgroup
constexpr int SC_RPT = 8;          // scatter: rows per thread per tile
// (12 rows per thread = 12 K-row tiles, 1.5x longer runs, measured: 84 spilled VGPRs, C2's scatter 1.94 -> 2.54 ms)
constexpr int SC_RPT_WIDE = 16;    // the wide tile (16 K rows per 1024-thread workgroup): twice the run length, staged in two halves
constexpr int SC_POS_BITS = 14;    // the scatter packs (partition << SC_POS_BITS | position in the tile)
constexpr int SC_TILE_MAX = 1024 * SC_RPT_WIDE;
static_assert(SC_TILE_MAX <= (1 << SC_POS_BITS), "scatter packs the tile position in SC_POS_BITS bits");
constexpr int AG_THREADS = 1024;   // aggregate workgroup
constexpr int MAX_ROUNDS = 16;

enum StateKind : int8_t { SK_ADD_F64 = 0, SK_ADD_I64, SK_MIN_F64, SK_MAX_F64, SK_MIN_I64, SK_MAX_I64 };

struct SrcDev {
    const uint64_t *vals;   // partitioned 8-byte values
    const uint8_t *valid;   // partitioned validity bytes (1 = valid) or nullptr
    int8_t kind;            // 0 = f64, 1 = i64
    int8_t st_add, st_min, st_max, st_nn;  // LDS state indices, -1 = none
    int8_t st_fadd;         // f64 sum of (double)x for an i64 column's Std/Var (aggregation.rs:559-563)
    int8_t st_ssq;          // sum of squared deviations from the group mean, filled by the second pass
    int8_t pad[1];
};

struct FinDev {
    int8_t op, kind, st_add, st_nn, st_min, st_max;   // st_*: LDS state index inside its round
    int8_t round, st_ssq;
    int8_t st_fadd, rowsrc_min, rowsrc_max, pad[5];   // rowsrc_*: LDS state of the first / last row index
    const void *col_data;       // First/Last: the ORIGINAL (un-partitioned) column and its null bitmap
    const uint8_t *col_null;
};

struct MoveDesc {
    const void *src;
    void *dst;
    int kind;   // 0: 8-byte element, 1: null bitmap -> validity byte, 2: byte copy, 3: row index -> u32,
                // 4: u32 source -> u64, 5: row index -> u64
    int pad;
};


struct ScatterArgs {
    KeyDesc key;
    uint64_t *pkeys;
    const uint32_t *offsets;   // partition-major exclusive scan of the histogram
    uint32_t *gcur;            // [P+1][8] shared write cursors per (partition, group); nullptr = private cursors
    // capacity mode (radix_partition_sampled): no histogram ran; region (p, g) is [gbeg, gend) rows, sized from a
    // sample.  A run that does not fit raises flags[0] and is dropped (the caller falls back to the exact path).
    const uint32_t *gend;      // nullptr = exact mode
    uint32_t *flags;
    uint32_t total_cap;        // rows of the partitioned arrays in capacity mode
    int64_t n_rows, chunk;
    uint32_t P, seed;
    int n_move, n_move8;       // mv[0 .. n_move8) are 8-byte columns, the rest byte-wide
    uint32_t hash_P, part_shift;   // 0, 0 normally.  First pass of a two-pass partition: partition id = part_of(hash, hash_P) >> part_shift
                                   // (a monotone coarsening of the final partition ids; P = the number of such buckets)
    int allow_two_pass;        // radix_partition may take two passes at fan-outs >= 6144: the caller has budgeted two_pass_workspace_bytes in c->work
    MoveDesc mv[MAX_MOVE];
};


struct Plan {
    int n_src = 0, n_states = 0, n_fin = 0;
    int src_col[MAX_SRC];          // index into vals[]
    int8_t src_kind[MAX_SRC];
    int8_t st_add[MAX_SRC], st_min[MAX_SRC], st_max[MAX_SRC], st_nn[MAX_SRC];   // absolute state ids
    int8_t st_fadd[MAX_SRC], st_ssq[MAX_SRC];
    int8_t st_firstrow = -1, st_lastrow = -1;      // group-level: min / max original row index
    bool needs_second_pass = false;                // any Std / Var
    bool mergeable = true;                         // false when Std/Var/First/Last are requested
    bool has_median = false;                       // Median aggregates are filled by median_pass after the engine run
    int8_t kinds[MAX_STATES];
    int8_t fin_op[MAX_AGGS], fin_kind[MAX_AGGS];
    int fin_src[MAX_AGGS];         // plan source of each aggregate, -1 for COUNT
};


// Row source handed to the engine: device pointers only.
struct PrePartitioned;
struct RowSource {
    KeyDesc key;
    int64_t n_rows = 0;
    const PrePartitioned *pre = nullptr;        // rows that a producer already wrote in the capacity layout (see below)
    // raw mode: per plan source, the value column and its null bitmap
    const void *val_data[MAX_SRC]{};
    const uint8_t *val_null_bits[MAX_SRC]{};
    const uint8_t *val_valid_bytes[MAX_SRC]{};   // alternative to null_bits: one byte per row, 1 = valid
    // First / Last read the value at the group's first / last ORIGINAL row: the un-partitioned column
    // (differs from val_data inside a two-level sub-run) and the original row index per row (nullptr = i)
    const void *fin_data[MAX_SRC]{};
    const uint8_t *fin_null_bits[MAX_SRC]{};
    const uint64_t *row_index = nullptr;
    const int64_t *merge_gsize = nullptr;        // merge mode alternative to merge_states column 0 (with merge_cols)
    const uint64_t *merge_cols[MAX_STATES]{};    // merge mode alternative: one pointer per state column
    // merge mode: partial state columns [1 + n_states][n_rows] (column 0 = group size)
    const uint64_t *merge_states = nullptr;
    size_t merge_stride = 0;
};


// Result of one radix partition pass: partition p holds rows [offsets[p*NB], offsets[(p+1)*NB]) of
// the partitioned arrays; partition P is the null-key partition.
struct PartInfo {
    uint32_t P = 0, NB = 0;
    uint32_t *offsets = nullptr;
    // capacity mode: partition p = the 8 row ranges [gbeg[p*8+g], min(gcur[p*8+g], gend[p*8+g]))
    uint32_t *gbeg = nullptr, *gcur = nullptr, *gend = nullptr, *flags = nullptr;
    uint32_t total_cap = 0;
};

// Rows that arrive already radix-partitioned in the capacity layout (the fused join's probe writes its (g, v) pairs
// straight into their partitions): run_engine skips estimate and partition and aggregates `part`'s regions.
// The producer must have placed every row of a key in ONE partition; which function it used does not matter.
// key / val_data of the RowSource are ignored; n_rows = rows actually written (for sizing).
struct PrePartitioned {
    PartInfo part;                     // gbeg / gcur / gend / flags, P
    const uint64_t *pkeys = nullptr;   // [total_cap + trash tile] key cells
    const uint64_t *pvals[MAX_SRC]{};  // per plan source, 8-byte values, same layout
    int64_t est_groups = 0;            // an upper bound on the number of groups (sizes nothing but the retry decision)
};

// histogram -> scan -> scatter.  The caller fills sa.key, sa.pkeys, sa.mv[0..n_move), sa.n_rows,
// sa.P, sa.seed; workspace comes from c->work (not reset here).
int32_t radix_partition(pandrs_hip_ctx *c, ScatterArgs &sa, PartInfo *out, int phase_hist, int phase_scan,
                        int phase_scatter);
// The same partition WITHOUT the exact histogram pass: per-(partition, group) region capacities come from a
// 1-in-8 sample of the keys (+ 6 sigma), the scatter appends with the shared cursors and drops what does not
// fit (flags[0]: the caller re-runs with radix_partition).  The caller allocates every destination column with
// sampled_partition_rows(n_rows, P) rows.
int32_t radix_partition_sampled(pandrs_hip_ctx *c, ScatterArgs &sa, PartInfo *out, int phase_hist, int phase_scatter);
uint32_t sampled_partition_rows(int64_t n_rows, int64_t P);
constexpr uint32_t SAMPLE_REPL = 16;        // replicas of a sampled partition histogram (same-address global atomics serialise)
// (clump = how many rows travel together on average: 1 for independent rows; widens the 6-sigma margin of the split)
// regions of the capacity layout from a sampled histogram hist[SAMPLE_REPL][P1 + 1] ([..][P1] = rows sampled) that stands for n_rows rows
void plan_sampled_regions(pandrs_hip_ctx *c, const uint32_t *hist, int64_t n_rows, uint32_t P1, uint32_t total_cap, uint32_t *gbeg,
                          uint32_t *gcur, uint32_t *gend, uint32_t *flags, double clump);
// LDS table slots of the lean aggregate kernel for a plan with `round_states` 8-byte states per group (what run_engine will use)
int64_t lean_table_slots(const pandrs_hip_ctx *c, int round_states);
bool sampled_partition_ok(int64_t n_rows, int64_t P);
int32_t exclusive_scan_u32(pandrs_hip_ctx *c, const uint32_t *in, size_t n, uint32_t *out, uint32_t *seg);
size_t scan_seg_count(size_t n);     // entries the `seg` scratch of exclusive_scan_u32 needs
// sampled cardinality estimate (also sets c->clustered_rows); synchronises the stream
// keep_table: the sample's key table stays armed with this call's keys until estimate_coverage / estimate_release
int32_t estimate_groups(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, int64_t *out_est, bool keep_table = false);
// share of the rows (by the kept sample) held by the `budget` most frequent keys: the absorb-and-spill decision
int32_t estimate_coverage(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, int64_t budget, double *out_share,
                          int64_t image_T = 0, uint32_t image_seed = 0, double min_share = 0.0, const uint64_t **out_image = nullptr);
void estimate_release(pandrs_hip_ctx *c);
// partition starts (offsets[p * NB]) of the first n partitions, gathered into a dense device array
void gather_part_offsets(pandrs_hip_ctx *c, const uint32_t *offsets, uint32_t NB, uint32_t n, uint32_t *out);

int32_t build_plan(const int32_t *val_dtypes, const uint8_t *val_has_nulls, int n_vals,
                   const pandrs_hip_agg_spec *aggs, int n_aggs, Plan &pl);
int32_t run_engine(pandrs_hip_ctx *c, const RowSource &rs, const Plan &pl, bool merge, bool partials,
                   int n_aggs, int key_dtype, int n_keys_out = 1, int res_slot = 0);
// work-arena bytes a two-pass radix_partition of n_rows rows with n_cols8 8-byte and n_cols1 byte-wide moved columns takes on top of the single pass
size_t two_pass_workspace_bytes(int64_t n_rows, int n_cols8, int n_cols1);
size_t engine_workspace_bytes(int64_t n_rows, int n_cols8, int n_cols1);

// segsort.hip: sorts every partition [0, n_parts) of (keys, payload) ascending by (key, payload),
// in place, whatever the partition sizes.  `enc`: 0 = payload compared as is, 1 = f64 bits and
// 2 = i64 rewritten to their order-preserving u64 encodings (and left encoded).
int32_t segmented_sort_u32(pandrs_hip_ctx *c, uint64_t *keys, uint32_t *pay, const uint32_t *offsets, uint32_t NB,
                           uint32_t n_parts, int64_t n_rows, const uint8_t *only = nullptr);
// `only` (optional, device, one byte per partition): sort just the partitions whose byte is non-zero.
// `tiles` (optional) receives the device task list the sort ran on — one SortTask per SS_TILE-row tile of
// every sorted partition, `counters[0]` of them, at most `max_tasks` — for kernels that walk the sorted rows.
constexpr uint32_t SS_TILE = 8192;
struct SortTask {
    uint32_t pbeg, pend;    // the partition's row range
    uint32_t tile;          // tile index inside the partition
    uint32_t multi;         // the partition has more than one tile
};
struct SortTiles {
    const SortTask *tasks = nullptr;
    const uint32_t *counters = nullptr;
    uint32_t max_tasks = 0;
};
int32_t segmented_sort_u64(pandrs_hip_ctx *c, uint64_t *keys, uint64_t *pay, const uint32_t *offsets, uint32_t NB,
                           uint32_t n_parts, int64_t n_rows, int enc, const uint8_t *only = nullptr, SortTiles *tiles = nullptr);
size_t segsort_workspace_bytes(int64_t n_rows, uint32_t n_parts, size_t pay_bytes);

// median.hip: fills aggregate `fin_index` of c->gb with the groups' medians of one value column
// (kind 0 = f64, 1 = i64); `key` is the key source the engine ran on.
// median.hip, LDS group-sort path for (key cell, row) partitions: see group_sort_kernel<uint32_t, true>
int32_t group_order_partitions(pandrs_hip_ctx *c, uint64_t *pk, uint32_t *prow, const uint32_t *offsets, uint32_t NB,
                               uint32_t P, uint8_t *only);
// Median and Nunique: no engine state, filled by median_pass (a per-group sort) after the engine run
inline bool is_sorted_pass_op(int op) { return op == PANDRS_HIP_AGG_MEDIAN || op == PANDRS_HIP_AGG_NUNIQUE; }
int32_t median_pass(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, const void *vdata, const uint8_t *vnull,
                    int kind, int fin_index, int mode = 0);   // mode 0: median, 1: number of distinct values

}  // namespace pandrs
