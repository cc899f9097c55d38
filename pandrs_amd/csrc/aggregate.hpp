// aggregate.hpp — argument block and state helpers shared by the aggregate kernels
// (groupby.hip: generic / RUNS / MERGE / direct kernels; aggregate2.hip: the lean persistent kernel).
#pragma once
#include "engine.hpp"

namespace pandrs {

struct AggTask { uint32_t part, beg, end, multi; };
// aggregate2: one LDS table = one partition, or one row slice of an oversized partition (multi), fed by
// the row ranges tasks[task_beg .. task_beg + n_tasks) (1 in the exact layout, up to 8 in the capacity layout)
struct AggTable { uint32_t task_beg, n_tasks, part, multi; };
// where a partition's rows are: exact layout (histogram) or capacity layout (radix_partition_sampled)
struct SegSource { const uint32_t *offsets; uint32_t NB; const uint32_t *gbeg, *gcur, *gend; };

// Std / Var over partial records (the row slices of an oversized partition): record i stands for n_i rows with sum s_i and
// squared deviations M2_i around ITS mean; the merged group has M2 = sum M2_i + sum n_i (s_i / n_i - S / N)^2 — the second
// term is the merge's own second pass over the records (a sum of non-negative terms: no cancellation)
struct MergeVar { const uint64_t *sum_col, *nn_col; int8_t ssq, sum, nn, pad[5]; };   // nn_col nullptr / nn -1: the group size
constexpr int MAX_MERGE_VAR = 8;

struct AggArgs {
    const uint64_t *pkeys;
    const uint32_t *offsets;     // partition p rows = [offsets[p*NB], offsets[(p+1)*NB])
    const int64_t *pgsize;       // merge mode: partitioned group sizes, else nullptr (=1 per row)
    uint32_t NB, P, T, seed;
    int n_src, n_states, n_fin, partials, n_rounds, round_states, second_pass;
    int direct;                  // 1: no radix partition — workgroup b pre-aggregates rows [b*chunk, (b+1)*chunk) of the
                                 // ORIGINAL columns (dkey, src[].vals, src[].valid = null BITMAP) and emits partial records
    KeyDesc dkey;
    uint32_t d_rows, d_chunk, launch_grid, d_task_cap;
    // oversized partitions are split into row slices handled by different workgroups; their groups
    // leave as partial records in the side buffers (counters[2]) and are merged afterwards
    const AggTask *tasks;        // nullptr: workgroup b = partition b
    const uint32_t *n_tasks;
    const AggTable *tables;      // aggregate2 only; n_tasks[1] = number of tables
    // aggregate2 in ROUNDS (more than 4 uniform columns; groupby.hip): launch r folds the sources [src_base, src_base + NSRC) of round
    // `cur_round` and writes the outputs whose states live in that round.  Launch 0 leaves, per table, a snapshot of its key table (keys,
    // tags, sentinel flag) and every slot's output position; the later launches start from the snapshot — same slots, same positions, no
    // second compaction — so the outputs of all rounds line up.  snap_* = nullptr: one round, as before.
    uint64_t *snap_keys;         // [n_tables][T + 2]
    uint8_t *snap_ctrl;          // [n_tables][T]
    uint32_t *snap_pos;          // [n_tables][T + 2] output position of the slot's group, 0xFFFFFFFF = free; [T + 1] = sentinel-key flag
    int32_t src_base, cur_round;
    const uint32_t *order;       // aggregate2, optional: the tables in the order they are handed out (largest first); nullptr = as listed
    uint64_t *side_keys; uint8_t *side_null; uint64_t *side_states; size_t side_cap;
    int8_t round_src_begin[MAX_ROUNDS + 1];   // sources of round r = [begin[r], begin[r+1])
    SrcDev src[MAX_STATES];      // st_* = LDS state index inside the source's round (raw rows: <= MAX_SRC columns; merges: one per state)
    int8_t kinds[MAX_STATES];    // by absolute state index (ABI / partial order)
    int8_t st_round[MAX_STATES], st_lds[MAX_STATES];
    FinDev fin[MAX_AGGS];
    // outputs (row capacity = cap)
    uint64_t *out_keys;
    uint8_t *out_null;
    double *out_aggs;            // [n_fin][cap]
    uint64_t *out_states;        // [1 + n_states][cap] when partials
    size_t cap;
    uint32_t *counters;          // [0] = n_groups, [1] = overflow flag, [2] = side records; aggregate2: [6] = overflow rows, [8] = workgroups done
    // aggregate2: the last workgroup to finish copies counters[0..2] and *scatter_flags to host_out[0..3] (pinned,
    // device-visible) and sets host_out[4] = 1, so the host polls one word instead of copying and synchronising
    const uint32_t *scatter_flags;
    uint32_t *host_out;
    // aggregate2, optional (ov_keys != nullptr): a row whose key finds no slot in a FULL table (the group estimate was too low)
    // is appended — key cell, values, validity bytes — at counters[6] instead of failing the call.  The keys of such rows are
    // in no table (a full table stays full), so they are a disjoint sub-problem: the host groups them on their own and appends
    // the groups.  Not for `multi` tables (another slice may hold the key); more than ov_cap rows => the overflow flag as before.
    uint32_t fold_min, fold_min_multi;             // aggregate2: lanes in one slot from which a wave folds them on the VALU (ordinary tables, pieces)
    MergeVar mvar[MAX_MERGE_VAR]; int n_mvar;      // merge mode with a second pass
    uint64_t *ov_keys; uint64_t *ov_vals[4]; uint8_t *ov_valid[4]; uint32_t ov_cap;      // (aggregate2 instantiates 1..4 sources)
    // aggregate2's SMALL mode (few rows: no estimate, no partition): workgroup b folds rows [b * s_chunk, ...) of the
    // ORIGINAL columns (dkey, src[].vals, src[].valid = null BITMAP) into its LDS table and flushes the table into
    // the context's global table with device-scope atomics; small_output_kernel turns that into the result
    uint32_t s_vec;                           // clustered.hip: every column (8-byte key cells included) starts on a 16-byte boundary: 16-byte loads
    uint32_t s_need_cnt;                      // 0: no output reads a group size (one global atomic less per group and workgroup)
    uint32_t s_rows, s_chunk, g_slots;        // g_slots: power of two; slot g_slots = sentinel-valued key, g_slots + 1 = NULL key
    uint64_t *g_keys;                          // [g_slots + 2], EMPTY_KEY when free
    uint32_t *g_cnt;                           // [g_slots + 2] group sizes
    uint64_t *g_states;                        // [round_states][g_slots + 2]; sums as they are, min as ~enc, max as enc (all
                                               // via atomicMax, so an all-zero table is an armed table)
};

__device__ __forceinline__ uint64_t state_identity(int8_t kind) {
    switch (kind) {
    case SK_MIN_F64: return enc_f64(__longlong_as_double(0x7FF0000000000000ll));
    case SK_MAX_F64: return enc_f64(__longlong_as_double((long long)0xFFF0000000000000ull));
    case SK_MIN_I64: return enc_i64(INT64_MAX);
    case SK_MAX_I64: return enc_i64(INT64_MIN);
    default: return 0ull;   // +0.0 / 0
    }
}
// ---- wave-wide reductions on the VALU alone (DPP row shifts + row broadcasts; no LDS crossbar): the total ends in lane 63
// and is read back as a wave-uniform value.  OP: 0 = f64 add, 1 = u64 add, 2 = u64 min, 3 = u64 max; `ident` is OP's identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_move64(uint64_t x, uint64_t ident) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)ident, (int)(uint32_t)x, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(ident >> 32), (int)(uint32_t)(x >> 32), CTRL, ROW_MASK, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
template <int OP>
__device__ __forceinline__ uint64_t wave_op64(uint64_t a, uint64_t b) {
    if (OP == 0) return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
    if (OP == 1) return a + b;
    if (OP == 2) return b < a ? b : a;
    return b > a ? b : a;
}
template <int OP>
__device__ __forceinline__ uint64_t wave_reduce64(uint64_t x, uint64_t ident) {
    x = wave_op64<OP>(x, dpp_move64<0x111, 0xF>(x, ident));          // row_shr:1
    x = wave_op64<OP>(x, dpp_move64<0x112, 0xF>(x, ident));          // row_shr:2
    x = wave_op64<OP>(x, dpp_move64<0x114, 0xF>(x, ident));          // row_shr:4
    x = wave_op64<OP>(x, dpp_move64<0x118, 0xF>(x, ident));          // row_shr:8   -> lane 15 of every row: the row's total
    x = wave_op64<OP>(x, dpp_move64<0x142, 0xA>(x, ident));          // row_bcast:15 into rows 1, 3
    x = wave_op64<OP>(x, dpp_move64<0x143, 0xC>(x, ident));          // row_bcast:31 into rows 2, 3 -> lane 63: the wave's total
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}

// natural (ABI / partial) representation of an LDS state cell
__device__ __forceinline__ uint64_t state_natural(int8_t kind, uint64_t cell) {
    switch (kind) {
    case SK_MIN_F64: case SK_MAX_F64: return (uint64_t)__double_as_longlong(dec_f64(cell));
    case SK_MIN_I64: case SK_MAX_I64: return (uint64_t)dec_i64(cell);
    default: return cell;
    }
}


// absorb.hip: hot-key absorb-and-spill pass in front of the radix path (uniform profiles, <= 4 value columns)
constexpr int MAX_ABS_SRC = 4;
struct AbsorbArgs {
    KeyDesc key;                                  // the ORIGINAL key column (any dtype, null bitmap in place)
    uint32_t n_rows, chunk, T, seed;              // workgroup w owns rows [w * chunk, (w + 1) * chunk)
    int n_src, n_lds_states;
    const uint64_t *vals[MAX_ABS_SRC];
    const uint8_t *null_bits[MAX_ABS_SRC];
    int8_t st_add[MAX_ABS_SRC], st_min[MAX_ABS_SRC], st_max[MAX_ABS_SRC], st_nn[MAX_ABS_SRC];   // LDS state index, -1 = none
    int8_t lds_kind[MAX_STATES];                  // StateKind of LDS state k
    int8_t lds_abs[MAX_STATES];                   // its absolute (ABI / partial) state index
    uint64_t *out_keys; uint8_t *out_null; uint64_t *out_states; size_t cap;      // partial records ([1 + n_states][cap])
    uint32_t *counters;                           // [1] a spill region overflowed, [2] partial records (shared with the lean aggregate's
                                                  // side records), [3] rows spilled
    // spill: row of partition p from workgroup w -> region (w * spill_P + p), spill_cap rows each, in the spill columns
    uint32_t spill_P, spill_cap;
    uint64_t *sp_keys; uint64_t *sp_vals[MAX_ABS_SRC]; uint8_t *sp_valid[MAX_ABS_SRC];
    uint32_t *sp_count;                           // [grid * spill_P] rows in each region
    int fold;                                     // 1: waves fold the lanes that sit in one slot (set when the sample's neighbours often share their key)
    int image_only;                               // 1: the table takes no key beyond the image (every workgroup then absorbs the SAME keys:
                                                  // the spilled rows' keys are disjoint from the absorbed ones — the compact spill relies on it)
    const uint64_t *hot_image;                    // optional [T]: the table every workgroup starts from (hot_image_kernel), nullptr = empty
};
constexpr uint32_t ABSORB_SEED = 0x9E3779B9u;     // bucket hash of the absorb table (the hot-key image is built with it)
bool absorb_has(int n_src, int profile);
bool launch_absorb(pandrs_hip_ctx *c, const AbsorbArgs &a, int n_src, int profile, size_t lds, uint32_t grid);
void launch_compact_spill(pandrs_hip_ctx *c, const AbsorbArgs &a, uint32_t n_wg, uint64_t *dst_keys, uint64_t *const *dst_vals, uint8_t *const *dst_valid,
                          bool has_v, uint32_t *total);
void launch_build_spill_tables(pandrs_hip_ctx *c, const uint32_t *sp_count, uint32_t n_wg, uint32_t PS, uint32_t cap_wp, uint32_t wpt,
                               AggTask *tasks, AggTable *tables, uint32_t *counts);

// aggregate2.hip: lean persistent aggregate for uniform profiles on raw partitioned rows (one round).
// Returns false when (n_src, profile) has no instantiation: the caller falls back to aggregate_kernel.
constexpr size_t AGG2_LDS_EXTRA = 16 * 128 * 4 + 64;   // per-wave retry queues of aggregate2_kernel
bool aggregate2_has(int n_src, int profile);
bool launch_aggregate2(pandrs_hip_ctx *c, const AggArgs &a, int n_src, int profile, size_t lds, uint32_t grid);
bool aggregate2_small_has(int n_src, int profile);
// clustered.hip: rows clustered by key — one pass over the ORIGINAL columns, every row chunk's groups leave as partial records
bool clustered_has(int n_src, int profile);
bool launch_clustered(pandrs_hip_ctx *c, const AggArgs &a, int n_src, int profile, size_t lds, uint32_t grid);
bool launch_clustered_parts(pandrs_hip_ctx *c, const AggArgs &a, int n_src, int profile, size_t lds, uint32_t grid);
// SMALL mode: false when (n_src, profile) has no small instantiation
bool launch_aggregate2_small(pandrs_hip_ctx *c, const AggArgs &a, int n_src, int profile, size_t lds, uint32_t grid);
void launch_small_output(pandrs_hip_ctx *c, const AggArgs &a, int n_src, int profile);
constexpr uint32_t SMALL_G_SLOTS = 1u << 15;      // global table of the small path: up to 16 K groups

}  // namespace pandrs
