// entries.hip — host entry points of the groupby family (gfx950, wave64): staging of host columns,
// multi-key packing / dictionary encoding, pandrs_hip_groupby_agg / _partials / _merge / _partials_split,
// group_by's own row -> group result (pandrs_hip_groupby_indices), and the multi-GPU row shuffle.
// The kernels of the aggregate pipeline are in groupby.hip / partition.hip.
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace pandrs {

// ---- staging helpers (host mem_space) -----------------------------------------------------------
struct Stager {
    pandrs_hip_ctx *c;
    int32_t space;
    int32_t status = 0;
    std::vector<const void *> pinned;      // host ranges page-locked for this call (GpuConfig.use_pinned_memory)
    // copies `bytes` from a caller pointer into the staging arena when it lives on the host
    const void *in(const void *p, size_t bytes) {
        if (!p || space == PANDRS_HIP_MEM_DEVICE || status || bytes == 0) return p;      // (an empty column is never dereferenced: nothing to stage)
        void *d = c->staging.take<uint8_t>(bytes + 16);
        if (!d) { status = fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small"); return nullptr; }
        if (config_use_pinned_memory() && bytes >= (size_t(1) << 20) &&
            hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess)
            pinned.push_back(p);                // (a range that cannot be registered is simply copied pageable)
        else
            (void)hipGetLastError();
        hipError_t e = hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) status = fail(PANDRS_HIP_ERR_COMPUTATION, "H2D copy failed: %s", hipGetErrorString(e));
        return d;
    }
    ~Stager() {
        if (pinned.empty()) return;
        (void)hipStreamSynchronize(c->stream);   // the DMA engines may still be reading the ranges
        for (const void *p : pinned) (void)hipHostUnregister(const_cast<void *>(p));
    }
};

static int32_t check_cols(const pandrs_hip_column *cols, int n, const char *what, bool keys = false) {
    for (int i = 0; i < n; i++) {
        if (cols[i].dtype < PANDRS_HIP_I64 || cols[i].dtype > (keys ? PANDRS_HIP_CELL64 : PANDRS_HIP_BOOLBITS))
            return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "%s column %d: bad dtype %d", what, i, cols[i].dtype);
    }
    return 0;
}

// ================================================================================================
// group_by's own result: the row -> group assignment (reference grouping.rs:22-115 builds
// HashMap<Vec<String>, Vec<usize>> with every group's row indices ascending, :98-103; GroupBy.groups
// is a pub field, types.rs:52, read by filter / transform / custom aggregations and par_groupby).
// Device form: CSR — group keys, offsets[G+1], rows[N] with each group's rows ascending.
//   radix partition of (key cell, row) -> segmented sort by (key, row) -> run starts -> scan -> emit.
// ================================================================================================
__global__ void zero_range_kernel(uint64_t *a, const uint32_t *beg, const uint32_t *end) {
    const uint32_t b = *beg, e = *end;
    for (uint32_t i = b + blockIdx.x * blockDim.x + threadIdx.x; i < e; i += gridDim.x * blockDim.x) a[i] = 0ull;
}
__global__ void run_start_flags_kernel(const uint64_t *keys, const uint32_t *null_beg, uint32_t n, uint32_t *flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || i == *null_beg || keys[i - 1] != keys[i]) ? 1u : 0u;
}
__global__ void emit_groups_kernel(const uint64_t *keys, const uint32_t *prow, const uint32_t *null_beg, uint32_t n,
                                   const uint32_t *flag, const uint32_t *gid, uint64_t *out_keys, uint8_t *out_null,
                                   int64_t *out_off, int64_t *out_rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_rows[i] = prow[i];
    if (flag[i]) {
        const uint32_t g = gid[i];
        const bool nul = i >= *null_beg;
        out_keys[g] = nul ? 0ull : keys[i];
        out_null[g] = nul ? 1 : 0;
        out_off[g] = i;
    }
    if (i == 0) out_off[gid[n]] = n;
}

// Rows sorted by (radix partition, key cell, row) with the run starts marked: the common first half
// of groupby_indices and of the dictionary encoding of wide multi-key columns.  Everything lives in
// c->work (which is re-sized here); synchronises the stream to learn the number of groups.
struct SortedGroups {
    uint64_t *pk = nullptr;         // key cells in sorted order (NULL-key rows last, cells zeroed)
    uint32_t *prow = nullptr;       // original row of every sorted position
    uint32_t *flag = nullptr;       // 1 at the first position of a group
    uint32_t *gid = nullptr;        // exclusive scan of flag; gid[n] = number of groups
    const uint32_t *null_beg = nullptr;   // device: first position of the NULL-key group
    int64_t G = 0;
};
static int32_t build_sorted_groups(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, SortedGroups *o) {
    const size_t ws = engine_workspace_bytes(n_rows, 1, 0) + two_pass_workspace_bytes(n_rows, 1, 1) + 3 * Arena::padded(size_t(n_rows + 2) * 4)
                    + Arena::padded(scan_seg_count((size_t)n_rows + 1) * 4) + segsort_workspace_bytes(n_rows, P_MAX + 2, 4) + (size_t(9) << 20);
    ST_TRY(c->work.ensure(ws, c->stream));
    o->pk = c->work.take<uint64_t>(n_rows + 1);
    o->prow = c->work.take<uint32_t>(n_rows + 2); o->flag = c->work.take<uint32_t>(n_rows + 2); o->gid = c->work.take<uint32_t>(n_rows + 2);
    uint32_t *seg = c->work.take<uint32_t>(scan_seg_count((size_t)n_rows + 1));
    if (!o->pk || !o->prow || !o->flag || !o->gid || !seg) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (group index)");
    // partitions of ~9 K rows: most are grouped and ordered in LDS (group_sort_kernel), the general sort takes the rest
    // (not with nearly unique keys — more than ~1 K groups per partition overflow the LDS table everywhere: a sampled
    // estimate decides for large inputs, small ones just try)
    bool fast = !c->opt.median_generic;
    if (fast && n_rows >= (int64_t(1) << 21)) {
        int64_t est = 0;
        ST_TRY(estimate_groups(c, key, n_rows, &est));
        const int64_t p_fast = std::min<int64_t>(std::max<int64_t>(1, (int64_t)std::ceil((double)n_rows / 9000.0)), P_MAX);
        fast = (double)est / (double)p_fast <= 1200.0;
    }
    int64_t P = std::min<int64_t>(std::max<int64_t>(1, (int64_t)std::ceil((double)n_rows / (fast ? 9000.0 : 4900.0))), P_MAX);
    uint8_t *only = fast ? c->work.take<uint8_t>((size_t)P + 16) : nullptr;
    if (fast && !only) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (group index)");
    PartInfo part{};
    ScatterArgs sa{};
    sa.key = key; sa.pkeys = o->pk; sa.n_rows = n_rows; sa.P = (uint32_t)P; sa.seed = 0x6A09E667u; sa.allow_two_pass = 1;
    sa.mv[sa.n_move++] = MoveDesc{nullptr, o->prow, 3, 0};
    ST_TRY(radix_partition(c, sa, &part, PANDRS_HIP_PHASE_HISTOGRAM, PANDRS_HIP_PHASE_SCAN, PANDRS_HIP_PHASE_SCATTER));
    o->null_beg = part.offsets + (size_t)P * part.NB;
    const uint32_t *null_end = part.offsets + (size_t)(P + 1) * part.NB;
    PhaseTimer pt(c, PANDRS_HIP_PHASE_OTHER);
    if (fast) ST_TRY(group_order_partitions(c, o->pk, o->prow, part.offsets, part.NB, (uint32_t)P, only));
    hipLaunchKernelGGL(zero_range_kernel, dim3(256), dim3(256), 0, c->stream, o->pk, o->null_beg, null_end);
    ST_TRY(segmented_sort_u32(c, o->pk, o->prow, part.offsets, part.NB, (uint32_t)P + 1, n_rows, only));
    hipLaunchKernelGGL(run_start_flags_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream,
                       o->pk, o->null_beg, (uint32_t)n_rows, o->flag);
    HIP_TRY(hipMemsetAsync(o->flag + n_rows, 0, 8, c->stream));
    ST_TRY(exclusive_scan_u32(c, o->flag, (size_t)n_rows + 1, o->gid, seg));
    uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
    HIP_TRY(hipMemcpyAsync(h, o->gid + n_rows, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    o->G = h[0];
    return 0;
}

// ---- multi-key groupby: composite keys packed into one 8-byte cell -------------------------------
// The reference groups on Vec<String> (grouping.rs:62-104).  Here every key column is reduced to
// an order-preserving code  code = sortable(cell) - min + (nullable ? 1 : 0)  (0 = NULL) of just
// enough bits, the codes are concatenated into one u64 cell, the single-key engine runs on it, and
// the group keys are unpacked afterwards.  Exact; fails cleanly when the codes need > 64 bits.
constexpr int MAX_KEYS = 8;
struct PackDesc {
    KeyDesc key[MAX_KEYS];
    uint64_t min_sortable[MAX_KEYS];
    uint32_t shift[MAX_KEYS], bits[MAX_KEYS], nullable[MAX_KEYS];
    // dictionary-encoded columns (too wide for their share of the 64 bits): dense[k][row] = the row's
    // group id in column k alone, dict[k][id] / dict_null[k][id] = that group's cell / null flag
    const uint32_t *dense[MAX_KEYS];
    const uint64_t *dict[MAX_KEYS];
    const uint8_t *dict_null[MAX_KEYS];
    int n_keys;
};
__device__ __forceinline__ uint64_t sortable_cell(int dtype, uint64_t cell) {
    if (dtype == PANDRS_HIP_I64) return cell ^ 0x8000000000000000ull;
    if (dtype == PANDRS_HIP_F64) return (cell >> 63) ? ~cell : (cell | 0x8000000000000000ull);
    return cell;
}
__device__ __forceinline__ uint64_t unsortable_cell(int dtype, uint64_t s) {
    if (dtype == PANDRS_HIP_I64) return s ^ 0x8000000000000000ull;
    if (dtype == PANDRS_HIP_F64) return (s >> 63) ? (s & 0x7FFFFFFFFFFFFFFFull) : ~s;
    return s;
}
// out[2k] = min sortable cell, out[2k+1] = max, over the non-null rows of key k
__global__ void key_minmax_kernel(PackDesc d, int64_t n, uint64_t *out) {
    for (int k = 0; k < d.n_keys; k++) {
        uint64_t mn = ~0ull, mx = 0;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            if (key_is_null(d.key[k], i)) continue;
            uint64_t s = sortable_cell(d.key[k].dtype, key_cell(d.key[k], i));
            mn = s < mn ? s : mn; mx = s > mx ? s : mx;
        }
        for (int o = 32; o >= 1; o >>= 1) {
            uint64_t a = __shfl_down(mn, o, 64), b = __shfl_down(mx, o, 64);
            mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin((unsigned long long *)&out[2 * k], mn);
            atomicMax((unsigned long long *)&out[2 * k + 1], mx);
        }
    }
}
__global__ void pack_keys_kernel(PackDesc d, int64_t n, uint64_t *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t cell = 0;
    for (int k = 0; k < d.n_keys; k++) {
        uint64_t code = 0;
        if (d.dense[k]) code = d.dense[k][i];
        else if (!key_is_null(d.key[k], i))
            code = sortable_cell(d.key[k].dtype, key_cell(d.key[k], i)) - d.min_sortable[k] + d.nullable[k];
        cell |= code << d.shift[k];
    }
    out[i] = cell;
}
// keys[0][g] holds the packed cell; rewrite keys[k][g] / key_null[k][g] for every key column
__global__ void unpack_keys_kernel(PackDesc d, int64_t g, size_t cap, uint64_t *keys, uint8_t *knull) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g) return;
    const uint64_t cell = keys[i];
    for (int k = 0; k < d.n_keys; k++) {
        uint64_t mask = d.bits[k] >= 64 ? ~0ull : ((1ull << d.bits[k]) - 1);
        uint64_t code = (cell >> d.shift[k]) & mask;
        if (d.dense[k]) {
            keys[(size_t)k * cap + i] = d.dict[k][code];
            knull[(size_t)k * cap + i] = d.dict_null[k][code];
            continue;
        }
        bool nul = d.nullable[k] && code == 0;
        keys[(size_t)k * cap + i] = nul ? 0ull : unsortable_cell(d.key[k].dtype, code - d.nullable[k] + d.min_sortable[k]);
        knull[(size_t)k * cap + i] = nul ? 1 : 0;
    }
}

// dictionary encoding of one key column from its sorted group structure
__global__ void dense_emit_kernel(const uint64_t *pk, const uint32_t *prow, const uint32_t *null_beg, uint32_t n,
                                  const uint32_t *flag, const uint32_t *gid, uint32_t *code, uint64_t *dict, uint8_t *dict_null) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = gid[i] + flag[i] - 1;      // gid = starts before i; a start opens group gid[i]
    code[prow[i]] = g;
    if (flag[i]) {
        const bool nul = i >= *null_beg;
        dict[g] = nul ? 0ull : pk[i];
        dict_null[g] = nul ? 1 : 0;
    }
}

// dictionary encoding of one key column WITHOUT ordering its rows: the ordinary engine lists the column's distinct cells
// (a count-only run), their position in that list is the code, and every row looks its cell up in an open-addressing
// table key -> position (16-byte entries, load <= 0.5: a few probes, L2 / MALL resident for all but huge dictionaries)
struct DictEntry { uint64_t key; uint32_t id, pad; };
__global__ void dict_build_kernel(const uint64_t *gkeys, const uint8_t *gnull, uint32_t G, DictEntry *table, uint32_t mask,
                                  uint32_t *special, uint64_t *dict, uint8_t *dict_null) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G) return;
    const uint64_t k = gkeys[i];
    const bool nul = gnull[i] != 0;
    dict[i] = nul ? 0ull : k;
    dict_null[i] = nul ? 1 : 0;
    if (nul) { special[1] = i; return; }
    if (k == EMPTY_KEY) { special[0] = i; return; }         // (the table's own free marker: kept beside it)
    uint32_t slot = hash32(k, 0x3C6EF372u) & mask;
    for (uint32_t probes = 0; probes <= mask; probes++) {
        if (atomicCAS((unsigned long long *)&table[slot].key, EMPTY_KEY, k) == EMPTY_KEY) { table[slot].id = i; return; }
        slot = (slot + 1) & mask;
    }
}
__global__ void dict_lookup_kernel(KeyDesc key, int64_t n, const DictEntry *table, uint32_t mask, const uint32_t *special,
                                   uint32_t *code) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t id = 0;
    if (key_is_null(key, i)) id = special[1];
    else {
        const uint64_t k = key_cell(key, i);
        if (k == EMPTY_KEY) id = special[0];
        else {
            uint32_t slot = hash32(k, 0x3C6EF372u) & mask;
            for (uint32_t probes = 0; probes <= mask; probes++) {
                const DictEntry e = table[slot];
                if (e.key == k) { id = e.id; break; }
                if (e.key == EMPTY_KEY) break;               // unreachable: every cell of the column is in the list
                slot = (slot + 1) & mask;
            }
        }
    }
    code[i] = id;
}
// -> 0 and *G_out, or SMALL_NOT_TAKEN-like -1 when the column has more distinct cells than one engine run lists (the caller
// then orders the rows instead: build_sorted_groups)
static int32_t hashed_dictionary(pandrs_hip_ctx *c, const KeyDesc &key, int dtype, int64_t n_rows, uint32_t *code, uint64_t *dict,
                                 uint8_t *dict_null, int64_t dict_cap, int64_t *G_out) {
    Plan cpl;
    int32_t cdt = PANDRS_HIP_I64; uint8_t chn = 0;
    pandrs_hip_agg_spec cspec{}; cspec.col = 0; cspec.op = PANDRS_HIP_AGG_COUNT;
    ST_TRY(build_plan(&cdt, &chn, 1, &cspec, 1, cpl));
    RowSource krs;
    krs.n_rows = n_rows; krs.key = key;
    Options saved = c->opt;
    c->opt.groups_hint = 0; c->opt.partitions = 0;
    pandrs_hip_timings tsave = c->timings;
    c->quiet++;
    int32_t st = run_engine(c, krs, cpl, /*merge=*/false, /*partials=*/false, 1, dtype, 1);
    c->quiet--;
    c->opt = saved;
    c->timings = tsave;
    if (st) {
        if (c->capacity_exceeded) { c->capacity_exceeded = false; return -1; }
        return st;
    }
    const int64_t G = c->gb.n_groups;
    if (G <= 0 || G > dict_cap) return -1;
    uint32_t slots = 1024;
    while ((int64_t)slots < 2 * G) slots <<= 1;
    ST_TRY(c->work.ensure((size_t)slots * sizeof(DictEntry) + 8192, c->stream));
    DictEntry *table = c->work.take<DictEntry>(slots);
    uint32_t *special = c->work.take<uint32_t>(64);
    if (!table || !special) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (dictionary)");
    HIP_TRY(hipMemsetAsync(table, 0xFF, (size_t)slots * sizeof(DictEntry), c->stream));
    HIP_TRY(hipMemsetAsync(special, 0, 256, c->stream));
    hipLaunchKernelGGL(dict_build_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, c->stream, c->gb.keys, c->gb.key_null,
                       (uint32_t)G, table, slots - 1, special, dict, dict_null);
    hipLaunchKernelGGL(dict_lookup_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, key, n_rows, table,
                       slots - 1, special, code);
    HIP_TRY(hipGetLastError());
    *G_out = G;
    return 0;
}

// Stages the remaining key columns, measures the code widths and replaces `key` (on entry: key
// column 0) by the packed cells.  `pd` is kept for unpack_keys_kernel.
static int32_t pack_multi_key(pandrs_hip_ctx *c, Stager &stg, const pandrs_hip_column *keys, int n_keys, int64_t n_rows,
                              KeyDesc &key, PackDesc &pd) {
    // composite key: per-column code widths from a min/max pass, then one packed cell per row
    PhaseTimer pt(c, PANDRS_HIP_PHASE_OTHER);
    pd.n_keys = n_keys;
    for (int k = 0; k < n_keys; k++)
        pd.key[k] = KeyDesc{k == 0 ? key.data : stg.in(keys[k].data, dtype_bytes(keys[k].dtype, n_rows)),
                            k == 0 ? key.null_bits : (const uint8_t *)stg.in(keys[k].null_mask, (n_rows + 7) / 8),
                            nullptr, keys[k].dtype};
    if (stg.status) return stg.status;
    ST_TRY(c->work.ensure(1 << 16, c->stream));
    uint64_t *mm = c->work.take<uint64_t>(2 * MAX_KEYS);
    if (!mm) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (multi-key)");
    uint64_t *h = reinterpret_cast<uint64_t *>(c->pinned);
    for (int k = 0; k < n_keys; k++) { h[2 * k] = ~0ull; h[2 * k + 1] = 0; }
    HIP_TRY(hipMemcpyAsync(mm, h, 16 * n_keys, hipMemcpyHostToDevice, c->stream));
    int blocks = (int)std::min<int64_t>(2048, (n_rows + 255) / 256);
    hipLaunchKernelGGL(key_minmax_kernel, dim3(blocks), dim3(256), 0, c->stream, pd, n_rows, mm);
    HIP_TRY(hipMemcpyAsync(h, mm, 16 * n_keys, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t need[MAX_KEYS], total = 0;
    for (int k = 0; k < n_keys; k++) {
        uint64_t mn = h[2 * k], mx = h[2 * k + 1];
        pd.nullable[k] = keys[k].null_mask ? 1u : 0u;
        if (mn > mx) { mn = mx = 0; }                       // every row null
        uint64_t span = mx - mn;                            // codes 0..span (+1 when nullable)
        uint32_t bits = 0;
        bool wide = span == ~0ull || (pd.nullable[k] && span + 1 == ~0ull);
        uint64_t top = span + pd.nullable[k];
        while (bits < 64 && (top >> bits)) bits++;
        if (wide) bits = 65;
        if (bits == 0) bits = 1;
        pd.min_sortable[k] = mn; need[k] = bits; total += bits;
    }
    {   // packed cells + (only when the codes do not fit) room for the dictionaries of the widest columns
        uint32_t t = total, n_dict = 0, nd[MAX_KEYS];
        for (int k = 0; k < n_keys; k++) nd[k] = need[k];
        while (t > 64 && n_dict < (uint32_t)n_keys) {       // every encoded column needs at most 32 bits
            int w = 0;
            for (int k = 1; k < n_keys; k++) if (nd[k] > nd[w]) w = k;
            t -= nd[w]; nd[w] = 0; t += 32; n_dict++;
        }
        ST_TRY(c->packed.ensure(Arena::padded(size_t(n_rows) * 8) + (size_t)n_dict * (Arena::padded(size_t(n_rows) * 4) +
                                Arena::padded(size_t(n_rows) * 8) + Arena::padded(size_t(n_rows))) + 8192, c->stream));
    }
    uint64_t *packed = c->packed.take<uint64_t>(n_rows);
    if (!packed) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "packed arena too small");
    // Too wide for one 64-bit cell (e.g. two hashed i64 ids): the widest columns are dictionary-encoded
    // — a column cannot have more distinct values than rows, so its dense group id needs <= 32 bits —
    // until the codes fit.  One group-index pass (partition + segmented sort) per encoded column.
    while (total > 64) {
        int w = -1;
        for (int k = 0; k < n_keys; k++)
            if (!pd.dense[k] && (w < 0 || need[k] > need[w])) w = k;
        if (w < 0 || need[w] <= 1)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED,
                        "multi-key groupby: the key columns need more than 64 bits even with every column "
                        "dictionary-encoded (%u bits); not supported on the device path", total);
        uint32_t *code = c->packed.take<uint32_t>(n_rows);
        uint64_t *dict = c->packed.take<uint64_t>(n_rows);
        uint8_t *dnull = c->packed.take<uint8_t>(n_rows);
        if (!code || !dict || !dnull) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "packed arena too small (dictionary)");
        int64_t G = 0;
        int32_t hst = c->opt.sorted_dictionary ? -1 : hashed_dictionary(c, pd.key[w], keys[w].dtype, n_rows, code, dict, dnull, n_rows, &G);
        if (hst > 0) return hst;
        if (hst < 0) {           // more distinct cells than one engine run lists: order the rows by the column instead
            SortedGroups sg;
            ST_TRY(build_sorted_groups(c, pd.key[w], n_rows, &sg));
            hipLaunchKernelGGL(dense_emit_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream,
                               sg.pk, sg.prow, sg.null_beg, (uint32_t)n_rows, sg.flag, sg.gid, code, dict, dnull);
            HIP_TRY(hipGetLastError());
            G = sg.G;
        }
        pd.dense[w] = code; pd.dict[w] = dict; pd.dict_null[w] = dnull;
        uint32_t bits = 1;
        while (bits < 32 && ((uint64_t)(G - 1) >> bits)) bits++;
        total -= need[w]; need[w] = bits; total += bits;
    }
    uint32_t shift = 0;
    for (int k = 0; k < n_keys; k++) { pd.shift[k] = shift; pd.bits[k] = need[k]; shift += need[k]; }
    hipLaunchKernelGGL(pack_keys_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, pd, n_rows, packed);
    HIP_TRY(hipGetLastError());
    key = KeyDesc{packed, nullptr, nullptr, DT_CELL};
    return 0;
}

static int32_t ordered_fold_pass(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, const RowSource &rs, const Plan &pl,
                                 const pandrs_hip_column *vals, const pandrs_hip_agg_spec *aggs, int n_aggs);

int32_t groupby_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *keys,
                      int32_t n_keys, int64_t n_rows, const pandrs_hip_column *vals, int32_t n_vals,
                      const pandrs_hip_agg_spec *aggs, int32_t n_aggs, bool partials,
                      int64_t *out_n_groups, int32_t *out_n_state) {
    if (!c || !out_n_groups || n_rows < 0 || n_keys < 1 || n_vals < 0 || n_aggs < 0 || !keys ||
        (n_vals && !vals) || (n_aggs && !aggs))
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "groupby: bad arguments");
    if (n_keys > MAX_KEYS)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "more than %d key columns", MAX_KEYS);
    if (n_keys > 1 && partials)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "multi-key partials are not mergeable across shards yet");
    ST_TRY(check_cols(keys, n_keys, "key", true));
    ST_TRY(check_cols(vals, n_vals, "value"));
    for (int k = 0; k < n_keys; k++)
        if (n_rows > 0 && !keys[k].data) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "key column %d has no data", k);
    std::vector<int32_t> dts(std::max(n_vals, 1));
    std::vector<uint8_t> hn(std::max(n_vals, 1));
    for (int i = 0; i < n_vals; i++) { dts[i] = vals[i].dtype; hn[i] = vals[i].null_mask != nullptr; }
    Plan pl;
    ST_TRY(build_plan(dts.data(), hn.data(), n_vals, aggs, n_aggs, pl));
    if (partials && pl.has_median)      // (nested merges of slices keep the Median slots as placeholders)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED,
                    "Std/Var/Median/First/Last partial states are not mergeable across shards yet");

    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    RowSource rs;
    rs.n_rows = n_rows;
    Stager stg{c, mem_space};
    if (mem_space == PANDRS_HIP_MEM_HOST && n_rows > 0) {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_STAGE_IN);
        size_t need = 0;
        for (int k = 0; k < n_keys; k++) need += dtype_bytes(keys[k].dtype, n_rows) + (n_rows + 7) / 8 + 1024;
        for (int s = 0; s < pl.n_src; s++) need += size_t(n_rows) * 8 + (n_rows + 7) / 8 + 1024;
        for (int a = 0; a < n_aggs; a++)
            if (is_sorted_pass_op(aggs[a].op)) need += size_t(n_rows) * 8 + (n_rows + 7) / 8 + 1024;
        ST_TRY(c->staging.ensure(need + (1 << 16), c->stream));
    }
    rs.key = KeyDesc{stg.in(keys[0].data, dtype_bytes(keys[0].dtype, n_rows)),
                     (const uint8_t *)stg.in(keys[0].null_mask, (n_rows + 7) / 8), nullptr, keys[0].dtype};
    for (int s = 0; s < pl.n_src; s++) {
        const pandrs_hip_column &v = vals[pl.src_col[s]];
        if (n_rows > 0 && !v.data) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "value column %d has no data", pl.src_col[s]);
        rs.val_data[s] = stg.in(v.data, size_t(n_rows) * 8);
        rs.val_null_bits[s] = (const uint8_t *)stg.in(v.null_mask, (n_rows + 7) / 8);
    }
    // Median columns: device views of the original columns (shared with the plan's sources when the
    // column is also aggregated otherwise)
    const void *med_data[MAX_AGGS]{};
    const uint8_t *med_null[MAX_AGGS]{};
    for (int a = 0; a < n_aggs && pl.has_median; a++) {
        if (!is_sorted_pass_op(aggs[a].op)) continue;
        const int col = aggs[a].col;
        for (int s = 0; s < pl.n_src; s++)
            if (pl.src_col[s] == col) { med_data[a] = rs.val_data[s]; med_null[a] = rs.val_null_bits[s]; }
        for (int b = 0; b < a && !med_data[a]; b++)
            if (is_sorted_pass_op(aggs[b].op) && aggs[b].col == col) { med_data[a] = med_data[b]; med_null[a] = med_null[b]; }
        if (!med_data[a]) {
            if (n_rows > 0 && !vals[col].data) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "value column %d has no data", col);
            med_data[a] = stg.in(vals[col].data, size_t(n_rows) * 8);
            med_null[a] = (const uint8_t *)stg.in(vals[col].null_mask, (n_rows + 7) / 8);
        }
    }
    if (stg.status) return stg.status;
    PackDesc pd{};
    if (n_keys > 1 && n_rows > 0) ST_TRY(pack_multi_key(c, stg, keys, n_keys, n_rows, rs.key, pd));
    ST_TRY(run_engine(c, rs, pl, /*merge=*/false, partials, n_aggs, keys[0].dtype, n_keys));
    for (int a = 0; a < n_aggs && pl.has_median; a++)       // Median: a per-group sort, one pass per column
        if (is_sorted_pass_op(aggs[a].op))
            ST_TRY(median_pass(c, rs.key, n_rows, med_data[a], med_null[a], pl.fin_kind[a], a,
                               aggs[a].op == PANDRS_HIP_AGG_NUNIQUE ? 1 : 0));
    if (c->opt.deterministic && !partials) ST_TRY(ordered_fold_pass(c, rs.key, n_rows, rs, pl, vals, aggs, n_aggs));
    if (n_keys > 1 && c->gb.n_groups > 0) {
        hipLaunchKernelGGL(unpack_keys_kernel, dim3((unsigned)((c->gb.n_groups + 255) / 256)), dim3(256), 0, c->stream,
                           pd, c->gb.n_groups, (size_t)c->gb.cap, c->gb.keys, c->gb.key_null);
        HIP_TRY(hipGetLastError());
    }
    // B = N (K + 8 C) + G (K + 8 A) (+ N/8 per masked column)
    {
        int64_t K = 0;
        for (int k = 0; k < n_keys; k++)
            K += keys[k].dtype == PANDRS_HIP_U32CODE ? 4 : (keys[k].dtype == PANDRS_HIP_BOOLBITS ? 0 : 8);
        int64_t b = n_rows * (K + 8 * (int64_t)pl.n_src) + c->gb.n_groups * (K + 8 * (int64_t)n_aggs);
        if (keys[0].null_mask) b += n_rows / 8;
        for (int s = 0; s < pl.n_src; s++) if (rs.val_null_bits[s]) b += n_rows / 8;
        c->timings.algorithmic_bytes = b;
    }
    ST_TRY(timings_end(c));
    *out_n_groups = c->gb.n_groups;
    if (out_n_state) *out_n_state = c->gb.n_state;
    return 0;
}

int32_t groupby_indices_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *keys, int32_t n_keys,
                              int64_t n_rows, int64_t *out_n_groups) {
    if (!c || !out_n_groups || n_rows < 0 || n_keys < 1 || !keys)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "groupby_indices: bad arguments");
    if (n_keys > MAX_KEYS) return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "more than %d key columns", MAX_KEYS);
    ST_TRY(check_cols(keys, n_keys, "key", true));
    for (int k = 0; k < n_keys; k++)
        if (n_rows > 0 && !keys[k].data) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "key column %d has no data", k);
    if (n_rows >= (int64_t(1) << 32) - 16384)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "groupby_indices: more than 2^32 rows per call");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    c->gr = GroupsResult{};
    Stager stg{c, mem_space};
    if (mem_space == PANDRS_HIP_MEM_HOST && n_rows > 0) {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_STAGE_IN);
        size_t need = 0;
        for (int k = 0; k < n_keys; k++) need += dtype_bytes(keys[k].dtype, n_rows) + (n_rows + 7) / 8 + 1024;
        ST_TRY(c->staging.ensure(need + (1 << 16), c->stream));
    }
    KeyDesc key{stg.in(keys[0].data, dtype_bytes(keys[0].dtype, n_rows)),
                (const uint8_t *)stg.in(keys[0].null_mask, (n_rows + 7) / 8), nullptr, keys[0].dtype};
    if (stg.status) return stg.status;
    PackDesc pd{};
    if (n_keys > 1 && n_rows > 0) ST_TRY(pack_multi_key(c, stg, keys, n_keys, n_rows, key, pd));
    int64_t G = 0;
    if (n_rows > 0) {
        SortedGroups sg;
        ST_TRY(build_sorted_groups(c, key, n_rows, &sg));
        G = sg.G;
        PhaseTimer pt(c, PANDRS_HIP_PHASE_OTHER);
        ST_TRY(c->groups.ensure((size_t)n_keys * (Arena::padded(size_t(G) * 8) + Arena::padded(size_t(G))) +
                                Arena::padded(size_t(G + 1) * 8) + Arena::padded(size_t(n_rows) * 8) + 4096, c->stream));
        GroupsResult &r = c->gr;
        r.cap = G; r.n_keys = n_keys; r.n_rows = n_rows; r.n_groups = G;
        r.keys = c->groups.take<uint64_t>((size_t)n_keys * G);
        r.key_null = c->groups.take<uint8_t>((size_t)n_keys * G);
        r.offsets = c->groups.take<int64_t>(G + 1);
        r.rows = c->groups.take<int64_t>(n_rows);
        if (!r.keys || !r.key_null || !r.offsets || !r.rows) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "groups arena too small");
        hipLaunchKernelGGL(emit_groups_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream,
                           sg.pk, sg.prow, sg.null_beg, (uint32_t)n_rows, sg.flag, sg.gid, r.keys, r.key_null, r.offsets, r.rows);
        if (n_keys > 1)
            hipLaunchKernelGGL(unpack_keys_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, c->stream,
                               pd, G, (size_t)G, r.keys, r.key_null);
        HIP_TRY(hipGetLastError());
    } else {
        ST_TRY(c->groups.ensure(4096, c->stream));
        c->gr.offsets = c->groups.take<int64_t>(1);
        HIP_TRY(hipMemsetAsync(c->gr.offsets, 0, 8, c->stream));
        c->gr.n_keys = n_keys;
    }
    c->gr.valid = true;
    {
        int64_t K = 0;
        for (int k = 0; k < n_keys; k++)
            K += keys[k].dtype == PANDRS_HIP_U32CODE ? 4 : (keys[k].dtype == PANDRS_HIP_BOOLBITS ? 0 : 8);
        c->timings.algorithmic_bytes = n_rows * (K + 8) + G * (K + 8);
    }
    ST_TRY(timings_end(c));
    *out_n_groups = G;
    return 0;
}

// ================================================================================================
// Row shuffle by key owner (multi-GPU): the radix partitioner with P = n_ranks.
// ================================================================================================
constexpr uint32_t OWNER_SEED = 0x1B873593u;    // independent of every partition seed used locally

__global__ void bytes_to_bitmap_kernel(const uint8_t *bytes, int64_t n, uint8_t *out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b * 8 >= n) return;
    uint32_t v = 0;
    for (int j = 0; j < 8; j++)
        if (b * 8 + j < n && bytes[b * 8 + j]) v |= 1u << j;
    out[b] = (uint8_t)v;
}

struct HashKeys { KeyDesc key[MAX_KEYS]; int n_keys; };
__global__ void key_hash_cells_kernel(HashKeys hk, int64_t n, uint64_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < hk.n_keys; k++) {
        const bool nul = key_is_null(hk.key[k], i);
        uint64_t x = nul ? 0xD1B54A32D192ED03ull : key_cell(hk.key[k], i);
        x ^= h + (nul ? 1 : 0);
        x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
        h = x;
    }
    out[i] = h;
}

int32_t key_hash_cells_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *keys, int32_t n_keys,
                             int64_t n_rows, uint64_t *out_cells) {
    if (!c || !keys || n_keys < 1 || n_keys > MAX_KEYS || n_rows < 0 || (n_rows && !out_cells))
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "key_hash_cells: bad arguments");
    ST_TRY(check_cols(keys, n_keys, "key", true));
    if (n_rows == 0) return 0;
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    Stager stg{c, mem_space};
    uint64_t *dst = out_cells;
    if (mem_space == PANDRS_HIP_MEM_HOST) {
        size_t need = size_t(n_rows) * 8 + 4096;
        for (int k = 0; k < n_keys; k++) need += dtype_bytes(keys[k].dtype, n_rows) + (n_rows + 7) / 8 + 1024;
        ST_TRY(c->staging.ensure(need + (1 << 16), c->stream));
        dst = c->staging.take<uint64_t>(n_rows);
        if (!dst) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
    }
    HashKeys hk{};
    hk.n_keys = n_keys;
    for (int k = 0; k < n_keys; k++) {
        if (!keys[k].data) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "key column %d has no data", k);
        hk.key[k] = KeyDesc{stg.in(keys[k].data, dtype_bytes(keys[k].dtype, n_rows)),
                            (const uint8_t *)stg.in(keys[k].null_mask, (n_rows + 7) / 8), nullptr, keys[k].dtype};
    }
    if (stg.status) return stg.status;
    hipLaunchKernelGGL(key_hash_cells_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream, hk, n_rows, dst);
    HIP_TRY(hipGetLastError());
    if (mem_space == PANDRS_HIP_MEM_HOST) HIP_TRY(hipMemcpyAsync(out_cells, dst, size_t(n_rows) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int32_t bytes_to_bitmap_entry(pandrs_hip_ctx *c, int32_t mem_space, const uint8_t *bytes, int64_t n, uint8_t *out) {
    if (!c || n < 0 || (n && (!bytes || !out))) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bytes_to_bitmap: bad arguments");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint8_t *src = bytes; uint8_t *dst = out;
    const size_t nb = (size_t)(n + 7) / 8;
    if (mem_space == PANDRS_HIP_MEM_HOST) {
        ST_TRY(c->staging.ensure((size_t)n + nb + 4096, c->stream));
        uint8_t *d = c->staging.take<uint8_t>(n); dst = c->staging.take<uint8_t>(nb);
        if (!d || !dst) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
        HIP_TRY(hipMemcpyAsync(d, bytes, (size_t)n, hipMemcpyHostToDevice, c->stream));
        src = d;
    }
    hipLaunchKernelGGL(bytes_to_bitmap_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, src, n, dst);
    HIP_TRY(hipGetLastError());
    if (mem_space == PANDRS_HIP_MEM_HOST) HIP_TRY(hipMemcpyAsync(out, dst, nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int32_t shuffle_split_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *key,
                            const pandrs_hip_column *payload, int32_t n_payload, int64_t n_rows, int32_t n_ranks,
                            int32_t drop_null_keys, int64_t *out_counts, int64_t *out_n_rows) {
    if (!c || !key || !out_counts || !out_n_rows || n_rows < 0 || n_payload < 0 || n_payload > 16 || (n_payload && !payload) ||
        n_ranks < 1 || n_ranks > 1024)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "shuffle_split: bad arguments");
    ST_TRY(check_cols(key, 1, "key", true));
    for (int p = 0; p < n_payload; p++)
        if (payload[p].dtype != PANDRS_HIP_I64 && payload[p].dtype != PANDRS_HIP_F64 && payload[p].dtype != PANDRS_HIP_U32CODE)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "shuffle_split: payload column %d has dtype %d (i64, f64 or u32 codes only)", p, payload[p].dtype);
    if (n_rows >= (int64_t(1) << 32) - 16384)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "shuffle_split: more than 2^32 rows per call");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    c->sh = ShuffleResult{};
    for (int r = 0; r < n_ranks; r++) out_counts[r] = 0;
    *out_n_rows = 0;
    if (n_rows == 0) { c->sh.valid = true; c->sh.n_payload = n_payload; return timings_end(c); }
    Stager stg{c, mem_space};
    if (mem_space == PANDRS_HIP_MEM_HOST) {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_STAGE_IN);
        size_t need = dtype_bytes(key->dtype, n_rows) + (n_rows + 7) / 8 + 1024;
        for (int p = 0; p < n_payload; p++) need += dtype_bytes(payload[p].dtype, n_rows) + (n_rows + 7) / 8 + 1024;
        ST_TRY(c->staging.ensure(need + (1 << 16), c->stream));
    }
    if (!key->data) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "shuffle_split: key column has no data");
    KeyDesc kd{stg.in(key->data, dtype_bytes(key->dtype, n_rows)), (const uint8_t *)stg.in(key->null_mask, (n_rows + 7) / 8), nullptr, key->dtype};
    ST_TRY(c->work.ensure(engine_workspace_bytes(n_rows, 0, 0) + (1 << 20), c->stream));
    size_t out_bytes = Arena::padded(size_t(n_rows) * 8) + Arena::padded(size_t(n_rows)) + 4096;
    for (int p = 0; p < n_payload; p++) out_bytes += Arena::padded(size_t(n_rows) * 8) + Arena::padded(size_t(n_rows));
    ST_TRY(c->shuf.ensure(out_bytes, c->stream));
    ShuffleResult &r = c->sh;
    r.n_payload = n_payload;
    r.cells = c->shuf.take<uint64_t>(n_rows);
    r.key_null = c->shuf.take<uint8_t>(n_rows);
    if (!r.cells || !r.key_null) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "shuffle arena too small");
    ScatterArgs sa{};
    sa.key = kd; sa.pkeys = r.cells; sa.n_rows = n_rows; sa.P = (uint32_t)n_ranks; sa.seed = OWNER_SEED;
    if (kd.null_bits) sa.mv[sa.n_move++] = MoveDesc{kd.null_bits, r.key_null, 6, 0};
    else HIP_TRY(hipMemsetAsync(r.key_null, 0, (size_t)n_rows, c->stream));
    for (int p = 0; p < n_payload; p++) {
        if (!payload[p].data) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "shuffle_split: payload column %d has no data", p);
        const void *d = stg.in(payload[p].data, dtype_bytes(payload[p].dtype, n_rows));
        const uint8_t *m = (const uint8_t *)stg.in(payload[p].null_mask, (n_rows + 7) / 8);
        r.pay[p] = c->shuf.take<uint64_t>(n_rows);
        if (!r.pay[p]) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "shuffle arena too small");
        sa.mv[sa.n_move++] = MoveDesc{d, r.pay[p], payload[p].dtype == PANDRS_HIP_U32CODE ? 4 : 0, 0};
        if (m) {
            r.pay_null[p] = c->shuf.take<uint8_t>(n_rows);
            if (!r.pay_null[p]) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "shuffle arena too small");
            sa.mv[sa.n_move++] = MoveDesc{m, r.pay_null[p], 6, 0};
        }
    }
    if (stg.status) return stg.status;
    PartInfo part{};
    ST_TRY(radix_partition(c, sa, &part, PANDRS_HIP_PHASE_HISTOGRAM, PANDRS_HIP_PHASE_SCAN, PANDRS_HIP_PHASE_SCATTER));
    uint32_t *bounds = c->work.take<uint32_t>(n_ranks + 2);
    if (!bounds) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (shuffle)");
    gather_part_offsets(c, part.offsets, part.NB, (uint32_t)n_ranks + 2, bounds);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> hb((size_t)n_ranks + 2);
    HIP_TRY(hipMemcpyAsync(hb.data(), bounds, ((size_t)n_ranks + 2) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int q = 0; q < n_ranks; q++) out_counts[q] = (int64_t)hb[q + 1] - (int64_t)hb[q];
    const int64_t n_null = (int64_t)hb[n_ranks + 1] - (int64_t)hb[n_ranks];
    if (!drop_null_keys) out_counts[n_ranks - 1] += n_null;     // the NULL-key partition sits right behind the last rank's rows
    r.n_rows = drop_null_keys ? (int64_t)hb[n_ranks] : (int64_t)hb[n_ranks + 1];
    r.valid = true;
    *out_n_rows = r.n_rows;
    c->timings.algorithmic_bytes = n_rows * (8 + 8 * (int64_t)n_payload) * 2;
    return timings_end(c);
}

// packed partial records [n][W] -> column arrays keys[n] | key_null[n] | states[W-2][n]
__global__ void unpack_records_kernel(const uint64_t *rec, int64_t n, int W, uint64_t *keys,
                                      uint8_t *knull, uint64_t *states) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *r = rec + (size_t)i * W;
    keys[i] = r[0];
    knull[i] = r[1] != 0;
    for (int s = 0; s < W - 2; s++) states[(size_t)s * n + i] = r[2 + s];
}

int32_t groupby_merge_entry(pandrs_hip_ctx *c, int32_t mem_space, int32_t key_dtype,
                            const uint64_t *records,
                            int64_t n_rows, const int32_t *val_dtypes, int32_t n_vals,
                            const uint8_t *val_has_nulls, const pandrs_hip_agg_spec *aggs,
                            int32_t n_aggs, int64_t *out_n_groups) {
    if (!c || !out_n_groups || n_rows < 0 || (n_rows && !records) || n_vals < 0 || n_aggs < 0)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "groupby_merge: bad arguments");
    Plan pl;
    ST_TRY(build_plan(val_dtypes, val_has_nulls, n_vals, aggs, n_aggs, pl));
    if (pl.has_median)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED,
                    "Std/Var/Median/First/Last partial states are not mergeable across shards yet");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    const size_t n_state = 1 + (size_t)pl.n_states, W = 2 + n_state;
    // staging arena: [records (host mode only)] keys | key_null | states
    ST_TRY(c->staging.ensure(size_t(n_rows) * 8 * W * (mem_space == PANDRS_HIP_MEM_HOST ? 2 : 1) + size_t(n_rows) * 16 + (1 << 16), c->stream));
    RowSource rs;
    rs.n_rows = n_rows;
    if (n_rows > 0) {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_STAGE_IN);
        Stager stg{c, mem_space};
        const uint64_t *drec = (const uint64_t *)stg.in(records, size_t(n_rows) * 8 * W);
        if (stg.status) return stg.status;
        uint64_t *dk = c->staging.take<uint64_t>(n_rows);
        uint8_t *dn = c->staging.take<uint8_t>(n_rows);
        uint64_t *ds = c->staging.take<uint64_t>(size_t(n_rows) * n_state);
        if (!dk || !dn || !ds) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
        hipLaunchKernelGGL(unpack_records_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream,
                           drec, n_rows, (int)W, dk, dn, ds);
        HIP_TRY(hipGetLastError());
        rs.key = KeyDesc{dk, nullptr, dn, DT_CELL};
        rs.merge_states = ds;
        rs.merge_stride = (size_t)n_rows;
    }
    ST_TRY(run_engine(c, rs, pl, /*merge=*/true, /*partials=*/false, n_aggs, key_dtype));
    c->timings.algorithmic_bytes = n_rows * (int64_t)(8 * W) + c->gb.n_groups * (8 + 8 * (int64_t)n_aggs);
    ST_TRY(timings_end(c));
    *out_n_groups = c->gb.n_groups;
    return 0;
}

// ---- deterministic mode ("deterministic" = 1): f64 Sum / Mean and Std / Var folded in ASCENDING ROW ORDER ----------
// The engine's f64 sums re-associate (LDS atomics in arrival order: within 1e-9 of the reference, not bit-identical,
// and not the same from run to run).  Here every group's rows are put in ascending row order (build_sorted_groups:
// group_by's own row lists), the values are gathered into that order, and ONE thread per group folds them
// sequentially — the reference's loop (aggregation.rs:625-648 Sum / Mean, :557-584 + :881-903 Std / Var), so the
// result is bit-identical to it.  Opt-in: one sort of the rows + one gather and one fold per aggregate
// (~10 ms per 100 M rows with ~100-row groups; a group of r rows costs r dependent adds on one lane).
struct OrdEntry { uint64_t key; double value; };

__global__ void group_starts_kernel(const uint32_t *flag, const uint32_t *gid, uint32_t n, uint32_t *goff) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) goff[gid[i]] = i;
    if (i == 0) goff[gid[n]] = n;
}
__global__ void gather_sorted_values_kernel(const uint32_t *prow, uint32_t n, const uint64_t *vdata, const uint8_t *vnull,
                                            uint64_t *sv, uint8_t *sn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = prow[i];
    sv[i] = vdata[r];
    sn[i] = vnull ? (uint8_t)bit_at(vnull, r) : 0;
}
__global__ void ordered_fold_kernel(const uint64_t *pk, const uint32_t *goff, uint32_t G, const uint32_t *null_beg,
                                    const uint64_t *sv, const uint8_t *sn, int kind, int op, OrdEntry *table, uint32_t mask) {
#pragma clang fp contract(off)      // the reference rounds the product and the sum separately: no fused multiply-add here
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint32_t beg = goff[g], end = goff[g + 1];
    auto val = [&](uint32_t i) { return kind == 0 ? __longlong_as_double((long long)sv[i]) : (double)(int64_t)sv[i]; };
    double sum = 0.0;
    uint64_t cnt = 0;
    for (uint32_t i = beg; i < end; i++)
        if (!sn[i]) { sum += val(i); cnt++; }
    double out;
    if (op == PANDRS_HIP_AGG_SUM) out = sum;
    else if (op == PANDRS_HIP_AGG_MEAN) out = cnt ? sum / (double)cnt : 0.0;
    else {                                               // calculate_variance (aggregation.rs:881-903)
        double var = 0.0;
        if (cnt > 0) {
            const double n = (double)cnt, mean = sum / n;
            double ssd = 0.0;
            for (uint32_t i = beg; i < end; i++)
                if (!sn[i]) { const double d = val(i) - mean; ssd += d * d; }
            var = cnt > 1 ? ssd / (n - 1.0) : 0.0;
        }
        out = op == PANDRS_HIP_AGG_STD ? sqrt(var) : var;
    }
    // publish under the group's key (the engine's result rows are in another order)
    if (beg >= *null_beg) { table[mask + 2].value = out; return; }
    const uint64_t key = pk[beg];
    if (key == EMPTY_KEY) { table[mask + 1].value = out; return; }
    uint32_t slot = hash32(key, 0x2545F491u) & mask;
    for (uint32_t probes = 0; probes <= mask; probes++) {
        const uint64_t old = atomicCAS((unsigned long long *)&table[slot].key, EMPTY_KEY, key);
        if (old == EMPTY_KEY) { table[slot].value = out; return; }
        slot = (slot + 1) & mask;
    }
}
__global__ void ordered_lookup_kernel(const uint64_t *gkeys, const uint8_t *gnull, int64_t n_groups, const OrdEntry *table,
                                      uint32_t mask, double *out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_groups) return;
    const uint64_t k = gkeys[j];
    double v = 0.0;
    if (gnull[j]) v = table[mask + 2].value;
    else if (k == EMPTY_KEY) v = table[mask + 1].value;
    else {
        uint32_t slot = hash32(k, 0x2545F491u) & mask;
        for (uint32_t probes = 0; probes <= mask; probes++) {
            const OrdEntry e = table[slot];
            if (e.key == k) { v = e.value; break; }
            if (e.key == EMPTY_KEY) break;
            slot = (slot + 1) & mask;
        }
    }
    out[j] = v;
}

static bool wants_ordered_fold(const pandrs_hip_agg_spec &a, int dtype) {
    return ((a.op == PANDRS_HIP_AGG_SUM || a.op == PANDRS_HIP_AGG_MEAN) && dtype == PANDRS_HIP_F64) ||
           a.op == PANDRS_HIP_AGG_STD || a.op == PANDRS_HIP_AGG_VAR;
}

// overwrites the order-dependent aggregates of c->gb with their ascending-row-order folds
static int32_t ordered_fold_pass(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, const RowSource &rs, const Plan &pl,
                                 const pandrs_hip_column *vals, const pandrs_hip_agg_spec *aggs, int n_aggs) {
    GroupbyResult &res = c->gb;
    const int64_t G = res.n_groups;
    if (G <= 0 || n_rows <= 0) return 0;
    bool any = false;
    for (int a = 0; a < n_aggs; a++) any = any || wants_ordered_fold(aggs[a], vals[aggs[a].col].dtype);
    if (!any) return 0;
    PhaseTimer pt(c, PANDRS_HIP_PHASE_OTHER);
    c->quiet++;
    struct Unquiet { pandrs_hip_ctx *c; ~Unquiet() { c->quiet--; } } unq{c};
    SortedGroups sg;
    ST_TRY(build_sorted_groups(c, key, n_rows, &sg));
    if (sg.G != G) return fail(PANDRS_HIP_ERR_COMPUTATION, "deterministic pass found %lld groups, the engine %lld", (long long)sg.G, (long long)G);
    uint32_t cap = 64;
    while ((double)cap < 1.5 * (double)G) cap <<= 1;
    ST_TRY(c->temp.ensure(Arena::padded(size_t(G + 2) * 4) + Arena::padded(size_t(n_rows) * 8) + Arena::padded(size_t(n_rows)) +
                          Arena::padded(size_t(cap + 4) * 16) + 8192, c->stream));
    uint32_t *goff = c->temp.take<uint32_t>(G + 2);
    uint64_t *sv = c->temp.take<uint64_t>(n_rows);
    uint8_t *sn = c->temp.take<uint8_t>(n_rows);
    OrdEntry *table = c->temp.take<OrdEntry>((size_t)cap + 4);
    if (!goff || !sv || !sn || !table) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "temp arena too small (deterministic)");
    const unsigned nb = (unsigned)((n_rows + 255) / 256);
    hipLaunchKernelGGL(group_starts_kernel, dim3(nb), dim3(256), 0, c->stream, sg.flag, sg.gid, (uint32_t)n_rows, goff);
    int gathered_src = -1;
    for (int a = 0; a < n_aggs; a++) {
        if (!wants_ordered_fold(aggs[a], vals[aggs[a].col].dtype)) continue;
        const int s = pl.fin_src[a];
        if (s != gathered_src) {
            hipLaunchKernelGGL(gather_sorted_values_kernel, dim3(nb), dim3(256), 0, c->stream, sg.prow, (uint32_t)n_rows,
                               reinterpret_cast<const uint64_t *>(rs.val_data[s]), rs.val_null_bits[s], sv, sn);
            gathered_src = s;
        }
        HIP_TRY(hipMemsetAsync(table, 0xFF, size_t(cap) * 16, c->stream));
        HIP_TRY(hipMemsetAsync(&table[cap], 0, 64, c->stream));
        hipLaunchKernelGGL(ordered_fold_kernel, dim3((unsigned)((G + 127) / 128)), dim3(128), 0, c->stream, sg.pk, goff, (uint32_t)G,
                           sg.null_beg, sv, sn, (int)pl.src_kind[s], (int)aggs[a].op, table, cap - 1);
        hipLaunchKernelGGL(ordered_lookup_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, c->stream, res.keys, res.key_null, G,
                           table, cap - 1, res.aggs + (size_t)a * res.cap);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// device-to-device copy of the retained result's first key row, its null bytes and its first aggregate row
// (the fused join's <= G local sums on their way into the groupby exchange, dist.hip)
int32_t export_result_columns(pandrs_hip_ctx *c, uint64_t *cells, uint8_t *nulls, double *agg0, int64_t *out_n) {
    std::lock_guard<std::mutex> lock(c->mu);
    GroupbyResult &res = c->gb;
    if (!res.valid || res.partials || res.n_aggs < 1) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no finished groupby result in this context");
    const size_t n = (size_t)res.n_groups;
    if (n) {
        HIP_TRY(hipMemcpyAsync(cells, res.keys, n * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(nulls, res.key_null, n, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(agg0, res.aggs, n * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    *out_n = res.n_groups;
    return 0;
}

// ---- partial split for the all-to-all -------------------------------------------------------------
// Owner bucketing with workgroup-level aggregation: LDS counters per rank, ONE global atomic per
// (workgroup, rank) — a per-record global atomic on n_ranks addresses serialises at the memory side.
constexpr int OS_THREADS = 256, OS_RPT = 16;

__global__ __launch_bounds__(OS_THREADS) void owner_count_kernel(const uint64_t *keys, const uint8_t *knull,
                                                                 int64_t n, uint32_t n_ranks, uint32_t *counts) {
    __shared__ uint32_t cnt[1024];
    for (uint32_t r = threadIdx.x; r < n_ranks; r += OS_THREADS) cnt[r] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * OS_THREADS * OS_RPT;
    for (int q = 0; q < OS_RPT; q++) {
        int64_t i = base + q * OS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&cnt[knull[i] ? 0u : owner_of(keys[i], n_ranks)], 1u);
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < n_ranks; r += OS_THREADS)
        if (cnt[r]) atomicAdd(&counts[r], cnt[r]);
}
// writes packed records [key, key_null, states...] rank-contiguously
__global__ __launch_bounds__(OS_THREADS) void owner_scatter_kernel(const uint64_t *keys, const uint8_t *knull,
                                                                   const uint64_t *states, size_t in_stride,
                                                                   int n_state, int64_t n, uint32_t n_ranks,
                                                                   uint32_t *cursors, uint64_t *out_records) {
    __shared__ uint32_t cnt[1024];
    for (uint32_t r = threadIdx.x; r < n_ranks; r += OS_THREADS) cnt[r] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * OS_THREADS * OS_RPT;
    uint32_t own[OS_RPT], rank[OS_RPT];
#pragma unroll
    for (int q = 0; q < OS_RPT; q++) {
        int64_t i = base + q * OS_THREADS + threadIdx.x;
        own[q] = 0xFFFFFFFFu;
        if (i < n) {
            own[q] = knull[i] ? 0u : owner_of(keys[i], n_ranks);
            rank[q] = atomicAdd(&cnt[own[q]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < n_ranks; r += OS_THREADS) {
        uint32_t c = cnt[r];
        cnt[r] = c ? atomicAdd(&cursors[r], c) : 0u;     // cnt[] now holds this workgroup's base per rank
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < OS_RPT; q++) {
        if (own[q] == 0xFFFFFFFFu) continue;
        int64_t i = base + q * OS_THREADS + threadIdx.x;
        uint64_t *o = out_records + (size_t)(cnt[own[q]] + rank[q]) * (2 + n_state);
        o[0] = keys[i];
        o[1] = knull[i];
        for (int s = 0; s < n_state; s++) o[2 + s] = states[(size_t)s * in_stride + i];
    }
}

// The in-library exchange's split (dist.hip).  The retained partial records leave as ONE BLOCK PER OWNER, each block a small
// column store: block o = [W][n_o] words, W = 2 + n_state (key cell, key-null word, states), at word W * (records of the owners
// before o) — one message per peer, written and read with unit stride.  Stream-ordered, no host round trip: the scatter derives
// the block bases from the device counts itself, and its first workgroup leaves the counts as the int64 row the count
// all-gather sends.
constexpr int OB_RPT = 4;
__global__ __launch_bounds__(OS_THREADS) void owner_scatter_blocks_kernel(const uint64_t *keys, const uint8_t *knull,
                                                                          const uint64_t *states, size_t in_stride,
                                                                          int n_state, int64_t n, uint32_t n_ranks,
                                                                          const uint32_t *counts, uint32_t *cursors,
                                                                          uint64_t *out, int64_t *count_row) {
    __shared__ uint32_t cnt[1024], base[1024], wt[17];
    const uint32_t tid = threadIdx.x;
    for (uint32_t r = tid; r < n_ranks; r += OS_THREADS) cnt[r] = 0;
    {   // exclusive prefix of the owners' counts (n_ranks <= 1024: four per thread)
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t r = tid * 4 + q; v[q] = r < n_ranks ? counts[r] : 0u; sum += v[q]; }
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<OS_THREADS>(sum, wt, &tot);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t r = tid * 4 + q;
            if (r < n_ranks) { base[r] = ex; if (blockIdx.x == 0) count_row[r] = (int64_t)v[q]; }
            ex += v[q];
        }
    }
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * OS_THREADS * OB_RPT;
    uint32_t own[OB_RPT], rank[OB_RPT];
#pragma unroll
    for (int q = 0; q < OB_RPT; q++) {
        const int64_t i = first + q * OS_THREADS + tid;
        own[q] = 0xFFFFFFFFu;
        if (i < n) {
            own[q] = knull[i] ? 0u : owner_of(keys[i], n_ranks);
            rank[q] = atomicAdd(&cnt[own[q]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t r = tid; r < n_ranks; r += OS_THREADS) {
        const uint32_t k = cnt[r];
        cnt[r] = k ? atomicAdd(&cursors[r], k) : 0u;          // this workgroup's first row inside owner r's block
    }
    __syncthreads();
    const size_t W = 2 + (size_t)n_state;
#pragma unroll
    for (int q = 0; q < OB_RPT; q++) {
        if (own[q] == 0xFFFFFFFFu) continue;
        const int64_t i = first + q * OS_THREADS + tid;
        const size_t n_o = counts[own[q]];
        uint64_t *blk = out + W * (size_t)base[own[q]] + (cnt[own[q]] + rank[q]);
        blk[0] = keys[i];
        blk[n_o] = knull[i];
        for (int st = 0; st < n_state; st++) blk[(size_t)(2 + st) * n_o] = states[(size_t)st * in_stride + i];
    }
}

// blocks as received (block r = [W][n_r] at word W * roff_r) -> the merge's columns: keys[n] | key_null[n] bytes | states[W - 2][n]
__global__ __launch_bounds__(256) void unblock_records_kernel(const uint64_t *blocks, const int64_t *roff, uint32_t n_src, int64_t n, int W,
                                                              uint64_t *keys, uint8_t *knull, uint64_t *states) {
    __shared__ int64_t off[1025];
    for (uint32_t r = threadIdx.x; r <= n_src; r += 256) off[r] = roff[r];
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    uint32_t lo = 0, hi = n_src;                               // the source block of row e: off[lo] <= e < off[lo + 1]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= e) lo = mid; else hi = mid; }
    const size_t n_r = (size_t)(off[lo + 1] - off[lo]);
    const uint64_t *blk = blocks + (size_t)W * (size_t)off[lo] + (size_t)(e - off[lo]);
    keys[e] = blk[0];
    knull[e] = blk[n_r] != 0;
    for (int st = 0; st < W - 2; st++) states[(size_t)st * (size_t)n + e] = blk[(size_t)(2 + st) * n_r];
}

// -> `out` (device, >= max(n, 1) * (2 + n_state) words), *count_row (device int64[n_ranks]: records per owner)
int32_t partials_split_blocks_entry(pandrs_hip_ctx *c, int32_t n_ranks, uint64_t *out, int64_t *count_row) {
    if (!c || n_ranks < 1 || n_ranks > 1024 || !count_row || !out) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "partials_split: bad arguments");
    std::lock_guard<std::mutex> lock(c->mu);
    GroupbyResult &res = c->gb;
    if (!res.valid || !res.partials) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no partials retained in this context");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t n = res.n_groups;
    HIP_TRY(hipMemsetAsync(count_row, 0, (size_t)n_ranks * 8, c->stream));
    if (n == 0) return 0;
    if (c->work.cap < (1 << 16)) ST_TRY(c->work.ensure(1 << 16, c->stream));
    c->work.off = 0;
    uint32_t *counts = c->work.take<uint32_t>(2048);
    uint32_t *cursors = counts + 1024;
    HIP_TRY(hipMemsetAsync(counts, 0, 2048 * 4, c->stream));
    hipLaunchKernelGGL(owner_count_kernel, dim3((unsigned)((n + OS_THREADS * OS_RPT - 1) / (OS_THREADS * OS_RPT))), dim3(OS_THREADS), 0, c->stream,
                       res.keys, res.key_null, n, (uint32_t)n_ranks, counts);
    hipLaunchKernelGGL(owner_scatter_blocks_kernel, dim3((unsigned)((n + OS_THREADS * OB_RPT - 1) / (OS_THREADS * OB_RPT))), dim3(OS_THREADS), 0,
                       c->stream, res.keys, res.key_null, res.states, (size_t)res.cap, res.n_state, n, (uint32_t)n_ranks, counts, cursors,
                       out, count_row);
    HIP_TRY(hipGetLastError());
    return 0;
}

// The merge over received blocks (see above): one unblock pass into the staging arena, then the ordinary merge.
// `roff`: HOST array of n_src + 1 row offsets of the source blocks.
int32_t groupby_merge_blocks_entry(pandrs_hip_ctx *c, int32_t key_dtype, const uint64_t *blocks, const int64_t *roff, int32_t n_src,
                                   const int32_t *val_dtypes, int32_t n_vals, const uint8_t *val_has_nulls,
                                   const pandrs_hip_agg_spec *aggs, int32_t n_aggs, int64_t *out_n_groups) {
    if (!c || !out_n_groups || !roff || n_src < 1 || n_src > 1024 || n_vals < 0 || n_aggs < 0)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "groupby_merge: bad arguments");
    const int64_t n_rows = roff[n_src];
    if (n_rows < 0 || (n_rows && !blocks)) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "groupby_merge: bad arguments");
    Plan pl;
    ST_TRY(build_plan(val_dtypes, val_has_nulls, n_vals, aggs, n_aggs, pl));
    if (pl.has_median)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED,
                    "Std/Var/Median/First/Last partial states are not mergeable across shards yet");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    const size_t n_state = 1 + (size_t)pl.n_states, W = 2 + n_state;
    ST_TRY(c->staging.ensure(size_t(n_rows) * 8 * W + size_t(n_rows) * 16 + (size_t)(n_src + 1) * 8 + (1 << 16), c->stream));
    RowSource rs;
    rs.n_rows = n_rows;
    if (n_rows > 0) {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_STAGE_IN);
        int64_t *d_off = c->staging.take<int64_t>((size_t)n_src + 1);
        uint64_t *dk = c->staging.take<uint64_t>(n_rows);
        uint8_t *dn = c->staging.take<uint8_t>(n_rows);
        uint64_t *ds = c->staging.take<uint64_t>(size_t(n_rows) * n_state);
        if (!d_off || !dk || !dn || !ds) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
        HIP_TRY(hipMemcpyAsync(d_off, roff, (size_t)(n_src + 1) * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(unblock_records_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, c->stream,
                           blocks, d_off, (uint32_t)n_src, n_rows, (int)W, dk, dn, ds);
        HIP_TRY(hipGetLastError());
        rs.key = KeyDesc{dk, nullptr, dn, DT_CELL};
        rs.merge_states = ds;
        rs.merge_stride = (size_t)n_rows;
    }
    ST_TRY(run_engine(c, rs, pl, /*merge=*/true, /*partials=*/false, n_aggs, key_dtype));
    c->timings.algorithmic_bytes = n_rows * (int64_t)(8 * W) + c->gb.n_groups * (8 + 8 * (int64_t)n_aggs);
    ST_TRY(timings_end(c));
    *out_n_groups = c->gb.n_groups;
    return 0;
}

int32_t partials_split_entry(pandrs_hip_ctx *c, int32_t mem_space, int32_t n_ranks,
                             uint64_t *out_records, int64_t *out_counts) {
    if (!c || n_ranks < 1 || n_ranks > 1024 || !out_counts)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "partials_split: bad arguments");
    std::lock_guard<std::mutex> lock(c->mu);
    GroupbyResult &res = c->gb;
    if (!res.valid || !res.partials) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no partials retained in this context");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t n = res.n_groups;
    for (int r = 0; r < n_ranks; r++) out_counts[r] = 0;
    if (n == 0) return 0;
    if (!out_records) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null output");
    const size_t W = 2 + (size_t)res.n_state;
    size_t need = (1 << 16) + (mem_space == PANDRS_HIP_MEM_HOST ? Arena::padded(size_t(n) * 8 * W) : 0);
    if (c->work.cap < need) ST_TRY(c->work.ensure(need, c->stream));
    c->work.off = 0;
    uint32_t *counts = c->work.take<uint32_t>(2048);
    uint32_t *cursors = counts + 1024;
    HIP_TRY(hipMemsetAsync(counts, 0, 2048 * 4, c->stream));
    unsigned grid = (unsigned)((n + OS_THREADS * OS_RPT - 1) / (OS_THREADS * OS_RPT));
    hipLaunchKernelGGL(owner_count_kernel, dim3(grid), dim3(OS_THREADS), 0, c->stream, res.keys, res.key_null, n,
                       (uint32_t)n_ranks, counts);
    uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
    HIP_TRY(hipMemcpyAsync(h, counts, n_ranks * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t run = 0;
    for (int r = 0; r < n_ranks; r++) { out_counts[r] = h[r]; uint32_t t = h[r]; h[r] = run; run += t; }
    HIP_TRY(hipMemcpyAsync(cursors, h, n_ranks * 4, hipMemcpyHostToDevice, c->stream));
    uint64_t *drec = out_records;
    if (mem_space == PANDRS_HIP_MEM_HOST) {
        drec = c->work.take<uint64_t>(size_t(n) * W);
        if (!drec) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small");
    }
    hipLaunchKernelGGL(owner_scatter_kernel, dim3(grid), dim3(OS_THREADS), 0, c->stream, res.keys, res.key_null,
                       res.states, (size_t)res.cap, res.n_state, n, (uint32_t)n_ranks, cursors, drec);
    HIP_TRY(hipGetLastError());
    if (mem_space == PANDRS_HIP_MEM_HOST)
        HIP_TRY(hipMemcpyAsync(out_records, drec, size_t(n) * 8 * W, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace pandrs
