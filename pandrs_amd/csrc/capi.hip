// capi.hip — extern "C" surface of libpandrs_hip.so (declarations: include/pandrs_hip.h).
#include "common.hpp"

#include <atomic>

namespace {
std::mutex g_mu;
bool g_inited = false;
pandrs_hip_config g_cfg{1, 0, 0, 1, 0, 10000};   // GpuConfig defaults, src/gpu/mod.rs:32-44
bool g_cfg_explicit = false;                     // a config was passed to pandrs_hip_init: thresholds are honoured
}  // namespace

namespace pandrs {
int64_t config_min_size_threshold() { return g_cfg_explicit ? g_cfg.min_size_threshold : 0; }
bool config_use_pinned_memory() { return g_cfg_explicit && g_cfg.use_pinned_memory != 0; }
}  // namespace pandrs

using pandrs::fail;

extern "C" {

int32_t pandrs_hip_abi_version(void) try { return PANDRS_HIP_ABI_VERSION; } catch (...) { return pandrs::on_exception("pandrs_hip_abi_version"); }

const char *pandrs_hip_last_error(void) { return pandrs::last_error().text; }

int32_t pandrs_hip_init(const pandrs_hip_config *cfg) try {
    std::lock_guard<std::mutex> lock(g_mu);
    if (cfg) { g_cfg = *cfg; g_cfg_explicit = true; }
    else g_cfg_explicit = false;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "no HIP device available: %s",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (g_cfg.device_id < 0 || g_cfg.device_id >= n)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "device_id %d out of range (%d devices)", g_cfg.device_id, n);
    g_inited = true;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_init"); }

int32_t pandrs_hip_shutdown(void) try {
    std::lock_guard<std::mutex> lock(g_mu);
    g_inited = false;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_shutdown"); }

int32_t pandrs_hip_device_count(int32_t *out_count) try {
    if (!out_count) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null out_count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *out_count = e == hipSuccess ? n : 0;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_device_count"); }

int32_t pandrs_hip_ctx_create(int32_t device_id, pandrs_hip_ctx **out_ctx) try {
    if (!out_ctx) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null out_ctx");
    *out_ctx = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (!g_inited) {
            int n = 0;
            hipError_t e = hipGetDeviceCount(&n);
            if (e != hipSuccess || n == 0)
                return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "no HIP device available");
            g_inited = true;
        }
    }
    if (!g_cfg.enabled)          // GpuConfig.enabled = false: the caller keeps its CPU path (src/gpu/mod.rs:20)
        return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "the device path is disabled by configuration (pandrs_hip_config.enabled = 0)");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0) device_id = g_cfg.device_id;
    if (device_id >= n) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "device_id %d out of range (%d devices)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    auto *c = new pandrs_hip_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    c->lds_bytes = (int)std::min<size_t>(prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin : prop.sharedMemPerBlock, 160 * 1024);
    if (c->lds_bytes < 64 * 1024) c->lds_bytes = 64 * 1024;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i < PANDRS_HIP_MAX_PHASES; i++) {
        HIP_TRY(hipEventCreate(&c->ev_begin[i]));
        HIP_TRY(hipEventCreate(&c->ev_end[i]));
    }
    HIP_TRY(hipEventCreate(&c->ev_call_begin));
    HIP_TRY(hipEventCreate(&c->ev_call_end));
    HIP_TRY(hipHostMalloc(&c->pinned, 1 << 16, hipHostMallocDefault));
    pandrs::arena_limit() = g_cfg.memory_limit > 0 ? (size_t)g_cfg.memory_limit : 0;   // GpuConfig.memory_limit
    *out_ctx = c;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_ctx_create"); }

int32_t pandrs_hip_ctx_destroy(pandrs_hip_ctx *c) try {
    if (!c) return PANDRS_HIP_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->work.release(); c->result.release(); c->staging.release(); c->temp.release(); c->result2.release(); c->result3.release(); c->side.release(); c->super.release(); c->packed.release(); c->pairs.release(); c->groups.release(); c->shuf.release(); c->absorb.release(); c->overflow.release();
    for (int i = 0; i < PANDRS_HIP_MAX_PHASES; i++) { (void)hipEventDestroy(c->ev_begin[i]); (void)hipEventDestroy(c->ev_end[i]); }
    (void)hipEventDestroy(c->ev_call_begin); (void)hipEventDestroy(c->ev_call_end);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->est_table) (void)hipFree(c->est_table);
    if (c->census_table) (void)hipFree(c->census_table);
    if (c->small_table) (void)hipFree(c->small_table);
    for (auto &kv : c->resident) (void)hipFree(kv.second.base);      // columns the caller never released
    (void)hipStreamDestroy(c->stream);
    delete c;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_ctx_destroy"); }

int32_t pandrs_hip_ctx_synchronize(pandrs_hip_ctx *c) try {
    if (!c) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null ctx");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_ctx_synchronize"); }

int32_t pandrs_hip_ctx_reserve(pandrs_hip_ctx *c, int64_t workspace_bytes) try {
    if (!c || workspace_bytes < 0) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bad arguments");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    return c->work.ensure((size_t)workspace_bytes, c->stream);
} catch (...) { return pandrs::on_exception("pandrs_hip_ctx_reserve"); }

// ---- resident columns -------------------------------------------------------------------------------------------
// The reference's columns are immutable Arc<[T]> (src/column/int64_column.rs:10) that every operator Arc-clones
// (src/optimized/dataframe/transformations.rs:524-577): uploaded ONCE, a column serves every later aggregate / join
// from HBM.  One hipMalloc per column: data, then the null bitmap on a 256-byte boundary.
int32_t pandrs_hip_column_upload(pandrs_hip_ctx *c, const pandrs_hip_column *host, int64_t n_rows, pandrs_hip_column *out) try {
    if (!c || !host || !out || n_rows < 0) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bad arguments");
    if (host->dtype < PANDRS_HIP_I64 || host->dtype > PANDRS_HIP_CELL64) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bad dtype %d", host->dtype);
    if (!host->data && n_rows > 0) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null data pointer");
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const size_t db = pandrs::dtype_bytes(host->dtype, n_rows), mb = host->null_mask ? (size_t)((n_rows + 7) / 8) : 0;
    const size_t dpad = (db + 255) & ~size_t(255);
    const size_t total = dpad + ((mb + 255) & ~size_t(255)) + 256;     // (+256: the kernels' 16-byte tail loads stay inside)
    if (pandrs::arena_limit() && c->resident_bytes + total > pandrs::arena_limit())
        return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "resident columns (%zu + %zu bytes) exceed pandrs_hip_config.memory_limit (%zu)",
                    c->resident_bytes, total, pandrs::arena_limit());
    char *base = nullptr;
    HIP_TRY(hipMalloc((void **)&base, total));
    pandrs::alloc_events()++;
    hipError_t e = db ? hipMemcpyAsync(base, host->data, db, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    if (e == hipSuccess && mb) e = hipMemcpyAsync(base + dpad, host->null_mask, mb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);       // the host buffers may be dropped as soon as we return
    if (e != hipSuccess) { (void)hipFree(base); return fail(PANDRS_HIP_ERR_COMPUTATION, "column upload failed: %s", hipGetErrorString(e)); }
    c->resident[base] = pandrs_hip_ctx::Resident{base, total};
    c->resident_bytes += total;
    out->data = base; out->null_mask = mb ? reinterpret_cast<const uint8_t *>(base + dpad) : nullptr;
    out->dtype = host->dtype; out->reserved = 0;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_column_upload"); }

int32_t pandrs_hip_column_release(pandrs_hip_ctx *c, const pandrs_hip_column *col) try {
    if (!c || !col) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bad arguments");
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->resident.find(col->data);
    if (it == c->resident.end()) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "not a column of pandrs_hip_column_upload on this context (or released twice)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));     // a call that reads the column may still be in flight
    HIP_TRY(hipFree(it->second.base));
    c->resident_bytes -= it->second.bytes;
    c->resident.erase(it);
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_column_release"); }

int32_t pandrs_hip_alloc_events(int64_t *out_device_allocations) try {
    if (!out_device_allocations) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null output");
    *out_device_allocations = pandrs::alloc_events().load();
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_alloc_events"); }

int32_t pandrs_hip_resident_bytes(pandrs_hip_ctx *c, int64_t *out_bytes, int64_t *out_columns) try {
    if (!c) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null ctx");
    std::lock_guard<std::mutex> lock(c->mu);
    if (out_bytes) *out_bytes = (int64_t)c->resident_bytes;
    if (out_columns) *out_columns = (int64_t)c->resident.size();
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_resident_bytes"); }

int32_t pandrs_hip_ctx_set_option(pandrs_hip_ctx *c, const char *name, int64_t value) try {
    if (name && !std::strcmp(name, "test_throw")) {
        // tests of the exception firewall (common.hpp on_exception; needs no context): the entry point's host code throws
        if (value == 1) throw std::bad_alloc();
        if (value == 2) { std::vector<int> v; (void)v.at(7); }                 // std::out_of_range, as a container would raise it
        if (value == 3) throw 42;                                              // not a std::exception
        if (value == 4) { std::vector<uint8_t> v; v.resize(~size_t(0) >> 1); } // a real oversized staging resize: std::length_error / bad_alloc
        return PANDRS_HIP_OK;
    }
    if (!c || !name) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bad arguments");
    std::lock_guard<std::mutex> lock(c->mu);
    if (!std::strcmp(name, "groups_hint")) c->opt.groups_hint = value;
    else if (!std::strcmp(name, "scatter_staged")) c->opt.scatter_staged = value;
    else if (!std::strcmp(name, "partitions")) c->opt.partitions = value;
    else if (!std::strcmp(name, "scatter_threads")) c->opt.scatter_threads = value;
    else if (!std::strcmp(name, "src_per_round")) c->opt.src_per_round = value;
    else if (!std::strcmp(name, "p_target")) c->opt.p_target = value;
    else if (!std::strcmp(name, "join_generic")) c->opt.join_generic = value;
    else if (!std::strcmp(name, "join_no_l2")) c->opt.join_no_l2 = value;
    else if (!std::strcmp(name, "join_pair_p")) c->opt.join_pair_p = value;
    else if (!std::strcmp(name, "two_pass")) c->opt.two_pass = value;
    else if (!std::strcmp(name, "scatter_wide")) c->opt.scatter_wide = value;
    else if (!std::strcmp(name, "two_pass_min_p")) c->opt.two_pass_min_p = value;
    else if (!std::strcmp(name, "join_no_pairpart")) c->opt.join_no_pairpart = value;
    else if (!std::strcmp(name, "no_runs")) c->opt.no_runs = value;
    else if (!std::strcmp(name, "join_one_pass")) c->opt.join_one_pass = value;
    else if (!std::strcmp(name, "load_pct")) c->opt.load_pct = value;
    else if (!std::strcmp(name, "generic_aggregate")) c->opt.generic_aggregate = value;
    else if (!std::strcmp(name, "median_generic")) c->opt.median_generic = value;
    else if (!std::strcmp(name, "shared_cursors")) c->opt.shared_cursors = value;
    else if (!std::strcmp(name, "no_direct")) c->opt.no_direct = value;
    else if (!std::strcmp(name, "no_absorb")) c->opt.no_absorb = value;
    else if (!std::strcmp(name, "no_chao")) c->opt.no_chao = value;
    else if (!std::strcmp(name, "no_census")) c->opt.no_census = value;
    else if (!std::strcmp(name, "no_overflow_run")) c->opt.no_overflow_run = value;
    else if (!std::strcmp(name, "tail_groups_hint")) c->opt.tail_groups_hint = value;
    else if (!std::strcmp(name, "sorted_dictionary")) c->opt.sorted_dictionary = value;
    else if (!std::strcmp(name, "wide_slices")) c->opt.wide_slices = value;
    else if (!std::strcmp(name, "no_table_order")) c->opt.no_table_order = value;
    else if (!std::strcmp(name, "no_lean_rounds")) c->opt.no_lean_rounds = value;
    else if (!std::strcmp(name, "no_window_bound")) c->opt.no_window_bound = value;
    else if (!std::strcmp(name, "no_burst_kernel")) c->opt.no_burst_kernel = value;
    else if (!std::strcmp(name, "no_profile_rounds")) c->opt.no_profile_rounds = value;
    else if (!std::strcmp(name, "no_clustered")) c->opt.no_clustered = value;
    else if (!std::strcmp(name, "clustered_chunk")) c->opt.clustered_chunk = value;
    else if (!std::strcmp(name, "clustered_max_runs_pct")) c->opt.clustered_max_runs_pct = value;
    else if (!std::strcmp(name, "fold_min")) c->opt.fold_min = value;
    else if (!std::strcmp(name, "fold_min_multi")) c->opt.fold_min_multi = value;
    else if (!std::strcmp(name, "slice_over")) c->opt.slice_over = value;
    else if (!std::strcmp(name, "no_hot_image")) c->opt.no_hot_image = value;
    else if (!std::strcmp(name, "no_slice")) c->opt.no_slice = value;
    else if (!std::strcmp(name, "p_max")) c->opt.p_max = value;
    else if (!std::strcmp(name, "slice_rows")) c->opt.slice_rows = value;
    else if (!std::strcmp(name, "agg_v1")) c->opt.agg_v1 = value;
    else if (!std::strcmp(name, "no_small")) { c->opt.no_small = value; c->small_skip = c->small_backoff = 0; }
    else if (!std::strcmp(name, "small_chunk")) c->opt.small_chunk = value;
    else if (!std::strcmp(name, "deterministic")) c->opt.deterministic = value;
    else if (!std::strcmp(name, "exact_partition")) c->opt.exact_partition = value;
    else if (!std::strcmp(name, "agg_ablate")) c->opt.agg_ablate = value;
    else if (!std::strcmp(name, "agg_depth")) c->opt.agg_depth = value;
    else return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_ctx_set_option"); }

int32_t pandrs_hip_get_timings(pandrs_hip_ctx *c, pandrs_hip_timings *out) try {
    if (!c || !out) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "bad arguments");
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->timings_pending) { (void)hipSetDevice(c->device); (void)pandrs::timings_resolve(c); }
    *out = c->timings;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_get_timings"); }

int32_t pandrs_hip_groupby_agg(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *keys,
                               int32_t n_keys, int64_t n_rows, const pandrs_hip_column *vals,
                               int32_t n_vals, const pandrs_hip_agg_spec *aggs, int32_t n_aggs,
                               int64_t *out_n_groups) try {
    ST_TRY(pandrs::below_threshold(n_rows));
    return pandrs::groupby_entry(ctx, mem_space, keys, n_keys, n_rows, vals, n_vals, aggs, n_aggs,
                                 false, out_n_groups, nullptr);
} catch (...) { return pandrs::on_exception("pandrs_hip_groupby_agg"); }

int32_t pandrs_hip_groupby_partials(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *keys,
                                    int32_t n_keys, int64_t n_rows, const pandrs_hip_column *vals,
                                    int32_t n_vals, const pandrs_hip_agg_spec *aggs, int32_t n_aggs,
                                    int64_t *out_n_groups, int32_t *out_n_state) try {
    return pandrs::groupby_entry(ctx, mem_space, keys, n_keys, n_rows, vals, n_vals, aggs, n_aggs,
                                 true, out_n_groups, out_n_state);
} catch (...) { return pandrs::on_exception("pandrs_hip_groupby_partials"); }

int32_t pandrs_hip_partials_split(pandrs_hip_ctx *ctx, int32_t mem_space, int32_t n_ranks,
                                  uint64_t *out_records, int64_t *out_counts) try {
    return pandrs::partials_split_entry(ctx, mem_space, n_ranks, out_records, out_counts);
} catch (...) { return pandrs::on_exception("pandrs_hip_partials_split"); }

int32_t pandrs_hip_groupby_merge(pandrs_hip_ctx *ctx, int32_t mem_space, int32_t key_dtype,
                                 const uint64_t *records,
                                 int64_t n_rows, const int32_t *val_dtypes, int32_t n_vals,
                                 const uint8_t *val_has_nulls, const pandrs_hip_agg_spec *aggs,
                                 int32_t n_aggs, int64_t *out_n_groups) try {
    return pandrs::groupby_merge_entry(ctx, mem_space, key_dtype, records, n_rows,
                                       val_dtypes, n_vals, val_has_nulls, aggs, n_aggs, out_n_groups);
} catch (...) { return pandrs::on_exception("pandrs_hip_groupby_merge"); }

int32_t pandrs_hip_groupby_fetch(pandrs_hip_ctx *c, int32_t mem_space, uint64_t *const *out_keys,
                                 uint8_t *const *out_key_null, double *const *out_aggs) try {
    if (!c) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null ctx");
    std::lock_guard<std::mutex> lock(c->mu);
    pandrs::GroupbyResult &r = c->gb;
    if (!r.valid) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no groupby result retained in this context");
    if (r.partials) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "context holds partials; use pandrs_hip_partials_split");
    HIP_TRY(hipSetDevice(c->device));
    const size_t g = (size_t)r.n_groups;
    if (g == 0) return PANDRS_HIP_OK;
    hipMemcpyKind kind = mem_space == PANDRS_HIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    for (int k = 0; k < r.n_keys; k++) {
        if (out_keys && out_keys[k])
            HIP_TRY(hipMemcpyAsync(out_keys[k], r.keys + (size_t)k * r.cap, g * 8, kind, c->stream));
        if (out_key_null && out_key_null[k])
            HIP_TRY(hipMemcpyAsync(out_key_null[k], r.key_null + (size_t)k * r.cap, g, kind, c->stream));
    }
    for (int a = 0; a < r.n_aggs; a++)
        if (out_aggs && out_aggs[a])
            HIP_TRY(hipMemcpyAsync(out_aggs[a], r.aggs + (size_t)a * r.cap, g * 8, kind, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_groupby_fetch"); }

int32_t pandrs_hip_groupby_indices(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *keys,
                                   int32_t n_keys, int64_t n_rows, int64_t *out_n_groups) try {
    ST_TRY(pandrs::below_threshold(n_rows));
    return pandrs::groupby_indices_entry(ctx, mem_space, keys, n_keys, n_rows, out_n_groups);
} catch (...) { return pandrs::on_exception("pandrs_hip_groupby_indices"); }

int32_t pandrs_hip_groupby_indices_fetch(pandrs_hip_ctx *c, int32_t mem_space, uint64_t *const *out_keys,
                                         uint8_t *const *out_key_null, int64_t *out_offsets, int64_t *out_rows) try {
    if (!c) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null ctx");
    std::lock_guard<std::mutex> lock(c->mu);
    pandrs::GroupsResult &r = c->gr;
    if (!r.valid) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no group index retained in this context");
    HIP_TRY(hipSetDevice(c->device));
    const size_t g = (size_t)r.n_groups, n = (size_t)r.n_rows;
    hipMemcpyKind kind = mem_space == PANDRS_HIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    for (int k = 0; k < r.n_keys && g > 0; k++) {
        if (out_keys && out_keys[k])
            HIP_TRY(hipMemcpyAsync(out_keys[k], r.keys + (size_t)k * r.cap, g * 8, kind, c->stream));
        if (out_key_null && out_key_null[k])
            HIP_TRY(hipMemcpyAsync(out_key_null[k], r.key_null + (size_t)k * r.cap, g, kind, c->stream));
    }
    if (out_offsets) HIP_TRY(hipMemcpyAsync(out_offsets, r.offsets, (g + 1) * 8, kind, c->stream));
    if (out_rows && n > 0) HIP_TRY(hipMemcpyAsync(out_rows, r.rows, n * 8, kind, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_groupby_indices_fetch"); }

int32_t pandrs_hip_shuffle_split(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *key,
                                 const pandrs_hip_column *payload, int32_t n_payload, int64_t n_rows,
                                 int32_t n_ranks, int32_t drop_null_keys, int64_t *out_counts, int64_t *out_n_rows) try {
    return pandrs::shuffle_split_entry(ctx, mem_space, key, payload, n_payload, n_rows, n_ranks, drop_null_keys,
                                       out_counts, out_n_rows);
} catch (...) { return pandrs::on_exception("pandrs_hip_shuffle_split"); }

int32_t pandrs_hip_shuffle_fetch(pandrs_hip_ctx *c, int32_t mem_space, uint64_t *out_cells, uint8_t *out_key_null,
                                 uint64_t *const *out_payload, uint8_t *const *out_payload_null) try {
    if (!c) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null ctx");
    std::lock_guard<std::mutex> lock(c->mu);
    pandrs::ShuffleResult &r = c->sh;
    if (!r.valid) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no shuffle result retained in this context");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)r.n_rows;
    if (n == 0) return PANDRS_HIP_OK;
    hipMemcpyKind kind = mem_space == PANDRS_HIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (out_cells) HIP_TRY(hipMemcpyAsync(out_cells, r.cells, n * 8, kind, c->stream));
    if (out_key_null) HIP_TRY(hipMemcpyAsync(out_key_null, r.key_null, n, kind, c->stream));
    for (int p = 0; p < r.n_payload; p++) {
        if (out_payload && out_payload[p]) HIP_TRY(hipMemcpyAsync(out_payload[p], r.pay[p], n * 8, kind, c->stream));
        if (out_payload_null && out_payload_null[p] && r.pay_null[p])
            HIP_TRY(hipMemcpyAsync(out_payload_null[p], r.pay_null[p], n, kind, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_shuffle_fetch"); }

int32_t pandrs_hip_key_hash_cells(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *keys,
                                  int32_t n_keys, int64_t n_rows, uint64_t *out_cells) try {
    return pandrs::key_hash_cells_entry(ctx, mem_space, keys, n_keys, n_rows, out_cells);
} catch (...) { return pandrs::on_exception("pandrs_hip_key_hash_cells"); }

int32_t pandrs_hip_bytes_to_bitmap(pandrs_hip_ctx *ctx, int32_t mem_space, const uint8_t *bytes, int64_t n,
                                   uint8_t *out_bitmap) try {
    return pandrs::bytes_to_bitmap_entry(ctx, mem_space, bytes, n, out_bitmap);
} catch (...) { return pandrs::on_exception("pandrs_hip_bytes_to_bitmap"); }

int32_t pandrs_hip_join_indices(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *left_key,
                                int64_t n_left, const pandrs_hip_column *right_key, int64_t n_right,
                                int32_t how, int64_t *out_n_rows) try {
    ST_TRY(pandrs::below_threshold(n_left > n_right ? n_left : n_right));
    return pandrs::join_entry(ctx, mem_space, left_key, n_left, right_key, n_right, how, out_n_rows);
} catch (...) { return pandrs::on_exception("pandrs_hip_join_indices"); }

int32_t pandrs_hip_join_fetch(pandrs_hip_ctx *c, int32_t mem_space, int64_t *out_left_idx,
                              int64_t *out_right_idx) try {
    if (!c) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null ctx");
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->jn.valid) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "no join result retained in this context");
    HIP_TRY(hipSetDevice(c->device));
    size_t n = (size_t)c->jn.n_rows;
    if (n == 0) return PANDRS_HIP_OK;
    hipMemcpyKind kind = mem_space == PANDRS_HIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (out_left_idx) HIP_TRY(hipMemcpyAsync(out_left_idx, c->jn.left_idx, n * 8, kind, c->stream));
    if (out_right_idx) HIP_TRY(hipMemcpyAsync(out_right_idx, c->jn.right_idx, n * 8, kind, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_join_fetch"); }

int32_t pandrs_hip_gather_i64(pandrs_hip_ctx *ctx, int32_t mem_space, const int64_t *src,
                              const uint8_t *m, const int64_t *idx, int64_t n, int64_t fill, int64_t *out) try {
    return pandrs::gather_entry(ctx, mem_space, 0, src, m, idx, n, (uint64_t)fill, out);
} catch (...) { return pandrs::on_exception("pandrs_hip_gather_i64"); }
int32_t pandrs_hip_gather_f64(pandrs_hip_ctx *ctx, int32_t mem_space, const double *src,
                              const uint8_t *m, const int64_t *idx, int64_t n, double fill, double *out) try {
    uint64_t b;
    std::memcpy(&b, &fill, 8);
    return pandrs::gather_entry(ctx, mem_space, 0, src, m, idx, n, b, out);
} catch (...) { return pandrs::on_exception("pandrs_hip_gather_f64"); }
int32_t pandrs_hip_gather_u32(pandrs_hip_ctx *ctx, int32_t mem_space, const uint32_t *src,
                              const uint8_t *m, const int64_t *idx, int64_t n, uint32_t fill, uint32_t *out) try {
    return pandrs::gather_entry(ctx, mem_space, 1, src, m, idx, n, fill, out);
} catch (...) { return pandrs::on_exception("pandrs_hip_gather_u32"); }
int32_t pandrs_hip_gather_bool(pandrs_hip_ctx *ctx, int32_t mem_space, const uint8_t *src_bits,
                               const uint8_t *m, const int64_t *idx, int64_t n, uint8_t fill, uint8_t *out) try {
    return pandrs::gather_entry(ctx, mem_space, 2, src_bits, m, idx, n, fill, out);
} catch (...) { return pandrs::on_exception("pandrs_hip_gather_bool"); }

int32_t pandrs_hip_join_groupby_sum(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *lk,
                                    const pandrs_hip_column *lv, int64_t nl, const pandrs_hip_column *rk,
                                    const pandrs_hip_column *rg, int64_t nr, int64_t *out_n_groups) try {
    ST_TRY(pandrs::below_threshold(nl > nr ? nl : nr));
    return pandrs::join_groupby_sum_entry(ctx, mem_space, lk, lv, nl, rk, rg, nr, out_n_groups);
} catch (...) { return pandrs::on_exception("pandrs_hip_join_groupby_sum"); }

int32_t pandrs_hip_reduce_column(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *col,
                                 int64_t n, double out[4], int64_t *out_count) try {
    ST_TRY(pandrs::below_threshold(n));
    return pandrs::reduce_entry(ctx, mem_space, col, n, out, out_count);
} catch (...) { return pandrs::on_exception("pandrs_hip_reduce_column"); }

int32_t pandrs_hip_gather_column(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *src, int64_t n_src,
                                 const int64_t *idx, int64_t n, uint64_t fill_bits, void *out) try {
    return pandrs::gather_column_entry(ctx, mem_space, src, n_src, idx, n, fill_bits, out);
} catch (...) { return pandrs::on_exception("pandrs_hip_gather_column"); }

int32_t pandrs_hip_join_gather(pandrs_hip_ctx *ctx, int32_t src_mem_space, const pandrs_hip_column *src, int64_t n_src,
                               int32_t side, uint64_t fill_bits, int32_t out_mem_space, void *out) try {
    return pandrs::join_gather_entry(ctx, src_mem_space, src, n_src, side, fill_bits, out_mem_space, out);
} catch (...) { return pandrs::on_exception("pandrs_hip_join_gather"); }

int32_t pandrs_hip_join_gather_key(pandrs_hip_ctx *ctx, int32_t src_mem_space, const pandrs_hip_column *left_key, int64_t n_left,
                                   const pandrs_hip_column *right_key, int64_t n_right, uint64_t fill_bits, int32_t out_mem_space, void *out) try {
    if (!right_key) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join_gather_key: null right key column");
    return pandrs::join_gather_entry(ctx, src_mem_space, left_key, n_left, 0, fill_bits, out_mem_space, out, right_key, n_right);
} catch (...) { return pandrs::on_exception("pandrs_hip_join_gather_key"); }

int32_t pandrs_hip_reduce_moments(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *col, int64_t n,
                                  double *out_sum, double *out_sum_sq, int64_t *out_count) try {
    if (!out_sum || !out_sum_sq || !out_count) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "reduce_moments: bad arguments");
    ST_TRY(pandrs::below_threshold(n));
    double o[4];
    int32_t st = pandrs::reduce_entry(ctx, mem_space, col, n, o, out_count, out_sum_sq);
    if (st) return st;
    *out_sum = o[0];
    return 0;
} catch (...) { return pandrs::on_exception("pandrs_hip_reduce_moments"); }

int32_t pandrs_hip_reduce_stats(pandrs_hip_ctx *ctx, int32_t mem_space, const pandrs_hip_column *col, int64_t n,
                                pandrs_hip_column_stats *out) try {
    ST_TRY(pandrs::below_threshold(n));
    return pandrs::reduce_stats_entry(ctx, mem_space, col, n, out);
} catch (...) { return pandrs::on_exception("pandrs_hip_reduce_stats"); }

}  // extern "C"
