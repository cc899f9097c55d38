// dist.hip — the multi-GPU exchange INSIDE the library (BASELINE.json north_star: "RCCL all-to-all over
// xGMI ... called from Rust host code through a thin extern "C" FFI").  One process per GPU; every rank calls the same
// entry point with its own row range:
//
//   groupby   local partial aggregation -> owner split (packed records, rank-contiguous) -> count exchange (one
//             ncclAllGather) -> ONE grouped ncclSend / ncclRecv all-to-all of the records on the context's stream
//             -> merge of what this rank owns.
//   join      the build side is all-gathered (shards padded with NULL-key rows, which an inner join skips by its own
//             semantics, join.rs:107-142), the probe side never leaves its GPU, local fused join -> groupby-sum, then
//             the groupby exchange above on the <= G partial sums.
//
// The reference has no counterpart (src/gpu/multi_gpu.rs:326-360 splits rows on the host; src/distributed is an
// in-process DataFusion wrapper).  RCCL is opened at run time (dlopen of librccl.so.1: the copy already mapped by the
// host process — PyTorch-ROCm bundles one — or /opt/rocm's), so single-GPU users need no RCCL at all.
#include "common.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <vector>

namespace pandrs {

// ---- the few RCCL entry points used, bound at first use --------------------------------------------------------
namespace {
constexpr int RCCL_UNIQUE_ID_BYTES = 128;             // NCCL_UNIQUE_ID_BYTES (rccl.h:40)
struct RcclUniqueId { char internal[RCCL_UNIQUE_ID_BYTES]; };
enum { RCCL_INT8 = 0, RCCL_INT64 = 4, RCCL_MAX = 2 };  // ncclDataType_t / ncclRedOp_t values (rccl.h)
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
Rccl &rccl() { static Rccl r; return r; }
std::mutex &rccl_mu() { static std::mutex m; return m; }

int32_t rccl_load() {
    std::lock_guard<std::mutex> lock(rccl_mu());
    Rccl &r = rccl();
    if (r.handle) return 0;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);     // the host process's copy, if it has one
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "RCCL is not available: %s", dlerror());
    auto sym = [&](const char *name, auto &fn) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name));
        return fn != nullptr;
    };
    if (!(sym("ncclGetUniqueId", r.GetUniqueId) && sym("ncclCommInitRank", r.CommInitRank) && sym("ncclCommDestroy", r.CommDestroy) &&
          sym("ncclGetErrorString", r.GetErrorString) && sym("ncclAllGather", r.AllGather) && sym("ncclAllReduce", r.AllReduce) &&
          sym("ncclSend", r.Send) && sym("ncclRecv", r.Recv) && sym("ncclGroupStart", r.GroupStart) && sym("ncclGroupEnd", r.GroupEnd)))
        return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "RCCL lacks a required entry point");
    r.handle = h;
    return 0;
}
#define RCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        int e__ = (expr);                                                                                     \
        if (e__ != 0) return ::pandrs::fail(PANDRS_HIP_ERR_COMPUTATION, "%s failed: %s", #expr, ::pandrs::rccl().GetErrorString(e__)); \
    } while (0)
}  // namespace

// copies the retained groupby result (cells, null bytes, one aggregate row) out of the context: join -> groupby hand-over
int32_t export_result_columns(pandrs_hip_ctx *c, uint64_t *cells, uint8_t *nulls, double *agg0, int64_t *out_n);

}  // namespace pandrs

struct pandrs_hip_comm {
    void *nccl = nullptr;
    int rank = 0, world = 1;
    bool owned = false;
    pandrs::Arena send, recv, small;         // exchange buffers: grown, never shrunk
    std::vector<int64_t> counts;             // world x world matrix of the last count exchange
};

namespace pandrs {

// all-reduce (max) of a few host integers: layout agreement before planning (a rank-local decision such as "this
// column has a null mask" must not change the partial-record width on one rank only)
static int32_t agree_max(pandrs_hip_ctx *c, pandrs_hip_comm *cm, int64_t *vals, int n) {
    ST_TRY(cm->small.ensure(4096 + (size_t)n * 16, c->stream));
    int64_t *d = cm->small.take<int64_t>(n);
    HIP_TRY(hipMemcpyAsync(d, vals, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(rccl().AllReduce(d, d, (size_t)n, RCCL_INT64, RCCL_MAX, cm->nccl, c->stream));
    HIP_TRY(hipMemcpyAsync(vals, d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// The exchange proper.  `send`: device records, rank-contiguous by owner, `send_counts[p]` records of W words for rank p.
// -> *out_recv (device, cm->recv arena), *out_n_recv.
static int32_t exchange_records(pandrs_hip_ctx *c, pandrs_hip_comm *cm, const uint64_t *send, const int64_t *send_counts,
                                size_t W, uint64_t **out_recv, int64_t *out_n_recv) {
    const int world = cm->world, me = cm->rank;
    // 1. counts: every rank learns the whole world x world matrix with one all-gather
    ST_TRY(cm->small.ensure(4096 + (size_t)world * (size_t)world * 8 + (size_t)world * 8, c->stream));
    int64_t *d_mine = cm->small.take<int64_t>(world);
    int64_t *d_all = cm->small.take<int64_t>((size_t)world * world);
    HIP_TRY(hipMemcpyAsync(d_mine, send_counts, (size_t)world * 8, hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(rccl().AllGather(d_mine, d_all, (size_t)world, RCCL_INT64, cm->nccl, c->stream));
    cm->counts.assign((size_t)world * world, 0);
    HIP_TRY(hipMemcpyAsync(cm->counts.data(), d_all, (size_t)world * world * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int64_t n_recv = 0;
    for (int r = 0; r < world; r++) {
        const int64_t v = cm->counts[(size_t)r * world + me];          // what rank r sends to this rank
        if (v < 0) return fail(PANDRS_HIP_ERR_COMPUTATION, "negative record count from rank %d", r);
        n_recv += v;
    }
    for (int p = 0; p < world; p++)
        if (cm->counts[(size_t)me * world + p] != send_counts[p])
            return fail(PANDRS_HIP_ERR_COMPUTATION, "count exchange returned a different row for this rank");
    // 2. ONE grouped all-to-all of the packed records (every xGMI link busy at once; no ring)
    ST_TRY(cm->recv.ensure((size_t)std::max<int64_t>(n_recv, 1) * W * 8 + 4096, c->stream));
    uint64_t *recv = cm->recv.take<uint64_t>((size_t)std::max<int64_t>(n_recv, 1) * W);
    if (!recv) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "exchange buffer too small");
    RCCL_TRY(rccl().GroupStart());
    int64_t soff = 0, roff = 0;
    int rc = 0;
    for (int p = 0; p < world && rc == 0; p++) {
        const int64_t ns = send_counts[p], nr = cm->counts[(size_t)p * world + me];
        if (ns > 0) rc = rccl().Send(send + (size_t)soff * W, (size_t)ns * W, RCCL_INT64, p, cm->nccl, c->stream);
        if (nr > 0 && rc == 0) rc = rccl().Recv(recv + (size_t)roff * W, (size_t)nr * W, RCCL_INT64, p, cm->nccl, c->stream);
        soff += ns; roff += nr;
    }
    const int rc_end = rccl().GroupEnd();
    if (rc != 0) return fail(PANDRS_HIP_ERR_COMPUTATION, "ncclSend / ncclRecv failed: %s", rccl().GetErrorString(rc));
    RCCL_TRY(rc_end);
    *out_recv = recv; *out_n_recv = n_recv;
    return 0;
}

static int32_t dist_groupby_impl(pandrs_hip_ctx *c, pandrs_hip_comm *cm, int32_t mem_space, const pandrs_hip_column *key,
                                 int64_t n_rows, const pandrs_hip_column *vals, int32_t n_vals, const pandrs_hip_agg_spec *aggs,
                                 int32_t n_aggs, int64_t *out_n_groups) {
    if (n_vals > 64) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_groupby_agg: more than 64 value columns");
    for (int a = 0; a < n_aggs; a++)
        if (aggs[a].op > PANDRS_HIP_AGG_COUNT)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "dist_groupby_agg exchanges partial states: Sum / Mean / Min / Max / Count only "
                                                         "(the row shuffle for the rest is pandrs_hip_shuffle_split + the host's all-to-all)");
    // 0. one layout on every rank: a value column has an nn state iff SOME rank passes a null mask for it
    int64_t flags[65] = {0};
    for (int i = 0; i < n_vals; i++) flags[i] = vals[i].null_mask ? 1 : 0;
    flags[n_vals] = n_rows > 0 ? 1 : 0;
    ST_TRY(agree_max(c, cm, flags, n_vals + 1));
    std::vector<pandrs_hip_column> v2(vals, vals + n_vals);
    std::vector<uint8_t> has_nulls((size_t)std::max(n_vals, 1), 0);
    std::vector<int32_t> dtypes((size_t)std::max(n_vals, 1), 0);
    std::vector<uint8_t> zero_host;
    uint8_t *zero_dev = nullptr;
    for (int i = 0; i < n_vals; i++) {
        has_nulls[i] = (uint8_t)flags[i];
        dtypes[i] = vals[i].dtype;
        if (flags[i] && !vals[i].null_mask) {           // another rank has nulls here: an all-valid bitmap keeps the plan identical
            const size_t nb = (size_t)(n_rows + 7) / 8 + 8;
            if (mem_space == PANDRS_HIP_MEM_HOST) {
                if (zero_host.size() < nb) zero_host.assign(nb, 0);
                v2[i].null_mask = zero_host.data();
            } else {
                if (!zero_dev) {
                    ST_TRY(cm->send.ensure(nb + 4096, c->stream));      // (the send arena is re-sized below, after the local pass)
                    HIP_TRY(hipMalloc((void **)&zero_dev, nb));
                    HIP_TRY(hipMemsetAsync(zero_dev, 0, nb, c->stream));
                }
                v2[i].null_mask = zero_dev;
            }
        }
    }
    struct Free { uint8_t *p; hipStream_t s; ~Free() { if (p) { (void)hipStreamSynchronize(s); (void)hipFree(p); } } } free_zero{zero_dev, c->stream};
    // 1. local partial aggregation (states retained in the context)
    int64_t ng = 0;
    int32_t n_state = 0;
    ST_TRY(groupby_entry(c, mem_space, key, 1, n_rows, v2.data(), n_vals, aggs, n_aggs, /*partials=*/true, &ng, &n_state));
    const size_t W = 2 + (size_t)n_state;
    // 2. owner split: packed records, rank-contiguous
    ST_TRY(cm->send.ensure((size_t)std::max<int64_t>(ng, 1) * W * 8 + 4096, c->stream));
    uint64_t *send = cm->send.take<uint64_t>((size_t)std::max<int64_t>(ng, 1) * W);
    std::vector<int64_t> send_counts((size_t)cm->world, 0);
    ST_TRY(partials_split_entry(c, PANDRS_HIP_MEM_DEVICE, cm->world, send, send_counts.data()));
    // 3. count exchange + ONE all-to-all
    uint64_t *recv = nullptr;
    int64_t n_recv = 0;
    ST_TRY(exchange_records(c, cm, send, send_counts.data(), W, &recv, &n_recv));
    // 4. merge what this rank owns (cardinality bounded by the records received: no sampling pass)
    const int64_t hint_saved = c->opt.groups_hint;
    c->opt.groups_hint = std::max<int64_t>(n_recv, 1);
    const int32_t st = groupby_merge_entry(c, PANDRS_HIP_MEM_DEVICE, key->dtype, recv, n_recv, dtypes.data(), n_vals, has_nulls.data(),
                                           aggs, n_aggs, out_n_groups);
    c->opt.groups_hint = hint_saved;
    return st;
}

}  // namespace pandrs

using pandrs::fail;

extern "C" {

int32_t pandrs_hip_comm_unique_id(char out_id[128]) {
    if (!out_id) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null out_id");
    ST_TRY(pandrs::rccl_load());
    pandrs::RcclUniqueId id;
    RCCL_TRY(pandrs::rccl().GetUniqueId(&id));
    std::memcpy(out_id, id.internal, 128);
    return PANDRS_HIP_OK;
}

int32_t pandrs_hip_comm_init(pandrs_hip_ctx *ctx, const char id[128], int32_t rank, int32_t world, pandrs_hip_comm **out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "comm_init: bad arguments");
    ST_TRY(pandrs::rccl_load());
    HIP_TRY(hipSetDevice(ctx->device));
    pandrs::RcclUniqueId uid;
    std::memcpy(uid.internal, id, 128);
    auto *cm = new pandrs_hip_comm();
    int e = pandrs::rccl().CommInitRank(&cm->nccl, world, uid, rank);
    if (e != 0) { delete cm; return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "ncclCommInitRank failed: %s", pandrs::rccl().GetErrorString(e)); }
    cm->rank = rank; cm->world = world; cm->owned = true;
    *out = cm;
    return PANDRS_HIP_OK;
}

int32_t pandrs_hip_comm_adopt(void *nccl_comm, int32_t rank, int32_t world, pandrs_hip_comm **out) {
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "comm_adopt: bad arguments");
    ST_TRY(pandrs::rccl_load());
    auto *cm = new pandrs_hip_comm();
    cm->nccl = nccl_comm; cm->rank = rank; cm->world = world; cm->owned = false;
    *out = cm;
    return PANDRS_HIP_OK;
}

int32_t pandrs_hip_comm_destroy(pandrs_hip_comm *cm) {
    if (!cm) return PANDRS_HIP_OK;
    cm->send.release(); cm->recv.release(); cm->small.release();
    if (cm->owned && cm->nccl) (void)pandrs::rccl().CommDestroy(cm->nccl);
    delete cm;
    return PANDRS_HIP_OK;
}

int32_t pandrs_hip_dist_groupby_agg(pandrs_hip_ctx *ctx, pandrs_hip_comm *comm, int32_t mem_space, const pandrs_hip_column *keys,
                                    int32_t n_keys, int64_t n_rows, const pandrs_hip_column *vals, int32_t n_vals,
                                    const pandrs_hip_agg_spec *aggs, int32_t n_aggs, int64_t *out_n_groups) {
    if (!ctx || !comm || !keys || !out_n_groups || n_rows < 0 || n_vals < 0 || n_aggs < 0 || (n_vals && !vals) || (n_aggs && !aggs))
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_groupby_agg: bad arguments");
    if (n_keys != 1)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "dist_groupby_agg takes one key column (composite keys: shuffle on pandrs_hip_key_hash_cells)");
    HIP_TRY(hipSetDevice(ctx->device));
    return pandrs::dist_groupby_impl(ctx, comm, mem_space, keys, n_rows, vals, n_vals, aggs, n_aggs, out_n_groups);
}

int32_t pandrs_hip_dist_join_groupby_sum(pandrs_hip_ctx *ctx, pandrs_hip_comm *cm, int32_t mem_space,
                                         const pandrs_hip_column *left_key, const pandrs_hip_column *left_val, int64_t n_left,
                                         const pandrs_hip_column *right_key, const pandrs_hip_column *right_group, int64_t n_right,
                                         int64_t *out_n_groups) {
    using namespace pandrs;
    if (!ctx || !cm || !left_key || !left_val || !right_key || !right_group || !out_n_groups || n_left < 0 || n_right < 0)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_join_groupby_sum: bad arguments");
    if (mem_space != PANDRS_HIP_MEM_DEVICE)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_join_groupby_sum takes device-resident shards (PANDRS_HIP_MEM_DEVICE)");
    if ((right_key->dtype != PANDRS_HIP_I64 && right_key->dtype != PANDRS_HIP_F64 && right_key->dtype != PANDRS_HIP_CELL64) ||
        (right_group->dtype != PANDRS_HIP_I64 && right_group->dtype != PANDRS_HIP_F64 && right_group->dtype != PANDRS_HIP_CELL64))
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "dist_join_groupby_sum: 8-byte build-side columns only");
    pandrs_hip_ctx *c = ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int world = cm->world;
    // 1. all-gather the build side; shards are padded to a common length with NULL-key rows
    int64_t info[3] = {n_right, right_key->null_mask ? 1 : 0, right_group->null_mask ? 1 : 0};
    int64_t mx[3] = {info[0], info[1], info[2]};
    ST_TRY(agree_max(c, cm, mx, 3));
    const int64_t n_pad = (mx[0] + 7) / 8 * 8, n_all = n_pad * world;
    // every rank needs its neighbours' true lengths to know whether padding exists anywhere
    int64_t mn[1] = {-n_right};
    ST_TRY(agree_max(c, cm, mn, 1));
    const bool padded = -mn[0] != n_pad;
    const bool key_mask = mx[1] != 0 || padded, grp_mask = mx[2] != 0;
    const size_t mb = (size_t)n_pad / 8;
    ST_TRY(cm->send.ensure(((size_t)n_pad * 16 + 2 * mb) + ((size_t)n_all * 16 + 2 * mb * world) + (1 << 16), c->stream));
    uint64_t *sk = cm->send.take<uint64_t>((size_t)std::max<int64_t>(n_pad, 1)), *sg = cm->send.take<uint64_t>((size_t)std::max<int64_t>(n_pad, 1));
    uint8_t *skm = cm->send.take<uint8_t>(mb + 16), *sgm = cm->send.take<uint8_t>(mb + 16);
    uint64_t *ak = cm->send.take<uint64_t>((size_t)std::max<int64_t>(n_all, 1)), *ag = cm->send.take<uint64_t>((size_t)std::max<int64_t>(n_all, 1));
    uint8_t *akm = cm->send.take<uint8_t>(mb * world + 16), *agm = cm->send.take<uint8_t>(mb * world + 16);
    if (!sk || !sg || !skm || !sgm || !ak || !ag || !akm || !agm) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "exchange buffer too small");
    HIP_TRY(hipMemsetAsync(sk, 0, (size_t)n_pad * 8, c->stream));
    HIP_TRY(hipMemsetAsync(sg, 0, (size_t)n_pad * 8, c->stream));
    HIP_TRY(hipMemcpyAsync(sk, right_key->data, (size_t)n_right * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(sg, right_group->data, (size_t)n_right * 8, hipMemcpyDeviceToDevice, c->stream));
    if (key_mask) {
        // host-built bitmap: the caller's bits for [0, n_right), 1 (null) for the padding
        std::vector<uint8_t> m(mb + 1, 0);
        if (right_key->null_mask) {
            HIP_TRY(hipMemcpyAsync(m.data(), right_key->null_mask, (size_t)(n_right + 7) / 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (n_right % 8) m[(size_t)n_right / 8] &= (uint8_t)((1u << (n_right % 8)) - 1);
        }
        for (int64_t i = n_right; i < n_pad; i++) m[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7));
        HIP_TRY(hipMemcpyAsync(skm, m.data(), mb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (grp_mask) {
        HIP_TRY(hipMemsetAsync(sgm, 0, mb, c->stream));
        if (right_group->null_mask) HIP_TRY(hipMemcpyAsync(sgm, right_group->null_mask, (size_t)(n_right + 7) / 8, hipMemcpyDeviceToDevice, c->stream));
    }
    RCCL_TRY(rccl().AllGather(sk, ak, (size_t)n_pad, RCCL_INT64, cm->nccl, c->stream));
    RCCL_TRY(rccl().AllGather(sg, ag, (size_t)n_pad, RCCL_INT64, cm->nccl, c->stream));
    if (key_mask) RCCL_TRY(rccl().AllGather(skm, akm, mb, RCCL_INT8, cm->nccl, c->stream));
    if (grp_mask) RCCL_TRY(rccl().AllGather(sgm, agm, mb, RCCL_INT8, cm->nccl, c->stream));
    // 2. local fused join -> groupby-sum against the whole build side
    pandrs_hip_column rk{ak, key_mask ? akm : nullptr, right_key->dtype, 0}, rg{ag, grp_mask ? agm : nullptr, right_group->dtype, 0};
    int64_t g_local = 0;
    ST_TRY(join_groupby_sum_entry(c, PANDRS_HIP_MEM_DEVICE, left_key, left_val, n_left, &rk, &rg, n_all, &g_local));
    // 3. the <= G local sums go through the groupby exchange, keyed on g's cell
    ST_TRY(cm->recv.ensure((size_t)std::max<int64_t>(g_local, 1) * 17 + ((size_t)g_local + 7) / 8 + (1 << 16), c->stream));
    uint64_t *cells = cm->recv.take<uint64_t>((size_t)std::max<int64_t>(g_local, 1));
    double *sums = cm->recv.take<double>((size_t)std::max<int64_t>(g_local, 1));
    uint8_t *nb = cm->recv.take<uint8_t>((size_t)std::max<int64_t>(g_local, 1)), *nbits = cm->recv.take<uint8_t>(((size_t)g_local + 7) / 8 + 16);
    int64_t got = 0;
    ST_TRY(export_result_columns(c, cells, nb, sums, &got));
    ST_TRY(bytes_to_bitmap_entry(c, PANDRS_HIP_MEM_DEVICE, nb, got, nbits));
    pandrs_hip_column gk{cells, nbits, PANDRS_HIP_CELL64, 0}, gv{sums, nullptr, PANDRS_HIP_F64, 0};
    const pandrs_hip_agg_spec sum_spec{0, PANDRS_HIP_AGG_SUM};
    // (exchange_records re-takes cm->recv: the columns above must outlive the local partial pass only, which reads them
    // before the exchange allocates — keep them in their own arena to be safe)
    pandrs::Arena keep;
    std::swap(keep, cm->recv);
    const int32_t st = dist_groupby_impl(c, cm, PANDRS_HIP_MEM_DEVICE, &gk, got, &gv, 1, &sum_spec, 1, out_n_groups);
    (void)hipStreamSynchronize(c->stream);
    keep.release();
    return st;
}

}  // extern "C"
