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
#include <chrono>
#include <cstdlib>

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
    int (*CommAbort)(void *) = nullptr;            // optional: releases peers blocked in a collective this rank cannot take part in
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
    r.CommAbort = reinterpret_cast<int (*)(void *)>(dlsym(h, "ncclCommAbort"));
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
    void *nccl = nullptr;                    // RCCL transport (the default)
    pandrs_hip_transport host{};             // host-callback transport (pandrs_hip_comm_adopt_transport); used when nccl == nullptr
    int rank = 0, world = 1;
    bool owned = false;
    // exchange buffers: grown, never shrunk, never released between calls (no hipMalloc in a steady-state step).
    // send / recv: the travelling columns; small: counts and flags of ONE collective; crow: this rank's count row (device); zeros: an all-valid null bitmap (written once per growth);
    // stage: the join's exported (g, sum) columns, which must outlive the nested groupby exchange
    pandrs::Arena send, recv, small, zeros, stage, crow;
    size_t zeros_valid = 0;                  // bytes of `zeros` known to be zero
    // the null-mask layout the ranks agreed on in the last dist_groupby call with the same column count (bit i: value column i carries
    // a non-null count): the next call ASSUMES it and verifies on the count exchange — no collective of its own in the steady state
    int64_t agreed_flags = 0;
    int32_t agreed_n_vals = -1;
    bool aborted = false;                    // a rank-local failure after the count exchange aborted the communicator
    std::vector<int64_t> counts;             // world x (world + 1) matrix of the last count exchange (last column: status)
    std::vector<uint8_t> hsend, hrecv;       // host staging of the callback transport
};

namespace pandrs {

// ---- the collectives, behind one small table: RCCL on device buffers (default), or host callbacks (tests, other fabrics)
// all-reduce (max) of a few host integers: layout agreement before planning (a rank-local decision such as "this
// column has a null mask" must not change the partial-record width on one rank only)
static int32_t agree_max(pandrs_hip_ctx *c, pandrs_hip_comm *cm, int64_t *vals, int n) {
    if (!cm->nccl) {
        const int32_t st = cm->host.all_reduce_max_i64(cm->host.user, vals, n);
        return st ? fail(PANDRS_HIP_ERR_COMPUTATION, "transport all_reduce_max_i64 failed (%d)", st) : 0;
    }
    ST_TRY(cm->small.ensure(4096 + (size_t)n * 16, c->stream));
    int64_t *d = cm->small.take<int64_t>(n);
    HIP_TRY(hipMemcpyAsync(d, vals, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(rccl().AllReduce(d, d, (size_t)n, RCCL_INT64, RCCL_MAX, cm->nccl, c->stream));
    HIP_TRY(hipMemcpyAsync(vals, d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// all-gather of `bytes` bytes per rank between DEVICE buffers (recv: world * bytes, rank-major), stream-ordered
static int32_t all_gather_dev(pandrs_hip_ctx *c, pandrs_hip_comm *cm, const void *send, void *recv, size_t bytes) {
    if (cm->nccl) {
        if (bytes % 8 == 0) RCCL_TRY(rccl().AllGather(send, recv, bytes / 8, RCCL_INT64, cm->nccl, c->stream));
        else RCCL_TRY(rccl().AllGather(send, recv, bytes, RCCL_INT8, cm->nccl, c->stream));
        return 0;
    }
    cm->hsend.resize(bytes + 8); cm->hrecv.resize(bytes * (size_t)cm->world + 8);
    HIP_TRY(hipMemcpyAsync(cm->hsend.data(), send, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int32_t st = cm->host.all_gather(cm->host.user, cm->hsend.data(), cm->hrecv.data(), (int64_t)bytes);
    if (st) return fail(PANDRS_HIP_ERR_COMPUTATION, "transport all_gather failed (%d)", st);
    HIP_TRY(hipMemcpyAsync(recv, cm->hrecv.data(), bytes * (size_t)cm->world, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));            // hrecv is reused by the next collective
    return 0;
}

// ---- exchange of row-aligned COLUMNS (partial records as columns; the general row shuffle below)
struct ExCol { const void *send; size_t elem; void *recv; };

// A rank that cannot take part in a collective its peers are about to enter (or are in) aborts the communicator: the peers' calls
// return an error instead of blocking for ever.  The communicator is unusable afterwards.
static void abort_comm(pandrs_hip_comm *cm) {
    if (cm->nccl && cm->world > 1 && rccl().CommAbort) { (void)rccl().CommAbort(cm->nccl); cm->nccl = nullptr; cm->owned = false; }
    cm->aborted = true;
}

// The count exchange: ONE all-gather of a row of world + 1 + n_extra int64 per rank — rows for every peer, this rank's status so
// far, and `extra` words every rank wants the others to see (dist_groupby's null-mask layout).  cm->counts holds the gathered
// matrix (row stride world + 1 + n_extra).  `local_status` rides on the exchange: when ANY rank reports a failure every rank
// returns an error together instead of leaving its peers blocked in the all-to-all.
// `d_count_row` (optional): this rank's counts as a DEVICE row (the split left them there; the status and extra words are written
// here): the all-gather starts from it and `send_counts` is an OUTPUT — the host first learns its own counts from the gathered
// matrix, one synchronisation for the whole count exchange.
static int32_t exchange_counts(pandrs_hip_ctx *c, pandrs_hip_comm *cm, int64_t *send_counts, int32_t local_status,
                               const int64_t *extra, int n_extra, int64_t *d_count_row = nullptr) {
    const int world = cm->world, me = cm->rank, row = world + 1 + n_extra;
    std::vector<int64_t> mine((size_t)row, 0);
    if (!d_count_row) for (int p = 0; p < world; p++) mine[p] = local_status ? 0 : send_counts[p];
    mine[world] = local_status;
    for (int e = 0; e < n_extra; e++) mine[world + 1 + e] = extra[e];
    cm->counts.assign((size_t)world * row, 0);
    if (d_count_row) {
        if (local_status) HIP_TRY(hipMemsetAsync(d_count_row, 0, (size_t)world * 8, c->stream));
        HIP_TRY(hipMemcpyAsync(d_count_row + world, &mine[world], (size_t)(1 + n_extra) * 8, hipMemcpyHostToDevice, c->stream));   // (pageable source: copied before the call returns)
    }
    if (cm->nccl) {
        if (const int32_t st = cm->small.ensure(4096 + (size_t)world * row * 8 + (size_t)row * 8, c->stream)) { abort_comm(cm); return st; }   // (a few KB, grown once)
        int64_t *d_mine = d_count_row ? d_count_row : cm->small.take<int64_t>(row);
        int64_t *d_all = cm->small.take<int64_t>((size_t)world * row);
        if (!d_count_row) HIP_TRY(hipMemcpyAsync(d_mine, mine.data(), (size_t)row * 8, hipMemcpyHostToDevice, c->stream));
        RCCL_TRY(rccl().AllGather(d_mine, d_all, (size_t)row, RCCL_INT64, cm->nccl, c->stream));
        HIP_TRY(hipMemcpyAsync(cm->counts.data(), d_all, (size_t)world * row * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        if (d_count_row) {
            HIP_TRY(hipMemcpyAsync(mine.data(), d_count_row, (size_t)row * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        const int32_t st = cm->host.all_gather(cm->host.user, mine.data(), cm->counts.data(), (int64_t)row * 8);
        if (st) return fail(PANDRS_HIP_ERR_COMPUTATION, "transport all_gather failed (%d)", st);
    }
    if (d_count_row) for (int p = 0; p < world; p++) send_counts[p] = cm->counts[(size_t)me * row + p];
    for (int r = 0; r < world; r++)
        if (cm->counts[(size_t)r * row + world] != 0) {
            if (r == me && local_status) return local_status;
            return fail(PANDRS_HIP_ERR_COMPUTATION, "rank %d failed in its local phase (status %lld): the exchange is abandoned on every rank",
                        r, (long long)cm->counts[(size_t)r * row + world]);
        }
    return 0;
}

// The payload: row-aligned device COLUMNS, rank-contiguous by the split `send_counts`, in ONE grouped send / recv.  recv pointers
// are filled in (cm->recv arena).  `row`: the row stride of cm->counts as the count exchange left it.  A failure here is rank-local
// and comes AFTER the ranks agreed to exchange (the receive buffer cannot be allocated): the peers are about to block in the
// all-to-all, so the communicator is aborted (ncclCommAbort) — they return an error instead of waiting for ever — and is unusable
// afterwards (cm->aborted).
static int32_t exchange_payload(pandrs_hip_ctx *c, pandrs_hip_comm *cm, const int64_t *send_counts, std::vector<ExCol> &cols, int row, int64_t *out_n_recv) {
    const int world = cm->world, me = cm->rank;
    int64_t n_recv = 0;
    for (int r = 0; r < world; r++) n_recv += cm->counts[(size_t)r * row + me];
    size_t need = 4096;
    for (auto &col : cols) need += Arena::padded((size_t)std::max<int64_t>(n_recv, 1) * col.elem + 256);
    int32_t st = cm->recv.ensure(need, c->stream);
    for (auto &col : cols) {
        if (st) break;
        col.recv = cm->recv.take<uint8_t>((size_t)std::max<int64_t>(n_recv, 1) * col.elem + 256);
        if (!col.recv) st = fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "exchange buffer too small");
    }
    if (st) { abort_comm(cm); return st; }
    if (cm->nccl) {
        RCCL_TRY(rccl().GroupStart());
        int rc = 0;
        for (auto &col : cols) {
            int64_t soff = 0, roff = 0;
            for (int p = 0; p < world && rc == 0; p++) {
                const int64_t ns = send_counts[p], nr = cm->counts[(size_t)p * row + me];
                if (ns > 0) rc = rccl().Send((const char *)col.send + (size_t)soff * col.elem, (size_t)ns * col.elem, RCCL_INT8, p, cm->nccl, c->stream);
                if (nr > 0 && rc == 0) rc = rccl().Recv((char *)col.recv + (size_t)roff * col.elem, (size_t)nr * col.elem, RCCL_INT8, p, cm->nccl, c->stream);
                soff += ns; roff += nr;
            }
        }
        const int rc_end = rccl().GroupEnd();
        if (rc != 0) return fail(PANDRS_HIP_ERR_COMPUTATION, "ncclSend / ncclRecv failed: %s", rccl().GetErrorString(rc));
        RCCL_TRY(rc_end);
    } else {
        std::vector<int64_t> sb((size_t)world), so((size_t)world), rb((size_t)world), ro((size_t)world);
        for (auto &col : cols) {
            int64_t soff = 0, roff = 0;
            for (int p = 0; p < world; p++) {
                sb[p] = send_counts[p] * (int64_t)col.elem; so[p] = soff; soff += sb[p];
                rb[p] = cm->counts[(size_t)p * row + me] * (int64_t)col.elem; ro[p] = roff; roff += rb[p];
            }
            cm->hsend.resize((size_t)soff + 8); cm->hrecv.resize((size_t)roff + 8);
            if (soff) HIP_TRY(hipMemcpyAsync(cm->hsend.data(), col.send, (size_t)soff, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const int32_t st2 = cm->host.all_to_all_v(cm->host.user, cm->hsend.data(), sb.data(), so.data(), cm->hrecv.data(), rb.data(), ro.data());
            if (st2) return fail(PANDRS_HIP_ERR_COMPUTATION, "transport all_to_all_v failed (%d)", st2);
            if (roff) HIP_TRY(hipMemcpyAsync(col.recv, cm->hrecv.data(), (size_t)roff, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    *out_n_recv = n_recv;
    return 0;
}

// both steps, for callers without extra words
static int32_t exchange_columns(pandrs_hip_ctx *c, pandrs_hip_comm *cm, int64_t *send_counts, std::vector<ExCol> &cols,
                                int32_t local_status, int64_t *out_n_recv, int64_t *d_count_row = nullptr) {
    ST_TRY(exchange_counts(c, cm, send_counts, local_status, nullptr, 0, d_count_row));
    return exchange_payload(c, cm, send_counts, cols, cm->world + 1, out_n_recv);
}

// `prior_status`: a failure of the caller's local phase (the join's fused pass): this rank still takes part in every
// collective below, contributes nothing, and all ranks return an error together.
static int32_t dist_groupby_impl(pandrs_hip_ctx *c, pandrs_hip_comm *cm, int32_t mem_space, const pandrs_hip_column *key,
                                 int64_t n_rows, const pandrs_hip_column *vals, int32_t n_vals, const pandrs_hip_agg_spec *aggs,
                                 int32_t n_aggs, int64_t *out_n_groups, int32_t prior_status = 0) {
    if (n_vals > 62) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_groupby_agg: more than 62 value columns");
    for (int a = 0; a < n_aggs; a++)
        if (aggs[a].op > PANDRS_HIP_AGG_COUNT)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "dist_groupby_agg exchanges partial states: Sum / Mean / Min / Max / Count only "
                                                         "(the row shuffle for the rest is pandrs_hip_shuffle_split + the host's all-to-all)");
    // PANDRS_HIP_DIST_TRACE=1: wall time of every stage on stderr (each mark synchronises the stream: a diagnostic, not a mode)
    static const bool trace = std::getenv("PANDRS_HIP_DIST_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace) return;
        (void)hipStreamSynchronize(c->stream);
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[dist rank %d] %-28s %8.3f ms\n", cm->rank, what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // 0. one layout on every rank: a value column has an nn state iff SOME rank passes a null mask for it.  No collective of its
    // own: every rank plans with the layout it ASSUMES — the one agreed in the previous call of this shape, plus its own masks — and
    // puts (its own masks, the layout it used) on the count exchange.  When every rank used the same layout and that layout covers
    // every rank's masks (the steady state: always), the exchange goes ahead; otherwise every rank repeats its local phase with the
    // union, which is then the same word everywhere by construction.  (Round 3: a host-synchronous all-reduce in front of every call.)
    int64_t my_flags = 0;
    for (int i = 0; i < n_vals; i++) if (!prior_status && vals[i].null_mask) my_flags |= int64_t(1) << i;
    int64_t used = my_flags | (cm->agreed_n_vals == n_vals ? cm->agreed_flags : 0);
    int32_t status = prior_status;
    int64_t ng = 0;
    int32_t n_state = 0;
    uint64_t *send = nullptr;
    const int n_extra = 2, row = cm->world + 1 + n_extra;
    std::vector<int64_t> send_counts((size_t)cm->world, 0);
    // (its own arena, sized when the communicator was made: `small` is re-used by every collective)
    if (cm->crow.cap < 4096 + (size_t)row * 8) status = status ? status : cm->crow.ensure(4096 + (size_t)row * 8, c->stream);
    cm->crow.off = 0;
    int64_t *d_count_row = cm->crow.take<int64_t>((size_t)row);
    if (!d_count_row && !status) status = fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "exchange buffer too small (counts)");
    std::vector<uint8_t> has_nulls((size_t)std::max(n_vals, 1), 0);
    std::vector<int32_t> dtypes((size_t)std::max(n_vals, 1), 0);
    for (int i = 0; i < n_vals; i++) dtypes[i] = vals[i].dtype;
    // the local phase: a failure here must not return before the count exchange (the peers are waiting in it)
    auto local_phase = [&]() -> int32_t {
        std::vector<pandrs_hip_column> v2(vals, vals + n_vals);
        std::vector<uint8_t> zero_host;
        for (int i = 0; i < n_vals; i++) {
            if (((used >> i) & 1) && !vals[i].null_mask) {           // another rank has nulls here: an all-valid bitmap keeps the plan identical
                const size_t nb = (size_t)(n_rows + 7) / 8 + 8;
                if (mem_space == PANDRS_HIP_MEM_HOST) {
                    if (zero_host.size() < nb) zero_host.assign(nb, 0);
                    v2[i].null_mask = zero_host.data();
                } else {
                    if (cm->zeros.cap < nb + 256 || cm->zeros_valid < nb) {       // kept in the communicator: zeroed once per growth
                        ST_TRY(cm->zeros.ensure(nb + 256, c->stream));
                        HIP_TRY(hipMemsetAsync(cm->zeros.base, 0, cm->zeros.cap, c->stream));
                        cm->zeros_valid = cm->zeros.cap;
                    }
                    v2[i].null_mask = reinterpret_cast<const uint8_t *>(cm->zeros.base);
                }
            }
        }
        // 1. local partial aggregation (states retained in the context)
        ST_TRY(groupby_entry(c, mem_space, key, 1, n_rows, v2.data(), n_vals, aggs, n_aggs, /*partials=*/true, &ng, &n_state));
        mark("local partials");
        // 2. owner split: one block of records per owner; the counts stay on the device (no host round trip here)
        const size_t W = 2 + (size_t)n_state;
        ST_TRY(cm->send.ensure((size_t)std::max<int64_t>(ng, 1) * W * 8 + 4096, c->stream));
        send = cm->send.take<uint64_t>((size_t)std::max<int64_t>(ng, 1) * W);
        ST_TRY(partials_split_blocks_entry(c, cm->world, send, d_count_row));
        mark("owner split");
        return 0;
    };
    for (int attempt = 0;; attempt++) {
        if (!status) status = local_phase();
        // 3. count exchange (+ status and layout agreement) straight from the device counts.  A rank whose local phase failed sends
        // nothing and receives nothing: every rank returns the error together.
        if (!status && n_state <= 0) status = fail(PANDRS_HIP_ERR_COMPUTATION, "dist_groupby_agg: the local phase left no partial states");
        const int64_t extra[2] = {my_flags, used};
        ST_TRY(exchange_counts(c, cm, send_counts.data(), status, extra, n_extra, d_count_row));
        int64_t all_masks = 0, all_used = 0;
        bool same = true;
        for (int r = 0; r < cm->world; r++) {
            all_masks |= cm->counts[(size_t)r * row + cm->world + 1];
            all_used |= cm->counts[(size_t)r * row + cm->world + 2];
            same = same && cm->counts[(size_t)r * row + cm->world + 2] == used;
        }
        if (same && (all_masks & ~used) == 0) break;              // one layout everywhere, and it covers every rank's masks
        if (attempt >= 1) return fail(PANDRS_HIP_ERR_COMPUTATION, "dist_groupby_agg: the ranks could not agree on a null-mask layout");
        used = all_masks | all_used;                               // the same word on every rank: the second round must agree
        mark("layout disagreement: local phase repeated");
    }
    cm->agreed_flags = used; cm->agreed_n_vals = n_vals;
    for (int i = 0; i < n_vals; i++) has_nulls[i] = (uint8_t)((used >> i) & 1);
    // ... and ONE grouped all-to-all, a block per peer.  The block width is a function of (dtypes, agreed null flags, aggs) alone:
    // identical on every rank.
    const size_t W = 2 + (size_t)std::max(n_state, 0);
    std::vector<ExCol> ex;
    ex.push_back(ExCol{send, W * 8, nullptr});
    int64_t n_recv = 0;
    ST_TRY(exchange_payload(c, cm, send_counts.data(), ex, row, &n_recv));
    mark("counts + all-to-all");
    // 4. merge what this rank owns (cardinality bounded by the records received: no sampling pass)
    std::vector<int64_t> roff((size_t)cm->world + 1, 0);
    for (int r = 0; r < cm->world; r++) roff[(size_t)r + 1] = roff[(size_t)r] + cm->counts[(size_t)r * row + cm->rank];
    const int64_t hint_saved = c->opt.groups_hint;
    c->opt.groups_hint = std::max<int64_t>(n_recv, 1);
    const int32_t st = groupby_merge_blocks_entry(c, key->dtype, (const uint64_t *)ex[0].recv, roff.data(), cm->world, dtypes.data(), n_vals,
                                                  has_nulls.data(), aggs, n_aggs, out_n_groups);
    c->opt.groups_hint = hint_saved;
    mark("merge");
    return st;
}

// ---- the GENERAL exchange: rows to the owner of their key --------------------------------------------------------------
// For what partial states cannot express (Std / Var / Median / Nunique, several key columns): every row goes to the rank that
// owns its key (pandrs_hip_shuffle_split: the radix partitioner with P = world), ONE count exchange, one grouped all-to-all of
// all columns, and the owner runs the ordinary groupby on what it received.  A composite key is shuffled on a hash cell of
// the whole tuple (pandrs_hip_key_hash_cells) with the key columns travelling as payload.
static int32_t dist_groupby_shuffle_impl(pandrs_hip_ctx *c, pandrs_hip_comm *cm, int32_t mem_space, const pandrs_hip_column *keys, int32_t n_keys,
                                         int64_t n_rows, const pandrs_hip_column *vals, int32_t n_vals, const pandrs_hip_agg_spec *aggs,
                                         int32_t n_aggs, int64_t *out_n_groups) {
    if (n_keys < 1 || n_keys > 8 || n_vals > 16 - n_keys) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_groupby_agg: 1..8 key columns, keys + values <= 16");
    for (int a = 0; a < n_aggs; a++)
        if (aggs[a].op == PANDRS_HIP_AGG_FIRST || aggs[a].op == PANDRS_HIP_AGG_LAST)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "First / Last need the global row order and are not sharded");
    for (int k = 0; k < n_keys; k++)
        if (n_keys > 1 && keys[k].dtype != PANDRS_HIP_I64 && keys[k].dtype != PANDRS_HIP_F64 && keys[k].dtype != PANDRS_HIP_U32CODE)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "dist_groupby_agg: key column %d of a composite key cannot travel as shuffle payload (dtype %d)", k, keys[k].dtype);
    for (int v = 0; v < n_vals; v++)
        if (vals[v].dtype != PANDRS_HIP_I64 && vals[v].dtype != PANDRS_HIP_F64)
            return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "dist_groupby_agg (row shuffle): value column %d has dtype %d (i64 / f64 only)", v, vals[v].dtype);
    // 0. one column layout on every rank: a column carries null flags iff SOME rank passes a mask for it
    const int n_cols = n_keys + n_vals;
    int64_t flags[17] = {0};
    for (int k = 0; k < n_keys; k++) flags[k] = keys[k].null_mask ? 1 : 0;
    for (int v = 0; v < n_vals; v++) flags[n_keys + v] = vals[v].null_mask ? 1 : 0;
    ST_TRY(agree_max(c, cm, flags, n_cols));
    std::vector<pandrs_hip_column> cols(keys, keys + n_keys);
    cols.insert(cols.end(), vals, vals + n_vals);
    std::vector<uint8_t> zero_host;
    std::vector<uint64_t> hcells_host;
    std::vector<int64_t> send_counts((size_t)cm->world, 0);
    int64_t n_send = 0;
    auto local_phase = [&]() -> int32_t {
        for (int i = 0; i < n_cols; i++) {
            if (flags[i] && !cols[i].null_mask) {
                const size_t nb = (size_t)(n_rows + 7) / 8 + 8;
                if (mem_space == PANDRS_HIP_MEM_HOST) {
                    if (zero_host.size() < nb) zero_host.assign(nb, 0);
                    cols[i].null_mask = zero_host.data();
                } else {
                    if (cm->zeros.cap < nb + 256 || cm->zeros_valid < nb) {
                        ST_TRY(cm->zeros.ensure(nb + 256, c->stream));
                        HIP_TRY(hipMemsetAsync(cm->zeros.base, 0, cm->zeros.cap, c->stream));
                        cm->zeros_valid = cm->zeros.cap;
                    }
                    cols[i].null_mask = reinterpret_cast<const uint8_t *>(cm->zeros.base);
                }
            }
        }
        if (n_keys == 1) {
            ST_TRY(shuffle_split_entry(c, mem_space, &cols[0], n_vals ? &cols[1] : nullptr, n_vals, n_rows, cm->world, 0, send_counts.data(), &n_send));
        } else {
            // the shuffle key: one hash cell per row of the whole tuple; the key columns travel as payload
            pandrs_hip_column hk{nullptr, nullptr, PANDRS_HIP_CELL64, 0};
            if (mem_space == PANDRS_HIP_MEM_HOST) {
                hcells_host.resize((size_t)std::max<int64_t>(n_rows, 1));
                ST_TRY(key_hash_cells_entry(c, mem_space, cols.data(), n_keys, n_rows, hcells_host.data()));
                hk.data = hcells_host.data();
            } else {
                ST_TRY(cm->stage.ensure((size_t)std::max<int64_t>(n_rows, 1) * 8 + 4096, c->stream));
                uint64_t *hc = cm->stage.take<uint64_t>((size_t)std::max<int64_t>(n_rows, 1));
                ST_TRY(key_hash_cells_entry(c, mem_space, cols.data(), n_keys, n_rows, hc));
                hk.data = hc;
            }
            ST_TRY(shuffle_split_entry(c, mem_space, &hk, cols.data(), n_cols, n_rows, cm->world, 0, send_counts.data(), &n_send));
        }
        return 0;
    };
    const int32_t status = local_phase();
    // 1. the columns that travel: [cells + key null bytes (single key)] + payload columns + their null bytes
    std::vector<ExCol> ex;
    const ShuffleResult &sh = c->sh;
    const int n_pay = n_keys == 1 ? n_vals : n_cols;
    if (!status) {
        if (n_keys == 1) { ex.push_back(ExCol{sh.cells, 8, nullptr}); ex.push_back(ExCol{sh.key_null, 1, nullptr}); }
        for (int p = 0; p < n_pay; p++) ex.push_back(ExCol{sh.pay[p], 8, nullptr});
        for (int p = 0; p < n_pay; p++) if (flags[n_keys == 1 ? 1 + p : p]) ex.push_back(ExCol{sh.pay_null[p], 1, nullptr});
    }
    int64_t n_recv = 0;
    ST_TRY(exchange_columns(c, cm, send_counts.data(), ex, status, &n_recv));
    // 2. the received columns as ordinary groupby inputs (null bytes -> bitmaps)
    const size_t nb = (size_t)(n_recv + 7) / 8 + 64;
    ST_TRY(cm->small.ensure((size_t)(n_cols + 1) * Arena::padded(nb) + 4096, c->stream));
    size_t at = 0;
    auto next_recv = [&]() { return ex[at++].recv; };
    std::vector<pandrs_hip_column> k2((size_t)n_keys), v2((size_t)std::max(n_vals, 1));
    std::vector<const uint8_t *> null_bytes((size_t)n_cols, nullptr);
    std::vector<void *> data((size_t)n_cols, nullptr);
    const uint8_t *key_null_bytes = nullptr;
    if (n_keys == 1) { data[0] = next_recv(); key_null_bytes = (const uint8_t *)next_recv(); }
    for (int p = 0; p < n_pay; p++) data[n_keys == 1 ? 1 + p : p] = next_recv();
    for (int p = 0; p < n_pay; p++) { const int i = n_keys == 1 ? 1 + p : p; if (flags[i]) null_bytes[i] = (const uint8_t *)next_recv(); }
    if (n_keys == 1 && flags[0]) null_bytes[0] = key_null_bytes;
    for (int i = 0; i < n_cols; i++) {
        uint8_t *bits = nullptr;
        if (null_bytes[i]) {
            bits = cm->small.take<uint8_t>(nb);
            if (!bits) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "exchange buffer too small (bitmaps)");
            if (n_recv) ST_TRY(bytes_to_bitmap_entry(c, PANDRS_HIP_MEM_DEVICE, null_bytes[i], n_recv, bits));
        }
        int32_t dt = i < n_keys ? keys[i].dtype : vals[i - n_keys].dtype;
        if (i < n_keys && (n_keys == 1 || dt == PANDRS_HIP_U32CODE)) dt = PANDRS_HIP_CELL64;      // normalised / zero-extended cells
        pandrs_hip_column col{data[i], bits, dt, 0};
        if (i < n_keys) k2[i] = col; else v2[i - n_keys] = col;
    }
    return groupby_entry(c, PANDRS_HIP_MEM_DEVICE, k2.data(), n_keys, n_recv, v2.data(), n_vals, aggs, n_aggs, /*partials=*/false, out_n_groups, nullptr);
}

}  // namespace pandrs

using pandrs::fail;

extern "C" {

int32_t pandrs_hip_comm_unique_id(char out_id[128]) try {
    if (!out_id) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "null out_id");
    ST_TRY(pandrs::rccl_load());
    pandrs::RcclUniqueId id;
    RCCL_TRY(pandrs::rccl().GetUniqueId(&id));
    std::memcpy(out_id, id.internal, 128);
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_comm_unique_id"); }

int32_t pandrs_hip_comm_init(pandrs_hip_ctx *ctx, const char id[128], int32_t rank, int32_t world, pandrs_hip_comm **out) try {
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
} catch (...) { return pandrs::on_exception("pandrs_hip_comm_init"); }

int32_t pandrs_hip_comm_adopt(void *nccl_comm, int32_t rank, int32_t world, pandrs_hip_comm **out) try {
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "comm_adopt: bad arguments");
    ST_TRY(pandrs::rccl_load());
    auto *cm = new pandrs_hip_comm();
    cm->nccl = nccl_comm; cm->rank = rank; cm->world = world; cm->owned = false;
    *out = cm;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_comm_adopt"); }

int32_t pandrs_hip_comm_adopt_transport(const pandrs_hip_transport *t, int32_t rank, int32_t world, pandrs_hip_comm **out) try {
    if (!t || !out || world < 1 || rank < 0 || rank >= world || !t->all_gather || !t->all_reduce_max_i64 || !t->all_to_all_v)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "comm_adopt_transport: bad arguments (all three callbacks are required)");
    auto *cm = new pandrs_hip_comm();
    cm->host = *t; cm->rank = rank; cm->world = world; cm->owned = false;
    *out = cm;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_comm_adopt_transport"); }

int32_t pandrs_hip_comm_destroy(pandrs_hip_comm *cm) try {
    if (!cm) return PANDRS_HIP_OK;
    cm->send.release(); cm->recv.release(); cm->small.release(); cm->zeros.release(); cm->stage.release(); cm->crow.release();
    if (cm->owned && cm->nccl) (void)pandrs::rccl().CommDestroy(cm->nccl);
    delete cm;
    return PANDRS_HIP_OK;
} catch (...) { return pandrs::on_exception("pandrs_hip_comm_destroy"); }

int32_t pandrs_hip_dist_groupby_agg(pandrs_hip_ctx *ctx, pandrs_hip_comm *comm, int32_t mem_space, const pandrs_hip_column *keys,
                                    int32_t n_keys, int64_t n_rows, const pandrs_hip_column *vals, int32_t n_vals,
                                    const pandrs_hip_agg_spec *aggs, int32_t n_aggs, int64_t *out_n_groups) try {
    if (!ctx || !comm || !keys || !out_n_groups || n_rows < 0 || n_vals < 0 || n_aggs < 0 || (n_vals && !vals) || (n_aggs && !aggs))
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_groupby_agg: bad arguments");
    if (comm->aborted) return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "this communicator was aborted after a rank-local failure: create a new one");
    HIP_TRY(hipSetDevice(ctx->device));
    bool mergeable = n_keys == 1;
    for (int a = 0; a < n_aggs; a++) mergeable = mergeable && aggs[a].op <= PANDRS_HIP_AGG_COUNT;
    if (!mergeable)       // Std / Var / Median / Nunique, or a composite key: the rows go to the owner of their key
        return pandrs::dist_groupby_shuffle_impl(ctx, comm, mem_space, keys, n_keys, n_rows, vals, n_vals, aggs, n_aggs, out_n_groups);
    return pandrs::dist_groupby_impl(ctx, comm, mem_space, keys, n_rows, vals, n_vals, aggs, n_aggs, out_n_groups);
} catch (...) { return pandrs::on_exception("pandrs_hip_dist_groupby_agg"); }

int32_t pandrs_hip_dist_join_groupby_sum(pandrs_hip_ctx *ctx, pandrs_hip_comm *cm, int32_t mem_space,
                                         const pandrs_hip_column *left_key, const pandrs_hip_column *left_val, int64_t n_left,
                                         const pandrs_hip_column *right_key, const pandrs_hip_column *right_group, int64_t n_right,
                                         int64_t *out_n_groups) try {
    using namespace pandrs;
    if (!ctx || !cm || !left_key || !left_val || !right_key || !right_group || !out_n_groups || n_left < 0 || n_right < 0)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_join_groupby_sum: bad arguments");
    if (mem_space != PANDRS_HIP_MEM_DEVICE)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "dist_join_groupby_sum takes device-resident shards (PANDRS_HIP_MEM_DEVICE)");
    if ((right_key->dtype != PANDRS_HIP_I64 && right_key->dtype != PANDRS_HIP_F64 && right_key->dtype != PANDRS_HIP_CELL64) ||
        (right_group->dtype != PANDRS_HIP_I64 && right_group->dtype != PANDRS_HIP_F64 && right_group->dtype != PANDRS_HIP_CELL64))
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "dist_join_groupby_sum: 8-byte build-side columns only");
    if (cm->aborted) return fail(PANDRS_HIP_ERR_NOT_INITIALIZED, "this communicator was aborted after a rank-local failure: create a new one");
    pandrs_hip_ctx *c = ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int world = cm->world;
    // 1. all-gather the build side; shards are padded to a common length with NULL-key rows
    // (one all-reduce: the longest shard, the mask flags, and — as the max of the negated lengths — the shortest shard: every rank needs
    // to know whether padding exists anywhere)
    int64_t mx[4] = {n_right, right_key->null_mask ? 1 : 0, right_group->null_mask ? 1 : 0, -n_right};
    ST_TRY(agree_max(c, cm, mx, 4));
    const int64_t n_pad = (mx[0] + 7) / 8 * 8, n_all = n_pad * world;
    const bool padded = -mx[3] != n_pad;
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
    ST_TRY(all_gather_dev(c, cm, sk, ak, (size_t)n_pad * 8));
    ST_TRY(all_gather_dev(c, cm, sg, ag, (size_t)n_pad * 8));
    if (key_mask) ST_TRY(all_gather_dev(c, cm, skm, akm, mb));
    if (grp_mask) ST_TRY(all_gather_dev(c, cm, sgm, agm, mb));
    // 2. local fused join -> groupby-sum against the whole build side.  From here to the groupby exchange's count
    // all-gather nothing may return early: a rank-local failure travels as `status` and every rank aborts together.
    pandrs_hip_column rk{ak, key_mask ? akm : nullptr, right_key->dtype, 0}, rg{ag, grp_mask ? agm : nullptr, right_group->dtype, 0};
    int64_t g_local = 0, got = 0;
    pandrs_hip_column gk{nullptr, nullptr, PANDRS_HIP_CELL64, 0}, gv{nullptr, nullptr, PANDRS_HIP_F64, 0};
    auto local_phase = [&]() -> int32_t {
        ST_TRY(join_groupby_sum_entry(c, PANDRS_HIP_MEM_DEVICE, left_key, left_val, n_left, &rk, &rg, n_all, &g_local));
        // 3. the <= G local sums go through the groupby exchange, keyed on g's cell; they live in the communicator's own
        // `stage` arena (grown, never released): the nested exchange re-takes send / recv
        ST_TRY(cm->stage.ensure((size_t)std::max<int64_t>(g_local, 1) * 17 + ((size_t)g_local + 7) / 8 + (1 << 16), c->stream));
        uint64_t *cells = cm->stage.take<uint64_t>((size_t)std::max<int64_t>(g_local, 1));
        double *sums = cm->stage.take<double>((size_t)std::max<int64_t>(g_local, 1));
        uint8_t *nb = cm->stage.take<uint8_t>((size_t)std::max<int64_t>(g_local, 1)), *nbits = cm->stage.take<uint8_t>(((size_t)g_local + 7) / 8 + 16);
        if (!cells || !sums || !nb || !nbits) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "stage arena too small");
        ST_TRY(export_result_columns(c, cells, nb, sums, &got));
        ST_TRY(bytes_to_bitmap_entry(c, PANDRS_HIP_MEM_DEVICE, nb, got, nbits));
        gk.data = cells; gk.null_mask = nbits; gv.data = sums;
        return 0;
    };
    const int32_t local_status = local_phase();
    const pandrs_hip_agg_spec sum_spec{0, PANDRS_HIP_AGG_SUM};
    return dist_groupby_impl(c, cm, PANDRS_HIP_MEM_DEVICE, &gk, local_status ? 0 : got, &gv, 1, &sum_spec, 1, out_n_groups, local_status);
} catch (...) { return pandrs::on_exception("pandrs_hip_dist_join_groupby_sum"); }

}  // extern "C"
