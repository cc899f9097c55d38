// join.hip — radix-partitioned hash join for MI355X (gfx950, wave64).
//
// Replaces OptimizedDataFrame::join_impl up to the join_indices vector
// (reference src/optimized/split_dataframe/join.rs:106-224), which stringifies both key columns,
// builds HashMap<String, Vec<usize>> over the right side and probes the left side on one thread.
//
// Device plan (inputs / outputs resident in HBM):
//   1. the BUILD (right) side is radix-partitioned on hash(key cell) with the groupby engine's
//      histogram / scan / LDS-staged scatter, carrying the original row index; null keys go to a
//      partition of their own and are never built (join.rs:112);
//   2. build pass, one workgroup per partition: the rows are grouped by key with an LDS hash table
//      (count per key -> scan -> place) and every key's short run is put in ascending right-row
//      order, the reference's per-key Vec<usize> order (join.rs:114, :156-158); that LDS table is the
//      partition's region of the global table of 16-byte {key, start, count} entries and is stored
//      whole, without global atomics.  Partitions with a long run fall back to an LDS bitonic sort,
//      partitions beyond LDS to the general segmented sort (segsort.hip);
//   3. probe pass over the LEFT rows in their ORIGINAL order (never partitioned): one lookup per row — its 64-byte
//      home bucket of the table, read whole — -> {first match, output rows}, and the output rows of every
//      2048-row tile; null left keys emit nothing (join.rs:152);
//   4. exclusive scan of the TILE sums (64-bit) = output offset of every tile in reference order
//      (left rows ascending, join.rs:151);
//   5. emit pass, tile by tile in original left order: a row's position = its tile's offset + the in-tile prefix
//      of the match records it re-reads (no per-row count / offset arrays); coalesced index-pair writes, the
//      only gather is from the sorted right-row array; left/outer misses write (left, -1) (join.rs:159-162);
//   6. right/outer: runs hit by the probe mark their right rows; unmatched right rows (null keys
//      included) are compacted ascending behind the probe output (join.rs:211-224).
// All of it is HBM-bound integer work: no MFMA.
#include "engine.hpp"

#include <algorithm>
#include <cmath>

namespace pandrs {

constexpr int JN_THREADS = 1024;
constexpr int JN_RCAP = 8192;           // right rows per partition that fit the LDS sort buffers
constexpr uint32_t JN_SEED = 0x51ED270Bu;
constexpr uint32_t NO_MATCH = 0xFFFFFFFFu;
constexpr uint32_t JN_DIRECT = 0x80000000u;   // JoinEntry.count flag: a single-row run, `start` IS the right row (no gather)

// ---- LDS bitonic sort of (key, payload) pairs, ascending by key then payload ----------------------
template <typename PT>
__device__ __forceinline__ void lds_bitonic_sort(uint64_t *sk, PT *sp, uint32_t n2) {
    for (uint32_t k = 2; k <= n2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n2; i += JN_THREADS) {
                uint32_t x = i ^ j;
                if (x > i) {
                    uint64_t ka = sk[i], kb = sk[x];
                    PT pa = sp[i], pb = sp[x];
                    bool gt = ka > kb || (ka == kb && pa > pb);
                    bool up = (i & k) == 0;
                    if (gt == up) { sk[i] = kb; sk[x] = ka; sp[i] = pb; sp[x] = pa; }
                }
            }
            __syncthreads();
        }
    }
}

// Global open-addressing table over the build side's DISTINCT keys: 16-byte entries
// {key cell, first position of the key's run in the sorted right arrays, run length}.
struct __attribute__((aligned(16))) JoinEntry {
    uint64_t key;
    uint32_t start, count;      // count & JN_DIRECT: run of one row, start = that row itself
};

// The table is cut into one REGION per build partition (a key's partition is a function of the key,
// so a lookup knows its region).  A build workgroup owns its region: it assembles the region's
// open-addressing layout in LDS and writes it out with plain coalesced stores — no global atomics
// (device-scope CAS on random addresses retires at only ~10 G/s on this chip, which used to BE the
// build time).  regions == 1 is the plain global table filled with atomics (general build path).
// The entry behind the last region serves the key ~0.
constexpr uint32_t BH_SLOTS = 8192;             // entries per region = slots of the build kernel's LDS table
constexpr uint32_t BH_SEED = 0x3243F6A8u;
struct TableRef {
    JoinEntry *t;
    uint32_t regions, rmask;        // rmask = region capacity - 1
    __device__ __forceinline__ uint32_t sentinel() const { return regions * (rmask + 1); }
    __device__ __forceinline__ uint32_t base_of(uint64_t k) const {
        return regions > 1 ? part_of(hash32(k, JN_SEED), regions) * (rmask + 1) : 0u;
    }
    // home slots are multiples of 4: a key sits in its 64-byte home BUCKET (4 entries) unless that overflowed, so a lookup that
    // reads the bucket whole needs ~1.1 line reads where entry-by-entry probing at load 0.6 needs 1.75 (the probe is bound by
    // random line reads out of MALL: 50 M x 5 M lookup kernel 1.43 -> see DESIGN §7)
    __device__ __forceinline__ uint32_t home_of(uint64_t k) const { return (hash32(k, BH_SEED) & (rmask >> 2)) << 2; }
    // claims the first free entry of the key's probe sequence (distinct keys only)
    __device__ __forceinline__ void insert(uint32_t base, uint64_t k, uint32_t start, uint32_t count) const {
        uint32_t o = home_of(k);
        for (uint32_t probes = 0; probes <= rmask; probes++) {       // bounded: a full region drops the entry instead of spinning
            uint64_t old = atomicCAS((unsigned long long *)&t[base + o].key, EMPTY_KEY, k);
            if (old == EMPTY_KEY) { t[base + o].start = start; t[base + o].count = count; return; }
            o = (o + 1) & rmask;
        }
    }
};

struct BuildArgs {
    uint64_t *rkeys; uint32_t *rrows;           // partitioned right side, sorted in place per partition
    const uint32_t *roff;                       // partition offsets (PartInfo.offsets)
    uint32_t rNB;
    TableRef tab;
    uint32_t *flags;                            // [0] = a right partition did not fit LDS
};

// Build pass, fallback body: bitonic sort of the whole partition by (key, right row) in LDS
// (sk[R2] u64 | sp[R2] u32), written back sorted, every run {key -> start, count} published.
__device__ __forceinline__ void build_partition_bitonic(const BuildArgs &a, unsigned char *smem, uint32_t rbeg, uint32_t nR, uint32_t tbase) {
    const uint32_t tid = threadIdx.x;
    uint32_t n2 = 64;
    while (n2 < nR) n2 <<= 1;
    uint64_t *sk = reinterpret_cast<uint64_t *>(smem);
    uint32_t *sp = reinterpret_cast<uint32_t *>(sk + JN_RCAP);
    for (uint32_t i = tid; i < n2; i += JN_THREADS) {
        sk[i] = i < nR ? a.rkeys[rbeg + i] : ~0ull;
        sp[i] = i < nR ? a.rrows[rbeg + i] : 0xFFFFFFFFu;
    }
    __syncthreads();
    lds_bitonic_sort<uint32_t>(sk, sp, n2);
    for (uint32_t i = tid; i < nR; i += JN_THREADS) {
        const uint64_t k = sk[i];
        a.rkeys[rbeg + i] = k; a.rrows[rbeg + i] = sp[i];
        if (i > 0 && sk[i - 1] == k) continue;          // not the start of a run
        uint32_t m = 1;
        while (i + m < nR && sk[i + m] == k) m++;
        const uint32_t e_start = m == 1 ? sp[i] : rbeg + i, e_count = m == 1 ? (1u | JN_DIRECT) : m;
        if (k == EMPTY_KEY) {                            // the sentinel-valued key has a dedicated entry
            a.tab.t[a.tab.sentinel()].start = e_start; a.tab.t[a.tab.sentinel()].count = e_count;
            continue;
        }
        a.tab.insert(tbase, k, e_start, e_count);
    }
}

// Build pass, one workgroup per right partition.  The join needs, per key, its right rows in
// ascending order (join.rs:114, :156-158) — not an order between keys.  So instead of sorting the
// partition: group the rows by key with an LDS hash table (count per key -> scan -> place), then
// order the rows inside each key's run (a run is one row for a primary-key build side, the common
// case, and a few rows otherwise).  ~4 LDS passes instead of the 91 stages of an 8192-element
// bitonic sort.  A key with more than BH_MAXRUN rows sends the partition to the bitonic fallback.
// LDS: sk[BH_SLOTS] u64 keys | pc[BH_SLOTS + 1] u32 (count | cursor << 16) | lrows[JN_RCAP] u32 | scan scratch
constexpr uint32_t BH_MAXRUN = 48;
constexpr int BH_RPT = JN_RCAP / JN_THREADS;
__global__ __launch_bounds__(JN_THREADS) void join_build_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t rbeg = a.roff[(size_t)p * a.rNB], rend = a.roff[(size_t)(p + 1) * a.rNB];
    const uint32_t nR = rend - rbeg;
    const uint32_t tbase = p * BH_SLOTS;
    if (nR == 0 || nR > BH_SLOTS / 8 * 7) {
        // nothing to build, or too much for a region that must keep free entries (the host retries): the
        // region is written EMPTY either way — the probe that is already queued must find its way out
        for (uint32_t s = tid; s < BH_SLOTS; s += JN_THREADS) a.tab.t[tbase + s] = JoinEntry{EMPTY_KEY, 0u, 0u};
        if (nR != 0 && tid == 0) a.flags[0] = 1;
        return;
    }
    uint64_t *sk = reinterpret_cast<uint64_t *>(smem);
    uint32_t *pc = reinterpret_cast<uint32_t *>(sk + BH_SLOTS);
    uint32_t *lrows = pc + BH_SLOTS + 8;
    uint32_t *wt = lrows + JN_RCAP;                  // 17 words of scan scratch, [20] = fallback flag
    for (uint32_t s = tid; s < BH_SLOTS; s += JN_THREADS) { sk[s] = EMPTY_KEY; pc[s] = 0; }
    if (tid < 8) pc[BH_SLOTS + tid] = 0;
    if (tid < 32) wt[tid] = 0;
    __syncthreads();
    // 1. count the rows of every distinct key; each thread remembers the slots of its rows
    uint64_t k[BH_RPT];
    uint32_t row[BH_RPT], slot[BH_RPT];
#pragma unroll
    for (int r = 0; r < BH_RPT; r++) {
        const uint32_t i = r * JN_THREADS + tid;
        if (i < nR) { k[r] = a.rkeys[rbeg + i]; row[r] = a.rrows[rbeg + i]; }
    }
#pragma unroll
    for (int r = 0; r < BH_RPT; r++) {
        const uint32_t i = r * JN_THREADS + tid;
        if (i >= nR) continue;
        uint32_t s = BH_SLOTS;                       // the sentinel-valued key counts in the extra entry
        if (k[r] != EMPTY_KEY) {
            s = (hash32(k[r], BH_SEED) & (BH_SLOTS / 4 - 1)) << 2;       // = TableRef::home_of: the home bucket's first slot
            for (;;) {
                const uint64_t old = atomicCAS((unsigned long long *)&sk[s], EMPTY_KEY, k[r]);
                if (old == EMPTY_KEY || old == k[r]) break;
                s = (s + 1) & (BH_SLOTS - 1);
            }
        }
        slot[r] = s;
        if ((atomicAdd(&pc[s], 1u) & 0xFFFFu) + 1 > BH_MAXRUN) wt[20] = 1;
    }
    __syncthreads();
    if (wt[20]) {                                    // a long run (hot build key): sort the partition instead,
        for (uint32_t s = tid; s < BH_SLOTS; s += JN_THREADS)      // publish into the (emptied) region with atomics
            a.tab.t[tbase + s] = JoinEntry{EMPTY_KEY, 0u, 0u};
        __threadfence();
        __syncthreads();
        build_partition_bitonic(a, smem, rbeg, nR, tbase);
        return;
    }
    // 2. exclusive scan of the counts = start of every key's run; cursor = start
    {
        uint32_t c[BH_RPT + 1], mine = 0;
        const uint32_t s0 = tid * BH_RPT;
#pragma unroll
        for (int j = 0; j < BH_RPT; j++) { c[j] = pc[s0 + j]; mine += c[j]; }
        c[BH_RPT] = tid == JN_THREADS - 1 ? pc[BH_SLOTS] : 0u;
        mine += c[BH_RPT];
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<JN_THREADS>(mine, wt, &tot);
#pragma unroll
        for (int j = 0; j < BH_RPT; j++) { pc[s0 + j] = c[j] | (ex << 16); ex += c[j]; }
        if (tid == JN_THREADS - 1) pc[BH_SLOTS] = c[BH_RPT] | (ex << 16);
    }
    __syncthreads();
    // 3. place every row in its key's run (order inside a run arbitrary for now)
#pragma unroll
    for (int r = 0; r < BH_RPT; r++) {
        const uint32_t i = r * JN_THREADS + tid;
        if (i < nR) lrows[atomicAdd(&pc[slot[r]], 1u << 16) >> 16] = row[r];
    }
    __syncthreads();
    // 4. per key: order its rows ascending (insertion sort of a short run); the LDS table IS the region's
    //    open-addressing layout (same hash, same probe order): write it out, entry by entry, coalesced
    for (uint32_t s = tid; s <= BH_SLOTS; s += JN_THREADS) {
        const uint32_t v = pc[s], m = v & 0xFFFFu;
        if (m == 0) { if (s < BH_SLOTS) a.tab.t[tbase + s] = JoinEntry{EMPTY_KEY, 0u, 0u}; continue; }
        const uint32_t start = (v >> 16) - m;
        for (uint32_t x = 1; x < m; x++) {
            const uint32_t rv = lrows[start + x];
            uint32_t y = x;
            while (y > 0 && lrows[start + y - 1] > rv) { lrows[start + y] = lrows[start + y - 1]; y--; }
            lrows[start + y] = rv;
        }
        const uint32_t e_start = m == 1 ? lrows[start] : rbeg + start, e_count = m == 1 ? (1u | JN_DIRECT) : m;
        if (s == BH_SLOTS) { a.tab.t[a.tab.sentinel()].start = e_start; a.tab.t[a.tab.sentinel()].count = e_count; continue; }
        a.tab.t[tbase + s] = JoinEntry{sk[s], e_start, e_count};
    }
    __syncthreads();
    for (uint32_t i = tid; i < nR; i += JN_THREADS) a.rrows[rbeg + i] = lrows[i];
}

// General build path (a partition larger than the LDS sort buffers: very many build rows, or one
// key with thousands of duplicates): the partitions were sorted by segmented_sort_u32; the sorted
// non-null rows [0, *n_bound) form one array in which equal keys are adjacent (a key lives in
// exactly one partition), so runs are published without looking at partition boundaries.
__global__ void publish_runs_kernel(const uint64_t *rkeys, const uint32_t *rrows, const uint32_t *n_bound, TableRef tab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = *n_bound;
    if (i >= n) return;
    const uint64_t k = rkeys[i];
    if (i > 0 && rkeys[i - 1] == k) return;
    const uint32_t m = sorted_run_length(rkeys, i, n);
    const uint32_t e_start = m == 1 ? rrows[i] : i, e_count = m == 1 ? (1u | JN_DIRECT) : m;
    if (k == EMPTY_KEY) { tab.t[tab.sentinel()].start = e_start; tab.t[tab.sentinel()].count = e_count; return; }
    tab.insert(tab.base_of(k), k, e_start, e_count);
}

// Lookup of 4 keys per thread: the 4 home buckets (4 x 4 entries of 16 B) are in flight together; a key that is not in its home
// bucket and saw no empty entry there walks on entry by entry (its bucket overflowed: ~10 % at load 0.6).
// skip[r]: no lookup (null key / row past the end); k == EMPTY_KEY: the sentinel-valued key's dedicated entry.
__device__ __forceinline__ void bucket_lookup4(const TableRef &tab, const uint64_t (&k)[4], const bool (&skip)[4],
                                               uint32_t (&slot)[4], JoinEntry (&e)[4], bool (&found)[4]) {
    const JoinEntry *table = tab.t;
    uint32_t tb[4];
    JoinEntry b[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        tb[r] = tab.base_of(k[r]);
        slot[r] = k[r] == EMPTY_KEY ? tab.sentinel() : tb[r] + tab.home_of(k[r]);     // (the table carries 8 spare entries behind sentinel())
#pragma unroll
        for (int j = 0; j < 4; j++) b[r][j] = table[slot[r] + j];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        found[r] = false;
        e[r] = b[r][0];
        if (skip[r]) continue;
        if (k[r] == EMPTY_KEY) { found[r] = b[r][0].count != 0; continue; }
        bool open = true;                                   // neither the key nor an empty entry seen yet
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool hit = open && b[r][j].key == k[r];
            if (hit) { found[r] = true; e[r] = b[r][j]; slot[r] += j; }
            open = open && !hit && b[r][j].key != EMPTY_KEY;
        }
        if (open) {                                         // the home bucket is full of other keys: walk on
            uint32_t sl = slot[r] - tb[r] + 3;
            JoinEntry x = b[r][3];
            for (uint32_t probes = 0; x.key != k[r] && x.key != EMPTY_KEY && probes < tab.rmask; probes++) {
                sl = (sl + 1) & tab.rmask;
                x = table[tb[r] + sl];
            }
            if (x.key == k[r]) { found[r] = true; e[r] = x; slot[r] = tb[r] + sl; }
        }
    }
}

// Probe pass over the left rows in ORIGINAL order.  LK_RPT rows per thread: all key loads, then all
// first-probe table reads (16-byte entries, cache resident for typical builds) are in flight together;
// only rows whose first entry is neither their key nor empty walk on.
constexpr int LK_THREADS = 256, LK_RPT = 8;
__global__ __launch_bounds__(LK_THREADS) void join_lookup_kernel(KeyDesc lkey, int64_t n_left, TableRef tab,
                                                                 int keep_left, int flag_right,
                                                                 uint2 *match, unsigned long long *tile_sum, uint8_t *hit) {
    const int64_t base = (int64_t)blockIdx.x * (LK_THREADS * LK_RPT) + threadIdx.x;
    unsigned long long mine = 0;                   // output rows of this thread's left rows
#pragma unroll
    for (int h = 0; h < LK_RPT; h += 4) {
        uint64_t k[4];
        bool skip[4], live[4], found[4];
        uint32_t slot[4];
        JoinEntry e[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t l0 = base + (int64_t)(h + r) * LK_THREADS, l = min(l0, n_left - 1);
            live[r] = l0 < n_left;
            skip[r] = !live[r] || key_is_null(lkey, l);     // null left keys are dropped even for left/outer (join.rs:152)
            k[r] = key_cell(lkey, l);
        }
        bucket_lookup4(tab, k, skip, slot, e, found);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (!live[r]) continue;
            const int64_t l = base + (int64_t)(h + r) * LK_THREADS;
            uint2 out = make_uint2(NO_MATCH, 0u);
            if (found[r]) {
                out = make_uint2(e[r].start, e[r].count);
                if (flag_right) hit[slot[r]] = 1;
            } else if (keep_left && !key_is_null(lkey, l)) {
                out.y = 1;
            }
            match[l] = out;
            mine += out.y & ~JN_DIRECT;
        }
    }
    // the tile's output rows: the emit pass re-derives every row's position from the scanned tile sums and the tile's own
    // match records, so no per-row count / offset arrays travel through HBM (16 B per left row less)
    __shared__ unsigned long long wsum[LK_THREADS / 64];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < LK_THREADS / 64; w++) t += wsum[w];
        tile_sum[blockIdx.x] = t;
    }
}

// exclusive scan of the tile sums (one workgroup; 64-bit: the total is checked against the 2^32-row limit), total -> *total
__global__ __launch_bounds__(1024) void tile_scan_kernel(const unsigned long long *tile_sum, uint32_t n_tiles, unsigned long long *tile_off,
                                                         unsigned long long *total) {
    __shared__ unsigned long long wtot[16];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + tid;
        const unsigned long long v = i < n_tiles ? tile_sum[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += t;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        unsigned long long before = carry;
        for (uint32_t w = 0; w < wave; w++) before += wtot[w];
        if (i < n_tiles) tile_off[i] = before + inc - v;
        __syncthreads();
        if (tid == 1023) carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

// ---- single-pass probe (opt-in, option join_one_pass): lookup + scan + emit in ONE kernel -----------
// The three-kernel probe above writes and re-reads 16 B of intermediates per left row (match, count,
// offset).  Here a tile of 2048 consecutive left rows looks its keys up, and the output position of
// the tile comes from a decoupled look-back over the tiles before it (each tile publishes first its
// own total, then its inclusive prefix, in one 64-bit word: flag << 62 | value), so the index pairs
// are written in the reference's order straight away.  Tiles take their number from an atomic ticket,
// which guarantees that every predecessor a tile waits for is already running.  The wait is bounded:
// on a timeout the kernel raises err[0] and the host falls back to the three-kernel probe.
constexpr int OP_THREADS = 256, OP_RPT = 8, OP_TILE = OP_THREADS * OP_RPT;
constexpr unsigned long long OP_VALUE = (1ull << 62) - 1;
struct OnePassArgs {
    KeyDesc lkey; int64_t n_left;
    TableRef tab;
    const uint32_t *rrows_sorted;
    int keep_left, flag_right;
    uint8_t *hit;
    unsigned long long *status;     // [n_tiles], zeroed
    uint32_t *ticket;               // zeroed; err = ticket + 1
    uint64_t cap;                   // capacity of out_left / out_right (rows); the totals keep counting past it
    int64_t *out_left, *out_right;
};
__global__ __launch_bounds__(OP_THREADS) void join_probe_onepass_kernel(OnePassArgs a) {
    constexpr int NW = OP_THREADS / 64;
    __shared__ uint32_t wsum[OP_RPT][NW];              // matches of (row slice r, wave w)
    __shared__ uint32_t s_tile;
    __shared__ unsigned long long s_excl;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    // row slice r of the tile = rows tile * OP_TILE + r * OP_THREADS + tid: coalesced key loads and,
    // for a unique-key build side, coalesced output stores (consecutive lanes, consecutive rows)
    const int64_t base = (int64_t)tile * OP_TILE + tid;
    uint32_t start[OP_RPT], cnt[OP_RPT], inc[OP_RPT];
    bool direct[OP_RPT];
#pragma unroll
    for (int h = 0; h < OP_RPT; h += 4) {
        uint64_t k4[4];
        bool skip[4], found[4];
        uint32_t slot4[4];
        JoinEntry e4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t l0 = base + (int64_t)(h + r) * OP_THREADS, l = min(l0, a.n_left - 1);
            skip[r] = key_is_null(a.lkey, l) || l0 >= a.n_left;       // null left keys are dropped even for left/outer (join.rs:152)
            k4[r] = key_cell(a.lkey, l);
        }
        bucket_lookup4(a.tab, k4, skip, slot4, e4, found);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            start[h + r] = NO_MATCH; cnt[h + r] = 0; direct[h + r] = false;
            if (found[r]) {
                start[h + r] = e4[r].start; cnt[h + r] = e4[r].count & ~JN_DIRECT; direct[h + r] = (e4[r].count & JN_DIRECT) != 0;
                if (a.flag_right) a.hit[slot4[r]] = 1;
            } else if (a.keep_left && !skip[r]) {
                cnt[h + r] = 1;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < OP_RPT; r++) {
        uint32_t v = cnt[r];                            // inclusive scan over the wave's lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(v, d, 64);
            if (lane >= (uint32_t)d) v += t;
        }
        inc[r] = v;
        if (lane == 63) wsum[r][wave] = v;
    }
    __syncthreads();
    uint32_t tot = 0, before[OP_RPT];                   // matches of the tile; matches before (r, wave)
#pragma unroll
    for (int r = 0; r < OP_RPT; r++) {
        before[r] = tot;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t x = wsum[r][w];
            if (w < (int)wave) before[r] += x;
            tot += x;
        }
    }
    if (tid < 64) {                                     // wave 0: publish, then look back
        unsigned long long excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&a.status[0], (2ull << 62) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&a.status[tile], (1ull << 62) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t j0 = (int64_t)tile - 1;
            for (;;) {
                const int64_t j = j0 - lane;
                unsigned long long v = 2ull << 62;      // before tile 0: inclusive prefix 0
                if (j >= 0) {
                    uint32_t spins = 0;
                    v = __hip_atomic_load(&a.status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while ((v >> 62) == 0) {
                        if (++spins > (1u << 22)) { a.ticket[1] = 1; v = 2ull << 62; break; }   // never hang the GPU
                        __builtin_amdgcn_s_sleep(2);
                        v = __hip_atomic_load(&a.status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                const unsigned long long incl = __ballot((v >> 62) == 2);
                const int first = incl ? __builtin_ctzll(incl) : 64;
                unsigned long long add = (int)lane <= first ? (v & OP_VALUE) : 0ull;
                for (int o = 32; o >= 1; o >>= 1) add += __shfl_xor(add, o, 64);
                excl += add;
                if (incl) break;
                j0 -= 64;
            }
            if (lane == 0) __hip_atomic_store(&a.status[tile], (2ull << 62) | ((excl + tot) & OP_VALUE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    const uint64_t tile_pos = s_excl;
    uint32_t first_row[OP_RPT];
#pragma unroll
    for (int r = 0; r < OP_RPT; r++)                    // the gathers of all slices in flight together
        first_row[r] = direct[r] ? start[r] : ((cnt[r] != 0 && start[r] != NO_MATCH) ? a.rrows_sorted[start[r]] : 0u);
#pragma unroll
    for (int r = 0; r < OP_RPT; r++) {
        if (cnt[r] == 0) continue;
        const int64_t l = base + (int64_t)r * OP_THREADS;
        const uint64_t pos = tile_pos + before[r] + inc[r] - cnt[r];
        if (pos + cnt[r] > a.cap) continue;
        if (start[r] == NO_MATCH) { a.out_left[pos] = l; a.out_right[pos] = -1; continue; }
        a.out_left[pos] = l; a.out_right[pos] = first_row[r];
        for (uint32_t q = 1; q < cnt[r]; q++) { a.out_left[pos + q] = l; a.out_right[pos + q] = a.rrows_sorted[start[r] + q]; }
    }
}

// right / outer: every right row of a probed run is matched
__global__ void mark_matched_kernel(const JoinEntry *table, const uint8_t *hit, uint32_t n_entries,
                                    const uint32_t *rrows_sorted, uint8_t *rmatched) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_entries || !hit[s]) return;
    const JoinEntry e = table[s];
    if (e.count & JN_DIRECT) { rmatched[e.start] = 1; return; }
    for (uint32_t j = 0; j < e.count; j++) rmatched[rrows_sorted[e.start + j]] = 1;
}

// Emit pass, in ORIGINAL left order: coalesced reads of (match, offset), coalesced index-pair writes;
// the only gather is from the sorted right-row array (n_right x 4 B, cache resident for typical builds).
// EM_RPT rows per thread so the gathers of several rows are in flight together.
constexpr int EM_THREADS = LK_THREADS, EM_RPT = LK_RPT;      // the emit tile IS the lookup tile
__global__ __launch_bounds__(EM_THREADS) void join_emit_kernel(const uint2 *match, const unsigned long long *tile_off,
                                                               const uint32_t *rrows_sorted, int64_t n_left,
                                                               int64_t *out_left, int64_t *out_right) {
    constexpr int NW = EM_THREADS / 64;
    __shared__ uint32_t wsum[EM_RPT][NW];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * (EM_THREADS * EM_RPT) + tid;
    uint2 mt[EM_RPT];
    uint32_t cntr[EM_RPT], inc[EM_RPT], first[EM_RPT];
#pragma unroll
    for (int r = 0; r < EM_RPT; r++) {
        const int64_t l = base + (int64_t)r * EM_THREADS;
        mt[r] = match[min(l, n_left - 1)];
        cntr[r] = l < n_left ? (mt[r].y & ~JN_DIRECT) : 0u;
    }
#pragma unroll
    for (int r = 0; r < EM_RPT; r++)
        first[r] = (mt[r].y & JN_DIRECT) ? mt[r].x : ((cntr[r] != 0 && mt[r].x != NO_MATCH) ? rrows_sorted[mt[r].x] : 0u);
    // positions in left-row order: row slice r (rows base + r * EM_THREADS) before slice r + 1, lanes in order inside a slice.
    // (a tile's output can exceed 2^32 only if the whole join does, which the host has already refused)
#pragma unroll
    for (int r = 0; r < EM_RPT; r++) {
        uint32_t v = cntr[r];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(v, d, 64);
            if (lane >= (uint32_t)d) v += t;
        }
        inc[r] = v;
        if (lane == 63) wsum[r][wave] = v;
    }
    __syncthreads();
    uint64_t pos0 = tile_off[blockIdx.x];
    uint32_t before[EM_RPT], run = 0;
#pragma unroll
    for (int r = 0; r < EM_RPT; r++) {
        before[r] = run;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t x = wsum[r][w];
            if (w < (int)wave) before[r] += x;
            run += x;
        }
    }
#pragma unroll
    for (int r = 0; r < EM_RPT; r++) {
        if (cntr[r] == 0) continue;
        const int64_t l = base + (int64_t)r * EM_THREADS;
        const uint64_t o = pos0 + before[r] + inc[r] - cntr[r];
        if (mt[r].x == NO_MATCH) { out_left[o] = l; out_right[o] = -1; continue; }
        out_left[o] = l; out_right[o] = first[r];
        for (uint32_t q = 1; q < cntr[r]; q++) { out_left[o + q] = l; out_right[o + q] = rrows_sorted[mt[r].x + q]; }
    }
}

__global__ void unmatched_pred_kernel(const uint8_t *rmatched, int64_t n, uint32_t *pred) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pred[i] = rmatched[i] ? 0u : 1u;
}
__global__ void append_unmatched_kernel(const uint8_t *rmatched, const uint32_t *off, int64_t n, int64_t base,
                                        int64_t *out_left, int64_t *out_right) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !rmatched[i]) { out_left[base + off[i]] = -1; out_right[base + off[i]] = i; }
}


static int32_t stage_key(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *col, int64_t n, KeyDesc *out) {
    const void *d = col->data; const uint8_t *m = col->null_mask;
    if (mem_space == PANDRS_HIP_MEM_HOST && n > 0) {
        size_t bytes = dtype_bytes(col->dtype, n);
        void *dd = c->staging.take<uint8_t>(bytes + 16);
        if (!dd) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
        HIP_TRY(hipMemcpyAsync(dd, d, bytes, hipMemcpyHostToDevice, c->stream));
        d = dd;
        if (m) {
            uint8_t *dm = c->staging.take<uint8_t>((n + 7) / 8 + 16);
            if (!dm) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "staging arena too small");
            HIP_TRY(hipMemcpyAsync(dm, m, (n + 7) / 8, hipMemcpyHostToDevice, c->stream));
            m = dm;
        }
    }
    *out = KeyDesc{d, m, nullptr, col->dtype};
    return 0;
}

static int32_t join_core(pandrs_hip_ctx *c, const KeyDesc &lkey, int64_t nl, const KeyDesc &rkey, int64_t nr, int32_t how);

int32_t join_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *lk, int64_t nl,
                   const pandrs_hip_column *rk, int64_t nr, int32_t how, int64_t *out_n) {
    if (!c || !lk || !rk || !out_n || nl < 0 || nr < 0 || how < PANDRS_HIP_JOIN_INNER || how > PANDRS_HIP_JOIN_OUTER)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: bad arguments");
    if (lk->dtype != rk->dtype)                 // join.rs:98-104
        return fail(PANDRS_HIP_ERR_TYPE_MISMATCH, "join key columns have different types (%d and %d)", lk->dtype, rk->dtype);
    if (lk->dtype < PANDRS_HIP_I64 || lk->dtype > PANDRS_HIP_CELL64)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: bad key dtype %d", lk->dtype);
    if ((nl && !lk->data) || (nr && !rk->data)) return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: key column has no data");
    if (nl >= (int64_t(1) << 32) - 16384 || nr >= (int64_t(1) << 32) - 16384)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: a side exceeds the 2^32-row per-call limit");

    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    timings_begin(c);
    KeyDesc lkey{}, rkey{};
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_STAGE_IN);
        if (mem_space == PANDRS_HIP_MEM_HOST)
            ST_TRY(c->staging.ensure(dtype_bytes(lk->dtype, nl) + dtype_bytes(rk->dtype, nr) + (nl + nr) / 8 + (1 << 16), c->stream));
        ST_TRY(stage_key(c, mem_space, lk, nl, &lkey));
        ST_TRY(stage_key(c, mem_space, rk, nr, &rkey));
    }
    ST_TRY(join_core(c, lkey, nl, rkey, nr, how));
    // B_join for index output: both key columns read once, 16 B per output row written
    int64_t K = lk->dtype == PANDRS_HIP_U32CODE ? 4 : (lk->dtype == PANDRS_HIP_BOOLBITS ? 0 : 8);
    c->timings.algorithmic_bytes = (nl + nr) * K + c->jn.n_rows * 16;
    ST_TRY(timings_end(c));
    *out_n = c->jn.n_rows;
    return 0;
}

// The join proper on device-resident key columns; the caller holds the context's mutex.
// Leaves the index pairs in c->jn (result arena).
static int32_t join_core(pandrs_hip_ctx *c, const KeyDesc &lkey, int64_t nl, const KeyDesc &rkey, int64_t nr, int32_t how) {
    const bool keep_left = how == PANDRS_HIP_JOIN_LEFT || how == PANDRS_HIP_JOIN_OUTER;
    const bool keep_right = how == PANDRS_HIP_JOIN_RIGHT || how == PANDRS_HIP_JOIN_OUTER;
    c->jn = JoinResult{};
    c->gb.valid = false;                        // the result arena is shared
    // workspace: one partition pass over the build side + the table + per-row arrays
    // table capacity: the larger of the global layout (general build path: load <= 0.77) and the regional one
    // (BH_SLOTS entries per build partition; up to P_MAX partitions after a retry); 16 B per entry
    uint32_t cap_glob = 64;
    while ((double)cap_glob < 1.3 * (double)nr) cap_glob <<= 1;
    auto partitions_for = [&](int64_t P0) { return std::min<int64_t>(P0 * 4, P_MAX); };
    int64_t P = c->opt.partitions > 0 ? c->opt.partitions
                                     : std::max<int64_t>(1, (int64_t)std::ceil((double)nr / (JN_RCAP * 0.6)));
    P = std::min<int64_t>(std::max<int64_t>(P, std::min<int64_t>(256, nr / 8192)), P_MAX);
    P = std::max<int64_t>(P, 1);
    const uint32_t cap_tab = (uint32_t)std::max<uint64_t>(cap_glob, (uint64_t)partitions_for(P) * BH_SLOTS);
    size_t ws = 2 * engine_workspace_bytes(0, 0, 0) + Arena::padded(size_t(nr + 1) * 8) + Arena::padded(size_t(nr + 1) * 4)
              + Arena::padded(size_t(cap_tab + 8) * 16) + Arena::padded(size_t(cap_tab) + 16)
              + Arena::padded(size_t(nl + 2) * 8) + 2 * Arena::padded((size_t(nl) / (LK_THREADS * LK_RPT) + 4) * 8)
              + 2 * Arena::padded(size_t(nr + 2) * 4) + Arena::padded(size_t(nr) + 8)
              + Arena::padded(scan_seg_count((size_t)nl + 1) * 4) + Arena::padded(scan_seg_count((size_t)nr + 1) * 4) + (1 << 16)
              + segsort_workspace_bytes(nr, P_MAX + 1, 4) + Arena::padded((size_t(nl) / OP_TILE + 2) * 8) + 4096;
    ST_TRY(c->work.ensure(ws, c->stream));

    uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
    int64_t M1 = 0, M2 = 0;
    // the LDS build cannot work when even the maximum fan-out leaves partitions above its capacity
    bool generic = c->opt.join_generic != 0 || (double)nr / (double)P > JN_RCAP * 0.95;
    bool onepass_failed = false;
    for (int attempt = 0;; attempt++) {
        c->work.off = 0;
        c->timings.n_partitions = P; c->timings.retries = attempt; c->timings.table_slots = JN_RCAP;
        uint32_t *flags = c->work.take<uint32_t>(64);
        uint64_t *prk = c->work.take<uint64_t>(nr + 1);
        uint32_t *prr = c->work.take<uint32_t>(nr + 1);
        JoinEntry *table = c->work.take<JoinEntry>((size_t)cap_tab + 8);       // + the sentinel-valued key's entry and the rest of its "bucket"
        uint8_t *hit = c->work.take<uint8_t>((size_t)cap_tab + 16);
        uint2 *match = c->work.take<uint2>(nl + 2);
        const uint32_t n_tiles = (uint32_t)((nl + LK_THREADS * LK_RPT - 1) / (LK_THREADS * LK_RPT));
        unsigned long long *tile_sum = c->work.take<unsigned long long>((size_t)n_tiles + 2);
        unsigned long long *tile_off = c->work.take<unsigned long long>((size_t)n_tiles + 2);
        uint8_t *rmatched = c->work.take<uint8_t>(nr + 8);
        if (!flags || !prk || !prr || !table || !hit || !match || !tile_sum || !tile_off || !rmatched)
            return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (join)");
        HIP_TRY(hipMemsetAsync(flags, 0, 256, c->stream));
        HIP_TRY(hipMemsetAsync(rmatched, 0, size_t(nr) + 8, c->stream));
        TableRef tab{table, 1u, cap_glob - 1};
        if (!generic) { tab.regions = (uint32_t)P; tab.rmask = BH_SLOTS - 1; }          // regions are written whole by their build workgroups
        const uint32_t n_entries = tab.regions * (tab.rmask + 1);
        if ((uint64_t)tab.regions * (tab.rmask + 1) > cap_tab) return fail(PANDRS_HIP_ERR_COMPUTATION, "join: table geometry exceeds its allocation");
        if (generic) HIP_TRY(hipMemsetAsync(table, 0xFF, size_t(n_entries) * 16, c->stream));   // keys = EMPTY, filled with atomics
        HIP_TRY(hipMemsetAsync(&table[n_entries], 0, 8 * 16, c->stream));              // the key ~0's own entry: count 0 (+ 7 spare entries)
        if (keep_right) HIP_TRY(hipMemsetAsync(hit, 0, size_t(n_entries) + 16, c->stream));

        // ---- build side: radix partition (null keys -> their own partition, never built), sort, publish
        PartInfo rpart{};
        ScatterArgs rs{};
        rs.key = rkey; rs.pkeys = prk; rs.n_rows = nr; rs.P = (uint32_t)P; rs.seed = JN_SEED;
        rs.mv[rs.n_move++] = MoveDesc{nullptr, prr, 3, 0};
        ST_TRY(radix_partition(c, rs, &rpart, PANDRS_HIP_PHASE_BUILD, PANDRS_HIP_PHASE_BUILD, PANDRS_HIP_PHASE_BUILD));
        {
            PhaseTimer pt(c, PANDRS_HIP_PHASE_BUILD);
            BuildArgs ba{};
            ba.rkeys = prk; ba.rrows = prr; ba.roff = rpart.offsets; ba.rNB = rpart.NB;
            ba.tab = tab; ba.flags = flags;
            const size_t lds = (size_t)BH_SLOTS * 12 + JN_RCAP * 4 + 256;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(join_build_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (!generic) {
                hipLaunchKernelGGL(join_build_kernel, dim3((unsigned)P), dim3(JN_THREADS), lds, c->stream, ba);
            } else {
                // a partition does not fit the LDS sort buffers (very many build rows, or one key with
                // thousands of duplicates): general segmented sort, runs published from the sorted array
                ST_TRY(segmented_sort_u32(c, prk, prr, rpart.offsets, rpart.NB, (uint32_t)P, nr));
                if (nr > 0)
                    hipLaunchKernelGGL(publish_runs_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream,
                                       prk, prr, rpart.offsets + (size_t)P * rpart.NB, tab);
            }
            HIP_TRY(hipGetLastError());
        }
        // ---- probe, single pass (opt-in: both probes are bound by the random table reads, ~40 G/s from
        // MALL, and measure within a few percent of each other); the result buffers are sized for a
        // unique-key build side up front and re-sized once if duplicate keys produce more rows
        if (c->opt.join_one_pass && !onepass_failed && nl > 0) {
            const uint32_t n_tiles = (uint32_t)((nl + OP_TILE - 1) / OP_TILE);
            unsigned long long *status = c->work.take<unsigned long long>((size_t)n_tiles + 1);
            uint32_t *ticket = c->work.take<uint32_t>(16);
            if (!status || !ticket) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (join)");
            uint64_t cap_out = (uint64_t)nl + (keep_right ? (uint64_t)nr : 0ull) + 16;
            uint64_t total = 0;
            bool bail = false;
            for (int pass = 0;; pass++) {
                ST_TRY(c->result.ensure(2 * Arena::padded(size_t(cap_out + 1) * 8) + 4096, c->stream));
                c->jn.left_idx = c->result.take<int64_t>(cap_out + 1);
                c->jn.right_idx = c->result.take<int64_t>(cap_out + 1);
                if (!c->jn.left_idx || !c->jn.right_idx) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "result arena too small");
                HIP_TRY(hipMemsetAsync(status, 0, ((size_t)n_tiles + 1) * 8, c->stream));
                HIP_TRY(hipMemsetAsync(ticket, 0, 64, c->stream));
                {
                    PhaseTimer pt(c, PANDRS_HIP_PHASE_PROBE);
                    OnePassArgs oa{};
                    oa.lkey = lkey; oa.n_left = nl; oa.tab = tab; oa.rrows_sorted = prr;
                    oa.keep_left = keep_left ? 1 : 0; oa.flag_right = keep_right ? 1 : 0; oa.hit = hit;
                    oa.status = status; oa.ticket = ticket; oa.cap = cap_out - (keep_right ? (uint64_t)nr : 0ull);
                    oa.out_left = c->jn.left_idx; oa.out_right = c->jn.right_idx;
                    hipLaunchKernelGGL(join_probe_onepass_kernel, dim3(n_tiles), dim3(OP_THREADS), 0, c->stream, oa);
                    HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipMemcpyAsync(h, flags, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipMemcpyAsync(h + 1, ticket + 1, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipMemcpyAsync(h + 2, status + (n_tiles - 1), 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (h[0] || h[1]) { bail = true; break; }
                total = ((uint64_t)h[2] | ((uint64_t)h[3] << 32)) & OP_VALUE;
                if (total + (uint64_t)nr >= (1ull << 32) - 16384)
                    return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: the output exceeds the 2^32-row per-call limit");
                if (total + (keep_right ? (uint64_t)nr : 0ull) <= cap_out || pass > 0) break;
                cap_out = total + (keep_right ? (uint64_t)nr : 0ull) + 16;     // duplicate build keys: exact size, probe again
                if (keep_right) HIP_TRY(hipMemsetAsync(hit, 0, size_t(n_entries) + 16, c->stream));
            }
            if (bail) {
                if (h[0]) {     // a build partition overflowed (see below)
                    if (attempt == 0 && P < P_MAX) P = std::min<int64_t>(P * 4, P_MAX);
                    else generic = true;
                } else {
                    onepass_failed = true;      // look-back timed out: use the three-kernel probe
                }
                continue;
            }
            M1 = (int64_t)total;
            if (keep_right && nr > 0) {
                hipLaunchKernelGGL(mark_matched_kernel, dim3((n_entries + 2 + 255) / 256), dim3(256), 0, c->stream,
                                   table, hit, n_entries + 2, prr, rmatched);
                uint32_t *pred = c->work.take<uint32_t>(nr + 2);
                uint32_t *roff2 = c->work.take<uint32_t>(nr + 2);
                uint32_t *seg2 = c->work.take<uint32_t>(scan_seg_count((size_t)nr + 1));
                if (!pred || !roff2 || !seg2) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (join)");
                HIP_TRY(hipMemsetAsync(pred + nr, 0, 8, c->stream));
                hipLaunchKernelGGL(unmatched_pred_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream, rmatched, nr, pred);
                ST_TRY(exclusive_scan_u32(c, pred, (size_t)nr + 1, roff2, seg2));
                hipLaunchKernelGGL(append_unmatched_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream,
                                   rmatched, roff2, nr, M1, c->jn.left_idx, c->jn.right_idx);
                HIP_TRY(hipMemcpyAsync(h, roff2 + nr, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                M2 = h[0];
            }
            c->jn.n_rows = M1 + M2;
            c->jn.valid = true;
            break;
        }
        // ---- probe in original left order, scan of the per-row output counts
        {
            PhaseTimer pt(c, PANDRS_HIP_PHASE_PROBE);
            if (nl > 0) {
                hipLaunchKernelGGL(join_lookup_kernel, dim3((unsigned)((nl + LK_THREADS * LK_RPT - 1) / (LK_THREADS * LK_RPT))), dim3(LK_THREADS), 0, c->stream,
                                   lkey, nl, tab, keep_left ? 1 : 0, keep_right ? 1 : 0, match, tile_sum, hit);
            }
            // output position of every tile (64-bit: flags[2..3] = rows from the probe, checked against the 2^32-row limit below)
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, tile_sum, n_tiles, tile_off,
                               reinterpret_cast<unsigned long long *>(flags + 2));
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(h, flags, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h[0]) {
            // first overflow: more partitions (unlucky hashing); second: the general sort handles any size
            if (attempt == 0 && P < P_MAX) P = std::min<int64_t>(P * 4, P_MAX);
            else generic = true;
            continue;
        }
        if (((uint64_t)h[2] | ((uint64_t)h[3] << 32)) + (uint64_t)nr >= (1ull << 32) - 16384)
            return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: the output exceeds the 2^32-row per-call limit");
        M1 = (int64_t)((uint64_t)h[2] | ((uint64_t)h[3] << 32));
        // ---- right / outer: unmatched right rows, ascending
        uint32_t *roff2 = nullptr;
        if (keep_right && nr > 0) {
            hipLaunchKernelGGL(mark_matched_kernel, dim3((n_entries + 2 + 255) / 256), dim3(256), 0, c->stream,
                               table, hit, n_entries + 2, prr, rmatched);
            uint32_t *pred = c->work.take<uint32_t>(nr + 2);
            roff2 = c->work.take<uint32_t>(nr + 2);
            uint32_t *seg2 = c->work.take<uint32_t>(scan_seg_count((size_t)nr + 1));
            if (!pred || !roff2 || !seg2) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (join)");
            HIP_TRY(hipMemsetAsync(pred + nr, 0, 8, c->stream));
            hipLaunchKernelGGL(unmatched_pred_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream, rmatched, nr, pred);
            ST_TRY(exclusive_scan_u32(c, pred, (size_t)nr + 1, roff2, seg2));
            HIP_TRY(hipMemcpyAsync(h, roff2 + nr, 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            M2 = h[0];
        }
        const int64_t M = M1 + M2;
        ST_TRY(c->result.ensure(2 * Arena::padded(size_t(M + 1) * 8) + 4096, c->stream));
        c->jn.left_idx = c->result.take<int64_t>(M + 1);
        c->jn.right_idx = c->result.take<int64_t>(M + 1);
        if (!c->jn.left_idx || !c->jn.right_idx) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "result arena too small");
        {
            PhaseTimer pt(c, PANDRS_HIP_PHASE_PROBE);
            if (M1 > 0)
                hipLaunchKernelGGL(join_emit_kernel, dim3((unsigned)((nl + EM_THREADS * EM_RPT - 1) / (EM_THREADS * EM_RPT))), dim3(EM_THREADS), 0, c->stream,
                                   match, tile_off, prr, nl, c->jn.left_idx, c->jn.right_idx);
            if (M2 > 0)
                hipLaunchKernelGGL(append_unmatched_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream,
                                   rmatched, roff2, nr, M1, c->jn.left_idx, c->jn.right_idx);
            HIP_TRY(hipGetLastError());
        }
        c->jn.n_rows = M;
        c->jn.valid = true;
        break;
    }
    return 0;
}

// ================================================================================================
// Fused inner join -> groupby(right payload g).sum(left payload v)   (BASELINE config 5)
// Equivalent to inner_join (join.rs:32) + group_by(g).aggregate([(v, Sum)]) (aggregation.rs:763)
// without materialising the join rows in reference order: matches are emitted partition by
// partition as (g, v) pairs (coalesced), then fed to the groupby engine.  Null g / null v become
// 0 / 0.0 and are NOT null afterwards, exactly as the reference's gathers fill them
// (join.rs:304-307, :319-322).
// ================================================================================================
struct FusedArgs {
    const uint64_t *rkeys, *rpay;   // partitioned build side: key cells, group payload g
    const uint64_t *lkeys, *lpay;   // partitioned probe side: key cells, value payload v
    const uint32_t *roff, *loff;
    uint32_t rNB, lNB, P;
    unsigned long long *cursor;     // pairs emitted so far (keeps counting past `cap`)
    uint64_t cap;                   // capacity of out_g / out_v
    uint64_t *out_g, *out_v;
    uint32_t *flags;
    uint32_t l_limit;               // > 0: a partition with more probe rows than this raises flags[5] and is skipped — one workgroup would
                                    // walk a hot probe key's rows alone (31 M rows: 46 ms); the host takes the L2-region path instead
};

constexpr uint32_t FJ_SLOTS = 8192;                 // LDS multimap slots per partition
constexpr uint32_t FJ_BUCKETS = FJ_SLOTS / 4;       // 4-slot buckets: one 32-byte LDS read compares 4 keys
constexpr uint32_t FJ_MAXROWS = FJ_SLOTS * 7 / 8;   // build rows a partition may hold
constexpr int FJ_RPT = 8;

// One workgroup per partition.  LDS: keys[FJ_SLOTS] u64 | g[FJ_SLOTS] u64 | fill[FJ_BUCKETS] u32.
// Build: a right row takes the next slot of its bucket (fill counter, so no key value is reserved
// and duplicate keys simply take several slots); a full bucket spills to the next one, and the
// counter keeps counting so that fill > 4 tells a reader to walk on.  Probe: a left row compares
// the 4 keys of its bucket at once and walks on only past an over-full bucket — the walk length is
// nearly uniform across a wave, unlike linear probing.  A sum does not depend on the order of its
// terms, so pairs are appended at a cursor reserved with one atomic per workgroup and
// FJ_RPT x 1024 rows — no count pass, no sort.
__global__ __launch_bounds__(JN_THREADS) void fused_probe_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t rbeg = a.roff[(size_t)p * a.rNB], rend = a.roff[(size_t)(p + 1) * a.rNB];
    const uint32_t lbeg = a.loff[(size_t)p * a.lNB], lend = a.loff[(size_t)(p + 1) * a.lNB];
    const uint32_t nR = rend - rbeg;
    if (nR > FJ_MAXROWS) { if (tid == 0) a.flags[0] = 1; return; }
    if (lbeg == lend || nR == 0) return;
    if (a.l_limit && lend - lbeg > a.l_limit) { if (tid == 0) a.flags[5] = 1; return; }
    uint64_t *sk = reinterpret_cast<uint64_t *>(smem);
    uint64_t *sg = sk + FJ_SLOTS;
    uint32_t *fill = reinterpret_cast<uint32_t *>(sg + FJ_SLOTS);
    uint32_t *wt = fill + FJ_BUCKETS;                                    // [0] pairs of the current iteration
    unsigned long long *s_base = reinterpret_cast<unsigned long long *>(wt + 2);
    uint32_t *wsum = wt + 4;                                             // FJ_RPT x 16 wave totals
    for (uint32_t i = tid; i < FJ_BUCKETS; i += JN_THREADS) fill[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < nR; i += JN_THREADS) {
        const uint64_t k = a.rkeys[rbeg + i], g = a.rpay[rbeg + i];
        // two-choice placement: the emptier of the key's two buckets, then the other one, and only when
        // both are full the buckets after the second (rare even at 90 % load)
        const uint32_t b1 = hash32(k, 0x7F4A7C15u) & (FJ_BUCKETS - 1), b2 = hash32(k, 0x165667B1u) & (FJ_BUCKETS - 1);
        const bool second_first = fill[b2] < fill[b1];
        uint32_t b = second_first ? b2 : b1;
        uint32_t c = atomicAdd(&fill[b], 1u);
        if (c >= 4) { b = second_first ? b1 : b2; c = atomicAdd(&fill[b], 1u); }
        if (c >= 4) {
            b = b2;
            do { b = (b + 1) & (FJ_BUCKETS - 1); c = atomicAdd(&fill[b], 1u); } while (c >= 4);
        }
        sk[b * 4 + c] = k; sg[b * 4 + c] = g;
    }
    __syncthreads();
    // every slot holding `key`, in a fixed order: the key's two buckets, then — only if both ever turned
    // a row away (fill > 4) — the buckets behind the second, up to the first one that never did
    auto visit_bucket = [&](uint32_t b, uint64_t key, auto &&fn) -> uint32_t {
        const uint32_t c = fill[b];
        const ulonglong2 lo = *reinterpret_cast<const ulonglong2 *>(&sk[b * 4]);
        const ulonglong2 hi = *reinterpret_cast<const ulonglong2 *>(&sk[b * 4 + 2]);
        if (c > 0 && lo.x == key) fn(b * 4);
        if (c > 1 && lo.y == key) fn(b * 4 + 1);
        if (c > 2 && hi.x == key) fn(b * 4 + 2);
        if (c > 3 && hi.y == key) fn(b * 4 + 3);
        return c;
    };
    auto visit_matches = [&](uint64_t key, auto &&fn) {
        const uint32_t b1 = hash32(key, 0x7F4A7C15u) & (FJ_BUCKETS - 1), b2 = hash32(key, 0x165667B1u) & (FJ_BUCKETS - 1);
        const uint32_t c1 = visit_bucket(b1, key, fn);
        uint32_t c2 = c1;
        if (b2 != b1) c2 = visit_bucket(b2, key, fn);
        if (c1 > 4 && c2 > 4) {
            uint32_t b = b2, c;
            do {
                b = (b + 1) & (FJ_BUCKETS - 1);
                c = (b == b1) ? 5u : visit_bucket(b, key, fn);     // b1 was visited already; it is over-full, walk on
            } while (c > 4);
        }
    };
    const uint32_t n_iter = (lend - lbeg + JN_THREADS * FJ_RPT - 1) / (JN_THREADS * FJ_RPT);
    for (uint32_t it = 0; it < n_iter; it++) {
        const uint32_t i0 = lbeg + it * (JN_THREADS * FJ_RPT) + tid;
        uint64_t k[FJ_RPT], v[FJ_RPT];
#pragma unroll
        for (int r = 0; r < FJ_RPT; r++) {
            const uint32_t i = min(i0 + (uint32_t)r * JN_THREADS, lend - 1);
            k[r] = __builtin_nontemporal_load(&a.lkeys[i]);
            v[r] = __builtin_nontemporal_load(&a.lpay[i]);
        }
        uint32_t m[FJ_RPT], first[FJ_RPT];
#pragma unroll
        for (int r = 0; r < FJ_RPT; r++) {
            m[r] = 0; first[r] = 0;
            if (i0 + (uint32_t)r * JN_THREADS >= lend) continue;
            visit_matches(k[r], [&](uint32_t slot) { if (m[r] == 0) first[r] = slot; m[r]++; });
        }
        // compaction of the workgroup's pairs, row-slice major: the pairs of slice r (rows i0 + r * 1024 + tid)
        // come before those of slice r + 1, and inside a slice in thread order — consecutive lanes write
        // consecutive pairs (coalesced stores; thread-major order would scatter every lane's 8 pairs 64 B
        // apart).  ONE global atomic per workgroup and FJ_RPT x 1024 rows (same-address atomics retire at
        // well under 100 M/s on this chip: never one per wave).
        constexpr int NW = JN_THREADS / 64;
        const uint32_t lane = tid & 63, wave = tid >> 6;
        uint32_t inc[FJ_RPT];
#pragma unroll
        for (int r = 0; r < FJ_RPT; r++) {
            uint32_t x = m[r];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(x, d, 64);
                if (lane >= (uint32_t)d) x += t;
            }
            inc[r] = x;
            if (lane == 63) wsum[r * NW + wave] = x;
        }
        __syncthreads();
        if (tid < 64) {                                 // exclusive scan of the FJ_RPT x NW (= 128) wave totals, 2 per lane
            const uint32_t a0 = wsum[2 * lane], a1 = wsum[2 * lane + 1];
            uint32_t x = a0 + a1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(x, d, 64);
                if (lane >= (uint32_t)d) x += t;
            }
            const uint32_t tot = __shfl(x, 63, 64);
            wsum[2 * lane] = x - a0 - a1; wsum[2 * lane + 1] = x - a1;
            if (lane == 0) { wt[0] = tot; *s_base = tot ? atomicAdd(a.cursor, (unsigned long long)tot) : 0ull; }
        }
        __syncthreads();
        const uint32_t tot = wt[0];
        const unsigned long long base = *s_base;
        uint32_t before[FJ_RPT];
#pragma unroll
        for (int r = 0; r < FJ_RPT; r++) before[r] = wsum[r * NW + wave];
        __syncthreads();                                // wsum / wt / s_base are rewritten by the next iteration
        if (tot == 0 || base + tot > a.cap) continue;   // uniform; on overflow the host grows the buffer and runs the pass again
#pragma unroll
        for (int r = 0; r < FJ_RPT; r++) {
            if (m[r] == 0) continue;
            uint64_t pos = base + before[r] + inc[r] - m[r];
            if (m[r] == 1) { a.out_g[pos] = sg[first[r]]; a.out_v[pos] = v[r]; continue; }
            const uint64_t vr = v[r];                   // a duplicated build key: visit its slots again
            visit_matches(k[r], [&](uint32_t slot) { a.out_g[pos] = sg[slot]; a.out_v[pos] = vr; pos++; });
        }
    }
}

// ================================================================================================
// The fused path for LARGE build sides (more than ~1024 LDS-sized partitions, C5: 50 M build rows).
// The LDS multimap above needs the PROBE side partitioned as finely as the build side, and at 8192
// partitions an 8192-row scatter tile leaves as 1-row runs (500 M probe rows: 12.4 ms for 16 GB moved).
// Here only the build side is partitioned finely (P_f).  One workgroup per fine partition assembles
// an open-addressing region of 16-byte {key, g} entries in LDS and stores it whole (no global
// atomics); 8 consecutive regions (1 MiB) belong to one COARSE partition c = f / 8.  The probe side
// is partitioned by c only (P_c = P_f / 8 <= 1024: 8-row runs, capacity mode without a histogram
// pass), and coarse partition c is probed by the workgroups of ONE XCD group (blockIdx % 8), whose L2
// then holds c's regions: random 16-byte lookups out of a 1-3 MiB region run at 130-155 G/s, out of
// MALL / HBM at 40-57 G/s (experiments/ubench/l2_probe.hip).  The XCD mapping is a matter of speed,
// never of correctness.  Unique build keys only (the primary-key side of a PK-FK join): the build
// raises a flag on a duplicate and the caller takes the LDS-multimap path instead.
// ================================================================================================
struct __attribute__((aligned(16))) L2Entry { uint64_t key, g; };
constexpr uint32_t L2_REG = 8192;                    // entries per fine region (128 KB: what one workgroup can assemble in LDS)
constexpr uint32_t L2_MAXROWS = L2_REG / 8 * 7;
constexpr uint32_t L2_SEED = 0x3C6EF372u;
constexpr int L2_FINE_PER_COARSE = 8;

struct L2BuildArgs {
    const uint64_t *rkeys, *rpay;                    // partitioned build side
    const uint32_t *roff; uint32_t rNB;
    L2Entry *table;                                  // [P_f * sub * L2_REG] + 1 entry for the sentinel-valued key
    uint32_t P_f, sub;                               // sub = 1 or 2 regions per fine partition (by one more hash bit): load <= ~0.4,
                                                     // linear-probe chains stay short (at 0.75 a wave waits for 30+ dependent reads)
    uint32_t *flags;                                 // [0] a partition does not fit, [1] duplicate build key, [4] sentinel-valued key present
};

__global__ __launch_bounds__(JN_THREADS) void fused_l2_build_kernel(L2BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t rbeg = a.roff[(size_t)p * a.rNB], rend = a.roff[(size_t)(p + 1) * a.rNB];
    const uint32_t nR = rend - rbeg;
    L2Entry *reg = a.table + (size_t)p * a.sub * L2_REG;
    if (nR == 0 || nR > L2_MAXROWS) {
        for (uint32_t s = tid; s < a.sub * L2_REG; s += JN_THREADS) reg[s] = L2Entry{EMPTY_KEY, 0ull};
        if (nR != 0 && tid == 0) a.flags[0] = 1;
        return;
    }
    uint64_t *sk = reinterpret_cast<uint64_t *>(smem);
    uint64_t *sg = sk + L2_REG;
    for (uint32_t sub = 0; sub < a.sub; sub++) {
        for (uint32_t s = tid; s < L2_REG; s += JN_THREADS) sk[s] = EMPTY_KEY;
        __syncthreads();
        for (uint32_t i = tid; i < nR; i += JN_THREADS) {
            const uint64_t k = a.rkeys[rbeg + i], g = a.rpay[rbeg + i];
            if (k == EMPTY_KEY) {                         // the sentinel-valued key lives behind the last region
                if (sub == 0) {
                    if (atomicExch(&a.flags[4], 1u)) a.flags[1] = 1;
                    else a.table[(size_t)a.P_f * a.sub * L2_REG].g = g;
                }
                continue;
            }
            const uint32_t h = hash32(k, L2_SEED);
            if (((h >> 13) & (a.sub - 1)) != sub) continue;
            uint32_t s = h & (L2_REG - 1);
            for (;;) {
                const uint64_t old = atomicCAS((unsigned long long *)&sk[s], EMPTY_KEY, k);
                if (old == EMPTY_KEY) { sg[s] = g; break; }
                if (old == k) { a.flags[1] = 1; break; }             // duplicate build key: not this path
                s = (s + 1) & (L2_REG - 1);
            }
        }
        __syncthreads();
        for (uint32_t s = tid; s < L2_REG; s += JN_THREADS) reg[(size_t)sub * L2_REG + s] = L2Entry{sk[s], sg[s]};
        __syncthreads();
    }
}

constexpr int L2_BATCH = 1;      // (4 tiles per ticket measured slower: the group spreads over more partitions, 7.0 -> 7.8 ms)
// 256-thread workgroups, four per CU: the tile is a chain of dependent round trips (rows, entries, walk, cursor), other tiles fill
// the gaps.  MODE 2 takes 512 threads (4096-row tiles, two per CU): half the reservations and barriers per row.
constexpr int L2_THREADS = 256, L2_THREADS_PART = 512, L2_RPT = 8;
struct L2ProbeArgs {
    const uint64_t *lkeys, *lpay;                    // probe side, partitioned by COARSE partition
    const uint32_t *loff; uint32_t lNB;              // exact layout, or (gbeg != nullptr) the capacity layout's 8 ranges per partition
    const uint32_t *gbeg, *gcur, *gend;
    uint32_t P_c, P_f, sub;
    uint32_t *ticket;                                // [8] zeroed: next tile of XCD group g
    uint32_t ablate;                                 // experiments: 1 = no pair output, 2 = no walk either
    const L2Entry *table;
    const uint32_t *flags;
    unsigned long long *cursor; uint64_t cap;
    uint64_t *out_g, *out_v;
    // MODE 1 (sample) / MODE 2 (pairs written straight into the groupby engine's capacity layout, partitioned by g)
    uint32_t pair_P, pair_seed;                      // pair partition = part_of(hash32(g, pair_seed), pair_P), <= L2_PAIR_PMAX
    uint32_t sample_stride;                          // MODE 1 looks at the first tile of every sample_stride
    uint32_t *pair_hist;                             // MODE 1: [SAMPLE_REPL][pair_P + 2] like sample_histogram_kernel's
    uint32_t *pair_cur; const uint32_t *pair_end;    // MODE 2: the plan's cursors / region ends, [pair_P + 1][8]
    uint32_t pair_cap;                               // rows of the regions; [pair_cap, pair_cap + tile) is the trash tile
    uint32_t *pair_flags;                            // [0] a region overflowed
};
constexpr uint32_t L2_PAIR_PMAX = 512;

// MODE 0: pairs appended at one global cursor (any order).  MODE 1: no output — a histogram of the pair partitions over 1 tile in
// 64, from which plan_regions_kernel sizes the regions.  MODE 2: every tile's pairs are ranked by pair partition in LDS and
// appended as contiguous runs to region (partition, XCD group) — the capacity layout aggregate2 reads — so the pairs are
// written once, already partitioned (MODE 0 + the engine's own scatter writes, reads, writes and reads them).
template <int MODE, int TH>
__global__ __launch_bounds__(TH, 4) void fused_l2_probe_kernel(L2ProbeArgs a) {
    constexpr int TILE = TH * L2_RPT;
    constexpr int NW = TH / 64;
    __shared__ uint32_t wsum[L2_RPT * NW];
    __shared__ uint32_t s_tot;
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t grp = blockIdx.x & 7;
    __shared__ uint32_t pcnt[MODE ? L2_PAIR_PMAX + 1 : 1], pdelta[MODE == 2 ? L2_PAIR_PMAX + 1 : 1], pwt[17];
    // MODE 2: dynamic LDS = (g, v) records of the tile in sorted order [TILE] x 16 B | their pair partitions [TILE] x u16
    extern __shared__ __attribute__((aligned(16))) unsigned char l2_dyn[];
    ulonglong2 *pstage = reinterpret_cast<ulonglong2 *>(l2_dyn);
    uint16_t *ppid = reinterpret_cast<uint16_t *>(l2_dyn + (MODE == 2 ? (size_t)TILE * 16 : 0));
    uint32_t rows_seen = 0;                            // MODE 1
    unsigned long long pairs_mine = 0;                 // MODE 2 (thread 0): pairs this workgroup wrote
    if (MODE == 1) {
        for (uint32_t p = tid; p <= a.pair_P; p += TH) pcnt[p] = 0;
        __syncthreads();
    }
    const bool have_sentinel = a.flags[4] != 0;
    const uint64_t sentinel_g = a.table[(size_t)a.P_f * a.sub * L2_REG].g;
    // The group's tiles, in the order (partition c0: tile 0, 1, ..), (c0 + 8: ..), ..., are handed out by a ticket per group:
    // every workgroup always takes the NEXT tile, so the group as a whole works on one or two partitions at a time whatever
    // the speed of its members (a static deal lets slow workgroups fall partitions behind and the L2 holds none of them).
    uint32_t c = grp, acc = 0;                         // acc = tiles of the group's partitions before c
    uint32_t sb[8], nt[8], se[8], total = 0;
    auto load_partition = [&]() {                      // the row ranges of coarse partition c and their tiles (wave-uniform)
        total = 0;
        if (c >= a.P_c) return;
        uint32_t n_seg;
        if (a.gbeg) {
            n_seg = 8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                sb[j] = a.gbeg[c * 8 + j];
                se[j] = max(min(a.gcur[c * 8 + j], a.gend[c * 8 + j]), sb[j]);
            }
        } else {
            n_seg = 1;
            sb[0] = a.loff[(size_t)c * a.lNB]; se[0] = a.loff[(size_t)(c + 1) * a.lNB];
#pragma unroll
            for (int j = 1; j < 8; j++) { sb[j] = 0; se[j] = 0; }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { nt[j] = (uint32_t)j < n_seg ? (se[j] - sb[j] + TILE - 1) / TILE : 0u; total += nt[j]; }
    };
    load_partition();
    uint32_t next_tile = 0, batch_left = 0;
    for (;;) {
        {
            // one ticket = L2_BATCH consecutive tiles (a returning device-scope atomic is a ~2 us round trip)
            if (batch_left == 0) {
                if (tid == 0) s_tot = atomicAdd(&a.ticket[grp], MODE == 1 ? a.sample_stride : (uint32_t)L2_BATCH);
                __syncthreads();
                next_tile = s_tot;
                __syncthreads();
                batch_left = L2_BATCH;
            }
            const uint32_t ticket = next_tile++;
            batch_left--;
            while (c < a.P_c && ticket >= acc + total) { acc += total; c += 8; load_partition(); }
            if (c >= a.P_c) break;
            const uint32_t tau = ticket - acc;
            uint32_t t = tau, beg = 0, end = 0;
            bool found = false;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (!found && t < nt[j]) { beg = sb[j] + t * TILE; end = se[j]; found = true; }
                if (!found) t -= nt[j];
            }
            const uint32_t i0 = beg + tid;
            uint64_t k[L2_RPT], v[L2_RPT];
#pragma unroll
            for (int r = 0; r < L2_RPT; r++) {
                const uint32_t i = min(i0 + (uint32_t)r * TH, end - 1);
                k[r] = __builtin_nontemporal_load(&a.lkeys[i]);
                v[r] = __builtin_nontemporal_load(&a.lpay[i]);
            }
            uint32_t slot[L2_RPT], rb[L2_RPT];
            L2Entry e[L2_RPT];
#pragma unroll
            for (int r = 0; r < L2_RPT; r++) {
                const uint32_t h = hash32(k[r], L2_SEED);
                rb[r] = (part_of(hash32(k[r], JN_SEED), a.P_f) * a.sub + ((h >> 13) & (a.sub - 1))) * L2_REG;
                slot[r] = h & (L2_REG - 1);
            }
#pragma unroll
            for (int r = 0; r < L2_RPT; r++) e[r] = a.table[(size_t)rb[r] + slot[r]];
            uint32_t m[L2_RPT], walk = 0;              // walk bit r: slice r has seen neither its key nor an empty entry yet
            uint64_t gv[L2_RPT];
#pragma unroll
            for (int r = 0; r < L2_RPT; r++) {
                const bool live = i0 + (uint32_t)r * TH < end;
                const bool is_sentinel = k[r] == EMPTY_KEY;
                const bool hit = live && (is_sentinel ? have_sentinel : e[r].key == k[r]);
                m[r] = hit ? 1u : 0u;
                gv[r] = is_sentinel ? sentinel_g : e[r].g;
                if (live && !is_sentinel && !hit && e[r].key != EMPTY_KEY) walk |= 1u << r;
            }
            // the walk, all slices in step (wave-uniform loop: the next entries of every walking slice are in flight together).
            // (A per-slice `for (; e.key != k && e.key != EMPTY; ) e = next` walk in front of the ballots below lost every row it
            // found past its home slot with hipcc 7.2 -O3; with two atomics added behind it, it did not.  This form has no divergent loop.)
            for (uint32_t probes = 0; a.ablate < 2 && __any(walk != 0) && probes < L2_REG; probes++) {
#pragma unroll
                for (int r = 0; r < L2_RPT; r++)
                    if ((walk >> r) & 1u) {
                        slot[r] = (slot[r] + 1) & (L2_REG - 1);
                        e[r] = a.table[(size_t)rb[r] + slot[r]];
                    }
#pragma unroll
                for (int r = 0; r < L2_RPT; r++)
                    if ((walk >> r) & 1u) {
                        if (e[r].key == k[r]) { m[r] = 1; gv[r] = e[r].g; walk &= ~(1u << r); }
                        else if (e[r].key == EMPTY_KEY) walk &= ~(1u << r);
                    }
            }
            if (a.ablate) {
                uint64_t x = 0;
#pragma unroll
                for (int r = 0; r < L2_RPT; r++) x += m[r] ? gv[r] ^ v[r] : 0ull;
                if (x == 0x123456789ull) a.out_g[0] = x;
                continue;
            }
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < L2_RPT; r++) {
                    if (i0 + (uint32_t)r * TH < end) rows_seen++;
                    if (m[r]) atomicAdd(&pcnt[part_of(hash32(gv[r], a.pair_seed), a.pair_P)], 1u);
                }
                continue;
            }
            if (MODE == 2) {
                const uint32_t PP1 = a.pair_P + 1;
                for (uint32_t p = tid; p < PP1; p += TH) pcnt[p] = 0;
                __syncthreads();
                uint32_t ps[L2_RPT], mm = 0;               // (pair partition << 16 | position in the tile's sorted order); match bits
#pragma unroll
                for (int r = 0; r < L2_RPT; r++) {
                    const uint32_t pp = part_of(hash32(gv[r], a.pair_seed), a.pair_P);
                    ps[r] = pp << 16;
                    if (m[r]) { mm |= 1u << r; ps[r] |= atomicAdd(&pcnt[pp], 1u); }
                }
                __syncthreads();
                {                                          // exclusive scan of pcnt -> pdelta (tile-local partition starts)
                    constexpr uint32_t IPT = (L2_PAIR_PMAX + 1 + TH - 1) / TH;
                    const uint32_t first = tid * IPT;
                    uint32_t sum = 0;
#pragma unroll
                    for (uint32_t q = 0; q < IPT; q++) if (first + q < PP1) sum += pcnt[first + q];
                    uint32_t tot;
                    uint32_t ex = block_exclusive_scan<TH>(sum, pwt, &tot);
#pragma unroll
                    for (uint32_t q = 0; q < IPT; q++) if (first + q < PP1) { pdelta[first + q] = ex; ex += pcnt[first + q]; }
                    if (tid == 0) { s_tot = tot; pairs_mine += tot; }
                }
                __syncthreads();
                const uint32_t tot = s_tot;
#pragma unroll
                for (int r = 0; r < L2_RPT; r++) ps[r] += pdelta[ps[r] >> 16];
                __syncthreads();
                // region reservation: one device-scope atomic per partition with pairs in this tile (<= pair_P + 1 per tile)
                for (uint32_t p = tid; p < PP1; p += TH) {
                    const uint32_t n = pcnt[p];
                    uint32_t c0 = n ? atomicAdd(&a.pair_cur[p * 8 + grp], n) : 0u;
                    if (n && c0 + n > a.pair_end[p * 8 + grp]) { a.pair_flags[0] = 1; c0 = a.pair_cap + pdelta[p]; }   // the trash tile
                    pdelta[p] = c0 - pdelta[p];            // destination = pdelta[partition] + sorted position
                }
#pragma unroll
                for (int r = 0; r < L2_RPT; r++)
                    if ((mm >> r) & 1u) { pstage[ps[r] & 0xFFFFu] = make_ulonglong2(gv[r], v[r]); ppid[ps[r] & 0xFFFFu] = (uint16_t)(ps[r] >> 16); }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < L2_RPT; r++) {          // one staging round for both columns (16-byte records)
                    const uint32_t j = (uint32_t)r * TH + tid;
                    if (j < tot) {
                        const ulonglong2 rec = pstage[j];
                        const uint32_t dst = pdelta[ppid[j]] + j;
                        __builtin_nontemporal_store(rec.x, &a.out_g[dst]); __builtin_nontemporal_store(rec.y, &a.out_v[dst]);
                    }
                }
                __syncthreads();
                continue;
            }
            // compaction of the tile's pairs, row-slice major (as in fused_probe_kernel): one global atomic per tile
            uint32_t inc[L2_RPT];
#pragma unroll
            for (int r = 0; r < L2_RPT; r++) {
                const unsigned long long bal = __ballot(m[r] != 0);
                inc[r] = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));           // exclusive inside the wave
                if (lane == 0) wsum[r * NW + wave] = (uint32_t)__popcll(bal);
            }
            __syncthreads();
            if (tid < 64) {                            // exclusive scan of the L2_RPT x NW (<= 64) wave totals
                const uint32_t a0 = lane < L2_RPT * NW ? wsum[lane] : 0u;
                uint32_t x = a0;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t tt = __shfl_up(x, d, 64);
                    if (lane >= (uint32_t)d) x += tt;
                }
                const uint32_t tot = __shfl(x, 63, 64);
                if (lane < L2_RPT * NW) wsum[lane] = x - a0;
                if (lane == 0) { s_tot = tot; s_base = tot ? atomicAdd(a.cursor, (unsigned long long)tot) : 0ull; }
            }
            __syncthreads();
            const uint32_t tot = s_tot;
            const unsigned long long base = s_base;
            uint32_t before[L2_RPT];
#pragma unroll
            for (int r = 0; r < L2_RPT; r++) before[r] = wsum[r * NW + wave];
            __syncthreads();
            if (tot == 0 || base + tot > a.cap) continue;
#pragma unroll
            for (int r = 0; r < L2_RPT; r++) {
                if (!m[r]) continue;
                const uint64_t pos = base + before[r] + inc[r];
                // streamed out: must not evict the table regions this XCD's L2 is holding
                __builtin_nontemporal_store(gv[r], &a.out_g[pos]); __builtin_nontemporal_store(v[r], &a.out_v[pos]);
            }
        }
    }
    if (MODE == 1) {
        __syncthreads();
        uint32_t *myhist = a.pair_hist + (size_t)(blockIdx.x % SAMPLE_REPL) * (a.pair_P + 2);
        for (uint32_t p = tid; p <= a.pair_P; p += TH) if (pcnt[p]) atomicAdd(&myhist[p], pcnt[p]);
        uint32_t w = rows_seen;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) w += __shfl_down(w, d, 64);
        if (lane == 0 && w) atomicAdd(&myhist[a.pair_P + 1], w);
    }
    if (MODE == 2 && tid == 0 && pairs_mine) atomicAdd(a.cursor, pairs_mine);
}

// General fallback of the fused path: (g, v) pairs from materialised join indices, with the
// reference's gather fill (null => 0, join.rs:304-307, :319-322); u32 group codes widened.
__global__ void pairs_from_indices_kernel(const int64_t *li, const int64_t *ri, int64_t n, KeyDesc g, int g_is_u32,
                                          KeyDesc v, uint64_t *out_g, uint64_t *out_v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t l = li[i], r = ri[i];
    const uint64_t gv = g_is_u32 ? (uint64_t)reinterpret_cast<const uint32_t *>(g.data)[r] : reinterpret_cast<const uint64_t *>(g.data)[r];
    out_g[i] = (g.null_bits && bit_at(g.null_bits, r)) ? 0ull : gv;
    out_v[i] = (v.null_bits && bit_at(v.null_bits, l)) ? 0ull : reinterpret_cast<const uint64_t *>(v.data)[l];
}

// payload with the reference's gather fill: null => 0 (join.rs:304-307); u32 sources widened
__global__ void clean_payload_kernel(const void *src, const uint8_t *null_bits, int is_u32, int64_t n, uint64_t *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t v = is_u32 ? (uint64_t)reinterpret_cast<const uint32_t *>(src)[i] : reinterpret_cast<const uint64_t *>(src)[i];
    out[i] = (null_bits && bit_at(null_bits, i)) ? 0ull : v;
}

constexpr int32_t FUSED_L2_NOT_TAKEN = -1001;
// The large-build fused path (see fused_l2_build_kernel).  Leaves the (g, v) pairs in c->pairs.
static int32_t fused_l2_path(pandrs_hip_ctx *c, const KeyDesc &lkey, const void *vsrc, int64_t nl, const KeyDesc &rkey,
                             const void *gsrc, int64_t nr, uint32_t *flags, uint64_t **out_g_p, uint64_t **out_v_p, int64_t *M_p,
                             PrePartitioned *pre, bool *use_pre) {
    int64_t P_f = (int64_t)std::ceil((double)nr / (L2_REG * 0.75));       // two sub-regions per fine partition: load ~0.38
    // a multiple of 64: every XCD group then owns the same number of coarse partitions (the pair regions assume an even 1/8 split)
    P_f = std::min<int64_t>(std::max<int64_t>((P_f + 63) / 64 * 64, 64), P_MAX);
    const int64_t P_c = P_f / L2_FINE_PER_COARSE;
    if ((double)nr / (double)P_f > L2_MAXROWS * 0.92) return FUSED_L2_NOT_TAKEN;
    const uint32_t sub = (double)nr / (double)P_f > L2_REG * 0.4 ? 2u : 1u;
    uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
    uint64_t *prk = c->work.take<uint64_t>(nr + 1), *prg = c->work.take<uint64_t>(nr + 1);
    L2Entry *table = reinterpret_cast<L2Entry *>(c->work.take<uint64_t>(((size_t)P_f * sub * L2_REG + 1) * 2));
    if (!prk || !prg || !table) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (fused join, table)");
    PartInfo rpart{};
    ScatterArgs rs{};
    rs.key = rkey; rs.pkeys = prk; rs.n_rows = nr; rs.P = (uint32_t)P_f; rs.seed = JN_SEED; rs.allow_two_pass = 1;
    rs.mv[rs.n_move++] = MoveDesc{gsrc, prg, 0, 0};
    ST_TRY(radix_partition(c, rs, &rpart, PANDRS_HIP_PHASE_BUILD, PANDRS_HIP_PHASE_BUILD, PANDRS_HIP_PHASE_BUILD));
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_BUILD);
        L2BuildArgs ba{prk, prg, rpart.offsets, rpart.NB, table, (uint32_t)P_f, sub, flags};
        const size_t lds = (size_t)L2_REG * 16;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_l2_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fused_l2_build_kernel, dim3((unsigned)P_f), dim3(JN_THREADS), lds, c->stream, ba);
        HIP_TRY(hipGetLastError());
    }
    // the probe side, by coarse partition: capacity mode when the sizes allow it (no histogram pass), else exact
    const bool sampled = sampled_partition_ok(nl, P_c) && !c->opt.exact_partition;
    PartInfo lpart{};
    uint64_t *plk = nullptr, *plv = nullptr;
    for (int pass = sampled ? 0 : 1; pass < 2; pass++) {
        const size_t NP = pass == 0 ? (size_t)sampled_partition_rows(nl, P_c) : (size_t)nl + 1;
        const size_t mark = c->work.off;
        plk = c->work.take<uint64_t>(NP); plv = c->work.take<uint64_t>(NP);
        if (!plk || !plv) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (fused join, probe side)");
        ScatterArgs ls{};
        ls.key = lkey; ls.pkeys = plk; ls.n_rows = nl; ls.P = (uint32_t)P_c; ls.seed = JN_SEED;
        ls.mv[ls.n_move++] = MoveDesc{vsrc, plv, 0, 0};
        lpart = PartInfo{};
        if (pass == 0) ST_TRY(radix_partition_sampled(c, ls, &lpart, PANDRS_HIP_PHASE_HISTOGRAM, PANDRS_HIP_PHASE_SCATTER));
        else ST_TRY(radix_partition(c, ls, &lpart, PANDRS_HIP_PHASE_HISTOGRAM, PANDRS_HIP_PHASE_SCAN, PANDRS_HIP_PHASE_SCATTER));
        // one read-back: build flags (+ the capacity-mode overflow flag)
        if (pass == 0) HIP_TRY(hipMemcpyAsync(flags + 5, lpart.flags, 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(h, flags, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h[0] || h[1]) return FUSED_L2_NOT_TAKEN;       // a fine partition too large, or a duplicate build key
        if (pass == 0 && h[5]) { c->work.off = mark; HIP_TRY(hipMemsetAsync(flags + 5, 0, 4, c->stream)); continue; }
        break;
    }
    const uint64_t cap_pairs = (uint64_t)std::max<int64_t>(nl, 1);            // unique build keys: at most one pair per probe row
    L2ProbeArgs pa{};
    pa.lkeys = plk; pa.lpay = plv; pa.loff = lpart.offsets; pa.lNB = lpart.NB;
    pa.gbeg = lpart.gbeg; pa.gcur = lpart.gcur; pa.gend = lpart.gend;
    pa.ablate = (uint32_t)c->opt.agg_ablate; pa.ticket = flags + 8;
    pa.P_c = (uint32_t)P_c; pa.P_f = (uint32_t)P_f; pa.sub = sub; pa.table = table; pa.flags = flags;
    pa.cursor = reinterpret_cast<unsigned long long *>(flags + 2); pa.cap = cap_pairs;
    const dim3 grid((unsigned)(8 * 4 * ((c->n_cu + 7) / 8)));

    // ---- pairs straight into the groupby engine's partitions (MODE 2) when the group count allows a fan-out the probe's LDS
    // holds: the group count is bounded by the distinct g of the build side (one estimate over n_right rows), the regions are
    // sized from the pair partitions of 1 tile in 256 at C5's size (MODE 1: a 0.4 % probe).  An overflowing region (a group the sample
    // under-weighted) raises a flag and the plain emission below runs instead.
    if (c->opt.join_no_pairpart != 1 && !pa.ablate) {
        int64_t est_g = 0;
        ST_TRY(estimate_groups(c, KeyDesc{gsrc, nullptr, nullptr, DT_CELL}, nr, &est_g));
        const int64_t T = lean_table_slots(c, 1);
        const int64_t pair_min = c->opt.join_pair_p > 0 ? c->opt.join_pair_p : 256;
        int64_t pair_P = std::max<int64_t>(pair_min, (int64_t)std::ceil((double)std::max<int64_t>(est_g, 1) / ((double)T * 0.70)));
        const bool lean_ok = !c->opt.agg_v1 && !c->opt.generic_aggregate && (c->opt.p_max <= 0 || pair_P <= c->opt.p_max);
        if (lean_ok && pair_P <= (int64_t)L2_PAIR_PMAX && sampled_partition_ok(nl, pair_P) && (double)nl * 1.5 < 4.0e9) {
            const uint32_t PP1 = (uint32_t)pair_P + 1;
            // 1.5 rows of capacity per probe row: the pairs of one build row (n_left / n_right of them on average) land in ONE region,
            // so the regions' margins are wider than for independent rows
            const size_t NPp = (size_t)((double)nl * 1.5) + (size_t)(pair_P + 1) * 8 * 96 + 65536 + SC_TILE_MAX;
            const size_t mark = c->work.off;
            uint32_t *hist = c->work.take<uint32_t>((size_t)SAMPLE_REPL * (PP1 + 1) + 64);
            uint32_t *gb = c->work.take<uint32_t>((size_t)PP1 * 8), *gc = c->work.take<uint32_t>((size_t)PP1 * 8), *ge = c->work.take<uint32_t>((size_t)PP1 * 8);
            ST_TRY(c->pairs.ensure(2 * Arena::padded(std::max(NPp, size_t(cap_pairs + 1)) * 8) + 4096, c->stream));
            uint64_t *pg = c->pairs.take<uint64_t>(NPp), *pv = c->pairs.take<uint64_t>(NPp);
            if (!hist || !gb || !gc || !ge || !pg || !pv) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (fused join, pair partitions)");
            uint32_t *pflags = hist + (size_t)SAMPLE_REPL * (PP1 + 1);
            pa.pair_P = (uint32_t)pair_P; pa.pair_seed = 0x9E3779B9u; pa.pair_hist = hist; pa.pair_cur = gc; pa.pair_end = ge;
            pa.pair_cap = (uint32_t)(NPp - SC_TILE_MAX); pa.pair_flags = pflags; pa.out_g = pg; pa.out_v = pv;
            // >= ~4 K sampled rows per pair partition (the budget of sampled_partition_rows assumes that much), at most 1 tile in 256
            pa.sample_stride = (uint32_t)std::min<int64_t>(256, std::max<int64_t>(1, nl / (pair_P * 4096)));
            {
                PhaseTimer pt(c, PANDRS_HIP_PHASE_PROBE);
                HIP_TRY(hipMemsetAsync(hist, 0, ((size_t)SAMPLE_REPL * (PP1 + 1) + 64) * 4, c->stream));
                HIP_TRY(hipMemsetAsync(flags + 8, 0, 32, c->stream));
                hipLaunchKernelGGL((fused_l2_probe_kernel<1, L2_THREADS>), grid, dim3(L2_THREADS), 0, c->stream, pa);
                // (join_no_pairpart = 2, tests: regions planned for an eighth of the rows, so that they overflow and the fallback runs)
                plan_sampled_regions(c, hist, c->opt.join_no_pairpart == 2 ? nl / 8 : nl, PP1, pa.pair_cap, gb, gc, ge, pflags,
                                     1.0 + (double)nl / (double)nr);
                HIP_TRY(hipMemsetAsync(flags + 8, 0, 32, c->stream));
                const size_t part_lds = (size_t)L2_THREADS_PART * L2_RPT * 18;
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_l2_probe_kernel<2, L2_THREADS_PART>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_lds));
                hipLaunchKernelGGL((fused_l2_probe_kernel<2, L2_THREADS_PART>), dim3((unsigned)(8 * 2 * ((c->n_cu + 7) / 8))), dim3(L2_THREADS_PART),
                                   part_lds, c->stream, pa);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipMemcpyAsync(flags + 6, pflags, 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(h, flags, 32, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const uint64_t total = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
            if (!h[6] && total <= cap_pairs) {
                pre->part = PartInfo{};
                pre->part.P = (uint32_t)pair_P; pre->part.gbeg = gb; pre->part.gcur = gc; pre->part.gend = ge; pre->part.flags = pflags;
                pre->part.total_cap = pa.pair_cap;
                pre->pkeys = pg; pre->pvals[0] = pv; pre->est_groups = est_g;
                *use_pre = true;
                c->timings.n_partitions = P_c;        // the probe side's (coarse) fan-out
                *out_g_p = pg; *out_v_p = pv; *M_p = (int64_t)total;
                return 0;
            }
            // a region overflowed: plain emission
            c->pair_fallback = true;
            c->work.off = mark;
            c->pairs.off = 0;
            HIP_TRY(hipMemsetAsync(flags + 2, 0, 8, c->stream));
            HIP_TRY(hipMemsetAsync(flags + 8, 0, 32, c->stream));
        }
    }
    ST_TRY(c->pairs.ensure(2 * Arena::padded(size_t(cap_pairs + 1) * 8) + 4096, c->stream));
    uint64_t *out_g = c->pairs.take<uint64_t>(cap_pairs + 1), *out_v = c->pairs.take<uint64_t>(cap_pairs + 1);
    if (!out_g || !out_v) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "pairs arena too small");
    pa.out_g = out_g; pa.out_v = out_v;
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_PROBE);
        hipLaunchKernelGGL((fused_l2_probe_kernel<0, L2_THREADS>), grid, dim3(L2_THREADS), 0, c->stream, pa);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(h, flags, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t total = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
    if (total > cap_pairs) return fail(PANDRS_HIP_ERR_COMPUTATION, "fused join: more pairs than probe rows with unique build keys");
    c->timings.n_partitions = P_c;        // the probe side's (coarse) fan-out
    *out_g_p = out_g; *out_v_p = out_v; *M_p = (int64_t)total;
    return 0;
}

int32_t join_groupby_sum_entry(pandrs_hip_ctx *c, int32_t mem_space, const pandrs_hip_column *lk,
                               const pandrs_hip_column *lv, int64_t nl,
                               const pandrs_hip_column *rk, const pandrs_hip_column *rg,
                               int64_t nr, int64_t *out_n_groups) {
    if (!c || !lk || !lv || !rk || !rg || !out_n_groups || nl < 0 || nr < 0)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join_groupby_sum: bad arguments");
    if (lk->dtype != rk->dtype)
        return fail(PANDRS_HIP_ERR_TYPE_MISMATCH, "join key columns have different types (%d and %d)", lk->dtype, rk->dtype);
    if (lv->dtype != PANDRS_HIP_I64 && lv->dtype != PANDRS_HIP_F64)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "Aggregation operation 0 is not supported for column type %d", lv->dtype);
    if (rg->dtype == PANDRS_HIP_BOOLBITS)
        return fail(PANDRS_HIP_ERR_OPERATION_FAILED, "join_groupby_sum: bit-packed group column is not supported on the device path");
    if (nl >= (int64_t(1) << 32) - 16384 || nr >= (int64_t(1) << 32) - 16384)
        return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: a side exceeds the 2^32-row per-call limit");
    int32_t vdt = lv->dtype; uint8_t vhn = 0;
    pandrs_hip_agg_spec spec{0, PANDRS_HIP_AGG_SUM};
    Plan pl;
    ST_TRY(build_plan(&vdt, &vhn, 1, &spec, 1, pl));

    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    // the whole call as one attempt: when the pairs went pre-partitioned into the groupby engine (MODE 2 of the L2-region probe)
    // and the engine could not take them as they were — a full LDS table (cardinality under-estimated under skew), or a plan
    // that rules the lean aggregate out — the attempt is repeated with the plain pair emission, which the engine partitions itself
    bool pre_failed = false;
    auto attempt_call = [&]() -> int32_t {
    timings_begin(c);
    c->jn.valid = false;
    KeyDesc lkey{}, rkey{}, lval{}, rgrp{};
    {
        PhaseTimer pt(c, PANDRS_HIP_PHASE_STAGE_IN);
        if (mem_space == PANDRS_HIP_MEM_HOST)
            ST_TRY(c->staging.ensure(dtype_bytes(lk->dtype, nl) + dtype_bytes(rk->dtype, nr) + size_t(nl) * 8 + size_t(nr) * 8 +
                                     (nl + nr) / 2 + (1 << 16), c->stream));
        ST_TRY(stage_key(c, mem_space, lk, nl, &lkey));
        ST_TRY(stage_key(c, mem_space, rk, nr, &rkey));
        ST_TRY(stage_key(c, mem_space, lv, nl, &lval));
        ST_TRY(stage_key(c, mem_space, rg, nr, &rgrp));
    }
    int64_t P = c->opt.partitions > 0 ? c->opt.partitions
                                     : std::max<int64_t>(1, (int64_t)std::ceil((double)nr / (JN_RCAP * 0.6)));
    // beyond ~1024 LDS-sized build partitions the probe side's scatter degrades (short runs): table regions in L2 instead
    // (and only when the probe side is at least twice the build side: the regions cost a 2 GB table store per 50 M build rows and
    // every entry must be looked up a few times for the L2 to matter.  Measured, 50 M build rows: 500 M probe rows 24.2 -> 19.6 ms,
    // 62.5 M probe rows 4.65 -> 4.9 ms)
    bool l2_path = c->opt.partitions <= 0 && !c->opt.join_generic && c->opt.join_no_l2 <= 0 && nl > 0 && nr > 0 &&
                   ((P > 1024 && nl >= 2 * nr) || c->opt.join_no_l2 < 0);
    size_t ws = 2 * engine_workspace_bytes(0, 0, 0) + 3 * Arena::padded(size_t(nl + 1) * 8) + 3 * Arena::padded(size_t(nr + 1) * 8) + (1 << 20);
    ws += two_pass_workspace_bytes(nl, 1, 0) + two_pass_workspace_bytes(nr, 1, 0);      // both sides may take the two-pass partition (fan-outs >= 6144)
    if (l2_path) ws += Arena::padded(((size_t)P_MAX * 2 * L2_REG + 1) * 16) + Arena::padded(size_t(nl) * 4 + (size_t(1) << 25)) + (1 << 20);
    ST_TRY(c->work.ensure(ws, c->stream));
    P = std::min<int64_t>(std::max<int64_t>(P, std::min<int64_t>(256, (nl + nr) / 32768)), P_MAX);
    P = std::max<int64_t>(P, 1);
    uint32_t *h = reinterpret_cast<uint32_t *>(c->pinned);
    int64_t M = 0;
    uint64_t *out_g = nullptr, *out_v = nullptr;
    uint64_t cap_pairs = (uint64_t)std::max<int64_t>(nl, 1);    // exact bound for unique build keys
    PrePartitioned pre{};
    bool use_pre = false;
    c->pair_fallback = false;
    // the LDS multimaps cannot hold the build side when even the maximum fan-out leaves partitions too large
    bool l2_tried = false, skew_rerouted = false;
    bool general = c->opt.join_generic != 0 || (double)nr / (double)P > FJ_MAXROWS * 0.95;
    for (int attempt = 0;; attempt++) {
        if (general) {
            // materialise the inner join with the general (table-based, segmented-sort) join, then
            // gather the (g, v) pairs: slower than the fused probe but without any size limit
            ST_TRY(join_core(c, lkey, nl, rkey, nr, PANDRS_HIP_JOIN_INNER));
            M = c->jn.n_rows;
            ST_TRY(c->pairs.ensure(2 * Arena::padded(size_t(M + 1) * 8) + 4096, c->stream));
            out_g = c->pairs.take<uint64_t>(M + 1);
            out_v = c->pairs.take<uint64_t>(M + 1);
            if (!out_g || !out_v) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "pairs arena too small");
            if (M > 0) {
                PhaseTimer pt(c, PANDRS_HIP_PHASE_PROBE);
                hipLaunchKernelGGL(pairs_from_indices_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->stream,
                                   c->jn.left_idx, c->jn.right_idx, M, rgrp, rg->dtype == PANDRS_HIP_U32CODE ? 1 : 0, lval, out_g, out_v);
                HIP_TRY(hipGetLastError());
            }
            c->jn.valid = false;                    // the result arena is about to hold the groupby result
            break;
        }
        c->work.off = 0;
        c->timings.n_partitions = P; c->timings.retries = attempt;
        uint32_t *flags = c->work.take<uint32_t>(64);               // [0] partition overflow, [2..3] pair cursor
        if (!flags) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (fused join)");
        HIP_TRY(hipMemsetAsync(flags, 0, 256, c->stream));
        // payload columns: plain 8-byte columns move as they are; masked / 4-byte ones are cleaned first
        const void *gsrc = rgrp.data, *vsrc = lval.data;
        if (nr > 0 && (rgrp.null_bits || rg->dtype == PANDRS_HIP_U32CODE)) {
            uint64_t *t = c->work.take<uint64_t>(nr + 1);
            if (!t) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (fused join)");
            hipLaunchKernelGGL(clean_payload_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, c->stream,
                               rgrp.data, rgrp.null_bits, rg->dtype == PANDRS_HIP_U32CODE ? 1 : 0, nr, t);
            gsrc = t;
        }
        if (nl > 0 && lval.null_bits) {
            uint64_t *t = c->work.take<uint64_t>(nl + 1);
            if (!t) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (fused join)");
            hipLaunchKernelGGL(clean_payload_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, c->stream,
                               lval.data, lval.null_bits, 0, nl, t);
            vsrc = t;
        }
        if (l2_path) {
            l2_path = false;                        // one try; whatever it declines goes down the LDS-multimap path below
            l2_tried = true;
            const size_t mark = c->work.off;
            int32_t st = fused_l2_path(c, lkey, vsrc, nl, rkey, gsrc, nr, flags, &out_g, &out_v, &M, &pre, &use_pre);
            if (st == 0) break;
            if (st != FUSED_L2_NOT_TAKEN) return st;
            c->work.off = mark;
            HIP_TRY(hipMemsetAsync(flags, 0, 256, c->stream));
        }
        uint64_t *prk = c->work.take<uint64_t>(nr + 1), *prg = c->work.take<uint64_t>(nr + 1);
        uint64_t *plk = c->work.take<uint64_t>(nl + 1), *plv = c->work.take<uint64_t>(nl + 1);
        if (!prk || !prg || !plk || !plv)
            return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (fused join)");
        PartInfo rpart{}, lpart{};
        ScatterArgs rs{}, ls{};
        rs.key = rkey; rs.pkeys = prk; rs.n_rows = nr; rs.P = (uint32_t)P; rs.seed = JN_SEED; rs.allow_two_pass = 1;
        rs.mv[rs.n_move++] = MoveDesc{gsrc, prg, 0, 0};
        ls.key = lkey; ls.pkeys = plk; ls.n_rows = nl; ls.P = (uint32_t)P; ls.seed = JN_SEED; ls.allow_two_pass = 1;
        ls.mv[ls.n_move++] = MoveDesc{vsrc, plv, 0, 0};
        ST_TRY(radix_partition(c, rs, &rpart, PANDRS_HIP_PHASE_BUILD, PANDRS_HIP_PHASE_BUILD, PANDRS_HIP_PHASE_BUILD));
        ST_TRY(radix_partition(c, ls, &lpart, PANDRS_HIP_PHASE_SCATTER, PANDRS_HIP_PHASE_SCATTER, PANDRS_HIP_PHASE_SCATTER));
        ST_TRY(c->pairs.ensure(2 * Arena::padded(size_t(cap_pairs + 1) * 8) + 4096, c->stream));
        out_g = c->pairs.take<uint64_t>(cap_pairs + 1);
        out_v = c->pairs.take<uint64_t>(cap_pairs + 1);
        if (!out_g || !out_v) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "pairs arena too small");
        FusedArgs fa{};
        fa.rkeys = prk; fa.rpay = prg; fa.lkeys = plk; fa.lpay = plv;
        fa.roff = rpart.offsets; fa.loff = lpart.offsets; fa.rNB = rpart.NB; fa.lNB = lpart.NB; fa.P = (uint32_t)P;
        fa.cursor = reinterpret_cast<unsigned long long *>(flags + 2); fa.cap = cap_pairs;
        fa.out_g = out_g; fa.out_v = out_v; fa.flags = flags;
        // a hot probe key: one partition with a large share of the probe rows, one workgroup for all of them.  The L2-region path
        // deals its probe tiles by ticket and does not care — taken instead when such a partition shows up (once)
        const bool may_reroute = !l2_tried && !skew_rerouted && c->opt.partitions <= 0 && !c->opt.join_generic && c->opt.join_no_l2 <= 0 && nl >= (int64_t(1) << 22);
        fa.l_limit = may_reroute ? (uint32_t)std::min<int64_t>(std::max<int64_t>(32 * (nl / P), 262144), 0x7FFFFFFF) : 0u;
        const size_t lds = (size_t)FJ_SLOTS * 16 + FJ_BUCKETS * 4 + 16 + FJ_RPT * 16 * 4 + 64;
        uint64_t total = 0;
        for (;;) {
            {
                PhaseTimer pt(c, PANDRS_HIP_PHASE_PROBE);
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_probe_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(fused_probe_kernel, dim3((unsigned)P), dim3(JN_THREADS), lds, c->stream, fa);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipMemcpyAsync(h, flags, 32, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            total = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
            if (h[5] && fa.l_limit) break;
            if (h[0] || total <= fa.cap) break;
            // duplicate build keys produced more pairs than probe rows: size the buffer exactly, probe again
            if (total >= (1ull << 32) - 16384)
                return fail(PANDRS_HIP_ERR_INVALID_ARGUMENT, "join: output exceeds the 2^32-row per-call limit");
            cap_pairs = total;
            ST_TRY(c->pairs.ensure(2 * Arena::padded(size_t(cap_pairs + 1) * 8) + 4096, c->stream));
            fa.out_g = out_g = c->pairs.take<uint64_t>(cap_pairs + 1);
            fa.out_v = out_v = c->pairs.take<uint64_t>(cap_pairs + 1);
            if (!out_g || !out_v) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "pairs arena too small");
            fa.cap = cap_pairs;
            HIP_TRY(hipMemsetAsync(flags, 0, 256, c->stream));
        }
        if (h[5] && fa.l_limit) {                    // a hot probe key: the L2-region path, whatever the sizes say
            skew_rerouted = true; l2_path = true;
            ST_TRY(c->work.ensure(ws + Arena::padded(((size_t)P_MAX * 2 * L2_REG + 1) * 16) + Arena::padded(size_t(nl) * 4 + (size_t(1) << 25)) + (1 << 20), c->stream));
            continue;
        }
        if (h[0]) {
            // first overflow: more partitions (unlucky hashing); then the general path (hot build keys, huge builds)
            if (attempt == 0 && P < P_MAX) P = std::min<int64_t>(P * 4, P_MAX);
            else general = true;
            continue;
        }
        M = (int64_t)total;
        break;
    }
    // groupby(g).sum(v) over the matched pairs
    RowSource rsrc;
    rsrc.n_rows = M;
    rsrc.key = KeyDesc{out_g, nullptr, nullptr, rg->dtype == PANDRS_HIP_U32CODE ? DT_CELL : rg->dtype};
    rsrc.val_data[0] = out_v;
    rsrc.val_null_bits[0] = nullptr;
    if (use_pre) rsrc.pre = &pre;
    const int64_t join_fanout = general ? 0 : c->timings.n_partitions;
    {
        const int32_t est = run_engine(c, rsrc, pl, /*merge=*/false, /*partials=*/false, 1, rg->dtype);
        if (est && use_pre) pre_failed = true;
        ST_TRY(est);
    }
    c->timings.n_partitions = join_fanout;          // the join's build-side fan-out (0: general path), not the pair groupby's
    c->timings.retries = c->pair_fallback ? 1 : 0;  // 1: the partitioned pair output overflowed a region and the plain emission answered
    {
        int64_t K = lk->dtype == PANDRS_HIP_U32CODE ? 4 : 8;
        c->timings.algorithmic_bytes = nl * (K + 8) + nr * (K + 8) + c->gb.n_groups * 16;   // SURVEY.md §8d, fused form
    }
    ST_TRY(timings_end(c));
    *out_n_groups = c->gb.n_groups;
    return 0;
    };
    int32_t st = attempt_call();
    if (st && pre_failed) {
        const int64_t saved = c->opt.join_no_pairpart;
        c->opt.join_no_pairpart = 1;
        pre_failed = false;
        st = attempt_call();
        c->opt.join_no_pairpart = saved;
        if (!st) c->timings.retries = 2;            // 2: the pre-partitioned pairs were refused by the engine, the plain emission answered
    }
    return st;
}

}  // namespace pandrs
