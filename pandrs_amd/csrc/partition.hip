// partition.hip — the radix partitioner shared by groupby, join, median and the shuffle (gfx950, wave64):
// cardinality estimate, histogram, 3-kernel exclusive scan, LDS-staged scatter with XCD-shared write
// cursors.  Split out of groupby.hip; the aggregate engine lives there.
#include "engine.hpp"

#include <algorithm>
#include <cmath>

namespace pandrs {

template <typename K>
static int32_t set_max_lds(K kernel, int bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

// ------------------------------------------------------------------------------------ estimate
// distinct[0] = distinct keys in the sample, [2] = adjacent pairs that differ, [3] = adjacent pairs (one u64, one atomic), [4] = blocks done,
// of the SUB-sample (every fourth block): [7] = distinct keys, [5] = keys sighted at least twice, [6] = at least three times
// (=> singletons f1 = [7] - [5], doubletons f2 = [5] - [6]: the Chao1 estimate).
// The last block to finish copies [0..2] to `host_out` (pinned, device-visible), so the host needs one stream
// synchronise and no copy; estimate_clear_kernel then re-arms table and counters for the next call (off the
// critical path: it runs while the host plans).
__device__ __forceinline__ void estimate_publish(uint32_t *distinct, uint32_t *host_out) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&distinct[4], 1u) == gridDim.x - 1) {
            __threadfence();
            host_out[0] = __hip_atomic_load(&distinct[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[1] = __hip_atomic_load(&distinct[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[2] = __hip_atomic_load(&distinct[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[4] = __hip_atomic_load(&distinct[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[5] = __hip_atomic_load(&distinct[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[6] = __hip_atomic_load(&distinct[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[7] = __hip_atomic_load(&distinct[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[8] = __hip_atomic_load(&distinct[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&host_out[3], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void estimate_clear_kernel(uint64_t *table, uint32_t slots, uint32_t *distinct, uint32_t *counts, uint32_t *sight) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < slots) { table[i] = EMPTY_KEY; sight[i] = 0; }
    if (i < 12) distinct[i] = 0;
    if (counts && i < slots) counts[i] = 0;
}

// ---- hot-key coverage of the sample (absorb-and-spill decision, groupby.hip) ---------------------------------------
// After an estimate whose table was KEPT: the same strided sample is walked again and every row adds 1 to its key's
// counter (a wave's repeats of one key are added by one lane) ...
__global__ __launch_bounds__(1024) void estimate_count_kernel(KeyDesc key, int64_t n_rows, int64_t stride, int64_t n_sample,
                                                              const uint64_t *table, uint32_t table_mask, uint32_t *counts) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = s < n_sample && s * stride < n_rows;
    uint64_t k = 0;
    if (live) {
        const int64_t i = sample_row(s, stride);
        k = key_cell(key, i);
        live = !key_is_null(key, i) && k != EMPTY_KEY;
    }
    uint32_t weight = live ? 1u : 0u;
    for (int round = 0; round < 8; round++) {                  // peel the wave's leading keys: a handful of dominant keys must not cost one global atomic per lane
        const unsigned long long m = __ballot(live && weight == 1u);
        if (!m) break;
        const int leader = __ffsll((long long)m) - 1;
        const uint64_t lk = __shfl(k, leader, 64);
        const unsigned long long same = __ballot(live && weight == 1u && k == lk);
        if (live && weight == 1u && k == lk) {
            if ((int)(threadIdx.x & 63) == leader) weight = (uint32_t)__popcll(same) + 0x80000000u;   // marked: done peeling
            else live = false;
        }
    }
    if (live) {
        uint32_t slot = hash32(k, 0x1234567u) & table_mask;
        for (uint32_t probe = 0; probe <= table_mask; probe++) {
            const uint64_t cur = table[slot];
            if (cur == k) { atomicAdd(&counts[slot], weight & 0x7FFFFFFFu); break; }
            if (cur == EMPTY_KEY) break;                        // cannot happen: the estimate inserted every sampled key
            slot = (slot + 1) & table_mask;
        }
    }
}
// ... and ONE workgroup-cooperative pass over the table turns the counters into: rows of the sample that belong to the
// `budget` most frequent keys.  A histogram of the counts (exact up to 4094, one overflow bin) is built in LDS by every
// workgroup for its part of the table and added to a global histogram; the last workgroup walks it from the top.
// out[0] = sampled rows covered, out[1] = keys taken, out[2] = rows counted in all, out[3] = 1 when done (host polls).
constexpr uint32_t COV_BINS = 4096;
__global__ __launch_bounds__(1024) void estimate_coverage_kernel(const uint64_t *table, const uint32_t *counts, uint32_t slots,
                                                                 uint32_t budget, uint32_t *g_hist /* [2][COV_BINS] + [1] done */,
                                                                 uint32_t *thr_out /* device [4] */, uint32_t *host_out) {
    __shared__ uint32_t hn[COV_BINS], hr[COV_BINS];
    __shared__ uint32_t is_last;
    for (uint32_t b = threadIdx.x; b < COV_BINS; b += 1024) { hn[b] = 0; hr[b] = 0; }
    __syncthreads();
    for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < slots; i += gridDim.x * 1024) {
        const uint32_t c = counts[i];
        if (c && table[i] != EMPTY_KEY) {
            const uint32_t b = min(c, COV_BINS - 1);
            atomicAdd(&hn[b], 1u);
            atomicAdd(&hr[b], c);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < COV_BINS; b += 1024) {
        if (hn[b]) { atomicAdd(&g_hist[b], hn[b]); atomicAdd(&g_hist[COV_BINS + b], hr[b]); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        is_last = atomicAdd(&g_hist[2 * COV_BINS], 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    for (uint32_t b = threadIdx.x; b < COV_BINS; b += 1024) {
        hn[b] = __hip_atomic_load(&g_hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hr[b] = __hip_atomic_load(&g_hist[COV_BINS + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // from the most frequent keys down: thread t owns the four bins COV_BINS - 1 - 4 t ... (descending), one block scan gives the
    // number of keys in front of them
    __shared__ uint32_t wt[17], tot[3];
    if (threadIdx.x < 3) tot[threadIdx.x] = 0;
    uint32_t ln[4], lr[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int b = (int)COV_BINS - 1 - (4 * (int)threadIdx.x + q);
        ln[q] = b >= 1 ? hn[b] : 0u; lr[q] = b >= 1 ? hr[b] : 0u;
        mine += ln[q];
    }
    const uint32_t before0 = block_exclusive_scan<1024>(mine, wt, nullptr);
    uint32_t before = before0;
    uint32_t rows = 0, keys = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        all += lr[q];
        if (ln[q] && before < budget) {
            const uint32_t take = min(ln[q], budget - before);
            rows += (uint32_t)((unsigned long long)lr[q] * take / ln[q]);
            keys += take;
        }
        before += ln[q];
    }
    if (rows) atomicAdd(&tot[0], rows);
    if (keys) atomicAdd(&tot[1], keys);
    if (all) atomicAdd(&tot[2], all);
    // the admission rule of the hot-key image (hot_image_kernel): every key counted more often than bin `thr`, and `take` keys of that bin
    __shared__ uint32_t s_thr, s_take;
    if (threadIdx.x == 0) { s_thr = COV_BINS; s_take = 0; }
    __syncthreads();
    {
        uint32_t bf = before0;                       // keys in front of this thread's first bin
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int b = (int)COV_BINS - 1 - (4 * (int)threadIdx.x + q);
            if (ln[q] && bf < budget) atomicMin(&s_thr, (uint32_t)b);
            bf += ln[q];
        }
    }
    __syncthreads();
    {
        uint32_t bf = before0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int b = (int)COV_BINS - 1 - (4 * (int)threadIdx.x + q);
            if ((uint32_t)b == s_thr && ln[q]) s_take = min(ln[q], budget - bf);
            bf += ln[q];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        thr_out[0] = s_thr; thr_out[1] = s_take; thr_out[2] = 0;          // [2]: keys of bin `thr` admitted so far
        host_out[0] = tot[0]; host_out[1] = tot[1]; host_out[2] = tot[2];
        __hip_atomic_store(&host_out[3], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (uint32_t b = threadIdx.x; b < 2 * COV_BINS + 1; b += 1024) g_hist[b] = 0;       // re-armed for the next call
}
// The hot-key image of the absorb pass (absorb.hip): the sample's most frequent keys — the ones the coverage pass counted —
// placed exactly where absorb_kernel looks for them (4-key buckets, home bucket then the next one), so that every workgroup
// starts from a table that already holds the keys worth a slot instead of the first keys it happens to see (first come, first
// served gave the cold keys 40 % of the slots: C3 absorbed 74 % of its rows where the hot set covers 88 %).
// `image` [T] is EMPTY on entry; a key whose two buckets are full is left out (its rows spill like a cold key's).
// `band`: 0 = keys counted >= 16 x the threshold, 1 = [4 x, 16 x), 2 = [1 x, 4 x) — three launches, the hottest keys first: a key
// that finds its two buckets full is left out, and that must be a cold one (all keys at once left 5-10 % of the HOT keys of a
// 500-key hot set out: 6 M of their rows spilled and met again, serialised, in one LDS slot of the spill run)
__global__ __launch_bounds__(1024) void hot_image_kernel(const uint64_t *table, const uint32_t *counts, uint32_t slots, uint32_t *thr,
                                                         uint64_t *image, uint32_t T, uint32_t seed, int band) {
    const uint32_t t = thr[0], take = thr[1], NBK = T >> 2;
    const uint32_t lo = band == 0 ? 16 * t : band == 1 ? 4 * t : t, hi = band == 0 ? 0xFFFFFFFFu : band == 1 ? 16 * t : 4 * t;
    for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < slots; i += gridDim.x * 1024) {
        const uint32_t cnt = counts[i];
        const uint64_t k = table[i];
        if (!cnt || k == EMPTY_KEY) continue;
        const uint32_t b = min(cnt, COV_BINS - 1);
        if (b < t || cnt < lo || cnt >= hi) continue;
        if (b == t && atomicAdd(&thr[2], 1u) >= take) continue;
        uint32_t bk = slot_of(hash32(k, seed), NBK);
        bool placed = false;
        for (int hop = 0; hop < 2 && !placed; hop++) {
            for (int q = 0; q < 4 && !placed; q++)
                placed = atomicCAS((unsigned long long *)&image[4 * bk + q], EMPTY_KEY, k) == EMPTY_KEY;
            bk = bk + 1 == NBK ? 0 : bk + 1;
        }
    }
}

__device__ __forceinline__ void estimate_body(KeyDesc key, int64_t n_rows, int64_t stride, int64_t n_sample,
                                uint64_t *table, uint32_t table_mask, uint32_t *distinct, uint32_t *sight) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = s < n_sample && s * stride < n_rows;
    uint64_t k = 0;
    // second signal, for CLUSTERED inputs (e.g. rows sorted by key), where a strided sample shows no
    // repeats at all: the share of adjacent row pairs whose keys differ.  #groups <= #runs = boundaries + 1
    // whatever the order, so it bounds the estimate from above.  distinct[2] = boundaries, [3] = pairs.
    // ... and its control, FAR pairs: this sampled row against the one 32 lanes on (32 x stride rows away).  A dominant key makes
    // neighbours share their key in ANY row order — and then far pairs share it as often; rows clustered by key do not.
    // distinct[8] = far pairs that differ, [9] = far pairs.
    bool pair = false, differs = false, nul = false;
    const bool in_range = live;
    if (live) {
        int64_t i = sample_row(s, stride);
        nul = key_is_null(key, i);
        k = key_cell(key, i);
        if (i + 1 < n_rows) {
            const bool nul2 = key_is_null(key, i + 1);
            pair = true;
            differs = nul != nul2 || (!nul && k != key_cell(key, i + 1));
        }
        live = !nul && k != EMPTY_KEY;
    }
    {   // one atomic pair per workgroup (same-address global atomics serialise)
        __shared__ uint32_t sb[4];
        if (threadIdx.x < 4) sb[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t kf = __shfl_down(k, 32, 64);
        const int nf = __shfl_down((int)nul, 32, 64), rf = __shfl_down((int)in_range, 32, 64);
        const bool fpair = in_range && (threadIdx.x & 63) < 32 && rf != 0;
        const bool fdiff = fpair && ((int)nul != nf || (!nul && k != kf));
        const unsigned long long mp = __ballot(pair), md = __ballot(differs), fp = __ballot(fpair), fd = __ballot(fdiff);
        if ((threadIdx.x & 63) == 0) {
            if (mp) atomicAdd(&sb[0], (uint32_t)__popcll(mp));
            if (md) atomicAdd(&sb[1], (uint32_t)__popcll(md));
            if (fp) atomicAdd(&sb[2], (uint32_t)__popcll(fp));
            if (fd) atomicAdd(&sb[3], (uint32_t)__popcll(fd));
        }
        __syncthreads();
        if (threadIdx.x == 0 && sb[0])
            atomicAdd(reinterpret_cast<unsigned long long *>(&distinct[2]), ((unsigned long long)sb[0] << 32) | sb[1]);
        if (threadIdx.x == 64 && sb[2])
            atomicAdd(reinterpret_cast<unsigned long long *>(&distinct[8]), ((unsigned long long)sb[2] << 32) | sb[3]);
    }
    // a dominant key would make every lane CAS the same address: peel the wave's leading keys first (one lane inserts for all lanes that hold the same key)
    bool peeled = false;       // this lane is (or was represented by) a leader already
    uint32_t mult = 1;         // sampled rows this lane stands for (rows clustered by key put one key many times into a wave:
                               // counted once, a hot key's few waves looked like a doubleton and Chao1 lost an 18 x long tail)
    for (int round = 0; round < 8; round++) {          // (8: with a handful of keys in all, every lane is represented — the sighting counters
                                                        // below would otherwise take one device atomic per sampled row on ten addresses)
        unsigned long long m = __ballot(live && !peeled);
        if (!m) break;
        int leader = __ffsll((long long)m) - 1;
        uint64_t lk = __shfl(k, leader, 64);
        const unsigned long long same = __ballot(live && !peeled && k == lk);
        if (live && !peeled && k == lk) {
            peeled = true;
            if ((int)(threadIdx.x & 63) != leader) live = false;           // the leader inserts on their behalf
            else mult = (uint32_t)__popcll(same);                          // ... and counts their sightings with its own
        }
    }
    // new keys are counted per workgroup in LDS and added to the global counter ONCE: a device-scope atomic per
    // inserted key on one address (even wave-aggregated: ~20 K of them) was most of this kernel's 80 us
    // ... and so are the keys that reach their SECOND and THIRD sighting (a saturating counter per slot; a hot key costs three
    // atomics in all, the pre-check keeps the rest away): singletons and doubletons of the sample are what tells "few keys, all
    // seen" from "a few hot keys in front of a long tail" — the uniform-occupancy model alone under-estimated a 2 K-hot-key /
    // 1 M-key column 16 x, and the engine paid with three full retries (C2's skewed variants: 12-29 ms instead of ~3.5)
    // The sighting counters are kept by every FOURTH block of the sample only (a 64 K-row sub-sample with its own distinct count:
    // Chao1 needs d, f1, f2 of ONE sample, and the coherent counter loads — a quarter of a million of them on a few hot lines when
    // the keys are few — were 20-30 us of a 40 us kernel).
    __shared__ uint32_t inserted, once, twice, thrice;
    if (threadIdx.x == 0) { inserted = 0; once = 0; twice = 0; thrice = 0; }
    __syncthreads();
    const bool sub = (blockIdx.x & 3u) == 0u;
    if (live) {
        uint32_t slot = hash32(k, 0x1234567u) & table_mask;
        uint32_t seen = 3;                               // the slot's sighting counter, loaded WITH the key (one round trip, not two)
        bool placed = false;
        for (uint32_t probe = 0; probe <= table_mask; probe++) {
            uint64_t cur = table[slot];
            if (sub) seen = __hip_atomic_load(&sight[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == k) { placed = true; break; }
            if (cur == EMPTY_KEY) {
                uint64_t old = atomicCAS((unsigned long long *)&table[slot], EMPTY_KEY, k);
                if (old == EMPTY_KEY) { atomicAdd(&inserted, 1u); placed = true; break; }
                if (old == k) { placed = true; break; }
            }
            slot = (slot + 1) & table_mask;
        }
        if (sub && placed && seen < 3u) {
            const uint32_t o = atomicAdd(&sight[slot], mult), n2 = o + mult;
            if (o == 0) atomicAdd(&once, 1u);
            if (o < 2u && n2 >= 2u) atomicAdd(&twice, 1u);
            if (o < 3u && n2 >= 3u) atomicAdd(&thrice, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && inserted) atomicAdd(distinct, inserted);
    if (threadIdx.x == 1 && twice) atomicAdd(&distinct[5], twice);
    if (threadIdx.x == 2 && thrice) atomicAdd(&distinct[6], thrice);
    if (threadIdx.x == 3 && once) atomicAdd(&distinct[7], once);
}

__global__ void estimate_kernel(KeyDesc key, int64_t n_rows, int64_t stride, int64_t n_sample,
                                uint64_t *table, uint32_t table_mask, uint32_t *distinct, uint32_t *host_out, uint32_t *sight) {
    estimate_body(key, n_rows, stride, n_sample, table, table_mask, distinct, sight);
    estimate_publish(distinct, host_out);
}

// ---- second stage of the estimate: a hash-slice CENSUS -----------------------------------------------------------------
// The strided sample cannot size a long tail behind a broad hot class (SURVEY 8d's own 80/20 variant of C2: 200 K keys with 80 % of
// the rows + 1 M keys sharing the rest): its 256 K rows hold the tail's keys once each, and "seen once" says nothing about how many
// there are.  What does: take every row of a TENTH of the column (16 K-row blocks spread evenly) whose key hashes into one slice of
// the key space (1 / R of the keys).  A key of that slice is then seen with its FULL multiplicity among the scanned rows — ~2 rows
// for the tail's keys instead of ~0.003 in the strided sample — so singletons and doubletons of the slice are meaningful for every
// class of keys at once, Chao1 over them (exact for Poisson counts of any one rate, a lower bound for mixtures) extrapolates from
// the scanned tenth to the whole column, and R x that is the group count.  The scan is a contiguous read (80 MB for C2: ~20 us), the
// ~150 K rows that pass the slice filter go through the same CAS table + saturating sighting counters as the first stage.
// counters: [0] distinct keys of the slice, [1] rows that passed, [4] blocks done, [5] keys sighted twice or more, [6] three times or more.
constexpr uint32_t CENSUS_RPT = 16, CENSUS_BLOCK_ROWS = 1024 * CENSUS_RPT, CENSUS_QCAP = 4096;
// one key of the slice into the census table: CAS insert, then the saturating sighting counter (`mult` sightings at once)
__device__ __forceinline__ void census_insert(uint64_t k, uint32_t mult, uint64_t *table, uint32_t table_mask, uint32_t *sight,
                                              uint32_t *inserted, uint32_t *twice, uint32_t *thrice) {
    uint32_t slot = hash32(k, 0x1234567u) & table_mask;
    uint32_t seen = 3;
    bool placed = false;
    for (uint32_t probe = 0; probe <= table_mask; probe++) {
        const uint64_t cur = table[slot];
        seen = __hip_atomic_load(&sight[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == k) { placed = true; break; }
        if (cur == EMPTY_KEY) {
            const uint64_t old = atomicCAS((unsigned long long *)&table[slot], EMPTY_KEY, k);
            if (old == EMPTY_KEY) { atomicAdd(inserted, 1u); placed = true; break; }
            if (old == k) { placed = true; break; }
        }
        slot = (slot + 1) & table_mask;
    }
    if (placed && seen < 3u) {
        const uint32_t o = atomicAdd(&sight[slot], mult), n2 = o + mult;
        if (o < 2u && n2 >= 2u) atomicAdd(twice, 1u);
        if (o < 3u && n2 >= 3u) atomicAdd(thrice, 1u);
    }
}
// A workgroup scans 16 K contiguous rows: 16 independent key loads per thread (the scan is a stream, not a chain of round trips), the
// ~1 / R of them that fall into the slice are collected in an LDS queue, and the queue is then worked off one key per thread — the
// table's round trips (load, CAS, counter) are paid once per workgroup, side by side, instead of once per passing row
// (the first form, 2 K-row blocks and the insert inline: 255 us for C2's 10 M scanned rows; this one: see DESIGN.md).
__global__ __launch_bounds__(1024) void census_kernel(KeyDesc key, int64_t n_rows, int64_t block_stride, uint32_t slice_mask,
                                                      uint64_t *table, uint32_t table_mask, uint32_t *counters, uint32_t *sight, uint32_t *host_out) {
    __shared__ uint64_t q[CENSUS_QCAP];
    __shared__ uint32_t qn, inserted, twice, thrice;
    if (threadIdx.x == 0) { qn = 0; inserted = 0; twice = 0; thrice = 0; }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * block_stride;
#pragma unroll
    for (uint32_t half = 0; half < 2; half++) {
        uint64_t k[CENSUS_RPT / 2];
        bool live[CENSUS_RPT / 2];
#pragma unroll
        for (uint32_t j = 0; j < CENSUS_RPT / 2; j++) {
            const int64_t i = base + (int64_t)((half * (CENSUS_RPT / 2) + j) * 1024 + threadIdx.x);
            live[j] = i < n_rows;
            k[j] = live[j] ? key_cell(key, i) : 0ull;
            live[j] = live[j] && !key_is_null(key, i);
        }
#pragma unroll
        for (uint32_t j = 0; j < CENSUS_RPT / 2; j++) {
            if (live[j] && k[j] != EMPTY_KEY && (hash32(k[j], 0x7F4A7C15u) & slice_mask) == 0u) {
                const uint32_t pos = atomicAdd(&qn, 1u);
                if (pos < CENSUS_QCAP) q[pos] = k[j];
                else census_insert(k[j], 1u, table, table_mask, sight, &inserted, &twice, &thrice);     // (a heavy key in the slice: beyond the queue, inline)
            }
        }
    }
    __syncthreads();
    const uint32_t n_q = min(qn, CENSUS_QCAP);
    for (uint32_t e0 = 0; e0 < n_q; e0 += 1024) {               // (wave-uniform trip count)
        const uint32_t e = e0 + threadIdx.x;
        bool live = e < n_q;
        const uint64_t k = live ? q[e] : 0ull;
        // a wave's repeats of one key are represented by one lane (rows clustered by key; a heavy key that falls into the slice)
        bool peeled = false;
        uint32_t mult = 1;
        for (int round = 0; round < 8; round++) {
            const unsigned long long m = __ballot(live && !peeled);
            if (!m) break;
            const int leader = __ffsll((long long)m) - 1;
            const uint64_t lk = __shfl(k, leader, 64);
            const unsigned long long same = __ballot(live && !peeled && k == lk);
            if (live && !peeled && k == lk) {
                peeled = true;
                if ((int)(threadIdx.x & 63) != leader) live = false;
                else mult = (uint32_t)__popcll(same);
            }
        }
        if (live) census_insert(k, mult, table, table_mask, sight, &inserted, &twice, &thrice);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (inserted) atomicAdd(&counters[0], inserted);
        if (qn) atomicAdd(&counters[1], qn);
        if (twice) atomicAdd(&counters[5], twice);
        if (thrice) atomicAdd(&counters[6], thrice);
        __threadfence();
        if (atomicAdd(&counters[4], 1u) == gridDim.x - 1) {
            __threadfence();
            host_out[0] = __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[1] = __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[2] = __hip_atomic_load(&counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_out[4] = __hip_atomic_load(&counters[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&host_out[3], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------ histogram
// ------------------------------------------------------------------------------------ scan
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_IPT = 4;
constexpr int SCAN_SEG = SCAN_THREADS * SCAN_IPT;

__global__ __launch_bounds__(SCAN_THREADS) void scan_partial_kernel(const uint32_t *in, size_t n,
                                                                    uint32_t *seg_sum) {
    __shared__ uint32_t wt[17];
    size_t base = (size_t)blockIdx.x * SCAN_SEG + (size_t)threadIdx.x * SCAN_IPT;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; j++) if (base + j < n) s += in[base + j];
    uint32_t total;
    block_exclusive_scan<SCAN_THREADS>(s, wt, &total);
    if (threadIdx.x == 0) seg_sum[blockIdx.x] = total;
}
// single workgroup: exclusive scan of up to SCAN_SEG segment sums, in place
__global__ __launch_bounds__(SCAN_THREADS) void scan_top_kernel(uint32_t *seg_sum, uint32_t n_seg) {
    __shared__ uint32_t wt[17];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_seg; base += SCAN_THREADS) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_seg ? seg_sum[i] : 0u, total;
        uint32_t ex = block_exclusive_scan<SCAN_THREADS>(v, wt, &total);
        uint32_t c = carry;
        if (i < n_seg) seg_sum[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
}
// out[i] = exclusive prefix; out[n] = grand total (written by the last segment)
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(const uint32_t *in, size_t n,
                                                                  const uint32_t *seg_base,
                                                                  uint32_t *out) {
    __shared__ uint32_t wt[17];
    size_t base = (size_t)blockIdx.x * SCAN_SEG + (size_t)threadIdx.x * SCAN_IPT;
    uint32_t v[SCAN_IPT], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; j++) { v[j] = base + j < n ? in[base + j] : 0u; s += v[j]; }
    uint32_t total;
    uint32_t ex = block_exclusive_scan<SCAN_THREADS>(s, wt, &total) + seg_base[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_IPT; j++) {
        if (base + j < n) out[base + j] = ex;
        ex += v[j];
        if (base + j == n - 1) out[n] = ex;
    }
}

// ------------------------------------------------------------------------------------ scatter
__device__ __forceinline__ uint64_t move_load(const MoveDesc &m, int64_t i) {
    switch (m.kind) {
    case 0: return reinterpret_cast<const uint64_t *>(m.src)[i];
    case 1: return bit_at(reinterpret_cast<const uint8_t *>(m.src), i) ? 0ull : 1ull;
    case 2: return reinterpret_cast<const uint8_t *>(m.src)[i];
    case 6: return bit_at(reinterpret_cast<const uint8_t *>(m.src), i) ? 1ull : 0ull;   // null bitmap -> null byte
    case 3: case 5: return (uint64_t)i;                                // row index
    default: return reinterpret_cast<const uint32_t *>(m.src)[i];      // 4: u32 source (widened), 7: u32 copy
    }
}
__device__ __forceinline__ void move_store(const MoveDesc &m, uint32_t dst, uint64_t v) {
    switch (m.kind) {
    case 0: case 4: case 5: reinterpret_cast<uint64_t *>(m.dst)[dst] = v; break;
    case 3: case 7: reinterpret_cast<uint32_t *>(m.dst)[dst] = (uint32_t)v; break;
    default: reinterpret_cast<uint8_t *>(m.dst)[dst] = (uint8_t)v;
    }
}

// Row r of a thread's tile slice is tile-local index r*THREADS + tid, clamped to the tile's last
// row so every load below is unconditional (eight back-to-back coalesced loads, no per-row
// branches).  Bases are wave-uniform (column + tbase), indices 32-bit: saddr + voffset addressing.
template <int THREADS>
__device__ __forceinline__ uint32_t tile_idx(uint32_t tid, int r, uint32_t tile_last) {
    return min((uint32_t)(r * THREADS) + tid, tile_last);
}

template <int THREADS, int RPT = SC_RPT>
__device__ __forceinline__ void load_column8(const void *col, int64_t tbase, uint32_t tid, uint32_t tile_last,
                                             uint64_t (&out)[RPT]) {
    // streamed once: non-temporal, so the input does not evict the partially written output lines
    // that the XCD's L2 is completing (shared-cursor write frontier)
    const uint64_t *src = reinterpret_cast<const uint64_t *>(col) + tbase;
#pragma unroll
    for (int r = 0; r < RPT; r++) out[r] = __builtin_nontemporal_load(src + tile_idx<THREADS>(tid, r, tile_last));
}

// key cells + null flags (bit r of *nulls) for the thread's SC_RPT rows; the dtype switch is
// wave-uniform and sits outside the unrolled loads.  tbase is a multiple of 8 (tile aligned), so
// bitmaps are addressed from a byte base.
template <int THREADS, int RPT = SC_RPT>
__device__ __forceinline__ void load_key_cells(const KeyDesc &k, int64_t tbase, uint32_t tid, uint32_t tile_last,
                                               uint64_t (&kc)[RPT], uint32_t *nulls) {
    switch (k.dtype) {
    case PANDRS_HIP_U32CODE: {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(k.data) + tbase;
#pragma unroll
        for (int r = 0; r < RPT; r++) kc[r] = __builtin_nontemporal_load(src + tile_idx<THREADS>(tid, r, tile_last));
        break;
    }
    case PANDRS_HIP_BOOLBITS: {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(k.data) + (tbase >> 3);
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            uint32_t j = tile_idx<THREADS>(tid, r, tile_last);
            kc[r] = (src[j >> 3] >> (j & 7)) & 1;
        }
        break;
    }
    default: {
        const uint64_t *src = reinterpret_cast<const uint64_t *>(k.data) + tbase;
#pragma unroll
        for (int r = 0; r < RPT; r++) kc[r] = __builtin_nontemporal_load(src + tile_idx<THREADS>(tid, r, tile_last));
        if (k.dtype == PANDRS_HIP_F64) {
#pragma unroll
            for (int r = 0; r < RPT; r++)
                if ((kc[r] & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) kc[r] = CANON_NAN;
        }
    }
    }
    uint32_t nm = 0;
    if (k.null_bits) {
        const uint8_t *src = k.null_bits + (tbase >> 3);
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            uint32_t j = tile_idx<THREADS>(tid, r, tile_last);
            nm |= ((src[j >> 3] >> (j & 7)) & 1u) << r;
        }
    } else if (k.null_bytes) {
        const uint8_t *src = k.null_bytes + tbase;
#pragma unroll
        for (int r = 0; r < RPT; r++) nm |= (src[tile_idx<THREADS>(tid, r, tile_last)] ? 1u : 0u) << r;
    }
    *nulls = nm;
}

// LDS: cursor[P+1] | cnt[P+1] | delta[P+1] | wave_tot[32] | pid[TILE] (u16) | stage[TILE] (u64)
// THREADS = 1024: one 8192-row tile per CU (longest per-partition runs);
// THREADS = 512 : 4096-row tiles, two workgroups per CU (loads of one overlap LDS work of the other).
// Workgroup b owns rows [b*chunk, (b+1)*chunk).  hist is partition-major: hist[p*NB + q(b)] with
// q(b) = (b % 8) * (NB/8) + b / 8, so the workgroups of one group g = b % 8 (the set that shares an
// XCD under round-robin dispatch; a label, never a correctness assumption) own ONE contiguous
// region of every partition.  NB is a multiple of 8.
__device__ __forceinline__ uint32_t group_slot(uint32_t b, uint32_t NB) { return (b & 7) * (NB >> 3) + (b >> 3); }

__global__ __launch_bounds__(HI_THREADS) void histogram_kernel(KeyDesc key, int64_t n_rows,
                                                               int64_t chunk, uint32_t P,
                                                               uint32_t seed, uint32_t *hist, uint32_t hash_P, uint32_t part_shift) {
    extern __shared__ uint32_t cnt[];  // P + 1
    const uint32_t NB = gridDim.x, b = blockIdx.x, qb = group_slot(b, NB), tid = threadIdx.x;
    for (uint32_t p = tid; p <= P; p += HI_THREADS) cnt[p] = 0;
    __syncthreads();
    const int64_t beg = (int64_t)b * chunk, end = min(beg + chunk, n_rows);
    // same tiling as the scatter: SC_RPT batched, branch-free key loads per thread and step
    constexpr int TILE = HI_THREADS * SC_RPT;
    for (int64_t tbase = beg; tbase < end; tbase += TILE) {
        const uint32_t tile_n = (uint32_t)min<int64_t>(TILE, end - tbase);
        uint64_t kc[SC_RPT];
        uint32_t nulls;
        load_key_cells<HI_THREADS>(key, tbase, tid, tile_n - 1, kc, &nulls);
#pragma unroll
        for (int r = 0; r < SC_RPT; r++) {
            const bool act = (uint32_t)(r * HI_THREADS) + tid < tile_n;
            const uint32_t p = ((nulls >> r) & 1) ? P : part_of(hash32(kc[r], seed), hash_P ? hash_P : P) >> part_shift;
            // consecutive lanes hold consecutive rows: a run of equal partition ids (rows clustered by
            // key) is counted by its first lane in ONE atomic instead of serialising on one LDS address
            const uint32_t lane = tid & 63;
            const uint32_t pp = __shfl_up(p, 1, 64);
            const unsigned long long am = __ballot(act);
            const bool head = act && (lane == 0 || pp != p || !((am >> (lane - 1)) & 1ull));
            const unsigned long long hm = __ballot(head);
            if (head) {
                const unsigned long long above = lane == 63 ? 0ull : ((hm | ~am) >> (lane + 1));   // next head or first inactive lane
                const uint32_t run = above ? (uint32_t)__ffsll((long long)above) : 64u - lane;
                atomicAdd(&cnt[p], run);
            }
        }
    }
    __syncthreads();
    for (uint32_t p = tid; p <= P; p += HI_THREADS) hist[(size_t)p * NB + qb] = cnt[p];
}

// group cursors: gcur[p*8 + g] = first row of group g's region inside partition p
__global__ void init_group_cursors_kernel(const uint32_t *offsets, uint32_t NB, uint32_t P1, uint32_t *gcur) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P1 * 8) gcur[i] = offsets[(size_t)(i >> 3) * NB + (i & 7) * (NB >> 3)];
}

// One tile of the scatter.  FULL = the tile has all TILE rows (every tile but the input's last):
// no per-row predicates anywhere on that path.
template <int THREADS, bool STAGED, bool FULL, bool CAPPED, int RPT = SC_RPT>
__device__ __forceinline__ void scatter_tile(const ScatterArgs &a, int64_t tbase, uint32_t tile_n,
                                             uint32_t *cursor, uint32_t *cnt, uint32_t *delta,
                                             uint32_t *wave_tot, uint16_t *pid, uint64_t *stage) {
    constexpr uint32_t SPM = (1u << SC_POS_BITS) - 1;
    const uint32_t P1 = a.P + 1, tid = threadIdx.x;
    const uint32_t ipt = (P1 + THREADS - 1) / THREADS;   // partition counters each thread scans
    const uint32_t tile_last = tile_n - 1;
    auto live = [&](int r) { return FULL || (uint32_t)(r * THREADS) + tid < tile_n; };

    for (uint32_t p = tid; p < P1; p += THREADS) cnt[p] = 0;
    block_sync_lds();
    // per row: key cell + packed (partition << SC_POS_BITS | position)
    uint64_t kc[RPT];
    uint32_t ps[RPT], nulls;
    load_key_cells<THREADS, RPT>(a.key, tbase, tid, tile_last, kc, &nulls);
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        bool nul = (nulls >> r) & 1;
        if (nul) kc[r] = 0ull;
        uint32_t p = nul ? a.P : part_of(hash32(kc[r], a.seed), a.hash_P ? a.hash_P : a.P) >> a.part_shift;
        ps[r] = p << SC_POS_BITS;
        if (live(r)) ps[r] |= atomicAdd(&cnt[p], 1u);   // rank inside (tile, partition)
    }
    block_sync_lds();
    // exclusive scan of cnt[] -> delta[] (tile-local partition starts)
    {
        uint32_t first = tid * ipt, s = 0;
        for (uint32_t q = 0; q < ipt; q++) if (first + q < P1) s += cnt[first + q];
        uint32_t ex = block_exclusive_scan<THREADS>(s, wave_tot, nullptr);
        for (uint32_t q = 0; q < ipt; q++)
            if (first + q < P1) { delta[first + q] = ex; ex += cnt[first + q]; }
    }
    block_sync_lds();
#pragma unroll
    for (int r = 0; r < RPT; r++) ps[r] += delta[ps[r] >> SC_POS_BITS];
    block_sync_lds();
    // delta[p] := global cursor - tile-local start, so dst = delta[p] + sorted position.
    // Shared cursors: the 32 CUs of an XCD append to the SAME region of each partition, so a
    // partition's 128-B lines are completed inside that XCD's 4 MiB L2 (frontier = P x columns x
    // a few lines) instead of leaving it half-written from P x 32 private regions.
    if (a.gcur) {
        const uint32_t g = blockIdx.x & 7;
        for (uint32_t p = tid; p < P1; p += THREADS) {
            uint32_t n = cnt[p];
            uint32_t c = n ? atomicAdd(&a.gcur[p * 8 + g], n) : 0u;
            if (CAPPED && n && c + n > a.gend[p * 8 + g]) {          // the sampled capacity was too small: drop the run
                a.flags[0] = 1;
                delta[p] = a.total_cap;                                // dst in [total_cap, total_cap + TILE): the trash tile every
                                                                       // partitioned column carries behind its regions (no guards below)
            } else
            delta[p] = c - delta[p];
        }
    } else {
        for (uint32_t p = tid; p < P1; p += THREADS) {
            uint32_t c = cursor[p];
            delta[p] = c - delta[p];
            cursor[p] = c + cnt[p];
        }
    }
    block_sync_lds();

    if constexpr (STAGED) {
#pragma unroll
        for (int r = 0; r < RPT; r++)
            if (live(r)) { stage[ps[r] & SPM] = kc[r]; pid[ps[r] & SPM] = (uint16_t)(ps[r] >> SC_POS_BITS); }
        // from here kc[] is dead: the first value column's loads go out before the barrier.
        // mv[0 .. n_move8) are 8-byte columns (software-pipelined), the rest byte-wide.
        // one register buffer for the value columns (a second one spilled: 128 VGPRs at 1024 threads): column m + 1's
        // loads are issued right after column m went to LDS and fly under its barrier, linear reads and global stores
        uint64_t v[RPT];
        if (a.n_move8 > 0) load_column8<THREADS, RPT>(a.mv[0].src, tbase, tid, tile_last, v);
        block_sync_lds();
        uint32_t dst[RPT];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            uint32_t j = r * THREADS + tid;
            if (live(r)) { dst[r] = delta[pid[j]] + j; a.pkeys[dst[r]] = stage[j]; }
        }
        for (int m = 0; m < a.n_move8; m++) {
            uint64_t *out = reinterpret_cast<uint64_t *>(a.mv[m].dst);
            block_sync_lds();   // previous column's linear reads done
#pragma unroll
            for (int r = 0; r < RPT; r++) if (live(r)) stage[ps[r] & SPM] = v[r];
            if (m + 1 < a.n_move8) load_column8<THREADS, RPT>(a.mv[m + 1].src, tbase, tid, tile_last, v);
            block_sync_lds();
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                uint32_t j = r * THREADS + tid;
                if (live(r)) out[dst[r]] = stage[j];
            }
        }
        for (int m = a.n_move8; m < a.n_move; m++) {      // validity bytes / byte columns
            const MoveDesc mv = a.mv[m];
            block_sync_lds();
#pragma unroll
            for (int r = 0; r < RPT; r++)
                if (live(r)) stage[ps[r] & SPM] = move_load(mv, tbase + tile_idx<THREADS>(tid, r, tile_last));
            block_sync_lds();
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                uint32_t j = r * THREADS + tid;
                if (live(r)) move_store(mv, dst[r], stage[j]);
            }
        }
        block_sync_lds();
    } else {
        uint32_t dst[RPT];
#pragma unroll
        for (int r = 0; r < RPT; r++)
            if (live(r)) { dst[r] = delta[ps[r] >> SC_POS_BITS] + (ps[r] & SPM); a.pkeys[dst[r]] = kc[r]; }
        for (int m = 0; m < a.n_move; m++) {
            const MoveDesc mv = a.mv[m];
#pragma unroll
            for (int r = 0; r < RPT; r++)
                if (live(r)) move_store(mv, dst[r], move_load(mv, tbase + tile_idx<THREADS>(tid, r, tile_last)));
        }
        block_sync_lds();
    }
}

// The WIDE tile: 16 rows per thread — 16 K rows per 1024-thread workgroup ranked at once, so a (tile, partition) run is twice as
// long as scatter_tile's (at P = 1024: 16 rows = one 128-byte line per column instead of half of one; with five columns the
// scatter of C2 behaves like P = 512's) — moved through the SAME 64 KB staging buffer in two halves of the tile's sorted order:
// a row is staged in the half its sorted position falls in (predicated LDS writes), each half leaves with unit-stride reads.
// Staged layouts only; LDS: counters | wave_tot | pid[16 K] u16 | pos16[16 K] u16 | stage[8 K] u64.
template <int THREADS, bool FULL, bool CAPPED, int RPT = SC_RPT_WIDE>
__device__ __forceinline__ void scatter_tile_wide(const ScatterArgs &a, int64_t tbase, uint32_t tile_n,
                                                  uint32_t *cursor, uint32_t *cnt, uint32_t *delta,
                                                  uint32_t *wave_tot, uint16_t *pid, uint64_t *stage, uint16_t *pos16) {
    constexpr int HR = RPT / 2;                          // rows per thread and half
    constexpr uint32_t HALF = THREADS * HR;              // sorted positions per half = rows of the staging buffer
    constexpr uint32_t SPM = (1u << SC_POS_BITS) - 1;
    static_assert(THREADS * RPT == 2 * HALF, "two halves");
    const uint32_t P1 = a.P + 1, tid = threadIdx.x;
    const uint32_t ipt = (P1 + THREADS - 1) / THREADS;
    const uint32_t tile_last = tile_n - 1;
    auto live = [&](int r) { return FULL || (uint32_t)(r * THREADS) + tid < tile_n; };

    for (uint32_t p = tid; p < P1; p += THREADS) cnt[p] = 0;
    block_sync_lds();
    uint64_t kc[RPT];
    uint32_t ps[RPT], nulls;
    load_key_cells<THREADS, RPT>(a.key, tbase, tid, tile_last, kc, &nulls);
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        bool nul = (nulls >> r) & 1;
        if (nul) kc[r] = 0ull;
        uint32_t p = nul ? a.P : part_of(hash32(kc[r], a.seed), a.hash_P ? a.hash_P : a.P) >> a.part_shift;
        ps[r] = p << SC_POS_BITS;
        if (live(r)) ps[r] |= atomicAdd(&cnt[p], 1u);
    }
    block_sync_lds();
    {
        uint32_t first = tid * ipt, s = 0;
        for (uint32_t q = 0; q < ipt; q++) if (first + q < P1) s += cnt[first + q];
        uint32_t ex = block_exclusive_scan<THREADS>(s, wave_tot, nullptr);
        for (uint32_t q = 0; q < ipt; q++)
            if (first + q < P1) { delta[first + q] = ex; ex += cnt[first + q]; }
    }
    block_sync_lds();
#pragma unroll
    for (int r = 0; r < RPT; r++) ps[r] += delta[ps[r] >> SC_POS_BITS];
    block_sync_lds();
    if (a.gcur) {
        const uint32_t g = blockIdx.x & 7;
        for (uint32_t p = tid; p < P1; p += THREADS) {
            uint32_t n = cnt[p];
            uint32_t c = n ? atomicAdd(&a.gcur[p * 8 + g], n) : 0u;
            if (CAPPED && n && c + n > a.gend[p * 8 + g]) {
                a.flags[0] = 1;
                delta[p] = a.total_cap;                                // the trash tile (SC_TILE_MAX rows behind the regions)
            } else
            delta[p] = c - delta[p];
        }
    } else {
        for (uint32_t p = tid; p < P1; p += THREADS) {
            uint32_t c = cursor[p];
            delta[p] = c - delta[p];
            cursor[p] = c + cnt[p];
        }
    }
    // a row's sorted position goes to LDS too (pos16[row]): the column loop reads it back per column instead of holding 16 more
    // registers per thread (rows of dead lanes get a position outside both halves)
    uint32_t *pos32 = reinterpret_cast<uint32_t *>(pos16);       // rows 2k, 2k + 1 of a thread share one word: 8 reads per staging half
#pragma unroll
    for (int r = 0; r < RPT; r++)
        if (live(r)) pid[ps[r] & SPM] = (uint16_t)(ps[r] >> SC_POS_BITS);
#pragma unroll
    for (int k = 0; k < RPT / 2; k++) {
        const uint32_t lo = live(2 * k) ? (ps[2 * k] & SPM) : 0xFFFFu, hi = live(2 * k + 1) ? (ps[2 * k + 1] & SPM) : 0xFFFFu;
        pos32[k * THREADS + tid] = lo | (hi << 16);
    }
    // ---- keys: two halves.  The destination of sorted position J is delta[pid[J]] + J: two small LDS reads per row and column
    // instead of 16 registers held across the column loop (the LDS pipe has room, the register file has none: with dst[16] kept
    // the kernel spilled 350 VGPRs)
    auto dst_of = [&](uint32_t J) { return delta[pid[J]] + J; };
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int r = 0; r < RPT; r++)
            if (live(r) && ((ps[r] & SPM) >= HALF) == (h == 1)) stage[(ps[r] & SPM) - h * HALF] = kc[r];
        block_sync_lds();                                 // (the first one also orders delta[] and pid[] before their readers)
#pragma unroll
        for (int q = 0; q < HR; q++) {
            const uint32_t j = q * THREADS + tid, J = h * HALF + j;
            if (FULL || J < tile_n) a.pkeys[dst_of(J)] = stage[j];
        }
        block_sync_lds();
    }
    uint64_t v[RPT];
    if (a.n_move8 > 0) load_column8<THREADS, RPT>(a.mv[0].src, tbase, tid, tile_last, v);
    for (int m = 0; m < a.n_move8; m++) {
        uint64_t *out = reinterpret_cast<uint64_t *>(a.mv[m].dst);
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int k = 0; k < RPT / 2; k++) {
                const uint32_t w = pos32[k * THREADS + tid];
                const uint32_t p0 = (w & 0xFFFFu) - h * HALF, p1 = (w >> 16) - h * HALF;      // (unsigned: the other half and dead rows fall outside)
                if (p0 < HALF) stage[p0] = v[2 * k];
                if (p1 < HALF) stage[p1] = v[2 * k + 1];
            }
            if (h == 1 && m + 1 < a.n_move8) load_column8<THREADS, RPT>(a.mv[m + 1].src, tbase, tid, tile_last, v);
            block_sync_lds();
#pragma unroll
            for (int q = 0; q < HR; q++) {
                const uint32_t j = q * THREADS + tid, J = h * HALF + j;
                if (FULL || J < tile_n) out[dst_of(J)] = stage[j];
            }
            block_sync_lds();
        }
    }
    for (int m = a.n_move8; m < a.n_move; m++) {          // validity bytes / byte columns
        const MoveDesc mv = a.mv[m];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const uint32_t pos = ((pos32[(r >> 1) * THREADS + tid] >> ((r & 1) * 16)) & 0xFFFFu) - h * HALF;
                if (pos < HALF) stage[pos] = move_load(mv, tbase + tile_idx<THREADS>(tid, r, tile_last));
            }
            block_sync_lds();
#pragma unroll
            for (int q = 0; q < HR; q++) {
                const uint32_t j = q * THREADS + tid, J = h * HALF + j;
                if (FULL || J < tile_n) move_store(mv, dst_of(J), stage[j]);
            }
            block_sync_lds();
        }
    }
}

template <int THREADS, bool STAGED, bool CAPPED = false, int RPT = SC_RPT>
__global__ __launch_bounds__(THREADS, 4) void scatter_kernel(ScatterArgs a) {
    constexpr int TILE = THREADS * RPT;
    constexpr bool WIDE = TILE == 1024 * SC_RPT_WIDE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t P1 = a.P + 1;
    // LDS: [cursor[P1] only with private cursors] | cnt[P1] | delta[P1] | wave_tot[32] | pid[TILE] | stage[TILE]
    uint32_t *cursor = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cnt = a.gcur ? cursor : cursor + P1;
    uint32_t *delta = cnt + P1;
    uint32_t *wave_tot = delta + P1;
    uint16_t *pid = reinterpret_cast<uint16_t *>(wave_tot + 32);
    uint16_t *pos16 = pid + TILE;                        // (wide tile only)
    uint64_t *stage = reinterpret_cast<uint64_t *>(
        (reinterpret_cast<uintptr_t>(pid + (WIDE ? 2 * TILE : TILE)) + 15) & ~uintptr_t(15));

    const uint32_t NB = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    if (!a.gcur) for (uint32_t p = tid; p < P1; p += THREADS) cursor[p] = a.offsets[(size_t)p * NB + group_slot(b, NB)];
    if (CAPPED) {
        // capacity mode: tiles are dealt round-robin, so group g = b % 8 takes every 8th tile of the input and
        // its share of a partition is 1/8 whenever the key distribution is stationary over 8 tiles (64 K rows)
        for (int64_t tbase = (int64_t)b * TILE; tbase < a.n_rows; tbase += (int64_t)NB * TILE) {
            const uint32_t tile_n = (uint32_t)min<int64_t>(TILE, a.n_rows - tbase);
            if constexpr (WIDE) {
                if (tile_n == TILE) scatter_tile_wide<THREADS, true, true, RPT>(a, tbase, tile_n, cursor, cnt, delta, wave_tot, pid, stage, pos16);
                else {                                       // the input's last tile: as ordinary tiles
                    constexpr uint32_t T8 = THREADS * SC_RPT;
                    for (uint32_t o = 0; o < tile_n; o += T8)
                        scatter_tile<THREADS, true, false, true, SC_RPT>(a, tbase + o, min(tile_n - o, T8), cursor, cnt, delta, wave_tot, pid, stage);
                }
            } else if (tile_n == TILE)
                scatter_tile<THREADS, STAGED, true, true, RPT>(a, tbase, tile_n, cursor, cnt, delta, wave_tot, pid, stage);
            else
                scatter_tile<THREADS, STAGED, false, true, RPT>(a, tbase, tile_n, cursor, cnt, delta, wave_tot, pid, stage);
        }
        return;
    }
    const int64_t beg = (int64_t)b * a.chunk, end = min(beg + a.chunk, a.n_rows);
    for (int64_t tbase = beg; tbase < end; tbase += TILE) {
        const uint32_t tile_n = (uint32_t)min<int64_t>(TILE, end - tbase);
        if constexpr (WIDE) {
            if (tile_n == TILE) scatter_tile_wide<THREADS, true, false, RPT>(a, tbase, tile_n, cursor, cnt, delta, wave_tot, pid, stage, pos16);
            else {
                constexpr uint32_t T8 = THREADS * SC_RPT;
                for (uint32_t o = 0; o < tile_n; o += T8)
                    scatter_tile<THREADS, true, false, false, SC_RPT>(a, tbase + o, min(tile_n - o, T8), cursor, cnt, delta, wave_tot, pid, stage);
            }
        } else if (tile_n == TILE)
            scatter_tile<THREADS, STAGED, true, false, RPT>(a, tbase, tile_n, cursor, cnt, delta, wave_tot, pid, stage);
        else
            scatter_tile<THREADS, STAGED, false, false, RPT>(a, tbase, tile_n, cursor, cnt, delta, wave_tot, pid, stage);
    }
}

// ---- capacity mode: regions from a sample instead of the exact histogram ---------------------------
constexpr uint32_t SAMPLE_BLOCK = 1024, SAMPLE_PERIOD = 8;     // rows of every 8th 1024-row block are counted
__global__ __launch_bounds__(1024) void sample_histogram_kernel(KeyDesc key, int64_t n_rows, uint32_t P, uint32_t seed,
                                                                uint32_t *hist /* [SAMPLE_REPL][P + 2]; [P + 1] = rows sampled */) {
    extern __shared__ uint32_t cnt[];  // P + 1
    const uint32_t tid = threadIdx.x;
    for (uint32_t p = tid; p <= P; p += 1024) cnt[p] = 0;
    __syncthreads();
    const int64_t n_blocks = (n_rows + (int64_t)SAMPLE_BLOCK * SAMPLE_PERIOD - 1) / ((int64_t)SAMPLE_BLOCK * SAMPLE_PERIOD);
    uint32_t mine = 0;
    constexpr int U = 4;                // sample blocks in flight per workgroup step
    for (int64_t j0 = (int64_t)blockIdx.x * U; j0 < n_blocks; j0 += (int64_t)gridDim.x * U) {
        uint64_t kc[U];
        bool nul[U], live[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = (j0 + u) * SAMPLE_BLOCK * SAMPLE_PERIOD + tid;
            live[u] = j0 + u < n_blocks && i < n_rows;
            const int64_t ic = live[u] ? i : 0;
            kc[u] = key_cell(key, ic);
            nul[u] = key_is_null(key, ic);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (live[u]) {
                atomicAdd(&cnt[nul[u] ? P : part_of(hash32(kc[u], seed), P)], 1u);
                mine++;
            }
        }
    }
    __syncthreads();
    uint32_t *myhist = hist + (size_t)(blockIdx.x % SAMPLE_REPL) * (P + 2);
    for (uint32_t p = tid; p <= P; p += 1024) if (cnt[p]) atomicAdd(&myhist[p], cnt[p]);
    const unsigned long long any = __ballot(mine != 0);
    (void)any;
    uint32_t w = mine;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) w += __shfl_down(w, d, 64);
    if ((tid & 63) == 0 && w) atomicAdd(&myhist[P + 1], w);
}

// one workgroup: region (p, g) = [gbeg, gend), 16-row aligned (128-byte lines), capacity = the partition's sampled
// share scaled up + 6 sigma of the sampling noise + 6 sigma of the split over the 8 groups + a constant
// `clump`: rows that travel together (the fused join's pairs: all pairs of one build row land in one region) widen the split's noise
__global__ __launch_bounds__(1024) void plan_regions_kernel(const uint32_t *hist, int64_t n_rows,
                                                            uint32_t P1, uint32_t total_cap, uint32_t *gbeg, uint32_t *gcur,
                                                            uint32_t *gend, uint32_t *flags, double clump) {
    __shared__ uint32_t wt[17];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    uint32_t n_sampled = 0;
#pragma unroll
    for (uint32_t r = 0; r < SAMPLE_REPL; r++) n_sampled += hist[(size_t)r * (P1 + 1) + P1];      // (unrolled: 16 loads in flight, not 16 round trips)
    const double scale = (double)n_rows / (double)max(n_sampled, 1u);
    for (uint32_t base = 0; base < P1; base += 1024) {
        const uint32_t p = base + threadIdx.x;
        uint32_t cap = 0;
        if (p < P1) {
            uint32_t ci = 0;
#pragma unroll
            for (uint32_t r = 0; r < SAMPLE_REPL; r++) ci += hist[(size_t)r * (P1 + 1) + p];
            const double c = (double)ci;
            const double share = (c + 6.0 * sqrt(c) + 4.0) * scale * 0.125;
            const double want = share + 6.0 * sqrt(share * clump) + 32.0;
            cap = want >= 4.0e9 ? 0xFFFFFFF0u : ((uint32_t)want + 15u) & ~15u;
        }
        uint32_t tot;
        // 8 regions per partition; saturate instead of wrapping when the plan is absurd (flagged below)
        const uint32_t mine = cap > 0x0FFFFFFFu ? 0x7FFFFFF8u : cap * 8u;
        const uint32_t ex = block_exclusive_scan<1024>(mine, wt, &tot) + carry;
        const bool fits = (uint64_t)ex + mine <= total_cap;
        if (p < P1) {
            for (uint32_t g = 0; g < 8; g++) {
                const uint32_t b = fits ? ex + g * cap : 0u;
                gbeg[p * 8 + g] = b; gcur[p * 8 + g] = b; gend[p * 8 + g] = fits ? b + cap : 0u;
            }
            if (!fits) flags[0] = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry = (uint64_t)carry + tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : carry + tot;
        __syncthreads();
    }
}

size_t scan_seg_count(size_t n) { return (n + SCAN_SEG - 1) / SCAN_SEG + 16; }

int32_t exclusive_scan_u32(pandrs_hip_ctx *c, const uint32_t *in, size_t n, uint32_t *out,
                                  uint32_t *seg) {
    uint32_t n_seg = (uint32_t)((n + SCAN_SEG - 1) / SCAN_SEG);
    hipLaunchKernelGGL(scan_partial_kernel, dim3(n_seg), dim3(SCAN_THREADS), 0, c->stream, in, n, seg);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_THREADS), 0, c->stream, seg, n_seg);
    hipLaunchKernelGGL(scan_final_kernel, dim3(n_seg), dim3(SCAN_THREADS), 0, c->stream, in, n, seg, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

constexpr uint32_t EST_SLOTS = 1u << 19;                  // >= 2 x the largest sample
// the estimate's block: [EST_SLOTS] u64 keys | 256 B counters | [EST_SLOTS] u32 per-key counts | coverage histogram | hot-key image
static uint32_t *est_counts(pandrs_hip_ctx *c) { return reinterpret_cast<uint32_t *>(c->est_table + EST_SLOTS) + 64; }
static uint32_t *est_cov_hist(pandrs_hip_ctx *c) { return est_counts(c) + EST_SLOTS; }      // [2 * COV_BINS + 1] + threshold words at + 2 * COV_BINS + 16
constexpr uint32_t EST_IMAGE_SLOTS = 1u << 15;            // the hot-key image (absorb pass): as many slots as an LDS table can have
static uint64_t *est_image(pandrs_hip_ctx *c) { return reinterpret_cast<uint64_t *>(est_cov_hist(c) + 2 * COV_BINS + 64); }
static uint32_t *est_sight(pandrs_hip_ctx *c) { return reinterpret_cast<uint32_t *>(est_image(c) + EST_IMAGE_SLOTS); }      // [EST_SLOTS] saturating sighting counters

// ---- keys that are LOCAL in position (nearly sorted input: event times arriving slightly out of order) -------------------------
// One sampled row per stride cannot see them repeat — a key's rows all lie inside one or two strides — so the sample reads "all
// distinct" and the model extrapolates far beyond the truth (100 rows per key shuffled within +-50 rows: 8.2 M for 1 M groups; 10 rows
// per key: 91 M for 10 M, the two-level path); the run bound does not help either, neighbours mostly differ.  What does: the distinct
// keys inside WINDOWS of consecutive rows.  Every key lies in at least one window, so (windows in the column) x (mean distinct keys
// per window) bounds the group count from above in ANY row order — loosely for random order (every window all distinct: the bound is N),
// tightly when keys are local.  256 windows of 4096 rows: 8 MB read, one LDS hash set per workgroup.
constexpr uint32_t WIN_ROWS = 4096, WIN_COUNT = 256, WIN_SLOTS = 8192;
__global__ __launch_bounds__(1024) void window_distinct_kernel(KeyDesc key, int64_t n_rows, int64_t win_stride, uint32_t *out /* [0] distinct, [1] rows */) {
    __shared__ uint64_t set[WIN_SLOTS];
    __shared__ uint32_t cnt[3];
    const uint32_t tid = threadIdx.x;
    for (uint32_t s = tid; s < WIN_SLOTS; s += 1024) set[s] = EMPTY_KEY;
    if (tid < 3) cnt[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * win_stride;
    uint32_t fresh = 0, rows = 0;
    bool saw_null = false, saw_sentinel = false;
    for (uint32_t r = tid; r < WIN_ROWS; r += 1024) {
        const int64_t i = base + r;
        if (i >= n_rows) break;
        rows++;
        if (key_is_null(key, i)) { saw_null = true; continue; }
        const uint64_t k = key_cell(key, i);
        if (k == EMPTY_KEY) { saw_sentinel = true; continue; }
        uint32_t slot = hash32(k, 0x3C6EF372u) & (WIN_SLOTS - 1);
        for (;;) {                                   // (4096 keys at most in 8192 slots: the set never fills)
            const uint64_t old = atomicCAS((unsigned long long *)&set[slot], EMPTY_KEY, k);
            if (old == EMPTY_KEY) { fresh++; break; }
            if (old == k) break;
            slot = (slot + 1) & (WIN_SLOTS - 1);
        }
    }
    if (saw_null) cnt[1] = 1;
    if (saw_sentinel) cnt[2] = 1;
    uint32_t w = fresh, rw = rows;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { w += __shfl_down(w, d, 64); rw += __shfl_down(rw, d, 64); }
    if ((tid & 63) == 0 && rw) { atomicAdd(&cnt[0], w); atomicAdd(&out[1], rw); }
    __syncthreads();
    if (tid == 0) atomicAdd(&out[0], cnt[0] + cnt[1] + cnt[2]);
}

// The census's own table (kept armed like the first stage's: cleared BEHIND a census, not in front of the next one).
// -> *out_est = 0 when the slice held too few keys to say anything.
static int32_t census_groups(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, double *out_est) {
    constexpr uint32_t slots = EST_SLOTS;
    *out_est = 0.0;
    if (!c->census_table) {
        const size_t bytes = size_t(slots) * 8 + 256 + size_t(slots) * 4;
        HIP_TRY(hipMalloc((void **)&c->census_table, bytes));
        alloc_events()++;
        hipLaunchKernelGGL(estimate_clear_kernel, dim3(slots / 256), dim3(256), 0, c->stream, c->census_table, slots,
                           reinterpret_cast<uint32_t *>(c->census_table + slots), (uint32_t *)nullptr, reinterpret_cast<uint32_t *>(c->census_table + slots) + 64);
    }
    uint32_t *counters = reinterpret_cast<uint32_t *>(c->census_table + slots), *sight = counters + 64;
    // a tenth of the rows, at least 4 M (all of a smaller column), in 16 K-row blocks spread evenly; slice width so that <= ~160 K rows pass
    const int64_t n_scan = std::min<int64_t>(n_rows, std::max<int64_t>(n_rows / 10, int64_t(4) << 20));
    const int64_t n_blocks = std::max<int64_t>(1, n_scan / CENSUS_BLOCK_ROWS);
    const int64_t block_stride = std::max<int64_t>(CENSUS_BLOCK_ROWS, (n_rows / n_blocks) & ~int64_t(15));
    uint32_t R = 16;
    while ((double)n_blocks * CENSUS_BLOCK_ROWS / R > 160000.0 && R < (1u << 16)) R *= 2;
    volatile uint32_t *h = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1200;
    h[3] = 0;
    hipLaunchKernelGGL(census_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, c->stream, key, n_rows, block_stride, R - 1, c->census_table, slots - 1,
                       counters, sight, const_cast<uint32_t *>(h));
    HIP_TRY(hipGetLastError());
    for (int spin = 0; spin < 400000 && h[3] != 1; spin++) __builtin_ia32_pause();
    if (h[3] != 1) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h[3] != 1) return fail(PANDRS_HIP_ERR_COMPUTATION, "census kernel did not publish its counters");
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const double d = h[0], rows = h[1], twice = h[2], thrice = h[4];
    hipLaunchKernelGGL(estimate_clear_kernel, dim3(slots / 256), dim3(256), 0, c->stream, c->census_table, slots, counters, (uint32_t *)nullptr, sight);
    const double scanned = (double)n_blocks * CENSUS_BLOCK_ROWS;
    if (d >= 256.0 && d < 0.45 * slots) {
        const double f1 = std::max(0.0, d - twice), f2 = std::max(0.0, twice - thrice);
        // Chao1 over the slice: from the scanned rows to all rows; every row scanned: the slice's distinct count is exact
        const double in_slice = scanned >= (double)n_rows ? d : d + f1 * std::max(0.0, f1 - 1.0) / (2.0 * (f2 + 1.0));
        *out_est = std::min((double)R * in_slice, (double)n_rows);
    }
    if (std::getenv("PANDRS_HIP_ENGINE_TRACE"))
        fprintf(stderr, "[census] scanned %.0f rows in %lld blocks, slice 1/%u: %.0f rows passed, %.0f distinct, %.0f twice, %.0f thrice -> %.0f groups\n",
                scanned, (long long)n_blocks, R, rows, d, twice, thrice, *out_est);
    return 0;
}

int32_t estimate_groups(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, int64_t *out_est, bool keep_table) {
    PhaseTimer pt(c, PANDRS_HIP_PHASE_ESTIMATE);
    const int64_t n_sample = std::min<int64_t>(n_rows, 1 << 18);
    const int64_t stride = n_rows / n_sample;
    constexpr uint32_t slots = EST_SLOTS;
    // a dedicated block that stays armed (table = EMPTY, counters = 0) between calls: cleared behind the previous
    // estimate instead of in front of this one
    if (!c->est_table) {
        const size_t bytes = size_t(slots) * 8 + 256 + size_t(slots) * 4 + (2 * COV_BINS + 64) * 4 + EST_IMAGE_SLOTS * 8 + size_t(slots) * 4;
        HIP_TRY(hipMalloc((void **)&c->est_table, bytes));
        alloc_events()++;
        HIP_TRY(hipMemsetAsync(c->est_table, 0, bytes, c->stream));
        uint32_t *cnt0 = reinterpret_cast<uint32_t *>(c->est_table + slots);
        hipLaunchKernelGGL(estimate_clear_kernel, dim3(slots / 256), dim3(256), 0, c->stream, c->est_table, slots, cnt0, est_counts(c), est_sight(c));
    }
    uint64_t *table = c->est_table;
    uint32_t *distinct = reinterpret_cast<uint32_t *>(c->est_table + slots);
    volatile uint32_t *h = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1024;     // own corner of the pinned block
    h[3] = 0;
    hipLaunchKernelGGL(estimate_kernel, dim3((unsigned)((n_sample + 1023) / 1024)), dim3(1024), 0, c->stream,
                       key, n_rows, stride, n_sample, table, slots - 1, distinct, const_cast<uint32_t *>(h), est_sight(c));
    HIP_TRY(hipGetLastError());
    // poll the pinned word the kernel's last block sets (a PCIe write away) before falling back to the runtime's
    // wait, which costs tens of microseconds to wake up
    for (int spin = 0; spin < 200000 && h[3] != 1; spin++) __builtin_ia32_pause();
    if (h[3] != 1) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h[3] != 1) return fail(PANDRS_HIP_ERR_COMPUTATION, "estimate kernel did not publish its counters");
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const uint32_t hv[3] = {h[0], h[1], h[2]};
    const double twice = h[4], thrice = h[5], d_sub = h[6];
    double s_sub = 0.0;                                  // rows of the sub-sample (every fourth 1024-row block of the sample)
    for (int64_t b = 0; b * 1024 < n_sample; b += 4) s_sub += (double)std::min<int64_t>(1024, n_sample - b * 1024);
    c->est_kept = keep_table;
    if (!keep_table) hipLaunchKernelGGL(estimate_clear_kernel, dim3(slots / 256), dim3(256), 0, c->stream, table, slots, distinct, (uint32_t *)nullptr, est_sight(c));
    double d = std::max<uint32_t>(hv[0], 1), s = (double)n_sample;
    double est;
    bool census_wanted = false;
    c->est_repeat_share = 0.0;
    if (n_sample == n_rows) est = d;
    else if (s - d < 64.0) est = (double)n_rows;                  // too few repeats in the sample to measure: (nearly) all distinct
    else {
        // uniform-occupancy model d = G (1 - exp(-s/G)); Newton on G
        double G = d;
        for (int it = 0; it < 50; it++) {
            double e = std::exp(-s / G), f = G * (1 - e) - d, fp = 1 - e - (s / G) * e;
            if (std::fabs(fp) < 1e-12) break;
            double Gn = G - f / fp;
            if (!(Gn > 0)) break;
            if (std::fabs(Gn - G) < 1e-6 * G) { G = Gn; break; }
            G = Gn;
        }
        // Chao1 from the sample's singletons and doubletons: d + f1 (f1 - 1) / (2 (f2 + 1)).  Equal to the model on uniform keys
        // (1 M groups: 0.998 M); on a few hot keys in front of a long tail it sees the tail the model cannot (2 K hot keys holding
        // 80 % of the rows + 1 M others: model 60 K, Chao1 0.96 M).
        const double f1 = std::max(0.0, d_sub - twice), f2 = std::max(0.0, twice - thrice);
        // share of the sub-sample's rows on keys sighted three times or more (s - f1 - 2 f2): uniform 1 M groups ~1 %, 2 K hot keys with 80 % of the rows ~75 %
        c->est_repeat_share = s_sub > 0 ? std::max(0.0, s_sub - f1 - 2.0 * f2) / s_sub : 0.0;
        const double chao = d_sub + f1 * std::max(0.0, f1 - 1.0) / (2.0 * (f2 + 1.0));
        // (only where it disagrees by more than its own noise — f2 is a few hundred keys of a 64 K-row sub-sample at 20 M groups —
        // so that uniform keys keep the model's steadier figure)
        if (!c->opt.no_chao && chao > 1.3 * G) G = chao;
        est = std::min<double>(std::max(G, d), (double)n_rows);
        // An unresolved tail?  Keys sighted three times or more against doubletons give the rate of the class that REPEATS in the
        // sub-sample (Poisson: P(>= 3) / P(2) is a function of the rate alone), and that class explains 2 f2 / rate singletons.
        // Singletons far beyond that belong to keys the sample cannot count — rows of a tail whose size neither the model nor Chao1
        // (a lower bound) can know.  When they are a real share of the rows, the second stage (a hash-slice census) sizes the column.
        // Uniform keys never get here: their singletons are what their own doubletons predict (or f3 is too small to tell, and then
        // the model's figure stands).
        census_wanted = false;
        // (few triple sightings are still evidence when they are many more than the doubletons predict: sigma below carries their noise)
        if (!c->opt.no_census && thrice >= 16.0 && f2 >= 200.0 && n_rows >= (int64_t(1) << 22)) {
            const double ratio = thrice / f2;                    // = (e^r - 1 - r - r^2 / 2) / (r^2 / 2)
            double lo = 1e-4, hi = 30.0;
            for (int it = 0; it < 60; it++) {
                const double r = 0.5 * (lo + hi), val = (std::expm1(r) - r - 0.5 * r * r) / (0.5 * r * r);
                (val < ratio ? lo : hi) = r;
            }
            const double rate = 0.5 * (lo + hi), f1_explained = 2.0 * f2 / rate;
            const double sigma = f1_explained * std::sqrt(1.0 / thrice + 1.0 / f2);      // (the rate's own noise, first order)
            const double excess = f1 - f1_explained;
            // (a loose bar: a false alarm on uniform keys — a 2.5 sigma event — costs one census, a miss costs an overflow run)
            census_wanted = excess > 0.06 * s_sub && excess > 2.5 * sigma;
            if (std::getenv("PANDRS_HIP_ENGINE_TRACE"))
                fprintf(stderr, "[estimate] repeat class: rate %.3f explains %.0f of %.0f singletons (sigma %.0f) -> census %d\n", rate, f1_explained, f1, sigma, (int)census_wanted);
        }
    }
    double run_bound = (double)n_rows;
    c->clustered_rows = false;
    c->est_far_same = 0.0; c->est_far_equal = 0;
    c->est_near_same = hv[2] >= 1024 ? 1.0 - (double)hv[1] / (double)hv[2] : 0.0;
    if (hv[2] >= 1024) {
        // most neighbours share their key — and far more often than rows far apart do (a dominant key alone puts equal keys next to
        // each other in any order: 90 % of the rows on one key read as "clustered", which kept the lean aggregate and the absorb pass away)
        const double near_same = 1.0 - (double)hv[1] / (double)hv[2];
        const double far_same = h[8] >= 256 ? 1.0 - (double)h[7] / (double)h[8] : 0.0;
        c->clustered_rows = near_same > 0.5 && far_same < 0.5 * near_same && !c->opt.no_runs;
        c->est_far_same = far_same; c->est_far_equal = h[8] >= 256 ? (int64_t)h[8] - (int64_t)h[7] : 0;
        // runs of equal keys: an upper bound on the group count in any row order (exact for sorted
        // input); + 3 sigma of the sampled share so that noise cannot push it below the truth
        const double pairs = (double)hv[2], b = (double)hv[1];
        const double share = std::min(1.0, (b + 3.0 * std::sqrt(b + 1.0)) / pairs);
        run_bound = std::max(d, share * (double)(n_rows - 1) + 1.0);
        est = std::min(est, run_bound);
    }
    // keys local in position (see window_distinct_kernel): looked for when neighbours share their key far more often than chance allows
    // although the sample is (nearly) all distinct — i.e. the estimate is an extrapolation — and the run bound did not already settle it
    c->clumped_rows = false;
    // (neighbours equal 20 x more often than `est` equally likely keys would make them: a dominant key does that too — its windows then
    // simply look like the sample, see below — at the price of this one small kernel and its read-back)
    if (n_sample < n_rows && n_rows >= (int64_t(1) << 22) && c->est_near_same * est > 20.0 && !(c->clustered_rows && c->est_near_same > 0.87) && !c->opt.no_window_bound) {
        const int64_t win_stride = std::max<int64_t>(WIN_ROWS, (n_rows / WIN_COUNT) & ~int64_t(15));
        const uint32_t n_win = (uint32_t)std::min<int64_t>(WIN_COUNT, (n_rows + win_stride - 1) / win_stride);
        uint32_t *wout = distinct + 16;                   // (two spare words of the estimate's counter block)
        HIP_TRY(hipMemsetAsync(wout, 0, 8, c->stream));
        hipLaunchKernelGGL(window_distinct_kernel, dim3(n_win), dim3(1024), 0, c->stream, key, n_rows, win_stride, wout);
        uint32_t *hw = reinterpret_cast<uint32_t *>(c->pinned) + 1232;
        HIP_TRY(hipMemcpyAsync(hw, wout, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (hw[1] > 0) {
            const double per_row = (double)hw[0] / (double)hw[1];                   // distinct keys per row inside a window
            // + 3 sigma of the windows' spread, taken as Poisson on the count, and the windows' edges (a key cut by a window's edge counts twice)
            const double bound = (double)n_rows * std::min(1.0, (hw[0] + 3.0 * std::sqrt((double)hw[0] + 1.0)) / (double)hw[1]);
            // what windows of a RANDOMLY ordered column would hold: at least the sub-sample's share of distinct keys (65536 rows taken at
            // random repeat their keys more often than 4096 do, whatever the key distribution); far fewer = keys local in position
            const double expect = s_sub > 0 ? d_sub / s_sub : 1.0;
            c->clumped_rows = per_row < 0.5 * expect;
            if (std::getenv("PANDRS_HIP_ENGINE_TRACE"))
                fprintf(stderr, "[estimate] windows: %u distinct keys in %u rows of %u windows -> bound %.0f (estimate before %.0f); random order would show %.3f per row: clumped %d\n",
                        hw[0], hw[1], n_win, bound, est, expect, (int)c->clumped_rows);
            if (bound * 1.5 < est) est = std::max(bound, d);             // (a bound in any row order; taken when it says something)
        }
    }
    c->timings_census = 0;
    if (census_wanted && !c->clustered_rows) {
        double est2 = 0.0;
        ST_TRY(census_groups(c, key, n_rows, &est2));
        est2 = std::min(est2, run_bound);
        if (est2 > est) { est = est2; c->timings_census = 1; }
    }
    if (std::getenv("PANDRS_HIP_ENGINE_TRACE"))
        fprintf(stderr, "[estimate] rows %lld sample %lld distinct %u | adjacent pairs %u differing %u (far pairs %u differing %u) | sub-sample %0.f distinct %0.f twice %0.f thrice %0.f | estimate %0.f clustered %d\n",
                (long long)n_rows, (long long)n_sample, hv[0], hv[2], hv[1], h[8], h[7], s_sub, d_sub, twice, thrice, est, (int)c->clustered_rows);
    *out_est = (int64_t)est;
    return 0;
}

// Releases a table kept by estimate_groups(.., keep_table = true) without using it.
void estimate_release(pandrs_hip_ctx *c) {
    if (!c->est_kept) return;
    c->est_kept = false;
    hipLaunchKernelGGL(estimate_clear_kernel, dim3(EST_SLOTS / 256), dim3(256), 0, c->stream, c->est_table, EST_SLOTS,
                       reinterpret_cast<uint32_t *>(c->est_table + EST_SLOTS), (uint32_t *)nullptr, est_sight(c));
}

// Share of the rows (by the kept sample) that belongs to the `budget` most frequent keys; releases the table.
// image_T > 0 and a share of at least min_share: *out_image = the hot-key image for an absorb table of image_T slots (hot_image_kernel;
// valid until the next estimate on this context), else nullptr.
int32_t estimate_coverage(pandrs_hip_ctx *c, const KeyDesc &key, int64_t n_rows, int64_t budget, double *out_share,
                          int64_t image_T, uint32_t image_seed, double min_share, const uint64_t **out_image) {
    *out_share = 0.0;
    if (out_image) *out_image = nullptr;
    if (!c->est_kept) return 0;
    PhaseTimer pt(c, PANDRS_HIP_PHASE_ESTIMATE);
    const int64_t n_sample = std::min<int64_t>(n_rows, 1 << 18);
    const int64_t stride = n_rows / n_sample;
    volatile uint32_t *h = reinterpret_cast<volatile uint32_t *>(c->pinned) + 1056;     // own corner of the pinned block
    h[3] = 0;
    hipLaunchKernelGGL(estimate_count_kernel, dim3((unsigned)((n_sample + 1023) / 1024)), dim3(1024), 0, c->stream,
                       key, n_rows, stride, n_sample, c->est_table, EST_SLOTS - 1, est_counts(c));
    hipLaunchKernelGGL(estimate_coverage_kernel, dim3(64), dim3(1024), 0, c->stream, c->est_table, est_counts(c), EST_SLOTS,
                       (uint32_t)std::min<int64_t>(std::max<int64_t>(budget, 1), 0x7FFFFFFF), est_cov_hist(c), est_cov_hist(c) + 2 * COV_BINS + 16,
                       const_cast<uint32_t *>(h));
    HIP_TRY(hipGetLastError());
    for (int spin = 0; spin < 200000 && h[3] != 1; spin++) __builtin_ia32_pause();
    if (h[3] != 1) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (h[3] != 1) return fail(PANDRS_HIP_ERR_COMPUTATION, "coverage kernel did not publish its counters");
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const uint32_t h0 = h[0], h2 = h[2];
    const double covered = h0, all = std::max<uint32_t>(h2, 1);
    if (out_image && image_T >= 16 && image_T <= (int64_t)EST_IMAGE_SLOTS && covered / all >= min_share) {
        HIP_TRY(hipMemsetAsync(est_image(c), 0xFF, (size_t)image_T * 8, c->stream));
        for (int band = 0; band < 3; band++)
            hipLaunchKernelGGL(hot_image_kernel, dim3(256), dim3(1024), 0, c->stream, c->est_table, est_counts(c), EST_SLOTS,
                               est_cov_hist(c) + 2 * COV_BINS + 16, est_image(c), (uint32_t)image_T, image_seed, band);
        *out_image = est_image(c);
    }
    c->est_kept = false;
    hipLaunchKernelGGL(estimate_clear_kernel, dim3(EST_SLOTS / 256), dim3(256), 0, c->stream, c->est_table, EST_SLOTS,
                       reinterpret_cast<uint32_t *>(c->est_table + EST_SLOTS), est_counts(c), est_sight(c));
    *out_share = covered / all;
    return 0;
}

// the wide tile (scatter_tile_wide) pays when the (tile, partition) runs of the ordinary tile are shorter than a 128-byte line, i.e.
// from a fan-out of ~1 K up, and its LDS fits; "scatter_wide": 1 = whenever it fits, -1 = never
static bool scatter_wide_ok(const pandrs_hip_ctx *c, const ScatterArgs &sa, bool shared_cursors, bool staged) {
    if (c->opt.scatter_wide < 0 || !staged || c->opt.scatter_threads == 512) return false;
    int n8 = 0;
    for (int i = 0; i < sa.n_move; i++) n8 += sa.mv[i].kind == 0;
    // measured (A/B on one box): C2 (5 columns, P = 1024) scatter 2.11 -> 2.04 ms, C4 whole (2 columns, P = 1792) 10.15 -> 9.79; at small
    // fan-outs the runs are long already and the two staging halves only cost (C5 shard's 64-bucket pass 1.66 -> 1.85)
    if (c->opt.scatter_wide == 0 && (sa.P < 1024 || (n8 < 2 && sa.P < 1536))) return false;
    const size_t lds = (size_t)(sa.P + 1) * (shared_cursors ? 8 : 12) + 32 * 4 + (size_t)1024 * SC_RPT_WIDE * 4 + 16 + (size_t)1024 * SC_RPT * 8;
    // (and only with several tiles per CU: a 1 M-record merge has 61 wide tiles for 256 CUs — its scatter took 0.17 ms instead of 0.08)
    return lds <= 160 * 1024 && sa.n_rows >= (c->opt.scatter_wide > 0 ? 4 : 1024) * 1024 * (int64_t)SC_RPT_WIDE;
}

template <int THREADS>
static int32_t launch_scatter(pandrs_hip_ctx *c, const ScatterArgs &sa, uint32_t NB, bool staged, bool wide = false) {
    constexpr int TILE = THREADS * SC_RPT;
    if (wide && THREADS == 1024) {
        const size_t ldsw = (size_t)(sa.P + 1) * (sa.gcur ? 8 : 12) + 32 * 4 + (size_t)THREADS * SC_RPT_WIDE * 4 + 16 + (size_t)TILE * 8;
        if (sa.gend) {
            ST_TRY(set_max_lds(scatter_kernel<1024, true, true, SC_RPT_WIDE>, (int)ldsw));
            hipLaunchKernelGGL((scatter_kernel<1024, true, true, SC_RPT_WIDE>), dim3(NB), dim3(1024), ldsw, c->stream, sa);
        } else {
            ST_TRY(set_max_lds(scatter_kernel<1024, true, false, SC_RPT_WIDE>, (int)ldsw));
            hipLaunchKernelGGL((scatter_kernel<1024, true, false, SC_RPT_WIDE>), dim3(NB), dim3(1024), ldsw, c->stream, sa);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    size_t lds = (size_t)(sa.P + 1) * (sa.gcur ? 8 : 12) + 32 * 4 + TILE * 2 + 16 + (staged ? TILE * 8 : 0);
    if (lds > 160 * 1024 && staged && !sa.gend) {          // private cursors at the widest fan-outs: no room for the staging tile,
        staged = false;                                     // the rows go out unstaged (slower, same result)
        lds -= (size_t)TILE * 8;
    }
    if (lds > 160 * 1024) return fail(PANDRS_HIP_ERR_COMPUTATION, "radix fan-out %u does not fit the scatter's LDS", sa.P);
    if (sa.gend) {
        ST_TRY(set_max_lds(scatter_kernel<THREADS, true, true>, (int)lds));
        hipLaunchKernelGGL((scatter_kernel<THREADS, true, true>), dim3(NB), dim3(THREADS), lds, c->stream, sa);
    } else if (staged) {
        ST_TRY(set_max_lds(scatter_kernel<THREADS, true>, (int)lds));
        hipLaunchKernelGGL((scatter_kernel<THREADS, true>), dim3(NB), dim3(THREADS), lds, c->stream, sa);
    } else {
        ST_TRY(set_max_lds(scatter_kernel<THREADS, false>, (int)lds));
        hipLaunchKernelGGL((scatter_kernel<THREADS, false>), dim3(NB), dim3(THREADS), lds, c->stream, sa);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t engine_workspace_bytes(int64_t n_rows, int n_cols8, int n_cols1) {
    const size_t NBmax = 1024 + 8;
    return Arena::padded(size_t(1 << 19) * 8) + 4096
         + 2 * Arena::padded((size_t(P_MAX + 1) * NBmax + 8) * 4) + Arena::padded(SCAN_SEG * 4 + 64)
         + Arena::padded(size_t(P_MAX + 1) * 32)
         + (size_t)n_cols8 * Arena::padded(size_t(n_rows) * 8) + (size_t)n_cols1 * Arena::padded(size_t(n_rows)) + (4 << 20);
}

static int32_t radix_partition_once(pandrs_hip_ctx *c, ScatterArgs &sa, PartInfo *out, int phase_hist, int phase_scan,
                                    int phase_scatter) {
    const int64_t N = sa.n_rows;
    const int SCT = c->opt.scatter_threads == 512 ? 512 : 1024;
    const bool wide = SCT == 1024 && scatter_wide_ok(c, sa, c->opt.shared_cursors != 0, c->opt.scatter_staged != 0);
    const int SC_TILE = SCT * (wide ? SC_RPT_WIDE : SC_RPT);
    const uint32_t P1 = sa.P + 1;
    int64_t n_tiles = (N + SC_TILE - 1) / SC_TILE;
    uint32_t NB = (uint32_t)std::min<int64_t>(std::max<int64_t>(n_tiles, 1), 1024);
    int64_t chunk = ((n_tiles + NB - 1) / NB) * SC_TILE;
    if (chunk == 0) chunk = SC_TILE;
    NB = (uint32_t)std::max<int64_t>((N + chunk - 1) / chunk, 1);
    NB = (NB + 7) & ~7u;                      // 8 groups of NB/8 workgroups (empty ones exit)
    size_t M = (size_t)P1 * NB;
    uint32_t *hist = c->work.take<uint32_t>(M + 8);
    uint32_t *offsets = c->work.take<uint32_t>(M + 8);
    uint32_t *seg = c->work.take<uint32_t>(scan_seg_count(M));
    uint32_t *gcur = c->work.take<uint32_t>((size_t)P1 * 8);
    if (!hist || !offsets || !seg || !gcur) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (partition)");
    {
        PhaseTimer pt(c, phase_hist);
        hipLaunchKernelGGL(histogram_kernel, dim3(NB), dim3(HI_THREADS), P1 * 4, c->stream,
                           sa.key, N, chunk, sa.P, sa.seed, hist, sa.hash_P, sa.part_shift);
        HIP_TRY(hipGetLastError());
    }
    {
        PhaseTimer pt(c, phase_scan);
        ST_TRY(exclusive_scan_u32(c, hist, M, offsets, seg));
        if (c->opt.shared_cursors)
            hipLaunchKernelGGL(init_group_cursors_kernel, dim3((P1 * 8 + 255) / 256), dim3(256), 0, c->stream,
                               offsets, NB, P1, gcur);
    }
    sa.offsets = offsets; sa.chunk = chunk;
    sa.gcur = c->opt.shared_cursors ? gcur : nullptr;
    // 8-byte columns first (the staged kernel pipelines those)
    std::stable_sort(sa.mv, sa.mv + sa.n_move, [](const MoveDesc &x, const MoveDesc &y) { return (x.kind != 0) < (y.kind != 0); });
    sa.n_move8 = 0;
    while (sa.n_move8 < sa.n_move && sa.mv[sa.n_move8].kind == 0) sa.n_move8++;
    {
        PhaseTimer pt(c, phase_scatter);
        if (SCT == 512) ST_TRY(launch_scatter<512>(c, sa, NB, c->opt.scatter_staged != 0));
        else ST_TRY(launch_scatter<1024>(c, sa, NB, c->opt.scatter_staged != 0, wide));
    }
    out->P = sa.P; out->NB = NB; out->offsets = offsets;
    return 0;
}

// The exact partition.  A fan-out of thousands leaves a scatter tile (8 K rows) as 1-row runs — 8-byte stores to 8 K places,
// 1.4 TB/s — so from P = 6144 up the rows move TWICE in long runs instead: 64-127 buckets first, then the full fan-out over the
// bucket-sorted rows, whose tiles see only ~P / 64 partitions each.  A bucket is the final partition id shifted down
// (ScatterArgs::hash_P / part_shift): a monotone coarsening, so every bucket is a contiguous range of partitions and the second
// pass is the ordinary partition of the first pass's output (key cells + a null byte per row; every moved column in its
// destination type).  62.5 M + 50 M rows of 16 bytes at P = 8192: 2.5 -> 2.1 ms; at P = 4096 and below the single pass is still
// the faster one.  Needs the temporaries in the work arena; without room (or with "two_pass" = -1) it is the single pass.
constexpr uint32_t TWO_PASS_BUCKETS = 64;
constexpr int64_t TWO_PASS_MIN_P = 6144;          // measured crossover (16-byte rows): 17.5 / 20.5 / 26.6 ns per K rows in one pass at P = 2.5 K / 4 K / 8 K, 21-25 in two
size_t two_pass_workspace_bytes(int64_t n_rows, int n_cols8, int n_cols1) {
    const size_t NBmax = 1024 + 8;
    return (size_t)(1 + n_cols8) * Arena::padded((size_t)(n_rows + 1) * 8) + (size_t)(1 + n_cols1) * Arena::padded((size_t)n_rows + 16) +
           2 * Arena::padded(((size_t)(256 + 1) * NBmax + 8) * 4) + Arena::padded(SCAN_SEG * 4 + 64) +
           Arena::padded((size_t)(256 + 1) * 32) + 65536;
}
int32_t radix_partition(pandrs_hip_ctx *c, ScatterArgs &sa, PartInfo *out, int phase_hist, int phase_scan,
                        int phase_scatter) {
    const int64_t N = sa.n_rows;
    const bool key_nulls = sa.key.null_bits || sa.key.null_bytes;
    // first-pass buckets: the final partition ids shifted down until <= 64 .. 127 of them are left (any P; a monotone coarsening)
    uint32_t shift = 0;
    while (((sa.P - 1) >> shift) + 1 > 2 * TWO_PASS_BUCKETS - 1) shift++;
    const uint32_t B1 = ((sa.P - 1) >> shift) + 1;
    bool two = sa.allow_two_pass && c->opt.two_pass >= 0 && sa.P >= (c->opt.two_pass_min_p > 0 ? c->opt.two_pass_min_p : TWO_PASS_MIN_P) && shift > 0 && !sa.hash_P && N >= (int64_t(1) << (c->opt.two_pass_min_p > 0 ? 16 : 22)) &&      // (an explicit threshold, tests: small inputs too)
              
               sa.n_move + (key_nulls ? 1 : 0) <= MAX_MOVE;
    if (!two) return radix_partition_once(c, sa, out, phase_hist, phase_scan, phase_scatter);
    auto elem_of = [](int kind) -> size_t { return kind == 0 || kind == 4 || kind == 5 ? 8 : (kind == 3 || kind == 7 ? 4 : 1); };
    size_t need = Arena::padded((size_t)(N + 1) * 8) + (key_nulls ? Arena::padded((size_t)N + 16) : 0);
    for (int i = 0; i < sa.n_move; i++) need += Arena::padded((size_t)(N + 1) * elem_of(sa.mv[i].kind));
    const size_t NBmax = 1024 + 8;
    const size_t sets = 2 * Arena::padded(((size_t)(B1 + 1) * NBmax + 8) * 4) + 2 * Arena::padded(((size_t)(sa.P + 1) * NBmax + 8) * 4) +
                        2 * Arena::padded(SCAN_SEG * 4 + 64) + 2 * Arena::padded((size_t)(sa.P + 1) * 32) + 65536;
    if (c->work.cap - c->work.off < need + sets)          // (a caller that budgeted less than it promised: the single pass, not a failure)
        return radix_partition_once(c, sa, out, phase_hist, phase_scan, phase_scatter);
    ScatterArgs s1 = sa, s2 = sa;
    s1.P = B1; s1.hash_P = sa.P; s1.part_shift = shift;
    uint64_t *tk = c->work.take<uint64_t>((size_t)N + 1);
    s1.pkeys = tk;
    for (int i = 0; i < sa.n_move; i++) {
        const int kind = sa.mv[i].kind;
        void *t = c->work.take<uint8_t>((size_t)(N + 1) * elem_of(kind));
        if (!t) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (two-pass partition)");
        s1.mv[i].dst = t;
        s2.mv[i].src = t;
        s2.mv[i].kind = elem_of(kind) == 8 ? 0 : (elem_of(kind) == 4 ? 7 : 2);       // the second pass copies what the first one produced
    }
    uint8_t *tnull = nullptr;
    if (key_nulls) {
        tnull = c->work.take<uint8_t>((size_t)N + 16);
        if (!tnull) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (two-pass partition)");
        s1.mv[s1.n_move++] = sa.key.null_bits ? MoveDesc{sa.key.null_bits, tnull, 6, 0} : MoveDesc{sa.key.null_bytes, tnull, 2, 0};
    }
    if (!tk) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (two-pass partition)");
    PartInfo p1{};
    ST_TRY(radix_partition_once(c, s1, &p1, PANDRS_HIP_PHASE_PREPARTITION, PANDRS_HIP_PHASE_PREPARTITION, PANDRS_HIP_PHASE_PREPARTITION));
    s2.key = KeyDesc{tk, nullptr, tnull, DT_CELL};
    ST_TRY(radix_partition_once(c, s2, out, phase_hist, phase_scan, phase_scatter));
    sa.offsets = s2.offsets; sa.chunk = s2.chunk; sa.gcur = s2.gcur;
    return 0;
}

void plan_sampled_regions(pandrs_hip_ctx *c, const uint32_t *hist, int64_t n_rows, uint32_t P1, uint32_t total_cap, uint32_t *gbeg,
                          uint32_t *gcur, uint32_t *gend, uint32_t *flags, double clump) {
    hipLaunchKernelGGL(plan_regions_kernel, dim3(1), dim3(1024), 0, c->stream, hist, n_rows, P1, total_cap, gbeg, gcur, gend, flags, clump);
}

uint32_t sampled_partition_rows(int64_t n_rows, int64_t P) {
    // budget: the sampled shares + 6 sigma twice (<= ~19 % at the sizes sampled_partition_ok admits) + per-region constants
    const double rows = (double)n_rows * 1.22 + (double)(P + 1) * 8.0 * 96.0 + 65536.0 + (double)SC_TILE_MAX;
    return rows >= 4.2e9 ? 0u : (uint32_t)rows;
}
bool sampled_partition_ok(int64_t n_rows, int64_t P) {
    // enough rows per (partition, group) region and enough samples per partition for tight 6-sigma margins
    return P >= 1 && P <= P_MAX && n_rows / (P * 8) >= 4096 && n_rows / ((int64_t)SAMPLE_PERIOD * P) >= 4096 &&
           sampled_partition_rows(n_rows, P) != 0;
}

int32_t radix_partition_sampled(pandrs_hip_ctx *c, ScatterArgs &sa, PartInfo *out, int phase_hist, int phase_scatter) {
    const int64_t N = sa.n_rows;
    const uint32_t P1 = sa.P + 1;
    const bool wide = scatter_wide_ok(c, sa, true, true);
    const int SC_TILE = 1024 * (wide ? SC_RPT_WIDE : SC_RPT);
    const int64_t n_tiles = (N + SC_TILE - 1) / SC_TILE;
    uint32_t NB = (uint32_t)std::min<int64_t>(std::max<int64_t>(n_tiles, 8), 1024);
    NB = (NB + 7) & ~7u;
    uint32_t *hist = c->work.take<uint32_t>((size_t)SAMPLE_REPL * (P1 + 1) + 64);      // + the flags word block
    uint32_t *gbeg = c->work.take<uint32_t>((size_t)P1 * 8);
    uint32_t *gcur = c->work.take<uint32_t>((size_t)P1 * 8);
    uint32_t *gend = c->work.take<uint32_t>((size_t)P1 * 8);
    uint32_t *flags = hist ? hist + (size_t)SAMPLE_REPL * (P1 + 1) : nullptr;
    if (!hist || !gbeg || !gcur || !gend || !flags) return fail(PANDRS_HIP_ERR_OUT_OF_MEMORY, "workspace too small (partition)");
    const uint32_t total_cap = sampled_partition_rows(N, sa.P) - SC_TILE_MAX;      // the last tile of every column: trash
    {
        PhaseTimer pt(c, phase_hist);
        HIP_TRY(hipMemsetAsync(hist, 0, ((size_t)SAMPLE_REPL * (P1 + 1) + 64) * 4, c->stream));
        const int64_t n_sb = (N + (int64_t)SAMPLE_BLOCK * SAMPLE_PERIOD - 1) / ((int64_t)SAMPLE_BLOCK * SAMPLE_PERIOD);
        hipLaunchKernelGGL(sample_histogram_kernel, dim3((unsigned)std::min<int64_t>((n_sb + 3) / 4, 256)), dim3(1024), P1 * 4, c->stream,
                           sa.key, N, sa.P, sa.seed, hist);
        hipLaunchKernelGGL(plan_regions_kernel, dim3(1), dim3(1024), 0, c->stream, hist, N, P1, total_cap,
                           gbeg, gcur, gend, flags, 1.0);
        HIP_TRY(hipGetLastError());
    }
    sa.offsets = nullptr; sa.chunk = 0; sa.gcur = gcur; sa.gend = gend; sa.flags = flags; sa.total_cap = total_cap;
    std::stable_sort(sa.mv, sa.mv + sa.n_move, [](const MoveDesc &x, const MoveDesc &y) { return (x.kind != 0) < (y.kind != 0); });
    sa.n_move8 = 0;
    while (sa.n_move8 < sa.n_move && sa.mv[sa.n_move8].kind == 0) sa.n_move8++;
    {
        PhaseTimer pt(c, phase_scatter);
        ST_TRY(launch_scatter<1024>(c, sa, NB, true, wide));
    }
    out->P = sa.P; out->NB = NB; out->offsets = nullptr;
    out->gbeg = gbeg; out->gcur = gcur; out->gend = gend; out->flags = flags; out->total_cap = total_cap;
    return 0;
}

__global__ void gather_part_offsets_kernel(const uint32_t *offsets, uint32_t NB, uint32_t n, uint32_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = offsets[(size_t)i * NB];
}


// partition starts (offsets[p * NB]) of the first n partitions, gathered into a dense array
void gather_part_offsets(pandrs_hip_ctx *c, const uint32_t *offsets, uint32_t NB, uint32_t n, uint32_t *out) {
    hipLaunchKernelGGL(gather_part_offsets_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, offsets, NB, n, out);
}


}  // namespace pandrs
