// swiss.hpp — the LDS Swiss table shared by the lean aggregate (aggregate2.hip) and the clustered-rows pass (clustered.hip)
#pragma once
#include "aggregate.hpp"

namespace pandrs {

// Key -> slot in the partition's LDS table, insert on first sight.  Swiss-table layout: groups of 16
// slots; a group's 16 one-byte tags (0 = empty, else 1 + hash byte) are ONE ds_read_b128, matched with
// SWAR byte compares; the candidate's 8-byte key is then verified with one ds_read_b64.  16-slot groups
// keep a wave's longest probe chain at ~1.4 groups at 65 % load (4-key buckets: 4.1 bucket reads per
// wave and lookup — the dominant cost of the round-1 kernel, instructions and LDS bytes alike).
// keys[] is the truth (claimed with ds_cmpst_b64); a tag is written after its key and may lag: a
// lagging tag only sends a lane to the CAS, which answers "already yours" or "taken".
// max_groups: SMALL gives up after a few groups (a chain that long means the table is nearly full; the caller falls back)
__device__ __forceinline__ uint32_t swiss_find(uint64_t k, uint64_t *keys, uint8_t *ctrl, uint32_t T, uint32_t seed, uint32_t max_groups = 0xFFFFFFFFu) {
    const uint32_t NG = T >> 4;
    const uint32_t h = hash32(k, seed);
    uint32_t g = slot_of(h, NG);
    const uint32_t tag = max(h & 0xFFu, 1u);          // never 0 (= empty); 255 values: half the false candidates of `| 1`
    const uint32_t tag4 = tag * 0x01010101u;
    const uint32_t n_probe = min(NG, max_groups);
    for (uint32_t probe = 0; probe < n_probe; probe++) {
        const uint4 cw = *reinterpret_cast<const uint4 *>(ctrl + 16 * g);
        const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
        uint32_t cand[4], emp[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t x = w[q] ^ tag4;
            cand[q] = (x - 0x01010101u) & ~x & 0x80808080u;          // bytes equal to the tag (a 0x01 byte above a match may be flagged too: verified below)
            emp[q] = (w[q] - 0x01010101u) & ~w[q] & 0x80808080u;     // empty bytes (same caveat; the CAS decides)
        }
        // candidates in slot order: verify the full key
        for (;;) {
            int q = cand[0] ? 0 : cand[1] ? 1 : cand[2] ? 2 : cand[3] ? 3 : -1;
            if (q < 0) break;
            const uint32_t z = q == 0 ? cand[0] : q == 1 ? cand[1] : q == 2 ? cand[2] : cand[3];
            const uint32_t bit = (uint32_t)__ffs((int)z) - 1u;
            const uint32_t idx = 16 * g + 4 * q + (bit >> 3);
            if (keys[idx] == k) return idx;
            const uint32_t clr = ~(1u << bit);
            if (q == 0) cand[0] &= clr; else if (q == 1) cand[1] &= clr; else if (q == 2) cand[2] &= clr; else cand[3] &= clr;
        }
        // not in this group: claim a free slot, or move on when the group has none.  The search starts at a
        // key-dependent position: first-free-from-0 would put the few keys of a lightly loaded table (low
        // cardinality per partition) all at positions 0-2 of their groups, i.e. on 6 of the 32 LDS bank pairs
        // (measured: C3's aggregate 0.9 -> 1.4 ms).
        uint32_t free16 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            free16 |= ((((emp[q] >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * q);          // byte flags -> one bit per slot
        const uint32_t r0 = (h >> 8) & 15u;
        while (free16) {
            const uint32_t rot = ((free16 >> r0) | (free16 << (16 - r0))) & 0xFFFFu;
            const uint32_t pos = ((uint32_t)__ffs((int)rot) - 1u + r0) & 15u;
            const uint32_t idx = 16 * g + pos;
            const uint64_t old = atomicCAS((unsigned long long *)&keys[idx], EMPTY_KEY, k);
            if (old == EMPTY_KEY) { ctrl[idx] = (uint8_t)tag; return idx; }
            if (old == k) return idx;
            free16 &= ~(1u << pos);
        }
        g = g + 1 == NG ? 0 : g + 1;
    }
    return T + 2;
}

// enc of a value for the min-type states: order-preserving u64 (f64: sign-magnitude flip, 4 VALU)
template <int KIND>
__device__ __forceinline__ uint64_t enc_val(uint64_t bits) {
    if (KIND == 1) return bits ^ 0x8000000000000000ull;
    const uint32_t hi = (uint32_t)(bits >> 32), lo = (uint32_t)bits;
    const uint32_t sm = (uint32_t)((int32_t)hi >> 31);
    return ((uint64_t)(hi ^ (sm | 0x80000000u)) << 32) | (uint32_t)(lo ^ sm);
}

}  // namespace pandrs
