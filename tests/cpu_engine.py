"""CPU stand-in for pandrs_amd.Context used ONLY by the gloo tests of the multi-GPU exchange logic
(tests/test_dist_gloo.py).  It implements the calls DistributedGroupBy / DistributedJoinGroupBy need with numpy so
that the routing / count-exchange / all-to-all / merge plumbing can run with world_size 2 on a
box without GPUs.  It is test infrastructure (like oracle/): never imported by pandrs_amd/."""
import numpy as np

I64, F64 = 0, 1
SUM, MEAN, MIN, MAX, COUNT = 0, 1, 2, 3, 4


class NumpyEngine:
    """State layout (its own, self-consistent): [group size] + per value column [sum, nn, min, max]."""

    def _reduce(self, keys, knull, states, ncols):
        # group rows by (null, key); add sizes/sums/nn, min/max the rest
        comp = np.stack([knull.astype(np.uint64), keys.astype(np.uint64)], axis=1)
        uniq, inv = np.unique(comp, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        g = len(uniq)
        out = np.zeros((1 + 4 * ncols, g), np.float64)
        np.add.at(out[0], inv, states[0])
        for c in range(ncols):
            b = 1 + 4 * c
            np.add.at(out[b], inv, states[b])
            np.add.at(out[b + 1], inv, states[b + 1])
            out[b + 2] = np.inf
            out[b + 3] = -np.inf
            np.minimum.at(out[b + 2], inv, states[b + 2])
            np.maximum.at(out[b + 3], inv, states[b + 3])
        return uniq[:, 1].copy(), uniq[:, 0].astype(np.uint8), out

    def groupby_partials(self, keys, n_rows, vals, aggs):
        (kdata, kmask, _), = keys
        knull = np.zeros(n_rows, np.uint8) if kmask is None else np.unpackbits(kmask, bitorder="little")[:n_rows]
        ncols = len(vals)
        st = np.zeros((1 + 4 * ncols, n_rows), np.float64)
        st[0] = 1
        for c, (v, m, _) in enumerate(vals):
            valid = np.ones(n_rows, bool) if m is None else ~np.unpackbits(m, bitorder="little")[:n_rows].astype(bool)
            v = np.asarray(v, np.float64)
            st[1 + 4 * c] = np.where(valid, v, 0.0)
            st[2 + 4 * c] = valid
            st[3 + 4 * c] = np.where(valid, v, np.inf)
            st[4 + 4 * c] = np.where(valid, v, -np.inf)
        k = np.where(knull.astype(bool), 0, np.asarray(kdata).view(np.uint64))
        self._p = self._reduce(k, knull, st, ncols)
        self._ncols = ncols
        return len(self._p[0]), 1 + 4 * ncols

    def partials_split(self, n_ranks):
        k, kn, st = self._p
        owner = np.where(kn.astype(bool), 0, (k * np.uint64(0x9E3779B97F4A7C15) >> np.uint64(40)) % np.uint64(n_ranks)).astype(np.int64)
        order = np.argsort(owner, kind="stable")
        rec = np.empty((len(k), 2 + st.shape[0]), np.uint64)
        rec[:, 0] = k[order]
        rec[:, 1] = kn[order]
        rec[:, 2:] = st[:, order].T.copy().view(np.uint64)
        counts = np.bincount(owner, minlength=n_ranks).tolist()
        return rec, counts

    def groupby_merge(self, key_dtype, records, val_dtypes, val_has_nulls, aggs):
        rec = np.ascontiguousarray(records, np.uint64)
        st = rec[:, 2:].copy().view(np.float64).T
        k, kn, out = self._reduce(rec[:, 0], rec[:, 1].astype(np.uint8), st, len(val_dtypes))
        res = np.zeros((len(aggs), len(k)))
        for a, (c, op) in enumerate(aggs):
            b = 1 + 4 * c
            if op == COUNT:
                res[a] = out[0]
            elif op == SUM:
                res[a] = out[b]
            elif op == MEAN:
                res[a] = np.where(out[b + 1] > 0, out[b] / np.maximum(out[b + 1], 1), 0.0)
            elif op == MIN:
                res[a] = np.where(out[b + 2] == np.inf, 0.0, out[b + 2])
            elif op == MAX:
                res[a] = np.where(out[b + 3] == -np.inf, 0.0, out[b + 3])
        self._r = (k[None, :].copy(), kn[None, :].copy(), res)
        return len(k)

    def groupby_fetch(self):
        return self._r

    # ---- the shuffle path: rows to the owner of their key, then ordinary calls on the received rows
    def shuffle_split(self, key, payload, n_rows, n_ranks, drop_null_keys=False):
        from oracle import oracle_np as ONP
        if key[2] == 4:
            data, mask, _ = key
            cell = np.asarray(data)[:n_rows].view(np.uint64).copy()
            nul = np.zeros(n_rows, np.uint8) if mask is None else np.unpackbits(np.asarray(mask, np.uint8), bitorder="little")[:n_rows]
        else:
            nul, cell = ONP.key_cells(key, n_rows)
        owner = ((cell * np.uint64(0xD6E8FEB86659FD93)) >> np.uint64(33)) % np.uint64(n_ranks)
        owner = np.where(nul.astype(bool), n_ranks - 1, owner).astype(np.int64)
        keep = np.ones(n_rows, bool) if not drop_null_keys else ~nul.astype(bool)
        order = np.argsort(owner, kind="stable")
        order = order[keep[order]]
        pays, pnull = [], []
        for data, mask, dt in payload:
            pays.append(np.asarray(data)[:n_rows][order].astype(np.int64 if dt != 1 else np.float64).view(np.uint64)
                        if dt != 2 else np.asarray(data, np.uint32)[:n_rows][order].astype(np.uint64))
            pnull.append(None if mask is None else np.unpackbits(np.asarray(mask, np.uint8), bitorder="little")[:n_rows][order])
        counts = np.bincount(owner[order], minlength=n_ranks).tolist()
        return cell[order], nul[order], pays, pnull, counts

    def key_hash_cells(self, keys, n_rows):
        from oracle import oracle_np as ONP
        h = np.full(n_rows, 0x9E3779B97F4A7C15, np.uint64)
        for col in keys:
            nul, cell = ONP.key_cells(col, n_rows)
            x = (cell ^ h) + nul.astype(np.uint64) * np.uint64(0x1234567)
            x ^= x >> np.uint64(33); x *= np.uint64(0xFF51AFD7ED558CCD); x ^= x >> np.uint64(29)
            h = x
        return h

    def bytes_to_bitmap(self, flags):
        return np.packbits(np.asarray(flags, np.uint8) != 0, bitorder="little")

    def groupby_agg(self, keys, n_rows, vals, aggs):
        from oracle import oracle as O
        return O.groupby_agg(keys, n_rows, vals, aggs)

    def join_groupby_sum(self, lkey, lval, n_left, rkey, rgroup, n_right):
        """The fused C5 call, answered by the oracle (test infrastructure on both sides)."""
        from oracle import oracle as O
        return O.join_groupby_sum(lkey, lval, n_left, rkey, rgroup, n_right)
