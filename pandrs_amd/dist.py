"""Row-range-sharded groupby across the GPUs of one node: one process per GPU, ONE exchange step.

Plan (the reference has no distributed path for this, src/distributed is an
in-process DataFusion wrapper):

  1. every rank pre-aggregates its own row range into mergeable partial rows
     (pandrs_hip_groupby_partials) — with G << N/P this shrinks the exchange from raw rows to
     at most G records per rank;
  2. the partial rows are bucketed by owner = hash(key) mod world and written rank-contiguous as
     packed records (pandrs_hip_partials_split);
  3. ONE all-to-all moves them (RCCL over xGMI via torch.distributed backend "nccl": grouped
     send/recv to every peer, all 7 links busy at once — not a ring), preceded by the world x
     world count exchange;
  4. every rank merges the records it received — the key sets of different ranks are disjoint —
     and finalises (pandrs_hip_groupby_merge).  The result stays sharded by key owner.

torch.distributed is transport only.  `engine` is the object that does the local work; in
production it is a pandrs_amd.Context (HIP).  Tests inject a CPU stand-in so that the exchange
logic runs under gloo without a GPU.
"""
import time


def np_contig(a):
    import numpy as np
    return np.ascontiguousarray(a)


class DistributedGroupBy:
    def __init__(self, engine, dist, device):
        self.engine = engine
        self.dist = dist
        self.device = device
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.last_timings = None

    def _torch(self):
        import torch
        return torch

    def exchange(self, records, counts):
        """records: [G, W] int64 tensor, rank-contiguous by owner; counts: list[world].
        -> [G_recv, W] tensor of the records this rank owns."""
        torch = self._torch()
        dist = self.dist
        send_counts = torch.tensor(counts, dtype=torch.int64, device=records.device)
        recv_counts = torch.empty(self.world, dtype=torch.int64, device=records.device)
        dist.all_to_all_single(recv_counts, send_counts)
        recv_list = [int(x) for x in recv_counts.tolist()]
        out = torch.empty((sum(recv_list), records.shape[1]), dtype=records.dtype, device=records.device)
        dist.all_to_all_single(out, records, output_split_sizes=recv_list, input_split_sizes=list(counts))
        return out

    MERGEABLE_OPS = (0, 1, 2, 3, 4)      # Sum, Mean, Min, Max, Count: pre-aggregate, exchange partial states

    def _agree_on_null_masks(self, cols, n_rows):
        """All-reduce (MAX) one has-null-mask bit per column; where another rank has a mask and this one does not, an
        all-valid bitmap is materialised so that every rank plans the same states and issues the same collectives."""
        if not cols:
            return cols
        torch = self._torch()
        dev = self.device if str(self.device) != "cpu" else "cpu"
        flags = torch.tensor([1 if c[1] is not None else 0 for c in cols], dtype=torch.int64, device=dev)
        self.dist.all_reduce(flags, op=self.dist.ReduceOp.MAX)
        out = []
        for (data, mask, dt), f in zip(cols, flags.tolist()):
            if f and mask is None:
                nb = (int(n_rows) + 7) // 8
                if hasattr(data, "is_cuda"):
                    mask = torch.zeros(nb, dtype=torch.uint8, device=data.device)
                else:
                    import numpy as np
                    mask = np.zeros(nb, np.uint8)
            out.append((data, mask, dt))
        return out

    def exchange_columns(self, columns, counts):
        """All-to-all of several row-aligned columns that share the same rank-contiguous split."""
        torch = self._torch()
        dist = self.dist
        first = next(c for c in columns if c is not None)
        was_numpy = not torch.is_tensor(first)
        def as_t(a):        # 64-bit unsigned words travel as int64 (gloo / RCCL have no uint64)
            if torch.is_tensor(a):
                return a
            a = np_contig(a)
            return torch.from_numpy(a.view("int64") if a.dtype.name == "uint64" else a)
        first = as_t(first)
        send_counts = torch.tensor(counts, dtype=torch.int64, device=first.device)
        recv_counts = torch.empty(self.world, dtype=torch.int64, device=first.device)
        dist.all_to_all_single(recv_counts, send_counts)
        recv_list = [int(x) for x in recv_counts.tolist()]
        out = []
        for c in columns:
            if c is None:
                out.append(None)
                continue
            t = as_t(c)
            r = torch.empty(sum(recv_list), dtype=t.dtype, device=t.device)
            dist.all_to_all_single(r, t, output_split_sizes=recv_list, input_split_sizes=list(counts))
            out.append(r.numpy().view(c.dtype) if was_numpy else r)
        return out

    def _received_column(self, data, null_bytes, dtype):
        """A payload column after the exchange -> (data, bitmap, dtype) for the ordinary entry points:
        f64 / i64 travel as raw 8-byte words, u32 codes zero-extended (key dtype CELL64 = 4)."""
        torch_like = hasattr(data, "is_cuda")
        if dtype == 1:
            data = data.view(self._torch().float64) if torch_like else data.view("float64")
        elif dtype == 0 and not torch_like:
            data = data.view("int64")
        mask = None if null_bytes is None else self.engine.bytes_to_bitmap(null_bytes)
        return (data, mask, 4 if dtype == 2 else dtype)

    def groupby_by_shuffle(self, keys, n_rows, vals, aggs):
        """The general path (any aggregate except First/Last, which need the global row order; any number
        of key columns): every row goes to the owner of its key — pandrs_hip_shuffle_split, one
        all-to-all per column — and the owner runs the ordinary groupby on the rows it received.
        A composite key is shuffled on a hash cell of the whole tuple (pandrs_hip_key_hash_cells) with
        the key columns travelling as payload, so each rank packs complete tuples only."""
        eng = self.engine
        vals = self._agree_on_null_masks(vals, n_rows)
        keys = self._agree_on_null_masks(keys, n_rows)
        if any(op in (8, 9) for _, op in aggs):
            raise NotImplementedError("First/Last need the global row order and are not sharded")
        if any(k[2] == 3 for k in keys) and len(keys) > 1:
            raise NotImplementedError("bit-packed key columns cannot travel as shuffle payload")
        if len(keys) == 1:
            cells, knull, pays, pnull, counts = eng.shuffle_split(keys[0], vals, n_rows, self.world, drop_null_keys=False)
            got = self.exchange_columns([cells, knull] + pays + pnull, counts)
            rc, rn = got[0], got[1]
            rp, rpn = got[2:2 + len(vals)], got[2 + len(vals):]
            kmask = eng.bytes_to_bitmap(rn) if keys[0][1] is not None else None
            keys2 = [(rc, kmask, 4)]
        else:
            hcell = eng.key_hash_cells(keys, n_rows)
            payload = list(keys) + list(vals)
            cells, _, pays, pnull, counts = eng.shuffle_split((hcell, None, 4), payload, n_rows, self.world, drop_null_keys=False)
            got = self.exchange_columns(pays + pnull, counts)
            rp_all, rpn_all = got[:len(payload)], got[len(payload):]
            keys2 = [self._received_column(rp_all[i], rpn_all[i], k[2]) for i, k in enumerate(keys)]
            rp, rpn = rp_all[len(keys):], rpn_all[len(keys):]
        n_recv = int(rp[0].shape[0]) if vals else int(keys2[0][0].shape[0])
        vals2 = [self._received_column(rp[i], rpn[i], v[2]) for i, v in enumerate(vals)]
        return eng.groupby_agg(keys2, n_recv, vals2, aggs)

    def groupby_agg(self, keys, n_rows, vals, aggs, fetch=True):
        """Same arguments as Context.groupby_agg; every rank passes its own row range.
        Returns this rank's share of the groups (keys owned by this rank)."""
        eng = self.engine
        general = len(keys) > 1 or any(op not in self.MERGEABLE_OPS for _, op in aggs)
        if general and not getattr(eng, "comm", None):
            return self.groupby_by_shuffle(keys, n_rows, vals, aggs)
        torch = self._torch()
        if getattr(eng, "comm", None):        # (also the general case: pandrs_hip_dist_groupby_agg shuffles the rows itself)
            # the exchange lives in the library (pandrs_hip_dist_groupby_agg: count all-gather + ONE grouped
            # ncclSend / ncclRecv all-to-all on the context's stream); this class is only its caller
            t0 = time.perf_counter()
            eng.dist_groupby_compute(keys, n_rows, vals, aggs)
            t = eng.timings()
            self.last_timings = {"total_ms": (time.perf_counter() - t0) * 1e3, "phase_ms": {"merge_" + k: v for k, v in t["phase_ms"].items()},
                                 "algorithmic_bytes": n_rows * (8 + 8 * len(vals)), "wall_ms": {"in_library": (time.perf_counter() - t0) * 1e3}}
            return eng.groupby_fetch() if fetch else None
        # torch.distributed transport (gloo rehearsals, engines without a communicator): agree on the partial-record
        # layout first — a null mask present on some ranks only must not change the record width on those ranks
        vals = self._agree_on_null_masks(vals, n_rows)
        t0 = time.perf_counter()
        ng_local, n_state = eng.groupby_partials(keys, n_rows, vals, aggs)
        t_local = eng.timings() if hasattr(eng, "timings") else None
        records, counts = eng.partials_split(self.world)
        if not torch.is_tensor(records):
            records = torch.from_numpy(records.view("int64"))
        t1 = time.perf_counter()
        recv = self.exchange(records, counts)
        if recv.is_cuda:
            torch.cuda.synchronize(recv.device)
        t2 = time.perf_counter()
        val_dtypes = [v[2] for v in vals]
        val_has_nulls = [v[1] is not None for v in vals]
        # the merge's cardinality is bounded by the records received: no sampling pass needed
        if hasattr(eng, "set_option"):
            eng.set_option("groups_hint", max(int(recv.shape[0]), 1))
        try:
            eng.groupby_merge(keys[0][2], recv if recv.is_cuda else recv.numpy().view("uint64"),
                              val_dtypes, val_has_nulls, aggs)
        finally:
            if hasattr(eng, "set_option"):
                eng.set_option("groups_hint", 0)
        t3 = time.perf_counter()
        t_merge = eng.timings() if hasattr(eng, "timings") else None
        if t_local is not None:
            phases = dict(t_local["phase_ms"])
            for k, v in t_merge["phase_ms"].items():
                phases["merge_" + k] = v
            phases["exchange_wall"] = (t2 - t1) * 1e3
            self.last_timings = {
                "total_ms": t_local["total_ms"] + t_merge["total_ms"] + (t2 - t1) * 1e3,
                "phase_ms": phases,
                "algorithmic_bytes": t_local["algorithmic_bytes"],
                "local_groups": ng_local, "records_sent": int(sum(counts)), "records_received": int(recv.shape[0]),
                "wall_ms": {"local+split": (t1 - t0) * 1e3, "exchange": (t2 - t1) * 1e3, "merge": (t3 - t2) * 1e3},
            }
        if fetch:
            return eng.groupby_fetch()
        return None


def _pack_bits(flags):
    """byte-per-row flags (numpy or torch, 0/1) -> LSB-first bitmap of the same kind
    (reference mask layout: src/core/column.rs:163-177)."""
    n = int(flags.shape[0])
    pad = (-n) % 8
    if hasattr(flags, "numpy") or hasattr(flags, "is_cuda"):
        import torch
        f = flags.to(torch.int32)
        if pad:
            f = torch.cat([f, torch.zeros(pad, dtype=torch.int32, device=f.device)])
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=f.device)
        return (f.view(-1, 8) * w).sum(dim=1).to(torch.uint8)
    import numpy as np
    return np.packbits(np.asarray(flags, dtype=np.uint8), bitorder="little")


class DistributedJoinGroupBy:
    """BASELINE config 5 across the GPUs of one node: inner join of a row-range-sharded probe side
    with a row-range-sharded build side, then groupby(g).sum(v) (reference path
    src/optimized/split_dataframe/join.rs:76-224 followed by group/aggregation.rs:763).

    Plan — the probe side (10 x the build side in C5) never leaves its GPU:
      1. all-gather the build side (key, g): 0.8 GB in total for C5, one RCCL all-gather per
         column with every xGMI link busy; shards are padded to a common length with NULL-key
         rows, which the join skips by its own semantics (join.rs:107-142), so no trimming pass;
      2. local fused join -> groupby-sum (pandrs_hip_join_groupby_sum): <= G partial sums per rank;
      3. the partial sums go through DistributedGroupBy (owner split -> ONE all-to-all -> merge).
    The result stays sharded by owner of g.  Radix-shuffling both sides by key instead would move
    the 8 GB probe side over xGMI; it only pays when the build side does not fit one GPU."""

    def __init__(self, engine, dist, device):
        self.engine = engine
        self.dist = dist
        self.device = device
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.groupby = DistributedGroupBy(engine, dist, device)
        self.last_wall_ms = None

    def _gather_build(self, data, mask, n, pad_null):
        """-> (data[world * n_pad], bitmap[world * n_pad / 8] or None) of every rank's shard.
        `pad_null` marks the padding rows null (join key) or leaves them valid (payload)."""
        import numpy as np
        import torch
        dist = self.dist
        is_t = torch.is_tensor(data)
        t = data if is_t else torch.from_numpy(np.ascontiguousarray(data))
        sizes = torch.tensor([int(n)], dtype=torch.int64, device=t.device)
        all_sizes = torch.empty(self.world, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(all_sizes, sizes)
        n_pad = (int(all_sizes.max().item()) + 7) // 8 * 8
        send = torch.zeros(n_pad, dtype=t.dtype, device=t.device)
        send[:n] = t[:n]
        out = torch.empty(self.world * n_pad, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, send)
        need_mask = mask is not None or (pad_null and any(int(s) != n_pad for s in all_sizes.tolist()))
        flags = torch.tensor([1 if need_mask else 0], dtype=torch.int64, device=t.device)
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        bits = None
        if int(flags.item()):
            mb = torch.zeros(n_pad // 8, dtype=torch.uint8, device=t.device)
            if mask is not None:
                m = mask if torch.is_tensor(mask) else torch.from_numpy(np.ascontiguousarray(mask))
                nb = (n + 7) // 8
                mb[:nb] = m[:nb].to(t.device)
                if n % 8:   # bits past n inside the last byte are undefined in the caller's bitmap
                    mb[nb - 1] &= (1 << (n % 8)) - 1
            if pad_null and n < n_pad:
                first = n // 8
                if n % 8:
                    mb[first] |= (0xFF << (n % 8)) & 0xFF
                    first += 1
                mb[first:] = 0xFF
            bits = torch.empty(self.world * (n_pad // 8), dtype=torch.uint8, device=t.device)
            dist.all_gather_into_tensor(bits, mb)
        if not is_t:
            out = out.numpy()
            bits = None if bits is None else bits.numpy()
        return out, bits, self.world * n_pad

    def _shuffled_side(self, key, payload, n_rows):
        """One side of the join after the row shuffle by key owner: (key column, payload column, rows)."""
        eng = self.engine
        cells, _, (pay,), (pnull,), counts = eng.shuffle_split(key, [payload], n_rows, self.world, drop_null_keys=True)
        rc, rp, rpn = self.groupby.exchange_columns([cells, pay, pnull], counts)
        if payload[2] == 1:
            rp = rp.view(self.groupby._torch().float64) if hasattr(rp, "is_cuda") else rp.view("float64")
        elif not hasattr(rp, "is_cuda"):
            rp = rp.view("int64") if payload[2] == 0 else rp
        pdt = 4 if payload[2] == 2 else payload[2]      # u32 codes travel zero-extended: 8-byte cells
        return (rc, None, 4), (rp, None if rpn is None else eng.bytes_to_bitmap(rpn), pdt), int(rc.shape[0])

    def join_groupby_sum(self, lkey, lval, n_left, rkey, rgroup, n_right, strategy="allgather"):
        """Columns are (data, mask, dtype) like Context.join_groupby_sum; every rank passes its own
        row ranges of both sides.  -> this rank's share of (g cells, g null flags, sums).
        strategy "auto" picks between the two; "allgather": the build side is replicated, the probe side stays put (small builds);
        "shuffle": both sides go to the owner of their join key (SURVEY.md 8e's radix all-to-all) —
        every GPU then builds only 1/world of the build side, at the price of moving the probe rows."""
        t0 = time.perf_counter()
        if strategy == "auto" and getattr(self.engine, "comm", None) and not hasattr(self.dist, "all_reduce"):
            strategy = "allgather"          # the library owns the fabric: no host collective to take the size vote with
        if strategy == "auto":
            # replicating the build side makes EVERY rank build all of it; shuffling moves the probe rows once.
            # By the single-GPU numbers the shuffle wins from 4 ranks up once the build side is large.
            torch = self.groupby._torch()
            tot = torch.tensor([int(n_right)], dtype=torch.int64, device=self.device if str(self.device) != "cpu" else "cpu")
            self.dist.all_reduce(tot)
            strategy = "shuffle" if self.world >= 4 and int(tot.item()) >= 8_000_000 else "allgather"
        # the in-library exchange (pandrs_hip_dist_join_groupby_sum) replicates the build side and takes 8-byte build-side
        # columns on the device: string-code keys / groups and host shards keep the path below
        if strategy == "allgather" and getattr(self.engine, "comm", None) and rkey[2] in (0, 1, 4) and rgroup[2] in (0, 1, 4) and \
                all(hasattr(c[0], "is_cuda") or type(c[0]).__name__ == "ResidentColumn" for c in (lkey, lval, rkey, rgroup)):
            out = self.engine.dist_join_groupby_sum(lkey, lval, n_left, rkey, rgroup, n_right)
            self.last_wall_ms = {"in_library": (time.perf_counter() - t0) * 1e3}
            return out
        if strategy == "shuffle":
            lk2, lv2, nl2 = self._shuffled_side(lkey, lval, n_left)
            rk2, rg2, nr2 = self._shuffled_side(rkey, rgroup, n_right)
            t1 = time.perf_counter()
            kc, kn, sums = self.engine.join_groupby_sum(lk2, lv2, nl2, rk2, rg2, nr2)
        else:
            rk, rk_bits, n_all = self._gather_build(rkey[0], rkey[1], n_right, pad_null=True)
            rg, rg_bits, _ = self._gather_build(rgroup[0], rgroup[1], n_right, pad_null=False)
            t1 = time.perf_counter()
            kc, kn, sums = self.engine.join_groupby_sum(lkey, lval, n_left, (rk, rk_bits, rkey[2]),
                                                        (rg, rg_bits, rgroup[2]), n_all)
        t2 = time.perf_counter()
        g = int(kc.shape[1])
        cells, nulls, part = kc[0], kn[0], sums[0]
        if hasattr(cells, "is_cuda"):
            import torch
            cells = cells.contiguous().view(torch.int64)
            has_null = bool(nulls.any().item()) if g else False
        else:
            import numpy as np
            cells = np.ascontiguousarray(cells).view(np.int64)
            has_null = bool(nulls.any()) if g else False
        # group cells are merged as plain 8-byte keys whatever g's dtype was: cell equality is
        # the reference's key equality (device_utils.hpp key_cell)
        out = self.groupby.groupby_agg([(cells, _pack_bits(nulls) if has_null else None, 0)], g,
                                       [(part, None, 1)], [(0, 0)])
        t3 = time.perf_counter()
        self.last_wall_ms = {("shuffle_rows" if strategy == "shuffle" else "allgather_build"): (t1 - t0) * 1e3, "local_join_groupby": (t2 - t1) * 1e3,
                             "exchange_merge": (t3 - t2) * 1e3}
        return out
