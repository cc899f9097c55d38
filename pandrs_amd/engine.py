"""Thin host wrapper over the C ABI: one `Context` = one HIP stream + workspace on one GPU.

Columns are `(data, null_mask, dtype)` triples.  `data`/`null_mask` are either numpy arrays
(host memory: the library stages them over PCIe) or torch CUDA tensors (device memory: nothing
leaves HBM).  torch is used only as the device allocator; no torch op is on the compute path.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_NP_OF = {L.I64: np.int64, L.F64: np.float64, L.U32CODE: np.uint32, L.BOOLBITS: np.uint8, L.CELL64: np.uint64}


class PandrsHipError(RuntimeError):
    """Maps onto pandrs::Error (reference src/core/error.rs): .status is the C status code."""

    def __init__(self, status, message):
        super().__init__("[status %d] %s" % (status, message))
        self.status = status
        self.message = message


class ColumnTypeMismatch(PandrsHipError):
    pass


class OperationFailed(PandrsHipError):
    pass


class BelowThreshold(PandrsHipError):
    """PANDRS_HIP_ERR_BELOW_THRESHOLD: fewer rows than GpuConfig.min_size_threshold — the caller keeps its CPU path
    (src/optimized/split_dataframe/gpu.rs:30-32).  This package has no CPU path: the error is the answer."""


class EmptyError(PandrsHipError):
    """Error::Empty (src/optimized/split_dataframe/aggregate.rs:87): no non-null value to aggregate."""


def _raise(status):
    msg = L.last_error()
    if status == L.ERR_TYPE_MISMATCH:
        raise ColumnTypeMismatch(status, msg)
    if status == L.ERR_OPERATION_FAILED:
        raise OperationFailed(status, msg)
    if status == L.ERR_BELOW_THRESHOLD:
        raise BelowThreshold(status, msg)
    raise PandrsHipError(status, msg)


def _is_torch(x):
    return x is not None and type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return x.data_ptr()
    return x.ctypes.data


class ResidentColumn:
    """A column uploaded ONCE with `Context.upload_column` (pandrs_hip_column_upload): usable wherever a
    (data, null_mask, dtype) triple is, any number of times, until `release()` / the context is closed.  The
    reference's columns are immutable Arc<[T]> (src/column/int64_column.rs:10), so a shim can keep them in HBM
    across aggregate / join calls (src/optimized/dataframe/transformations.rs:524-577)."""

    def __init__(self, ctx, desc, n_rows):
        self.ctx, self.desc, self.n_rows, self.dtype = ctx, desc, n_rows, desc.dtype

    def release(self):
        if self.desc is not None and getattr(self.ctx, "h", None):
            st = self.ctx.lib.pandrs_hip_column_release(self.ctx.h, C.byref(self.desc))
            self.desc = None
            if st:
                _raise(st)
        self.desc = None

    def __iter__(self):        # unpacks like a triple: (self, None, dtype)
        return iter((self, None, self.dtype))

    def __getitem__(self, i):  # ... and indexes like one (col[0] is the data, col[2] the dtype)
        return (self, None, self.dtype)[i]

    def __len__(self):
        return 3


class Context:
    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        st = self.lib.pandrs_hip_ctx_create(int(device), C.byref(h))
        if st:
            _raise(st)
        self.h = h
        self.device = device

    def close(self):
        self.comm_destroy()
        if getattr(self, "h", None):
            self.lib.pandrs_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------------
    def _cols(self, cols, keep, wait=True):
        arr = (L.Column * max(len(cols), 1))()
        space = None
        for i, (data, mask, dt) in enumerate(cols):
            if isinstance(data, ResidentColumn):
                if data.desc is None or data.ctx is not self:
                    raise ValueError("resident column released, or uploaded through another context")
                if space not in (None, L.MEM_DEVICE):
                    raise ValueError("columns of one call must all be host or all device")
                space = L.MEM_DEVICE
                keep.append(data)
                arr[i].data, arr[i].null_mask, arr[i].dtype = data.desc.data, data.desc.null_mask, data.desc.dtype
                continue
            if _is_torch(data):
                sp = L.MEM_DEVICE
                if not data.is_contiguous():
                    data = data.contiguous()
                if mask is not None and not _is_torch(mask):
                    raise ValueError("device column with a host null mask")
            else:
                sp = L.MEM_HOST
                data = np.ascontiguousarray(data, dtype=_NP_OF[dt])
                if mask is not None:
                    mask = np.ascontiguousarray(mask, dtype=np.uint8)
            if space is None:
                space = sp
            elif space != sp:
                raise ValueError("columns of one call must all be host or all device")
            keep.extend([data, mask])
            arr[i].data = _ptr(data)
            arr[i].null_mask = _ptr(mask)
            arr[i].dtype = int(dt)
        if space == L.MEM_DEVICE and wait and any(_is_torch(next(iter(c))) for c in cols):
            self._wait_for_producer()
        return arr, (space if space is not None else L.MEM_HOST)

    def _wait_for_producer(self):
        """The library runs on its own non-blocking stream and (like any C ABI over raw device
        pointers) requires its device inputs to be complete when the call starts.  Tensors handed
        over from torch may still be being written by kernels queued on torch's current stream."""
        import torch
        torch.cuda.current_stream(self.device).synchronize()

    @staticmethod
    def _aggs(aggs):
        arr = (L.AggSpec * max(len(aggs), 1))()
        for i, (c, op) in enumerate(aggs):
            arr[i].col, arr[i].op = int(c), int(op)
        return arr

    # -- resident columns -------------------------------------------------------------------------
    def upload_column(self, data, mask, dtype):
        """Host column -> HBM, once (pandrs_hip_column_upload).  -> ResidentColumn."""
        data = np.ascontiguousarray(data, dtype=_NP_OF[dtype])
        n = int(data.shape[0]) * (8 if dtype == L.BOOLBITS else 1)
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            if dtype != L.BOOLBITS:
                assert mask.shape[0] >= (n + 7) // 8
        return self.upload_column_n(data, mask, dtype, n)

    def upload_column_n(self, data, mask, dtype, n_rows):
        host = L.Column(_ptr(data), _ptr(mask), int(dtype), 0)
        dev = L.Column()
        st = self.lib.pandrs_hip_column_upload(self.h, C.byref(host), int(n_rows), C.byref(dev))
        if st:
            _raise(st)
        return ResidentColumn(self, dev, int(n_rows))

    def resident_bytes(self):
        b, n = C.c_int64(0), C.c_int64(0)
        st = self.lib.pandrs_hip_resident_bytes(self.h, C.byref(b), C.byref(n))
        if st:
            _raise(st)
        return b.value, n.value

    def synchronize(self):
        st = self.lib.pandrs_hip_ctx_synchronize(self.h)
        if st:
            _raise(st)

    def set_option(self, name, value):
        st = self.lib.pandrs_hip_ctx_set_option(self.h, name.encode(), int(value))
        if st:
            _raise(st)

    def reserve(self, nbytes):
        st = self.lib.pandrs_hip_ctx_reserve(self.h, int(nbytes))
        if st:
            _raise(st)

    def timings(self):
        t = L.Timings()
        st = self.lib.pandrs_hip_get_timings(self.h, C.byref(t))
        if st:
            _raise(st)
        return {
            "total_ms": t.total_ms,
            "phase_ms": {L.PHASE_NAMES[i]: t.phase_ms[i] for i in range(L.MAX_PHASES)
                         if L.PHASE_NAMES[i] and t.phase_ms[i] > 0},
            "algorithmic_bytes": t.algorithmic_bytes, "n_partitions": t.n_partitions,
            "table_slots": t.table_slots, "retries": t.retries,
            "estimated_groups": t.estimated_groups, "absorbed_rows": t.absorbed_rows,
        }

    # -- groupby -----------------------------------------------------------------------------------
    def groupby_compute(self, keys, n_rows, vals, aggs):
        """Runs the device pipeline and keeps the result in the context.  -> n_groups."""
        keep = []
        kc, sp1 = self._cols(keys, keep, wait=not vals)       # one wait covers both column lists
        vc, sp2 = self._cols(vals, keep) if vals else ((L.Column * 1)(), sp1)
        if vals and sp1 != sp2:
            raise ValueError("keys and values must live in the same memory space")
        ng = C.c_int64(0)
        st = self.lib.pandrs_hip_groupby_agg(self.h, sp1, kc, len(keys), int(n_rows), vc, len(vals),
                                             self._aggs(aggs), len(aggs), C.byref(ng))
        if st:
            _raise(st)
        self._last = (len(keys), len(aggs), sp1, ng.value)
        return ng.value

    def groupby_fetch(self, to_device=None):
        """-> (key_cells[n_keys, G] u64, key_null[n_keys, G] u8, aggs[n_aggs, G] f64)."""
        n_keys, n_aggs, space, g = self._last
        dev = space == L.MEM_DEVICE if to_device is None else to_device
        if dev:
            import torch
            d = "cuda:%d" % self.device
            kc = torch.empty((n_keys, g), dtype=torch.int64, device=d)
            kn = torch.empty((n_keys, g), dtype=torch.uint8, device=d)
            oa = torch.empty((n_aggs, g), dtype=torch.float64, device=d)
            row = lambda t, i: t[i].data_ptr()
        else:
            kc = np.empty((n_keys, g), np.uint64)
            kn = np.empty((n_keys, g), np.uint8)
            oa = np.empty((n_aggs, g), np.float64)
            row = lambda t, i: t[i].ctypes.data
        pk = (C.c_void_p * max(n_keys, 1))(*[row(kc, i) for i in range(n_keys)])
        pn = (C.c_void_p * max(n_keys, 1))(*[row(kn, i) for i in range(n_keys)])
        pa = (C.c_void_p * max(n_aggs, 1))(*[row(oa, i) for i in range(n_aggs)])
        st = self.lib.pandrs_hip_groupby_fetch(self.h, L.MEM_DEVICE if dev else L.MEM_HOST, pk, pn, pa)
        if st:
            _raise(st)
        return kc, kn, oa

    def groupby_agg(self, keys, n_rows, vals, aggs):
        self.groupby_compute(keys, n_rows, vals, aggs)
        return self.groupby_fetch()

    def groupby_indices(self, keys, n_rows):
        """group_by's row -> group assignment (grouping.rs:22-115) in CSR form.
        -> (key_cells[n_keys, G] u64, key_null[n_keys, G] u8, offsets[G + 1] i64, rows[n_rows] i64);
        group g = rows[offsets[g]:offsets[g + 1]], ascending."""
        keep = []
        kc, sp = self._cols(keys, keep)
        ng = C.c_int64(0)
        st = self.lib.pandrs_hip_groupby_indices(self.h, sp, kc, len(keys), int(n_rows), C.byref(ng))
        if st:
            _raise(st)
        g, nk = ng.value, len(keys)
        if sp == L.MEM_DEVICE:
            import torch
            d = "cuda:%d" % self.device
            cells = torch.empty((nk, g), dtype=torch.int64, device=d)
            nulls = torch.empty((nk, g), dtype=torch.uint8, device=d)
            off = torch.empty(g + 1, dtype=torch.int64, device=d)
            rows = torch.empty(int(n_rows), dtype=torch.int64, device=d)
            row = lambda t, i: t[i].data_ptr()
        else:
            cells = np.empty((nk, g), np.uint64)
            nulls = np.empty((nk, g), np.uint8)
            off = np.empty(g + 1, np.int64)
            rows = np.empty(int(n_rows), np.int64)
            row = lambda t, i: t[i].ctypes.data
        pk = (C.c_void_p * nk)(*[row(cells, i) for i in range(nk)])
        pn = (C.c_void_p * nk)(*[row(nulls, i) for i in range(nk)])
        st = self.lib.pandrs_hip_groupby_indices_fetch(self.h, sp, pk, pn, _ptr(off), _ptr(rows))
        if st:
            _raise(st)
        return cells, nulls, off, rows

    def groupby_agg_chunked(self, keys, n_rows, vals, aggs, chunk_rows):
        """Groupby over more rows than one call takes (the per-call limit is 2^32 rows: 32-bit row
        offsets) or than the workspace should hold at once: the rows are processed in chunks of
        `chunk_rows` (a multiple of 8, so that null bitmaps split on byte boundaries), every chunk
        leaves mergeable partial states, and one merge finishes.  Single key column and the mergeable
        aggregates (Sum / Mean / Min / Max / Count), like the multi-GPU path it reuses."""
        if chunk_rows <= 0 or chunk_rows % 8:
            raise ValueError("chunk_rows must be a positive multiple of 8")
        if len(keys) != 1:
            raise NotImplementedError("chunked groupby takes one key column")
        def cut(col, lo, hi):
            data, mask, dt = col
            if dt == L.BOOLBITS:
                data = data[lo // 8:(hi + 7) // 8]
            else:
                data = data[lo:hi]
            return (data, None if mask is None else mask[lo // 8:(hi + 7) // 8], dt)
        recs = []
        for lo in range(0, int(n_rows), int(chunk_rows)):
            hi = min(lo + int(chunk_rows), int(n_rows))
            self.groupby_partials([cut(keys[0], lo, hi)], hi - lo, [cut(v, lo, hi) for v in vals], aggs)
            rec, _ = self.partials_split(1)
            recs.append(rec)
        if not recs:
            return self.groupby_agg(keys, 0, vals, aggs)
        if _is_torch(recs[0]):
            import torch
            allrec = torch.cat(recs, dim=0)
        else:
            allrec = np.concatenate(recs, axis=0)
        self.groupby_merge(keys[0][2], allrec, [v[2] for v in vals], [v[1] is not None for v in vals], aggs)
        return self.groupby_fetch()

    # -- row shuffle by key owner (multi-GPU) ------------------------------------------------------------
    def shuffle_split(self, key, payload, n_rows, n_ranks, drop_null_keys=False):
        """Buckets this shard's rows by the owner rank of their key, rank-contiguous.
        -> (cells[n] u64, key_null[n] u8, [payload u64/i64 [n]], [payload null bytes [n] or None], counts[n_ranks])"""
        keep = []
        kc, sp = self._cols([key], keep)
        pc, sp2 = self._cols(payload, keep) if payload else ((L.Column * 1)(), sp)
        if payload and sp != sp2:
            raise ValueError("key and payload must live in the same memory space")
        counts = (C.c_int64 * n_ranks)()
        n_out = C.c_int64(0)
        st = self.lib.pandrs_hip_shuffle_split(self.h, sp, kc, pc, len(payload), int(n_rows), int(n_ranks),
                                               1 if drop_null_keys else 0, counts, C.byref(n_out))
        if st:
            _raise(st)
        n, npay = n_out.value, len(payload)
        masked = [p[1] is not None for p in payload]
        if sp == L.MEM_DEVICE:
            import torch
            d = "cuda:%d" % self.device
            new = lambda dt: torch.empty(n, dtype=dt, device=d)
            cells, knull = new(torch.int64), new(torch.uint8)
            pays = [new(torch.int64) for _ in range(npay)]
            pnull = [new(torch.uint8) if m else None for m in masked]
        else:
            cells, knull = np.empty(n, np.uint64), np.empty(n, np.uint8)
            pays = [np.empty(n, np.uint64) for _ in range(npay)]
            pnull = [np.empty(n, np.uint8) if m else None for m in masked]
        pp = (C.c_void_p * max(npay, 1))(*[_ptr(a) for a in pays])
        pn = (C.c_void_p * max(npay, 1))(*[_ptr(a) for a in pnull])
        st = self.lib.pandrs_hip_shuffle_fetch(self.h, sp, _ptr(cells), _ptr(knull), pp, pn)
        if st:
            _raise(st)
        return cells, knull, pays, pnull, [int(x) for x in counts]

    def key_hash_cells(self, keys, n_rows):
        """One 64-bit hash cell per row of a composite key (the shuffle key of a multi-key groupby)."""
        keep = []
        kc, sp = self._cols(keys, keep)
        if sp == L.MEM_DEVICE:
            import torch
            out = torch.empty(int(n_rows), dtype=torch.int64, device="cuda:%d" % self.device)
        else:
            out = np.empty(int(n_rows), np.uint64)
        st = self.lib.pandrs_hip_key_hash_cells(self.h, sp, kc, len(keys), int(n_rows), _ptr(out))
        if st:
            _raise(st)
        return out

    def bytes_to_bitmap(self, flags):
        """One byte per row (non-zero = set) -> LSB-first bitmap, same kind (numpy / torch) as the input."""
        n = int(flags.shape[0])
        if _is_torch(flags):
            import torch
            self._wait_for_producer()
            flags = flags.contiguous()
            out = torch.empty((n + 7) // 8, dtype=torch.uint8, device=flags.device)
            sp = L.MEM_DEVICE
        else:
            flags = np.ascontiguousarray(flags, np.uint8)
            out = np.empty((n + 7) // 8, np.uint8)
            sp = L.MEM_HOST
        st = self.lib.pandrs_hip_bytes_to_bitmap(self.h, sp, _ptr(flags), n, _ptr(out))
        if st:
            _raise(st)
        return out

    # -- mergeable partials (multi-GPU) --------------------------------------------------------------
    def groupby_partials(self, keys, n_rows, vals, aggs):
        """-> (n_groups, n_state); partial rows stay in the context until partials_split."""
        keep = []
        kc, sp1 = self._cols(keys, keep)
        vc, _ = self._cols(vals, keep) if vals else ((L.Column * 1)(), sp1)
        ng, ns = C.c_int64(0), C.c_int32(0)
        st = self.lib.pandrs_hip_groupby_partials(self.h, sp1, kc, len(keys), int(n_rows), vc, len(vals),
                                                  self._aggs(aggs), len(aggs), C.byref(ng), C.byref(ns))
        if st:
            _raise(st)
        self._last_partials = (ng.value, ns.value, sp1)
        return ng.value, ns.value

    def partials_split(self, n_ranks, device=None):
        """Buckets the retained partial rows by owner rank.
        -> (records[G, 2 + n_state] int64/uint64: key, key_null, states..., counts[n_ranks]);
        records are rank-contiguous, ready for one all-to-all."""
        g, ns, space = self._last_partials
        dev = space == L.MEM_DEVICE if device is None else device
        counts = (C.c_int64 * n_ranks)()
        if dev:
            import torch
            rec = torch.empty((g, 2 + ns), dtype=torch.int64, device="cuda:%d" % self.device)
        else:
            rec = np.empty((g, 2 + ns), np.uint64)
        st = self.lib.pandrs_hip_partials_split(self.h, L.MEM_DEVICE if dev else L.MEM_HOST, n_ranks,
                                                _ptr(rec), counts)
        if st:
            _raise(st)
        return rec, [int(c) for c in counts]

    def groupby_merge(self, key_dtype, records, val_dtypes, val_has_nulls, aggs):
        """records: [n, 2 + n_state] packed partial rows gathered from all peers."""
        dev = _is_torch(records)
        if not dev:
            records = np.ascontiguousarray(records, np.uint64)
        else:
            records = records.contiguous()
            self._wait_for_producer()
        n_rows = int(records.shape[0])
        nv = len(val_dtypes)
        vd = (C.c_int32 * max(nv, 1))(*[int(x) for x in val_dtypes])
        vh = (C.c_uint8 * max(nv, 1))(*[1 if x else 0 for x in val_has_nulls])
        ng = C.c_int64(0)
        st = self.lib.pandrs_hip_groupby_merge(self.h, L.MEM_DEVICE if dev else L.MEM_HOST, int(key_dtype),
                                               _ptr(records) if n_rows else None, n_rows,
                                               vd, nv, vh, self._aggs(aggs), len(aggs), C.byref(ng))
        if st:
            _raise(st)
        self._last = (1, len(aggs), L.MEM_DEVICE if dev else L.MEM_HOST, ng.value)
        return ng.value

    # -- join --------------------------------------------------------------------------------------------
    def join_indices_compute(self, lkey, n_left, rkey, n_right, how):
        """Runs the join and keeps the pairs in the context (pandrs_hip_join_indices).  -> (n_pairs, memory space)"""
        keep = []
        lc, sp1 = self._cols([lkey], keep)
        rc, sp2 = self._cols([rkey], keep)
        if sp1 != sp2:
            raise ValueError("both key columns must live in the same memory space")
        n = C.c_int64(0)
        st = self.lib.pandrs_hip_join_indices(self.h, sp1, lc, int(n_left), rc, int(n_right), int(how), C.byref(n))
        if st:
            _raise(st)
        return n.value, sp1

    def join_indices(self, lkey, n_left, rkey, n_right, how):
        n, sp1 = self.join_indices_compute(lkey, n_left, rkey, n_right, how)
        if sp1 == L.MEM_DEVICE:
            import torch
            d = "cuda:%d" % self.device
            li = torch.empty(n, dtype=torch.int64, device=d)
            ri = torch.empty(n, dtype=torch.int64, device=d)
        else:
            li, ri = np.empty(n, np.int64), np.empty(n, np.int64)
        st = self.lib.pandrs_hip_join_fetch(self.h, sp1, _ptr(li), _ptr(ri))
        if st:
            _raise(st)
        return li, ri

    def gather(self, src, mask, idx, fill, dtype):
        """Device tensors only.  -> torch tensor (uint8 per row for BOOLBITS sources)."""
        import torch
        self._wait_for_producer()        # src / mask / idx may still be being written on torch's current stream
        n = idx.numel()
        d = idx.device
        fn, out = {
            L.I64: (self.lib.pandrs_hip_gather_i64, torch.empty(n, dtype=torch.int64, device=d)),
            L.F64: (self.lib.pandrs_hip_gather_f64, torch.empty(n, dtype=torch.float64, device=d)),
            L.U32CODE: (self.lib.pandrs_hip_gather_u32, torch.empty(n, dtype=torch.int32, device=d)),
            L.BOOLBITS: (self.lib.pandrs_hip_gather_bool, torch.empty(n, dtype=torch.uint8, device=d)),
        }[dtype]
        fill = float(fill) if dtype == L.F64 else int(fill)
        st = fn(self.h, L.MEM_DEVICE, _ptr(src), _ptr(mask), _ptr(idx), n, fill, _ptr(out))
        if st:
            _raise(st)
        return out

    def join_gather(self, col, n_src, n_out, side, fill=0, key_right=None, n_right=0):
        """One column of the joined frame through the pairs this context retains (the last join_indices / join_indices_compute):
        pandrs_hip_join_gather (side 0 = the pairs' left rows, 1 = their right rows), or — key_right given — pandrs_hip_join_gather_key
        (the left key's value, else the right key's: join.rs:364-470).  `col` is a (data, mask, dtype) triple on the host or the
        device, or a ResidentColumn; -> numpy array of n_out elements (uint8 per row for BOOLBITS sources)."""
        keep = []
        cc, sp = self._cols([tuple(col)], keep)
        dtype = tuple(col)[2]
        out = np.empty(int(n_out), {L.I64: np.int64, L.F64: np.float64, L.U32CODE: np.uint32, L.BOOLBITS: np.uint8}[dtype])
        fill_bits = int(np.float64(fill).view(np.uint64)) if dtype == L.F64 else int(fill) & 0xFFFFFFFFFFFFFFFF
        if key_right is None:
            st = self.lib.pandrs_hip_join_gather(self.h, sp, cc, int(n_src), int(side), fill_bits, L.MEM_HOST, _ptr(out))
        else:
            kc, sp2 = self._cols([tuple(key_right)], keep)
            if sp2 != sp:
                raise ValueError("both key columns must live in one memory space")
            st = self.lib.pandrs_hip_join_gather_key(self.h, sp, cc, int(n_src), kc, int(n_right), fill_bits, L.MEM_HOST, _ptr(out))
        if st:
            _raise(st)
        return out

    def join_groupby_sum(self, lkey, lval, n_left, rkey, rgroup, n_right):
        keep = []
        a, sp = self._cols([lkey], keep)
        b, _ = self._cols([lval], keep)
        c, _ = self._cols([rkey], keep)
        d, _ = self._cols([rgroup], keep)
        ng = C.c_int64(0)
        st = self.lib.pandrs_hip_join_groupby_sum(self.h, sp, a, b, int(n_left), c, d, int(n_right), C.byref(ng))
        if st:
            _raise(st)
        self._last = (1, 1, sp, ng.value)
        return self.groupby_fetch()

    def column_std(self, col, n, population=True):
        """parallel_std_f64_value / parallel_var (jit/parallel.rs:222-250): sqrt(max(sum_sq / n - mean^2, 0)),
        n <= 1 => 0.0.  -> (std, variance)."""
        keep = []
        cc, sp = self._cols([col], keep)
        s, q, cnt = C.c_double(0), C.c_double(0), C.c_int64(0)
        st = self.lib.pandrs_hip_reduce_moments(self.h, sp, cc, int(n), C.byref(s), C.byref(q), C.byref(cnt))
        if st:
            _raise(st)
        m = cnt.value
        if m <= 1:
            return 0.0, 0.0
        mean = s.value / m
        var = max(q.value / m - mean * mean, 0.0)
        if not population:
            var *= m / (m - 1.0)
        return var ** 0.5, var

    # -- the in-library RCCL exchange (pandrs_hip_dist_*) -------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from rank 0 (ncclGetUniqueId); the host hands them to every rank over any channel."""
        buf = C.create_string_buffer(128)
        st = L.load().pandrs_hip_comm_unique_id(buf)
        if st:
            _raise(st)
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        h = C.c_void_p()
        st = self.lib.pandrs_hip_comm_init(self.h, C.c_char_p(bytes(unique_id)), int(rank), int(world), C.byref(h))
        if st:
            _raise(st)
        self.comm = h
        self.comm_world = world
        return h

    def comm_adopt_transport(self, all_gather, all_reduce_max, all_to_all_v, rank, world):
        """pandrs_hip_comm_adopt_transport: the in-library exchange over host callbacks instead of RCCL.
        all_gather(send: bytes) -> list of world bytes objects; all_reduce_max(list of int) -> list of int;
        all_to_all_v(list of world bytes objects) -> list of world bytes objects (what each rank sent to this one)."""
        def _ag(_user, send, recv, nbytes):
            try:
                parts = all_gather(C.string_at(send, nbytes))
                for r, part in enumerate(parts):
                    assert len(part) == nbytes
                    C.memmove(recv + r * nbytes, part, nbytes)
                return 0
            except Exception:        # noqa: BLE001 — the library turns a non-zero status into an error on every rank
                import traceback
                traceback.print_exc()
                return 1

        def _ar(_user, vals, n):
            try:
                out = all_reduce_max([int(vals[i]) for i in range(n)])
                for i in range(n):
                    vals[i] = int(out[i])
                return 0
            except Exception:        # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        def _a2a(_user, send, sb, so, recv, rb, ro):
            try:
                parts = all_to_all_v([C.string_at(send + so[p], sb[p]) if sb[p] else b"" for p in range(world)])
                for p in range(world):
                    assert len(parts[p]) == rb[p], (p, len(parts[p]), rb[p])
                    if rb[p]:
                        C.memmove(recv + ro[p], parts[p], rb[p])
                return 0
            except Exception:        # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        t = L.Transport(None, L.ALL_GATHER_FN(_ag), L.ALL_REDUCE_MAX_FN(_ar), L.ALL_TO_ALL_V_FN(_a2a))
        h = C.c_void_p()
        st = self.lib.pandrs_hip_comm_adopt_transport(C.byref(t), int(rank), int(world), C.byref(h))
        if st:
            _raise(st)
        self._transport = t            # the callbacks must outlive the communicator
        self.comm = h
        self.comm_world = world
        return h

    @staticmethod
    def alloc_events():
        n = C.c_int64(0)
        L.load().pandrs_hip_alloc_events(C.byref(n))
        return n.value

    def comm_destroy(self):
        if getattr(self, "comm", None):
            self.lib.pandrs_hip_comm_destroy(self.comm)
            self.comm = None

    def dist_groupby_compute(self, keys, n_rows, vals, aggs):
        """pandrs_hip_dist_groupby_agg: every rank passes its own row range.  -> groups owned by this rank."""
        keep = []
        kc, sp1 = self._cols(keys, keep)
        vc, _ = self._cols(vals, keep) if vals else ((L.Column * 1)(), sp1)
        ng = C.c_int64(0)
        st = self.lib.pandrs_hip_dist_groupby_agg(self.h, self.comm, sp1, kc, len(keys), int(n_rows), vc, len(vals),
                                                  self._aggs(aggs), len(aggs), C.byref(ng))
        if st:
            _raise(st)
        general = len(keys) > 1 or any(int(op) > L.COUNT for _, op in aggs)
        self._last = (len(keys), len(aggs), L.MEM_DEVICE if general else sp1, ng.value)     # the row shuffle leaves its result on the device
        return ng.value

    def dist_join_groupby_sum(self, lkey, lval, n_left, rkey, rgroup, n_right):
        keep = []
        a, sp = self._cols([lkey], keep)
        b, _ = self._cols([lval], keep)
        c, _ = self._cols([rkey], keep)
        d, _ = self._cols([rgroup], keep)
        ng = C.c_int64(0)
        st = self.lib.pandrs_hip_dist_join_groupby_sum(self.h, self.comm, sp, a, b, int(n_left), c, d, int(n_right), C.byref(ng))
        if st:
            _raise(st)
        self._last = (1, 1, sp, ng.value)
        return self.groupby_fetch()

    def column_stats(self, col, n):
        """pandrs_hip_reduce_stats: one pass, everything K1's reference functions need.  -> dict."""
        keep = []
        cc, sp = self._cols([col], keep)
        out = L.ColumnStats()
        st = self.lib.pandrs_hip_reduce_stats(self.h, sp, cc, int(n), C.byref(out))
        if st:
            _raise(st)
        return {name: getattr(out, name) for name, _ in L.ColumnStats._fields_}

    def reduce_column(self, col, n):
        keep = []
        cc, sp = self._cols([col], keep)
        out = (C.c_double * 4)()
        cnt = C.c_int64(0)
        st = self.lib.pandrs_hip_reduce_column(self.h, sp, cc, int(n), out, C.byref(cnt))
        if st:
            _raise(st)
        return np.array(list(out)), cnt.value
