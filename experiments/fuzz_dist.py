#!/usr/bin/env python3
"""Randomised parity of the IN-LIBRARY distributed groupby / join (dist.hip) with REAL ranks on one GPU: W processes, each a pandrs
context on cuda:0, the exchange over a host transport (pandrs_hip_comm_adopt_transport over gloo; RCCL refuses two ranks on one
device).  Every case: all ranks derive the same global frame from the seed, take random (uneven, possibly empty) row ranges, pass
null masks on random subsets of ranks (the layout agreement), host or device shards, mergeable or shuffled aggregate sets, 1-2 key
columns; rank 0 gathers the owners' results and compares the union with the oracle on the whole frame.  GPU box only.
usage: fuzz_dist.py [world] [n_cases] [seed]"""
import os, sys, socket, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def worker(rank, world, port, n_cases, seed0):
    import datetime, torch, torch.distributed as dist
    import pandrs_amd as pa
    from oracle import oracle as O
    from tests.helpers import assert_groupby_equal
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.cuda.set_device(0)
    ctx = pa.Context(0)

    def all_gather(b):
        parts = [None] * world; dist.all_gather_object(parts, b); return parts
    def all_reduce_max(vals):
        t = torch.tensor(vals, dtype=torch.int64); dist.all_reduce(t, op=dist.ReduceOp.MAX); return t.tolist()
    def all_to_all_v(parts):
        everyone = [None] * world; dist.all_gather_object(everyone, parts); return [everyone[src][rank] for src in range(world)]
    ctx.comm_adopt_transport(all_gather, all_reduce_max, all_to_all_v, rank, world)
    fails = 0
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 * 104729 + case)          # the SAME stream on every rank
        desc = "?"
        try:
            if rng.random() < float(os.environ.get("FUZZ_DIST_JOIN_FRAC", "0.3")):
                # ---- pandrs_hip_dist_join_groupby_sum: both sides row-range-sharded (uneven, possibly empty), device shards
                nl = int(rng.choice([0, 7, 4000, 300_000])); nr = int(rng.choice([0, 5, 3000, 80_000]))
                unique = rng.random() < 0.6
                rcells = (rng.permutation(max(nr * 3, 8))[:nr] if unique else rng.integers(0, max(nr // 2, 3), nr)).astype(np.int64) * 7919 - 5
                lcells = (rcells[rng.integers(0, nr, nl)] if nr else rng.integers(0, 9, nl).astype(np.int64)).copy()
                if nl: lcells[rng.random(nl) < 0.15] = 123456789
                vdt = int(rng.choice([O.I64, O.F64]))
                rg = rng.integers(0, int(rng.choice([3, 400, 30_000])), nr).astype(np.int64) - 7
                lv = rng.integers(-1000, 1000, nl).astype(np.int64) if vdt == O.I64 else rng.normal(10, 5, nl)
                lkm, lvm = rng.random(nl) < rng.choice([0, 0.02]), rng.random(nl) < rng.choice([0, 0.05])
                rkm, rgm = rng.random(nr) < rng.choice([0, 0.02]), rng.random(nr) < rng.choice([0, 0.02])
                cut = lambda m: [0] + [int(x) for x in np.sort(rng.integers(0, m + 1, world - 1))] + [m]
                bl, br = cut(nl), cut(nr)
                pass_rkm, pass_rgm = rng.random(world) < 0.6, rng.random(world) < 0.6
                for r in range(world):
                    if not pass_rkm[r]: rkm[br[r]:br[r + 1]] = False
                    if not pass_rgm[r]: rgm[br[r]:br[r + 1]] = False
                d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
                l0, l1, r0, r1 = bl[rank], bl[rank + 1], br[rank], br[rank + 1]
                desc = "JOIN nl=%d nr=%d unique=%d vdt=%d cuts=%s / %s masks=%s %s" % (nl, nr, unique, vdt, bl, br, pass_rkm.astype(int).tolist(), pass_rgm.astype(int).tolist())
                kc, kn, oa = ctx.dist_join_groupby_sum((d(lcells[l0:l1]), d(O.pack_mask(lkm[l0:l1])) if lkm.any() else None, O.I64),
                                                       (d(lv[l0:l1]), d(O.pack_mask(lvm[l0:l1])) if lvm.any() else None, vdt), l1 - l0,
                                                       (d(rcells[r0:r1]), d(O.pack_mask(rkm[r0:r1])) if pass_rkm[rank] else None, O.I64),
                                                       (d(rg[r0:r1]), d(O.pack_mask(rgm[r0:r1])) if pass_rgm[rank] else None, O.I64), r1 - r0)
                parts = [None] * world
                dist.all_gather_object(parts, (kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy()))
                if rank == 0:
                    got = tuple(np.concatenate([p[i] for p in parts], axis=1) for i in range(3))
                    want = O.join_groupby_sum((lcells, O.pack_mask(lkm) if lkm.any() else None, O.I64), (lv, O.pack_mask(lvm) if lvm.any() else None, vdt), nl,
                                              (rcells, O.pack_mask(rkm) if rkm.any() else None, O.I64), (rg, O.pack_mask(rgm) if rgm.any() else None, O.I64), nr)
                    assert got[0].shape[1] == want[0].shape[1], "groups: %d vs %d" % (got[0].shape[1], want[0].shape[1])
                    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[0] if vdt == O.I64 else [], rtol=1e-9)
                    print("ok   %3d %s" % (case, desc), flush=True)
                dist.barrier()
                continue
            n = int(rng.choice([0, 5, 3000, 200_000, 900_000]))
            g = int(rng.choice([1, 7, 500, 40_000, 300_000]))
            cuts = np.sort(rng.integers(0, n + 1, world - 1)) if rng.random() < 0.7 else np.array([n * (r + 1) // world for r in range(world - 1)])
            if rng.random() < 0.2 and world > 1: cuts[0] = 0                         # rank 0 empty
            b = [0] + [int(x) for x in cuts] + [n]
            lo, hi = b[rank], b[rank + 1]
            kd = int(rng.choice([O.I64, O.I64, O.U32CODE, O.F64]))
            ids = rng.integers(0, g, n)
            if rng.random() < 0.3 and n: ids[rng.random(n) < 0.5] = 0
            if kd == O.I64: k = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
            elif kd == O.U32CODE: k = ids.astype(np.uint32)
            else: k = ids.astype(np.float64) * 0.5 - 3.0
            km = rng.random(n) < rng.choice([0, 0, 0.01])
            nk = 1 if rng.random() < 0.75 else 2
            nv = int(rng.integers(1, 4))
            vdata, vmask, vdt = [], [], []
            for _ in range(nv):
                if rng.random() < 0.6: vdata.append(np.round(rng.normal(50, 20, n), 2)); vdt.append(O.F64)
                else: vdata.append(rng.integers(-10**6, 10**6, n).astype(np.int64)); vdt.append(O.I64)
                vmask.append(rng.random(n) < rng.choice([0, 0.1, 0.5]))
            mask_ranks = [rng.random(world) < 0.6 for _ in range(nv)]               # which ranks PASS a mask for column c
            for c in range(nv):                                                      # a rank that passes none must hold no null there
                for r in range(world):
                    if not mask_ranks[c][r]: vmask[c][b[r]:b[r + 1]] = False
            general = rng.random() < 0.35
            ops = [O.SUM, O.MEAN, O.MIN, O.MAX, O.COUNT] + ([O.STD, O.MEDIAN, O.NUNIQUE, O.VAR] if general else [])
            aggs = [(int(rng.integers(0, nv)), int(rng.choice(ops))) for _ in range(int(rng.integers(1, 7)))]
            host = rng.random() < 0.3
            dev = (lambda a: np.ascontiguousarray(a)) if host else (lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda())
            key_mask_passed = km.any() or rng.random() < 0.3
            keys_all = [(k, O.pack_mask(km) if key_mask_passed else None, kd)]
            keys = [(dev(k[lo:hi]), dev(O.pack_mask(km[lo:hi])) if key_mask_passed else None, kd)]
            kdts = [kd]
            if nk == 2:
                k2 = (ids % 5).astype(np.uint32)
                keys_all.append((k2, None, O.U32CODE)); keys.append((dev(k2[lo:hi]), None, O.U32CODE)); kdts.append(O.U32CODE)
                if kd == O.F64:       # composite keys travel as i64 / f64 / u32 payload: keep the first column narrow
                    k0 = (ids % 1000).astype(np.int64) - 50
                    keys_all[0] = (k0, keys_all[0][1], O.I64); keys[0] = (dev(k0[lo:hi]), keys[0][1], O.I64); kdts[0] = O.I64
            vals_all = [(vdata[c], O.pack_mask(vmask[c]) if vmask[c].any() else None, vdt[c]) for c in range(nv)]
            vals = [(dev(vdata[c][lo:hi]), dev(O.pack_mask(vmask[c][lo:hi])) if mask_ranks[c][rank] else None, vdt[c]) for c in range(nv)]
            desc = "n=%d g=%d cuts=%s kd=%d nk=%d nv=%d vdt=%s aggs=%s host=%d masks=%s" % (n, g, b, kd, nk, nv, vdt, aggs, host, [m.astype(int).tolist() for m in mask_ranks])
            ctx.dist_groupby_compute(keys, hi - lo, vals, aggs)
            kc, kn, oa = ctx.groupby_fetch(to_device=False)
            parts = [None] * world
            dist.all_gather_object(parts, (np.asarray(kc), np.asarray(kn), np.asarray(oa)))
            if rank == 0:
                got = tuple(np.concatenate([p[i] for p in parts], axis=1) for i in range(3))
                want = O.groupby_agg(keys_all, n, vals_all, aggs)
                assert got[0].shape[1] == want[0].shape[1], "groups: %d vs %d (owners must be disjoint)" % (got[0].shape[1], want[0].shape[1])
                exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT, O.MEDIAN, O.NUNIQUE) or (vdt[c] == O.I64 and op == O.SUM)]
                assert_groupby_equal(got, want, kdts, int_exact_rows=exact, rtol=1e-9)
                print("ok   %3d %s" % (case, desc), flush=True)
        except Exception:
            fails += 1
            print("FAIL %3d rank %d %s" % (case, rank, desc), flush=True); traceback.print_exc()
        dist.barrier()
    tot = torch.tensor([fails]); dist.all_reduce(tot)
    if rank == 0: print("fuzz_dist done: world %d, %d cases, %d failing rank-cases" % (world, n_cases, int(tot.item())), flush=True)
    dist.barrier(); dist.destroy_process_group(); ctx.close()
    if int(tot.item()): sys.exit(1)

if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(world, port, n_cases, seed0), nprocs=world, join=True)
