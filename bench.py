#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE config 2.

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the groupby-aggregate hot path over one batch of synthetic rows that are
already resident in HBM: 100 M rows per GPU, one sparse i64 key column with 1 M groups, four f64
value columns, sum/mean/min/max over each (16 aggregates).  N > 1 is weak scaling: every rank
holds its own 100 M-row shard; per step it pre-aggregates locally, exchanges the partial rows
with ONE all-to-all (RCCL) keyed on hash(key) mod N, and merges the partitions it owns.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable


def make_shard(torch, n_rows, n_groups, n_cols, seed, device):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    ids = torch.randint(0, n_groups, (n_rows,), device=device, generator=gen, dtype=torch.int64)
    # bit-mix dense ids to sparse i64 keys: x * 0x9E3779B97F4A7C15 ^ const, wrapping
    keys = ids * -7046029254386353131 ^ 0x5555AAAA5555AAAA
    del ids
    vals = [torch.randn(n_rows, device=device, generator=gen, dtype=torch.float64) * 10 + 100
            for _ in range(n_cols)]
    return keys, vals


def pmc_traffic(n, g, ncol):
    """HBM bytes per step from the committed rocprofv3 PMC passes (profiles/): counters cannot be read
    from inside the process, so the corrected per-launch figure is loaded — a STATIC number measured in a
    separate `rocprofv3 --pmc` run of this command, not in this run — when it was collected on this exact
    workload; otherwise null.  -> (bytes or None, source)"""
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if t["config"] == {"rows": n, "groups": g, "value_cols": ncol}:
                return t["pipeline_hbm_bytes_per_step"], "profiles/%s (static: separate rocprofv3 --pmc passes, not this run)" % name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def cpu_baseline(n_sample, n_groups, n_cols, aggs):
    """The oracle's faithful restatement of the reference path (string keys + HashMap + per-group
    gather/fold, lazy.rs:186-404), single thread, on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(43)
    ids = rng.integers(0, n_groups, n_sample).astype(np.uint64)
    keys = (ids * np.uint64(0x9E3779B97F4A7C15) ^ np.uint64(0x5555AAAA5555AAAA)).view(np.int64)
    vals = [(rng.normal(100, 10, n_sample), None, O.F64) for _ in range(n_cols)]
    O.lib()
    t0 = time.perf_counter()
    O.groupby_agg([(keys, None, O.I64)], n_sample, vals, aggs, faithful=True)
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt / 1e6, "unit": "Mrows/s", "cores": 1, "kind": "port",
            "sample": "%d rows, %d-group key space, %d f64 cols, same 16 aggregates; string-keyed "
                      "HashMap restatement of lazy.rs:186-404 (oracle_groupby_agg_ref), %.1f s"
                      % (n_sample, n_groups, n_cols, dt)}


def cpu_baseline_parallel(n_sample, n_groups, n_cols, aggs):
    """SURVEY.md 8(d)(i), second half: the faithful restatement in the reference's PARALLEL shape
    (par_groupby chunk-map + serial merge, par_aggregate folds) on the box's host-core share."""
    import numpy as np
    from oracle import oracle as O
    cores = min(len(os.sched_getaffinity(0)), 16)
    rng = np.random.default_rng(43)
    ids = rng.integers(0, n_groups, n_sample).astype(np.uint64)
    keys = (ids * np.uint64(0x9E3779B97F4A7C15) ^ np.uint64(0x5555AAAA5555AAAA)).view(np.int64)
    vals = [(rng.normal(100, 10, n_sample), None, O.F64) for _ in range(n_cols)]
    O.lib()
    t0 = time.perf_counter()
    O.groupby_agg([(keys, None, O.I64)], n_sample, vals, aggs, faithful=True, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt / 1e6, "unit": "Mrows/s", "cores": cores, "kind": "port",
            "sample": "%d rows, same shape; par_groupby / par_aggregate shaped restatement (oracle_groupby_agg_ref_mt), %.1f s" % (n_sample, dt)}


def cpu_baseline_typed(n_sample, n_groups, n_cols):
    """a FAIR typed CPU baseline beside the faithful one — i64 open-addressing
    hash, all host cores, count/sum/min/max per column (mean = sum / count) — so the GPU/CPU ratio is
    not just 'removed the strings'.  Not the reference's algorithm; reported as extra information."""
    import numpy as np
    from oracle import oracle as O
    cores = min(len(os.sched_getaffinity(0)), 16)          # the GPU box gives one GPU's job a 16-core share
    rng = np.random.default_rng(44)
    ids = rng.integers(0, n_groups, n_sample).astype(np.uint64)
    keys = (ids * np.uint64(0x9E3779B97F4A7C15) ^ np.uint64(0x5555AAAA5555AAAA)).view(np.int64)
    vals = [rng.normal(100, 10, n_sample) for _ in range(n_cols)]
    O.lib()
    t0 = time.perf_counter()
    O.groupby_typed_mt(keys, vals, cores)
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt / 1e6, "unit": "Mrows/s", "cores": cores, "kind": "typed-hash (not the reference's algorithm)",
            "sample": "%d rows, %d-group key space, %d f64 cols, count/sum/min/max per column, %.2f s" % (n_sample, n_groups, n_cols, dt)}


def cpu_baseline_join(n_left, n_right):
    """join_impl's index build in the reference's own shape — HashMap<String, Vec<usize>> over
    the right side, one formatted String + SipHash-1-3 per probed left row (join.rs:107-224), serial like the
    reference — on a bounded sample of the C5 shape (probe : build = 10 : 1, unique build keys, every probe row hits)."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(45)
    mix = np.uint64(0x9E3779B97F4A7C15)
    rk = (rng.permutation(n_right).astype(np.uint64) * mix).view(np.int64)
    lk = (rng.integers(0, n_right, n_left).astype(np.uint64) * mix).view(np.int64)
    O.lib()
    t0 = time.perf_counter()
    li, _ = O.join_indices((lk, None, O.I64), n_left, (rk, None, O.I64), n_right, O.INNER, faithful=True)
    dt = time.perf_counter() - t0
    assert len(li) == n_left
    return {"value": n_left / dt / 1e6, "unit": "Mrows/s (probe rows)", "cores": 1, "kind": "port",
            "sample": "inner join %d probe x %d build rows (unique i64 keys, every probe row matches); string-keyed HashMap "
                      "restatement of join.rs:107-224 (oracle_join_indices_ref), serial, %.1f s" % (n_left, n_right, dt)}


def extra_configs(torch, pa, ctx, device, steps=5):
    """The other single-GPU configurations of BASELINE.json beside the headline: same library,
    same timing (hipEvent time of the whole call on the library stream, mean over `steps` after one warm-up),
    inputs resident in HBM.  -> {name: {"ms", "algorithmic_bytes", "frac", "Mrows_per_s", "config"}}"""
    MIX = -7046029254386353131
    out = {}

    def timed(name, rows, config, fn, bytes_alg=None):
        fn()
        ms = 0.0
        for _ in range(steps):
            fn()
            t = ctx.timings()
            ms += t["total_ms"]
        ms /= steps
        b = bytes_alg if bytes_alg is not None else t["algorithmic_bytes"]
        out[name] = {"ms": ms, "algorithmic_bytes": b, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "Mrows_per_s": rows / ms / 1e3, "config": config}

    gen = torch.Generator(device=device)
    gen.manual_seed(4242)
    # north-star line: 100 M rows, sparse i64 key with 1 M groups, ONE f64 sum (the config the 60 % target is quoted on)
    n, g = 100_000_000, 1_000_000
    k = torch.randint(0, g, (n,), device=device, generator=gen, dtype=torch.int64) * MIX ^ 0x5555AAAA5555AAAA
    v = torch.randn(n, device=device, generator=gen, dtype=torch.float64) * 10 + 100
    timed("north_star_sum", n, "100M rows, i64 key (1M groups), sum of one f64 column",
          lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)]))
    # C1's shape (1 K groups, one f64 sum: the reference's own example, examples/optimized_groupby_example.rs) at 100 M rows: every group
    # fits a workgroup's LDS table, so the input is read once and nothing else moves (absorb.hip with nothing to spill)
    k1 = torch.randint(0, 1000, (n,), device=device, generator=gen, dtype=torch.int64) * MIX
    timed("c1_shape_100m", n, "C1's shape at 100M rows: i64 key (1K groups), sum of one f64 column",
          lambda: ctx.groupby_compute([(k1, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)]))
    del k, k1
    # C2's 80/20-skew variant (benches/enhanced_comprehensive_benchmark.rs:53-59): 80 % of the rows on a fifth of the 1 M keys,
    # the same four f64 columns x sum/mean/min/max.  (The sampled estimate cannot see the tail here; full tables hand their unplaced
    # rows to a run of their own.)
    g = 1_000_000
    hot = torch.rand(n, device=device, generator=gen) < 0.8
    ks = torch.where(hot, torch.randint(0, g // 5, (n,), device=device, generator=gen), torch.randint(0, g, (n,), device=device, generator=gen)) * MIX
    del hot
    vs = [v] + [torch.randn(n, device=device, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(3)]
    aggs16 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
    timed("c2_skew_80_20", n, "C2 with 80/20 skew: 100M rows, i64 key (1M groups, 80% of the rows on 200K of them), sum/mean/min/max over 4 f64 cols",
          lambda: ctx.groupby_compute([(ks, None, pa.I64)], n, [(x, None, pa.F64) for x in vs], aggs16),
          bytes_alg=n * 40 + g * (8 + 8 * 16))
    # ... and with the rows SORTED by key (round 4: no partition, one pass over the original columns — clustered.hip)
    ks = torch.sort(torch.randint(0, g, (n,), device=device, generator=gen))[0] * MIX
    timed("c2_sorted", n, "C2 with the rows sorted by key: 100M rows, i64 key (1M groups), sum/mean/min/max over 4 f64 cols",
          lambda: ctx.groupby_compute([(ks, None, pa.I64)], n, [(x, None, pa.F64) for x in vs], aggs16),
          bytes_alg=n * 40 + g * (8 + 8 * 16))
    del ks, vs
    # C3: 100 M rows, u32 string-pool codes, 10 K groups with 80/20 skew, 2 f64 columns x sum/mean/min/max + count
    g = 10_000
    hot = torch.rand(n, device=device, generator=gen) < 0.8
    codes = torch.where(hot, torch.randint(0, g // 5, (n,), device=device, generator=gen),
                        torch.randint(0, g, (n,), device=device, generator=gen)).to(torch.int32)
    del hot
    v2 = torch.randn(n, device=device, generator=gen, dtype=torch.float64)
    aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)] + [(0, pa.COUNT)]
    timed("c3", n, "100M rows, u32 string-pool code key (10K groups, 80/20 skew), sum/mean/min/max over 2 f64 cols + count",
          lambda: ctx.groupby_compute([(codes, None, pa.U32CODE)], n, [(v, None, pa.F64), (v2, None, pa.F64)], aggs))
    del codes, v2, v
    # C4 per-GPU shard: 125 M rows, sparse i64 key, 10 M groups, sum + count
    n, g = 125_000_000, 10_000_000
    k = torch.randint(0, g, (n,), device=device, generator=gen, dtype=torch.int64) * MIX
    v = torch.randn(n, device=device, generator=gen, dtype=torch.float64)
    timed("c4_shard", n, "C4 per-GPU shard: 125M rows, i64 key (10M groups), sum + count",
          lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.COUNT)]))
    del k, v
    # C5 per-GPU shard after the build side's all-gather: 62.5 M probe rows x 50 M build rows, fused join -> groupby-sum
    nl, nr, g = 62_500_000, 50_000_000, 100_000
    rkey = torch.randperm(nr, device=device, generator=gen) * MIX
    rgrp = torch.randint(0, g, (nr,), device=device, generator=gen, dtype=torch.int64)
    lkey = torch.randint(0, nr, (nl,), device=device, generator=gen, dtype=torch.int64) * MIX
    lval = torch.randn(nl, device=device, generator=gen, dtype=torch.float64)
    timed("c5_shard", nl, "C5 per-GPU shard: inner join 62.5M probe x 50M build (unique i64 keys), then groupby(g in [0,100000)).sum(v), fused",
          lambda: ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr),
          bytes_alg=nl * 16 + nr * 16 + g * 16)
    del lkey, lval
    # C5 whole on one GPU: 500 M probe rows x the same build side (the probe side dominates: L2-resident table regions)
    nl = 500_000_000
    lkey = torch.randint(0, nr, (nl,), device=device, generator=gen, dtype=torch.int64) * MIX
    lval = torch.randn(nl, device=device, generator=gen, dtype=torch.float64)
    timed("c5_one_gpu", nl, "C5 on one GPU: inner join 500M probe x 50M build (unique i64 keys), then groupby(g in [0,100000)).sum(v), fused",
          lambda: ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr),
          bytes_alg=nl * 16 + nr * 16 + g * 16)
    del lkey, lval, rkey, rgrp
    # join_indices (join_impl's index build, all pairs materialised in the reference's order): 50 M probe x 5 M build
    nl, nr = 50_000_000, 5_000_000
    rkey = torch.randperm(nr, device=device, generator=gen) * MIX
    lkey = torch.randint(0, nr, (nl,), device=device, generator=gen, dtype=torch.int64) * MIX
    timed("join_indices_50m_5m", nl, "inner join_indices 50M probe x 5M build (unique i64 keys, every probe row matches): (left row, right row) pairs in the reference's order",
          lambda: ctx.join_indices_compute((lkey, None, pa.I64), nl, (rkey, None, pa.I64), nr, pa.INNER),
          bytes_alg=nl * 8 + nr * 8 + nl * 16)
    del lkey, rkey
    torch.cuda.empty_cache()
    # C4 WHOLE on one GPU (1 B rows, 10 M groups, sum + count) — one call, N < 2^32
    n, g = 1_000_000_000, 10_000_000
    k = torch.randint(0, g, (n,), device=device, generator=gen, dtype=torch.int64)
    k.mul_(MIX)
    v = torch.randn(n, device=device, generator=gen, dtype=torch.float64)
    timed("c4_one_gpu", n, "C4 on one GPU: 1B rows, i64 key (10M groups), sum + count, one call",
          lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.COUNT)]))
    del k, v
    torch.cuda.empty_cache()
    # C1 (the reference's own CPU-runnable case): 1 M rows, 1 K groups, one f64 sum — wall time per call, launch-bound
    n, g = 1_000_000, 1_000
    k = torch.randint(0, g, (n,), device=device, generator=gen, dtype=torch.int64)
    v = torch.randn(n, device=device, generator=gen, dtype=torch.float64)
    f = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)])
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        f()
    wall_ms = (time.perf_counter() - t0) / 100 * 1e3
    b = n * 16 + g * 16
    out["c1"] = {"ms": wall_ms, "algorithmic_bytes": b, "frac": b / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "Mrows_per_s": n / wall_ms / 1e3, "config": "C1: 1M rows, i64 key (1K groups), sum of one f64 column; HOST WALL time per call "
                 "(Python + C ABI + two launches + completion poll), mean of 100"}
    return out


_REAL_STDOUT = None


def emit(record):
    """The one JSON line, on the process's real stdout."""
    sys.stdout.flush()
    if _REAL_STDOUT is not None:
        os.dup2(_REAL_STDOUT, 1)
    print(json.dumps(record), flush=True)
    if _REAL_STDOUT is not None:
        os.dup2(2, 1)


def dry_run(args):
    """PANDRS_BENCH_BACKEND=gloo: a rehearsal of the LAUNCH path on a box without GPUs (tests/test_bench_launch.py).
    Same rendezvous, barriers, max-over-ranks timing and record as the real run, but the local engine is the
    numpy stand-in of tests/cpu_engine.py and the record says so.  Never a measurement."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from pandrs_amd.dist import DistributedGroupBy
    from tests.cpu_engine import NumpyEngine
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist.init_process_group("gloo")
    n, g, ncol = min(args.rows, 200_000), min(args.groups, 5_000), args.cols
    rng = np.random.default_rng(42 + 1 + 1000 * rank)
    keys = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    vals = [(rng.normal(100, 10, n), None, 1) for _ in range(ncol)]
    aggs = [(c, op) for c in range(ncol) for op in (0, 1, 2, 3)]
    dgb = DistributedGroupBy(NumpyEngine(), dist, "cpu")
    step = lambda: dgb.groupby_agg([(keys, None, 0)], n, vals, aggs, fetch=False)
    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({"metric": "Mrows/sec groupby-agg", "value": n * world / (dt / args.steps) / 1e6, "unit": "Mrows/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                          "data": "synthetic", "dry_run": "gloo rehearsal of the launch path on CPU with the numpy stand-in engine: NOT a measurement",
                          "config": {"workload": "dry run: %d rows/rank, %d groups, %d f64 cols" % (n, g, ncol)}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start N fresh ranks (one per GPU) under
    torch.distributed.run and relay rank 0's JSON line.  Runs BEFORE torch or any HIP call is made in
    this process: a process that has touched the GPU must never exec or fork workers."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif p.returncode == 0:
        sys.stderr.write("bench.py: the ranks printed no record\n")
        return 1
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU")
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--cols", type=int, default=4)
    ap.add_argument("--cpu-sample", type=int, default=16_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra configs (north-star sum line, C3, C4 shard, C5 shard) reported beside the headline")
    ap.add_argument("--workload", choices=["groupby", "join", "c4"], default="groupby",
                    help="groupby = BASELINE config 2 (the headline line); join = config 5 shape "
                         "(probe --rows per GPU, build rows/10 per GPU, inner join -> groupby(g).sum(v)); "
                         "c4 = BASELINE config 4's per-GPU shard (125 M rows/GPU, 10 M groups, one f64 column, sum + count): "
                         "the configuration north_star names for the 8-GPU radix all-to-all")
    ap.add_argument("--join-strategy", choices=["auto", "allgather", "shuffle"], default="auto",
                    help="N > 1, --workload join: replicate the build side, or shuffle both sides to the owner of their key")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("PANDRS_BENCH_BACKEND") == "gloo":
        return dry_run(args)

    # stdout carries exactly ONE line, the JSON record: native libraries (RCCL prints a version banner
    # on its first collective) write to file descriptor 1 behind Python's back, so fd 1 points at stderr
    # until the record is printed
    sys.stdout.flush()
    global _REAL_STDOUT
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)

    import torch
    import pandrs_amd as pa

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(device))

    if args.workload == "join":
        return bench_join(args, torch, pa, dist, rank, local_rank, world, device)

    n, g, ncol = args.rows, args.groups, args.cols
    c4 = args.workload == "c4"
    if c4:       # BASELINE config 4: 1 B rows over 8 GPUs = 125 M per GPU, 10 M groups over the WHOLE job, sum + count of one column
        n, g, ncol = (125_000_000 if args.rows == 100_000_000 else args.rows), (10_000_000 if args.groups == 1_000_000 else args.groups), 1
    keys, vals = make_shard(torch, n, g, ncol, 42 + (3 if c4 else 1) + 1000 * rank, device)
    aggs = [(0, pa.SUM), (0, pa.COUNT)] if c4 else [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
    key_cols = [(keys, None, pa.I64)]
    val_cols = [(v, None, pa.F64) for v in vals]
    ctx = pa.Context(local_rank)

    force_dist = os.environ.get("PANDRS_BENCH_FORCE_DIST") == "1" and dist is not None
    if world > 1 or force_dist:
        from pandrs_amd.dist import DistributedGroupBy
        if os.environ.get("PANDRS_BENCH_DIST", "library") != "torch":
            # the exchange runs INSIDE libpandrs_hip.so (pandrs_hip_dist_groupby_agg: RCCL behind the C ABI); torch.distributed
            # only carries the 128-byte communicator id from rank 0 to the others
            box = [pa.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.comm_init(box[0], rank, world)
        dgb = DistributedGroupBy(ctx, dist, device)

        def step():
            return dgb.groupby_agg(key_cols, n, val_cols, aggs, fetch=False)
    else:
        def step():
            return ctx.groupby_compute(key_cols, n, val_cols, aggs)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    phase_sum, total_kernel_ms, bytes_alg = {}, 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = dgb.last_timings if (world > 1 or force_dist) else ctx.timings()
        total_kernel_ms += t["total_ms"]
        bytes_alg = t["algorithmic_bytes"]
        for k, v in t["phase_ms"].items():
            phase_sum[k] = phase_sum.get(k, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        k = args.steps
        ms_per_step = dt / k * 1e3
        dev_ms = total_kernel_ms / k                    # hipEvent time of the pipeline on its stream
        achieved = bytes_alg / (dev_ms * 1e-3) / 1e9
        phases = {p: v / k for p, v in sorted(phase_sum.items())}
        out = {
            "metric": "Mrows/sec groupby-agg", "value": n * world / (dt / k) / 1e6, "unit": "Mrows/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("BASELINE config 4 (per-GPU shard): %d rows/GPU, 1 sparse i64 key (%d groups over the whole job), "
                                    "sum + count of %d f64 col" if c4 else
                                    "BASELINE config 2: %d rows/GPU, 1 sparse i64 key (%d groups), "
                                    "sum/mean/min/max over %d f64 cols") % (n, g, ncol),
                       "rows_per_gpu": n, "groups": g, "value_cols": ncol, "aggregates": len(aggs),
                       "parallelism": "row-range shards + 1 RCCL all-to-all of partial records inside the library (pandrs_hip_dist_groupby_agg)" if world > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(n, g, ncol)[0] if world == 1 else None,
                         "traffic_source": pmc_traffic(n, g, ncol)[1] if world == 1 else None,
                         "algorithmic_bytes": bytes_alg, "device_ms": dev_ms, "phase_ms": phases,
                         "note": "whole groupby pipeline (estimate+histogram+scan+scatter+aggregate) "
                                 "timed with hipEvents on the library stream; B = N(K+8C)+G(K+8A)"},
        }
        if "scatter" in phases and phases["scatter"] > 0:
            # the task's literal reading (dominant kernel only) beside the whole-pipeline figure above
            dom = max((p for p in phases if not p.startswith("merge_") and p != "exchange_wall"), key=lambda p: phases[p])
            out["roofline"]["dominant_kernel"] = {"phase": dom, "ms": phases[dom], "achieved": bytes_alg / (phases[dom] * 1e-3) / 1e9,
                                                  "frac": bytes_alg / (phases[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                  "note": "algorithmic bytes of the whole call / this kernel's hipEvent time"}
        if world > 1 or force_dist:
            out["roofline"]["wall_ms_last_step"] = dgb.last_timings.get("wall_ms")
            out["roofline"]["note"] = ("local partial aggregation + split + all-to-all + merge; device_ms = hipEvent time of the "
                                       "local and merge pipelines + wall time of the exchange; B = N(K+8C)+G(K+8A) per GPU")
        if not args.no_extras and world == 1 and not c4:
            del keys, vals, key_cols, val_cols
            torch.cuda.empty_cache()
            out.update(extra_configs(torch, pa, ctx, device))
        if not args.no_cpu_baseline and world == 1 and not c4:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, g, ncol, aggs)
            out["cpu_baseline_parallel"] = cpu_baseline_parallel(min(n, 2 * args.cpu_sample), g, ncol, aggs)
            out["cpu_baseline_typed"] = cpu_baseline_typed(min(n, 3 * args.cpu_sample), g, ncol)
            if not args.no_extras:
                out["cpu_baseline_join"] = cpu_baseline_join(10_000_000, 1_000_000)      # beside c5_* / join_indices_50m_5m
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def bench_join(args, torch, pa, dist, rank, local_rank, world, device):
    """BASELINE config 5 shape per GPU: probe rows (i64 key, f64 v) x build rows/10 (unique i64 key,
    i64 g in [0, 100 000)), inner join then groupby(g).sum(v) through the fused entry point.
    N > 1: build side all-gathered, probe side stays local (pandrs_amd/dist.py)."""
    nl, nr, g = args.rows, max(args.rows // 10, 1), 100_000
    gen = torch.Generator(device=device)
    gen.manual_seed(42 + 5 + 1000 * rank)
    mix = lambda ids: ids * (-7046029254386353131)          # odd multiplier: a bijection on i64
    base = rank * nr
    rkey = mix(torch.randperm(nr, device=device, generator=gen) + base)
    rgrp = torch.randint(0, g, (nr,), device=device, generator=gen, dtype=torch.int64)
    lkey = mix(torch.randint(0, nr * world, (nl,), device=device, generator=gen, dtype=torch.int64))
    lval = torch.randn(nl, device=device, generator=gen, dtype=torch.float64) * 10 + 100
    ctx = pa.Context(local_rank)
    cols = ((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
    if world > 1 or (os.environ.get("PANDRS_BENCH_FORCE_DIST") == "1" and dist is not None):
        from pandrs_amd.dist import DistributedJoinGroupBy
        if os.environ.get("PANDRS_BENCH_DIST", "library") != "torch":
            box = [pa.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.comm_init(box[0], rank, world)
        djg = DistributedJoinGroupBy(ctx, dist, device)
        step = lambda: djg.join_groupby_sum(*cols, strategy=args.join_strategy)
    else:
        djg = None
        step = lambda: ctx.join_groupby_sum(*cols)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    dev_ms, phases = 0.0, {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        if djg is None:
            t = ctx.timings()
            dev_ms += t["total_ms"]
            for k, v in t["phase_ms"].items():
                phases[k] = phases.get(k, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        k = args.steps
        n_groups = int(out[0].shape[1])
        bytes_alg = nl * 16 + nr * world * 16 + n_groups * 16      # fused join->groupby
        ms = dt / k * 1e3
        dms = dev_ms / k if djg is None else ms
        res = {"metric": "Mrows/sec hash-join (probe rows, fused join->groupby-sum)",
               "value": nl * world / (dt / k) / 1e6, "unit": "Mrows/s", "n_gpus": world, "steps": k,
               "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE config 5 shape: %d probe rows/GPU x %d build rows/GPU, "
                                      "inner join -> groupby(g in [0,100000)).sum(v)" % (nl, nr),
                          "parallelism": ("%s + local fused join + 1 all-to-all of partial sums" %
                                          {"allgather": "all-gather of the build side", "shuffle": "row shuffle of both sides by key owner",
                                           "auto": "all-gather of the build side or row shuffle by key owner (by size and world)"}[args.join_strategy])
                          if world > 1 else "1 GPU"},
               "roofline": {"bound": "hbm", "achieved": bytes_alg / (dms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": bytes_alg / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                            "algorithmic_bytes": bytes_alg, "device_ms": dms,
                            "phase_ms": {p: v / k for p, v in sorted(phases.items())}}}
        if djg is not None:
            res["roofline"]["wall_ms_last_step"] = djg.last_wall_ms
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_join(10_000_000, 1_000_000)
        emit(res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
