#!/usr/bin/env python3
"""C2 (100 M rows, 1 M groups, 4 f64 cols x sum/mean/min/max): sweep scatter workgroup size and
aggregated-columns-per-round; prints per-phase hipEvent times.  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
from bench import make_shard

n = int(os.environ.get("ROWS", 100_000_000)); g = int(os.environ.get("GROUPS", 1_000_000)); ncol = int(os.environ.get("COLS", 4))
keys, vals = make_shard(torch, n, g, ncol, 43, "cuda:0")
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
ctx = pa.Context(0)
for sct in [int(x) for x in os.environ.get("SCT", "1024").split(",")]:
    for spr in [int(x) for x in os.environ.get("SPR", "4,2").split(",")]:
        for load, gen in [(int(x.split(":")[0]), int(x.split(":")[1])) for x in os.environ.get("LOADGEN", "70:0,70:1,80:0,55:0").split(",")]:
          for shared in [int(x) for x in os.environ.get("SHARED", "1,0").split(",")]:
            ctx.set_option("scatter_threads", sct); ctx.set_option("src_per_round", spr); ctx.set_option("load_pct", load); ctx.set_option("generic_aggregate", gen); ctx.set_option("shared_cursors", shared)
            best = None
            for it in range(3):
                ctx.groupby_compute([(keys, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs)
                t = ctx.timings()
                if best is None or t["total_ms"] < best["total_ms"]: best = t
            print(json.dumps({"shared": shared, "sct": sct, "spr": spr, "load": load, "generic": gen, "P": best["n_partitions"], "T": best["table_slots"], "total_ms": round(best["total_ms"], 3),
                              **{k: round(v, 3) for k, v in best["phase_ms"].items()}, "retries": best["retries"]}), flush=True)
