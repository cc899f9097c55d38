#!/usr/bin/env python3
"""ONE benchmark configuration, a few calls, per-phase hipEvent times — the target of the per-config rocprofv3 runs
(experiments/profile_configs.sh).  usage: one_config.py {north_star|c3|c4_shard|c4_one_gpu|c5_shard|c5_one_gpu|join_indices} [reps]
Same shapes and seeds as bench.py's extra_configs.  GPU box only."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa

cfg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = "cuda:0"
ctx = pa.Context(0)
MIX = -7046029254386353131
gen = torch.Generator(device=d)
gen.manual_seed(4242)
if cfg == "north_star":
    n, g = 100_000_000, 1_000_000
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * MIX ^ 0x5555AAAA5555AAAA
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100
    fn = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)])
elif cfg == "c3":
    n, g = 100_000_000, 10_000
    hot = torch.rand(n, device=d, generator=gen) < 0.8
    codes = torch.where(hot, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)).to(torch.int32)
    del hot
    v = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(2)]
    aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)] + [(0, pa.COUNT)]
    fn = lambda: ctx.groupby_compute([(codes, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs)
elif cfg in ("c4_shard", "c4_one_gpu"):
    n, g = (125_000_000, 10_000_000) if cfg == "c4_shard" else (1_000_000_000, 10_000_000)
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    k.mul_(MIX)
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    fn = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.COUNT)])
elif cfg in ("c5_shard", "c5_one_gpu"):
    n, nr, g = (62_500_000 if cfg == "c5_shard" else 500_000_000), 50_000_000, 100_000
    rkey = torch.randperm(nr, device=d, generator=gen) * MIX
    rgrp = torch.randint(0, g, (nr,), device=d, generator=gen, dtype=torch.int64)
    lkey = torch.randint(0, nr, (n,), device=d, generator=gen, dtype=torch.int64) * MIX
    lval = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    fn = lambda: ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), n, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
elif cfg == "join_indices":
    n, nr = 50_000_000, 5_000_000
    rkey = torch.randperm(nr, device=d, generator=gen) * MIX
    lkey = torch.randint(0, nr, (n,), device=d, generator=gen, dtype=torch.int64) * MIX
    fn = lambda: ctx.join_indices_compute((lkey, None, pa.I64), n, (rkey, None, pa.I64), nr, pa.INNER)
else:
    raise SystemExit("unknown config " + cfg)
for kv in os.environ.get("PANDRS_OPTS", "").split(","):          # PANDRS_OPTS=name=value,... (A/B of one option on one box)
    if kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
torch.cuda.synchronize()
fn()
best = None
for _ in range(reps):
    fn()
    t = ctx.timings()
    if best is None or t["total_ms"] < best["total_ms"]:
        best = t
print(json.dumps({"cfg": cfg, "rows": n, "calls": reps + 1, "best_ms": round(best["total_ms"], 4), "P": best["n_partitions"],
                  "absorbed_rows": best.get("absorbed_rows", 0), "phase_ms": {k: round(x, 4) for k, x in best["phase_ms"].items()}}), flush=True)
ctx.close()
