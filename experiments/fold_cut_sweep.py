#!/usr/bin/env python3
"""Cut threshold (slice_over) x fold thresholds (fold_min, fold_min_multi) over hot-key shapes at C2's shape.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
g, n, ncol = 1_000_000, 100_000_000, 4
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
ids = torch.randint(0, g, (n,), device=d, generator=gen)
variants = [("default", {}), ("fold 32", {"fold_min": 32}), ("fold 40", {"fold_min": 40}), ("fold 48", {"fold_min": 48}), ("fold 32, over 3", {"fold_min": 32, "slice_over": 3}),
            ("fold 32, over 4", {"fold_min": 32, "slice_over": 4}), ("fold 40, over 3", {"fold_min": 40, "slice_over": 3})]
if os.environ.get("SWEEP1"):
    variants = [("default 2/8/16", {}), ("over 3", {"slice_over": 3}), ("over 4", {"slice_over": 4}), ("multi 16", {"fold_min_multi": 16}), ("multi 32", {"fold_min_multi": 32}),
                ("fold 32", {"fold_min": 32}), ("no fold", {"fold_min": 65, "fold_min_multi": 65}), ("over 3, multi 16", {"slice_over": 3, "fold_min_multi": 16})]
print("%-28s" % "shape" + "".join("%18s" % nme for nme, _ in variants), flush=True)
for share, hotn in ((0.0, 1), (0.8, 2000), (0.5, 200), (0.5, 20), (0.5, 1), (0.9, 1), (0.8, 200_000), (0.95, 2000), (0.3, 50)):
    k = torch.where(torch.rand(n, device=d, generator=gen) < share, torch.randint(0, hotn, (n,), device=d, generator=gen), ids) * -7046029254386353131
    row = []
    for nme, opts in variants:
        for o, val in opts.items(): ctx.set_option(o, val)
        best = None
        for _ in range(3):
            ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs); t = ctx.timings()["total_ms"]
            best = t if best is None else min(best, t)
        for o in opts: ctx.set_option(o, 0)
        row.append("%18.2f" % best)
    print("%-28s" % ("%.0f %% on %d keys" % (share * 100, hotn)) + "".join(row), flush=True)
    del k
