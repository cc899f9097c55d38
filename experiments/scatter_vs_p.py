"""Scatter time against the fan-out P on the C2 columns (1 key + 4 f64 values, 100 M rows): is the
scatter bound by bytes or by the number of (tile, partition) cursor reservations (global atomics)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 100_000_000, int(os.environ.get("G", "1000000"))
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
for shared in (1, 0):
    ctx.set_option("shared_cursors", shared)
    for P in (32, 64, 128, 256, 512, 1152, 2304, 4608):
        ctx.set_option("partitions", P)
        best = None
        for _ in range(3):
            try:
                ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs)
            except pa.PandrsHipError as e:
                best = None; break
            t = ctx.timings()
            if best is None or t["phase_ms"]["scatter"] < best["phase_ms"]["scatter"]: best = t
        if best:
            print("shared_cursors=%d P=%5d (used %5d, retries %d)  scatter %.3f ms  aggregate %.3f ms" % (
                shared, P, best["n_partitions"], best["retries"], best["phase_ms"]["scatter"], best["phase_ms"].get("aggregate", 0)), flush=True)
