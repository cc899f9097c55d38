"""The PCIe-inclusive rate: BASELINE config 2 handed over as HOST buffers (numpy, pageable) through
PANDRS_HIP_MEM_HOST — what a shim that leaves the columns in host RAM would see.  Never the bench value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandrs_amd as pa
ctx = pa.Context(0)
n, g = 100_000_000, 1_000_000
rng = np.random.default_rng(1)
keys = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
vals = [rng.normal(100, 10, n) for _ in range(4)]
aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
for it in range(4):
    t0 = time.perf_counter()
    ng = ctx.groupby_compute([(keys, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs)
    dt = time.perf_counter() - t0
    t = ctx.timings()
    print("host-buffer C2: %.1f ms wall (%.2f Grows/s), stage_in %.1f ms, device pipeline %.2f ms, %d groups" % (
        dt * 1e3, n / dt / 1e9, t["phase_ms"].get("stage_in", 0), t["total_ms"] - t["phase_ms"].get("stage_in", 0), ng), flush=True)
