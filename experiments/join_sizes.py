import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
ctx = pa.Context(0); d = "cuda:0"
if os.environ.get("ONE_PASS"): ctx.set_option("join_one_pass", 1)
import sys as _s
CASES = [(5_000_000, 50_000_000), (30_000_000, 60_000_000), (1_000_000, 100_000_000)]
if len(_s.argv) > 1 and _s.argv[1] == 'big':      # beyond one LDS partition per build bucket: general segmented sort
    CASES = [(100_000_000, 100_000_000), (200_000_000, 50_000_000)]
for nb, npb in CASES:
    rk = torch.randperm(nb * 2, device=d)[:nb].to(torch.int64) * -7046029254386353131
    pick = torch.randint(0, nb, (npb,), device=d)
    lk = rk[pick]
    best = None
    for it in range(3):
        li, ri = ctx.join_indices((lk, None, pa.I64), npb, (rk, None, pa.I64), nb, pa.INNER)
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    assert li.numel() == npb and bool((ri == pick).all())
    print("join %dM x %dM: %.2f ms  P %d  %s" % (npb // 10**6, nb // 10**6, best["total_ms"], best["n_partitions"], {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
    del rk, pick, lk, li, ri
