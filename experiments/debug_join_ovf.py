import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pandrs_amd as pa
d = "cuda:0"
nr, nl = 8_000_000, 17_000_000
rng = np.random.default_rng(31)
rk = (rng.permutation(nr).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
hot = rng.random(nr) < 0.6
rg = np.where(hot, rng.integers(0, 16, nr), 1_000 + np.arange(nr)).astype(np.int64)
pick = rng.integers(0, nr, nl)
lk = rk[pick]
lv = rng.integers(-8, 9, nl).astype(np.float64)
dev = lambda a: torch.from_numpy(a).to(d)
c = pa.Context(0)
groups = np.unique(rg[pick])
for no_chao, novf in ((0, 0), (1, 0), (1, 1)):
    c.set_option("no_chao", no_chao); c.set_option("no_overflow_run", novf)
    kc, kn, oa = c.join_groupby_sum((dev(lk), None, pa.I64), (dev(lv), None, pa.F64), nl, (dev(rk), None, pa.I64), (dev(rg), None, pa.I64), nr)
    t = c.timings()
    print(no_chao, novf, "groups", kc.shape, "want", len(groups), "retries", t["retries"], "est", t["estimated_groups"], "P", t["n_partitions"], flush=True)
