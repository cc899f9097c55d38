import sys; sys.path.insert(0, "/root/repo")
import torch, pandrs_amd as pa
from oracle import oracle as O
c = pa.Context(0)
x = torch.randn(100_000_000, dtype=torch.float64, device="cuda:0")
ts = []
for _ in range(10):
    st = c.column_stats((x, None, O.F64), x.numel()); ts.append(c.timings()["phase_ms"]["other"])
print(["%.3f" % t for t in ts], "best %.2f TB/s" % (0.8 / min(ts)))
