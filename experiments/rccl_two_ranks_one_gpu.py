#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE device?  (The builder's box has one GPU; RCCL has only ever run at world 1 here.)
Two processes, each a pandrs context on cuda:0, pandrs_hip_comm_init with world = 2.  Prints what happens.  GPU box only;
run under `timeout`: a refused or hung init must not hold the box."""
import os, sys, socket, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def worker(rank, world, port, uid_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time, torch, pandrs_amd as pa
    ctx = pa.Context(0)
    if rank == 0:
        uid = pa.Context.comm_unique_id()
        open(uid_path + ".tmp", "wb").write(bytes(uid)); os.rename(uid_path + ".tmp", uid_path)
    else:
        while not os.path.exists(uid_path): time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    try:
        ctx.comm_init(uid, rank, world)
        print("rank %d: comm_init world %d on one device OK" % (rank, world), flush=True)
        rng = np.random.default_rng(5 + rank)
        n = 400_000
        k = torch.from_numpy((rng.integers(0, 30_000, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)).cuda()
        v = torch.from_numpy(rng.normal(size=n)).cuda()
        ng = ctx.dist_groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.COUNT)])
        print("rank %d: dist groupby over RCCL -> %d groups owned" % (rank, ng), flush=True)
    except Exception as e:
        print("rank %d: FAILED: %s" % (rank, str(e)[:300]), flush=True)
    ctx.close()

if __name__ == "__main__":
    import torch.multiprocessing as mp, tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = tempfile.mkdtemp()
    mp.spawn(worker, args=(2, port, os.path.join(d, "uid")), nprocs=2, join=True)
