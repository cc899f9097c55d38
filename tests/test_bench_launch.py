"""`python bench.py --gpus N` must start its own ranks (VERDICT r1 / ADVICE): rehearsed here on CPU with
the gloo backend and the numpy stand-in engine (PANDRS_BENCH_BACKEND=gloo); the record it relays is marked
as a dry run, never as a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=240)


def test_plain_invocation_with_two_gpus_spawns_ranks():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "20000", "--groups", "500", "--cols", "2"],
             {"PANDRS_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and "dry_run" in rec
    assert rec["value"] > 0


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "2"], {"PANDRS_BENCH_BACKEND": "gloo", "RANK": "0", "WORLD_SIZE": "3"})
    assert p.returncode != 0
    assert b"WORLD_SIZE" in p.stderr
