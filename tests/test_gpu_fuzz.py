"""A slice of the randomised parity sweep inside the GPU test suite (the builder-run fuzz —
profiles/r0N_fuzz_summary.txt, thousands of cases — was not part of GPUTEST).  experiments/fuzz_parity.py draws shapes, dtypes,
null rates, skews, row layouts, aggregate sets, join types and engine knobs at random and compares the HIP engine (through the C
ABI) with the CPU oracle, bit for bit where the reference is; experiments/fuzz_absorb.py forces the hot-key absorb pass.  Fixed
seeds: the same cases every run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, env=None):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "experiments", script), *map(str, args)], capture_output=True, text=True,
                       timeout=280, env=e, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-40:])
    assert r.returncode == 0, tail
    return r.stdout


def test_fuzz_slice_default_mix():
    out = _run("fuzz_parity.py", 70, 4001)
    assert "fuzz done: 70 cases, 0 failures" in out


def test_fuzz_slice_joins_and_fused():
    out = _run("fuzz_parity.py", 40, 4002, env={"FUZZ_GROUPBY_FRAC": "0.0"})
    assert "fuzz done: 40 cases, 0 failures" in out


def test_fuzz_slice_absorb_forced():
    out = _run("fuzz_absorb.py", 25, 43)
    assert "0 failures" in out


def test_fuzz_slice_distributed_real_ranks():
    """experiments/fuzz_dist.py: the in-library exchange (dist.hip) with three REAL ranks on the one GPU over a host transport —
    random uneven / empty row ranges, null masks passed by random subsets of the ranks (the layout agreement that rides on the
    count exchange), host and device shards, mergeable and shuffled aggregate sets, one or two key columns; the owners' results,
    concatenated, equal the oracle on the whole frame."""
    out = _run("fuzz_dist.py", 3, 14, 5)
    assert "fuzz_dist done: world 3, 14 cases, 0 failing rank-cases" in out
