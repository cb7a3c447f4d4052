"""interval.py:610-612 (`dist.sort()`; `interp1d(linspace(0, 1, n), dist)`) for a sample whose shards stay on their
ranks: quantpy_amd.distributed.ShardedSample against np.sort + scipy's interp1d, world sizes 1-3 on gloo (CPU).
The NumPy form of the four selection steps runs here; tests/test_gpu_selection.py runs the HIP kernels on the same cases."""
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy.interpolate import interp1d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sample_cases():
    """name -> (sample, levels): sizes around the shard boundaries, ties, a NaN, the extreme levels."""
    rng = np.random.default_rng(2024)
    cases = {}
    few = np.array([0.5, 0.9, 0.95])
    cases["gamma_20000"] = (rng.gamma(3.0, 0.01, 20000), few)
    cases["uniform_20001_edges"] = (rng.random(20001), np.array([0.0, 1.0, 1e-9, 1 - 1e-12, 0.5]))
    cases["ties_30000"] = (np.round(rng.random(30000), 2), few)          # 300 copies of each value
    cases["heavy_ties_30000"] = (np.round(rng.random(30000), 1), few)    # 3000 copies: the windows overflow
    cases["mild_ties_30000"] = (np.round(rng.random(30000), 5), few)
    cases["sorted_input_8192"] = (np.sort(rng.standard_normal(8192)), few)  # rank r holds the r-th quantile range
    cases["reversed_9000"] = (np.sort(rng.standard_normal(9000))[::-1].copy(), few)
    z = rng.random(12000)
    z[::7] = 0.0                                                          # hs_dst returns exact zeros below 1e-15
    cases["zeros_12000"] = (z, np.array([0.05, 0.14, 0.15, 0.9]))
    w = rng.random(10000)
    w[123] = np.nan                                                       # a failed trial: np.sort puts it last
    cases["nan_10000"] = (w, few)
    cases["small_24"] = (rng.random(24), few)                             # gather path (selection would move more bytes)
    cases["tiny_3"] = (rng.random(3), few)
    cases["one_value"] = (np.array([0.25]), few)
    cases["many_levels_5000"] = (rng.random(5000), np.linspace(1e-3, 1 - 1e-3, 1000))  # the reference's default levels
    return cases


def reference_quantiles(sample, levels):
    srt = np.sort(sample)
    if len(srt) == 1:
        return np.full(len(levels), srt[0])
    return interp1d(np.linspace(0, 1, len(srt)), srt)(levels)


def test_single_rank_is_sort_plus_interp1d():
    from quantpy_amd.distributed import ShardedSample, interp_cell

    for name, (x, lv) in sample_cases().items():
        got = ShardedSample(x.copy(), len(x)).quantiles(lv)
        assert np.array_equal(got, reference_quantiles(x, lv), equal_nan=True), name
    # the cell walk equals searchsorted on the actual grid, also where q * (n - 1) rounds across a grid point
    for n in (2, 3, 7, 1000, 2000, 2097152):
        grid = np.linspace(0, 1, n)
        qs = np.concatenate([grid[:: max(1, n // 257)], np.nextafter(grid[:: max(1, n // 263)], 2), [0.0, 1.0, 0.3, 0.999999]])
        for q in qs[(qs >= 0) & (qs <= 1)]:
            hi = min(max(np.searchsorted(grid, q, "left"), 1), n - 1)
            lo, xl, xh = interp_cell(n, float(q))
            assert (lo, xl, xh) == (hi - 1, grid[hi - 1], grid[hi]), (n, q)


def test_selection_plan_moves_less_than_the_sample():
    from quantpy_amd.distributed import selection_plan

    for n_total, ws, nl in ((2097152, 8, 3), (2097152, 2, 3), (2000, 8, 3), (20000, 3, 5), (10**7, 8, 1000)):
        plan = selection_plan(n_total, ws, nl)
        if plan is None:
            continue
        stride, p, width = plan
        n_max = -(-n_total // ws)
        assert p * stride >= n_max and (p - 1) * stride < n_max
        assert width >= min((2 * ws + 3) * stride, n_max)
        assert ws * (p + nl * (2 + width)) < n_total
    assert selection_plan(2097152, 8, 3) is not None and selection_plan(24, 2, 3) is None
    s, p, w = selection_plan(2097152, 8, 3)
    assert 8 * 8 * (p + 3 * (2 + w)) < 0.05 * 8 * 2097152  # < 5 % of the bytes of gathering the sample


_WORKER = r'''
import sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
from quantpy_amd import distributed as qd
from test_sharded_quantiles import sample_cases, reference_quantiles
dist.init_process_group("gloo")
rank, ws = qd.world()
paths = {}
for name, (x, lv) in sample_cases().items():
    lo, hi = qd.shard_bounds(len(x))
    smp = qd.ShardedSample(x[lo:hi].copy(), len(x))
    got = smp.quantiles(lv)
    want = reference_quantiles(x, lv)
    assert np.array_equal(got, want, equal_nan=True), (rank, name, got, want)
    paths[name] = smp.last_path
    full = smp.gather_sorted()
    assert np.array_equal(full, np.sort(x), equal_nan=True), (rank, name)
    assert np.array_equal(smp.quantiles(lv), want, equal_nan=True)       # (now from the gathered sample)
assert paths["gamma_20000"] == "selection" and paths["small_24"] == "gather", paths
assert paths["heavy_ties_30000"] == "gather" and paths["zeros_12000"] == "gather", paths   # windows overflow on massive ties
assert paths["mild_ties_30000"] == "selection" and paths["nan_10000"] == "selection", paths
assert paths["many_levels_5000"] == "gather", paths
print(f"rank {rank}/{ws} ok", flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world_size", [2, 3, 4])
def test_sharded_quantiles_equal_sort_plus_interp1d_gloo(tmp_path, world_size):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world_size}",
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + world_size), str(script), ROOT]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for r in range(world_size):
        assert f"rank {r}/{world_size} ok" in res.stdout
