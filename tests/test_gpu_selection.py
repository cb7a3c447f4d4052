"""GPU: round-3 pieces of the bootstrap path (interval.py:598-612).

* qt_lin_dist_batch / qt_mle_dist_batch: reconstruct + Hilbert-Schmidt distance in ONE pass (rho nullable) against the
  two-pass form (qt_*_batch + qt_hs_dist_batch) and against the oracle, n = 1 ... 5, fused and split MLE kernels;
* the distributed-selection kernels (qt_select_*) and qt_merge_sorted against np.sort + scipy's interp1d on the cases of
  tests/test_sharded_quantiles.py, with the ranks simulated on one GPU and as two gloo ranks sharing cuda:0;
* the engine follows torch's current stream."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_sharded_quantiles import reference_quantiles, sample_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qp():
    import quantpy_amd

    return quantpy_amd


def ginibre(rng, d, rank=None):
    r = d if rank is None else rank
    g = rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r))
    rho = g @ g.conj().T
    return rho / np.trace(rho)


@pytest.mark.parametrize("n,shots,batch", [(1, 200, 37), (2, 1000, 50), (3, 1000, 70), (3, 100000, 1500), (4, 100000, 12), (5, 1000000, 6)])
def test_fused_distance_equals_two_pass(qp, oracle, n, shots, batch):
    import torch

    from quantpy_amd.tomography.state import simulate_counts

    d = 2**n
    rng = np.random.default_rng(10 + n)
    rho = ginibre(rng, d, rank=None if shots >= 1000 else 1)
    povm = qp.generate_measurement_matrix("proj-set", n)
    ns = np.ones(povm.shape[0]) * shots
    np.random.seed(100 + n)
    counts = simulate_counts(povm, qp.Qobj(rho).bloch, ns, repeats=batch)
    eng = qp.get_engine(n)
    eng.set_povm(povm, ns)
    centre = eng.mle(counts[0])
    for init in ("lin", "mixed"):
        if init == "mixed" and n >= 4 and batch > 8:
            counts_i = counts[:4]
        else:
            counts_i = counts
        two_pass_rho, info = eng.mle(counts_i, init=init, return_info=True)
        two_pass = eng.hs_dist(two_pass_rho, centre)
        got, ginfo = eng.mle_dist(counts_i, centre, init=init, return_info=True)
        assert np.array_equal(ginfo["nit"], info["nit"])
        assert np.abs(got - two_pass).max() <= 1e-15 + 4e-16 * np.abs(two_pass).max(), (n, init)
        assert got[0 if init == "lin" else -1] >= 0
        if init == "lin":
            assert got[0] == 0.0  # the centre's own counts: hs_dst returns an exact 0 below 1e-15
        for i in range(min(3, len(counts_i))):  # the oracle's hs_dst on the two-pass matrices (geometry.py:16-20)
            assert abs(got[i] - oracle.hs_dst(two_pass_rho[i], centre)) < 1e-13
    # hs_dst is defined for any pair of matrices (geometry.py:16: sqrt(|Tr((A - B)^2)|) / sqrt(2), no conjugation): a centre
    # that is not Hermitian goes through the same formula
    odd = centre + 0.05 * (rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))
    got_odd = eng.mle_dist(counts[:3], odd)
    for i in range(3):
        assert abs(got_odd[i] - oracle.hs_dst(eng.mle(counts[i]), odd)) < 1e-13
    for physical in (True, False):
        lin_rho = eng.lin(counts, physical=physical)
        assert np.abs(eng.lin_dist(counts, centre, physical=physical) - eng.hs_dist(lin_rho, centre)).max() < 1e-15
    # device pointers: rho optional; with rho the matrices equal the plain call's bit for bit
    cd = torch.from_numpy(counts).cuda()
    cen = torch.from_numpy(np.ascontiguousarray(centre)).cuda()
    dist_d = torch.full((batch,), -1.0, dtype=torch.float64, device="cuda")
    rho_d = torch.zeros((batch, d, d), dtype=torch.complex128, device="cuda")
    st = torch.full((batch,), -1, dtype=torch.int32, device="cuda")
    eng.mle_dist_dev(cd, cen, dist_d, rho=rho_d, status=st)
    eng.sync()
    plain = eng.mle(counts)
    assert np.array_equal(rho_d.cpu().numpy(), plain) and int(st.abs().sum()) == 0
    assert np.array_equal(dist_d.cpu().numpy(), eng.mle_dist(counts, centre))
    dist2 = torch.empty_like(dist_d)
    eng.lin_dist_dev(cd, cen, dist2)
    eng.sync()
    assert np.array_equal(dist2.cpu().numpy(), eng.lin_dist(counts, centre))
    # the split (start + BFGS) kernels write the same distances as the one-launch kernel
    if n <= 3:
        from quantpy_amd import _capi

        eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, 0)
        try:
            split = eng.mle_dist(counts, centre, init="mixed")
        finally:
            eng.set_option(_capi.QT_OPT_MLE_FUSED_MAX_WAVES, 1024)
        fused = eng.mle_dist(counts, centre, init="mixed")
        assert np.abs(split - fused).max() < 1e-9  # (two BFGS forms: same iterates to rounding)


def test_fused_distance_argument_errors(qp):
    eng = qp.get_engine(1)
    povm = qp.generate_measurement_matrix("proj-set", 1)
    eng.set_povm(povm, np.ones(3) * 10)
    c = np.array([[[5, 5], [5, 5], [10, 0]]], dtype=np.int64)
    dist = np.empty(1)
    from quantpy_amd.engine import _ptr

    assert eng.lib.qt_mle_dist_batch(eng._h, _ptr(c), 1, 0, 100, 1e-3, None, None, _ptr(dist), None, None, None, None, 0) < 0
    assert eng.lib.qt_lin_dist_batch(eng._h, _ptr(c), 1, 1, None, None, None, None, 0) < 0
    assert eng.lib.qt_mle_batch(eng._h, _ptr(c), 1, 0, 100, 1e-3, None, None, None, None, None, 0) < 0


def simulated_rank_quantiles(eng, x, levels, n_ranks, plan):
    """The four selection steps with the ranks of a process group played one after the other on one GPU."""
    import torch

    from quantpy_amd import distributed as qd

    n = len(x)
    stride, n_split, width = plan
    bounds = [qd.shard_bounds(n, r, n_ranks) for r in range(n_ranks)]
    shards = []
    for lo, hi in bounds:
        t = torch.from_numpy(x[lo:hi].copy()).cuda()
        shards.append(eng.sort_dev(t) if hi - lo > 1 else t)
    spl = torch.empty((n_ranks, n_split), dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        eng.select_splitters(s, stride, n_split, spl[r])
    sizes = torch.tensor([hi - lo for lo, hi in bounds], dtype=torch.int64, device="cuda")
    q = torch.from_numpy(np.ascontiguousarray(levels)).cuda()
    nl = len(levels)
    lo_k = torch.empty(nl, dtype=torch.int64, device="cuda")
    hi_k = torch.empty(nl, dtype=torch.int64, device="cuda")
    eng.select_bracket(spl, sizes, stride, n, q, lo_k, hi_k)
    win = torch.empty((n_ranks, nl, 2 + width), dtype=torch.float64, device="cuda")
    for r, s in enumerate(shards):
        eng.select_window(s, lo_k, hi_k, width, win[r])
    out = torch.empty(nl, dtype=torch.float64, device="cuda")
    flag = torch.zeros(2, dtype=torch.int32, device="cuda")
    eng.select_finish(win, n, q, out, flag)
    eng.sync()
    return out.cpu().numpy(), int(flag[0].item()), shards


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_selection_kernels_equal_sort_plus_interp1d(qp, n_ranks):
    import torch

    from quantpy_amd import distributed as qd

    eng = qp.get_engine(1)
    for name, (x, lv) in sample_cases().items():
        want = reference_quantiles(x, lv)
        plan = qd.selection_plan(len(x), n_ranks, len(lv))
        if plan is None:  # sizes at which the product gathers instead: the kernels must still be right (or raise the flag)
            n_max = -(-len(x) // n_ranks)
            stride = max(1, n_max // 16)
            plan = (stride, -(-n_max // stride), min((2 * n_ranks + 3) * stride, n_max))
        got, flag, shards = simulated_rank_quantiles(eng, x, lv, n_ranks, plan)
        if flag == 0:
            assert np.array_equal(got, want, equal_nan=True), (name, n_ranks, got, want)
        else:
            assert name in ("heavy_ties_30000", "ties_30000", "zeros_12000", "small_24", "tiny_3", "one_value", "many_levels_5000",
                            "mild_ties_30000"), (name, flag)
        # the gather path: merge of the sorted shards = np.sort
        runs = torch.cat(shards)
        merged = eng.merge_sorted(runs, [s.numel() for s in shards])
        eng.sync()
        assert np.array_equal(merged.cpu().numpy(), np.sort(x), equal_nan=True), (name, n_ranks)
    # a case the selection must settle itself (no overflow), at the size of the bench's large leg
    rng = np.random.default_rng(5)
    x = rng.gamma(3.0, 0.01, 2097152)
    lv = np.array([0.5, 0.9, 0.95])
    plan = qd.selection_plan(len(x), max(n_ranks, 2), 3)
    got, flag, _ = simulated_rank_quantiles(eng, x, lv, max(n_ranks, 2), plan)
    assert flag == 0 and np.array_equal(got, reference_quantiles(x, lv))


def test_merge_sorted_host_pointers_and_odd_run_counts(qp):
    eng = qp.get_engine(1)
    rng = np.random.default_rng(9)
    for lengths in ([5], [0, 7], [3, 0, 0, 9, 1], [1000, 1, 999, 37, 4096], [8192] * 7):
        runs = [np.sort(rng.standard_normal(m)) for m in lengths]
        got = eng.merge_sorted(np.concatenate(runs) if sum(lengths) else np.empty(0), lengths)
        assert np.array_equal(got, np.sort(np.concatenate(runs)))


_WORKER = r'''
import sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
import quantpy_amd as qp
from quantpy_amd import distributed as qd
from test_sharded_quantiles import sample_cases, reference_quantiles
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, ws = qd.world()
eng = qp.get_engine(1, 0)
paths = {}
for name, (x, lv) in sample_cases().items():
    lo, hi = qd.shard_bounds(len(x))
    smp = qd.ShardedSample(torch.from_numpy(x[lo:hi].copy()).cuda(), len(x), engine=eng)
    got = smp.quantiles(lv)
    want = reference_quantiles(x, lv)
    assert np.array_equal(got, want, equal_nan=True), (rank, name, got, want)
    paths[name] = smp.last_path
    full = smp.gather_sorted()
    eng.sync()
    assert np.array_equal(full.cpu().numpy(), np.sort(x), equal_nan=True), (rank, name)
assert paths["gamma_20000"] == "selection" and paths["small_24"] == "gather" and paths["heavy_ties_30000"] == "gather", paths
print(f"rank {rank}/{ws} ok", flush=True)
dist.destroy_process_group()
'''


def test_sharded_sample_on_device_two_gloo_ranks_sharing_gpu0(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29731", str(script), ROOT]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "rank 0/2 ok" in res.stdout and "rank 1/2 ok" in res.stdout


def test_engine_follows_torchs_current_stream(qp):
    """the handle was bound to torch's stream once.  Inputs produced on a side stream, a device-pointer
    call under `torch.cuda.stream(side)` and no synchronisation in between: the call must be enqueued on `side`."""
    import torch

    from quantpy_amd.tomography.state import simulate_counts

    n = 3
    eng = qp.get_engine(n)
    eng.follow_torch_stream()
    povm = qp.generate_measurement_matrix("proj-set", n)
    ns = np.ones(27) * 1000
    eng.set_povm(povm, ns)
    rng = np.random.default_rng(1)
    np.random.seed(1)
    counts = simulate_counts(povm, qp.Qobj(ginibre(rng, 8)).bloch, ns, repeats=4096)
    want = eng.lin(counts)
    host = torch.from_numpy(counts).pin_memory()
    rho_d = torch.zeros((4096, 8, 8), dtype=torch.complex128, device="cuda")
    eng.lin_dev(torch.from_numpy(counts[:8]).cuda(), rho_d[:8])  # binds to the default stream first
    side = torch.cuda.Stream()
    big = torch.empty((64 << 20,), dtype=torch.float64, device="cuda")
    for _ in range(3):
        with torch.cuda.stream(side):
            big.normal_()                              # ~10 ms of work in front of the copy on `side`
            cd = host.to("cuda", non_blocking=True)    # the producer of the counts: asynchronous, on `side`
            eng.lin_dev(cd, rho_d)                     # must queue behind it
            ptr_side = eng._bound_ptr
        side.synchronize()
        assert ptr_side == side.cuda_stream
        assert np.array_equal(rho_d.cpu().numpy(), want)
        rho_d.zero_()
        eng.lin_dev(cd, rho_d)                          # back on the default stream
        assert eng._bound_ptr != side.cuda_stream
        torch.cuda.synchronize()
        assert np.array_equal(rho_d.cpu().numpy(), want)
        rho_d.zero_()


_BOOT_WORKER = r'''
import json, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import quantpy_amd as qp
from quantpy_amd import distributed as qd
torch.cuda.set_device(0)
world = int(sys.argv[3])
if world > 1:
    dist.init_process_group("gloo")
rank, ws = qd.world()
rng = np.random.default_rng(8)
g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
rho = g @ g.conj().T
rho /= np.trace(rho)
out = {}
levels = np.array([0.05, 0.5, 0.9, 0.95])
for sampler, method, n_points in (("numpy", "mle", 301), ("device", "mle", 4001), ("device", "lin", 30000)):
    np.random.seed(21)
    tmg = qp.StateTomograph(qp.Qobj(rho))
    tmg.experiment(1000, "proj-set")
    tmg.point_estimate("mle")
    iv = qp.BootstrapStateInterval(tmg, n_points=n_points, method=method, sampler=sampler, seed=None if sampler == "numpy" else 77)
    d, cl = iv(levels)
    key = f"{sampler}-{method}-{n_points}"
    out[key] = {"q": [float(x) for x in d], "path": iv.sample.last_path,
                "shard": int(iv.sample.local.shape[0]),
                "first": [float(x) for x in iv.boot_dist[:5]], "n": int(len(iv.boot_dist)),
                "default_levels": [float(x) for x in iv()[0][::250]],
                "sorted_ok": bool(np.array_equal(iv.cl_to_dist.y, np.sort(iv.boot_dist)))}
if rank == 0:
    json.dump(out, open(sys.argv[2], "w"))
if world > 1:
    dist.destroy_process_group()
'''


def test_bootstrap_interval_shards_over_ranks_with_identical_quantiles(tmp_path):
    """sampler='device' draws only the rank's shard, distances never leave the rank unsorted, the
    quantiles come from the distributed selection -- and equal the one-rank run's bit for bit, for both samplers."""
    import json

    script = tmp_path / "worker.py"
    script.write_text(_BOOT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one = tmp_path / "one.json"
    res = subprocess.run([sys.executable, str(script), ROOT, str(one), "1"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    two = tmp_path / "two.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", str(script), ROOT, str(two), "2"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    a, b = json.load(open(one)), json.load(open(two))
    for key in a:
        assert a[key]["q"] == b[key]["q"] and a[key]["first"] == b[key]["first"], (key, a[key], b[key])
        assert a[key]["default_levels"] == b[key]["default_levels"], key
        assert a[key]["n"] == b[key]["n"] and a[key]["sorted_ok"] and b[key]["sorted_ok"]
        assert a[key]["path"] == "local" and b[key]["shard"] == -(-a[key]["n"] // 2)
    assert b["device-lin-30000"]["path"] == "selection"


_RCCL_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from quantpy_amd import distributed as qd
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29751")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and qd.world() == (0, 1)
# the exchanges of ShardedSample / BootstrapStateInterval as RCCL issues them, on a one-rank group
spl = torch.arange(3856, dtype=torch.float64, device="cuda") * 0.5
got = qd.allgather_equal(spl, force_collective=True)
assert got.shape == (1, 3856) and got.is_cuda and torch.equal(got[0], spl)
win = torch.rand((3, 2 + 1292), dtype=torch.float64, device="cuda")
got = qd.allgather_equal(win, force_collective=True)
assert got.shape == (1, 3, 1294) and torch.equal(got[0], win)
keys = np.array([0, 2**63 - 1, -1, 12345], dtype=np.int64)             # bit patterns travel as int64
got = qd.allgather_equal(keys, force_collective=True)
assert isinstance(got, np.ndarray) and got.dtype == np.int64 and np.array_equal(got[0], keys)
bad = qd.allgather_equal(np.array([0, 1]), force_collective=True).max(axis=0)
assert bad.tolist() == [0, 1]
seed = np.array([0xDEADBEEFCAFEF00D], dtype=np.uint64)
assert int(qd.broadcast_array(seed.view(np.int64)).view(np.uint64)[0]) == 0xDEADBEEFCAFEF00D
full = qd.allgather_device(torch.arange(5, dtype=torch.float64, device="cuda"), 5)
assert full.shape == (5,)
torch.cuda.synchronize()
print("rccl ok", flush=True)
dist.destroy_process_group()
'''


def test_the_collectives_of_the_sharded_path_on_a_one_rank_rccl_group(tmp_path):
    """No multi-GPU box here: the RCCL calls the N > 1 path makes (all_gather_into_tensor of float64 / int64 device tensors
    on torch's current stream, NumPy payloads through the device) are at least issued for real, on a one-rank nccl group."""
    script = tmp_path / "worker.py"
    script.write_text(_RCCL_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    res = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "rccl ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_quantile_kernels_on_grid_points_and_random_samples(qp):
    """numpy.interp's semantics on the device (what interp1d calls for real 1-D data): levels that ARE grid points return the
    order statistic itself (round 2's kernel was one ulp off in a third of such cases: 0.5 with n = 2001); and a random
    sweep of the selection kernels (sizes, rank counts, tie structures, levels on and between grid points) against
    np.sort + interp1d -- bit for bit whenever the kernels do not raise their overflow flag."""
    import torch
    from scipy.interpolate import interp1d

    from quantpy_amd import distributed as qd

    eng = qp.get_engine(1)
    rng = np.random.default_rng(77)
    for n in (2, 5, 102, 2001, 8192, 20001):
        y = rng.gamma(2.0, 0.01, n)
        grid = np.linspace(0, 1, n)
        qs = np.unique(np.clip(np.concatenate([grid[:: max(1, n // 400)], np.nextafter(grid[:: max(1, n // 300)], 2),
                                               np.nextafter(grid[:: max(1, n // 300)], -1), [0.0, 0.5, 1.0]]), 0, 1))
        srt, got = eng.sort_quantiles(y, qs)
        assert np.array_equal(srt, np.sort(y)) and np.array_equal(got, interp1d(grid, np.sort(y))(qs)), n
    declined = 0
    for case in range(160):
        n_ranks, n = int(rng.integers(1, 10)), int(rng.integers(1, 4000))
        kind = ("cont", "r3", "r1", "zeros", "nan", "const")[case % 6]
        x = rng.gamma(2.0, 0.01, n)
        if kind == "r3":
            x = np.round(x, 3)
        elif kind == "r1":
            x = np.round(x, 1)
        elif kind == "zeros":
            x[rng.random(n) < 0.3] = 0.0
        elif kind == "nan":
            x[rng.integers(0, n)] = np.nan
        elif kind == "const":
            x[:] = 0.25
        grid = np.linspace(0, 1, n) if n > 1 else np.array([0.0])
        levels = np.concatenate([rng.random(3), grid[rng.integers(0, len(grid), 2)], [0.0, 1.0]])
        plan = qd.selection_plan(n, n_ranks, len(levels)) if case % 2 else None
        if plan is None:
            n_max = -(-n // n_ranks)
            stride = int(rng.integers(1, max(2, n_max // 2 + 1)))
            plan = (stride, -(-n_max // stride), int(rng.integers(1, min(3 * n_max + 2, 1500))))
        got, flag, _ = simulated_rank_quantiles(eng, x, levels, n_ranks, plan)
        if flag:
            declined += 1
            continue
        assert np.array_equal(got, reference_quantiles(x, levels), equal_nan=True), (case, n_ranks, n, kind, plan)
    assert declined < 100
