"""Sharding of independent tomography trials over the GPUs of a node.

Trials / bootstrap resamples share only read-only operands, so each rank reconstructs a
contiguous slice of the batch on its own GPU and the only exchange is one all-gather of the
per-trial results (float64 distances: 16 KB for 2000 resamples) -- RCCL over xGMI when the
process group is `nccl`, gloo in the CPU tests.  One process per GPU (torchrun); without an
initialised process group everything degenerates to a single shard.
"""
import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover - torch is part of the image
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
    """(rank, world_size) of the default process group, (0, 1) when none is initialised."""
    dist = _dist()
    return (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)


def shard_bounds(n_items, rank=None, world_size=None):
    """Contiguous, balanced [lo, hi) slice of range(n_items) owned by `rank`; the first
    n_items % world_size ranks get one extra item.  Empty slices are legal."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_array(array, src=0):
    """Make every rank hold rank `src`'s ndarray (e.g. the counts drawn from rank 0's RNG stream)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return array
    import torch

    dev = _comm_device()
    t = torch.from_numpy(np.ascontiguousarray(array)).to(dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def _comm_device():
    import torch

    dist = _dist()
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allgather_concat(local, n_total):
    """Concatenate the ranks' 1-D float64 shards (in rank order) into the full length-`n_total`
    vector on every rank.  Shards may differ in length by one; they are padded to the largest."""
    dist = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or dist.get_world_size() == 1:
        return local
    import torch

    ws = dist.get_world_size()
    sizes = [shard_bounds(n_total, r, ws) for r in range(ws)]
    width = max(hi - lo for lo, hi in sizes)
    dev = _comm_device()
    buf = torch.zeros(width, dtype=torch.float64, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(local).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(gathered, buf)
    parts = [g[: hi - lo].cpu().numpy() for g, (lo, hi) in zip(gathered, sizes)]
    return np.concatenate(parts)


def allgather_device(local, n_total):
    """Device-tensor form of `allgather_concat`: `local` is this rank's 1-D float64 torch tensor (its
    `shard_bounds(n_total)` slice); returns the full length-`n_total` tensor on the same device on every rank.
    With the `nccl` group this is ONE RCCL all-gather over xGMI, enqueued behind the producer of `local` on
    torch's current stream (the engine's device-pointer calls run there), and nothing goes through the host;
    with `gloo` the shard takes the host path."""
    import torch

    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return local
    ws = dist.get_world_size()
    sizes = [shard_bounds(n_total, r, ws) for r in range(ws)]
    width = max(hi - lo for lo, hi in sizes)
    if dist.get_backend() != "nccl":
        full = allgather_concat(local.detach().cpu().numpy(), n_total)
        return torch.from_numpy(full).to(local.device)
    if local.shape[0] == width:
        buf = local.contiguous()
    else:
        buf = torch.zeros(width, dtype=torch.float64, device=local.device)
        buf[: local.shape[0]] = local
    out = torch.empty(ws * width, dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    if all(hi - lo == width for lo, hi in sizes):
        return out
    return torch.cat([out[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])


def allgather_equal(x, force_collective=False):
    """Stack the ranks' equally shaped tensors / arrays along a new first axis: (N, *x.shape) on every rank.  torch CUDA
    tensors travel as ONE RCCL all-gather on torch's current stream (`nccl` group) or through the host (`gloo`);
    NumPy arrays come back as NumPy arrays.  A one-rank group returns `x[None]` without a collective unless
    `force_collective` (the rehearsal of the RCCL calls on a one-GPU box: tests/test_gpu_selection.py)."""
    import torch

    dist = _dist()
    is_np = isinstance(x, np.ndarray)
    if dist is None or (dist.get_world_size() == 1 and not force_collective):
        return x[None]
    ws = dist.get_world_size()
    if dist.get_backend() == "nccl":
        t = torch.from_numpy(np.ascontiguousarray(x)).to(_comm_device()) if is_np else x.contiguous()
        out = torch.empty((ws,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.view(-1))
        return out.cpu().numpy() if is_np else out
    t = torch.from_numpy(np.ascontiguousarray(x)) if is_np else x.detach().cpu().contiguous()
    parts = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(parts, t)
    out = torch.stack(parts)
    return out.numpy() if is_np else out.to(x.device)


# ---- interval.py:610-612 for a sample whose shards stay on their ranks --------------------------------------------------
def _sort_keys(a):
    """Order-preserving uint64 keys of float64 values: the order of np.sort / of the device radix sort (NaN last)."""
    b = np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    return np.where(b >> np.uint64(63), ~b, b | np.uint64(1 << 63))


def interp_cell(n, q):
    """The cell of np.linspace(0, 1, n) that numpy.interp -- which is what scipy's interp1d(kind='linear') calls for real
    1-D data -- evaluates q in: (j, x_j, x_(j+1), exact) with x_j <= q < x_(j+1) (j = n - 1 for q = 1); `exact`: q is the
    grid point x_j, the result is y_j itself.  The same walk as the device's interp_cell (csrc/qt_ops.h)."""
    step = 1.0 / (n - 1)

    def grid(i):
        return 1.0 if i == n - 1 else i * step

    j = min(int(q * (n - 1)), n - 1)
    while j > 0 and grid(j) > q:
        j -= 1
    while j < n - 1 and grid(j + 1) <= q:
        j += 1
    xj = grid(j)
    return j, xj, (grid(j + 1) if j < n - 1 else 1.0), (j == n - 1 or xj == q)


def interp_value(cell, q, yj, yj1):
    """numpy.interp's arithmetic for one query (compiled_base.c, arr_interp), operation by operation."""
    _, xj, xj1, exact = cell
    if exact:
        return yj
    with np.errstate(invalid="ignore"):
        slope = (yj1 - yj) / (xj1 - xj)
        res = slope * (q - xj) + yj
        if np.isnan(res):
            res = slope * (q - xj1) + yj1
            if np.isnan(res) and yj == yj1:
                res = yj
    return res


def selection_plan(n_total, world_size, n_levels):
    """(stride, n_splitters, window_width) of the two-exchange selection, or None when gathering the sorted shards is
    the smaller exchange.  Between two consecutive merged splitters a rank holds fewer than `stride` values and the
    bracket of a level spans at most 2 N + 2 splitters, hence the window bound (2 N + 3) * stride (heavy ties aside:
    those raise the overflow flag and take the gather path)."""
    n_max = -(-n_total // world_size)
    if n_max < 1 or n_levels < 1:
        return None
    factor = 2 * world_size + 3
    p = int(round((n_levels * factor * n_max) ** 0.5))
    p = max(16, min(p, 8192, n_max))
    stride = -(-n_max // p)
    p = -(-n_max // stride)
    width = min(factor * stride, n_max)
    if world_size * (p + n_levels * (2 + width)) >= n_total or n_levels > 64:
        return None
    return stride, p, width


class ShardedSample:
    """A sample of `n_total` float64 values of which this rank holds its `shard_bounds(n_total)` slice -- the bootstrap
    distances of interval.py:598-609 where they were computed.  `quantiles(levels)` is
    `interp1d(np.linspace(0, 1, n), np.sort(sample))(levels)` (interval.py:610-612) evaluated WITHOUT assembling the
    sample: each rank sorts its shard, then either the two small exchanges of the distributed selection (few levels;
    kernels in csrc/qt_ops.h, argument in include/qtomo.h) or one gather of the sorted shards and a merge.  Collective:
    every rank calls the same methods in the same order.

    `local` is a NumPy array (host arithmetic below, the form the gloo tests run) or a float64 torch CUDA tensor (then
    `engine` does every step on the GPU and only the L results cross to the host).  The shard is sorted in place."""

    def __init__(self, local, n_total, engine=None):
        self.n_total = int(n_total)
        self.rank, self.world = world()
        self.sizes = np.array([hi - lo for lo, hi in (shard_bounds(self.n_total, r, self.world) for r in range(self.world))],
                              dtype=np.int64)
        self.device = not isinstance(local, np.ndarray)
        if local.shape[0] != self.sizes[self.rank]:
            raise ValueError(f"rank {self.rank} holds {local.shape[0]} values, its shard of {self.n_total} is {self.sizes[self.rank]}")
        self.engine = engine
        if self.device:
            if engine is None:
                raise ValueError("a device-resident shard needs the engine that sorts and selects on it")
            self.local = engine.sort_dev(local) if local.numel() > 1 else local
        else:
            self.local = np.sort(np.ascontiguousarray(local, dtype=np.float64))
        self._full = None
        self.last_path = None  # 'local' | 'selection' | 'gather': what the last quantiles() call did (tests, bench)

    # -- the whole sorted sample on every rank (one all-gather of the sorted shards + a merge) ---------------------------
    def gather_sorted(self):
        if self._full is not None:
            return self._full
        if self.world == 1:
            self._full = self.local
            return self._full
        width = int(self.sizes.max())
        if self.device:
            import torch

            buf = self.local
            if buf.shape[0] != width:
                buf = torch.empty(width, dtype=torch.float64, device=self.local.device)
                buf[: self.local.shape[0]] = self.local
            runs = allgather_equal(buf)  # (N, width)
            if int(self.sizes.min()) != width:
                runs = torch.cat([runs[r, : int(self.sizes[r])] for r in range(self.world)])
            self._full = self.engine.merge_sorted(runs.reshape(-1), self.sizes)
        else:
            buf = np.full(width, np.nan)
            buf[: self.local.shape[0]] = self.local
            runs = allgather_equal(buf)
            self._full = np.sort(np.concatenate([runs[r, : int(self.sizes[r])] for r in range(self.world)]), kind="stable")
        return self._full

    def _quantiles_of(self, srt, levels):
        if self.device:
            return self.engine.quantiles_of_sorted(srt, levels)
        n = len(srt)
        if n == 1:
            return np.full(len(levels), srt[0])
        out = np.empty(len(levels))
        for t, q in enumerate(levels):
            cell = interp_cell(n, float(q))
            out[t] = interp_value(cell, q, srt[cell[0]], srt[min(cell[0] + 1, n - 1)])
        return out

    def quantiles(self, levels):
        levels = np.ascontiguousarray(np.atleast_1d(levels), dtype=np.float64)
        if levels.size and (levels.min() < 0 or levels.max() > 1 or np.isnan(levels).any()):
            raise ValueError("A value in x_new is outside the interpolation range.")
        if self.n_total < 1:
            raise ValueError("empty sample")
        if levels.size == 0:
            return np.empty(0)
        if self.world == 1 or self._full is not None:
            self.last_path = "local"
            return self._quantiles_of(self.gather_sorted(), levels)
        plan = selection_plan(self.n_total, self.world, levels.size)
        if plan is not None:
            out = self._select_device(levels, *plan) if self.device else self._select_host(levels, *plan)
            if out is not None:
                self.last_path = "selection"
                return out
        self.last_path = "gather"
        return self._quantiles_of(self.gather_sorted(), levels)

    __call__ = quantiles

    # -- distributed selection on the GPU: four launches, two small all-gathers, one read-back -----------------------------
    def _select_device(self, levels, stride, n_split, width):
        import torch

        eng, dev = self.engine, self.local.device
        q = torch.from_numpy(levels).to(dev)
        spl = torch.empty(n_split, dtype=torch.float64, device=dev)
        eng.select_splitters(self.local, stride, n_split, spl)
        all_spl = allgather_equal(spl)
        sizes = torch.from_numpy(self.sizes).to(dev)
        lo = torch.empty(levels.size, dtype=torch.int64, device=dev)
        hi = torch.empty(levels.size, dtype=torch.int64, device=dev)
        eng.select_bracket(all_spl, sizes, stride, self.n_total, q, lo, hi)
        win = torch.empty((levels.size, 2 + width), dtype=torch.float64, device=dev)
        eng.select_window(self.local, lo, hi, width, win)
        all_win = allgather_equal(win)
        res = torch.empty(levels.size + 1, dtype=torch.float64, device=dev)
        flag = torch.zeros(2, dtype=torch.int32, device=dev)
        eng.select_finish(all_win, self.n_total, q, res[: levels.size], flag)
        eng.sync()
        if int(flag[0].item()) != 0:  # the same on every rank (computed from the gathered windows): take the gather path
            return None
        return res[: levels.size].cpu().numpy()

    # -- the same four steps in NumPy (host-resident shards: the reference's own distances, the gloo tests) ---------------
    def _select_host(self, levels, stride, n_split, width):
        spl = host_splitters(self.local, stride, n_split)
        all_spl = allgather_equal(spl.view(np.int64)).view(np.uint64)  # (N, P), bit patterns
        lo, hi = host_bracket(all_spl, self.sizes, stride, self.n_total, levels)
        win = host_window(self.local, lo, hi, width)
        all_win = allgather_equal(win)  # (N, L, 2 + W)
        return host_finish(all_win, self.n_total, levels, width)


# The four local steps of the selection as plain functions of NumPy arrays (the kernels of csrc/qt_ops.h restated; the
# product's path for host-resident shards and the checker of the kernels): with the all-gathers replaced by np.stack they
# run N simulated ranks in one process (tests/test_sharded_quantiles.py sweeps random samples that way).
_KEY_NONE_LO, _KEY_NONE_HI = np.uint64(0), np.uint64(0xFFFFFFFFFFFFFFFF)


def host_splitters(local_sorted, stride, n_split):
    """Radix-sort keys of local_sorted[::stride], padded behind everything to n_split entries."""
    spl = np.full(n_split, _KEY_NONE_HI, dtype=np.uint64)
    take = _sort_keys(local_sorted)[::stride]
    spl[: len(take)] = take
    return spl


def host_bracket(all_spl, sizes, stride, n_total, levels):
    """Per level the tightest splitter pair (lo, hi] that provably holds the order statistics k0 = cell(q) and k1 of the union
    = min(k0 + 1, n - 1) (keys; 0 / ~0 = none): lo = largest splitter with sum_r min(n_r, cnt_r s) <= k0, hi = smallest with
    sum_r ((cnt_r - 1) s + 1) >= k1 + 1, cnt_r(v) = rank r's splitters <= v."""
    nr = len(sizes)
    valid = [int(-(-int(sizes[r]) // stride)) for r in range(nr)]
    cand = np.concatenate([all_spl[r, : valid[r]] for r in range(nr)]) if sum(valid) else np.empty(0, dtype=np.uint64)
    up = np.zeros(len(cand), dtype=np.int64)
    low = np.zeros(len(cand), dtype=np.int64)
    for r in range(nr):
        cnt = np.searchsorted(all_spl[r, : valid[r]], cand, side="right").astype(np.int64)
        up += np.minimum(int(sizes[r]), cnt * stride)
        low += np.where(cnt > 0, (cnt - 1) * stride + 1, 0)
    lo = np.full(len(levels), _KEY_NONE_LO, dtype=np.uint64)
    hi = np.full(len(levels), _KEY_NONE_HI, dtype=np.uint64)
    if n_total >= 2:
        for t, qv in enumerate(levels):
            k0 = interp_cell(n_total, float(qv))[0]
            k1 = min(k0 + 1, n_total - 1)
            sel_lo, sel_hi = cand[up <= k0], cand[low >= k1 + 1]
            if sel_lo.size:
                lo[t] = sel_lo.max()
            if sel_hi.size:
                hi[t] = sel_hi.min()
    return lo, hi


def host_window(local_sorted, lo, hi, width):
    """win[l] = [#values <= lo, w = #values in (lo, hi], the first min(w, width) of them]."""
    keys = _sort_keys(local_sorted)
    win = np.zeros((len(lo), 2 + width))
    for t in range(len(lo)):
        below = 0 if lo[t] == _KEY_NONE_LO else int(np.searchsorted(keys, lo[t], side="right"))
        upto = len(keys) if hi[t] == _KEY_NONE_HI else int(np.searchsorted(keys, hi[t], side="right"))
        w = upto - below
        win[t, 0], win[t, 1] = below, w
        win[t, 2: 2 + min(w, width)] = local_sorted[below: below + min(w, width)]
    return win


def host_finish(all_win, n_total, levels, width):
    """interp1d(linspace(0, 1, n_total), sorted union)(levels) from the gathered windows, or None when a window was
    clipped (w > width: heavy ties -- the caller gathers the sorted shards instead)."""
    if (all_win[:, :, 1] > width).any():
        return None
    nr = all_win.shape[0]
    out = np.empty(len(levels))
    for t, qv in enumerate(levels):
        below = int(all_win[:, t, 0].sum())
        union = np.concatenate([all_win[r, t, 2: 2 + int(all_win[r, t, 1])] for r in range(nr)])
        union = union[np.argsort(_sort_keys(union), kind="stable")]
        if n_total == 1:
            out[t] = union[0 - below]
            continue
        cell = interp_cell(n_total, float(qv))
        k0 = cell[0]
        k1 = min(k0 + 1, n_total - 1)
        out[t] = interp_value(cell, qv, union[k0 - below], union[k1 - below])
    return out


def sharded_map(items, fn):
    """Apply `fn(items[lo:hi]) -> 1-D float64 array` to this rank's slice and all-gather."""
    n = len(items)
    lo, hi = shard_bounds(n)
    local = fn(items[lo:hi]) if hi > lo else np.empty(0)
    return allgather_concat(local, n)
