"""the driver's bench read 3.49 ms per device draw where every builder run read 0.07 ms, from ONE
perf_counter pair around 20 draws.  What a GPU that idled does to the next launches: after `idle` ms without work, 40
draws (qt_device_multinomial, 54 000 rows: the 2000-resample table of configs[3]) are timed ONE BY ONE with HIP events
and with the host clock.  Usage: python scripts/idle_ramp.py [idle_ms ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import quantpy_amd as qp  # noqa: E402
from quantpy_amd.tomography.state import born_probabilities  # noqa: E402

idles = [float(a) for a in sys.argv[1:]] or [0, 5, 20, 50, 200, 1000]
n = 3
rng = np.random.default_rng(1234)
g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
rho = g @ g.conj().T
rho /= np.trace(rho)
povm = qp.generate_measurement_matrix("proj-set", n)
eng = qp.get_engine(n)
pv = torch.from_numpy(born_probabilities(povm, qp.Qobj(rho).bloch)).cuda()
shots = torch.full((27,), 100000, dtype=torch.int64, device="cuda")
rows = 2000 * 27
out = torch.empty((rows, 8), dtype=torch.int64, device="cuda")
draw = lambda: eng.device_multinomial(shots, pv, rows, 4242, out=out)  # noqa: E731
for _ in range(200):  # bring the clocks up first (~15 ms of work)
    draw()
eng.sync()
print(f"# {rows} rows per draw; columns: kernel time by HIP events / host wall per call, microseconds")
for idle in idles:
    for _ in range(300):
        draw()
    eng.sync()
    time.sleep(idle * 1e-3)
    ev, host = [], []
    for _ in range(40):
        th = time.perf_counter()
        eng.timer_begin()
        draw()
        ev.append(eng.timer_end() * 1e3)
        host.append((time.perf_counter() - th) * 1e6)
    fmt = lambda v: " ".join(f"{x:7.1f}" for x in v)  # noqa: E731
    print(f"idle {idle:7.1f} ms  events: first 8 [{fmt(ev[:8])}]  median {np.median(ev):7.1f}  max {max(ev):8.1f}")
    print(f"                 host  : first 8 [{fmt(host[:8])}]  median {np.median(host):7.1f}  max {max(host):8.1f}")
    # the same 20 + 20 draws as round 2's loop: one clock pair around 20 back-to-back draws after 20 warm-up draws
    for _ in range(300):
        draw()
    eng.sync()
    time.sleep(idle * 1e-3)
    for _ in range(20):
        draw()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        draw()
    eng.sync()
    print(f"                 round-2 style loop (20 warm-up, then 20 draws / one clock pair): {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us per draw")
