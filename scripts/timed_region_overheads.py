import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, quantpy_amd as qp
from quantpy_amd.tomography.state import simulate_counts
rng = np.random.default_rng(1234); g = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8)); rho = g @ g.conj().T; rho /= np.trace(rho).real
povm = qp.generate_measurement_matrix("proj-set", 3); shots = np.ones(27) * 100000
np.random.seed(7); counts = simulate_counts(povm, qp.Qobj(rho).bloch, shots, repeats=1000)
eng = qp.get_engine(3); eng.set_povm(povm, shots)
cd = torch.from_numpy(counts).cuda(); out = torch.empty((1000, 8, 8), dtype=torch.complex128, device="cuda")
def t(f, n=200):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e6
for _ in range(10): eng.mle_dev(cd, out)
print("torch.cuda.synchronize() idle: %.1f us" % t(torch.cuda.synchronize))
print("timer_begin (event record): %.1f us" % t(eng.timer_begin))
print("timer_begin+timer_end idle: %.1f us" % t(lambda: (eng.timer_begin(), eng.timer_end())))
print("eng.sync idle: %.1f us" % t(eng.sync))
def one():
    eng.mle_dev(cd, out); eng.sync()
print("one step + spin sync: %.1f us (kernel ~15.9)" % t(one))
def one_t():
    eng.mle_dev(cd, out); torch.cuda.synchronize()
print("one step + torch sync: %.1f us" % t(one_t))
def twenty(sync):
    def f():
        for _ in range(20): eng.mle_dev(cd, out)
        sync()
    return f
print("20 steps + spin sync: %.1f us" % t(twenty(eng.sync), 50))
print("20 steps + torch sync: %.1f us" % t(twenty(torch.cuda.synchronize), 50))
def twenty_ev():
    eng.timer_begin()
    for _ in range(20): eng.mle_dev(cd, out)
    eng.timer_end(); torch.cuda.synchronize()
print("20 steps with events + torch sync (bench's region): %.1f us" % t(twenty_ev, 50))
print("python call overhead of mle_dev alone (async, back to back 200): %.1f us per call" % t(lambda: eng.mle_dev(cd, out), 200))
