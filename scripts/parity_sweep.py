#!/usr/bin/env python3
"""Wide randomized parity sweep of 'lin' and 'mle' against the oracle (n, POVM, shots, start point, state
rank): slower than the unit tests, run by hand after kernel changes.  Prints the worst deviations."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import quantpy_oracle as oracle  # noqa: E402  (checker only)

import quantpy_amd as qp  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
split = len(sys.argv) > 2 and sys.argv[2] == "split"  # force k_mle_start + k_mle_bfgs (two-loop BFGS) for every batch
rng = np.random.default_rng(2718)
worst = dict(lin=0.0, mle_infid=0.0, nit_mismatch=0, status=0, total=0)
for n in (1, 2, 3):
    d = 2**n
    for povm_name in ("proj-set", "sic", "proj"):
        povm = oracle.measurement_matrix(povm_name, n)
        for shots_n in (100, 1000, 100000):
            for rank in (0, 1, 2):
                g = rng.standard_normal((d, rank or d)) + 1j * rng.standard_normal((d, rank or d))
                rho = g @ g.conj().T
                rho /= np.trace(rho).real
                shots = np.ones(povm.shape[0]) * shots_n
                np.random.seed(int(rng.integers(0, 2**31)))
                counts = np.stack([oracle.sample_counts(povm, oracle.bloch_from_matrix(rho), shots) for _ in range(trials)])
                eng = qp.get_engine(n)
                eng.set_povm(qp.generate_measurement_matrix(povm_name, n), shots)
                eng.set_option(2, 0 if split else 1024)  # QT_OPT_MLE_FUSED_MAX_WAVES
                lin = eng.lin(counts, physical=True)
                for init in ("lin", "mixed"):
                    got, info = eng.mle(counts, init=init, return_info=True)
                    for b in range(trials):
                        if init == "lin":
                            worst["lin"] = max(worst["lin"], np.abs(lin[b] - oracle.lin_estimate(counts[b], povm)).max())
                        try:
                            want, winfo = oracle.mle_estimate(counts[b], povm, init=init, jac="analytic", solver="port",
                                                              return_info=True)
                        except np.linalg.LinAlgError:
                            assert info["status"][b] == 1
                            continue
                        worst["total"] += 1
                        if info["status"][b] != 0:
                            worst["status"] += 1
                            continue
                        if info["nit"][b] != winfo["nit"]:
                            worst["nit_mismatch"] += 1
                            print("nit", n, povm_name, shots_n, rank, init, info["nit"][b], winfo["nit"],
                                  oracle.infidelity(got[b], want), flush=True)
                        worst["mle_infid"] = max(worst["mle_infid"], oracle.infidelity(got[b], want))
        print(n, povm_name, worst, flush=True)
print("SUMMARY", worst)
