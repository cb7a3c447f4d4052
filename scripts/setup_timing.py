#!/usr/bin/env python3
"""Time qt_set_povm / qt_set_povm_product (one-off operator set-up) for n = 1..5."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantpy_amd as qp

for n in (1, 2, 3, 4, 5):
    eng = qp.get_engine(n)
    t0 = time.perf_counter(); a = qp.generate_measurement_matrix("proj-set", n); t1 = time.perf_counter()
    eng.set_povm(a, np.full(a.shape[0], 1000)); t2 = time.perf_counter()
    eng.set_povm(np.array(a), np.full(a.shape[0], 1001)); t3 = time.perf_counter()
    print(f"n={n} kron {1e3*(t1-t0):9.2f} ms   set_povm_product {1e3*(t2-t1):9.2f} ms   set_povm (dense) {1e3*(t3-t2):9.2f} ms", flush=True)
