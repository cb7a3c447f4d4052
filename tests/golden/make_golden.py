#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the *imported reference*.

Run ONLY in the development container, where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (nordmtr/quantpy, pure Python) is imported read-only from
/root/reference.  `quantpy/__init__.py` pulls in `tomography/interval.py`, which imports
`cvxopt` at module level for SOCP/LP interval classes that are NOT on the hot path; cvxopt
is not installed and there is no network, so an inert in-process placeholder module is
registered for that one import.  Nothing under /root/reference is
written to, and nothing of the reference is copied: the outputs are numeric input/output
vectors only (npz / json), which is what travels to the GPU box.

Versions recorded in golden/meta.json (reference pins: python 3.9 / numpy 1.23.4 /
scipy 1.9.3; oracle container: see meta.json).
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    placeholder = types.ModuleType("cvxopt")
    placeholder.matrix = lambda *a, **k: None
    placeholder.solvers = types.SimpleNamespace(options={})
    sys.modules["cvxopt"] = placeholder
    sys.path.insert(0, REF)
    import quantpy as qp  # noqa

    return qp


qp = _import_reference()
import quantpy.tomography.state as ref_state  # noqa: E402
from quantpy.routines import (  # noqa: E402
    _left_inv,
    _matrix_to_real_tril_vec,
    _real_tril_vec_to_matrix,
    generate_pauli,
)
from scipy.optimize import minimize as _scipy_minimize  # noqa: E402

warnings.filterwarnings("ignore")

# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------


def ginibre_state(rng, d, rank=None):
    """rho = G G^dagger / Tr, G (d, rank) complex standard normal."""
    r = d if rank is None else rank
    g = rng.standard_normal((d, r)) + 1j * rng.standard_normal((d, r))
    rho = g @ g.conj().T
    return rho / np.trace(rho)


_CAPTURE = {}


def _recording_minimize(fun, x0, *args, **kwargs):
    """Wrap scipy.optimize.minimize as called at reference state.py:213 and record the
    OptimizeResult plus the per-iteration iterates (return_all) without changing what
    the reference computes."""
    opts = dict(kwargs.get("options", {}))
    if kwargs.get("method") == "BFGS":
        opts["return_all"] = True
    kwargs["options"] = opts
    res = _scipy_minimize(fun, x0, *args, **kwargs)
    _CAPTURE["res"] = res
    _CAPTURE["x0"] = np.array(x0, dtype=float)
    return res


ref_state.minimize = _recording_minimize


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------------------
# a1/a2: Pauli bases and POVM tensors
# --------------------------------------------------------------------------------------
def gen_operators():
    out = {}
    for n in (1, 2, 3):
        out[f"pauli_n{n}"] = np.asarray(generate_pauli(n))
        for povm in ("proj", "proj-set", "proj4", "sic"):
            out[f"povm_{povm}_n{n}"] = qp.generate_measurement_matrix(povm, n)
    # custom 1-qubit table (2-D and 3-D) expanded by kron
    tab = np.array([[0.5, 0.1, 0.2, 0.3], [0.5, -0.1, -0.2, -0.3]])
    out["povm_custom2d_in"] = tab
    out["povm_custom2d_n2"] = qp.generate_measurement_matrix(tab, 2)
    # n = 4, 5 'proj-set': shape + 64 sampled entries (full tensor is 63.7 MB at n=5)
    rng = np.random.default_rng(5)
    for n in (4, 5):
        A = qp.generate_measurement_matrix("proj-set", n)
        idx = np.stack([rng.integers(0, s, 64) for s in A.shape], axis=1)
        out[f"povm_proj-set_n{n}_shape"] = np.array(A.shape)
        out[f"povm_proj-set_n{n}_idx"] = idx
        out[f"povm_proj-set_n{n}_val"] = A[idx[:, 0], idx[:, 1], idx[:, 2]]
        out[f"povm_proj-set_n{n}_sum"] = np.array([A.sum(), np.abs(A).sum()])
    save("operators", **out)


# --------------------------------------------------------------------------------------
# a3/a4: Bloch <-> matrix, Born probabilities
# --------------------------------------------------------------------------------------
def gen_states_and_born():
    out = {}
    rng = np.random.default_rng(2024)
    for n in (1, 2, 3, 4):
        d = 2**n
        rhos = np.stack([ginibre_state(rng, d, rank=(None if i % 2 == 0 else 1)) for i in range(6)])
        blochs = np.stack([qp.Qobj(r).bloch for r in rhos])
        back = np.stack([qp.Qobj(b).matrix for b in blochs])
        out[f"rho_n{n}"] = rhos
        out[f"bloch_n{n}"] = blochs
        out[f"rho_from_bloch_n{n}"] = back
        # a non-Hermitian matrix: the reference keeps Re Tr(P_k M^dagger)/d  (qobj.py:132)
        m = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        out[f"nonherm_n{n}"] = m
        out[f"nonherm_bloch_n{n}"] = qp.Qobj(m).bloch
        if n <= 3:
            for povm in ("proj", "proj-set", "sic"):
                A = qp.generate_measurement_matrix(povm, n)
                p = np.stack([np.einsum("ijk,k->ij", A, b) * d for b in blochs])
                out[f"born_{povm}_n{n}"] = p
    save("states_born", **out)


# --------------------------------------------------------------------------------------
# a4: counts (pins the legacy RNG call order), a6/a7: linear inversion
# --------------------------------------------------------------------------------------
def gen_counts_lin():
    out = {}
    # C1: |0>, 10 000 shots / setting, seed 0
    np.random.seed(0)
    t = qp.StateTomograph(qp.qobj.zero(1))
    t.experiment(10000)
    out["C1_counts"] = t.results
    out["C1_n_meas"] = t.n_measurements
    out["C1_lin"] = t.point_estimate("lin").matrix
    out["C1_lin_unphys"] = t.point_estimate("lin", physical=False).matrix
    out["C1_lin_bloch_unphys"] = t.point_estimate("lin", physical=False).bloch
    out["C1_mle"] = t.point_estimate("mle").matrix

    # C2: Ginibre rho (rng 1234), seed 7, 100 000 shots / setting, 8 consecutive trials
    rng = np.random.default_rng(1234)
    rho = ginibre_state(rng, 8)
    out["C2_rho_true"] = rho
    np.random.seed(7)
    t = qp.StateTomograph(qp.Qobj(rho))
    counts, lin, lin_u, mle, nit, nfev = [], [], [], [], [], []
    for _ in range(8):
        t.experiment(100000, "proj-set")
        counts.append(t.results.copy())
        lin_u.append(t.point_estimate("lin", physical=False).matrix)
        lin.append(t.point_estimate("lin").matrix)
        mle.append(t.point_estimate("mle").matrix)
        nit.append(_CAPTURE["res"].nit)
        nfev.append(_CAPTURE["res"].nfev)
    out["C2_counts"] = np.stack(counts)
    out["C2_lin"] = np.stack(lin)
    out["C2_lin_unphys"] = np.stack(lin_u)
    out["C2_mle"] = np.stack(mle)
    out["C2_nit"] = np.array(nit)
    out["C2_nfev"] = np.array(nfev)

    # per-setting shot vector + other POVMs + n = 1, 2 (ragged N_s exercises the weights)
    rng = np.random.default_rng(99)
    k = 0
    for n, povm, nm in (
        (1, "proj-set", np.array([100, 2000, 30000])),
        (2, "proj-set", 5000),
        (2, "proj", 20000),
        (2, "sic", 20000),
        (3, "proj", 200000),
        (3, "sic", 50000),
        (3, "proj-set", 1000),
    ):
        rho = ginibre_state(rng, 2**n)
        np.random.seed(100 + k)
        t = qp.StateTomograph(qp.Qobj(rho))
        t.experiment(nm, povm)
        key = f"L{k}"
        out[key + "_n"] = np.array(n)
        out[key + "_povm"] = np.array(povm)
        out[key + "_seed"] = np.array(100 + k)
        out[key + "_rho_true"] = rho
        out[key + "_nmeas_arg"] = np.asarray(nm)
        out[key + "_counts"] = t.results
        out[key + "_povm_matrix"] = t.povm_matrix
        A = np.reshape(
            t.povm_matrix * t.n_measurements[:, None, None] / np.sum(t.n_measurements),
            (-1, t.povm_matrix.shape[-1]),
        )
        out[key + "_leftinv"] = _left_inv(A)
        r_u = t.point_estimate("lin", physical=False)
        out[key + "_lin_unphys"] = r_u.matrix
        out[key + "_lin_bloch"] = r_u.bloch
        out[key + "_lin_eigs"] = np.linalg.eigvalsh(r_u.matrix)
        out[key + "_lin"] = t.point_estimate("lin").matrix
        k += 1
    out["n_lin_cases"] = np.array(k)
    save("counts_lin", **out)


# --------------------------------------------------------------------------------------
# a8/a9: Cholesky parametrisation, NLL, finite-difference gradient as SciPy forms it
# --------------------------------------------------------------------------------------
def gen_chol_nll():
    from scipy.optimize._numdiff import approx_derivative

    out = {}
    rng = np.random.default_rng(31)
    k = 0
    for n in (1, 2, 3):
        d = 2**n
        for povm, shots in (("proj-set", 100000), ("proj", 300), ("sic", 1000)):
            rho = ginibre_state(rng, d)
            np.random.seed(500 + k)
            t = qp.StateTomograph(qp.Qobj(rho))
            t.experiment(shots, povm)
            x = _matrix_to_real_tril_vec(rho)
            xr = x + 0.05 * rng.standard_normal(x.shape)  # a generic (unnormalised) point
            key = f"N{k}"
            out[key + "_n"] = np.array(n)
            out[key + "_counts"] = t.results
            out[key + "_povm_matrix"] = t.povm_matrix
            out[key + "_rho"] = rho
            out[key + "_x"] = x
            out[key + "_LLh"] = _real_tril_vec_to_matrix(x)
            out[key + "_xr"] = xr
            out[key + "_LLh_r"] = _real_tril_vec_to_matrix(xr)
            out[key + "_nll_x"] = np.array(t._nll(x))
            out[key + "_nll_xr"] = np.array(t._nll(xr))
            # SciPy's BFGS gradient: 2-point, absolute step sqrt(eps)  (optimize/_optimize.py)
            eps = np.sqrt(np.finfo(float).eps)
            out[key + "_fdgrad_xr"] = approx_derivative(t._nll, xr, method="2-point", abs_step=eps)
            out[key + "_cgrad_xr"] = approx_derivative(t._nll, xr, method="3-point", abs_step=1e-6)
            k += 1
    out["n_nll_cases"] = np.array(k)
    save("chol_nll", **out)


# --------------------------------------------------------------------------------------
# a10: MLE (BFGS) trials incl. iterating ones, with the optimizer trajectory
# --------------------------------------------------------------------------------------
def gen_mle():
    out = {}
    rng = np.random.default_rng(77)
    cases = []
    for n in (1, 2, 3):
        d = 2**n
        full = ginibre_state(rng, d)
        rank1 = ginibre_state(rng, d, rank=1)
        rank2 = ginibre_state(rng, d, rank=min(2, d))
        zero = qp.qobj.zero(n).matrix
        mixed = qp.qobj.fully_mixed(n).matrix
        named = [("full", full), ("rank1", rank1), ("zero", zero), ("mixed", mixed)]
        if n >= 2:
            named.append(("ghz", qp.qobj.GHZ(n).matrix))
            named.append(("rank2", rank2))
        for name, rho in named:
            for shots in (100, 100000):
                for init in ("lin", "mixed"):
                    cases.append((n, name, rho, shots, init, "proj-set"))
        cases.append((n, "full", full, 1000, "lin", "proj"))
        cases.append((n, "rank1", rank1, 1000, "lin", "sic"))
    k = 0
    for n, name, rho, shots, init, povm in cases:
        np.random.seed(1000 + k)
        t = qp.StateTomograph(qp.Qobj(rho))
        t.experiment(shots, povm)
        key = f"M{k}"
        out[key + "_n"] = np.array(n)
        out[key + "_name"] = np.array(name)
        out[key + "_init"] = np.array(init)
        out[key + "_povm"] = np.array(povm)
        out[key + "_shots"] = np.array(shots)
        out[key + "_seed"] = np.array(1000 + k)
        out[key + "_rho_true"] = rho
        out[key + "_counts"] = t.results
        _CAPTURE.clear()
        try:
            r = t.point_estimate("mle", init=init)
            res = _CAPTURE["res"]
            out[key + "_rho"] = r.matrix
            out[key + "_ok"] = np.array(1)
            out[key + "_x0"] = _CAPTURE["x0"]
            out[key + "_xfinal"] = res.x
            out[key + "_fun"] = np.array(res.fun)
            out[key + "_jac"] = res.jac
            out[key + "_nit"] = np.array(res.nit)
            out[key + "_nfev"] = np.array(res.nfev)
            out[key + "_njev"] = np.array(res.njev)
            out[key + "_status"] = np.array(res.status)
            av = np.asarray(res.allvecs)
            out[key + "_allvecs"] = av
            out[key + "_fvals"] = np.array([t._nll(v) for v in av])
            print(f"   M{k}: n={n} {name:6s} shots={shots:6d} init={init:5s} {povm:8s} "
                  f"nit={res.nit} nfev={res.nfev} status={res.status}")
        except Exception as e:  # e.g. LinAlgError from la.cholesky (routines.py:86)
            out[key + "_ok"] = np.array(0)
            out[key + "_error"] = np.array(type(e).__name__)
            print(f"   M{k}: n={n} {name} shots={shots} init={init}: {type(e).__name__}: {e}")
        k += 1
    out["n_mle_cases"] = np.array(k)
    save("mle", **out)


# --------------------------------------------------------------------------------------
# a11-a15: process tomography (C3, n=1 cases, notebook known answer)
# --------------------------------------------------------------------------------------
def gen_process():
    out = {}
    # --- notebook known answer (notebooks/Moments.ipynb cells 3, 5, 6, 7; input.json:18-23)
    s = 0.2886751345948129
    s4 = 0.28867513459481287
    sic_states = [[0.5, s, s, s4], [0.5, s, -s, -s4], [0.5, -s, s, -s4], [0.5, -s, -s, s4]]
    target = [0.5, 0, 0, 0, 0, 0, 0, 0.5, 0, 0, 0.5, 0, 0, 0.5, 0, 0]
    counts = np.array(json.load(open(os.path.join(REF, "input.json")))["outcomes"])
    tmg = qp.ProcessTomograph(qp.Channel(qp.Qobj(target)), input_states=[qp.Qobj(b) for b in sic_states])
    np.random.seed(0)
    tmg.experiment(10000, "proj-set")
    tmg.results = counts
    ch = tmg.point_estimate(cptp=False)
    out["NB_input_blochs"] = np.array(sic_states)
    out["NB_counts"] = counts
    out["NB_choi_nocptp"] = ch.choi.matrix
    out["NB_choi_bloch_nocptp"] = ch.choi.bloch
    out["NB_lifp_oper"] = tmg._lifp_oper
    out["NB_lifp_oper_inv"] = tmg._lifp_oper_inv
    out["NB_choi_cptp"] = tmg.point_estimate(cptp=True).choi.matrix
    # values printed in the notebook itself (Moments.ipynb cell 6 / cell 7 outputs)
    out["NB_printed_bloch_nocptp"] = np.array(
        [5.00000000e-01, 1.20000000e-03, 1.27500000e-03, 2.47500000e-03,
         -2.77555756e-17, -3.63730670e-03, -1.42894192e-03, 5.05109317e-01,
         0.00000000e+00, -6.49519053e-03, 5.01818420e-01, 8.87676039e-03,
         5.55111512e-17, 4.96665569e-01, -4.33012702e-05, 4.54663337e-03])
    out["NB_printed_choi_cptp"] = np.array(
        [[0.50321637 + 0.0j, 0.49729217 + 2.15697599e-03j, 0.49706507 + 2.56306202e-03j, -0.50099188 - 4.82421850e-03j],
         [0.49729217 - 2.15697599e-03j, 0.49678363 - 2.16840434e-19j, 0.49479818 + 2.48192079e-03j, -0.49706507 - 2.56306202e-03j],
         [0.49706507 - 2.56306202e-03j, 0.49479818 - 2.48192079e-03j, 0.49856023 + 0.0j, -0.49742196 - 2.79553158e-03j],
         [-0.50099188 + 4.82421850e-03j, -0.49706507 + 2.56306202e-03j, -0.49742196 + 2.79553158e-03j, 0.50143977 + 0.0j]])

    # --- generic cases: (n, channel builder, shots, povm, seed)
    def dyk_iters(tmg_, choi_vec):
        """count Dykstra iterations the way process.py:243-256 stops."""
        import scipy.linalg as la
        from quantpy.routines import _vec2mat

        x = choi_vec.copy()
        p = q = y = 0
        for i in range(1000):
            crit = 0
            y_diff = tmg_.tp_projection(qp.Channel(_vec2mat(x + p)), vectorized=True) - y
            y = y + y_diff
            x_diff = tmg_.cp_projection(qp.Channel(_vec2mat(y + q)), vectorized=True) - x
            x = x + x_diff
            crit += 2 * (np.abs(np.sum(y_diff.T.conj() * q)) + np.abs(np.sum(x_diff.T.conj() * p)))
            p_diff = x - y
            p = p + p_diff
            q_diff = y - x
            q = q + q_diff
            crit += la.norm(p_diff) ** 2 + la.norm(q_diff) ** 2
            if crit < 1e-12:
                break
        return i + 1

    from quantpy.routines import _mat2vec

    cases = [
        ("P0", 1, lambda: qp.channel.depolarizing(0.1, 1), 10000, "proj-set", 11),
        ("P1", 1, lambda: qp.operator.H.as_channel(), 1000, "proj-set", 12),
        ("P2", 1, lambda: qp.channel.amplitude_damping(0.3), 5000, "sic", 13),
        ("C3", 2, lambda: qp.channel.depolarizing(0.1, 2), 10000, "proj-set", 11),
        ("P4", 2, lambda: qp.operator.CNOT.as_channel(), 1000, "proj-set", 14),
    ]
    for key, n, mk, shots, povm, seed in cases:
        np.random.seed(seed)
        tmg = qp.ProcessTomograph(mk())
        tmg.experiment(shots, povm)
        out[key + "_n"] = np.array(n)
        out[key + "_seed"] = np.array(seed)
        out[key + "_shots"] = np.array(shots)
        out[key + "_povm"] = np.array(povm)
        out[key + "_true_choi"] = tmg.channel.choi.matrix
        out[key + "_input_states"] = np.stack([s_.matrix for s_ in tmg.input_basis.elements])
        out[key + "_output_states"] = np.stack([t_.state.matrix for t_ in tmg.tomographs])
        out[key + "_counts"] = tmg.results
        ch = tmg.point_estimate("lifp", cptp=False)
        out[key + "_choi_nocptp"] = ch.choi.matrix
        if n == 1:
            out[key + "_lifp_oper"] = tmg._lifp_oper
            out[key + "_lifp_oper_inv"] = tmg._lifp_oper_inv
        else:
            # 576 x 256 complex: keep a strided sample + checksum (full = 2.4 MB x 2)
            out[key + "_lifp_oper_rows"] = tmg._lifp_oper[::37]
            out[key + "_lifp_oper_inv_cols"] = tmg._lifp_oper_inv[:, ::37]
            out[key + "_lifp_oper_abs_sum"] = np.array(np.abs(tmg._lifp_oper).sum())
        out[key + "_dykstra_iters"] = np.array(dyk_iters(tmg, _mat2vec(ch.choi.matrix)))
        out[key + "_choi_cptp"] = tmg.point_estimate("lifp", cptp=True).choi.matrix
        out[key + "_states_lin"] = tmg.point_estimate("states", cptp=False).choi.matrix
        out[key + "_states_lin_cptp"] = tmg.point_estimate("states").choi.matrix
        if n == 1:
            out[key + "_states_mle_cptp"] = tmg.point_estimate("states", states_est_method="mle").choi.matrix
        out[key + "_tp_only"] = tmg.tp_projection(ch).choi.matrix
        out[key + "_cp_only"] = tmg.cp_projection(ch).choi.matrix
        print(f"   {key}: n={n} dykstra iters={int(out[key + '_dykstra_iters'])}")
    save("process", **out)


# --------------------------------------------------------------------------------------
# a11-a15 at n = 3 (process.py:142-229 is size-generic; 13824 x 4096 complex design matrix, minutes in the reference)
# --------------------------------------------------------------------------------------
def gen_process3():
    import scipy.linalg as la
    import time as _time
    from quantpy.routines import _mat2vec, _vec2mat

    def dyk_iters(tmg_, choi_vec):
        x = choi_vec.copy()
        p = q = y = 0
        for i in range(1000):
            y_diff = tmg_.tp_projection(qp.Channel(_vec2mat(x + p)), vectorized=True) - y
            y = y + y_diff
            x_diff = tmg_.cp_projection(qp.Channel(_vec2mat(y + q)), vectorized=True) - x
            x = x + x_diff
            crit = 2 * (np.abs(np.sum(y_diff.T.conj() * q)) + np.abs(np.sum(x_diff.T.conj() * p)))
            p_diff = x - y
            p = p + p_diff
            q_diff = y - x
            q = q + q_diff
            crit += la.norm(p_diff) ** 2 + la.norm(q_diff) ** 2
            if crit < 1e-12:
                break
        return i + 1

    out = {}
    t0 = _time.time()
    np.random.seed(31)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
    tmg.experiment(10000, "proj-set")
    out["Q0_seed"], out["Q0_shots"] = np.array(31), np.array(10000)
    out["Q_true_choi"] = tmg.channel.choi.matrix
    out["Q_input_states"] = np.stack([s_.matrix for s_ in tmg.input_basis.elements])
    out["Q0_counts"] = tmg.results
    ch = tmg.point_estimate("lifp", cptp=False)  # builds _lifp_oper (13824 x 4096) and its left inverse
    print(f"   design matrix + left inverse + lifp: {_time.time() - t0:.0f} s")
    out["Q_lifp_oper_rows"] = tmg._lifp_oper[::997]
    out["Q_lifp_oper_inv_cols"] = tmg._lifp_oper_inv[:, ::1999]
    out["Q_lifp_oper_abs_sum"] = np.array(np.abs(tmg._lifp_oper).sum())
    # a second data set on the same operators (fewer shots would change the operator only through the
    # shots ratio N_s / sum N, which is the same): the estimators below reuse tmg._lifp_oper_inv, as
    # point_estimate() itself would recompute it identically
    for key, counts in (("Q0", tmg.results), ("Q1", None)):
        if counts is None:
            np.random.seed(32)
            tmg.experiment(10000, "proj-set")
            out["Q1_seed"], out["Q1_shots"] = np.array(32), np.array(10000)
            out["Q1_counts"] = tmg.results
        # (copies: _vec2mat hands back a transposed view, _mat2vec of that is again a view, and _cptp_projection_vec
        #  updates its argument in place -- cptp_projection(ch) would overwrite ch's own matrix)
        raw = tmg._point_estimate_lifp(cptp=False).choi.matrix.copy()
        out[key + "_choi_nocptp"] = raw
        out[key + "_dykstra_iters"] = np.array(dyk_iters(tmg, _mat2vec(raw.copy())))
        out[key + "_choi_cptp"] = tmg.cptp_projection(qp.Channel(raw.copy())).choi.matrix.copy()
        out[key + "_tp_only"] = tmg.tp_projection(qp.Channel(raw.copy())).choi.matrix.copy()
        out[key + "_cp_only"] = tmg.cp_projection(qp.Channel(raw.copy())).choi.matrix.copy()
        assert np.abs(raw - out[key + "_choi_cptp"]).max() > 1e-6
        out[key + "_states_lin"] = tmg._point_estimate_states(False, "lin", True, "lin", 1000, 1e-10).choi.matrix
        out[key + "_states_lin_cptp"] = tmg._point_estimate_states(True, "lin", True, "lin", 1000, 1e-10).choi.matrix
        print(f"   {key}: dykstra iters={int(out[key + '_dykstra_iters'])}  ({_time.time() - t0:.0f} s)")
    save("process3", **out)


# --------------------------------------------------------------------------------------
# (f) MomentInterval: closed-form CI of the CLI scripts  (interval.py:59-110, stats.py)
# --------------------------------------------------------------------------------------
def gen_moment():
    out = {}
    cls = np.array([0.5, 0.75, 0.9, 0.99])
    k = 0
    for n, povm, shots, seed in ((1, "proj-set", 1000, 1), (2, "proj-set", 10000, 2), (3, "proj-set", 100000, 3),
                                 (2, "sic", 5000, 4)):
        rho = ginibre_state(np.random.default_rng(300 + k), 2**n)
        np.random.seed(seed)
        t = qp.StateTomograph(qp.Qobj(rho))
        t.experiment(shots, povm)
        key = f"S{k}"
        out[key + "_n"] = np.array(n)
        out[key + "_povm"] = np.array(povm)
        out[key + "_counts"] = t.results
        for distr in ("gamma", "norm", "exp"):
            iv = qp.MomentInterval(t, distr_type=distr)
            out[key + "_" + distr] = iv(cls)[0]
        k += 1
    out["n_state_cases"] = np.array(k)
    out["conf_levels"] = cls
    # process: the notebook case (Moments.ipynb cell 4 prints [0.01529795, 0.01764183, 0.01983802])
    s_ = 0.2886751345948129
    s4 = 0.28867513459481287
    sic_states = [[0.5, s_, s_, s4], [0.5, s_, -s_, -s4], [0.5, -s_, s_, -s4], [0.5, -s_, -s_, s4]]
    target = [0.5, 0, 0, 0, 0, 0, 0, 0.5, 0, 0, 0.5, 0, 0, 0.5, 0, 0]
    counts = np.array(json.load(open(os.path.join(REF, "input.json")))["outcomes"])
    tmg = qp.ProcessTomograph(qp.Channel(qp.Qobj(target)), input_states=[qp.Qobj(b) for b in sic_states])
    np.random.seed(0)
    tmg.experiment(10000, "proj-set")
    tmg.results = counts
    out["NB_process_radii"] = qp.MomentInterval(tmg)([0.5, 0.75, 0.9])[0]
    out["NB_process_printed"] = np.array([0.01529795, 0.01764183, 0.01983802])
    np.random.seed(11)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 2))
    tmg.experiment(10000, "proj-set")
    out["C3_counts"] = tmg.results
    out["C3_process_radii"] = qp.MomentInterval(tmg)(cls)[0]
    save("moment", **out)


# --------------------------------------------------------------------------------------
# a16: bootstrap (small): counts per resample, distances, quantiles
# --------------------------------------------------------------------------------------
def gen_bootstrap():
    out = {}
    rng = np.random.default_rng(1234)
    rho = ginibre_state(rng, 8)
    for tag, n, method, n_points, shots in (("B3lin", 3, "lin", 64, 100000), ("B3mle", 3, "mle", 24, 100000),
                                             ("B1mle", 1, "mle", 40, 1000), ("B2mle", 2, "mle", 24, 200)):
        if n == 3:
            state = qp.Qobj(rho)
        else:
            state = qp.Qobj(ginibre_state(np.random.default_rng(50 + n), 2**n))
        np.random.seed(7)
        t = qp.StateTomograph(state)
        t.experiment(shots, "proj-set")
        centre = t.point_estimate(method)
        out[tag + "_true"] = state.matrix
        out[tag + "_counts0"] = t.results
        out[tag + "_centre"] = centre.matrix
        # replay of interval.py:598-609 that also records each resample's counts
        np.random.seed(4242)
        boot = qp.StateTomograph(centre, t.dst)
        cs, ds, rs = [], [], []
        for _ in range(n_points):
            boot.experiment(t.n_measurements, t.povm_matrix)
            r = boot.point_estimate(method=method)
            cs.append(boot.results.copy())
            rs.append(r.matrix)
            ds.append(t.dst(r, centre))
        # and the reference class itself on the same RNG stream: must agree with the replay
        np.random.seed(4242)
        iv = qp.BootstrapStateInterval(t, n_points=n_points, method=method)
        dist, cl = iv([0.5, 0.9, 0.95])
        assert np.allclose(np.sort(ds), np.sort(iv.cl_to_dist.y), rtol=0, atol=0), "replay differs"
        out[tag + "_boot_counts"] = np.stack(cs)
        out[tag + "_boot_rho"] = np.stack(rs)
        out[tag + "_boot_dist"] = np.array(ds)
        out[tag + "_cl"] = np.array(cl)
        out[tag + "_cl_dist"] = np.array(dist)
        out[tag + "_nmeas"] = t.n_measurements
        print(f"   {tag}: quantiles {dist}")
    save("bootstrap", **out)


# --------------------------------------------------------------------------------------
# n = 4, 5: 'lin' and one NLL value (full MLE is infeasible in the reference: SURVEY 6.2)
# --------------------------------------------------------------------------------------
def gen_large():
    out = {}
    for n, shots in ((4, 100000), (5, 1000000)):
        rng = np.random.default_rng(1234 + n)
        rho = ginibre_state(rng, 2**n)
        np.random.seed(7)
        t = qp.StateTomograph(qp.Qobj(rho))
        t.experiment(shots, "proj-set")
        out[f"n{n}_rho_true"] = rho
        out[f"n{n}_counts"] = t.results
        out[f"n{n}_lin_unphys"] = t.point_estimate("lin", physical=False).matrix
        lin = t.point_estimate("lin").matrix
        out[f"n{n}_lin"] = lin
        x = _matrix_to_real_tril_vec(lin)
        out[f"n{n}_x"] = x
        out[f"n{n}_nll"] = np.array(t._nll(x))
        print(f"   n={n}: nll={float(out[f'n{n}_nll']):.12f}")
    save("large", **out)


# --------------------------------------------------------------------------------------
# (f) 'pgdb' process estimator (process.py:291-308): what the reference returns, the pieces of its
# first iteration, and the same loop driven to convergence through the reference's own methods
# --------------------------------------------------------------------------------------
def gen_pgdb():
    from quantpy.routines import _mat2vec, _vec2mat
    from quantpy.qobj import fully_mixed

    out = {}
    cases = [
        ("P0", 1, lambda: qp.channel.depolarizing(0.1, 1), 10000, "proj-set", 11),
        ("P2", 1, lambda: qp.channel.amplitude_damping(0.3), 5000, "sic", 13),
        ("C3", 2, lambda: qp.channel.depolarizing(0.1, 2), 10000, "proj-set", 11),
    ]
    for key, n, mk, shots, povm, seed in cases:
        np.random.seed(seed)
        tmg = qp.ProcessTomograph(mk())
        tmg.experiment(shots, povm)
        out[key + "_n"] = np.array(n)
        out[key + "_povm"] = np.array(povm)
        out[key + "_counts"] = tmg.results
        out[key + "_input_states"] = np.stack([s_.matrix for s_ in tmg.input_basis.elements])
        ch = tmg.point_estimate("pgdb")  # the reference as it is
        out[key + "_returned"] = ch.choi.matrix
        # first iteration, piece by piece, with the reference's own operators and methods
        v = _mat2vec(fully_mixed(n * 2).matrix)
        mu, gamma = 1.5 / 4**n, 0.3
        probas = tmg._lifp_oper @ v
        grad = -tmg._lifp_oper.T.conj() @ (tmg._unnorm_results / probas)
        direction = tmg._cptp_projection_vec(v - grad / mu) - v
        alpha = 1
        while tmg._nll(v + alpha * direction) - tmg._nll(v) > gamma * alpha * np.dot(direction, grad):
            alpha /= 2
        out[key + "_it0_probas"] = probas
        out[key + "_it0_grad"] = grad
        out[key + "_it0_direction"] = direction
        out[key + "_it0_alpha"] = np.array(alpha)
        out[key + "_it0_dot"] = np.array(np.dot(direction, grad))
        out[key + "_it0_nll"] = np.array([tmg._nll(v), tmg._nll(v + alpha * direction)])
        # the loop with the step accepted and the exit on a small decrease, capped at 20 iterations
        cap, tol = 20, 1e-10
        nlls = []
        for it in range(cap):
            probas = tmg._lifp_oper @ v
            grad = -tmg._lifp_oper.T.conj() @ (tmg._unnorm_results / probas)
            direction = tmg._cptp_projection_vec(v - grad / mu) - v
            alpha = 1
            while tmg._nll(v + alpha * direction) - tmg._nll(v) > gamma * alpha * np.dot(direction, grad):
                alpha /= 2
            new = v + alpha * direction
            f0, f1 = tmg._nll(v), tmg._nll(new)
            nlls.append([f0, f1, alpha])
            v = new
            if not (f0 - f1 > tol):
                break
        out[key + "_conv_cap"] = np.array(cap)
        out[key + "_conv_choi"] = _vec2mat(v)
        out[key + "_conv_trace"] = np.array(nlls)
        print(f"   {key}: returned == start: {np.allclose(ch.choi.matrix, fully_mixed(n * 2).matrix)}; "
              f"converged variant: {len(nlls)} steps, nll {nlls[0][0].real:.6f} -> {nlls[-1][1].real:.6f}")
    save("pgdb", **out)


# --------------------------------------------------------------------------------------
# (f) 'mle-constr' (state.py:231-253): SLSQP under the unit-trace constraint
# --------------------------------------------------------------------------------------
def gen_constr():
    out = {}
    k = 0
    for n, povm, shots, seed in ((1, "proj-set", 1000, 1), (2, "proj-set", 10000, 2), (3, "proj-set", 100000, 3),
                                 (3, "proj-set", 1000, 4), (2, "sic", 100, 5)):
        rho = ginibre_state(np.random.default_rng(500 + k), 2**n)
        np.random.seed(seed)
        t = qp.StateTomograph(qp.Qobj(rho))
        t.experiment(shots, povm)
        key = f"K{k}"
        out[key + "_n"] = np.array(n)
        out[key + "_povm"] = np.array(povm)
        out[key + "_counts"] = t.results
        for init in ("lin", "mixed"):
            out[key + "_" + init] = t.point_estimate("mle-constr", init=init).matrix
        print(f"   {key}: n={n} {povm} shots={shots}")
        k += 1
    out["n_cases"] = np.array(k)
    save("constr", **out)


# --------------------------------------------------------------------------------------
# (f) MHMCStateInterval (interval.py:689-750, mhmc.py)
# --------------------------------------------------------------------------------------
def gen_mhmc():
    out = {}
    cls = np.array([0.1, 0.5, 0.9, 0.99])
    k = 0
    for n, povm, shots, seed, n_points, burn, step, thin in ((1, "proj-set", 1000, 1, 150, 60, 0.01, 1),
                                                             (2, "proj-set", 10000, 2, 120, 40, 0.005, 2),
                                                             (3, "proj-set", 100000, 3, 100, 30, 0.001, 1)):
        rho = ginibre_state(np.random.default_rng(700 + k), 2**n)
        np.random.seed(seed)
        t = qp.StateTomograph(qp.Qobj(rho))
        t.experiment(shots, povm)
        est = t.point_estimate("mle")
        key = f"H{k}"
        np.random.seed(100 + seed)
        iv = qp.MHMCStateInterval(t, n_points=n_points, step=step, burn_steps=burn, thinning=thin)
        radii = iv(cls)[0]
        out[key + "_n"] = np.array(n)
        out[key + "_povm"] = np.array(povm)
        out[key + "_counts"] = t.results
        out[key + "_state"] = est.matrix
        out[key + "_args"] = np.array([n_points, burn, thin])
        out[key + "_step"] = np.array(step)
        out[key + "_rng_seed"] = np.array(100 + seed)
        out[key + "_radii"] = radii
        out[key + "_all_dist"] = iv.cl_to_dist(np.linspace(0, 1, n_points))
        out[key + "_final_x"] = iv.chain.x_t
        print(f"   {key}: n={n} radii {radii}")
        k += 1
    out["n_cases"] = np.array(k)
    out["conf_levels"] = cls
    # process chains (interval.py:763-850): every proposal goes through the Dykstra projection
    for key, n, mk, shots, seed, n_points, burn, step in (("Q0", 1, lambda: qp.channel.depolarizing(0.1, 1), 2000, 51, 40, 15, 0.01),
                                                          ("Q1", 2, lambda: qp.channel.depolarizing(0.2, 2), 5000, 52, 12, 5, 0.003)):
        np.random.seed(seed)
        tmg = qp.ProcessTomograph(mk())
        tmg.experiment(shots, "proj-set")
        ch = tmg.point_estimate("lifp")
        np.random.seed(200 + seed)
        iv = qp.MHMCProcessInterval(tmg, n_points=n_points, step=step, burn_steps=burn, return_samples=True)
        dist, cl, rate, mats = iv.setup()
        out[key + "_n"] = np.array(n)
        out[key + "_seed"] = np.array(seed)
        out[key + "_shots"] = np.array(shots)
        out[key + "_counts"] = tmg.results
        out[key + "_channel"] = ch.choi.matrix
        out[key + "_args"] = np.array([n_points, burn])
        out[key + "_step"] = np.array(step)
        out[key + "_dist"] = dist
        out[key + "_rate"] = np.array(rate)
        out[key + "_samples"] = np.stack(mats)
        print(f"   {key}: n={n} acceptance {rate:.3f} dist {dist[:3]} ...")
    save("mhmc", **out)


# --------------------------------------------------------------------------------------
# (f) SugiyamaInterval (interval.py:219-265) and HolderInterval (interval.py:421-539)
# --------------------------------------------------------------------------------------
def gen_holder():
    out = {}
    cls = np.array([0.3, 0.6, 0.9])
    k = 0
    for n, povm, shots, seed, dst in ((1, "proj-set", 1000, 1, "hs"), (2, "proj-set", 10000, 2, "trace"),
                                      (2, "sic", 5000, 3, "if"), (3, "proj-set", 100000, 4, "hs")):
        rho = ginibre_state(np.random.default_rng(900 + k), 2**n)
        np.random.seed(seed)
        t = qp.StateTomograph(qp.Qobj(rho), dst)
        t.experiment(shots, povm)
        key = f"S{k}"
        out[key + "_n"] = np.array(n)
        out[key + "_povm"] = np.array(povm)
        out[key + "_dst"] = np.array(dst)
        out[key + "_counts"] = t.results
        out[key + "_radii"] = qp.SugiyamaInterval(t, n_points=400)(cls)[0]
        k += 1
    out["n_state_cases"] = np.array(k)
    out["conf_levels"] = cls
    # Holder: 1-qubit process, three kinds
    np.random.seed(21)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 1))
    tmg.experiment(2000, "proj-set")
    tmg.point_estimate("states")  # gives every inner tomograph its reconstructed_state
    out["H_counts"] = tmg.results
    out["H_states"] = np.stack([t_.reconstructed_state.matrix for t_ in tmg.tomographs])
    r = qp.HolderInterval(tmg, n_points=300, kind="sugiyama")(cls)
    out["H_sugiyama_dist"], out["H_sugiyama_cl"] = r
    np.random.seed(31)
    r = qp.HolderInterval(tmg, n_points=40, kind="bootstrap", method="lin")(cls)
    out["H_bootstrap_dist"], out["H_bootstrap_cl"] = r
    np.random.seed(41)
    r = qp.HolderInterval(tmg, n_points=60, kind="mhmc", step=0.01, burn_steps=20)(cls)
    out["H_mhmc_dist"], out["H_mhmc_cl"] = r
    for bad in ("wang", "moment"):
        try:
            qp.HolderInterval(tmg, kind=bad)(cls)
            out["H_error_" + bad] = np.array("none")
        except Exception as e:  # noqa: BLE001
            out["H_error_" + bad] = np.array(type(e).__name__)
    print("   holder:", out["H_sugiyama_dist"], out["H_bootstrap_dist"], out["H_mhmc_dist"], out["H_error_wang"],
          out["H_error_moment"])
    save("holder", **out)


# --------------------------------------------------------------------------------------
# round 2: what round 1 left unpinned -- if_dst / trace_dst (a17), warm_start accumulation
# (state.py:116-124, process.py:122-129), BootstrapProcessInterval from the reference itself
# (interval.py:615-685) incl. n = 2 and the 'states' / 'pgdb' methods
# --------------------------------------------------------------------------------------
def gen_leftovers():
    import time

    out = {}
    # ---- a17: geometry.py:5-56 on full-rank, rank-deficient, pure and identical pairs -----------------
    pairs = []
    for n in (1, 2, 3):
        d = 2**n
        rng = np.random.default_rng(900 + n)
        full_a, full_b = ginibre_state(rng, d), ginibre_state(rng, d)
        low_a, low_b = ginibre_state(rng, d, rank=max(1, d // 2)), ginibre_state(rng, d, rank=max(1, d // 2))
        pure_a, pure_b = ginibre_state(rng, d, rank=1), ginibre_state(rng, d, rank=1)
        near = full_a + 1e-6 * (full_b - full_a)
        pairs += [(full_a, full_b), (full_a, low_a), (low_a, low_b), (pure_a, full_a), (pure_a, pure_b),
                  (pure_a, pure_a), (full_a, near), (np.eye(d) / d, pure_b)]
    out["geo_n_pairs"] = np.array(len(pairs))
    for i, (a, b) in enumerate(pairs):
        out[f"geo{i}_a"], out[f"geo{i}_b"] = a, b
        out[f"geo{i}_hs"] = np.array(float(qp.hs_dst(a, b)))
        out[f"geo{i}_trace"] = np.array(float(qp.trace_dst(qp.Qobj(a), qp.Qobj(b))))
        out[f"geo{i}_if"] = np.array(float(qp.if_dst(a, b)))
    print("   geometry:", len(pairs), "pairs; if_dst of a pure state with itself =", float(out["geo5_if"]))

    # ---- warm_start, state tomography ------------------------------------------------------------------
    for tag, n, first, second in (("W1", 1, 1000, 400), ("W2", 2, 1000, np.arange(1, 10) * 100), ("W3", 3, 5000, 5000)):
        state = qp.Qobj(ginibre_state(np.random.default_rng(70 + n), 2**n))
        np.random.seed(100 + n)
        t = qp.StateTomograph(state)
        t.experiment(first, "proj-set")
        t.experiment(second, "proj-set", warm_start=True)
        out[tag + "_state"] = state.matrix
        out[tag + "_second"] = np.asarray(second)
        out[tag + "_povm"] = t.povm_matrix
        out[tag + "_results"] = t.results
        out[tag + "_nmeas"] = t.n_measurements
        out[tag + "_lin_unphys"] = t.point_estimate("lin", physical=False).matrix
        out[tag + "_lin"] = t.point_estimate("lin").matrix
        out[tag + "_mle"] = t.point_estimate("mle").matrix
        out[tag + "_mle_nit"] = np.array(_CAPTURE["res"].nit)
        # a third round on top (still warm)
        t.experiment(first, "proj-set", warm_start=True)
        out[tag + "_results3"] = t.results
        out[tag + "_povm3"] = t.povm_matrix
        out[tag + "_lin3"] = t.point_estimate("lin").matrix
    # ---- warm_start, process tomography ---------------------------------------------------------------
    for tag, n, ch in (("WP1", 1, qp.channel.depolarizing(0.15, 1)), ("WP2", 2, qp.channel.depolarizing(0.1, 2))):
        np.random.seed(200 + n)
        pt = qp.ProcessTomograph(ch)
        pt.experiment(2000, "proj-set")
        pt.experiment(1000, "proj-set", warm_start=True)
        out[tag + "_results"] = pt.results
        out[tag + "_povm"] = pt.tomographs[0].povm_matrix
        out[tag + "_nmeas"] = pt.tomographs[0].n_measurements
        out[tag + "_choi_raw"] = pt.point_estimate("lifp", cptp=False).choi.matrix
        out[tag + "_choi"] = pt.point_estimate("lifp", cptp=True).choi.matrix
        print(f"   {tag}: warm-started process, results {pt.results.shape}")

    # ---- BootstrapProcessInterval from the reference itself -------------------------------------------
    cls = np.array([0.5, 0.9, 0.95])
    for tag, n, method, n_points, shots in (("BP2lifp", 2, "lifp", 6, 3000), ("BP1lifp", 1, "lifp", 40, 1000),
                                             ("BP1states", 1, "states", 24, 1000), ("BP1pgdb", 1, "pgdb", 3, 1000)):
        t0 = time.time()
        np.random.seed(300 + n)
        pt = qp.ProcessTomograph(qp.channel.depolarizing(0.1, n))
        pt.experiment(shots, "proj-set")
        centre = pt.point_estimate(method)
        out[tag + "_counts0"] = pt.results
        out[tag + "_centre"] = centre.choi.matrix
        # replay of interval.py:673-682 recording every resample
        np.random.seed(5150)
        boot = qp.ProcessTomograph(centre, pt.input_states, pt.dst)
        cs, ds, ch_ = [], [], []
        for _ in range(n_points):
            boot.experiment(pt.tomographs[0].n_measurements, povm=pt.tomographs[0].povm_matrix)
            est = boot.point_estimate(method=method, states_physical=True, states_init="lin", cptp=True)
            cs.append(boot.results.copy())
            ch_.append(est.choi.matrix)
            ds.append(pt.dst(est.choi, centre.choi))
        np.random.seed(5150)
        iv = qp.BootstrapProcessInterval(pt, n_points=n_points, method=method)
        dist, cl = iv(cls)
        assert np.array_equal(np.sort(ds), iv.cl_to_dist.y), "replay differs from the reference class"
        out[tag + "_boot_counts"] = np.stack(cs)
        out[tag + "_boot_choi"] = np.stack(ch_)
        out[tag + "_boot_dist"] = np.array(ds)
        out[tag + "_cl_dist"] = np.array(dist)
        out[tag + "_nmeas"] = pt.tomographs[0].n_measurements
        print(f"   {tag}: quantiles {dist}  ({time.time() - t0:.1f} s)")
    out["conf_levels"] = cls
    save("leftovers", **out)


# --------------------------------------------------------------------------------------
# (f) 'pgdb' at n = 3 (process.py:291-314 is size-generic): what the reference returns on the Q0 data of
# gen_process3, and its loop with the step accepted (the evident intent), three iterations, piece by piece
# --------------------------------------------------------------------------------------
def gen_pgdb3():
    import time as _time
    from quantpy.routines import _mat2vec, _vec2mat
    from quantpy.qobj import fully_mixed

    out = {}
    t0 = _time.time()
    np.random.seed(31)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
    tmg.experiment(10000, "proj-set")
    out["R0_counts"] = tmg.results
    out["R0_input_states"] = np.stack([s_.matrix for s_ in tmg.input_basis.elements])
    ch = tmg.point_estimate("pgdb", n_iter=2)  # the reference as it is (leaves in its first iteration)
    out["R0_returned"] = ch.choi.matrix
    print(f"   reference pgdb(n_iter=2): {_time.time() - t0:.0f} s; returned == start: "
          f"{np.allclose(ch.choi.matrix, fully_mixed(6).matrix)}")
    v = _mat2vec(fully_mixed(6).matrix)
    mu, gamma, tol, cap = 1.5 / 4**3, 0.3, 1e-10, 3
    trace, iterates = [], []
    for it in range(cap):
        probas = tmg._lifp_oper @ v
        grad = -tmg._lifp_oper.T.conj() @ (tmg._unnorm_results / probas)
        direction = tmg._cptp_projection_vec(v - grad / mu) - v
        alpha = 1
        while tmg._nll(v + alpha * direction) - tmg._nll(v) > gamma * alpha * np.dot(direction, grad):
            alpha /= 2
        if it == 0:
            out["R0_it0_probas"] = probas
            out["R0_it0_grad"] = grad
            out["R0_it0_direction"] = direction
            out["R0_it0_dot"] = np.array(np.dot(direction, grad))
        new = v + alpha * direction
        f0, f1 = tmg._nll(v), tmg._nll(new)
        trace.append([f0, f1, alpha])
        v = new
        iterates.append(_vec2mat(v).copy())
        print(f"   step {it}: nll {np.real(f0):.6f} -> {np.real(f1):.6f}, alpha {alpha}  ({_time.time() - t0:.0f} s)")
        if not (f0 - f1 > tol):
            break
    out["R0_conv_cap"] = np.array(cap)
    out["R0_conv_iterates"] = np.stack(iterates)
    out["R0_conv_trace"] = np.array(trace)
    save("pgdb3", **out)


# --------------------------------------------------------------------------------------
# (f) MHMCProcessInterval at n = 3 (interval.py:763-850 is size-generic): two short chains on the Q0 data
# --------------------------------------------------------------------------------------
def gen_mhmc3():
    import time as _time

    out = {}
    t0 = _time.time()
    np.random.seed(31)
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(0.1, 3))
    tmg.experiment(10000, "proj-set")
    ch = tmg.point_estimate("lifp")
    out["M_counts"] = tmg.results
    out["M_channel"] = ch.choi.matrix
    print(f"   lifp + cptp: {_time.time() - t0:.0f} s")
    for key, seed, step, n_points, burn in (("M0", 231, 1e-5, 6, 4), ("M1", 232, 2e-6, 6, 4)):
        np.random.seed(seed)
        iv = qp.MHMCProcessInterval(tmg, n_points=n_points, step=step, burn_steps=burn, return_samples=True)
        dist, cl, rate, mats = iv.setup()
        out[key + "_seed"] = np.array(seed)
        out[key + "_args"] = np.array([n_points, burn])
        out[key + "_step"] = np.array(step)
        out[key + "_dist"] = dist
        out[key + "_rate"] = np.array(rate)
        out[key + "_samples"] = np.stack(mats)
        print(f"   {key}: step {step} acceptance {rate:.3f} dist {dist[:3]}  ({_time.time() - t0:.0f} s)", flush=True)
    save("mhmc3", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["operators", "states", "counts", "chol", "mle", "process", "moment", "bootstrap", "large",
                             "pgdb", "constr", "mhmc", "holder", "leftovers", "process3", "pgdb3", "mhmc3"]
    table = {
        "operators": gen_operators,
        "states": gen_states_and_born,
        "counts": gen_counts_lin,
        "chol": gen_chol_nll,
        "mle": gen_mle,
        "process": gen_process,
        "process3": gen_process3,
        "moment": gen_moment,
        "bootstrap": gen_bootstrap,
        "large": gen_large,
        "pgdb": gen_pgdb,
        "pgdb3": gen_pgdb3,
        "mhmc3": gen_mhmc3,
        "constr": gen_constr,
        "mhmc": gen_mhmc,
        "holder": gen_holder,
        "leftovers": gen_leftovers,
    }
    for w in which:
        print(f"[{w}]")
        table[w]()
    meta = {
        "generated_by": "tests/golden/make_golden.py (imports /root/reference read-only)",
        "python": sys.version.split()[0],
        "numpy": np.__version__,
        "scipy": scipy.__version__,
        "reference_pins": {"python": "^3.9,<3.10", "numpy": "1.23.4", "scipy": "1.9.3"},
    }
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
