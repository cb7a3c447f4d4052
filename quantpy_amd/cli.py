"""Command-line front ends with the reference's JSON wire format (reference scripts/state_interval.py,
scripts/process_interval.py, input.json): experimental counts in, point estimate and Hilbert-Schmidt
confidence radii out.

Input keys: `povm_matrix` (S, K, 4^n), `outcomes` (state: (S, K); process: (D, S, K)), `input_states`
(process: D Bloch vectors), `conf_levels`; optional `target_state` / `target_process`.  Output keys:
`state` / `process` (Bloch vector of the unconstrained linear-inversion estimate), `hs_radius`.
The fidelity bounds the reference adds when a target is given come from a cvxopt SOCP
(MomentFidelity*Interval) and are not produced; `hs_radius` is the same MomentInterval either way.
"""
import json
import sys
from argparse import ArgumentParser
from pprint import pprint

import numpy as np


def _parser():
    parser = ArgumentParser()
    parser.add_argument("-i", "--input", type=str, required=True, help="path to input data file")
    parser.add_argument("-o", "--output", type=str, default=None, help="path to output file")
    parser.add_argument("--no-ci", action="store_true", default=False, help="removes confidence intervals")
    return parser


def _emit(output, path):
    if path:
        with open(path, "w") as fp:
            json.dump(output, fp, indent=4)
    else:
        pprint(output)


def _radius(qp, tmg, data, output, target_key):
    if target_key in data:
        print(f"note: `{target_key}` given -- fidelity bounds need the reference's cvxopt SOCP and are skipped",
              file=sys.stderr)
    interval = qp.MomentInterval(tmg)
    interval.setup()
    output["hs_radius"] = list(interval.cl_to_dist(data["conf_levels"]))


def state_interval(argv=None):
    import quantpy_amd as qp

    args = _parser().parse_args(argv)
    with open(args.input) as fp:
        data = json.load(fp)
    results = np.asarray(data["outcomes"])
    povm_matrix = np.asarray(data["povm_matrix"], dtype=np.float64)
    n_qubits = int(np.log2(povm_matrix.shape[-1]) / 2)
    tmg = qp.StateTomograph(qp.qobj.fully_mixed(n_qubits))
    tmg.povm_matrix = qp.generate_measurement_matrix(povm_matrix, n_qubits)
    tmg.results = results  # the setter derives the shots per setting from the counts
    output = {"state": list(tmg.point_estimate(physical=False).bloch)}
    if not args.no_ci:
        _radius(qp, tmg, data, output, "target_state")
    _emit(output, args.output)
    return output


def process_interval(argv=None):
    import quantpy_amd as qp

    args = _parser().parse_args(argv)
    with open(args.input) as fp:
        data = json.load(fp)
    results = np.asarray(data["outcomes"])
    povm_matrix = np.asarray(data["povm_matrix"], dtype=np.float64)
    n_qubits = int(np.log2(povm_matrix.shape[-1]) / 2)
    inputs = [qp.Qobj(bloch) for bloch in data["input_states"]]
    tmg = qp.ProcessTomograph(qp.channel.depolarizing(n_qubits=n_qubits), input_states=inputs)
    povm = qp.generate_measurement_matrix(povm_matrix, n_qubits)
    tmg.tomographs = []
    for state, counts in zip(tmg.input_basis.elements, results):
        t = qp.StateTomograph(tmg.channel.transform(state))
        t.povm_matrix = povm
        t.results = np.asarray(counts)
        tmg.tomographs.append(t)
    output = {"process": list(tmg.point_estimate(cptp=False).choi.bloch)}
    if not args.no_ci:
        _radius(qp, tmg, data, output, "target_process")
    _emit(output, args.output)
    return output
