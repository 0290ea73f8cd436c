"""Generates tests/golden/oracle_golden.json from the ORACLE (oracle/lfpsqp_ref.py).

Provenance: the reference (ksil/LFPSQP.jl) is Julia-only and cannot run in the build image, so these
vectors are NOT outputs of the reference itself -- they freeze the oracle's outputs (which are pinned on
the README Rosenbrock known answer and the reference's test properties, FINDINGS.md §3) on small seeded
instances of BASELINE configs 1-4, so that (a) the oracle cannot drift silently and (b) the GPU path is
checked against committed data.  The one reference-generated golden is README.md:31-36 (Rosenbrock),
asserted directly in tests/test_oracle_reference_properties.py.

    python tests/golden/make_golden.py        # rewrites oracle_golden.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lfpsqp_ref as R  # noqa: E402
from oracle import synth  # noqa: E402
from tests.helpers import DiagOpRef  # noqa: E402
from tests.test_oracle_reference_properties import rosenbrock  # noqa: E402

OFF = dict(disp=R.DisplayOption.off)


def trace_summary(tr):
    return [{k: (None if t.get(k) is None else (float(t[k]) if isinstance(t[k], float) else int(t[k])))
             for k in ("tn_iter", "steptype", "mtype", "retract_iter1", "retract_iter2", "ls_flag", "rank")} |
            {"alpha": t.get("alpha"), "x_norm": float(np.linalg.norm(t["x"])), "fval": float(t["fval"])} for t in tr]


def pack(x, obj, lam, ti, tr):
    return dict(iters=ti.iter, condition=ti.condition.name, obj_values=[float(v) for v in obj], lam=[float(v) for v in lam],
                x_head=[float(v) for v in x[:8]], x_norm=float(np.linalg.norm(x)), x_sum=float(np.sum(x)), trace=trace_summary(tr))


def main():
    G = {}
    f, dv = rosenbrock()
    tr = []
    x, obj, lam, ti = R.optimize(f, np.zeros(2), R.LFPSQPParams(**OFF), derivatives=dv, trace=tr)
    G["config1_rosenbrock"] = pack(x, obj, lam, ti, tr)
    for tag, dpr in (("nr", False), ("pp", True)):
        prob, x0 = synth.config2(50)
        tr = []
        out = R.optimize(prob.f, prob.grad_, prob.c_, prob.jac_, prob.hess_lag_vec_, x0, None, None, 1,
                         R.LFPSQPParams(do_project_retract=dpr, **OFF), trace=tr)
        G[f"config2_n50_{tag}"] = pack(*out, tr)
        prob, x0 = synth.config3(2000, 8)
        tr = []
        out = R.optimize(prob.f, prob.grad_, prob.c_, prob.jac_, prob.hess_lag_vec_, x0, None, None, 8,
                         R.LFPSQPParams(do_project_retract=dpr, **OFF), trace=tr)
        G[f"config3_n2000_m8_{tag}"] = pack(*out, tr)
    P = synth.BallBoxProblem(400, 6)
    tr = []
    out = R.optimize(P.f, P.c_, P.d_, P.x0, P.xl, P.xu, P.m, P.p, R.LFPSQPParams(do_project_retract=False, **OFF),
                     derivatives=P.derivatives(), trace=tr)
    G["config4_n400_m6_nr"] = pack(*out, tr)
    # projcg on the bench workload shape (hash basis, A = diag(5 + 4u), b = u)
    n, m = 1000, 10
    U, _ = np.linalg.qr(synth.hash_matrix(1, n, m))
    a = 4.0 * synth.hash_vector(3, n) + 5.0
    b = synth.hash_vector(4, n)
    runs = {}
    for e in (6, 10, 14):
        x, lam = np.zeros(n), np.zeros(m)
        i, nr = R.projcg_(x, lam, DiagOpRef(a), np.asfortranarray(U), b, np.zeros(m), tol=10.0 ** -e)
        runs[f"tol1e-{e}"] = dict(iters=i, nr=nr, x_norm=float(np.linalg.norm(x)), x_head=[float(v) for v in x[:8]],
                                  lam=[float(v) for v in lam])
    G["projcg_n1000_m10"] = runs
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.json"), "w") as fh:
        json.dump(G, fh, indent=1)
    print("wrote", len(G), "entries")


if __name__ == "__main__":
    main()
