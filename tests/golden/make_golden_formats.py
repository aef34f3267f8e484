#!/usr/bin/env python3
"""Golden vectors for the on-disk instance formats, produced by RUNNING THE REFERENCE's converters
(scripts/min2mcf.py::parse_min_file, scripts/mnist2ot.py::normalize_and_amplify / create_cost_matrix /
make_opt_transport_instances) on small inputs written here.  Build-container only (needs /root/reference).
``idx2numpy`` is not installed; mnist2ot.py imports it at module top, so an empty placeholder module of
that name lets the import resolve -- its reader is not used (the images below are generated arrays).

Outputs (data only): g6_dimacs_small.min (input text), g6_formats.npz, g6_summaries.json.
Usage:  python tests/golden/make_golden_formats.py
"""
import importlib.util
import io
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
sys.modules.setdefault("idx2numpy", types.ModuleType("idx2numpy"))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MIN_TEXT = """c small min-cost-flow instance: 6 nodes, 11 arcs, one self loop, one parallel pair,
c one node line repeated (the later one wins), one arc with a non-zero lower bound (ignored)
p min 6 11
n 1 7
n 6 -7
n 3 2
n 3 0
a 1 2 0 5 3
a 1 3 0 4 1
a 2 4 0 3 2
a 3 4 0 6 7
a 2 3 1 2 1
a 4 5 0 9 2
a 5 6 0 9 1
a 4 6 0 2 8
a 4 4 0 1 5
a 1 2 0 1 9
a 3 5 0 2 4
"""


def main():
    min_path = os.path.join(HERE, "g6_dimacs_small.min")
    with open(min_path, "w") as fh:
        fh.write(MIN_TEXT)
    m2m = load("/root/reference/scripts/min2mcf.py", "ref_min2mcf")
    mcf = m2m.parse_min_file(min_path, "g6")
    A = mcf.A.tocsr()
    A.sort_indices()

    m2o = load("/root/reference/scripts/mnist2ot.py", "ref_mnist2ot")
    rng = np.random.default_rng(6)
    imgs = rng.integers(0, 256, size=(4, 28, 28)).astype(np.uint8)
    imgs[rng.random(imgs.shape) < 0.8] = 0                     # MNIST-like: mostly background
    norm = [m2o.normalize_and_amplify(im, 1) for im in imgs]
    cost = m2o.create_cost_matrix(1)
    with redirect_stdout(io.StringIO()):
        ots = m2o.make_opt_transport_instances(norm, cost, 1)
    small = m2o.normalize_and_amplify(imgs[0][:3, :4], 2)

    np.savez_compressed(
        os.path.join(HERE, "g6_formats.npz"),
        mcf_indptr=A.indptr, mcf_indices=A.indices, mcf_data=np.asarray(A.data, dtype=np.float64),
        mcf_shape=np.array(A.shape), mcf_b=mcf.b, mcf_c=mcf.c, mcf_u=mcf.u,
        images=imgs, cost_sample=cost[::37, ::41], cost_sum=np.array([cost.sum()]),
        amp_in=imgs[0][:3, :4], amp_out=small,
        ot0_s=ots[0].s, ot0_d=ots[0].d, ot0_M=ots[0].M, ot1_s=ots[1].s, ot1_d=ots[1].d, ot1_M=ots[1].M,
        ot_names=np.array([o.name for o in ots]))
    print("wrote g6_dimacs_small.min, g6_formats.npz;", A.shape, len(ots), "OT instances")
    summaries()


def summaries():
    """Summary statistics of the reference's analysis (visualization.py:181-195 called as is; :415 is an inline
    pandas expression, evaluated here on a small frame exactly as written there)."""
    import json
    import pandas as pd
    # visualization.py imports filehandling.py, which imports gurobipy and evaluates get_project_root() at import
    # time: an empty placeholder module lets the import statement resolve (no solver behaviour is provided), and
    # the import is made from a directory named like the project, which is all that function looks for
    for _name in ("gurobipy", "cplex", "mosek", "mosek.fusion"):
        _m = types.ModuleType(_name)
        for _a in ("GRB", "Model", "Cplex"):
            setattr(_m, _a, type(_a, (), {}))
        sys.modules.setdefault(_name, _m)
    import tempfile
    _cwd = os.getcwd()
    _tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(_tmp, "smart-crossover"))
    os.chdir(os.path.join(_tmp, "smart-crossover"))
    try:
        import smart_crossover.visualization as viz
    finally:
        os.chdir(_cwd)
    rng = np.random.default_rng(16)
    ptime = rng.uniform(0.5, 400.0, 12)
    cross = rng.uniform(0.5, 400.0, 12)
    ptime[[2, 7]] = np.nan
    cross[[5]] = np.nan
    ptime[9], cross[9] = np.nan, np.nan
    df = pd.DataFrame({"Ptime": ptime.copy(), "Crossover(ori)": cross.copy()})
    with redirect_stdout(io.StringIO()) as text:
        avg = viz.calculate_average_improvement_lp(df, "ptb")
    names = [f"{g}_{k}" for g in ("goto", "netgen", "road") for k in range(4)]
    net = pd.DataFrame({"grb_runtime": rng.uniform(0.0, 30.0, 12), "cnet_runtime": rng.uniform(0.0, 30.0, 12)}, index=names)
    net["group"] = net.index.str.split("_").str[0]
    grouped = net.groupby("group").agg(lambda x: np.exp(np.log(x + 0.01).mean())).round(2)
    out = {"ptime": [None if np.isnan(v) else float(v) for v in ptime],
           "crossover_ori": [None if np.isnan(v) else float(v) for v in cross],
           "averages": [None if (isinstance(v, float) and np.isnan(v)) else float(v) for v in avg],
           "printed": text.getvalue().strip(),
           "net_rows": {n: {c: float(net.loc[n, c]) for c in ("grb_runtime", "cnet_runtime")} for n in names},
           "net_grouped": {g: {c: float(grouped.loc[g, c]) for c in ("grb_runtime", "cnet_runtime")} for g in grouped.index}}
    with open(os.path.join(HERE, "g6_summaries.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote g6_summaries.json")


if __name__ == "__main__":
    main()
