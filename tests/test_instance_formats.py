"""On-disk instance formats (smart_crossover/instances.py) against goldens produced by the reference's own
converters (tests/golden/make_golden_formats.py): DIMACS .min -> MinCostFlow, MNIST idx images -> OptTransport."""
import os
import struct

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN as GOLDEN_DIR, bits_equal
from smart_crossover import instances as I


@pytest.fixture(scope="module")
def g6():
    return np.load(os.path.join(GOLDEN_DIR, "g6_formats.npz"), allow_pickle=False)


def test_dimacs_min_matches_reference_converter(g6):
    mcf = I.read_dimacs_min(os.path.join(GOLDEN_DIR, "g6_dimacs_small.min"), "g6")
    A = sp.csr_matrix(mcf.A)
    A.sort_indices()
    assert tuple(g6["mcf_shape"]) == A.shape == (6, 11)
    assert np.array_equal(A.indptr, g6["mcf_indptr"]) and np.array_equal(A.indices, g6["mcf_indices"])
    assert np.array_equal(A.data, g6["mcf_data"])
    assert np.array_equal(mcf.b, g6["mcf_b"]) and mcf.b[2] == 0.0          # repeated node line: the later one wins
    assert np.array_equal(mcf.c, g6["mcf_c"]) and np.array_equal(mcf.u, g6["mcf_u"])
    col = A.tocsc()
    assert col[:, 8].nnz == 1 and col[3, 8] == 1.0                          # the self loop keeps +1 only
    assert mcf.name == "g6"


def test_dimacs_min_errors(tmp_path):
    p = tmp_path / "bad.min"
    p.write_text("c nothing here\na 1 2 0 1 1\n")
    with pytest.raises(ValueError):
        I.read_dimacs_min(str(p))
    p.write_text("p min 2 2\nn 1 1\nn 2 -1\na 1 2 0 1 1\n")
    with pytest.raises(ValueError):
        I.read_dimacs_min(str(p))


def test_idx_reader_roundtrip_and_errors(tmp_path, g6):
    imgs = g6["images"]
    p = tmp_path / "train-images-idx3-ubyte"
    p.write_bytes(struct.pack(">IIII", 0x00000803, *imgs.shape) + imgs.tobytes())
    got = I.read_idx_images(str(p))
    assert got.dtype == np.uint8 and np.array_equal(got, imgs)
    p.write_bytes(struct.pack(">IIII", 0x00000801, *imgs.shape) + imgs.tobytes())
    with pytest.raises(ValueError):
        I.read_idx_images(str(p))
    p.write_bytes(struct.pack(">IIII", 0x00000803, *imgs.shape) + imgs.tobytes()[:-5])
    with pytest.raises(ValueError):
        I.read_idx_images(str(p))


def test_ot_instances_match_reference_converter(g6):
    assert bits_equal(I.amplify_and_normalise(g6["amp_in"], 2), g6["amp_out"])
    cost = I.manhattan_cost(28)
    assert cost.shape == (784, 784) and int(cost.sum()) == int(g6["cost_sum"][0])
    assert np.array_equal(cost[::37, ::41], g6["cost_sample"])
    norm = [I.amplify_and_normalise(im, 1) for im in g6["images"]]
    ots = I.ot_instances_from_images(norm, cost, 1)
    assert [o.name for o in ots] == list(g6["ot_names"])
    for k, o in enumerate(ots):
        assert bits_equal(o.s, g6[f"ot{k}_s"]) and bits_equal(o.d, g6[f"ot{k}_d"])
        assert np.array_equal(o.M, g6[f"ot{k}_M"])
        assert abs(o.s.sum() - 1.0) < 1e-12 and abs(o.d.sum() - 1.0) < 1e-12


def test_result_pickles_round_trip(tmp_path):
    from smart_crossover import results as R
    payload = {"goto_8_13a": {"grb_runtime": 1.5, "x": np.arange(4)}, "note": "anything picklable"}
    R.write_results_to_pickle(payload, "ot/run1.pickle", root=str(tmp_path))
    assert os.path.exists(tmp_path / "results" / "ot" / "run1.pickle")      # <root>/results/<path>, filehandling.py:103,109
    back = R.read_results_from_pickle("ot/run1.pickle", root=str(tmp_path))
    assert back["note"] == payload["note"] and np.array_equal(back["goto_8_13a"]["x"], payload["goto_8_13a"]["x"])
    with pytest.raises(FileNotFoundError):
        R.read_results_from_pickle("missing.pickle", root=str(tmp_path))


def test_summary_statistics_match_the_reference_analysis():
    """visualization.py:181-195 (called as is by the golden generator) and the grouped shifted geometric mean of
    :415/:428 (the reference's inline pandas expression evaluated on the same rows)."""
    import json
    from smart_crossover import results as R
    g = json.load(open(os.path.join(GOLDEN_DIR, "g6_summaries.json")))
    nan = float("nan")
    ptime = [nan if v is None else v for v in g["ptime"]]
    cross = [nan if v is None else v for v in g["crossover_ori"]]
    avg, avg_solver, avg_par, improved = R.average_improvement_lp(ptime, cross)
    assert (avg, avg_solver, avg_par) == pytest.approx(tuple(g["averages"]), rel=1e-12)
    assert g["printed"] == f"Number of improved problems: {improved}"
    assert R.grouped_geometric_mean(g["net_rows"]) == g["net_grouped"]
