"""Host-side analysis drop-ins (utils / metrics / kmc_event_rates helpers) vs reference outputs
(fixture F6)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, load

import kmc_event_rates
import metrics
import utils

META = json.load(open(os.path.join(GOLDEN, "metrics_meta.json")))


def _same(a, b):
    if isinstance(b, float):
        return a == pytest.approx(b, rel=1e-13, abs=0) or (a == b)
    return a == b


@pytest.mark.parametrize("name", ["m_L6", "m_L9", "m_L12", "m_empty"])
def test_clusters_and_metrics(name):
    z = load("metrics")
    state = z[name + "_state"].astype(np.int64)
    theta, phi, defects = z[name + "_theta"], z[name + "_phi"], z[name + "_defects"].astype(np.int64)
    clusters, visited = utils.get_clusters(state, theta, phi, theta_threshold=0.5)
    assert np.array_equal(np.asarray(visited), z[name + "_visited"])
    assert [len(c) for c in clusters] == z[name + "_cluster_sizes"].tolist()
    if clusters:
        assert np.array_equal(np.array([c[0] for c in clusters]), z[name + "_cluster_first"])
        assert np.array_equal(np.array([utils.calculate_aspect_ratio(c) for c in clusters]), z[name + "_cluster_ar"])
    m = metrics.compute_metrics(state, theta, phi, defects=defects, W_mask=(state == 1), Re_mask=(state == 2),
                                C_mask=(state == 3), grain_ids=visited, rng_seed=7)
    want = META[name]["full"]
    assert list(m.keys()) == list(metrics.compute_metrics(state * 0, theta, phi).keys())
    assert set(m.keys()) == set(want.keys())
    for k, v in want.items():
        assert _same(m[k], v), (k, m[k], v)
    m2 = metrics.compute_metrics(state, theta, phi)
    for k, v in META[name]["default"].items():
        assert _same(m2[k], v), (k, m2[k], v)
    assert metrics.compute_CET(state, theta, phi) == META[name]["cet"]
    assert bool(metrics.detect_CET_transition(m)) == META[name]["detect"]


def test_run_simulation_smoke_values():
    """run_simulation.py of the reference: 10^3 fake lattice -> 17 keys (run_simulation.py:5-30)."""
    import importlib
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    with redirect_stdout(buf):
        mod = importlib.import_module("run_simulation")
        importlib.reload(mod)
    want = META["run_simulation"]
    assert list(mod.m.keys()) == want["keys"]
    for k, v in want["values"].items():
        assert _same(mod.m[k], v), k
    assert "Keys:" in buf.getvalue() and "Values:" in buf.getvalue()


def test_neighbors_and_misorientation_helpers():
    for key, want in META["neighbors"].items():
        i, j, k, L = map(int, key.split(","))
        got = kmc_event_rates.get_bcc_neighbors(i, j, k, L)
        assert got.dtype == np.int64 and got.tolist() == want
    z = load("metrics")
    got = np.array([kmc_event_rates.compute_misorientation(*row) for row in z["misor_in"]])
    assert np.max(np.abs(got - z["misor_out"])) < 2e-8


def test_dropin_engine_caches_are_bounded():
    """kmc_event_rates / thermal_solver keep device lattices between calls: least recently used first out, close_engines()."""
    import kmc_event_rates
    import thermal_solver

    class Fake:
        def __init__(self):
            self.closed = False

        def close(self):
            self.closed = True
    for mod in (kmc_event_rates, thermal_solver):
        c = mod._EngineCache(2)
        a, b = c.get(8, Fake), c.get(9, Fake)
        assert c.get(8, Fake) is a
        d = c.get(10, Fake)
        assert b.closed and not a.closed and len(c) == 2
        c.close()
        assert a.closed and d.closed and len(c) == 0
        assert isinstance(mod._engines, mod._EngineCache) and callable(mod.close_engines)
