"""Device-side analysis helpers: grain clustering, species counts, sparse
carbon-site queries -- against the host implementations (which are pinned to the reference by
tests/test_host_metrics.py) and the reference fixtures."""
import numpy as np
import pytest

from helpers import load, random_lattice

pytestmark = pytest.mark.gpu


def _host_clusters(state, theta, phi):
    import utils
    clusters, visited = utils.get_clusters(state, theta, phi, theta_threshold=0.5)
    return clusters, np.asarray(visited)


@pytest.mark.parametrize("name", ["m_L6", "m_L9", "m_L12", "m_empty"])
def test_cluster_vs_reference_fixture(name):
    import cetkmc
    z = load("metrics")
    state = z[name + "_state"].astype(np.int64)
    theta, phi = z[name + "_theta"], z[name + "_phi"]
    L = state.shape[0]
    e = cetkmc.Engine(L)
    e.upload(state, theta, phi, np.full(state.shape, 3000.0), state * 0)
    cl = e.clusters(0.5, labels=True)
    assert np.array_equal(cl["labels"], z[name + "_visited"])
    assert cl["size"].tolist() == z[name + "_cluster_sizes"].tolist()
    if len(cl["size"]):
        assert np.array_equal(cl["first"], z[name + "_cluster_first"])
        dims = cl["bbox"][:, 3:] - cl["bbox"][:, :3] + 1
        ar = dims.max(axis=1) / np.maximum(dims.min(axis=1), 1)
        assert np.array_equal(ar, z[name + "_cluster_ar"])


@pytest.mark.parametrize("L,seed,fill,npal", [(20, 1, 0.4, 3), (33, 2, 0.15, 5), (48, 3, 0.6, 2)])
def test_cluster_vs_host_random(L, seed, fill, npal):
    import cetkmc
    import metrics
    rs = np.random.RandomState(seed)
    state = np.zeros((L, L, L), np.int64)
    occ = rs.random_sample((L, L, L)) < fill
    state[occ] = rs.choice([1, 2, 3, 4], size=int(occ.sum()), p=[0.6, 0.15, 0.2, 0.05])
    pal_t, pal_p = rs.uniform(0, np.pi, npal), rs.uniform(0, 2 * np.pi, npal)
    pick = rs.randint(0, npal, (L, L, L))
    theta = np.where(state != 0, pal_t[pick], 0.0)
    phi = np.where(state != 0, pal_p[pick], 0.0)
    clusters, visited = _host_clusters(state, theta, phi)
    e = cetkmc.Engine(L)
    e.upload(state, theta, phi, np.full(state.shape, 3000.0), state * 0)
    cl = e.clusters(0.5, labels=True)
    assert np.array_equal(cl["labels"], visited)
    assert cl["size"].tolist() == [len(c) for c in clusters]
    m_host = metrics.compute_metrics(state, theta, phi, defects=(state == 3).astype(int))
    m_dev = metrics.compute_metrics_device(e, state.size, defects_count=int((state == 3).sum()))
    assert list(m_host.keys()) == list(m_dev.keys())
    for k in m_host:
        assert m_host[k] == m_dev[k], k


def test_sparse_site_queries_and_defect_refresh():
    import cetkmc
    import defects
    L = 17
    state, theta, phi, T, dmask = random_lattice(L, 4, fill=0.5)
    e = cetkmc.Engine(L, n_slabs=2)
    e.upload(state, theta, phi, T, dmask)
    assert e.species_counts().tolist() == [int((state == s).sum()) for s in range(5)] + [0]
    idx, Tv = e.gather_species(3)
    want = np.flatnonzero(state.ravel() == 3)
    assert np.array_equal(idx, want) and np.array_equal(Tv, T.ravel()[want])
    np.random.seed(5)
    mask_ref, dens_ref = defects.introduce_defects(state, state, T)
    np.random.seed(5)
    n_flag, dens = defects.refresh_defects_device(e)
    assert n_flag == int(mask_ref.sum()) and dens == dens_ref
    assert np.array_equal(e.download(state=False, theta=False, phi=False, T=False, defects=True)["defects"], mask_ref)
    assert np.random.random() == np.random.RandomState(5).random_sample(len(want) + 1)[-1]   # same stream position
