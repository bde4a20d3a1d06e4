"""tgcn_amd/coarsening.py (the matching loop runs as host code inside libtgcn_hip.so) against fixtures produced by running the
reference's gcn/coarsening.py (tools/make_golden.py): coarsen() under the same numpy seed, metis() with a given visiting order,
compute_perm, perm_data, perm_adjacency -- graphs, parents and permutations equal entry for entry.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import golden_files, golden_ids, load_golden

FILES = golden_files("coarsen_")


def _A(g):
    return sp.csr_matrix((g["a_val"], g["a_col"], g["a_rowptr"]), shape=(int(g["n"]), int(g["n"])))


@pytest.mark.parametrize("path", FILES, ids=golden_ids(FILES))
def test_coarsen_matches_reference(path):
    from tgcn_amd import coarsening as co
    g = load_golden(path)
    levels = int(g["levels"])
    np.random.seed(int(g["seed"]))
    graphs, perm = co.coarsen(_A(g), levels=levels, self_connections=False)
    assert np.array_equal(np.asarray(perm), g["perm"])
    assert len(graphs) == levels + 1
    for i, G in enumerate(graphs):
        G = G.tocsr()
        G.sort_indices()
        assert G.shape[0] == int(g["g%d_n" % i])
        assert np.array_equal(G.indptr, g["g%d_rowptr" % i]) and np.array_equal(G.indices, g["g%d_col" % i])
        assert np.array_equal(G.data.astype(np.float32), g["g%d_val" % i])
    assert np.array_equal(co.perm_data(g["x"], perm), g["x_perm"])


@pytest.mark.parametrize("path", FILES, ids=golden_ids(FILES))
def test_metis_and_perm_match_reference(path):
    from tgcn_amd import coarsening as co
    g = load_golden(path)
    levels = int(g["levels"])
    graphs, parents = co.metis(_A(g), levels, rid=g["rid"])
    assert len(parents) == levels
    for i, p in enumerate(parents):
        assert np.array_equal(np.asarray(p), g["parents%d" % i])
    perms = co.compute_perm(parents)
    for i, p in enumerate(perms):
        assert np.array_equal(np.asarray(p), g["perms%d" % i])
    # every coarse vertex has exactly two (real or fake) children, siblings adjacent: what gcn_pool / gcn_pool_4 rely on
    for i in range(levels):
        assert len(perms[i]) == 2 * len(perms[i + 1])


def test_compute_perm_known_answer():
    """the reference's import-time assert (gcn/coarsening.py:219-220)"""
    from tgcn_amd.coarsening import compute_perm
    want = [[3, 4, 0, 9, 1, 2, 5, 8, 6, 7, 10, 11], [2, 4, 1, 3, 0, 5], [0, 1, 2]]
    assert compute_perm([np.array([4, 1, 1, 2, 2, 3, 0, 0, 3]), np.array([2, 1, 0, 1, 0])]) == want


def test_perm_adjacency_and_device_data():
    import torch
    from tgcn_amd import coarsening as co
    rng = np.random.default_rng(0)
    A = sp.random(7, 7, 0.4, random_state=1, format="csr", dtype=np.float32)
    A = A + A.T
    idx = [3, 7, 0, 8, 1, 2, 9, 5, 6, 4]
    B = co.perm_adjacency(A, idx).toarray()
    pos = np.argsort(idx)
    full = np.zeros((10, 10), np.float32)
    full[:7, :7] = A.toarray()
    assert np.array_equal(B[np.ix_(pos, pos)], full)
    x = rng.standard_normal((4, 7))
    assert np.array_equal(co.perm_data_device(torch.as_tensor(x), idx).numpy(), co.perm_data(x, idx))
