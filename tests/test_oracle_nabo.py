"""The reference's one approximation on the ICP path: libnabo's eps = 3.16 nearest neighbour
(/root/reference/registrators/icp_fast.cc:174).  libnabo is not vendored; oracle/nabo.py and the nabo_* part of
oracle/csrc/smref_icp.c restate its KDTREE_LINEAR_HEAP tree and search.  These tests (1) check the two restatements
against each other and against the (1 + eps) contract, and (2) QUANTIFY what the exact search of the GPU path (and of the
default oracle) changes in a whole IcpFast::Align -- the number every "within tolerance of the reference" sentence in
DESIGN.md rests on."""
import numpy as np
import pytest

from oracle import cref, nabo
from oracle import icp_fast as onp
from staticmapping_amd import synth


def _tie_free_cloud(n, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(0, 5, (n, 3)) * [4, 3, 0.3]


def test_python_and_c_nabo_restatements_agree():
    tgt = _tie_free_cloud(700, 1)
    qry = _tie_free_cloud(300, 2)
    tree = nabo.NaboTree(tgt)
    for eps in (0.0, 0.5, 3.16):
        ids_c, d2_c, leaves_c = cref.nn_nabo(tgt, qry, eps)
        leaves_py = 0
        for k, q in enumerate(qry):
            j, d2, lv = tree.knn1(q, eps)
            leaves_py += lv
            assert j == ids_c[k] and d2 == d2_c[k], (eps, k)
        assert leaves_py == leaves_c


def test_nabo_eps0_is_the_exact_neighbour_and_eps_keeps_its_contract():
    tgt = _tie_free_cloud(20000, 3)
    qry = _tie_free_cloud(20000, 4) * 1.05
    ids_x, d2_x = cref.nn(tgt, qry)
    ids_0, d2_0, leaves_0 = cref.nn_nabo(tgt, qry, 0.0)
    assert np.array_equal(ids_x, ids_0) and np.array_equal(d2_x, d2_0)
    ids_e, d2_e, leaves_e = cref.nn_nabo(tgt, qry, 3.16)
    ratio = np.sqrt(d2_e) / np.sqrt(d2_x)
    assert ratio.min() >= 1.0 and ratio.max() <= 1.0 + 3.16 + 1e-12       # libnabo's documented guarantee
    assert leaves_e < leaves_0                                              # that is what eps buys
    assert 0.02 < (ids_e != ids_x).mean() < 0.6                             # and what it costs: many matches are not the nearest


CASES = [("cfg1", 5000), ("cfg2", 20000), ("cfg2", 120000)]


@pytest.mark.parametrize("name,n_points", CASES)
def test_exact_search_vs_reference_eps_search_whole_align(name, n_points, capsys):
    """Deviation of IcpFast::Align with the exact neighbour (GPU path, default oracle) from the same Align with the
    reference's eps = 3.16 libnabo search, and both against the known motion.  Measured (and asserted as a band so a
    change of either restatement shows up): the eps search moves the result by a few millimetres -- MORE than the
    1e-3 m north-star tolerance -- and the exact search is the one closer to the truth."""
    if name == "cfg1":
        tgt, src, T = synth.three_planes_pair(n_points, seed=1)
        guess = np.eye(4)
    else:
        tgt, src, T = synth.scan_pair("cfg2", n_points=n_points)
        guess = synth.make_pose(t=(0.6, 0.0, 0.0))
    q, n, _ = cref.calculate_normals(tgt[:, :3].astype(np.float64))
    ok = np.isfinite(n).all(axis=1)
    q, n = q[ok], n[ok]
    s = src[:, :3].astype(np.float64)
    exact = cref.icp_fast_align(s, q, n, guess=guess)
    approx = cref.icp_fast_align(s, q, n, guess=guess, nn_eps=3.16)
    da, dt = onp.se3_error(exact["result"], approx["result"])
    ea, et = onp.se3_error(exact["result"], T)
    aa, at = onp.se3_error(approx["result"], T)
    with capsys.disabled():
        print(f"\n[{name} {n_points}] exact vs eps=3.16: {da:.2e} rad {dt:.2e} m (iterations {exact['iterations']} / "
              f"{approx['iterations']}); vs truth: exact {ea:.2e} rad {et:.2e} m, eps {aa:.2e} rad {at:.2e} m")
    assert 1e-6 < da < 2e-3 and 1e-4 < dt < 2e-2
    if name == "cfg2":
        assert et < at                       # the exact search lands closer to the known motion
