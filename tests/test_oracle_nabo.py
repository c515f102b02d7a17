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


def test_restatements_on_ties_and_duplicates_hypothesis():
    """Property test of the two libnabo restatements where a k-d tree's arbitrary choices show: duplicated points, coordinates on a
    coarse lattice (many equal split values and equal distances), queries that coincide with targets, clouds of 1-40 points (bucket
    size 8: trees of depth 0-3).  libnabo's own answer is not defined there -- which of two points at the same distance wins follows
    from the order std::nth_element leaves equal keys in (kdtree_opencl / kdtree_cpu build, 1.0.7) -- so the two restatements may name
    different points; what each must keep: the distance it reports is the distance to the point it names, at eps = 0 that distance is
    the exact minimum, and for every eps it is within libnabo's (1 + eps) contract.  The same inputs jittered (no ties left): id,
    squared distance and visited-leaf count agree exactly."""
    from hypothesis import given, settings, strategies as st

    coord = st.integers(min_value=-6, max_value=6).map(lambda v: v * 0.25)
    point = st.tuples(coord, coord, coord)
    differ = [0, 0]

    @settings(max_examples=150, deadline=None)
    @given(st.lists(point, min_size=1, max_size=40), st.lists(point, min_size=1, max_size=12), st.sampled_from([0.0, 0.5, 3.16]),
           st.integers(min_value=0, max_value=3), st.integers(min_value=0, max_value=2 ** 31 - 1))
    def check(tp, qp, eps, dup, seed):
        tgt = np.array(tp, dtype=np.float64)
        if dup:
            tgt = np.concatenate([tgt, tgt[:: max(1, 4 - dup)]])           # exact duplicates of some of the points
        qry = np.concatenate([np.array(qp, dtype=np.float64), tgt[:2]])   # and queries that sit on targets
        for jitter in (False, True):
            t, q = tgt, qry
            if jitter:
                rng = np.random.default_rng(seed)
                t = tgt + rng.uniform(-0.05, 0.05, tgt.shape); q = qry + rng.uniform(-0.05, 0.05, qry.shape)
            tree = nabo.NaboTree(t)
            ids_c, d2_c, leaves_c = cref.nn_nabo(t, q, eps)
            brute = ((q[:, None, :] - t[None, :, :]) ** 2).sum(axis=2)
            leaves_py = 0
            for k, qq in enumerate(q):
                j, d2, lv = tree.knn1(qq, eps)
                leaves_py += lv
                for jj, dd in ((j, d2), (int(ids_c[k]), float(d2_c[k]))):
                    assert dd == brute[k, jj]                                       # the reported distance is the named point's
                    assert np.sqrt(dd) <= (1.0 + eps) * np.sqrt(brute[k].min()) + 1e-12
                    if eps == 0.0:
                        assert dd == brute[k].min()
                if jitter:
                    assert j == ids_c[k] and d2 == d2_c[k], (eps, k, j, ids_c[k])
                else:
                    differ[0] += int(j != ids_c[k]); differ[1] += 1
            if jitter:
                assert leaves_py == leaves_c

    check()
    assert differ[1] > 0
    print(f"\n[nabo ties] the restatements named different (equally far or equally admissible) points for {differ[0]} of {differ[1]} tied queries")
