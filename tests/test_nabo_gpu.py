"""nn_mode = NABO: the reference's own nearest-neighbour search on the device -- libnabo 1.0.7's KDTREE_LINEAR_HEAP tree
rebuilt per Align and its epsilon = 3.16 approximate knn (/root/reference/registrators/icp_fast.cc:169-180, 464-467) --
against the restatement of that library in oracle/csrc/smref_icp.c (nabo_*, cross-checked with oracle/nabo.py).
The device works in float32 on the centred clouds, libnabo in float64: a pruning test within a float ulp of its threshold
can fall the other way, so neighbour ids are compared as a fraction and whole alignments within the north-star tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _find_closests(case, eps, n_points):
    import staticmapping_amd as sm
    from oracle import cref
    m = sm.IcpFastHip(max_source_points=n_points, max_target_points=len(case["q"]), nn_mode=sm.NN_NABO, nn_epsilon=eps)
    m.set_input_source(case["src"]); m.set_input_target(case["q"], case["n"])
    ids, d2 = m.find_closests(case["guess"], len(case["src"]))
    m.close()
    mean = case["q"].mean(axis=0)
    G = case["guess"]
    moved = case["src"][:, :3].astype(np.float64) @ G[:3, :3].T + G[:3, 3]
    ids_o, d2_o, _ = cref.nn_nabo(case["q"] - mean, moved - mean, eps)
    return ids, d2, ids_o, d2_o


@pytest.mark.parametrize("eps", [3.16, 0.5, 0.0])
def test_find_closests_equals_the_libnabo_restatement(velo20k, eps):
    ids, d2, ids_o, d2_o = _find_closests(velo20k, eps, 20000)
    same = ids == ids_o
    assert same.mean() > 0.999, same.mean()
    assert np.allclose(d2[same], d2_o[same], rtol=2e-4, atol=1e-9)
    # where they differ both are legal answers of the (1 + eps) contract: never more than (1 + eps) x the true distance
    from oracle import cref
    mean = velo20k["q"].mean(axis=0)
    G = velo20k["guess"]
    moved = velo20k["src"][:, :3].astype(np.float64) @ G[:3, :3].T + G[:3, 3]
    _, d2_x = cref.nn(velo20k["q"] - mean, moved - mean)
    assert (np.sqrt(d2) <= (1.0 + eps) * np.sqrt(d2_x) * (1 + 1e-4) + 1e-6).all()


def test_eps0_through_the_tree_is_the_exact_neighbour(velo20k):
    import staticmapping_amd as sm
    ids, d2, ids_o, d2_o = _find_closests(velo20k, 0.0, 20000)
    m = sm.IcpFastHip(max_source_points=20000, max_target_points=len(velo20k["q"]))          # the exact grid search
    m.set_input_source(velo20k["src"]); m.set_input_target(velo20k["q"], velo20k["n"])
    ids_g, d2_g = m.find_closests(velo20k["guess"], 20000)
    m.close()
    assert (ids == ids_g).mean() > 0.9999
    assert np.array_equal(d2[ids == ids_g], d2_g[ids == ids_g])


@pytest.mark.parametrize("fixture_name,n_points", [("velo20k", 20000), ("cfg2", 120000)])
def test_whole_align_matches_the_reference_search_semantics(request, fixture_name, n_points, capsys):
    """IcpFast::Align with the reference's epsilon = 3.16 search: device vs the C restatement with nn_eps = 3.16 -- the parity
    the exact search cannot have (it sits 3-4 mm away from this result)."""
    import staticmapping_amd as sm
    from oracle import cref
    case = request.getfixturevalue(fixture_name)
    m = sm.IcpFastHip(max_source_points=n_points, max_target_points=len(case["q"]), nn_mode=sm.NN_NABO, nn_epsilon=3.16, max_iteration=100)
    m.set_input_source(case["src"]); m.set_input_target(case["q"], case["n"])
    ok, R = m.align(case["guess"])
    st = m.last_stats[0]
    score = m.get_fitness_score()
    m.set_options(nn_mode=sm.NN_GRID)
    ok, Rx = m.align(case["guess"])
    m.close()
    src = case["src"][:, :3].astype(np.float64)
    ref = cref.icp_fast_align(src, case["q"], case["n"], guess=case["guess"], nn_eps=3.16, nthreads=cref.usable_cores())
    da, dt = sm.se3_error(R, ref["result"])
    dxa, dxt = sm.se3_error(Rx, ref["result"])
    with capsys.disabled():
        print(f"\\n[{fixture_name}] device libnabo-mode vs eps = 3.16 oracle: {da:.2e} rad {dt:.2e} m (iterations {st['iterations']} / {ref['iterations']}); "
              f"device exact mode vs the same oracle: {dxa:.2e} rad {dxt:.2e} m")
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert st["iterations"] == ref["iterations"]
    assert abs(score - ref["score"]) < 1e-4


def test_tree_build_survives_ties_and_tiny_clouds():
    """Lattice targets (every coordinate value shared by many points: ties on every median) and clouds of a few points."""
    import staticmapping_amd as sm
    from oracle import cref
    rng = np.random.default_rng(3)
    g = np.stack(np.meshgrid(np.arange(12.0), np.arange(9.0), np.arange(5.0), indexing="ij"), axis=-1).reshape(-1, 3) * 0.5
    nrm = np.tile([0.0, 0.0, 1.0], (len(g), 1))
    qry = (rng.uniform(-0.5, 6.0, size=(4000, 3)) * [1, 0.75, 0.4]).astype(np.float32)
    for tgt in (g, g[:7], g[:9], g[:1]):
        m = sm.IcpFastHip(max_source_points=4096, max_target_points=1024, nn_mode=sm.NN_NABO, nn_epsilon=0.0)
        m.set_input_source(qry); m.set_input_target(tgt, nrm[:len(tgt)])
        ids, d2 = m.find_closests(np.eye(4), len(qry))
        m.close()
        _, d2_x = cref.nn(tgt, qry.astype(np.float64))
        assert (ids >= 0).all() and (ids < len(tgt)).all()
        assert np.allclose(d2, d2_x, rtol=1e-4, atol=1e-8)           # eps = 0: the exact distance, whichever of the tied points
        assert np.allclose(np.linalg.norm(qry - tgt[ids], axis=1) ** 2, d2, rtol=1e-3, atol=1e-6)


def test_device_search_on_duplicates_keeps_libnabo_contract():
    """Targets with exact duplicates and lattice ties, eps = 0 and the reference's 3.16: which of the equally admissible points libnabo
    names is not defined (std::nth_element's order of equal keys), so the device tree, the C and the Python restatement may differ
    there (tests/test_oracle_nabo.py counts how often); what the device search must keep is what libnabo guarantees: the distance it
    reports belongs to the point it names, is the exact minimum at eps = 0 and within (1 + eps) of it otherwise."""
    import staticmapping_amd as sm
    from oracle import cref
    rng = np.random.default_rng(11)
    g = np.stack(np.meshgrid(np.arange(10.0), np.arange(8.0), np.arange(4.0), indexing="ij"), axis=-1).reshape(-1, 3) * 0.25
    tgt = np.concatenate([g, g[::3], g[::7], g[:5]])                            # duplicates, some threefold
    nrm = np.tile([0.0, 0.0, 1.0], (len(tgt), 1))
    qry = np.concatenate([rng.uniform(-0.3, 2.6, size=(3000, 3)) * [1, 0.8, 0.4], g[::5]]).astype(np.float32)   # and queries on targets
    _, d2_x = cref.nn(tgt, qry.astype(np.float64))
    for eps in (0.0, 3.16):
        m = sm.IcpFastHip(max_source_points=4096, max_target_points=1024, nn_mode=sm.NN_NABO, nn_epsilon=eps)
        m.set_input_source(qry); m.set_input_target(tgt, nrm)
        ids, d2 = m.find_closests(np.eye(4), len(qry))
        m.close()
        assert (ids >= 0).all() and (ids < len(tgt)).all()
        named = np.linalg.norm(qry.astype(np.float64) - tgt[ids], axis=1) ** 2
        assert np.allclose(named, d2, rtol=1e-3, atol=1e-6)
        assert (np.sqrt(named) <= (1.0 + eps) * np.sqrt(d2_x) * (1 + 1e-4) + 1e-4).all()
        if eps == 0.0:
            assert np.allclose(d2, d2_x, rtol=1e-4, atol=1e-8)


def test_large_target_tree(capsys):
    """A 150 k-point target: levels with more than 8192 segments (1-bit radix passes, fill counters in global memory)."""
    import staticmapping_amd as sm
    from oracle import cref
    rng = np.random.default_rng(11)
    tgt = rng.normal(0, 1, (150_000, 3)) * [40, 30, 2.0]
    nrm = np.tile([0.0, 0.0, 1.0], (len(tgt), 1))
    qry = (rng.normal(0, 1, (30_000, 3)) * [42, 31, 2.2]).astype(np.float32)
    m = sm.IcpFastHip(max_source_points=len(qry), max_target_points=len(tgt), nn_mode=sm.NN_NABO, nn_epsilon=3.16)
    m.set_input_source(qry); m.set_input_target(tgt, nrm)
    ids, d2 = m.find_closests(np.eye(4), len(qry))
    m.close()
    mean = tgt.mean(axis=0)
    ids_o, d2_o, _ = cref.nn_nabo(tgt - mean, qry.astype(np.float64) - mean, 3.16)
    same = ids == ids_o
    assert same.mean() > 0.998, same.mean()
    assert np.allclose(d2[same], d2_o[same], rtol=2e-4, atol=1e-9)
    _, d2_x = cref.nn(tgt - mean, qry.astype(np.float64) - mean)
    assert (np.sqrt(d2) <= 4.16 * np.sqrt(d2_x) * (1 + 1e-4) + 1e-6).all()


def test_batch_equals_single_calls_in_the_reference_search_mode(velo20k):
    import staticmapping_amd as sm
    guesses = [velo20k["guess"], velo20k["guess"] @ sm.synth.make_pose(t=(0.05, 0.02, 0.0), rpy_deg=(0, 0, 0.4)), np.eye(4)]
    single = []
    for G in guesses:
        m = sm.IcpFastHip(max_source_points=20000, max_target_points=len(velo20k["q"]), nn_mode=sm.NN_NABO, max_iteration=25)
        m.set_input_source(velo20k["src"]); m.set_input_target(velo20k["q"], velo20k["n"])
        single.append(m.align(G)[1]); m.close()
    m = sm.IcpFastHip(pair_slots=3, max_source_points=20000, max_target_points=len(velo20k["q"]), nn_mode=sm.NN_NABO, max_iteration=25)
    for s in range(3):
        m.set_input_source(velo20k["src"], slot=s); m.set_input_target(velo20k["q"], velo20k["n"], slot=s)
    R, sc, st = m.align_batch(3, guesses)
    m.close()
    for k in range(3):
        da, dt = sm.se3_error(R[k], single[k])
        assert da < 1e-9 and dt < 1e-8, (k, da, dt)


@pytest.mark.parametrize("guess_name", ["offset", "identity", "truth"])
def test_nabo_certificates_change_no_bit(cfg2, guess_name, capsys):
    """A 120 k-point Align in the reference's search mode with traversal certificates (iterations >= 1 re-walk only the
    queries that moved further than their recorded slack) against the same Align with every query walked in every
    iteration (no_certify): the same pose bits, kept count and limit, and after the last iteration the same match of every
    query -- from a good, a poor and an exact guess, for a batch as for a single pair.  The certificate is a bound on the
    walk's own float comparisons; it is conservative, never wrong."""
    import staticmapping_amd as sm
    src, q, n = cfg2["src"], cfg2["q"], cfg2["n"]
    guess = {"offset": cfg2["guess"], "identity": np.eye(4), "truth": cfg2["T"]}[guess_name]
    ref = None
    searched = {}
    for name, opts in (("no_certify", dict(no_certify=1)), ("certify", dict())):
        m = sm.IcpFastHip(max_source_points=len(src), max_target_points=len(q), max_iteration=20, early_exit=0,
                          nn_mode=sm.NN_NABO, nn_epsilon=3.16, **opts)
        m.set_input_source(src); m.set_input_target(q, n)
        ok, R = m.align(guess)
        st = m.last_stats[0]
        ids, d2 = m.get_matches(len(src))
        m.close()
        key = (R.tobytes(), st["kept"], st["limit_d2"], st["iterations"], ids.tobytes(), d2.tobytes())
        searched[name] = st["searched_queries"]
        if ref is None:
            ref = key
        assert key[:4] == ref[:4], (name, st)
        assert key[4] == ref[4], (name, "match ids differ", int((np.frombuffer(key[4], np.int32) != np.frombuffer(ref[4], np.int32)).sum()))
        assert key[5] == ref[5], (name, "match distances differ")
    with capsys.disabled():
        print(f"\n[{guess_name}] queries walked in 20 iterations: {searched['certify']} with certificates, {searched['no_certify']} without")
    assert searched["no_certify"] == 20 * len(src)
    # how many certificates hold depends on how far the pose still moves per iteration: from identity it never settles
    assert searched["certify"] < {"offset": 0.7, "truth": 0.5, "identity": 0.95}[guess_name] * searched["no_certify"]


def test_nabo_certificates_in_a_batch(velo20k):
    """The batched launches (20 rounds per certificate workgroup, the strided list walk) against the single-pair ones.  (The sums
    the plain way in both runs: the fused certificate pass adds the same terms in another order -- its own test is below.)"""
    import staticmapping_amd as sm
    guesses = [velo20k["guess"], np.eye(4), velo20k["T"]] * 6
    out = {}
    for name, opts in (("no_certify", dict(no_certify=1)), ("certify", dict())):
        m = sm.IcpFastHip(pair_slots=len(guesses), max_source_points=20000, max_target_points=len(velo20k["q"]), nn_mode=sm.NN_NABO,
                          max_iteration=15, early_exit=0, no_fused_sums=1, **opts)
        for s in range(len(guesses)):
            m.set_input_source(velo20k["src"], slot=s); m.set_input_target(velo20k["q"], velo20k["n"], slot=s)
        R, sc, st = m.align_batch(len(guesses), guesses)
        m.close()
        out[name] = (np.asarray(R).tobytes(), [s["kept"] for s in st])
    assert out["certify"] == out["no_certify"]


@pytest.mark.parametrize("rho", [0.7, 0.4])
def test_fused_sums_in_the_reference_search_mode(velo20k, cfg1, cfg2, rho):
    """nn_mode NABO with the fused path (nn_certify_acc<., true>: traversal certificates + the sums below the predicted quantile
    band in one pass; accumulate_listed: the walked queries by the same rule and the check of the prediction) against the same
    ragged batch with certificate pass and accumulate kept apart: 18 pairs of three sizes, NaN points in some, early exit on --
    the same iteration counts, kept sets and quantiles, poses to 1e-10."""
    import staticmapping_amd as sm
    from staticmapping_amd import synth
    cases = [cfg2, velo20k, cfg1] * 6
    cap_s = max(len(c["src"]) for c in cases); cap_t = max(len(c["q"]) for c in cases)
    guesses = [c.get("guess", np.eye(4)) @ synth.make_pose(t=(0.01 * (k % 3), 0.0, 0.0), rpy_deg=(0, 0, 0.03 * (k % 4))) for k, c in enumerate(cases)]
    out = {}
    for name, opts in (("separate", dict(no_fused_sums=1)), ("fused", dict())):
        m = sm.IcpFastHip(pair_slots=len(cases), max_source_points=cap_s, max_target_points=cap_t, max_iteration=30, early_exit=1,
                          dist_outlier_ratio=rho, nn_mode=sm.NN_NABO, **opts)
        for s_, c in enumerate(cases):
            src = np.array(c["src"], dtype=np.float32, copy=True)
            if s_ % 5 == 1:
                src[7, 0] = np.nan; src[100, 2] = np.inf
            m.set_input_source(src, slot=s_); m.set_input_target(c["q"], c["n"], slot=s_)
        out[name] = m.align_batch(len(cases), guesses)
        m.close()
    Rs, scs, sts = out["separate"]; Rf, scf, stf = out["fused"]
    for s_ in range(len(cases)):
        assert stf[s_]["iterations"] == sts[s_]["iterations"] and stf[s_]["kept"] == sts[s_]["kept"] and stf[s_]["limit_d2"] == sts[s_]["limit_d2"], (s_, stf[s_], sts[s_])
        da, dt = sm.se3_error(Rf[s_], Rs[s_])
        assert da < 1e-10 and dt < 1e-9, (s_, da, dt)
        assert abs(scf[s_] - scs[s_]) < 1e-11
        assert stf[s_]["searched_queries"] == sts[s_]["searched_queries"]
    assert max(s_["fused_iterations"] for s_ in stf) > 0 and max(s_["fused_iterations"] for s_ in sts) == 0
