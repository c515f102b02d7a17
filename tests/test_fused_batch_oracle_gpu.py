"""The batch path that carries 17 of the headline's 20 iterations -- the fused certificate pass (nn_certify_acc), the
balanced listed search and finalize's listed / record phases -- compared DIRECTLY with the CPU oracle (not with the
device's own separate passes): 32 distinct BASELINE-config-#2-size pairs of the synthetic drive through the library's
default options, with the three kinds of guess the bench times (icp_fast.cc:455-529; builder/map_builder.cc:302-333).
Tolerance: BASELINE.json's 1e-4 rad / 1e-3 m, equal iteration counts, scores to 1e-4."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-4      # rad, BASELINE.json north_star
TRANS_TOL = 1e-3    # m
N_PAIRS = 32


@pytest.fixture(scope="module")
def drive():
    import torch
    from staticmapping_amd import synth
    return synth.drive_pairs(N_PAIRS, 120_000, torch.device("cuda", 0), seed=5)


def _guesses(work, kind):
    if kind == "extrapolated":
        return [w["guess_cv"] for w in work]
    if kind == "identity":
        return [w["guess_id"] for w in work]
    return [w["guess_cv" if k % 2 == 0 else "guess_id"] for k, w in enumerate(work)]


def _batch(sm, work, guesses, **opts):
    ns = max(len(w["src"]) for w in work)
    nt = max(len(w["q"]) for w in work)
    m = sm.IcpFastHip(pair_slots=len(work), max_source_points=ns, max_target_points=nt, max_iteration=20, early_exit=0, **opts)
    for s, w in enumerate(work):
        m.set_input_source(w["src"], slot=s)
        m.set_input_target(w["q"], w["n"], slot=s)
    # twice: the second batch places the switch to certificate pass + listed search from the first one's search counts, as
    # every batch after a handle's first does -- the state the bench times
    m.align_batch(len(work), guesses)
    R, sc, st = m.align_batch(len(work), guesses)
    split = m.get_profile()["split_after_used"]
    m.close()
    return R, sc, st, split


@pytest.mark.parametrize("kind", ["extrapolated", "identity", "alternating"])
def test_default_batch_path_against_the_oracle(drive, kind, capsys):
    import staticmapping_amd as sm
    from oracle import cref
    g = _guesses(drive, kind)
    R, sc, st, split = _batch(sm, drive, g)
    fused = [s["fused_iterations"] for s in st]
    assert min(fused) > 0, fused                           # every pair's later iterations did come from the fused pass
    worst = [0.0, 0.0, 0.0]
    cores = cref.usable_cores()
    for s, w in enumerate(drive):
        ref = cref.icp_fast_align(w["src"][:, :3].astype(np.float64), w["q"], w["n"], guess=g[s], max_iteration=20,
                                  dist_outlier_ratio=0.7, early_exit=False, nthreads=cores)
        da, dt = sm.se3_error(R[s], ref["result"])
        assert da < ROT_TOL and dt < TRANS_TOL, (kind, s, da, dt)
        assert st[s]["iterations"] == 20 == ref["iterations"]
        assert abs(sc[s] - ref["score"]) < 1e-4, (kind, s, sc[s], ref["score"])
        worst = [max(worst[0], da), max(worst[1], dt), max(worst[2], abs(sc[s] - ref["score"]))]
    with capsys.disabled():
        print(f"\n[fused batch vs oracle, {kind} guess] {N_PAIRS} pairs: worst {worst[0]:.2e} rad / {worst[1]:.2e} m, score {worst[2]:.1e}; "
              f"switch at iteration {split}, fused iterations {min(fused)}..{max(fused)}")


def test_reference_search_batch_against_the_eps_oracle(drive, capsys):
    """nn_mode NABO (libnabo's tree and epsilon = 3.16 search on the device, icp_fast.cc:174, 464-467) through the batch path with
    its fused form, against the oracle run with the same approximate search: the parity the exact modes cannot have."""
    import staticmapping_amd as sm
    from oracle import cref
    g = _guesses(drive, "extrapolated")
    R, sc, st, _ = _batch(sm, drive, g, nn_mode=sm.NN_NABO, nn_epsilon=3.16)
    assert max(s["fused_iterations"] for s in st) > 0
    worst = [0.0, 0.0]
    for s, w in enumerate(drive):
        ref = cref.icp_fast_align(w["src"][:, :3].astype(np.float64), w["q"], w["n"], guess=g[s], max_iteration=20,
                                  dist_outlier_ratio=0.7, early_exit=False, nn_eps=3.16)
        da, dt = sm.se3_error(R[s], ref["result"])
        assert da < ROT_TOL and dt < TRANS_TOL, (s, da, dt)
        assert st[s]["iterations"] == 20 == ref["iterations"]
        assert abs(sc[s] - ref["score"]) < 1e-4
        worst = [max(worst[0], da), max(worst[1], dt)]
    with capsys.disabled():
        print(f"\n[reference-search batch vs eps oracle] {N_PAIRS} pairs: worst {worst[0]:.2e} rad / {worst[1]:.2e} m")


def _coincident_case():
    """Three families of axis-aligned planes on dyadic coordinates, symmetric about the origin (the target mean is exactly 0, so
    centring changes no bit), the z planes' points stored TWICE.  Source: six exact copies of every target point (distance 0; over
    a doubled point the runner-up is at distance 0 too, so the match's certificate bound is 0 and its certificate fails in every
    iteration: it is listed every time) and four copies displaced inside the plane by 2, 4, 7, 11 sixty-fourths of a metre
    (residual (p - q) . n = 0 exactly).  Every kept residual is 0, so the pose stays the identity guess, and the quantile -- rank
    int(n * 0.7f) = 0.7 n - 1, icp_fast.cc:86 -- is the last of the 2 / 64 m group: kept = 6 + 1 of every 10 points."""
    g = np.arange(-6.0, 6.0 + 1e-9, 0.5)
    zs = np.arange(-1.5, 1.5 + 1e-9, 0.5)
    pts, nrm, dup, tang = [], [], [], []
    for sgn in (-1.0, 1.0):
        for x in g:
            for y in g:
                pts.append((x, y, 2.0 * sgn)); nrm.append((0, 0, 1)); dup.append(True); tang.append((1, 0, 0))
        for a in g:
            for z in zs:
                pts.append((8.0 * sgn, a, z)); nrm.append((1, 0, 0)); dup.append(False); tang.append((0, 1, 0))
                pts.append((a, 8.0 * sgn, z)); nrm.append((0, 1, 0)); dup.append(False); tang.append((1, 0, 0))
    pts = np.array(pts); nrm = np.array(nrm, dtype=np.float64); dup = np.array(dup); tang = np.array(tang, dtype=np.float64)
    q = np.concatenate([pts, pts[dup]]); n = np.concatenate([nrm, nrm[dup]])
    src = [pts] * 6 + [pts + tang * (k / 64.0) for k in (2, 4, 7, 11)]
    src = np.concatenate(src)
    rs = np.random.default_rng(3)
    src = src[rs.permutation(len(src))]
    return q, n, src.astype(np.float32), len(pts), int(dup.sum())


def test_fused_sums_with_coincident_duplicated_targets():
    """A listed match whose runner-up is at distance 0 carries the certificate bound 0; finalize's listed phase has to sum it like
    any other match below the band, as `accumulate` does (kept = #(d2 <= limit), icp_fast.cc:497-498): the fused path against the
    separate passes and against the count the construction gives."""
    import staticmapping_amd as sm
    q, n, src, n_unique, n_dup = _coincident_case()
    assert abs(q.sum(axis=0)).max() == 0.0
    B = 16
    out = {}
    for name, opts in (("separate", dict(no_fused_sums=1)), ("fused", dict())):
        m = sm.IcpFastHip(pair_slots=B, max_source_points=len(src), max_target_points=len(q), max_iteration=12, early_exit=0, split_after=1, **opts)
        for s in range(B):
            m.set_input_source(src, slot=s); m.set_input_target(q, n, slot=s)
        out[name] = m.align_batch(B, [np.eye(4)] * B)
        m.close()
    Rs, scs, sts = out["separate"]; Rf, scf, stf = out["fused"]
    assert min(s["fused_iterations"] for s in stf) >= 8 and max(s["fused_iterations"] for s in sts) == 0
    for s in range(B):
        assert sts[s]["kept"] == 7 * n_unique and sts[s]["limit_d2"] == (2.0 / 64.0) ** 2, sts[s]
        assert stf[s]["kept"] == sts[s]["kept"] and stf[s]["limit_d2"] == sts[s]["limit_d2"], (s, stf[s], sts[s])
        assert np.array_equal(Rf[s], np.eye(4)) and np.array_equal(Rs[s], np.eye(4))
        assert abs(scf[s] - scs[s]) < 1e-12
        # the doubled points' exact copies are listed in every iteration after the first
        assert stf[s]["searched_queries"] >= len(src) + 10 * 6 * n_dup


def test_fused_path_with_odd_sizes_and_pair_counts(velo20k):
    """The fixed-grid kernels of the fused iteration (iteration_sums: work items per XCD from a plan over the launch's pairs; the
    listed search's tickets) with a pair count that is no multiple of 8 and source sizes around every block boundary (one certificate
    block = 8 192 points, one short accumulate block = 2 048, one long = 8 192; 5 120 was the certificate block until the re-tune): fused against separate passes -- the same kept
    sets and quantiles, poses to 1e-10 (icp_fast.cc:484-523)."""
    import staticmapping_amd as sm
    from staticmapping_amd import synth
    sizes = [20000, 5120, 5121, 2048, 2049, 8192, 8193, 10240, 1000, 19999, 4097, 16384, 12345, 6000, 5119, 15361, 20000, 777, 10241]
    src0 = np.asarray(velo20k["src"], dtype=np.float32)
    out = {}
    for name, opts in (("separate", dict(no_fused_sums=1)), ("fused", dict())):
        m = sm.IcpFastHip(pair_slots=len(sizes), max_source_points=max(sizes), max_target_points=len(velo20k["q"]), max_iteration=15, early_exit=0,
                          split_after=1, **opts)
        for s_, n_ in enumerate(sizes):
            pick = np.sort(np.random.default_rng(100 + s_).choice(len(src0), size=n_, replace=False))
            m.set_input_source(src0[pick], slot=s_); m.set_input_target(velo20k["q"], velo20k["n"], slot=s_)
        g = [velo20k["guess"] @ synth.make_pose(t=(0.01 * (k % 3), 0.005 * (k % 2), 0.0), rpy_deg=(0, 0, 0.02 * (k % 5))) for k in range(len(sizes))]
        out[name] = m.align_batch(len(sizes), g)
        m.close()
    Rs, scs, sts = out["separate"]; Rf, scf, stf = out["fused"]
    assert max(s["fused_iterations"] for s in stf) > 0 and max(s["fused_iterations"] for s in sts) == 0
    for s_ in range(len(sizes)):
        assert stf[s_]["iterations"] == sts[s_]["iterations"] == 15
        assert stf[s_]["kept"] == sts[s_]["kept"] and stf[s_]["limit_d2"] == sts[s_]["limit_d2"], (s_, sizes[s_], stf[s_], sts[s_])
        da, dt = sm.se3_error(Rf[s_], Rs[s_])
        assert da < 1e-10 and dt < 1e-9, (s_, sizes[s_], da, dt)
        assert abs(scf[s_] - scs[s_]) < 1e-11
