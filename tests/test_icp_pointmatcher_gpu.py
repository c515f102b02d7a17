"""registrators::IcpUsingPointMatcher chain (BASELINE config #1 names it) on the GPU engine vs its restatement.

The chain runs on the device end to end (random sampling, CalculateNormals of the reference, the 150-iteration loop and
the post-hoc score pass); the oracle is handed (a) the sampling mask the device drew -- a pure function of (seed, row) --
and (b) either its own CalculateNormals (the whole chain, tolerance = the north-star 1e-4 rad / 1e-3 m) or the target
the device prepared (isolates the loop + score: same iteration count, same score)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_target_fn(m):
    """normals_fn for the oracle that returns the ICP target resident in slot 0 (what the device CalculateNormals made)."""
    def fn(_points):
        p, n = m._m.get_target(m.target_points, slot=0)
        return p.astype(np.float64), n.astype(np.float64), None
    return fn


def _check_chain(m, src, tgt, guess, truth=None):
    import staticmapping_amd as sm
    from oracle import icp_pointmatcher as opm
    from oracle import cref
    ok, R = m.align(guess)
    # (b) the loop and the score pass against the oracle on the same prepared target
    ok_d, R_d, score_d, it_d = opm.align(src, tgt, guess, m.last_mask, normals_fn=_device_target_fn(m))
    da, dt = sm.se3_error(R, R_d)
    assert da < 1e-5 and dt < 1e-4, ("loop vs oracle on the device-prepared target", da, dt)
    assert m.iterations == it_d
    assert abs(m.get_fitness_score() - score_d) < 1e-5 and ok == ok_d
    # (a) the whole chain, the oracle preparing its own target
    ok_o, R_o, score_o, it_o = opm.align(src, tgt, guess, m.last_mask, normals_fn=lambda p: cref.calculate_normals(p))
    da, dt = sm.se3_error(R, R_o)
    assert da < 1e-4 and dt < 1e-3, ("whole chain vs oracle", da, dt)
    assert abs(m.get_fitness_score() - score_o) < 1e-4 and ok == ok_o
    if truth is not None:
        da, dt = sm.se3_error(R, truth)
        assert da < 3e-3 and dt < 3e-2
    return ok, R


def test_cfg1_pointmatcher_chain_parity(cfg1):
    """Config #1: two 5k-pt clouds on three noisy planes, known SE(3) offset."""
    import staticmapping_amd as sm
    m = sm.IcpPointMatcherHip(max_points=8192, prob=0.9, seed=3)
    m.set_input_source(cfg1["src"])
    m.set_input_target(cfg1["tgt"])
    _check_chain(m, cfg1["src"], cfg1["tgt"], np.eye(4), cfg1["T"])
    m.close()


def test_velodyne_pointmatcher_chain_and_nan_drop(velo20k):
    import staticmapping_amd as sm
    src = velo20k["src"].copy()
    src[::211, 1] = np.nan                       # InnerCloudToPmPoints drops these (:57-66)
    m = sm.IcpPointMatcherHip(max_points=32768, prob=0.9, seed=11)
    m.set_input_source(src)
    m.set_input_target(velo20k["tgt"])
    _check_chain(m, src, velo20k["tgt"], velo20k["guess"])
    m.close()


def test_device_chain_equals_host_chain(velo20k):
    """The device chain (2 uploads) against the host-side chain it replaced (host CalculateNormals, 4 uploads, a second
    Align for the score): same sampled set, results within the tolerance, same accept decision."""
    import staticmapping_amd as sm
    a = sm.IcpPointMatcherHip(max_points=32768, prob=0.9, seed=5)
    b = sm.IcpPointMatcherHip(max_points=32768, prob=0.9, seed=5, device_chain=False)
    for m in (a, b):
        m.set_input_source(velo20k["src"]); m.set_input_target(velo20k["tgt"])
    ok_a, Ra = a.align(velo20k["guess"])
    ok_b, Rb = b.align(velo20k["guess"])
    assert np.array_equal(a.last_mask, b.last_mask) and ok_a == ok_b
    da, dt = sm.se3_error(Ra, Rb)
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert abs(a.get_fitness_score() - b.get_fitness_score()) < 1e-4
    a.close(); b.close()


def test_sampled_source_keeps_caller_order_indices(velo20k):
    """smhip_sample_source: the kept set is the host-reproducible mask and get_matches of the sampled slot is indexed by
    the sampled cloud's rows in caller order."""
    import ctypes
    import staticmapping_amd as sm
    from oracle import cref
    m = sm.IcpPointMatcherHip(max_points=32768, prob=0.7, seed=9)
    src = np.ascontiguousarray(velo20k["src"][:, :3], dtype=np.float32)
    mask = m.sampling_mask(len(src))
    h = m._m
    h.set_input_source(src, slot=1)
    n_out = ctypes.c_int32()
    h._check(h._lib.smhip_sample_source(h._h, 1, 0, 0.7, 9, ctypes.byref(n_out)))
    assert n_out.value == int(mask.sum())
    q, nrm = sm.calculate_normals(velo20k["tgt"][:, :3].astype(np.float64))
    h.set_input_target(q, nrm, slot=0)
    ids, d2 = h.find_closests(velo20k["guess"], n_out.value)
    moved = src[mask].astype(np.float64) @ velo20k["guess"][:3, :3].T + velo20k["guess"][:3, 3]
    ids_ref, d2_ref = cref.nn(q, moved)
    assert (ids == ids_ref).mean() > 0.9995
    assert np.allclose(d2, d2_ref, rtol=1e-4, atol=1e-8)
    m.close()


def test_chain_with_the_reference_matcher_settings(velo20k):
    """KDTreeMatcher as icp_pointmatcher.cc:186-191 configures it (knn 1, epsilon 3.16 = libnabo's approximate search) in
    the loop AND in the post-hoc score pass: device (nn_mode NABO) vs the oracle run the same way."""
    import staticmapping_amd as sm
    from oracle import icp_pointmatcher as opm
    m = sm.IcpPointMatcherHip(max_points=32768, prob=0.9, seed=4, nn_mode=sm.NN_NABO, nn_epsilon=3.16)
    m.set_input_source(velo20k["src"]); m.set_input_target(velo20k["tgt"])
    ok, R = m.align(velo20k["guess"])
    ok_o, R_o, score_o, it_o = opm.align(velo20k["src"], velo20k["tgt"], velo20k["guess"], m.last_mask,
                                          normals_fn=_device_target_fn(m), nn_eps=3.16)
    da, dt = sm.se3_error(R, R_o)
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert m.iterations == it_o and ok == ok_o
    assert abs(m.get_fitness_score() - score_o) < 1e-4
    m.close()
