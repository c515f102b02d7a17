"""The single-pair Align as one cooperative launch (csrc/icp_one.hip) against the separate launches per iteration
(no_single_kernel = 1) and against the oracle: the front end's call shape, builder/map_builder.cc:317-333 -> icp_fast.cc:455-529.
Matches, distances, histogram, quantile and kept set are the same bits in both forms; the normal-equation sums are added in a
different fixed order, so poses agree to rounding, and a run is reproducible bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-4      # rad, BASELINE.json north_star
TRANS_TOL = 1e-3    # m


@pytest.fixture(scope="module")
def smhip():
    import staticmapping_amd as sm
    from staticmapping_amd import _capi
    lib = _capi.load_library()
    assert lib.smhip_device_count() >= 1, "no gfx950 device visible"
    return sm


def _run(sm, c, guess, repeats=1, **opts):
    m = sm.IcpFastHip(max_source_points=len(c["src"]), max_target_points=len(c["q"]), **opts)
    m.set_input_source(c["src"])
    m.set_input_target(c["q"], c["n"])
    _, R = m.align(guess)
    out = (R.copy(), m.get_fitness_score(), dict(m.last_stats[0]))
    for rep in range(repeats):                             # target structures kept; every run the same bits
        _, R2 = m.align(guess)
        assert np.array_equal(R, R2) and m.get_fitness_score() == out[1] and dict(m.last_stats[0]) == out[2], f"Align {rep + 2} differs from the first"
    used, fell = m.single_launch_counts()
    # no launch stopped itself; and a handle told to keep to the separate launches never took the one launch
    assert fell == 0 and (used == 0 if opts.get("no_single_kernel") else used in (0, repeats + 1)), (used, fell)
    m.close()
    return out


def _same_alignment(sm, a, b, what):
    """The two forms add the normal equations in different orders: poses differ in their last bits (1e-15), so a borderline
    certificate or a distance's last float bit may fall the other way -- the counts of searched queries may differ by a few, the
    quantile by an ulp.  Everything else is equal."""
    Ra, sa, ta = a
    Rb, sb, tb = b
    for k in ("iterations", "status", "refined_iterations", "fallback_queries"):
        assert ta[k] == tb[k], (what, k, ta[k], tb[k])
    assert abs(ta["kept"] - tb["kept"]) <= 2, (what, ta["kept"], tb["kept"])
    assert abs(ta["limit_d2"] - tb["limit_d2"]) <= 1e-6 * abs(tb["limit_d2"]), (what, ta["limit_d2"], tb["limit_d2"])
    # (how many queries had to be searched again is a property of the recorded bounds, not of the result: the one launch searches
    # its failing queries with the listed search, whose radius margin is the query's own motion; the separate launches of a single
    # pair with nn_ball_lds, whose margin is the pair's motion bound)
    for k, tol in (("hard_queries", 0.10), ("searched_queries", 0.10)):
        assert abs(ta[k] - tb[k]) <= tol * max(ta[k], tb[k]) + 2, (what, k, ta[k], tb[k])
    da, dt = sm.se3_error(Ra, Rb)
    assert da < 1e-11 and dt < 1e-11, (what, da, dt)
    assert abs(sa - sb) <= 1e-12 * max(1.0, abs(sa)), (what, sa, sb)


@pytest.mark.parametrize("fixture_name", ["cfg1", "velo20k", "cfg2"])
@pytest.mark.parametrize("early_exit", [0, 1])
def test_one_launch_equals_the_separate_launches(request, smhip, fixture_name, early_exit):
    c = request.getfixturevalue(fixture_name)
    guess = c.get("guess", np.eye(4))
    opts = dict(max_iteration=20 if not early_exit else 100, early_exit=early_exit)
    one = _run(smhip, c, guess, **opts)
    sep = _run(smhip, c, guess, no_single_kernel=1, **opts)
    _same_alignment(smhip, one, sep, fixture_name)


def test_one_launch_against_the_oracle_full_size(smhip, cfg2):
    from oracle import cref
    c = cfg2
    R, score, st = _run(smhip, c, c["guess"], max_iteration=20, early_exit=0)
    ref = cref.icp_fast_align(c["src"][:, :3].astype(np.float64), c["q"], c["n"], guess=c["guess"], max_iteration=20, early_exit=False)
    da, dt = smhip.se3_error(R, ref["result"])
    assert da < ROT_TOL and dt < TRANS_TOL, (da, dt)
    assert st["iterations"] == ref["iterations"] == 20
    assert abs(score - ref["score"]) < 1e-4


def test_poor_guess_refines_bounds_in_both_forms(smhip, velo20k):
    """A search radius that the first iterations' quantile exceeds: the lower bounds must be refined to matches (ring search +
    fallback, shared by all workgroups of the one launch)."""
    c = velo20k
    opts = dict(max_iteration=12, early_exit=0, ball_radius=0.1, grid_cell=0.5)
    one = _run(smhip, c, c["guess"], **opts)
    sep = _run(smhip, c, c["guess"], no_single_kernel=1, **opts)
    assert one[2]["refined_iterations"] > 0
    _same_alignment(smhip, one, sep, "refine")


def test_far_source_goes_through_the_fallback_in_both_forms(smhip, velo20k):
    from staticmapping_amd import synth
    c = velo20k
    guess = synth.make_pose(t=(40.0, 0.0, 0.0))            # most queries far outside the target's box
    opts = dict(max_iteration=6, early_exit=0, grid_max_ring=1)
    one = _run(smhip, c, guess, **opts)
    sep = _run(smhip, c, guess, no_single_kernel=1, **opts)
    _same_alignment(smhip, one, sep, "far")


@pytest.mark.parametrize("n", [1, 63, 256, 257, 300, 513, 1000, 4097])
def test_ragged_sizes(smhip, cfg1, n):
    c = dict(cfg1)
    c["src"] = cfg1["src"][:n]
    opts = dict(max_iteration=8, early_exit=0)
    one = _run(smhip, c, np.eye(4), repeats=25, **opts)    # (a handful of workgroups, most of them idle: the tightest timing the barriers see)
    sep = _run(smhip, c, np.eye(4), no_single_kernel=1, **opts)
    _same_alignment(smhip, one, sep, f"n={n}")


def test_several_rounds_per_workgroup(smhip, velo20k, monkeypatch):
    """A grid of 16 workgroups for 20 000 points: five rounds of 256 points each (the form large clouds take)."""
    c = velo20k
    opts = dict(max_iteration=15, early_exit=0)
    sep = _run(smhip, c, c["guess"], no_single_kernel=1, **opts)
    full = _run(smhip, c, c["guess"], **opts)
    monkeypatch.setenv("SMHIP_ONE_BLOCKS", "16")
    few = _run(smhip, c, c["guess"], **opts)
    monkeypatch.delenv("SMHIP_ONE_BLOCKS")
    _same_alignment(smhip, few, sep, "16 workgroups")
    _same_alignment(smhip, full, sep, "full grid")


def test_too_many_rounds_take_the_separate_launches(smhip, cfg2, monkeypatch):
    """120 000 points over 8 workgroups would be 59 rounds each: the handle falls back to the launches per iteration (same result)."""
    c = cfg2
    opts = dict(max_iteration=5, early_exit=0)
    sep = _run(smhip, c, c["guess"], no_single_kernel=1, **opts)
    monkeypatch.setenv("SMHIP_ONE_BLOCKS", "8")
    few = _run(smhip, c, c["guess"], **opts)
    monkeypatch.delenv("SMHIP_ONE_BLOCKS")
    for k in ("iterations", "kept", "limit_d2", "searched_queries"):
        assert few[2][k] == sep[2][k], k
    assert np.array_equal(few[0], sep[0])


def test_identical_clouds_and_no_match_cases(smhip, cfg1):
    """The NaN rule (identical clouds -> identity, icp_fast.cc:315-321) and the no-correspondence status through the one launch."""
    c = cfg1
    m = smhip.IcpFastHip(max_source_points=len(c["q"]), max_target_points=len(c["q"]), max_iteration=5, early_exit=0)
    m.set_input_source(c["q"]); m.set_input_target(c["q"], c["n"])
    _, R = m.align(np.eye(4))
    assert np.allclose(R, np.eye(4), atol=1e-5) and m.get_fitness_score() > 0.999
    m.close()
    m = smhip.IcpFastHip(max_source_points=8, max_target_points=len(c["q"]), max_iteration=5, early_exit=0)
    m.set_input_source(np.full((8, 3), np.nan)); m.set_input_target(c["q"], c["n"])
    with pytest.raises(smhip.SmhipError):
        m.align(np.eye(4))
    assert m.last_stats[0]["status"] != 0
    m.close()


def test_six_matchers_align_at_once(smhip, cfg2):
    """Six host threads, a matcher each, single 120 k-point Aligns at the same time (the back end's pool, builder/map_builder.cc:399-446,
    655): six grids of 256 resident workgroups do not fit the device together -- the cooperative launches must queue, not deadlock, and
    every result must be the single-threaded run's bits."""
    import threading
    c = cfg2
    opts = dict(max_iteration=20, early_exit=0)
    ref = _run(smhip, c, c["guess"], **opts)[0]
    err = []

    def work(k):
        try:
            m = smhip.IcpFastHip(max_source_points=len(c["src"]), max_target_points=len(c["q"]), **opts)
            m.set_input_source(c["src"]); m.set_input_target(c["q"], c["n"])
            for _ in range(5):
                if not np.array_equal(m.align(c["guess"])[1], ref):
                    err.append((k, "differs"))
            m.close()
        except Exception as e:      # noqa: BLE001
            err.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a matcher did not return"
    assert not err, err


@pytest.mark.parametrize("npairs", [2, 3, 6, 8])
def test_small_batches_take_the_one_launch(smhip, velo20k, cfg1, cfg2, npairs):
    """Up to eight pairs (the back end's handful of concurrent submap pairs) share one cooperative launch, a row of the grid each:
    clouds of three very different sizes in one batch (what exposed the launch's two missing barriers), against the same batch as
    separate launches and against a single Align; ten repeated launches, every one the first one's bits."""
    cases = [cfg2, cfg1, velo20k, cfg2, velo20k, cfg1, cfg2, velo20k][:npairs]
    guesses = [c.get("guess", np.eye(4)) for c in cases]
    cap_s = max(len(c["src"]) for c in cases); cap_t = max(len(c["q"]) for c in cases)
    out = {}
    for name, opts in (("one", dict()), ("separate", dict(no_single_kernel=1))):
        m = smhip.IcpFastHip(pair_slots=npairs, max_source_points=cap_s, max_target_points=cap_t, max_iteration=20, early_exit=1, **opts)
        for s, c in enumerate(cases):
            m.set_input_source(c["src"], slot=s); m.set_input_target(c["q"], c["n"], slot=s)
        R, sc, st = m.align_batch(npairs, guesses)
        for rep in range(10 if name == "one" else 1):
            R2, sc2, st2 = m.align_batch(npairs, guesses)
            assert R.tobytes() == R2.tobytes() and sc.tobytes() == sc2.tobytes(), (name, rep)
        out[name] = (R, sc, [dict(x) for x in st])
        m.close()
    for s in range(npairs):
        a, b_ = out["one"], out["separate"]
        assert a[2][s]["iterations"] == b_[2][s]["iterations"] and abs(a[2][s]["kept"] - b_[2][s]["kept"]) <= 2, (s, a[2][s], b_[2][s])
        da, dt = smhip.se3_error(a[0][s], b_[0][s])
        assert da < 1e-10 and dt < 1e-10, (s, da, dt)
        assert abs(a[1][s] - b_[1][s]) < 1e-11
    # a pair's result does not depend on what shares the launch with it: slot 0 against the same pair aligned alone (its grid differs)
    m1 = smhip.IcpFastHip(max_source_points=cap_s, max_target_points=cap_t, max_iteration=20, early_exit=1)
    m1.set_input_source(cases[0]["src"]); m1.set_input_target(cases[0]["q"], cases[0]["n"])
    _, R1 = m1.align(guesses[0])
    m1.close()
    da, dt = smhip.se3_error(out["one"][0][0], R1)
    assert da < 1e-10 and dt < 1e-10, (da, dt)


def test_mixed_small_batch_many_iterations(smhip, cfg1, velo20k):
    """Pairs of different sizes, 25 iterations each, bounds refined in most iterations of the small ones: the shape in which a fast
    workgroup used to take bounds out of the histogram a slow one was still reading, and in which the score's fold used to overwrite
    the last iteration's rows.  40 launches of 7 pairs, every one the first one's bits."""
    cases = [cfg1, velo20k, velo20k, cfg1, velo20k, velo20k, cfg1]
    guesses = [c.get("guess", np.eye(4)) for c in cases]
    cap_s = max(len(c["src"]) for c in cases); cap_t = max(len(c["q"]) for c in cases)
    m = smhip.IcpFastHip(pair_slots=7, max_source_points=cap_s, max_target_points=cap_t, max_iteration=25, early_exit=0)
    for s, c in enumerate(cases):
        m.set_input_source(c["src"], slot=s); m.set_input_target(c["q"], c["n"], slot=s)
    R, sc, st = m.align_batch(7, guesses)
    assert max(x["refined_iterations"] for x in st) > 5
    for rep in range(40):
        R2, sc2, st2 = m.align_batch(7, guesses)
        assert R.tobytes() == R2.tobytes() and sc.tobytes() == sc2.tobytes(), rep
    m.close()
