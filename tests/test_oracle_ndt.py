"""Self-consistency KATs of the NDT restatement (oracle/ndt.py); the reference pins nothing here."""
import numpy as np
import pytest

from oracle import ndt as o


def test_gauss_constants_match_the_survey_card():
    d1, d2, d3 = o.gauss_constants()
    assert d1 == pytest.approx(-2.2172, abs=1e-4)
    assert d2 == pytest.approx(0.4331, abs=1e-4)
    assert d3 == pytest.approx(0.5978, abs=1e-4)


def test_euler_round_trip_including_the_first_angle_range():
    for rpy in [(0.01, -0.02, 0.03), (-0.01, 0.02, -0.5), (0.3, 0.2, 1.0), (-0.4, -0.3, 2.5)]:
        T = o.pose_to_matrix_f32(np.array([0, 0, 0, *rpy])).astype(np.float64)
        e = o.euler_xyz_from_matrix(T[:3, :3])
        assert 0.0 <= e[0] <= np.pi + 1e-6 or abs(e[0]) < 1e-6      # Eigen: first angle in [0, pi]
        T2 = o.pose_to_matrix_f32(np.array([0, 0, 0, *e])).astype(np.float64)
        assert np.abs(T - T2).max() < 1e-6


def test_svd_solve_equals_direct_solve_for_a_regular_matrix():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6)); A = A @ A.T + np.eye(6)
    b = rng.normal(size=6)
    assert np.allclose(o.svd_solve(A, b), np.linalg.solve(A, b), atol=1e-10)
    A[:, 5] = A[:, 4]; A[5, :] = A[4, :]                               # singular: minimum-norm solution
    x = o.svd_solve(A, b)
    assert np.allclose(x, np.linalg.pinv(A) @ b, atol=1e-8)


def test_voxel_grid_keeps_only_voxels_with_six_points_and_adds_identity():
    rng = np.random.default_rng(1)
    dense = rng.uniform(0.05, 0.95, (50, 3)).astype(np.float32)        # one voxel, 50 points
    sparse = (rng.uniform(0.05, 0.95, (5, 3)) + [3, 0, 0]).astype(np.float32)      # 5 points: not searchable
    g = o.VoxelGrid(np.concatenate([dense, sparse]))
    assert len(g.mean) == 1 and g.valid[0]
    p = dense.astype(np.float64)
    n = len(p)
    mean = p.mean(axis=0)
    cov = ((np.eye(3) + p.T @ p) - 2 * np.outer(p.sum(0), mean)) / n + np.outer(mean, mean)     # cov_ starts as I (Leaf ctor)
    cov *= (n - 1.0) / n
    assert np.allclose(g.mean[0], mean) and np.allclose(np.linalg.inv(g.icov[0]), cov, rtol=1e-8)


def test_gradient_is_the_derivative_of_the_score():
    """computeDerivatives' gradient vs finite differences of its own score (float inner math: loose)."""
    from staticmapping_amd import synth
    scene = synth.make_scene(0)
    tgt = synth.velodyne_scan(scene, synth.make_pose(), seed=1, n_points=20000)
    src = synth.velodyne_scan(scene, synth.make_pose(t=(0.3, 0, 0)), seed=2, n_points=3000)
    grid = o.VoxelGrid(tgt)
    d1, d2, _ = o.gauss_constants()
    p = np.array([0.25, 0.02, 0.0, 0.004, -0.003, 0.01])
    pairs = grid.radius_pairs(o.transform_cloud_f32(src, o.pose_to_matrix_f32(p)))     # frozen neighbourhoods

    def score_at(pp):
        T = o.pose_to_matrix_f32(pp)
        return o.compute_derivatives(grid, src, o.transform_cloud_f32(src, T), pp, d1, d2, False, pairs=pairs)[0]

    _, g, H, _ = o.compute_derivatives(grid, src, o.transform_cloud_f32(src, o.pose_to_matrix_f32(p)), p, d1, d2, True, pairs=pairs)
    for k, h in [(0, 2e-3), (1, 2e-3), (2, 2e-3), (5, 5e-4)]:
        e = np.zeros(6); e[k] = h
        fd = (score_at(p + e) - score_at(p - e)) / (2 * h)
        assert abs(fd - g[k]) <= 0.03 * max(abs(g[k]), 50.0), (k, fd, g[k])
    assert np.allclose(H, H.T, rtol=1e-3, atol=1e-3 * np.abs(H).max())               # eq. 6.13 is symmetric
