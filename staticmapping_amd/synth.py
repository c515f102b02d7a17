"""Seeded synthetic Velodyne-64 workload generator.

This is *workload generation*, not a checker: `bench.py`, the tests and the
oracle all draw their clouds from here so that every leg sees identical bytes.

Scene: ground plane z = -1.73 m, axis-aligned boxes (buildings / cars) and
vertical cylinders (poles / trunks).  Sensor: 64 rings with elevation linearly
spaced +2.0 deg ... -24.8 deg, 1875 azimuth steps => 120 000 rays, ring-major
order, max range 80 m, Gaussian range noise.  Misses are dropped and the
scan is padded with extra random rays so it holds exactly `n_points` returns.
The output layout is the KITTI `.bin` one the reference reads
(ros_node/kitti_reader.cc:91-121): float32 [N,4] = x, y, z, intensity in the
sensor frame.
"""
from __future__ import annotations

import dataclasses
import numpy as np

GROUND_Z = -1.73
MAX_RANGE = 80.0
N_RINGS = 64
N_AZIMUTH = 1875
ELEV_TOP_DEG = 2.0
ELEV_BOTTOM_DEG = -24.8


@dataclasses.dataclass
class Scene:
    box_min: np.ndarray   # [B,3]
    box_max: np.ndarray   # [B,3]
    cyl_xy: np.ndarray    # [C,2]
    cyl_r: np.ndarray     # [C]
    cyl_top: np.ndarray   # [C]
    ground_z: float = GROUND_Z


def make_scene(seed: int = 0, n_boxes: int = 40, n_cyl: int = 30,
               half_extent: float = 60.0, keep_clear: float = 4.0) -> Scene:
    """40 boxes (edge U[1,15] m) + 30 cylinders (r U[0.1,0.4] m) within +-60 m.

    Primitives whose footprint comes within `keep_clear` metres of the x axis
    segment [-10, 40] (where the synthetic sensor drives) are re-drawn.
    """
    rng = np.random.default_rng(seed)
    bmin, bmax = [], []
    while len(bmin) < n_boxes:
        c = rng.uniform(-half_extent, half_extent, size=2)
        sz = rng.uniform(1.0, 15.0, size=3)
        lo = np.array([c[0] - sz[0] / 2, c[1] - sz[1] / 2, GROUND_Z])
        hi = np.array([c[0] + sz[0] / 2, c[1] + sz[1] / 2, GROUND_Z + sz[2]])
        if lo[1] - keep_clear < 0.0 < hi[1] + keep_clear and hi[0] > -10.0 - keep_clear and lo[0] < 40.0 + keep_clear:
            continue
        bmin.append(lo)
        bmax.append(hi)
    cxy, cr, ct = [], [], []
    while len(cxy) < n_cyl:
        c = rng.uniform(-half_extent, half_extent, size=2)
        r = rng.uniform(0.1, 0.4)
        if abs(c[1]) < keep_clear + r and -10.0 - keep_clear < c[0] < 40.0 + keep_clear:
            continue
        cxy.append(c)
        cr.append(r)
        ct.append(GROUND_Z + rng.uniform(2.0, 8.0))
    return Scene(np.array(bmin), np.array(bmax), np.array(cxy), np.array(cr), np.array(ct))


def ring_directions(n_rings: int = N_RINGS, n_az: int = N_AZIMUTH) -> np.ndarray:
    """Unit ray directions in the sensor frame, ring-major, [n_rings*n_az, 3]."""
    elev = np.deg2rad(np.linspace(ELEV_TOP_DEG, ELEV_BOTTOM_DEG, n_rings))
    az = np.linspace(0.0, 2.0 * np.pi, n_az, endpoint=False)
    e, a = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], axis=-1)
    return d.reshape(-1, 3)


def _raycast(scene: Scene, origin: np.ndarray, dirs: np.ndarray) -> np.ndarray:
    """Range of the first hit along each world-frame ray (inf = miss)."""
    n = dirs.shape[0]
    best = np.full(n, np.inf)
    # ground
    dz = dirs[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (scene.ground_z - origin[2]) / dz
    tg = np.where((dz < 0) & (tg > 0), tg, np.inf)
    best = np.minimum(best, tg)
    # boxes: slab method, all rays x all boxes
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dirs                                    # [n,3]
        t0 = (scene.box_min[None, :, :] - origin[None, None, :]) * inv[:, None, :]
        t1 = (scene.box_max[None, :, :] - origin[None, None, :]) * inv[:, None, :]
    tn = np.nanmax(np.minimum(t0, t1), axis=2)
    tf = np.nanmin(np.maximum(t0, t1), axis=2)
    hit = (tf >= tn) & (tf > 0)
    tb = np.where(hit, np.where(tn > 0, tn, np.inf), np.inf)
    best = np.minimum(best, tb.min(axis=1))
    # vertical cylinders (side surface only)
    ox = origin[0] - scene.cyl_xy[:, 0]                      # [C]
    oy = origin[1] - scene.cyl_xy[:, 1]
    dx = dirs[:, 0:1]
    dy = dirs[:, 1:2]
    a = dx * dx + dy * dy                                    # [n,1]
    b = 2.0 * (dx * ox[None, :] + dy * oy[None, :])          # [n,C]
    c = (ox * ox + oy * oy - scene.cyl_r ** 2)[None, :]
    disc = b * b - 4.0 * a * c
    with np.errstate(divide="ignore", invalid="ignore"):
        tc = (-b - np.sqrt(np.where(disc >= 0, disc, np.nan))) / (2.0 * a)
    zc = origin[2] + tc * dirs[:, 2:3]
    ok = (disc >= 0) & (tc > 0) & (zc >= scene.ground_z) & (zc <= scene.cyl_top[None, :])
    tc = np.where(ok, tc, np.inf)
    best = np.minimum(best, tc.min(axis=1))
    best[best > MAX_RANGE] = np.inf
    return best


def _raycast_torch(scene: Scene, origin: np.ndarray, dirs: np.ndarray, device) -> np.ndarray:
    """_raycast with the rays-by-primitives arithmetic in float64 torch ops on `device` (the numpy form takes ~3 s per
    120k-ray scan; benches that need dozens of scans generate them on the GPU).  Same formulas, same result up to
    floating-point association."""
    import torch
    f64 = torch.float64
    d = torch.as_tensor(dirs, dtype=f64, device=device)
    o = torch.as_tensor(origin, dtype=f64, device=device)
    inf = torch.tensor(float("inf"), dtype=f64, device=device)
    dz = d[:, 2]
    tg = (scene.ground_z - o[2]) / dz
    best = torch.where((dz < 0) & (tg > 0), tg, inf)
    if len(scene.box_min):
        bmin = torch.as_tensor(scene.box_min, dtype=f64, device=device)
        bmax = torch.as_tensor(scene.box_max, dtype=f64, device=device)
        inv = 1.0 / d
        t0 = (bmin[None, :, :] - o[None, None, :]) * inv[:, None, :]
        t1 = (bmax[None, :, :] - o[None, None, :]) * inv[:, None, :]
        lo = torch.nan_to_num(torch.minimum(t0, t1), nan=-float("inf"))
        hi = torch.nan_to_num(torch.maximum(t0, t1), nan=float("inf"))
        tn = lo.max(dim=2).values
        tf = hi.min(dim=2).values
        hit = (tf >= tn) & (tf > 0) & (tn > 0)
        best = torch.minimum(best, torch.where(hit, tn, inf).min(dim=1).values)
    if len(scene.cyl_r):
        cxy = torch.as_tensor(scene.cyl_xy, dtype=f64, device=device)
        cr = torch.as_tensor(scene.cyl_r, dtype=f64, device=device)
        ct = torch.as_tensor(scene.cyl_top, dtype=f64, device=device)
        ox = o[0] - cxy[:, 0]
        oy = o[1] - cxy[:, 1]
        dx, dy = d[:, 0:1], d[:, 1:2]
        a = dx * dx + dy * dy
        b = 2.0 * (dx * ox[None, :] + dy * oy[None, :])
        c = (ox * ox + oy * oy - cr ** 2)[None, :]
        disc = b * b - 4.0 * a * c
        tc = (-b - torch.sqrt(torch.clamp(disc, min=0.0))) / (2.0 * a)
        zc = o[2] + tc * d[:, 2:3]
        ok = (disc >= 0) & (tc > 0) & (zc >= scene.ground_z) & (zc <= ct[None, :])
        best = torch.minimum(best, torch.where(ok, tc, inf).min(dim=1).values)
    best = torch.where(best > MAX_RANGE, inf, best)
    return best.cpu().numpy()


def scene_near(scene: Scene, origin: np.ndarray, radius: float = MAX_RANGE + 12.0) -> Scene:
    """The primitives a sensor at `origin` can reach (a drive scene holds thousands; one scan sees a few dozen)."""
    c = 0.5 * (scene.box_min[:, :2] + scene.box_max[:, :2]) if len(scene.box_min) else np.zeros((0, 2))
    kb = np.linalg.norm(c - origin[:2], axis=1) < radius if len(c) else np.zeros(0, bool)
    kc = np.linalg.norm(scene.cyl_xy - origin[:2], axis=1) < radius if len(scene.cyl_xy) else np.zeros(0, bool)
    return Scene(scene.box_min[kb], scene.box_max[kb], scene.cyl_xy[kc], scene.cyl_r[kc], scene.cyl_top[kc], scene.ground_z)


def drive_poses(n_poses: int, seed: int = 5, speed: float = 8.0, hz: float = 10.0, yaw_rate_max: float = 0.2,
                segment_s: float = 1.0, speed_spread: float = 0.0) -> list:
    """SURVEY.md §8(d) cfg 4: a drive at `speed` m/s sampled at `hz`, yaw rate ~ U[-yaw_rate_max, yaw_rate_max] rad/s
    drawn every `segment_s` seconds and interpolated linearly in between (a vehicle does not jump its yaw rate); with
    `speed_spread` > 0 the speed is drawn the same way from U[speed - spread, speed + spread].  Planar, starting at
    the origin heading +x.  Deterministic in its arguments."""
    rng = np.random.default_rng(seed)
    dt = 1.0 / hz
    per_seg = max(1, int(round(segment_s * hz)))
    n_knots = n_poses // per_seg + 2
    rate_k = rng.uniform(-yaw_rate_max, yaw_rate_max, size=n_knots)
    speed_k = speed + rng.uniform(-speed_spread, speed_spread, size=n_knots)
    x = y = yaw = 0.0
    poses = []
    for k in range(n_poses):
        seg, f = divmod(k, per_seg)
        w = f / per_seg
        rate = (1 - w) * rate_k[seg] + w * rate_k[seg + 1]
        v = (1 - w) * speed_k[seg] + w * speed_k[seg + 1]
        T = np.eye(4)
        T[:3, :3] = rpy_to_matrix(0.0, 0.0, yaw)
        T[:3, 3] = (x, y, 0.0)
        poses.append(T)
        x += v * dt * np.cos(yaw)
        y += v * dt * np.sin(yaw)
        yaw += rate * dt
    return poses


def make_drive_scene(poses, seed: int = 5, keep_clear: float = 4.0, margin: float = 90.0) -> Scene:
    """make_scene's primitives at the same density (40 boxes + 30 cylinders per 120 m x 120 m) over the bounding box of a
    drive +- `margin`, minus everything within `keep_clear` metres of the path."""
    rng = np.random.default_rng(seed)
    path = np.array([P[:2, 3] for P in poses])
    lo, hi = path.min(axis=0) - margin, path.max(axis=0) + margin
    area = float(np.prod(hi - lo))
    n_boxes = max(8, int(round(area * 40 / 14400.0)))
    n_cyl = max(6, int(round(area * 30 / 14400.0)))
    sub = path[:: max(1, len(path) // 4000)]

    def clear(lo2, hi2):
        """min distance from the path to each axis-aligned rectangle [lo2, hi2] (rows) > keep_clear"""
        out = np.ones(len(lo2), dtype=bool)
        for s0 in range(0, len(lo2), 512):
            a, b = lo2[s0:s0 + 512], hi2[s0:s0 + 512]
            d = np.maximum(np.maximum(a[:, None, :] - sub[None, :, :], sub[None, :, :] - b[:, None, :]), 0.0)
            out[s0:s0 + 512] = np.sqrt((d * d).sum(axis=2)).min(axis=1) > keep_clear
        return out
    c = rng.uniform(lo, hi, size=(n_boxes, 2))
    sz = rng.uniform(1.0, 15.0, size=(n_boxes, 3))
    bmin = np.column_stack([c - sz[:, :2] / 2, np.full(n_boxes, GROUND_Z)])
    bmax = np.column_stack([c + sz[:, :2] / 2, GROUND_Z + sz[:, 2]])
    kb = clear(bmin[:, :2], bmax[:, :2])
    cxy = rng.uniform(lo, hi, size=(n_cyl, 2))
    cr = rng.uniform(0.1, 0.4, size=n_cyl)
    ct = GROUND_Z + rng.uniform(2.0, 8.0, size=n_cyl)
    kc = clear(cxy - cr[:, None], cxy + cr[:, None])
    return Scene(bmin[kb], bmax[kb], cxy[kc], cr[kc], ct[kc])


def rpy_to_matrix(roll: float, pitch: float, yaw: float) -> np.ndarray:
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def make_pose(t=(0.0, 0.0, 0.0), rpy_deg=(0.0, 0.0, 0.0)) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = rpy_to_matrix(*np.deg2rad(np.asarray(rpy_deg, dtype=np.float64)))
    T[:3, 3] = t
    return T


def velodyne_scan(scene: Scene, pose: np.ndarray, seed: int, n_points: int = 120_000,
                  sigma_range: float = 0.02, n_rings: int = N_RINGS,
                  n_az: int | None = None, device=None) -> np.ndarray:
    """One scan taken at world pose `pose` (4x4), returned in the SENSOR frame.

    Returns float32 [n_points, 4] (x, y, z, intensity).  Deterministic in
    (scene, pose, seed, n_points).  `device`: a torch device to ray-cast on (same scan up to floating-point
    association); None = numpy.
    """
    rng = np.random.default_rng(seed)
    if device is not None:
        _np_raycast = lambda sc, o, d: _raycast_torch(sc, o, d, device)
    else:
        _np_raycast = _raycast
    if n_az is None:
        n_az = max(8, int(round(n_points / n_rings)))
    R = pose[:3, :3]
    origin = pose[:3, 3]
    d_local = ring_directions(n_rings, n_az)
    rng_hit = _np_raycast(scene, origin, d_local @ R.T)
    keep = np.isfinite(rng_hit)
    d_keep = d_local[keep]
    r_keep = rng_hit[keep]
    # pad with random extra rays (random ring, random azimuth) until full
    elev = np.deg2rad(np.linspace(ELEV_TOP_DEG, ELEV_BOTTOM_DEG, n_rings))
    while d_keep.shape[0] < n_points:
        need = n_points - d_keep.shape[0]
        m = int(need * 1.6) + 64
        e = elev[rng.integers(0, n_rings, size=m)]
        a = rng.uniform(0.0, 2.0 * np.pi, size=m)
        dl = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], axis=-1)
        rr = _np_raycast(scene, origin, dl @ R.T)
        ok = np.isfinite(rr)
        d_keep = np.concatenate([d_keep, dl[ok][:need]], axis=0)
        r_keep = np.concatenate([r_keep, rr[ok][:need]], axis=0)
    d_keep = d_keep[:n_points]
    r_keep = r_keep[:n_points] + rng.normal(0.0, sigma_range, size=n_points)
    pts = d_keep * r_keep[:, None]
    out = np.empty((n_points, 4), dtype=np.float32)
    out[:, :3] = pts.astype(np.float32)
    out[:, 3] = rng.uniform(0.0, 1.0, size=n_points).astype(np.float32)
    return out


def scan_pair(cfg: str = "cfg2", n_points: int = 120_000, scene_seed: int = 0):
    """Named scan pairs of SURVEY.md §8(d).

    Returns (target_scan[N,4] f32, source_scan[N,4] f32, T_true 4x4) where
    T_true maps SOURCE-frame points into the TARGET frame (the reference's
    `result` convention, builder/map_builder.cc:354).
    """
    scene = make_scene(scene_seed)
    if cfg == "cfg2":
        pose_a = make_pose()
        pose_b = make_pose(t=(0.80, 0.05, 0.01), rpy_deg=(0.2, 0.2, 1.5))
        a = velodyne_scan(scene, pose_a, seed=2, n_points=n_points)
        b = velodyne_scan(scene, pose_b, seed=3, n_points=n_points)
        return a, b, np.linalg.inv(pose_a) @ pose_b
    raise ValueError(cfg)


def three_planes_pair(n_points: int = 5000, seed: int = 1, sigma: float = 0.01):
    """cfg1 plumbing case: points on three orthogonal noisy planes, known offset
    t = (0.30, -0.20, 0.10) m, rpy = (1, -2, 3) deg (SURVEY.md §8(d) cfg 1)."""
    rng = np.random.default_rng(seed)

    def cloud(n, rs):
        k = n // 3
        u = rs.uniform(0.0, 10.0, size=(n, 2))
        p = np.zeros((n, 3))
        p[:k, 0], p[:k, 1] = u[:k, 0], u[:k, 1]                    # z = 0
        p[k:2 * k, 0], p[k:2 * k, 2] = u[k:2 * k, 0], u[k:2 * k, 1]  # y = 0
        p[2 * k:, 1], p[2 * k:, 2] = u[2 * k:, 0], u[2 * k:, 1]      # x = 0
        return p + rs.normal(0.0, sigma, size=(n, 3))

    T_true = make_pose(t=(0.30, -0.20, 0.10), rpy_deg=(1.0, -2.0, 3.0))
    tgt = cloud(n_points, rng)
    src_in_tgt = cloud(n_points, rng)
    Ti = np.linalg.inv(T_true)
    src = src_in_tgt @ Ti[:3, :3].T + Ti[:3, 3]
    f = lambda p: np.concatenate([p, np.zeros((p.shape[0], 1))], axis=1).astype(np.float32)
    return f(tgt), f(src), T_true


def drive_pairs(n_pairs: int, n_points: int = 120_000, device=None, seed: int = 5):
    """`n_pairs` consecutive scan pairs (i, i + 1) of the synthetic drive (SURVEY.md 8(d) cfg 4: 10 Hz, speed 6-10 m/s, yaw rate
    +-0.2 rad/s); scan i prepared as pair i's target by the caller-side CalculateNormals (builder/map_builder.cc:286, 389).
    Each entry: src (the 120 k-point scan i + 1), q / n (target points / normals of scan i), T (true motion), guess_cv (the motion
    one frame earlier: what a constant-velocity extrapolator supplies, map_builder.cc:302-308), guess_id (identity)."""
    from .matcher import calculate_normals
    poses = drive_poses(n_pairs + 2, seed=seed, speed=8.0, speed_spread=2.0, yaw_rate_max=0.2, segment_s=1.0)
    scene = make_drive_scene(poses, seed=seed)
    scans = [velodyne_scan(scene_near(scene, P[:3, 3]), P, seed=500 + k, n_points=n_points, device=device)
             for k, P in enumerate(poses[1:])]
    rel = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(len(poses) - 1)]     # rel[k]: scan k+1 -> scan k
    pairs = []
    for k in range(n_pairs):
        q, n = calculate_normals(scans[k][:, :3].astype(np.float64))
        # pair k = (target scan k, source scan k + 1) of `scans`; its true motion is rel[k + 1], the motion one frame
        # earlier (what a constant-velocity extrapolator predicts) is rel[k]
        pairs.append(dict(src=scans[k + 1], q=q, n=n, T=rel[k + 1], guess_cv=rel[k], guess_id=np.eye(4)))
    return pairs
