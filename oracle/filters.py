"""numpy restatement of the front end's pre-filters -- TEST ORACLE (see oracle/__init__.py).

Unlike the registrators, these filters have reference tests, and this oracle is PINNED on them: the known answers
of /root/reference/pre_processors/test/test_filter_{range,axis_range,bounding_box,random_sample,voxel_grid}.cc are
replayed against it in tests/test_oracle_filters.py (identity with default parameters, ConfigsValid() verdicts,
the 100 / 36 / 9 voxel counts of the 10 x 10 lattice, the empty output of a default BoundingBoxRemoval, the
0.48 .. 0.52 keep fraction of RandomSampler(0.5), bounds of every kept point).

Restates (paths relative to /root/reference/pre_processors)
  filter_range.cc:46-91          Range
  filter_axis_range.cc:45-103    AxisRange
  filter_bounding_box.cc:53-83   BoundingBoxRemoval (common/bounding_box.cc:117-121)
  filter_random_sample.cc:41-85  RandomSampler -- the reference draws from a mt19937 seeded by std::random_device on
                                 every call, so no stream is reproducible; the law (one uniform per input point, keep
                                 when u <= rate) is kept with the counter-based generator the device uses
  filter_voxel_grid.cc:38-80     VoxelGrid (output order unspecified there: unordered_map; sorted by voxel here)
  filter_factory.cc:83-106       Factory::Filter
Clouds are float32 [N, 5] arrays of InnerPointType rows (x, y, z, intensity, factor).
"""
from __future__ import annotations

import numpy as np

F = np.float32
FLT_MAX = np.finfo(np.float32).max
RANGE, AXIS_RANGE, RANDOM_SAMPLER, VOXEL_GRID, BOUNDING_BOX_REMOVAL = 1, 2, 3, 4, 5


def default(kind: int) -> dict:
    """constructor defaults"""
    return {RANGE: dict(type=RANGE, min_range=0.0, max_range=FLT_MAX),
            AXIS_RANGE: dict(type=AXIS_RANGE, min=-FLT_MAX, max=FLT_MAX, axis_index=2),
            RANDOM_SAMPLER: dict(type=RANDOM_SAMPLER, sampling_rate=1.0, seed=0),
            VOXEL_GRID: dict(type=VOXEL_GRID, voxel_size=0.1),
            BOUNDING_BOX_REMOVAL: dict(type=BOUNDING_BOX_REMOVAL, min_x=-FLT_MAX, min_y=-FLT_MAX, min_z=-FLT_MAX,
                                       max_x=FLT_MAX, max_y=FLT_MAX, max_z=FLT_MAX)}[kind].copy()


def config_valid(f: dict) -> bool:
    t = f["type"]
    if t == RANGE:
        return True
    if t == AXIS_RANGE:
        return bool(F(f["max"]) > F(f["min"])) and 0 <= f["axis_index"] <= 2
    if t == RANDOM_SAMPLER:
        return bool(F(0) <= F(f["sampling_rate"]) <= F(1))
    if t == VOXEL_GRID:
        return bool(F(f["voxel_size"]) > 1.e-6)
    if t == BOUNDING_BOX_REMOVAL:
        return bool(F(f["min_x"]) < F(f["max_x"]) and F(f["min_y"]) < F(f["max_y"]) and F(f["min_z"]) < F(f["max_z"]))
    return False


def sampler_uniform(seed: int, n: int) -> np.ndarray:
    """splitmix64 of (seed << 32 | i): the device's counter-based stream, 53-bit uniforms in [0, 1)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) << np.uint64(32) | np.arange(n, dtype=np.uint64)) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def keep_mask(f: dict, pts: np.ndarray) -> np.ndarray:
    t = f["type"]
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    if t == RANGE:
        with np.errstate(over="ignore", invalid="ignore"):
            r = np.sqrt((x * x + y * y) + z * z)                                   # float, left to right
        return (r >= F(f["min_range"])) & (r <= F(f["max_range"]))
    if t == AXIS_RANGE:
        v = pts[:, f["axis_index"]]
        return ~((v < F(f["min"])) | (v > F(f["max"])))
    if t == BOUNDING_BOX_REMOVAL:
        lo = np.array([F(f["min_x"]), F(f["min_y"]), F(f["min_z"])], dtype=np.float64)
        hi = np.array([F(f["max_x"]), F(f["max_y"]), F(f["max_z"])], dtype=np.float64)
        p = pts[:, :3].astype(np.float64)
        return ~np.all((p >= lo) & (p <= hi), axis=1)
    if t == RANDOM_SAMPLER:
        if float(F(f["sampling_rate"])) > 0.999:      # filter_random_sample.cc:44: the float rate against the double literal
            return np.ones(len(pts), dtype=bool)
        return sampler_uniform(f.get("seed", 0), len(pts)) <= np.float64(F(f["sampling_rate"]))
    raise ValueError(t)


def voxel_grid(pts: np.ndarray, voxel_size: float) -> np.ndarray:
    s = F(voxel_size)
    q = pts[:, :3] / s                                                              # float division
    ijk = np.where(q >= 0, np.floor(q.astype(np.float64) + 0.5), np.ceil(q.astype(np.float64) - 0.5)).astype(np.int64)   # lround
    order = np.lexsort((np.arange(len(pts)), ijk[:, 2], ijk[:, 1], ijk[:, 0]))      # by voxel, arrival order inside
    sk = ijk[order]
    head = np.ones(len(pts), dtype=bool)
    head[1:] = np.any(sk[1:] != sk[:-1], axis=1)
    starts = np.flatnonzero(head)
    ends = np.append(starts[1:], len(pts))
    out = np.zeros((len(starts), 5), dtype=F)
    p64 = pts[:, :4].astype(np.float64)
    for v, (a, b) in enumerate(zip(starts, ends)):
        acc = np.zeros(4)
        for k in order[a:b]:
            acc += p64[k]                                                           # double sums in arrival order
        out[v, :4] = (acc / (b - a)).astype(F)
    return out


def run_chain(points5: np.ndarray, chain: list[dict]):
    """Factory::Filter: returns (filtered [M,5] float32, source_index [M] int32; -1 after a VoxelGrid)."""
    pts = np.asarray(points5, dtype=F)
    src = np.arange(len(pts), dtype=np.int32)
    for f in chain:
        if not config_valid(f):
            raise ValueError(f"ConfigsValid() is false for {f}")
        if len(pts) == 0:
            break
        if f["type"] == VOXEL_GRID:
            pts = voxel_grid(pts, f["voxel_size"])
            src = np.full(len(pts), -1, dtype=np.int32)
        else:
            m = keep_mask(f, pts)
            pts, src = pts[m], src[m]
    return pts, src


def with_factor(rows4: np.ndarray) -> np.ndarray:
    """KITTI rows (x y z intensity) -> InnerPointType rows, factor = i / size (data_collector.h:202-204)."""
    r = np.asarray(rows4, dtype=F)
    n = len(r)
    return np.concatenate([r[:, :4], (np.arange(n, dtype=np.float64) / max(n, 1)).astype(F)[:, None]], axis=1)
