"""KITTI odometry wire formats either side of the registration path.

* `.bin` scans: float32 rows x, y, z, reflectance -- read exactly like the reference's KittiReader
  (/root/reference/ros_node/kitti_reader.cc:91-121: at most 1 000 000 floats per file, intensity x 255,
  files addressed as `%010d.bin` or in sorted directory order, :124-149).
* `kitti_pose.txt`: 12 floats per line, the row-major top 3x4 of each pose with setprecision(8)
  (/root/reference/builder/map_builder.cc:626-641).

`scan_to_scan_sequence` is the sharded benchmark driver of BASELINE config #4: consecutive pairs
(i, i+1) are independent units dealt round-robin over the ranks (shard.pairs_of_rank), aligned in
batches on each GPU, gathered once and chained into a trajectory on the host.
"""
from __future__ import annotations

import os
import numpy as np

from . import shard

MAX_FLOATS_PER_FILE = 1_000_000          # kitti_reader.cc:93


def read_bin(path: str, scale_intensity: bool = True) -> np.ndarray:
    """One scan as float32 [N, 4] (x, y, z, intensity * 255)."""
    data = np.fromfile(path, dtype=np.float32, count=MAX_FLOATS_PER_FILE)
    n = data.size // 4                                                  # :104
    pts = data[:4 * n].reshape(n, 4).copy()
    if scale_intensity:
        pts[:, 3] *= 255.0                                              # :110
    return pts


def write_bin(path: str, points: np.ndarray) -> None:
    np.ascontiguousarray(points[:, :4], dtype=np.float32).tofile(path)


def list_scans(directory: str) -> list[str]:
    """Sorted directory listing, as KittiReader::SetPointCloudDataPath does (:124-131)."""
    return sorted(os.path.join(directory, f) for f in os.listdir(directory) if f.endswith(".bin"))


def scan_path(directory: str, index: int) -> str:
    return os.path.join(directory, f"{index:010d}.bin")                 # :136-137


def write_poses(path: str, poses: np.ndarray) -> None:
    with open(path, "w") as f:
        for line in shard.poses_to_kitti_lines(poses):
            f.write(line + "\n")


def read_poses(path: str) -> np.ndarray:
    rows = np.loadtxt(path).reshape(-1, 12)
    poses = np.tile(np.eye(4), (len(rows), 1, 1))
    poses[:, :3, :] = rows.reshape(-1, 3, 4)
    return poses


def scan_to_scan_sequence(scans, matcher, batch: int, guesses=None, rank: int = 0, world: int = 1,
                          prepare_target=None, filter_chain=None):
    """Align every consecutive pair (scan i = target, scan i+1 = source) owned by this rank.

    scans: list of float32 [N,4] arrays (or callables returning one, for lazy loading);
    matcher: an IcpFastHip with at least `batch` pair slots;  guesses: optional [n_pairs,4,4].
    Target preparation (CalculateNormals, builder/map_builder.cc:286,389) runs on the GPU; when the
    previous pair of the chunk already uploaded scan i as its source, that resident copy is re-used.
    `prepare_target`: optional host callable scan -> (points, normals) to override that.
    `filter_chain`: optional list of filters (staticmapping_amd.filters.make_filter / chain_from_xml): the front end's
    <filters> chain (config/lidar_only_kitti.xml:18-41) applied on the device to every scan as it is loaded; the
    filtered cloud goes to its slot without returning to the host.
    Returns (pair_indices, transforms [k,4,4], scores [k], iterations [k]) for this rank's pairs.
    """
    n_pairs = len(scans) - 1
    mine = shard.pairs_of_rank(n_pairs, rank, world)
    get = lambda i: scans[i]() if callable(scans[i]) else scans[i]
    out_T, out_s, out_it = [], [], []
    spare = matcher.pair_slots - 1                         # holds a target scan nobody uploaded as a source
    if filter_chain is not None:
        from . import filters as _filters

    def load_source(index, slot):
        if filter_chain is None:
            matcher.set_input_source(get(index), slot=slot)
        else:
            _filters.run_chain_resident(matcher, get(index), filter_chain)
            _filters.output_to_source(matcher, slot)
    for b0 in range(0, len(mine), batch):
        chunk = mine[b0:b0 + batch]
        g = []
        for s, pair in enumerate(chunk):                  # every source once
            load_source(pair + 1, s)
            g.append(np.eye(4) if guesses is None else guesses[pair])
        if prepare_target is not None:
            for s, pair in enumerate(chunk):
                q, n = prepare_target(get(pair))
                matcher.set_input_target(q, n, slot=s)
        else:
            # targets: scan `pair` is the previous slot's source when the chunk is consecutive; all of those are
            # prepared in ONE batched device pass, the others are uploaded and prepared one by one
            fr, to = [], []
            for s, pair in enumerate(chunk):
                if s > 0 and chunk[s - 1] == pair - 1:
                    fr.append(s - 1); to.append(s)
                elif spare >= len(chunk) and not fr:      # first pair of a consecutive chunk: park its target scan
                    load_source(pair, spare)
                    fr.append(spare); to.append(s)
                elif filter_chain is None:
                    matcher.prepare_target(get(pair), slot=s)
                else:                                     # filtered target with no resident copy: park it in the spare slot first
                    if fr:
                        matcher.prepare_targets_from_sources(fr, to); fr, to = [], []
                    load_source(pair, spare)
                    matcher.prepare_targets_from_sources([spare], [s])
            if fr:
                matcher.prepare_targets_from_sources(fr, to)
        T, sc, st = matcher.align_batch(len(chunk), g)
        out_T.append(T); out_s.append(sc); out_it.append([x["iterations"] for x in st])
    if not out_T:
        return [], np.zeros((0, 4, 4)), np.zeros(0), np.zeros(0, int)
    return mine, np.concatenate(out_T), np.concatenate(out_s), np.concatenate(out_it).astype(int)
