"""Regenerates tests/golden/*.json from the numpy oracle (oracle/icp_fast.py, oracle/ndt.py).

The reference has no golden vectors of its own, so these pin the ORACLE against
regressions and give the GPU path a committed target; they are not reference outputs.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from staticmapping_amd import synth          # noqa: E402
from oracle import icp_fast as o, ndt as ondt, ndt_gicp as ong, filters as of    # noqa: E402
import hashlib    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def icp_case():
    tgt, src, T = synth.three_planes_pair(2000, seed=11, sigma=0.01)
    q, n, _ = o.calculate_normals(tgt[:, :3].astype(np.float64))
    trace = []
    R, score, it = o.icp_fast_align(src[:, :3].astype(np.float64), q, n, trace=trace)
    first = trace[0]
    return dict(generator="synth.three_planes_pair(2000, seed=11, sigma=0.01); oracle.icp_fast.calculate_normals; icp_fast_align(defaults)",
                target_points=int(len(q)), result=R.tolist(), score=score, iterations=it,
                first_limit_d2=first["limit"], first_kept=int(first["keep"].sum()),
                first_A_trace=float(np.trace(first["A"])), truth=T.tolist())


def ndt_case():
    scene = synth.make_scene(0)
    poses = [synth.make_pose(t=(0.8 * k, 0.0, 0.0)) for k in range(3)]
    scans = [synth.velodyne_scan(scene, P, seed=20 + k, n_points=8000) for k, P in enumerate(poses)]
    tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:2], poses[:2])]).astype(np.float32)
    G = poses[2].copy(); G[0, 3] -= 0.25
    r = ondt.ndt_align(scans[2], tgt, guess=G)
    grid = ondt.VoxelGrid(tgt)
    return dict(generator="two 8000-pt scans merged as target, third as source, guess 0.25 m short; oracle.ndt.ndt_align(defaults)",
                voxels=int(len(grid.mean)), result=r["result"].tolist(), score=r["score"], iterations=r["iterations"],
                derivative_calls=r["derivative_calls"], trans_probability=r["trans_probability"])


def ndt_gicp_inputs():
    a, b, T = synth.scan_pair("cfg2", n_points=12000)
    return b[:, :3].copy(), a[:, :3].copy(), synth.make_pose(t=(0.6, 0, 0)), T


def ndt_gicp_case():
    src, tgt, G, T = ndt_gicp_inputs()
    r = ong.ndt_gicp_align(src, tgt, G)
    ds = ong.approximate_voxel_grid(src, 0.2)
    return dict(generator="synth.scan_pair('cfg2', n_points=12000); guess 0.6 m along x; oracle.ndt_gicp.ndt_gicp_align(defaults)",
                n_source=r["n_source"], n_target=r["n_target"], ok=r["ok"], result=r["result"].tolist(), score=r["score"],
                ndt_iterations=r["ndt"]["iterations"], ndt_score=r["ndt"]["score"], gicp_iterations=r["gicp"]["iterations"],
                downsampled_source_sha256=hashlib.sha256(ds.tobytes()).hexdigest(), truth=T.tolist())


def filter_inputs():
    a, b, T = synth.scan_pair("cfg2", n_points=12000)
    chain = [dict(of.default(of.RANGE), min_range=5.0), dict(of.default(of.AXIS_RANGE), min=-2.0),
             dict(of.default(of.RANDOM_SAMPLER), sampling_rate=0.5, seed=21), dict(of.default(of.VOXEL_GRID), voxel_size=0.3)]
    return np.ascontiguousarray(a[:, :4]), chain


def filter_case():
    rows, chain = filter_inputs()
    out3, src3 = of.run_chain(of.with_factor(rows), chain[:3])
    out4, _ = of.run_chain(of.with_factor(rows), chain)
    return dict(generator="synth.scan_pair('cfg2', n_points=12000)[0] as KITTI rows; Range(5) -> AxisRange(z >= -2) -> RandomSampler(0.5, seed 21) [-> VoxelGrid(0.3)]",
                n_after_sampler=int(len(out3)), sampler_sha256=hashlib.sha256(out3.tobytes()).hexdigest(),
                index_sha256=hashlib.sha256(src3.tobytes()).hexdigest(),
                n_after_voxel_grid=int(len(out4)), voxel_sha256=hashlib.sha256(out4.tobytes()).hexdigest())


if __name__ == "__main__":
    json.dump(ndt_gicp_case(), open(os.path.join(HERE, "ndt_gicp_cfg2_12000.json"), "w"), indent=1)
    json.dump(filter_case(), open(os.path.join(HERE, "filters_kitti_chain_12000.json"), "w"), indent=1)
    json.dump(icp_case(), open(os.path.join(HERE, "icp_three_planes_2000.json"), "w"), indent=1)
    json.dump(ndt_case(), open(os.path.join(HERE, "ndt_two_scans_8000.json"), "w"), indent=1)
    print("written")
