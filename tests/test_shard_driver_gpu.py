"""BASELINE config #4's driver: the C++ program (staticmapping_amd/csrc/shard_driver.cc: KITTI .bin in, device
CalculateNormals + IcpFast per pair, ONE ncclAllGather of the poses, kitti_pose.txt out) against the Python driver, and a
two-rank run with the real matcher on every rank."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _drive(tmp_path, n_scans=7, n_points=20000):
    from staticmapping_amd import synth, kitti
    d = tmp_path / "seq"
    d.mkdir()
    poses = synth.drive_poses(n_scans, seed=5, speed=8.0, hz=10.0, yaw_rate_max=0.2)
    scene = synth.make_drive_scene(poses, seed=5)
    for k, P in enumerate(poses):
        kitti.write_bin(kitti.scan_path(str(d), k), synth.velodyne_scan(synth.scene_near(scene, P[:3, 3]), P, seed=1000 + k, n_points=n_points))
    return str(d), poses


def test_shard_driver_is_built_and_linked_against_rccl():
    from staticmapping_amd import build
    build.build()
    exe = build.build_shard_driver()
    assert os.path.exists(exe)
    ldd = subprocess.check_output(["ldd", exe], text=True)
    assert "librccl" in ldd and "libsmhip" in ldd and "not found" not in ldd
    r = subprocess.run([exe, "--scans", "/nonexistent-dir"], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_cpp_driver_matches_the_python_driver(tmp_path):
    import staticmapping_amd as sm
    from staticmapping_amd import build, kitti, shard
    seq, poses = _drive(tmp_path)
    exe = build.build_shard_driver()
    out = tmp_path / "kitti_pose.txt"
    r = subprocess.run([exe, "--scans", seq, "--gpus", "1", "--out", str(out), "--batch", "4", "--iterations", "20", "--guess-tx", "0.6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["pairs"] == 6 and line["unfinished_pairs"] == 0 and line["mean_iterations"] == 20
    got = kitti.read_poses(str(out))
    assert got.shape == (7, 4, 4) and np.array_equal(got[0], np.eye(4))
    # the Python driver on the same files takes the same steps: every scan uploaded once as a source, the first target of a
    # batch parked in a spare slot, the targets of a batch prepared in one device pass, batches of 4, 20 fixed iterations
    files = kitti.list_scans(seq)
    scans = [(lambda f=f: kitti.read_bin(f, scale_intensity=False)) for f in files]
    G = np.eye(4); G[0, 3] = 0.6
    m = sm.IcpFastHip(pair_slots=5, max_source_points=32768, max_target_points=32768, max_iteration=20, early_exit=0)
    idx, T, sc, it = kitti.scan_to_scan_sequence(scans, m, batch=4, guesses=[G] * 6)
    m.close()
    traj = shard.chain_poses(T)
    for k in range(7):                                                    # the file carries 8 significant digits
        da, dt = sm.se3_error(got[k], traj[k])
        assert da < 1e-6 and dt < 1e-5, (k, da, dt)
    truth = np.linalg.inv(poses[0]) @ poses[-1]
    assert np.linalg.norm(got[-1][:3, 3] - truth[:3, 3]) < 0.08
    # the warm-up batch the driver sends through the same calls before its clock starts leaves nothing behind: same poses without it
    assert line["warmup_batch_before_the_clock_s"] > 0
    out0 = tmp_path / "kitti_pose_no_warmup.txt"
    r0 = subprocess.run([exe, "--scans", seq, "--gpus", "1", "--out", str(out0), "--batch", "4", "--iterations", "20", "--guess-tx", "0.6", "--warmup", "0"],
                        capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0, r0.stderr
    assert json.loads(r0.stdout.strip().splitlines()[-1])["warmup_batch_before_the_clock_s"] == 0
    assert out0.read_text() == out.read_text()


@pytest.mark.gpu
def test_two_ranks_with_the_real_matcher(tmp_path):
    """Two processes, each with its own IcpFastHip on the one GPU, pairs round-robin, one gather: the interleaved rows
    equal a single-process run pair by pair."""
    import staticmapping_amd as sm
    from staticmapping_amd import kitti, shard
    import torch
    seq, poses = _drive(tmp_path, n_scans=6)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "rows.npy"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "shard_worker.py"), seq, str(out)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = np.load(out)
    assert rows.shape == (5, 18)
    T2, s2, it2 = shard.unpack_pose_rows(torch.from_numpy(rows))
    files = kitti.list_scans(seq)
    scans = [(lambda f=f: kitti.read_bin(f)) for f in files]
    G = np.eye(4); G[0, 3] = 0.6
    m = sm.IcpFastHip(device=0, pair_slots=3, max_source_points=32768, max_target_points=32768, max_iteration=20, early_exit=0)
    idx, T1, s1, it1 = kitti.scan_to_scan_sequence(scans, m, batch=2, guesses=[G] * 5)
    m.close()
    assert list(it2) == [20] * 5
    for k in range(5):
        da, dt = sm.se3_error(T2[k], T1[k])
        assert da < 1e-7 and dt < 1e-6, (k, da, dt)              # same kernels; only the batch composition differs
        assert abs(s2[k] - s1[k]) < 1e-7


@pytest.mark.gpu
def test_bench_runs_its_multi_rank_sequence_to_the_end(tmp_path):
    """bench.py under torch.distributed.run with two ranks, as the driver launches it for N > 1 -- on this one-GPU box in its
    dry-run mode (SMHIP_BENCH_SHARED_GPU=1: the ranks share the device, gloo carries the collectives): every rank must
    reach the same barriers, gathers, the reduce of the timing and the broadcast of the profiled class, and rank 0 must
    print one line for two GPUs whose poses are right."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SMHIP_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--pairs", "32", "--distinct", "8", "--cpu-pairs", "2"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]                  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["parity"]["worst_trans_err_vs_truth_m"] < 0.05
    assert "cpu_baseline" not in d                           # the CPU legs are a one-GPU report
    assert set(d["figures"]) >= {"extrapolated_guess", "identity_guess", "reference_search_eps3.16", "early_exit"}
    # the line certifies what the collective library saw: an all-reduce of ones over the communicator counted both ranks
    assert d["config"]["rccl_ranks"] == 2 and len(d["config"]["rank_devices"]) == 2 and len(d["config"]["split_after_by_rank"]) == 2
    assert len(set(d["config"]["profiled_class_by_rank"])) == 1
