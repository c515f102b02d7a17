"""MapBuilder::ScanMatchProcessing restated over the GPU matcher (include/smhip/front_end.h): key-frame logic, CTRV
extrapolation and pose chaining over a synthetic drive, against the drive's true poses."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_exe():
    from staticmapping_amd import build
    lib = build.build()
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_front_end")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_front_end.cc")
    hdrs = [os.path.join(ROOT, "include", "smhip", h) for h in ("front_end.h", "back_end.h", "registrator.h")]
    if (not os.path.exists(exe)) or max([os.path.getmtime(src), os.path.getmtime(lib)] + [os.path.getmtime(h) for h in hdrs]) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L", os.path.dirname(lib), "-lsmhip", "-Wl,-rpath," + os.path.dirname(lib),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_front_end_mirror_compiles():
    assert os.path.exists(_build_exe())


@pytest.mark.gpu
def test_front_end_drive(tmp_path):
    import staticmapping_amd as sm
    from staticmapping_amd import synth, kitti
    scene = synth.make_scene(0)
    n = 9
    poses = [synth.make_pose(t=(0.3 * k, 0.01 * k, 0.0), rpy_deg=(0, 0, 0.4 * k)) for k in range(n)]   # 3 m/s at 10 Hz
    for k, P in enumerate(poses):
        kitti.write_bin(kitti.scan_path(str(tmp_path), k), synth.velodyne_scan(scene, P, seed=120 + k, n_points=30000))
    # the drive's rough speed primes the extrapolator (PoseExtrapolator::InitRoughLinearVelocity): with a zero first guess the
    # trimmed point-to-plane ICP stalls along the road in this scene (the reference's early-exit rule, see DESIGN_HISTORY.md section 2)
    out = subprocess.check_output([_build_exe(), str(n), str(tmp_path), "3.0", "0.1"], text=True, timeout=600)
    frames = json.loads(out.strip().splitlines()[-1])["frames"]
    assert len(frames) == n and frames[0]["key"] and not frames[0]["matched"]
    keys = [k for k, f in enumerate(frames) if f["key"]]
    # 0.3 m per scan against translation_range 0.5: every second scan becomes a key frame
    assert keys == [0, 2, 4, 6, 8], keys
    for k in range(1, n):
        P = np.array(frames[k]["pose"]).reshape(4, 4)
        da, dt = sm.se3_error(P, np.linalg.inv(poses[0]) @ poses[k])
        assert frames[k]["matched"] and frames[k]["score"] > 0.8
        assert da < 4e-3 and dt < 0.06, (k, da, dt)
    # the GPU-resident key frame (device normals, no per-scan target upload) follows the same drive: the device and the host
    # CalculateNormals keep the same points and agree on the normals to float rounding (tests/test_prepare_target_gpu.py)
    out2 = subprocess.check_output([_build_exe(), str(n), str(tmp_path), "3.0", "0.1", "1"], text=True, timeout=600)
    frames2 = json.loads(out2.strip().splitlines()[-1])["frames"]
    assert [k for k, f in enumerate(frames2) if f["key"]] == keys
    for k in range(1, n):
        da, dt = sm.se3_error(np.array(frames2[k]["pose"]).reshape(4, 4), np.array(frames[k]["pose"]).reshape(4, 4))
        assert da < 5e-4 and dt < 5e-3, (k, da, dt)
    host_ms = np.median([f["ms"] for f in frames[2:]])
    dev_ms = np.median([f["ms"] for f in frames2[2:]])
    print("front end, median ms per scan (30 k points): host target prep %.2f, device-resident %.2f" % (host_ms, dev_ms))
