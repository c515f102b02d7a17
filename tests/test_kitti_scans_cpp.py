"""The KITTI `.bin` reader and the read-ahead pool of the sharded sequence driver (include/smhip/kitti_scans.h; reference
ros_node/kitti_reader.cc:91-149) are host-only C++: compiled with g++ and exercised on the CPU box."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kitti_scan_reader_and_prefetcher(tmp_path):
    build = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "test_kitti_scans")
    src = os.path.join(ROOT, "tests", "cpp", "test_kitti_scans.cc")
    hdr = os.path.join(ROOT, "include", "smhip", "kitti_scans.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe, str(tmp_path)], text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
