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


def test_host_side_dry_run_at_8_ranks(tmp_path, capsys):
    """What the ranks of smhip_shard do on the HOST (read order, read-ahead pools, staging copies) for 1, 2 and 8 concurrent
    ranks on this box's cores, no GPU: every rank reads exactly its files, and the aggregate scans/s is printed -- the
    host-side ceiling of an 8-GPU node (tests/cpp/host_dry_run.cc)."""
    import json
    build = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "host_dry_run")
    src = os.path.join(ROOT, "tests", "cpp", "host_dry_run.cc")
    hdr = os.path.join(ROOT, "include", "smhip", "kitti_scans.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe, str(tmp_path), "65", "120000", "4", "6"], text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0
    runs = {r["ranks"]: r for r in res["runs"]}
    assert runs[1]["scans_read"] == 6 * 65 and runs[8]["scans_read"] == 6 * 2 * 64   # one rank re-uses the previous source; 8 ranks read both scans of a pair
    with capsys.disabled():
        print("\n[host dry run] " + "; ".join(f"{r['ranks']} rank(s): {r['scans_per_s']:.0f} scans/s = {r['pairs_per_s']:.0f} pairs/s, "
                                              f"{r['GB_per_s']:.2f} GB/s" for r in res["runs"]))
