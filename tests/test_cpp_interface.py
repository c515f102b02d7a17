"""The C++ registrator::Interface mirror (include/smhip/registrator.h) driven the way
builder/map_builder.cc drives the reference, through a small C++ program."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_registrator")


def _build_exe():
    from staticmapping_amd import build
    lib = build.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_registrator.cc")
    hdr = os.path.join(ROOT, "include", "smhip", "registrator.h")
    if (not os.path.exists(EXE)) or max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(lib)) > os.path.getmtime(EXE):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                               "-L", os.path.dirname(lib), "-lsmhip", "-Wl,-rpath," + os.path.dirname(lib),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def _build_filters_exe():
    from staticmapping_amd import build
    lib = build.build()
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_filters")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_filters.cc")
    hdr = os.path.join(ROOT, "include", "smhip", "filters.h")
    if (not os.path.exists(exe)) or max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(lib)) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L", os.path.dirname(lib), "-lsmhip", "-Wl,-rpath," + os.path.dirname(lib),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_compiles_and_links():
    assert os.path.exists(_build_exe())
    assert os.path.exists(_build_filters_exe())


@pytest.mark.parametrize("name", ["test_back_end", "test_front_end", "test_mrvm", "test_registrator", "test_filters", "test_kitti_scans", "host_dry_run"])
def test_every_cpp_program_compiles_without_a_gpu(name):
    """The header-only C++ mirrors (registrator.h, back_end.h, front_end.h, mrvm.h, filters.h, kitti_scans.h) are compiled here
    through every program of tests/cpp -- the GPU tests build and run them; a header that no longer compiles shows up on the CPU."""
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cc")])


@pytest.mark.gpu
def test_cpp_filters_replay_the_reference_tests():
    """tests/cpp/test_filters.cc = the reference's five filter tests + the Factory chain, against include/smhip/filters.h"""
    out = subprocess.run([_build_filters_exe()], text=True, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["failed"] == 0


@pytest.mark.gpu
def test_cpp_interface_matches_python_path_and_oracle(tmp_path, velo20k):
    import staticmapping_amd as sm
    from oracle import cref
    exe = _build_exe()
    c = velo20k
    tgt_bin, src_bin = tmp_path / "t.bin", tmp_path / "s.bin"
    c["tgt"].astype(np.float32).tofile(tgt_bin)
    c["src"].astype(np.float32).tofile(src_bin)
    xml = '<param name="max_iteration"> 30 </param><param name="dist_outlier_ratio"> 0.7 </param>'
    out = subprocess.check_output([exe, str(tgt_bin), str(src_bin), "0.6", xml], text=True, timeout=300)
    res = json.loads(out.strip().splitlines()[-1])
    assert res["ok"] and res["type"] == 6
    assert res["unknown_option_check"] and res["wrong_type_null"] and res["no_normals_check"]
    assert res["refused_target_fails"] and res["recovers_after_good_target"]
    assert res["compensation_refused"] and res["compensation_off_runs"]          # EnableInnerCompensation is refused, not ignored
    R = np.array(res["result"]).reshape(4, 4)
    # same clouds through the oracle: target prepared by the product's host CalculateNormals
    q, n = sm.calculate_normals(c["tgt"][:, :3].astype(np.float64))
    assert res["target_points"] >= len(q)
    ref = cref.icp_fast_align(c["src"][:, :3].astype(np.float64), q, n, guess=c["guess"], max_iteration=30)
    da, dt = sm.se3_error(R, ref["result"])
    # the C++ path keeps leaves with non-finite normals exactly like the reference does; allow for that
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert abs(res["score"] - ref["score"]) < 1e-3
    # registrators::IcpUsingPointMatcher chain (sampling off) vs its restatement
    from oracle import icp_pointmatcher as opm
    ok_o, R_o, score_o, _ = opm.align(c["src"], c["tgt"], c["guess"], np.ones(len(c["src"]), bool),
                                      normals_fn=lambda p: sm.calculate_normals(p) + (None,))
    Rp = np.array(res["pm_result"]).reshape(4, 4)
    da, dt = sm.se3_error(Rp, R_o)
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert abs(res["pm_score"] - score_o) < 1e-4 and res["pm_ok"] == ok_o
    # registrators::Ndt through the same C++ surface vs the numpy restatement
    from oracle import ndt as ondt
    assert res["ndt_ok"] and res["ndt_type"] == 5
    nref = ondt.ndt_align(c["src"], c["tgt"], guess=c["guess"])
    Rn = np.array(res["ndt_result"]).reshape(4, 4)
    da, dt = sm.se3_error(Rn, nref["result"])
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    assert abs(res["ndt_score"] - nref["score"]) <= 1e-3 * nref["score"]
    # registrators::NdtWithGicp through the same C++ surface vs its restatement
    from oracle import ndt_gicp as ong
    assert res["gicp_type"] == 3
    gref = ong.ndt_gicp_align(c["src"][:, :3], c["tgt"][:, :3], c["guess"])
    assert res["gicp_ok"] == gref["ok"]
    Rg = np.array(res["gicp_result"]).reshape(4, 4)
    da, dt = sm.se3_error(Rg, gref["result"])
    assert da < 3e-3 and dt < 5e-2, (da, dt)            # GICP's own repeatability, see tests/test_ndt_gicp_gpu.py
    assert abs(res["gicp_score"] - gref["score"]) < 2e-2


@pytest.mark.gpu
def test_cpp_interface_with_the_reference_search_semantics(tmp_path, velo20k):
    """The XML a maintainer would write to get the reference's own neighbour search (libnabo's tree, epsilon = 3.16) from
    the C++ matcher: `nn_mode` 2 / `nn_epsilon`, against the oracle run with the same search."""
    import staticmapping_amd as sm
    from oracle import cref
    exe = _build_exe()
    c = velo20k
    tgt_bin, src_bin = tmp_path / "t.bin", tmp_path / "s.bin"
    c["tgt"].astype(np.float32).tofile(tgt_bin)
    c["src"].astype(np.float32).tofile(src_bin)
    xml = '<param name="nn_mode"> 2 </param><param name="nn_epsilon"> 3.16 </param><!-- <param name="bogus"> 1 </param> -->'
    out = subprocess.check_output([exe, str(tgt_bin), str(src_bin), "0.6", xml], text=True, timeout=300)
    res = json.loads(out.strip().splitlines()[-1])
    assert res["ok"] and res["type"] == 6
    R = np.array(res["result"]).reshape(4, 4)
    q, n = sm.calculate_normals(c["tgt"][:, :3].astype(np.float64))
    ref = cref.icp_fast_align(c["src"][:, :3].astype(np.float64), q, n, guess=c["guess"], nn_eps=3.16)
    da, dt = sm.se3_error(R, ref["result"])
    assert da < 1e-4 and dt < 1e-3, (da, dt)
    exact = cref.icp_fast_align(c["src"][:, :3].astype(np.float64), q, n, guess=c["guess"])
    assert sm.se3_error(R, exact["result"])[1] > 1e-3          # and it is NOT the exact-search result
