"""The ctypes mirrors of the C ABI's structs (staticmapping_amd/_capi.py) against the header itself: a small C program compiled
with gcc from include/smhip.h prints sizeof and the offset of every field the mirror names; size, offsets and field order must
agree.  (The library's symbols are checked in test_capi_symbols.py; this is the other half of header / binding drift.)"""
import ctypes
import os
import subprocess

import pytest

from staticmapping_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [("smhip_icp_options", _capi.IcpOptions), ("smhip_icp_stats", _capi.IcpStats), ("smhip_icp_profile", _capi.IcpProfile),
         ("smhip_ndt_options", _capi.NdtOptions), ("smhip_ndt_stats", _capi.NdtStats), ("smhip_ndt_gicp_options", _capi.NdtGicpOptions),
         ("smhip_ndt_gicp_stats", _capi.NdtGicpStats), ("smhip_filter_desc", _capi.FilterDesc), ("smhip_mrvm_settings", _capi.MrvmSettings)]


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    d = tmp_path_factory.mktemp("layout")
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "smhip.h"', "int main(void) {"]
    for cname, cls in PAIRS:
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu %zu\\n", offsetof({cname}, {fname}), sizeof((({cname}*)0)->{fname}));')
    lines += ["  return 0;", "}"]
    src = d / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = d / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    layout = {}
    for ln in out.splitlines():
        p = ln.split()
        layout.setdefault(p[0], {})[p[1]] = tuple(int(x) for x in p[2:])
    return layout


@pytest.mark.parametrize("cname,cls", PAIRS)
def test_ctypes_mirror_has_the_headers_layout(c_layout, cname, cls):
    want = c_layout[cname]
    assert ctypes.sizeof(cls) == want["size"][0], (cname, ctypes.sizeof(cls), want["size"][0])
    covered = 0
    for fname, ftype in cls._fields_:
        off, size = want[fname]            # (a field the header does not have fails the C compile above)
        assert getattr(cls, fname).offset == off and ctypes.sizeof(ftype) == size, (cname, fname, getattr(cls, fname).offset, off, ctypes.sizeof(ftype), size)
        covered = max(covered, off + size)
    # nothing of the C struct beyond the mirror's last field but padding
    assert want["size"][0] - covered < 8, (cname, want["size"][0], covered)
