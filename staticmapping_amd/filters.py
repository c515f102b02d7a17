"""ctypes view of the device pre-filters (include/smhip.h, `smhip_filter_*`): the Python mirror of
pre_processers::filter::{Range, AxisRange, BoundingBoxRemoval, RandomSampler, VoxelGrid, Factory}
(/root/reference/pre_processors/filter_*.cc).  Filters are dicts with the reference's parameter names."""
from __future__ import annotations

import ctypes
import re

import numpy as np

from . import _capi

RANGE, AXIS_RANGE, RANDOM_SAMPLER, VOXEL_GRID, BOUNDING_BOX_REMOVAL = 1, 2, 3, 4, 5
NAMES = {"Range": RANGE, "AxisRange": AXIS_RANGE, "RandomSampler": RANDOM_SAMPLER, "VoxelGrid": VOXEL_GRID,
         "BoundingBoxRemoval": BOUNDING_BOX_REMOVAL}
_PARAMS = {RANGE: ("min_range", "max_range"), AXIS_RANGE: ("min", "max"), RANDOM_SAMPLER: ("sampling_rate",),
           VOXEL_GRID: ("voxel_size",), BOUNDING_BOX_REMOVAL: ("min_x", "min_y", "min_z", "max_x", "max_y", "max_z")}


def make_filter(name_or_type, **params) -> _capi.FilterDesc:
    """A filter with its constructor defaults, then `params` (reference names; plus `axis_index`, `seed`)."""
    t = NAMES[name_or_type] if isinstance(name_or_type, str) else int(name_or_type)
    d = _capi.FilterDesc()
    _capi.load_library().smhip_filter_default(t, ctypes.byref(d))
    for k, v in params.items():
        if k == "axis_index":
            d.axis_index = int(v)
        elif k == "seed":
            d.seed = int(v)
        elif k in _PARAMS[t]:
            d.p[_PARAMS[t].index(k)] = float(v)
        else:
            raise KeyError(f"{k} is not a parameter of filter type {t}")       # SetValue -> CHECK(all_right), filter_interface.cc:58
    return d


def config_valid(d: _capi.FilterDesc) -> bool:
    return bool(_capi.load_library().smhip_filter_config_valid(ctypes.byref(d)))


def chain_from_xml(text: str, seed: int = 0) -> list:
    """The <filters> element of the reference's configs (filter_factory.cc:47-81): unsupported names are skipped."""
    out = []
    for m in re.finditer(r'<filter\s+name="([^"]+)"\s*(?:/>|>(.*?)</filter>)', text, flags=re.S):
        name, body = m.group(1), m.group(2) or ""
        if name not in NAMES:
            continue
        params = {}
        for p in re.finditer(r'<param\s+type="(\d)"\s+name="([^"]+)"\s*>\s*([^<]*?)\s*</param>', body):
            params[p.group(2)] = int(float(p.group(3))) if p.group(1) == "0" else float(p.group(3))
        if NAMES[name] == RANDOM_SAMPLER:
            params.setdefault("seed", seed + len(out))
        out.append(make_filter(name, **params))
    return out


def run_chain(matcher, points, chain):
    """Factory::Filter on the device.  points: float32 [N,4] (KITTI rows) or [N,5] (InnerPointType rows).
    Returns (filtered [M,5] float32, source_index [M] int32)."""
    a = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
    arr = (_capi.FilterDesc * max(1, len(chain)))(*chain)
    n_out = ctypes.c_int()
    matcher._check(matcher._lib.smhip_filter_chain_f32(matcher._h, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0],
                                                       arr, len(chain), ctypes.byref(n_out)))
    out = np.zeros((n_out.value, 5), np.float32)
    src = np.zeros(n_out.value, np.int32)
    matcher._check(matcher._lib.smhip_filter_get_output(matcher._h, out.ctypes.data_as(_capi.c_float_p),
                                                        src.ctypes.data_as(_capi.c_int32_p), n_out.value))
    return out, src


def run_chain_resident(matcher, points, chain) -> int:
    """The same without reading the result back: returns the filtered size; follow with output_to_source()."""
    a = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
    arr = (_capi.FilterDesc * max(1, len(chain)))(*chain)
    n_out = ctypes.c_int()
    matcher._check(matcher._lib.smhip_filter_chain_f32(matcher._h, a.ctypes.data_as(_capi.c_float_p), a.shape[1], a.shape[0],
                                                       arr, len(chain), ctypes.byref(n_out)))
    return n_out.value


def output_to_source(matcher, slot: int = 0):
    matcher._check(matcher._lib.smhip_filter_output_to_source(matcher._h, slot))
