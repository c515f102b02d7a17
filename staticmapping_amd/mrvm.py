"""Python view of static_map::MultiResolutionVoxelMap on the GPU (/root/reference/builder/multi_resolution_voxel_map.{h,cc}):
Initialise (constructor) / insert_point_cloud / output_to_point_cloud, all compute inside libsmhip.so."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _capi
from .matcher import SmhipError


class MultiResolutionVoxelMapHip:
    def __init__(self, device: int = 0, table_log2: int = 22, max_cloud_points: int = 262144, max_table_log2: int | None = None, **settings):
        """table_log2: the table's FIRST size; it doubles between inserts as the map grows, up to 2^max_table_log2 (default 28)."""
        self._lib = _capi.load_library()
        self.settings = _capi.MrvmSettings()
        self._lib.smhip_mrvm_default_settings(ctypes.byref(self.settings))
        for k, v in settings.items():
            if not hasattr(self.settings, k):
                raise KeyError(f"unknown MRVM setting {k}")
            setattr(self.settings, k, v)
        self._h = ctypes.c_void_p()
        st = self._lib.smhip_mrvm_create(device, table_log2, max_cloud_points, ctypes.byref(self.settings), ctypes.byref(self._h))
        if st != 0:
            self._h = ctypes.c_void_p()
            raise SmhipError(st, self._lib.smhip_status_string(st).decode())
        self.last_warning = ""
        self.last_skipped = 0
        if max_table_log2 is not None:
            self._check(self._lib.smhip_mrvm_set_max_table_log2(self._h, max_table_log2))

    @property
    def table_log2(self) -> int:
        return int(self._lib.smhip_mrvm_table_log2(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.smhip_mrvm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise SmhipError(st, self._lib.smhip_mrvm_last_error(self._h).decode())

    def insert_point_cloud(self, points, origin):
        """points [N, 4+] float32 rows x y z intensity [factor] in the map frame; origin = the frame's sensor position."""
        p = np.ascontiguousarray(points, dtype=np.float32)
        o = np.ascontiguousarray(origin, dtype=np.float32)
        self._check(self._lib.smhip_mrvm_insert_f32(self._h, p.ctypes.data_as(_capi.c_float_p), p.shape[1], p.shape[0], o.ctypes.data_as(_capi.c_float_p)))
        # applied; what the device had to say about it (points beyond +-2^20 voxels skipped, table filling up) -- "" if nothing
        self.last_warning = self._lib.smhip_mrvm_last_error(self._h).decode()
        n = ctypes.c_int32()
        self._lib.smhip_mrvm_last_skipped(self._h, ctypes.byref(n))
        self.last_skipped = n.value

    def voxel_count(self) -> int:
        n = ctypes.c_int32()
        self._check(self._lib.smhip_mrvm_voxel_count(self._h, ctypes.byref(n)))
        return n.value

    def output_to_point_cloud(self, threshold: float | None = None, average: bool = False, rgb: bool = False) -> np.ndarray:
        """OutputToPointCloud (multi_resolution_voxel_map.cc:125-216): rows x y z intensity; average = MrvmSettings::output_average
        (one mean point per voxel); rgb = the PointXYZRGB overload, 4th column = the grey level 0..255 (the packed colour's byte)."""
        thr = self.settings.prob_threshold if threshold is None else threshold
        flags = (1 if average else 0) | (2 if rgb else 0)
        n = ctypes.c_int32()
        self._check(self._lib.smhip_mrvm_output_ex(self._h, thr, flags, None, 0, ctypes.byref(n)))
        out = np.zeros((max(n.value, 1), 4), np.float32)
        self._check(self._lib.smhip_mrvm_output_ex(self._h, thr, flags, out.ctypes.data_as(_capi.c_float_p), len(out), ctypes.byref(n)))
        out = out[:n.value]
        if rgb:                                   # packed r << 16 | g << 8 | b in the float's bits, r = g = b
            out[:, 3] = (out[:, 3].copy().view(np.uint32) & 0xff).astype(np.float32)
        return out

    def dump(self):
        """(keys [V,3], prob [V], max_intensity [V], npoints [V], points [V, max, 5]) sorted by key -- the parity tests' view."""
        V = max(self.voxel_count(), 1)
        P = self.settings.max_point_num_in_cell
        keys = np.zeros((V, 3), np.int32); prob = np.zeros(V, np.uint8); mi = np.zeros(V, np.int32); npts = np.zeros(V, np.int32)
        pts = np.zeros((V, P, 5), np.float32)
        n = ctypes.c_int32()
        self._check(self._lib.smhip_mrvm_dump(self._h, keys.ctypes.data_as(_capi.c_int32_p), prob.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                              mi.ctypes.data_as(_capi.c_int32_p), npts.ctypes.data_as(_capi.c_int32_p), pts.ctypes.data_as(_capi.c_float_p),
                                              V, ctypes.byref(n)))
        keys, prob, mi, npts, pts = keys[:n.value], prob[:n.value], mi[:n.value], npts[:n.value], pts[:n.value]
        o = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
        return keys[o], prob[o], mi[o], npts[o], pts[o]
