"""TEST ORACLE (not product code): static_map::MultiResolutionVoxelMap restated a second time, in plain Python, to cross-check
oracle/csrc/smref_mrvm.c -- the reference holds no test or fixture for this class (PARITY UNPINNED), so the two
restatements were written separately from the reference text and are compared with each other and with hand-computed
cases (tests/test_oracle_mrvm.py).  Pure-Python loops: small clouds only.

  /root/reference/builder/multi_resolution_voxel_map.h:49-50, 119-139    Probability = uint8, kTableSize 256, kUnknown 128,
                                                                         ProbabilityToOdd / OddToProbability
  /root/reference/builder/multi_resolution_voxel_map.cc:36-53            odds table, Initialise (clamps)
  /root/reference/builder/multi_resolution_voxel_map.cc:59-131           InsertPointCloud, in point order (the loop without OpenMP)
  /root/reference/builder/multi_resolution_voxel_map.cc:133-170          OutputToPointCloud (PointXYZI)
  /root/reference/common/math.cc:35-93                                   VoxelCastingBresenham
"""
import math

import numpy as np

F = np.float32
TABLE = 256
UNKNOWN = 128


def _clamp(v, lo, hi):                      # common::Clamp (common/math.h:66-75)
    return hi if v > hi else (lo if v < lo else v)


def prob_to_odd(p):                          # header :134-136: float in, double arithmetic (the literal 1.), float out
    p = float(F(p))
    with np.errstate(divide="ignore"):
        return F(np.log(np.float64(p) / (1.0 - np.float64(p))))


def odd_to_prob(odd):                        # header :138-140: std::exp(float) is the float overload
    e = F(np.exp(F(odd)))
    return F(1.0 - 1.0 / (1.0 + np.float64(e)))


def bresenham(start, end, step):             # common/math.cc:35-93
    step = F(step)
    c0 = [int(math.floor(float(F(F(v) / step)))) for v in start]
    c1 = [int(math.floor(float(F(F(v) / step)))) for v in end]
    d = [abs(c1[k] - c0[k]) for k in range(3)]
    s = [1 if c0[k] < c1[k] else -1 for k in range(3)]
    dm = max(d)
    err = [dm >> 1] * 3
    cur = list(c0)
    out = []
    i = dm
    while True:
        out.append(tuple(cur))
        if i == 0:
            break
        i -= 1
        for k in range(3):
            err[k] -= d[k]
            if err[k] < 0:
                err[k] += dm
                cur[k] += s[k]
    assert cur == c1
    return out


class Mrvm:
    def __init__(self, high_resolution=0.1, hit_prob=0.55, miss_prob=0.48, z_offset=0.0, max_point_num_in_cell=10):
        assert max_point_num_in_cell > 0                                        # :48 CHECK_GT
        self.res = F(high_resolution)
        self.hit = F(_clamp(F(hit_prob), F(0.501), F(0.9)))                      # :50
        self.miss = F(_clamp(F(miss_prob), F(0.1), F(0.499)))                    # :51
        self.z_offset = F(z_offset)
        self.max_points = max_point_num_in_cell
        self.odds = [prob_to_odd(F(i) / F(TABLE)) for i in range(TABLE)]        # :37-39
        self.vox = {}            # key -> [probability, need_update, max_intensity, points]

    def _update(self, former, hit):                                             # :70-74
        odd = F(self.odds[former] + (prob_to_odd(self.hit) if hit else prob_to_odd(self.miss)))
        return F(_clamp(odd_to_prob(odd), F(0.1), F(0.9)))

    def _voxel(self, key):
        return self.vox.setdefault(key, [UNKNOWN, True, 0, []])                 # header :120-121

    def insert(self, points5, origin):
        pts = np.asarray(points5, dtype=np.float32)
        if len(pts) == 0:
            return
        o = [F(origin[0]), F(origin[1]), F(F(origin[2]) + self.z_offset)]       # :66-67
        ends = {}
        for p in pts:
            ray = bresenham(o, (p[0], p[1], p[2]), self.res)
            if not ray:
                continue
            end = ray[-1]
            v = self._voxel(end)
            v[1] = False                                                        # :91
            ends[end] = True
            if int(p[3]) > int(v[2]):                                           # :98-101 (truncation toward zero)
                v[2] = int(p[3])
            v[0] = int(F(self._update(v[0], True) * F(TABLE))) & 0xff           # :102
            if len(v[3]) < self.max_points:                                     # :103-106
                v[3].append(np.array(p, dtype=np.float32))
            for key in ray[:-1]:                                                # :109-119
                w = self.vox.get(key)
                if w is not None and w[1]:
                    w[0] = int(F(self._update(w[0], False) * F(TABLE))) & 0xff
        for key in ends:                                                        # :123-125
            self.vox[key][1] = True

    def dump(self):
        keys = sorted(self.vox)
        return (np.array(keys, np.int32).reshape(-1, 3), np.array([self.vox[k][0] for k in keys], np.uint8),
                np.array([self.vox[k][2] for k in keys], np.int32), np.array([len(self.vox[k][3]) for k in keys], np.int32))

    def output(self, threshold=0.6, use_max_intensity=True, average=False, rgb=False):
        """OutputToPointCloud, both overloads (:125-170 PointXYZI, :172-216 PointXYZRGB), with settings_.output_average.  Rows
        x y z c: c = intensity, or for rgb the grey level r = g = b."""
        thr = int(F(F(threshold) * F(TABLE))) & 0xff                            # :132 (float -> uint8)
        rows = []
        for key in sorted(self.vox):
            prob, _, mi, pts = self.vox[key]
            if prob < thr:
                continue
            grey = min(255, int(float((int(mi) & 0xffffffff)) * 1.4))           # :181-186: uint32_t intensity *= 1.4 (double), clamp
            if average:                                                         # :136-151 / :188-199
                ax = ay = az = F(0)                                             # pcl points start at 0
                for p in pts:
                    ax = F(ax + F(p[0])); ay = F(ay + F(p[1])); az = F(az + F(p[2]))
                size = F(len(pts))
                c = F(grey) if rgb else (F(int(mi)) if use_max_intensity else F(0))   # an averaged PointXYZI's intensity is only set with use_max_intensity
                rows.append([F(ax / size), F(ay / size), F(az / size), c])
            else:
                for p in pts:
                    rows.append([p[0], p[1], p[2], F(grey) if rgb else (F(int(mi)) if use_max_intensity else p[3])])
        return np.array(rows, np.float32).reshape(-1, 4)
