"""libnabo 1.0.7 `KDTREE_LINEAR_HEAP` restated in plain Python (TEST ORACLE; small cases only).

The reference asks libnabo for an eps = 3.16 APPROXIMATE nearest neighbour
(/root/reference/registrators/icp_fast.cc:174-178, tree built at :464-467; via libpointmatcher at
registrators/icp_pointmatcher.cc:186-191).  libnabo is a git dependency pinned to tags/1.0.7
(/root/reference/setup/install_libnabo.sh:16-18) and is NOT in /root/reference, so its algorithm is restated
here from the published source (nabo/kdtree_cpu.cpp, class KDTreeUnbalancedPtInLeavesImplicitBoundsStackOpt):

  buildNodes   leaf when count <= bucketSize (8); cut dimension = argMax(maxValues - minValues) of the box INHERITED
               from the parent (root: the cloud's bounds; a child gets the parent's box cut at cutVal), argMax starting
               from (index 0, value 0); leftCount = count - count / 2; std::nth_element at first + leftCount;
               cutVal = coordinate of that element; left child is stored right after its parent.
  recurseKnn   leaf: every bucket entry with dist < heap head replaces it (k = 1: strict "<", first seen wins);
               inner node: new_off = q[cd] - cutVal; descend the side of the query first (right when new_off > 0),
               then rd += new_off^2 - old_off^2 and the other side is visited only if rd * (1 + eps)^2 < head.
This module is the slow, independent cross-check of the C version in oracle/csrc/smref_icp.c (nabo_*).
PARITY UNPINNED like the rest of the registrator oracle: no libnabo binary exists here to run against.
"""
from __future__ import annotations

import numpy as np

BUCKET_SIZE = 8


def _arg_max(v):
    max_val, max_idx = 0.0, 0
    for i in range(len(v)):
        if v[i] > max_val:
            max_val, max_idx = v[i], i
    return max_idx


class NaboTree:
    def __init__(self, pts: np.ndarray):
        self.pts = np.asarray(pts, dtype=np.float64)
        self.nodes = []          # (dim, cut, right_child) or (3, first, count)
        self.perm = np.arange(len(self.pts))
        if len(self.pts):
            self._build(0, len(self.pts), self.pts.min(axis=0).copy(), self.pts.max(axis=0).copy())

    def _build(self, first, last, mn, mx):
        count = last - first
        pos = len(self.nodes)
        if count <= BUCKET_SIZE:
            self.nodes.append((3, first, count))
            return pos
        cd = _arg_max(mx - mn)
        left = count - count // 2
        seg = self.perm[first:last]
        order = np.argpartition(self.pts[seg, cd], left)
        self.perm[first:last] = seg[order]
        cut = self.pts[self.perm[first + left], cd]
        lmx = mx.copy(); lmx[cd] = cut
        rmn = mn.copy(); rmn[cd] = cut
        self.nodes.append(None)
        self._build(first, first + left, mn, lmx)
        right = self._build(first + left, last, rmn, mx)
        self.nodes[pos] = (cd, cut, right)
        return pos

    def knn1(self, q, eps: float = 0.0):
        """(index, squared distance, leaves visited) of libnabo's knn(k = 1, epsilon = eps, ALLOW_SELF_MATCH)."""
        best = [-1, np.inf, 0]
        off = [0.0, 0.0, 0.0]
        max_error2 = (1.0 + eps) * (1.0 + eps)

        def rec(n, rd):
            node = self.nodes[n]
            if node[0] == 3:
                best[2] += 1
                for e in self.perm[node[1]:node[1] + node[2]]:
                    dist = 0.0
                    for d in range(3):
                        diff = q[d] - self.pts[e, d]
                        dist += diff * diff
                    if dist < best[1]:
                        best[0], best[1] = int(e), dist
                return
            cd, cut, right = node
            old_off, new_off = off[cd], q[cd] - cut
            near, far = (right, n + 1) if new_off > 0 else (n + 1, right)
            rec(near, rd)
            rd += -old_off * old_off + new_off * new_off
            if rd * max_error2 < best[1]:
                off[cd] = new_off
                rec(far, rd)
                off[cd] = old_off

        if self.nodes:
            rec(0, 0.0)
        return best[0], best[1], best[2]
