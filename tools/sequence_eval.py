"""Compare a kitti_pose.txt trajectory with the generating poses of a synthetic drive (tools/make_drive.py).
Usage: sequence_eval.py POSES TRUTH -> one JSON line (relative pose errors per pair, end-point drift)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from staticmapping_amd import kitti
import staticmapping_amd as sm  # noqa: F401  (se3_error)
from staticmapping_amd.matcher import se3_error

got, truth = kitti.read_poses(sys.argv[1]), kitti.read_poses(sys.argv[2])
n = min(len(got), len(truth))
rot, trans = [], []
for k in range(n - 1):
    Tg = np.linalg.inv(got[k]) @ got[k + 1]
    Tt = np.linalg.inv(truth[k]) @ truth[k + 1]
    a, t = se3_error(Tg, Tt)
    rot.append(a); trans.append(t)
rot, trans = np.array(rot), np.array(trans)
path = float(np.sum(np.linalg.norm(np.diff(truth[:n, :3, 3], axis=0), axis=1)))
print(json.dumps({"poses": n, "path_m": round(path, 1),
                  "rel_trans_err_m": {"median": float(np.median(trans)), "p95": float(np.percentile(trans, 95)), "max": float(trans.max())},
                  "rel_rot_err_rad": {"median": float(np.median(rot)), "p95": float(np.percentile(rot, 95)), "max": float(rot.max())},
                  "pairs_within_2cm": float((trans < 0.02).mean()),
                  "end_point_drift_m": float(np.linalg.norm(got[n - 1][:3, 3] - truth[n - 1][:3, 3])),
                  "end_point_drift_pct_of_path": float(100 * np.linalg.norm(got[n - 1][:3, 3] - truth[n - 1][:3, 3]) / max(path, 1e-9))}))
