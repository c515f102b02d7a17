"""One rank of a sharded scan-to-scan run with the REAL matcher (tests/test_shard_driver_gpu.py launches two of these on
one GPU under torch.distributed.run): round-robin pairs -> IcpFastHip on cuda:0 -> one gather of the 18-double pose rows
(gloo here, because RCCL refuses two ranks on one device; on an 8-GPU node bench.py and smhip_shard use RCCL).
Usage: shard_worker.py SCAN_DIR OUT_NPY"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm  # noqa: E402
from staticmapping_amd import kitti, shard  # noqa: E402


def main():
    scan_dir, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    files = kitti.list_scans(scan_dir)
    scans = [(lambda f=f: kitti.read_bin(f)) for f in files]
    n_pairs = len(files) - 1
    G = np.eye(4); G[0, 3] = 0.6
    m = sm.IcpFastHip(device=0, pair_slots=3, max_source_points=32768, max_target_points=32768, max_iteration=20, early_exit=0)
    idx, T, sc, it = kitti.scan_to_scan_sequence(scans, m, batch=2, guesses=[G] * n_pairs, rank=rank, world=world)
    m.close()
    per = shard.padded_local_count(n_pairs, world)
    local = torch.zeros((per, shard.POSE_DOUBLES), dtype=torch.float64)
    for s, (Ti, si, ii) in enumerate(zip(T, sc, it)):
        local[s, :16] = torch.from_numpy(np.ascontiguousarray(Ti.T).reshape(-1))       # column-major on the wire
        local[s, 16] = float(si); local[s, 17] = float(ii)
    rows = shard.gather_poses(local, n_pairs)
    if rank == 0:
        np.save(out, rows.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
