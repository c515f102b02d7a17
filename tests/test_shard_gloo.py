"""The N>1 path on CPU: world_size-2 gloo run of the pose gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from staticmapping_amd import shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.pairs_of_rank(n_pairs, rank, world)
    per = shard.padded_local_count(n_pairs, world)
    local = torch.zeros((per, shard.POSE_DOUBLES), dtype=torch.float64)
    for s, g in enumerate(mine):
        local[s] = torch.arange(18, dtype=torch.float64) + 100.0 * g     # fake "pose" of global pair g
    out = shard.gather_poses(local, n_pairs)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_partition():
    assert shard.pairs_of_rank(10, 0, 4) == [0, 4, 8]
    assert shard.pairs_of_rank(10, 3, 4) == [3, 7]
    allp = sorted(sum((shard.pairs_of_rank(4541, r, 8) for r in range(8)), []))
    assert allp == list(range(4541))
    assert shard.padded_local_count(4541, 8) == 568


def test_gather_poses_world2_gloo():
    world, n_pairs = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([np.arange(18) + 100.0 * g for g in range(n_pairs)])
    for r in range(world):
        assert np.array_equal(res[r], expect)


def test_chain_and_kitti_format():
    T = np.eye(4); T[0, 3] = 0.8
    poses = shard.chain_poses(np.stack([T, T, T]))
    assert poses.shape == (4, 4, 4) and abs(poses[3][0, 3] - 2.4) < 1e-12
    lines = shard.poses_to_kitti_lines(poses)
    assert len(lines) == 4 and len(lines[1].split()) == 12


def test_unpack_rows():
    rows = torch.zeros((2, 18), dtype=torch.float64)
    M = np.arange(16.0).reshape(4, 4)
    rows[1, :16] = torch.from_numpy(M.T.reshape(-1).copy())     # column-major on the wire
    rows[1, 16] = 0.9; rows[1, 17] = 20
    T, s, it = shard.unpack_pose_rows(rows)
    assert np.array_equal(T[1], M) and s[1] == 0.9 and it[1] == 20
