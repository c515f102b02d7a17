"""Scan-pair sharding across the GPUs of one node.

Independent scan pairs are dealt round-robin -- pair i goes to rank i mod G -- and every
rank aligns its own pairs with no data-path communication.  The only exchange step is ONE
gather of the resulting SE(3) poses (18 doubles per pair: 16 column-major transform +
score + iterations), done with torch.distributed: backend "nccl" is RCCL over xGMI on
ROCm, "gloo" is used by the CPU tests.  At 144 B per pair the collective is pure latency.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

POSE_DOUBLES = 18


def pairs_of_rank(n_pairs: int, rank: int, world: int) -> list[int]:
    """Global pair indices owned by `rank` (round-robin: pair i -> rank i mod world)."""
    return list(range(rank, n_pairs, world))


def padded_local_count(n_pairs: int, world: int) -> int:
    return (n_pairs + world - 1) // world


def gather_poses(local: torch.Tensor, n_pairs: int, group=None) -> torch.Tensor:
    """all_gather of each rank's [padded_local_count, 18] block and re-interleave to pair order.

    `local` may live on the GPU (RCCL) or the CPU (gloo).  Returns [n_pairs, 18] on the same
    device, identical on every rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    per = padded_local_count(n_pairs, world)
    assert local.shape == (per, POSE_DOUBLES) and local.dtype == torch.float64, local.shape
    if world == 1 and not dist.is_initialized():
        return local[:n_pairs].clone()
    out = torch.empty((world * per, POSE_DOUBLES), dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    # rank r, local slot s  ->  global pair s * world + r
    out = out.view(world, per, POSE_DOUBLES).transpose(0, 1).reshape(world * per, POSE_DOUBLES)
    return out[:n_pairs].contiguous()


def chain_poses(rel: np.ndarray) -> np.ndarray:
    """pose_{i+1} = pose_i * T_i (builder/map_builder.cc:354) -> [n+1,4,4] starting at identity."""
    poses = [np.eye(4)]
    for T in rel:
        poses.append(poses[-1] @ T)
    return np.stack(poses)


def poses_to_kitti_lines(poses: np.ndarray) -> list[str]:
    """12 floats per line, row-major 3x4 -- the kitti_pose.txt format of builder/map_builder.cc:626-641."""
    return [" ".join(f"{v:.8g}" for v in P[:3, :].reshape(-1)) for P in poses]


def unpack_pose_rows(rows: torch.Tensor):
    """[n,18] -> (transforms [n,4,4] numpy, scores [n], iterations [n])."""
    a = rows.detach().cpu().numpy()
    T = a[:, :16].reshape(-1, 4, 4).transpose(0, 2, 1).copy()
    return T, a[:, 16].copy(), a[:, 17].astype(np.int64)
