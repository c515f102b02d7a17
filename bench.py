#!/usr/bin/env python
"""bench.py -- scan-pair alignments/s of the IcpFast hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank aligns `--pairs` independent
120k-point Velodyne-64 scan pairs (BASELINE config #2: point-to-plane, exactly 20 iterations, early
exit disabled) with its inputs already resident in HBM, then the SE(3) poses of all ranks are
gathered once (RCCL all_gather over xGMI; no-op at N=1).  value = pairs aligned by all ranks per
second of the slowest rank.

  python bench.py                      # N=1, finishes in about a minute
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
N_POINTS = 120_000
ICP_ITERS = 20
RHO = 0.7


def algorithmic_bytes_per_alignment(ns: int, nt: int, iters: int = ICP_ITERS, rho: float = RHO) -> float:
    """24 N_t' + I N_s (12 + 8 + 24 rho)."""
    return 24.0 * nt + iters * ns * (12.0 + 8.0 + 24.0 * rho)


def nn_bytes_per_launch(pairs: int, ns: int) -> float:
    """Dominant kernel (FindClosests, one launch = one iteration of every pair in the batch):
    12 B source xyz read + 8 B (id, d2) written per source point -- DESIGN.md §4."""
    return pairs * ns * 20.0


def build_workload(n_distinct: int, n_points: int):
    """`n_distinct` consecutive synthetic scan pairs along a straight 0.8 m/frame drive."""
    import staticmapping_amd as sm
    from staticmapping_amd import synth
    scene = synth.make_scene(0)
    poses = [synth.make_pose(t=(0.8 * k, 0.05 * k, 0.0), rpy_deg=(0.0, 0.0, 1.5 * k)) for k in range(n_distinct + 1)]
    scans = [synth.velodyne_scan(scene, P, seed=100 + k, n_points=n_points) for k, P in enumerate(poses)]
    pairs = []
    for k in range(n_distinct):
        q, n = sm.calculate_normals(scans[k][:, :3].astype(np.float64))      # caller-side target prep
        T_true = np.linalg.inv(poses[k]) @ poses[k + 1]
        guess = T_true.copy()
        guess[:3, 3] *= 0.75                                                 # constant-velocity-like prediction
        guess[:3, :3] = np.eye(3)
        pairs.append(dict(src=scans[k + 1], q=q, n=n, T=T_true, guess=guess))
    return pairs


def main():
    # The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With RCCL in the process
    # (its own streams) the two streams a handle overlaps its half-batches on can end up sharing one queue, which
    # serialises them (measured: 61.7 instead of 54.7 ms per step).  Must be set before the runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=512, help="scan pairs per GPU per step")
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic scan pairs (replicated over the slots)")
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--nn-mode", choices=["grid", "brute"], default="grid")
    ap.add_argument("--cell", type=float, default=0.25)
    ap.add_argument("--ring", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import staticmapping_amd as sm
    from staticmapping_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # rank 0 prints ONE JSON line on stdout: RCCL's version banner and its warnings (it logs to stdout by default,
    # some of them at process exit) go to stderr instead
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    under_torchrun = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or under_torchrun:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B = args.pairs
    n_total = B * world
    work = build_workload(args.distinct, args.points)
    ns = max(len(w["src"]) for w in work)
    nt = max(len(w["q"]) for w in work)
    # one non-default torch stream carries everything (kernels, export, collective hand-off), so
    # torch.cuda.synchronize / torch events and the library see the same queue
    tstream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    m = sm.IcpFastHip(device=local_rank, pair_slots=B, max_source_points=ns, max_target_points=nt, stream=stream,
                      max_iteration=ICP_ITERS, early_exit=0, dist_outlier_ratio=RHO,
                      nn_mode=1 if args.nn_mode == "grid" else 0, grid_cell=args.cell, grid_max_ring=args.ring)
    # round-robin shard: slot s of this rank is global pair s * world + rank; its cloud is distinct pair (g mod D)
    mine = shard.pairs_of_rank(n_total, rank, world)
    first_slot = {}
    guesses = []
    for s, g in enumerate(mine):
        d = g % len(work)
        if d in first_slot:
            m.copy_slot(first_slot[d], s)
        else:
            m.set_input_source(work[d]["src"], slot=s)
            m.set_input_target(work[d]["q"], work[d]["n"], slot=s)
            first_slot[d] = s
        guesses.append(work[d]["guess"])
    m.synchronize()
    poses_local = torch.zeros((B, shard.POSE_DOUBLES), dtype=torch.float64, device=dev)

    def step():
        m.enqueue_batch(B, guesses)
        m.export_results_device(B, poses_local.data_ptr())
        gathered = shard.gather_poses(poses_local, n_total)
        return gathered

    def sync_all():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()
    m.enable_profile(2)          # HIP events around the dominant NN kernel only, on the stream each launch goes to
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res, scores, stats = m.fetch_batch(B)
    nn_prof = m.get_profile()    # the timed region's launches of the dominant kernel
    m.enable_profile(False)

    # correctness of what was timed: every gathered pose is the known motion of its pair
    T_all, sc_all, it_all = shard.unpack_pose_rows(gathered)
    worst_rot = worst_t = 0.0
    for g in range(n_total):
        da, dt = sm.se3_error(T_all[g], work[g % len(work)]["T"])
        worst_rot, worst_t = max(worst_rot, da), max(worst_t, dt)
    assert int(it_all.min()) == ICP_ITERS == int(it_all.max()), "a pair did not run exactly 20 iterations"

    out = None
    if rank == 0:
        value = n_total * args.steps / elapsed
        # ---- per-kernel breakdown: HIP events around every launch (untimed extra step)
        m.enable_profile(True)
        m.enqueue_batch(B, guesses)
        m.fetch_batch(B)
        prof = m.get_profile()
        m.enable_profile(False)
        # ---- the same kernel with the GPU to itself (one stream, untimed extra step): how long a launch takes when it
        # does not share the machine with the other half-batch's kernels
        m.set_options(no_overlap=1)
        m.enable_profile(2)
        m.enqueue_batch(B, guesses)
        m.fetch_batch(B)
        alone = m.get_profile()
        m.enable_profile(False)
        m.set_options(no_overlap=0)
        alone_ms = alone["ms_nn_main"] / max(1, alone["launches_nn_main"])
        alone_pairs = int(round(alone["pairs_nn_main"] / max(1, alone["launches_nn_main"])))
        alone_gbs = nn_bytes_per_launch(alone_pairs, ns) / (alone_ms * 1e-3) / 1e9
        # ---- roofline of the dominant kernel from the events of the TIMED region
        nn_ms = nn_prof["ms_nn_main"] / max(1, nn_prof["launches_nn_main"])
        # with >= 16 pairs a step is two half-batches on two streams: one launch covers B / 2 pairs
        pairs_per_launch = int(round(nn_prof["pairs_nn_main"] / max(1, nn_prof["launches_nn_main"])))
        nn_bytes = nn_bytes_per_launch(pairs_per_launch, ns)
        achieved = nn_bytes / (nn_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic_nn_main.json")
        if os.path.exists(tj):
            try:
                tdat = json.load(open(tj))
                if tdat.get("nn_mode") == args.nn_mode and tdat.get("source_points") in (None, ns):
                    # counters are per launch of `pairs_per_launch` pairs; traffic is per point, so it scales with the pairs
                    traffic = int(tdat["hbm_bytes_per_launch"] * pairs_per_launch / tdat["pairs_per_launch"])
            except Exception:
                traffic = None
        alg_bytes = algorithmic_bytes_per_alignment(ns, nt)
        out = {
            "metric": "scan-pair alignments/sec (120k-pt KITTI-64, 20 ICP iters)",
            "value": round(value, 2), "unit": "alignments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE config #2: IcpFast point-to-plane, 120k-pt synthetic Velodyne-64 scan pair "
                                   f"vs CalculateNormals target ({nt} pts), exactly 20 iterations, rho 0.7",
                       "pairs_per_gpu": B, "global_pairs_per_step": n_total, "source_points": ns,
                       "target_points": nt, "iterations": ICP_ITERS, "nn_mode": args.nn_mode,
                       "grid_cell_m": args.cell, "parallelism": f"pairs round-robin over {world} GPU(s), one RCCL gather of poses"},
            "roofline": {"bound": "hbm", "kernel": "nn_ball_lds" if args.nn_mode == "grid" else "nn_brute",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "bytes_per_launch": nn_bytes, "pairs_per_launch": pairs_per_launch, "avg_launch_ms": round(nn_ms, 4),
                         "launches_timed": nn_prof["launches_nn_main"],
                         "alone": {"note": "same kernel on one stream, not sharing the GPU with the other half-batch",
                                   "pairs_per_launch": alone_pairs, "avg_launch_ms": round(alone_ms, 4),
                                   "achieved": round(alone_gbs, 2), "frac": round(alone_gbs / HBM_PEAK_GBS, 5)},
                         "whole_alignment": {"algorithmic_bytes": alg_bytes,
                                             "achieved_GBs": round(alg_bytes * value / world / 1e9, 2),
                                             "frac": round(alg_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5)}},
            "kernel_ms_per_step": {k: round(v, 3) for k, v in prof.items() if k.startswith("ms_")},
            "parity": {"worst_rot_err_vs_truth_rad": worst_rot, "worst_trans_err_vs_truth_m": worst_t},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(work[0], args.cpu_seconds)
    m.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def cpu_baseline(w, budget_s: float):
    """The oracle's C restatement of IcpFast::Align timed on the host (reference-faithful threading:
    icp_fast.cc has no pragma, so 1 thread), on a bounded sample of the SAME workload."""
    from oracle import cref
    src = w["src"][:, :3].astype(np.float64)
    n_done, t_used = 0, 0.0
    while t_used < budget_s and n_done < 64:
        t = time.perf_counter()
        cref.icp_fast_align(src, w["q"], w["n"], guess=w["guess"], max_iteration=ICP_ITERS,
                            dist_outlier_ratio=RHO, early_exit=False, nthreads=1)
        t_used += time.perf_counter() - t
        n_done += 1
    ncores = os.cpu_count() or 1
    t = time.perf_counter()
    cref.icp_fast_align(src, w["q"], w["n"], guess=w["guess"], max_iteration=ICP_ITERS, dist_outlier_ratio=RHO,
                        early_exit=False, nthreads=ncores)
    t_all = time.perf_counter() - t
    return {"value": round(n_done / t_used, 3), "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": f"{n_done} alignments of one 120k-pt pair (20 iterations each), C restatement oracle/csrc/smref_icp.c, "
                      f"exact kd-tree 1-NN, gcc -O2",
            "all_cores": {"value": round(1.0 / t_all, 3), "cores": ncores,
                          "note": "same code with the FindClosests loop under OpenMP"}}


if __name__ == "__main__":
    main()
