#!/usr/bin/env python
"""bench.py -- scan-pair alignments/s of the IcpFast hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank aligns `--pairs` independent 120k-point Velodyne-64
scan pairs (BASELINE config #2: point-to-plane, exactly 20 iterations, early exit disabled) with its inputs already
resident in HBM, then the SE(3) poses of all ranks are gathered once (RCCL all_gather over xGMI; no-op at N=1).
value = pairs aligned by all ranks per second of the slowest rank.

Workload: `--distinct` (64) DIFFERENT consecutive scan pairs of a synthetic drive (SURVEY.md §8(d) cfg 4: 10 Hz, speed and
yaw rate varying, seed 5), each uploaded to its own slots -- no device-side replication of one pair.  The line carries three
figures for the same clouds:
  value / figures.extrapolated_guess guess = the previous pair's motion (what the front end's extrapolator supplies,
                                     builder/map_builder.cc:302-308), exactly 20 iterations              <- headline
  figures.identity_guess             guess = identity, exactly 20 iterations (SURVEY cfg 2 / cfg 4 literal).  With 0.6-1.0 m
                                     between the scans, point-to-plane ICP on a road scene does not recover the along-track
                                     motion from identity -- neither here nor in the CPU oracle (both end ~0.86 m from the
                                     truth, and agree with each other): the reference's front end never calls Align that way
  figures.early_exit                 extrapolated guess, CheckConvergence on, max 100 iterations (the reference default)
  figures.reference_search_eps3.16   extrapolated guess, 20 iterations, nn_mode NABO: libnabo's tree and its epsilon = 3.16 approximate
                                     search on the device -- the reference's own semantics, compared with the oracle run the same way
and the other matchers of the path, measured on one GPU (rank 0, N = 1 only), under `other_workloads`:
  ndt        BASELINE config #3: registrators::Ndt, 120k scan vs 500k-pt submap, 1.0 m voxels
  ndt_gicp   BASELINE config #5: registrators::NdtWithGicp, 120k scan vs 2M-pt submap
Every timed pose is compared with the CPU oracle run on the same clouds and guess (`parity.*_vs_oracle`).

  python bench.py                      # N=1, a few minutes (most of it the CPU oracle legs)
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


def algorithmic_bytes_per_alignment(ns: int, nt: int, iters: float = ICP_ITERS, rho: float = RHO) -> float:
    """24 N_t' + I N_s (12 + 8 + 24 rho)."""
    return 24.0 * nt + iters * ns * (12.0 + 8.0 + 24.0 * rho)


def _device_id(torch, index: int) -> str:
    p = torch.cuda.get_device_properties(index)
    return f"{index}:{getattr(p, 'uuid', '')}:{getattr(p, 'pci_bus_id', '')}:{getattr(p, 'gcnArchName', p.name)}"


def build_workload(n_distinct: int, n_points: int, device):
    """`n_distinct` consecutive scan pairs (i, i + 1) of the synthetic drive; scan i prepared as pair i's target by
    the caller-side CalculateNormals (builder/map_builder.cc:286,389)."""
    from staticmapping_amd import synth
    return synth.drive_pairs(n_distinct, n_points, device, seed=5)


def oracle_pose(w, guess, max_iteration, early_exit, nthreads=1, nn_eps=None):
    from oracle import cref
    return cref.icp_fast_align(w["src"][:, :3].astype(np.float64), w["q"], w["n"], guess=guess, max_iteration=max_iteration,
                               dist_outlier_ratio=RHO, early_exit=early_exit, nthreads=nthreads, nn_eps=nn_eps)


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
    ap.add_argument("--distinct", type=int, default=512, help="distinct synthetic scan pairs (cycled over the slots; each slot uploaded).  Default: every "
                    "slot of the default batch holds a pair of its own (513 scans of the synthetic drive, ~25 s to generate and prepare)")
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--nn-mode", choices=["grid", "brute"], default="grid")
    ap.add_argument("--cell", type=float, default=0.25)
    ap.add_argument("--ring", type=int, default=8)
    ap.add_argument("--split-after", type=int, default=0, help="smhip_icp_options.split_after (0 = the library's default)")
    ap.add_argument("--ball-radius", type=float, default=0.0, help="smhip_icp_options.ball_radius: first-iteration search radius (0 = the library's default, 0.3 m)")
    ap.add_argument("--streams", type=int, default=0, help="smhip_icp_options.overlap_streams: parts a batch is split into (0 = the library's default, 2)")
    ap.add_argument("--headline", choices=["identity", "extrapolated"], default="extrapolated")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="oracle / cpu_baseline sample: distinct pairs run on the host (0 = the first 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (no parity-vs-oracle either)")
    ap.add_argument("--no-figures", action="store_true", help="skip the extra figures")
    ap.add_argument("--no-other", action="store_true", help="skip other_workloads (NDT / NdtWithGicp), single_pair and end_to_end")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end_to_end leg (generated drive through the C++ sequence driver)")
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
    # SMHIP_BENCH_SHARED_GPU=1: a dry run of the multi-rank control flow on a box with fewer GPUs than ranks (ranks share
    # devices, gloo instead of RCCL, the collectives' tensors on the host): the same sequence of collectives, no valid timing
    shared_gpu = os.environ.get("SMHIP_BENCH_SHARED_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = torch.device("cpu") if shared_gpu else dev
    # rank 0 prints ONE JSON line on stdout: RCCL's version banner and its warnings (it logs to stdout by default,
    # some of them at process exit) go to stderr instead
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    under_torchrun = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or under_torchrun:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # What the collective library actually saw: an all-reduce of ones over the communicator the poses are gathered on (its result IS
    # the number of ranks that took part) and every rank's device, so that a multi-GPU line certifies itself.
    coll = {"backend": None, "ranks_in_all_reduce": 1, "devices": [_device_id(torch, dev_index)]}
    if dist.is_initialized():
        ones = torch.ones(1, dtype=torch.int32, device=coll_dev)
        dist.all_reduce(ones)
        got = [None] * world
        dist.all_gather_object(got, (rank, _device_id(torch, dev_index)))
        coll = {"backend": "gloo (shared-GPU dry run)" if shared_gpu else "nccl (RCCL)", "ranks_in_all_reduce": int(ones.item()),
                "devices": [d for _, d in sorted(got)]}
        assert coll["ranks_in_all_reduce"] == world, coll
        if not shared_gpu:
            assert len(set(coll["devices"])) == world, f"two ranks on one device: {coll['devices']}"

    B = args.pairs
    n_total = B * world
    t_gen = time.perf_counter()
    work = build_workload(args.distinct, args.points, dev)
    t_gen = time.perf_counter() - t_gen
    D = len(work)
    ns = max(len(w["src"]) for w in work)
    nt = max(len(w["q"]) for w in work)
    nt_mean = float(np.mean([len(w["q"]) for w in work]))
    # one non-default torch stream carries everything (kernels, export, collective hand-off), so
    # torch.cuda.synchronize / torch events and the library see the same queue
    tstream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    m = sm.IcpFastHip(device=dev_index, pair_slots=B, max_source_points=ns, max_target_points=nt, stream=stream,
                      max_iteration=ICP_ITERS, early_exit=0, dist_outlier_ratio=RHO,
                      nn_mode=1 if args.nn_mode == "grid" else 0, grid_cell=args.cell, grid_max_ring=args.ring,
                      split_after=args.split_after, ball_radius=args.ball_radius, overlap_streams=args.streams)
    # round-robin shard: slot s of this rank is global pair s * world + rank; its clouds are distinct pair (g mod D),
    # uploaded into the slot (every slot has its own copy in HBM: 512 slots x (120k + 21.7k) points)
    mine = shard.pairs_of_rank(n_total, rank, world)
    for s, g in enumerate(mine):
        w = work[g % D]
        m.set_input_source(w["src"], slot=s)
        m.set_input_target(w["q"], w["n"], slot=s)
    m.synchronize()
    poses_local = torch.zeros((B, shard.POSE_DOUBLES), dtype=torch.float64, device=dev)

    def sync_all():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_run(guess_key, steps, warmup, profile_nn=0):
        if guess_key == "mixed":     # one batch holding both kinds: even pairs the extrapolated guess, odd pairs the identity
            guesses = [work[g % D]["guess_cv" if g % 2 == 0 else "guess_id"] for g in mine]
        else:
            guesses = [work[g % D][guess_key] for g in mine]

        def step():
            m.enqueue_batch(B, guesses)
            m.export_results_device(B, poses_local.data_ptr())
            if shared_gpu:
                return shard.gather_poses(poses_local.cpu(), n_total)
            return shard.gather_poses(poses_local, n_total)
        for _ in range(warmup):
            step()
        sync_all()
        if profile_nn:
            m.enable_profile(profile_nn)      # HIP events around ONE kernel class, on the stream each launch goes to
        t0 = time.perf_counter()
        for _ in range(steps):
            gathered = step()
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        res, scores, stats = m.fetch_batch(B)
        prof = m.get_profile()
        split = prof["split_after_used"]     # where the batch switched from the fused search to certify + listed search
        if profile_nn:
            m.enable_profile(False)
        else:
            prof = None
        T_all, sc_all, it_all = shard.unpack_pose_rows(gathered)
        return dict(elapsed=elapsed, value=n_total * steps / elapsed, T=T_all, it=it_all, stats=stats, prof=prof, guesses=guesses, split=split)

    def truth_errors(T_all):
        worst_rot = worst_t = 0.0
        errs = []
        for g in range(n_total):
            da, dt = sm.se3_error(T_all[g], work[g % D]["T"])
            worst_rot, worst_t = max(worst_rot, da), max(worst_t, dt)
            errs.append(dt)
        return worst_rot, worst_t, float(np.median(errs))

    def searched(stats):
        return float(np.mean([s["searched_queries"] for s in stats]))

    head_key = "guess_id" if args.headline == "identity" else "guess_cv"
    other_key = "guess_cv" if args.headline == "identity" else "guess_id"
    # Which kernel class does this workload spend the most time in?  One untimed step with events around every launch (after a
    # plain one, so that the library has placed the search-form switch from a batch of these guesses); the timed region then
    # carries events around that class only -- every bracket is a barrier between two launches, and bracketing all classes
    # took 6 % off the batch rate.
    guesses0 = [work[g % D][head_key] for g in mine]
    m.enqueue_batch(B, guesses0); m.fetch_batch(B)
    m.enable_profile(True)
    m.enqueue_batch(B, guesses0); m.fetch_batch(B)
    pre = m.get_profile()
    m.enable_profile(False)
    # the class of the single kernel with the largest summed time
    single = {"nn_main": pre["ms_nn_main"], "nn_certify": pre["ms_nn_certify"], "accumulate": pre["ms_error_elements"], "nn_listed": pre["ms_nn_listed"]}
    dom_single = max(single, key=lambda k: single[k])
    prof_class = {"nn_main": 2, "nn_certify": 2, "accumulate": 3, "nn_listed": 4}[dom_single]
    if world > 1:      # every rank must time the same thing
        t = torch.tensor([prof_class], dtype=torch.int32, device=coll_dev)
        dist.broadcast(t, 0)
        prof_class = int(t.item())
    head = timed_run(head_key, args.steps, args.warmup, profile_nn=prof_class)
    by_rank = [(rank, prof_class, int(head["split"]))]
    if world > 1:       # what every rank timed: the same kernel class (broadcast above) and where its library placed the search-form switch
        got = [None] * world
        dist.all_gather_object(got, by_rank[0])
        by_rank = sorted(got)
        assert len({c for _, c, _ in by_rank}) == 1, by_rank
    assert int(head["it"].min()) == ICP_ITERS == int(head["it"].max()), "a pair did not run exactly 20 iterations"
    elapsed, value, nn_prof = head["elapsed"], head["value"], head["prof"]
    h_rot, h_t, h_med = truth_errors(head["T"])

    out = None
    figures = {}
    if rank == 0 or world > 1:
        # every rank takes part in the extra figures' gathers; only rank 0 reports
        if not args.no_figures:
            fs, fw = max(2, args.steps // 2), 1
            oth = timed_run(other_key, fs, fw)
            o_rot, o_t, o_med = truth_errors(oth["T"])
            m.set_options(max_iteration=100, early_exit=1)
            ee = timed_run("guess_cv", fs, fw)
            m.set_options(max_iteration=ICP_ITERS, early_exit=0)
            e_rot, e_t, e_med = truth_errors(ee["T"])
            name_h = "identity_guess" if args.headline == "identity" else "extrapolated_guess"
            name_o = "extrapolated_guess" if args.headline == "identity" else "identity_guess"
            figures[name_h] = dict(value=round(head["value"], 2), iterations=ICP_ITERS, searched_queries_per_alignment=searched(head["stats"]), split_after=head["split"],
                                   worst_trans_err_vs_truth_m=h_t, median_trans_err_vs_truth_m=h_med, T=head["T"], guess_key=head_key,
                                   max_iteration=ICP_ITERS, early_exit=False)
            figures[name_o] = dict(value=round(oth["value"], 2), iterations=ICP_ITERS, searched_queries_per_alignment=searched(oth["stats"]), split_after=oth["split"],
                                   worst_trans_err_vs_truth_m=o_t, median_trans_err_vs_truth_m=o_med, T=oth["T"], guess_key=other_key,
                                   max_iteration=ICP_ITERS, early_exit=False)
            # a heterogeneous batch: alternating extrapolated / identity guesses in ONE 512-pair batch.  The library
            # places the switch from the fused search to certificate pass + listed search once per batch (from the previous
            # batch's per-iteration search counts); here half the pairs would want it at 2 and half at 5.
            mx = timed_run("mixed", fs, fw)
            mx_err = [sm.se3_error(mx["T"][g], work[g % D]["T"])[1] for g in range(n_total)]
            hm = 2.0 / (1.0 / head["value"] + 1.0 / oth["value"])
            figures["mixed_guess"] = dict(value=round(mx["value"], 2), iterations=ICP_ITERS, searched_queries_per_alignment=searched(mx["stats"]), split_after=mx["split"],
                                          harmonic_mean_of_the_two_homogeneous_figures=round(hm, 2), ratio_to_harmonic_mean=round(mx["value"] / hm, 4),
                                          median_trans_err_vs_truth_m_even_pairs=float(np.median(mx_err[0::2])),
                                          median_trans_err_vs_truth_m_odd_pairs=float(np.median(mx_err[1::2])),
                                          fused_iterations_mean_even_odd=[float(np.mean([s_["fused_iterations"] for s_ in mx["stats"][0::2]])),
                                                                          float(np.mean([s_["fused_iterations"] for s_ in mx["stats"][1::2]]))],
                                          T=mx["T"], guess_key="mixed", max_iteration=ICP_ITERS, early_exit=False)
            # the reference's own search semantics: libnabo's tree + epsilon = 3.16 approximate knn on the device (nn_mode NABO)
            m.set_options(nn_mode=sm.NN_NABO, nn_epsilon=3.16)
            nb = timed_run("guess_cv", fs, fw)
            if rank == 0:                        # where this mode's step goes: events around every launch of one untimed step
                m.enable_profile(True)
                m.enqueue_batch(B, nb["guesses"]); m.fetch_batch(B)
                nb["prof_all"] = m.get_profile()
                m.enable_profile(False)
            m.set_options(nn_mode=1 if args.nn_mode == "grid" else 0)
            n_rot, n_t, n_med = truth_errors(nb["T"])
            figures["reference_search_eps3.16"] = dict(value=round(nb["value"], 2), iterations=ICP_ITERS, searched_queries_per_alignment=searched(nb["stats"]),
                                                       worst_trans_err_vs_truth_m=n_t, median_trans_err_vs_truth_m=n_med, T=nb["T"], guess_key="guess_cv",
                                                       max_iteration=ICP_ITERS, early_exit=False, nn_eps=3.16, prof_all=nb.get("prof_all"))
            figures["early_exit"] = dict(value=round(ee["value"], 2), iterations=float(ee["it"].mean()), iterations_max=int(ee["it"].max()),
                                         searched_queries_per_alignment=searched(ee["stats"]), split_after=ee["split"], worst_trans_err_vs_truth_m=e_t,
                                         median_trans_err_vs_truth_m=e_med, T=ee["T"], guess_key="guess_cv", max_iteration=100, early_exit=True)
    if rank == 0:
        guesses = head["guesses"]
        # ---- per-kernel breakdown: HIP events around every launch (untimed extra step; one plain step first, so that the
        # library places the search-form switch from a batch of these guesses, as in the timed region, not from whatever
        # figure ran last)
        m.enqueue_batch(B, guesses)
        m.fetch_batch(B)
        m.enable_profile(True)
        m.enqueue_batch(B, guesses)
        m.fetch_batch(B)
        prof = m.get_profile()
        m.enable_profile(False)
        # ---- the same kernel with the GPU to itself (one stream, untimed extra step): how long a launch takes when it
        # does not share the machine with the other half-batch's kernels
        m.set_options(no_overlap=1)
        m.enable_profile(prof_class)
        m.enqueue_batch(B, guesses)
        m.fetch_batch(B)
        alone = m.get_profile()
        m.enable_profile(False)
        m.set_options(no_overlap=0)
        # ---- roofline from the events of the TIMED region (HIP events on the stream each launch goes to).  An iteration's
        # FindClosests runs either as the fused search nn_ball_lds (+ the three refinement launches validate / ring / fallback)
        # or as the certificate pass nn_certify + the listed search nn_ball_listed (+ the same refinement launches); then
        # accumulate (ErrorElements + ComputePointToPlane) and finalize.  Every kernel is priced with the ALGORITHMIC bytes of
        # the reference function it implements: FindClosests 12 B read + 8 B written = 20 B per source point,
        # ErrorElements/ComputePointToPlane 24 B gathered per kept point = 24 rho B per source point -- and the line's kernel is
        # the one the timed region spent the most time in, whichever it is.
        def kernel_figures(p, ms, launches, pairs_sum, bytes_per_point):
            n_l = max(1, launches)
            avg_ms = ms / n_l
            pairs = int(round(pairs_sum / n_l))
            if bytes_per_point is None:          # a kernel that handles a varying PART of the points: no per-launch byte figure, no fraction
                return dict(total_ms=ms, launches=launches, avg_launch_ms=avg_ms, pairs_per_launch=pairs, bytes_per_point=None,
                            bytes_per_launch=None, achieved=None)
            by = pairs * ns * bytes_per_point
            return dict(total_ms=ms, launches=launches, avg_launch_ms=avg_ms, pairs_per_launch=pairs, bytes_per_point=bytes_per_point,
                        bytes_per_launch=by, achieved=by / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0)

        # From the switch on a batch runs the certificate pass and the normal-equation sums as ONE kernel (nn_certify_acc: FindClosests for
        # the certified queries + ErrorElements / ComputePointToPlane below the predicted quantile band, 20 + 24 rho B per source point);
        # `accumulate` is then launched every iteration but returns at once unless a pair's prediction missed.
        fused_on = float(np.mean([s_["fused_iterations"] for s_ in head["stats"]])) > 0
        cert_name = "nn_certify_acc" if fused_on else "nn_certify"
        cert_bytes = 20.0 + 24.0 * RHO if fused_on else 20.0

        def all_kernels(p):
            kf = {}
            main_name = "nn_ball_lds" if args.nn_mode == "grid" else "nn_brute"
            if p["launches_nn_main"] > 0:
                kf[main_name] = kernel_figures(p, p["ms_nn_main"], p["launches_nn_main"], p["pairs_nn_main"], 20.0)
            if p["launches_nn_certify"] > 0:
                kf[cert_name] = kernel_figures(p, p["ms_nn_certify"], p["launches_nn_certify"], p["pairs_nn_certify"], cert_bytes)
            if p["launches_nn_listed"] > 0:
                # (the listed search finishes the ~2 % of the queries whose certificate failed: priced with the whole of FindClosests' 20 B/pt
                # it read "51 % of the roofline"; it has no fraction of its own -- find_closests_per_iteration prices the iteration's search as a whole)
                kf["nn_ball_listed"] = kernel_figures(p, p["ms_nn_listed"], p["launches_nn_listed"], p["pairs_nn_listed"], None)
            if p["launches_error_elements"] > 0:
                # (fused path: most of these launches are the early-exit form, so the average says nothing about the kernel;
                # the entry is kept for the time it takes per step)
                kf["accumulate"] = kernel_figures(p, p["ms_error_elements"], p["launches_error_elements"], p["pairs_error_elements"], None if fused_on else 24.0 * RHO)
                if fused_on:
                    kf["accumulate"]["mostly_early_exit_launches"] = True
            # FindClosests of one iteration as a whole: every launch that belongs to it, 20 B per source point once
            it_fused, it_split = p["launches_nn_main"], p["launches_nn_certify"]
            refine_per_it = p["ms_nn_refine"] / max(1, it_fused + it_split)
            fc = {}
            if it_split > 0:
                ms = (p["ms_nn_certify"] + p["ms_nn_listed"]) / it_split + refine_per_it
                pairs = kf[cert_name]["pairs_per_launch"]
                # (fused path: these launches also carry the iteration's ErrorElements sums -- priced with 20 B/pt all the same here)
                fc["certify_listed_refine"] = dict(iteration_launches=it_split, ms_per_iteration_launch=ms, pairs_per_launch=pairs,
                                                   achieved=pairs * ns * 20.0 / (ms * 1e-3) / 1e9)
            if it_fused > 0:
                ms = p["ms_nn_main"] / it_fused + refine_per_it
                pairs = kf[main_name]["pairs_per_launch"]
                fc["fused_refine"] = dict(iteration_launches=it_fused, ms_per_iteration_launch=ms, pairs_per_launch=pairs,
                                          achieved=pairs * ns * 20.0 / (ms * 1e-3) / 1e9)
            return kf, fc
        # timed region: the one class carrying events; every other kernel: the untimed step with events around every launch
        kf_timed, _ = all_kernels(nn_prof)
        kf_all, fc = all_kernels(prof)
        kf = {k: dict(kf_timed[k], timed=True) if k in kf_timed else dict(v, timed=False) for k, v in kf_all.items()}
        cands = [k for k in kf_timed if kf_timed[k]["achieved"] is not None] or [k for k in kf if kf[k]["achieved"] is not None]
        dom = max(cands, key=lambda k: kf[k]["total_ms"])
        nn_ms, pairs_per_launch, nn_bytes, achieved = (kf[dom][k] for k in ("avg_launch_ms", "pairs_per_launch", "bytes_per_launch", "achieved"))
        kf_alone, _ = all_kernels(alone)
        alone_f = kf_alone.get(dom, dict(avg_launch_ms=0.0, pairs_per_launch=0, achieved=0.0))
        alone_ms, alone_pairs, alone_gbs = alone_f["avg_launch_ms"], alone_f["pairs_per_launch"], alone_f["achieved"]

        def traffic_of(kernel, pairs, scattered=False):
            """HBM bytes per launch from the committed counter passes (profiles/traffic_<kernel>.json: FETCH_SIZE / WRITE_SIZE passes
            of the same batch, corrected with the factors calibrated on known-byte kernels of the same access shape)."""
            tj = os.path.join(ROOT, "profiles", f"traffic_{kernel}.json")
            if not os.path.exists(tj):
                return None
            try:
                tdat = json.load(open(tj))
                # (a record is only used while the kernel sources it names are what it was measured on; one that names none is undated)
                if not tdat.get("source_files") or tdat.get("source_sha") != _source_sha(tdat["source_files"]):
                    return None
                if tdat.get("nn_mode") == args.nn_mode and tdat.get("source_points") in (None, ns):
                    # scattered 32-64 B sectors (the listed search): the counters at factor 1, as the scatter calibration found
                    # (profiles/r04_traffic_calibration.json calib_scatter); streaming shapes: FETCH_SIZE x 2
                    key = "hbm_bytes_per_launch_uncorrected" if scattered and "hbm_bytes_per_launch_uncorrected" in tdat else "hbm_bytes_per_launch"
                    return int(tdat[key] * pairs / tdat["pairs_per_launch"])   # per point, so it scales with the pairs
            except Exception:
                pass
            return None
        traffic = traffic_of(dom, pairs_per_launch)
        alg_bytes = algorithmic_bytes_per_alignment(ns, nt_mean)
        out = {
            "metric": "scan-pair alignments/sec (120k-pt KITTI-64, 20 ICP iters)",
            "value": round(value, 2), "unit": "alignments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE config #2: IcpFast point-to-plane, 120k-pt synthetic Velodyne-64 scan pair "
                                   f"vs CalculateNormals target (~{int(nt_mean)} pts), exactly 20 iterations, rho 0.7; {D} distinct consecutive "
                                   f"pairs of a synthetic drive (speed 6-10 m/s, yaw rate +-0.2 rad/s, seed 5), guess = "
                                   + ("identity (SURVEY cfg 2 / cfg 4)" if args.headline == "identity" else "previous pair's motion"),
                       "pairs_per_gpu": B, "global_pairs_per_step": n_total, "distinct_pairs": D, "source_points": ns,
                       "target_points_mean": int(nt_mean), "iterations": ICP_ITERS, "nn_mode": args.nn_mode,
                       "grid_cell_m": args.cell, "guess": args.headline, "split_after": head["split"],
                       "parallelism": f"pairs round-robin over {world} GPU(s), one RCCL gather of poses",
                       "rccl_ranks": coll["ranks_in_all_reduce"], "collective_backend": coll["backend"], "rank_devices": coll["devices"],
                       "split_after_by_rank": [sp for _, _, sp in by_rank], "profiled_class_by_rank": [c for _, c, _ in by_rank]},
            "roofline": {"bound": "hbm", "kernel": dom,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "bytes_per_launch": nn_bytes, "bytes_per_point": kf[dom]["bytes_per_point"], "pairs_per_launch": pairs_per_launch,
                         "avg_launch_ms": round(nn_ms, 4), "launches_timed": kf[dom]["launches"],
                         "ms_per_step_by_kernel": {k: round(v["total_ms"] / (args.steps if v["timed"] else 1), 3) for k, v in kf.items()},
                         "kernels": {k: {"avg_launch_ms": round(v["avg_launch_ms"], 4), "pairs_per_launch": v["pairs_per_launch"],
                                         "launches": v["launches"], "in_timed_region": v["timed"], "algorithmic_bytes_per_point": v["bytes_per_point"],
                                         "algorithmic_bytes_per_launch": v["bytes_per_launch"],
                                         "achieved": None if v["achieved"] is None else round(v["achieved"], 2),
                                         "frac": None if v["achieved"] is None else round(v["achieved"] / HBM_PEAK_GBS, 5),
                                         "traffic": traffic_of(k, v["pairs_per_launch"], scattered=(k == "nn_ball_listed")),
                                         **({"mostly_early_exit_launches": True} if v.get("mostly_early_exit_launches") else {})}
                                     for k, v in kf.items()},
                         "fused_certificate_pass": {"on": bool(fused_on), "iterations_carried_mean": float(np.mean([s_["fused_iterations"] for s_ in head["stats"]])),
                                                    "iterations_carried_min": int(min(s_["fused_iterations"] for s_ in head["stats"])),
                                                    "note": "iterations (of 20) whose normal-equation sums came from nn_certify_acc; the rest: nn_ball_lds iterations "
                                                            "before the switch and predictions that missed, summed by accumulate"},
                         "find_closests_per_iteration": {k: {"iteration_launches": v["iteration_launches"], "ms_per_iteration_launch": round(v["ms_per_iteration_launch"], 4),
                                                             "pairs_per_launch": v["pairs_per_launch"], "algorithmic_bytes_per_point": 20.0,
                                                             "achieved": round(v["achieved"], 2), "frac": round(v["achieved"] / HBM_PEAK_GBS, 5)}
                                                         for k, v in fc.items()},
                         "refinement_launches_ms_per_step": round(prof["ms_nn_refine"], 3),
                         "note": "The timed region carries HIP events around ONE kernel class -- the one an untimed step with events around every launch found "
                                 "the largest (kernels.*.in_timed_region) -- because every event pair is a barrier between two launches and bracketing all "
                                 "classes took 6 % off the batch rate; the other kernels and find_closests_per_iteration come from that untimed step of the "
                                 "same batch.  kernels.*: each kernel alone, priced with the algorithmic bytes of the reference function(s) it implements (nn_certify and "
                                 "nn_ball_listed each carry the full 20 B/pt of FindClosests; nn_certify_acc = FindClosests + ErrorElements / ComputePointToPlane "
                                 "in one pass = 20 + 24 rho B/pt; read find_closests_per_iteration for FindClosests as a "
                                 "whole: certificate pass + listed search + validate / ring / fallback launches of one iteration, 20 B/pt once)",
                         "alone": {"note": "same kernel on one stream, not sharing the GPU with the other half-batch",
                                   "pairs_per_launch": alone_pairs, "avg_launch_ms": round(alone_ms, 4),
                                   "achieved": round(alone_gbs, 2), "frac": round(alone_gbs / HBM_PEAK_GBS, 5)},
                         "reference_search": _reference_search_roofline(figures.get("reference_search_eps3.16"), ns, nt_mean, B, world),
                         "shape_rate": _shape_rate(),
                         "whole_alignment": {"algorithmic_bytes": alg_bytes,
                                             "achieved_GBs": round(alg_bytes * value / world / 1e9, 2),
                                             "frac": round(alg_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5)}},
            "kernel_ms_per_step": {k: round(v, 3) for k, v in prof.items() if k.startswith("ms_")},
            "profiled_class": {2: "nn kernels (fused search, certificate pass)", 3: "accumulate", 4: "listed search"}[prof_class],
            "parity": {"worst_rot_err_vs_truth_rad": h_rot, "worst_trans_err_vs_truth_m": h_t, "median_trans_err_vs_truth_m": h_med},
            "workload_generation_s": round(t_gen, 1),
        }
        if not args.no_cpu_baseline and rank == 0:     # (the CPU oracle legs: rank 0 alone, the other ranks wait at the final barrier)
            cpu, par = cpu_baseline_and_parity(work, figures, head, head_key, args.cpu_pairs or min(D, 64), world)
            out["parity"].update(par)
            if world == 1:
                out["cpu_baseline"] = cpu
        out["figures"] = {k: {kk: vv for kk, vv in f.items() if kk not in ("T", "guess_key", "max_iteration", "early_exit", "nn_eps", "prof_all")}
                          for k, f in figures.items()}
        for k, f in out["figures"].items():      # the same whole-alignment roofline for every figure (its own iteration count)
            by = algorithmic_bytes_per_alignment(ns, nt_mean, iters=float(f["iterations"]))
            f["whole_alignment_roofline"] = {"algorithmic_bytes": by, "achieved_GBs": round(by * f["value"] / world / 1e9, 2),
                                             "frac": round(by * f["value"] / world / 1e9 / HBM_PEAK_GBS, 5)}
        if "reference_search_eps3.16" in out["figures"]:
            out["figures"]["reference_search_eps3.16"]["note"] = (
                "nn_mode NABO: libnabo 1.0.7's KDTREE_LINEAR_HEAP tree rebuilt per Align and its epsilon = 3.16 approximate knn "
                "(icp_fast.cc:169-180, 464-467) walked on the device; *_vs_oracle here is against the oracle run with the SAME "
                "approximate search -- the parity the exact modes cannot have (see parity.exact_vs_reference_eps3.16)")
        # the reference's own search semantics as a peer of the headline: the only mode a maintainer can diff against a libnabo build
        fr = out["figures"].get("reference_search_eps3.16")
        if fr:
            rs = (out.get("roofline") or {}).get("reference_search") or {}
            out["reference_search"] = {"value": fr["value"], "unit": "alignments/s", "nn_mode": "nabo", "nn_epsilon": 3.16,
                                       "whole_alignment_frac": (fr.get("whole_alignment_roofline") or {}).get("frac"),
                                       "worst_rot_vs_oracle_rad": fr.get("worst_rot_vs_oracle_rad"), "worst_trans_vs_oracle_m": fr.get("worst_trans_vs_oracle_m"),
                                       "pairs_checked_vs_oracle": fr.get("pairs_checked_vs_oracle"),
                                       "walked_queries_per_alignment_after_iteration_0": rs.get("walked_queries_per_alignment_after_iteration_0"),
                                       "dominant_class": rs.get("kernel"), "dominant_class_ms_per_step": rs.get("kernel_ms_per_step"),
                                       "oracle": "oracle/csrc/smref_icp.c with libnabo's tree and epsilon = 3.16 knn restated (the SAME approximate search)"}
    m.close()
    if rank == 0 and world == 1 and not args.no_other:
        out["single_pair"] = single_pair_latency(work[0], dev_index)
        out["other_workloads"] = other_workloads(dev, not args.no_cpu_baseline)
        if not args.no_end_to_end:
            out["end_to_end"] = end_to_end(dev, n_scans=1025)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def _reference_search_roofline(f, ns, nt_mean, pairs_per_gpu, world):
    """The reference's own search semantics (nn_mode NABO) against the same roofline: the whole alignment, and the kernel class the
    step spends the most time in (events around every launch of one untimed step).  Classes: the full walk of iteration 0
    (nn_nabo<4, false>, every query: 20 B/pt), the certificate pass (nn_certify<., true> / nn_certify_acc<., true>: 20 B/pt, + 24 rho
    where it also sums), the list walk of the queries whose traversal certificate failed (nn_nabo<1, true>: priced with FindClosests'
    20 B for every query it walks, from the per-alignment count of walked queries), accumulate."""
    if not f or not f.get("prof_all"):
        return None
    p = f["prof_all"]
    by = algorithmic_bytes_per_alignment(ns, nt_mean)
    walked = max(0.0, f["searched_queries_per_alignment"] - ns)           # iterations >= 1 (iteration 0 walks every query)
    cls = {"nn_nabo_full_walk": (p["ms_nn_main"], 20.0 * ns * p["pairs_nn_main"]),          # (pairs_*: pairs covered, summed over the class's launches)
           "nn_certify": (p["ms_nn_certify"], 20.0 * ns * p["pairs_nn_certify"]),
           "nn_nabo_list_walk": (p["ms_nn_listed"], 20.0 * walked * pairs_per_gpu),
           "accumulate": (p["ms_error_elements"], None)}
    dom = max(cls, key=lambda k: cls[k][0])
    ms, bytes_step = cls[dom]
    ach = None if bytes_step is None or ms <= 0 else bytes_step / (ms * 1e-3) / 1e9
    return {"value": f["value"], "unit": "alignments/s", "whole_alignment_frac": round(by * f["value"] / world / 1e9 / HBM_PEAK_GBS, 5),
            "kernel": dom, "kernel_ms_per_step": round(ms, 3), "kernel_bytes_per_step": bytes_step,
            "achieved": None if ach is None else round(ach, 2), "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 5),
            "walked_queries_per_alignment_after_iteration_0": walked,
            "ms_per_step_by_class": {k: round(v[0], 3) for k, v in cls.items()} | {"refine_validate": round(p["ms_nn_refine"], 3), "finalize": round(p["ms_solve"], 3), "kd_build_and_grid": round(p["ms_prepare"], 3)},
            "note": "nn_mode NABO: libnabo's tree + epsilon = 3.16 search on the device (icp_fast.cc:174, 464-467); one untimed step with HIP events around every launch"}


def cpu_baseline_and_parity(work, figures, head, head_key, n_cpu, world):
    """The oracle's C restatement of IcpFast::Align run on the host on the SAME clouds and guesses as the timed batch:
    (1) every timed pose of the sampled pairs is compared with it (parity vs the oracle, not vs the truth), and
    (2) its wall time is the cpu_baseline: 1 thread = icp_fast.cc as written; all usable cores = FindClosests under
    OpenMP, which is how libnabo's knn runs by default."""
    import staticmapping_amd as sm
    from oracle import cref
    D = len(work)
    n_cpu = min(n_cpu, D)
    par = {}
    t_used, blocks = 0.0, {}
    worst_rot = worst_t = 0.0
    for d in range(n_cpu):
        t = time.perf_counter()
        ref = oracle_pose(work[d], work[d][head_key], ICP_ITERS, False)
        t_used += time.perf_counter() - t
        for k, v in ref["block_times"].items():
            blocks[k] = blocks.get(k, 0.0) + v
        da, dt = sm.se3_error(head["T"][d], ref["result"])        # global pair d of the timed batch IS distinct pair d
        worst_rot, worst_t = max(worst_rot, da), max(worst_t, dt)
    par["worst_rot_vs_oracle_rad"] = worst_rot
    par["worst_trans_vs_oracle_m"] = worst_t
    par["pairs_checked_vs_oracle"] = n_cpu
    par["oracle"] = "oracle/csrc/smref_icp.c, exact 1-NN, same clouds and guess as the timed batch"
    # the other figures: a bounded subset each
    for name, f in figures.items():
        if f["guess_key"] == head_key and not f["early_exit"] and not f.get("nn_eps"):
            f["worst_rot_vs_oracle_rad"], f["worst_trans_vs_oracle_m"], f["pairs_checked_vs_oracle"] = worst_rot, worst_t, n_cpu
            continue
        wr = wt = 0.0
        it_ok = True
        sub = list(range(0, D, max(1, D // 8)))[:8]
        if f.get("nn_eps"):                      # the reference's own search: the figure inside north_star's tolerance of the reference -- 64 pairs
            sub = list(range(min(D, 64)))
        if f["guess_key"] == "mixed":            # pairs of both kinds: even global index = extrapolated guess, odd = identity
            sub = [min(D - 1, d + (k % 2)) for k, d in enumerate(sub)]
        many = cref.usable_cores() if len(sub) > 8 else 1
        for d in sub:
            gkey = ("guess_cv" if d % 2 == 0 else "guess_id") if f["guess_key"] == "mixed" else f["guess_key"]
            ref = oracle_pose(work[d], work[d][gkey], f["max_iteration"], f["early_exit"], nthreads=many, nn_eps=f.get("nn_eps"))
            da, dt = sm.se3_error(f["T"][d], ref["result"])
            wr, wt = max(wr, da), max(wt, dt)
        f["worst_rot_vs_oracle_rad"], f["worst_trans_vs_oracle_m"], f["pairs_checked_vs_oracle"] = wr, wt, len(sub)
    # the reference's own approximation, on the same pairs: exact search (GPU, oracle) vs libnabo eps = 3.16 restated
    sub = list(range(0, D, max(1, D // 4)))[:4]
    wr = wt = 0.0
    for d in sub:
        ex = oracle_pose(work[d], work[d][head_key], ICP_ITERS, False)
        ap = oracle_pose(work[d], work[d][head_key], ICP_ITERS, False, nn_eps=3.16)
        da, dt = sm.se3_error(ex["result"], ap["result"])
        wr, wt = max(wr, da), max(wt, dt)
    par["exact_vs_reference_eps3.16"] = {"worst_rot_rad": wr, "worst_trans_m": wt, "pairs": len(sub),
                                         "note": "oracle with exact 1-NN vs oracle with libnabo's eps = 3.16 search restated "
                                                 "(icp_fast.cc:174): the reference's own approximation, not a GPU error"}
    cores = cref.usable_cores()
    sub = list(range(min(8, D)))
    t = time.perf_counter()
    for d in sub:
        oracle_pose(work[d], work[d][head_key], ICP_ITERS, False, nthreads=cores)
    t_all = time.perf_counter() - t
    # the baseline proper: the reference's own search, libnabo's epsilon = 3.16 approximate knn (icp_fast.cc:174), which is what
    # its CPU path runs and the faster of the two on a CPU; the exact kd-tree figures above stay as a sub-field
    n_eps = min(D, 32)
    t = time.perf_counter()
    eps_blocks = {}
    for d in range(n_eps):
        ref = oracle_pose(work[d], work[d][head_key], ICP_ITERS, False, nn_eps=3.16)
        for k, v in ref["block_times"].items():
            eps_blocks[k] = eps_blocks.get(k, 0.0) + v
    t_eps = time.perf_counter() - t
    t = time.perf_counter()
    for d in sub:
        oracle_pose(work[d], work[d][head_key], ICP_ITERS, False, nthreads=cores, nn_eps=3.16)
    t_eps_all = time.perf_counter() - t
    cpu = {"value": round(n_eps / t_eps, 3), "unit": "alignments/s", "cores": 1, "kind": "port",
           "sample": f"{n_eps} distinct 120k-pt pairs of the timed batch, one 20-iteration alignment each, C restatement "
                     f"oracle/csrc/smref_icp.c with the reference's search (libnabo tree rebuilt per Align + epsilon = 3.16 knn restated, "
                     f"icp_fast.cc:169-180, 464-467; gcc -O2), 1 thread as icp_fast.cc is written",
           "block_seconds_per_alignment": {k: round(v / n_eps, 5) for k, v in eps_blocks.items()},
           "all_cores": {"value": round(len(sub) / t_eps_all, 3), "cores": cores,
                         "note": "same code, ApplyTransform / FindClosests / normal equations under OpenMP on every usable host "
                                 "core (affinity mask, cgroup quota and physical cores respected); libnabo's knn is OpenMP-parallel "
                                 "by default, so this is the reference's likely deployment"},
           "exact_search": {"value": round(n_cpu / t_used, 3), "cores": 1,
                            "sample": f"{n_cpu} pairs, the same code with an exact kd-tree 1-NN (the oracle of the headline's parity check)",
                            "block_seconds_per_alignment": {k: round(v / n_cpu, 5) for k, v in blocks.items()},
                            "all_cores": {"value": round(len(sub) / t_all, 3), "cores": cores}}}
    return cpu, par


def single_pair_latency(w, device):
    """What a drop-in user of the sequential front end gets (builder/map_builder.cc:260-397): ONE Align at a time."""
    import staticmapping_amd as sm
    m = sm.IcpFastHip(device=device, pair_slots=1, max_source_points=len(w["src"]), max_target_points=len(w["q"]),
                      max_iteration=ICP_ITERS, early_exit=0, dist_outlier_ratio=RHO)
    m.set_input_source(w["src"]); m.set_input_target(w["q"], w["n"])
    out = {"workload": "one 120k-pt pair of the batch, guess = previous pair's motion, blocking smhip_icp_align calls"}

    def timed(reps=40):                 # the median call (a mean of 20 moved by 0.1 ms with one hiccup of the host)
        m.align(w["guess_cv"])
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            m.align(w["guess_cv"])
            ts.append(time.perf_counter() - t)
        return round(float(np.median(ts)) * 1e3, 4)
    m.set_target_cache(False)
    out["ms_20_iterations_rebuild_every_align"] = timed()
    m.set_target_cache(True)
    out["ms_20_iterations_target_kept"] = timed()
    out["fused_iterations_of_20"] = int(m.last_stats[0]["fused_iterations"])
    m.set_options(max_iteration=100, early_exit=1)
    out["ms_early_exit_target_kept"] = timed()
    out["iterations_early_exit"] = int(m.last_stats[0]["iterations"])
    used, fell = m.single_launch_counts()
    out["aligns_as_one_launch"] = used
    out["launches_that_stopped_themselves"] = fell      # (each would have been redone as separate launches: smhip_icp_single_launch_counts)
    # the same Align as separate launches per iteration (round 5's form): what the one cooperative launch (csrc/icp_one.hip) replaced
    m.set_options(max_iteration=ICP_ITERS, early_exit=0, no_single_kernel=1)
    out["ms_20_iterations_target_kept_separate_launches"] = timed()
    out["form"] = "one cooperative launch per Align: the whole loop of icp_fast.cc:484-523 + the score, workgroups resident, grid barriers (csrc/icp_one.hip)"
    m.close()
    return out


# ----------------------------------------------------------------------------------------------------------------
# End to end: files -> poses through the C++ driver (BASELINE config #4 on one GPU)
# ----------------------------------------------------------------------------------------------------------------
def end_to_end(dev, n_scans=513, n_points=N_POINTS):
    """What a user of the sequence driver gets per second, nothing resident beforehand: a generated drive of `n_scans` KITTI .bin
    files -> staticmapping_amd/lib/smhip_shard (read, upload, device CalculateNormals of every scan as the next pair's
    target, 20-iteration IcpFast, one RCCL gather, kitti_pose.txt).  The resident-input figure of the headline leaves all of
    that out; this object is where the two are put side by side."""
    import shutil
    import subprocess
    import tempfile
    from staticmapping_amd import kitti, synth, build
    exe = build.SHARD_EXE
    if not os.path.exists(exe):
        return {"error": "smhip_shard not built"}
    root = tempfile.mkdtemp(prefix="smhip_bench_drive_", dir="/tmp")
    try:
        free = shutil.disk_usage(root).free
        while n_scans > 65 and n_scans * n_points * 16 + (1 << 30) > free:
            n_scans = (n_scans - 1) // 2 + 1
        drive = os.path.join(root, "drive")
        os.makedirs(drive)
        t = time.perf_counter()
        poses = synth.drive_poses(n_scans, seed=5, speed=8.0, hz=10.0, yaw_rate_max=0.2)
        scene = synth.make_drive_scene(poses, seed=5)
        for k, P in enumerate(poses):
            kitti.write_bin(kitti.scan_path(drive, k), synth.velodyne_scan(synth.scene_near(scene, P[:3, 3]), P, seed=1000 + k, n_points=n_points, device=dev))
        t_gen = time.perf_counter() - t
        out = {"workload": f"BASELINE config #4 on one GPU: {n_scans}-scan synthetic drive (8 m/s, 10 Hz, yaw rate +-0.2 rad/s, seed 5; {n_points} points "
                           f"per scan) as KITTI .bin files -> smhip_shard (C++ driver: read, upload, device CalculateNormals, 20-iteration IcpFast, "
                           f"RCCL gather, kitti_pose.txt), guess = 0.8 m forward", "scans": n_scans, "generation_s": round(t_gen, 1)}
        pose_file = os.path.join(root, "kitti_pose.txt")
        best = None
        for rep in range(2):            # the first run pays the page-cache misses of the fresh files and the library's first-use allocations
            r = subprocess.run([exe, "--scans", drive, "--gpus", "1", "--guess-tx", "0.8", "--iterations", str(ICP_ITERS), "--early-exit", "0",
                                "--out", pose_file], text=True, capture_output=True, timeout=600)
            if r.returncode != 0:
                return dict(out, error=(r.stderr or r.stdout)[-400:])
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
            if best is None or j["pairs_per_s"] > best["pairs_per_s"]:
                best = j
        est = kitti.read_poses(pose_file)
        base = np.linalg.inv(poses[0])
        truth = np.stack([base @ P for P in poses])
        errs = []
        for k in range(len(truth) - 1):
            Te = np.linalg.inv(est[k]) @ est[k + 1]
            Tt = np.linalg.inv(truth[k]) @ truth[k + 1]
            errs.append(np.linalg.norm(Te[:3, 3] - Tt[:3, 3]))
        out.update({"value": best["pairs_per_s"], "unit": "pairs/s", "pairs": best["pairs"], "seconds": best["seconds"],
                    "host_side_split_s": {"wait_for_readers": best.get("wait_for_readers_s_rank0"), "upload_and_morton_order": best.get("upload_s_rank0"),
                                          "device_calculate_normals": best.get("prepare_targets_s_rank0"),
                                          "read_upload_prepare_total": best["read_upload_prepare_s_rank0"]},
                    "clock": "starts before the driver's readers open the first file, stops when the gathered poses are on the host side of the "
                             "collective; before it: the handle, its workspaces and one batch of 32 made-up 8 192-point scans through the same calls "
                             "(code objects loaded, copy and side streams created -- a process pays that once, not per sequence; its search history forgotten again)",
                    "warmup_batch_before_the_clock_s": best.get("warmup_batch_before_the_clock_s"),
                    "steady_state_pairs_per_s": best.get("steady_state_pairs_per_s_rank0"),
                    "steady_state_note": "pairs of the batches after the first / time from the first batch's alignments being enqueued to the last "
                                         "batch's: the batch period once the pipeline is full (a 1 024-pair run is four batches: a quarter of it "
                                         "is filling and draining); `value` is the whole run",
                    "host_side_split_note": "device_calculate_normals is the host blocked in the batch's target preparation, which runs on the stream "
                                            "behind the previous batch's alignments: it holds their remaining time too (per 256-pair batch: alignments "
                                            "~15 ms, preparation ~15 ms of which kd_forest_build 11-13; side by side on two streams they take the same "
                                            "30 ms -- both are bound by instruction issue on the CUs, not by HBM)",
                    "mean_iterations": best["mean_iterations"], "unfinished_pairs": best["unfinished_pairs"],
                    "relative_pose_error_vs_generating_motion_m": {"median": float(np.median(errs)), "p95": float(np.percentile(errs, 95)), "max": float(np.max(errs))}})
        return out
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


# ----------------------------------------------------------------------------------------------------------------
# BASELINE configs #3 and #5 on one GPU
# ----------------------------------------------------------------------------------------------------------------
def _submap_case(n_scans, n_target, seed, device, n_points=N_POINTS):
    from staticmapping_amd import synth
    poses = synth.drive_poses(n_scans + 1, seed=seed, speed=8.0, yaw_rate_max=0.1)
    scene = synth.make_drive_scene(poses, seed=seed)
    scans = [synth.velodyne_scan(synth.scene_near(scene, P[:3, 3]), P, seed=40 * seed + k, n_points=n_points, device=device)
             for k, P in enumerate(poses)]
    tgt = np.concatenate([s[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3] for s, P in zip(scans[:n_scans], poses[:n_scans])])
    rng = np.random.default_rng(seed)
    if n_target < len(tgt):
        tgt = tgt[np.sort(rng.choice(len(tgt), size=n_target, replace=False))]
    tgt = np.ascontiguousarray(tgt.astype(np.float32))
    T = poses[n_scans]
    G = T.copy()
    G[:3, 3] += T[:3, :3] @ np.array([-0.3, 0.0, 0.0])                     # truth perturbed by 0.3 m, 1 degree (SURVEY cfg 3)
    c, s = np.cos(np.deg2rad(1.0)), np.sin(np.deg2rad(1.0))
    G[:3, :3] = T[:3, :3] @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    return np.ascontiguousarray(scans[n_scans][:, :3]), tgt, T, G


def _source_sha(files):
    """sha256 over the named kernel sources (relative to the repository): what a recorded counter pass is dated with"""
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted(files):
        with open(os.path.join(ROOT, f), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()


def _kernel_traffic(kernel, units):
    """HBM bytes per launch of `kernel` from its recorded counter passes (profiles/traffic_<kernel>.json: separate FETCH_SIZE and
    WRITE_SIZE passes), scaled to `units` of work per launch -- or None when the record is missing or was taken from other source
    than the tree holds now (the record carries the sha256 of the kernel's source files)."""
    tj = os.path.join(ROOT, "profiles", f"traffic_{kernel}.json")
    try:
        d = json.load(open(tj))
        if not d.get("source_files") or d.get("source_sha") != _source_sha(d["source_files"]):
            return None
        return int(d["hbm_bytes_per_launch"] * units / d["units_per_launch"])
    except Exception:
        return None


def _shape_rate():
    """What the dominant kernel's streaming accesses alone reach on this GPU: a recorded run of tools/traffic_calib.sh (a kernel that
    reads 12 + 4 + 4 B and writes 4 B per element and computes nothing).  A reference point next to `peak`, not a replacement for it."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "r04_traffic_calibration.json")))["kernels"]["calib_certify"]
        return {"kernel": "calib_certify (tools/traffic_calib.hip)", "achieved": round(1e3 * c["streamed_TB_per_s"], 1), "unit": "GB/s",
                "measured": "recorded run, profiles/r04_traffic_calibration.json",
                "note": "HBM rate of a no-arithmetic kernel with nn_certify_acc's streaming accesses (12-byte row + int + float in, one float out)"}
    except Exception:
        return None


def other_workloads(dev, with_cpu):
    import staticmapping_amd as sm
    out = {}
    # ---- config #3: registrators::Ndt
    try:
        src, tgt, T, G = _submap_case(5, 500_000, 4, dev)
        m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
        m.set_input_source(src); m.set_input_target(tgt)
        reps = 10

        def time_align():
            m.align(G)
            t = time.perf_counter()
            for _ in range(reps):
                ok_, R_ = m.align(G)
            return (time.perf_counter() - t) / reps, R_
        # value: every Align rebuilds the voxel table and the fitness search structure, as the reference does (ndt.cc:54);
        # target_kept: both kept while the target is unchanged (many scans against one key frame, map_builder.cc:379-392)
        m.set_target_cache(False)
        dt, R = time_align()
        m.set_target_cache(True)
        dt_kept, R_kept = time_align()
        assert np.array_equal(R, R_kept)
        st = m.last_ndt_stats
        ns, nt_, V, C = len(src), len(tgt), st["voxels"], st["derivative_calls"]
        mbar = st["pairs_last"] / ns
        b_ndt = 12.0 * nt_ + 40.0 * V + C * ns * (12.0 + 36.0 * mbar) + 12.0 * ns + 4.0 * ns     # SURVEY §8(d) B_ndt
        # the path's dominant kernel by itself: computeDerivatives (ndt_derivatives_ctl), HIP events around back-to-back launches on
        # the matcher's stream at the pose the Align ended with; its algorithmic bytes = B_ndt's per-call term N_s (12 + 36 m)
        ms_k, pairs_k = m.time_derivatives(npairs=1, launches=50)
        b_call = 12.0 * ns + 36.0 * pairs_k
        entry = {"workload": "BASELINE config #3: registrators::Ndt, 120k-pt scan vs 500k-pt submap (5 merged scans), 1.0 m voxels, "
                             "guess = truth perturbed by 0.3 m / 1 deg, clouds resident",
                 "value": round(1.0 / dt, 2), "unit": "alignments/s", "ms_per_alignment": round(dt * 1e3, 3),
                 "iterations": st["iterations"], "derivative_calls": C, "voxels": V, "mean_neighbours": round(mbar, 3),
                 "roofline": {"bound": "hbm", "kernel": "ndt_derivatives_ctl", "achieved": round(b_call / ms_k / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(b_call / ms_k / 1e6 / HBM_PEAK_GBS, 5), "bytes_per_launch": b_call, "avg_launch_ms": round(ms_k, 5),
                              "launches_timed": 50, "pairs_per_launch": pairs_k,
                              "traffic": _kernel_traffic("ndt_derivatives_ctl", 1),
                              "note": "one evaluation of one pair (469 workgroups: the launch is as long as a workgroup lives, not bandwidth); "
                                      "the batch of 64 below is the kernel under load",
                              "whole_alignment": {"achieved": round(b_ndt / dt / 1e9, 2), "frac": round(b_ndt / dt / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": b_ndt,
                                                  "note": "whole Align (voxel table build + C computeDerivatives + fitness pass), SURVEY §8(d) B_ndt with measured V, C, m"}},
                 "submission": "one per Align: table build, the rounds of the device-resident Newton / More-Thuente driver, fitness pass; one synchronise",
                 "trans_err_vs_truth_m": sm.se3_error(R, T)[1],
                 "target_kept": {"value": round(1.0 / dt_kept, 2), "ms_per_alignment": round(dt_kept * 1e3, 3),
                                 "note": "voxel table (which is also the fitness search structure) kept across Aligns on an unchanged target "
                                         "(smhip_set_target_cache, default on); identical result"}}
        entry["workload"] = ("BASELINE config #3: registrators::Ndt, 120k-pt scan vs 500k-pt submap (5 merged scans), 1.0 m voxels, "
                             "guess = truth perturbed by 0.3 m / 1 deg, clouds resident")
        # the back end's form: six SubmapPairMatch tasks at once, a matcher each (map_builder.cc:655, 706-708) -- six handles
        # on six host threads (include/smhip/back_end.h SubmapMatcherPool); every Align still rebuilds everything
        try:
            import threading
            pool = [m]
            for _ in range(5):
                mk = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt))
                mk.set_input_source(src); mk.set_input_target(tgt)
                pool.append(mk)
            for mk in pool:
                mk.set_target_cache(False); mk.align(G)
            res = [None] * len(pool)

            def run(k):
                for _ in range(reps):
                    res[k] = pool[k].align(G)[1]
            ths = [threading.Thread(target=run, args=(k,)) for k in range(len(pool))]
            t = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt_pool = (time.perf_counter() - t) / (reps * len(pool))
            entry["six_concurrent_matchers"] = {"value": round(1.0 / dt_pool, 2), "unit": "alignments/s", "matchers": len(pool),
                                                "identical_to_single": bool(all(np.array_equal(r, R) for r in res)),
                                                "note": "six handles (own arena + stream) driven by six host threads, as the reference's "
                                                        "thread pool runs six SubmapPairMatch tasks; everything rebuilt per Align"}
            for mk in pool[1:]:
                mk.close()
            m.set_target_cache(True)
        except Exception as e:
            entry["six_concurrent_matchers"] = {"error": repr(e)}
        # ... and as ONE lock-step batch through the pair slots of one matcher (smhip_ndt_align_batch): the K Newton / More-Thuente
        # state machines advance together, every round's computeDerivatives calls are one launch, the K voxel tables are built in
        # one pass (one radix sort) and the K fitness scores in another.  64 slots hold this pair with 64 different guesses (the
        # pairs leave the rounds at different times); slot 0 carries the single call's guess and must return its bits.
        try:
            from staticmapping_amd import synth
            K = 64
            mb = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt), pair_slots=K)
            for k in range(K):
                mb.set_input_source(src, slot=k); mb.set_input_target(tgt, slot=k)
            gs = [G @ synth.make_pose(t=(0.01 * (k % 8), -0.01 * (k % 5), 0.0), rpy_deg=(0, 0, 0.05 * (k % 7))) for k in range(K)]
            gs[0] = G

            def time_batch():
                mb.align_batch(K, gs)
                t_ = time.perf_counter()
                for _ in range(3):
                    Rb_, scb_, stb_ = mb.align_batch(K, gs)
                return (time.perf_counter() - t_) / 3, Rb_, stb_
            mb.set_target_cache(False)
            dtb, Rb, stb = time_batch()
            mb.set_target_cache(True)
            dtb_kept, Rb_kept, _ = time_batch()
            calls = float(np.sum([s_["derivative_calls"] for s_ in stb]))
            b_batch = K * (12.0 * nt_ + 40.0 * V + 16.0 * ns) + calls * ns * (12.0 + 36.0 * mbar)       # SURVEY §8(d) B_ndt, summed over the pairs
            ms_kb, pairs_kb = mb.time_derivatives(npairs=K, launches=20)
            b_call_b = 12.0 * ns * K + 36.0 * pairs_kb
            entry["batch64"] = {"value": round(K / dtb, 2), "unit": "alignments/s", "pairs": K, "ms_per_batch": round(dtb * 1e3, 3),
                                "identical_to_single": bool(np.array_equal(Rb[0], R)), "iterations_min_max": [int(min(s_["iterations"] for s_ in stb)), int(max(s_["iterations"] for s_ in stb))],
                                "derivative_calls_total": int(calls),
                                "roofline": {"bound": "hbm", "kernel": "ndt_derivatives_ctl", "achieved": round(b_call_b / ms_kb / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": round(b_call_b / ms_kb / 1e6 / HBM_PEAK_GBS, 5), "bytes_per_launch": b_call_b, "avg_launch_ms": round(ms_kb, 5),
                                             "launches_timed": 20, "pairs_per_launch": pairs_kb, "evaluations_per_launch": K,
                                             "traffic": _kernel_traffic("ndt_derivatives_ctl", K),
                                             "whole_batch": {"achieved": round(b_batch / dtb / 1e9, 2), "frac": round(b_batch / dtb / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": b_batch}},
                                "target_kept": {"value": round(K / dtb_kept, 2), "ms_per_batch": round(dtb_kept * 1e3, 3), "identical_result": bool(np.array_equal(Rb, Rb_kept))},
                                "note": "smhip_ndt_align_batch: device-resident state machines, one ndt_derivatives_ctl + one ndt_ctl_step launch per round over all running "
                                        "pairs, one submission per batch (pclomp/ndt_omp_impl.hpp:81-171, 757-916; builder/map_builder.cc:399-446, 655); value = everything rebuilt per Align"}
            mb.close()
        except Exception as e:
            entry["batch64"] = {"error": repr(e)}
        if with_cpu:
            from oracle import cref
            t = time.perf_counter()
            ref = cref.ndt_align(src, tgt, guess=G, nthreads_deriv=6, nthreads_other=1)
            t_cpu = time.perf_counter() - t
            da, dtt = sm.se3_error(R, ref["result"])
            entry["parity"] = {"rot_vs_oracle_rad": da, "trans_vs_oracle_m": dtt, "iterations_oracle": ref["iterations"],
                               "derivative_calls_oracle": ref["derivative_calls"],
                               "fitness_gpu": m.get_fitness_score(), "fitness_oracle": ref["score"]}
            cores = cref.usable_cores()
            t = time.perf_counter()
            cref.ndt_align(src, tgt, guess=G, nthreads_deriv=cores, nthreads_other=cores)
            t_all = time.perf_counter() - t
            entry["cpu_baseline"] = {"value": round(1.0 / t_cpu, 3), "unit": "alignments/s", "cores": min(6, cores), "kind": "port",
                                     "sample": "one alignment of the same clouds and guess, C restatement oracle/csrc/smref_ndt.c, computeDerivatives "
                                               "on 6 OpenMP threads as ndt.cc:32 sets, everything else 1 thread",
                                     "block_seconds": {k: round(v, 4) for k, v in ref["block_times"].items()},
                                     "all_cores": {"value": round(1.0 / t_all, 3), "cores": cores}}
        m.close()
        out["ndt"] = entry
    except Exception as e:           # a failure here must not take the headline line down
        out["ndt"] = {"error": repr(e)}
    # ---- config #5: registrators::NdtWithGicp
    try:
        src, tgt, T, G = _submap_case(20, 2_000_000, 6, dev)
        m = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt))
        m.set_input_source(src); m.set_input_target(tgt)
        reps = 5

        def time_gicp():
            ok, R = m.align(G)
            t = time.perf_counter()
            for _ in range(reps):
                ok, R = m.align(G)
            return (time.perf_counter() - t) / reps, ok, R
        m.set_target_cache(False)                 # everything rebuilt in every Align, as ndt_gicp.cc does: the figure
        dt, ok, R = time_gicp()
        m.set_target_cache(True)
        dt_kept, ok_k, R_k = time_gicp()
        st = m.last_gicp_stats
        n_s, n_t = st["n_source"], st["n_target"]
        # SURVEY §8(d) GICP: covariance build (N_s + N_t)(12 + 20 * 12) + per outer iteration N_s (12 + 8) + N_corr (12 + 36 + 36),
        # plus the voxel filter reading both raw clouds once (12 B / point)
        b = 12.0 * (len(src) + len(tgt)) + (n_s + n_t) * (12.0 + 240.0) + max(1, st["gicp_iterations"]) * (n_s * 20.0 + st["gicp_correspondences"] * 84.0)
        out["ndt_gicp"] = {"workload": "BASELINE config #5: registrators::NdtWithGicp, 120k-pt scan vs 2M-pt submap (20 merged scans), "
                                       "ApproximateVoxelGrid 0.2 m -> NDT -> GICP, clouds resident",
                           "value": round(1.0 / dt, 2), "unit": "alignments/s", "ms_per_alignment": round(dt * 1e3, 3), "ok": bool(ok),
                           "stats": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in st.items()},
                           "roofline": {"bound": "hbm", "achieved": round(b / dt / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(b / dt / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": b,
                                        "note": "whole Align; SURVEY §8(d) GICP bytes with the measured down-sampled sizes"},
                           "trans_err_vs_truth_m": sm.se3_error(R, T)[1],
                           "target_kept": {"value": round(1.0 / dt_kept, 2), "ms_per_alignment": round(dt_kept * 1e3, 3),
                                           "identical_result": bool(np.array_equal(R, R_k)),
                                           "note": "down-sampled target, NDT voxel table, correspondence grid and the target's GICP covariances kept "
                                                   "across Aligns on an unchanged target (smhip_set_target_cache, default on)"},
                           "cpu_baseline": None}
        # the back end's form (six SubmapPairMatch tasks, a matcher each): six handles on six host threads, everything rebuilt
        try:
            import threading
            pool = [m]
            for _ in range(5):
                mk = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt))
                mk.set_input_source(src); mk.set_input_target(tgt)
                pool.append(mk)
            for mk in pool:
                mk.set_target_cache(False); mk.align(G)
            res = [None] * len(pool)

            def run(k):
                for _ in range(reps):
                    res[k] = pool[k].align(G)[1]
            ths = [threading.Thread(target=run, args=(k,)) for k in range(len(pool))]
            t = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt_pool = (time.perf_counter() - t) / (reps * len(pool))
            out["ndt_gicp"]["six_concurrent_matchers"] = {"value": round(1.0 / dt_pool, 2), "unit": "alignments/s", "matchers": len(pool),
                                                          "identical_to_single": bool(all(np.array_equal(r, R) for r in res)),
                                                          "note": "six handles (own arena + stream) on six host threads, everything rebuilt per Align"}
            for mk in pool[1:]:
                mk.close()
            m.set_target_cache(True)
        except Exception as e:
            out["ndt_gicp"]["six_concurrent_matchers"] = {"error": repr(e)}
        # ... and as ONE lock-step batch through the jobs of one matcher (smhip_ndt_gicp_align_batch): the stages run over all jobs
        # at once (the NDT rounds as in ndt.batch64; the 20-NN covariances, the correspondence steps and every round of functor
        # evaluations one launch each over the jobs still running), every job's BFGS keeps its own sequence.  16 jobs hold this
        # pair with 16 different guesses; job 0 carries the single call's guess and must return its bits.
        try:
            from staticmapping_amd import synth
            K = 16
            mb = sm.NdtGicpHip(max_source_points=len(src), max_target_points=len(tgt), jobs=K)
            for k in range(K):
                mb.set_input_source(src, slot=k); mb.set_input_target(tgt, slot=k)
            gs = [G @ synth.make_pose(t=(0.01 * (k % 8), -0.01 * (k % 5), 0.0), rpy_deg=(0, 0, 0.05 * (k % 7))) for k in range(K)]
            gs[0] = G

            def time_gbatch():
                mb.align_batch(K, gs)
                t_ = time.perf_counter()
                for _ in range(3):
                    Rb_, scb_, stb_ = mb.align_batch(K, gs)
                return (time.perf_counter() - t_) / 3, Rb_, stb_
            mb.set_target_cache(False)
            dtb, Rb, stb = time_gbatch()
            mb.set_target_cache(True)
            dtb_kept, Rb_kept, _ = time_gbatch()
            b_batch = sum(12.0 * (len(src) + len(tgt)) + (s_["n_source"] + s_["n_target"]) * (12.0 + 240.0) +
                          max(1, s_["gicp_iterations"]) * (s_["n_source"] * 20.0 + s_["gicp_correspondences"] * 84.0) for s_ in stb)
            out["ndt_gicp"]["batch16"] = {
                "value": round(K / dtb, 2), "unit": "alignments/s", "jobs": K, "ms_per_batch": round(dtb * 1e3, 3),
                "identical_to_single": bool(np.array_equal(Rb[0], R)),
                "gicp_iterations_min_max": [int(min(s_["gicp_iterations"] for s_ in stb)), int(max(s_["gicp_iterations"] for s_ in stb))],
                "function_evaluations_min_max": [int(min(s_["gicp_function_evaluations"] for s_ in stb)), int(max(s_["gicp_function_evaluations"] for s_ in stb))],
                "roofline": {"bound": "hbm", "achieved": round(b_batch / dtb / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(b_batch / dtb / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": b_batch},
                "target_kept": {"value": round(K / dtb_kept, 2), "ms_per_batch": round(dtb_kept * 1e3, 3), "identical_result": bool(np.array_equal(Rb, Rb_kept))},
                "note": "smhip_ndt_gicp_align_batch: lock-step jobs, one gicp_fdf launch per round of functor evaluations over all running jobs "
                        "(pclomp/gicp_omp_impl.hpp:381-514; builder/map_builder.cc:399-446, 655); value = everything rebuilt per Align"}
            mb.close()
        except Exception as e:
            out["ndt_gicp"]["batch16"] = {"error": repr(e)}
        recorded = os.path.join(ROOT, "profiles", "r02_gicp_cpu_baseline.json")
        if not (with_cpu and os.environ.get("SMHIP_BENCH_GICP_CPU", "1") == "1") and os.path.exists(recorded):
            # the numpy oracle needs ~30 s for this case on the GPU box: when it is switched off (SMHIP_BENCH_GICP_CPU=0 or
            # --no-cpu-baseline) the line carries the figure of the recorded run
            # (tools/evidence_bench.sh wrote the file)
            try:
                out["ndt_gicp"]["cpu_baseline"] = dict(json.load(open(recorded)), measured="recorded run, profiles/r02_gicp_cpu_baseline.json")
            except Exception:
                pass
        if with_cpu and os.environ.get("SMHIP_BENCH_GICP_CPU", "1") == "1":
            # the only CPU statement of this matcher is the numpy / scipy oracle (PCL is not vendored by the reference and no
            # C restatement of its GICP exists here): one whole Align of the same clouds, vectorised numpy on one core
            from oracle import ndt_gicp as og
            t = time.perf_counter()
            ds_o = og.approximate_voxel_grid_runs(src, 0.2)
            dt_o = og.approximate_voxel_grid_runs(tgt, 0.2)
            ref = og.ndt_gicp_align(src, tgt, guess=G, downsampled=(ds_o, dt_o))
            t_cpu = time.perf_counter() - t
            da, dtt = sm.se3_error(R, ref["result"])
            out["ndt_gicp"]["cpu_baseline"] = {"value": round(1.0 / t_cpu, 4), "unit": "alignments/s", "cores": 1, "kind": "port",
                                               "sample": "one whole Align of the same clouds and guess through oracle/ndt_gicp.py (numpy + scipy cKDTree, "
                                                         "single thread; stock PCL runs this matcher single-threaded too)"}
            out["ndt_gicp"]["parity"] = {"rot_vs_oracle_rad": da, "trans_vs_oracle_m": dtt, "ok_oracle": bool(ref["ok"]),
                                         "n_source_oracle": int(ref["n_source"]), "n_target_oracle": int(ref["n_target"]),
                                         "note": "a whole GICP run is comparable only to GICP's own repeatability (DESIGN_HISTORY.md section 9; on this case they do agree: tests/test_fullsize_gpu.py asserts 1e-4 rad / 1e-3 m)"}
        m.close()
    except Exception as e:
        out["ndt_gicp"] = {"error": repr(e)}
    # ---- the static-map output: MultiResolutionVoxelMap::InsertPointCloud per frame (builder/map_builder.cc:832-900)
    try:
        from staticmapping_amd import synth
        poses = synth.drive_poses(5, seed=5, speed=8.0)
        scene = synth.make_drive_scene(poses, seed=5)
        frames = []
        for k, P in enumerate(poses):
            sc = synth.velodyne_scan(synth.scene_near(scene, P[:3, 3]), P, seed=700 + k, n_points=N_POINTS, device=dev)
            w = (sc[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3]).astype(np.float32)
            frames.append((np.ascontiguousarray(np.concatenate([w, np.round(sc[:, 3:4] * 255), np.zeros((len(sc), 1), np.float32)], axis=1).astype(np.float32)),
                           P[:3, 3].astype(np.float32)))
        m = sm.MultiResolutionVoxelMapHip(table_log2=22, max_cloud_points=N_POINTS)
        m.insert_point_cloud(*frames[0])
        t = time.perf_counter()
        for f in frames[1:]:
            m.insert_point_cloud(*f)
        dt = (time.perf_counter() - t) / (len(frames) - 1)
        res = 0.1
        steps = float(np.mean([np.abs(np.floor(f[0][:, :3] / res) - np.floor(f[1] / res)).max(axis=1).sum() for f in frames[1:]]))
        entry = {"workload": "static-map output: MultiResolutionVoxelMap::InsertPointCloud of 120k-pt frames (0.1 m voxels, hit 0.55 / miss 0.48, "
                             "10 points per voxel), host rows in, map resident on the device",
                 "value": round(1.0 / dt, 2), "unit": "frames/s", "ms_per_frame": round(dt * 1e3, 3), "voxels": m.voxel_count(),
                 "ray_voxel_visits_per_frame": steps, "ray_voxel_visits_per_s": round(steps / dt, 1)}
        if with_cpu:
            from oracle import cref
            o = cref.Mrvm()
            o.insert(*frames[0])
            t = time.perf_counter()
            for f in frames[1:]:
                o.insert(*f)
            t_cpu = (time.perf_counter() - t) / (len(frames) - 1)
            kd, pd, md, nd, qd = m.dump()
            ko, po, mo, no, qo = o.dump()
            entry["parity"] = {"identical_map": bool(np.array_equal(kd, ko) and np.array_equal(pd, po) and np.array_equal(md, mo) and np.array_equal(nd, no) and np.array_equal(qd, qo)),
                               "voxels_oracle": int(len(ko))}
            entry["cpu_baseline"] = {"value": round(1.0 / t_cpu, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                     "sample": "the same frames through oracle/csrc/smref_mrvm.c: the reference's insert loop in point order, 1 thread "
                                               "(the reference's OpenMP form races on the probabilities and has no defined result)"}
            o.close()
        m.close()
        out["mrvm"] = entry
    except Exception as e:
        out["mrvm"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    main()
