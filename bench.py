#!/usr/bin/env python
"""Benchmark of the CrossLoc localisation hot path on MI355X (BASELINE.json metric: images/s localised,
480x720 frames, 256 RANSAC hypotheses; median pose error in cm / deg).

A step = one batch of synthetic frames through the whole path on one GPU:
    single-task scene-coordinate CNN forward (fp32 MFMA kernels)  ->  HIP dsacstar, 256 hypotheses per image.
No trained weights or datasets exist offline: the CNN runs seeded random weights on uniform-random images and
the solver consumes synthetic scene-coordinate maps (ray-cast terrain, 0.5 m noise, 30 % outliers) that are
resident in HBM before the timed region; both stages execute in full for every image of every step.
Multi-GPU (driver: torch.distributed.run, one rank per GPU): images shard as independent batches (weak
scaling, no data-path collective); one RCCL all-gather of the per-image errors for the median.

Prints ONE JSON line on rank 0 (contract in the task statement) including
    roofline     — the dominant kernel (3x3 512->512 implicit-GEMM conv, 78 % of forward FLOPs) timed with HIP
                   events on its launch stream inside the timed region, against the fp32 MFMA peak;
    cpu_baseline — the CPU oracle (restated reference path: OpenMP C solver + PyTorch-CPU network) on a bounded
                   sample, rank 0 at N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA = vector fp32 peak
# HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (profiles/r1_conv512_pmc.csv):
# FETCH_SIZE (KB, doubled per the gfx950 note of MI355X_MICROARCH.md §HBM) + WRITE_SIZE (KB), batch 24.
# PMC counters cannot be read from inside the process, so the profiled value is recorded per batch size.
CONV512_TRAFFIC_BYTES = {24: (755733 * 2 + 259200) * 1024}
# the batched Winograd GEMM launch of a 3x3 512->512 layer (profiles/r1_wino512_pmc.csv)
# by GEMMs per launch, then batch (profiles/r1_wino512_pmc.csv: F(4x4) batch 24, r1_wino512_b44_pmc.csv: F(4x4) batch 44,
# r1_wino512_f6_b44_pmc.csv: F(6x6) batch 44)
WINO512_TRAFFIC_BYTES = {16: {24: (762829 * 2 + 1036800) * 1024},
                         36: {24: (472038 * 2 + 596160) * 1024, 44: (832879 * 2 + 1092960) * 1024},
                         64: {44: (652157 * 2 + 844800) * 1024}}
WINO_NAME = {16: "F(2x2,3x3)", 36: "F(4x4,3x3)", 64: "F(6x6,3x3)"}
FWD_GFLOP_PER_IMAGE = 295.41        # SURVEY.md §8(d), single-task net, 480x720
FWD_GFLOP_PER_IMAGE_3ENC = 755.96   # SURVEY.md §8(d), CrossLoc 3-encoder net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None,
                    help="images per step per GPU.  44: the batched F(6x6,3x3) GEMM launch has 52 x 4 x 64 = 13312 "
                         "workgroups = exactly 26 rounds of the 512 resident ones, and the fixed per-launch costs of "
                         "the ~100 kernels of a forward are amortised over more frames (24: 997, 44: 1006 images/s); "
                         "47 is the per-launch maximum at 480x720 (32-bit byte offsets).  Default 44; 24 with --mlr 3 "
                         "(the Winograd buffers of the 1536-channel fusion layer stay below 2 GiB)")
    ap.add_argument("--hyps", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cnn-streams", type=int, default=1,
                    help="2: a step's CNN forward runs as two sub-batches on two HIP streams (the GEMMs of one overlap the "
                         "HBM-bound passes of the other: +9 %% images/s).  Not the default: kernel durations measured "
                         "while another stream shares the chip are not a roofline measurement")
    ap.add_argument("--mlr", type=int, default=0, choices=[0, 3],
                    help="3: BASELINE configs[4], the 3-encoder CrossLoc network (755.96 GFLOP per frame) instead of "
                         "the single-task one the headline metric is quoted on")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 24 if args.mlr else 44

    import torch
    from crossloc_amd import networks, synth, evaluation
    from crossloc_amd.weights import seeded_state_dict
    import dsacstar

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B, K, W, H, IMW, NH = args.batch, args.steps, args.warmup, 480, 720, args.hyps
    mean = torch.tensor(synth.SCENE_MEAN, dtype=torch.float32)
    net = networks.TransPoseNet(mean, False, False, 2, 2, 3, 1, num_mlr=args.mlr)   # utils/learning.py:302-305 sizes
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.to(dev).eval()

    g = torch.Generator(device="cpu").manual_seed(2021 + rank)
    images = torch.rand((B, 3, H, IMW), generator=g).to(dev)             # un-normalised [0,1] RGB at eval
    coords_np, _, poses_np = synth.make_batch(2021 + 1000 * rank, B, noise=0.5, outlier_ratio=0.3)
    coords = torch.from_numpy(coords_np).to(dev)
    gt_poses = torch.from_numpy(poses_np).to(dev)

    # pipeline: the CNN of a step runs as `cnn_streams` sub-batches on their own streams; the latency-bound solver of
    # step s runs on a side stream under the CNN of step s+1 (evaluation.PipelinedLocalizer)
    pipe = evaluation.PipelinedLocalizer(net, NH, synth.FOCAL, H, IMW, cnn_streams=args.cnn_streams)

    def step(s):
        image0 = (s * world + rank) * B                                   # global image index keys the sampler
        return pipe.submit(images, image0=image0, scene_coords=coords)

    for s in range(W):
        step(s)
    pipe.finish()
    torch.cuda.synchronize()

    L = networks._bind()
    L.xl_cnn_prof_begin.argtypes = [ctypes.c_int]
    L.xl_cnn_prof_end.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    n_sub = min(len(pipe.cnn), B)
    sub_b = [round((i + 1) * B / n_sub) - round(i * B / n_sub) for i in range(n_sub)]
    plan = net._plans[(sub_b[0], H, IMW, dev.index, False, 1)]           # sub-batch 0 (all sub-batches have the same ops)
    n_ops = len(plan.op_array)
    # events only around the launches the roofline object reports (the batched Winograd GEMMs; the conv ops when the
    # plan has none), every op with XL_BENCH_VERBOSE: ~100 extra event pairs per step cost 1.5 % of the step
    has_wino = any(op.type == 1 and op.nchunks2 > 1 for op in plan.ops)
    if os.environ.get("XL_BENCH_VERBOSE"):
        L.xl_cnn_prof_filter(-1, 0)
    else:
        L.xl_cnn_prof_filter(1, 2 if has_wino else 0)
    L.xl_cnn_prof_begin(n_ops * K * n_sub)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    all_poses = []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        ev[s][0].record(pipe.cnn[0])
        pred, done = pipe.forward_cnn(images)
        ev[s][1].record(pipe.cnn[0])                                     # stream 0 joins the others before the concat
        poses = torch.zeros((B, 4, 4), dtype=torch.float32, device=dev)
        for e in done:
            pipe.side.wait_event(e)                                      # solver(s) after CNN(s)
        pipe.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(pipe.side):
            ev[s][2].record()
            dsacstar.forward_rgb_batch(coords, poses, NH, 10.0, synth.FOCAL, IMW / 2.0, H / 2.0, 100.0, 100.0, 8,
                                       image0=(s * world + rank) * B)
            ev[s][3].record()
        pred.record_stream(pipe.side)
        all_poses.append(poses)
    pipe.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- per-kernel durations from the HIP events recorded inside the timed region
    cap = n_ops * K * n_sub
    idx = (ctypes.c_int32 * cap)()
    typ = (ctypes.c_int32 * cap)()
    ms = (ctypes.c_float * cap)()
    nrec = L.xl_cnn_prof_end(idx, typ, ms, cap)
    conv_ms, by_type, wino, wino_tiles = [], {}, 0, 0
    for i in range(max(nrec, 0)):
        op = plan.op_array[idx[i]]
        by_type[typ[i]] = by_type.get(typ[i], 0.0) + ms[i]
        if typ[i] == networks.XL_OP_CONV and op.Cin == 512 and op.Cout == 512 and op.stride == 1 and (
                (op.ksize == 3 and op.nchunks2 <= 1) or (op.ksize == 1 and op.nchunks2 > 1)):
            conv_ms.append(ms[i])
            wino = int(op.nchunks2) if op.nchunks2 > 1 else 0
            wino_tiles = op.Hi * op.Wi
    if rank == 0 and os.environ.get("XL_BENCH_VERBOSE"):
        per_op = {}
        for i in range(max(nrec, 0)):
            per_op.setdefault(idx[i], []).append(ms[i])
        names = {0: "conv1", 1: "conv", 2: "gn_stats", 3: "gn_apply", 4: "head", 11: "gn_final", 12: "wino_in",
                 13: "wino_out"}
        for i in sorted(per_op):
            op = plan.op_array[i]
            sys.stderr.write("op %3d %-8s k%d s%d %4d->%4d %3dx%3d  %.4f ms\n" % (
                i, names[op.type], op.ksize, op.stride, op.Cin, op.Cout, op.Hi, op.Wi, float(np.mean(per_op[i]))))
        sys.stderr.write("by type (ms/step): %s\n" % {names[k]: round(v / K, 3) for k, v in by_type.items()})
    conv_avg_ms = float(np.mean(conv_ms)) if conv_ms else float("nan")
    # dominant kernel: the 3x3 512->512 layers.  Direct form: one implicit GEMM of 2*M*512*4608 FLOP.  Winograd
    # F(2x2,3x3) form (inference plans): one batched launch of 16 GEMMs [M/4 x 512] x [512 x 512]; the FLOPs counted
    # are the ones that launch executes (2.25x fewer multiplies than the direct form for the same layer)
    Bl = sub_b[0]                                    # frames per launch (one sub-batch)
    conv_flop = wino * 2.0 * (Bl * wino_tiles) * 512 * 512 if wino else 2.0 * (Bl * 60 * 90) * 512 * (9 * 512)
    conv_tflops = conv_flop / (conv_avg_ms * 1e-3) / 1e12 if conv_ms else float("nan")
    cnn_ms = float(np.mean([ev[s][0].elapsed_time(ev[s][1]) for s in range(K)]))
    dsac_ms = float(np.mean([ev[s][2].elapsed_time(ev[s][3]) for s in range(K)]))

    # ---- pose errors: every image of every step, gathered over ranks with one all-gather
    est = torch.cat(all_poses, 0)
    t_err, r_err = evaluation.pose_errors(gt_poses.repeat(K, 1, 1), est)
    local = torch.stack([t_err, r_err], 1)
    total_imgs = world * B * K
    if world > 1:
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        allerr = torch.cat(gathered, 0)
    else:
        allerr = local
    med_t_cm = float(torch.median(allerr[:, 0]).item() * 100.0)
    med_r_deg = float(torch.median(allerr[:, 1]).item())

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(net, images, coords_np, NH, args.mlr)

    if rank == 0:
        value = total_imgs / elapsed
        out = {
            "metric": "images/sec localized (480x720, 256 hyps)", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: CrossLoc 3-encoder (coord+depth+normal) fusion network forward"
                                    if args.mlr else
                                    "BASELINE configs[2]: single-task coord CNN forward (2+2 extra res blocks)")
                                   + " + HIP dsacstar.forward_rgb, 480x720 frames, 60x90 coordinate grid",
                       "hypotheses": NH, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "images sharded over %d GPU(s), no data-path collective" % world,
                       "solver_input": "synthetic scene coordinates (0.5 m noise, 30% outliers); CNN runs seeded "
                                       "random weights on random images (no trained weights offline)",
                       "cnn_ms_per_batch": round(cnn_ms, 3), "dsac_ms_per_batch": round(dsac_ms, 3),
                       "pipeline": "CNN of a step as %d sub-batches on %d streams; solver(s) on a side stream under "
                                   "CNN(s+1), ordered by events" % (n_sub, n_sub),
                       # north_star asks for the RANSAC stage against the HBM roofline as well: algorithmic bytes =
                       # nHyp*N*12 B + 64 B per image; the stage is LDS-resident and fp64/latency-bound
                       "dsac_algorithmic_GBps": round(B * (NH * 5400 * 12 + 64) / (dsac_ms * 1e-3) / 1e9, 1),
                       "dsac_hbm_roofline_frac": round(B * (NH * 5400 * 12 + 64) / (dsac_ms * 1e-3) / 8e12, 5),
                       # direct-convolution FLOP count of the network over the CNN time; with the Winograd
                       # layers fewer multiplies are executed, so this "algorithmic" rate may exceed the MFMA peak
                       "cnn_fwd_algorithmic_tflops": round((FWD_GFLOP_PER_IMAGE_3ENC if args.mlr else FWD_GFLOP_PER_IMAGE) * B / cnn_ms, 2),
                       "conv3x3_s1_algorithm": ("winograd " + WINO_NAME.get(wino, "?")) if wino else "direct implicit GEMM",
                       "median_err_cm": round(med_t_cm, 3), "median_err_deg": round(med_r_deg, 5)},
            "roofline": {"bound": "mfma",
                         "kernel": (("igemm_conv_kernel<1,1,128,512,0,128,1> batched x%d: the Winograd %s GEMMs of a 3x3 "
                                     "512->512 layer @60x90" % (wino, WINO_NAME.get(wino, "?")) if wino else
                                     "igemm_conv_kernel<3,1,128,512> (3x3 512->512 @60x90") + " x%d images per launch)" % Bl),
                         "achieved": round(conv_tflops, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(conv_tflops / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": (WINO512_TRAFFIC_BYTES.get(wino, {}) if wino else CONV512_TRAFFIC_BYTES).get(Bl),
                         "algorithmic_bytes_per_launch": (wino * (2 * Bl * wino_tiles * 512 + 512 * 512) * 4 if wino else
                                                          2 * Bl * 5400 * 512 * 4 + 512 * 4608 * 4),
                         "avg_launch_ms": round(conv_avg_ms, 4), "launches_timed": len(conv_ms),
                         "algorithmic_gflop_per_launch": round(conv_flop / 1e9, 2)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(net, images, coords_np, n_hyp, num_mlr=0):
    """The restated reference CPU path on this box's host cores: PyTorch-CPU fp32 network + OpenMP C solver.
    Bounded sample (a few frames) so the default run stays within minutes."""
    import torch
    from oracle import cnn_oracle, dsac_oracle
    dsac_oracle.build()
    # all hardware threads of a 2-socket host oversubscribe both OpenMP runtimes badly (measured 22 s per
    # frame at 256 threads); use up to 64 and report that number as `cores`
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
    torch.set_num_threads(cores)
    dsac_oracle.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    x = images[:1].detach().cpu()
    cnn_oracle.transposenet_forward(sd, x, num_mlr, 2, 2)                # warm-up
    n_cnn = 3
    t0 = time.perf_counter()
    for _ in range(n_cnn):
        cnn_oracle.transposenet_forward(sd, x, num_mlr, 2, 2)
    t_cnn = (time.perf_counter() - t0) / n_cnn
    n_dsac = min(16, coords_np.shape[0])
    dsac_oracle.forward_rgb(coords_np[0], n_hyp, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8)   # warm-up
    t0 = time.perf_counter()
    for b in range(n_dsac):
        dsac_oracle.forward_rgb(coords_np[b], n_hyp, 10.0, 480.0, 360.0, 240.0, 100.0, 100.0, 8, image=b)
    t_dsac = (time.perf_counter() - t0) / n_dsac
    return {"value": round(1.0 / (t_cnn + t_dsac), 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d frames CNN forward (PyTorch CPU fp32, batch 1) + %d frames oracle dsacstar %d hyps "
                      "(C/OpenMP, %d threads); reference binary unbuildable (needs OpenCV)" % (
                          n_cnn, n_dsac, n_hyp, dsac_oracle.num_threads()),
            "cnn_s_per_image": round(t_cnn, 4), "dsac_s_per_image": round(t_dsac, 5)}


if __name__ == "__main__":
    main()
