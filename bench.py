#!/usr/bin/env python3
"""Benchmark of the MI355X hot path (matching + tracking; triangulation and float
matching as separate workloads).

  python bench.py --gpus N --steps K --warmup W [--workload framepair|c3|c4|c5]

N > 1 runs one rank per GPU over RCCL: either the caller launches this file under
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or, when
those are absent, this process starts `python -m torch.distributed.run` itself as a
child -- before anything here touches the GPU -- waits for it and forwards rank 0's
line.  Rank 0 prints ONE JSON line.  The default workload is the one BASELINE.json's metric is
quoted on: per step every rank processes PAIRS frame pairs, each = robust
matching of two 8192-keypoint ORB frames (2-NN both ways, ratio test,
cross-check) + pyramidal LK of 5000 points on a 1280x720 pair (21x21 window,
maxLevel 3); with N > 1 the ranks then all-gather their match lists.  Inputs
are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "matched+tracked features/sec (ms/frame pair @8k kps) at 1/2/4/8 MI355X"
VALU_PEAK_TLANEOPS = 78.6   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
FP4_MFMA_PEAK_TOPS = 10066.0  # dense 4-bit: 256 CU x 4 SIMD x 4096 ops/clk (32x32x64 per 32 clk) x 2.4 GHz


def hamming_roofline(prof, n_directions, rows, desc_bytes, launches_note):
    """Roofline entry of whichever Hamming kernel ran.  n_directions: 2-NN problems of
    rows x rows per launch.  The algorithmic work is 2 * ceil(D/4) lane-ops per descriptor
    pair (xor + popcount per 32-bit word); the matrix-core kernel executes it as a contraction
    of 8 D multiply-adds of 4-bit operands per pair and is priced against the dense 4-bit MFMA
    peak, with the lane-op rate of the same launch beside it."""
    words = (desc_bytes + 3) // 4
    lane_ops = float(n_directions) * rows * rows * 2 * words
    for name in ("hamming_mfma_kernel", "hamming_pair_kernel", "hamming_knn2_kernel"):
        if name in prof and prof[name][1]:
            break
    else:
        return {"kernel": None, "bound": "valu", "achieved": None, "peak": VALU_PEAK_TLANEOPS,
                "unit": "Tlane-op/s", "frac": None, "traffic": None}
    ms, n = prof[name]
    sec = ms / n * 1e-3
    lane_rate = lane_ops / sec / 1e12
    if name == "hamming_mfma_kernel":
        mac_ops = float(n_directions) * rows * rows * (32 * words) * 2
        ach = mac_ops / sec / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(ach, 1), "peak": FP4_MFMA_PEAK_TOPS,
                "unit": "TOP/s (4-bit operands, multiply-add = 2)",
                "frac": round(ach / FP4_MFMA_PEAK_TOPS, 4),
                "avg_launch_us": round(1e6 * sec, 2),
                "algorithmic_per_launch": f"{mac_ops:.4g} 4-bit ops = {launches_note} x {rows}^2 x "
                                          f"{32 * words} bits x 2 (the xor/popcount form of the same "
                                          f"launch is {lane_ops:.4g} lane-ops)",
                "valu_equivalent": {"achieved": round(lane_rate, 2), "peak": VALU_PEAK_TLANEOPS,
                                    "unit": "Tlane-op/s",
                                    "frac": round(lane_rate / VALU_PEAK_TLANEOPS, 4)},
                "traffic": None}
    return {"kernel": name, "bound": "valu", "achieved": round(lane_rate, 3),
            "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
            "frac": round(lane_rate / VALU_PEAK_TLANEOPS, 4), "avg_launch_us": round(1e6 * sec, 2),
            "algorithmic_per_launch": f"{lane_ops:.4g} lane-ops = {launches_note} x {rows}^2 x "
                                      f"{2 * words} (xor + popcount per 32-bit word and direction)",
            "traffic": None}
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0  # bf16 and fp16, dense, no sparsity

KPS = 8192
DESC_BYTES = 32
LK_POINTS = 5000
DEFAULT_PAIRS = 64   # frame pairs per rank and step (the tracking configuration is batched by 64)
LK_W, LK_H = 1280, 720
LK_WIN, LK_MAXLEVEL, LK_COUNT, LK_EPS = (21, 21), 3, 30, 0.01


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="framepair", choices=["framepair", "c3", "c4", "c5", "f1", "f2", "f3", "vo", "ranks"],
                    help="'ranks' only reports the ranks a multi-process launch created (gloo, no GPU)")
    ap.add_argument("--pairs", type=int, default=None,
                    help="frame pairs (problems, frames) per rank and step; default 64 for the "
                         "frame-pair workload, 32 for f1 / f2 / f3 / vo")
    ap.add_argument("--frames", type=int, default=512, help="c5: frames in the window")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain", type=float, default=2.0,
                    help="framepair: seconds the step is looped after the timed region for the "
                         "'sustained' rate (0: off)")
    ap.add_argument("--no-sequence", action="store_true",
                    help="framepair: skip the consecutive-frames figure (pyramids built once per frame)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="framepair: skip the short runs of the other configurations "
                         "(triangulation, float matching, sharded window) reported under 'other_configs'")
    ap.add_argument("--one-stream", action="store_true",
                    help="framepair: matching and tracking on ONE HIP stream.  By default tracking "
                         "goes to a second stream and runs beside matching (the tracker is bound by "
                         "vector issue, the Hamming search by the matrix pipe); the per-kernel times "
                         "behind 'roofline' and 'kernels' then come from a separate one-stream pass "
                         "after the timed region, so that they are not stretched by the overlap")
    ap.add_argument("--force-gather", action="store_true",
                    help="framepair / c5 with one rank: run the RCCL all-gather of the match lists anyway "
                         "(a one-rank process group), to exercise that path on a single GPU")
    return ap.parse_args()


def dist_setup(n_gpus, force_group=False):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or force_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # SFM_BENCH_REHEARSE=1: the multi-rank code on fewer GPUs than ranks (ranks share the
        # cards round-robin and the collectives go through gloo; RCCL refuses two ranks on one
        # device).  It checks the N > 1 control flow on a one-GPU box; its timings mean nothing.
        rehearse = os.environ.get("SFM_BENCH_REHEARSE", "0") == "1"
        if rehearse:
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is for
        # the one JSON line, so the banner goes to stderr
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            warm = torch.zeros(1, device=torch.device("cuda", local))
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    else:
        torch.cuda.set_device(0)
        local = 0
    if n_gpus != world and rank == 0:
        print(f"warning: --gpus {n_gpus} but WORLD_SIZE {world}", file=sys.stderr)
    return rank, world, local


def barrier_sync(world):
    import torch
    import torch.distributed as dist

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(seconds, world, dev):
    import torch
    import torch.distributed as dist

    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_steps(step, steps, warmup, world, dev, ctx, more_ctx=(), profile_step=None, profile_ctxs=None):
    """W untimed steps, then K timed ones between barrier + device-wide synchronize (all
    streams), with the library's per-kernel timing off: its HIP events around every launch
    would sit in the timed streams.  The per-kernel times come from a second pass of K steps
    after the timed region (`profile_step`, default the same step) with the events on."""
    # ranks that share one card (rehearsal) take seconds per step: say so while they run
    rehearsal = bool(os.environ.get("SFM_BENCH_REHEARSE"))
    for _ in range(warmup):
        step()
    barrier_sync(world)
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        if rehearsal:
            import torch

            torch.cuda.synchronize()
            print(f"rehearsal: step {i + 1} of {steps}", file=sys.stderr, flush=True)
    barrier_sync(world)
    dt = time.perf_counter() - t0
    ctxs = tuple(profile_ctxs) if profile_ctxs is not None else (ctx,) + tuple(more_ctx)
    for c in ctxs:
        c.set_profiling(True)
    pstep = profile_step or step
    for _ in range(steps):
        pstep()
    barrier_sync(world)
    prof = {}
    for c in ctxs:
        prof.update(c.profile())
        c.set_profiling(False)
    return max_over_ranks(dt, world, dev), prof


LK_ITERS_PER_POINT = 12.9   # mean Lucas-Kanade iterations per point summed over the 4 levels
                            # of the synthetic set (the CPU leg measures it; the GPU runs the
                            # same iterations: positions and status are identical)


def lk_roofline(prof, n_pairs, iters_per_point):
    """Roofline entry of the tracker: vector-ALU bound.  Algorithmic work per point and level =
    6615 + k * 2646 multiply-adds for a 21 x 21 window (441 px x (3 bilinear sums of 4 taps + 3
    products) to set up I, dI/dx, dI/dy and the normal matrix; 441 x (4 + 2) per iteration),
    counted as lane-ops against the vector peak.  k comes from the CPU leg of this run when it
    ran, else from the constant above."""
    name = "lk_track_kernel"
    if name not in prof or not prof[name][1]:
        return {"kernel": None, "bound": "valu", "achieved": None, "peak": VALU_PEAK_TLANEOPS,
                "unit": "Tlane-op/s", "frac": None, "traffic": None}
    k = iters_per_point if iters_per_point else LK_ITERS_PER_POINT
    levels = LK_MAXLEVEL + 1
    macs = float(n_pairs) * LK_POINTS * (levels * 6615 + k * 2646)
    ms, n = prof[name]
    sec = ms / n * 1e-3
    ach = macs / sec / 1e12
    compulsory = n_pairs * (2 * LK_W * LK_H + 21 * LK_POINTS)
    return {"kernel": name, "bound": "valu", "achieved": round(ach, 3), "peak": VALU_PEAK_TLANEOPS,
            "unit": "Tlane-op/s", "frac": round(ach / VALU_PEAK_TLANEOPS, 4),
            "avg_launch_us": round(1e6 * sec, 2),
            "algorithmic_per_launch": f"{macs:.4g} multiply-adds = {n_pairs} pairs x {LK_POINTS} points x "
                                      f"({levels} levels x 6615 + {k:.2f} iterations x 2646)",
            "iterations_per_point": round(k, 3),
            "iterations_source": "cpu_baseline leg of this run" if iters_per_point else "constant",
            "compulsory_hbm_bytes": compulsory,
            "compulsory_hbm_GBps": round(compulsory / sec / 1e9, 1),
            "second_limit": "vector issue: about 12 250 vector instructions per wave of eight points "
                            "(SQ_INSTS_VALU, tools/pmc_sq.sh; 7100 per wave of four with sixteen lanes per "
                            "point, SFM_LK_LANES=16), most of them byte shuffles and 16-bit dot products that "
                            "the multiply-add count does not see; the SIMDs issue one of them in about 85 % "
                            "of their cycles",
            "traffic": None}


def pmc_traffic(workload, kernels):
    """HBM bytes per launch of the first of `kernels` found in the newest committed
    profiles/rNN_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this
    same workload (tools/profile_round.sh), (2 x FETCH_SIZE + WRITE_SIZE) x 1024 with the gfx950
    half-count correction of FETCH_SIZE.  None when no such record exists."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        rec = json.load(f).get(workload, {})
    for k in kernels:
        if k in rec and rec[k].get("hbm_bytes") is not None:
            return rec[k]["hbm_bytes"], os.path.relpath(files[-1], ROOT)
    return None, None


def kernel_line(prof, name):
    ms, n = prof.get(name, (0.0, 0))
    return {"total_ms": round(ms, 4), "launches": n, "avg_us": round(1e3 * ms / n, 3) if n else None}


# --------------------------------------------------------------------------- framepair
def lk_inputs(pairs, seed):
    """`pairs` distinct 1280x720 image pairs (4 textures x different warps) and point sets."""
    from sfm_amd import synth

    n_tex = min(4, pairs)
    rng = np.random.default_rng(seed)
    prevs, currs, pts = [], [], []
    texes = [synth.texture(LK_H, LK_W, seed=seed + 17 * i) for i in range(n_tex)]
    for p in range(pairs):
        tex = texes[p % n_tex]
        shift = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)))
        ang = float(rng.uniform(-0.5, 0.5))
        prevs.append(synth.to_u8(tex))
        currs.append(synth.to_u8(synth.affine_warp(tex, shift, ang)))
        pts.append(synth.grid_points(LK_H, LK_W, LK_POINTS, seed=seed + p, margin=15))
    return np.stack(prevs), np.stack(currs), np.stack(pts)


def lk_sequence_frames(n_frames, seed, dev):
    """n_frames consecutive 1280x720 frames of one textured scene under a slow camera motion
    (0.9, -0.6 px and 0.02 degrees per frame), sampled bilinearly from the scene on the device."""
    import torch

    from sfm_amd import synth

    tex = torch.from_numpy(synth.texture(LK_H, LK_W, seed=seed)).to(dev, torch.float32)
    ys, xs = torch.meshgrid(torch.arange(LK_H, device=dev, dtype=torch.float32),
                            torch.arange(LK_W, device=dev, dtype=torch.float32), indexing="ij")
    cx, cy = (LK_W - 1) / 2.0, (LK_H - 1) / 2.0
    out = torch.empty((n_frames, LK_H, LK_W), dtype=torch.uint8, device=dev)
    for k in range(n_frames):
        a = np.deg2rad(0.02 * k)
        ca, sa = float(np.cos(a)), float(np.sin(a))
        dx, dy = xs - cx - 0.9 * k, ys - cy + 0.6 * k
        sx, sy = ca * dx + sa * dy + cx, -sa * dx + ca * dy + cy
        x0 = sx.floor().clamp(0, LK_W - 2)
        y0 = sy.floor().clamp(0, LK_H - 2)
        fx, fy = (sx - x0).clamp(0, 1), (sy - y0).clamp(0, 1)
        x0, y0 = x0.long(), y0.long()
        v = (tex[y0, x0] * (1 - fx) * (1 - fy) + tex[y0, x0 + 1] * fx * (1 - fy) +
             tex[y0 + 1, x0] * (1 - fx) * fy + tex[y0 + 1, x0 + 1] * fx * fy)
        out[k] = (v * 255.0 + 0.5).clamp(0, 255).to(torch.uint8)
    return out


def cpu_baseline_framepair(desc, prevs, currs, pts):
    """The CPU oracle on the same workload, bounded to ~10-20 s: all host threads OpenMP gives
    this process (OMP_NUM_THREADS), float accumulators (OpenCV's accumulator type)."""
    from oracle import pyoracle

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    pyoracle.robust_matches(pyoracle.NORM_HAMMING, desc[0][:512], desc[1][:512], 0.5)  # warm-up
    done, t0, lk_iters = 0, time.perf_counter(), 0
    while True:
        i = done % (len(desc) - 1)
        pyoracle.robust_matches(pyoracle.NORM_HAMMING, desc[i], desc[i + 1], 0.5, threads=0)
        j = done % len(prevs)
        lk_iters += pyoracle.pyrlk(prevs[j], currs[j], pts[j], win=LK_WIN, max_level=LK_MAXLEVEL,
                                   crit_type=3, max_count=LK_COUNT, epsilon=LK_EPS, acc_exact=False,
                                   threads=0, return_iters=True)[-1]
        done += 1
        el = time.perf_counter() - t0
        if (el > 10.0 and done >= 2) or el > 25.0:
            break
    feats = done * (2 * KPS + LK_POINTS)
    return {"value": round(feats / el, 1), "unit": "features/s", "cores": threads, "kind": "port",
            "sample": f"{done} frame pairs (robust matching 2x{KPS}x{KPS}x{DESC_BYTES}B + LK "
                      f"{LK_POINTS} pts {LK_W}x{LK_H}) in {el:.1f} s, restatement not OpenCV",
            "ms_per_frame_pair": round(1e3 * el / done, 2),
            "lk_iterations_per_point": round(lk_iters / (done * LK_POINTS), 3)}


def run_framepair(args, rank, world, local):
    import torch

    from sfm_amd import allpairs, capi, synth

    dev = torch.device("cuda", local)
    P = (args.pairs or DEFAULT_PAIRS)
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    # Matching and tracking of a frame pair are independent: the tracking kernels go to a
    # second stream (a context of its own) and run concurrently with matching.
    two_streams = not args.one_stream
    if not two_streams:
        lk_stream, ctx_lk = None, ctx
    else:
        lk_stream = torch.cuda.Stream(device=dev)
        ctx_lk = capi.Context(local, stream=lk_stream.cuda_stream)
    # matching inputs: a chain of P+1 frames, pairs (i, i+1); own chain per rank (weak scaling)
    desc = synth.descriptor_chain(P + 1, KPS, DESC_BYTES, seed=1234 + rank)
    pairs = np.stack([np.arange(P), np.arange(P) + 1], 1).astype(np.int32)
    d_desc = torch.from_numpy(desc).to(dev)
    d_pairs = torch.from_numpy(pairs).to(dev)
    cap = KPS
    d_matches = torch.empty((P, cap, 4), dtype=torch.int32, device=dev)
    d_mcount = torch.empty(P, dtype=torch.int32, device=dev)
    # tracking inputs
    prevs, currs, pts = lk_inputs(P, seed=42 + 1000 * rank)
    d_prev = torch.from_numpy(prevs).to(dev)
    d_curr = torch.from_numpy(currs).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    d_out = torch.empty((P, LK_POINTS, 2), dtype=torch.float32, device=dev)
    d_status = torch.empty((P, LK_POINTS), dtype=torch.uint8, device=dev)
    d_err = torch.empty((P, LK_POINTS), dtype=torch.float32, device=dev)
    ctx.reserve(max(capi.Context.match_pairs_workspace(0, P + 1, KPS, DESC_BYTES, P),
                    capi.Context.pyrlk_workspace(P, LK_W, LK_H, LK_WIN, LK_MAXLEVEL)))
    if two_streams:
        ctx_lk.reserve(capi.Context.pyrlk_workspace(P, LK_W, LK_H, LK_WIN, LK_MAXLEVEL))
    torch.cuda.synchronize()

    # N > 1: every rank matches its own pairs, then all ranks exchange their match lists
    # (fixed-capacity buffers + counts, no host round trip); the collective is asynchronous
    # and overlaps the tracking kernels of the same step.
    # The exchanged lists are the compact 6-byte records (a third of the bytes of the DMatch
    # buffers); the matcher then writes them back to back, still without a host round trip.
    compact = os.environ.get('SFM_BENCH_GATHER', 'compact') == 'compact'
    gather = allpairs.PaddedGather(P, cap, world, dev, compact=compact) if world > 1 or args.force_gather else None
    gathered = {}
    if gather is not None:
        d_compact = torch.empty((P * cap, 3), dtype=torch.int16, device=dev)
        d_offsets = torch.empty(P + 1, dtype=torch.int64, device=dev)
        ctx.reserve(capi.Context.match_pairs_compact_workspace(0, P + 1, KPS, DESC_BYTES, P))
        torch.cuda.synchronize()

    def make_step(track_ctx):
        def step():
            if gather is None or not compact:
                ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, P + 1, KPS * DESC_BYTES, KPS, None,
                                    DESC_BYTES, d_pairs, P, 0.5, d_matches, cap, d_mcount)
                if gather is not None:
                    gather.start(d_matches, d_mcount)
            else:
                ctx.match_pairs_compact_dev(capi.NORM_HAMMING, d_desc, P + 1, KPS * DESC_BYTES, KPS, None,
                                            DESC_BYTES, d_pairs, P, 0.5, d_compact, P * cap, d_mcount, d_offsets)
                gather.start(d_compact, d_mcount)
            track_ctx.pyrlk_batch_dev(d_prev, d_curr, P, LK_W * LK_H, LK_W, LK_H, LK_W, d_pts, LK_POINTS,
                                      LK_WIN, LK_MAXLEVEL, 3, LK_COUNT, LK_EPS, 1e-4, 10.0, d_out,
                                      d_status, d_err)
            if gather is not None:
                gather.wait()
        return step

    step = make_step(ctx_lk)
    # kernel durations without the overlap: the same step on one stream, after the timed
    # region (not part of `value`)
    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx,
                           profile_step=make_step(ctx), profile_ctxs=(ctx,))
    feats_step = world * P * (2 * KPS + LK_POINTS)
    value = feats_step * args.steps / dt
    # the same step looped for at least --sustain seconds: a rate at the clock the chip holds
    # under continuous load, beside the short timed region's
    sustained = None
    if args.sustain > 0:
        barrier_sync(world)
        t0, n_sus = time.perf_counter(), 0
        while True:
            for _ in range(10):
                step()
            n_sus += 10
            torch.cuda.synchronize()
            go_on = torch.tensor([1.0 if time.perf_counter() - t0 < args.sustain else 0.0], device=dev)
            if world > 1:   # every rank leaves the loop after the same number of steps
                torch.distributed.all_reduce(go_on, op=torch.distributed.ReduceOp.MIN)
            if go_on.item() == 0.0:
                break
        barrier_sync(world)
        dt_sus = max_over_ranks(time.perf_counter() - t0, world, dev)
        sustained = {"value": round(feats_step * n_sus / dt_sus, 1), "unit": "features/s", "steps": n_sus,
                     "seconds": round(dt_sus, 3), "ms_per_step": round(1e3 * dt_sus / n_sus, 4)}
    matched = int(d_mcount.sum().item())
    if gather is not None:
        gathered["n"] = int(gather.counts.sum().item())
    tracked = int(d_status.sum().item())

    # The reference's own use: frame k -> k+1 on every iteration (app_solver.cpp:68, :366).  P + 1
    # consecutive frames, every frame's pyramid built once and used as the next image of one pair
    # and the previous image of the following one.  A separate figure; `value` stays the
    # independent pairs above.
    sequence = None
    if not args.no_sequence:
        d_frames = lk_sequence_frames(P + 1, seed=4242 + rank, dev=dev)
        pyr_bytes = capi.Context.lk_pyramid_bytes(LK_W, LK_H, LK_WIN, LK_MAXLEVEL)
        d_pyr = torch.empty((P + 1) * pyr_bytes, dtype=torch.uint8, device=dev)
        ctx_lk.reserve(capi.Context.pyrlk_pyramids_workspace(P, LK_W, LK_H, LK_WIN, LK_MAXLEVEL))
        d_seq_status = torch.empty((P, LK_POINTS), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def seq_step():
            ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, P + 1, KPS * DESC_BYTES, KPS, None,
                                DESC_BYTES, d_pairs, P, 0.5, d_matches, cap, d_mcount)
            ctx_lk.lk_build_pyramids_dev(d_frames, P + 1, LK_W * LK_H, LK_W, LK_H, LK_W, LK_WIN, LK_MAXLEVEL, d_pyr)
            ctx_lk.pyrlk_pyramids_dev(d_pyr, P + 1, d_pairs, P, LK_W, LK_H, d_pts, LK_POINTS, LK_WIN, LK_MAXLEVEL,
                                      3, LK_COUNT, LK_EPS, 1e-4, 10.0, d_out, d_seq_status, d_err)

        for _ in range(args.warmup):
            seq_step()
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            seq_step()
        barrier_sync(world)
        dt_seq = max_over_ranks(time.perf_counter() - t0, world, dev)
        sequence = {"value": round(feats_step * args.steps / dt_seq, 1), "unit": "features/s",
                    "ms_per_step": round(1e3 * dt_seq / args.steps, 4),
                    "workload": f"{P + 1} consecutive frames, {P} pairs (k, k+1): matching as above; "
                                f"{P + 1} pyramids built once each (sfm_lk_build_pyramids_dev) instead of "
                                f"{2 * P}, tracking by pair list (sfm_pyrlk_pyramids_dev)",
                    "tracked_per_step": int(d_seq_status.sum().item())}
        del d_frames, d_pyr

    # roofline of the dominant kernel (by summed device time over the timed region)
    dominant = max(prof.items(), key=lambda kv: kv[1][0])[0]
    lk_ms, lk_n = prof.get("lk_track_kernel", (0.0, 0))
    roof_ham = hamming_roofline(prof, 2 * P, KPS, DESC_BYTES, f"2 dirs x {P} pairs")
    if P == DEFAULT_PAIRS and roof_ham["kernel"]:  # the PMC passes were taken at the default pairs per launch
        roof_ham["traffic"], roof_ham["traffic_source"] = pmc_traffic("framepair",
                                                                      [roof_ham["kernel"]])
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_framepair(desc, prevs, currs, pts)
    roof_lk = lk_roofline(prof, P, cpu["lk_iterations_per_point"] if cpu else None)
    if P == DEFAULT_PAIRS and roof_lk["kernel"]:
        roof_lk["traffic"], roof_lk["traffic_source"] = pmc_traffic("framepair", ["lk_track_kernel"])
    # `roofline` describes the kernel with the largest summed device time of the step
    roofs = {roof_ham["kernel"]: roof_ham, roof_lk["kernel"]: roof_lk}
    roof_main = roofs.get(dominant, roof_ham)
    roof_other = roof_lk if roof_main is roof_ham else roof_ham
    out = {
        "metric": METRIC, "value": round(value, 1), "unit": "features/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"framepair: {P} frame pairs/rank/step; robust matching ORB "
                               f"{KPS} kps x {DESC_BYTES} B (2-NN both ways, ratio 0.5, "
                               f"cross-check) + pyrLK {LK_POINTS} pts {LK_W}x{LK_H} win 21x21 "
                               f"maxLevel {LK_MAXLEVEL} (levels 0..3) crit COUNT+EPS(30,0.01)",
                   "pairs_per_rank_step": P, "sharding": "by frame pair, all-gather of match lists"
                   if world > 1 else "single GPU",
                   "streams": "matching and tracking on two HIP streams; kernel times and roofline "
                              "from a one-stream pass of the same step after the timed region"
                   if two_streams else "one"},
        "ms_per_frame_pair": round(1e3 * dt / args.steps / P, 4),
        "dominant_kernel": dominant,
        "roofline": roof_main,
        "roofline_second_kernel": roof_other,
        "kernels": {k: kernel_line(prof, k) for k in sorted(prof)},
        "kernel_sum_ms_per_step": round(sum(ms for ms, _ in prof.values()) / args.steps, 4),
        "check": {"cross_checked_matches_per_step": matched, "tracked_per_step": tracked,
                  "gathered_matches": gathered.get("n")},
    }
    if sustained is not None:
        out["sustained"] = sustained
    if sequence is not None:
        out["sequence"] = sequence
    if lk_n:
        out["lk_track"] = {"avg_launch_us": round(1e3 * lk_ms / lk_n, 2),
                           "points_per_s": round(P * LK_POINTS / (lk_ms / lk_n * 1e-3), 1)}
        # the whole tracking stage per step (padding, pyramids, a derivative image where the
        # window size still has one, the tracker): the tracker of 21 x 21 windows forms the
        # derivatives itself, so its own time carries work that was another kernel's before
        stage_ms = sum(prof[k][0] for k in prof if k.startswith("lk_")) / args.steps
        macs = roof_lk.get("achieved", 0) and roof_lk["achieved"] * 1e12 * (lk_ms / lk_n * 1e-3)
        out["lk_stage"] = {"ms_per_step": round(stage_ms, 4),
                           "kernels": sorted(k for k in prof if k.startswith("lk_")),
                           "achieved": round(macs / (stage_ms * 1e-3) / 1e12, 3) if macs else None,
                           "unit": "Tlane-op/s over the stage (same multiply-add count as 'roofline')",
                           "frac_of_valu_peak": round(macs / (stage_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS, 4) if macs else None}
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    if ctx_lk is not ctx:
        ctx_lk.close()
    ctx.close()
    return out


# --------------------------------------------------------------------------- c3 triangulation
def run_c3(args, rank, world, local):
    import torch

    from sfm_amd import capi, synth

    dev = torch.device("cuda", local)
    n = 10_000_000
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    s = synth.triangulation_set(n, seed=7 + rank)
    d_p1 = torch.from_numpy(s["p1"]).to(dev)
    d_p2 = torch.from_numpy(s["p2"]).to(dev)
    d_X = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_mask = torch.empty(n, dtype=torch.uint8, device=dev)

    def step():
        ctx.triangulate_dev(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], d_p1, d_p2, n,
                            None, 0, 0, 0, 1e-4, 250.0, 3.0, d_X, None, d_mask, None)

    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx)
    ms, cnt = prof["triangulate_dlt_kernel"]
    avg_s = ms / cnt * 1e-3
    out = {
        "metric": "triangulated correspondences/sec (BASELINE config 3)",
        "value": round(world * n * args.steps / dt, 1), "unit": "correspondences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "c3: DLT triangulation + reprojection filter, 10M correspondences"},
        # the library's timer label "triangulate_dlt_kernel" covers whichever first-pass kernel ran:
        # triangulate_invit_kernel unless SFM_TRIANGULATE_JACOBI=1 selects the Jacobi kernel
        "roofline": {"kernel": ("triangulate_dlt_kernel" if os.environ.get("SFM_TRIANGULATE_JACOBI", "")[:1] == "1"
                                else "triangulate_invit_kernel"), "bound": "hbm",
                     "achieved": round(41.0 * n / avg_s / 1e9, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(41.0 * n / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                     "avg_launch_us": round(avg_s * 1e6, 2), "traffic": None,
                     "algorithmic_per_launch": "41 B x 1e7 correspondences = 410 MB"},
        "check": {"mask_true_fraction": round(float(d_mask.float().mean().item()), 4)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle

        m = 200_000
        t0 = time.perf_counter()
        pyoracle.triangulate(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"], s["p1"][:m],
                             s["p2"][:m], threads=1)
        t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            pyoracle.triangulate(s["K"], s["prev_pose"], s["curr_pose"], s["R"], s["t"],
                                 s["p1"][:m * 5], s["p2"][:m * 5], threads=0)
            reps += 1
        tall = (time.perf_counter() - t0) / reps
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        out["cpu_baseline"] = {"value": round(m * 5 / tall, 1), "unit": "correspondences/s",
                               "cores": threads, "kind": "port",
                               "sample": f"{m * 5} correspondences x {reps} passes, all threads; "
                                         f"1 thread: {m / t1:.0f}/s on {m}",
                               "one_thread_value": round(m / t1, 1)}
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc_traffic(
        "c3", ["triangulate_invit_kernel", "triangulate_dlt_kernel"])
    ctx.close()
    return out


# --------------------------------------------------------------------------- c4 float matching
def run_c4(args, rank, world, local):
    import torch

    from sfm_amd import capi, synth

    dev = torch.device("cuda", local)
    nq = nt = 20000
    dim = 128
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    res = {}
    fp32_pass = None
    for name, integer in (("sift_like_integer", True), ("unit_norm_real", False)):
        q, t = synth.float_descriptors(nq, nt, dim=dim, seed=99 + rank, integer=integer)
        d_q = torch.from_numpy(q).to(dev)
        d_t = torch.from_numpy(t).to(dev)
        d_idx = torch.empty((nq, 2), dtype=torch.int32, device=dev)
        d_dist = torch.empty((nq, 2), dtype=torch.float32, device=dev)

        def step():
            ctx.knn2_dev(capi.NORM_L2, d_q, nq, dim * 4, d_t, nt, dim * 4, dim, d_idx, d_dist)

        dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx)
        ms, cnt = prof["l2_candidates_kernel"]
        avg_s = ms / cnt * 1e-3
        flop = 2.0 * nq * nt * dim
        tf = flop / avg_s / 1e12
        # The candidate pass runs on the fp16 matrix cores (one product per term on fp16 copies of
        # the rows) unless SFM_L2_MFMA selects the bfloat16 split (three products) or the fp32 ones.
        env = os.environ.get("SFM_L2_MFMA", "")
        mode = "f32" if env.startswith("f3") else "bf16" if env.startswith("b") else "f16"
        peak = MFMA_F32_PEAK_TFLOPS if mode == "f32" else MFMA_BF16_PEAK_TFLOPS
        roof = {"kernel": "l2_candidates_kernel" if mode == "f32" else "l2_candidates_narrow_kernel",
                "operands": mode, "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(tf / peak, 4), "avg_launch_us": round(avg_s * 1e6, 2), "traffic": None,
                "algorithmic_per_launch": f"2 x {nq} x {nt} x {dim} = {flop:.4g} FLOP"}
        if mode == "bf16":
            roof["issued"] = {"achieved": round(3 * tf, 2), "frac": round(3 * tf / peak, 4),
                              "note": "three bf16 products per term of the dot product "
                                      "(high*high, high*low, low*high)"}
        if mode != "f32":
            roof["vs_fp32_matrix_peak"] = round(tf / MFMA_F32_PEAK_TFLOPS, 4)
        res[name] = {"ms_per_step": round(1e3 * dt / args.steps, 4),
                     "kernels": {k: kernel_line(prof, k) for k in sorted(prof)},
                     "exhaustive_queries_last_step": ctx.l2_exhaustive_count(),
                     "roofline": roof}
        res[name]["_dt"] = dt
        if not integer and not os.environ.get("SFM_L2_MFMA"):
            # the candidate pass on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), same data, same
            # exact result: the bound the metric names, beside the default fp16 pass
            os.environ["SFM_L2_MFMA"] = "f32"
            try:
                k32 = max(args.steps // 5, 3)
                dt32, prof32 = timed_steps(step, k32, 2, world, dev, ctx)
            finally:
                del os.environ["SFM_L2_MFMA"]
            ms32, cnt32 = prof32["l2_candidates_kernel"]
            tf32 = flop / (ms32 / cnt32 * 1e-3) / 1e12
            fp32_pass = {"ms_per_step": round(1e3 * dt32 / k32, 4), "steps": k32,
                         "roofline": {"kernel": "l2_candidates_kernel", "operands": "f32", "bound": "mfma",
                                      "achieved": round(tf32, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(tf32 / MFMA_F32_PEAK_TFLOPS, 4),
                                      "avg_launch_us": round(1e3 * ms32 / cnt32, 2)},
                         "exhaustive_queries_last_step": ctx.l2_exhaustive_count()}
    main = res["unit_norm_real"]
    out = {
        "metric": "float-descriptor 2-NN query rows/sec (BASELINE config 4)",
        "value": round(world * nq * args.steps / main["_dt"], 1), "unit": "query rows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "c4: L2 2-NN 20000 x 20000 x 128 f32 (unit-norm real-valued; the "
                               "integer-valued variant is under 'variants')"},
        "roofline": main["roofline"],
        "variants": {k: {kk: vv for kk, vv in v.items() if kk != "_dt"} for k, v in res.items()},
    }
    if fp32_pass is not None:
        out["fp32_mfma_pass"] = fp32_pass
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc_traffic(
        "c4", [out["roofline"]["kernel"]])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle

        q, t = synth.float_descriptors(2000, nt, dim=dim, seed=99, integer=False)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            pyoracle.knn2_l2(q, t, squared=False, threads=0)
            reps += 1
        el = (time.perf_counter() - t0) / reps
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        out["cpu_baseline"] = {"value": round(2000 / el, 1), "unit": "query rows/s",
                               "cores": threads, "kind": "port",
                               "sample": f"2000 query rows x {nt} train rows x {reps} passes"}
    ctx.close()
    return out


# --------------------------------------------------------------------------- c5 all pairs
def run_c5(args, rank, world, local):
    import torch

    from sfm_amd import allpairs, capi, synth

    dev = torch.device("cuda", local)
    F = args.frames
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    # Every rank uploads only the frames f with f % world == rank (the frames it would have
    # produced or received); one all-gather over the ranks then replicates the stack -- the one
    # sizeable payload of this configuration, outside the timed region and reported by itself.
    desc = synth.descriptor_chain(F, KPS, DESC_BYTES, seed=1234)
    per = (F + world - 1) // world
    local_frames = np.zeros((per, KPS, DESC_BYTES), np.uint8)
    mine = desc[rank::world]
    local_frames[: len(mine)] = mine
    d_local = torch.from_numpy(local_frames).to(dev)
    del desc, local_frames
    barrier_sync(world)
    t0 = time.perf_counter()
    d_desc, recv_bytes = allpairs.replicate_frames(d_local, F, rank, world)
    barrier_sync(world)
    replicate = {"ms": round(1e3 * max_over_ranks(time.perf_counter() - t0, world, dev), 3),
                 "bytes_received_per_rank": int(recv_bytes), "stack_bytes": F * KPS * DESC_BYTES,
                 "collective": "all_gather_into_tensor of per-rank shares (frame f on rank f % N), then a "
                               "strided copy into frame order" if world > 1 else "none (one rank holds every frame)"}
    del d_local
    pairs = allpairs.window_pairs(F)
    matcher = allpairs.HipPairMatcher(ctx, capi.NORM_HAMMING, d_desc, KPS, DESC_BYTES, 0.5)
    n_local = len(allpairs.shard_indices(len(pairs), rank, world))
    ctx.reserve(capi.Context.match_pairs_compact_workspace(0, F, KPS, DESC_BYTES, n_local))
    result = {}

    def step():
        c, off, m = allpairs.match_all_pairs(pairs, matcher, rank, world, always_exchange=args.force_gather)
        result["n"], result["sum_counts"], result["bytes"] = int(off[-1]), int(c.sum()), m.numel() * m.element_size()

    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx)
    roof = hamming_roofline(prof, 2 * n_local, KPS, DESC_BYTES, f"2 dirs x {n_local} pairs")
    if roof["kernel"]:  # the pairs go out in several launches: price the whole step
        ms, cnt = prof[roof["kernel"]]
        scale = cnt / args.steps
        roof["launches_per_step"] = scale
        roof["achieved"] = round(roof["achieved"] / scale, 3)
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        if "valu_equivalent" in roof:
            ve = roof["valu_equivalent"]
            ve["achieved"] = round(ve["achieved"] / scale, 3)
            ve["frac"] = round(ve["achieved"] / ve["peak"], 4)
        roof["avg_launch_us"] = round(1e3 * ms / cnt, 2)
    out = {
        "metric": METRIC, "value": round(len(pairs) * 2 * KPS * args.steps / dt, 1),
        "unit": "features/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"c5: all {len(pairs)} pairs of a {F}-frame window, {KPS} kps x "
                               f"{DESC_BYTES} B, round-robin by pair over ranks, all-gather of "
                               f"match lists"},
        "ms_per_frame_pair": round(1e3 * dt / args.steps / len(pairs), 5),
        "roofline": roof,
        "kernels": {k: kernel_line(prof, k) for k in sorted(prof)},
        "check": {"gathered_matches": result.get("n"), "sum_of_pair_counts": result.get("sum_counts"),
                  "gathered_bytes": result.get("bytes"),
                  "record": "6 B (u16 prev row, u16 curr row, u16 Hamming distance)"},
        "replicate": replicate,
    }
    if result.get("n") != result.get("sum_counts"):
        raise RuntimeError(f"c5: gathered {result.get('n')} matches, the pair counts sum to {result.get('sum_counts')}")
    ctx.close()
    return out


# --------------------------------------------------------------------------- f1 epipolar filter
F64_VALU_PEAK_TLANEOPS = 39.3   # 256 CU x 4 SIMD x 16 fp64 lanes x 2.4 GHz
FM_ERROR_OPS = 48               # unfused fp64 lane-ops of one symmetric epipolar inlier test


def f1_inputs(P, seed):
    """2P frames of KPS key points: frame 2p+1 re-observes 70 % of frame 2p's features (second
    view of the same 3D points, 0.5 px noise) of which 20 % sit at a wrong place."""
    from sfm_amd import synth

    rng = np.random.default_rng(seed)
    desc = np.empty((2 * P, KPS, DESC_BYTES), np.uint8)
    kps = np.zeros((2 * P, KPS, 7), np.float32)   # 28-byte records like cv::KeyPoint, .pt first
    for p in range(P):
        s = synth.triangulation_set(KPS, seed=seed + 31 * p, outliers=0.0)
        a = rng.integers(0, 256, size=(KPS, DESC_BYTES), dtype=np.uint8)
        perm = rng.permutation(KPS)
        b = a[perm].copy()
        b ^= np.packbits(rng.random(size=(KPS, 8 * DESC_BYTES)) < 0.05, axis=1, bitorder="little")
        fresh = rng.random(KPS) < 0.3
        b[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), DESC_BYTES), dtype=np.uint8)
        kb = s["p2"][perm].copy()
        wrong = rng.random(KPS) < 0.2
        kb[wrong] = np.stack([rng.uniform(0, LK_W, wrong.sum()), rng.uniform(0, LK_H, wrong.sum())], 1)
        desc[2 * p], desc[2 * p + 1] = a, b
        kps[2 * p, :, :2], kps[2 * p + 1, :, :2] = s["p1"], kb
    return desc, kps


def run_f1(args, rank, world, local):
    """The epipolar filter of findRobustMatches (feature_processing.cpp:187-204) on the match
    lists of P frame pairs, device-resident: key point gather, findFundamentalMat (RANSAC),
    ordered compaction of the inliers."""
    import torch

    from sfm_amd import capi

    dev = torch.device("cuda", local)
    P = (args.pairs or 32)
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    desc, kps = f1_inputs(P, seed=777 + rank)
    pairs = np.stack([2 * np.arange(P), 2 * np.arange(P) + 1], 1).astype(np.int32)
    d_desc, d_kps = torch.from_numpy(desc).to(dev), torch.from_numpy(kps).to(dev)
    d_pairs = torch.from_numpy(pairs).to(dev)
    cap = KPS
    d_m0 = torch.empty((P, cap, 4), dtype=torch.int32, device=dev)
    d_c0 = torch.empty(P, dtype=torch.int32, device=dev)
    ctx.match_pairs_dev(capi.NORM_HAMMING, d_desc, 2 * P, KPS * DESC_BYTES, KPS, None, DESC_BYTES,
                        d_pairs, P, 0.5, d_m0, cap, d_c0)
    ctx.synchronize()
    n_in = d_c0.cpu().numpy().astype(np.int64)
    d_m, d_c = torch.empty_like(d_m0), torch.empty_like(d_c0)
    ctx.reserve(capi.Context.epipolar_filter_pairs_workspace(P, cap, 1000))

    def step():
        d_m.copy_(d_m0)      # the filter works in place: every step starts from the match lists
        d_c.copy_(d_c0)
        ctx.epipolar_filter_pairs_dev(d_kps, KPS * 28, 28, d_pairs, P, d_m, cap, d_c)

    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx)
    n_out = d_c.cpu().numpy().astype(np.int64)
    out = {
        "metric": "epipolar-filtered correspondences/sec (findFundamentalMat RANSAC per frame pair)",
        "value": round(world * float(n_in.sum()) * args.steps / dt, 1), "unit": "correspondences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"f1: epipolar filter of {P} frame pairs/rank/step, "
                               f"{int(n_in.mean())} cross-checked matches per pair of {KPS} key "
                               f"points, RANSAC 3 px / 0.99 / 1000"},
        "ms_per_frame_pair": round(1e3 * dt / args.steps / P, 4),
        "kernels": {k: kernel_line(prof, k) for k in sorted(prof)},
        "check": {"matches_in": int(n_in.sum()), "matches_kept": int(n_out.sum())},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle

        m = d_m0.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(P, cap)
        sets = []
        for p in range(P):
            mm = m[p][: n_in[p]]
            sets.append((kps[2 * p, mm["queryIdx"], :2].copy(), kps[2 * p + 1, mm["trainIdx"], :2].copy()))
        kept, iters, evals = 0, 0, 0.0
        t0 = time.perf_counter()
        reps = 0
        while True:
            kept = 0
            for a, b in sets:
                _, _, mask, it = pyoracle.find_fundamental(a, b)
                kept += int(mask.sum())
                if reps == 0:
                    iters += it
                    evals += float(it) * len(a)
            reps += 1
            el = time.perf_counter() - t0
            if el > 10.0 or (el > 3.0 and reps >= 3):
                break
        out["cpu_baseline"] = {"value": round(float(n_in.sum()) * reps / el, 1),
                               "unit": "correspondences/s", "cores": 1, "kind": "port",
                               "sample": f"the same {P} point sets x {reps} passes in {el:.1f} s, one "
                                         f"thread, restatement not OpenCV; mean RANSAC iterations "
                                         f"run {iters / P:.1f}",
                               "ms_per_frame_pair": round(1e3 * el / reps / P, 4)}
        out["check"]["cpu_matches_kept"] = kept
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
        # the errors the sequential loop evaluates: iterations run x points (x models, not
        # counted: at least one per iteration)
        ms, cnt = prof.get("fm_score_kernel", (0.0, 0))
        if cnt:
            per_step_s = ms / args.steps * 1e-3   # all rounds' scoring launches of one step
            ach = evals * FM_ERROR_OPS / per_step_s / 1e12
            out["roofline"] = {"kernel": "fm_score_kernel", "bound": "valu_f64",
                               "achieved": round(ach, 4), "peak": F64_VALU_PEAK_TLANEOPS,
                               "unit": "Tlane-op/s", "frac": round(ach / F64_VALU_PEAK_TLANEOPS, 5),
                               "launches_per_step": cnt // args.steps,
                               "us_per_step": round(per_step_s * 1e6, 2), "traffic": None,
                               "algorithmic_per_launch": f"{evals:.4g} inlier tests per step (iterations "
                                                         f"the sequential loop runs x points, one model "
                                                         f"each; the rounds of 64/256/1024 iterations "
                                                         f"evaluate more) x {FM_ERROR_OPS} fp64 lane-ops"}
    ctx.close()
    return out


# --------------------------------------------------------------------------- f2 relative pose
def run_f2(args, rank, world, local):
    """Tracking::findCameraPose (tracking.cpp:91-113) for P device-resident sets of tracked
    points per step: findEssentialMat (five-point RANSAC, prob 0.99, threshold 0.5 px: the
    reference's defaults), the residual check, recoverPose."""
    import torch

    from sfm_amd import capi, synth

    dev = torch.device("cuda", local)
    P, N = (args.pairs or 32), LK_POINTS
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(4242 + rank)
    pts1 = np.zeros((P, N, 2), np.float32)
    pts2 = np.zeros((P, N, 2), np.float32)
    K = None
    for p in range(P):
        s = synth.triangulation_set(N, seed=100 + p + 1000 * rank, sigma=0.2, outliers=0.0)
        K = s["K"]
        b = s["p2"].copy()
        wrong = rng.random(N) < 0.2      # mistracked points
        b[wrong] = np.stack([rng.uniform(0, LK_W, wrong.sum()), rng.uniform(0, LK_H, wrong.sum())], 1)
        pts1[p], pts2[p] = s["p1"], b
    d1, d2 = torch.from_numpy(pts1).to(dev), torch.from_numpy(pts2).to(dev)
    dE = torch.zeros((P, 10, 9), dtype=torch.float64, device=dev)
    dR = torch.zeros((P, 9), dtype=torch.float64, device=dev)
    dt_ = torch.zeros((P, 3), dtype=torch.float64, device=dev)
    dm = torch.zeros((P, N), dtype=torch.uint8, device=dev)
    df = torch.zeros(P, dtype=torch.int32, device=dev)
    di = torch.zeros(P, dtype=torch.int32, device=dev)
    ctx.reserve(capi.Context.find_camera_pose_workspace(P, N))

    def step():
        ctx.find_camera_pose_dev(d1, d2, N, None, N, P, K, 0, 0.99, 0.5, dE, dR, dt_, dm, df, di)

    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx)
    inl = di.cpu().numpy()
    out = {
        "metric": "relative poses/sec (findEssentialMat five-point RANSAC + recoverPose per frame pair)",
        "value": round(world * P * args.steps / dt, 1), "unit": "poses/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"f2: findCameraPose for {P} frame pairs/rank/step, {N} tracked points each "
                               f"(20 % mistracked, 0.2 px noise), RANSAC 0.99 / 0.5 px / 1000"},
        "ms_per_frame_pair": round(1e3 * dt / args.steps / P, 4),
        "kernels": {k: kernel_line(prof, k) for k in sorted(prof)},
        "check": {"mean_inliers": float(inl.mean()), "poses_found": int((inl > 10).sum())},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle

        t0 = time.perf_counter()
        reps, same, iters = 0, True, 0
        while True:
            p = reps % P
            E, mask, it = pyoracle.find_essential(pts1[p], pts2[p], K, 0, 0.99, 0.5)
            cnt = 0
            if len(E) == 1 and pyoracle.essential_residual(E[0], pts1[p, 0], pts2[p, 0], K) <= 1e-3:
                cnt, _, _, _ = pyoracle.recover_pose(E[0], pts1[p], pts2[p], K, mask)
            if reps < P:
                same = same and cnt == int(inl[p])
                iters += it
            reps += 1
            el = time.perf_counter() - t0
            if el > 10.0 and reps >= P:
                break
        out["cpu_baseline"] = {"value": round(reps / el, 2), "unit": "poses/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} of the same point sets in {el:.1f} s, one thread, restatement "
                                         f"not OpenCV; mean RANSAC iterations run {iters / P:.1f}",
                               "ms_per_frame_pair": round(1e3 * el / reps, 3)}
        out["check"]["identical_inlier_counts"] = bool(same)
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        # the Sampson tests the sequential loop makes: iterations x points, at least one model each
        ms, cnt_l = prof.get("em_score_kernel", (0.0, 0))
        if cnt_l:
            evals = float(iters) * N
            per_step_s = ms / args.steps * 1e-3
            ach = evals * 36 / per_step_s / 1e12
            out["roofline"] = {"kernel": "em_score_kernel", "bound": "valu_f64", "achieved": round(ach, 4),
                               "peak": F64_VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                               "frac": round(ach / F64_VALU_PEAK_TLANEOPS, 5),
                               "us_per_step": round(per_step_s * 1e6, 2), "traffic": None,
                               "algorithmic_per_launch": f"{evals:.4g} Sampson tests per step (iterations the "
                                                         f"sequential loop runs x points, one model each; a "
                                                         f"sample yields up to ten) x 36 fp64 lane-ops"}
    ctx.close()
    return out


# --------------------------------------------------------------------------- f3 corners
def run_f3(args, rank, world, local):
    """Shi-Tomasi corners (goodFeaturesToTrack, feature_processing.cpp:87-99) on P device-resident
    1280x720 frames per step, with the reference application's settings (2000 corners, quality
    0.01, minimum distance 10)."""
    import torch

    from sfm_amd import capi, synth

    dev = torch.device("cuda", local)
    P = (args.pairs or 32)
    MAXC, QUAL, MIND = 2000, 0.01, 10.0
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    n_tex = min(4, P)
    frames = np.stack([synth.to_u8(synth.texture(LK_H, LK_W, seed=42 + 17 * i + 1000 * rank))
                       for i in range(n_tex)])
    d_frames = torch.from_numpy(frames[np.arange(P) % n_tex]).to(dev)   # P frames, one batched call
    d_pts = torch.zeros((P, MAXC, 2), dtype=torch.float32, device=dev)
    d_info = torch.zeros((P, 3), dtype=torch.int32, device=dev)
    ctx.reserve(capi.Context.good_features_batch_workspace(P, LK_W, LK_H))

    def step():
        ctx.good_features_batch_dev(d_frames, P, LK_W * LK_H, LK_W, LK_H, LK_W, MAXC, QUAL, MIND, d_pts, MAXC,
                                    d_info)

    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx)
    info = d_info.cpu().numpy()
    eig_ms, eig_n = prof.get("corner_eig_kernel", (0.0, 0))
    px = LK_W * LK_H
    out = {
        "metric": "frames/sec of Shi-Tomasi corner detection (goodFeaturesToTrack, 1280x720)",
        "value": round(world * P * args.steps / dt, 1), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"f3: goodFeaturesToTrack on {P} frames {LK_W}x{LK_H}/rank/step, "
                               f"maxCorners {MAXC}, quality {QUAL}, minDistance {MIND}"},
        "ms_per_frame": round(1e3 * dt / args.steps / P, 4),
        "kernels": {k: kernel_line(prof, k) for k in sorted(prof)},
        "check": {"corners_per_frame": int(info[:, 0].mean()), "candidates_per_frame": int(info[:, 1].mean()),
                  "overflow": int(info[:, 2].max())},
    }
    if eig_n:
        avg_s = eig_ms / eig_n * 1e-3
        # the strength map is the one full-image pass: 1 B per pixel in, 4 B per pixel out
        out["roofline"] = {"kernel": "corner_eig_kernel", "bound": "hbm",
                           "achieved": round(5.0 * px / avg_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(5.0 * px / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                           "avg_launch_us": round(avg_s * 1e6, 2), "traffic": None,
                           "algorithmic_per_launch": f"5 B x {px} pixels = {5 * px / 1e6:.2f} MB "
                                                     f"(u8 image in, float strength map out)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle

        t0 = time.perf_counter()
        reps, same = 0, True
        while True:
            want = pyoracle.good_features(frames[reps % n_tex], MAXC, QUAL, MIND)
            if reps < n_tex:
                got = d_pts[reps].cpu().numpy()[: info[reps, 0]]
                same = same and np.array_equal(got, want)
            reps += 1
            el = time.perf_counter() - t0
            if el > 8.0 and reps >= n_tex:
                break
        out["cpu_baseline"] = {"value": round(reps / el, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} frames of the same set in {el:.1f} s, one thread, "
                                         f"restatement not OpenCV",
                               "ms_per_frame": round(1e3 * el / reps, 3)}
        out["check"]["identical_to_cpu"] = bool(same)
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    ctx.close()
    return out


# --------------------------------------------------------------------------- vo front end
def run_vo(args, rank, world, local):
    """The tracking front end of the reference's main loop (AppSolver::findGoodImages,
    app_solver.cpp:40-79) as one device-resident chain per step, for P frame pairs: Shi-Tomasi
    corners on the previous frame (2000, 0.01, 10), pyramidal LK to the current frame (21x21,
    maxLevel 3), the status / boundary filter, findCameraPose (RANSAC 0.99 / 0.5 px)."""
    import torch

    from sfm_amd import capi, synth

    dev = torch.device("cuda", local)
    P, MAXC = (args.pairs or 32), 2000
    ctx = capi.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    # frames of a fronto-parallel textured plane under a sideways camera move with some roll:
    # 24-36 px of image motion, i.e. a depth of about 30 baselines (recoverPose discards
    # points beyond 50, so the few-pixel motion of consecutive video frames yields no pose;
    # the reference skips frames until the motion is large enough, app_solver.cpp:45-79)
    rng = np.random.default_rng(4242 + 1000 * rank)
    texes = [synth.texture(LK_H, LK_W, seed=4242 + 17 * i + 1000 * rank) for i in range(min(4, P))]
    prevs, currs = [], []
    for p in range(P):
        ang = float(rng.uniform(0, 2 * np.pi))
        mag = float(rng.uniform(24, 36))
        tex = texes[p % len(texes)]
        prevs.append(synth.to_u8(tex))
        currs.append(synth.to_u8(synth.affine_warp(tex, (mag * np.cos(ang), mag * np.sin(ang)),
                                                   float(rng.uniform(-0.5, 0.5)))))
    prevs, currs = np.stack(prevs), np.stack(currs)
    K = np.array([[1000.0, 0, LK_W / 2], [0, 1000.0, LK_H / 2], [0, 0, 1]])
    d_prev, d_curr = torch.from_numpy(prevs).to(dev), torch.from_numpy(currs).to(dev)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    d_corners, d_info = z((P, MAXC, 2), torch.float32), z((P, 3), torch.int32)
    d_next, d_status, d_err = z((P, MAXC, 2), torch.float32), z((P, MAXC), torch.uint8), z((P, MAXC), torch.float32)
    d_p1, d_p2, d_counts = z((P, MAXC, 2), torch.float32), z((P, MAXC, 2), torch.float32), z(P, torch.int32)
    d_E, d_R, d_t = z((P, 10, 9), torch.float64), z((P, 9), torch.float64), z((P, 3), torch.float64)
    d_mask, d_found, d_inl = z((P, MAXC), torch.uint8), z(P, torch.int32), z(P, torch.int32)
    d_valid = z(P, torch.int32)
    ctx.reserve(max(capi.Context.good_features_batch_workspace(P, LK_W, LK_H),
                    capi.Context.pyrlk_workspace(P, LK_W, LK_H, LK_WIN, LK_MAXLEVEL),
                    capi.Context.find_camera_pose_workspace(P, MAXC)))
    bound = (8, 8, LK_W - 16, LK_H - 16)

    def step():
        ctx.good_features_batch_dev(d_prev, P, LK_W * LK_H, LK_W, LK_H, LK_W, MAXC, 0.01, 10.0, d_corners, MAXC,
                                    d_info)
        ctx.pyrlk_batch_dev(d_prev, d_curr, P, LK_W * LK_H, LK_W, LK_H, LK_W, d_corners, MAXC, LK_WIN,
                            LK_MAXLEVEL, 3, LK_COUNT, LK_EPS, 1e-4, 10.0, d_next, d_status, d_err)
        d_valid.copy_(d_info[:, 0])
        ctx.filter_tracked_dev(d_corners, d_next, d_status, d_valid, P, MAXC, bound, d_p1, d_p2, d_counts)
        ctx.find_camera_pose_dev(d_p1, d_p2, MAXC, d_counts, 0, P, K, 0, 0.99, 0.5, d_E, d_R, d_t, d_mask,
                                 d_found, d_inl)

    dt, prof = timed_steps(step, args.steps, args.warmup, world, dev, ctx)
    counts, inl = d_counts.cpu().numpy(), d_inl.cpu().numpy()
    groups = {"corners": ("corner_",), "tracking": ("lk_",), "pose": ("em_", "pose_")}
    by_stage = {g: round(sum(ms for k, (ms, _) in prof.items() if k.startswith(pre)) / args.steps, 4)
                for g, pre in groups.items()}
    out = {
        "metric": "frame pairs/sec of the tracking front end (corners + pyrLK + filter + findCameraPose)",
        "value": round(world * P * args.steps / dt, 1), "unit": "frame pairs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "mixed (u8 / f32 / f64)",
        "data": "synthetic",
        "config": {"workload": f"vo: {P} frame pairs {LK_W}x{LK_H}/rank/step, device-resident chain "
                               f"goodFeaturesToTrack(2000, 0.01, 10) -> pyrLK 21x21 L3 -> status/boundary "
                               f"filter -> findEssentialMat RANSAC + recoverPose"},
        "ms_per_frame_pair": round(1e3 * dt / args.steps / P, 4),
        "stage_ms_per_step": by_stage,
        "kernels": {k: kernel_line(prof, k) for k in sorted(prof)},
        "check": {"tracked_per_pair": float(counts.mean()), "pose_inliers_per_pair": float(inl.mean())},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle

        t0 = time.perf_counter()
        reps, same = 0, True
        while True:
            p = reps % P
            c = pyoracle.good_features(prevs[p], MAXC, 0.01, 10.0)
            nxt, st, _ = pyoracle.pyrlk(prevs[p], currs[p], c, win=LK_WIN, max_level=LK_MAXLEVEL, crit_type=3,
                                        max_count=LK_COUNT, epsilon=LK_EPS, max_error=10.0, acc_exact=True, threads=1)
            keep = st.astype(bool) & (bound[0] <= nxt[:, 0]) & (nxt[:, 0] < bound[0] + bound[2]) \
                & (bound[1] <= nxt[:, 1]) & (nxt[:, 1] < bound[1] + bound[3])
            a, b = c[keep], nxt[keep]
            E, m, _ = pyoracle.find_essential(a, b, K, 0, 0.99, 0.5)
            cnt = 0
            if len(E) == 1 and pyoracle.essential_residual(E[0], a[0], b[0], K) <= 1e-3:
                cnt, _, _, _ = pyoracle.recover_pose(E[0], a, b, K, m)
            if reps < P:
                same = same and len(a) == int(counts[p]) and cnt == int(inl[p])
            reps += 1
            el = time.perf_counter() - t0
            if el > 12.0 and reps >= min(P, 8):
                break
        out["cpu_baseline"] = {"value": round(reps / el, 2), "unit": "frame pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} of the same frame pairs in {el:.1f} s, one thread, "
                                         f"restatement not OpenCV",
                               "ms_per_frame_pair": round(1e3 * el / reps, 2)}
        out["check"]["identical_to_cpu"] = bool(same)
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    ctx.close()
    return out


def other_configs(args, rank, world, local):
    """Short runs of the other configurations, so that one default invocation carries a measured
    line for each: triangulation (10 M correspondences) and float matching (20 000^2 x 128) on
    rank 0's GPU only (they do not shard: replicas), and the sharded matching window over all
    ranks (strong scaling: the window is fixed, pairs are dealt over the ranks)."""
    import copy

    keep = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "dtype", "config",
            "roofline", "cpu_baseline", "check", "ms_per_frame_pair", "kernels", "variants", "fp32_mfma_pass",
            "replicate")
    res = {}
    a = copy.copy(args)
    # the window at the size the configuration is quoted on: 512 frames, 130 816 pairs (one
    # untimed and one timed pass, about 1.8 s each on one GPU)
    a.steps, a.warmup, a.frames = 1, 1, 512
    res["c5_window"] = {k: v for k, v in run_c5(a, rank, world, local).items() if k in keep}
    if world == 1:
        a = copy.copy(args)
        a.steps, a.warmup = 50, 10   # 10 ms of GPU time each; ten steps read 25 % high on a cold clock
        res["c3"] = {k: v for k, v in run_c3(a, rank, world, local).items() if k in keep}
        res["c4"] = {k: v for k, v in run_c4(a, rank, world, local).items() if k in keep}
    return res


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks as a child `torch.distributed.run` (fresh
    processes; this parent never initialises the GPU and is not replaced), forward the one JSON
    line rank 0 prints and return the child's exit status."""
    import socket
    import subprocess

    with socket.socket() as sock:   # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank launch failed with exit status {rc}", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(line)
    return 0


def run_ranks():
    """Launch check without a GPU: every rank reports its RANK / LOCAL_RANK over gloo."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
        mine = torch.tensor([rank, local, os.getpid()], dtype=torch.int64)
        every = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(every, mine)
        dist.barrier()
        dist.destroy_process_group()
    else:
        every = [torch.tensor([rank, local, os.getpid()])]
    if rank == 0:
        print(json.dumps({"metric": "ranks launched", "value": world, "unit": "ranks", "n_gpus": world,
                          "ranks": [int(e[0]) for e in every], "local_ranks": [int(e[1]) for e in every],
                          "distinct_processes": len({int(e[2]) for e in every})}))


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.workload == "ranks":
        return run_ranks()
    if os.environ.get("SFM_BENCH_WATCHDOG"):   # seconds: a hung rank prints its stacks and exits
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["SFM_BENCH_WATCHDOG"]), exit=True)
    rank, world, local = dist_setup(args.gpus, args.force_gather)
    run = {"framepair": run_framepair, "c3": run_c3, "c4": run_c4, "c5": run_c5, "f1": run_f1, "f2": run_f2, "f3": run_f3, "vo": run_vo}[args.workload]
    out = run(args, rank, world, local)
    if args.workload == "framepair" and not args.no_other_configs:
        out["other_configs"] = other_configs(args, rank, world, local)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or args.force_gather:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
