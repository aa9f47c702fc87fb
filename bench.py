#!/usr/bin/env python3
"""Benchmark of the MI355X Gaussian-splat hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config C3|C4|C1|C2|C5] [--views V]

N > 1 either arrives already launched (python -m torch.distributed.run ... bench.py --gpus N: WORLD_SIZE == N) or, from a
bare shell, starts its N ranks itself as child processes of `python -m torch.distributed.run` BEFORE anything touches the
GPU (never an exec of a process that initialised HIP) and exits with their return code.

A step = one pass of the hot path over one VIEW BATCH of the synthetic scene; per view
    preprocess -> depth sort + two-level tile lists -> composite forward -> composite backward -> per-gaussian backward
with the parameter gradients ACCUMULATING over the views of the batch.

  N = 1 (default config C3, BASELINE.json's headline): a batch is ONE camera view at 1 M gaussians / 1920x1080 / SH3.  The
        steps cycle over two cameras (views 0 and 4: the scene seen from the front and from behind, same statistics), each
        with its own view slot, so no step re-renders the view of the step before it -- a training loop cycles over a fixed
        camera set in the same way -- and the forward's launch order comes from what the same camera measured two steps ago.
  N > 1 (default config C4): a batch is the EIGHT views of the 8-GPU training batch.  Rank r renders its
        8 / N views (gaussiansplat_amd.distributed.shard_views) one after the other, then the ranks sum the flat 59 N-float
        gradient buffer over RCCL.  `--config C4 --gpus 1` is the same batch on one GPU (the anchor of the scaling curve; the
        default N = 1 line carries it as "c4_batch").  Strong scaling: the batch is fixed, value = 8 n steps / time.

Inputs (model, dC of every view) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def algorithmic_bytes(stage: str, N: int, I: int, I1: int, Iw_f: int, Iw_b: int, P: int, Tn: int, K: int, I_listed: int | None = None) -> float:
    """Algorithmic HBM bytes of one launch of a stage.  SURVEY.md section 8(d) figures for the per-gaussian and the composite
    stages (I = tile instances, Iw = list entries actually walked).  For the binning the survey prices the reference-style
    pipeline (12 I key build + 24 I sort + 8 I ranges); the two-level binning that replaced it has less to move, and its own
    minimum is used so that the fractions cannot exceed 1: per gaussian the rectangle in and out of depth order, per coarse
    instance I1 (gaussian x super-tile of 8x8 or 16x16 tiles) 6 B written once and read twice, per LISTED instance the 4-byte id
    (I_listed < I when the lists are capped, gs_config.list_cap: the ids nobody walks are not written)."""
    per_g_params = 4 * (3 + 3 + 4 + 1 + 3 * K)
    if I_listed is None:
        I_listed = I
    return {
        "preprocess": (per_g_params + 48) * N,
        "depth_sort": 16 * N,                       # one read + one write of the 8-byte (depth|id) pair
        "count_scan": 20 * N,                       # level-1 histogram: perm 4 + rectangle 8 read, rectangle in list order 8 written
        "emit": 12 * I,
        "tile_sort": 8 * N + 18 * I1 + 4 * I_listed + 8 * Tn,   # level-1 scatter + level 2 (counts, ranges, lists)
        "ranges": 8 * I + 8 * Tn,
        "composite_fwd": 40 * Iw_f + 16 * P,
        "composite_bwd": 40 * Iw_b + 20 * P + 36 * Iw_b,
        "preprocess_bwd": (36 + per_g_params) * N + per_g_params * N,
    }[stage]


def csrc_sha() -> str:
    """Identity of the device code a profile was taken with: sha256 over gaussiansplat_amd/csrc/* and include/gsplat.h."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gaussiansplat_amd", "csrc")
    for f in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "gsplat.h")]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


STAGE_KERNELS = {"composite_fwd": "composite_fwd_kernel", "composite_bwd": "composite_bwd_kernel", "preprocess": "gs_preprocess_kernel"}


def _pmc_summary(config: str):
    """The committed rocprofv3 PMC summary (profiles/*pmc_summary.json, written by tools/pmc_summary.py from separate --pmc
    passes of this same bench command) stamped with THIS build's csrc_sha and this config, else (None, None)."""
    import glob
    sha = csrc_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("csrc_sha") == sha and d.get("config", "C3") == config:
            return d, os.path.relpath(f, ROOT)
    return None, None


def _pmc_entry(summary, stage: str, early: bool):
    if not summary or stage not in STAGE_KERNELS:
        return None
    for name, v in summary["kernels"].items():
        if STAGE_KERNELS[stage] in name and (stage == "preprocess" or ("<true" in name) == early):
            return v
    return None


def valu_view(v, avg_ms: float, src: str):
    """Compute-side roofline of a kernel from its PMC entry: SQ_INSTS_VALU wave-instructions per launch against the issue peak
    of 1024 SIMDs x one wave64 VALU instruction per 2 cycles (v_fma_f32, MI355X_MICROARCH.md cycle constants)."""
    if not v or "SQ_INSTS_VALU" not in v or not v.get("clock_GHz") or avg_ms <= 0:
        return None
    cyc = v["clock_GHz"] * 1e9 * avg_ms * 1e-3                   # kernel cycles at the clock measured under the counters
    return {"valu_wave_insts": v["SQ_INSTS_VALU"], "clock_GHz": v["clock_GHz"],
            "valu_issue_frac": v["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc),
            "cycles_per_valu_inst_per_simd": 1024.0 * cyc / v["SQ_INSTS_VALU"],
            "mean_waves_per_simd": v.get("mean_waves_per_simd"), "lds_insts": v.get("SQ_INSTS_LDS"), "source": src}


def loop_cost_model(kernel_stage: str, avg_ms: float, evaluated: int):
    """Compute-side view of a composite kernel: VALU wave-instructions and microbenchmark-priced issue cycles per evaluated
    (tile, splat) entry from the disassembly of THIS build (tools/loop_cost.py -> profiles/*loop_cost.json, matched by
    csrc_sha), next to the measured SIMD-time per entry (kernel time x 1024 SIMDs / entries)."""
    import glob
    if kernel_stage not in ("composite_fwd", "composite_bwd") or evaluated <= 0 or avg_ms <= 0:
        return None
    sha = csrc_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*loop_cost.json")), reverse=True):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("csrc_sha") == sha and kernel_stage in d:
            k = d[kernel_stage]
            ns = avg_ms * 1e6 * 1024.0 / evaluated
            return {"valu_per_entry": k["valu_per_entry_all_strips"], "model_issue_cycles_per_entry": k["model_cycles_per_entry_all_strips"],
                    "measured_simd_ns_per_entry": ns, "measured_cycles_per_entry_at_2p1GHz": ns * 2.1,
                    "vgpr": k.get("vgpr"), "source": os.path.relpath(f, ROOT),
                    "note": "model = per-form issue costs of tools/valu_ubench3.hip summed over the loop body with all four strips live; "
                            "the backward evaluates ~3 of 4 strips per entry at C3"}
    return None


def launch_command(n: int, argv: list, port: int | None = None) -> list:
    """The child command bench.py starts for --gpus N > 1 from a bare shell (one rank per GPU, rendezvous on 127.0.0.1)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def cpu_baseline(t_min: float, order: int):
    """The oracle (CPU restatement of the reference arithmetic, OpenMP build) on a bounded sample of the
    workload: one fwd+bwd of the full C3 scene when the host has >= 32 threads (about 5-10 s of wall
    time), else of BASELINE config C2 (100k gaussians, 800x800, SH3)."""
    from oracle import oracle as O
    from gaussiansplat_amd import camera as gcam, synthetic
    cores = int(O.lib(omp=True).gso_num_threads())
    cfg = "C3" if cores >= 32 else "C2"
    n, W, H, deg = synthetic.CONFIGS[cfg]
    sc = synthetic.make_scene(n, W, H, deg, seed=1234 + list(synthetic.CONFIGS).index(cfg))
    cam = synthetic.scene_camera(W)
    ocam = O.camera_from_arrays(gcam.compute_transform(cam), gcam.compute_projection(cam, W, H), np.float32(cam.fx), np.float32(cam.fy),
                                np.float32(cam.near), np.float32(cam.far), cam.eye, cam.lookAt, W, H)
    dC = synthetic.make_dC(W, H, 1235)
    t0 = time.perf_counter()
    r = O.render(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, order=order, t_min=t_min, omp=True)
    O.backward(sc["means"], sc["scales"], sc["quats"], sc["opacities"], sc["shs"], deg, ocam, r["ranges"], r["ids"], dC, t_min=t_min, omp=True)
    dt = time.perf_counter() - t0
    return {"value": n / dt / 1e6, "unit": "Msplats/s", "cores": cores, "kind": "port",
            "sample": f"{cfg}: {n} gaussians, {W}x{H}, SH{deg}, one fwd+bwd of view 0, t_min={t_min:g}, {dt:.2f} s wall "
                      "(oracle/gs_oracle.c built gcc -O3 -march=native -fopenmp -ffp-contract=off; OpenMP over gaussians, tile-row "
                      "bands (lists) and tiles (composite); depth order by qsort on one thread; fp64 adjoint)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, help="C3 (default at --gpus 1), C4 = the 8-view batch (default at --gpus > 1), C1, C2, C5")
    ap.add_argument("--views", type=int, default=0, help="views per step (default: 8 for C4, else 1)")
    ap.add_argument("--view-cycle", default="0,4", help="single-view configs: the cameras the steps cycle over (view k = eye rotated by k x 45 degrees)")
    ap.add_argument("--t-min", type=float, default=1e-5, help="transmittance early-out (0 = literal reference)")
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--rank-mode", type=int, default=1, help="radix-sort stable ranks (depth sort): 1 wave64 ballots (default), 0 LDS atomic-add-return (probed at gs_create)")
    ap.add_argument("--schedule", type=int, default=0, help="gs_config.schedule (0 = library default = 3; 1 = tile order; 4 = forward by the previous frame when its slot has no history)")
    ap.add_argument("--no-view-slots", action="store_true", help="do not name view slots (the forward then launches in tile order)")
    ap.add_argument("--list-cap", type=int, default=0, help="gs_config.list_cap: 0 automatic (tile lists written as far as the view slot's previous frame walked them), 1 never, 2 also on small grids")
    ap.add_argument("--bin-path", type=int, default=0, help="gs_config.bin_path: 0 default (two-level lists; small frames -- C1 -- by the two-launch small path), 3 two-level whatever the size (A/B), 2 / 1 the radix paths")
    ap.add_argument("--debug-flags", type=int, default=0, help="gs_config.debug_flags (A/B runs: 16 = super-tiles of 8 x 8 tiles on every grid, 8 = of 16 x 16)")
    ap.add_argument("--no-cull", action="store_true", help="gs_config.alpha_cull = 0: evaluate every walked entry per pixel")
    ap.add_argument("--settle-frames", type=int, default=24, help="untimed frames per rank the timed pass's renderer runs BEFORE its W warm-up "
                    "steps (0: none; the same count on every rank).  A fresh renderer's first ~20 frames run up to 4 %% slower than its "
                    "steady state whatever W is (tools/frames_probe.py: 1.54, 1.48, 1.42, 1.40, 1.39 ms over groups of four C3 frames); "
                    "the driver's W = 5 would time that transient, not the rate a training run sees")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-literal", action="store_true", help="skip the extra literal (t_min=0) measurement")
    ap.add_argument("--no-clustered", action="store_true", help="skip the extra heavy-tailed scene (synthetic.make_scene(clustered=True)) measurement")
    ap.add_argument("--no-train-iteration", action="store_true", help="skip the extra training-iteration measurement (loss + SGD, N = 1)")
    ap.add_argument("--no-c4-anchor", action="store_true", help="N = 1, C3: skip the extra 8-view batch (the N = 1 anchor of the C4 scaling curve)")
    ap.add_argument("--grad-sync", default="allreduce", choices=["factored", "allreduce"],
                    help="N > 1, the exchange of the headline number: 'allreduce' (default, the north_star collective) = the sum of the "
                         "flat 59N-float gradient buffer; 'factored' = the same gradients from an all-reduce of the 11N geometry "
                         "floats + an all-gather of 3N colour-gradient floats per view (gaussiansplat_amd/distributed.py), 2.6x less "
                         "xGMI traffic.  The other mode is timed as well and reported beside it.")
    ap.add_argument("--no-overlap", action="store_true", help="allreduce: literally ONE collective after the last kernel (default: the Δshs "
                    "segment starts as soon as the last view's SH kernel has run, beside the geometry chain)")
    ap.add_argument("--no-pipeline", action="store_true", help="view batches: one renderer, one stream (default: the views of a rank alternate between "
                    "--pipeline-depth renderers / streams over the same model, so the lists of view k+1 are built beside the composite kernels of view k)")
    ap.add_argument("--pipeline-depth", type=int, default=3, help="view batches: views of a rank in flight at once (renderers / streams over the same model; 3: + 2 % over 2 at C4 on one GPU, 4 and 5 no better)")
    ap.add_argument("--own-renderer-in-flight", action="store_true", help="A/B: the pipelined view batches use the rank's own renderer + depth-1 new ones instead of depth new ones (measured slower: distributed.HipViewRenderer)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    args = ap.parse_args()
    if args.config is None:
        args.config = "C3" if args.gpus == 1 else "C4"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks as children of a fresh launcher process.  Nothing in THIS
        # process has imported torch or touched HIP, and it is a child process, not an exec.
        cmd = launch_command(args.gpus, sys.argv[1:])
        if os.environ.get("GS_BENCH_DRY_LAUNCH") == "1":                # tests/test_host.py: show, do not start
            print(json.dumps({"launch": cmd}))
            return
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist
    from gaussiansplat_amd import renderer as R, synthetic
    from gaussiansplat_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if os.environ.get("GS_BENCH_SINGLE_DEVICE") == "1":      # rehearsal: every rank on cuda:0 (use with --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    devices = None
    if world > 1:
        # which physical device every rank sits on (PCI bus id): a launcher that maps two ranks onto one GPU would still "scale" on paper
        pr = torch.cuda.get_device_properties(local)
        me = {"rank": rank, "local_rank": local, "name": pr.name,
              "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xFF, getattr(pr, "pci_device_id", 0)),
              "uuid": str(getattr(pr, "uuid", ""))}
        devices = [None] * world
        dist.all_gather_object(devices, me)
        ids = [(d["pci"], d["uuid"]) for d in devices]
        if len(set(ids)) != world and os.environ.get("GS_BENCH_SINGLE_DEVICE") != "1":
            raise SystemExit(f"bench.py --gpus {world}: two ranks share a device {ids}; one process per GPU is the contract")

    n, W, H, deg = synthetic.CONFIGS[args.config]
    seed = 1234 + list(synthetic.CONFIGS).index("C3" if args.config == "C4" else args.config)     # C4 = C3's scene, eight views
    scene = synthetic.make_scene(n, W, H, deg, seed=seed)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    views_per_step = args.views or (8 if args.config == "C4" else 1)
    if views_per_step > 1:
        batches = [list(range(views_per_step))]                                  # every step renders the same batch of views 0 .. V-1
    else:
        batches = [[int(v)] for v in args.view_cycle.split(",")]                 # step k renders view cycle[k % len]
    if world > 1 and views_per_step % world:
        raise SystemExit(f"--views {views_per_step} is not a multiple of --gpus {world}")
    all_views = sorted({v for b in batches for v in b})

    def camera_of(v):
        cam = synthetic.scene_camera(W, view=v)                                  # eye rotated about +y by v * 45 degrees; cam.id = v
        if args.no_view_slots:
            cam.id = None
        return cam

    cams = {v: camera_of(v) for v in all_views}
    dCs = {v: torch.as_tensor(synthetic.make_dC(W, H, seed + v)).cuda() for v in all_views}

    def make(t_min, profile_stages, sc=None):
        return R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene if sc is None else sc, device=local, order=args.order, t_min=t_min,
                             profile_stages=profile_stages, alpha_cull=not args.no_cull, rank_mode=args.rank_mode, schedule=args.schedule,
                             list_cap=args.list_cap, debug_flags=args.debug_flags, bin_path=args.bin_path)

    sync_mode = [args.grad_sync]                                                # the exchange `step` uses (N > 1)

    exchange_on = [True]                                                        # False: the same step without its collectives (compute_ms)

    def step(r, k):
        """one step = one view batch: this rank's views one after the other (gradients accumulate), then the exchange"""
        batch = batches[k % len(batches)]
        hv = r.__dict__.setdefault("_bench_hv", D.HipViewRenderer(r, args.pipeline_depth, args.own_renderer_in_flight))   # (cycle r <-> hv: collected by gc below)
        D.multi_view_step(hv, [cams[v] for v in batch], [dCs[v] for v in batch], sync=sync_mode[0], overlap=not args.no_overlap,
                          pipeline=not args.no_pipeline, exchange=exchange_on[0])

    def stage_stats(r, reset=False):
        """per-stage hipEvent sums of every ctx the renderer's views ran through (the pipelined view batches use two more)"""
        tot = {}
        for ctx in r._bench_hv.contexts() if "_bench_hv" in r.__dict__ else [r.ctx]:
            for k, (sm, ct) in ctx.stage_stats(reset=reset).items():
                a = tot.setdefault(k, [0.0, 0]); a[0] += sm; a[1] += ct
        return {k: (v[0], v[1]) for k, v in tot.items()}

    settled = {"steps": 0}
    host_enqueue = {"ms_per_step": None}

    def timed(r, steps, warmup, k0=0, settle_frames=0):
        if settle_frames > 0:                                                # untimed: the fresh renderer (and the chip) reach their steady state
            per_rank = max(1, views_per_step // world)
            ks = -(-settle_frames // per_rank)                               # steps: the same number on every rank (the steps hold collectives)
            ks += (-ks) % len(batches)
            for k in range(ks):
                step(r, k0 + k)
            settled["steps"] = ks
            k0 += ks
        for k in range(warmup):
            step(r, k0 + k)
        torch.cuda.synchronize()
        if int(r.ctx.cfg.profile_stages):
            stage_stats(r, reset=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(r, k0 + warmup + k)
        host_enqueue["ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3   # the host's share: when it approaches ms_per_step the frame is bound by the caller, not the GPU
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # Pass A (untimed survey): hipEvents around EVERY stage for a few steps -> per-stage table, dominant kernel.
    # Recording 18 events per frame costs ~3 % of a C3 frame, so the timed region below keeps only the dominant
    # kernel's pair (gs_config.profile_stages = 2 + stage).
    from gaussiansplat_amd.backend import STAGES
    ra = make(args.t_min, 1)
    ka = max(2 * len(batches), min(args.steps, 6))
    timed(ra, ka, max(len(batches), min(args.warmup, 4)))
    survey = stage_stats(ra)
    stage_ms = {k: (s / c if c else 0.0) for k, (s, c) in survey.items()}
    dom = max(stage_ms, key=stage_ms.get)
    if world > 1:                                                          # every rank must time the same stage
        di = torch.tensor([STAGES.index(dom)], device="cuda")
        dist.broadcast(di, 0)
        dom = STAGES[int(di.item())]
    del ra
    import gc
    gc.collect()
    torch.cuda.empty_cache()

    # Beside the headline (N = 1, single-view configs): the same K steps (a) on a FRESH renderer with the driver's W warm-up steps
    # only -- no settle frames: what the first frames of a run cost (settle transient) -- and (b) without view slots: no
    # launch-order history and no list caps, i.e. every frame rendered as if its camera had never been seen.
    side_runs = {}
    if world == 1 and views_per_step == 1 and args.settle_frames > 0 and not args.no_view_slots:
        ru = make(args.t_min, 0)
        side_runs["ms_per_step_unsettled"] = timed(ru, args.steps, args.warmup) / args.steps * 1e3
        del ru
        gc.collect(); torch.cuda.empty_cache()
        keep = {v: c.id for v, c in cams.items()}
        for c in cams.values():
            c.id = None
        rn = make(args.t_min, 0)
        side_runs["no_view_slot_history"] = {"ms_per_step": timed(rn, args.steps, args.warmup, settle_frames=args.settle_frames) / args.steps * 1e3,
                                             "what": "the same steps with no view slot named: the forward launches in tile order and every tile list is written in full"}
        del rn
        for v, c in cams.items():
            c.id = keep[v]
        gc.collect(); torch.cuda.empty_cache()

    # Pass B: THE timed region -- W warmup steps, then exactly K steps between barriers + synchronize.
    r = make(args.t_min, 2 + STAGES.index(dom))
    dt = timed(r, args.steps, args.warmup, settle_frames=args.settle_frames)
    host_ms = host_enqueue["ms_per_step"]
    dom_sum, dom_cnt = stage_stats(r)[dom]
    dom_ms = dom_sum / dom_cnt if dom_cnt else 0.0
    lctx = r._bench_hv.last_ctx                              # the ctx of the last view rendered
    I = lctx.num_instances
    I1 = lctx.num_coarse_instances
    wc = lctx.work_counters_ex()
    ls = lctx.list_stats()
    tile_parts = lctx.tile_parts_of_frame()                                      # waves per tile of the composite launches (gs_config.tile_parts)
    wf, wb = wc["walked_fwd"], wc["walked_bwd"]
    value = views_per_step * n * args.steps / dt / 1e6
    nranks = dist.get_world_size() if world > 1 else 1
    other = None
    if world > 1:                                              # the other exchange, same renderer, same K steps, beside the headline
        head = sync_mode[0]
        sync_mode[0] = "factored" if head == "allreduce" else "allreduce"
        dt_o = timed(r, args.steps, max(2, args.warmup // 2))
        other = {"grad_sync": sync_mode[0], "value": views_per_step * n * args.steps / dt_o / 1e6, "unit": "Msplats/s",
                 "ms_per_step": dt_o / args.steps * 1e3, "steps": args.steps,
                 "note": "same gradients (tests/test_distributed_gloo.py, tests/test_gpu_api.py); factored = all-reduce of the 11N geometry "
                         "floats + all-gather of 3N colour-gradient floats per view, rebuilt into the SH gradient locally"}
        sync_mode[0] = head
    factored = world > 1 and args.grad_sync == "factored"
    split_record = None
    if world > 1:
        # So that a shortfall against DESIGN.md's prediction can be split into compute and exchange from this one line:
        # (a) compute_ms -- the same K steps with every collective left out (per rank, and the slowest rank);
        # (b) exchange_ms -- the collectives of a step alone on the final gradient buffer, back to back, timed with events on the
        #     stream they are posted on: the flat all-reduce of 59 N floats, and the colour-factored pair (all-reduce of 11 N floats +
        #     all-gather of 3 N floats per view).  ms_per_step - compute_ms is what the exchange costs INSIDE a step (it partly
        #     overlaps the geometry chain); exchange_ms is what it costs alone.
        exchange_on[0] = False
        kc = max(2, min(args.steps, 10))
        t_c = timed(r, kc, 2) / kc * 1e3                                 # (MAX over ranks)
        exchange_on[0] = True
        mine_c = torch.tensor([0.0], dtype=torch.float64, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        exchange_on[0] = False
        for k in range(kc):
            step(r, k)
        torch.cuda.synchronize()
        exchange_on[0] = True
        mine_c[0] = (time.perf_counter() - t0) / kc * 1e3
        per_rank = [torch.zeros_like(mine_c) for _ in range(world)]
        dist.all_gather(per_rank, mine_c)
        hvb = r._bench_hv
        flat = hvb.flat
        geo = hvb.geometry_floats
        vpr = views_per_step // world
        slots = torch.zeros((vpr, n, 3), dtype=torch.float32, device="cuda")
        allc = torch.empty(world * slots.numel(), dtype=torch.float32, device="cuda")

        def ev_time(fn, reps=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        ex_flat = ev_time(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM))
        ex_two = ev_time(lambda: (dist.all_reduce(flat[geo:], op=dist.ReduceOp.SUM), dist.all_reduce(flat[:geo], op=dist.ReduceOp.SUM)))
        ex_fact = ev_time(lambda: (dist.all_reduce(flat[:geo], op=dist.ReduceOp.SUM), dist.all_gather_into_tensor(allc, slots.reshape(-1))))
        split_record = {"compute_ms": t_c, "compute_ms_by_rank": [float(x.item()) for x in per_rank], "compute_steps": kc,
                        "exchange_ms": {"allreduce_flat_59N": ex_flat, "allreduce_two_segments": ex_two, "factored_allreduce_11N_plus_allgather": ex_fact},
                        "exchange_bytes": {"allreduce_flat_59N": int(flat.numel()) * 4, "factored": int(geo) * 4 + int(slots.numel()) * 4 * world},
                        "note": "compute_ms: the K steps of the headline with every collective left out (max over ranks; per rank beside it).  "
                                "exchange_ms: the step's collectives alone on the final buffers, ten back to back between events on the posting stream "
                                "(max over ranks).  ms_per_step - compute_ms = what the exchange adds inside a step."}
        del slots, allc

    out = None
    if rank == 0:
        P, Tn, K = W * H, gx * gy, (deg + 1) ** 2
        early = args.t_min > 0
        pmc, pmc_src = _pmc_summary(args.config) if (abs(args.t_min - 1e-5) < 1e-12 or args.t_min == 0.0) else (None, None)
        by = algorithmic_bytes(dom, n, I, I1, wf, wb, P, Tn, K, ls["listed"])
        ach = by / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        dom_pmc = _pmc_entry(pmc, dom, early)
        dom_valu = valu_view(dom_pmc, dom_ms, pmc_src)
        # a kernel is called VALU-bound only on counter evidence of THIS build: its share of the VALU issue peak above its share of HBM peak
        bound = "valu" if dom_valu and dom_valu["valu_issue_frac"] > ach / HBM_PEAK_GBS else "hbm"
        stages = {}
        for st, ms in stage_ms.items():
            if ms <= 0:
                continue
            b = algorithmic_bytes(st, n, I, I1, wf, wb, P, Tn, K, ls["listed"])
            e = {"algorithmic_bytes": b, "ms": round(ms, 5), "achieved_GBs": round(b / (ms * 1e-3) / 1e9, 1), "frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            v = valu_view(_pmc_entry(pmc, st, early), ms, pmc_src)
            if v:
                e["valu_issue_frac"] = round(v["valu_issue_frac"], 4)
                e["mean_waves_per_simd"] = v["mean_waves_per_simd"]
                e["bound"] = "valu" if v["valu_issue_frac"] > e["frac"] else "hbm"
            stages[st] = e
        single = views_per_step == 1
        if single:
            what = (f"{args.config}: {n} gaussians, {W}x{H}, SH{deg}, ONE camera view per step, fwd+bwd; the steps cycle over the cameras "
                    f"{[b[0] for b in batches]} (view k: eye rotated about +y by 45k degrees; one view slot per camera)")
        else:
            what = (f"{args.config}: {n} gaussians, {W}x{H}, SH{deg}, a batch of {views_per_step} camera views per step (view k: eye rotated about +y "
                    f"by 45k degrees), fwd+bwd per view with the gradients accumulating; {views_per_step // world} view(s) per GPU"
                    + (f" (consecutive views alternate between {args.pipeline_depth} renderers / HIP streams over the same model: the lists of view k+1 are built beside "
                       "the composite kernels of view k)" if views_per_step // world > 1 and not args.no_pipeline else "")
                    + ((", then all-reduce of 11N f32 + all-gather of 3N f32 per view (colour-factored) over RCCL" if factored
                        else ", then the sum of the flat 59N-f32 gradient buffer over RCCL ("
                             + ("ONE all-reduce" if args.no_overlap else "reduced as its two segments: Δshs starts behind the last SH kernel, beside the geometry chain")
                             + ")") if world > 1 else ", no collective (one GPU)"))
        out = {
            "metric": "fwd+bwd Msplats/sec at 1M Gaussians, 1920x1080, SH deg 3" if args.config in ("C3", "C4") else f"fwd+bwd Msplats/sec ({args.config})",
            "value": value, "unit": "Msplats/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            **side_runs,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": what + "; synthetic scene of SURVEY 8d with the quaternions NORMALISED (8d leaves N(0,1)^4 raw: the reference "
                                          "never normalises q and |q| scales every footprint by |q|^4; un-normalised q is parity-tested, "
                                          "tests/test_gpu_sizes.py)",
                       "views_per_step": views_per_step, "views_per_rank": views_per_step // world, "nranks": nranks,
                       "ms_per_view": dt / args.steps * 1e3 / (views_per_step // world),
                       "settle_steps": settled["steps"],      # untimed steps before the W warm-up steps (--settle-frames)
                       "pipeline": (not args.no_pipeline) if views_per_step // world > 1 else None,
                       "grad_sync": args.grad_sync if world > 1 else None, "allreduce_overlap": (not args.no_overlap) if world > 1 else None,
                       "backend": args.backend if world > 1 else None,
                       "rccl_ranks": nranks if (world > 1 and args.backend == "nccl") else None, "devices": devices,
                       "rank_mode": int(r.ctx.cfg.rank_mode), "binning_rounds": lctx.num_rounds, "schedule": int(r.ctx.cfg.schedule) or 3,
                       "view_slots": not args.no_view_slots,
                       "order": ["index", "depth_desc", "depth_asc"][args.order], "t_min": args.t_min, "tile": 16,
                       "instances": I, "coarse_instances": I1, "walked_fwd": wf, "walked_bwd": wb, "alpha_cull": not args.no_cull,
                       "list_cap": args.list_cap, "bin_path_of_frame": lctx.bin_path_of_frame(), "host_enqueue_ms_per_step": round(host_ms, 4), "tile_parts": tile_parts, "lists_capped": ls["capped"], "listed_entries": ls["listed"], "list_segments_appended_by_waves": ls["extended_segments"],
                       "evaluated_fwd": wc["evaluated_fwd"], "evaluated_bwd": wc["evaluated_bwd"], "counters_of": "the last view rendered",
                       "seed": seed},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_note": f"survey pass ({ka} steps, hipEvents around every stage, averaged over the views rendered; not the timed region)",
            "roofline": {"kernel": dom, "bound": bound, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": dom_pmc.get("hbm_bytes_total") if dom_pmc else None, "algorithmic_bytes": by, "avg_ms": dom_ms, "launches": dom_cnt,
                         "valu": dom_valu, "csrc_sha": csrc_sha(), "stages": stages,
                         "note": "achieved / frac: algorithmic HBM bytes over the kernel time, against the 8 TB/s spec peak, as BASELINE.json "
                                 "asks -- for every stage under `stages`.  avg_ms: hipEvents around the kernel launch on the ctx stream, inside "
                                 "the timed region.  traffic (HBM bytes per launch, FETCH_SIZE x2 + WRITE_SIZE), valu and `bound` come from "
                                 "the rocprofv3 --pmc summary under profiles/ stamped with this csrc_sha (else null / 'hbm'): bound = 'valu' when "
                                 "the kernel's share of the VALU issue peak (1024 SIMDs x 1 wave64 instruction per 2 cycles) exceeds its share of "
                                 "the HBM peak (DESIGN.md s5)"},
        }
        out["roofline"]["loop_cost"] = loop_cost_model(dom, dom_ms, wc["evaluated_bwd"] if dom == "composite_bwd" else wc["evaluated_fwd"])
        if other:
            out["factored_exchange" if other["grad_sync"] == "factored" else "allreduce_exchange"] = other
        if split_record:
            out.update({"compute_ms": split_record["compute_ms"], "exchange_ms": split_record["exchange_ms"]})
            out["compute_exchange_split"] = split_record
    extras = world == 1 and views_per_step == 1
    if extras and not args.no_c4_anchor and args.config == "C3":
        # the N = 1 anchor of the C4 scaling curve, in the same run: the 8-view batch on one GPU, gradients accumulating
        v8 = list(range(8))
        cams8 = [camera_of(v) for v in v8]
        dC8 = [torch.as_tensor(synthetic.make_dC(W, H, seed + v)).cuda() for v in v8]
        r8 = make(args.t_min, 0)
        hv = D.HipViewRenderer(r8, args.pipeline_depth)
        for _ in range(3):
            D.multi_view_step(hv, cams8, dC8, pipeline=not args.no_pipeline)
        torch.cuda.synchronize()
        k8 = max(2, min(args.steps // 4, 10))
        t0 = time.perf_counter()
        for _ in range(k8):
            D.multi_view_step(hv, cams8, dC8, pipeline=not args.no_pipeline)
        torch.cuda.synchronize()
        d8 = time.perf_counter() - t0
        out["c4_batch"] = {"value": 8 * n * k8 / d8 / 1e6, "unit": "Msplats/s", "ms_per_step": d8 / k8 * 1e3, "ms_per_view": d8 / k8 / 8 * 1e3,
                           "views_per_step": 8, "n_gpus": 1, "steps": k8, "pipeline": not args.no_pipeline,
                           "what": "config C4 on ONE GPU: the eight views of the 8-GPU batch one after the other, gradients accumulating, no collective "
                                   "(= `bench.py --config C4 --gpus 1`): the N = 1 point of the strong-scaling curve `--gpus 2/4/8` continues"}
        del hv, r8
    if not args.no_literal and args.t_min > 0 and extras:          # extra measurements only at N = 1
        del r
        gc.collect()
        torch.cuda.empty_cache()
        r0 = make(0.0, 1)
        k0 = max(2, min(args.steps, 4))
        dt0 = timed(r0, k0, 2)
        st0 = stage_stats(r0)
        if rank == 0:
            out["literal_t_min_0"] = {"value": n * k0 / dt0 / 1e6, "unit": "Msplats/s", "ms_per_step": dt0 / k0 * 1e3, "steps": k0,
                                      "stage_ms": {k: round(s / c if c else 0.0, 4) for k, (s, c) in st0.items()}}
        del r0
    if not args.no_clustered and args.t_min > 0 and extras and args.config in ("C3", "C5"):
        # A heavy-tailed scene of the same size beside the spatially uniform BASELINE one (trained scenes, splat.jl:106-119):
        # 60 % of the gaussians in three faint blobs on 5 % of the frame, 0.1 % huge.  Reported, never `value`.
        gc.collect()
        torch.cuda.empty_cache()
        rc = make(args.t_min, 1, synthetic.make_scene(n, W, H, deg, seed=seed, clustered=True))
        kc = max(2, min(args.steps, 10))
        dtc = timed(rc, kc, 6, settle_frames=args.settle_frames)
        stc = stage_stats(rc)
        cctx = rc._bench_hv.last_ctx
        wcc = cctx.work_counters_ex()
        import numpy as _np
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from tile_tail import analyse as _analyse
        ev = (cctx.tile_clock(1, 30)[:, 3] & _np.uint64(0xFFFFFFFF)).astype(_np.int64)          # whole tiles: the work as the scene has it
        try:
            as_run = _analyse(cctx.tile_clock(1, -30))                                            # the backward launch as production runs it
            tail = {"mean_over_peak": as_run["mean_over_peak"], "span_us": as_run["span_us"], "simd_time_share_by_resident_waves_0_to_8": as_run["simd_time_share_by_resident_waves_0_to_8"]}
        except Exception as e:                                                                    # (no launch order on this grid)
            tail = {"error": str(e)}
        if rank == 0:
            ms_c = dtc / kc * 1e3
            same_work = out["ms_per_step"] * wcc["evaluated_fwd"] / max(wc["evaluated_fwd"], 1)
            out["clustered"] = {"value": n * kc / dtc / 1e6, "unit": "Msplats/s", "ms_per_step": ms_c, "steps": kc, "instances": cctx.num_instances,
                                "walked_fwd": wcc["walked_fwd"], "evaluated_fwd": wcc["evaluated_fwd"],
                                "evaluated_per_tile_max": int(ev.max()), "evaluated_per_tile_median": float(_np.median(ev)),
                                "stage_ms": {k: round(s / c if c else 0.0, 4) for k, (s, c) in stc.items()},
                                "backward_tile_tail_as_run": tail,
                                "uniform_frame_scaled_to_the_same_evaluated_entries_ms": same_work, "over_that": ms_c / same_work,
                                "what": "synthetic.make_scene(clustered=True): 60 % of the gaussians in three faint blobs covering 5 % of the frame, 0.1 % "
                                        "with footprints of hundreds of pixels; tiles above an even share of the frame's work run as two or four waves"}
        del rc
    if not args.no_train_iteration and extras and args.config in ("C1", "C2", "C3"):
        # SURVEY 8(f).2 beside the headline: one iteration of src/train.jl as intended = the fwd+bwd step + the L1/DSSIM loss
        # with its image gradient (gs_loss.hip) + the SGD update; reported, never `value`
        from gaussiansplat_amd import train as TR
        gc.collect()
        torch.cuda.empty_cache()
        rt = make(args.t_min, 0)
        gt = torch.rand((3, H, W), device="cuda")
        lf = TR.getLossFunction((W, H, 3), 11, 3, renderer=rt)
        cam0 = cams[batches[0][0]]
        for _ in range(3):
            TR.trainStep(rt, gt, 1e-4, lf, cam0, want_loss=False)
        torch.cuda.synchronize()
        kt = max(2, min(args.steps, 20))
        t0 = time.perf_counter()
        for _ in range(kt):
            TR.trainStep(rt, gt, 1e-4, lf, cam0, want_loss=False)
        torch.cuda.synchronize()
        out["train_iteration"] = {"ms": (time.perf_counter() - t0) / kt * 1e3, "iterations": kt,
                                  "what": "preprocess, lists, forward, L1+DSSIM loss and image gradient, backward, SGD step (train.jl:33-56)"}
        for _ in range(2):
            TR.trainStep(rt, gt, 1e-4, lf, cam0, want_loss=False, fused_sgd=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(kt):
            TR.trainStep(rt, gt, 1e-4, lf, cam0, want_loss=False, fused_sgd=True)
        torch.cuda.synchronize()
        out["train_iteration"]["ms_with_fused_backward_sgd"] = (time.perf_counter() - t0) / kt * 1e3
        del rt
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:                # rank 0, N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.t_min, args.order)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
