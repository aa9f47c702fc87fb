#!/usr/bin/env python3
"""Benchmark of the MI355X Gaussian-splat hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 either arrives already launched (python -m torch.distributed.run ... bench.py --gpus N: WORLD_SIZE == N) or, from a
bare shell, starts its N ranks itself as child processes of `python -m torch.distributed.run` BEFORE anything touches the
GPU (never an exec of a process that initialised HIP) and exits with their return code.

A step = one fwd+bwd pass of the hot path over one camera view of the synthetic scene:
preprocess -> tile|depth keys + radix sort -> composite forward -> composite backward ->
per-gaussian backward (+ ONE RCCL all-reduce of the flat 59N-float gradient buffer when N > 1; the colour-factored
exchange is timed beside it and reported under "factored_exchange").
Weak scaling: every GPU renders its own view (one camera per GPU) of the replicated model.
Inputs (model, dC) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
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


def algorithmic_bytes(stage: str, N: int, I: int, Iw_f: int, Iw_b: int, P: int, Tn: int, K: int) -> float:
    """SURVEY.md section 8(d) per-unit figures (I = instances, Iw = list entries actually walked)."""
    per_g_params = 4 * (3 + 3 + 4 + 1 + 3 * K)
    return {
        "preprocess": (per_g_params + 48) * N,
        "depth_sort": 16 * N,                       # one read + one write of the 8-byte (depth|id) pair
        "count_scan": 12 * N,
        "emit": 12 * I,
        "tile_sort": 24 * I,
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


def _pmc_entry(kernel_stage: str, early: bool, config: str):
    """The dominant kernel's entry in a committed rocprofv3 PMC summary (profiles/*pmc_summary.json, written by
    tools/pmc_summary.py from separate --pmc passes of this same bench command: SQ/GRBM counters, FETCH_SIZE, WRITE_SIZE)
    -- only a summary stamped with THIS build's csrc_sha and this config counts; a stale profile gives None."""
    import glob
    sha = csrc_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("csrc_sha") != sha or d.get("config", "C3") != config:
            continue
        for name, v in d["kernels"].items():
            if kernel_stage in name and (("<true" in name) == early):
                return v, os.path.relpath(f, ROOT)
    return None, None


def measured_counters(kernel_stage: str, config: str, t_min: float, avg_ms: float):
    """(traffic, valu) of the dominant kernel from the PMC summary of THIS build (else None, None).
    traffic: HBM bytes per launch = FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE.
    valu: the compute-side roofline -- SQ_INSTS_VALU wave-instructions per launch against the issue peak of 1024 SIMDs x
    one wave64 VALU instruction per 2 cycles (v_fma_f32, MI355X_MICROARCH.md cycle constants)."""
    if abs(t_min - 1e-5) > 1e-12 and t_min != 0.0:
        return None, None
    v, src = _pmc_entry(kernel_stage, t_min > 0, config)
    if not v:
        return None, None
    valu = None
    if "SQ_INSTS_VALU" in v and v.get("clock_GHz") and avg_ms > 0:
        cyc = v["clock_GHz"] * 1e9 * avg_ms * 1e-3                   # kernel cycles at the clock measured under the counters
        valu = {"valu_wave_insts": v["SQ_INSTS_VALU"], "clock_GHz": v["clock_GHz"],
                "valu_issue_frac": v["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc),
                "cycles_per_valu_inst_per_simd": 1024.0 * cyc / v["SQ_INSTS_VALU"],
                "mean_waves_per_simd": v.get("mean_waves_per_simd"), "lds_insts": v.get("SQ_INSTS_LDS"), "source": src}
    return v.get("hbm_bytes_total"), valu


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
            "sample": f"{cfg}: {n} gaussians, {W}x{H}, SH{deg}, one fwd+bwd, t_min={t_min:g}, {dt:.2f} s wall "
                      "(oracle/gs_oracle.c, OpenMP over gaussians/tiles; list building single-threaded; fp64 adjoint)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--t-min", type=float, default=1e-5, help="transmittance early-out (0 = literal reference)")
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--rank-mode", type=int, default=1, help="radix-sort stable ranks (depth sort): 1 wave64 ballots (default), 0 LDS atomic-add-return (probed at gs_create; -4 us per C3 frame)")
    ap.add_argument("--schedule", type=int, default=3, help="gs_config.schedule (3 default; 4 = forward tiles ordered by the previous frame's per-tile work)")
    ap.add_argument("--no-cull", action="store_true", help="gs_config.alpha_cull = 0: evaluate every walked entry per pixel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-literal", action="store_true", help="skip the extra literal (t_min=0) measurement")
    ap.add_argument("--no-train-iteration", action="store_true", help="skip the extra training-iteration measurement (loss + SGD, N = 1)")
    ap.add_argument("--grad-sync", default="allreduce", choices=["factored", "allreduce"],
                    help="N > 1, the exchange of the headline number: 'allreduce' (default, the north_star collective) = ONE all-reduce "
                         "of the flat 59N-float gradient buffer; 'factored' = the same gradients from an all-reduce of the 11N geometry "
                         "floats + an all-gather of 3N colour-gradient floats per view (gaussiansplat_amd/distributed.py), 2.6x less "
                         "xGMI traffic.  The other mode is timed as well and reported beside it.")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    args = ap.parse_args()

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

    n, W, H, deg = synthetic.CONFIGS[args.config]
    seed = 1234 + list(synthetic.CONFIGS).index(args.config)
    scene = synthetic.make_scene(n, W, H, deg, seed=seed)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    view = int(os.environ.get("GS_BENCH_VIEW", rank % 8))      # one camera per GPU: eye rotated about +y by k*45 degrees
    cam = synthetic.scene_camera(W, view=view)
    dC = torch.as_tensor(synthetic.make_dC(W, H, seed + rank)).cuda()

    def make(t_min, profile_stages):
        return R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, device=local, order=args.order, t_min=t_min,
                             profile_stages=profile_stages, alpha_cull=not args.no_cull, rank_mode=args.rank_mode, schedule=args.schedule)

    from gaussiansplat_amd import distributed as D
    sync_mode = [args.grad_sync]                                                # the exchange `step` uses (N > 1)
    all_cams = [synthetic.scene_camera(W, view=int(os.environ.get("GS_BENCH_VIEW", k % 8))) for k in range(world)]   # rank k renders view k
    cam_records = D.view_records(all_cams, W, H)                                # host, once: the cameras of all ranks' views

    def step(r):
        R.resetGrads(r)
        if not (world > 1 and sync_mode[0] == "factored"):
            tps = R.preprocess(r, cam)
            R.compactIdxs(r, (16, 16), (gx, gy))
            R.forward(r, tps, (16, 16), (gx, gy))
            R.backward(r, dC)
            if world > 1:
                dist.all_reduce(r.splatGrads.flat)       # ONE flat RCCL all-reduce (59 N floats at SH3)
            return
        # colour-factored exchange: same gradients, 11N floats all-reduced + 3N per view all-gathered; the gather
        # overlaps the per-gaussian backward (gaussiansplat_amd/distributed.py)
        if "_bench_hv" not in r.__dict__:                                   # (cycle r <-> hv: collected by gc below)
            r._bench_hv = D.HipViewRenderer(r)
            r._bench_allc = torch.empty(world * 3 * n, dtype=torch.float32, device="cuda")
        D.factored_one_view_step(r._bench_hv, cam, dC, cam_records, r._bench_allc)

    def timed(r, steps, warmup):
        for _ in range(warmup):
            step(r)
        torch.cuda.synchronize()
        r.ctx.stage_stats(reset=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(r)
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
    ka = max(2, min(args.steps, 5))
    timed(ra, ka, max(1, args.warmup))
    survey = ra.ctx.stage_stats()
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

    # Pass B: THE timed region -- W warmup steps, then exactly K steps between barriers + synchronize.
    r = make(args.t_min, 2 + STAGES.index(dom))
    dt = timed(r, args.steps, args.warmup)
    dom_sum, dom_cnt = r.ctx.stage_stats()[dom]
    dom_ms = dom_sum / dom_cnt if dom_cnt else 0.0
    I = r.ctx.num_instances
    wc = r.ctx.work_counters_ex()
    wf, wb = wc["walked_fwd"], wc["walked_bwd"]
    value = world * n * args.steps / dt / 1e6
    nranks = dist.get_world_size() if world > 1 else 1
    other = None
    if world > 1:                                              # the other exchange, same renderer, same K steps, beside the headline
        head = sync_mode[0]
        sync_mode[0] = "factored" if head == "allreduce" else "allreduce"
        dt_o = timed(r, args.steps, max(2, args.warmup // 2))
        other = {"grad_sync": sync_mode[0], "value": world * n * args.steps / dt_o / 1e6, "unit": "Msplats/s",
                 "ms_per_step": dt_o / args.steps * 1e3, "steps": args.steps,
                 "note": "same gradients (tests/test_distributed_gloo.py, tests/test_gpu_api.py); factored = all-reduce of the 11N geometry "
                         "floats + all-gather of 3N colour-gradient floats per view, rebuilt into the SH gradient locally"}
        sync_mode[0] = head
    factored = world > 1 and args.grad_sync == "factored"

    out = None
    if rank == 0:
        P, Tn, K = W * H, gx * gy, (deg + 1) ** 2
        by = algorithmic_bytes(dom, n, I, wf, wb, P, Tn, K)
        ach = by / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        out = {
            "metric": "fwd+bwd Msplats/sec at 1M Gaussians, 1920x1080, SH deg 3" if args.config == "C3" else f"fwd+bwd Msplats/sec ({args.config})",
            "value": value, "unit": "Msplats/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {n} gaussians, {W}x{H}, SH{deg}, one camera view per GPU per step, fwd+bwd"
                                   + ((", RCCL all-reduce of 11N f32 + all-gather of 3N f32 per view (colour-factored)" if factored
                                       else ", one RCCL all-reduce of 59N f32") if world > 1 else "")
                                   + "; synthetic scene of SURVEY 8d with the quaternions NORMALISED (8d leaves N(0,1)^4 raw: the reference "
                                     "never normalises q and |q| scales every footprint by |q|^4; un-normalised q is parity-tested, "
                                     "tests/test_gpu_sizes.py)",
                       "nranks": nranks, "grad_sync": args.grad_sync if world > 1 else None, "backend": args.backend if world > 1 else None,
                       "rank_mode": int(r.ctx.cfg.rank_mode), "binning_rounds": r.ctx.num_rounds, "schedule": args.schedule,
                       "order": ["index", "depth_desc", "depth_asc"][args.order], "t_min": args.t_min, "tile": 16,
                       "instances": I, "walked_fwd": wf, "walked_bwd": wb, "alpha_cull": not args.no_cull,
                       "evaluated_fwd": wc["evaluated_fwd"], "evaluated_bwd": wc["evaluated_bwd"], "seed": seed},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "stage_ms_note": f"survey pass ({ka} steps, hipEvents around every stage, not the timed region)",
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes": by, "avg_ms": dom_ms, "launches": dom_cnt, "valu": None,
                         "csrc_sha": csrc_sha(),
                         "note": "avg_ms: hipEvents around the kernel launch on the ctx stream, inside the timed region.  traffic (HBM "
                                 "bytes per launch, FETCH_SIZE x2 + WRITE_SIZE) and valu (SQ_INSTS_VALU against the issue peak of 1024 "
                                 "SIMDs x 1 wave64 instruction per 2 cycles) come from the rocprofv3 --pmc summary under profiles/ "
                                 "stamped with this csrc_sha, else null.  The composite kernels are VALU-bound (DESIGN.md s5); the HBM "
                                 "fraction is reported as measured"},
        }
        out["roofline"]["traffic"], out["roofline"]["valu"] = measured_counters(dom, args.config, args.t_min, dom_ms)
        out["roofline"]["loop_cost"] = loop_cost_model(dom, dom_ms, wc["evaluated_bwd"] if dom == "composite_bwd" else wc["evaluated_fwd"])
        if other:
            out["factored_exchange" if other["grad_sync"] == "factored" else "allreduce_exchange"] = other
    if not args.no_literal and args.t_min > 0 and world == 1:      # extra measurements only at N = 1
        del r
        torch.cuda.empty_cache()
        r0 = make(0.0, 1)
        k0 = max(2, min(args.steps, 5))
        dt0 = timed(r0, k0, 1)
        st0 = r0.ctx.stage_stats()
        if rank == 0:
            out["literal_t_min_0"] = {"value": world * n * k0 / dt0 / 1e6, "unit": "Msplats/s", "ms_per_step": dt0 / k0 * 1e3, "steps": k0,
                                      "stage_ms": {k: round(s / c if c else 0.0, 4) for k, (s, c) in st0.items()}}
        del r0
    if not args.no_train_iteration and world == 1 and args.config in ("C1", "C2", "C3"):
        # SURVEY 8(f).2 beside the headline: one iteration of src/train.jl as intended = the fwd+bwd step + the L1/DSSIM loss
        # with its image gradient (gs_loss.hip) + the SGD update; reported, never `value`
        from gaussiansplat_amd import train as TR
        torch.cuda.empty_cache()
        rt = make(args.t_min, 0)
        gt = torch.rand((3, H, W), device="cuda")
        lf = TR.getLossFunction((W, H, 3), 11, 3, renderer=rt)
        for _ in range(3):
            TR.trainStep(rt, gt, 1e-4, lf, cam, want_loss=False)
        torch.cuda.synchronize()
        kt = max(2, min(args.steps, 20))
        t0 = time.perf_counter()
        for _ in range(kt):
            TR.trainStep(rt, gt, 1e-4, lf, cam, want_loss=False)
        torch.cuda.synchronize()
        out["train_iteration"] = {"ms": (time.perf_counter() - t0) / kt * 1e3, "iterations": kt,
                                  "what": "preprocess, lists, forward, L1+DSSIM loss and image gradient, backward, SGD step (train.jl:33-56)"}
        for _ in range(2):
            TR.trainStep(rt, gt, 1e-4, lf, cam, want_loss=False, fused_sgd=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(kt):
            TR.trainStep(rt, gt, 1e-4, lf, cam, want_loss=False, fused_sgd=True)
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
