#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Mpaths/s (+ Mrays/s) on the Room scene at 256x256.
A step = one CLI iteration per GPU: every rank traces 256 samples per pixel of
room.json at 256x256 (reference src/CLI/main.cpp:64-70: fov pi/4, 256 samples per
iteration; depth 5) = 16.8 M paths, 83.9 M rays per GPU per step.  With N ranks the
ranks take disjoint sample ranges of one N*256-spp iteration (weak scaling, the default),
their per-GPU HDR sum buffers are combined with one RCCL reduce to rank 0, and rank 0
applies the running mean (abstract_pathtracer.cpp:88-90).  --scaling strong keeps the
job fixed instead, the way BASELINE.json's config 4 runs: ONE 256-spp iteration per step,
each rank 256/N samples of every pixel.  Either way the reduce and the running mean run on
a second stream under the next step's render (parallel.IterationPipeline: double-buffered
sums, events, no host synchronisation inside the timed region).  Scene tables and ray
queues are resident in HBM before the timed region; nothing crosses PCIe inside it.

Prints ONE JSON line on rank 0 (see the round contract): value = whole-job Mpaths/s.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "assets", "res", "scenes")
SCENE = os.path.join(SCENES, "room.json")
W = H = 256
SPI = 256
FOV = math.pi / 4.0

# BASELINE.json configs.  The headline metric is quoted on Room 256x256 (the default); the others are
# reachable with --workload for the per-config numbers in DESIGN.md / profiles/.
WORKLOADS = {
    "room": dict(scene="room.json", w=256, h=256, desc="room.json 256x256, 10 nodes / 36 triangles / 2 spheres"),
    "spheres": dict(scene="materials.json", w=512, h=512, desc="materials.json 512x512 (config 2), 25 spheres, equirect bg0.hdr"),
    "mario": dict(scene="mario.json", w=128, h=128, desc="mario.json 128x128 (config 3), 14552-triangle KD mesh + 2048^2 texture"),
    "synthetic": dict(scene=None, w=2048, h=2048, desc="synthetic displaced sphere (config 5), textured, equirect bg0.hdr"),
}


def load_workload(name, triangles):
    from pathtracing_amd import load_scene, scene as S, workloads
    wl = WORKLOADS[name]
    if wl["scene"] is not None:
        return load_scene(os.path.join(SCENES, wl["scene"]))
    bg = S.load_image(os.path.join(ROOT, "tests", "golden", "assets", "res", "textures", "bg0.hdr"))
    return workloads.synthetic_scene(triangles, 4096, bg)
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector (FMA = 2 flop)
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E spec
PEAK_L2_GBS = 34500.0      # MI355X_MICROARCH.md: aggregate L2 (8 XCDs x 4 MiB)
PTK_DEPTH = 5              # pathtracer.cpp:16


def load_pmc_traffic(workload):
    """HBM bytes per launch of the traced kernels from the newest committed rocprofv3 PMC summary of this
    workload (profiles/rNN_<tag>_pmc.json, written by scripts/collect_profiles.py from separate --pmc passes,
    FETCH_SIZE doubled for gfx950's 128-B requests as MI355X_MICROARCH.md prescribes).  bench.py cannot
    collect counters itself; an absent file gives traffic = null."""
    import glob
    tag = {"room": "room", "spheres": "spheres", "mario": "mario", "synthetic": "synth5m"}[workload]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_pmc.json")))
    if not files:
        return {}
    out = {"source": os.path.relpath(files[-1], ROOT)}
    with open(files[-1]) as f:
        data = json.load(f)
    for k, v in data.items():
        d = v.get("derived", {})
        if "hbm_read_bytes_per_dispatch" in d:
            name = k.split(",")[0].replace("<false", "<false>").replace("<true", "<true>") if "k_bounce<" in k else k.split("<")[0]
            if name.startswith("k_bounce_split"):
                name = "k_bounce<false>" if "false" in k else "k_bounce<true>"
            out[name] = d["hbm_read_bytes_per_dispatch"] + d.get("hbm_write_bytes_per_dispatch", 0.0)
    return out


def flops_from_stats(st, costs, equirect):
    """Algorithmic flops of a run from the oracle's operation counts (DESIGN.md 'Flop model')."""
    f = 0
    for k in ("path", "node", "kd_box", "tri_a", "tri_b", "tri_full", "tri_hit", "sphere", "sphere_hit", "node_hit",
              "shade", "mesh_win"):
        f += st[k] * costs[k]
    f += (st["shade"] - st["mesh_win"]) * costs["sphere_win"]
    f += st["miss"] * (costs["miss_equirect"] if equirect else costs["miss_color"])
    return f


def cpu_baseline(scene, workload, target_seconds=12.0):
    """The CPU restatement of the reference ("port"), timed on this box's host cores on a bounded
    sample of the same workload (same scene and resolution, fewer samples per pixel).  Also returns
    the operation counts the roofline needs: flops per ray from the reference-order run, and the
    nodes / triangles / texels per ray of the library's pruned KD walk (oracle walk model 1)."""
    import numpy as np
    from oracle import oracle
    ora = oracle.OracleScene(scene)
    # the GPU box's CPU share is 16; os.cpu_count() shows the whole machine there
    cores = min(16, os.cpu_count() or 1)
    cores = int(os.environ.get("PTK_BENCH_CPU_THREADS", cores))
    sw, sh = min(W, 512), min(H, 512)   # bounded sample: at most 512x512 of the frame's field of view
    t0 = time.perf_counter()
    ora.render_samples(sw, sh, FOV, 0, 0, 1, threads=cores)
    probe = time.perf_counter() - t0
    spp = int(max(1, min(4096, target_seconds / max(probe, 1e-3))))
    t0 = time.perf_counter()
    _, rays, st = ora.render_samples_stats(sw, sh, FOV, 0, 0, spp, threads=cores)
    dt = time.perf_counter() - t0
    with open(os.path.join(ROOT, "tests", "golden", "flop_costs.json")) as f:
        costs = json.load(f)
    from pathtracing_amd import scene as S
    flops = flops_from_stats(st, costs, equirect=isinstance(scene.background, S.EquirectangularMap))
    paths = sw * sh * spp
    try:
        oracle.set_walk_model(1)
        _, mrays, mst = ora.render_samples_stats(sw, sh, FOV, 0, 0, max(1, spp // 8), threads=cores)
    finally:
        oracle.set_walk_model(0)
    # algorithmic HBM bytes per ray of the GPU path (DESIGN.md "Roofline"): 64 B per KD node tested,
    # 64 B per triangle hit record, 64 B shading record per triangle winner, 12 B per texel, and the
    # SoA ray record in + out (52 + 52 B)
    bytes_per_ray = (64 * mst["kd_box"] + 64 * mst["tri_a"] + 64 * mst["mesh_win"] + 12 * mst["texel"]) / mrays + 104
    return {
        "value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
        "sample": f"{WORKLOADS[workload]['desc']}; {sw}x{sh}, {spp} spp ({paths} paths, {rays} rays) in {dt:.1f} s, "
                  f"oracle/pt_oracle.c, {cores} pthreads; the reference itself is unbuildable here (glm/spdlog absent)",
        "mrays_per_s": rays / dt / 1e6,
    }, {"flops_per_ray": flops / rays, "bytes_per_ray": bytes_per_ray,
        "kd_nodes_per_ray": mst["kd_box"] / mrays, "triangles_per_ray": mst["tri_a"] / mrays,
        "texels_per_ray": mst["texel"] / mrays,
        "ref_kd_nodes_per_ray": st["kd_box"] / rays, "ref_triangles_per_ray": st["tri_a"] / rays}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true", help="also time the CPU port for workloads other than room")
    ap.add_argument("--traversal", type=int, default=None, help="PTK_OPT_TRAVERSAL override (0/1), for A/B runs")
    ap.add_argument("--kd-prune", type=int, default=None, help="PTK_OPT_KD_PRUNE override (0/1), for A/B runs")
    ap.add_argument("--node-cull", type=int, default=None, help="PTK_OPT_NODE_CULL override (0/1/2), for A/B runs")
    ap.add_argument("--balance", type=int, default=None, help="PTK_OPT_BALANCE override (0/1/2), for A/B runs")
    ap.add_argument("--split", type=int, default=None, help="PTK_OPT_SPLIT override (0/1/2), for A/B runs")
    ap.add_argument("--compact", type=int, default=None, help="PTK_OPT_COMPACT override (0/1/2), for A/B runs")
    ap.add_argument("--fuse", type=int, default=None, help="PTK_OPT_FUSE override (0/1/2), for A/B runs")
    ap.add_argument("--overlap", type=int, default=None, help="PTK_OPT_OVERLAP override (0/1), for A/B runs and profiles")
    ap.add_argument("--walk-stack", type=int, default=None, help="PTK_OPT_WALK_STACK override (entries of the walker's short stack), for A/B runs")
    ap.add_argument("--no-timing", action="store_true", help="no per-launch HIP events in the timed region (no roofline object)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N>1 path on a box with ONE GPU: every rank uses cuda:0 and the HDR sum "
                         "buffers are reduced through host memory (not a measurement)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank traces 256 spp per step; strong: one 256-spp iteration per step split "
                         "over the ranks by sample range (config 4's shape)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="room")
    ap.add_argument("--triangles", type=int, default=5_000_000, help="synthetic workload: triangle count")
    ap.add_argument("--resolution", type=int, nargs=2, default=None)
    args = ap.parse_args()
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when the process group comes up (seen with
    # the nccl backend on ROCm 7), and anything else a library writes there would break a parser too: from here on file
    # descriptor 1 is stderr, and the JSON line goes to the saved original.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    global W, H
    W, H = args.resolution or (WORKLOADS[args.workload]["w"], WORKLOADS[args.workload]["h"])

    import torch
    import torch.distributed as dist
    from pathtracing_amd import Context, load_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # PTK_BENCH_FORCE_DIST=1: a one-rank job still goes through torch.distributed (rehearsal of the RCCL leg on a one-GPU
    # box: process group, reduce on the exchange stream, barriers)
    distributed = world > 1 or os.environ.get("PTK_BENCH_FORCE_DIST") == "1"
    if args.backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    scene = load_workload(args.workload, args.triangles)
    drop = os.environ.get("PTK_BENCH_DROP")  # experiment knob: drop nodes by kind ("sphere", "leaf12", ...)
    if drop:
        from pathtracing_amd import scene as S
        keep = []
        for n in scene.nodes:
            kind = "sphere" if isinstance(n.shape, S.Sphere) else f"leaf{n.shape.triangles.shape[0]}"
            if kind not in drop.split(","):
                keep.append(n)
        scene.nodes = keep
    ctx = Context(local_rank)
    # (the render and exchange streams are the pipeline's: parallel.IterationPipeline hands the render stream to the
    # context and runs the reduce, which torch.distributed orders against the current stream, on the exchange stream)
    if args.traversal is not None:
        ctx.set_option(1, args.traversal)
    if args.kd_prune is not None:
        ctx.set_option(2, args.kd_prune)
    if args.node_cull is not None:
        ctx.set_option(3, args.node_cull)
    if args.balance is not None:
        ctx.set_option(4, args.balance)
    if args.split is not None:
        ctx.set_option(5, args.split)
    if args.compact is not None:
        ctx.set_option(8, args.compact)
    if args.fuse is not None:
        ctx.set_option(9, args.fuse)
    if args.overlap is not None:
        ctx.set_option(10, args.overlap)
    if args.walk_stack is not None:
        ctx.set_option(7, args.walk_stack)
    t_up = time.perf_counter()
    ctx.upload(scene)
    torch.cuda.synchronize(dev)
    upload_ms = (time.perf_counter() - t_up) * 1e3  # scene flattening + KD build + tables to HBM, outside the timed steps
    ctx.render_begin(W, H, FOV)
    from pathtracing_amd import parallel
    pipe = parallel.IterationPipeline(ctx, W, H, dev, rank, world, dist if distributed else None,
                                      host_reduce=(args.backend == "gloo"), force_exchange=distributed)
    # samples per pixel of one step over all ranks: weak = 256 per rank, strong = 256 in total (rank r renders the
    # r-th part of the step's global sample range either way: parallel.iteration_share)
    step_spi = SPI * world if args.scaling == "weak" else SPI

    def step(it):
        pipe.step(it, step_spi, seed=0)

    for it in range(args.warmup):
        step(it)
    pipe.synchronize()
    torch.cuda.synchronize(dev)
    ctx.reset_counters()
    # per-launch HIP events on the launch stream for the roofline's kernel durations, on every 4th step: the events
    # themselves cost GPU time (1 % of a Room step, 5 % of a materials step), a quarter of the steps gives the same average
    ctx.enable_timing(0 if args.no_timing else 4)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for it in range(args.steps):
        step(args.warmup + it)
    pipe.synchronize()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    image = pipe.image
    # ranks that took part in the exchange (RCCL when the backend is nccl), read after the first reduce has run
    rccl_ranks = dist.get_world_size() if (distributed and pipe.reduces > 0) else 1

    paths, rays = ctx.counters()
    timing = ctx.timing()
    ctx.enable_timing(False)
    stats = torch.tensor([elapsed, float(paths), float(rays)], dtype=torch.float64,
                         device="cpu" if args.backend == "gloo" else dev)
    if distributed:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        paths, rays = int(sm[1]), int(sm[2])

    if rank == 0:
        finite = bool(torch.isfinite(image).all())
        assert finite and float(image.mean()) > 0.0, "render produced no radiance"
        assert paths == W * H * step_spi * args.steps, (paths, W * H * step_spi * args.steps)
        wl_scene = WORKLOADS[args.workload]["scene"]
        data_desc = (f"{wl_scene} (the reference's own scene file, src/res/scenes/{wl_scene}; no dataset or checkpoint involved)"
                     if wl_scene else f"synthetic ({args.triangles}-triangle displaced sphere generated in code, workloads.synthetic_scene)")
        out = {
            "metric": "Mpaths/s", "value": paths / elapsed / 1e6, "unit": "Mpaths/s",
            "mrays_per_s": rays / elapsed / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "rccl_ranks": rccl_ranks if args.backend == "nccl" else None,
            "vs_baseline": None, "dtype": "f32", "data": data_desc, "scene_upload_ms": upload_ms,
            "config": {"workload": f"{WORKLOADS[args.workload]['desc']}; {W}x{H}, " +
                                   (f"{SPI} spp per GPU per step (one CLI iteration per GPU)" if args.scaling == "weak"
                                    else f"{SPI} spp per step over all GPUs (one CLI iteration split by sample range)") +
                                   ", fov pi/4, depth 5",
                       "paths_per_step_per_gpu": W * H * step_spi // world, "rays_per_path": rays / paths,
                       "parallelism": f"sample-range x{world}" + (" + RCCL reduce(sum) to rank 0 on a second stream, "
                                                                  "double-buffered sums" if distributed else "")},
        }
        model = None
        model_tables = {"kd_bytes": 0}
        for m in range(64):   # tree tables of the uploaded scene: 64-B nodes, 64-B hit records + 8 B of indices per leaf slot
            try:
                ki = ctx.kd_info(m)
            except Exception:
                break
            if ki.num_nodes > 1:
                model_tables["kd_bytes"] += 64 * ki.num_nodes + 72 * ki.num_leaf_prims
        if not args.no_cpu_baseline and (args.workload == "room" or args.cpu_baseline):
            # the timed CPU baseline is an N = 1 figure; with more ranks only a short sample runs, for the
            # operation counts the roofline needs
            out["cpu_baseline"], model = cpu_baseline(scene, args.workload, 12.0 if world == 1 else 1.5)
            if world > 1:
                out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = None
        # dominant kernel, rank 0's launches: k_bounce<false> (bounces 1..4), or in split mode (scenes with
        # KD-trees) the walker k_walk, which runs once per bounce
        # the events bracket the launches of every 4th step (ctx.enable_timing(4)): per-launch work = totals / all launches
        timed_steps = (args.steps + 3) // 4
        launch_scale = args.steps / timed_steps if timed_steps else 1.0
        n_launch, ms = timing["k_bounce<false>"]
        n0, ms0 = timing["k_bounce<true>"]
        nw, msw = timing["k_walk"]
        nsp, mssp = timing["k_shade_parked"]
        nf, msf = timing["k_bounce_fused"]
        pmc = load_pmc_traffic(args.workload)
        if nf and msf >= ms and model is not None and not nw:
            # A closed scene (Room): all five bounces of a chain run in k_bounce_fused with the rays in registers.  The scene
            # tables (a few KB) are cache resident and there is no ray queue: HBM sees 12 B of radiance per path -> VALU-bound.
            all_launches = nf * launch_scale
            chains = (nf + n0) * launch_scale                      # launch chains in the timed region (fused or not), equal in size
            rays_launch = rays / world / chains                    # rays a chain casts = rays per k_bounce_fused launch
            avg_ms = msf / nf
            tflops = model["flops_per_ray"] * rays_launch / (avg_ms * 1e-3) / 1e12
            hbm_bytes = 12.0 * (paths / world) / chains
            traffic = pmc.get("k_bounce_fused")
            out["roofline"] = {
                "bound": "valu", "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP32_TFLOPS,
                "kernel": "k_bounce_fused", "traffic": traffic, "traffic_source": pmc.get("source"),
                "avg_launch_ms": avg_ms, "launches": all_launches, "timed_launches": nf, "rays_per_launch": rays_launch,
                "flops_per_ray": model["flops_per_ray"],
                "hbm_algorithmic_bytes_per_launch": hbm_bytes, "hbm_algorithmic_GBs": hbm_bytes / (avg_ms * 1e-3) / 1e9,
                "hbm_measured_GBs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                "hbm_measured_frac": (traffic / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if traffic else None, "hbm_peak_GBs": PEAK_HBM_GBS,
                "unfused_launches_in_timed_region": n_launch * launch_scale,
                "note": ("no MFMA on this path (nothing is a dense contraction).  k_bounce_fused generates the primary rays and runs all "
                         "five casts and shadings of a path with the ray in registers: the scene tables (a few KB) are cache resident and "
                         "the only HBM traffic is the 12 B of radiance a path writes when it ends (hbm_algorithmic_*), so the kernel is "
                         "bound by VALU issue.  peak = fp32 vector peak with packed FMA; hit arithmetic is built unfused "
                         "(-ffp-contract=off, bit-exact ids), so 78.6 T op/s is the reachable ceiling.  flops = the REFERENCE's "
                         "operation count per ray (it tests every node; tests/golden/flop_costs.json x the oracle's counts); the "
                         "kernel's conservative cull skips most of that work, valu_busy_frac is in profiles/*_pmc.json"),
            }
        elif n_launch and model is not None and not nw:
            # Room / materials: the whole scene is cache resident, HBM sees the ray queue only -> VALU-bound.
            all_launches = n_launch * launch_scale
            rays_next = (rays - paths) / world / all_launches      # rays cast per k_bounce<false> launch
            avg_ms = ms / n_launch
            tflops = model["flops_per_ray"] * rays_next / (avg_ms * 1e-3) / 1e12
            # algorithmic HBM bytes of one launch: every ray is read (52 B); a survivor is written back (52 B), a path
            # that ends writes its radiance (12 B).  Survivors of launch b = rays of launch b + 1 (0 after the last).
            rays_in = (rays - paths) / world                        # all k_bounce<false> launches together
            paths_w = paths / world
            # rays entering bounces 2..4 are the survivors written by bounces 1..3; bounce 0's survivors (= rays of
            # bounce 1) are written by k_bounce<true>, not by this kernel
            queue_bytes = (52.0 * rays_in + 52.0 * max(0.0, rays_in - rays_in / (PTK_DEPTH - 1)) + 12.0 * paths_w) / all_launches
            gbs = queue_bytes / (avg_ms * 1e-3) / 1e9
            measured = (pmc["k_bounce<false>"] / (avg_ms * 1e-3) / 1e9) if pmc.get("k_bounce<false>") else None
            out["roofline"] = {
                "bound": "valu", "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP32_TFLOPS,
                "kernel": "k_bounce<false>", "traffic": pmc.get("k_bounce<false>"), "traffic_source": pmc.get("source"),
                "avg_launch_ms": avg_ms, "launches": all_launches, "timed_launches": n_launch, "rays_per_launch": rays_next,
                "first_bounce_avg_ms": (ms0 / n0) if n0 else None,
                "flops_per_ray": model["flops_per_ray"],
                "hbm_algorithmic_bytes_per_launch": queue_bytes, "hbm_algorithmic_GBs": gbs, "hbm_measured_GBs": measured,
                "hbm_measured_frac": (measured / PEAK_HBM_GBS) if measured else None, "hbm_peak_GBs": PEAK_HBM_GBS,
                "note": ("no MFMA on this path (nothing is a dense contraction); the scene tables (a few KB) are cache resident, so HBM "
                         "carries the ray queue only (hbm_algorithmic_*: 52 B per ray read, 52 B per survivor or 12 B per ended path "
                         "written; an estimate from the counters: survivors of a launch = rays of the next) and the kernel is bound by "
                         "VALU issue.  peak = fp32 vector peak with packed FMA; hit arithmetic is built unfused "
                         "(-ffp-contract=off, bit-exact ids), so 78.6 T op/s is the reachable ceiling.  flops = the REFERENCE's "
                         "operation count per ray (it tests every node; tests/golden/flop_costs.json x the oracle's counts); the "
                         "kernel's conservative cull skips most of that work, valu_busy_frac is in profiles/*_pmc.json"),
            }
        elif nw and model is not None:
            # Scenes with KD-trees (Mario, config 5): one bounce = k_bounce_split (spheres, one-leaf meshes, parking) +
            # k_walk (KD-trees of the parked rays) + k_shade_parked; k_walk dominates.  SURVEY 8d: HBM-bound line from
            # the build's own pruned walk (V nodes, T triangles per ray: oracle walk model 1).
            rays_launch = rays / world / (nw * launch_scale)
            walk_ms = msw / nw
            V, T = model["kd_nodes_per_ray"], model["triangles_per_ray"]
            touched = (64.0 * V + 64.0 * T) * rays_launch          # bytes the walk requests (mostly cache hits)
            tables = float(model_tables["kd_bytes"])               # what HBM must deliver at least once per launch ...
            parked = 112.0 * rays_launch                            # ... plus the parked record in (76 B) and out (36 B)
            compulsory = min(touched, tables) + parked
            traffic = pmc.get("k_walk")
            out["roofline"] = {
                "bound": "hbm", "achieved": compulsory / (walk_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": compulsory / (walk_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "kernel": "k_walk", "traffic": traffic, "traffic_source": pmc.get("source"),
                "avg_launch_ms": walk_ms, "launches": nw, "rays_per_launch": rays_launch,
                "kd_nodes_per_ray": V, "triangles_per_ray": T, "texels_per_ray": model["texels_per_ray"],
                "ref_kd_nodes_per_ray": model["ref_kd_nodes_per_ray"], "ref_triangles_per_ray": model["ref_triangles_per_ray"],
                "algorithmic_bytes_per_launch": compulsory, "tree_table_bytes": tables,
                "requested_bytes_per_launch": touched, "requested_GBs": touched / (walk_ms * 1e-3) / 1e9,
                "requested_frac_of_l2_peak": touched / (walk_ms * 1e-3) / 1e9 / PEAK_L2_GBS, "l2_peak_GBs": PEAK_L2_GBS,
                "hbm_measured_GBs": (traffic / (walk_ms * 1e-3) / 1e9) if traffic else None,
                "bounce_kernels_avg_ms": {"k_bounce_split": (ms + ms0) / max(1, n_launch + n0), "k_walk": walk_ms,
                                          "k_shade_parked": mssp / nsp if nsp else None},
                "note": ("achieved = algorithmic HBM bytes per k_walk launch / its average duration.  Algorithmic bytes = what HBM has to "
                         "deliver at least once: min(bytes the pruned walk requests = 64 B x (V nodes + T triangles) per ray, size of "
                         "the tree tables) + the parked record in and out (112 B per ray).  requested_* is the walk's request stream, "
                         "served mostly by L2 (34.5 TB/s aggregate) and the Infinity Cache; traffic = PMC HBM bytes per launch "
                         "(profiles/): traffic / algorithmic > 1 = re-reads after cache eviction.  The walker is limited by the issue "
                         "rate of its divergent walk (valu_busy_frac in profiles/*_pmc.json), not by HBM: see DESIGN.md.  ref_* = the "
                         "reference's unpruned walk (every box-hit node), for comparison only"),
            }
        else:
            out["roofline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
