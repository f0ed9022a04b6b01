#!/usr/bin/env python3
"""bench.py — throughput of the radiance inner loop on MI355X (BASELINE.json metric: Msamples/s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): models/CornellBox.obj, 1920x1080,
bounces = 2, f64.  A "step" is one wavefront pass of the hot path over `--spp` frames of the whole
image (generate -> [extend -> shade] x 3 -> accumulate).  With N > 1 the work is split by
sample ranges: every rank renders the full image for its own, disjoint frame indices (fixed
per-GPU work => weak scaling) and the f64 accumulation buffers are summed onto rank 0 with one
RCCL reduce at the end of the timed region (partitioning 2).  `--partition tiles` is
the other form (§8e partitioning 1): row bands of the one image, every rank the same frames, one
RCCL all-gather at the end (fixed total work => strong scaling).  The scene and the accumulation
buffer are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0; see DESIGN.md "Measurement" for the roofline/cpu_baseline fields.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spp", type=int, default=64, help="frames per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="CornellBox", help="models/<scene>.obj, or 'heightfield' (SURVEY config 5)")
    ap.add_argument("--cells", type=int, default=2236, help="heightfield resolution (2236 -> 9,999,392 triangles)")
    ap.add_argument("--bounces", type=int, default=2)
    ap.add_argument("--partition", choices=("samples", "tiles"), default="samples",
                    help="N > 1: 'samples' = every rank renders the whole image for its own frames, one RCCL reduce "
                         "(weak scaling, the measured default); 'tiles' = row bands of the one image, one RCCL "
                         "all-gather (SURVEY.md §8e partitioning 1, strong scaling)")
    ap.add_argument("--nee", action="store_true", help="next-event estimation (shadow rays) on")
    ap.add_argument("--textures", action="store_true", help="load map_Kd textures (BASELINE configs[3]: --scene Test)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    args = ap.parse_args()

    import pathtracer_amd as pt  # loads the HIP library (and pins the process to one HIP runtime)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs for a 1-GPU box (never set by the driver): all ranks on one device, gloo
    # instead of RCCL (RCCL refuses two ranks on one GPU).  The measured configuration is nccl.
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    if "PT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PT_BENCH_DEVICE"])
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    n_gpus = world
    if args.gpus != n_gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run for N>1",
              file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.scene == "heightfield":
        from pathtracer_amd.synthetic import heightfield_box
        data = heightfield_box(args.cells)
    else:
        data = pt.scene_data(args.scene, textures=args.textures)
    scene = pt.Scene(data, device=local_rank)
    cfg = pt.Config(width=args.width, height=args.height, bounces=args.bounces, seed=1, spp_per_batch=args.spp,
                    nee=args.nee)
    accum = torch.zeros((args.height, args.width, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    npix = args.width * args.height

    tiles = world > 1 and args.partition == "tiles"
    if tiles and args.height % world:
        raise SystemExit(f"--partition tiles needs --height divisible by the number of ranks ({args.height} % {world})")
    band = args.height // world if tiles else args.height
    y0 = rank * band if tiles else 0

    def step(i):
        if tiles:   # every rank the same frames, its own rows
            scene.render_device(cfg, accum.data_ptr(), i * args.spp, args.spp, tile=(0, y0, args.width, y0 + band),
                                stream=stream)
        else:       # frames [base, base + spp): disjoint across ranks and steps
            scene.render_device(cfg, accum.data_ptr(), (i * world + rank) * args.spp, args.spp, stream=stream)

    def exchange():   # the job's one exchange, RCCL over xGMI (gloo through the host in rehearsals)
        if tiles:
            mine = accum[y0:y0 + band].contiguous()
            if backend == "nccl":
                dist.all_gather_into_tensor(accum.view(-1), mine.view(-1))
            else:
                parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
                dist.all_gather(parts, mine.cpu())
                accum.copy_(torch.cat(parts, dim=0))
        elif backend == "nccl":
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        else:
            host = accum.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            accum.copy_(host)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    if world > 1 and backend == "nccl":
        # first use of a collective builds channels and loads its kernel: not part of a step
        if tiles:
            dist.all_gather_into_tensor(torch.zeros_like(accum).view(-1), torch.zeros_like(accum[y0:y0 + band]).view(-1))
        else:
            dist.reduce(torch.zeros_like(accum), dst=0, op=dist.ReduceOp.SUM)
    fence()
    scene.reset_stats()
    scene.set_profiling(1)  # HIP events around every kernel launch of the timed region
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if world > 1:
        exchange()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = scene.stats()
    scene.set_profiling(0)

    samples_per_rank = (band * args.width) * args.spp * args.steps
    rays_t = torch.tensor([float(st["rays"])], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
    total_rays = float(rays_t.item())
    total_samples = samples_per_rank * world
    value = total_samples / dt / 1e6

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (rank 0's launches, HIP-event durations) ----------
        kern = st["kernel"]
        dom = max(kern, key=lambda k: kern[k]["ms"])
        total_kernel_ms = sum(k["ms"] for k in kern.values())
        # V (child boxes slab-tested) and T (triangles tested) per ray: one instrumented step
        scene.reset_stats()
        scene.set_profiling(2)
        step(args.warmup + args.steps)
        torch.cuda.synchronize(dev)
        st2 = scene.stats()
        scene.set_profiling(0)
        rays2 = max(1, st2["rays"])
        V = st2["nodes_visited"] / rays2
        T = st2["tris_tested"] / rays2
        # the same for the primary rays alone (the same frames with bounces = 0)
        scene.reset_stats()
        scene.set_profiling(2)
        cfg0 = pt.Config(width=args.width, height=args.height, bounces=0, seed=1, spp_per_batch=args.spp, nee=args.nee)
        scratch = torch.zeros_like(accum)
        scene.render_device(cfg0, scratch.data_ptr(), ((args.warmup + args.steps) * world + rank) * args.spp, args.spp,
                            stream=stream)
        torch.cuda.synchronize(dev)
        st0 = scene.stats()
        scene.set_profiling(0)
        del scratch
        V0 = st0["nodes_visited"] / max(1, st0["rays"])
        T0 = st0["tris_tested"] / max(1, st0["rays"])
        # Algorithmic bytes (DESIGN.md "Measurement").  SURVEY.md §8d prices a ray at
        # B_ray = 32 V + 36 T + 128: V child boxes (32 B each), T triangles (36 B) and one path
        # record written by the producer and read by the consumer.  Per kernel, with this build's
        # records (path record 76 B = o, d, T as f64 + slot, +8 B pdf with NEE; hit record 12 B):
        #   extend      primary ray: 12 written; secondary ray: 48 read + 12 written
        #   shade       per ray 12 read; secondary ray: +76 read; per continuing path 76 written;
        #               per sample 1 bit, per sample that gathered radiance 24 written
        #   path        (extend+shade fused) secondary ray: 76 read; continuing path: 76 written;
        #               per sample 1 bit, per lit sample 24 written
        #   accumulate  per sample 1 bit, per lit sample 24 read, per pixel and launch 48 (accum
        #               read-modify-write)
        # (the radiance of a sample is stored only if it gathered any: DESIGN.md "sparse radiance";
        # `lit` is counted by the instrumented step)
        # The scene terms 32 V + 36 T are added to the traversing kernel only when the scene does
        # not fit one XCD's L2 (config 5); for a cache-resident scene they are reported as L2-served.
        scene_bytes = data.n_tris * 48 + len(scene.bvh()[0]) * 64
        resident = scene_bytes < (4 << 20)  # one XCD's L2
        scene_ray = 32.0 * V + 36.0 * T
        rec, hit = 76.0, 12.0
        P = float(samples_per_rank)                 # primary rays of the timed region
        lit = P * st2["lit_samples"] / max(1, st2["samples"])   # samples that gathered radiance
        # primary rays actually traced (those through pixels outside the scene's screen bounds are not)
        P_traced = P * st0["rays"] / max(1, st0["samples"])
        S = max(0.0, float(st["rays"]) - P_traced)  # secondary rays == records written
        scene_all = float(st["rays"]) * scene_ray
        per_kind = {
            "extend": P * hit + S * (48.0 + hit),
            "shade": (P + S) * hit + S * rec + S * rec + P / 8.0 + lit * 24.0,
            "path": S * rec + S * rec + P / 8.0 + lit * 24.0,
            "accumulate": P / 8.0 + lit * 24.0 + float(npix) * 48.0 * max(1, kern["accumulate"]["launches"]),
            "shadow": float(st["shadow_rays"]) * (56.0 + 24.0 + 4.0),
            "generate": 0.0,
        }
        # which kernel walked the tree for which rays: the fused path kernel every bounce of a shallow
        # tree, or only the primary bounce of a deep one (the extend kernel then has the secondary rays)
        if kern["path"]["launches"] and kern["extend"]["launches"]:
            scene_primary = P_traced * (32.0 * V0 + 36.0 * T0)
            scene_of = {"path": scene_primary, "extend": scene_all - scene_primary}
            # deep tree: path = primary bounce only (writes the records of the secondary rays), extend and
            # shade = the secondary rays (records written by shade: the few survivors of the cull, ignored)
            per_kind["path"] = S * rec + P / 8.0 + lit * 24.0
            per_kind["extend"] = S * (48.0 + hit)
            per_kind["shade"] = S * (hit + rec) + lit * 24.0
        elif kern["path"]["launches"]:
            scene_of = {"path": scene_all}
        else:
            scene_of = {"extend": scene_all}
        launches = max(1, kern[dom]["launches"])
        avg_ms = kern[dom]["ms"] / launches
        hbm_bytes = per_kind[dom] + (0.0 if resident else scene_of.get(dom, 0.0))
        l2_bytes = scene_of.get(dom, 0.0) if resident else 0.0
        bytes_per_launch = hbm_bytes / launches
        gbs = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        achieved = gbs(bytes_per_launch, avg_ms)
        whole = sum(per_kind[k] for k in kern if kern[k]["launches"]) + (0.0 if resident else scene_all)
        # HBM traffic per launch of the dominant kernel from the rocprofv3 PMC passes of this same
        # command (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE; profiles/traffic.json, scripts/pmc_traffic.py)
        traffic, pmc = None, {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)
            key = f"{args.scene}_{args.width}x{args.height}_spp{args.spp}_b{args.bounces}"
            pmc = tj.get(key, {}).get(dom, {})
            traffic = pmc.get("traffic_bytes_per_launch")
        except (OSError, ValueError):
            pass
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            # from the committed rocprofv3 PMC passes of this command (profiles/traffic.json), not live:
            "pmc": {k: pmc.get(k) for k in ("valu_busy", "lane_utilisation", "waves_per_simd", "l2_hit")} if pmc else None,
            "avg_launch_ms": round(avg_ms, 4), "launches": launches,
            "algorithmic_bytes_per_launch": round(bytes_per_launch),
            "scene_cache_resident": resident, "scene_bytes": scene_bytes,
            "l2_served_GBs": round(gbs(l2_bytes / launches, avg_ms), 2),
            "nodes_per_ray": round(V, 3), "tris_per_ray": round(T, 3),
            "nodes_per_primary_ray": round(V0, 3), "tris_per_primary_ray": round(T0, 3),
            "rays_per_sample": round(st["rays"] / max(1, samples_per_rank), 4),
            "lit_samples": round(lit / P, 4), "primary_rays_traced": round(P_traced / P, 4),
            "kernel_ms": {k: round(v["ms"], 3) for k, v in kern.items()},
            "kernel_launches": {k: v["launches"] for k, v in kern.items()},
            "kernel_share": {k: round(v["ms"] / total_kernel_ms, 4) for k, v in kern.items()} if total_kernel_ms else {},
            "kernel_GBs": {k: round(gbs(per_kind[k] + (0.0 if resident else scene_of.get(k, 0.0)), v["ms"]), 1)
                           for k, v in kern.items() if v["launches"]},
            "whole_path_GBs": round(gbs(whole, total_kernel_ms), 2),
        }
        # ---- CPU baseline: the oracle (C++ restatement, libm, f64) on this box's host cores ----
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import numpy as np
            import oracle_lib as O
            oracle = O.OracleScene(data)
            threads = max(1, min(O.lib.orc_hardware_threads(), 16, len(os.sched_getaffinity(0))))
            acc = np.zeros((args.height, args.width, 3))
            cpu_cfg = pt.Config(width=args.width, height=args.height, bounces=args.bounces, seed=1, nee=args.nee)
            n_frames, c0 = 0, time.perf_counter()
            while True:
                oracle.render(cpu_cfg, acc, sample_begin=n_frames, sample_count=1, threads=threads, use_bvh=True)
                n_frames += 1
                if time.perf_counter() - c0 >= args.cpu_seconds:
                    break
            cdt = time.perf_counter() - c0
            cpu = {"value": round(npix * n_frames / cdt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                   "sample": f"{n_frames} frames (spp) of the same {args.width}x{args.height} {args.scene} workload, "
                             f"{cdt:.1f} s; C++ f64 restatement of the Rust loop (rustc absent, SURVEY F1)"}
        out = {
            "metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": "f64",
            "data": (f"synthetic heightfield box, {data.n_tris} triangles, seed 12345" if args.scene == "heightfield"
                     else f"reference scene models/{args.scene}.obj") + ", counter-based RNG (no dataset)",
            "config": {"workload": f"{args.scene}{'' if args.scene == 'heightfield' else '.obj'} {args.width}x{args.height} "
                                   f"bounces={args.bounces}{', NEE' if args.nee else ''}{', textures' if args.textures else ''}, {args.spp} spp per step per GPU"
                                   + (" (BASELINE.json configs[1])" if args.scene == "CornellBox" else ""),
                       "triangles": data.n_tris,
                       "exact_culls": {k: os.environ.get(k, "1") != "0" for k in ("PT_SCREEN_CULL", "PT_LIGHT_CULL")},
                       "partitioning": ("row-band tiles + one RCCL all-gather" if tiles else "sample ranges + one RCCL reduce") if world > 1 else "single GPU",
                       "samples_per_step": band * args.width * args.spp * world},
            "mrays_per_s": round(total_rays / dt / 1e6, 3),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1) if cpu["value"] else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
