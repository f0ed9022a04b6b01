#!/usr/bin/env python3
"""bench.py — throughput of the radiance inner loop on MI355X (BASELINE.json metric: Msamples/s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): models/CornellBox.obj, 1920x1080,
bounces = 2, f64.  A "step" is one wavefront pass of the hot path over `--spp` frames of the whole
image (generate -> [extend -> shade] x 3 -> accumulate).  With N > 1 the work is split by
sample ranges: every rank renders the full image for its own, disjoint frame indices (fixed
per-GPU work => weak scaling) and the f64 accumulation buffers are summed onto rank 0 with one
RCCL reduce at the end of the timed region (partitioning 2).  The scene and the
accumulation buffer are resident in HBM before the timed region starts.

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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spp", type=int, default=8, help="frames per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="CornellBox", help="models/<scene>.obj, or 'heightfield' (SURVEY config 5)")
    ap.add_argument("--cells", type=int, default=2236, help="heightfield resolution (2236 -> 9,999,392 triangles)")
    ap.add_argument("--bounces", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    args = ap.parse_args()

    import pathtracer_amd as pt  # loads the HIP library (and pins the process to one HIP runtime)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
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
        data = pt.scene_data(args.scene)
    scene = pt.Scene(data, device=local_rank)
    cfg = pt.Config(width=args.width, height=args.height, bounces=args.bounces, seed=1, spp_per_batch=args.spp)
    accum = torch.zeros((args.height, args.width, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    npix = args.width * args.height

    def step(i):
        # frames [base, base + spp): disjoint across ranks and steps
        base = (i * world + rank) * args.spp
        scene.render_device(cfg, accum.data_ptr(), base, args.spp, stream=stream)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    scene.reset_stats()
    scene.set_profiling(1)  # HIP events around every kernel launch of the timed region
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if world > 1:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI: the job's one exchange
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = scene.stats()
    scene.set_profiling(0)

    samples_per_rank = npix * args.spp * args.steps
    rays_t = torch.tensor([float(st["rays"])], dtype=torch.float64, device=dev)
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
        # SURVEY.md §8d per-ray figure: B_ray = 32 V + 36 T + 128.  The 128 B queue record always
        # crosses HBM; the 32 V + 36 T scene bytes do only when the scene does not fit the caches
        # (config 5).  For a cache-resident scene they are L2/L1-served and priced separately.
        scene_bytes = data.n_tris * 48 + len(scene.bvh()[0]) * 64
        resident = scene_bytes < (4 << 20)  # one XCD's L2
        scene_ray = 32.0 * V + 36.0 * T
        b_ray_hbm = 128.0 + (0.0 if resident else scene_ray)
        b_ray = 128.0 + scene_ray
        rays_per_sample = st["rays"] / max(1, samples_per_rank)
        launches = max(1, kern[dom]["launches"])
        avg_ms = kern[dom]["ms"] / launches
        if dom in ("extend", "shadow"):
            units = st["rays"] / max(1, kern["extend"]["launches"])  # rays per extend launch
            bytes_per_launch, all_bytes = units * b_ray_hbm, units * b_ray
        else:
            units = samples_per_rank / launches
            bytes_per_launch = units * (rays_per_sample * b_ray_hbm + 24.0)
            all_bytes = units * (rays_per_sample * b_ray + 24.0)
        gbs = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        achieved = gbs(bytes_per_launch, avg_ms)
        b_sample = rays_per_sample * b_ray_hbm + 24.0
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "avg_launch_ms": round(avg_ms, 4), "launches": launches,
            "algorithmic_bytes_per_launch": round(bytes_per_launch),
            "scene_cache_resident": resident, "scene_bytes": scene_bytes,
            "l2_served_GBs": round(gbs(all_bytes - bytes_per_launch, avg_ms), 2) if resident else 0.0,
            "nodes_per_ray": round(V, 3), "tris_per_ray": round(T, 3),
            "bytes_per_ray_hbm": round(b_ray_hbm, 1), "bytes_per_ray_with_scene": round(b_ray, 1),
            "kernel_ms": {k: round(v["ms"], 3) for k, v in kern.items()},
            "kernel_share": {k: round(v["ms"] / total_kernel_ms, 4) for k, v in kern.items()} if total_kernel_ms else {},
            "whole_path_GBs": round(gbs(b_sample * samples_per_rank, total_kernel_ms), 2),
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
            cpu_cfg = pt.Config(width=args.width, height=args.height, bounces=args.bounces, seed=1)
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
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": (f"synthetic heightfield box, {data.n_tris} triangles, seed 12345" if args.scene == "heightfield"
                     else f"reference scene models/{args.scene}.obj") + ", counter-based RNG (no dataset)",
            "config": {"workload": f"{args.scene}{'' if args.scene == 'heightfield' else '.obj'} {args.width}x{args.height} "
                                   f"bounces={args.bounces}, {args.spp} spp per step per GPU"
                                   + (" (BASELINE.json configs[1])" if args.scene == "CornellBox" else ""),
                       "triangles": data.n_tris,
                       "partitioning": "sample ranges + one RCCL reduce" if world > 1 else "single GPU",
                       "samples_per_step": npix * args.spp * world},
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
