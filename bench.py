#!/usr/bin/env python3
"""bench.py — throughput of the radiance inner loop on MI355X (BASELINE.json metric: Msamples/s).

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1]): models/CornellBox.obj, 1920x1080,
1024 spp, bounces = 2, f64.  A "step" is one pass of the hot path over the config as named: all 1024
frames of the image (generate -> [extend -> shade] x 3 -> accumulate), rendered as 16 batches of 64
frames.  The scene and the accumulation buffer are resident in HBM before the timed region starts.

With N > 1 and no launcher in the environment, bench.py starts its own N ranks (a child
`python -m torch.distributed.run`, before this process touches the GPU).  The work is cut by
pathtracer_amd/dist.py.  Default `--partition tiles`: the ONE image of the N = 1 workload, every rank the
row groups g % N == rank of it (fixed total work => "scaling": "strong"; `samples_per_step` equals the
1-GPU line's), and the rows are gathered to rank 0 INSIDE every step - an image is done when rank 0 holds
it - with the exchange's share reported as `exchange_ms_per_step`.  After the headline run the same ranks
time `--partition samples` (every rank the whole image for its own 1024 frames per step, partial sums added
on rank 0 in rank order once per step: fixed per-GPU work, weak scaling) and both appear under "partitions".

At N = 1 the other BASELINE configs (glossy_planes 1080p, textured Test.obj 4K, the 10 M-triangle
heightfield 4K) are timed for a few steps each after the headline and reported under "configs", each
with its own roofline; `--configs none` skips them.

Prints ONE JSON line on rank 0; see DESIGN.md "Measurement" for the roofline / cpu_baseline fields.
`value` counts every sample of the frame; `value_on_screen` only those whose primary ray was generated (the
scene's screen bounds skip the others), `value_no_culls` is a short run of the same workload with both exact
culls off (every ray the reference's render() would trace is traced).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MAX_CLOCK_HZ = 2.4e9       # MI355X_MICROARCH.md: max clock
N_SIMD = 256 * 4           # 256 CUs x 4 SIMDs
FP64_VALU_PEAK_TFLOPS = 78.6   # spec vector FP64 (FMA = 2 flops); the reference's arithmetic has no FMA: half of it

# The BASELINE.json configs that run on a GPU.  `spp` = frames per step (the config as named), `batch` =
# frames in flight per wavefront pass.
WORKLOADS = {
    "CornellBox": dict(scene="CornellBox", width=1920, height=1080, spp=1024, batch=64, label="BASELINE.json configs[1]"),
    "glossy_planes": dict(scene="glossy_planes", width=1920, height=1080, spp=256, batch=64, label="BASELINE.json configs[2]"),
    "Test_textured": dict(scene="Test", textures=True, width=3840, height=2160, spp=64, batch=16, label="BASELINE.json configs[3]"),
    "heightfield": dict(scene="heightfield", width=3840, height=2160, spp=4096, batch=64, label="BASELINE.json configs[4]"),
}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pmc_key(wl, bounces, nee):
    return (f"{wl['scene']}{'_tex' if wl.get('textures') else ''}{'_nee' if nee else ''}_{wl['width']}x{wl['height']}"
            f"_batch{wl['batch']}_b{bounces}")


def load_pmc():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def pmc_is_stale(entry):
    """The counters of profiles/pmc.json belong to the kernel sources they were collected on (their git blob
    hashes, scripts/pmc_digest.py): stale when the working tree's differ, or when none were recorded."""
    from pathtracer_amd.buildinfo import kernel_source_hashes
    return entry.get("_sources") != kernel_source_hashes()


class Workload:
    """One scene + config on this rank's GPU, with the step / instrumentation / roofline logic."""

    def __init__(self, pt, torch, wl, args, dev, local_rank, part=None, data=None, options=None):
        self.pt, self.torch, self.wl, self.args, self.dev, self.part = pt, torch, wl, args, dev, part
        if data is not None:
            self.data = data
        elif wl["scene"] == "heightfield":
            from pathtracer_amd.synthetic import heightfield_box
            self.data = heightfield_box(args.cells)
        else:
            self.data = pt.scene_data(wl["scene"], textures=bool(wl.get("textures")))
        self.scene = pt.Scene(self.data, device=local_rank, options=options)
        self.w, self.h, self.spp, self.batch = wl["width"], wl["height"], wl["spp"], wl["batch"]
        if part is not None and part.mode != "samples":
            # a rank renders 1 / world of the image: world x the frames per pass keep the paths in flight (and the
            # workspace) what they are at N = 1; the image is bitwise independent of the batch size (tested)
            self.batch = min(self.spp, self.batch * part.world)
        self.exchange_events = []
        self.cfg = pt.Config(width=self.w, height=self.h, bounces=args.bounces, seed=1, spp_per_batch=self.batch, nee=args.nee)
        self.accum = torch.zeros((self.h, self.w, 3), dtype=torch.float64, device=dev)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.world = part.world if part is not None else 1
        self.rank = part.rank if part is not None else 0

    def render_fn(self, accum, tile, s0, n, rows):
        self.scene.render_device(self.cfg, accum.data_ptr(), s0, n, tile=tile, stream=self.stream, rows=rows)

    def step(self, i):
        """Step i.  One GPU: the next spp frames are added to the running sums.  N > 1: one complete image per
        step - the ranks render their shares into zeroed sums and hand them to rank 0 (Partition.exchange), inside
        the step; HIP events on the launch stream bracket the exchange."""
        if self.part is None:
            self.render_fn(self.accum, None, i * self.spp, self.spp, None)
            return
        self.accum.zero_()
        if self.part.mode == "samples":   # the job's step = world x spp frames, spp of them here
            self.part.render(self.render_fn, self.accum, i * self.spp * self.world, self.spp * self.world)
        else:                             # every rank the same spp frames, its own rows
            self.part.render(self.render_fn, self.accum, i * self.spp, self.spp)
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        a.record()
        self.part.exchange(self.accum)
        b.record()
        self.exchange_events.append((a, b))

    def exchange_ms(self):
        """Sum over the recorded steps (call after a synchronize), and forget them."""
        ms = sum(a.elapsed_time(b) for a, b in self.exchange_events)
        self.exchange_events = []
        return ms

    def pixels(self):
        return self.part.pixels() if (self.part is not None and self.part.mode != "samples") else self.w * self.h

    # ---- V, T and lit samples per ray from one instrumented batch (profiling level 2) -------------------
    def instrument(self, first_frame):
        pt, torch, scene = self.pt, self.torch, self.scene
        out = {}
        for label, bounces in (("all", self.args.bounces), ("primary", 0)):
            scene.reset_stats()
            scene.set_profiling(2)
            cfg = pt.Config(width=self.w, height=self.h, bounces=bounces, seed=1, spp_per_batch=self.batch, nee=self.args.nee)
            scratch = torch.zeros_like(self.accum)
            scene.render_device(cfg, scratch.data_ptr(), first_frame, self.batch, stream=self.stream)
            torch.cuda.synchronize(self.dev)
            out[label] = scene.stats()
            scene.set_profiling(0)
            del scratch
        return out["all"], out["primary"]

    # ---- roofline of the dominant kernel -----------------------------------------------------------------
    def roofline(self, st, samples_per_rank, first_free_frame):
        args, data, scene = self.args, self.data, self.scene
        npix = self.w * self.h
        kern = st["kernel"]
        dom = max(kern, key=lambda k: kern[k]["ms"])
        total_kernel_ms = sum(k["ms"] for k in kern.values())
        st2, st0 = self.instrument(first_free_frame)
        rays2 = max(1, st2["rays"])
        V, T = st2["nodes_visited"] / rays2, st2["tris_tested"] / rays2
        V0, T0 = st0["nodes_visited"] / max(1, st0["rays"]), st0["tris_tested"] / max(1, st0["rays"])
        # Algorithmic bytes (DESIGN.md "Measurement").  SURVEY.md §8d prices a ray at
        # B_ray = 32 V + 36 T + 128: V child boxes (32 B each), T triangles (36 B) and one path
        # record written by the producer and read by the consumer.  Per kernel, with this build's
        # records (path record 76 B = o, d, T as f64 + slot, +8 B pdf with NEE; hit record 12 B):
        #   extend      primary ray: 12 written; secondary ray: 48 read + 12 written
        #   shade       per ray 12 read; secondary ray: +76 read; per continuing path 76 written;
        #               per sample 1 bit, per sample that gathered radiance 24 written
        #   path        (extend+shade fused) secondary ray: 76 read; continuing path: 76 written;
        #               per sample 1 bit, per lit sample 24 written.  Where bounces 0 and 1 run in one launch
        #               (pt_stats.two_bounce) the rays of bounce 1 never exist as records: only the rays of
        #               bounce >= 2 are written and read
        #   accumulate  per sample 1 bit, per lit sample 24 read, per pixel and launch 48 (accum
        #               read-modify-write)
        # The scene terms 32 V + 36 T are HBM bytes of the traversing kernel only when the scene does
        # not fit one XCD's L2 (config 5); for a cache-resident scene they are reported as L2/LDS-served.
        scene_bytes = st["scene_bytes"]
        resident = scene_bytes < (4 << 20)  # one XCD's L2
        scene_ray = 32.0 * V + 36.0 * T
        rec, hit = 76.0, 12.0
        P = float(samples_per_rank)                 # primary rays of the timed region
        lit = P * st2["lit_samples"] / max(1, st2["samples"])   # samples that gathered radiance
        P_traced = P * st0["rays"] / max(1, st0["samples"])     # primary rays actually traced (screen bounds)
        S = max(0.0, float(st["rays"]) - P_traced)  # secondary rays == records written
        # ... minus the rays of bounce 1 where bounces 0 and 1 run in one launch: they stay in registers (pt_stats.unqueued_rays)
        S_rec = max(0.0, S - float(st.get("unqueued_rays", 0)))
        scene_all = float(st["rays"]) * scene_ray
        per_kind = {
            "extend": P * hit + S * (48.0 + hit),
            "shade": (P + S) * hit + S * rec + S * rec + P / 8.0 + lit * 24.0,
            "path": S_rec * rec + S_rec * rec + P / 8.0 + lit * 24.0,
            "accumulate": P / 8.0 + lit * 24.0 + float(npix) * 48.0 * max(1, kern["accumulate"]["launches"]),
            "shadow": float(st["shadow_rays"]) * (56.0 + 24.0 + 4.0),
            "generate": 0.0,
        }
        if kern["path"]["launches"] and kern["extend"]["launches"]:
            # deep tree: path = primary bounce only (writes the records of the secondary rays), extend and
            # shade = the secondary rays (records written by shade: the few survivors of the cull, ignored)
            scene_primary = P_traced * (32.0 * V0 + 36.0 * T0)
            scene_of = {"path": scene_primary, "extend": scene_all - scene_primary}
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
        # rocprofv3 PMC passes of this same workload, committed under profiles/ (scripts/pmc_collect.sh ->
        # scripts/pmc_digest.py -> profiles/pmc.json): per launch of each kernel kind, HBM traffic
        # ((FETCH_SIZE x 2 on gfx950) + WRITE_SIZE), VALU busy cycles, VALU instructions, lane cycles
        pmc_entry = load_pmc().get(pmc_key(self.wl, args.bounces, args.nee), {})
        pmc = pmc_entry.get(dom, {})
        pmc_stale = pmc_is_stale(pmc_entry) if pmc else None
        hbm = {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc.get("traffic_bytes_per_launch"),
            "algorithmic_bytes_per_launch": round(bytes_per_launch),
        }
        if hbm["traffic"]:
            # what the memory system actually moved (counters) over the same live duration: above the algorithmic
            # bytes on re-reads, below them where the caches serve part (the deep tree's upper levels)
            hbm["traffic_GBs"] = round(gbs(hbm["traffic"], avg_ms), 2)
            hbm["traffic_frac"] = round(hbm["traffic_GBs"] / HBM_PEAK_GBS, 5)
            if not resident and hbm["traffic"] < bytes_per_launch:
                # SURVEY §8d prices every node and triangle fetch of a scene that exceeds L2 as an HBM byte; L2 and
                # the Infinity Cache serve more than half of them, and the rate on those bytes can pass the HBM
                # peak.  The fraction of the roof is then stated on the bytes HBM really moved.
                hbm["algorithmic_GBs"], hbm["algorithmic_frac"] = hbm["achieved"], hbm["frac"]
                hbm["achieved"], hbm["frac"] = hbm["traffic_GBs"], hbm["traffic_frac"]
                hbm["basis"] = "counter traffic: the caches serve part of the algorithmic bytes (algorithmic_GBs, algorithmic_frac)"
        valu = None
        if pmc.get("valu_busy_cycles_per_launch"):
            # VALU issue against 1024 SIMDs x the 2.4 GHz maximum clock (counts fixed by the workload: the RNG is
            # counter-based).  lane_utilisation = active lanes per issued instruction / 64
            # (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU); useful = the two multiplied.
            # achieved = vector instructions issued per launch (SQ_INSTS_VALU, a count fixed by the workload) x the 4
            # cycles a wave64 instruction occupies a 16-lane SIMD, over the live launch duration.  The busy-cycle
            # figure beside it (SQ_ACTIVE_INST_VALU x 4: per-wave cycles with a vector instruction in flight, which
            # weighs quarter-rate f64 rcp / sqrt fully but overlaps between waves by the pipeline's depth, so it can
            # pass 1) brackets the same thing from the other side.
            cyc = pmc.get("valu_insts_per_launch", 0) * 4.0
            busy = pmc["valu_busy_cycles_per_launch"]
            ach = cyc / (avg_ms * 1e-3) / 1e9
            peak = N_SIMD * MAX_CLOCK_HZ / 1e9
            lane = pmc.get("lane_utilisation")
            flops = pmc.get("fp64_flops_per_launch")  # wave-level F64 add/mul/fma(x2)/trans instructions x active lanes
            valu = {
                "bound": "valu-issue", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G SIMD-cycles/s",
                "frac": round(ach / peak, 4), "traffic": pmc.get("traffic_bytes_per_launch"),
                "basis": "vector instructions issued x 4 cycles per wave64 instruction, over 1024 SIMDs x 2.4 GHz",
                "busy_cycles_frac": round(busy / (avg_ms * 1e-3) / 1e9 / peak, 4),
                "valu_busy_at_effective_clock": pmc.get("valu_busy"), "effective_clock_ghz": pmc.get("clock_ghz"),
                "lane_utilisation": lane, "useful_lane_frac": round(min(1.0, ach / peak) * lane, 4) if lane else None,
                "valu_insts_per_launch": pmc.get("valu_insts_per_launch"),
                "fp64_tflops": round(flops / (avg_ms * 1e-3) / 1e12, 2) if flops else None,
                "fp64_peak_tflops_no_fma": FP64_VALU_PEAK_TFLOPS / 2,
                # how much of the issue rate is f64 arithmetic on live lanes (the rest: f32 box tests, integer RNG and
                # address work, selects, moves, idle lanes): the useful-work fraction beside the issue-slot fraction
                "fp64_frac": round(flops / (avg_ms * 1e-3) / 1e12 / (FP64_VALU_PEAK_TFLOPS / 2), 4) if flops else None,
                "waves_per_simd": pmc.get("waves_per_simd"),
            }
        # the bound: a scene that lives in cache / LDS is bound by vector issue, the 10 M-triangle scene by HBM gathers
        main = dict(valu if (resident and valu) else hbm)
        main.update({
            "kernel": dom, "avg_launch_ms": round(avg_ms, 4), "launches": launches,
            # counter-derived figures (traffic, VALU cycles, lane utilisation, flops) come from profiles/pmc.json:
            # true = a kernel source has changed since they were collected (durations are always live)
            "pmc_stale": pmc_stale,
            "hbm": hbm if (resident and valu) else None, "valu": valu if not resident else None,
            "pmc": {k: pmc.get(k) for k in ("valu_busy", "lane_utilisation", "waves_per_simd", "l2_hit", "clock_ghz")} if pmc else None,
            "scene_cache_resident": resident, "scene_bytes": scene_bytes, "tree_width": st["tree_width"], "stack_need": st["stack_need"],
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
        })
        return main

    def describe(self):
        wl, args = self.wl, self.args
        name = wl["scene"] + ("" if wl["scene"] == "heightfield" else ".obj")
        if self.part is None:
            share = f"{self.spp} spp per step"
        elif self.part.mode == "samples":
            share = f"{self.spp} spp per step per GPU ({self.spp * self.world} per step)"
        else:
            share = f"{self.spp} spp per step, the image's rows shared by {self.world} GPUs"
        return (f"{name} {self.w}x{self.h} bounces={args.bounces}{', NEE' if args.nee else ''}"
                f"{', textures' if wl.get('textures') else ''}, {share}, in batches of {self.batch} ({wl['label']})")


def timed_run(torch, dist, wk, steps, warmup, world, backend):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (seconds [max over ranks], stats,
    exchange ms of this rank over the K steps)."""
    dev = wk.dev

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(warmup):
        wk.step(i)
    fence()
    wk.exchange_ms()
    wk.scene.reset_stats()
    wk.scene.set_profiling(1)  # HIP events around every kernel launch of the timed region, on the launch stream
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        wk.step(warmup + i)
    fence()
    dt = time.perf_counter() - t0
    ex_ms = wk.exchange_ms()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = wk.scene.stats()
    wk.scene.set_profiling(0)
    return dt, st, ex_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="CornellBox", choices=sorted(WORKLOADS), help="the headline workload")
    ap.add_argument("--spp", type=int, default=0, help="frames per step (default: the config as named)")
    ap.add_argument("--batch", type=int, default=0, help="frames in flight per wavefront pass (default per workload)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--cells", type=int, default=2236, help="heightfield resolution (2236 -> 9,999,392 triangles)")
    ap.add_argument("--bounces", type=int, default=2)
    ap.add_argument("--partition", choices=("tiles", "bands", "samples"), default="tiles",
                    help="N > 1, the headline run: 'tiles' = interleaved groups of 8 rows of the ONE image of the N = 1 "
                         "workload, gathered to rank 0 every step (SURVEY.md §8e.1; fixed total work, strong scaling); "
                         "'bands' = contiguous bands; 'samples' = every rank the whole image for its own frames, partial "
                         "sums added on rank 0 in rank order every step (fixed per-GPU work, weak scaling)")
    ap.add_argument("--other-partition", choices=("auto", "none"), default="auto",
                    help="N > 1: after the headline run, time the other of tiles / samples too (under \"partitions\")")
    ap.add_argument("--no-cull-steps", type=int, default=3,
                    help="N = 1: steps of the run with both exact culls off behind `value_no_culls` (0 = skip)")
    ap.add_argument("--nee", action="store_true", help="next-event estimation (shadow rays) on")
    ap.add_argument("--configs", choices=("all", "none"), default="all",
                    help="N = 1: also time the other BASELINE configs for a few steps each")
    ap.add_argument("--config-steps", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the instrumented batches behind `roofline` (counter passes: only the timed kernels run)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start the ranks ourselves, before anything here touches the GPU ------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import pathtracer_amd as pt  # loads the HIP library (and pins the process to one HIP runtime)
    import torch
    import torch.distributed as dist
    from pathtracer_amd.dist import Partition

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
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: measuring {world} rank(s)", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    wl = dict(WORKLOADS[args.workload])
    for k in ("spp", "batch", "width", "height"):
        if getattr(args, k):
            wl[k] = getattr(args, k)
    wl["batch"] = min(wl["batch"], wl["spp"])
    tdev = dev if backend == "nccl" else "cpu"
    PARTITIONING = {"samples": "sample ranges: every rank the whole image for its own frames, partial sums gathered and "
                               "added in rank order on rank 0, every step",
                    "tiles": "interleaved groups of 8 rows of the one image, gathered to rank 0 every step",
                    "bands": "contiguous row bands of the one image, gathered to rank 0 every step"}

    def measure(mode, data=None):
        """The K timed steps under partition `mode` (None: one GPU).  Returns the workload and what it achieved."""
        part = Partition(mode, wl["width"], wl["height"]) if world > 1 else None
        wk = Workload(pt, torch, wl, args, dev, local_rank, part, data=data)
        if world > 1 and backend == "nccl":
            # first use of a collective builds channels and loads its kernels: not part of a step
            part.exchange(torch.zeros_like(wk.accum))
        dt, st, ex_ms = timed_run(torch, dist, wk, args.steps, args.warmup, world, backend)
        samples_per_rank = wk.pixels() * wk.spp * args.steps
        tot = torch.tensor([float(st["rays"]), float(samples_per_rank)], dtype=torch.float64, device=tdev)
        ex = torch.tensor([ex_ms], dtype=torch.float64, device=tdev)
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(ex, op=dist.ReduceOp.MAX)   # rank 0 waits for every sender: its share is the exchange's
        res = {"dt": dt, "st": st, "samples_per_rank": samples_per_rank, "total_rays": float(tot[0].item()),
               "total_samples": float(tot[1].item()), "exchange_ms_per_step": float(ex.item()) / args.steps}
        res["value"] = res["total_samples"] / dt / 1e6
        return wk, res

    def summary(mode, r):
        return {"value": round(r["value"], 3), "unit": "Msamples/s", "scaling": "weak" if mode == "samples" else "strong",
                "ms_per_step": round(r["dt"] / args.steps * 1e3, 3),
                "exchange_ms_per_step": round(r["exchange_ms_per_step"], 4),
                "samples_per_step": int(r["total_samples"] / args.steps), "partitioning": PARTITIONING[mode]}

    mode = args.partition if world > 1 else None
    wk, res = measure(mode)
    dt, st, value = res["dt"], res["st"], res["value"]
    samples_per_rank, total_rays, total_samples = res["samples_per_rank"], res["total_rays"], res["total_samples"]

    out = None
    if rank == 0:
        next_frame = (args.warmup + args.steps) * wk.spp * world
        roofline = None if args.no_roofline else wk.roofline(st, samples_per_rank, next_frame)
        # ---- CPU baseline: the oracle (C++ restatement, libm, f64) on this box's host cores ----
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import numpy as np
            import oracle_lib as O
            oracle = O.OracleScene(wk.data)
            threads = max(1, min(O.lib.orc_hardware_threads(), 16, len(os.sched_getaffinity(0))))
            acc = np.zeros((wk.h, wk.w, 3))
            cpu_cfg = pt.Config(width=wk.w, height=wk.h, bounces=args.bounces, seed=1, nee=args.nee)
            n_frames, c0 = 0, time.perf_counter()
            while True:
                oracle.render(cpu_cfg, acc, sample_begin=n_frames, sample_count=1, threads=threads, use_bvh=True)
                n_frames += 1
                if time.perf_counter() - c0 >= args.cpu_seconds:
                    break
            cdt = time.perf_counter() - c0
            cpu = {"value": round(wk.w * wk.h * n_frames / cdt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                   "sample": f"{n_frames} frames (spp) of the same {wk.w}x{wk.h} {wl['scene']} workload, "
                             f"{cdt:.1f} s; C++ f64 restatement of the Rust loop (rustc absent, SURVEY F1)"}
            del oracle, acc
        out = {
            "metric": "Msamples/s", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if mode == "samples" else "strong",
            "vs_baseline": None, "dtype": "f64",
            "data": (f"synthetic heightfield box, {wk.data.n_tris} triangles, seed 12345" if wl["scene"] == "heightfield"
                     else f"reference scene models/{wl['scene']}.obj") + ", counter-based RNG (no dataset)",
            "config": {"workload": wk.describe(), "triangles": wk.data.n_tris,
                       "exact_culls": {k: os.environ.get(k, "1") != "0" for k in ("PT_SCREEN_CULL", "PT_LIGHT_CULL")},
                       "partitioning": PARTITIONING[mode] if mode else "single GPU",
                       "samples_per_step": int(total_samples / args.steps)},
            "mrays_per_s": round(total_rays / dt / 1e6, 3),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if world > 1:
            out["exchange_ms_per_step"] = round(res["exchange_ms_per_step"], 4)
            out["partitions"] = {mode: summary(mode, res)}
        if roofline and roofline.get("primary_rays_traced") is not None:
            # the samples whose primary ray was generated (the others lie outside the scene's screen bounds: their
            # value is exactly +0 and they cost nothing)
            out["value_on_screen"] = round(value * roofline["primary_rays_traced"], 3)
        if cpu:
            out["gpu_over_cpu"] = round(value / cpu["value"], 1) if cpu["value"] else None
    data = wk.data
    del wk

    # ---- N > 1: the other partitioning on the same ranks (tiles <-> samples) ------------------------------
    if world > 1 and args.other_partition == "auto":
        other = "samples" if mode != "samples" else "tiles"
        owk, ores = measure(other, data=data)
        del owk
        if rank == 0:
            out["partitions"][other] = summary(other, ores)

    # ---- N = 1: the same workload with both exact culls off (every ray of the reference's render() is traced) --
    if rank == 0 and world == 1 and args.no_cull_steps > 0:
        nwk = Workload(pt, torch, wl, args, dev, local_rank, None, data=data, options={"light_cull": 0, "screen_cull": 0})
        ndt, nst, _ = timed_run(torch, dist, nwk, args.no_cull_steps, 1, 1, backend)
        n = nwk.pixels() * nwk.spp * args.no_cull_steps
        out["value_no_culls"] = round(n / ndt / 1e6, 3)
        out["no_culls"] = {"steps": args.no_cull_steps, "warmup": 1, "ms_per_step": round(ndt / args.no_cull_steps * 1e3, 3),
                           "mrays_per_s": round(nst["rays"] / ndt / 1e6, 3),
                           "rays_per_sample": round(nst["rays"] / max(1, n), 4)}
        del nwk
    del data

    # ---- the other BASELINE configs, a few steps each (single GPU only) ----------------------------------
    if rank == 0 and world == 1 and args.configs == "all":
        others = {}
        for name, owl in WORKLOADS.items():
            if name == args.workload:
                continue
            owk = Workload(pt, torch, dict(owl), args, dev, local_rank, None)
            odt, ost, _ = timed_run(torch, dist, owk, args.config_steps, 1, 1, backend)
            n = owk.pixels() * owk.spp * args.config_steps
            others[name] = {
                "workload": owk.describe(), "triangles": owk.data.n_tris,
                "value": round(n / odt / 1e6, 3), "unit": "Msamples/s", "steps": args.config_steps, "warmup": 1,
                "ms_per_step": round(odt / args.config_steps * 1e3, 3),
                "mrays_per_s": round(ost["rays"] / odt / 1e6, 3),
                "roofline": owk.roofline(ost, n, (1 + args.config_steps) * owk.spp),
            }
            del owk
        out["configs"] = others

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
