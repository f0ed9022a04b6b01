#!/usr/bin/env python3
"""Digest of the rocprofv3 PMC passes of scripts/pmc_collect.sh.

    python scripts/pmc_digest.py <workload key> gpurun_out/pmc_<tag> [profiles/pmc.json] > digest.txt

Per kernel kind (path / extend / shade / shadow / accumulate), per launch, merged into the JSON file bench.py
reads (`roofline.traffic`, the VALU-issue roofline), and a per-instantiation text digest on stdout.

Units and corrections (MI355X_MICROARCH.md "HBM" and the cycle-constants table):
  * FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE tallies 128-B read requests at 64 B on gfx950 and is
    doubled (calibrated in round 1 on k_accumulate, whose byte count is known exactly); WRITE_SIZE is exact;
  * SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* count quad-cycles (x4 = cycles, summed over waves / SIMDs);
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 (/8 = cycles the launch took); over the
    dispatch's own duration in that pass it gives the effective clock of the profiled run (clock_ghz).
  * the instrumented extend kernel (k_extend<.., true, ..>, node / triangle counting) is not a timed kernel.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd.buildinfo import kernel_source_hashes  # noqa: E402

KINDS = ("k_extend", "k_shade", "k_path", "k_shadow", "k_accumulate")
N_SIMD = 1024.0


def main():
    key, pmc_dir = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc.json"
    tot = collections.defaultdict(lambda: collections.defaultdict(float))    # kind -> counter -> sum
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))      # kind -> counter -> dispatches
    inst = collections.defaultdict(lambda: collections.defaultdict(float))   # instantiation -> counter -> sum
    icnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(pmc_dir + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"ptd::(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
            if not m or m.group(1) not in KINDS:
                continue
            if re.search(r"k_extend<\d+, true", r["Kernel_Name"]):
                continue
            c, v = r["Counter_Name"], float(r["Counter_Value"])
            kind, full = m.group(1)[2:], m.group(1) + (m.group(2) or "")
            tot[kind][c] += v; cnt[kind][c] += 1
            inst[full][c] += v; icnt[full][c] += 1
            if c == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):   # the dispatch's duration in that pass (ns)
                ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                for t_, n_, k_ in ((tot, cnt, kind), (inst, icnt, full)):
                    t_[k_]["_gui_ns"] += ns; n_[k_]["_gui_ns"] += 1

    def derive(t, n):
        L = max(n.get("SQ_WAVES", 0), n.get("FETCH_SIZE", 0), 1)
        per = lambda c: t.get(c, 0.0) / max(n.get(c, 0), 1)
        cyc = per("GRBM_GUI_ACTIVE") / 8.0
        busy = per("SQ_ACTIVE_INST_VALU") * 4.0
        d = {"launches_profiled": L}
        if t.get("FETCH_SIZE") is not None and n.get("FETCH_SIZE"):
            d["fetch_bytes_per_launch"] = round(per("FETCH_SIZE") * 1024.0 * 2.0)
            d["write_bytes_per_launch"] = round(per("WRITE_SIZE") * 1024.0)
            d["traffic_bytes_per_launch"] = d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"]
        if busy:
            d["valu_busy_cycles_per_launch"] = round(busy)
            d["valu_insts_per_launch"] = round(per("SQ_INSTS_VALU"))
            d["lane_utilisation"] = round(per("SQ_THREAD_CYCLES_VALU") / (per("SQ_ACTIVE_INST_VALU") * 64.0), 3)
            d["waves_per_launch"] = round(per("SQ_WAVES"))
            d["valu_insts_per_wave"] = round(per("SQ_INSTS_VALU") / max(per("SQ_WAVES"), 1), 1)
            if cyc:
                d["gui_cycles_per_launch"] = round(cyc)
                if per("_gui_ns"):
                    d["clock_ghz"] = round(cyc / per("_gui_ns"), 3)   # effective clock of the profiled pass
                d["valu_busy"] = round(busy / (N_SIMD * cyc), 3)
                d["waves_per_simd"] = round(per("SQ_WAVE_CYCLES") * 4.0 / (N_SIMD * cyc), 2)
        f64 = {k: per("SQ_INSTS_VALU_" + k) for k in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64")}
        if any(f64.values()):
            d["f64_insts_per_launch"] = {k.lower(): round(v) for k, v in f64.items()}
            d["other_insts_per_launch"] = {k.lower(): round(per("SQ_INSTS_VALU_" + k)) for k in ("INT32", "INT64", "CVT", "MUL_F32")}
            lane = d.get("lane_utilisation") or 1.0
            # flops actually done: wave instructions x 64 lanes x the share of lanes that are on (FMA = 2)
            d["fp64_flops_per_launch"] = round((f64["ADD_F64"] + f64["MUL_F64"] + f64["TRANS_F64"] + 2.0 * f64["FMA_F64"]) * 64.0 * lane)
        if per("SQ_INSTS_VALU") and any(f64.values()):
            # instruction mix per wave: what the vector issue slots are spent on.  Classified by the SQ counters; the
            # remainder ("other") is what no counter names: moves, selects, compares, min / max, bit operations, lane ops
            w = max(per("SQ_WAVES"), 1.0)
            mix = {
                "f64_arith": (f64["ADD_F64"] + f64["MUL_F64"] + f64["FMA_F64"] + f64["TRANS_F64"]) / w,
                "f32_arith": sum(per("SQ_INSTS_VALU_" + k) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")) / w,
                "int32": per("SQ_INSTS_VALU_INT32") / w, "int64": per("SQ_INSTS_VALU_INT64") / w, "cvt": per("SQ_INSTS_VALU_CVT") / w,
            }
            mix["other_valu"] = per("SQ_INSTS_VALU") / w - sum(mix.values())
            d["valu_mix_per_wave"] = {k: round(v, 1) for k, v in mix.items()}
            d["valu_mix_share"] = {k: round(v * w / per("SQ_INSTS_VALU"), 3) for k, v in mix.items()}
            d["non_valu_per_wave"] = {k.lower(): round(per("SQ_INSTS_" + k) / w, 1) for k in
                                      ("SALU", "SMEM", "BRANCH", "LDS", "LDS_LOAD", "LDS_STORE", "VMEM_RD", "VMEM_WR") if per("SQ_INSTS_" + k)}
        hit, miss = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
        if hit + miss:
            d["l2_hit"] = round(hit / (hit + miss), 3)
        if per("SQ_WAIT_ANY"):
            d["wait_any_frac_of_wave_cycles"] = round(per("SQ_WAIT_ANY") / max(per("SQ_WAVE_CYCLES"), 1), 3)
        if per("TA_BUSY_avr") and cyc:
            d["ta_busy"] = round(per("TA_BUSY_avr") / cyc, 3)
        return d

    res = {kind: derive(tot[kind], cnt[kind]) for kind in tot}
    for kind in list(res):
        res[kind]["correction"] = "FETCH_SIZE KiB x1024 x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE KiB x1024; SQ cycle counters x4"
    try:
        allj = json.load(open(out_path))
    except (OSError, ValueError):
        allj = {}
    # which device code the counters belong to: bench.py marks them stale when a kernel source has changed since
    # (run this digest from the tree the passes were collected on)
    res["_sources"] = kernel_source_hashes()
    allj[key] = res
    json.dump(allj, open(out_path, "w"), indent=1, sort_keys=True)
    print(f"# {key}: per kernel instantiation, per launch (scripts/pmc_digest.py)")
    for full in sorted(inst):
        d = derive(inst[full], icnt[full])
        print(full)
        for k, v in d.items():
            print(f"    {k}: {v}")


if __name__ == "__main__":
    main()
