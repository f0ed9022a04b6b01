#!/usr/bin/env python3
"""HBM traffic per launch and kernel kind from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_passes.sh.

    python scripts/pmc_traffic.py <workload key> gpurun_out/pmc_<tag> [profiles/traffic.json]

Units and corrections (MI355X_MICROARCH.md "HBM", calibrated on this build's own k_accumulate, whose
byte count is known exactly: 24 B per sample + 24 B per pixel read, 24 B per pixel written):
  * both counters are in KiB;
  * FETCH_SIZE tallies 128-B read requests at 64 B on gfx950: doubled (k_accumulate reads
    447.9 MB per CornellBox launch, the counter says 219.0 MiB-KiB -> x2 = 448.6 MB);
  * WRITE_SIZE is exact (49.8 MB written, counter 49.8 MB).
The result is merged into the JSON file bench.py reads for `roofline.traffic`.
"""
import collections
import csv
import glob
import json
import re
import sys

KINDS = ("k_extend", "k_shade", "k_path", "k_shadow", "k_accumulate")


def main():
    key, pmc_dir = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(pmc_dir + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c not in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES",
                         "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"):
                continue
            m = re.search(r"ptd::(k_\w+)", r["Kernel_Name"])
            if not m or m.group(1) not in KINDS:
                continue
            if re.search(r"k_extend<\d+, true", r["Kernel_Name"]):
                continue  # the instrumented (node/triangle counting) step is not the timed kernel
            kind = m.group(1)[2:]
            tot[kind][c] += float(r["Counter_Value"])
            n[kind][c] += 1
    res = {}
    for kind in tot:
        launches = max(n[kind]["FETCH_SIZE"], 1)
        fetch = tot[kind]["FETCH_SIZE"] * 1024.0 * 2.0 / launches
        write = tot[kind]["WRITE_SIZE"] * 1024.0 / max(n[kind]["WRITE_SIZE"], 1)
        t = tot[kind]
        cyc = t["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs by rocprofv3
        simds = 1024.0
        occ = {
            # SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md constants table)
            "valu_busy": round(t["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * cyc), 3) if cyc else None,
            "lane_utilisation": round(t["SQ_THREAD_CYCLES_VALU"] / (t["SQ_ACTIVE_INST_VALU"] * 64.0), 3)
            if t["SQ_ACTIVE_INST_VALU"] else None,
            "waves_per_simd": round(t["SQ_WAVE_CYCLES"] * 4.0 / (simds * cyc), 2) if cyc else None,
            "l2_hit": round(t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]), 3)
            if (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]) else None,
        }
        res[kind] = {"launches_profiled": launches, "fetch_bytes_per_launch": round(fetch), **occ,
                     "write_bytes_per_launch": round(write), "traffic_bytes_per_launch": round(fetch + write),
                     "correction": "FETCH_SIZE KiB x1024 x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE KiB x1024"}
    try:
        allj = json.load(open(out_path))
    except (OSError, ValueError):
        allj = {}
    allj[key] = res
    json.dump(allj, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({key: res}, indent=1))


if __name__ == "__main__":
    main()
