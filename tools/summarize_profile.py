#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one table:
per kernel the launch count, average duration and the PMC counters averaged
per launch.  FETCH_SIZE / WRITE_SIZE are reported in bytes with the gfx950
correction for wide coalesced reads (FETCH_SIZE counts 64 B per 128-B request:
doubled here, flagged as an upper bound for narrow access patterns)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    return name.split("(")[0].replace("void ", "")


stats = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        stats[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3, float(row["Percentage"]))

# medians from the raw trace: first launches (code upload, cold caches) skew the averages
durations = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        durations[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
durations = {k: [d for _, d in sorted(v)] for k, v in durations.items()}       # in launch order


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


pmc = defaultdict(lambda: defaultdict(list))
# GRBM_GUI_ACTIVE of each counter pass: the cycles the kernel took IN THAT PASS (counter collection serialises launches
# and stretches them, each pass differently: a fraction is a pass's counter over that same pass's cycles).  Launches that
# are not the benchmark's ordinary steps -- the first frame that loads the code, the counting step -- are left out the
# way steady() leaves them out of the durations: by their cycles against the pass's median.
active = defaultdict(lambda: defaultdict(list))         # pass -> kernel -> GRBM_GUI_ACTIVE
pass_of = {}                                            # counter -> pass
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    which = os.path.relpath(f, out).split(os.sep)[0]
    launches = defaultdict(dict)                        # (kernel, dispatch) -> counter -> value
    for row in csv.DictReader(open(f)):
        launches[(short(row["Kernel_Name"]), row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    cycles = defaultdict(list)
    for (kernel, _), values in launches.items():
        if "GRBM_GUI_ACTIVE" in values:
            cycles[kernel].append(values["GRBM_GUI_ACTIVE"])
    for (kernel, _), values in launches.items():
        if "GRBM_GUI_ACTIVE" in values:
            m = median(cycles[kernel])
            if not 0.5 * m <= values["GRBM_GUI_ACTIVE"] <= 2 * m:
                continue
        for counter, value in values.items():
            if counter == "GRBM_GUI_ACTIVE":
                active[which][kernel].append(value)
                if which != "fetch":
                    continue
            pmc[kernel][counter].append(value)
            pass_of[counter] = which
XCDS = 8


def cycles_in_pass(kernel, counter):
    """average cycles per launch of `kernel` in the pass that collected `counter` (GRBM_GUI_ACTIVE is summed over the
    eight XCDs), or None when that pass did not collect it"""
    v = active.get(pass_of.get(counter, ""), {}).get(kernel)
    return sum(v) / len(v) / XCDS if v else None

def steady(v):
    """average over the launches of the benchmark's ordinary steps: without those that take more than twice
    the median (the untimed step in which bench.py turns the device-side fragment counters on, whose atomics
    make every launch 30 times longer) or less than half of it (the small frame bench.py draws first to
    load the code and the scene)"""
    m = median(v)
    kept = [d for d in v if 0.5 * m <= d <= 2 * m]
    return sum(kept) / len(kept) if kept else float("nan")


print("avg us = rocprofv3's own average over all launches, including bench.py's one untimed counting step (device-side")
print("fragment counters on: every launch takes 30 times longer) and the small first frame; steady avg us = without the")
print("launches longer than twice or shorter than half the median, i.e. the ordinary steps, which is what bench.py's")
print("HIP-event figure measures")
print(f"{'kernel':34s} {'calls':>6s} {'avg us':>9s} {'%':>6s} {'steady avg us':>14s} {'median us':>10s}")
for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:34s} {calls:6d} {avg:9.2f} {pct:6.2f} {steady(durations.get(k, [])):14.2f} {median(durations.get(k, [])):10.2f}")
    for c, vals in sorted(pmc.get(k, {}).items()):
        mean = sum(vals) / len(vals)
        note = ""
        if c == "FETCH_SIZE":
            note = f"  KiB -> {mean * 1024 * 2 / 1e6:.2f} MB/launch (x2 gfx950 correction)"
        if c == "WRITE_SIZE":
            note = f"  KiB -> {mean * 1024 / 1e6:.2f} MB/launch"
        print(f"    {c:28s} {mean:16.1f}{note}")

# machine-readable figures per kernel for bench.py's roofline objects
import json
frames = int(sys.argv[2]) if len(sys.argv) > 2 else None
HBM_PEAK = 8000e9       # bytes/s
COMPUTE_UNITS = 256
CLOCK_MHZ = 2400.0
traffic = {"frames_per_launch": frames, "unit": "bytes per launch; fractions over the kernel's steady average duration in this profile",
           "note": "traffic_frac = (FETCH_SIZE + WRITE_SIZE, KiB -> bytes) / duration / 8 TB/s; traffic_frac_x2 doubles FETCH_SIZE, the gfx950 "
                   "correction for wide coalesced reads (an upper bound for 4- and 16-byte gathers); valu_issue_frac = wave-level VALU "
                   "instructions x 4 cycles over the cycles of the 1024 SIMDs; td_busy_frac = TD_TD_BUSY_sum over the "
                   "cycles of the 256 compute units' texture data paths; cycles = GRBM_GUI_ACTIVE / 8 XCDs of the same counter pass "
                   "(2.4 GHz x the trace's duration where a pass did not collect it)",
           "kernels": {}}
for k in stats:
    c = pmc.get(k, {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024
        write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
        us = steady(durations.get(k, []))
        entry = {"fetch_size_bytes_raw": fetch, "write_size_bytes": write,
                 "hbm_bytes_per_launch": 2 * fetch + write,
                 "median_us": median(durations.get(k, [])), "steady_avg_us": us, "launches": len(durations.get(k, []))}
        if us == us and us > 0:
            entry["traffic_frac"] = (fetch + write) / (us * 1e-6) / HBM_PEAK
            entry["traffic_frac_x2"] = (2 * fetch + write) / (us * 1e-6) / HBM_PEAK
            # VALU issue utilisation: wave-level VALU instructions x 4 cycles (a wave64 op on a 16-lane SIMD)
            # over the cycles the 1024 SIMDs had during the launch, at the 2.4 GHz peak engine clock
            if "SQ_INSTS_VALU" in c:
                valu = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])
                salu = sum(c["SQ_INSTS_SALU"]) / len(c["SQ_INSTS_SALU"]) if "SQ_INSTS_SALU" in c else 0.0
                entry["valu_insts_per_launch"] = valu
                entry["salu_insts_per_launch"] = salu
                cyc = cycles_in_pass(k, "SQ_INSTS_VALU") or us * CLOCK_MHZ
                entry["valu_issue_frac"] = valu * 4 / (4 * COMPUTE_UNITS * cyc)
                # scalar issue: a compute unit's scalar unit takes one instruction per cycle from its four SIMDs in turn,
                # i.e. one per SIMD every 4 cycles, like the vector unit; scalar memory instructions go through it too
                smem = sum(c["SQ_INSTS_SMEM"]) / len(c["SQ_INSTS_SMEM"]) if "SQ_INSTS_SMEM" in c else 0.0
                entry["salu_issue_frac"] = (salu + smem) * 4 / (4 * COMPUTE_UNITS * (cycles_in_pass(k, "SQ_INSTS_SALU") or us * CLOCK_MHZ))
            if "TD_TD_BUSY_sum" in c:
                busy = sum(c["TD_TD_BUSY_sum"]) / len(c["TD_TD_BUSY_sum"])
                entry["td_busy_frac"] = busy / (COMPUTE_UNITS * (cycles_in_pass(k, "TD_TD_BUSY_sum") or us * CLOCK_MHZ))
            if "SQ_INSTS_VMEM_RD" in c:
                entry["vmem_rd_insts_per_launch"] = sum(c["SQ_INSTS_VMEM_RD"]) / len(c["SQ_INSTS_VMEM_RD"])
        traffic["kernels"][k] = entry
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
