#!/usr/bin/env python3
"""Turns one tools/profile.sh output directory into a short markdown summary (stdout).
Usage: python tools/summarize_profile.py gpurun_out/prof_<tag> "<title>"
HBM bytes follow the gfx950 corrections: FETCH_SIZE and WRITE_SIZE are reported in KiB;
FETCH_SIZE counts 64 B per 128-B request on wide streaming reads, so the fetched bytes lie
between 1x and 2x the counter (this kernel's loads are 16-B gathers, not wide streams; both
bounds are printed)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else d
print(f"# {title}\n")
import os
command = "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-both-hierarchies --no-walk-stats"
if os.path.exists(f"{d}/args.txt"):
    command = open(f"{d}/args.txt").read().strip()
print(f"Command per pass: `rocprofv3 <opts> --output-format csv -- {command}`\n")
stats = glob.glob(f"{d}/trace/*/*_kernel_stats.csv")
if stats:
    print("## rocprofv3 --kernel-trace --stats\n")
    print("| kernel | calls | total ns | average ns | % | min ns | max ns |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats[0])):
        print(f"| {r['Name']} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} | {r['MinNs']} | {r['MaxNs']} |")
    print()
trace = glob.glob(f"{d}/trace/*/*_kernel_trace.csv")
if trace:
    seen = set()
    for r in csv.DictReader(open(trace[0])):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "_kernel" in kn and kn not in seen:
            seen.add(kn)
            print(f"{kn} launch: grid {r['Grid_Size_X']}, workgroup {r['Workgroup_Size_X']}, VGPR {r.get('VGPR_Count')}, "
                  f"SGPR {r.get('SGPR_Count')}, LDS {r.get('LDS_Block_Size')} B, scratch {r.get('Scratch_Size')} B/lane\n")
FRAMES = 6   # bench.py --steps 5 --warmup 1
print(f"## PMC passes (sum over a frame's dispatches of each kernel, mean of {FRAMES} frames)\n")
print("| pass | kernel | counter | per frame |")
print("|---|---|---|---|")
vals = {}
seen_counter = set()
perk = collections.defaultdict(dict)
for f in sorted(glob.glob(f"{d}/pmc_*/*/*_counter_collection.csv")):
    name = f.split("/")[-3]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        # once-per-upload kernels (leaf records, light grids) are not part of a frame
        if "_kernel" in kn and not any(w in kn for w in ("expand", "untile", "grid_kernel", "sum_kernel", "copy_kernel", "longest_kernel", "Scan", "rocprim", "hipcub")):
            agg[(kn, r["Counter_Name"])] += float(r["Counter_Value"])
    for (kn, k), v in sorted(agg.items()):
        perk[kn][k] = v / FRAMES
        if (kn, k) not in seen_counter:        # a counter collected in several passes is reported once
            seen_counter.add((kn, k))
            vals[k] = vals.get(k, 0.0) + v / FRAMES
        print(f"| {name} | {kn} | {k} | {v / FRAMES:.6g} |")
print()
for kn, kv in perk.items():
    if "SQ_WAVE_CYCLES" in kv:
        print(f"- {kn}: active {kv.get('SQ_ACTIVE_INST_ANY', 0) / kv['SQ_WAVE_CYCLES']:.3f}, waiting {kv.get('SQ_WAIT_ANY', 0) / kv['SQ_WAVE_CYCLES']:.3f}, "
              f"issue-stalled {kv.get('SQ_WAIT_INST_ANY', 0) / kv['SQ_WAVE_CYCLES']:.3f} of wave cycles; VALU instructions {kv.get('SQ_INSTS_VALU', 0):.4g}")
print()
if "SQ_WAVE_CYCLES" in vals:
    wc = vals["SQ_WAVE_CYCLES"]
    print("## Derived\n")
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in vals:
            print(f"- {k} / SQ_WAVE_CYCLES = {vals[k] / wc:.3f}")
    if "SQ_INSTS_VALU" in vals and "SQ_WAVES" in vals:
        print(f"- VALU wave-instructions per wavefront = {vals['SQ_INSTS_VALU'] / vals['SQ_WAVES']:.0f}")
if "SQ_THREAD_CYCLES_VALU" in vals and "SQ_ACTIVE_INST_VALU" in vals:
    print(f"- live lanes per vector instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = "
          f"{vals['SQ_THREAD_CYCLES_VALU'] / vals['SQ_ACTIVE_INST_VALU']:.1f} of 64")
if "GRBM_GUI_ACTIVE" in vals and "SQ_INSTS_VALU" in vals:
    simd_cycles = vals["GRBM_GUI_ACTIVE"] / 8 * 1024      # the counter is summed over the 8 XCDs; 256 CUs x 4 SIMDs
    print(f"- kernel duration {vals['GRBM_GUI_ACTIVE'] / 8:.4g} cycles; {simd_cycles / vals['SQ_INSTS_VALU']:.2f} SIMD-cycles per vector instruction")
    if "SQ_ACTIVE_INST_VALU" in vals:
        # the profiler's VALUBusy: SQ_ACTIVE_INST_VALU ticks once per four cycles a wavefront has a vector instruction executing;
        # instructions of different wavefronts overlap in a SIMD's pipeline, so the figure can pass 1
        busy = 4.0 * vals["SQ_ACTIVE_INST_VALU"]
        print(f"- VALUBusy = 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles = {busy:.4g} / {simd_cycles:.4g} = {busy / simd_cycles:.3f} "
              f"(at or above 1: the vector ALUs never waited for work)")
if "TCC_HIT_sum" in vals:
    print(f"- L2 hit rate = {vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):.4f}")
if "TCP_TOTAL_CACHE_ACCESSES_sum" in vals and "TCP_TCC_READ_REQ_sum" in vals:
    print(f"- vector L1: {vals['TCP_TOTAL_CACHE_ACCESSES_sum']:.4g} accesses, {vals['TCP_TCC_READ_REQ_sum']:.4g} read requests to L2 "
          f"({1 - vals['TCP_TCC_READ_REQ_sum'] / vals['TCP_TOTAL_CACHE_ACCESSES_sum']:.4f} served by L1)")
if "FETCH_SIZE" in vals:
    lo = vals["FETCH_SIZE"] * 1024
    wr = vals.get("WRITE_SIZE", 0) * 1024
    print(f"- HBM read bytes per dispatch: {lo / 1e6:.1f} MB (counter) .. {2 * lo / 1e6:.1f} MB (x2 bound); written: {wr / 1e6:.1f} MB")

# machine-readable line for tools/update_traffic.py
import json
keep = {k: vals[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE",
                             "SQ_WAIT_ANY", "SQ_WAVE_CYCLES") if k in vals}
print(f"\n<!-- counters {json.dumps(keep)} -->")
