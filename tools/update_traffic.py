#!/usr/bin/env python3
"""Puts the counters of one profile summary (tools/summarize_profile.py output, kept under profiles/) into
profiles/traffic.json, where bench.py reads the HBM traffic and the vector-ALU figures of a workload from.
Usage: python tools/update_traffic.py profiles/<summary>.md <workload> <key: bvh|octree|...>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary, workload, key = sys.argv[1:4]
text = open(summary).read()
found = re.search(r"<!-- counters (\{.*\}) -->", text)
if not found:
    raise SystemExit(f"{summary}: no counters line")
c = json.loads(found.group(1))
path = os.path.join(ROOT, "profiles", "traffic.json")
table = json.load(open(path))
simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024
entry = {"fetch_counter_bytes": int(c["FETCH_SIZE"] * 1024), "write_bytes": int(c["WRITE_SIZE"] * 1024),
         "source": os.path.relpath(os.path.abspath(summary), ROOT),
         "valu": {"insts_valu": c["SQ_INSTS_VALU"], "simd_cycles": simd_cycles,
                  "simd_cycles_per_inst": round(simd_cycles / c["SQ_INSTS_VALU"], 3),
                  "lanes_per_inst": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 1),
                  "active_cycles": 4.0 * c["SQ_ACTIVE_INST_VALU"],
                  "cycles_per_inst_measured": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], 2),
                  "valu_busy_formula": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles, 3),
                  "busy_frac": round(min(1.0, 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles), 3),
                  "waiting_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3) if "SQ_WAIT_ANY" in c else None}}
table.setdefault(workload, {})[key] = entry
table["_comment"] = ("Counter evidence per trace_kernel launch from separate rocprofv3 --pmc passes (tools/profile.sh, tools/update_traffic.py): "
                     "FETCH_SIZE and WRITE_SIZE in KiB x 1024; FETCH_SIZE doubled by the reader (gfx950 tallies 128-B requests at 64 B). "
                     "valu: vector instructions issued (SQ_INSTS_VALU), SIMD-cycles of the launch (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), "
                     "live lanes per instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), and valu_busy_formula = 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles, "
                     "the profiler's own VALUBusy figure (busy_frac is that capped at 1): the counter ticks once per four cycles a wavefront has a vector "
                     "instruction executing, and with several wavefronts issuing to one SIMD their instructions overlap, so the figure can pass 1; "
                     "what it says at or above 1 is that the vector ALUs never waited for work.  Entries without active_cycles are from before "
                     "that change (busy_frac = min(1, 4 / simd_cycles_per_inst)).  What an instruction of each class costs at this kernel's "
                     "occupancy: profiles/r3_micro_valu_ops.txt.")
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
