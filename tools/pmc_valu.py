"""Per kernel of one bench step: VALU instructions and VALU-busy share of the SIMD cycles (tools/pmc_valu.sh).
SQ_ACTIVE_INST_VALU counts, per SIMD, the cycles (in units of 4) its vector ALU executes an instruction; GRBM_GUI_ACTIVE is
summed over the 8 XCDs, so SIMD-cycles of a dispatch = GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs."""
import collections
import csv
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    name = re.sub(r"^smt::", "", name)
    m = re.match(r"_ZN3smt\d+([A-Za-z0-9_]+?)I", name)
    return (m.group(1) if m else name)[:40]


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    if not os.path.exists(path):
        return acc, launches
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"])
            launches[k] += 1
    return acc, launches


a, la = load(sys.argv[1])
b, _ = load(sys.argv[2]) if len(sys.argv) > 2 else ({}, None)
rows = []
for k, v in a.items():
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    simd_cycles = gui / 8 * 256 * 4
    rows.append((gui, k, la[k], v.get("SQ_INSTS_VALU", 0), v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles,
                 v.get("SQ_WAVE_CYCLES", 0) * 4 / simd_cycles, b.get(k, {})))
rows.sort(reverse=True)
print(f"{'kernel':40s} {'launches':>8s} {'Mcycles/XCD':>11s} {'VALU insts':>12s} {'VALU busy':>9s} {'waves/SIMD':>10s}  LDS insts / busy, VMEM rd / wr insts")
for gui, k, n, iv, busy, occ, bb in rows[:28]:
    extra = ""
    if bb:
        simd_cycles = gui / 8 * 256 * 4
        extra = f"  {bb.get('SQ_INSTS_LDS', 0):12.0f} / {bb.get('SQ_ACTIVE_INST_LDS', 0) * 4 / simd_cycles:5.2f}, {bb.get('SQ_INSTS_VMEM_RD', 0):10.0f} / {bb.get('SQ_INSTS_VMEM_WR', 0):10.0f}"
    print(f"{k:40s} {n:8d} {gui / 8 / 1e6:11.2f} {iv:12.0f} {busy:9.2f} {occ:10.2f}{extra}")
