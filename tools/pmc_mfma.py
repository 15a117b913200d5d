#!/usr/bin/env python3
"""MFMA-pipe utilisation and effective clock per kernel group from one rocprofv3 PMC pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, with --kernel-trace only) of
`python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events --no_fp32`.

    python tools/pmc_mfma.py <counter_collection.csv> <kernel_trace.csv> <out_prefix>

MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles of the matrix pipe (32 per v_mfma_f32_32x32x16_bf16), summed
over every SIMD; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so a dispatch's clock = GRBM_GUI_ACTIVE / 8 / duration and
the SIMD-cycles it had are GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs.  mfma_busy = MFMA cycles / SIMD-cycles.
Writes <out_prefix>.txt and <out_prefix>.json."""
import collections
import csv
import json
import sys

GROUPS = [
    ("conv_ws", ["conv_ws_kernel", "conv_ws_pipe_kernel", "conv_ws2_kernel"]),
    ("conv_wgrad_shift", ["conv_wgrad_shift_kernel"]),
    ("conv_wgrad_other", ["conv_wgrad_kernel", "conv_wgrad_dma_kernel"]),
    ("conv_gemm_dma", ["conv_gemm_dma_kernel"]),
    ("conv_gemm", ["conv_gemm_kernel"]),
    ("conv1x1_fold", ["conv1x1_fold_kernel"]),
    ("conv_k3gate", ["conv_k3gate_kernel"]),
    ("conv1x1_bwd", ["conv1x1_bwd_kernel"]),
    ("conv_k1act", ["conv_k1act_kernel"]),
    ("conv_k1_bwd", ["conv_k1_bwd_kernel"]),
    ("vq_search", ["vq_search_kernel"]),
]
N_SIMD = 256 * 4


def group_of(name):
    for key, subs in GROUPS:
        if any(s in name for s in subs):
            return key
    return None


def main():
    counters_csv, trace_csv, prefix = sys.argv[1:4]
    dur = {}
    with open(trace_csv) as f:
        for row in csv.DictReader(f):
            dur[row["Dispatch_Id"]] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-9
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    with open(counters_csv) as f:
        for row in csv.DictReader(f):
            key = group_of(row["Kernel_Name"])
            if key is None:
                continue
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Dispatch_Id"] not in seen[key]:
                seen[key].add(row["Dispatch_Id"])
                acc[key]["seconds"] += dur.get(row["Dispatch_Id"], 0.0)
    out, lines = {}, [__doc__.split("\n\n")[2].strip(), ""]
    for key, _ in GROUPS:
        a = acc.get(key)
        if not a or not a.get("GRBM_GUI_ACTIVE"):
            continue
        simd_cycles = a["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
        clock = a["GRBM_GUI_ACTIVE"] / 8.0 / a["seconds"] * 1e-9 if a["seconds"] else float("nan")
        out[key] = {"launches_profiled": len(seen[key]), "mfma_busy": busy, "effective_clock_ghz": clock,
                    "ms_profiled": a["seconds"] * 1e3,
                    "sq_busy": a.get("SQ_BUSY_CYCLES", 0.0) / (a["GRBM_GUI_ACTIVE"] / 8.0 * 8 * 4) if a.get("SQ_BUSY_CYCLES") else None}
        lines.append(f"{key:18s} launches {len(seen[key]):4d}  {a['seconds'] * 1e3:8.3f} ms  MFMA pipe busy {100 * busy:5.1f} %"
                     f"   effective clock {clock:5.2f} GHz")
    open(prefix + ".json", "w").write(json.dumps(out, indent=1))
    open(prefix + ".txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
