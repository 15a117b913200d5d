#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only) of
`python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_kernel_events` into per-kernel-group HBM bytes per launch.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix>

Counter unit = KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the
bytes of wide (16 B/lane) coalesced reads, global_load and LDS-DMA alike -> x2; WRITE_SIZE is exact for 16 B/lane
stores.  Writes <out_prefix>.txt and <out_prefix>.json (bench.py reads the .json for roofline.traffic)."""
import csv, json, sys, collections

GROUPS = [  # (key, substrings of the kernel name)
    ("conv_ws_kernel", ["conv_ws_kernel", "conv_ws_pipe_kernel", "conv_ws2_kernel"]),
    ("conv_gemm_dma_kernel", ["conv_gemm_dma_kernel"]),
    ("conv1x1_dma_kernel", ["conv1x1_dma_kernel"]),
    ("conv1x1_fold_kernel", ["conv1x1_fold_kernel"]),
    ("conv_k3gate_kernel", ["conv_k3gate_kernel"]),
    ("conv1x1_bwd_kernel", ["conv1x1_bwd_kernel"]),
    ("conv_k1act_kernel", ["conv_k1act_kernel"]),
    ("conv_k1_bwd_kernel", ["conv_k1_bwd_kernel"]),
    ("conv_gemm_kernel", ["conv_gemm_kernel"]),
    ("conv_wgrad_shift_kernel", ["conv_wgrad_shift_kernel"]),
    ("conv_wgrad", ["conv_wgrad_kernel", "conv_wgrad_dma_kernel", "conv_wgrad_reduce_kernel"]),
    ("gate_mix", ["gate_mix_fwd_kernel", "gate_mix_bwd_kernel"]),
    ("vq_forward", ["vq_search", "vq_candidates", "vq_exact", "vq_reduce"]),
    ("recon_loss", ["recon_loss_fwd", "recon_loss_bwd"]),
    ("resample", ["conv4s2_kernel", "convt4s2_kernel"]),
    ("gate_mix_bwd", ["gate_mix_bwd_kernel"]),
    ("conv_gate_bwd", ["conv_gate_bwd_kernel"]),
    ("melspec", ["melspec_kernel", "melspec_wave_kernel"]),
    ("stft_loss", ["stft_loss_fwd", "stft_loss_bwd", "stft_loss_gather", "stft_overlap_gather"]),
    ("vq_ema_accumulate", ["vq_ema_accumulate", "vq_ema_sort", "vq_ema_segment"]),
]


def load(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for key, subs in GROUPS:
                if any(s in name for s in subs):
                    tot[key] += float(row["Counter_Value"]); n[key] += 1
                    break
    return tot, n


def main():
    fetch_csv, write_csv, prefix = sys.argv[1:4]
    ft, fn = load(fetch_csv, "FETCH_SIZE")
    wt, wn = load(write_csv, "WRITE_SIZE")
    out, lines = {}, [__doc__.split("\n\n")[2].strip(), ""]
    for key, _ in GROUPS:
        if not fn.get(key):
            continue
        fpl = ft[key] / fn[key] * 1024.0 * 2.0
        wpl = wt[key] / max(1, wn[key]) * 1024.0
        out[key] = {"launches_profiled": fn[key], "fetch_bytes_per_launch": fpl, "write_bytes_per_launch": wpl,
                    "hbm_bytes_per_launch": fpl + wpl}
        lines.append(f"{key:26s} launches {fn[key]:5d}  FETCH_SIZE/launch {ft[key] / fn[key]:12.1f} KiB -> {fpl / 1e6:9.1f} MB (x2)"
                     f"   WRITE_SIZE/launch {wt[key] / max(1, wn[key]):12.1f} KiB -> {wpl / 1e6:9.1f} MB   HBM {(fpl + wpl) / 1e6:9.1f} MB")
    open(prefix + ".json", "w").write(json.dumps(out, indent=1))
    open(prefix + ".txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
