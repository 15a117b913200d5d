#!/usr/bin/env python3
"""Headline benchmark: LJSpeech-shaped utterances/sec of the VQ-VAE train step.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full train step (train.py:train_step: zero_grad, forward, NaN guard, backward,
gradient all-reduce, AdamW, scheduler, parameter-EMA hook) on one batch of 32 synthetic
22.05 kHz clips of 145,408 samples per GPU (BASELINE.json configs[1]: codebook 1024, bf16).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "speech-masters-thesis_amd")
for _p in (PKG, REPO):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CLIP_LEN = 145408
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--workload", choices=["vqvae", "transformer_lm", "aux", "glow_tts"], default="vqvae",
                   help="vqvae = the headline metric (BASELINE.json); transformer_lm = the SURVEY 8(f2) train step, reported "
                        "in the same format under its own metric name; aux = the SURVEY 8(f1)/(f3)/(f4) paths (encode-only "
                        "pass, STFT.inverse, monotonic alignment search), each with its roofline and its CPU leg")
    p.add_argument("--tts_batch", type=int, default=32, help="glow_tts workload: utterances per GPU (scripts/train_glow_tts.sh trains with 32)")
    p.add_argument("--lm_batch", type=int, default=8, help="sequences per GPU (scripts/train_transformer_lm.sh: 8)")
    p.add_argument("--lm_len", type=int, default=258, help="tokens per sequence (<bos> + 256 codes + pad)")
    p.add_argument("--lm_tune_gemm", action="store_true",
                   help="transformer_lm workload: let PyTorch's TunableOp pick the hipBLASLt / rocBLAS solution of every GEMM shape "
                        "during warm-up (a few seconds; still plain library GEMMs)")
    p.add_argument("--lm_graph", action="store_true",
                   help="transformer_lm workload: replay forward + backward as one captured hipGraph (smt_amd/graph.py; single "
                        "GPU; no per-kernel events inside a graph)")
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=32, help="clips per GPU")
    p.add_argument("--model", type=str, default="vqvae_k1024")
    p.add_argument("--clip_len", type=int, default=CLIP_LEN)
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_kernel_events", action="store_true",
                   help="do not bracket kernels with HIP events in the timed region (no roofline objects)")
    p.add_argument("--event_every", type=int, default=5,
                   help="bracket every native call with HIP events on every N-th timed step (events serialise the "
                        "kernels around them: ~4 %% of the step when every step is instrumented)")
    p.add_argument("--cpu_clip_len", type=int, default=CLIP_LEN // 2,
                   help="clip length of the bounded CPU-baseline sample (half an utterance keeps 3 + 5 steps near 20-30 s)")
    p.add_argument("--cpu_steps", type=int, default=10, help="timed steps of the CPU-baseline leg (SURVEY 8(d): 3 warm-up + 10)")
    p.add_argument("--no_fp32", action="store_true", help="skip the secondary fp32 (parity path) measurement")
    p.add_argument("--no_graph", action="store_true", help="skip the secondary hipGraph-replay measurement")
    p.add_argument("--no_ragged", action="store_true", help="skip the secondary ragged-length run (SURVEY 8(d))")
    p.add_argument("--ragged_steps", type=int, default=6)
    p.add_argument("--no_micro", action="store_true", help="skip the VQ micro-benchmark and the mel-STFT kernel measurement")
    p.add_argument("--graph_leg", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--fp32_steps", type=int, default=3)
    return p.parse_args(argv)


def make_config(args):
    from utils import config as C
    cfg = C.merge(C.load(os.path.join(PKG, "configs/models", args.model + ".yaml")),
                  C.load(os.path.join(PKG, "configs/datasets/synthetic_ljspeech.yaml")),
                  C.create({"train": {"batch_size": args.batch, "n_gpus": args.gpus, "ema": False,
                                      "grad_clip_norm": None, "seed": 0, "log_dir": "/tmp/smt_bench"}}))
    return cfg


def synthetic_batches(n_batches, batch, length, rank, device):
    """Seeded clips of SURVEY 8(d): seed = 1000*rank + step."""
    from datasets.synthetic import synth_clip
    out = []
    for step in range(n_batches):
        clips = torch.stack([synth_clip(length, (1000 * rank + step) * 64 + i) for i in range(batch)])
        lens = torch.full((batch,), length, dtype=torch.long)
        out.append([None, None, None, None, clips.unsqueeze(1).to(device), lens.to(device), None])
    return out


def usable_cpus():
    """CPUs this process may really use: the scheduler affinity capped by the cgroup CPU quota (a GPU box shows 256
    logical CPUs but grants one GPU's share, 16: more threads than that are throttled -- 128 threads ran 9x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path, oracle/vqvae_oracle.py) timed on this box's host cores on
    a bounded sample, protocol of SURVEY 8(d) / BASELINE.md 3: one thread per usable core, 3 warm-up + 10 timed train
    steps, median.  The sample is ONE clip of --cpu_clip_len samples per step (default half an utterance, so the leg
    stays near 20-30 s); the rate is converted to full-length utterances/s by the sample ratio (the conv stacks, which
    are the step, cost the same per sample at any length)."""
    import statistics
    from oracle import vqvae_oracle as orc
    from utils import config as C
    mcfg = C.load(os.path.join(PKG, "configs/models", args.model + ".yaml")).model
    cfg = orc.VQVAEConfig.from_dict(mcfg.to_dict())
    cores = usable_cpus()
    torch.set_num_threads(cores)
    threads = torch.get_num_threads()
    trainer = orc.OracleTrainer(cfg, seed=0)
    x = orc.synthetic_clip_batch(1, args.cpu_clip_len, 123)
    lens = torch.tensor([args.cpu_clip_len])
    for _ in range(3):
        trainer.step(x, lens)
    times = []
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        trainer.step(x, lens)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    frac = args.cpu_clip_len / float(args.clip_len)
    return {"value": frac / med, "unit": "utterances/s", "cores": cores, "threads": threads,
            "logical_cpus_visible": os.cpu_count(), "kind": "port",
            "sample": f"oracle train step (fp32, torch-CPU), batch 1 x {args.cpu_clip_len} samples "
                      f"({frac:.3f} of a {args.clip_len}-sample utterance), 3 warm-up + {args.cpu_steps} timed steps, median "
                      f"{med:.2f} s/step (min {min(times):.2f}, max {max(times):.2f}); value = {frac:.3f} / median"}


def lm_cpu_baseline(args):
    """The TransformerLM oracle (oracle/lm_oracle.py: the reference's forward restated op by op, torch-CPU fp32, autograd
    backward) on this box's host cores: one warm-up + 3 timed forward + backward passes of the SAME batch shape, median."""
    import statistics
    from oracle import lm_oracle as lmo
    cores = usable_cpus()
    torch.set_num_threads(cores)
    p = {k: v.requires_grad_(True) for k, v in lmo.init_params(512, 512, 16, 2048, 12, seed=0).items()}
    x, lens = lmo.synthetic_tokens(args.lm_batch, args.lm_len, 512, seed=1, ragged=False)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        logits = lmo.lm_logits(x, lens, p, heads=16, num_layers=12, drop=lmo.CounterDropout(seed=i, p=0.1))
        loss, _ = lmo.lm_loss(x, logits)
        loss.backward()
        for v in p.values():
            v.grad = None
        if i:
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": args.lm_batch * args.lm_len / med, "unit": "tokens/s", "cores": cores, "threads": torch.get_num_threads(),
            "kind": "port", "sample": f"oracle forward + backward (no optimizer), fp32 torch-CPU, {args.lm_batch} x {args.lm_len} "
                                      f"tokens, 1 warm-up + 3 timed, median {med:.2f} s"}


def lm_main(args, rank, world, device, rehearsal):
    """`--workload transformer_lm`: tokens/s of the TransformerLM train step on the reference's
    configuration (configs/models/transformer_lm.yaml) and batch shape, same protocol as the headline run."""
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_lm
    from smt_amd import native, profiler
    from smt_amd.dist import GradSync
    native.lib()
    if args.lm_tune_gemm:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(30)
        tunable.set_filename(os.path.join(tempfile.gettempdir(), f"smt_lm_tunableop_{os.getpid()}.csv"))
    with tempfile.TemporaryDirectory() as tmp:
        model, optimizer, scheduler = bench_lm.build(tmp, "fp32", device)
    grad_sync = GradSync([p for p in model.parameters() if p.requires_grad], timing=True) if world > 1 else None
    g = torch.Generator().manual_seed(1 + rank)
    pool = []
    for _ in range(4):
        x = torch.randint(2, 514, (args.lm_batch, args.lm_len), generator=g)
        x[:, 0], x[:, -1] = 1, 0
        pool.append((x.to(device), torch.full((args.lm_batch,), args.lm_len - 1).to(device)))
    model.train()
    graphed = None
    if args.lm_graph:
        if world != 1:
            sys.exit("bench.py --lm_graph is a single-GPU mode (the gradient exchange is issued from autograd hooks)")
        from smt_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, optimizer, scheduler, pool[0][0], pool[0][1])
        args.no_kernel_events = True

    def step(i):
        x, lens = pool[i % len(pool)]
        if graphed is not None:
            return graphed.step(x, lens)
        if grad_sync is not None:
            grad_sync.zero_grad()                 # gradients live in the flat all-reduce buffer
        else:
            optimizer.zero_grad(set_to_none=True)
        out, _ = model(x, lens, None, None)
        out["loss"].backward()
        if grad_sync is not None:
            grad_sync.finish()
        optimizer.step()
        scheduler.step()
        return out["loss"]

    for i in range(args.warmup):
        step(i)
    profiler.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_prof = 0
    for i in range(args.steps):
        prof = (not args.no_kernel_events) and i % max(1, args.event_every) == 0
        profiler.enable(prof)
        n_prof += int(prof)
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    profiler.enable(False)
    if world > 1:
        mine = torch.tensor([elapsed], device=device)
        every_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_rank, mine)
        elapsed = max(t.item() for t in every_rank)
    if rank == 0:
        kernels = profiler.summary()
        dom = next((k for k in kernels if k["name"] == "lm_attention:bwd"), None)
        roofline = None if dom is None else {
            "kernel": "lm_attention:bwd (lm_attn_dq_kernel + lm_attn_dkv_kernel)", "bound": "mfma", "achieved": dom["achieved"],
            "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"], "traffic": None, "avg_us": dom["avg_us"],
            "note": "largest hand-written kernel of the step (the dense projections are hipBLASLt f32 GEMMs); algorithmic FLOPs = "
                    "5 products x 64 per visible (query, key) pair against the f32-input MFMA peak"}
        line = {"metric": "TransformerLM train tokens/sec (SURVEY 8(f2); not the BASELINE.json headline)",
                "value": args.lm_batch * args.lm_len * world * args.steps / elapsed, "unit": "tokens/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "rehearsal_shared_gpu": bool(rehearsal), "hip_graph": bool(args.lm_graph), "tuned_gemm": bool(args.lm_tune_gemm),
                "config": {"workload": "models/transformer_lm (12 x d512, 16 heads, ff 2048, dropout 0.1, CE), "
                                       f"batch {args.lm_batch}/GPU x {args.lm_len} tokens, fp32, AdamW",
                           "global_batch": args.lm_batch * world, "seq_len": args.lm_len, "parallelism": f"dp{world}"},
                "loss": float(loss.detach()), "roofline": roofline, "kernels": kernels, "kernel_event_steps": n_prof}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = lm_cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def glow_main(args, rank, world, device):
    """`--workload glow_tts`: utterances/s of the GlowTTS train step (BASELINE.json configs[4]) on the reference's
    configuration (configs/models/glow_tts.yaml: 6-layer relative-attention encoder, 12 flow blocks x 4 WN layers over 80 mels
    x n_sqz 2), synthetic LJSpeech-shaped (token, mel) pairs (configs/datasets/synthetic_tts.yaml), fp32, AdamW + Noam, dropout
    on; data parallel like the headline run.  CPU leg: the oracle's forward + backward on this box's host cores."""
    from smt_amd import native, profiler
    from smt_amd.dist import GradSync
    native.lib()
    from utils import config as C
    from utils.commons import get_model, get_optimizer
    from datasets.synthetic import SyntheticTTS
    import train as trainlib
    cfg = C.merge(C.load(os.path.join(PKG, "configs/models/glow_tts.yaml")), C.load(os.path.join(PKG, "configs/datasets/synthetic_tts.yaml")),
                  C.create({"train": {"batch_size": args.tts_batch, "n_gpus": world, "ema": False, "grad_clip_norm": None, "seed": 0,
                                      "log_dir": "/tmp/smt_bench_tts"}}))
    torch.manual_seed(0)
    model, ema = get_model(cfg, device, rank)
    optimizer, scheduler = get_optimizer(cfg, model)
    grad_sync = GradSync([p for p in model.parameters() if p.requires_grad], timing=True) if world > 1 else None
    ds = SyntheticTTS(cfg, "train")
    pool, frames, tokens = [], 0, 0
    for bi in range(4):
        items = [ds[(rank * 4 + bi) * args.tts_batch + i] for i in range(args.tts_batch)]
        batch = SyntheticTTS.collate(items)
        frames += int(batch[3].sum()); tokens += int(batch[1].sum())
        pool.append([b.to(device) if torch.is_tensor(b) else b for b in batch])
    model.train()

    def step(i):
        return trainlib.train_step(global_step=i, batch=pool[i % len(pool)], config=cfg, model=model, ema=ema, optimizer=optimizer,
                                   scheduler=scheduler, device=device, rank=rank, grad_sync=grad_sync)
    for i in range(args.warmup):
        step(i)
    profiler.reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_prof = 0
    for i in range(args.steps):
        prof = (not args.no_kernel_events) and i % max(1, args.event_every) == 0
        profiler.enable(prof); n_prof += int(prof)
        loss_dict, _ = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    profiler.enable(False)
    if world > 1:
        mine = torch.tensor([elapsed], device=device)
        every_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_rank, mine)
        elapsed = max(t.item() for t in every_rank)
    if rank == 0:
        kernels = profiler.summary()
        convs = [k for k in kernels if k["name"].startswith(("conv_gemm", "conv_wgrad"))]
        tot_us = sum(k["total_ms"] for k in convs) * 1e3
        flops = sum(k["alg_flops"] * k["launches"] for k in convs)
        roofline = None if not convs or tot_us == 0 else {
            "kernel": "smt::conv_gemm_kernel<float> / conv_wgrad_kernel<float> (every convolution of the model, forward + both gradients)",
            "bound": "mfma", "achieved": flops / tot_us * 1e-6, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / tot_us * 1e-6 / F32_MFMA_PEAK_TFLOPS, "traffic": None, "ms_per_step": tot_us * 1e-3 / max(1, n_prof),
            "note": "fp32 matrix pipe (v_mfma_f32_32x32x2_f32), 160-channel flows and 192-channel hidden layers on 128-row tiles; the step "
                    "is ~2,000 small launches and host-paced"}
        line = {"metric": "GlowTTS train utterances/sec (SURVEY 8(f4), BASELINE.json configs[4]; not the headline)",
                "value": args.tts_batch * world * args.steps / elapsed, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"models/glow_tts (configs/models/glow_tts.yaml), batch {args.tts_batch}/GPU, mean {tokens / (4 * args.tts_batch):.0f} tokens / "
                                       f"{frames / (4 * args.tts_batch):.0f} mel frames per utterance, fp32, AdamW + Noam, dropout on",
                           "global_batch": args.tts_batch * world, "parallelism": f"dp{world}"},
                "loss": float(loss_dict["loss"].detach()), "native_launches_per_step": sum(k["launches"] for k in kernels) / max(1, n_prof),
                "native_kernel_ms_per_step": sum(k["total_ms"] for k in kernels) / max(1, n_prof), "roofline": roofline, "kernels": kernels}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = glow_cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def glow_cpu_baseline(cfg):
    """The GlowTTS oracle (oracle/glow_oracle.py: the reference's forward restated, torch-CPU fp32, autograd backward, numpy
    alignment search as in the reference) on this box's host cores: 4 utterances, one warm-up + 3 timed passes, median."""
    import statistics
    from oracle import glow_oracle as go
    cores = usable_cpus()
    torch.set_num_threads(cores)
    m = cfg.model.to_dict()
    ocfg = dict(encoder=m["encoder"], decoder=m["decoder"], zero_out=False)
    params = {k: v.requires_grad_(True) for k, v in go.init_params(ocfg, 149, 80, seed=0).items()}
    tokens, x_lens, y, y_lens = go.synthetic_batch(4, 120, 620, 149, 80, seed=1)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        out, _ = go.glow_tts_forward(tokens, x_lens, y, y_lens, params, ocfg, True)
        out["loss"].backward()
        for v in params.values():
            v.grad = None
        if i:
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": 4 / med, "unit": "utterances/s", "cores": cores, "threads": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle forward + backward (no optimizer, dropout off), fp32 torch-CPU, 4 utterances of <= 120 tokens / 620 frames, "
                      f"1 warm-up + 3 timed, median {med:.2f} s"}


def _timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps        # ms per call, on the launch stream


def aux_main(args, device):
    """`--workload aux`: the paths either side of the train step that SURVEY 8(f) names, single GPU.
    (f1) encode-only pass of scripts/generate_vq_dataset.py (encoder + exact VQ search, no gradients): utterances/s;
    (f3) STFT.inverse (datasets/transforms.py:125-156) at the dataset's analysis parameters: GB/s of algorithmic traffic;
    (f4) GlowTTS maximum_path (models/glow_tts/submodules.py:28-67) at LJSpeech-like shapes: lattice cells/s.
    CPU legs: the oracle restatements on this box's host cores (numpy / torch-CPU), bounded samples."""
    import numpy as np
    from smt_amd import native
    native.lib()
    from utils.commons import get_model
    from datasets.transforms import STFT
    from models.glow_tts.submodules import maximum_path
    cores = usable_cpus()
    torch.set_num_threads(cores)
    out = {"metric": "auxiliary paths of SURVEY 8(f) (not the BASELINE.json headline)", "n_gpus": 1, "data": "synthetic",
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None}
    # ---- (f1) encode-only pass
    cfg = make_config(args)
    model, _ = get_model(cfg, device, 0)
    model.eval()
    pool = synthetic_batches(1, args.batch, args.clip_len, 0, device)
    x, lens = pool[0][4], pool[0][5]
    ms = _timed(lambda: model.encode_and_quantize(x, lens), args.steps, args.warmup)
    out["encode_only"] = {"value": args.batch / ms * 1e3, "unit": "utterances/s", "ms_per_batch": ms, "dtype": cfg.model.get("compute_dtype", "fp32"),
                          "config": {"workload": f"VQVAE.encode_and_quantize, {args.model}, batch {args.batch} x {args.clip_len} samples"}}
    del model
    # ---- (f3) STFT.inverse
    b, n_fft, hop, frames = args.batch, 1024, 256, args.clip_len // 256 + 1
    g = torch.Generator().manual_seed(3)
    mag = torch.rand(b, n_fft // 2 + 1, frames, generator=g).to(device)
    ph = ((torch.rand(b, n_fft // 2 + 1, frames, generator=g) - 0.5) * 6.28).to(device)
    stft = STFT(n_fft=n_fft, hop_length=hop, win_length=n_fft, window="hann").to(device)
    ms = _timed(lambda: stft.inverse(mag, ph), args.steps, args.warmup)
    t_out = stft.inverse(mag, ph).shape[-1]
    nbytes = 2 * mag.numel() * 4 + b * t_out * 4
    leg = {}
    if not args.no_cpu_baseline:
        from oracle import vqvae_oracle as orc
        m1, p1 = mag[:2].cpu(), ph[:2].cpu()
        orc.stft_inverse(m1, p1, n_fft, hop, n_fft)
        t0 = time.perf_counter()
        orc.stft_inverse(m1, p1, n_fft, hop, n_fft)
        dt = time.perf_counter() - t0
        leg = {"value": 2 * (2 * m1[0].numel() * 4 + t_out * 4) / dt * 1e-9, "unit": "GB/s", "cores": cores, "kind": "port",
               "sample": f"oracle stft_inverse, 2 clips of {frames} frames, one timed call {dt:.3f} s (includes rebuilding the "
                         "pseudo-inverse basis, which the reference builds once in STFT.__init__)"}
    out["stft_inverse"] = {"value": nbytes / ms * 1e-6, "unit": "GB/s", "ms_per_call": ms, "dtype": "f32",
                           "roofline": {"bound": "hbm", "achieved": nbytes / ms * 1e-6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": nbytes / ms * 1e-6 / HBM_PEAK_GBS, "traffic": None},
                           "config": {"workload": f"STFT.inverse n_fft {n_fft} hop {hop}, batch {b} x {frames} frames"},
                           "cpu_baseline": leg or None}
    # ---- (f4) monotonic alignment search
    bb, t_x, t_y = 32, 160, 800
    value = (torch.randn(bb, t_x, t_y, generator=g) * 3).to(device)
    xl = torch.randint(t_x // 2, t_x + 1, (bb,), generator=g)
    yl = torch.randint(t_y // 2, t_y + 1, (bb,), generator=g)
    mask = ((torch.arange(t_x)[None, :, None] < xl[:, None, None]) & (torch.arange(t_y)[None, None, :] < yl[:, None, None])).float().to(device)
    ms = _timed(lambda: maximum_path(value, mask), args.steps, args.warmup)
    cells = bb * t_x * t_y
    leg = {}
    if not args.no_cpu_baseline:
        from oracle import mas_oracle
        v, m = value.cpu().numpy(), mask.cpu().numpy()
        t0 = time.perf_counter()
        ref = mas_oracle.maximum_path(v, m)
        dt = time.perf_counter() - t0
        same = bool(np.array_equal(ref, maximum_path(value, mask).cpu().numpy()))
        leg = {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
               "sample": f"oracle maximum_path (the reference's numpy algorithm) on the same batch, one call {dt:.3f} s; paths identical: {same}"}
    out["maximum_path"] = {"value": cells / ms * 1e3, "unit": "cells/s", "ms_per_call": ms, "dtype": "f32",
                           "roofline": {"bound": "hbm", "achieved": 12 * cells / ms * 1e-6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": 12 * cells / ms * 1e-6 / HBM_PEAK_GBS, "traffic": None,
                                        "note": "latency-bound by construction: t_y sequential lattice columns with a barrier each; the "
                                                "fraction is against the HBM bound its 12 B/cell would allow"},
                           "config": {"workload": f"maximum_path batch {bb}, t_x {t_x}, t_y {t_y}, ragged"}, "cpu_baseline": leg or None}
    print(json.dumps(out), flush=True)


def ragged_lengths(n, seed=0):
    """SURVEY 8(d) secondary run: clip lengths ~ U[24,064, 222,720] rounded down to a multiple of 512 (the dataset's rule,
    reference datasets/ljspeech.py:14,82), seeded."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(24064, 222720 + 1, (n,), generator=g) // 512) * 512


def timed_ragged(args, step_fn_factory, rank, device, note):
    """Secondary measurement: the SAME train step on ragged batches -- lengths U[24,064, 222,720] x 512, seed 0, each batch
    zero-padded to its own longest clip exactly as the reference's collate does (datasets/ljspeech.py:135-138) -- to show what
    padding costs.  Four distinct batches are cycled; `padded_fraction` = padded samples / processed samples."""
    from datasets.synthetic import synth_clip
    n_batches = 4
    lens_all = ragged_lengths(n_batches * args.batch, seed=0).view(n_batches, args.batch)
    pool, real, padded = [], 0, 0
    for bi in range(n_batches):
        lens = lens_all[bi]
        tmax = int(lens.max())
        clips = torch.zeros(args.batch, tmax)
        for i in range(args.batch):
            clips[i, :int(lens[i])] = synth_clip(int(lens[i]), 900000 + (1000 * rank + bi) * 64 + i)
        pool.append([None, None, None, None, clips.unsqueeze(1).to(device), lens.to(device), None])
        real += int(lens.sum()); padded += args.batch * tmax
    step = step_fn_factory(pool)
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.ragged_steps):
        step(2 + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    utt = args.batch * args.ragged_steps / el
    # fixed-length equivalent: utterances of the headline length carrying the same number of REAL samples per second
    mean_len = real / (n_batches * args.batch)
    note(f"ragged run: {utt:.1f} utt/s, padded fraction {1 - real / padded:.3f}")
    return {"value": utt, "unit": "utterances/s", "ms_per_step": el / args.ragged_steps * 1e3, "steps": args.ragged_steps,
            "warmup": 2, "mean_len": mean_len, "max_len": int(lens_all.max()), "min_len": int(lens_all.min()),
            "padded_fraction": 1 - real / padded, "real_samples_per_s": utt * mean_len,
            "note": "lengths ~ U[24064, 222720] // 512 * 512, torch.Generator seed 0, 4 batches of "
                    f"{args.batch} cycled, each padded to its own maximum (reference datasets/ljspeech.py:135-138); "
                    "padded_fraction = 1 - real samples / processed samples"}


def vq_micro(device, note):
    """SURVEY 8(d) VQ micro-benchmark: x ~ N(0,1) [N,128] seed 0, k ~ N(0,1) [K,128] seed 1, K in {256, 1024}, N in
    {4,544; 36,352}, all-ones mask; smt_vq_forward (search + candidates + exact + reduce, x_d written) timed with HIP events
    on the launch stream.  Reported against BOTH bounds the survey names: HBM by algorithmic bytes (1,036 B/vector with x_d
    + the codebook once) and the fp32-FMA roof by 2 N K D FLOP; plus the bf16-MFMA fraction of the 3 x 2 N K D FLOP the
    filter really issues, and the number of rows that had to be re-scored exactly (fp64 path)."""
    from smt_amd import vq
    out = []
    for k_bins in (256, 1024):
        for n in (4544, 36352):
            x = torch.randn(n, 128, generator=torch.Generator().manual_seed(0)).to(device)
            cb = torch.randn(k_bins, 128, generator=torch.Generator().manual_seed(1)).to(device)
            prep = vq.prepare(cb)
            sums = vq.vq_forward_raw(x, cb, None, prep=prep)[3]
            ms = _timed(lambda: vq.vq_forward_raw(x, cb, None, prep=prep), 50, 5)
            us = ms * 1e3
            nbytes = n * (4 * 128 + 8 + 4 + 4 * 128) + 4 * k_bins * 128
            flops = 2.0 * n * k_bins * 128
            out.append({"K": k_bins, "N": n, "us": us, "exact_rows": int(sums[3].item()),
                        "hbm": {"achieved": nbytes / us * 1e-3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / us * 1e-3 / HBM_PEAK_GBS,
                                "alg_bytes": nbytes},
                        "f32_fma": {"achieved": flops / us * 1e-6, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": flops / us * 1e-6 / F32_MFMA_PEAK_TFLOPS},
                        "bf16_mfma": {"achieved": 3 * flops / us * 1e-6, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": 3 * flops / us * 1e-6 / BF16_MFMA_PEAK_TFLOPS}})
    note("vq micro: " + ", ".join(f"K{r['K']}xN{r['N']} {r['us']:.0f} us" for r in out))
    return out


def melspec_roofline(args, audio, device, note):
    """The first kernel north_star names: windowed STFT + mel-filterbank contraction + log (reference MelSpectrogram.forward,
    datasets/transforms.py:48-65; n_fft 1024, hop 256, 80 mels, 22.05 kHz, 0-8 kHz: configs/datasets/ljspeech.yaml) on the
    headline batch (32 x 145,408 samples), one launch, HIP events on the launch stream.  Algorithmic bytes: 4 B/sample in +
    80 x 4 / 256 = 1.25 B/sample out = 5.25 B/sample."""
    from datasets.transforms import MelSpectrogram
    from smt_amd import spectral
    mel = MelSpectrogram(n_fft=1024, hop_length=256, win_length=1024, n_mels=80, sample_rate=22050, f_min=0.0, f_max=8000.0).to(device)
    x = audio.reshape(audio.shape[0], -1).contiguous()
    out = mel(x)
    # the kernel alone: MelSpectrogram.forward also asserts the [-1, 1] range (two reductions and a host sync, transforms.py:49)
    ms = _timed(lambda: spectral.log_mel(x, mel.mel_basis, mel.band, 1024, 256, 1024), 50, 5)
    us = ms * 1e3
    nbytes = x.numel() * 4 + out.numel() * 4
    note(f"melspec: {us:.1f} us = {nbytes / us * 1e-3:.0f} GB/s")
    return {"kernel": "smt::melspec_wave_kernel (windowed STFT + mel contraction + log, one launch)", "bound": "hbm",
            "achieved": nbytes / us * 1e-3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / us * 1e-3 / HBM_PEAK_GBS,
            "avg_us": us, "launches_timed": 50, "alg_bytes_per_launch": nbytes, "bytes_per_sample": nbytes / x.numel(),
            "frames": int(out.shape[-1]), "traffic": None,
            "note": f"batch {x.shape[0]} x {x.shape[1]} samples; not on the VQ-VAE train step (SURVEY D1), measured stand-alone; "
                    "VALU-bound (in-LDS FFT), frac is against the HBM bound of its algorithmic bytes"}


def launch_or_none(args, argv):
    """`python bench.py --gpus N` with N > 1 outside torchrun: start N rank processes of this script (fresh
    interpreters; this parent makes no HIP call) and return their exit code.  None = run in this process."""
    from smt_amd import launcher
    if launcher.under_launcher() or args.gpus <= 1:
        return None
    visible = torch.cuda.device_count()          # does not initialise the GPU on this image
    if visible < args.gpus and os.environ.get("SMT_BENCH_REHEARSAL") == "1" and visible >= 1:
        visible = args.gpus                       # rehearsal of the N > 1 path on fewer cards (ranks share GPUs over gloo)
    if visible < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {visible} GPU(s) visible on this node", file=sys.stderr)
        return 2
    return launcher.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + list(argv))


def graph_leg_main(args, device):
    """Child process of `timed_graph`: the SAME train step with forward + backward replayed from one captured hipGraph
    (smt_amd/graph.py; dropout keys in device memory, packed weights refreshed inside the graph); NaN guard, AdamW, scheduler
    and the parameter-EMA hook stay eager, exactly as in train.train_step.  Prints one small JSON object."""
    from smt_amd import native
    native.lib()
    from smt_amd.graph import GraphedStep
    from utils.commons import get_model, get_optimizer
    from utils.train_utils import seed_all_rng
    import train as trainlib
    cfg = make_config(args)
    seed_all_rng(cfg.train.seed)
    model, ema = get_model(cfg, device, 0)
    optimizer, scheduler = get_optimizer(cfg, model)
    model.train()
    pool = synthetic_batches(min(4, args.steps + args.warmup), args.batch, args.clip_len, 0, device)
    for i in range(2):      # eager steps first: codebook initialisation and every lazily built buffer
        trainlib.train_step(global_step=i, batch=pool[i % len(pool)], config=cfg, model=model, ema=ema, optimizer=optimizer,
                            scheduler=scheduler, device=device, rank=0, grad_sync=None)
    graph = GraphedStep(model, lambda *slots: model.supervised_step(list(slots)), pool[0],
                        lambda: optimizer.zero_grad(set_to_none=True), warmup=1)

    def step(i):
        loss_dict, _ = graph.replay(*pool[i % len(pool)])
        if torch.isnan(loss_dict["loss"]):
            raise RuntimeError("Nan detected in loss")
        optimizer.step()
        scheduler.step()
        ema.step()
        return loss_dict
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss_dict = step(args.warmup + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"value": args.batch * args.steps / el, "unit": "utterances/s", "ms_per_step": el / args.steps * 1e3,
                      "steps": args.steps, "warmup": args.warmup, "loss": float(loss_dict["loss"]),
                      "note": "same step in a fresh process, forward + backward as one captured hipGraph; optimizer / scheduler / "
                              "NaN guard / EMA hook eager"}), flush=True)


def timed_graph(args, note):
    """Secondary measurement, in a CHILD process (a crash inside graph capture must not take the headline line with it)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--graph_leg", "--steps", str(args.steps), "--warmup", "2", "--batch",
           str(args.batch), "--model", args.model, "--clip_len", str(args.clip_len)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            note(f"hipGraph leg failed (exit code {r.returncode})")
            return {"error": f"exit code {r.returncode}: " + r.stderr[-300:]}
        out = json.loads(lines[-1])
        note(f"hipGraph replay: {out['ms_per_step']:.1f} ms/step")
        return out
    except Exception as e:
        note(f"hipGraph leg failed: {e}")
        return {"error": str(e)[:300]}


def timed_fp32(args, pool, device, note):
    """Secondary measurement: the fp32 parity path (same model, same batch, compute_dtype fp32) for a few steps."""
    from utils.commons import get_model, get_optimizer
    from utils.train_utils import seed_all_rng
    import train as trainlib
    cfg = make_config(args)
    cfg.model.compute_dtype = "fp32"
    seed_all_rng(cfg.train.seed)
    model, ema = get_model(cfg, device, 0)
    optimizer, scheduler = get_optimizer(cfg, model)
    model.train()

    def step(i):
        return trainlib.train_step(global_step=i, batch=pool[i % len(pool)], config=cfg, model=model, ema=ema,
                                   optimizer=optimizer, scheduler=scheduler, device=device, rank=0, grad_sync=None)
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.fp32_steps):
        loss_dict, _ = step(1 + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    note(f"fp32 parity path: {el / args.fp32_steps * 1e3:.1f} ms/step")
    return {"value": args.batch * args.fp32_steps / el, "unit": "utterances/s", "ms_per_step": el / args.fp32_steps * 1e3,
            "steps": args.fp32_steps, "warmup": 1, "dtype": "fp32", "loss": float(loss_dict["loss"].detach()),
            "note": "same workload on the fp32 parity path (conv stacks on v_mfma_f32_32x32x2_f32)"}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    rc = launch_or_none(args, argv)
    if rc is not None:
        sys.exit(rc)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs MI355X GPUs (no CPU fallback for the hot path)")
    rehearsal = os.environ.get("SMT_BENCH_REHEARSAL") == "1" and torch.cuda.device_count() < world
    if rehearsal:
        local = local % torch.cuda.device_count()   # several ranks per card: RCCL cannot do that, gloo carries the tensors
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend="gloo" if rehearsal else "nccl", init_method="env://")

    if args.graph_leg:
        return graph_leg_main(args, device)
    if args.workload == "transformer_lm":
        return lm_main(args, rank, world, device, rehearsal)
    if args.workload == "glow_tts":
        return glow_main(args, rank, world, device)
    if args.workload == "aux":
        if world != 1:
            sys.exit("bench.py --workload aux is a single-GPU measurement")
        return aux_main(args, device)

    from smt_amd import native, profiler
    native.lib()  # fail loudly if the HIP library is missing
    from utils.commons import get_model, get_optimizer
    from utils.train_utils import seed_all_rng
    import train as trainlib

    cfg = make_config(args)
    seed_all_rng(cfg.train.seed)
    model, ema = get_model(cfg, device, rank)
    optimizer, scheduler = get_optimizer(cfg, model)
    grad_sync = None
    if world > 1:
        from smt_amd.dist import GradSync
        grad_sync = GradSync(model.parameters(), timing=True)
    model.train()

    t_start = time.perf_counter()
    pool = synthetic_batches(min(4, args.steps + args.warmup), args.batch, args.clip_len, rank, device)

    def step(i):
        return trainlib.train_step(global_step=i, batch=pool[i % len(pool)], config=cfg, model=model, ema=ema,
                                   optimizer=optimizer, scheduler=scheduler, device=device, rank=rank,
                                   grad_sync=grad_sync)

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    note(f"model + {len(pool)} resident batches ready")
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        note(f"warm-up step {i} done")
    profiler.reset()
    profiler.enable(False)
    if grad_sync is not None:
        grad_sync.exposed_ms()            # drop the warm-up steps' events
    every = max(1, args.event_every)
    n_prof = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        prof_step = (not args.no_kernel_events) and i % every == 0   # HIP events around every native call of this step
        profiler.enable(prof_step)
        n_prof += int(prof_step)
        loss_dict, _ = step(args.warmup + i)
    host_elapsed = time.perf_counter() - t0     # all launches enqueued (the host runs ahead of the GPU)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    profiler.enable(False)
    per_rank_ms = [elapsed / args.steps * 1e3]
    sync_ms = None
    if world > 1:
        mine = torch.tensor([elapsed], device=device)
        every_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_rank, mine)
        per_rank_ms = [t.item() / args.steps * 1e3 for t in every_rank]
        elapsed = max(t.item() for t in every_rank)            # MAX over ranks
        sync_ms = grad_sync.exposed_ms()

    if rank == 0:
        kernels = profiler.summary()
        m = cfg.model
        n_rows = args.batch * (args.clip_len // 128)

        pmc, pmc_source = {}, None
        for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            pmc_path = os.path.join(REPO, "profiles", name)
            if os.path.exists(pmc_path):
                with open(pmc_path) as f:
                    pmc = json.load(f)
                pmc_source = f"profiles/{name} (stored rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                             "benchmark, tools/collect_profiles.sh; not re-measured in this run)"
                break

        def group(prefixes, label, bound, dtype, pmc_key=None):
            recs = [k for k in kernels if any(k["name"] == q or k["name"].startswith(q + ":") or
                                              k["name"].startswith(q + "_k") for q in prefixes)]
            if not recs:
                return None
            launches = sum(k["launches"] for k in recs)
            total_us = sum(k["total_ms"] for k in recs) * 1e3
            flops = sum(k["alg_flops"] * k["launches"] for k in recs)
            nbytes = sum(k["alg_bytes"] * k["launches"] for k in recs)
            if bound == "mfma":
                achieved, peak, unit = flops / total_us * 1e-6, profiler.PEAK_TFLOPS[dtype], "TFLOP/s"
            else:
                achieved, peak, unit = nbytes / total_us * 1e-3, profiler.HBM_PEAK_GBS, "GB/s"
            # HBM bytes per launch from the rocprofv3 PMC passes (profiles/r01_pmc_traffic.json), if recorded
            traffic = pmc.get(pmc_key, {}).get("hbm_bytes_per_launch") if pmc_key else None
            return {"kernel": label, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
                    "frac": achieved / peak, "launches_per_step": launches / max(1, n_prof),
                    "avg_us": total_us / launches, "ms_per_step": total_us * 1e-3 / max(1, n_prof),
                    "alg_flops_per_launch": flops / launches, "alg_bytes_per_launch": nbytes / launches,
                    "traffic": traffic, "traffic_source": pmc_source if traffic is not None else None}

        dt = "bf16" if m.get("compute_dtype") == "bf16" else "f32"
        # dominant kernel by time: the weight-stationary implicit-GEMM conv kernel (dilated 128->128 convs, fwd + dgrad)
        roofline = group({"conv_ws", "conv_ws_pipe", "conv_ws2"}, "smt::conv_ws2_kernel / conv_ws_pipe_kernel / conv_ws_kernel (dilated 128->128 convs, forward + data gradient)",
                         "mfma", dt, "conv_ws_kernel")
        if roofline is None:   # fp32 configuration: everything runs on the generic kernel
            roofline = group({"conv_gemm"}, "smt::conv_gemm_kernel", "mfma", dt, "conv_gemm_kernel")
        extra_rooflines = {
            "conv_wgrad": group({"conv_wgrad", "conv_wgrad_shift", "conv_wgrad_dma"}, "smt::conv_wgrad_shift_kernel / conv_wgrad{,_dma}_kernel + reduce", "mfma",
                                dt, "conv_wgrad_shift_kernel"),
            "conv1x1_bwd": group({"conv1x1_bwd"}, "smt::conv1x1_bwd_kernel (fused K3 backward, HBM-bound)", "hbm", dt,
                                 "conv1x1_bwd_kernel"),
            "conv1x1_fold": group({"conv1x1_fold"}, "smt::conv1x1_fold_kernel (K3 + recomputed K1 residual, HBM-bound)",
                                  "hbm", dt, "conv1x1_fold_kernel"),
            "conv_k3gate": group({"conv_k3gate"}, "smt::conv_k3gate_kernel (K3 of the four branches + gate, HBM-bound)", "hbm",
                                 dt, "conv_k3gate_kernel"),
            "resample": group({"conv4s2", "convt4s2"}, "smt::conv4s2_kernel / convt4s2_kernel (k4 s2 resampling convs, HBM-bound)",
                              "hbm", dt, "resample"),
            "conv_k1act": group({"conv_k1act"}, "smt::conv_k1act_kernel (K1, activated output only, HBM-bound)", "hbm",
                                dt, "conv_k1act_kernel"),
            "conv1x1_c64": group({"conv1x1_c64"}, "smt::conv1x1_c64_kernel (gate conv forward + residual, HBM-bound)", "hbm", dt),
            "conv_k1_bwd": group({"conv_k1_bwd"}, "smt::conv_k1_bwd_kernel (fused K1 backward, HBM-bound)", "hbm", dt,
                                 "conv_k1_bwd_kernel"),
            "conv_gemm_dma": group({"conv_gemm_dma"}, "smt::conv_gemm_dma_kernel (LDS-DMA streaming kernel, small levels)",
                                   "mfma", dt, "conv_gemm_dma_kernel"),
            "conv1x1_dma": group({"conv1x1_dma"}, "smt::conv1x1_dma_kernel (persistent 1x1, HBM-bound)", "hbm", dt,
                                 "conv1x1_dma_kernel"),
            "conv_gemm": group({"conv_gemm"}, "smt::conv_gemm_kernel (register-staged generic path)", "mfma", dt,
                               "conv_gemm_kernel"),
            "vq_forward": group({"vq_forward"}, "smt_vq_forward (search, candidates, exact, reduce)", "hbm", "bf16",
                                "vq_forward"),
            "vq_ema_accumulate": group({"vq_ema_accumulate"}, "smt::vq_ema_accumulate_kernel", "hbm", "f32"),
            "gate_mix": group({"gate_mix_fwd", "gate_mix_bwd"}, "smt::gate_mix_{fwd,bwd}_kernel", "hbm", dt, "gate_mix"),
            "stft_loss": group({"stft_loss_fwd", "stft_loss_bwd"}, "smt::stft_loss_{fwd,bwd}_kernel", "hbm", "f32"),
        }
        extra_rooflines = {k: v for k, v in extra_rooflines.items() if v is not None}
        if "stft_loss" in extra_rooflines:
            extra_rooflines["stft_loss"]["note"] = ("not HBM-bound: one wave per frame, radix-8/16 in-LDS passes; the PMC pass shows the "
                                                    "vector ALU ~70 % busy (1,430 VALU instructions per 1,024-point frame). "
                                                    "frac is against the HBM bound its algorithmic bytes would allow")
        if "vq_forward" in extra_rooflines:
            # the north-star kernel against BOTH bounds: HBM by algorithmic bytes (above) and the bf16 matrix pipe by the
            # 3 x 2 N K D FLOP of its filter (its intensity, ~1500 FLOP/B at K = 1024, puts it on the MFMA side)
            v = extra_rooflines["vq_forward"]
            tf = v["alg_flops_per_launch"] / v["avg_us"] * 1e-6
            v["mfma"] = {"achieved": tf, "peak": profiler.PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                         "frac": tf / profiler.PEAK_TFLOPS["bf16"]}
        line = {
            "metric": "LJSpeech utterances/sec per VQ-VAE train step",
            "value": args.batch * world * args.steps / elapsed,
            "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "world_size": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() if world > 1 else None),
            "rehearsal_shared_gpu": bool(rehearsal),
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_per_rank": per_rank_ms,
            "grad_sync_exposed_ms_per_step": sync_ms,
            "native_launches_per_step": sum(k["launches"] for k in kernels) / max(1, n_prof),
            "native_kernel_ms_per_step": sum(k["total_ms"] for k in kernels) / max(1, n_prof),
            "host_enqueue_ms_per_step": host_elapsed / args.steps * 1e3,
            "kernel_event_steps": n_prof,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": m.get("compute_dtype", "fp32"), "data": "synthetic",
            "config": {"workload": f"configs[1]: models/vqvae codebook={m.l_bins}, batch={args.batch}/GPU, "
                                   f"{args.clip_len}-sample 22.05 kHz clips, {m.get('compute_dtype')} conv stacks, "
                                   f"fp32 VQ/losses/AdamW",
                       "global_batch": args.batch * world, "clip_len": args.clip_len, "codebook": m.l_bins,
                       "latent_rows_per_gpu": n_rows, "parallelism": f"dp{world}"},
            "loss": float(loss_dict["loss"].detach()),
            "roofline": roofline,
            "rooflines_other": extra_rooflines,
            "kernels": kernels,
        }
        note(f"timed region done: {line['value']:.2f} utt/s, {line['ms_per_step']:.1f} ms/step")
        for key in ("vq_forward", "vq_ema_accumulate"):        # spread over the timed steps, not only the mean
            rec = next((k for k in kernels if k["name"] == key), None)
            if rec is not None and key in extra_rooflines:
                extra_rooflines[key]["min_us"], extra_rooflines[key]["max_us"] = rec["min_us"], rec["max_us"]
        if world == 1 and not args.no_micro:
            line["rooflines_other"]["melspec"] = melspec_roofline(args, pool[0][4], device, note)
            mel_traffic = pmc.get("melspec", {}).get("hbm_bytes_per_launch")
            if mel_traffic is not None:
                line["rooflines_other"]["melspec"].update(traffic=mel_traffic, traffic_source=pmc_source)
            st_traffic = pmc.get("stft_loss", {}).get("hbm_bytes_per_launch")
            if st_traffic is not None and "stft_loss" in line["rooflines_other"]:
                line["rooflines_other"]["stft_loss"].update(traffic=st_traffic, traffic_source=pmc_source)
            line["vq_micro"] = vq_micro(device, note)
        if world == 1 and not args.no_ragged:
            def factory(rpool):
                return lambda i: trainlib.train_step(global_step=i, batch=rpool[i % len(rpool)], config=cfg, model=model, ema=ema,
                                                     optimizer=optimizer, scheduler=scheduler, device=device, rank=rank,
                                                     grad_sync=grad_sync)
            line["ragged"] = timed_ragged(args, factory, rank, device, note)
        if world == 1 and not args.no_graph:
            line["hip_graph"] = timed_graph(args, note)
        if world == 1 and not args.no_fp32 and m.get("compute_dtype") == "bf16":
            del model, optimizer, scheduler, ema
            torch.cuda.empty_cache()
            line["fp32"] = timed_fp32(args, pool, device, note)
        if world == 1 and not args.no_cpu_baseline:       # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(args)
            note("cpu baseline done")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
