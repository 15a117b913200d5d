"""How long the host needs to ENQUEUE one forward + backward (no sync inside) vs how long the GPU needs to run it."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
import bench
args = bench.parse()
device = torch.device("cuda", 0)
from utils.commons import get_model, get_optimizer
cfg = bench.make_config(args)
model, ema = get_model(cfg, device, 0)
optimizer, scheduler = get_optimizer(cfg, model)
model.train()
pool = bench.synthetic_batches(2, args.batch, args.clip_len, 0, device)
for i in range(6):
    optimizer.zero_grad()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss_dict, _ = model.supervised_step(pool[i % 2])
    t1 = time.perf_counter()
    loss_dict["loss"].backward()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    optimizer.step()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"iter {i}: host enqueue fwd {1e3*(t1-t0):6.1f} ms, bwd {1e3*(t2-t1):6.1f} ms; GPU done {1e3*(t3-t0):6.1f} ms after start; optimizer {1e3*(t4-t3):5.1f} ms", flush=True)
