"""How many latent rows of the bench model go through the exact fp64 re-scoring, and what each VQ kernel costs there."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
import bench
from smt_amd import vq
args = bench.parse()
device = torch.device("cuda", 0)
from utils.commons import get_model, get_optimizer
import train as trainlib
cfg = bench.make_config(args)
model, ema = get_model(cfg, device, 0)
optimizer, scheduler = get_optimizer(cfg, model)
model.train()
pool = bench.synthetic_batches(1, args.batch, args.clip_len, 0, device)
orig = vq.vq_forward_raw
def spy(x, cb, *a, **k):
    out = orig(x, cb, *a, **k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(x, cb, *a, **k); e.record(); torch.cuda.synchronize()
    print(f"vq_forward rows={x.shape[0]} K={cb.shape[0]} D={x.shape[1]}: queued for fp64 re-scoring = {int(out[3][3].item())}, {s.elapsed_time(e)*1e3:.0f} us", flush=True)
    return out
vq.vq_forward_raw = spy
for i in range(4):
    trainlib.train_step(global_step=i, batch=pool[0], config=cfg, model=model, ema=ema, optimizer=optimizer,
                        scheduler=scheduler, device=device, rank=0, grad_sync=None)
