"""Which host ops launch the per-step fill kernels (torch profiler, one train step)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
import bench
import train as trainlib
args = bench.parse()
device = torch.device("cuda", 0)
from utils.commons import get_model, get_optimizer
cfg = bench.make_config(args)
model, ema = get_model(cfg, device, 0)
optimizer, scheduler = get_optimizer(cfg, model)
model.train()
pool = bench.synthetic_batches(1, args.batch, args.clip_len, 0, device)
def step(i):
    return trainlib.train_step(global_step=i, batch=pool[0], config=cfg, model=model, ema=ema, optimizer=optimizer,
                               scheduler=scheduler, device=device, rank=0, grad_sync=None)
for i in range(2): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::full")]
import collections
cnt = collections.Counter()
for e in evs:
    par = e.cpu_parent
    chain = []
    while par is not None and len(chain) < 4:
        chain.append(par.name); par = par.cpu_parent
    stack = [fr for fr in (e.stack or []) if "site-packages/torch" not in fr][:3]
    cnt[(e.name, tuple(chain), tuple(stack))] += 1
for (name, chain, stack), c in cnt.most_common(12):
    print(c, name, "<-", " <- ".join(chain))
    for fr in stack: print("      ", fr)
