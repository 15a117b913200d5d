"""Which torch operations of one eager VQ-VAE train step synchronise the host with the device
(torch.cuda.set_sync_debug_mode("warn")): every one is a point the host cannot enqueue past."""
import os, sys, warnings, collections, traceback
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd")); sys.path.insert(0, REPO)
from utils import config as C
from utils.commons import get_model, get_optimizer
from oracle import vqvae_oracle as orc
cfg = C.merge(C.load(os.path.join(REPO, "speech-masters-thesis_amd/configs/models/vqvae_k1024.yaml")),
              C.load(os.path.join(REPO, "speech-masters-thesis_amd/configs/datasets/synthetic_ljspeech.yaml")),
              C.create({"train": {"batch_size": 2, "n_gpus": 1, "ema": True, "grad_clip_norm": None, "seed": 0, "log_dir": "/tmp/x", "total_epochs": 1}}))
torch.manual_seed(0)
model, ema = get_model(cfg, "cuda:0")
opt, sched = get_optimizer(cfg, model)
x = orc.synthetic_clip_batch(2, 32768, 5).cuda(); lens = torch.tensor([32768, 20000]).cuda()
model.train()
def step():
    opt.zero_grad(set_to_none=True)
    loss_dict, _ = model.supervised_step((None, None, None, None, x, lens, None))
    loss_dict["loss"].backward()
    opt.step(); sched.step()
    if ema is not None: ema.step()
for _ in range(2): step()
torch.cuda.synchronize()
seen = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f"{os.path.basename(s.filename)}:{s.lineno}" for s in traceback.extract_stack()[:-2] if "speech-masters-thesis_amd" in s.filename or "tools/" in s.filename]
    seen[(str(message)[:60], " <- ".join(reversed(st[-4:])))] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
for (m, st), n in seen.most_common(): print(n, m, "|", st)
print("synchronising operations in one step:", sum(seen.values()))
