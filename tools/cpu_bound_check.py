"""How much of a step is host-side enqueue time?  Prints enqueue ms/step (no sync inside) and total ms/step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mivp_amd
from mivp_amd import train
from mivp_amd.swin_unetr import SwinUnetR

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
conf, size, batch = train.make_conf(wl)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SwinUnetR(conf).to(dev).train()
opt = train.build_optimizer(model, conf)
x, y = train.synthetic_batch(conf, batch, size, dev)
for _ in range(5):
    train.train_step(model, opt, conf, x, y)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        train.train_step(model, opt, conf, x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{wl}: enqueue {1e3 * (t1 - t0) / steps:.2f} ms/step, total {1e3 * (t2 - t0) / steps:.2f} ms/step", flush=True)
