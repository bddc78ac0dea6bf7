"""Which ops of a cfg1 training step end in Memcpy DtoD (copyBuffer)?  torch.profiler with stacks."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import mivp_amd
from mivp_amd import train
from mivp_amd.swin_unetr import SwinUnetR
from torch.profiler import profile, ProfilerActivity
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
conf, size, batch = train.make_conf(wl)
dev = torch.device("cuda")
torch.manual_seed(0)
model = SwinUnetR(conf).to(dev).train()
opt = train.build_optimizer(model, conf)
x, y = train.synthetic_batch(conf, batch, size, dev, 0)
for _ in range(5):
    train.train_step(model, opt, conf, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train.train_step(model, opt, conf, x, y)
    torch.cuda.synchronize()
evs = prof.events()
for e in evs:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy") and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if "visual-prompts_amd" in s or "train.py" in s][:2]
        print(e.name, e.input_shapes, f"{e.device_time_total:.1f}us", st)
print("---- memcpy/memset kernels")
for e in evs:
    if "Memcpy" in e.name or "copyBuffer" in e.name or "Memset" in e.name or "fillBuffer" in e.name:
        print(e.name, f"{e.device_time_total:.1f}us")
