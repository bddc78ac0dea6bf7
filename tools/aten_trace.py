"""Which Python lines issue torch-native (aten) ops during a training step: a TorchDispatchMode with stack capture."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import mivp_amd
from mivp_amd import train
from mivp_amd.swin_unetr import SwinUnetR
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
agg = collections.Counter()
SKIP = ("aten.empty", "aten.view", "aten.detach", "aten.as_strided", "aten.permute", "aten.reshape", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.t.", "aten.transpose", "aten.alias", "aten._unsafe_view", "aten.unbind",
        "aten.split", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.lift_fresh", "aten.narrow", "aten.empty_like")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if ("mivp_amd" in fr.filename or "visual-prompts_amd" in fr.filename or fr.filename.endswith("train.py")) and "aten_trace" not in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            agg[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    train.train_step(model, opt, conf, x, y)
torch.cuda.synchronize()
for (n, f), c in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0]))[:80]:
    print(c, n, f)
