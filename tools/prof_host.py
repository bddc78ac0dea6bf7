import sys, os, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import torch, mivp_amd
from mivp_amd import train
from mivp_amd.swin_unetr import SwinUnetR
wl = sys.argv[1]
conf, size, batch = train.make_conf(wl)
dev = torch.device("cuda")
torch.manual_seed(0)
model = SwinUnetR(conf).to(dev).train()
opt = train.build_optimizer(model, conf)
x, y = train.synthetic_batch(conf, batch, size, dev, 0)
for _ in range(20): train.train_step(model, opt, conf, x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): train.train_step(model, opt, conf, x, y)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
