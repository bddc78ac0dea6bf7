"""Are the noise-level gradients (true value zero) reproducible between .backward(), autograd.grad and a graph replay?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import mivp_amd
from mivp_amd import train
from mivp_amd.swin_unetr import SwinUnetR
dev = torch.device("cuda", 0)
conf, size, batch = train.make_conf("tiny", (7, 7, 7), 0.0)
conf.training_mode, conf.use_encoder_prompting, conf.use_decoder_prompting = "supervised_learning_all", False, False
torch.manual_seed(100)
model = SwinUnetR(conf).to(dev).train()
x, y = train.synthetic_batch(conf, batch, size, dev, 0)
names = [n for n, p in model.named_parameters() if p.requires_grad]
params = [p for n, p in model.named_parameters() if p.requires_grad]
def run_backward():
    model.zero_grad(set_to_none=True)
    loss = train.step_loss(model(x), conf, y)
    loss.backward()
    return [p.grad.detach().clone() for p in params]
def run_grad():
    loss = train.step_loss(model(x), conf, y)
    return [g.detach().clone() for g in torch.autograd.grad(loss, params)]
a = run_backward(); b = run_backward(); c = run_grad()
def cmp(u, v, tag):
    bad = [(n, float((s - t).abs().max()), float(s.abs().max())) for n, s, t in zip(names, u, v) if not torch.equal(s, t)]
    print(tag, "differing tensors:", len(bad), bad[:6])
cmp(a, b, "backward vs backward")
cmp(a, c, "backward vs autograd.grad")
# graph replay vs eager at the same parameters (single process)
opt = train.build_optimizer(model, conf, capturable=True)
step = train.graphed_train_step(model, opt, conf, x, y, warmup=1)
e = run_grad()
step()
torch.cuda.synchronize()
g = [p.grad.detach().clone() for p in step.params]
names2 = {id(p): n for n, p in model.named_parameters()}
names = [names2[id(p)] for p in step.params]
e2 = dict(zip([n for n, p in model.named_parameters() if p.requires_grad], e))
cmp([e2[n] for n in names], g, "eager (autograd.grad) vs graph replay")
