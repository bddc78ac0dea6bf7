"""Are replays of a recorded step reproducible at fixed parameters (lr = 0)?  usage: replay_determinism.py <mode> <n_replays> [prompts]
single process; run two copies concurrently to share the GPU between processes."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import mivp_amd
from mivp_amd import train
from mivp_amd.swin_unetr import SwinUnetR
mode, n = sys.argv[1], int(sys.argv[2])
prompts = len(sys.argv) > 3
dev = torch.device("cuda", 0)
conf, size, batch = train.make_conf("tiny", (7, 7, 7), 0.0)
conf.training_mode, conf.use_encoder_prompting, conf.use_decoder_prompting = mode, prompts, prompts
for k in list(vars(conf)):
    if k.startswith("lr_") or k.startswith("weight_decay"):
        setattr(conf, k, 0.0)
torch.manual_seed(100)
model = SwinUnetR(conf).to(dev).train()
x, y = train.synthetic_batch(conf, batch, size, dev, 0)
opt = train.build_optimizer(model, conf, capturable=True)
step = train.graphed_train_step(model, opt, conf, x, y, warmup=1)
names = {id(p): k for k, p in model.named_parameters()}
prev, bad = None, 0
for it in range(n):
    step(); torch.cuda.synchronize()
    cur = [p.grad.clone() for p in step.params]
    if prev is not None:
        d = [(names[id(p)], float((a - b).abs().max())) for p, a, b in zip(step.params, prev, cur) if not torch.equal(a, b)]
        if d:
            bad += 1
            print(f"replay {it}: {len(d)} gradient tensors differ from the previous replay, e.g. {d[:3]}", flush=True)
    prev = cur
print(f"{mode} prompts={prompts}: {bad} of {n - 1} consecutive replay pairs differ", flush=True)
