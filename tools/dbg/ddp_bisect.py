"""world 2 over gloo on one GPU, lr = 0: which gradient tensor of the recorded backward is the FIRST to differ between replays?"""
import os, sys, socket, torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

def worker(rank, world, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import mivp_amd
    from mivp_amd import train
    from mivp_amd.swin_unetr import SwinUnetR
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    train.init_distributed(dev, "gloo")
    conf, size, batch = train.make_conf("tiny", (7, 7, 7), 0.0)
    conf.training_mode, conf.use_encoder_prompting, conf.use_decoder_prompting = "supervised_learning_all", False, False
    for k in list(vars(conf)):
        if k.startswith("lr_") or k.startswith("weight_decay"):
            setattr(conf, k, 0.0)
    torch.manual_seed(100 + rank)
    model = SwinUnetR(conf).to(dev).train()
    x, y = train.synthetic_batch(conf, batch, size, dev, rank)
    opt = train.build_optimizer(model, conf, capturable=True)
    stash = {}
    order = []
    def tap(name):
        def fwd_hook(mod, inp, out):
            t = out if torch.is_tensor(out) else (out[0] if isinstance(out, (tuple, list)) else None)
            if t is not None and t.requires_grad:
                if name not in order:
                    order.append(name)
                stash["fwd:" + name] = t
                t.register_hook(lambda g, n=name: stash.__setitem__("bwd:" + n, g))
        return fwd_hook
    for n, m in model.named_modules():
        if n and n.count(".") <= 1 and not n.startswith("prompt"):
            m.register_forward_hook(tap(n))
    def forward_backward():
        out = model(x)
        stash["fwd:seg_pred"] = out["seg_pred"]
        out["seg_pred"].register_hook(lambda g: stash.__setitem__("bwd:seg_pred", g))
        loss = train.step_loss(out, conf, y)
        opt.zero_grad(set_to_none=True)
        loss.backward(train.unit_grad(loss))
        return loss.detach()
    orig_fwd = train._DiceFocalFn.forward
    orig_bwd = train._DiceFocalFn.backward
    def fwd(ctx, logits_cl, target, include_background, gamma):
        return orig_fwd(ctx, logits_cl, target, include_background, gamma)
    def bwd(ctx, g):
        stash["fwd:gin"] = g
        stash["fwd:ws"] = ctx.saved_tensors[2]
        return orig_bwd(ctx, g)
    train._DiceFocalFn.forward = staticmethod(fwd)
    train._DiceFocalFn.backward = staticmethod(bwd)
    step = train.GraphedStep(forward_backward, opt, None, None, 1)
    keys = sorted(stash)
    prev = None
    for it in range(12):
        train.barrier_sync(dev)
        l = step(); torch.cuda.synchronize()
        cur = {k: stash[k].detach().clone() for k in keys}
        cur["loss"] = l.clone()
        for i, g in enumerate(getattr(step, "local_grads", [p.grad for p in step.params])):
            cur[f"grad{i:03d}"] = g.clone()
        if prev is not None:
            d = [k for k in cur if not torch.equal(cur[k], prev[k])]
            if d:
                a, b = cur["bwd:seg_pred"].float(), prev["bwd:seg_pred"].float()
                dd = (a - b).abs()
                nz = (dd > 0)
                idx = nz.nonzero()
                print(f"[rank {rank}] replay {it}: dz differs in {int(nz.sum())} of {nz.numel()} elements, max abs diff {float(dd.max()):.3e}, max |dz| {float(a.abs().max()):.3e}; first idx {idx[:4].tolist()} last idx {idx[-2:].tolist()}; shape {tuple(a.shape)}", flush=True)
                ratio = (a / b.clamp_min(1e-30))[b.abs() > 1e-12]
                print(f"[rank {rank}] replay {it}: dz ratio min {float(ratio.min()):.6f} max {float(ratio.max()):.6f}; ws equal {torch.equal(cur['fwd:ws'], prev['fwd:ws'])} gin {cur['fwd:gin'].flatten().tolist()} prev {prev['fwd:gin'].flatten().tolist()}", flush=True)
                print(f"[rank {rank}] replay {it}: differing: fwd {[k for k in d if k.startswith('fwd')]} bwd {[k for k in d if k.startswith('bwd')]} loss {'loss' in d} grads {sum(k.startswith('grad') for k in d)}", flush=True)
        prev = cur
    print(f"[rank {rank}] done; taps: {order}", flush=True)
    train.barrier_sync(dev)
    torch.distributed.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
