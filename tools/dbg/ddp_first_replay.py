"""Two ranks on one GPU (gloo): is the FIRST replay of a recorded step different from later replays at the same parameters?"""
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
            setattr(conf, k, 0.0)                                  # parameters never move: every replay sees the same state
    torch.manual_seed(100 + rank)
    model = SwinUnetR(conf).to(dev).train()
    x, y = train.synthetic_batch(conf, batch, size, dev, rank)
    opt = train.build_optimizer(model, conf, capturable=True)
    step = train.graphed_train_step(model, opt, conf, x, y, warmup=1)
    loss = train.step_loss(model(x), conf, y)
    ge = [g.clone() for g in torch.autograd.grad(loss, step.params)]
    reps = []
    for it in range(3):
        train.barrier_sync(dev)
        step(); torch.cuda.synchronize()
        reps.append([g.clone() for g in step.local_grads])
    def nd(a, b):
        return sum(0 if torch.equal(u, v) else 1 for u, v in zip(a, b))
    print(f"[rank {rank}] eager vs replay1 {nd(ge, reps[0])}, replay1 vs replay2 {nd(reps[0], reps[1])}, replay2 vs replay3 {nd(reps[1], reps[2])}, eager vs replay2 {nd(ge, reps[1])}", flush=True)
    train.barrier_sync(dev)
    torch.distributed.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
