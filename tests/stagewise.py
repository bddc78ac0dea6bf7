"""Stage-by-stage parity of a full forward (test helper, GPU): every fused stage of the HIP model is compared with the
rounding-aware oracle twice -- END TO END (the oracle runs its own chain) and RESTARTED (the oracle's stage is fed the HIP
path's own input of that stage).  The restarted figure isolates the stage's arithmetic from the error inherited from
upstream: two implementations that both STORE bf16 drift apart by ~1e-3 per Swin block no matter how exact their
arithmetic is (a fp32-level difference flips the final bf16 rounding of a fraction of the elements), and that drift
adds up over the 12 blocks of the network to the ~1e-2 end-to-end figure which the reference's own bf16-autocast path
shows against its fp32 path (1.1-1.4e-2)."""
import torch
import torch.nn.functional as F


def _r16(t):
    return t.to(torch.bfloat16).float()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def round_weights(sd):
    return {k: (_r16(v) if v.is_floating_point() and v.dim() >= 2 and not k.startswith("prompt_tokens") and ".pe." not in k
                and not k.startswith("input_layer.0") else v.clone()) for k, v in sd.items()}


def hip_stages(conf, sd, x, device="cuda", gout=None):
    """Run the product stage by stage (training-mode BatchNorm), recording every stage output as channels-first f32.
    With ``gout`` (channels-first [B, classes, H, W, D]) the forward keeps its graph, ``sum(logits * gout)`` is
    back-propagated and the gradient w.r.t. every stage output is recorded too: returns (rec, grads, model)."""
    import mivp_amd  # noqa: F401
    from mivp_amd import functional as Fn
    from mivp_amd.swin_unetr import SwinUnetR
    model = SwinUnetR(conf)
    model.load_state_dict(sd, strict=True)
    model.to(device).train()
    live = []

    def cf(t):
        if gout is not None:                                   # keep every stage output so that its .grad can be read
            if t.requires_grad:
                t.retain_grad()
            live.append(t)
        return t.detach().float().cpu().permute(0, 4, 1, 2, 3).contiguous()

    rec = {}
    with torch.set_grad_enabled(gout is not None):
        enc = Fn.patch_embed(model, model.input_layer[0], model.input_layer[1], x.to(device))
        rec["embed"] = cf(enc)
        feats = [enc]
        for j in range(conf.depth_unet):
            blk = model.encoder_blocks[j]
            pr = model._prompts("enc", j)
            a = blk.swin_blocks[0](enc, pr[0]); rec[f"enc{j}.b0"] = cf(a)
            b = blk.swin_blocks[1](a, pr[1]); rec[f"enc{j}.b1"] = cf(b)
            enc = blk.merge(b); rec[f"enc{j}.merge"] = cf(enc)
            feats.insert(0, enc)
        dec = Fn.conv3d_plain(model, "bottleneck", model.bottleneck, feats[0], residual=feats[0]); rec["bottleneck"] = cf(dec)
        for j in range(conf.depth_unet):
            db = model.decoder_blocks[j]
            pr = model._prompts("dec", j)
            y = Fn.upcat(dec, feats[j + 1], db.strides); rec[f"dec{j}.upcat"] = cf(y)
            y = Fn.bn_act_conv(db, db.norm_concat, db.conv_concat.conv, y, lrelu=True); rec[f"dec{j}.conv"] = cf(y)
            a = db.swin_layer.swin_blocks[0](y, pr[0]); rec[f"dec{j}.b0"] = cf(a)
            dec = db.swin_layer.swin_blocks[1](a, pr[1]); rec[f"dec{j}.b1"] = cf(dec)
        head = model.extra_heads["downstream"]
        if Fn.uphead_applicable(dec, head[0], head[1]):
            out = Fn.uphead(head[0], head[1], dec)
        else:
            out = Fn.bn_act_conv(model, head[0], head[1], Fn.upcat(dec, None, (2, 2, 2)), lrelu=False, out_f32=True,
                                 key="head_downstream")
        rec["logits"] = out.detach().float().cpu().permute(0, 4, 1, 2, 3).contiguous()
        Fn.flush_counters()
    if gout is not None:
        (out * gout.to(device).permute(0, 2, 3, 4, 1)).sum().backward()
        torch.cuda.synchronize()
        names = [k for k in rec if k != "logits"]
        grads = {}
        for i, name in enumerate(names):
            g = live[i].grad
            grads[name] = None if g is None else g.float().cpu().permute(0, 4, 1, 2, 3).contiguous()
        return rec, grads, model
    torch.cuda.synchronize()
    return rec


def stagewise_errors(conf, sd, x, emul=True, report=None):
    """{stage: (end-to-end rel-L2, restarted rel-L2)} of the HIP forward against the (rounding-aware) oracle."""
    from oracle import swin_ref as S
    rec = hip_stages(conf, sd, x)                             # gradients disabled: every block is a forward-only call, whose
    rr = S.r16 if emul else (lambda t: t)                     # attention keeps the softmax reference at zero (oracle: zero_ref)
    win, E = conf.attn_window_size, conf.pos_bias_embed_dim
    shift = tuple(w // 2 for w in win)
    res = {}

    def note(name, e2e, restarted):
        res[name] = (rel(rec[name], e2e), rel(rec[name], restarted))
        if report:
            report(f"{name:16s} {res[name][0]:11.3e} {res[name][1]:11.3e}")

    with torch.no_grad():

        e = F.conv3d(x, sd["input_layer.0.weight"], sd["input_layer.0.bias"], stride=tuple(conf.input_patch_size))
        e = rr(S.batch_norm_train(e, sd, "input_layer.1.", 1e-6, True, {}))
        note("embed", e, e)
        ofe = [e]
        for j in range(conf.depth_unet):
            heads = conf.num_heads_encoder * 2 ** j
            pr = (sd[f"prompt_tokens.enc.{2*j}"], sd[f"prompt_tokens.enc.{2*j+1}"]) if conf.use_encoder_prompting else (None, None)
            pre = f"encoder_blocks.{j}."
            prev_name = "embed" if j == 0 else f"enc{j-1}.merge"
            a = S.swin_block(e, pr[0], sd, pre + "swin_blocks.0.", win, (0, 0, 0), heads, E, emulate_bf16=emul, zero_ref=emul)
            a_r = S.swin_block(rec[prev_name], pr[0], sd, pre + "swin_blocks.0.", win, (0, 0, 0), heads, E, emulate_bf16=emul, zero_ref=emul)
            note(f"enc{j}.b0", a, a_r)
            b = S.swin_block(a, pr[1], sd, pre + "swin_blocks.1.", win, shift, heads, E, emulate_bf16=emul, zero_ref=emul)
            b_r = S.swin_block(rec[f"enc{j}.b0"], pr[1], sd, pre + "swin_blocks.1.", win, shift, heads, E, emulate_bf16=emul, zero_ref=emul)
            note(f"enc{j}.b1", b, b_r)
            e = S.patch_merge(b, sd, pre + "merge.", j < 1, emul)
            e_r = S.patch_merge(rec[f"enc{j}.b1"], sd, pre + "merge.", j < 1, emul)
            note(f"enc{j}.merge", e, e_r)
            ofe.insert(0, e)
        d = rr(F.conv3d(ofe[0], sd["bottleneck.weight"], sd["bottleneck.bias"], padding=1) + ofe[0])
        hin = rec[f"enc{conf.depth_unet-1}.merge"]
        d_r = rr(F.conv3d(hin, sd["bottleneck.weight"], sd["bottleneck.bias"], padding=1) + hin)
        note("bottleneck", d, d_r)
        prev = "bottleneck"
        for j in range(conf.depth_unet):
            pre = f"decoder_blocks.{j}."
            strides = (2, 2, 1 if j < conf.depth_unet - 1 else 2)
            pr = (sd[f"prompt_tokens.dec.{2*j}"], sd[f"prompt_tokens.dec.{2*j+1}"]) if conf.use_decoder_prompting else (None, None)

            def upc(xin, skip):
                up = F.interpolate(xin, scale_factor=tuple(float(s) for s in strides), mode="trilinear", align_corners=False)
                up = up[..., :skip.shape[2], :skip.shape[3], :skip.shape[4]]
                return rr(torch.cat([up, skip], dim=1))

            def bnconv(c):
                y = S.batch_norm_train(c, sd, pre + "norm_concat.", 1e-5, True, {})
                y = rr(F.leaky_relu(y, 0.01))
                return rr(F.conv3d(y, sd[pre + "conv_concat.conv.weight"], sd[pre + "conv_concat.conv.bias"], padding=1))

            skipname = f"enc{conf.depth_unet-2-j}.merge" if j < conf.depth_unet - 1 else "embed"
            c = upc(d, ofe[j + 1]); c_r = upc(rec[prev], rec[skipname])
            note(f"dec{j}.upcat", c, c_r)
            y = bnconv(c); y_r = bnconv(rec[f"dec{j}.upcat"])
            note(f"dec{j}.conv", y, y_r)
            a = S.swin_block(y, pr[0], sd, pre + "swin_layer.swin_blocks.0.", win, (0, 0, 0), conf.num_heads_decoder, E, emulate_bf16=emul, zero_ref=emul)
            a_r = S.swin_block(rec[f"dec{j}.conv"], pr[0], sd, pre + "swin_layer.swin_blocks.0.", win, (0, 0, 0), conf.num_heads_decoder, E, emulate_bf16=emul, zero_ref=emul)
            note(f"dec{j}.b0", a, a_r)
            d = S.swin_block(a, pr[1], sd, pre + "swin_layer.swin_blocks.1.", win, shift, conf.num_heads_decoder, E, emulate_bf16=emul, zero_ref=emul)
            d_r = S.swin_block(rec[f"dec{j}.b0"], pr[1], sd, pre + "swin_layer.swin_blocks.1.", win, shift, conf.num_heads_decoder, E, emulate_bf16=emul, zero_ref=emul)
            note(f"dec{j}.b1", d, d_r)
            prev = f"dec{j}.b1"

        def headf(lat):
            up = F.interpolate(lat, scale_factor=(2.0, 2.0, 2.0), mode="trilinear", align_corners=False)
            y = S.batch_norm_train(up, sd, "extra_heads.downstream.0.", 1e-5, True, {})
            return F.conv3d(y, sd["extra_heads.downstream.1.weight"], sd["extra_heads.downstream.1.bias"], padding=1)

        note("logits", headf(d), headf(rec[prev]))
    return res


def stagewise_block_backward(conf, sd, x, gout, report=None):
    """Backward of every Swin block at full size, RESTARTED: the rounding-aware oracle's block is run on the HIP path's own
    block input and back-propagated from the HIP path's own gradient w.r.t. the block output; compared are the gradient
    w.r.t. the block input (where the block is the only consumer of its input: every b1 block and the decoder's b0 blocks)
    and the block's own prompt-token / prompt-bias parameter gradients.  Returns {block: {what: rel-L2}}."""
    from oracle import swin_ref as S
    rec, grads, model = hip_stages(conf, sd, x, gout=gout)
    params = dict(model.named_parameters())
    win, E = conf.attn_window_size, conf.pos_bias_embed_dim
    shift = tuple(w // 2 for w in win)
    res = {}
    blocks = []
    for j in range(conf.depth_unet):
        prev = "embed" if j == 0 else f"enc{j-1}.merge"
        heads = conf.num_heads_encoder * 2 ** j
        blocks.append((f"enc{j}.b0", prev, f"encoder_blocks.{j}.swin_blocks.0.", (0, 0, 0), heads, ("enc", 2 * j), False))
        blocks.append((f"enc{j}.b1", f"enc{j}.b0", f"encoder_blocks.{j}.swin_blocks.1.", shift, heads, ("enc", 2 * j + 1), True))
    for j in range(conf.depth_unet):
        pre = f"decoder_blocks.{j}.swin_layer.swin_blocks."
        blocks.append((f"dec{j}.b0", f"dec{j}.conv", pre + "0.", (0, 0, 0), conf.num_heads_decoder, ("dec", 2 * j), True))
        blocks.append((f"dec{j}.b1", f"dec{j}.b0", pre + "1.", shift, conf.num_heads_decoder, ("dec", 2 * j + 1), True))
    for name, inp, prefix, sh, heads, (side, pi), sole in blocks:
        dy = grads[name]
        if dy is None:
            continue
        prompted = conf.use_encoder_prompting if side == "enc" else conf.use_decoder_prompting
        osd = {k: v.clone() for k, v in sd.items() if k.startswith(prefix)}
        leaves = {}
        prm = None
        if prompted:
            prm = sd[f"prompt_tokens.{side}.{pi}"].clone().requires_grad_(True)
            leaves[f"prompt_tokens.{side}.{pi}"] = prm
            for k in (prefix + "pe.weights_token", prefix + "pe.enc_token.0"):
                osd[k].requires_grad_(True)
                leaves[k] = osd[k]
        xin = rec[inp].clone().requires_grad_(True)
        y = S.swin_block(xin, prm, osd, prefix, win, sh, heads, E, emulate_bf16=True)
        y.backward(dy)
        out = {}
        if sole and grads.get(inp) is not None:
            out["dx"] = rel(grads[inp], xin.grad)
        for k, leaf in leaves.items():
            out["d" + k.split(".")[-2 if k.endswith(".0") else -1] if not k.startswith("prompt_tokens") else "dprompt"] = \
                rel(params[k].grad.float().cpu(), leaf.grad)
        res[name] = out
        if report:
            report(f"{name:10s} " + "  ".join(f"{k} {v:.2e}" for k, v in out.items()))
    return res
