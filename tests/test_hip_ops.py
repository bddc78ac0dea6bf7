"""GPU parity of the non-attention HIP kernels (through the C ABI) against plain
PyTorch fp32 on the CPU, fed the same bf16-rounded inputs.

Tolerances: bf16 operands, fp32 accumulation.  Outputs stored as bf16 carry a
2^-9 relative rounding (rel-L2 ~1.5e-3 on its own); asserts use rel-L2 <= 4e-3
for bf16 outputs and <= 2e-3 for fp32 outputs (statistics, weight gradients).
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_fixture, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def cl(t):       # [B,C,H,W,D] -> channels-last bf16 on device
    return t.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16)


def cf(t):       # channels-last device -> [B,C,H,W,D] float cpu
    return t.float().cpu().permute(0, 4, 1, 2, 3)


@pytest.mark.parametrize("cin,cout,dims,bias", [
    (16, 16, (4, 8, 16), True),            # exactly one brick
    (144, 48, (9, 13, 21), True),          # dec2 conv_concat channels; ragged bricks on every axis
    (48, 8, (6, 5, 4), False),             # volume smaller than a brick
    (32, 36, (8, 16, 32), True),           # Cout not a multiple of 16
    (48, 144, (5, 9, 17), True),           # three output-channel groups (dgrad of the dec2 conv)
    (96, 96, (8, 8, 16), False),           # two groups
    (16, 48, (9, 13, 21), False),          # one input chunk: the data gradient of a few-class head padded to 16 channels
])
def test_conv3d_halo_brick_kernel(cin, cout, dims, bias):
    """The halo-brick form (csrc/conv3d_halo.hip) against F.conv3d and against the im2col kernel on the same input."""
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = r16(torch.randn(2, cin, *dims, generator=g))
    w = r16(torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1 if bias else None
    res = r16(torch.randn(2, cout, *dims, generator=g)) if cin == cout else None      # bottleneck-style residual
    want = F.conv3d(x, w, b, padding=1) + (res if res is not None else 0)
    wp = ops.pack_conv_weight(w.to(DEV))
    bd = None if b is None else b.to(DEV)
    rd = None if res is None else cl(res)
    y = ops.conv3d(cl(x), wp, bd, cout, residual=rd, force_halo=8)
    y4 = ops.conv3d(cl(x), wp, bd, cout, residual=rd, force_halo=4)      # 4x4x16 bricks, 4 waves
    y6 = ops.conv3d(cl(x), wp, bd, cout, residual=rd, force_halo=6)      # 6x6x16 bricks, 12 waves x 3 tiles
    y66 = ops.conv3d(cl(x), wp, bd, cout, residual=rd, force_halo=66)    # 6x6x8 bricks of 2 x 8 voxel tiles, 6 waves x 3 tiles
    y36 = ops.conv3d(cl(x), wp, bd, cout, residual=rd, force_halo=36)    # 3x6x8 bricks of 2 x 8 voxel tiles, 3 waves x 3 tiles
    saved = ops.halo_brick
    ops.halo_brick = lambda *a: 0
    try:
        y_ref = ops.conv3d(cl(x), wp, bd, cout, residual=rd)    # the im2col kernel
    finally:
        ops.halo_brick = saved
    torch.cuda.synchronize()
    assert getattr(wp, "_mivp_halo", None) is not None     # the halo path really ran
    for got in (y, y4, y6, y66, y36):
        assert rel_l2(cf(got), want) < 4e-3 and rel_l2(cf(got), cf(y_ref)) < 2e-3
    # every brick geometry adds the same products in the same order (chunk, k-step): identical bits
    for got in (y4, y6, y66, y36):
        assert torch.equal(got, y)


@pytest.mark.parametrize("cin,cout,dims,lrelu,bw", [(144, 48, (9, 13, 21), True, 8), (32, 96, (6, 6, 24), False, 4),
                                                    (144, 48, (13, 9, 21), True, 6), (64, 48, (7, 12, 21), True, 66),
                                                    (32, 96, (6, 6, 24), True, 36)])
def test_conv3d_halo_fused_prologue(cin, cout, dims, lrelu, bw):
    """BatchNorm affine (+ LeakyReLU) applied while the halo is staged: zero padding stays zero AFTER the activation."""
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cin + bw)
    x = r16(torch.randn(2, cin, *dims, generator=g))
    w = r16(torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    scale, shift = 1 + 0.2 * torch.randn(cin, generator=g), 0.2 * torch.randn(cin, generator=g)
    xin = x * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    if lrelu:
        xin = F.leaky_relu(xin, 0.01)
    want = F.conv3d(r16(xin), w, b, padding=1)
    wp = ops.pack_conv_weight(w.to(DEV))
    y = ops.conv3d(cl(x), wp, b.to(DEV), cout, scale.to(DEV), shift.to(DEV), lrelu, force_halo=bw)
    torch.cuda.synchronize()
    assert getattr(wp, "_mivp_halo", None) is not None
    assert rel_l2(cf(y), want) < 4e-3


@pytest.mark.parametrize("cin,cout,dims,affine,lrelu,res,f32", [
    (16, 16, (5, 6, 7), False, False, False, False),
    (24, 48, (6, 6, 8), True, True, False, False),      # Cin % 32 != 0: k-steps straddle taps
    (64, 64, (4, 4, 6), False, False, True, False),     # bottleneck-style residual
    (48, 2, (8, 8, 8), True, False, False, True),       # segmentation head: BN prologue, fp32 logits
    (144, 48, (6, 5, 4), True, True, False, False),     # dec2 conv_concat channel counts
    (8, 96, (3, 3, 3), False, False, False, False),
])
def test_conv3d_fwd(cin, cout, dims, affine, lrelu, res, f32):
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = r16(torch.randn(2, cin, *dims, generator=g))
    w = r16(torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    scale = 1 + 0.2 * torch.randn(cin, generator=g) if affine else None
    shift = 0.2 * torch.randn(cin, generator=g) if affine else None
    resid = r16(torch.randn(2, cout, *dims, generator=g)) if res else None
    xin = x
    if affine:
        xin = x * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
        if lrelu:
            xin = F.leaky_relu(xin, 0.01)
        xin = r16(xin)
    want = F.conv3d(xin, w, b, padding=1)
    if res:
        want = want + resid
    wp = ops.pack_conv_weight(w.to(DEV))
    y = ops.conv3d(cl(x), wp, b.to(DEV), cout, None if scale is None else scale.to(DEV),
                   None if shift is None else shift.to(DEV), lrelu, None if resid is None else cl(resid), f32)
    torch.cuda.synchronize()
    assert rel_l2(cf(y), want) < (2e-3 if f32 else 4e-3)


def test_conv3d_dgrad_and_head_wgrad():
    from mivp_amd import ops
    g = torch.Generator().manual_seed(3)
    cin, cout, dims = 48, 2, (6, 8, 8)
    x = r16(torch.randn(2, cin, *dims, generator=g)).requires_grad_(True)
    w = r16(torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    scale = 1 + 0.2 * torch.randn(cin, generator=g)
    shift = 0.2 * torch.randn(cin, generator=g)
    xin = r16((x * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)).detach()).requires_grad_(True)
    y = F.conv3d(xin, w, b, padding=1)
    dy = r16(torch.randn(y.shape, generator=g))
    y.backward(dy)
    # data gradient = same kernel with flipped / transposed weights; dy carries 8 channels (6 zero)
    wd, cpad = ops.pack_conv_weight_dgrad(w.detach().to(DEV))
    dyp = torch.zeros(2, *dims, cpad, dtype=torch.bfloat16, device=DEV)
    dyp[..., :cout] = cl(dy)
    dx = ops.conv3d(dyp, wd, None, cin)
    torch.cuda.synchronize()
    assert rel_l2(cf(dx), xin.grad) < 4e-3
    dw, db = ops.conv3d_wgrad_small(cl(x.detach()), scale.to(DEV), shift.to(DEV), False, dyp, cout)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu(), w.grad) < 2e-3
    assert rel_l2(db.cpu(), b.grad) < 2e-3


@pytest.mark.parametrize("C,dims", [(48, (6, 6, 6)), (144, (4, 5, 6)), (8, (3, 3, 2))])
def test_batchnorm_train_fwd_bwd(C, dims):
    from mivp_amd import ops
    g = torch.Generator().manual_seed(C)
    x = r16(torch.randn(2, C, *dims, generator=g) * 1.5 + 0.3).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.leaky_relu(F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5), 0.01)
    dy = r16(torch.randn(y.shape, generator=g))
    y.backward(dy)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    xd = cl(x.detach())
    scale, shift, mr = ops.bn_batch_stats(xd, w.detach().to(DEV), b.detach().to(DEV), 1e-5, rmd, rvd)
    yd = ops.affine_act(xd, scale, shift, True)
    dx, dg, dbeta = ops.bn_backward(xd, cl(dy), scale, shift, mr, True)
    torch.cuda.synchronize()
    assert rel_l2(rmd.cpu(), rm) < 1e-4 and rel_l2(rvd.cpu(), rv) < 1e-4
    assert rel_l2(cf(yd), y) < 4e-3
    assert rel_l2(cf(dx), x.grad) < 6e-3
    assert rel_l2(dg.cpu(), w.grad) < 3e-3
    assert rel_l2(dbeta.cpu(), b.grad) < 3e-3


@pytest.mark.parametrize("cin,dims", [(1, (16, 16, 16)), (4, (8, 12, 8))])
def test_patch_embed(cin, dims):
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cin)
    C = 48
    x = torch.rand(2, cin, *dims, generator=g)
    w = torch.randn(C, cin, 2, 2, 2, generator=g) * 0.3
    b = torch.randn(C, generator=g) * 0.1
    bw, bb = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    rm, rv = torch.zeros(C), torch.ones(C)
    want = F.batch_norm(F.conv3d(x, w, b, stride=2), rm, rv, bw, bb, True, 0.1, 1e-6)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y = ops.patch_embed(x.to(DEV), w.to(DEV), b.to(DEV), bw.to(DEV), bb.to(DEV), 1e-6, rmd, rvd)
    torch.cuda.synchronize()
    assert rel_l2(cf(y), want) < 4e-3
    assert rel_l2(rmd.cpu(), rm) < 1e-4 and rel_l2(rvd.cpu(), rv) < 1e-4


@pytest.mark.parametrize("tag", ["even_T", "odd_T", "even_F", "odd_F"])
def test_patch_merge_golden(tag):
    from mivp_amd import ops
    from oracle import swin_ref as S
    fx = load_fixture(f"merge_{tag}")
    sd = dict(fx["sd"])
    sd["reduction.weight"] = r16(sd["reduction.weight"])
    x = r16(fx["in"]["x"])
    want = S.patch_merge(x, sd, "", fx.meta["merge_last_dim"])
    y = ops.patch_merge(cl(x), sd["norm.weight"].to(DEV), sd["norm.bias"].to(DEV),
                        sd["reduction.weight"].to(DEV, torch.bfloat16), fx.meta["merge_last_dim"])
    torch.cuda.synchronize()
    assert cf(y).shape == want.shape
    assert rel_l2(cf(y), want) < 5e-3


@pytest.mark.parametrize("C,dims,last", [(48, (8, 8, 8), True), (96, (6, 6, 8), False), (192, (4, 6, 5), False)])
def test_patch_merge_real_channels(C, dims, last):
    from mivp_amd import ops
    from oracle import swin_ref as S
    g = torch.Generator().manual_seed(C)
    k = 8 if last else 4
    sd = {"norm.weight": 1 + 0.2 * torch.randn(k * C, generator=g), "norm.bias": 0.1 * torch.randn(k * C, generator=g),
          "reduction.weight": r16(torch.randn(2 * C, k * C, generator=g) / (k * C) ** 0.5)}
    x = r16(torch.randn(2, C, *dims, generator=g))
    want = S.patch_merge(x, sd, "", last)
    y = ops.patch_merge(cl(x), sd["norm.weight"].to(DEV), sd["norm.bias"].to(DEV),
                        sd["reduction.weight"].to(DEV, torch.bfloat16), last)
    torch.cuda.synchronize()
    assert rel_l2(cf(y), want) < 5e-3


@pytest.mark.parametrize("idims,sdims,scale,cx,cs", [
    ((3, 3, 4), (6, 6, 4), (2, 2, 1), 16, 8),
    ((3, 4, 2), (5, 7, 4), (2, 2, 2), 16, 8),          # skip smaller than 2x: crop
    ((6, 6, 6), None, (2, 2, 2), 48, 0),               # output layer: plain upsample
])
def test_upcat_fwd_bwd(idims, sdims, scale, cx, cs):
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cx + cs)
    x = r16(torch.randn(2, cx, *idims, generator=g)).requires_grad_(True)
    up = F.interpolate(x, scale_factor=tuple(float(s) for s in scale), mode="trilinear", align_corners=False)
    skip = None
    if cs:
        skip = r16(torch.randn(2, cs, *sdims, generator=g)).requires_grad_(True)
        want = torch.cat([up[..., :sdims[0], :sdims[1], :sdims[2]], skip], 1)
    else:
        want = up
    dy = r16(torch.randn(want.shape, generator=g))
    want.backward(dy)
    y = ops.upcat(cl(x.detach()), None if skip is None else cl(skip.detach()), scale)
    dx, dskip = ops.upcat_backward(cl(dy), idims, scale, cx, cs)
    torch.cuda.synchronize()
    assert rel_l2(cf(y), want) < 4e-3
    assert rel_l2(cf(dx), x.grad) < 4e-3
    if cs:
        assert rel_l2(cf(dskip), skip.grad) < 1e-6


@pytest.mark.parametrize("idims,sdims,scale,cx,cs", [
    ((3, 3, 4), (6, 6, 4), (2, 2, 1), 16, 8),
    ((3, 4, 2), (5, 7, 4), (2, 2, 2), 16, 8),          # crop
    ((12, 12, 12), (24, 24, 24), (2, 2, 2), 96, 48),   # decoder stage shape (channels), several rows per workgroup
    ((3, 3, 3), (6, 6, 6), (2, 2, 2), 384, 192),       # widest decoder stage
    ((2, 3, 24), (4, 6, 24), (2, 2, 1), 384, 192),     # its 96^3 shape: d not halved, 1152 pieces per source row
    ((2, 2, 12), (4, 4, 24), (2, 2, 2), 384, 192),     # 576 pieces
])
def test_upcat_statistics_and_affine_without_the_concat_tensor(idims, sdims, scale, cx, cs):
    """mivp_upcat_stats / mivp_upcat_affine_fwd (SwinUpBlock's frozen-decoder path) against the materialised pipeline
    upcat -> bn_stats -> affine_act: the activations bit for bit, the statistics to f32 summation order."""
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cx + cs + idims[0])
    x = cl(r16(torch.randn(2, cx, *idims, generator=g) + 0.3))
    skip = cl(r16(torch.randn(2, cs, *sdims, generator=g) - 0.1))
    ct = cx + cs
    gamma = (1 + 0.2 * torch.randn(ct, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(ct, generator=g)).to(DEV)
    cat = ops.upcat(x, skip, scale)
    rm0, rv0 = torch.zeros(ct, device=DEV), torch.ones(ct, device=DEV)
    sc0, sh0, mr0 = ops.bn_batch_stats(cat, gamma, beta, 1e-5, rm0, rv0, 0.1)
    part, nblk, n_vox = ops.upcat_stats(x, skip, scale)
    assert n_vox == cat.numel() // ct
    rm1, rv1 = torch.zeros(ct, device=DEV), torch.ones(ct, device=DEV)
    sc1, sh1, mr1 = ops.bn_finalize(part, nblk, ct, n_vox, gamma, beta, 1e-5, rm1, rv1, 0.1)
    torch.cuda.synchronize()
    assert rel_l2(sc1, sc0) < 1e-6 and rel_l2(sh1, sh0) < 1e-5 and rel_l2(mr1, mr0) < 1e-6
    assert rel_l2(rm1, rm0) < 1e-5 and rel_l2(rv1, rv0) < 1e-6
    for lrelu in (True, False):
        want = ops.affine_act(cat, sc0, sh0, lrelu)
        got = ops.upcat_affine(x, skip, scale, sc0, sh0, lrelu)
        torch.cuda.synchronize()
        assert torch.equal(got, want)


@pytest.mark.parametrize("cin,cout,dims,training", [(48, 2, (6, 8, 8), True), (48, 5, (4, 5, 7), False), (8, 2, (5, 4, 33), True)])
def test_head_backward_from_one_mfma_pass(cin, cout, dims, training):
    """conv dW/db and BatchNorm dgamma/dbeta of the (BN -> conv) head all come from (G, S) of
    mivp_conv3d_wgrad_rows; compare with autograd through batch_norm + conv3d."""
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = r16(torch.randn(2, cin, *dims, generator=g) + 0.2)
    gamma = (1 + 0.2 * torch.randn(cin, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(cin, generator=g)).requires_grad_(True)
    w = r16(torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    rm, rv = 0.1 * torch.randn(cin, generator=g), 1 + 0.2 * torch.rand(cin, generator=g)
    y = F.conv3d(F.batch_norm(x, rm.clone(), rv.clone(), gamma, beta, training, 0.1, 1e-5), w, b, padding=1)
    dy = r16(torch.randn(y.shape, generator=g))
    y.backward(dy)
    xd = cl(x)
    if training:
        scale, shift, mr = ops.bn_batch_stats(xd, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5)
    else:
        scale, shift, mr = ops.bn_eval_affine(gamma.detach().to(DEV), beta.detach().to(DEV), rm.to(DEV), rv.to(DEV), 1e-5)
    dyp = torch.zeros(2, *dims, 8, dtype=torch.bfloat16, device=DEV)
    dyp[..., :cout] = cl(dy)
    G, S = ops.conv3d_wgrad_rows(xd, dyp, cout)
    dW, db, dgamma, dbeta = ops.head_grads_from_gs(G, S, w.detach().to(DEV), scale, shift, mr)
    torch.cuda.synchronize()
    assert rel_l2(dW.cpu(), w.grad) < 3e-3
    assert rel_l2(db.cpu(), b.grad) < 1e-3
    if training:
        # training-mode BN: gradients also flow through the batch statistics; dgamma/dbeta do not depend on that
        pass
    assert rel_l2(dgamma.cpu(), gamma.grad) < 5e-3
    assert rel_l2(dbeta.cpu(), beta.grad) < 5e-3


@pytest.mark.parametrize("C,inc_bg", [(2, True), (2, False), (5, True)])
def test_dice_focal_loss_value_and_gradient(C, inc_bg):
    """fused HIP loss vs the oracle's restatement of MONAI DiceFocalLoss (parity unpinned: MONAI absent)."""
    from mivp_amd import train
    from oracle.loss_ref import dice_focal_loss as oracle_loss
    g = torch.Generator().manual_seed(C)
    logits = (2.0 * torch.randn(2, C, 6, 7, 8, generator=g)).requires_grad_(True)
    y = torch.randint(0, C, (2, 1, 6, 7, 8), generator=g).float()
    want = oracle_loss(logits, y, inc_bg, 4.0)
    want.backward()
    base = logits.detach().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    got = train.dice_focal_loss(base.permute(0, 4, 1, 2, 3), y.to(DEV), inc_bg, 4.0)
    (got * 2.5).backward()                                   # the incoming gradient is applied inside the gradient pass
    torch.cuda.synchronize()
    assert abs(float(got) - float(want)) < 2e-5 * max(1.0, abs(float(want)))
    assert rel_l2(base.grad.cpu().permute(0, 4, 1, 2, 3), 2.5 * logits.grad) < 1e-4


@pytest.mark.parametrize("C,inc_bg,dims", [(2, True, (6, 7, 8)), (2, False, (8, 8, 8)), (5, True, (6, 7, 8)), (5, False, (4, 6, 5))])
def test_dice_loss_value_and_gradient(C, inc_bg, dims):
    """losses.dice_loss (the Dice-focal kernels without the focal term, csrc/loss.hip gamma < 0) vs the oracle's restatement of
    MONAI DiceLoss(include_background, to_onehot_y, softmax) -- the supervised modes' segmentation term
    (students_teacher.py:96-100,190-197).  Parity unpinned (MONAI absent).  Both the channels-last storage the HIP model
    returns and a plain contiguous channels-first tensor are fed (the latter is copied to channels-last, never an eager path)."""
    from mivp_amd.losses import dice_loss
    from oracle.loss_ref import dice_loss as oracle_loss
    g = torch.Generator().manual_seed(10 * C + dims[0])
    logits = (2.0 * torch.randn(2, C, *dims, generator=g)).requires_grad_(True)
    y = torch.randint(0, C, (2, 1, *dims), generator=g).float()
    want = oracle_loss(logits, y, inc_bg)
    want.backward()
    base = logits.detach().permute(0, 2, 3, 4, 1).contiguous().to(DEV).requires_grad_(True)
    got = dice_loss(base.permute(0, 4, 1, 2, 3), y.to(DEV), inc_bg)
    (got * 0.7).backward()
    plain = logits.detach().to(DEV).requires_grad_(True)
    got2 = dice_loss(plain, y.to(DEV), inc_bg)
    got2.backward()
    torch.cuda.synchronize()
    assert abs(float(got) - float(want)) < 2e-5 * max(1.0, abs(float(want)))
    assert float(got2) == float(got)
    assert rel_l2(base.grad.cpu().permute(0, 4, 1, 2, 3), 0.7 * logits.grad) < 1e-4
    assert rel_l2(plain.grad.cpu(), logits.grad) < 1e-4
    with pytest.raises(RuntimeError):
        dice_loss(logits.detach(), y, inc_bg)                  # CPU tensors: the product has no CPU path


@pytest.mark.parametrize("cin,cout,dims", [(48, 2, (8, 8, 16)), (48, 2, (9, 6, 21)), (8, 2, (4, 5, 7)), (32, 1, (6, 6, 6))])
def test_head_conv_fused(cin, cout, dims):
    """BatchNorm-affine + 3x3x3 conv to <= 2 channels through the per-voxel-GEMM + gather kernel."""
    from mivp_amd import ops
    g = torch.Generator().manual_seed(cin + cout + dims[2])
    x = r16(torch.randn(2, cin, *dims, generator=g))
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    b = 0.1 * torch.randn(cout, generator=g)
    scale, shift = 1 + 0.2 * torch.randn(cin, generator=g), 0.2 * torch.randn(cin, generator=g)
    want = F.conv3d(x * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1), w, b, padding=1)
    y = ops.head_conv(cl(x), w.to(DEV), b.to(DEV), scale.to(DEV), shift.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(cf(y), want) < 3e-3       # bf16 weights (scale folded in) and fp16 tap partials


@pytest.mark.parametrize("C,cout,dims,training", [(48, 2, (6, 5, 7), True), (48, 2, (4, 4, 4), False), (8, 1, (3, 9, 2), True),
                                                   (16, 2, (1, 2, 3), True)])
def test_uphead_low_resolution_head(C, cout, dims, training):
    """upsample x2 (trilinear, align_corners=False) -> BatchNorm3d -> Conv3d 3^3 evaluated from the low-resolution
    tensor (csrc/uphead.hip): logits and the four parameter gradients against torch on the same bf16 input.
    Tolerances: logits 4e-3 (fp16 tap planes), gradients 6e-3 (bf16 adjoint operand, fp32 sums)."""
    import torch.nn as nn
    import mivp_amd
    from mivp_amd import functional as Fn
    g = torch.Generator().manual_seed(C + cout + dims[0])
    x = r16(torch.randn(2, C, *dims, generator=g) + 0.3).requires_grad_(True)
    bn, conv = nn.BatchNorm3d(C), nn.Conv3d(C, cout, 3, 1, 1)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        bn.bias.copy_(0.1 * torch.randn(C, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(C, generator=g))
        bn.running_var.copy_(1 + 0.2 * torch.rand(C, generator=g))
    bn.train(training)
    import copy
    bn2, conv2 = copy.deepcopy(bn).to(DEV), copy.deepcopy(conv).to(DEV)
    up = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=False)
    y = conv(bn(up))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = cl(x.detach()).requires_grad_(True)
    assert Fn.uphead_applicable(xd, bn2, conv2)
    out = Fn.uphead(bn2, conv2, xd)
    out.backward(dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(cf(xd.grad), x.grad) < 1e-2
    assert out.shape == (2, 2 * dims[0], 2 * dims[1], 2 * dims[2], cout) and out.dtype == torch.float32
    assert rel_l2(out.detach().cpu().permute(0, 4, 1, 2, 3), y.detach()) < 4e-3
    assert rel_l2(conv2.weight.grad.cpu(), conv.weight.grad) < 6e-3
    assert rel_l2(conv2.bias.grad.cpu(), conv.bias.grad) < 6e-3
    assert rel_l2(bn2.weight.grad.cpu(), bn.weight.grad) < 6e-3
    assert rel_l2(bn2.bias.grad.cpu(), bn.bias.grad) < 6e-3
    if training:
        assert rel_l2(bn2.running_mean.cpu(), bn.running_mean) < 2e-3
        assert rel_l2(bn2.running_var.cpu(), bn.running_var) < 2e-3
        Fn.flush_counters()          # the step counters are batched per model forward (functional.bump_counter)
        assert int(bn2.num_batches_tracked) == int(bn.num_batches_tracked)


def test_reconstruction_head_matches_torch_modules():
    """The phase-1 reconstruction head (swin_unetr.py:185-212): [Conv3d 3^3 -> InstanceNorm3d -> LeakyReLU ->
    Upsample(trilinear, align_corners=True)] x 4 -> Conv3d 1^3 are stock torch modules, so torch itself is the oracle:
    forward, input gradient and every parameter gradient on a small deepest-feature tensor.
    Tolerances: output 1e-2 (four bf16 stages incl. instance norms on few voxels); gradients against a measured
    bf16-storage yardstick (see below)."""
    import copy
    import torch.nn as nn
    import mivp_amd
    from mivp_amd import functional as Fn
    g = torch.Generator().manual_seed(3)
    chs = [64, 32, 16, 16, 16]
    layers = []
    for i in range(4):
        layers += [nn.Conv3d(chs[i], chs[i + 1], 3, 1, 1), nn.InstanceNorm3d(chs[i + 1]), nn.LeakyReLU(),
                   nn.Upsample(scale_factor=(2, 2, 1 if i < 2 else 2), mode="trilinear", align_corners=True)]
    layers.append(nn.Conv3d(chs[-1], 2, 1, 1))
    head = nn.Sequential(*layers)
    with torch.no_grad():
        for m in head:
            if isinstance(m, nn.Conv3d):
                m.weight.copy_(r16(m.weight))
    x = r16(torch.randn(2, 64, 4, 4, 6, generator=g)).requires_grad_(True)
    y = head(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)

    class Owner:
        pass
    owner = Owner()
    owner._wcache = Fn.WeightCache()
    head2 = copy.deepcopy(head).to(DEV)
    for p in head2.parameters():
        p.grad = None
    xd = cl(x.detach()).requires_grad_(True)
    out = Fn.reconstruction_head(owner, head2, xd)
    out.backward(dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV))
    torch.cuda.synchronize()
    assert tuple(out.shape) == (2, 64, 64, 24, 2) == tuple(y.permute(0, 2, 3, 4, 1).shape) and out.dtype == torch.float32
    assert rel_l2(out.detach().cpu().permute(0, 4, 1, 2, 3), y.detach()) < 1e-2
    errs = {"dx": rel_l2(cf(xd.grad), x.grad)}
    for (n1, p1), (n2, p2) in zip(head.named_parameters(), head2.named_parameters()):
        if n1.endswith(".bias") and n1 != "16.bias":         # conv bias in front of an instance norm: true gradient zero
            assert float(p2.grad.abs().max()) < 2e-2 * float(dict(head2.named_parameters())[n1.replace("bias", "weight")].grad.abs().max()), n1
            continue
        errs[n1] = rel_l2(p2.grad.cpu(), p1.grad)
    # Yardstick: the same torch chain in fp32 with every stage boundary (conv output, activation, upsample) rounded to
    # bf16 in both directions - the storage precision of the HIP path.  The instance norms' backward projects out the
    # mean and the xhat component, so the surviving gradients are differences of much larger terms and a 2^-9 rounding
    # shows up as 4-8 % in the deepest gradients of this chain (measured, seed 3); the HIP path has to stay within
    # 1.25x that yardstick, and within 1e-2 where the yardstick itself is small.
    class _Round(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return r16(t)

        @staticmethod
        def backward(ctx, gr):
            return r16(gr)

    class _RoundM(nn.Module):
        def forward(self, t):
            return _Round.apply(t)

    mods = []
    for m in copy.deepcopy(head):
        mods.append(m)
        if isinstance(m, (nn.Conv3d, nn.LeakyReLU, nn.Upsample)):
            mods.append(_RoundM())
    head_r = nn.Sequential(*mods)
    for p in head_r.parameters():
        p.grad = None
    xr = x.detach().clone().requires_grad_(True)
    head_r(xr).backward(dy)
    yard = {"dx": rel_l2(xr.grad, x.grad)}
    wn = [n for n, _ in head.named_parameters()]
    for n, pr in zip(wn, head_r.parameters()):
        if n in errs:
            yard[n] = rel_l2(pr.grad, dict(head.named_parameters())[n].grad)
    for k, e in errs.items():
        assert e < max(1.25 * yard[k], 1e-2), (k, e, yard[k], errs, yard)


def test_pooled_linear_heads():
    import copy
    import torch.nn as nn
    import mivp_amd
    from mivp_amd import functional as Fn
    g = torch.Generator().manual_seed(4)
    x = r16(torch.randn(2, 64, 3, 3, 6, generator=g)).requires_grad_(True)
    lin = nn.Linear(64, 4)
    y = lin(nn.AdaptiveAvgPool3d((1, 1, 1))(x).flatten(1))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    lin2 = copy.deepcopy(lin).to(DEV)
    for p in lin2.parameters():
        p.grad = None
    xd = cl(x.detach()).requires_grad_(True)
    out = Fn.pooled_linear(lin2, xd)
    out.backward(dy.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(out.detach().cpu(), y.detach()) < 1e-5
    assert rel_l2(cf(xd.grad), x.grad) < 4e-3
    assert rel_l2(lin2.weight.grad.cpu(), lin.weight.grad) < 1e-5


@pytest.mark.parametrize("scale,dims", [((2, 2, 1), (3, 4, 5)), ((2, 2, 2), (2, 3, 4)), ((2, 2, 2), (1, 1, 3))])
def test_upsample_align_corners(scale, dims):
    from mivp_amd import ops
    g = torch.Generator().manual_seed(dims[0])
    x = r16(torch.randn(2, 16, *dims, generator=g)).requires_grad_(True)
    want = F.interpolate(x, scale_factor=tuple(float(s) for s in scale), mode="trilinear", align_corners=True)
    dy = r16(torch.randn(want.shape, generator=g))
    want.backward(dy)
    y = ops.upcat(cl(x.detach()), None, scale, align_corners=True)
    dx, _ = ops.upcat_backward(cl(dy), dims, scale, 16, 0, align_corners=True)
    torch.cuda.synchronize()
    assert rel_l2(cf(y), want.detach()) < 4e-3
    assert rel_l2(cf(dx), x.grad) < 4e-3


@pytest.mark.parametrize("stride,cin,cout,dims", [((2, 2, 2), 96, 48, (6, 5, 7)), ((2, 2, 1), 384, 192, (3, 3, 6)), ((2, 2, 2), 16, 8, (4, 4, 4))])
def test_conv_transpose_k2s2(stride, cin, cout, dims):
    """ConvTranspose3d with kernel == stride, no bias (MONAI UnetrUpBlock's up-sampling step, swin_unetr.py:338-348): forward,
    data gradient and weight gradient against torch's own nn.functional.conv_transpose3d in fp32 on the bf16-rounded operands
    (a stock layer: torch is the oracle; parity unpinned at the MONAI boundary)."""
    import mivp_amd
    from mivp_amd import functional as Fn
    g = torch.Generator().manual_seed(3)
    conv = torch.nn.ConvTranspose3d(cin, cout, kernel_size=stride, stride=stride, bias=False)
    with torch.no_grad():
        conv.weight.copy_(r16(conv.weight))
    x = r16(torch.randn(2, cin, *dims, generator=g))
    xr = x.clone().requires_grad_(True)
    want = torch.nn.functional.conv_transpose3d(xr, conv.weight, None, stride=stride)
    gout = r16(torch.randn(want.shape, generator=g))
    want.backward(gout)
    dw_want = conv.weight.grad.clone()
    conv.weight.grad = None
    conv = conv.to(DEV)

    class Owner:
        _wcache = Fn.WeightCache()

    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16).requires_grad_(True)
    y = Fn.conv_transpose(Owner, "t", conv, xc)
    y.backward(gout.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16))
    torch.cuda.synchronize()
    assert rel_l2(y.detach().float().cpu().permute(0, 4, 1, 2, 3), want.detach()) < 3e-3       # bf16 output rounding
    assert rel_l2(xc.grad.float().cpu().permute(0, 4, 1, 2, 3), xr.grad) < 3e-3
    assert rel_l2(conv.weight.grad.cpu(), dw_want) < 2e-3


@pytest.mark.parametrize("rows,cols,k_steps,paired", [(48, 48, 0, False), (48, 48, 0, True), (144, 48, 0, False),
                                                      (48, 144, 5, False), (8, 24, 2, False), (192, 576, 0, False)])
def test_pack_weight_frags(rows, cols, k_steps, paired):
    """mivp_pack_weight_frags against the layout formula of include/mivp.h ("Weight fragment images")."""
    import mivp_amd
    from mivp_amd import swin_ops
    gen = torch.Generator().manual_seed(5)
    w = torch.randn(rows, cols, generator=gen).to(torch.bfloat16)
    img = swin_ops.pack_weight_frags(w.to(DEV), paired=paired, k_steps=k_steps).cpu().float()
    nt, ks = (rows + 15) // 16, max((cols + 31) // 32, k_steps)
    assert img.shape == (nt, ks, 64, 8)
    wp = torch.zeros(nt * 16, ks * 32)
    wp[:rows, :cols] = w.float()
    want = torch.empty(nt, ks, 64, 8)
    for g in range(4):
        for e in range(8):
            col = (16 * (e >> 2) + 4 * g + (e & 3)) if paired else 8 * g + e
            # lane = 16 g + r
            want[:, :, 16 * g:16 * g + 16, e] = wp.view(nt, 16, ks, 32)[:, :, :, col].permute(0, 2, 1)
    assert torch.equal(img, want)
