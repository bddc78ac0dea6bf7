"""GPU parity of the weight-gradient kernels (csrc/wgrad.hip) against fp32 torch on the same bf16 operands.

Tolerance: operands are identical bf16 values on both sides and accumulation is fp32 on both, so only the summation
order differs: rel-L2 <= 1e-5 (written below)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    import mivp_amd
    from mivp_amd import ops
    return ops


@pytest.mark.parametrize("T,M,N", [(128, 64, 64), (1000, 48, 48), (4097, 8, 24), (70000, 96, 192), (300, 384, 130 * 4)])
def test_gemm_tn_rows(T, M, N):
    ops = _ops()
    g = torch.Generator().manual_seed(T + M)
    lda, ldb = M + 8, N
    a = torch.randn(T, lda, generator=g).bfloat16().to(DEV)
    b = torch.randn(T, ldb, generator=g).bfloat16().to(DEV)
    out = ops.gemm_tn(a, ops.operand_rows(lda), b, ops.operand_rows(ldb), T, M, N, alpha=0.5)
    want = 0.5 * a[:, :M].float().T @ b.float()
    assert rel_l2(out.cpu(), want.cpu()) < 1e-5
    # accumulate on top
    out2 = ops.gemm_tn(a, ops.operand_rows(lda), b, ops.operand_rows(ldb), T, M, N, out=out.clone(), alpha=0.5,
                       accumulate=True)
    assert rel_l2(out2.cpu(), 2 * want.cpu()) < 1e-5


@pytest.mark.parametrize("win,heads,rows,hd", [(5, 4, 352, 12), (3, 2, 32, 4), (7, 8, 48, 24)])
def test_gemm_tn_head_split(win, heads, rows, hd):
    ops = _ops()
    g = torch.Generator().manual_seed(win)
    Cc = heads * hd
    a = torch.randn(win, heads, rows, hd, generator=g).bfloat16().to(DEV)       # e.g. dq
    b = torch.randn(win * rows, Cc, generator=g).bfloat16().to(DEV)             # e.g. LN(x) rows
    out = ops.gemm_tn(a, ops.operand_heads(rows, hd), b, ops.operand_rows(Cc), win * rows, Cc, Cc)
    a_rows = a.float().permute(0, 2, 1, 3).reshape(win * rows, Cc)
    want = a_rows.T @ b.float()
    assert rel_l2(out.cpu(), want.cpu()) < 1e-5


@pytest.mark.parametrize("B,dims,cin,cout", [(2, (6, 5, 7), 8, 16), (1, (12, 12, 24), 48, 24), (2, (4, 4, 4), 4, 72)])
def test_gemm_tn_conv_taps(B, dims, cin, cout):
    """dW of a 3x3x3 'same' convolution in nn.Conv3d's layout, against autograd of F.conv3d."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(B, *dims, cin, generator=g).bfloat16().to(DEV)
    dy = torch.randn(B, *dims, cout, generator=g).bfloat16().to(DEV)
    vox = B * dims[0] * dims[1] * dims[2]
    out = ops.gemm_tn(dy, ops.operand_rows(cout), x, ops.operand_conv_taps(dims, cin, cin), vox, cout, 27 * cin,
                      perm_cin=cin)
    w = torch.zeros(cout, cin, 3, 3, 3, device=DEV, requires_grad=True)
    y = torch.nn.functional.conv3d(x.float().permute(0, 4, 1, 2, 3), w, padding=1)
    y.backward(dy.float().permute(0, 4, 1, 2, 3))
    assert rel_l2(out.view(cout, cin, 3, 3, 3).cpu(), w.grad.cpu()) < 1e-5


@pytest.mark.parametrize("cin,cout,dims,lrelu", [(24, 16, (6, 6, 8), True), (48, 5, (2, 3, 96), False), (144, 48, (8, 8, 8), True),
                                                (48, 5, (9, 13, 20), False)])     # 5-class head: data gradient as one 16-channel halo chunk
def test_bn_act_conv_all_parameter_gradients(cin, cout, dims, lrelu):
    """BatchNorm -> (LeakyReLU) -> conv3x3x3 with every parameter trainable, through the autograd glue: the decoder's
    conv_concat stage and a 5-class head whose depth (96) is outside the one-pass head kernel's window.
    Tolerance 6e-3: bf16 operands, fp32 sums (the BatchNorm gradients pass through one more bf16 tensor: 1e-2)."""
    import torch.nn as nn
    import torch.nn.functional as F
    import mivp_amd
    from mivp_amd import functional as Fn
    g = torch.Generator().manual_seed(cin + cout)
    r16 = lambda t: t.bfloat16().float()
    x = r16(torch.randn(2, cin, *dims, generator=g) + 0.2).requires_grad_(True)
    bn = nn.BatchNorm3d(cin)
    conv = nn.Conv3d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(cin, generator=g))
        bn.bias.copy_(0.1 * torch.randn(cin, generator=g))
        conv.weight.copy_(r16(conv.weight))
    z = bn(x)
    y = conv(F.leaky_relu(z, 0.01) if lrelu else z)
    dy = r16(torch.randn(y.shape, generator=g))
    y.backward(dy)
    want = {"x": x.grad, "bn_w": bn.weight.grad, "bn_b": bn.bias.grad, "w": conv.weight.grad, "b": conv.bias.grad}

    class Owner:
        pass
    owner = Owner()
    owner._wcache = Fn.WeightCache()
    bn2, conv2 = nn.BatchNorm3d(cin).to(DEV), nn.Conv3d(cin, cout, 3, 1, 1).to(DEV)
    bn2.load_state_dict({k: v for k, v in nn.BatchNorm3d(cin).state_dict().items()})
    with torch.no_grad():
        bn2.weight.copy_(bn.weight); bn2.bias.copy_(bn.bias); conv2.weight.copy_(conv.weight); conv2.bias.copy_(conv.bias)
    xd = x.detach().permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16).requires_grad_(True)
    out = Fn.bn_act_conv(owner, bn2, conv2, xd, lrelu=lrelu, out_f32=(cout <= 8), key="t")
    out.backward(dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV, out.dtype))
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu().permute(0, 4, 1, 2, 3), y.detach()) < 6e-3
    assert rel_l2(xd.grad.float().cpu().permute(0, 4, 1, 2, 3), want["x"]) < 1e-2
    assert rel_l2(conv2.weight.grad.cpu(), want["w"]) < 6e-3
    assert rel_l2(conv2.bias.grad.cpu(), want["b"]) < 6e-3
    assert rel_l2(bn2.weight.grad.cpu(), want["bn_w"]) < 1e-2
    assert rel_l2(bn2.bias.grad.cpu(), want["bn_b"]) < 1e-2


@pytest.mark.parametrize("cin", [1, 4])
def test_patch_embed_parameter_gradients(cin):
    import torch.nn as nn
    import mivp_amd
    from mivp_amd import functional as Fn
    g = torch.Generator().manual_seed(cin)
    x = torch.rand(2, cin, 12, 8, 10, generator=g)
    conv, bn = nn.Conv3d(cin, 48, 2, 2), nn.BatchNorm3d(48, eps=1e-6)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(48, generator=g))
        bn.bias.copy_(0.1 * torch.randn(48, generator=g))
    y = bn(conv(x))
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    import copy
    conv2, bn2 = copy.deepcopy(conv).to(DEV), copy.deepcopy(bn).to(DEV)
    for p in list(conv2.parameters()) + list(bn2.parameters()):
        p.grad = None
    bn2.running_mean.zero_(); bn2.running_var.fill_(1.0)
    out = Fn.patch_embed(None, conv2, bn2, x.to(DEV))
    out.backward(dy.permute(0, 2, 3, 4, 1).contiguous().to(DEV, torch.bfloat16))
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu().permute(0, 4, 1, 2, 3), y.detach()) < 6e-3
    assert rel_l2(conv2.weight.grad.cpu(), conv.weight.grad) < 6e-3
    assert rel_l2(bn2.weight.grad.cpu(), bn.weight.grad) < 6e-3
    assert rel_l2(bn2.bias.grad.cpu(), bn.bias.grad) < 6e-3
    # the conv bias sits in front of a training-mode BatchNorm: its true gradient is zero
    assert float(conv2.bias.grad.abs().max()) < 2e-2 * float(conv.weight.grad.abs().max())
