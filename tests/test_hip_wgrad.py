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
