"""Known answers for the CPU restatement of the reference's asynchronous-inference kernels (oracle/asy_ref.c;
src/dagr/asynchronous/asy_tools/main.cu:14-188): hand-derivable cases, and agreement with the plain torch formulas the
kernels stand for (torch.nn.functional.linear / batch_norm in eval mode) on the listed rows, other rows untouched."""
import torch
import torch.nn.functional as F

from oracle import asy


def test_masked_isdiff_known_answers():
    x_old = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 8.0]])
    x_new = x_old.clone()
    x_new[1, 1] = 4.1                      # differs
    x_new[3, 0] = 7.0 + 1e-7               # within atol + rtol * x_new = 1e-8 + 1e-5 * 7
    idx = torch.tensor([3, 1, 0], dtype=torch.int64)
    out = asy.masked_isdiff(idx, x_old, x_new, 1e-8, 1e-5)
    assert idx.tolist() == [-1, 1, -1] and out.tolist() == [1]
    # the tolerance uses x_new WITHOUT an absolute value (main.cu:35): a negative entry shrinks it below zero
    a, b = torch.tensor([[-100.0]]), torch.tensor([[-100.0]])
    assert asy.masked_isdiff(torch.tensor([0]), a, b, 1e-8, 1e-5).tolist() == [0]      # |0| > 1e-8 - 1e-3 is true


def test_masked_bn_and_linear_match_torch_on_listed_rows():
    g = torch.Generator().manual_seed(0)
    n, cin, cout = 40, 19, 16
    x = torch.randn(n, cin, generator=g)
    idx = torch.tensor([5, 17, 0, 39, 8], dtype=torch.int64)
    w, b = torch.randn(cout, cin, generator=g), torch.randn(cout, generator=g)
    base = torch.randn(n, cout, generator=g)
    for add in (False, True):
        for bias in (b, None):
            out = base.clone()
            if bias is None:
                asy.masked_lin_no_bias(idx, x, out, w, add)
            else:
                asy.masked_lin(idx, x, out, w, bias, add)
            want = base.clone()
            want[idx] = F.linear(x[idx], w, bias) + (base[idx] if add else 0)
            assert torch.allclose(out, want, atol=1e-5, rtol=1e-5)
            untouched = torch.ones(n, dtype=torch.bool); untouched[idx] = False
            assert torch.equal(out[untouched], base[untouched])
    mean, var = torch.randn(cin, generator=g), torch.rand(cin, generator=g) + 0.5
    gamma, beta = torch.randn(cin, generator=g), torch.randn(cin, generator=g)
    out = torch.zeros(n, cin)
    asy.masked_inplace_BN(idx, x, out, mean, var, gamma, beta, 1e-5)
    want = torch.zeros(n, cin)
    want[idx] = F.batch_norm(x[idx], mean, var, gamma, beta, False, 0.1, 1e-5)
    assert torch.allclose(out, want, atol=1e-5, rtol=1e-5)
