"""GPU: the HIP versions of the reference's `asy_tools` kernels, through the module `import asy_tools` resolves to after
compat.install(), against the C restatement (oracle/asy_ref.c) -- bit for bit, on the same seeded inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_asy_tools_module_equals_oracle_bitwise():
    import eventad_amd.compat as compat
    compat.install()
    import asy_tools
    from oracle import asy
    g = torch.Generator().manual_seed(1)
    for n, cin, cout, k in ((500, 19, 16, 37), (3000, 130, 64, 900), (64, 64, 64, 64), (10, 3, 70, 1)):
        x = torch.randn(n, cin, generator=g)
        idx = torch.randperm(n, generator=g)[:k].to(torch.int64).contiguous()
        w, b = torch.randn(cout, cin, generator=g), torch.randn(cout, generator=g)
        base = torch.randn(n, cout, generator=g)
        for add in (False, True):
            for bias in (b, None):
                want, got = base.clone(), base.clone().cuda()
                if bias is None:
                    asy.masked_lin_no_bias(idx, x, want, w, add)
                    asy_tools.masked_lin_no_bias(idx.cuda(), x.cuda(), got, w.cuda(), add)
                else:
                    asy.masked_lin(idx, x, want, w, bias, add)
                    asy_tools.masked_lin(idx.cuda(), x.cuda(), got, w.cuda(), bias.cuda(), add)
                assert torch.equal(got.cpu(), want), (n, cin, cout, add, bias is None)
        mean, var = torch.randn(cin, generator=g), torch.rand(cin, generator=g) + 0.5
        gamma, beta = torch.randn(cin, generator=g), torch.randn(cin, generator=g)
        want, got = torch.zeros(n, cin), torch.zeros(n, cin).cuda()
        asy.masked_inplace_BN(idx, x, want, mean, var, gamma, beta, 1e-5)
        asy_tools.masked_inplace_BN(idx.cuda(), x.cuda(), got, mean.cuda(), var.cuda(), gamma.cuda(), beta.cuda(), 1e-5)
        assert torch.equal(got.cpu(), want)
        x_new = x.clone()
        changed = idx[:: 3]
        x_new[changed, torch.randint(0, cin, (len(changed),), generator=g)] += 0.01
        i_ref, i_gpu = idx.clone(), idx.clone().cuda()
        out_ref = asy.masked_isdiff(i_ref, x, x_new, 1e-8, 1e-5)
        out_gpu = asy_tools.masked_isdiff(i_gpu, x.cuda(), x_new.cuda(), 1e-8, 1e-5)
        assert torch.equal(i_gpu.cpu(), i_ref) and torch.equal(out_gpu.cpu(), out_ref) and len(out_ref) >= len(changed)
    with pytest.raises(RuntimeError):
        asy_tools.masked_lin(idx, x.cuda(), base.cuda(), w.cuda(), b.cuda(), False)      # CPU tensor: as AT_ASSERTM
