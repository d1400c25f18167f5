"""Known answers for the restated third-party arithmetic in oracle/dagr_ref.py
(spline basis, LUT, grid_cluster incl. the t==1.0 quirk, pooling, sampler, BN)."""
from types import SimpleNamespace

import torch

from oracle import dagr_ref as R


def test_spline_basis_degree1_open():
    b, i = R.spline_basis(torch.tensor([[0.0, 0.0], [0.5, 0.5], [1.0, 1.0], [0.3, 0.8]]))
    assert i[0].tolist() == [0, 1, 5, 6] and torch.allclose(b[0], torch.tensor([1.0, 0, 0, 0]))
    # attr .5 -> v=2 exactly -> taps (2,2) weight 1
    assert i[1].tolist() == [12, 13, 17, 18] and torch.allclose(b[1], torch.tensor([1.0, 0, 0, 0]))
    # attr 1 -> v=4: second tap wraps (4+1)%5 = 0 with weight 0
    assert i[2].tolist() == [24, 20, 4, 0] and torch.allclose(b[2], torch.tensor([1.0, 0, 0, 0]))
    # general: v=(1.2,3.2): taps x {1,2} w {.8,.2}; y {3,4} w {.8,.2}; index = ix + 5*iy
    assert i[3].tolist() == [16, 17, 21, 22]
    assert torch.allclose(b[3], torch.tensor([0.64, 0.16, 0.16, 0.04]), atol=1e-6)
    assert torch.allclose(b.sum(1), torch.ones(4), atol=1e-6)


def test_lut_equals_on_the_fly_basis_at_grid_points():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(25, 3, 2, generator=g)
    rx, M, W, H = 3, 0.03125, 256, 256
    lut, remap = R.init_lut(w, H, W, rx=rx, Mx=M)
    assert lut.shape == (7, 7, 3, 2)
    for dx, dy in [(-3, 2), (0, 0), (3, 3), (1, -2)]:
        attr = torch.tensor([[dx / (2 * M * W) + 0.5, dy / (2 * M * H) + 0.5]])
        b, i = R.spline_basis(attr)
        direct = (b[0, :, None, None] * w[i[0]]).sum(0)
        assert torch.allclose(lut[dx + rx, dy + rx], direct, atol=1e-6)
        ix, iy = R.lut_indices(attr, remap)
        assert (int(ix), int(iy)) == (dx + rx, dy + rx)


def test_spline_conv_single_edge_by_hand():
    w = torch.zeros(25, 2, 1)
    w[12] = torch.tensor([[1.0], [2.0]])           # centre cell (attr .5,.5)
    root = torch.tensor([[10.0, 0.0]])
    x = torch.tensor([[1.0, 1.0], [3.0, 4.0]])
    ei = torch.tensor([[0, 1, 0], [0, 1, 1]])      # self loops + 0->1, all at zero offset
    attr = torch.full((3, 2), 0.5)
    out = R.spline_conv(x, ei, attr, w, root, (3, 3, 0.03125), 256, 256)
    assert torch.allclose(out, torch.tensor([[3.0 + 10.0], [11.0 + 3.0 + 30.0]]))
    out2 = R.spline_conv(x, ei, attr, w, root, (3, 3, 0.03125), 256, 256, mode="grouped")
    assert torch.allclose(out, out2)
    empty = R.spline_conv(x, ei[:, :0], attr[:0], w, root, (3, 3, 0.03125), 256, 256)
    assert torch.allclose(empty, torch.tensor([[10.0], [30.0]]))     # spline_conv.py:67-73


def test_grid_cluster_and_t_equals_one_quirk():
    cfg = R.make_config(256, 256, 3)
    vs = cfg.voxel[0]
    assert [int(v) for v in ((torch.Tensor([.9999999] * 3) / vs[:3]).long() + 1)] == [56, 40, 1]
    pos = torch.tensor([[0.0, 0.0, 0.5, 0], [255 / 256, 255 / 256, 0.2, 0], [0.0, 0.0, 1.0, 0],
                        [0.0, 0.0, 0.1, 1], [10 / 256, 0.0, 0.3, 2]])
    c = R.grid_cluster(pos, vs, torch.Tensor([0, 0, 0, 0]), torch.Tensor([.9999999, .9999999, .9999999, 2]))
    # the t==1.0 event of clip 0 lands in clip 1's voxel (pooling.py:51-58 clusters t == time_window into the next plane)
    assert c.tolist() == [0, 55 + 56 * 39, 2240, 2240, 2 + 2 * 2240]


def test_pooling_max_mean_edges_and_batch():
    x = torch.tensor([[1.0, 5.0], [3.0, 2.0], [0.0, 0.0], [7.0, 7.0]])
    pos = torch.tensor([[1 / 256, 1 / 256, 0.1], [2 / 256, 1 / 256, 0.3], [100 / 256, 0.0, 0.2], [1 / 256, 0.0, 1.0]])
    batch = torch.tensor([0, 0, 0, 0])
    ei = torch.tensor([[0, 1, 0, 2, 2, 0], [0, 1, 1, 2, 0, 2]])
    vs = R.make_config(256, 256, 1).voxel[0]
    for aggr, want in (("max", [[3.0, 5.0], [0.0, 0.0], [7.0, 7.0]]), ("mean", [[2.0, 3.5], [0.0, 0.0], [7.0, 7.0]])):
        nx, npos, nb, nei, nattr = R.pooling(x, pos, batch, ei, vs, 0.0625, aggr, 256, 256, 1)
        assert nx.tolist() == want
        # edges: 0->1 collapses to a self loop (dropped); 2->0 and 0->2 survive, sorted by (src,dst)
        assert nei.t().tolist() == [[0, 1], [1, 0]]
        assert nb.tolist() == [0, 0, 0]
        # mean pos (1.5/256, 1/256) -> floored to the pixel grid
        assert torch.allclose(npos[0, :2], torch.tensor([1 / 256, 1 / 256]))
        assert torch.allclose(nattr[0, :2], (npos[0, :2] - npos[1, :2]) / (2 * 0.0625) + 0.5)


def test_sampler_matches_explicit_bilinear_b1_and_b3():
    g = torch.Generator().manual_seed(1)
    W = H = 16
    for B in (1, 3):
        feat = torch.randn(B, 4, 8, 8, generator=g)
        px = torch.randint(0, W, (20,), generator=g).float()
        py = torch.randint(0, H, (20,), generator=g).float()
        b = torch.randint(0, B, (20,), generator=g)
        got = R.sample_features(torch.stack([px / W, py / H], 1), b, feat, W, H)
        ix = px / (W - 1) * 7
        iy = py / (H - 1) * 7
        x0, y0 = ix.floor().long(), iy.floor().long()
        fx, fy = ix - x0, iy - y0
        x1, y1 = (x0 + 1).clamp(max=7), (y0 + 1).clamp(max=7)
        f = feat[b]                                                    # [n,C,h,w]
        n = torch.arange(20)
        want = (f[n, :, y0, x0] * ((1 - fx) * (1 - fy))[:, None] + f[n, :, y0, x1] * (fx * (1 - fy))[:, None]
                + f[n, :, y1, x0] * ((1 - fx) * fy)[:, None] + f[n, :, y1, x1] * (fx * fy)[:, None])
        assert torch.allclose(got, want, atol=1e-5)


def test_bn_eval_is_affine_fold():
    sd = {"p.weight": torch.tensor([2.0]), "p.bias": torch.tensor([1.0]),
          "p.running_mean": torch.tensor([0.5]), "p.running_var": torch.tensor([4.0])}
    y = R.batch_norm_eval(torch.tensor([[2.5]]), sd, "p.")
    assert torch.allclose(y, torch.tensor([[2.0 * (2.5 - 0.5) / (4.0 + 1e-5) ** 0.5 + 1.0]]))


def test_config_constants_match_survey():
    c2 = R.make_config(256, 256, 8)
    assert c2.lut[0] == (3, 3, 0.03125) and abs(c2.eff_r - 0.03125) < 1e-12
    assert [(rx, ry) for rx, ry, _ in c2.lut[1:]] == [(10, 13), (19, 26), (37, 52), (74, 103)]
    c4 = R.make_config(1280, 720, 8)
    assert c4.lut[0][:2] == (13, 13) and abs(c4.eff_r - 0.021875) < 1e-12


def test_on_demand_lut_cells_equal_the_full_lut():
    """lut_cells (LUT entries evaluated per used cell) == init_lut, bit for bit, for every conv geometry at 256x256
    (all cells; the layer-5 table is sampled), and spline_conv(mode="ondemand") == mode="gather"."""
    from eventad_amd import weights
    sd = weights.random_graph_state()
    cfg = R.make_config(256, 256, 2)
    g = torch.Generator().manual_seed(5)
    for li, name in enumerate(["conv_block1.", "layer2.", "layer3.", "layer4.", "layer5."]):
        rx, ry, M = cfg.lut[li]
        w = sd["backbone." + name + "conv_block2.conv.weight"]
        lut, _ = R.init_lut(w, 256, 256, rx=rx, Mx=M, ry=ry)
        n = (2 * rx + 1) * (2 * ry + 1)
        ix = torch.arange(2 * rx + 1).repeat_interleave(2 * ry + 1)
        iy = torch.arange(2 * ry + 1).repeat(2 * rx + 1)
        if n > 20000:
            sel = torch.randperm(n, generator=g)[:20000]
            ix, iy = ix[sel], iy[sel]
        assert torch.equal(R.lut_cells(w, 256, 256, rx, M, ix, iy, ry=ry), lut[ix, iy]), name
    # a small conv through both modes
    rx, ry, M = cfg.lut[0]
    n, E = 300, 2000
    x = torch.randn(n, 16, generator=g)
    ei = torch.randint(0, n, (2, E), generator=g)
    attr = torch.randint(-rx, rx + 1, (E, 3), generator=g).float() / (2 * M * 256) + 0.5      # pixel offsets inside the radius
    w, root = sd["backbone.conv_block1.conv_block2.conv.weight"], sd["backbone.conv_block1.conv_block2.conv.lin.weight"]
    a = R.spline_conv(x, ei, attr, w, root, cfg.lut[0], 256, 256, mode="gather")
    b = R.spline_conv(x, ei, attr, w, root, cfg.lut[0], 256, 256, mode="ondemand", chunk=512)
    assert torch.equal(a, b)
