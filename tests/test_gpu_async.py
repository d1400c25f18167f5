"""Asynchronous (event-by-event) inference, checked the way the reference checks its own asynchronous layers
(src/dagr/asynchronous/evaluate_flops.py:99-152: a dense run on all events against an initial dense run followed by
asynchronous updates): dense pass on N + k events == dense pass on N events + appended updates.  The dense pass itself is
checked against the oracle in test_gpu_backbone.py."""
import pytest
import torch

from eventad_amd import synthetic, weights
from eventad_amd._lib import call, ptr

pytestmark = pytest.mark.gpu


def _split(raw, fracs):
    """Per clip: the first part of its (time-ordered) events is the dense batch, the rest comes in len(fracs) later chunks.
    -> (row index of the dense batch, [row index of each update]) into the full batch."""
    batch = raw.batch
    dense, chunks = [], [[] for _ in fracs]
    for b in range(int(raw.num_graphs)):
        rows = torch.nonzero(batch == b).flatten()
        n = rows.numel()
        cuts = [int(n * f) for f in fracs] + [n]
        dense.append(rows[: cuts[0]])
        for c in range(len(fracs)):
            chunks[c].append(rows[cuts[c]: cuts[c + 1]])
    return torch.cat(dense), [torch.cat(c) for c in chunks]


def _edges_by_event(nb, n, to_full):
    """ELL of the first n nodes -> {destination event (row of the full batch): [source events in slot order]}."""
    ws = nb.ws
    K = 16 * n
    edges = torch.empty((2, K), dtype=torch.int64, device="cuda")
    call("eventad_ell_to_edges", ptr(ws["ell_src"]), ptr(ws["deg"]), ptr(ws["sid"]), n, 16, K, ptr(edges),
         torch.cuda.current_stream().cuda_stream)
    e = edges.cpu()
    e = e[:, e[1] >= 0]
    return torch.stack([to_full[e[0]], to_full[e[1]]])


def _outputs(res):
    out = []
    for key in ("out3", "out4"):
        o = res[key]
        n = int(o["n_dev"].item())
        out.append((n, o["x"][:n].clone(), o["pos"][:n].clone(), o["batch"][:n].clone()))
    L = res["ws"]["hand"][0]
    n, e = int(L.n_dev.item()), int(L.e_dev.item())
    out.append((n, L.x[:n].clone(), L.npos[:n].clone(), L.nbatch[:n].clone()))
    out.append((e, L.csr_src[:e].clone(), L.csr_dxy[:e].clone(), L.rowptr[: n + 1].clone()))
    return out


@pytest.mark.parametrize("W,H,n,B,fracs", [(64, 64, 3000, 2, (0.9, 0.95)),          # two updates of 150 events per clip; 0.7 events per pixel
                                           (256, 256, 20000, 3, (0.97,)),           # sparse: the row-scan dense pass
                                           (64, 64, 6000, 1, (0.5, 0.75, 0.99)),    # one clip, updates of 1500 / 1440 / 60 events, several per pixel
                                           (640, 480, 30000, 1, (0.98,)),           # radius 7
                                           (1280, 720, 60000, 1, (0.6, 0.99))])     # radius 13: twelve 64-step passes of the walk
@pytest.mark.parametrize("graph", [True, False])             # pooled stage of an update replayed from a recorded graph | launched
def test_dense_pass_equals_dense_pass_plus_asynchronous_updates(W, H, n, B, fracs, graph):
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    dev = torch.device("cuda:0")
    raw = synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True)
    if B > 1:
        # An event at t == time_window is clustered into the NEXT clip's voxel (pooling.py:31,55-56), and that voxel's clip
        # index is the one of whichever member the reference's scatter writes last (pooling.py:12-16): not defined by the
        # events alone, so not comparable between two event orders.  The one-clip cases keep such an event.
        raw.t[raw.t == 1000000] = 999999
    data = dagr_ref.format_data(raw)
    sd = weights.random_graph_state()
    maps = [m.to(dev) for m in data.image_feats]
    pos, x, batch = data.pos.to(dev), data.x.to(dev), data.batch.to(dev)
    full = NativeBackbone(sd, W, H, dev)
    want = _outputs(full.forward(pos, x, batch, B, maps))
    N = pos.shape[0]
    want_edges = _edges_by_event(full, N, torch.arange(N))
    assert full.check_flags() & ~1 == 0

    dense_rows, chunks = _split(data, fracs)
    nb = NativeBackbone(sd, W, H, dev)
    nb.async_capacity, nb.async_graph = N, graph
    d = dense_rows.to(dev)
    nb.forward(pos[d], x[d], batch[d], B, maps)
    got = None
    for rows in chunks:
        r = rows.to(dev)
        got = nb.append_events(pos[r], x[r], batch[r])
    torch.cuda.synchronize()
    assert nb.check_flags() & ~1 == 0 and nb._n_cur == N
    # node i of the incremental run is row to_full[i] of the full batch
    to_full = torch.cat([dense_rows] + chunks)
    got_edges = _edges_by_event(nb, N, to_full)
    key = lambda e: e[:, torch.argsort(e[1], stable=True)]   # by destination event, slot order kept
    a, b = key(want_edges), key(got_edges)
    assert a.shape == b.shape and torch.equal(a, b)          # the same in-edges, in the same order, for every event
    got = _outputs(got)
    for w_, g_ in zip(want[:3], got[:3]):
        assert w_[0] == g_[0] and w_[0] > 0
        assert torch.equal(w_[2], g_[2]) and torch.equal(w_[3], g_[3])              # node positions and clips
        assert float((w_[1] - g_[1]).abs().max()) <= 1e-3 * max(1.0, float(w_[1].abs().max()))
        assert torch.equal(w_[1], g_[1])                                            # ... in fact bit for bit
    for w_, g_ in zip(want[3], got[3]):
        assert torch.equal(torch.as_tensor(w_), torch.as_tensor(g_))                # pooled level-0 edges


def test_make_model_asynchronous_on_a_growing_batch():
    """The reference's usage: convert the model once, then call it with a graph that keeps growing
    (src/dagr/asynchronous/__init__.py:41-110).  Through the DAGR mirror: logits of the head after the updates == logits
    after one dense pass over everything; a batch that does not extend the previous one falls back to a dense pass."""
    import eventad_amd.compat as compat
    compat.install()
    from dagr.asynchronous import make_model_asynchronous, make_model_synchronous
    from tests.test_gpu_model import _build
    W = H = 64; B = 1; n = 4000
    args, model, sd, head_sd, raw, data = _build(W, H, B, n)
    model.eval()
    with torch.no_grad():
        want = torch.cat([o for f in model(data, data.bbox[:, 4])[1] for o in f], 0).clone()
    make_model_asynchronous(model, capacity=2048)
    net = model.dagr_model.backbone

    def prefix(m):
        d = data.clone()
        d.pos, d.x, d.batch = data.pos[:m], data.x[:m], data.batch[:m]
        return d

    calls = []
    orig = net._native.append_events
    net._native.append_events = lambda *a: (calls.append(int(a[0].shape[0])), orig(*a))[1]
    with torch.no_grad():
        for m in (3000, 3500, 3990, 4000):
            got = torch.cat([o for f in model(prefix(m), data.bbox[:, 4])[1] for o in f], 0)
    assert calls == [500, 490, 10]
    assert float((got - want).abs().max()) < 1e-3 and torch.equal(got.argmax(1), want.argmax(1))
    with torch.no_grad():
        model(prefix(2000), data.bbox[:, 4])                  # fewer events than seen: a dense pass again
    assert calls == [500, 490, 10] and net._native._n_dense == 2000
    make_model_synchronous(model)
    with torch.no_grad():
        got = torch.cat([o for f in model(data, data.bbox[:, 4])[1] for o in f], 0)
    assert torch.equal(got, want) and net._native.async_capacity == 0


def test_capacity_and_crowded_pixel_are_reported():
    from oracle import dagr_ref
    from eventad_amd.backbone import NativeBackbone
    dev = torch.device("cuda:0")
    W = H = 64
    data = dagr_ref.format_data(synthetic.make_batch(W, H, 2000, 1, with_image=False, precomputed_feats=True))
    maps = [m.to(dev) for m in data.image_feats]
    pos, x, batch = data.pos.to(dev), data.x.to(dev), data.batch.to(dev)
    nb = NativeBackbone(weights.random_graph_state(), W, H, dev)
    with pytest.raises(RuntimeError, match="async_capacity"):
        nb.forward(pos[:1000], x[:1000], batch[:1000], 1, maps)
        nb.append_events(pos[1000:1010], x[1000:1010], batch[1000:1010])
    nb.async_capacity = 64
    nb.forward(pos[:1000], x[:1000], batch[:1000], 1, maps)
    with pytest.raises(RuntimeError, match="capacity"):
        nb.append_events(pos[1000:1100], x[1000:1100], batch[1000:1100])
    crowd = pos[1000:1040].clone()
    crowd[:, :2] = crowd[0, :2]                               # 40 appended events in one pixel: more than the search collects
    nb.append_events(crowd, x[1000:1040], batch[1000:1040])
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="appended events in one pixel"):
        nb.check_flags()
