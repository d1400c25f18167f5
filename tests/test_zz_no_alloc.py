"""eventad_sort_events_static (no workspace arguments, allocates nothing) == eventad_sort_events, bit for bit, on two
calls in a row: the second with a smaller input, so stale contents of the library's fixed workspace would show."""
import ctypes

import pytest
import torch

from eventad_amd import synthetic
from eventad_amd._lib import call, load, ptr
from eventad_amd.geometry import Geometry


def _run(static, data, B, W, H, g, tabs, dev):
    N = int(data.pos.shape[0])
    P = B * H * W
    st = torch.cuda.current_stream().cuda_stream
    i32 = lambda *s: torch.full(s, -7, dtype=torch.int32, device=dev)
    f32 = lambda *s: torch.full(s, -7.0, dtype=torch.float32, device=dev)
    o = dict(key=i32(N), tq=i32(N), bev=i32(N), pix_start=i32(P + 1), sid=i32(N), counters=i32(4), key_s=i32(N), ts_s=i32(N),
             ev2=i32(N, 2), posn_s=f32(N, 3), x0s=f32(N, 20), node_slot=i32(N), node_vox=i32(N), ovf_list=i32(max(N, 1)))
    pos, pol, batch = data.pos.to(dev).contiguous(), data.x.reshape(-1).float().to(dev).contiguous(), data.batch.to(dev).contiguous()
    m0 = data.image_feats[0].to(dev).permute(0, 2, 3, 1).contiguous()
    common_out = (ptr(o["key_s"]), ptr(o["ts_s"]), ptr(o["ev2"]), ptr(o["posn_s"]), ptr(o["x0s"]), g.voxel[0][0], g.voxel[0][1],
                  g.grid[0][0], g.grid[0][1], ptr(tabs[0]), ptr(tabs[1]), ptr(o["node_slot"]), ptr(o["node_vox"]), ptr(o["ovf_list"]))
    if static:
        call("eventad_sort_events_static", ptr(pos), ptr(batch), ptr(pol), ptr(m0), m0.shape[1], m0.shape[2], N, B, W, H, g.T,
             ptr(o["key"]), ptr(o["tq"]), ptr(o["bev"]), ptr(o["pix_start"]), ptr(o["sid"]), ptr(o["counters"]), *common_out, st)
    else:
        pix_count = torch.full((P + 1,), 12345, dtype=torch.int32, device=dev)
        scan_tmp = i32(2 * (max(P, 1 << 16) // 2048 + 4))
        sid_tmp, rec = i32(N), i32(max(N, 1), 8)               # with `rec` the workspace form places 32-byte records, the
        call("eventad_sort_events", ptr(pos), ptr(batch), ptr(pol), ptr(m0), m0.shape[1], m0.shape[2], N, B, W, H, g.T,   # static form bare ids
             ptr(o["key"]), ptr(o["tq"]), ptr(o["bev"]), ptr(pix_count), ptr(o["pix_start"]), ptr(scan_tmp), ptr(sid_tmp),
             ptr(rec), ptr(o["sid"]), ptr(o["counters"]), *common_out, 0, st)
    torch.cuda.synchronize()
    n_ovf = int(o["counters"][2].item())
    o["ovf_list"] = torch.sort(o["ovf_list"][:n_ovf]).values          # appended with an atomic counter: a set, not a sequence
    return o


@pytest.mark.gpu
def test_static_sort_equals_workspace_sort_twice():
    from oracle import dagr_ref
    dev = torch.device("cuda:0")
    W = H = 256
    g = Geometry(W, H, 0.01, 16, "5x7", 1000000)
    tabs = [t.to(dev).contiguous() for t in g.pixel_voxel_tables()]
    for n, B in ((50000, 8), (9000, 3)):                              # the second call is smaller in N and in B*H*W
        data = dagr_ref.format_data(synthetic.make_batch(W, H, n, B, with_image=False, precomputed_feats=True))
        want = _run(False, data, B, W, H, g, tabs, dev)
        got = _run(True, data, B, W, H, g, tabs, dev)
        for k in want:
            assert torch.equal(got[k], want[k]), (n, B, k)
        assert int(got["counters"][3].item()) >= 1                    # longest pixel segment: the scan really ran


def test_static_sort_rejects_sizes_beyond_its_workspace():
    """Needs no GPU: the size check comes before any device work."""
    lib = load()
    z = ctypes.c_void_p(0)
    rc = lib.eventad_sort_events_static(z, z, z, z, 1, 1, 10, 32, 256, 256, 1000000, z, z, z, z, z, z, z, z, z, z, z,
                                        0.1, 0.1, 1, 1, z, z, z, z, z, z)
    assert rc == -3                                                    # EVENTAD_ERR_TOO_LARGE: 32*256*256 pixels > 2^20
