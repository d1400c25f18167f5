"""CPU-side checks: the C-ABI library loads and exports every symbol include/eventad_hip.h declares
(no compute calls without a GPU), host-only helpers, config merge, geometry constants."""
import ctypes

import pytest
import torch


def test_library_exports_every_declared_symbol():
    from eventad_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 30
    lib = _lib.load()
    for name, argtypes in protos.items():
        fn = getattr(lib, name)
        assert fn.argtypes == argtypes and fn.restype is ctypes.c_int, name
    for must in ("eventad_insert_in_queue", "eventad_insert_in_queue_single", "eventad_fill_edges",   # = ev_graph_cuda
                 "eventad_sort_events", "eventad_build_ell", "eventad_l1_conv1", "eventad_l1_conv2",
                 "eventad_pool_scatter", "eventad_gemm_f32", "eventad_tail_blend", "eventad_roi_pool",
                 "eventad_gru_att_fwd_group", "eventad_gru_att_bwd_group", "eventad_clip_adamw"):
        assert must in protos


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    from eventad_amd import _lib
    lib = _lib.load()
    assert lib.eventad_insert_in_queue(None, None, None, None, 1, 1, 1, 1, 1, None) == -1
    assert lib.eventad_gemm_f32(None, 4, None, 4, None, 4, 1, None, 1, 1, None, None, None, None, 0, 0, None) == -1
    assert lib.eventad_roi_pool(None, None, None, None, 0, None, None, 0, 0, 2, 64, 64, 1, None, None) == -1
    # entry points added for the pooled levels' live-cell products, the asynchronous update and the dropout masks
    assert lib.eventad_tail_live_cells(None, None, None, 8, 1.0, 1.0, None, 1, None) == -1
    assert lib.eventad_tail_gemm(None, 84, None, 1728, None, 1728, 64, None, 1728, 84, None, None) == -1
    assert lib.eventad_incr_reset(None, 16, None) == -1
    assert lib.eventad_incr_edges(None, None, None, None, None, None, 0, 1, 0, 8, 8, 3, 1000, 16, 8, None, None, None, None) == -1
    assert lib.eventad_dropout_masks(None, 16, 0.7, 1, 0, None, None) == -1
    assert lib.eventad_conv2d_workspace_floats(1, 8, 8, 16, 16, 3, 3, None) == -1
    assert lib.eventad_clip_adamw(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, None, None, None) == -1
    assert lib.eventad_clip_adamw_out(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, None, None, None, None, None) == -1
    assert lib.eventad_gemm_group64(0, None, None) == -1 and lib.eventad_gemm_group64(9, None, None) == -1
    import ctypes
    desc = (ctypes.c_int64 * 16)()                           # one problem with null operands
    assert lib.eventad_gemm_group64(1, desc, None) == -1
    assert lib.eventad_copy_to_device(1, None, None, None, None) == -1 and lib.eventad_copy_to_device(0, None, None, None, None) == 0


def test_spiral_table_matches_oracle():
    from eventad_amd import spline
    from oracle import graph as G
    for r in (1, 3, 7):
        assert torch.equal(spline.spiral_offsets(r), G.spiral((2 * r + 1) ** 2).long())


def test_geometry_matches_oracle_config():
    from eventad_amd.geometry import Geometry
    from oracle import dagr_ref
    for W, H in ((64, 64), (256, 256), (640, 480), (1280, 720)):
        g, c = Geometry(W, H), dagr_ref.make_config(W, H, 4)
        assert (g.radius_px, g.radius_px, g.eff_r) == c.lut[0]
        for lv in range(4):
            assert g.voxel[lv] == (float(c.voxel[lv][0]), float(c.voxel[lv][1]))
            assert abs(g.cart_max[lv + 1] - float(c.cart_max[lv + 1])) == 0
        assert g.grid == [(56, 40), (28, 20), (14, 10), (7, 5)]
    assert Geometry(1280, 720).radius_px == 13 and Geometry(640, 480).radius_px == 7


def test_layer1_lut_equals_reference_lut_reordered():
    from eventad_amd import spline
    from eventad_amd.geometry import Geometry
    from oracle import dagr_ref
    g = Geometry(256, 256)
    w = torch.randn(25, 19, 16, generator=torch.Generator().manual_seed(0))
    mine = spline.lut_by_spiral_step(w, g, 20)
    ref, _ = dagr_ref.init_lut(w, 256, 256, rx=3, Mx=g.eff_r)
    d = spline.spiral_offsets(3)
    for s in range(49):
        assert torch.allclose(mine[s, :19], ref[d[s, 0] + 3, d[s, 1] + 3], atol=1e-7)
    assert float(mine[:, 19].abs().max()) == 0.0


def test_config_merge_rule_and_defaults(tmp_path):
    from eventad_amd.config.eventad_config import parse_eventad_args
    a = parse_eventad_args([])
    assert (a.radius, a.max_neighbors, a.pooling_dim_at_output, a.kernel_size) == (0.01, 16, "5x7", 5)
    assert (a.batch_size, a.learning_rate, a.grad_clip, a.x_dim, a.h_dim, a.num_workers) == (6, 0.001, 1.0, 64, 256, 4)
    b = parse_eventad_args(["--radius", "0.02", "--batch_size", "3"])      # a flag wins over the YAML
    assert b.radius == 0.02 and b.batch_size == 3 and b.max_neighbors == 16
    y = tmp_path / "e.yaml"
    y.write_text("h_dim: 999\nextra_key: 5\n")
    c = parse_eventad_args(["--eventad_config", str(y)])
    assert c.h_dim == 256 and c.extra_key == 5                             # argparse default present -> YAML ignored


def test_cpu_model_has_no_fallback():
    from eventad_amd.models.EventAD import EventADModel
    m = EventADModel(torch.nn.Module())
    with pytest.raises(RuntimeError):
        m.native_head()


def test_format_data_matches_oracle():
    from eventad_amd import synthetic
    from eventad_amd.dagr.utils.buffers import format_data
    from oracle import dagr_ref
    raw = synthetic.make_batch(64, 48, 500, 2, with_image=True)
    a, b = format_data(raw.clone()), dagr_ref.format_data(raw.clone())
    assert torch.equal(a.pos, b.pos) and torch.equal(a.x, b.x) and torch.equal(a.image, b.image) and a.t is None


def test_reference_checkpoints_with_detection_head_keys_round_trip():
    """The reference saves `model.state_dict()` of EventADModel -- including the YOLOX detection head under
    `dagr_model.head.*` -- and loads it strictly at the MODEL level (utils/model.py:87,116; utils/base.py:85).  The head's
    tensors are taken out in a hook (a parent's load never calls DAGR.load_state_dict), kept, and written back."""
    from types import SimpleNamespace
    from eventad_amd.config.eventad_config import parse_eventad_args
    from eventad_amd.dagr.model.networks.dagr import DAGR
    from eventad_amd.models.EventAD import EventADModel
    from eventad_amd import weights
    args = parse_eventad_args([])
    model = EventADModel(DAGR(args, height=64, width=64))
    net = model.dagr_model.backbone
    net._native = "stale"
    net.net._native = "stale"
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    extra = {"dagr_model.head.stem1.conv.weight": torch.randn(25, 64, 32), "dagr_model.head.cls_pred1.bias": torch.randn(2),
             "dagr_model.head.cnn_head.stems.0.conv.weight": torch.randn(8, 8, 1, 1)}
    sd.update(extra)
    sd["dagr_model.backbone.layer3.conv_block1.conv.weight"] = sd["dagr_model.backbone.layer3.conv_block1.conv.weight"] + 1.0
    res = model.load_state_dict(sd, strict=True)                       # model level, strict: as the reference does
    assert not res.missing_keys and not res.unexpected_keys
    assert net._native is None and net.net._native is None             # native weight images are rebuilt on next use
    out = model.state_dict()
    assert set(out) == set(sd)
    for k, v in extra.items():
        assert torch.equal(out[k], v)
    # a DAGR-level strict load of checkpoint['ema'] (utils/model.py:31-32) takes the same route
    dagr = DAGR(args, height=64, width=64)
    ema_sd = {k[len("dagr_model."):]: v for k, v in sd.items() if k.startswith("dagr_model.")}
    res = dagr.load_state_dict(ema_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys and set(dagr.state_dict()) == set(ema_sd)
    with pytest.raises(RuntimeError):                                  # anything else unknown is still an error
        model.load_state_dict(dict(sd, **{"dagr_model.backbone.bogus": torch.zeros(1)}), strict=True)


def test_unsupported_network_flags_raise():
    from eventad_amd.config.eventad_config import parse_eventad_args
    from eventad_amd.dagr.model.networks.dagr import DAGR
    for flag in (["--keep_temporal_ordering"], ["--no_events"], ["--img_net", "resnet18"], ["--pooling_aggr", "mean"]):
        with pytest.raises(NotImplementedError):
            DAGR(parse_eventad_args(flag), height=64, width=64)


def test_shim_serves_the_reference_import_names_without_editing_train_py(tmp_path):
    """`PYTHONPATH=<repo>/shim python train.py`: a script directory that has its own `models/EventAD.py` and
    `config/eventad_config.py` (as the reference's has) must still get this package's modules."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pkg, mod in (("models", "EventAD"), ("config", "eventad_config")):
        d = tmp_path / pkg
        d.mkdir()
        (d / "__init__.py").write_text("")
        (d / (mod + ".py")).write_text("raise ImportError('the script directory copy must be shadowed')\n")
    script = tmp_path / "train.py"
    script.write_text(
        "import sys\n"
        "from config.eventad_config import parse_eventad_args\n"
        "from models.EventAD import EventADModel\n"
        "from dagr.model.networks.dagr import DAGR\n"
        "from dagr.model.networks.ema import ModelEMA\n"
        "from dagr.utils.buffers import format_data\n"
        "import ev_graph_cuda, asy_tools\n"
        "from utils.test import collect_predictions\n"
        "a = parse_eventad_args([])\n"
        "print('OK', EventADModel.__module__, DAGR.__module__, a.radius, ev_graph_cuda.fill_edges_cuda.__name__)\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "shim"))
    out = subprocess.run([sys.executable, str(script)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK eventad_amd.models.EventAD eventad_amd.dagr.model.networks.dagr 0.01 fill_edges_cuda" in out.stdout
