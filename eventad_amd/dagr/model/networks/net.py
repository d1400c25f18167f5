"""Mirror of `dagr.model.networks.net.Net` (reference src/dagr/model/networks/net.py:30-197):
same constructor arguments, same parameter names (conv_block1 / layer2..5 with
conv_block{1,2}.conv.{weight,lin.weight}, norm.module.*, lin.mlp.weight, norm_skip.module.*, and
`net.*` for the CNN branch), so reference state dicts load.  `forward` does not run torch_geometric
layers: it hands the batch to the native HIP pipeline (eventad_amd/backbone.py)."""
import torch
import torch.nn as nn

from ....backbone import NativeBackbone
from ....weights import LAYER_NAMES, LAYER_DIMS
from .net_img import HookModule


class _Holder(nn.Module):
    """Empty module used to build the reference's dotted parameter names."""
    pass


def _bn_holder(c):
    h = _Holder()
    h.module = nn.BatchNorm1d(c)
    return h


def _conv_holder(cin, cout):
    h = _Holder()
    h.weight = nn.Parameter(torch.empty(25, cin, cout).uniform_(-1, 1) / (4.0 * cin) ** 0.5)
    h.lin = nn.Linear(cin, cout, bias=False)
    return h


class GraphOut:
    """What the head reads of a PyG `Data` (x, pos, batch); padded device arrays + device count."""

    def __init__(self, d):
        self._d = d
        self.n_dev = d["n_dev"]
        self.x_padded, self.pos_padded, self.batch_padded = d["x"], d["pos"], d["batch"]
        self._n = None

    def _count(self):
        if self._n is None:
            self._n = int(self.n_dev.item())          # the one host sync of the API-level path
        return self._n

    @property
    def x(self):
        return self.x_padded[: self._count()]

    @property
    def pos(self):
        return self.pos_padded[: self._count()]

    @property
    def batch(self):
        return self.batch_padded[: self._count()].long()

    @property
    def edge_index(self):
        L, n = self._d["level"], self._count()
        e = int(L.e_dev.item())
        rp = L.rowptr[: n + 1].long()
        dst = torch.repeat_interleave(torch.arange(n, device=rp.device), rp[1:] - rp[:-1])
        return torch.stack([L.csr_src[:e].long(), dst])


class GraphOutGrad(GraphOut):
    """GraphOut whose `x` carries autograd history (Net.forward_with_grad)."""

    def __init__(self, d, x):
        super().__init__(d)
        self._x, self._n = x, int(x.shape[0])

    @property
    def x(self):
        return self._x


class _BackboneFn(torch.autograd.Function):
    """Net.forward as ONE differentiable operation: the forward is the native pass, the backward is
    graph_backward.BackboneBackward on the workspace that pass left behind -- what the reference gets from autograd through
    MySplineConv / Pooling / grid_sample (spline_conv.py:39-78, pooling.py:51-97, net.py:200-228).  Inputs that receive a
    gradient: the five CNN feature maps and the 25 weight tensors of the five Layers (the weights' VALUES are read from
    the backbone's native images, built from the same parameters just before; they are arguments so that autograd routes
    their gradients).  No gradient for BatchNorm parameters (eval-mode statistics are folded) or node positions."""

    @staticmethod
    def forward(ctx, net, data, nhwc, n_maps, *tensors):
        nb = net._native
        maps = [t.detach() for t in tensors[:n_maps]]
        res = nb.forward(data.pos, data.x, data.batch, data.num_graphs, maps, feats_are_nhwc=nhwc)
        n3, n4 = int(res["out3"]["n_dev"].item()), int(res["out4"]["n_dev"].item())
        ctx.net, ctx.nhwc, ctx.n_maps, ctx.pass_id, ctx.res = net, nhwc, n_maps, nb._pass_id, res
        return res["out3"]["x"][:n3].clone(), res["out4"]["x"][:n4].clone()

    @staticmethod
    def backward(ctx, g3, g4):
        from ....graph_backward import BackboneBackward
        net, nb = ctx.net, ctx.net._native
        if nb is None or nb._pass_id != ctx.pass_id:
            raise RuntimeError("backward through Net.forward_with_grad must run before the backbone's next pass "
                               "(the gradients are computed on the workspace of the forward pass)")
        sd = {"backbone." + k: v for k, v in net.state_dict().items() if not k.startswith("net.")}
        r = BackboneBackward(nb, sd).backward(g3.contiguous().float(), g4.contiguous().float())
        dmaps = [m if ctx.nhwc else m.permute(0, 3, 1, 2) for m in r["dmaps"]]
        dw = [r["weights"]["backbone." + k] for k in net.grad_weight_keys()]
        return (None, None, None, None) + tuple(dmaps) + tuple(dw)


class Net(nn.Module):
    def __init__(self, args, height, width):
        super().__init__()
        self.height, self.width = int(height), int(width)
        self.use_image = bool(getattr(args, "use_image", True))
        self.num_scales = int(getattr(args, "num_scales", 2))
        self.radius = float(args.radius)
        self.max_neighbors = int(args.max_neighbors)
        self.pooling_dim_at_output = str(getattr(args, "pooling_dim_at_output", "5x7"))
        if getattr(args, "activation", "relu") != "relu" or getattr(args, "aggr", "sum") != "sum" or \
                int(getattr(args, "kernel_size", 5)) != 5 or int(getattr(args, "edge_attr_dim", 2)) != 2:
            raise NotImplementedError("native path implements the dagr-s configuration (relu, sum, kernel_size 5, dim 2)")
        if not self.use_image or self.num_scales != 2:
            raise NotImplementedError("EventAD needs use_image=True and num_scales=2 (models/EventAD.py:362)")
        if getattr(args, "keep_temporal_ordering", False) or getattr(args, "no_events", False):
            raise NotImplementedError("--keep_temporal_ordering (pooling.py:70-73) and --no_events are not implemented natively")
        if getattr(args, "img_net", "resnet50") != "resnet50":
            raise NotImplementedError("the native CNN branch is ResNet-50 (img_net=%r asked)" % (args.img_net,))
        if str(getattr(args, "pooling_aggr", "max")) != "max":
            raise NotImplementedError("native pooling implements pooling_aggr=max (pools 1-3; pool 4 is the mean, net.py:95)")
        self.net = HookModule()
        for name, (cin, cout) in zip(LAYER_NAMES, LAYER_DIMS):
            layer = _Holder()
            layer.conv_block1 = _Holder()
            layer.conv_block1.conv = _conv_holder(cin, cout)
            layer.conv_block1.norm = _bn_holder(cout)
            layer.conv_block2 = _Holder()
            layer.conv_block2.conv = _conv_holder(cout, cout)
            layer.conv_block2.norm = _bn_holder(cout)
            layer.conv_block2.lin = _Holder()
            layer.conv_block2.lin.mlp = nn.Linear(cin, cout, bias=False)
            layer.conv_block2.norm_skip = _bn_holder(cout)
            setattr(self, name, layer)
        self.out_channels = [64, 64]
        self.out_channels_cnn = [256, 256]
        self._native = None
        self._async_capacity, self._async_seen = 0, None
        # any load that reaches this module (its own, DAGR's or EventADModel's) makes the native weight images stale
        self.register_load_state_dict_post_hook(Net._drop_native)

    @staticmethod
    def _drop_native(module, incompatible_keys):
        module._native = None

    def prepare(self, device=None):
        """(Re)build the native weight images from the current parameters (called by DAGR.cache_luts)."""
        device = device or next(self.parameters()).device
        sd = {"backbone." + k: v for k, v in self.state_dict().items() if not k.startswith("net.")}
        self._native = NativeBackbone(sd, self.width, self.height, device, self.radius, self.max_neighbors,
                                      self.pooling_dim_at_output)
        return self._native

    def _maps(self, data):
        feats = getattr(data, "image_feats", None)
        if feats is not None:
            return feats, False                              # precomputed CNN features, NCHW like the reference's maps
        return self.net(data.image), True                    # live CNN branch (NHWC out of csrc/cnn.hip)

    @torch.no_grad()
    def forward(self, data, reset=True):
        if self._async_capacity > 0:
            return self.async_forward(data, reset=False)
        return self.sync_forward(data)

    @torch.no_grad()
    def sync_forward(self, data, reset=True):
        self.front(data)
        return self.tail()

    # ---- fine-tuning the graph branch: loss.backward() reaches the Layers' weights
    def grad_weight_keys(self):
        """The 25 weight tensors that forward_with_grad differentiates (state-dict names)."""
        keys = []
        for name in LAYER_NAMES:
            keys += ["%s.conv_block1.conv.weight" % name, "%s.conv_block1.conv.lin.weight" % name,
                     "%s.conv_block2.conv.weight" % name, "%s.conv_block2.conv.lin.weight" % name, "%s.conv_block2.lin.mlp.weight" % name]
        return keys

    def forward_with_grad(self, data):
        """forward() whose out3.x / out4.x carry autograd history to the weights of the five Layers and to the CNN feature
        maps (precomputed maps that require grad; the live trunk's maps are detached: its backward is not built).  The native
        weight images are rebuilt from the current parameters first (an optimiser step may have changed them); eval-mode
        BatchNorm.  backward() must run before this backbone's next pass."""
        self.prepare(data.pos.device)
        with torch.no_grad():
            feats, nhwc = self._maps(data)
        named = dict(self.named_parameters())
        params = [named[k] for k in self.grad_weight_keys()]
        x3, x4 = _BackboneFn.apply(self, data, nhwc, len(feats), *feats, *params)
        res = self._native._last_result
        self.last = res
        return [GraphOutGrad(res["out3"], x3), GraphOutGrad(res["out4"], x4)], None

    # ---- asynchronous mode (dagr.asynchronous.make_model_asynchronous)
    def enable_asynchronous(self, capacity):
        self._async_capacity, self._async_seen = int(capacity), None

    def disable_asynchronous(self):
        self._async_capacity, self._async_seen = 0, None
        if self._native is not None:
            self._native.async_capacity = 0

    @torch.no_grad()
    def async_forward(self, data, reset=False):
        """The first call, a call with reset=True, or a batch that does not extend the events seen so far: dense pass (the
        reference's graph initialisation, asynchronous/conv.py:28-58).  A batch with the same clips whose first rows are
        the events already seen and whose further rows are new events (asynchronous/conv.py:108-110: more nodes than the
        stored graph): only the new events are processed (NativeBackbone.append_events).  The caller vouches that the rows
        already seen are unchanged and that the new events are newer than every event of their clip so far."""
        if self._native is None:
            self.prepare(data.pos.device)
        nb = self._native
        nb.async_capacity = self._async_capacity
        N, B = int(data.pos.shape[0]), int(data.num_graphs)
        seen = self._async_seen
        extend = (not reset and seen is not None and seen[1] == B and seen[0] <= N and nb._n_cur == seen[0]
                  and N - nb._n_dense <= self._async_capacity)
        if extend:
            n0 = seen[0]
            res = nb.append_events(data.pos[n0:], data.x[n0:], data.batch[n0:])
        else:
            self.front(data)
            res = nb.tail(nb.ws["hand"][0])
        self._async_seen = (N, B)
        self.last = res
        return [GraphOut(res["out3"]), GraphOut(res["out4"])], None

    @torch.no_grad()
    def front(self, data):
        """Event-level stage (graph build, Layer 1, pool1) -> level-0 graph in the native hand-off set A."""
        if self._native is None:
            self.prepare(data.pos.device)
        feats, nhwc = self._maps(data)
        return self._native.front(data.pos, data.x, data.batch, data.num_graphs, feats, feats_are_nhwc=nhwc)

    @torch.no_grad()
    def tail(self, hand=None):
        """Pooled stage (Layers 2-5, pools 2-4) on a hand-off set (default: the one front() just wrote)."""
        res = self._native.tail(hand if hand is not None else self._native.ws["hand"][0])
        self.last = res
        return [GraphOut(res["out3"]), GraphOut(res["out4"])], None
