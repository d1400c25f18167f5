"""CNN feature branch: mirror of `dagr.model.networks.net_img.HookModule` over torchvision's
resnet50 (reference src/dagr/model/networks/net_img.py:42-134, net.py:41-49).

The nn.Conv2d / nn.BatchNorm2d modules below only HOLD the parameters under torchvision's key names
(`module.conv1.weight`, `module.layer1.0.bn1.running_mean`, ..., `feature_dconv.{0..4}`,
`output_dconv.{0,1}`) so reference checkpoints load; `forward` runs the hand-written NHWC implicit-GEMM
kernels of csrc/cnn.hip through the C ABI (`NativeCNN`).  Only what EventAD consumes is computed: the
five feature taps (raw conv1 output, layer1..4) through their 1x1 reducers; avgpool/fc and
`output_dconv` (detection-head inputs, net_img.py:131-133) are skipped."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...._lib import call, ptr, stream_handle

BLOCKS = [3, 4, 6, 3]
PLANES = [64, 128, 256, 512]
TAPS = [64, 256, 512, 1024, 2048]


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        return F.relu(self.bn3(self.conv3(y)) + idt)


class _ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, (nb, pl) in enumerate(zip(BLOCKS, PLANES)):
            blocks = []
            for bi in range(nb):
                blocks.append(_Bottleneck(inpl, pl, 2 if (bi == 0 and li > 0) else 1, bi == 0))
                inpl = pl * 4
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
        self.fc = nn.Linear(2048, 1000)          # unused; kept for checkpoint key compatibility


class HookModule(nn.Module):
    def __init__(self, feature_channels=(16, 64, 64, 64, 64), output_channels=(256, 256)):
        super().__init__()
        self.module = _ResNet50()
        self.feature_dconv = nn.ModuleList([nn.Conv2d(ci, co, 1) for ci, co in zip(TAPS, feature_channels)])
        self.output_dconv = nn.ModuleList([nn.Conv2d(ci, co, 1) for ci, co in zip(TAPS[3:], output_channels)])
        self.feature_channels = list(feature_channels)
        self.register_load_state_dict_post_hook(HookModule._drop_native)

    # (train()/eval() do not touch the native weight images: the trunk always runs with eval-mode BatchNorm folded into
    # its convolutions, and EventADModel.extract_features calls dagr_model.eval() on every step -- dropping the images
    # there meant 58 strided weight copies, 100 MB, per forward pass)

    def _apply(self, fn, *a, **k):                      # .to()/.cuda(): weights move -> rebuild the native images
        self._native = None
        return super()._apply(fn, *a, **k)

    @staticmethod
    def _drop_native(module, incompatible_keys):            # runs for nested loads too (a parent's load_state_dict)
        module._native = None

    @torch.no_grad()
    def forward(self, image):
        """image f32[B,3,H,W] in [0,1] -> five NHWC feature maps [B,h,w,C]."""
        if image.device.type != "cuda":
            raise RuntimeError("HookModule: the CNN branch runs on the GPU only (no CPU fallback)")
        if getattr(self, "_native", None) is None:
            object.__setattr__(self, "_native", NativeCNN(self, image.device))
        return self._native.forward(image)


def _fold(bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


def _wk(conv, cin_pad=None):
    """Conv2d weight [Cout,Cin,KH,KW] -> implicit-GEMM operand [KH*KW*Cin, Cout]."""
    w = conv.weight.detach().float()
    if cin_pad and cin_pad > w.shape[1]:
        w = torch.cat([w, torch.zeros(w.shape[0], cin_pad - w.shape[1], *w.shape[2:], device=w.device)], 1)
    return w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous()


class NativeCNN:
    """Launch sequence of the trunk over csrc/cnn.hip (eval-mode BatchNorm folded into the conv epilogues)."""

    def __init__(self, hook, device):
        m = hook.module
        self.dev = device
        self.stem_w = _wk(m.conv1, 4)
        self.stem_bn = _fold(m.bn1)
        self.blocks = []
        for li in range(4):
            for blk in getattr(m, "layer%d" % (li + 1)):
                d = dict(w1=_wk(blk.conv1), bn1=_fold(blk.bn1), w2=_wk(blk.conv2), bn2=_fold(blk.bn2), w3=_wk(blk.conv3),
                         bn3=_fold(blk.bn3), stride=blk.conv2.stride[0], planes=blk.conv1.out_channels, tap=False, wd=None)
                if blk.downsample is not None:
                    d["wd"], d["bnd"] = _wk(blk.downsample[0]), _fold(blk.downsample[1])
                self.blocks.append(d)
            self.blocks[-1]["tap"] = True
        self.red = [(_wk(c), c.bias.detach().float().contiguous()) for c in hook.feature_dconv]
        self._need, self._plan, self._ws = {}, None, None

    def _conv(self, x, w, B, IH, IW, Cin, Cout, K, stride, pad, bias=None, bn=None, residual=None, relu=0, out=None):
        OH, OW = (IH + 2 * pad - K) // stride + 1, (IW + 2 * pad - K) // stride + 1
        if self._plan is not None:                          # sizing pass: which split-K scratch does this shape want?
            n = ctypes.c_int64(0)
            call("eventad_conv2d_workspace_floats", B, OH, OW, Cin, Cout, K, K, ctypes.byref(n))
            self._plan = max(self._plan, n.value)
            return None, OH, OW
        if out is None:
            out = torch.empty((B, OH, OW, Cout), device=self.dev)
        ws = self._ws
        call("eventad_conv2d_nhwc", ptr(x), ptr(w), ptr(out), ptr(bias), ptr(bn[0]) if bn else None, ptr(bn[1]) if bn else None,
             ptr(residual), B, IH, IW, Cin, OH, OW, Cout, K, K, stride, pad, relu, ptr(ws), ws.numel() if ws is not None else 0,
             stream_handle())
        return out, OH, OW

    def workspace_floats(self, B, H, W):
        """Largest split-K scratch any convolution of a pass over [B,3,H,W] images wants (shapes that cannot fill 256 CUs
        on their own: layers 3-4)."""
        key = (B, H, W)
        if key not in self._need:
            self._plan = 0
            try:
                self._run(None, B, H, W)
                self._need[key] = self._plan
            finally:
                self._plan = None
        return self._need[key]

    def forward(self, image):
        B, _, H, W = image.shape
        # The split-K scratch belongs to THIS pass: slab partials written by a convolution are summed by the launch behind
        # it, so two passes in flight (backbone graphs of consecutive batches on different streams) must not share one.
        # Allocated per call, it comes from the stream's pool of torch's allocator when run eagerly and from the graph's
        # private pool under capture.
        need = self.workspace_floats(B, H, W)
        self._ws = torch.empty(need, device=self.dev) if need else None
        try:
            return self._run(image.contiguous().float(), B, H, W)
        finally:
            self._ws = None

    def _run(self, image, B, H, W):
        plan = image is None
        st = None if plan else stream_handle()
        x0 = None
        if not plan:
            x0 = torch.empty((B, H, W, 4), device=self.dev)
            call("eventad_image_to_nhwc4", ptr(image), ptr(x0), B, H, W, st)
        c1, h, w = self._conv(x0, self.stem_w, B, H, W, 4, 64, 7, 2, 3)                     # hook on `conv1`: raw output
        taps = [(c1, h, w, 64)]
        ph, pw = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        x = None
        if not plan:
            x = torch.empty((B, ph, pw, 64), device=self.dev)
            call("eventad_bn_relu_maxpool", ptr(c1), ptr(self.stem_bn[0]), ptr(self.stem_bn[1]), ptr(x), B, h, w, 64, ph, pw, st)
        h, w, cin = ph, pw, 64
        for d in self.blocks:
            pl, s = d["planes"], d["stride"]
            y1, _, _ = self._conv(x, d["w1"], B, h, w, cin, pl, 1, 1, 0, bn=d["bn1"], relu=1)
            y2, oh, ow = self._conv(y1, d["w2"], B, h, w, pl, pl, 3, s, 1, bn=d["bn2"], relu=1)
            idt = x
            if d["wd"] is not None:
                idt, _, _ = self._conv(x, d["wd"], B, h, w, cin, pl * 4, 1, s, 0, bn=d["bnd"])
            x, _, _ = self._conv(y2, d["w3"], B, oh, ow, pl, pl * 4, 1, 1, 0, bn=d["bn3"], residual=idt, relu=1)
            h, w, cin = oh, ow, pl * 4
            if d["tap"]:
                taps.append((x, h, w, cin))
        feats = []
        for (t, th, tw, tc), (rw, rb) in zip(taps, self.red):
            f, _, _ = self._conv(t, rw, B, th, tw, tc, rw.shape[1], 1, 1, 0, bias=rb)
            feats.append(f)
        return feats
