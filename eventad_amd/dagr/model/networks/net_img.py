"""CNN feature branch: mirror of `dagr.model.networks.net_img.HookModule` over torchvision's
resnet50 (reference src/dagr/model/networks/net_img.py:42-134, net.py:41-49).

Parameters keep torchvision's key names under `module.` plus `feature_dconv.{0..4}` /
`output_dconv.{0,1}` so reference checkpoints load.  Only what EventAD consumes is computed:
the five feature taps (raw conv1 output, layer1..4) through their 1x1 reducers; avgpool/fc and
`output_dconv` (detection-head inputs, net_img.py:131-133) are skipped.
ROUND-1 NOTE: the trunk's convolutions run through torch's GPU conv ops (MIOpen) in NHWC; the
hand-written implicit-GEMM kernels for it are the next step (DESIGN.md 'next').  The bench
configuration named by BASELINE.json (C2) uses precomputed CNN features and does not run it."""
import torch
import torch.nn as nn
import torch.nn.functional as F

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

    @torch.no_grad()
    def forward(self, image):
        """image f32[B,3,H,W] in [0,1] -> five NHWC feature maps [B,h,w,C] (channels_last views)."""
        m = self.module
        x = image.contiguous(memory_format=torch.channels_last)
        c1 = m.conv1(x)
        taps = [c1]
        x = F.max_pool2d(F.relu(m.bn1(c1)), 3, 2, 1)
        for li in range(4):
            x = getattr(m, "layer%d" % (li + 1))(x)
            taps.append(x)
        feats = [d(t) for d, t in zip(self.feature_dconv, taps)]
        return [f.permute(0, 2, 3, 1).contiguous() for f in feats]
