"""oracle/resnet_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the CNN feature branch:
HookModule.forward (src/dagr/model/networks/net_img.py:122-134) over
torchvision 0.16 `resnet50` (net.py:41-49; torchvision is absent here, so the
published ResNet-50 v1.5 architecture is restated: 7x7/2 stem, 3x3/2 max-pool,
bottleneck blocks [3,4,6,3] with the stride on the 3x3 conv, 1x1 downsample).

Hooks (net.py:46-47): features = raw `conv1` output (before bn1/relu), layer1..layer4;
each goes through its 1x1 `feature_dconv` (with bias).  The unused outputs
(`output_dconv` on layer3/4, avgpool, fc) are not computed -- EventAD never reads them.
State-dict keys follow torchvision's names under `backbone.net.module.`.
PARITY UNPINNED w.r.t. reference outputs: torchvision is not installed here and the reference holds no fixtures for
the trunk; the restated architecture is checked against shape/known-answer tests only (tests/test_oracle_ops.py).
"""
import torch.nn.functional as F

BLOCKS = [3, 4, 6, 3]
PLANES = [64, 128, 256, 512]
BN_EPS = 1e-5


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"],
                        sd[p + "bias"], False, 0.1, BN_EPS)


def _bottleneck(x, sd, p, stride, has_down):
    idt = x
    y = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    y = F.relu(_bn(F.conv2d(y, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2."))
    y = _bn(F.conv2d(y, sd[p + "conv3.weight"]), sd, p + "bn3.")
    if has_down:
        idt = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.")
    return F.relu(y + idt)


def hook_module_forward(sd, image, prefix="backbone.net."):
    """image f32[B,3,H,W] in [0,1] -> list of 5 maps [B,16,H/2,W/2],[B,64,H/4,W/4],...,[B,64,H/32,W/32]."""
    m = prefix + "module."
    feats = []
    c1 = F.conv2d(image, sd[m + "conv1.weight"], stride=2, padding=3)
    feats.append(c1)                                            # hook on `conv1` = raw conv output
    x = F.relu(_bn(c1, sd, m + "bn1."))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li in range(4):
        for bi in range(BLOCKS[li]):
            stride = 2 if (bi == 0 and li > 0) else 1
            x = _bottleneck(x, sd, "%slayer%d.%d." % (m, li + 1, bi), stride, bi == 0)
        feats.append(x)                                         # hooks on layer1..layer4
    out = []
    for i, f in enumerate(feats):                               # net_img.py:127-129
        out.append(F.conv2d(f, sd["%sfeature_dconv.%d.weight" % (prefix, i)],
                            sd["%sfeature_dconv.%d.bias" % (prefix, i)]))
    return out
