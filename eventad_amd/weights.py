"""Seeded random-init weights with the reference's state-dict key names.

No checkpoint exists offline (dagr_s_50.pth / best_rol.pth are external downloads), so
benchmarks and parity tests use random-init weights of the same architecture: the same
state dict is loaded into the product modules and handed to the CPU oracle.
"""
import math

import torch

LAYER_NAMES = ["conv_block1", "layer2", "layer3", "layer4", "layer5"]
LAYER_DIMS = [(19, 16), (82, 64), (130, 64), (130, 64), (130, 64)]     # net.py:73-97 (+2 pos channels)
RESNET_BLOCKS = [3, 4, 6, 3]
RESNET_PLANES = [64, 128, 256, 512]
FEATURE_TAPS = [(64, 16), (256, 64), (512, 64), (1024, 64), (2048, 64)]  # net_img.py:70-80


def _bn(sd, p, c, g):
    sd[p + "weight"] = 0.5 + torch.rand(c, generator=g)
    sd[p + "bias"] = 0.1 * torch.randn(c, generator=g)
    sd[p + "running_mean"] = 0.1 * torch.randn(c, generator=g)
    sd[p + "running_var"] = 0.5 + torch.rand(c, generator=g)
    sd[p + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def random_graph_state(seed=42, prefix="backbone."):
    """Spline-conv layers of Net (conv.py:10-72): weight[25,Cin,Cout], lin (root) [Cout,Cin],
    BatchNorm, skip Linear, skip BatchNorm."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (cin, cout) in zip(LAYER_NAMES, LAYER_DIMS):
        for blk, ci in (("conv_block1", cin), ("conv_block2", cout)):
            p = "%s%s.%s." % (prefix, name, blk)
            sd[p + "conv.weight"] = torch.randn(25, ci, cout, generator=g) / math.sqrt(4.0 * ci)
            sd[p + "conv.lin.weight"] = torch.randn(cout, ci, generator=g) / math.sqrt(ci)
            _bn(sd, p + "norm.module.", cout, g)
        p = "%s%s.conv_block2." % (prefix, name)
        sd[p + "lin.mlp.weight"] = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
        _bn(sd, p + "norm_skip.module.", cout, g)
    return sd


def random_cnn_state(seed=43, prefix="backbone.net."):
    """torchvision-named ResNet-50 (net.py:43) + the HookModule 1x1 reducers (net_img.py:70-90)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    m = prefix + "module."

    def conv(k, cout, cin, ks):
        sd[k] = torch.randn(cout, cin, ks, ks, generator=g) * math.sqrt(2.0 / (cin * ks * ks))

    conv(m + "conv1.weight", 64, 3, 7)
    _bn(sd, m + "bn1.", 64, g)
    inplanes = 64
    for li, (nb, planes) in enumerate(zip(RESNET_BLOCKS, RESNET_PLANES)):
        for bi in range(nb):
            p = "%slayer%d.%d." % (m, li + 1, bi)
            conv(p + "conv1.weight", planes, inplanes, 1)
            _bn(sd, p + "bn1.", planes, g)
            conv(p + "conv2.weight", planes, planes, 3)
            _bn(sd, p + "bn2.", planes, g)
            conv(p + "conv3.weight", planes * 4, planes, 1)
            _bn(sd, p + "bn3.", planes * 4, g)
            if bi == 0:
                conv(p + "downsample.0.weight", planes * 4, inplanes, 1)
                _bn(sd, p + "downsample.1.", planes * 4, g)
            inplanes = planes * 4
    sd[m + "fc.weight"] = torch.randn(1000, 2048, generator=g) / math.sqrt(2048)
    sd[m + "fc.bias"] = torch.zeros(1000)
    for i, (cin, cout) in enumerate(FEATURE_TAPS):
        sd["%sfeature_dconv.%d.weight" % (prefix, i)] = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
        sd["%sfeature_dconv.%d.bias" % (prefix, i)] = 0.1 * torch.randn(cout, generator=g)
    for i, cin in enumerate((1024, 2048)):
        sd["%soutput_dconv.%d.weight" % (prefix, i)] = torch.randn(256, cin, 1, 1, generator=g) / math.sqrt(cin)
        sd["%soutput_dconv.%d.bias" % (prefix, i)] = torch.zeros(256)
    return sd


HEAD_SHAPES = {
    "gru_net_event.gru.weight_ih_l0": (768, 64), "gru_net_event.gru.weight_hh_l0": (768, 256),
    "gru_net_event.gru.bias_ih_l0": (768,), "gru_net_event.gru.bias_hh_l0": (768,),
    "gru_net_event.gru.weight_ih_l1": (768, 256), "gru_net_event.gru.weight_hh_l1": (768, 256),
    "gru_net_event.gru.bias_ih_l1": (768,), "gru_net_event.gru.bias_hh_l1": (768,),
    "gru_net_cor.gru.weight_ih_l0": (96, 4), "gru_net_cor.gru.weight_hh_l0": (96, 32),
    "gru_net_cor.gru.bias_ih_l0": (96,), "gru_net_cor.gru.bias_hh_l0": (96,),
    "fusion_module.event_proj.weight": (256, 256), "fusion_module.event_proj.bias": (256,),
    "fusion_module.coord_proj.weight": (256, 32), "fusion_module.coord_proj.bias": (256,),
    "fusion_module.attention.in_proj_weight": (768, 256), "fusion_module.attention.in_proj_bias": (768,),
    "fusion_module.attention.out_proj.weight": (256, 256), "fusion_module.attention.out_proj.bias": (256,),
    "fusion_module.fusion.0.weight": (256, 512), "fusion_module.fusion.0.bias": (256,),
    "fusion_module.fusion.3.weight": (2, 256), "fusion_module.fusion.3.bias": (2,),
    "soft_attention.weight": (256, 1), "soft_attention_cor.weight": (32, 1),
}
