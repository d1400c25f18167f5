"""A plain-torch ResNet-50 (v1.5 layout, torchvision's parameter names) with a forward pass, for fixtures that need a
trunk under the reference's HookModule: torchvision is not installed, and the product's `_ResNet50` only holds
parameters (its arithmetic runs in csrc/cnn.hip)."""
import torch
import torch.nn.functional as F

from eventad_amd.dagr.model.networks.net_img import _ResNet50


class TorchResNet50(_ResNet50):
    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for li in range(4):
            x = getattr(self, "layer%d" % (li + 1))(x)
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.fc(x)
