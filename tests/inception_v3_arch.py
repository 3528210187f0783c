"""Inception-v3, ARCHITECTURE ONLY, in plain torch - test infrastructure, not product code.

The reference's CNN_ENCODER (util.py:263-306) wraps `torchvision.models.inception_v3()` with downloaded weights
(`inception_v3_google-1a9a5a14.pth`); neither torchvision nor the weights exist in this image, and its arithmetic lives in a
third-party package, not under /root/reference (parity UNPINNED for the trunk).  This file restates the published
topology (Szegedy et al., "Rethinking the Inception Architecture", the block names and channel counts the reference's
`define_module` copies, util.py:282-298) with seeded random weights, so that `tgsr_amd.util.CNN_ENCODER(nef, inception=...)`
can execute its real walk - 299 x 299 bilinear resize, sixteen blocks, 17 x 17 x 768 region features, 8 x 8 average pool, HIP
heads - on the GPU at the configured batch.  It says nothing about the trained network's outputs.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvBnRelu(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


def _avg3(x):
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1)


class BlockA(nn.Module):                      # 35 x 35: 1x1 | 1x1-5x5 | 1x1-3x3-3x3 | pool-1x1
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = ConvBnRelu(cin, 64, kernel_size=1)
        self.branch5x5_1 = ConvBnRelu(cin, 48, kernel_size=1)
        self.branch5x5_2 = ConvBnRelu(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = ConvBnRelu(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = ConvBnRelu(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = ConvBnRelu(96, 96, kernel_size=3, padding=1)
        self.branch_pool = ConvBnRelu(cin, pool_features, kernel_size=1)

    def forward(self, x):
        return torch.cat((self.branch1x1(x), self.branch5x5_2(self.branch5x5_1(x)),
                          self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))), self.branch_pool(_avg3(x))), 1)


class BlockB(nn.Module):                      # 35 -> 17: 3x3 s2 | 1x1-3x3-3x3 s2 | max pool
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = ConvBnRelu(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = ConvBnRelu(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = ConvBnRelu(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = ConvBnRelu(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        return torch.cat((self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          F.max_pool2d(x, kernel_size=3, stride=2)), 1)


class BlockC(nn.Module):                      # 17 x 17: factorised 7x7
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = ConvBnRelu(cin, 192, kernel_size=1)
        self.branch7x7_1 = ConvBnRelu(cin, c7, kernel_size=1)
        self.branch7x7_2 = ConvBnRelu(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = ConvBnRelu(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = ConvBnRelu(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = ConvBnRelu(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = ConvBnRelu(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = ConvBnRelu(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = ConvBnRelu(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = ConvBnRelu(cin, 192, kernel_size=1)

    def forward(self, x):
        d = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        return torch.cat((self.branch1x1(x), self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x))), d, self.branch_pool(_avg3(x))), 1)


class BlockD(nn.Module):                      # 17 -> 8
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = ConvBnRelu(cin, 192, kernel_size=1)
        self.branch3x3_2 = ConvBnRelu(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = ConvBnRelu(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = ConvBnRelu(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = ConvBnRelu(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = ConvBnRelu(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        return torch.cat((self.branch3x3_2(self.branch3x3_1(x)),
                          self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x)))),
                          F.max_pool2d(x, kernel_size=3, stride=2)), 1)


class BlockE(nn.Module):                      # 8 x 8: expanded filter bank
    def __init__(self, cin):
        super().__init__()
        self.branch1x1 = ConvBnRelu(cin, 320, kernel_size=1)
        self.branch3x3_1 = ConvBnRelu(cin, 384, kernel_size=1)
        self.branch3x3_2a = ConvBnRelu(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = ConvBnRelu(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = ConvBnRelu(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = ConvBnRelu(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = ConvBnRelu(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = ConvBnRelu(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = ConvBnRelu(cin, 192, kernel_size=1)

    def forward(self, x):
        a = self.branch3x3_1(x)
        b = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        return torch.cat((self.branch1x1(x), self.branch3x3_2a(a), self.branch3x3_2b(a), self.branch3x3dbl_3a(b),
                          self.branch3x3dbl_3b(b), self.branch_pool(_avg3(x))), 1)


class InceptionV3Arch(nn.Module):
    """The sixteen blocks CNN_ENCODER.define_module copies (util.py:282-298), under torchvision's attribute names."""

    def __init__(self, seed=0):
        super().__init__()
        self.Conv2d_1a_3x3 = ConvBnRelu(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = ConvBnRelu(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = ConvBnRelu(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = ConvBnRelu(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = ConvBnRelu(80, 192, kernel_size=3)
        self.Mixed_5b = BlockA(192, 32)
        self.Mixed_5c = BlockA(256, 64)
        self.Mixed_5d = BlockA(288, 64)
        self.Mixed_6a = BlockB(288)
        self.Mixed_6b = BlockC(768, 128)
        self.Mixed_6c = BlockC(768, 160)
        self.Mixed_6d = BlockC(768, 160)
        self.Mixed_6e = BlockC(768, 192)
        self.Mixed_7a = BlockD(768)
        self.Mixed_7b = BlockE(1280)
        self.Mixed_7c = BlockE(2048)
        g = torch.Generator().manual_seed(seed)
        for m in self.modules():                 # variance-preserving weights: activations stay O(1) through 48 layers
            if isinstance(m, nn.Conv2d):
                fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                with torch.no_grad():
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
            elif isinstance(m, nn.BatchNorm2d):
                with torch.no_grad():
                    m.running_mean.copy_(0.05 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.copy_(0.75 + 0.5 * torch.rand(m.running_var.shape, generator=g))
