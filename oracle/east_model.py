"""CPU fp32 restatement of the reference EAST network (TEST INFRASTRUCTURE).

Follows /root/reference/src/manuscript/detectors/_east/east.py:
  DecoderBlock                 east.py:13-30
  ResNetFeatureExtractor       east.py:33-67   (torchvision resnet50 taps layer1..4)
  FeatureMergingBranchResNet   east.py:70-93
  OutputHead                   east.py:96-105
  EAST.forward                 east.py:135-139

The backbone itself is torchvision's ResNet-50 (v1.5: stride on the 3x3 of
each Bottleneck), a third-party dependency absent from this image; it is
restated here from the published definition, keeping torchvision's module
names so the reference's state_dict keys
(`backbone.extractor.layer2.0.downsample.1.running_var`, ...) load unchanged.
Backbone parity is therefore UNPINNED; decoder + head are pinned by
tests/golden/east_decoder_head.npz (generated from the reference file).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet50Trunk(nn.Module):
    """conv1/bn1/maxpool/layer1..4 of ResNet-50 ([3,4,6,3] Bottlenecks)."""

    def __init__(self, layers=(3, 4, 6, 3)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4),
            )
        blks = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            blks.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*blks)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        r1 = self.layer1(x)
        r2 = self.layer2(r1)
        r3 = self.layer3(r2)
        r4 = self.layer4(r3)
        return {"res1": r1, "res2": r2, "res3": r3, "res4": r4}


class ResNetFeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        self.extractor = ResNet50Trunk()

    def forward(self, x):
        return self.extractor(x)


class DecoderBlock(nn.Module):  # east.py:13-30
    def __init__(self, cin, cmid, cout):
        super().__init__()
        self.conv1x1 = nn.Sequential(nn.Conv2d(cin, cmid, 1), nn.BatchNorm2d(cmid), nn.ReLU(inplace=True))
        self.conv3x3 = nn.Sequential(nn.Conv2d(cmid, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv3x3(self.conv1x1(x))


class FeatureMergingBranchResNet(nn.Module):  # east.py:70-93
    def __init__(self):
        super().__init__()
        self.block1 = DecoderBlock(2048, 512, 512)
        self.block2 = DecoderBlock(512 + 1024, 256, 256)
        self.block3 = DecoderBlock(256 + 512, 128, 128)
        self.block4 = DecoderBlock(128 + 256, 64, 32)

    def forward(self, feats):
        f1, f2, f3, f4 = feats["res1"], feats["res2"], feats["res3"], feats["res4"]
        up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
        h4 = self.block1(f4)
        h3 = self.block2(torch.cat([up(h4), f3], dim=1))
        h2 = self.block3(torch.cat([up(h3), f2], dim=1))
        h1 = self.block4(torch.cat([up(h2), f1], dim=1))
        return h1


class OutputHead(nn.Module):  # east.py:96-105
    def __init__(self):
        super().__init__()
        self.score_map = nn.Conv2d(32, 1, 1)
        self.geo_map = nn.Conv2d(32, 8, 1)

    def forward(self, x):
        return torch.sigmoid(self.score_map(x)), self.geo_map(x)


class EASTNet(nn.Module):  # east.py:108-139
    def __init__(self):
        super().__init__()
        self.backbone = ResNetFeatureExtractor()
        self.decoder = FeatureMergingBranchResNet()
        self.output_head = OutputHead()

    def forward(self, x):
        feats = self.backbone(x)
        score, geometry = self.output_head(self.decoder(feats))
        return {"score": score, "geometry": geometry}
