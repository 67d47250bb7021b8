"""BASELINE.json configs[3]: a PSPNet-ResNet50-shaped network in stock PyTorch-ROCm producing
class + log-spiral-offset maps that go straight into the HIP merger on the same GPU.

The reference's own models need torchvision and downloaded weights (models/pspnet.py:100-108,
models/resnet.py:21-24), neither available offline, so this is our own definition of the same
shape with random weights: a plumbing / throughput configuration, not a model-parity claim.
The hand-off is device resident: sigmoid outputs (utils/inference_utils.py:44,96) -> Merger with
the binding's clip fused into the loads; no .npy files (utils/inference_utils.py:122-126), no host
copy.
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride=1, dilation=1):
        super().__init__()
        cout = mid * 4
        self.c1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.b1 = nn.BatchNorm2d(mid)
        self.c2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.b2 = nn.BatchNorm2d(mid)
        self.c3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.b3 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                      nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.b1(self.c1(x)))
        y = F.relu(self.b2(self.c2(y)))
        y = self.b3(self.c3(y))
        return F.relu(y + (x if self.down is None else self.down(x)))


def _stage(cin, mid, blocks, stride, dilation):
    layers = [Bottleneck(cin, mid, stride=stride, dilation=dilation)]
    layers += [Bottleneck(mid * 4, mid, dilation=dilation) for _ in range(blocks - 1)]
    return nn.Sequential(*layers)


class PSPNetResNet50(nn.Module):
    """ResNet-50 (3-4-6-3 bottlenecks, output stride 8 by dilation) + pyramid pooling (1,2,3,6)."""

    def __init__(self, num_classes: int, num_offsets: int, width: int = 64):
        super().__init__()
        w = width
        self.stem = nn.Sequential(nn.Conv2d(3, w, 7, stride=2, padding=3, bias=False),
                                  nn.BatchNorm2d(w), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(3, stride=2, padding=1))
        self.l1 = _stage(w, w, 3, 1, 1)
        self.l2 = _stage(w * 4, w * 2, 4, 2, 1)
        self.l3 = _stage(w * 8, w * 4, 6, 1, 2)
        self.l4 = _stage(w * 16, w * 8, 3, 1, 4)
        feat = w * 32
        self.bins = (1, 2, 3, 6)
        self.ppm = nn.ModuleList([nn.Sequential(nn.Conv2d(feat, feat // 4, 1, bias=False),
                                                nn.BatchNorm2d(feat // 4), nn.ReLU(inplace=True))
                                  for _ in self.bins])
        self.head = nn.Sequential(nn.Conv2d(feat * 2, w * 8, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(w * 8), nn.ReLU(inplace=True),
                                  nn.Conv2d(w * 8, num_classes + num_offsets, 1))
        self.num_classes = num_classes

    def forward(self, x):
        size = x.shape[-2:]
        y = self.l4(self.l3(self.l2(self.l1(self.stem(x)))))
        hw = y.shape[-2:]
        pyramid = [y]
        for b, m in zip(self.bins, self.ppm):
            pyramid.append(F.interpolate(m(F.adaptive_avg_pool2d(y, b)), size=hw, mode="bilinear",
                                         align_corners=False))
        y = self.head(torch.cat(pyramid, 1))
        return F.interpolate(y, size=size, mode="bilinear", align_corners=False)


@torch.no_grad()
def segment_image(model: PSPNetResNet50, image, offsets, merger, opts):
    """image [1,3,H,W] on the GPU -> (mask, class_table, stats); everything stays on the device."""
    logits = model(image)[0]
    probs = torch.sigmoid(logits).float().contiguous()          # inference_utils.py:44,96
    C = model.num_classes
    return merger.segment(probs[:C].contiguous(), probs[C:].contiguous(), offsets, opts)


if __name__ == "__main__":
    import sys, time
    sys.path.insert(0, ".")
    from mergenet_amd import synth, segmenter as seg
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
    offs = synth.generate_offsets(40, 10)
    torch.manual_seed(0)
    model = PSPNetResNet50(9, len(offs)).cuda().eval()
    img = torch.rand(1, 3, H, W, device="cuda")
    merger = seg.Merger(H, W, 9, len(offs))
    opts = seg.default_options(clip_inputs=1, mode=seg.MN_MODE_ROUNDS)
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        mask, table, _, st = segment_image(model, img, offs, merger, opts)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("PSPNet-ResNet50 forward + merge %dx%d: %.1f ms (merger %.1f ms), %d instances"
          % (H, W, dt * 1e3, st["ms_total"], st["num_instances"]))
