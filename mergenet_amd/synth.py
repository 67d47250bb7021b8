"""Synthetic class/sameness maps (generator ``synth-v1``) and the log-spiral offset list.

Host-side input tooling for tests and ``bench.py``; nothing here is on the timed path.

* :func:`generate_offsets` restates the offset list contract of the reference
  (``utils/train_utils.py:317-328``): 100-degree steps on a geometric spiral whose
  last radius is ``max_offset``.  Offsets are ``(d_row, d_col)`` as consumed by the
  merger (``utils/segmenter.py:279-282``, ``utils/csegment/segment.cc:215-218``).
* :func:`synth_v1` builds piecewise-constant instance layouts (ellipses, painter's
  order, id 0 = background) and turns them into per-pixel class probabilities and
  per-offset "same instance" probabilities in the format the merger takes
  (``class[C,H,W]``, ``same[O,H,W]``, float32).  The sameness target definition
  (``inst(p + o) == inst(p)``) follows ``utils/dataset.py:259-277``.

All randomness is counter based (splitmix64 of ``(seed, stream, index)``), so any
host or device implementation can reproduce the same bits.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

_M64 = (1 << 64) - 1


def generate_offsets(max_offset: int = 20, num_offsets: int = 10) -> List[Tuple[int, int]]:
    """Log-spiral offsets; same numbers as ``utils/train_utils.py:317-328``."""
    step = math.pi * 5.0 / 9.0
    last = (num_offsets - 1) * step
    reach = max(abs(math.cos(last)), abs(math.sin(last)))
    growth = math.pow(abs(max_offset / reach), 1.0 / float(num_offsets - 1))
    out = []
    for n in range(num_offsets):
        radius = math.pow(growth, n)
        out.append((int(round(math.cos(n * step) * radius)),
                    int(round(math.sin(n * step) * radius))))
    return out


def validate_offsets(offsets: Sequence[Tuple[int, int]]) -> None:
    """Reject lists holding both ``o`` and ``-o`` (``utils/core_config.py:66-73``) or (0, 0)."""
    seen = set()
    for (i, j) in offsets:
        if (i, j) == (0, 0):
            raise ValueError("offset (0, 0) is not a valid offset")
        if (-i, -j) in seen or (i, j) in seen:
            raise ValueError("offset list holds an offset twice or together with its negation: %r"
                             % ((i, j),))
        seen.add((i, j))


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
        x = x ^ (x >> np.uint64(31))
    return x


def uniform01(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
    """``n`` float32 values in [0, 1) for counters ``start .. start+n-1`` of ``(seed, stream)``.

    value = (splitmix64(splitmix64(seed * 2^32 + stream) + counter) >> 40) * 2^-24
    """
    key = _splitmix64(np.array([((seed << 32) + stream) & _M64], dtype=np.uint64))[0]
    ctr = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64((ctr + key).astype(np.uint64))
    return ((bits >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


@dataclass
class SynthImage:
    class_probs: np.ndarray          # float32 [C, H, W]
    sameness_probs: np.ndarray       # float32 [O, H, W]
    instances: np.ndarray            # int32 [H, W], 0 = background
    instance_class: List[int]        # class of instance id i (index 0 = background = class 0)
    offsets: List[Tuple[int, int]]
    seed: int
    params: dict = field(default_factory=dict)


def _layout(H: int, W: int, num_classes: int, seed: int, max_reach: int,
            num_instances: int | None, occlusion: bool):
    N = H * W
    K = num_instances if num_instances is not None else max(4, N // 87381)
    u = uniform01(seed, 0xA11CE, 8 * K + 64).astype(np.float64)
    inst = np.zeros((H, W), dtype=np.int32)
    rows = np.arange(H, dtype=np.float64)[:, None]
    cols = np.arange(W, dtype=np.float64)[None, :]
    boxes = []
    for k in range(K):
        cy = u[8 * k + 0] * H
        cx = u[8 * k + 1] * W
        ry = (0.02 + 0.10 * u[8 * k + 2]) * H
        rx = (0.02 + 0.10 * u[8 * k + 3]) * W
        ry = max(ry, 2.0)
        rx = max(rx, 2.0)
        inside = ((rows - cy) / ry) ** 2 + ((cols - cx) / rx) ** 2 <= 1.0
        inst[inside] = k + 1
        boxes.append((cy - ry, cy + ry, cx - rx, cx + rx))
    groups = list(range(K + 1))   # instance id -> identity used by the sameness target
    if occlusion:
        # two "occluded" instances: a bar painted over the middle of an ellipse splits it into
        # two parts that keep ONE identity; only the long offsets connect the parts
        # (reference README.md:17).  The bar is a separate instance.
        for t in range(2):
            k = t  # split the first two ellipses
            y0, y1, x0, x1 = boxes[k]
            cx = 0.5 * (x0 + x1)
            half = max(2.0, 0.04 * (x1 - x0))
            bar = (np.abs(cols - cx) <= half) & (rows >= y0 - 2) & (rows <= y1 + 2)
            bar = bar & (inst == k + 1)
            K += 1
            inst[bar] = K
            groups.append(K)
            boxes.append((y0 - 2, y1 + 2, cx - half, cx + half))
    # classes: uniform in 1..C-1, neighbours (boxes closer than the longest offset) differ
    cls = [0]
    uc = uniform01(seed, 0xC1A55, K + 8)
    for k in range(K):
        want = 1 + int(uc[k] * (num_classes - 1)) % max(1, num_classes - 1)
        taken = set()
        for j in range(k):
            a, b = boxes[k], boxes[j]
            if (a[0] - max_reach <= b[1] and b[0] - max_reach <= a[1] and
                    a[2] - max_reach <= b[3] and b[2] - max_reach <= a[3]):
                taken.add(cls[j + 1])
        c = want
        for _ in range(num_classes):
            if c not in taken:
                break
            c = 1 + (c % (num_classes - 1)) if num_classes > 2 else 1
        cls.append(c)
    return inst, cls


def synth_v1(H: int, W: int, num_classes: int, offsets: Sequence[Tuple[int, int]], seed: int,
             noise: float = 0.15, num_instances: int | None = None,
             occlusion: bool = False) -> SynthImage:
    """Well-conditioned synthetic maps (generator ``synth-v1``).

    class[c,p] = clip((c == cls(p) ? 0.9 : 0.05) + 0.3 * U(-noise, noise), 0.01, 0.99)
    same[k,p]  = clip((inst(p+o_k) == inst(p) ? 0.9 : 0.1) + U(-noise, noise), 0.01, 0.99)
    (1.0 where p+o_k is outside the image; the merger never reads those).
    """
    offsets = [(int(i), int(j)) for (i, j) in offsets]
    validate_offsets(offsets)
    reach = max(max(abs(i), abs(j)) for (i, j) in offsets)
    inst, cls = _layout(H, W, num_classes, seed, reach, num_instances, occlusion)
    N = H * W
    pix_cls = np.asarray(cls, dtype=np.int32)[inst]
    class_probs = np.empty((num_classes, H, W), dtype=np.float32)
    for c in range(num_classes):
        base = np.where(pix_cls == c, np.float32(0.9), np.float32(0.05)).astype(np.float32)
        un = uniform01(seed, 0x100 + c, N).reshape(H, W)
        jitter = (np.float32(2.0) * un - np.float32(1.0)) * np.float32(noise) * np.float32(0.3)
        class_probs[c] = np.clip(base + jitter, np.float32(0.01), np.float32(0.99))
    same = np.empty((len(offsets), H, W), dtype=np.float32)
    for k, (di, dj) in enumerate(offsets):
        tgt = np.zeros((H, W), dtype=bool)
        valid = np.zeros((H, W), dtype=bool)
        r0, r1 = max(0, -di), min(H, H - di)
        c0, c1 = max(0, -dj), min(W, W - dj)
        if r0 < r1 and c0 < c1:
            tgt[r0:r1, c0:c1] = inst[r0:r1, c0:c1] == inst[r0 + di:r1 + di, c0 + dj:c1 + dj]
            valid[r0:r1, c0:c1] = True
        base = np.where(tgt, np.float32(0.9), np.float32(0.1)).astype(np.float32)
        un = uniform01(seed, 0x200 + k, N).reshape(H, W)
        jitter = (np.float32(2.0) * un - np.float32(1.0)) * np.float32(noise)
        plane = np.clip(base + jitter, np.float32(0.01), np.float32(0.99))
        plane[~valid] = np.float32(1.0)
        same[k] = plane
    return SynthImage(class_probs, same, inst, cls, list(offsets), seed,
                      dict(H=H, W=W, C=num_classes, noise=noise, occlusion=occlusion))


def adversarial(H: int, W: int, num_classes: int, offsets: Sequence[Tuple[int, int]],
                seed: int) -> SynthImage:
    """Uniform(0.02, 0.98) everything: order-dependent, for exact-mode tests only."""
    offsets = [(int(i), int(j)) for (i, j) in offsets]
    validate_offsets(offsets)
    N = H * W
    cp = np.empty((num_classes, H, W), dtype=np.float32)
    for c in range(num_classes):
        cp[c] = (np.float32(0.02) + np.float32(0.96) * uniform01(seed, 0x300 + c, N)).reshape(H, W)
    sp = np.empty((len(offsets), H, W), dtype=np.float32)
    for k in range(len(offsets)):
        sp[k] = (np.float32(0.02) + np.float32(0.96) * uniform01(seed, 0x400 + k, N)).reshape(H, W)
    return SynthImage(cp, sp, np.zeros((H, W), np.int32), [0], list(offsets), seed,
                      dict(H=H, W=W, C=num_classes, adversarial=True))


def count_edges(H: int, W: int, offsets: Sequence[Tuple[int, int]]) -> int:
    """Number of in-bounds (pixel, offset) pairs = initial adjacency records."""
    return sum(max(0, H - abs(i)) * max(0, W - abs(j)) for (i, j) in offsets)


def _box_blur(a: np.ndarray, r: int) -> np.ndarray:
    """Mean over a (2r+1)^2 window, edges replicated; ``a`` is [K, H, W]."""
    if r == 0:
        return a
    p = np.pad(a, ((0, 0), (r, r), (r, r)), mode="edge").astype(np.float64)
    cs = p.cumsum(1).cumsum(2)
    cs = np.pad(cs, ((0, 0), (1, 0), (1, 0)))
    k = 2 * r + 1
    s = cs[:, k:, k:] - cs[:, :-k, k:] - cs[:, k:, :-k] + cs[:, :-k, :-k]
    return (s / (k * k)).astype(np.float32)


def blurred_v1(H: int, W: int, num_classes: int, offsets: Sequence[Tuple[int, int]], seed: int,
               radius: int = 2, noise: float = 0.05,
               num_instances: int | None = None) -> SynthImage:
    """Network-like maps: the noise-free synth-v1 maps box-blurred (certainty fades towards the
    instance boundaries and passes through 0.5 next to them), then the synth-v1 noise added.
    Such maps are NOT sign-separable; the reference's result on them depends on its sequential
    order."""
    clean = synth_v1(H, W, num_classes, offsets, seed, noise=0.0, num_instances=num_instances)
    noisy = synth_v1(H, W, num_classes, offsets, seed, noise=noise, num_instances=num_instances)
    cp = _box_blur(clean.class_probs, radius) + (noisy.class_probs - clean.class_probs)
    sp = _box_blur(clean.sameness_probs, radius) + (noisy.sameness_probs - clean.sameness_probs)
    cp = np.clip(cp, 0.01, 0.99).astype(np.float32)
    sp = np.clip(sp, 0.01, 0.99).astype(np.float32)
    return SynthImage(cp, sp, clean.instances, clean.instance_class, list(clean.offsets), seed,
                      dict(H=H, W=W, C=num_classes, noise=noise, radius=radius, blurred=True))


def checkerboard(H: int, W: int, num_classes: int, offsets: Sequence[Tuple[int, int]],
                 cell_px: int, seed: int) -> SynthImage:
    """Cells of ``cell_px`` pixels with a random class each: hundreds of small components whose
    further merging is driven by ``merge_logprob_bias`` alone (bias-dominated second phase)."""
    offsets = [(int(i), int(j)) for (i, j) in offsets]
    validate_offsets(offsets)
    rng = np.random.default_rng(seed)
    cell = (np.arange(H)[:, None] // cell_px) * ((W + cell_px - 1) // cell_px) + \
        (np.arange(W)[None, :] // cell_px)
    cmap = rng.integers(0, num_classes, size=cell.max() + 1)[cell]
    cp = np.full((num_classes, H, W), 0.05, np.float32)
    for c in range(num_classes):
        cp[c][cmap == c] = 0.9
    cp += rng.uniform(0, 0.01, size=cp.shape).astype(np.float32)
    sp = np.zeros((len(offsets), H, W), np.float32)
    for k, (di, dj) in enumerate(offsets):
        other = np.full((H, W), -1)
        r0, r1 = max(0, -di), min(H, H - di)
        c0, c1 = max(0, -dj), min(W, W - dj)
        other[r0:r1, c0:c1] = cell[r0 + di:r1 + di, c0 + dj:c1 + dj]
        sp[k] = np.where(other == cell, 0.9, 0.1) + rng.uniform(-0.05, 0.05, size=(H, W))
    return SynthImage(cp.astype(np.float32), sp.astype(np.float32), cell.astype(np.int32), [0],
                      list(offsets), seed, dict(H=H, W=W, C=num_classes, checkerboard=cell_px))
