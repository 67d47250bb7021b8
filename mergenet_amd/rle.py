"""COCO run-length encoding from the change points of the column-major label scan.

The reference stores every instance as ``maskUtils.encode(np.asfortranarray(mask == i))``
(egs/cityscape/local/segment.py:165-186; pycocotools is a third-party dependency of the
reference, listed in its requirements.txt without a pinned version and absent from this image).
Its published format is restated here: the counts are the lengths of alternating 0/1 runs of
the Fortran-ordered binary mask, starting with a run of zeros (possibly empty); the string packs
each count -- from the third on as the difference to the count two places earlier -- in 5-bit
groups, low group first, bit 0x20 = "more groups follow", bit 0x10 of the last group = sign,
offset by 48 into printable ASCII.
"""

from __future__ import annotations

from typing import List

import numpy as np


def counts_to_string(counts) -> bytes:
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def string_to_counts(s: bytes) -> List[int]:
    counts: List[int] = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            ch = s[p] - 48
            x |= (ch & 0x1F) << (5 * k)
            more = bool(ch & 0x20)
            p += 1
            k += 1
            if not more and (ch & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def binary_mask_counts(b: np.ndarray) -> List[int]:
    """Run lengths of a binary [H,W] mask in Fortran order, first run = zeros (host restatement,
    used by the tests as the checker of the device path)."""
    flat = np.asarray(b, dtype=np.uint8).reshape(-1, order="F")
    if flat.size == 0:
        return [0]
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    if flat[0] == 1:
        counts = [0] + counts
    return counts


def decode(counts, H: int, W: int) -> np.ndarray:
    flat = np.zeros(H * W, np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape((H, W), order="F")


def from_change_points(pos, prev, cur, H: int, W: int, num_instances: int):
    """Group the change points (scan position, label before, label at) per instance."""
    N = H * W
    pos = np.asarray(pos, np.int64)
    prev = np.asarray(prev, np.int64)
    cur = np.asarray(cur, np.int64)
    results = []
    # a change point (j, a, b) ends a run of label a and starts a run of label b at j
    starts_lab = cur
    ends_lab = prev
    order_s = np.argsort(starts_lab, kind="stable")
    order_e = np.argsort(ends_lab, kind="stable")
    s_lab, s_pos = starts_lab[order_s], pos[order_s]
    e_lab, e_pos = ends_lab[order_e], pos[order_e]
    for k in range(1, num_instances + 1):
        s0, s1 = np.searchsorted(s_lab, k, "left"), np.searchsorted(s_lab, k, "right")
        e0, e1 = np.searchsorted(e_lab, k, "left"), np.searchsorted(e_lab, k, "right")
        starts = s_pos[s0:s1]
        ends = e_pos[e0:e1]
        if starts.size > ends.size:          # the last run reaches the end of the scan
            ends = np.concatenate([ends, [N]])
        counts = []
        last = 0
        for a, b in zip(starts.tolist(), ends.tolist()):
            counts.append(a - last)
            counts.append(b - a)
            last = b
        if last < N or not counts:
            counts.append(N - last)
        results.append({"size": [H, W], "counts": counts_to_string(counts)})
    return results
