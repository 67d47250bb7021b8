"""Comparing label maps up to a permutation of the labels (plain numpy helpers).

Instance ids have no canonical numbering: the reference numbers instances in hash-map iteration
order (``utils/csegment/segment.cc:503``), this library in ascending surviving pixel id.  Results
are therefore compared as partitions plus per-instance class.
"""

from __future__ import annotations

import numpy as np


def canonical(labels: np.ndarray) -> np.ndarray:
    """Relabel by first occurrence in row-major order (0, 1, 2, ...)."""
    flat = np.asarray(labels).reshape(-1)
    _, first, inv = np.unique(flat, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))
    return order[inv].reshape(np.asarray(labels).shape).astype(np.int64)


def same_partition(a: np.ndarray, b: np.ndarray) -> bool:
    return bool(np.array_equal(canonical(a), canonical(b)))


def partition_mismatch(a: np.ndarray, b: np.ndarray) -> int:
    """Number of pixels whose part in ``a`` is not matched 1:1 to a part of ``b``."""
    ca, cb = canonical(a).reshape(-1), canonical(b).reshape(-1)
    pairs, counts = np.unique(np.stack([ca, cb], 1), axis=0, return_counts=True)
    best_a = {}
    for (x, y), n in zip(pairs, counts):
        if n > best_a.get(x, (None, 0))[1]:
            best_a[x] = (y, n)
    used = {}
    good = 0
    for x, (y, n) in best_a.items():
        if y in used:
            continue
        size_a = int((ca == x).sum())
        size_b = int((cb == y).sum())
        if size_a == n and size_b == n:
            good += n
            used[y] = x
    return int(ca.size - good)


def masks_equivalent(mask_a, classes_a, mask_b, classes_b) -> bool:
    """Instance masks equal up to a permutation of labels 1..K, with equal per-label class."""
    ma, mb = np.asarray(mask_a), np.asarray(mask_b)
    if ma.shape != mb.shape or len(classes_a) != len(classes_b):
        return False
    if not np.array_equal(ma == 0, mb == 0):
        return False
    if not same_partition(ma, mb):
        return False
    flat_a, flat_b = ma.reshape(-1), mb.reshape(-1)
    _, idx = np.unique(flat_a, return_index=True)
    for i in idx:
        la, lb = int(flat_a[i]), int(flat_b[i])
        if la == 0:
            continue
        if classes_a[la - 1] != classes_b[lb - 1]:
            return False
    return True


def agreement(mask_a, mask_b) -> int:
    """Pixels on which two label maps agree under the best one-to-one matching of their labels by
    overlap (greedy on the contingency table, largest overlap first)."""
    a = np.asarray(mask_a).astype(np.int64).reshape(-1)
    b = np.asarray(mask_b).astype(np.int64).reshape(-1)
    cont = np.zeros((int(a.max()) + 1, int(b.max()) + 1), np.int64)
    np.add.at(cont, (a, b), 1)
    agree = 0
    for _ in range(min(cont.shape)):
        i, j = np.unravel_index(int(np.argmax(cont)), cont.shape)
        if cont[i, j] <= 0:
            break
        agree += int(cont[i, j])
        cont[i, :] = -1
        cont[:, j] = -1
    return agree
