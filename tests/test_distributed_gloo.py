"""The N>1 exchange path on CPU: world_size-2 gloo all-gather of masks and class tables."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mergenet_amd import distributed as mnd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 6, 10
    mask = torch.full((H, W), rank + 1, dtype=torch.int32)
    mask[0, 0] = 0
    table = torch.full((H * W,), -1, dtype=torch.int32)
    k = rank + 2
    table[:k] = torch.arange(1, k + 1, dtype=torch.int32) + 10 * rank
    masks, tabs, counts = mnd.gather_masks(mask, table, k)
    ok = masks.shape == (world, H, W) and tabs.shape == (world, mnd.MAX_INSTANCES)
    for r in range(world):
        ok &= bool((masks[r, 1:, :] == r + 1).all()) and int(masks[r, 0, 0]) == 0
        ok &= int(counts[r]) == r + 2
        ok &= tabs[r, : r + 2].tolist() == [i + 10 * r for i in range(1, r + 3)]
        ok &= bool((tabs[r, r + 2:] == -1).all())
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_gather_masks_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_gather_masks_single_process():
    mask = torch.ones((3, 4), dtype=torch.int32)
    table = torch.tensor([5, 7, -1, -1], dtype=torch.int32)
    masks, tabs, counts = mnd.gather_masks(mask, table, 2)
    assert masks.shape == (1, 3, 4) and tabs[0, :3].tolist() == [5, 7, -1] and int(counts[0]) == 2


def test_shard_indices_cover_every_image_once():
    for n, world in [(8, 8), (8, 4), (13, 4), (3, 8)]:
        seen = sorted(i for r in range(world) for i in mnd.shard_indices(n, r, world))
        assert seen == list(range(n))
    with pytest.raises(ValueError):
        mnd.shard_indices(4, 4, 4)


def _exchange_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 5, 7
    ex = mnd.MaskExchange(H, W, torch.device("cpu"), depth=2)
    ok = True
    slots = []
    # more steps than buffers: slot reuse must wait for the earlier collective
    for step in range(5):
        mask = torch.full((H, W), (rank + step) % 7, dtype=torch.int32)
        k = 1 + (rank + step) % 3
        table = torch.full((H * W,), -1, dtype=torch.int32)
        table[:k] = torch.arange(k, dtype=torch.int32) + 3 * rank + step
        slots.append(ex.submit(mask, table, k, -1000.5 * (rank + 1) - step))
        masks, tabs, counts = ex.result(slots[-1])
        ok &= ex.logprobs(slots[-1]).tolist() == [-1000.5 * (r + 1) - step for r in range(world)]
        ok &= masks.dtype == torch.int16 and masks.shape == (world, H, W)
        ok &= tabs.shape == (world, mnd.MAX_INSTANCES)
        for r in range(world):
            kr = 1 + (r + step) % 3
            ok &= bool((masks[r] == (r + step) % 7).all())
            ok &= int(counts[r]) == kr
            ok &= tabs[r, :kr].tolist() == [i + 3 * r + step for i in range(kr)]
            ok &= bool((tabs[r, kr:] == -1).all())
    ex.drain()
    ok &= slots == [0, 1, 0, 1, 0]
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_mask_exchange_async_world2_gloo():
    """The double-buffered int16 exchange bench.py uses for N > 1 (CPU tensors, gloo)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_mask_exchange_single_process_and_limits():
    ex = mnd.MaskExchange(3, 4, torch.device("cpu"))
    mask = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    table = torch.tensor([5, 7, -1, -1], dtype=torch.int32)
    slot = ex.submit(mask, table, 2, -3.25)
    masks, tabs, counts = ex.result(slot)
    assert ex.logprobs(slot).tolist() == [-3.25]
    assert masks.shape == (1, 3, 4) and masks[0].tolist() == mask.tolist()
    assert tabs[0, :3].tolist() == [5, 7, -1] and int(counts[0]) == 2
    with pytest.raises(ValueError):
        ex.submit(mask, table, mnd.MAX_INSTANCES + 1)
