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


def _test_mask(H, W, rank, step):
    """A piecewise-constant label map with a few rectangles (what a final instance mask looks like)."""
    m = torch.zeros((H, W), dtype=torch.int32)
    m[1:3, 1:4] = 1 + (rank + step) % 5
    m[2:H, W - 3:] = 2 + (rank + 2 * step) % 4
    m[H - 1, 0] = 1
    return m


def _exchange_worker(rank, world, port, out, fmt):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W = 5, 7
    ex = mnd.MaskExchange(H, W, torch.device("cpu"), depth=2, fmt=fmt)
    ok = ex.world == world and ex.backend == "gloo"
    slots = []
    # more steps than buffers: slot reuse must wait for the earlier collective
    for step in range(5):
        mask = _test_mask(H, W, rank, step)
        k = 1 + (rank + step) % 3
        table = torch.full((H * W,), -1, dtype=torch.int32)
        table[:k] = torch.arange(k, dtype=torch.int32) + 3 * rank + step
        slots.append(ex.submit(mask, table, k, -1000.5 * (rank + 1) - step))
        masks, tabs, counts = ex.result(slots[-1])
        ok &= ex.logprobs(slots[-1]).tolist() == [-1000.5 * (r + 1) - step for r in range(world)]
        ok &= masks.dtype == (torch.int32 if fmt == "runs" else torch.int16) and masks.shape == (world, H, W)
        ok &= tabs.shape == (world, mnd.MAX_INSTANCES)
        for r in range(world):
            kr = 1 + (r + step) % 3
            ok &= bool((masks[r].to(torch.int32) == _test_mask(H, W, r, step)).all())
            ok &= int(counts[r]) == kr
            ok &= tabs[r, :kr].tolist() == [i + 3 * r + step for i in range(kr)]
            ok &= bool((tabs[r, kr:] == -1).all())
    ex.drain()
    ok &= slots == [0, 1, 0, 1, 0]
    ok &= ex.wait_ms >= 0.0
    out[rank] = (bool(ok), ex.bytes_per_rank)
    dist.destroy_process_group()


@pytest.mark.parametrize("fmt", ["runs", "int16"])
def test_mask_exchange_async_world2_gloo(fmt):
    """The double-buffered exchange bench.py uses for N > 1 (CPU tensors, gloo): the run-length wire
    (default) and the int16 map it replaces deliver the same masks, tables, counts, log-likelihoods."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, port, out, fmt), nprocs=world, join=True)
    assert {k: v[0] for k, v in dict(out).items()} == {0: True, 1: True}


def _exchange_batch_worker(rank, world, port, out, fmt):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, B = 5, 7, 3
    ex = mnd.MaskExchange(H, W, torch.device("cpu"), depth=2, fmt=fmt, batch=B)
    ok = True
    handles = []

    def check(step, handle):
        good = True
        masks, tabs, counts = ex.result(handle)
        good &= ex.logprobs(handle).tolist() == [-7.25 * (r + 1) - step for r in range(world)]
        for r in range(world):
            kr = 1 + (r + step) % 3
            good &= bool((masks[r].to(torch.int32) == _test_mask(H, W, r, step)).all())
            good &= int(counts[r]) == kr and tabs[r, :kr].tolist() == [i + 3 * r + step for i in range(kr)]
        return good

    # 8 submits with batches of 3: collectives after submits 3 and 6, the last two flushed by result()
    for step in range(8):
        k = 1 + (rank + step) % 3
        table = torch.full((H * W,), -1, dtype=torch.int32)
        table[:k] = torch.arange(k, dtype=torch.int32) + 3 * rank + step
        handles.append(ex.submit(_test_mask(H, W, rank, step), table, k, -7.25 * (rank + 1) - step))
        if step % B == B - 1:                      # a batch has gone out: read all of it
            for j in range(step - B + 1, step + 1):
                ok &= check(j, handles[j])
    ok &= handles == [0, 1, 2, 3, 4, 5, 0, 1]      # slot * B + position, two slots in rotation
    ok &= check(7, handles[7]) and check(6, handles[6])    # partly filled batch: flushed on demand
    # after a flush the next submit starts a new batch in the next slot
    h = ex.submit(_test_mask(H, W, rank, 9), torch.full((H * W,), -1, dtype=torch.int32), 0, 0.0)
    ok &= h == 3
    ex.drain()
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("fmt", ["runs", "int16"])
def test_mask_exchange_batched_world2_gloo(fmt):
    """batch=3: three submits share one collective; a partly filled batch goes out when one of its
    results is asked for; every delivered mask, table, count and log-likelihood is checked."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_batch_worker, args=(world, port, out, fmt), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_run_length_wire_round_trip_and_size():
    """pack_runs_cpu / unpack_runs_cpu (the CPU twins of mn_pack_runs_device / mn_unpack_runs_device):
    random piecewise-constant masks come back exactly; a mask with too many changes says so; at
    1024x2048 the wire is >= 10x smaller than the int16 map."""
    g = torch.Generator().manual_seed(5)
    for H, W in [(1, 1), (3, 5), (16, 64), (33, 70)]:
        cap = H * W                      # (every mask fits; the production capacity is n_pixels / 32)
        coarse = torch.randint(0, 6, ((H + 7) // 8, (W + 7) // 8), generator=g, dtype=torch.int32)
        mask = coarse.repeat_interleave(8, 0).repeat_interleave(8, 1)[:H, :W].contiguous()
        k = int(mask.max())
        table = torch.arange(1, k + 1, dtype=torch.int32)
        wire = mnd.pack_runs_cpu(mask, table, k, -12.5, cap)
        m2, cls, k2, ll = mnd.unpack_runs_cpu(wire, H, W, cap)
        assert torch.equal(m2, mask) and k2 == k and ll == -12.5
        assert cls[:k].tolist() == table.tolist() and bool((cls[k:] == -1).all())
    noisy = torch.randint(0, 9, (16, 64), generator=g, dtype=torch.int32)      # ~900 changes > capacity 64
    wire = mnd.pack_runs_cpu(noisy, torch.arange(1, 9, dtype=torch.int32), 8, 0.0, mnd.runs_capacity(16 * 64))
    assert int(wire[0]) == -1
    with pytest.raises(ValueError):
        mnd.unpack_runs_cpu(wire, 16, 64, mnd.runs_capacity(16 * 64))
    n = 1024 * 2048
    runs = mnd.MaskExchange(1024, 2048, torch.device("cpu"), fmt="runs").bytes_per_rank
    i16 = mnd.MaskExchange(1024, 2048, torch.device("cpu"), fmt="int16").bytes_per_rank
    assert i16 >= 10 * runs and runs < n // 5


def test_mask_exchange_single_process_and_limits():
    ex = mnd.MaskExchange(3, 4, torch.device("cpu"), fmt="int16")
    mask = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    table = torch.tensor([5, 7, -1, -1], dtype=torch.int32)
    slot = ex.submit(mask, table, 2, -3.25)
    masks, tabs, counts = ex.result(slot)
    assert ex.logprobs(slot).tolist() == [-3.25]
    assert masks.shape == (1, 3, 4) and masks[0].tolist() == mask.tolist()
    assert tabs[0, :3].tolist() == [5, 7, -1] and int(counts[0]) == 2
    with pytest.raises(ValueError):
        ex.submit(mask, table, mnd.MAX_INSTANCES + 1)


def _fallback_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, B = 16, 64, 2                   # capacity max(64, 1024 / 32) = 64 change points
    ex = mnd.MaskExchange(H, W, torch.device("cpu"), depth=2, fmt="runs", batch=B)
    g = torch.Generator().manual_seed(11)
    noisy = torch.randint(0, 9, (H, W), generator=g, dtype=torch.int32)        # ~900 label changes: does not fit
    calm = _test_mask(H, W, rank, 0)
    ok = True
    table = torch.arange(1, 10, dtype=torch.int32)
    # step 0: only rank 1's mask overflows; step 1: nobody's; both in one batch
    buf = (noisy if rank == 1 else calm).clone()       # the caller's output buffer ...
    h0 = ex.submit(buf, table, 8, -1.5 * (rank + 1))
    buf.fill_(7)                                       # ... reused before the result is read (a pipelined loop
    h1 = ex.submit(calm, table, 3, -2.5 * (rank + 1))  # does that): the fallback must not read it again
    masks, tabs, counts = ex.result(h0)       # every rank sees rank 1's header -1 and joins the int16 exchange
    ok &= masks.shape == (world, H, W) and masks.dtype == torch.int32
    ok &= bool((masks[1] == noisy).all()) and bool((masks[0] == _test_mask(H, W, 0, 0)).all())
    ok &= counts.tolist() == [8, 8] and tabs[1, :8].tolist() == list(range(1, 9))
    ok &= ex.logprobs(h0).tolist() == [-1.5, -3.0]
    masks, tabs, counts = ex.result(h1)       # the other submit of the batch went through the run-length wire
    ok &= all(bool((masks[r] == _test_mask(H, W, r, 0)).all()) for r in range(world))
    ok &= counts.tolist() == [3, 3] and ex.logprobs(h1).tolist() == [-2.5, -5.0]
    ex.drain()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_mask_exchange_overflowing_mask_falls_back_to_int16_world2_gloo():
    """A mask with more label changes than the run-length wire holds: its header says -1, every rank sees
    it, and that ONE submit is exchanged again as an int16 map (round 2: result() raised) -- from a copy packed
    at submit time, so the caller may reuse its mask buffer before reading the result (advisor, round 3)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_fallback_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_run_length_wire_with_a_damaged_header_is_rejected():
    """A header that claims more change points than the capacity (a peer built with another capacity)
    must not send the reader past the wire's sections."""
    H, W = 8, 16
    cap = mnd.runs_capacity(H * W)
    wire = mnd.pack_runs_cpu(_test_mask(H, W, 0, 0), torch.arange(1, 4, dtype=torch.int32), 3, 0.0, cap)
    wire[0] = cap + 5
    with pytest.raises(ValueError):
        mnd.unpack_runs_cpu(wire, H, W, cap)
