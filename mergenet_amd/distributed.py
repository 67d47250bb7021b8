"""Image-level sharding and the one exchange step of the path: all-gather of the final masks.

Images are independent units (the reference shards them by process: ``--job/--num-jobs``,
``utils/dataset.py:56-63``, ``egs/cityscape/local/segment.py:59-61``), so the merger itself needs
no collective.  One process per GPU; at the end every rank contributes its int32 mask and a
fixed-length class table and receives everybody's (``torch.distributed`` all_gather: RCCL over
xGMI with the ``nccl`` backend, gloo on CPU in the tests).
"""

from __future__ import annotations

from typing import List, Tuple

MAX_INSTANCES = 4096   # fixed class-table length on the wire


def shard_indices(num_images: int, rank: int, world: int) -> List[int]:
    """Image i goes to rank i mod world (round-robin keeps the per-rank load within one image)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, num_images, world))


def gather_masks(mask, class_table, num_instances: int):
    """All-gather one image per rank.

    mask [H,W] int32, class_table [>=K] int32 (class of label k at k-1, as written by the merger),
    num_instances K.  Returns (masks [world,H,W], tables [world,MAX_INSTANCES] padded with -1,
    counts [world]).  Without an initialised process group the inputs are returned with a
    leading axis of 1.
    """
    import torch
    import torch.distributed as dist

    if num_instances > MAX_INSTANCES:
        raise ValueError("more than %d instances in one image" % MAX_INSTANCES)
    packed = torch.full((MAX_INSTANCES + 1,), -1, dtype=torch.int32, device=mask.device)
    packed[0] = num_instances
    packed[1:1 + num_instances] = class_table[:num_instances]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return mask.unsqueeze(0), packed[1:].unsqueeze(0), packed[:1].clone()
    world = dist.get_world_size()
    masks = [torch.empty_like(mask) for _ in range(world)]
    tabs = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(masks, mask.contiguous())
    dist.all_gather(tabs, packed)
    tabs = torch.stack(tabs)
    return torch.stack(masks), tabs[:, 1:], tabs[:, 0].clone()
