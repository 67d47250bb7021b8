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


class MaskExchange:
    """Asynchronous, double-buffered all-gather of one image per rank and step.

    The merger needs 0.4 ms per 1024x2048 image; an int32 mask is 8 MiB, so a blocking exchange of
    8 of them over xGMI would cost more than the merge.  Two things keep the links off the critical
    path: the wire format is one int16 buffer per image -- ``[H*W labels][K][MAX_INSTANCES classes,
    -1 padded][float64 log-likelihood as 4 words]`` (labels <= 4096, classes < 128), half the
    bytes and ONE collective instead of two
    -- and the collective of step i runs on the backend's own stream while the kernels of step
    i+1 run on the compute stream (``async_op``); a buffer is reused only after its collective
    has been waited for.

        ex = MaskExchange(H, W, device)
        slot = ex.submit(mask, class_table, K)      # returns at once
        ...                                          # next image
        masks, tables, counts = ex.result(slot)      # int16 [world,H,W], int16 [world,4096], [world]
        logliks = ex.logprobs(slot)                  # float64 [world]
        ex.drain()
    """

    def __init__(self, height: int, width: int, device, depth: int = 2):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.H, self.W, self.n = height, width, height * width
        self.words = self.n + 1 + MAX_INSTANCES + 4
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.depth = depth
        self.send = [torch.empty(self.words, dtype=torch.int16, device=device) for _ in range(depth)]
        self.recv = [torch.empty(self.world * self.words, dtype=torch.int16, device=device)
                     for _ in range(depth)]
        self.work = [None] * depth
        self.count = 0

    def _pack(self, mask, class_table, num_instances, wire, total_logprob):
        torch = self.torch
        if mask.is_cuda:                      # HIP kernel of the library, on the current stream
            from . import segmenter
            segmenter.pack_wire(mask, class_table, num_instances, wire, MAX_INSTANCES, total_logprob)
            return
        wire[: self.n] = mask.reshape(-1)     # CPU tensors: the gloo tests
        wire[self.n] = num_instances
        wire[self.n + 1: self.n + 1 + MAX_INSTANCES] = -1
        wire[self.n + 1: self.n + 1 + num_instances] = class_table[:num_instances]
        wire[self.n + 1 + MAX_INSTANCES:] = torch.tensor([total_logprob], dtype=torch.float64).view(torch.int16)

    def submit(self, mask, class_table, num_instances: int, total_logprob: float = float("nan")) -> int:
        if num_instances > MAX_INSTANCES:
            raise ValueError("more than %d instances in one image" % MAX_INSTANCES)
        torch = self.torch
        slot = self.count % self.depth
        self.wait(slot)
        self._pack(mask, class_table, num_instances, self.send[slot], total_logprob)
        if self.world == 1:
            self.recv[slot].copy_(self.send[slot])
        else:
            # as bytes: neither NCCL/RCCL nor gloo has an int16 type, and a gather needs none
            self.work[slot] = self.dist.all_gather_into_tensor(self.recv[slot].view(torch.uint8),
                                                               self.send[slot].view(torch.uint8),
                                                               async_op=True)
        self.count += 1
        return slot

    def wait(self, slot: int) -> None:
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None

    def result(self, slot: int):
        self.wait(slot)
        r = self.recv[slot].view(self.world, self.words)
        return (r[:, : self.n].view(self.world, self.H, self.W),
                r[:, self.n + 1: self.n + 1 + MAX_INSTANCES], r[:, self.n])

    def logprobs(self, slot: int):
        """Total log-likelihood of every rank's image (float64 [world])."""
        self.wait(slot)
        r = self.recv[slot].view(self.world, self.words)
        return r[:, self.n + 1 + MAX_INSTANCES:].reshape(-1).clone().view(self.torch.float64)

    def drain(self) -> None:
        for slot in range(self.depth):
            self.wait(slot)

