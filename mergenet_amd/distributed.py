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


def runs_capacity(n_pixels: int) -> int:
    """Change points the run-length wire holds: n_pixels / 32 (10.6x below the int16 map)."""
    return max(64, n_pixels // 32)


def pack_runs_cpu(mask, class_table, num_instances: int, total_logprob: float, capacity: int,
                  max_instances: int = MAX_INSTANCES):
    """CPU twin of ``mn_pack_runs_device`` (torch tensors; the gloo tests and the checker of the
    device kernels): int32 wire of ``4 + cap + (cap+1)//2 + (max_instances+3)//4`` words."""
    import torch
    flat = mask.reshape(-1).to(torch.int32)
    prev = torch.cat([torch.zeros(1, dtype=torch.int32), flat[:-1]])
    pos = torch.nonzero(flat != prev).reshape(-1).to(torch.int32)
    words = 4 + capacity + (capacity + 1) // 2 + (max_instances + 3) // 4
    wire = torch.zeros(words, dtype=torch.int32)
    n = int(pos.numel())
    wire[0] = n if n <= capacity else -1
    wire[1] = int(num_instances)
    wire[2:4] = torch.tensor([total_logprob], dtype=torch.float64).view(torch.int32)
    m = min(n, capacity)
    wire[4:4 + m] = pos[:m]
    labels = wire[4 + capacity: 4 + capacity + (capacity + 1) // 2].view(torch.int16)
    labels[:m] = flat[pos[:m].long()].to(torch.int16)
    classes = wire[4 + capacity + (capacity + 1) // 2:].view(torch.int8)
    classes[:max_instances] = -1
    classes[:num_instances] = class_table[:num_instances].to(torch.int8)
    return wire


def unpack_runs_cpu(wire, height: int, width: int, capacity: int, max_instances: int = MAX_INSTANCES):
    """CPU twin of ``mn_unpack_runs_device``: (mask int32 [H,W], classes int32 [max_instances], K, loglik)."""
    import torch
    n = int(wire[0])
    if n < 0:
        raise ValueError("run-length wire overflowed its capacity on the sending rank")
    if n > capacity:
        raise ValueError("run-length wire header claims %d change points, capacity is %d (a peer built with "
                         "another capacity, or a damaged buffer)" % (n, capacity))
    pos = wire[4:4 + n].long()
    labels = wire[4 + capacity: 4 + capacity + (capacity + 1) // 2].view(torch.int16)[:n].to(torch.int32)
    N = height * width
    idx = torch.searchsorted(pos, torch.arange(N), right=True)
    full = torch.cat([torch.zeros(1, dtype=torch.int32), labels])
    mask = full[idx].reshape(height, width)
    classes = wire[4 + capacity + (capacity + 1) // 2:].view(torch.int8)[:max_instances].to(torch.int32)
    loglik = float(wire[2:4].clone().view(torch.float64)[0])
    return mask, classes, int(wire[1]), loglik


class MaskExchange:
    """Asynchronous, double-buffered all-gather of one image per rank and step.

    The merger needs ~0.2 ms per 1024x2048 image; an int32 mask is 8 MiB, so a blocking exchange of
    8 of them over xGMI would cost more than the merge.  Two things keep the links off the critical
    path: the wire format and the overlap.

    * ``fmt="runs"`` (default): the row-major label change points of the mask -- the masks are
      piecewise constant -- with K, the class table and the log-likelihood in the same buffer:
      397 KB per 1024x2048 image (``mn_pack_runs_device``; capacity n_pixels / 32 change points; a
      mask with more reports -1 in its header, which every rank sees, and ``result`` exchanges that one
      submit again as an int16 map).  ``fmt="int16"``: one int16 per pixel
      (4.2 MB; ``mn_pack_wire_device``), for masks that do not compress and as the checked
      reference of the tests.  Either way ONE collective per step.
    * the collective of step i runs on the backend's own stream while the kernels of step i+1 run
      on the compute stream (``async_op``); a buffer is reused only after its collective has been
      waited for.  ``wait_ms``: how long collectives held the loop up -- host time inside ``wait()`` (gloo
      blocks there) plus, on the GPU, the stall of the compute stream behind each collective measured
      by event pairs (under nccl ``wait()`` returns at once and only the stream waits).

        ex = MaskExchange(H, W, device, merger=merger)
        slot = ex.submit(mask, class_table, K)      # returns at once
        ...                                          # next image
        masks, tables, counts = ex.result(slot)      # [world,H,W], [world,4096], [world]
        logliks = ex.logprobs(slot)                  # float64 [world]
        ex.drain()

    * ``batch=B``: B consecutive submits share ONE collective (fewer, larger all-gathers: at 0.14 ms per
      image the host cost of issuing a collective per step is what limits a rank, not the links).  The
      collective goes out with the B-th submit, or earlier when a result of the batch is asked for
      (``result`` / ``logprobs`` / ``drain`` flush a partly filled batch; every rank must do so at the
      same point of its loop, as with any collective).
    """

    def __init__(self, height: int, width: int, device, depth: int = 2, fmt: str = "runs", merger=None,
                 batch: int = 1):
        import torch
        import torch.distributed as dist
        if fmt not in ("runs", "int16"):
            raise ValueError("fmt is 'runs' or 'int16'")
        self.torch, self.dist = torch, dist
        self.H, self.W, self.n = height, width, height * width
        self.fmt, self.merger = fmt, merger
        self.cap = runs_capacity(self.n)
        if fmt == "runs":
            self.words = 4 + self.cap + (self.cap + 1) // 2 + (MAX_INSTANCES + 3) // 4
            dtype = torch.int32
        else:
            self.words = self.n + 1 + MAX_INSTANCES + 4
            dtype = torch.int16
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.backend = dist.get_backend() if self.world > 1 else "none"
        self.depth = depth
        self.batch = max(1, int(batch))
        self.send = [torch.empty(self.batch * self.words, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [torch.empty(self.world * self.batch * self.words, dtype=dtype, device=device)
                     for _ in range(depth)]
        self.work = [None] * depth
        self.filled = [0] * depth          # submits packed into the slot's send buffer and not yet sent
        # int16 copy of every submit of a batch, packed AT SUBMIT TIME (the fallback of an overflowing run-length
        # wire must not depend on the caller keeping its mask alive or unchanged: a pipelined loop reuses its
        # output buffers long before the results of a batch are read)
        self.words16 = self.n + 1 + MAX_INSTANCES + 4
        self.stage = ([torch.empty(self.batch * self.words16, dtype=torch.int16, device=device) for _ in range(depth)]
                      if fmt == "runs" else None)
        self.fallback = {}                 # (slot, pos) -> gathered int16 wires of an overflowed submit
        self.serial = [0] * depth          # number of the batch a slot holds (tags the fallback exchange)
        self.count = 0
        self._host_wait_ms = 0.0
        self._stall_events = []            # (before, after) event pairs around work.wait() on the compute stream
        self._stall_done_ms = 0.0          # ... finished pairs, summed
        self.bytes_per_rank = self.words * (4 if fmt == "runs" else 2)

    def _pack(self, mask, class_table, num_instances, wire, total_logprob):
        torch = self.torch
        if self.fmt == "runs":
            if mask.is_cuda:
                from . import segmenter
                if self.merger is None:
                    raise ValueError("MaskExchange(fmt='runs') on the GPU needs merger= (scratch of the pack kernels)")
                segmenter.pack_runs(self.merger, mask, class_table, num_instances, wire, self.cap,
                                    MAX_INSTANCES, total_logprob)
            else:
                wire.copy_(pack_runs_cpu(mask, class_table, num_instances, total_logprob, self.cap))
            return
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
        slot = (self.count // self.batch) % self.depth
        pos = self.count % self.batch
        if pos == 0:
            self.wait(slot)               # the slot's previous collective (results of it are gone after this)
        self._pack(mask, class_table, num_instances,
                   self.send[slot][pos * self.words:(pos + 1) * self.words], total_logprob)
        if self.fmt == "runs":
            saved = self.fmt
            self.fmt = "int16"
            try:
                self._pack(mask, class_table, num_instances,
                           self.stage[slot][pos * self.words16:(pos + 1) * self.words16], total_logprob)
            finally:
                self.fmt = saved
            self.fallback.pop((slot, pos), None)
        if pos == 0:
            self.serial[slot] = self.count // self.batch
        self.filled[slot] = pos + 1
        if pos == self.batch - 1:
            self._launch(slot)
        self.count += 1
        return slot * self.batch + pos

    def _launch(self, slot: int) -> None:
        torch = self.torch
        self.filled[slot] = 0
        if self.world == 1:
            self.recv[slot].copy_(self.send[slot])
        else:
            # as bytes: neither NCCL/RCCL nor gloo has an int16 type, and a gather needs none
            self.work[slot] = self.dist.all_gather_into_tensor(self.recv[slot].view(torch.uint8),
                                                               self.send[slot].view(torch.uint8),
                                                               async_op=True)

    def wait(self, slot: int) -> None:
        if self.filled[slot]:             # a partly filled batch: send what is there
            self._launch(slot)
            if self.batch > 1:            # (the next submit starts a new batch)
                self.count += (self.batch - self.count % self.batch) % self.batch
        if self.work[slot] is not None:
            import time
            # How long the exchange holds the loop up.  With gloo (CPU tensors) wait() blocks the host and
            # the host time is the answer.  With nccl (= RCCL) wait() only makes the CURRENT STREAM wait for
            # the collective and returns at once, so host time would say "0" whatever happens: the stall
            # is the time between two events recorded on the compute stream around the wait.
            t = time.perf_counter()
            cuda = self.send[slot].is_cuda
            if cuda:
                a = self.torch.cuda.Event(enable_timing=True)
                b = self.torch.cuda.Event(enable_timing=True)
                a.record()
            self.work[slot].wait()
            if cuda:
                b.record()
                self._stall_events.append((a, b))
                # pairs that have completed are folded into the running sum (a long loop would otherwise keep
                # two events per collective for ever and re-read all of them on every wait_ms)
                while len(self._stall_events) > 8 and self._stall_events[0][1].query():
                    a0, b0 = self._stall_events.pop(0)
                    self._stall_done_ms += a0.elapsed_time(b0)
            self._host_wait_ms += (time.perf_counter() - t) * 1e3
            self.work[slot] = None

    @property
    def wait_ms(self) -> float:
        """Milliseconds the loop was held up by collectives so far: host time inside wait() plus, on the
        GPU, the stall of the compute stream behind each collective (event pairs; reading it synchronises
        with the last of them)."""
        for a, b in self._stall_events:
            b.synchronize()
            self._stall_done_ms += a.elapsed_time(b)
        self._stall_events = []
        return self._host_wait_ms + self._stall_done_ms

    @wait_ms.setter
    def wait_ms(self, v: float) -> None:
        self._host_wait_ms = float(v)
        self._stall_events = []
        self._stall_done_ms = 0.0

    def _int16_fallback(self, slot: int, pos: int):
        """A rank's mask had more label changes than the run-length wire holds (header -1, seen by every
        rank in the gathered batch): that submit is exchanged again as an int16 map, from the copy packed at
        submit time.  This is a COLLECTIVE issued from ``result`` / ``logprobs``: as with any collective, every
        rank must ask for the results of its handles at the same points of its loop and in the same order.
        The exchange carries the handle's number; ranks that disagree raise instead of delivering another
        submit's mask."""
        key = (slot, pos)
        if key in self.fallback:
            return self.fallback[key]
        torch = self.torch
        words = self.words16 + 4
        send = torch.empty(words, dtype=torch.int16, device=self.send[slot].device)
        send[: self.words16] = self.stage[slot][pos * self.words16:(pos + 1) * self.words16]
        tag = self.serial[slot] * self.batch + pos
        send[self.words16:] = torch.tensor([tag], dtype=torch.int64).view(torch.int16).to(send.device)
        recv = torch.empty(self.world * words, dtype=torch.int16, device=send.device)
        if self.world == 1:
            recv.copy_(send)
        else:
            self.dist.all_gather_into_tensor(recv.view(torch.uint8), send.view(torch.uint8))
        got = recv.view(self.world, words)
        tags = got[:, self.words16:].reshape(-1).clone().view(torch.int64)
        if bool((tags != tag).any()):
            raise RuntimeError("MaskExchange: ranks asked for the results of different submits (%s, here %d): "
                               "result() / logprobs() must be called for the same handles in the same order on "
                               "every rank" % (tags.tolist(), tag))
        self.fallback[key] = got[:, : self.words16]
        return self.fallback[key]

    def result(self, slot: int):
        """(masks [world,H,W], class tables [world,MAX_INSTANCES] padded with -1, counts [world])."""
        slot, pos = divmod(slot, self.batch)
        self.wait(slot)
        torch = self.torch
        r = self.recv[slot].view(self.world, self.batch, self.words)[:, pos]
        if self.fmt == "int16":
            return (r[:, : self.n].view(self.world, self.H, self.W),
                    r[:, self.n + 1: self.n + 1 + MAX_INSTANCES], r[:, self.n])
        if bool((r[:, 0] > self.cap).any()):
            raise ValueError("a rank's wire claims more change points than the capacity %d: peers disagree "
                             "about the wire layout" % self.cap)
        if bool((r[:, 0] < 0).any()):         # some rank's mask does not fit the run-length wire
            f = self._int16_fallback(slot, pos)
            return (f[:, : self.n].view(self.world, self.H, self.W).to(torch.int32),
                    f[:, self.n + 1: self.n + 1 + MAX_INSTANCES].to(torch.int32), f[:, self.n].to(torch.int32))
        if r.is_cuda:                          # every rank's wire in ONE launch
            from . import segmenter
            masks, tabs = segmenter.unpack_runs_batch(r, self.H, self.W, self.cap, MAX_INSTANCES)
            return masks, tabs, r[:, 1].clone()
        masks, tabs = [], []
        for w in range(self.world):
            m, t, _, _ = unpack_runs_cpu(r[w], self.H, self.W, self.cap)
            masks.append(m)
            tabs.append(t)
        return torch.stack(masks), torch.stack(tabs), r[:, 1].clone()

    def logprobs(self, slot: int):
        """Total log-likelihood of every rank's image (float64 [world])."""
        slot, pos = divmod(slot, self.batch)
        self.wait(slot)
        r = self.recv[slot].view(self.world, self.batch, self.words)[:, pos]
        if self.fmt == "runs" and bool((r[:, 0] < 0).any()):
            f = self._int16_fallback(slot, pos)
            return f[:, self.n + 1 + MAX_INSTANCES:].reshape(-1).clone().view(self.torch.float64)
        if self.fmt == "runs":
            return r[:, 2:4].reshape(-1).clone().view(self.torch.float64)
        return r[:, self.n + 1 + MAX_INSTANCES:].reshape(-1).clone().view(self.torch.float64)

    def drain(self) -> None:
        for slot in range(self.depth):
            self.wait(slot)
