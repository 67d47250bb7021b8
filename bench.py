#!/usr/bin/env python
"""bench.py -- merged Mpixels/s of the HIP pixel merger on synthetic 1024x2048 maps.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: every rank merges ONE
1024x2048 Cityscapes-shape image (C=9 classes, O=10 log-spiral offsets, options 0/1/0.03 of
egs/cityscape/local/segment.py:134-136) whose probability maps are already resident in HBM, and --
for N > 1 -- the final masks and class tables are all-gathered over RCCL (the only exchange step
of the path; images are independent; run-length wire format, 8 steps per collective, the gathers overlap
the merges of the following steps, all of them complete inside the timed region).  Weak scaling: per-GPU work is fixed.  Each rank cycles
through POOL different images (636 MB of maps, more than the 256 MiB Infinity Cache), so a step
reads its maps from HBM and not from a cache warmed by the previous step.

Rank 0 prints ONE JSON line with BOTH paths on the same images:

* `value` = pixels merged by all ranks / wall time of the K timed steps (max over ranks), in Mpixel/s, on the
  library's SPECULATIVE fast path (mode AUTO with require_proof = -1: the component contraction when the maps are
  sign-separable, else the general rounds -- `mode_used` says which ran).  On these images the fast path's answer
  is not certified (the 0.03 bias lets the background swallow small instances in a second phase whose order
  matters), so its equality with the reference is MEASURED on every image of the pool in every run (`id_match`,
  against the reference's own outputs) instead of proven;
* `value_proven_path` / `default_mode` = the library's DEFAULT behaviour on the same pool, outside the timed region:
  one image through mn_default_options (AUTO: the speculative attempt fails its certificate, the exact engine --
  the reference's sequential order on the GPU -- redoes the image) and ONE mn_segment_exact_batch launch of as many
  1024x2048 images as fit the GPU's memory (a wavefront per image), every result compared with the reference's;
  `exact_engine` holds single images at smaller sizes.

`roofline` prices the slowest streaming kernel of the timed path with HIP events taken on
the launch stream inside the library (algorithmic bytes: 4 B per plane value, SURVEY.md section
8d: C class planes or O sameness planes per pass); `passes` lists every streaming pass of the
step the same way, and `scoring_pass_general_path` is the class pass + edge pass of the general
rounds (the north-star kernel), measured live on the same images outside the timed region.
`cpu_baseline` times the reference's own compiled segment.cc (oracle/_ref, kind "reference")
or, if that is absent, our C++ restatement (kind "port") on one core over a bounded 256x512
sample of the same generator.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, C = 1024, 2048, 9
OFFSETS_ARGS = (40, 10)
OPTS = (0.0, 1.0, 0.03)
PIPELINED_DEPTH = 4            # images in flight in the `pipelined` side measurement
EVENTS_EVERY = 16              # one timed step in 16 records every HIP event of the library; all record the sweep's
POOL = 4                       # images per rank, cycled: 4 x 159 MB of maps > 256 MiB Infinity Cache
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def _cpu_one(seed):
    """One 256x512 image through the CPU merger (worker of cpu_baseline)."""
    from mergenet_amd import synth
    from oracle import checker as ck
    offs = synth.generate_offsets(*OFFSETS_ARGS)
    s = synth.synth_v1(256, 512, C, offs, seed)
    fn = ck.run_reference if ck.have_reference() else ck.run_csegment
    t = time.perf_counter()
    fn(s.class_probs, s.sameness_probs, C, offs, *OPTS)
    return time.perf_counter() - t


def cpu_baseline():
    """Bounded CPU sample (about 15-20 s): 256x512 images of the same generator through the
    reference's compiled merger -- one image on one core, then one image per core on up to 8
    cores at once (the reference's own scaling model: independent jobs, --num-jobs)."""
    import multiprocessing as mp
    from oracle import checker as ck
    kind = "reference" if ck.have_reference() else "port"
    dt = _cpu_one(1000)
    out = {"value": round(256 * 512 / dt / 1e6, 6), "unit": "Mpixel/s", "cores": 1, "kind": kind,
           "sample": "one 256x512 synth-v1 image (seed 1000, C=9, O=10, opts 0/1/0.03): "
                     "%.1f s on one host core; the reference is single-threaded and "
                     "super-linear in pixels (435 s for 1024x2048 in the build container)" % dt,
           "seconds": round(dt, 3)}
    procs = max(1, min(8, (os.cpu_count() or 1) - 1))
    if procs > 1:
        t = time.perf_counter()
        with mp.get_context("spawn").Pool(procs) as pool:
            pool.map(_cpu_one, [1000 + i for i in range(procs)])
        wall = time.perf_counter() - t
        out["all_jobs"] = {"value": round(procs * 256 * 512 / wall / 1e6, 6), "unit": "Mpixel/s",
                           "cores": procs, "seconds": round(wall, 3),
                           "sample": "%d independent processes, one 256x512 image each (seeds 1000..%d), "
                                     "wall time incl. process start and input generation"
                                     % (procs, 999 + procs)}
    return out


def exact_engine_sample(seg, synth, offs, device):
    """One 512x1024 synth-v1 image (the reference caller's working size; the reference needs 84 s) and one
    blurred 256x512 map through MN_MODE_EXACT, each checked against the reference's own output."""
    import numpy as np
    import torch
    from mergenet_amd import labels as ck
    out = []
    for (name, hh, ww, maker) in (("cseg_synth_512x1024_s1000", 512, 1024, lambda: synth.synth_v1(512, 1024, C, offs, 1000)),
                                  ("cseg_blur_256x512_r2", 256, 512, None),
                                  # an image whose instance borders the ORDER AMONG BIT-EQUAL PRIORITIES decides: the
                                  # exact engine meets tied pops and the image is redone in the reference's own
                                  # order (its std::priority_queue and unordered_map restated: mn_reforder.h)
                                  ("cseg_blur4_128x256_s5100", 128, 256, None)):
        golden = os.path.join(ROOT, "tests", "golden", name + ".npz")
        if not os.path.exists(golden):
            continue
        z = np.load(golden)
        if maker is None:
            spec = json.loads(str(z["spec"]))
            img = synth.blurred_v1(hh, ww, C, offs, spec["seed"], radius=spec["radius"], noise=spec["noise"],
                                   num_instances=spec.get("num_instances"))
        else:
            img = maker()
        m = seg.Merger(hh, ww, C, len(offs), device=device)
        cp = torch.from_numpy(img.class_probs).cuda(device)
        sp = torch.from_numpy(img.sameness_probs).cuda(device)
        o = seg.default_options(same_different_bias=OPTS[0], object_merge_factor=OPTS[1],
                                merge_logprob_bias=OPTS[2], mode=seg.MN_MODE_EXACT, clip_inputs=1)
        torch.cuda.synchronize()
        t = time.perf_counter()
        mask, table, _, st = m.segment(cp, sp, offs, o)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        got = [int(c) for c in table.cpu().numpy()[: st["num_instances"]]]
        same = bool(ck.masks_equivalent(mask.cpu().numpy(), got, z["mask"], [int(c) for c in z["object_class"]]))
        out.append({"image": name, "seconds": round(dt, 3), "value": round(hh * ww / dt / 1e6, 4), "unit": "Mpixel/s",
                    "steps": st["finisher_steps"], "merges": st["merges"], "proof": st["proof"],
                    "tied_steps": st["tied_steps"], "tied_conflicts": st["tied_conflicts"],
                    "tie_order_used": st["tie_order_used"], "equals_reference": same})
        m.close()
    pool_out = None            # (the batch measurement moved to `default_mode`, at the benchmark's own size)
    return {"what": "MN_MODE_EXACT: the reference's sequential order (segment.cc:539-727) with its float32 "
                    "arithmetic, one wavefront per image; what AUTO falls back to when the fast path cannot "
                    "certify its answer.  The reference itself: 84 s at 512x1024, 12-15 s at 256x512 (BASELINE.md)",
            "samples": out, "batch": pool_out}


def default_mode_block(seg, pool_images, seeds, offs, device, max_batch):
    """The PROVEN path on the benchmark's own images (configs[1], 1024x2048): the library's default options
    (MN_MODE_AUTO, require_proof = 0).  The speculative attempt cannot certify these images (second-phase
    merges), so each is redone by the exact engine -- the reference's sequential order itself.  (a) ONE image
    through mn_segment_device with default options: the latency of the default call; (b) ONE
    mn_segment_exact_batch of as many images as fit (a workgroup per image in one launch: how the sequential
    order gets throughput; the reference scales the same way, by processes).  Every result is compared with
    the reference's own output (golden vectors)."""
    import numpy as np
    import torch
    from mergenet_amd import labels as ck
    goldens = []
    for sd in seeds:
        gname = "cseg_synth_1024x2048_cfg2.npz" if sd == 1000 else "cseg_synth_1024x2048_s%d.npz" % sd
        path = os.path.join(ROOT, "tests", "golden", gname)
        goldens.append(np.load(path) if os.path.exists(path) else None)

    def equal(mask, table, st, z):
        if z is None:
            return None
        got = [int(c) for c in table.cpu().numpy()[: st["num_instances"]]]
        return bool(ck.masks_equivalent(mask.cpu().numpy(), got, z["mask"], [int(c) for c in z["object_class"]]))

    O = len(offs)
    o_default = seg.default_options(same_different_bias=OPTS[0], object_merge_factor=OPTS[1],
                                    merge_logprob_bias=OPTS[2], clip_inputs=1)          # mode AUTO, require_proof 0
    m = seg.Merger(H, W, C, O, device=device)
    cp, sp = pool_images[0]
    torch.cuda.synchronize()
    t = time.perf_counter()
    mask, table, _, st = m.segment(cp, sp, offs, o_default)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    single = {"seconds": round(dt, 3), "value": round(H * W / dt / 1e6, 4), "unit": "Mpixel/s",
              "mode_used": {1: "exact", 2: "rounds", 3: "components"}.get(st["mode_used"]), "proof": st["proof"],
              "steps": st["finisher_steps"], "merges": st["merges"], "tied_steps": st["tied_steps"],
              "tied_conflicts": st["tied_conflicts"], "tie_order_used": st["tie_order_used"],
              "equals_reference": equal(mask, table, st, goldens[0]),
              "how": "one mn_segment_device call with mn_default_options (AUTO): speculative attempt, certificate "
                     "fails, exact engine"}
    # as many images as fit: a context that only serves the exact engine holds what this one holds minus the fast
    # path's own arrays (fixed-point class sums, best-record slots, edge masks: 8 C + 28 B per pixel), + its outputs
    per_image = int(1.03 * (m.workspace_bytes() - (8 * C + 28) * H * W) + 2 * 4 * H * W)
    m.close()
    del mask, table
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info(device)
    count = int(max(1, min(max_batch, (0.92 * free) // per_image)))
    batch_out = None
    o_exact = seg.default_options(same_different_bias=OPTS[0], object_merge_factor=OPTS[1],
                                  merge_logprob_bias=OPTS[2], mode=seg.MN_MODE_EXACT, clip_inputs=1)
    for attempt in range(3):
        batch = None
        try:
            batch = seg.ExactBatch(H, W, C, O, count, device=device)
            cps = [pool_images[i % len(pool_images)][0] for i in range(count)]
            sps = [pool_images[i % len(pool_images)][1] for i in range(count)]
            # two launches: the first allocates and sets up the workspaces (what a first call pays), the second is
            # what a service that keeps its workspaces pays per batch -- the figure reported as `value`
            secs = []
            res = None
            for _launch in range(2):
                del res
                torch.cuda.synchronize()
                t = time.perf_counter()
                res = batch.segment(cps, sps, offs, o_exact)
                torch.cuda.synchronize()
                secs.append(time.perf_counter() - t)
            dt = secs[1]
            eq = [equal(mk, tb, s_, goldens[i % len(pool_images)]) for i, (mk, tb, _, s_) in enumerate(res)]
            proofs = sorted({r[3]["proof"] for r in res})
            ws = batch.mergers[0].workspace_bytes()
            batch_out = {"images_per_launch": count, "seconds": round(dt, 3),
                         "value": round(count * H * W / dt / 1e6, 4), "unit": "Mpixel/s",
                         "first_launch_seconds": round(secs[0], 3),
                         "first_launch_value": round(count * H * W / secs[0] / 1e6, 4),
                         "proof": proofs, "tied_conflicts_any": bool(any(r[3]["tied_conflicts"] > 0 for r in res)),
                         "all_equal_reference": bool(all(e for e in eq if e is not None)) if any(e is not None for e in eq) else None,
                         "workspace_bytes_per_image": int(ws),
                         "how": "mn_segment_exact_batch, a workgroup per image in ONE launch, run twice on the same %d "
                                "workspaces: `seconds` / `value` are the second launch (a service keeps its workspaces; "
                                "its results are the ones compared with the reference), `first_launch_*` include "
                                "allocating and setting them up" % count}
            del res
            break
        except Exception as e:                                # noqa: BLE001 -- a side measurement must not fail the line
            batch_out = {"error": repr(e), "images_tried": count}
            count = max(1, int(count * 0.7))
        finally:
            if batch is not None:
                batch.close()
            torch.cuda.empty_cache()
    return {"what": "the library's DEFAULT behaviour on the timed workload's own images (1024x2048, the pool of this "
                    "rank): results are the reference's sequential order (proof 2: nothing left to a tie rule; 3: "
                    "equal priorities popped in creation order where the reference pops by heap position -- these "
                    "maps clip a third of their values to 0.99, so ties abound)",
            "single_image": single, "batch": batch_out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-general-path", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-engine samples at smaller sizes")
    ap.add_argument("--no-default-mode", action="store_true",
                    help="skip the default-mode (proven path) measurement at 1024x2048: one image + one batch")
    ap.add_argument("--default-batch", type=int, default=160,
                    help="most images in the default-mode batch launch (fewer if they do not fit in memory)")
    ap.add_argument("--mode", type=int, default=0, help="0 auto (default), 2 rounds, 3 components")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="drop the per-kernel HIP events inside the library (roofline then reads 0)")
    ap.add_argument("--contexts", type=int, default=0,
                    help="merger contexts in rotation on the one compute stream (launch of image i precedes the "
                         "read-back of image i - contexts + 1); 0 = by run length: 16 for 200 steps and more "
                         "(0.128 against 0.131 ms per step with 8), 6 for short runs, where a deeper ring only "
                         "lengthens the drain at the end (20 steps, three runs each on one box: 0.157 ms with 4, "
                         "0.153 with 6, 0.159 with 8, 0.163 with 12: profiles/r04_ring_depth_sweep.log)")
    ap.add_argument("--spin-seconds", type=float, default=2.5,
                    help="untimed: run the loop this long before the warm-up steps, so that the timed steps see "
                         "the GPU's sustained clocks instead of its idle power state")
    ap.add_argument("--streams", type=int, default=1,
                    help="compute streams the ring's contexts are dealt over (1: every sweep runs alone on the "
                         "one compute stream and its HIP-event duration is the kernel's)")
    ap.add_argument("--replay", action="store_true",
                    help="fixed output buffers per context + hipGraph replay of the launches after the sweep "
                         "(debug_flags bit 5): less host time per image")
    ap.add_argument("--exchange-batch", type=int, default=8,
                    help="N > 1: this many steps' masks share one all-gather (fewer, larger collectives; every "
                         "one of them completes inside the timed region)")
    ap.add_argument("--wire", default="runs", choices=["runs", "int16"],
                    help="wire format of the mask exchange for N > 1 (run-length change points | int16 map)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="images in flight per GPU (contexts + host threads + streams); 1 = serial")
    args = ap.parse_args()
    if args.contexts <= 0:
        args.contexts = 16 if args.steps >= 200 else 6

    import numpy as np
    import torch
    from mergenet_amd import synth
    from mergenet_amd import segmenter as seg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MN_BENCH_REHEARSAL=1 (development only): ranks share the GPUs that exist and talk over gloo,
    # to rehearse the N>1 control flow on a one-GPU box; the real runs are one rank per GPU on RCCL
    rehearsal = os.environ.get("MN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus > 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    offs = synth.generate_offsets(*OFFSETS_ARGS)
    O = len(offs)
    seeds = [1000 + (rank + j) % 8 for j in range(POOL)]
    pool_images = []
    for sd in seeds:
        img = synth.synth_v1(H, W, C, offs, sd)
        pool_images.append((torch.from_numpy(img.class_probs).to(dev), torch.from_numpy(img.sameness_probs).to(dev)))
    # `--pipeline D`: a mergenet_amd.segmenter.MergerPool of D contexts, each driven by its own
    # host thread on its own HIP stream: the merge of one image has a host round trip during which
    # the GPU would idle, and most of its kernels are latency-bound, so D images in flight fill
    # the gaps (4 in flight: 1.8x the images per second).  Results are collected in step order.
    # The default is 1: kernels of concurrent images share the chip, so their
    # HIP-event durations -- and with them the roofline fractions -- would describe the
    # contention, not the kernels; the pipelined rate is reported beside the line (`pipelined`).
    depth = max(1, args.pipeline)
    merger = seg.Merger(H, W, C, O, device=local_rank)
    # contexts in rotation on ONE stream: image i is launched before image i - CONTEXTS + 1 is read back
    mergers_ring = [merger] + [seg.Merger(H, W, C, O, device=local_rank) for _ in range(max(1, args.contexts) - 1)]
    main_pool = seg.MergerPool(H, W, C, O, depth=depth, device=local_rank) if depth > 1 else None
    opts = seg.default_options(same_different_bias=OPTS[0], object_merge_factor=OPTS[1],
                               merge_logprob_bias=OPTS[2], mode=args.mode, require_proof=seg.MN_PROVE_NEVER,
                               debug_flags=(2 if args.no_kernel_events else 0) | int(os.environ.get("MN_BENCH_FLAGS", "0")))
    # HIP events are host work (~3.5 us to record, ~8 us to read): every timed step carries the pair
    # around the sweep (the roofline kernel); one step in EVENTS_EVERY carries all of them (the
    # per-phase report), so that the host does not become the bottleneck of the loop it measures
    opts_lean = seg.default_options(same_different_bias=OPTS[0], object_merge_factor=OPTS[1],
                                    merge_logprob_bias=OPTS[2], mode=args.mode, require_proof=seg.MN_PROVE_NEVER,
                                    debug_flags=(2 if args.no_kernel_events else (48 if args.replay else 16))
                                    | int(os.environ.get("MN_BENCH_FLAGS", "0")))
    ring_out = [(torch.empty((H, W), dtype=torch.int32, device=dev), torch.empty((H * W,), dtype=torch.int32, device=dev))
                for _ in range(max(1, args.contexts))] if args.replay else None
    ring_streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(max(1, args.streams) - 1)]
    from mergenet_amd.distributed import MaskExchange
    # the exchange of step i (int16 wire format, one all-gather) overlaps the merge of step i+1
    ex = MaskExchange(H, W, dev, fmt=args.wire, merger=merger, batch=args.exchange_batch) if world > 1 else None

    from collections import deque

    def run_steps(first, count, pool=None):
        """Steps first .. first+count-1, serially on `merger` or through a MergerPool (a window of
        images in flight, results collected in step order); returns the last result and the sums."""
        last = None
        sums = {k: 0.0 for k in KEYS}
        sums["full_event_steps"] = 0
        modes = set()

        def collect(res):
            nonlocal last
            mask, table, _, st = res
            slot = None
            if ex is not None:                 # exchanges are issued by ONE thread, in step order
                slot = ex.submit(mask, table, st["num_instances"], st["total_logprob"])
            for k in KEYS:
                sums[k] += st[k]
            sums["full_event_steps"] += 1 if st["ms_total"] > 0 else 0
            modes.add(st["mode_used"])
            last = (mask, table, st, slot)

        if pool is None:
            # one stream, a ring of contexts: image i is launched before image i - contexts + 1 is read
            # back, so the host never waits for the tail of an image (a latency chain of ~0.2 ms on the
            # context's side stream) while the sweeps of different images still run one after the
            # other on the one compute stream (their HIP-event durations stay those of the kernels)
            ring = deque()
            for i in range(first, first + count):
                cp, sp = pool_images[i % POOL]
                k = i % len(mergers_ring)
                with torch.cuda.stream(ring_streams[k % len(ring_streams)]):
                    ring.append(mergers_ring[k].segment_async(
                        cp, sp, offs, opts if (i - first) % EVENTS_EVERY == 0 else opts_lean,
                        out=ring_out[k] if ring_out else None))
                if len(ring) >= len(mergers_ring):
                    collect(ring.popleft().result())
            while ring:
                collect(ring.popleft().result())
        else:
            window = deque()
            for i in range(first, first + count):
                cp, sp = pool_images[i % POOL]
                window.append(pool.submit(cp, sp, offs, opts))
                if len(window) >= 4 * len(pool.mergers):
                    collect(window.popleft().result())
            while window:
                collect(window.popleft().result())
        return last, sums, modes

    def fence():
        if ex is not None:
            ex.drain()                 # every all-gather issued so far has completed
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    KEYS = ("ms_class_pass", "ms_edge_pass", "ms_cc_label", "ms_cc_sums", "ms_cc_edges", "ms_cc_cross", "ms_merge",
            "ms_output", "ms_total")
    keys = KEYS
    # initialisation, not a step: one call loads the kernels and sets their attributes, one
    # exchange of an empty mask brings up the communicator's channels for the message size used
    merger.segment(pool_images[0][0], pool_images[0][1], offs, opts)
    if ex is not None:
        zmask = torch.zeros((H, W), dtype=torch.int32, device=dev)
        ex.result(ex.submit(zmask, torch.zeros((1,), dtype=torch.int32, device=dev), 0))
    # ... and the GPU's power state: after the seconds of host work above (map generation, upload) the
    # chip sits at idle clocks, and a few milliseconds of warm-up steps do not bring them up -- the same
    # 2000 timed steps ran at 0.176 ms behind 20 warm-up steps and at 0.141 ms behind 20000 (the
    # bandwidth-bound sweep took 53.9 us either way: it is the latency-bound kernels that follow the
    # core clock).  A service that merges images all day runs at the sustained clocks, so the loop is
    # spun for --spin-seconds before the W warm-up steps; nothing of it is timed.
    if args.spin_seconds > 0:
        # (a step count every rank agrees on: the steps hold collectives when N > 1)
        fence()
        t_probe = time.perf_counter()
        run_steps(0, 32, main_pool)
        fence()
        per_step = (time.perf_counter() - t_probe) / 32
        if world > 1:
            t = torch.tensor([per_step], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            per_step = float(t.item())
        spin_steps = int(min(1 << 16, max(0, args.spin_seconds / max(per_step, 1e-6))))
        if spin_steps:
            run_steps(32, spin_steps, main_pool)
            fence()
    if args.warmup:
        run_steps(0, args.warmup, main_pool)
    fence()
    if ex is not None:
        ex.wait_ms = 0.0
    t0 = time.perf_counter()
    (mask, table, st, gathered), acc, modes = run_steps(args.warmup, args.steps, main_pool)
    fence()
    elapsed = time.perf_counter() - t0
    exchange_wait_ms = ex.wait_ms if ex is not None else 0.0

    # the same steps with PIPELINED_DEPTH images in flight (one GPU; outside the contract line)
    pipelined = None
    if world == 1 and not args.no_pipelined and depth == 1:
        side_pool = seg.MergerPool(H, W, C, O, depth=PIPELINED_DEPTH, device=local_rank)
        run_steps(0, 2 * PIPELINED_DEPTH, side_pool)
        fence()
        tp = time.perf_counter()
        (pm, pt, pst, _), _, _ = run_steps(args.warmup, args.steps, side_pool)
        fence()
        dtp = time.perf_counter() - tp
        side_pool.close()
        pipelined = {"depth": PIPELINED_DEPTH, "value": round(args.steps * H * W / dtp / 1e6, 2),
                     "unit": "Mpixel/s", "ms_per_step": round(dtp / args.steps * 1e3, 4),
                     "last_mask_equals_serial_run": bool(torch.equal(pm, mask)),
                     "how": "mergenet_amd.segmenter.MergerPool: %d contexts, host threads and HIP streams; "
                            "same images; kernels of concurrent images share the chip, so per-kernel "
                            "durations roughly double (round 1's way of keeping the chip busy; since round 2 "
                            "the ring of contexts on one stream above is faster)" % PIPELINED_DEPTH}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    nfull = max(1, acc["full_event_steps"])
    avg = {k: acc[k] / (args.steps if k == "ms_cc_edges" else nfull) for k in keys}
    if avg["ms_cc_edges"] == 0 and avg["ms_edge_pass"] > 0:      # (general path: the sweep is reported as the edge pass)
        pass

    # what the last exchange delivered: every rank's own slice must be its own mask and table
    exchange_ok = None
    if ex is not None:
        masks_w, tabs_w, counts_w = ex.result(gathered)
        k = st["num_instances"]
        good = bool((masks_w[rank].to(torch.int32) == mask).all()) and int(counts_w[rank]) == k
        good &= bool((tabs_w[rank, :k].to(torch.int32) == table[:k]).all())
        good &= bool((counts_w > 0).all()) and float(ex.logprobs(gathered)[rank]) == st["total_logprob"]
        t = torch.tensor([int(good)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        exchange_ok = bool(t.item())

    # instance-id match of EVERY image of every rank's pool against the reference's own result for
    # that image (golden vectors tests/golden/cseg_synth_1024x2048_*.npz, produced by the
    # reference's segment.cc in 380-540 s per image); outside the timed region
    from mergenet_amd import labels as ck      # plain numpy label-map comparison
    checked, equal = 0, 0
    for j, sd in enumerate(seeds):
        gname = "cseg_synth_1024x2048_cfg2.npz" if sd == 1000 else "cseg_synth_1024x2048_s%d.npz" % sd
        golden = os.path.join(ROOT, "tests", "golden", gname)
        if not os.path.exists(golden):
            continue
        z = np.load(golden)
        m_j, t_j, _, st_j = merger.segment(pool_images[j][0], pool_images[j][1], offs, opts)
        got_cls = [int(c) for c in t_j.cpu().numpy()[: st_j["num_instances"]]]
        checked += 1
        equal += int(ck.masks_equivalent(m_j.cpu().numpy(), got_cls, z["mask"],
                                         [int(c) for c in z["object_class"]]))
    if world > 1:
        t = torch.tensor([checked, equal], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        checked, equal = int(t[0].item()), int(t[1].item())
    id_match = {"vs": "reference segment.cc (golden vectors, seeds 1000..1007)",
                "images_checked": checked, "images_equal": equal, "equal": bool(checked and checked == equal)}

    # the general path (rounds) on the same images, outside the timed region: its affinity-scoring
    # pass (class pass + edge pass) is the kernel the north star names
    general = None
    if rank == 0 and not args.no_general_path:
        ropts = seg.default_options(same_different_bias=OPTS[0], object_merge_factor=OPTS[1],
                                    merge_logprob_bias=OPTS[2], mode=seg.MN_MODE_ROUNDS)
        g = {"ms_class_pass": 0.0, "ms_edge_pass": 0.0, "ms_total": 0.0}
        merger.segment(pool_images[0][0], pool_images[0][1], offs, ropts)
        for j in range(POOL):
            _, _, _, st_r = merger.segment(pool_images[j][0], pool_images[j][1], offs, ropts)
            for k in g:
                g[k] += st_r[k] / POOL
        general = g

    # the exact engine (the reference's sequential order itself, MN_MODE_EXACT) on a bounded sample,
    # outside the timed region: what the library's default AUTO falls back to when the fast path's
    # answer is not certified
    exact = None
    if rank == 0 and world == 1 and not args.no_exact:
        exact = exact_engine_sample(seg, synth, offs, local_rank)

    # the proven path (library defaults) on this rank's own pool, outside the timed region; the ring's
    # contexts are released first (the batch takes what memory there is)
    default_mode = None
    if not args.no_default_mode:
        for mg in mergers_ring[1:]:
            mg.close()
        default_mode = default_mode_block(seg, pool_images, seeds, offs, local_rank, args.default_batch)
        if world > 1:
            # whole-job figures: the ranks' batches run side by side
            b = default_mode.get("batch") or {}
            t = torch.tensor([float(b.get("seconds", 0.0)), float(b.get("images_per_launch", 0)),
                              float(default_mode["single_image"]["seconds"]),
                              1.0 if b.get("all_equal_reference") else 0.0], dtype=torch.float64, device=dev)
            tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone(); dist.all_reduce(tsum)
            default_mode["all_ranks"] = {
                "batch_images": int(tsum[1].item()), "batch_seconds_max": round(float(tmax[0].item()), 3),
                "batch_value": round(float(tsum[1].item()) * H * W / max(float(tmax[0].item()), 1e-9) / 1e6, 4),
                "unit": "Mpixel/s", "single_image_seconds_max": round(float(tmax[2].item()), 3),
                "ranks_all_equal_reference": int(tsum[3].item())}

    if rank == 0:
        plane_bytes = 4.0 * H * W
        pmc = {}
        pmc_path = ""
        for rnd in ("r04", "r03", "r02"):            # the latest round's counter passes (tools/pmc_kernel.sh)
            pmc_path = os.path.join(ROOT, "profiles", "%s_pmc_components_1024x2048.json" % rnd)
            if os.path.exists(pmc_path):
                break
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh).get("hbm_bytes_per_launch", {})

        def price(name, ms, planes, what):
            gbs = planes * plane_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "reads": what, "algorithmic_bytes": planes * plane_bytes,
                    "avg_launch_ms": round(ms, 5), "achieved": round(gbs, 2),
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc.get(name),
                    "traffic_source": "profiles/%s (separate --pmc passes; NOT measured in this run)"
                                      % os.path.basename(pmc_path) if pmc.get(name) else None}

        if avg["ms_cc_sums"] > 0:      # components mode
            # the affinity-scoring sweep (mn_cc_sign, class-plane form) reads EVERY input plane once:
            # 4 * (C + O) bytes per pixel; everything after it works on what
            # it leaves (4 B/pixel edge masks, 9 B/pixel class log-products, the negative-edge list)
            passes = [price("mn_cc_sign", avg["ms_cc_edges"], C + O,
                            "C class planes + O sameness planes (the only read of the input tensors)"),
                      price("mn_cc_tiles+mn_cc_borders+mn_cc_flatten+mn_cc_hook", avg["ms_cc_label"], 1,
                            "the 4 B/pixel edge masks (union-find: latency-bound, no roofline claim)"),
                      price("mn_cc_sums", avg["ms_cc_sums"], (C + 5) / 4.0,
                            "per-lane class log-products + roots + arg-max classes, C + 5 B/pixel "
                            "(dependent round trips, no roofline claim; on the side stream)")]
        elif avg["ms_class_pass"] == 0:   # general path starting from the cores: the same single sweep
            passes = [price("mn_cc_sign", avg["ms_edge_pass"], C + O,
                            "C class planes + O sameness planes (the only read of the input tensors)")]
        else:                          # general path on the pixel graph: class pass + edge pass, SURVEY 8d
            passes = [price("mn_class_pass", avg["ms_class_pass"], C, "C class planes"),
                      price("mn_edge_pass_fast", avg["ms_edge_pass"], O, "O sameness planes")]
        streaming = [p for p in passes if "no roofline claim" not in p["reads"]]
        dom = max(streaming, key=lambda p: p["avg_launch_ms"])
        roofline = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": dom["traffic"],
                    "traffic_source": dom["traffic_source"],
                    "algorithmic_bytes": dom["algorithmic_bytes"], "avg_launch_ms": dom["avg_launch_ms"],
                    "timing": "hipEvent pair attached to the kernel's own dispatch on the launch stream "
                              "(hipExtLaunchKernel start/stop events) in EVERY timed step; rocprofv3 "
                              "durations of the same loop are in profiles/r03_bench_kernel_stats.csv; "
                              "the other phases' events are recorded "
                              "in one step of %d, because events are host work and the host must not "
                              "become the bottleneck of the loop" % EVENTS_EVERY,
                    "why_this_kernel": "the one HBM-streaming kernel of the timed path: it reads every input "
                                       "plane (class + sameness) exactly once"}
        out = {
            "metric": "merged Mpixels/sec at 1024x2048 (speculative fast path; proven path: value_proven_path)",
            "value": round(world * args.steps * H * W / elapsed / 1e6, 4),
            "unit": "Mpixel/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: one 1024x2048 Cityscapes-shape class+offset map per "
                                   "GPU and step (C=9, O=10 generate_offsets(40,10), opts 0/1/0.03, "
                                   "variant csegment, synth-v1 seeds 1000..1007, %d images per rank "
                                   "in rotation)" % POOL,
                       "images_per_step": world, "H": H, "W": W, "C": C, "O": O,
                       "pipeline_depth": depth,
                       "untimed_spin_up_s": args.spin_seconds,
                       "host_overlap": "launch of step i queued before the read-back of step i - %d "
                                       "(mn_segment_launch / mn_segment_finish, %d contexts in rotation on "
                                       "ONE compute stream: the sweeps of different images run one after the "
                                       "other; the latency-bound tail of an image -- class sums from the "
                                       "sweep's products, records, second phase, labels, mask -- runs on its "
                                       "context's side stream beside the next images' sweeps)"
                                       % (len(mergers_ring) - 1, len(mergers_ring)),
                       "mode": {0: "auto", 1: "exact", 2: "rounds", 3: "components"}.get(args.mode),
                       "proof": "speculative fast path (require_proof = -1): on these images its answer is not "
                                "certified -- equality with the reference's own outputs is measured on every image "
                                "of the pool in this run (id_match); the library's default AUTO redoes an "
                                "uncertified image in the exact engine (exact_engine)",
                       "mode_used": sorted({1: "exact", 2: "rounds", 3: "components"}.get(m, m) for m in modes),
                       "exchange": ("one all_gather per step of the masks (%s wire) + class tables + "
                                    "log-likelihoods, overlapped with the next step's merge; delivered "
                                    "data checked: %s" % (args.wire, exchange_ok)) if world > 1
                       else "none (single GPU)"},
            "distributed": {"world_size": (dist.get_world_size() if world > 1 else 1),
                            "backend": (ex.backend if ex is not None else "none"),
                            "wire_format": (args.wire if world > 1 else None),
                            "wire_bytes_per_rank_per_step": (ex.bytes_per_rank if ex is not None else 0),
                            "steps_per_collective": (ex.batch if ex is not None else None),
                            "exchange_wait_ms_per_step": round(exchange_wait_ms / max(1, args.steps), 5),
                            "delivered_data_checked": exchange_ok},
            "roofline": roofline,
            "passes": passes,
            "phases_ms": {"score": round(avg["ms_class_pass"] + avg["ms_edge_pass"], 4),
                          "merge": round(avg["ms_merge"], 4),
                          "output": round(avg["ms_output"], 4),
                          "device_total": round(avg["ms_total"], 4)},
            "merge_stats": {"rounds": st["rounds"], "finisher_steps": st["finisher_steps"],
                            "merges": st["merges"], "certified": st["certified"],
                            "instances": st["num_instances"]},
            "id_match": id_match,
        }
        if pipelined is not None:
            out["pipelined"] = pipelined
        if general is not None:
            sc_ms = general["ms_class_pass"] + general["ms_edge_pass"]
            sc = (C + O) * plane_bytes / (sc_ms * 1e-3) / 1e9
            from_cores = general["ms_class_pass"] == 0      # the general rounds start from the cores: same sweep
            sc_traffic = pmc.get("mn_cc_sign") if from_cores else None
            out["scoring_pass_general_path"] = {
                "bound": "hbm",
                "kernel": "mn_cc_sign (mode rounds: the same single sweep, then cores, clusters, rounds, finisher)"
                          if from_cores else "mn_class_pass + mn_edge_pass_fast (mode rounds from single pixels)",
                "achieved": round(sc, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(sc / HBM_PEAK_GBS, 4), "traffic": sc_traffic,
                "algorithmic_bytes": (C + O) * plane_bytes, "avg_launch_ms": round(sc_ms, 5),
                "whole_image_ms_rounds_mode": round(general["ms_total"], 3),
                "note": "measured live with the same HIP events, outside the timed steps"}
        if default_mode is not None:
            out["default_mode"] = default_mode
            sb = default_mode.get("all_ranks") or default_mode.get("batch") or {}
            out["value_proven_path"] = sb.get("batch_value", sb.get("value"))
            out["value_is"] = ("the SPECULATIVE fast path (require_proof = -1; equality with the reference measured, "
                               "not proven: id_match); the same images through the library's default (proven) mode: "
                               "default_mode / value_proven_path")
        if exact is not None:
            out["exact_engine"] = exact
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
