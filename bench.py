#!/usr/bin/env python
"""bench.py -- merged Mpixels/s of the HIP pixel merger on synthetic 1024x2048 maps.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: every rank merges ONE
1024x2048 Cityscapes-shape image (C=9 classes, O=10 log-spiral offsets, options 0/1/0.03 of
egs/cityscape/local/segment.py:134-136) whose probability maps are already resident in HBM, and --
for N > 1 -- the final int32 masks and class tables are all-gathered over RCCL (the only exchange
step of the path; images are independent).  Weak scaling: per-GPU work is fixed.

Rank 0 prints ONE JSON line.  `value` = pixels merged by all ranks / wall time of the K timed
steps (max over ranks), in Mpixel/s.  `roofline` prices the affinity-scoring pass (class pass +
edge pass) with HIP events taken on the launch stream inside the library: algorithmic bytes =
4*(C+O) B/pixel.  `cpu_baseline` times the reference's own compiled
segment.cc (oracle/_ref, kind "reference") or, if that is absent, our C++ restatement (kind
"port") on one core over a bounded 256x512 sample of the same generator.
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
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def cpu_baseline():
    """Bounded CPU sample (about 10-20 s): one 256x512 image of the same generator, 1 core."""
    from mergenet_amd import synth
    from oracle import checker as ck
    offs = synth.generate_offsets(*OFFSETS_ARGS)
    s = synth.synth_v1(256, 512, C, offs, 1000)
    if ck.have_reference():
        kind, fn = "reference", ck.run_reference
    else:
        kind, fn = "port", ck.run_csegment
    t = time.perf_counter()
    fn(s.class_probs, s.sameness_probs, C, offs, *OPTS)
    dt = time.perf_counter() - t
    return {"value": round(256 * 512 / dt / 1e6, 6), "unit": "Mpixel/s", "cores": 1, "kind": kind,
            "sample": "one 256x512 synth-v1 image (seed 1000, C=9, O=10, opts 0/1/0.03): "
                      "%.1f s on one host core; the reference is single-threaded and "
                      "super-linear in pixels (435 s for 1024x2048 in the build container)" % dt,
            "seconds": round(dt, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from mergenet_amd import synth
    from mergenet_amd import segmenter as seg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus > 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    offs = synth.generate_offsets(*OFFSETS_ARGS)
    img = synth.synth_v1(H, W, C, offs, 1000 + rank)
    cp = torch.from_numpy(img.class_probs).to(dev)
    sp = torch.from_numpy(img.sameness_probs).to(dev)
    merger = seg.Merger(H, W, C, len(offs), device=local_rank)
    opts = seg.default_options(same_different_bias=OPTS[0], object_merge_factor=OPTS[1],
                               merge_logprob_bias=OPTS[2], mode=seg.MN_MODE_ROUNDS)
    from mergenet_amd.distributed import gather_masks

    def step():
        mask, table, _, st = merger.segment(cp, sp, offs, opts)
        gathered = gather_masks(mask, table, st["num_instances"]) if world > 1 else None
        return mask, table, st, gathered

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    score_ms, class_ms, edge_ms, merge_ms, out_ms = [], [], [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mask, table, st, gathered = step()
        score_ms.append(st["ms_score"]); class_ms.append(st["ms_class_pass"])
        edge_ms.append(st["ms_edge_pass"]); merge_ms.append(st["ms_merge"]); out_ms.append(st["ms_output"])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # instance-id match of EVERY rank's image against the reference's own result for that image
    # (golden vectors tests/golden/cseg_synth_1024x2048_*.npz, produced by the reference's
    # segment.cc in 380-540 s per image); outside the timed region
    from mergenet_amd import labels as ck      # plain numpy label-map comparison
    gname = "cseg_synth_1024x2048_cfg2.npz" if rank == 0 else "cseg_synth_1024x2048_s%d.npz" % (1000 + rank)
    golden = os.path.join(ROOT, "tests", "golden", gname)
    checked, equal = 0, 0
    if os.path.exists(golden):
        z = np.load(golden)
        got_cls = [int(c) for c in table.cpu().numpy()[: st["num_instances"]]]
        checked = 1
        equal = int(ck.masks_equivalent(mask.cpu().numpy(), got_cls, z["mask"],
                                        [int(c) for c in z["object_class"]]))
    if world > 1:
        t = torch.tensor([checked, equal], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        checked, equal = int(t[0].item()), int(t[1].item())
    id_match = {"vs": "reference segment.cc (golden vectors, seeds 1000+rank)",
                "images_checked": checked, "images_equal": equal, "equal": bool(checked and checked == equal)}

    if rank == 0:
        avg_score_ms = sum(score_ms) / len(score_ms)
        algo_bytes = 4.0 * (C + len(offs)) * H * W
        achieved = algo_bytes / (avg_score_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, FETCH_SIZE x2 on gfx950): collected once per round, kept in profiles/
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_score_1024x2048.json")
        if os.path.exists(pmc):
            with open(pmc) as fh:
                traffic = json.load(fh).get("affinity_scoring_pass_hbm_bytes_per_launch")
        out = {
            "metric": "merged Mpixels/sec at 1024x2048",
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
                                   "GPU (C=9, O=10 generate_offsets(40,10), opts 0/1/0.03, "
                                   "variant csegment, synth-v1 seeds 1000+rank)",
                       "images_per_step": world, "H": H, "W": W, "C": C, "O": len(offs),
                       "exchange": "all_gather of int32 masks + class tables over RCCL" if world > 1
                                   else "none (single GPU)"},
            "roofline": {"bound": "hbm", "kernel": "affinity-scoring pass = mn_class_pass + mn_edge_pass_fast<10,true>",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes": algo_bytes, "avg_launch_ms": round(avg_score_ms, 5),
                         "class_pass_ms": round(sum(class_ms) / len(class_ms), 5),
                         "edge_pass_ms": round(sum(edge_ms) / len(edge_ms), 5),
                         "timing": "hipEvent pairs on the launch stream around each kernel (includes "
                                   "the ~5 us dispatch gap per kernel; rocprofv3 kernel-only "
                                   "durations are in profiles/r01_bench_kernel_stats.csv)"},
            "phases_ms": {"score": round(avg_score_ms, 4),
                          "merge": round(sum(merge_ms) / len(merge_ms), 4),
                          "output": round(sum(out_ms) / len(out_ms), 4)},
            "merge_stats": {"rounds": st["rounds"], "finisher_steps": st["finisher_steps"],
                            "merges": st["merges"], "certified": st["certified"],
                            "instances": st["num_instances"]},
            "id_match": id_match,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
