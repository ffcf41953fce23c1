#!/usr/bin/env python
"""The complete stage-1 schedule of the shipped config (pre_train_mapping 100 x F steps, 10 001 loop iterations, final
evaluation render + PSNR; reference: src/stage1_neural_atlas.py:137-251 / stage1_neural_atlas_seg.py:173-324) on a
synthetic video resident in HBM, timed end to end.  Prints one JSON line.

    python tools/full_run.py [--two-layer] [--frames 80 --resx 768 --resy 432] [--iters 10001]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--two-layer", action="store_true")
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--resx", type=int, default=768)
    ap.add_argument("--resy", type=int, default=432)
    ap.add_argument("--iters", type=int, default=10001)
    ap.add_argument("--pretrain-iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    import aiod_amd
    import bench
    dev = torch.device("cuda", 0)
    cfg = aiod_amd.default_config(args.resx, args.resy, args.frames, two_layer=args.two_layer)
    af = aiod_amd.AtlasFit(cfg)
    video = bench.synth_video_device(args.resx, args.resy, args.frames, seed=args.seed, device=dev)
    if args.two_layer:
        video = video + (bench.synth_fg_mask_device(args.resx, args.resy, args.frames, seed=args.seed, device=dev),)
    af.upload_video(*video)
    sds = bench.init_state_dicts(1234, args.two_layer)
    for net in af.nets:
        af.load_state_dict(net, sds[net])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    af.pre_train_mapping(args.pretrain_iters, seed=1)
    if args.two_layer:
        af.pre_train_mapping(args.pretrain_iters, seed=2, net=aiod_amd.NET_MAPPING2)
    af.sync(); t1 = time.perf_counter()
    p0, _ = af.psnr()
    af.sync(); t1b = time.perf_counter()
    block = 1000
    first_losses = last_losses = None
    i = 0
    while i < args.iters:
        n = min(block, args.iters - i)
        l = af.train_steps(i, n, None, seed=3)
        first_losses = l[0] if first_losses is None else first_losses
        last_losses = l[-1]
        i += n
    af.sync(); t2 = time.perf_counter()
    p1, per = af.psnr()
    af.sync(); t3 = time.perf_counter()
    N = cfg.samples_batch
    tot_col = 11 if args.two_layer else 5
    print(json.dumps({
        "workload": "%s, %d frames %dx%d, samples_batch %d, pretrain %d x F, %d iterations" % ("two-layer" if args.two_layer else "single atlas", args.frames, args.resx, args.resy, N, args.pretrain_iters, args.iters),
        "pretrain_s": t1 - t0, "loop_s": t2 - t1b, "render_psnr_s": t3 - t2, "total_s": (t1 - t0) + (t2 - t1b) + (t3 - t2),
        "loop_points_per_s": N * args.iters / (t2 - t1b),
        "psnr_after_pretrain_db": p0, "psnr_final_db": p1, "psnr_min_frame_db": float(per.min()),
        "total_loss_first": float(first_losses[tot_col]), "total_loss_last": float(last_losses[tot_col]),
    }))
    af.close()


if __name__ == "__main__":
    main()
