#!/usr/bin/env python3
"""Whole-image ZeST evaluation (DyMVSNeRF_G.forward_val) at the NSFF geometry: 288 x 512 pixels,
128 samples, 3 source views + 3 neighbour frames, random weights, chunks of --chunk rays.
Reports the split between the two volume builders, ray sampling + rendering, and pixels/s.

    python tools/bench_image.py [--chunk 1024] [--precision 16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench  # noqa: F401  (sets sys.path for the package)
import zest_networks as networks
import test_generators as tg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--precision", type=int, default=16)
    ap.add_argument("--images", type=int, default=3)
    ap.add_argument("--val-rays", type=int, default=None, help="args.zest_val_rays: rays per rendering launch of forward_val")
    ap.add_argument("--serial-builders", action="store_true", help="build the two encoding volumes one after the other")
    ap.add_argument("--no-conv-autotune", action="store_true",
                    help="leave torch.backends.cudnn.benchmark off (the reference's Trainer sets benchmark=True, train.py:1331: "
                         "MIOpen then times its convolution solvers on first use)")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = not a.no_conv_autotune
    H, W = 288, 512
    x = tg._batch(7, H=H, W=W)
    args = tg._args(chunk=a.chunk, precision=a.precision, N_samples=128, pad=24, batch_size=a.chunk,
                    zest_overlap_builders=not a.serial_builders,
                    **({} if a.val_rays is None else {"zest_val_rays": a.val_rays}))
    gen = tg._generator(args)

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()
    with torch.no_grad():
        gen.forward_val(x)                                  # warm-up (MIOpen picks its algorithms)
        t0 = sync()
        for _ in range(a.images):
            res = gen.forward_val(x)
        t1 = sync()
        for _ in range(a.images):
            sc = gen._scene(x, bn_batch_stats=True)
        t2 = sync()
    per_img, per_vol = (t1 - t0) / a.images, (t2 - t1) / a.images
    rgb = torch.cat(res[1])
    print(json.dumps({"op": "forward_val, %dx%d, %d samples, chunk %d, precision %d" % (H, W, 128, a.chunk, a.precision),
                      "conv_autotune": bool(torch.backends.cudnn.benchmark),
                      "ms_per_image": round(per_img * 1e3, 2), "ms_volume_builders": round(per_vol * 1e3, 2),
                      "ms_rays_and_render": round((per_img - per_vol) * 1e3, 2),
                      "pixels_per_s": round(H * W / per_img), "finite": bool(torch.isfinite(rgb).all())}))


if __name__ == "__main__":
    main()
