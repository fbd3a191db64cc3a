#!/usr/bin/env python3
"""Microbenchmark of the single-op convolution entry point (dmme_conv2d) on UNet layer shapes.
usage: python tools/bench_conv.py [--shapes 32x128x128,16x256x256] [--batch 128] [--variants plain,gn]
Prints per shape and variant: average launch time (HIP events on the launch stream) and bf16 TFLOP/s."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="32x128x128,32x256x128,16x256x256,16x512x256,8x256x256,8x512x256,4x256x256")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--variants", default="plain,gn")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--taps", type=int, default=9)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = _lib.BF16
    lib = _lib.lib()
    B = args.batch
    for shp in args.shapes.split(","):
        hw, cin, cout = (int(v) for v in shp.split("x"))
        x = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(cout, args.taps, cin, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.randn(cout, device=dev)
        scale = torch.rand(B, cin, device=dev) + 0.5
        shift = torch.randn(B, cin, device=dev) * 0.1
        out = torch.empty(B, hw, hw, cout, device=dev, dtype=torch.bfloat16)
        d = _lib.ConvDesc()
        d.dtype, d.N, d.Hin, d.Win, d.C1, d.C2 = dt, B, hw, hw, cin, 0
        d.upsample, d.stride, d.taps, d.Cout = 0, 1, args.taps, cout
        d.out_silu = d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = d.force_generic = 0
        flops = 2.0 * B * hw * hw * cin * cout * args.taps
        for var in args.variants.split(","):
            d.pro_silu = 1 if var == "gn" else 0  # "aff": GroupNorm affine without SiLU
            sc, sh = (scale, shift) if var in ("gn", "aff") else (None, None)
            st = _lib.stream_ptr()

            def run():
                _lib.check(lib.dmme_conv2d(C.byref(d), _lib.ptr(x), None, _lib.ptr(w), _lib.ptr(b), _lib.ptr(sc), _lib.ptr(sh), None, None, None,
                                           None, cout, _lib.ptr(out), st), "dmme_conv2d")

            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            print(f"{shp:14s} taps={args.taps} {var:6s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
