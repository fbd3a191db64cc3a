#!/usr/bin/env python3
"""What one CU can draw from L2: every workgroup streams the same L2-resident buffer (dmme_debug_l2_stream).
usage: python tools/l2_stream.py   -> bytes/clk/CU (at 2.4 GHz) and aggregate TB/s per (buffer size, loads in flight, mode, workgroups)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
sink = torch.zeros(4096, dtype=torch.int32, device=dev)
for kib in (256, 1024, 16384):
    buf = torch.randint(0, 2**31 - 1, (kib * 256,), dtype=torch.int32, device=dev)
    for mode in (0, 1, 288):  # registers, LDS-DMA contiguous, LDS-DMA gathering 128-byte rows 4608 bytes apart (a 3x3 filter row stride)
        for depth in (1, 4, 16):
            for blocks in (256, 512, 1024):
                iters = max(1, (64 << 20) // (kib << 10))
                st = _lib.stream_ptr()

                def run():
                    _lib.check(lib.dmme_debug_l2_stream(_lib.ptr(buf), kib << 10, iters, mode, depth, blocks, _lib.ptr(sink), st))

                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                sec = e0.elapsed_time(e1) * 1e-3
                per_wg = iters * (kib << 10)
                total = per_wg * blocks
                wg_per_cu = max(1, blocks // 256)
                print(f"buffer {kib:6d} KiB  mode {'reg' if mode == 0 else 'dma' if mode == 1 else 'dma-gather'}  depth {depth:2d}  workgroups {blocks:5d}: {total / sec / 1e12:6.2f} TB/s aggregate, "
                      f"{total / 256 / sec / 2.4e9:6.1f} B/clk per CU ({wg_per_cu} workgroups per CU)", flush=True)
