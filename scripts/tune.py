#!/usr/bin/env python3
"""Sweep-kernel tuning grid on one GPU: variant x n_chunk x block at a given shape.
Prints ms per sweep launch and the algorithmic GB/s (28 B per point-sweep).
Variants 97, 98, 99 (the same kernels with the arithmetic removed: their memory floors)
exist only in a library built with `make EXTRA=-DSLB_TIMING_VARIANTS`; elsewhere those
numbers select the default kernel."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-harmonics", type=int, default=128)
    ap.add_argument("--g-grid", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--variants", default="2,3")
    ap.add_argument("--chunks", default="0,8,16,32,64,128")
    ap.add_argument("--blocks", default="256,128,64")
    ap.add_argument("--fused", action="store_true", help="time the one-launch step (variants 6, 7)")
    ap.add_argument("--double", action="store_true", help="time the double step (variants 21, 23, 25)")
    ap.add_argument("--multi", type=int, default=0, help="time the S-steps-per-launch kernel (S = 1..4)")
    args = ap.parse_args()
    import torch
    slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
    torch.zeros(1, device="cuda")
    N, M = args.n_harmonics, args.g_grid
    slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    L = slb.lib()
    sol = slb.Solver()
    sol.tiptoe()
    sol.set_fused(10 + args.multi if args.multi else (2 if args.double else args.fused))
    per_launch = 1.0 / args.multi if args.multi else (0.5 if args.double else (1 if args.fused else 2))
    bytes_per_point = 36.0 if (args.fused or args.double or args.multi) else 28.0
    if args.double or args.multi:
        for _ in range(12 // (args.multi or 2)):   # spares made, ping-pong parity back to the start
            sol.multi_step(args.multi or 2)
    if args.multi:
        args.steps = args.steps // 12 * 12 or 12
    for v in [int(x) for x in args.variants.split(",")]:
        for nc in [int(x) for x in args.chunks.split(",")]:
            for b in [int(x) for x in args.blocks.split(",")]:
                L.slb_set_tuning(v, nc, b)
                sol.run_timed(3)
                ms = min(sol.run_timed(args.steps) for _ in range(3)) / (per_launch * args.steps)
                gbs = bytes_per_point * N * (M + 0.5) / (ms * 1e-3) / 1e9
                ups = N * M / (ms * per_launch * 1e-3)
                print(f"variant={v} n_chunk={nc:4d} block={b:4d}  {ms:8.4f} ms/launch  {gbs:8.1f} GB/s  "
                      f"{gbs / 80.0:5.1f}% of 8 TB/s  {ups:.3e} updates/s", flush=True)
    sol.close()


if __name__ == "__main__":
    main()
