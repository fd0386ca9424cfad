#!/usr/bin/env python3
"""What the C slab driver and RCCL cost per launch of two and of three steps, measured on ONE GPU: a slab of
2^20 columns (n-harmonics=128) stepped (a) as a lattice of its own by the plain engine,
(b) by the slab loop with one rank (no neighbours), (c) as the middle slab of three with both
neighbours' messages sent to the process itself over RCCL (edge launches, pack, grouped
ncclSend/ncclRecv, unpack on the side stream beside the interior launch).  (c)'s ghost columns
hold the slab's own edges, so its lattice is not physical; its launches and messages are those
of a real run."""
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    L = slb.lib()
    N, W, pairs = 128, 1 << 20, 60

    def configure(M):
        slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                      PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)

    def timed(step, n):
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        return best * 1e3

    for S in (2, 3):
        print(f"## {S} steps per launch", flush=True)
        configure(W)
        L.slb_set_stream(None)
        sol = slb.Solver()
        sol.tiptoe()
        print(f"plain engine, one lattice of 2^20 columns:          {timed(lambda: sol.multi_step(S) if S == 3 else sol.double_step(), pairs):.4f} ms per launch", flush=True)
        sol.close()

        r = L.slb_slabs_create(0, 1, None)
        L.slb_slabs_tiptoe(r)
        print(f"slab loop in C, one rank:                           {timed(lambda: L.slb_slabs_multi_step(r, S, None), pairs):.4f} ms", flush=True)
        L.slb_slabs_destroy(r)

        configure(3 * W)
        ident = C.create_string_buffer(128)
        assert L.slb_link_unique_id(ident) == 0
        link = L.slb_link_open_rccl(0, 1, ident.raw)
        r = L.slb_slabs_create_loopback(1, 3, link)
        L.slb_slabs_tiptoe(r)
        print(f"slab loop in C, middle slab, RCCL messages to self: {timed(lambda: L.slb_slabs_multi_step(r, S, None), pairs):.4f} ms", flush=True)
        L.slb_slabs_destroy(r)
        L.slb_link_close(link)


if __name__ == "__main__":
    main()
