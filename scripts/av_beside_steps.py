#!/usr/bin/env python3
"""What the averaging period costs the three-steps-per-launch loop at n-harmonics=128, g-grid=2^20 on one
GPU, piece by piece: the launch without and with the rows av() reads, the exact-order sums of six lattices
alone, and both at once (sums on a second stream, one launch of them per two step launches, as the program
issues them)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
    torch.zeros(1, device="cuda")
    N, M = 128, 1 << 20
    slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    L = slb.lib()
    main_stream = torch.cuda.Stream()
    side = torch.cuda.Stream()
    torch.cuda.set_stream(main_stream)
    L.slb_set_stream(main_stream.cuda_stream)
    sol = slb.Solver()
    sol.tiptoe()
    sol.set_fused(3)
    sol.double_step()
    for _ in range(6):
        sol.multi_step(3)
    pitch = sol.s.pitch
    rows = [torch.zeros((3, pitch), dtype=torch.float32, device="cuda") for _ in range(6)]
    tot = [torch.zeros(3, dtype=torch.float32, device="cuda") for _ in range(6)]
    slab = slb.Slab(pitch, 0, 1, M)
    a_ptrs = (C.c_void_p * 6)(*[r.data_ptr() for r in rows])
    b_ptrs = (C.c_void_p * 6)(*[r.data_ptr() + 4 * pitch for r in rows])
    t_ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in tot])

    def sums(n=6):
        L.slb_set_stream(side.cuda_stream)
        assert L.slb_av_sums_exact_n(C.byref(slab), n, a_ptrs, b_ptrs, t_ptrs) == 0
        L.slb_set_stream(main_stream.cuda_stream)

    def timed(fn, n, stream=main_stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for i in range(n):
            fn(i)
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    plain = lambda i: sol.multi_step(3)
    with_rows = lambda i: sol.multi_step_rows(3, [rows[3 * (i % 2) + k].data_ptr() for k in range(3)])

    def both(i):
        with_rows(i)
        if i % 2 == 1:
            side.wait_stream(main_stream)       # the sums read the rows of the last two launches
            sums()

    for _ in range(3):
        timed(plain, 10); timed(with_rows, 10)
    for _ in range(4):
        print(f"plain {timed(plain, 40):.4f} ms   with rows {timed(with_rows, 40):.4f} ms   "
              f"with rows and the sums beside {timed(both, 40):.4f} ms per launch of three steps", flush=True)
    for n in (1, 3, 6, 8 if len(rows) >= 8 else 6):
        print(f"exact-order sums of {n} lattices alone: {timed(lambda i: sums(n), 10, side):.4f} ms per launch", flush=True)
    sol.close()


if __name__ == "__main__":
    main()
