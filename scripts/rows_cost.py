#!/usr/bin/env python3
"""What the double step pays for also writing the three rows av() reads of both of its
iterations (slb_solver_double_step_rows) on one GPU: the two kernels back to back, alternating."""
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
    sol = slb.Solver()
    sol.tiptoe()
    sol.set_fused(2)
    for _ in range(6):
        sol.double_step()
    rows = [torch.zeros((3, sol.s.pitch), dtype=torch.float32, device="cuda") for _ in range(2)]

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    plain = lambda: sol.double_step()
    with_rows = lambda: sol.double_step_rows(rows[0].data_ptr(), rows[1].data_ptr())
    timed(plain, 10); timed(with_rows, 10)
    for _ in range(4):
        print(f"plain {timed(plain, 40):.4f} ms   with rows {timed(with_rows, 40):.4f} ms per double step", flush=True)
    sol.close()


if __name__ == "__main__":
    main()
