#!/usr/bin/env python3
"""Long full-lattice parity run at the headline shape: n-harmonics=128, g-grid=2^20, `--steps`
loop iterations taken on the GPU the way the program takes them (--form auto: three per launch on a
lattice this wide, a remainder as a double and a single step; --form double: two per launch) and as
single steps by the CPU oracle (OpenMP), then every cell of the current lattices compared bit for
bit, and av() of the final lattice.
A one-off check (about a minute of CPU per 100 steps), not part of the test suite."""
import argparse
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--n-harmonics", type=int, default=128)
    ap.add_argument("--g-grid", type=int, default=1 << 20)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--form", choices=("auto", "double"), default="auto")
    args = ap.parse_args()
    import torch
    import oracle_binding as ob
    slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
    torch.zeros(1, device="cuda")
    N, M = args.n_harmonics, args.g_grid
    p = ob.make_params(N=N, M=M, B=4000.0 / M, omega=100.0)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    sol = slb.Solver()
    sol.tiptoe()
    t0 = time.perf_counter()
    per_launch = sol.steps_per_launch() if args.form == "auto" else 2
    left, launches = args.steps, {3: 0, 2: 0, 1: 0}
    while left > 0:
        k = per_launch if left >= per_launch else (2 if left >= 2 else 1)
        {3: lambda: sol.multi_step(3), 2: sol.double_step, 1: sol.step_fused}[k]()
        launches[k] += 1
        left -= k
    torch.cuda.synchronize()
    print(f"GPU: {args.steps} iterations in {time.perf_counter() - t0:.3f} s "
          f"({launches[3]} launches of three, {launches[2]} of two, {launches[1]} of one)", flush=True)
    orc = ob.OracleState(p, threads=args.threads)
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:                         # a progress line every few seconds
        k = min(10, args.steps - done)
        orc.step(k)
        done += k
        print(f"oracle: {done} iterations, {time.perf_counter() - t0:.1f} s", flush=True)
    s = sol.s
    assert (s.current, s.current_hs) == (orc.st.current, orc.st.current_hs)
    assert s.t == orc.st.t
    bad = 0
    for which, i, arr in (("a", s.current, 1 + s.current), ("b", s.current, 5 + s.current),
                          ("a", s.current_hs, 1 + s.current_hs), ("b", s.current_hs, 5 + s.current_hs)):
        got = sol.rows(arr)[:, :M + 3].view(np.uint32)
        exp = orc.array(which, i).view(np.uint32)
        diff = int(np.count_nonzero(got != exp))
        bad += diff
        print(f"{which}[{i}]: {got.size} cells, {diff} differ", flush=True)
    av_gpu = torch.zeros(6, dtype=torch.float32, device="cuda")
    av_cpu = np.zeros(6, np.float32)
    L = slb.lib()
    L.av(0, s.a[s.current], s.b[s.current], av_gpu.data_ptr(), np.float32(0.3))
    ob.lib().orc_av(C.byref(p), np.ascontiguousarray(orc.array("a", s.current)),
                    np.ascontiguousarray(orc.array("b", s.current)), av_cpu, np.float32(0.3))
    torch.cuda.synchronize()
    same_av = np.array_equal(av_gpu.cpu().numpy().view(np.uint32), av_cpu.view(np.uint32))
    print("av() of the final lattice:", "identical" if same_av else f"DIFFERS {av_gpu.cpu().numpy()} {av_cpu}")
    print("RESULT:", "bit-identical" if bad == 0 and same_av else "MISMATCH")
    sol.close()
    orc.close()
    return 0 if bad == 0 and same_av else 1


if __name__ == "__main__":
    sys.exit(main())
