#!/usr/bin/env python3
"""Does one GPU gain from running the double step of a lattice as column domains on separate
streams, so that the thin end of one domain's launch overlaps the bulk of another's?

Timing only: mode `free` issues the domains' launches with no ordering between the streams
(the results are then wrong -- it bounds what the overlap could give); mode `edges` orders them
the way a correct pipeline would have to (a domain's launch of step k+1 waits for the launches
of step k that touched the columns next to it)."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-harmonics", type=int, default=128)
    ap.add_argument("--g-grid", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=40, help="double steps per timed block")
    ap.add_argument("--domains", default="1,2,3,4")
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
    sol.set_fused(2)
    for _ in range(6):
        sol.multi_step(2)
    s = sol.s
    first, last = s.half_slab.j_first, s.half_slab.j_last
    strip = 4 * 240                                   # columns of one workgroup's strips

    def run(nd, mode, steps):
        """steps double steps over nd domains; returns ms per double step"""
        streams = [torch.cuda.Stream() for _ in range(nd)]
        width = ((last - first + 1) // nd + strip - 1) // strip * strip
        dom = [(first + d * width, min(last, first + (d + 1) * width - 1)) for d in range(nd)]
        dom = [d for d in dom if d[0] <= d[1]]
        ev_edge = [[None, None] for _ in dom]         # per domain: its left and right edge launches of the last step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            c = slb.step2_cosines(sol.s.t)
            new_ev = [[None, None] for _ in dom]
            for d, (lo, hi) in enumerate(dom):
                st = streams[d]
                L.slb_push_stream(st.cuda_stream)
                if nd == 1 or mode == "free":
                    sol.double_step_range(lo, hi, c)
                else:
                    # edges first: the strips next to the neighbouring domains, after the
                    # neighbours' edge launches of the step before
                    l_edge = (lo, min(hi, lo + strip - 1)) if d > 0 else None
                    r_edge = (max(lo, hi - strip + 1), hi) if d < len(dom) - 1 else None
                    if l_edge and r_edge and l_edge[1] >= r_edge[0]:
                        r_edge = (l_edge[1] + 1, hi) if l_edge[1] < hi else None
                    if l_edge:
                        if ev_edge[d - 1][1] is not None:
                            st.wait_event(ev_edge[d - 1][1])
                        sol.double_step_range(l_edge[0], l_edge[1], c)
                        new_ev[d][0] = torch.cuda.Event()
                        new_ev[d][0].record(st)
                    if r_edge:
                        if ev_edge[d + 1][0] is not None:
                            st.wait_event(ev_edge[d + 1][0])
                        sol.double_step_range(r_edge[0], r_edge[1], c)
                        new_ev[d][1] = torch.cuda.Event()
                        new_ev[d][1].record(st)
                    ilo = l_edge[1] + 1 if l_edge else lo
                    ihi = r_edge[0] - 1 if r_edge else hi
                    if ilo <= ihi:
                        sol.double_step_range(ilo, ihi, c)
                L.slb_pop_stream()
            ev_edge = new_ev
            sol.double_step_finish()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps

    for rep in range(2):
        for nd in [int(x) for x in args.domains.split(",")]:
            for mode in (("one",) if nd == 1 else ("free", "edges")):
                run(nd, mode, 4)
                ms = min(run(nd, mode, args.steps) for _ in range(3))
                print(f"domains={nd} mode={mode:5s}  {ms:8.4f} ms per double step", flush=True)
    sol.close()


if __name__ == "__main__":
    main()
