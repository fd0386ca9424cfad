"""One rank of a slab run driven through the C loop (host/slb_slabs.c) with its messages over
inherited sockets: started by tests/test_slabs_c_gpu.py, one process per rank, all on one GPU.

usage: slab_c_worker.py RANK NRANKS OUTDIR PLAN N M B OMEGA FD_0 ... FD_{NRANKS-1}
PLAN: letters t (three steps in one launch), d (double step), s (single step with a ghost refresh), l (last step)
FD_q: this rank's socket to rank q (-1 for itself)
"""
import ctypes as C
import faulthandler
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_binding as ob          # noqa: E402  (only for make_params: the shared parameter record)
import slab_helpers as sh            # noqa: E402


def main():
    faulthandler.dump_traceback_later(90, exit=True)      # a stuck rank says where, then ends
    rank, nranks, outdir, plan = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    N, M, B, omega = int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7]), float(sys.argv[8])
    fds = [int(x) for x in sys.argv[9:9 + nranks]]
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    slb, _ = sh.import_pkg()
    p = ob.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=p.N, M=p.M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    L = slb.lib()
    link = L.slb_link_open_fds(rank, nranks, (C.c_int * nranks)(*fds))
    assert link
    r = L.slb_slabs_create(rank, nranks, link)
    assert r
    L.slb_slabs_tiptoe(r)
    for step in plan:
        if step == "t":
            L.slb_slabs_multi_step(r, 3, None)
        elif step == "d":
            L.slb_slabs_double_step(r, None, None)
        elif step == "s":
            L.slb_slabs_step(r)
        elif step == "l":
            L.slb_slabs_last_step(r)
        else:
            raise SystemExit(f"unknown step {step!r}")
    s = L.slb_slabs_solver(r).contents
    total = np.empty((N + 1, M + 3), np.float32)
    out = {"t": np.float32(s.t), "idx": np.array([s.current, s.current_hs])}
    for name, arr in (("a_main", 1 + s.current), ("b_main", 5 + s.current),
                      ("a_half", 1 + s.current_hs), ("b_half", 5 + s.current_hs)):
        L.slb_slabs_gather_rows(r, arr, 0, N + 1, total.ctypes.data)
        if rank == 0:
            out[name] = total.copy()
    if rank == 0:
        np.savez(os.path.join(outdir, "gathered.npz"), **out)
    L.slb_slabs_destroy(r)
    L.slb_link_close(link)


if __name__ == "__main__":
    main()
