#!/usr/bin/env python3
"""How the parallel sequential-sum kernel fares on av()'s three term streams:
passes, one-by-one runs and time per stream, at several g-grid sizes."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
torch.zeros(1, device="cuda")
L = slb.lib()
SHAPES = {0: "8x8", 41: "4x4", 42: "8x4", 43: "16x2", 44: "4x8", 45: "16x4", 46: "1x16", 47: "2x8"}
variants = [int(v) for v in sys.argv[1:]] or [0]
for M, steps in ((4096, 200), (262144, 50), (1 << 20, 20)):
    N = 8
    slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    sol = slb.Solver(); sol.tiptoe()
    for _ in range(steps):
        sol.step()
    s = sol.s
    a = sol.rows(1 + s.current, 0, 2)
    b = sol.rows(5 + s.current, 1, 1)
    dPhi = np.float32(slb.get_global("host_dPhi"))
    m = np.arange(1, M + 1)
    phi = (np.float32(-4.0) + dPhi * (m - 1).astype(np.float32)).astype(np.float32)
    streams = {"b1*dPhi": b[0, 1:M + 1] * dPhi, "a0*phi*dPhi": a[0, 1:M + 1] * phi * dPhi, "a1*dPhi": a[1, 1:M + 1] * dPhi}
    for variant in variants:
      L.slb_set_tuning(variant, 0, 0)
      for name, x in streams.items():
        x = np.ascontiguousarray(x, dtype=np.float32)
        dx = torch.from_numpy(x).cuda()
        out = torch.zeros(1, device="cuda")
        st = torch.zeros(2, dtype=torch.int32, device="cuda")
        L.slb_seq_sum_stats(dx.data_ptr(), len(x), out.data_ptr(), st.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.slb_seq_sum_stats(dx.data_ptr(), len(x), out.data_ptr(), st.data_ptr())
        e1.record(); torch.cuda.synchronize()
        print(f"g-grid={M:8d} {name:12s} shape {SHAPES[variant]:5s}: sum={out.item(): .9e} passes={st[0].item()} "
              f"bursts={st[1].item()} {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)
    L.slb_set_tuning(0, 0, 0)
    sol.close()
