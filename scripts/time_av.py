#!/usr/bin/env python3
"""Time av() in its three modes at a given g-grid (HIP events via torch on the null stream)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
torch.zeros(1, device="cuda")
for M in (4096, 262144, 1 << 20):
    N = 8
    slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    L = slb.lib()
    sol = slb.Solver(); sol.tiptoe(); sol.step()
    s = sol.s
    for mode, name in ((slb.SLB_AV_EXACT, "exact"), (slb.SLB_AV_CHAIN, "chain"), (slb.SLB_AV_TREE, "tree")):
        L.slb_set_av_mode(mode)
        L.av(0, s.a[s.current], s.b[s.current], s.av_data, 0.1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.av(0, s.a[s.current], s.b[s.current], s.av_data, 0.1)
        e1.record(); torch.cuda.synchronize()
        print(f"g-grid={M:8d} av {name:5s}: {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us per call", flush=True)
    L.slb_set_av_mode(slb.SLB_AV_EXACT)
    sol.close()

# av between stepping kernels (the GPU is then at its working clock)
M, N = 1 << 20, 128
slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
              PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
sol = slb.Solver(); sol.tiptoe()
for _ in range(5):
    sol.step_fused()
s = sol.s
for with_av in (False, True, False, True):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sol.step_fused()
        if with_av:
            L.av(0, s.a[s.current], s.b[s.current], s.av_data, 0.1)
    e1.record(); torch.cuda.synchronize()
    print(f"n-harmonics=128 g-grid=2^20: step_fused{' + av exact' if with_av else '           '}: "
          f"{e0.elapsed_time(e1) / 20:7.3f} ms per step", flush=True)
sol.close()
