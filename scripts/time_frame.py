"""Time the display=8 frame kernel at bench size (n-harmonics=128, g-grid=2^20)."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
slb = importlib.import_module("super-lattice-boltzmann-2d_amd")

N, M = 128, 1 << 20
slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
              PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
sol = slb.Solver()
sol.tiptoe()
L = slb.lib()
s = sol.s
for nphi in (8, 64):
    phi = np.linspace(-3.14, 3.14, nphi, dtype=np.float32)
    n = np.arange(N + 1, dtype=np.float32)[None, :]
    arg = (n * phi[:, None]).astype(np.float64)
    tab = torch.from_numpy(np.stack([np.cos(arg), np.sin(arg)], axis=-1).copy()).cuda()
    out = torch.empty((nphi, M + 1), dtype=torch.float32, device="cuda")
    L.slb_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    a, b = s.a[s.current], s.b[s.current]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.slb_frame_values(a, b, s.pitch, N + 1, 1, M + 1, tab.data_ptr(), nphi, out.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert rc == 0
    terms = nphi * (M + 1) * (N + 1)
    print(f"nphi={nphi}: {dt * 1e3:.2f} ms, {terms / dt / 1e9:.1f} G terms/s, "
          f"{6 * terms / dt / 1e12:.2f} T f64-op/s, whole 629-value frame ~{dt * 629 / nphi:.2f} s")
sol.close()
