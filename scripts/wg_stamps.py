#!/usr/bin/env python3
"""Timing build only (make EXTRA=-DSLB_TIMING_VARIANTS): when did each workgroup of one launch of the
three-step ring kernel start and end (100 MHz counter), and on which XCD?  Prints the spread."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
N, M = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M, PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
L = slb.lib()
sol = slb.Solver(); sol.tiptoe()
for _ in range(4): sol.multi_step(3)
L.slb_set_tuning(int(os.environ.get("STAMPS_VARIANT", "95")), 0, 0)
buf = torch.zeros(4 * 4 * 20000, dtype=torch.int64, device="cuda")
L.slb_debug_set_stamps.argtypes = [C.c_void_p]
L.slb_debug_set_stamps(C.c_void_p(buf.data_ptr()))
for _ in range(6): sol.multi_step(3)
torch.cuda.synchronize()
buf.zero_(); torch.cuda.synchronize()
sol.multi_step(3)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 4, 4)
out = os.environ.get("STAMPS_OUT")
if out: np.save(out, s)
used = s[:, 0, 1] != 0
s = s[used]
t0 = s[:, :, 0].min()
start = (s[:, :, 0].min(axis=1) - t0) / 100.0      # us
end = (s[:, :, 1].max(axis=1) - t0) / 100.0
xcc = s[:, 0, 3] & 15
dur = end - start
print(f"workgroups {len(s)}  launch spans {end.max():.1f} us")
order = np.argsort(start)
for lo, hi in [(0, 512), (512, 1024), (1024, len(s))]:
    idx = np.arange(len(s))[lo:hi]
    if len(idx) == 0: continue
    print(f"wg {lo}..{hi}: start {start[idx].min():.1f}..{start[idx].max():.1f}  end {end[idx].min():.1f}..{end[idx].max():.1f}  dur mean {dur[idx].mean():.1f} min {dur[idx].min():.1f} max {dur[idx].max():.1f}")
for x in range(8):
    idx = xcc == x
    print(f"xcc {x}: {idx.sum()} wgs, last end {end[idx].max():.1f} us, mean dur of whole strips {dur[idx & (np.arange(len(s)) < 1024)].mean():.1f}")
# per-wave spread inside a workgroup
wdur = (s[:, :, 1] - s[:, :, 0]) / 100.0
print("per-wave duration, first 8 workgroups:", np.round(wdur[:8], 1).tolist())
hist, edges = np.histogram(end, bins=16)
print("ends histogram:", hist.tolist(), [round(e) for e in edges.tolist()])
sol.close()
