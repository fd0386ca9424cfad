#!/usr/bin/env python3
"""What the GPU's clocks and power do while each step form runs back to back for a few seconds at
n-harmonics=128, g-grid=2^20: a thread asks rocm-smi for clocks and power a few times a second while
the main thread keeps the launch stream full and times blocks of launches with HIP events."""
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sample(stop, out):
    while not stop.is_set():
        t = time.perf_counter()
        try:
            r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"],
                               capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            pick = {k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk", "mclk", "fclk", "power", "junction", "edge"))}
            out.append((t, pick))
        except Exception as e:            # noqa: BLE001  (a probe: report and go on)
            out.append((t, {"error": repr(e)[:200]}))
        time.sleep(0.1)


def main():
    import torch
    slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
    torch.zeros(1, device="cuda")
    N, M = 128, 1 << 20
    slb.configure(N=N, M=M, t_start=1.0, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    slb.lib().slb_set_stream(torch.cuda.current_stream().cuda_stream)
    idle = []
    stop = threading.Event()
    th = threading.Thread(target=sample, args=(stop, idle))
    th.start(); time.sleep(1.5); stop.set(); th.join()
    print("idle:", idle[-1][1])
    for form, fused, per, block in (("triple", 3, 3, 300), ("double", 2, 2, 300), ("sweeps", 0, 0.5, 150)):
        sol = slb.Solver(0, 1)
        sol.tiptoe()
        sol.set_fused(fused)
        sol.double_step()
        torch.cuda.synchronize()
        time.sleep(2.0)                                  # let the device idle first
        samples = []
        stop = threading.Event()
        th = threading.Thread(target=sample, args=(stop, samples))
        th.start()
        t0 = time.perf_counter()
        rows = []
        while time.perf_counter() - t0 < 6.0:
            ms = sol.run_timed(block)
            rows.append((time.perf_counter() - t0, ms / (block / per)))
        stop.set(); th.join()
        sol.close()
        print(f"## {form}: ms per launch by block of {block} steps (time since start s: ms)")
        print("  " + "  ".join(f"{t:.2f}:{m:.4f}" for t, m in rows))
        print("  samples (s since start):")
        for t, p in samples:
            print(f"   {t - t0:6.2f}  {p}")


if __name__ == "__main__":
    main()
