#!/usr/bin/env python3
"""Throughput of the Boltzmann stepping loop on MI355X.

One "step" = one iteration of the time loop = main-lattice sweep + half-step
lattice sweep over the whole (n-harmonics x g-grid) lattice.  The workload is
the roofline shape, n-harmonics=128 and g-grid=2^20 PER GPU: at
--gpus 1 that is config 3; at --gpus 8 the eight phi_y slabs make up config 4
(g-grid=2^23), so the scaling reported is weak scaling.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints one JSON line (rank 0).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_HARMONICS = 128
G_GRID_PER_GPU = 1 << 20
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, spec
BYTES_PER_POINT_SWEEP = 28     # two-sweep form: 5 fp32 reads + 2 fp32 writes per point per sweep
BYTES_PER_UPDATE_FUSED = 36    # one-launch step: 5 reads + 4 writes per lattice-point-update
BYTES_PER_UPDATE_DOUBLE = 18   # two steps per launch: the same 9 accesses serve two updates


def cpu_baseline(n_harm, g_grid, phys, steps):
    """The CPU oracle (a restatement of the reference's OpenMP solver) on the
    host cores, same lattice shape, a few steps."""
    import oracle_binding as ob
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    p = ob.make_params(N=n_harm, M=g_grid, **phys)
    st = ob.OracleState(p, threads=threads)
    st.step(1)                                   # touch every page once
    t0 = time.perf_counter()
    st.step(steps)
    dt = time.perf_counter() - t0
    st.close()
    return {"value": n_harm * g_grid * steps / dt, "unit": "lattice-point-updates/s",
            "cores": threads, "kind": "port",
            "sample": f"{steps} steps of the same n-harmonics={n_harm} g-grid={g_grid} lattice "
                      f"(OpenMP over phi_y, {threads} threads), after 1 untimed step"}


def config1_case(slb, torch):
    """The reference's own CPU-runnable case (n-harmonics=32, g-grid=4096, B=1, omega=1): the
    GPU time loop (two steps per launch) next to the serial CPU oracle, for orientation."""
    import oracle_binding as ob
    n_harm, g_grid, steps = 32, 4096, 2000
    phys = dict(E_dc=1.0, E_omega=0.1, omega=1.0, mu=5.0, alpha=1.0, B=1.0, PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    slb.configure(N=n_harm, M=g_grid, t_start=10.0, **phys)
    sol = slb.Solver(0, 1)
    sol.tiptoe()
    sol.set_fused(2)
    sol.double_step()
    sol.run_timed(100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sol.run_timed(steps)
    torch.cuda.synchronize()
    gpu = n_harm * g_grid * steps / (time.perf_counter() - t0)
    sol.close()
    st = ob.OracleState(ob.make_params(N=n_harm, M=g_grid, **phys), threads=1)
    st.step(10)
    cpu_steps = 300
    t0 = time.perf_counter()
    st.step(cpu_steps)
    cpu = n_harm * g_grid * cpu_steps / (time.perf_counter() - t0)
    st.close()
    return {"workload": f"n-harmonics={n_harm} g-grid={g_grid} B=1 omega=1 dt=0.001",
            "gpu_value": gpu, "cpu_serial_value": cpu, "unit": "lattice-point-updates/s",
            "gpu_steps": steps, "cpu_steps": cpu_steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-harmonics", type=int, default=N_HARMONICS)
    ap.add_argument("--g-grid-per-gpu", type=int, default=G_GRID_PER_GPU)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--n-chunk", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--form", choices=("double", "fused", "sweeps"), default="double",
                    help="double: two steps (four sweeps) per launch (default); "
                         "fused: both sweeps of a step in one launch; "
                         "sweeps: the reference's two launches per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-steps", type=int, default=20)
    args = ap.parse_args()

    import torch
    slb = importlib.import_module("super-lattice-boltzmann-2d_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    torch.zeros(1, device="cuda")              # make this device current for the C library too
    # SLB_DIST_BACKEND=gloo: rehearsal of the multi-rank path where RCCL cannot run (several
    # ranks on one GPU); halo messages are then staged through host memory
    backend = os.environ.get("SLB_DIST_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)

    n_harm = args.n_harmonics
    g_grid = args.g_grid_per_gpu * world
    phys = dict(E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / g_grid,
                PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    slb.configure(N=n_harm, M=g_grid, t_start=1.0, **phys)
    L = slb.lib()
    if world > 1:
        # not the legacy null stream: its implicit synchronisation with other streams would
        # serialise the halo messages with the interior launches (measured: 1.26 ms instead
        # of 1.18 ms per double step with RCCL in the loop)
        torch.cuda.set_stream(torch.cuda.Stream())
    L.slb_set_stream(torch.cuda.current_stream().cuda_stream)
    L.slb_set_tuning(args.variant, args.n_chunk, args.block)

    kernel_ms = None
    if world == 1:
        sol = slb.Solver(0, 1)
        sol.tiptoe()
        sol.set_fused({"double": 2, "fused": 1, "sweeps": 0}[args.form])
        warm = args.warmup
        if args.form == "double":
            # an even number of warm-up steps keeps the ping-pong parity the spare arrays were
            # made for (an odd count would re-copy them inside the timed region)
            warm += warm % 2
            sol.double_step()                    # allocates the spare arrays outside the timed region
        sol.run_timed(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_ms = sol.run_timed(args.steps)        # HIP events on the launch stream
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        n_launches = {"double": args.steps // 2 + args.steps % 2, "fused": args.steps,
                      "sweeps": 2 * args.steps}[args.form]
        kernel_ms = ev_ms / n_launches           # launches are back to back
        sol.close()
    else:
        import torch.distributed as dist
        drv = importlib.import_module("super-lattice-boltzmann-2d_amd.dist")
        run = drv.SlabRun(slb, rank, world)
        run.tiptoe()
        def advance(nsteps):
            k = 0
            if args.form == "double":
                for _ in range(nsteps // 2):
                    run.double_step()
                k = nsteps - nsteps % 2
            for _ in range(nsteps - k):
                (run.step if args.form == "sweeps" else run.step_fused)()

        advance(max(args.warmup + args.warmup % 2, 2))
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()                                 # on the stream of the interior launches
        advance(args.steps)
        ev1.record()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if args.form == "double" and args.steps % 2 == 0:
            # this rank's stream time per double step: the interior launch (the edge launches
            # and the halo traffic run beside it) plus whatever it had to wait for
            kernel_ms = ev0.elapsed_time(ev1) / (args.steps // 2)
        w = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
        run.close()

    updates = float(n_harm) * float(g_grid) * args.steps
    value = updates / wall
    out = {
        "metric": "lattice-point-updates/sec (n-harmonics x g-grid per step)",
        "value": value, "unit": "lattice-point-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (analytic equilibrium initial state, no RNG)",
        "config": {"workload": f"n-harmonics={n_harm} g-grid={g_grid} "
                               f"({args.g_grid_per_gpu} columns per GPU, phi_y slabs) "
                               f"omega=100 B={phys['B']:.6g} dt=0.001, 2 sweeps per step",
                   "form": args.form, "parallelism": f"phi_y-slab x{world}"},
        # whole-job rate against the per-GPU HBM peak, with the bytes this form must move ...
        "hbm_roofline_frac_per_gpu": {"double": BYTES_PER_UPDATE_DOUBLE, "fused": BYTES_PER_UPDATE_FUSED,
                                      "sweeps": 2 * BYTES_PER_POINT_SWEEP}[args.form]
                                     * value / world / (HBM_PEAK_GBS * 1e9),
        # ... and with the 56 B/update of the reference's two-sweep formulation, for comparison
        "equiv_two_sweep_traffic_frac_per_gpu": 2 * BYTES_PER_POINT_SWEEP * value / world / (HBM_PEAK_GBS * 1e9),
    }
    if rank == 0:
        if kernel_ms is not None:
            # dominant kernel; one launch = one step (fused) or one sweep of the whole lattice
            points = n_harm * (g_grid / world + 0.5)      # main sweep M+1 columns, half sweep M; per GPU
            if args.form == "double":      # one launch = two steps, the same 9 array passes
                kernel, alg_bytes = "fused2_step_kernel", 2 * BYTES_PER_UPDATE_DOUBLE * points
            elif args.form == "fused":
                kernel, alg_bytes = "fused_step_kernel", BYTES_PER_UPDATE_FUSED * points
            else:
                kernel, alg_bytes = "sweep_pipe_kernel", BYTES_PER_POINT_SWEEP * points
            achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tfile):
                with open(tfile) as f:
                    tj = json.load(f)
                tj = tj.get(args.form, {})
                if world == 1 and tj.get("n_harmonics") == n_harm and tj.get("g_grid") == g_grid:
                    traffic = tj.get("bytes_per_launch")
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": kernel, "avg_launch_ms": kernel_ms,
                               "algorithmic_bytes_per_launch": alg_bytes}
            if world > 1:
                out["roofline"]["note"] = ("rank 0's slab; stream time per double step including waits "
                                           "for halo columns, not the bare kernel")
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(n_harm, g_grid, phys, args.cpu_steps)
            out["config1"] = config1_case(slb, torch)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
