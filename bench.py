#!/usr/bin/env python3
"""Throughput of the Boltzmann stepping loop on MI355X.

One "step" = one iteration of the time loop = main-lattice sweep + half-step
lattice sweep over the whole (n-harmonics x g-grid) lattice.  The workload is
the roofline shape, n-harmonics=128 and g-grid=2^20 PER GPU: at
--gpus 1 that is config 3; at --gpus 8 the eight phi_y slabs make up config 4
(g-grid=2^23), so the scaling reported is weak scaling.

    python bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --scaling strong --g-grid 8388608     # config 4 itself, cut into N slabs

With N > 1 the phi_y slabs are driven by the C slab loop, which calls RCCL itself (one
communicator; torch.distributed, over gloo, only hands round the identifier, holds the
barriers and takes the maximum of the wall times).  Before anything is timed every rank
checks that the ghost columns its neighbours sent it are bit for bit the columns those
neighbours own, and the run ends non-zero, naming rank and peer, if they are not.

With N > 1 and no WORLD_SIZE in the environment this process only starts the N
ranks (python -m torch.distributed.run ... bench.py, one per GPU) before it has
touched a GPU, passes their output through and exits with their status; started
by torch.distributed.run itself it is one of the ranks.

The timed block of exactly K steps (barrier + device synchronisation on both
sides, maximum over ranks) is repeated --repeats times; `value` and
`ms_per_step` are those of the median block, the spread is reported beside them.
Prints one JSON line (rank 0).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
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
BYTES_PER_UPDATE_TRIPLE = 12   # three steps per launch: ... three updates
TRAFFIC_FILE = os.path.join("profiles", "hbm_traffic.json")


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_threads():
    """Threads for the CPU baseline: OMP_NUM_THREADS where the machine sets it (a GPU box gives
    a job its share of the host's cores that way), else every CPU this process may run on."""
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 0:
        return env, "OMP_NUM_THREADS"
    try:
        return len(os.sched_getaffinity(0)), "sched_getaffinity"
    except AttributeError:
        return os.cpu_count() or 1, "cpu_count"


def cpu_baseline(n_harm, g_grid, phys, steps):
    """The CPU oracle (a restatement of the reference's OpenMP solver) on the
    host cores, same lattice shape, a few steps."""
    import oracle_binding as ob
    threads, source = cpu_threads()
    p = ob.make_params(N=n_harm, M=g_grid, **phys)
    st = ob.OracleState(p, threads=threads)
    st.step(1)                                   # touch every page once
    t0 = time.perf_counter()
    st.step(steps)
    dt = time.perf_counter() - t0
    st.close()
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    return {"value": n_harm * g_grid * steps / dt, "unit": "lattice-point-updates/s",
            "cores": threads, "threads_from": source, "nproc": os.cpu_count(), "affinity": affinity,
            "cpu_model": cpu_model(), "kind": "port", "seconds": dt,
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed blocks of --steps steps each; the median block is reported")
    ap.add_argument("--n-harmonics", type=int, default=N_HARMONICS)
    ap.add_argument("--g-grid-per-gpu", type=int, default=G_GRID_PER_GPU)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): --g-grid-per-gpu columns on every GPU; strong: --g-grid columns "
                         "in all, cut into one slab per GPU")
    ap.add_argument("--g-grid", type=int, default=1 << 23,
                    help="--scaling strong: columns of the whole lattice (default 2^23 = config 4)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--n-chunk", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--form", choices=("auto", "triple", "double", "fused", "sweeps"), default="auto",
                    help="triple: three steps (six sweeps) per launch, a remainder of the K steps in a "
                         "double and a single step (default); "
                         "double: two steps (four sweeps) per launch; "
                         "fused: both sweeps of a step in one launch; "
                         "sweeps: the reference's two launches per step")
    ap.add_argument("--driver", choices=("auto", "c", "torch"), default="auto",
                    help="N > 1: who orders the slab launches and moves the halo columns. c: the C slab "
                         "driver calling RCCL itself (host/slb_slabs.c); torch: torch.distributed "
                         "point-to-point operations (dist.py); auto: c for the double step (the run fails "
                         "if that driver cannot start -- it never changes silently), torch for the other "
                         "forms.  SLB_DIST_BACKEND=gloo (rehearsals with several ranks on one GPU): the C "
                         "driver's messages go over sockets instead of RCCL")
    ap.add_argument("--data", choices=("analytic", "stress"), default="analytic",
                    help="analytic (default): the equilibrium initial state, whose high harmonics are zeros and "
                         "denormals; stress (one GPU): every array filled with sine products of (n, m), values "
                         "around 0.1 throughout -- shows whether the rate depends on what the lattice holds")
    ap.add_argument("--no-self-check", action="store_true", help="N > 1: skip the ghost-column check before timing")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-two-sweep", action="store_true", help="skip the two-sweep leg of the default run")
    ap.add_argument("--two-sweep-steps", type=int, default=10)
    ap.add_argument("--cpu-steps", type=int, default=20)
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N without a rank environment: start the N ranks as children.  Nothing here has
    touched a GPU (torch is not even imported), so the children are the only GPU processes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def alg_bytes_per_launch(form, n_harm, cols):
    """Algorithmic bytes of the dominant kernel's launch on `cols` grid columns: the main sweep
    covers cols+1 columns and the half-step sweep cols, hence the half column."""
    points = n_harm * (cols + 0.5)
    if form == "triple":       # one launch = three steps, the same 9 array passes
        return "ring_step_kernel", 3 * BYTES_PER_UPDATE_TRIPLE * points
    if form == "double":       # one launch = two steps, the same 9 array passes
        return "fused2_step_kernel", 2 * BYTES_PER_UPDATE_DOUBLE * points
    if form == "fused":
        return "fused_step_kernel", BYTES_PER_UPDATE_FUSED * points
    return "sweep_pipe_kernel", BYTES_PER_POINT_SWEEP * points


def recorded_traffic(form, n_harm, g_grid):
    """HBM bytes per launch counted by rocprofv3 --pmc for this form and lattice in an earlier
    run of this command (profiles/hbm_traffic.json), or None: counters cannot be read from
    inside the timed process."""
    tfile = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(tfile):
        return None, None
    with open(tfile) as f:
        tj = json.load(f)
    for key, e in tj.items():
        if isinstance(e, dict) and e.get("form", key) == form and e.get("n_harmonics") == n_harm \
                and e.get("g_grid") == g_grid:
            return e.get("bytes_per_launch"), f"{TRAFFIC_FILE}:{key} ({e.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')})"
    return None, None


def roofline(form, n_harm, cols, kernel_ms, traffic=None, traffic_source=None):
    kernel, alg = alg_bytes_per_launch(form, n_harm, cols)
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kernel, "avg_launch_ms": kernel_ms, "algorithmic_bytes_per_launch": alg}


def launches_per(form, steps):
    if form == "triple":
        return steps // 3 + (steps % 3 + 1) // 2
    return {"double": steps // 2 + steps % 2, "fused": steps, "sweeps": 2 * steps}[form]


def median_block(blocks):
    """blocks: list of (wall seconds, kernel ms or None).  The block with the median wall time."""
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    return blocks[order[(len(blocks) - 1) // 2]]


def fill_without_tails(sol, torch):
    """--data stress: every lattice array (a0 included) overwritten with products of sines of (n, m),
    magnitudes around 0.1 everywhere -- no zeros, no denormals, none of the tails of the analytic
    state at high harmonics.  Deterministic, no RNG."""
    import ctypes
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    hip = ctypes.CDLL(paths[0])
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    s = sol.s
    n = torch.arange(s.N + 1, device="cuda", dtype=torch.float32)[:, None]
    m = torch.arange(s.pitch, device="cuda", dtype=torch.float32)[None, :]
    arrays = [s.a0] + [s.a[i] for i in range(4)] + [s.b[i] for i in range(4)]
    for k, ptr in enumerate(arrays):
        x = (0.1 + 0.2 * torch.sin(0.37 * n + 0.0113 * m + k) * torch.cos(0.05 * n - 0.0031 * m + 2 * k)).contiguous()
        assert hip.hipMemcpy(ptr, x.data_ptr(), x.numel() * 4, 3) == 0          # device to device
    torch.cuda.synchronize()


def run_single(args, slb, torch, form, steps, warmup, repeats):
    sol = slb.Solver(0, 1)
    sol.tiptoe()
    if args.data == "stress":
        fill_without_tails(sol, torch)
    sol.set_fused({"triple": 3, "double": 2, "fused": 1, "sweeps": 0}[form])
    warm = warmup
    if form in ("double", "triple"):
        # an even number of warm-up steps keeps the ping-pong parity the spare arrays were
        # made for (an odd count would re-copy them inside the timed region)
        warm += warm % 2
        sol.double_step()                    # allocates the spare arrays outside the timed region
    sol.run_timed(warm)
    blocks = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_ms = sol.run_timed(steps)             # HIP events on the launch stream
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        blocks.append((wall, ev_ms / launches_per(form, steps)))   # launches are back to back
    if form == "triple" and steps % 3:
        # the dominant kernel's own launch time: blocks of three-step launches only, same events
        n3 = 2 * max(steps // 6, 1)
        pure = sorted(sol.run_timed(3 * n3) / n3 for _ in range(max(repeats, 3)))
        k3 = pure[(len(pure) - 1) // 2]
        blocks = [(w, k3) for w, _ in blocks]
    sol.close()
    return blocks


def ghost_self_check(slb, run, driver, rank, world, dist):
    """After the warm-up steps, before anything is timed: the ghost columns of both current
    lattices on every rank must be, bit for bit, the outermost columns its neighbours own --
    they were sent through the driver's own exchange (C driver, slb_slabs_multi_step: pack, RCCL
    send/recv, unpack, G-1 = 5 columns of the main lattice and G = 6 of the half-step lattice
    after every launch; torch driver: 3 and 4 after a double step).  Every rank learns the
    verdict; a mismatch ends the run non-zero naming rank, peer and lattice."""
    import numpy as np
    import torch
    run.join() if hasattr(run, "join") else None
    torch.cuda.synchronize()
    sol = run.s if driver == "c" else run.eng.sol.s
    solp = slb.lib().slb_slabs_solver(run.r) if driver == "c" else run.eng.sol._p
    G, W, rows = sol.G, sol.W, sol.N + 1

    def window(arr, col0, ncols):
        out = np.empty((rows, ncols), dtype=np.float32)
        rc = slb.lib().slb_solver_get_window(solp, arr, col0, ncols, out.ctypes.data)
        if rc:
            raise SystemExit(f"rank {rank}: slb_solver_get_window -> {rc}")
        return out

    d_main, d_half = (G - 1, G) if driver == "c" else (3, 4)
    lattices = (("main", 1 + sol.current, 5 + sol.current, d_main), ("half-step", 1 + sol.current_hs, 5 + sol.current_hs, d_half))
    mine = {}
    for name, ia, ib, depth in lattices:
        for arr_name, arr in (("a", ia), ("b", ib)):
            mine[(name, arr_name, "own_left")] = window(arr, G, depth)                 # what rank-1 should hold as its right ghosts
            mine[(name, arr_name, "own_right")] = window(arr, G + W - depth, depth)    # what rank+1 should hold as its left ghosts
            mine[(name, arr_name, "ghost_left")] = window(arr, G - depth, depth)
            mine[(name, arr_name, "ghost_right")] = window(arr, G + W, depth)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    bad = []
    for name, _, _, depth in lattices:
        for arr_name in ("a", "b"):
            if rank > 0:
                exp = everyone[rank - 1][(name, arr_name, "own_right")]
                if not np.array_equal(mine[(name, arr_name, "ghost_left")].view(np.uint32), exp.view(np.uint32)):
                    bad.append(f"rank {rank}: left ghost columns of {arr_name} ({name} lattice) differ from what peer {rank - 1} owns")
            if rank < world - 1:
                exp = everyone[rank + 1][(name, arr_name, "own_left")]
                if not np.array_equal(mine[(name, arr_name, "ghost_right")].view(np.uint32), exp.view(np.uint32)):
                    bad.append(f"rank {rank}: right ghost columns of {arr_name} ({name} lattice) differ from what peer {rank + 1} owns")
    # a lattice that never moved would pass trivially: the edge columns must have changed since t = 0
    moved = bool(np.any(mine[("half-step", "b", "own_left")] != 0))
    if not moved:
        bad.append(f"rank {rank}: the half-step lattice's b is still zero after the warm-up steps")
    verdicts = [None] * world
    dist.all_gather_object(verdicts, bad)
    problems = [m for v in verdicts for m in v]
    if problems:
        if rank == 0:
            sys.stderr.write("bench: halo self-check FAILED\n" + "\n".join(problems) + "\n")
        dist.barrier()
        raise SystemExit(3)
    return {"ok": True, "ranks": world,
            "checked": f"ghost columns of both current lattices ({d_main} main, {d_half} half-step per side, a and b, all rows) "
                       "equal the neighbours' owned edge columns bit for bit after the warm-up steps"}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # Standard output is kept for the one JSON line: whatever libraries print there meanwhile
    # (gloo announces its connections on it) goes to standard error instead.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    slb = importlib.import_module("super-lattice-boltzmann-2d_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.data == "stress" and world > 1:
        raise SystemExit("--data stress fills one GPU's lattice")
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    torch.zeros(1, device="cuda")              # make this device current for the C library too
    # SLB_DIST_BACKEND=gloo: rehearsal of the multi-rank path where RCCL cannot run (several
    # ranks on one GPU); halo messages are then staged through host memory
    backend = os.environ.get("SLB_DIST_BACKEND", "nccl")
    ranks_seen = 1
    driver = None
    if args.form == "auto":
        args.form = "double" if world > 1 and args.driver == "torch" else "triple"
    if world > 1:
        driver = args.driver
        if driver == "auto":
            driver = "c" if args.form in ("double", "triple") else "torch"
        if driver == "c" and args.form not in ("double", "triple"):
            raise SystemExit("--driver c runs two or three steps per launch only")
        if driver == "torch" and args.form == "triple":
            raise SystemExit("--form triple over several GPUs is the C slab driver's (--driver c)")
        import torch.distributed as dist
        # The C driver owns the one RCCL communicator of the run (ncclCommInitRank in
        # host/slb_link.c); torch.distributed is then control plane only and runs over gloo.
        # The torch driver moves the halo columns with torch's own point-to-point operations
        # and needs the process group on RCCL.
        if driver == "torch" and backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group("gloo")
        control_on_gpu = dist.get_backend() == "nccl"
        ranks_seen = dist.get_world_size()

    n_harm = args.n_harmonics
    g_grid = args.g_grid if args.scaling == "strong" else args.g_grid_per_gpu * world
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
    repeats = max(args.repeats, 1)

    per_rank_ms = None
    self_check = None
    if world == 1:
        blocks = run_single(args, slb, torch, args.form, args.steps, args.warmup, repeats)
    else:
        drv = importlib.import_module("super-lattice-boltzmann-2d_amd.dist")
        if driver == "c":
            run = drv.CSlabRun(slb, rank, world, backend)        # raises on every rank alike
            transport = run.transport()                          # "rccl", or "fds" in a rehearsal
            ranks_seen = run.ranks_seen()                        # ncclCommCount of the C driver's communicator
            if ranks_seen != world:
                raise SystemExit(f"rank {rank}: RCCL reports {ranks_seen} ranks, expected {world}")
        else:
            run = drv.SlabRun(slb, rank, world)
            transport = "torch.distributed " + dist.get_backend()
        run.tiptoe()

        def advance(nsteps):
            k = 0
            if args.form == "triple":
                for _ in range(nsteps // 3):
                    run.multi_step(3)
                k = nsteps - nsteps % 3
            if args.form in ("double", "triple"):
                for _ in range((nsteps - k) // 2):
                    run.double_step()
                k += (nsteps - k) // 2 * 2
            for _ in range(nsteps - k):
                (run.step if args.form == "sweeps" else run.step_fused)()

        if args.form == "triple":
            advance(max((args.warmup + 2) // 3 * 3, 3))      # whole launches of three: all their ghost columns are fresh
        else:
            advance(max(args.warmup + args.warmup % 2, 2))
        if args.no_self_check or args.form not in ("double", "triple"):
            self_check = {"ok": None, "skipped": "--no-self-check" if args.no_self_check else
                          "the check knows the ghost depth of the multi-step launches only"}
        else:
            self_check = ghost_self_check(slb, run, driver, rank, world, dist)
        blocks = []
        mine = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()                                 # on the stream of the interior launches
            advance(args.steps)
            ev1.record()
            torch.cuda.synchronize()
            t_mine = time.perf_counter() - t0            # this rank's own loop, before it waits for the others
            dist.barrier()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            kernel_ms = None
            if args.form == "double" and args.steps % 2 == 0:
                # this rank's stream time per double step: the interior launch (the edge launches
                # and the halo traffic run beside it) plus whatever it had to wait for
                kernel_ms = ev0.elapsed_time(ev1) / (args.steps // 2)
            if args.form == "triple" and args.steps % 3 == 0:
                kernel_ms = ev0.elapsed_time(ev1) / (args.steps // 3)
            w = torch.tensor([wall], dtype=torch.float64, device="cuda" if control_on_gpu else "cpu")
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            blocks.append((float(w.item()), kernel_ms))
            mine.append(t_mine / args.steps * 1e3)
        per = {"triple": 3, "double": 2}.get(args.form)
        if per and args.steps % per:
            # this rank's stream time per launch of the dominant kernel: blocks of whole launches
            # only, same events, outside the timed region
            n = max(args.steps // per, 2)
            times = []
            for _ in range(3):
                torch.cuda.synchronize()
                dist.barrier()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                advance(per * n)
                ev1.record()
                torch.cuda.synchronize()
                times.append(ev0.elapsed_time(ev1) / n)
            times.sort()
            blocks = [(w, times[1]) for w, _ in blocks]
        mine.sort()
        gathered = [None] * world
        dist.all_gather_object(gathered, mine[(len(mine) - 1) // 2])
        per_rank_ms = gathered
        run.close()

    wall, kernel_ms = median_block(blocks)
    walls = [b[0] for b in blocks]
    updates = float(n_harm) * float(g_grid) * args.steps
    value = updates / wall
    bytes_per_update = {"triple": BYTES_PER_UPDATE_TRIPLE, "double": BYTES_PER_UPDATE_DOUBLE,
                        "fused": BYTES_PER_UPDATE_FUSED, "sweeps": 2 * BYTES_PER_POINT_SWEEP}[args.form]
    out = {
        "metric": "lattice-point-updates/sec (n-harmonics x g-grid per step)",
        "value": value, "unit": "lattice-point-updates/s",
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "repeats": repeats,
        "ms_per_step_min": min(walls) / args.steps * 1e3, "ms_per_step_max": max(walls) / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (analytic equilibrium initial state, no RNG)" if args.data == "analytic" else
                                 "synthetic (sine products of (n, m) in every array, no tails, no RNG)",
        "config": {"workload": f"n-harmonics={n_harm} g-grid={g_grid} "
                               f"({g_grid // world} columns per GPU, phi_y slabs) "
                               f"omega=100 B={phys['B']:.6g} dt=0.001, 2 sweeps per step",
                   "form": args.form, "parallelism": f"phi_y-slab x{world}"},
        # whole-job rate against the per-GPU HBM peak, with the bytes this form must move ...
        "hbm_roofline_frac_per_gpu": bytes_per_update * value / world / (HBM_PEAK_GBS * 1e9),
        # ... and with the 56 B/update of the reference's two-sweep formulation, for comparison
        "equiv_two_sweep_traffic_frac_per_gpu": 2 * BYTES_PER_POINT_SWEEP * value / world / (HBM_PEAK_GBS * 1e9),
    }
    if driver is not None:
        out["config"]["driver"] = {"c": "C slab driver, RCCL send/recv called from C",
                                   "torch": "torch.distributed point-to-point operations"}[driver]
        out["config"]["backend"] = backend
        out["config"]["transport"] = transport
        out["config"]["control_plane"] = "torch.distributed over " + dist.get_backend()
        out["driver_ran"] = driver
        out["ms_per_step_by_rank"] = per_rank_ms
        out["self_check"] = self_check
    if rank == 0:
        if kernel_ms is not None:
            traffic, source = (None, None)
            if world == 1:
                traffic, source = recorded_traffic(args.form, n_harm, g_grid)
            out["roofline"] = roofline(args.form, n_harm, g_grid / world, kernel_ms, traffic, source)
            if world > 1:
                out["roofline"]["note"] = ("rank 0's slab; stream time per launch including waits "
                                           "for halo columns, not the bare kernel")
            elif args.form == "triple":
                out["roofline"]["note"] = ("the kernel is bound by vector-instruction issue, not by HBM: the fraction says how "
                                           "much of the HBM peak its 12 B per update use; avg_launch_ms is measured over "
                                           "blocks of three-step launches only (HIP events on the launch stream)")
        if world == 1 and args.form == "triple" and not args.no_two_sweep:
            # the two-steps-per-launch kernel (the slab driver's form, last round's default), same process
            bd = run_single(args, slb, torch, "double", 20, 2, 3)
            wd, kd = median_block(bd)
            td, sd = recorded_traffic("double", n_harm, g_grid)
            dbl = roofline("double", n_harm, g_grid, kd, td, sd)
            dbl.update({"steps": 20, "repeats": 3, "value": n_harm * g_grid * 20 / wd, "ms_per_step": wd / 20 * 1e3})
            out["double_step"] = dbl
        if world == 1 and args.form != "sweeps" and not args.no_two_sweep:
            # the reference's own formulation (two launches per step, 28 B per point per sweep),
            # timed in the same process
            b2 = run_single(args, slb, torch, "sweeps", args.two_sweep_steps, 2, 3)
            w2, k2 = median_block(b2)
            t2, s2 = recorded_traffic("sweeps", n_harm, g_grid)
            two = roofline("sweeps", n_harm, g_grid, k2, t2, s2)
            two.update({"steps": args.two_sweep_steps, "repeats": 3,
                        "value": n_harm * g_grid * args.two_sweep_steps / w2,
                        "ms_per_step": w2 / args.two_sweep_steps * 1e3})
            out["two_sweep"] = two
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(n_harm, g_grid, phys, args.cpu_steps)
            out["config1"] = config1_case(slb, torch)
        sys.stdout.flush()
        import ctypes
        ctypes.CDLL(None).fflush(None)              # C-level buffers of the redirected output go where they were meant to
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
