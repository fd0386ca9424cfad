"""Two phi_y slabs, two processes, ONE GPU (gloo moves the halo columns through
host memory): the HIP kernels with slab offsets + the slab driver reproduce the
single-domain oracle bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_binding as ob          # noqa: E402
import slab_helpers as sh            # noqa: E402

pytestmark = pytest.mark.gpu

PHYS = dict(N=24, M=5003, B=0.8, omega=40.0, E_omega=0.3)
STEPS = 9


def _worker(rank, world, port, outdir, mode):
    import faulthandler
    import torch
    faulthandler.dump_traceback_later(90, exit=True)      # a stuck rank says where, then ends
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    slb, drv = sh.import_pkg()
    p = ob.make_params(**PHYS)
    slb.configure(N=p.N, M=p.M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    run = drv.SlabRun(slb, rank, world)
    run.tiptoe()
    k = 0
    while k < STEPS:
        # the lattices in between two fused iterations are never stored, so finish on a
        # single step: then every buffer, not only the current ones, equals the oracle's
        if mode == "double" and k + 2 < STEPS:
            run.double_step()
            k += 2
            continue
        if mode == "mixed" and k % 4 == 1 and k + 2 < STEPS:
            run.double_step()
            k += 2
            continue
        if mode in ("fused", "double") or (mode == "mixed" and k % 2 == 0):
            run.step_fused()
        else:
            run.step()
        k += 1
    # av() over slabs after the last step, twice (running averages)
    av_data = torch.zeros(6, dtype=torch.float32, device="cuda")
    run.av(av_data, 0.25)
    run.av(av_data, 0.251)
    torch.cuda.synchronize()
    sol = run.eng.sol
    s = sol.s
    lo = s.G - 1 if rank == 0 else s.G
    hi = s.G + s.W + 2 if rank == world - 1 else s.G + s.W
    np.savez(os.path.join(outdir, f"r{rank}.npz"), t=np.float32(s.t), av=av_data.cpu().numpy(),
             idx=np.array([s.current, s.next, s.current_hs, s.next_hs]),
             **{f"a{i}": sol.rows(1 + i)[:, lo:hi] for i in range(4)},
             **{f"b{i}": sol.rows(5 + i)[:, lo:hi] for i in range(4)})
    run.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "sweeps"), (2, "fused"), (2, "double"), (3, "double"), (3, "mixed")])
def test_slabs_on_one_gpu(world, mode, tmp_path):
    mp.spawn(_worker, args=(world, sh.free_port(), str(tmp_path), mode), nprocs=world, join=True)
    p = ob.make_params(**PHYS)
    ref = ob.OracleState(p, threads=4)
    ref.step(STEPS)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert all(np.float32(x["t"]) == np.float32(ref.st.t) for x in parts)
    assert list(parts[0]["idx"]) == [ref.st.current, ref.st.next, ref.st.current_hs, ref.st.next_hs]
    for w in "ab":
        for i in range(4):
            got = np.concatenate([x[f"{w}{i}"] for x in parts], axis=1)
            exp = ref.array(w, i)
            assert got.shape == exp.shape
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (w, i)
    # the slab-wise av() agrees on every rank and with the oracle's sequential sums up to
    # the rounding noise of a float running sum
    import ctypes as C
    av_ref = np.zeros(6, dtype=np.float32)
    a_cur, b_cur = ref.array("a", ref.st.current), ref.array("b", ref.st.current)
    for t in (0.25, 0.251):
        ob.lib().orc_av(C.byref(p), np.ascontiguousarray(a_cur), np.ascontiguousarray(b_cur), av_ref, np.float32(t))
    for x in parts:
        assert np.array_equal(x["av"], parts[0]["av"])
    assert parts[0]["av"][0] == 2.0
    scale = np.abs(av_ref[1:]).max()
    assert np.all(np.abs(parts[0]["av"][1:] - av_ref[1:]) <= 2e-5 * scale + 1e-9)
    ref.close()


def _av_worker(rank, world, port, outdir, phys):
    import faulthandler
    import torch
    faulthandler.dump_traceback_later(90, exit=True)
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    slb, drv = sh.import_pkg()
    p = ob.make_params(**phys)
    slb.configure(N=p.N, M=p.M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    torch.cuda.set_stream(torch.cuda.Stream())
    slb.lib().slb_set_stream(torch.cuda.current_stream().cuda_stream)
    run = drv.SlabRun(slb, rank, world)
    chain = drv.ExactAvChain(run, slots=4)          # fewer slots than iterations: buffers come back
    run.tiptoe()
    av = sh.run_av_plan(run, chain, lambda: run.eng.sol.s.t, np.float32(p.dt))
    np.save(os.path.join(outdir, f"av{rank}.npy"), av)
    run.close()
    slb.lib().slb_set_stream(None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,phys", [
    (1, PHYS), (2, PHYS), (3, PHYS),
    (3, dict(N=8, M=120001, B=0.8, omega=40.0, E_omega=0.3)),     # slabs wide enough for the parallel form of the sums
])
def test_exact_av_over_slabs_on_one_gpu(world, phys, tmp_path):
    """av() with the running sums handed from slab to slab (HIP kernels, `world` processes on
    one GPU): av_data on every rank has the bits of the oracle's av() over the whole lattice."""
    # always in processes of their own: the worker makes a new stream the current one
    mp.spawn(_av_worker, args=(world, sh.free_port(), str(tmp_path), phys), nprocs=world, join=True)
    av_ref = sh.reference_av_plan(ob, ob.make_params(**phys))
    for r in range(world):
        got = np.load(tmp_path / f"av{r}.npy")
        assert np.array_equal(got.view(np.uint32), av_ref.view(np.uint32)), (r, got, av_ref)


@pytest.mark.parametrize("world,lockstep", [(1, None), (3, None), (3, "0")])
def test_slab_program_writes_the_single_gpu_record(world, lockstep, tmp_path, solver_bin):
    """slab_program.py (display=4 over slabs; here `world` processes on one GPU over gloo)
    writes the single-GPU program's record, every column to the last digit: the lattices are
    the same and the running sums of av() are handed from slab to slab."""
    import subprocess
    args = ["display=4", "E_dc=1", "E_omega=0.1", "omega=100", "mu=5", "alpha=1", "B=0.5", "PhiYmin=-4",
            "PhiYmax=4", "dt=0.001", "n-harmonics=16", "g-grid=4001", "t-max=0.03", "quiet=1"]
    one = subprocess.run([solver_bin] + args + ["o=one.txt"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    script = os.path.join(sh.ROOT, sh.PKG, "slab_program.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(sh.free_port()), script] + args + ["o=slabs.txt"]
    env = {**os.environ, "SLB_DIST_BACKEND": "gloo"}
    if lockstep is not None:
        # iterations left in flight, as over RCCL (the averaging period here has fewer
        # iterations than the chain has slots, so no rank ever waits for a slot)
        env["SLB_AV_LOCKSTEP"] = lockstep
    many = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert many.returncode == 0, many.stdout + many.stderr
    a, b = (tmp_path / "one.txt").read_text().splitlines(), (tmp_path / "slabs.txt").read_text().splitlines()
    assert a == b and len(a) == 3


@pytest.mark.parametrize("world", [1, 3])
def test_slab_program_display77_trace(world, tmp_path, solver_bin):
    """display=77 over slabs: every record of the trace (sums over the CPU solver's flat index,
    averaging data rescaled in place at each record) equals the single-GPU program's."""
    import subprocess
    args = ["display=77", "E_dc=1", "E_omega=0.1", "omega=100", "mu=5", "alpha=1", "B=0.5", "PhiYmin=-4",
            "PhiYmax=4", "dt=0.001", "n-harmonics=16", "g-grid=4001", "t-max=0.05", "quiet=1"]
    one = subprocess.run([solver_bin] + args + ["o=one.txt"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    script = os.path.join(sh.ROOT, sh.PKG, "slab_program.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(sh.free_port()), script] + args + ["o=slabs.txt"]
    many = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300,
                          env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert many.returncode == 0, many.stdout + many.stderr
    a, b = (tmp_path / "one.txt").read_text().splitlines(), (tmp_path / "slabs.txt").read_text().splitlines()
    assert len(a) == 22                          # 11 records of header + data
    assert a == b
    pick = lambda s: [l for l in s.splitlines() if l.startswith("# t=")]
    assert pick(one.stdout) == pick(many.stdout)


@pytest.mark.parametrize("world,grid", [(1, "61"), (3, "61"), (3, "2500")])
def test_slab_program_display8_frame(world, grid, tmp_path, solver_bin):
    """display=8 over slabs: every slab forms the frame values of its columns, rank 0 writes
    frame.data; the file equals the single-GPU program's byte for byte."""
    import subprocess
    args = ["display=8", "E_dc=1", "E_omega=0.1", "omega=100", "mu=5", "alpha=1", "B=0.5", "PhiYmin=-4",
            "PhiYmax=4", "dt=0.001", "n-harmonics=8", "g-grid=" + grid, "t-max=0.01", "quiet=1", "o=out.txt"]
    (tmp_path / "one").mkdir(), (tmp_path / "many").mkdir()
    one = subprocess.run([solver_bin] + args, cwd=tmp_path / "one", capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    script = os.path.join(sh.ROOT, sh.PKG, "slab_program.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(sh.free_port()), script] + args
    many = subprocess.run(cmd, cwd=tmp_path / "many", capture_output=True, text=True, timeout=300,
                          env={**os.environ, "SLB_DIST_BACKEND": "gloo"})
    assert many.returncode == 0, many.stdout + many.stderr
    a, b = (tmp_path / "one" / "frame.data").read_bytes(), (tmp_path / "many" / "frame.data").read_bytes()
    assert len(a.splitlines()) == 629 * (int(grid) + 1) + 2
    assert a == b
    assert [l for l in one.stdout.splitlines() if "norm" in l] == [l for l in many.stdout.splitlines() if "norm" in l]


@pytest.mark.parametrize("world", [1, 3])
def test_slab_program_with_tree_sums(world, tmp_path, solver_bin):
    """The same with SLB_AV=tree (per-slab sums in double, all-reduced): the instantaneous
    columns and the norm are the single-GPU program's to the last digit, the period averages
    agree to the rounding noise of a float running sum."""
    import subprocess
    args = ["display=4", "E_dc=1", "E_omega=0.1", "omega=100", "mu=5", "alpha=1", "B=0.5", "PhiYmin=-4",
            "PhiYmax=4", "dt=0.001", "n-harmonics=16", "g-grid=4001", "t-max=0.03", "quiet=1"]
    one = subprocess.run([solver_bin] + args + ["o=one.txt"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=300, env={**os.environ, "SLB_AV": "tree"})
    assert one.returncode == 0, one.stdout + one.stderr
    script = os.path.join(sh.ROOT, sh.PKG, "slab_program.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(sh.free_port()), script] + args + ["o=slabs.txt"]
    many = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600,
                          env={**os.environ, "SLB_DIST_BACKEND": "gloo", "SLB_AV": "tree"})
    assert many.returncode == 0, many.stdout + many.stderr
    a, b = (tmp_path / "one.txt").read_text().splitlines(), (tmp_path / "slabs.txt").read_text().splitlines()
    assert len(a) == len(b) == 3 and a[:2] == b[:2]
    ca, cb = a[2].split(), b[2].split()
    exact = (0, 1, 2, 3, 4, 6, 7, 8)            # parameters, v_dr, NORM, v_y, m/m_x
    assert [ca[i] for i in exact] == [cb[i] for i in exact]
    for i in (5, 9, 10, 11, 12):                # A(omega), the three averages, Asin
        x, y = float(ca[i]), float(cb[i])
        assert abs(x - y) <= 2e-6 * max(abs(x), abs(y)) + 1e-9, (i, x, y)
