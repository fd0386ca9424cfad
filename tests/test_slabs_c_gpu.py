"""The slab time loop in C (host/slb_slabs.c, host/slb_link.c).  `boltzmann_solver ranks=N`
forks N processes; on a one-GPU box they share the GPU and talk over sockets (RCCL refuses
two ranks on one device), which runs the same launch order, the same halo traffic and the
same schedule of the av chain as a run over RCCL.  The record must be the single-GPU
program's, every column to the last digit: same lattices, and av()'s running sums handed
from slab to slab."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = ["display=4", "E_dc=1", "E_omega=0.1", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4", "quiet=1"]


@pytest.mark.parametrize("ranks,extra", [
    (2, ["omega=100", "B=0.5", "dt=0.001", "n-harmonics=16", "g-grid=4001", "t-max=0.03"]),    # 93 iterations
    (3, ["omega=100", "B=0.5", "dt=0.001", "n-harmonics=16", "g-grid=4001", "t-max=0.03"]),
    (5, ["omega=100", "B=0.5", "dt=0.001", "n-harmonics=8", "g-grid=1999", "t-max=0.0305"]),   # averaging starts between the two iterations of a launch
    (8, ["omega=40", "B=0.8", "dt=0.001", "n-harmonics=24", "g-grid=5003", "t-max=0.02"]),      # 178 iterations, 157 averaged: the chain's ring of batches turns over
    (3, ["omega=90", "B=0.03", "dt=0.0007", "n-harmonics=5", "g-grid=120001", "t-max=0.01"]),    # slabs wide enough for the parallel form of the sums
    (4, ["omega=100", "B=0.5", "dt=0.001", "n-harmonics=16", "g-grid=4001", "t-max=0.03", "E_omega=0"]),   # no averaging at all
    (2, ["omega=100", "B=0.0019", "dt=0.001", "n-harmonics=64", "g-grid=2000001", "t-max=0.003"]),   # slabs of more than one round of workgroups: whole column strips first
])
@pytest.mark.parametrize("per_launch", ["2", "3"])
def test_program_over_ranks_writes_the_single_gpu_record(ranks, extra, per_launch, tmp_path, solver_bin):
    """Two iterations per launch (the double step), and three (the LDS-ring kernel: six ghost
    columns' worth of halo, batches of the av chain of six lattices)."""
    args = BASE + extra
    one = subprocess.run([solver_bin] + args + ["o=one.txt"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    many = subprocess.run([solver_bin] + args + [f"ranks={ranks}", "o=slabs.txt"], cwd=tmp_path, capture_output=True,
                          text=True, timeout=300, env={**os.environ, "SLB_STEPS_PER_LAUNCH": per_launch})
    assert many.returncode == 0, many.stdout + many.stderr
    a, b = (tmp_path / "one.txt").read_text().splitlines(), (tmp_path / "slabs.txt").read_text().splitlines()
    assert len(a) == 3 and a == b


@pytest.mark.parametrize("per_launch", ["2", "3"])
@pytest.mark.parametrize("ranks,grid", [(1, "4001"), (3, "4001"), (4, "70001")])
def test_program_over_ranks_writes_the_single_gpu_trace(ranks, grid, per_launch, tmp_path, solver_bin):
    """display=77 over ranks: every record of the trace (sums over the CPU solver's flat index,
    averaging data rescaled in place at each record and handed back to the chain) and the norm
    log equal the single-GPU program's; single steps with a ghost refresh alternate with
    double steps here."""
    args = ["display=77", "E_dc=1", "E_omega=0.1", "omega=100", "mu=5", "alpha=1", "B=0.05", "PhiYmin=-4",
            "PhiYmax=4", "dt=0.001", "n-harmonics=16", "g-grid=" + grid, "t-max=0.05", "quiet=1"]
    one = subprocess.run([solver_bin] + args + ["o=one.txt"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    many = subprocess.run([solver_bin] + args + [f"ranks={ranks}", "o=slabs.txt"], cwd=tmp_path, capture_output=True,
                          text=True, timeout=300, env={**os.environ, "SLB_STEPS_PER_LAUNCH": per_launch,
                                                       **({"SLB_FORCE_SLABS": "1"} if ranks == 1 else {})})
    assert many.returncode == 0, many.stdout + many.stderr
    a, b = (tmp_path / "one.txt").read_text().splitlines(), (tmp_path / "slabs.txt").read_text().splitlines()
    assert len(a) == 22 and a == b
    pick = lambda s: [l for l in s.splitlines() if l.startswith("# t=")]
    assert pick(one.stdout) == pick(many.stdout) and len(pick(one.stdout)) == 11


@pytest.mark.parametrize("ranks,grid", [(3, "61"), (4, "2500")])
def test_program_over_ranks_writes_the_single_gpu_frame(ranks, grid, tmp_path, solver_bin):
    """display=8 over ranks: every slab forms the frame values of its columns, rank 0 writes
    frame.data; the file equals the single-GPU program's byte for byte."""
    args = ["display=8", "E_dc=1", "E_omega=0.1", "omega=100", "mu=5", "alpha=1", "B=0.5", "PhiYmin=-4",
            "PhiYmax=4", "dt=0.001", "n-harmonics=8", "g-grid=" + grid, "t-max=0.01", "quiet=1", "o=out.txt"]
    (tmp_path / "one").mkdir(), (tmp_path / "many").mkdir()
    one = subprocess.run([solver_bin] + args, cwd=tmp_path / "one", capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    many = subprocess.run([solver_bin] + args + [f"ranks={ranks}"], cwd=tmp_path / "many", capture_output=True,
                          text=True, timeout=300)
    assert many.returncode == 0, many.stdout + many.stderr
    a, b = (tmp_path / "one" / "frame.data").read_bytes(), (tmp_path / "many" / "frame.data").read_bytes()
    assert len(a.splitlines()) == 629 * (int(grid) + 1) + 2
    assert a == b
    assert [l for l in one.stdout.splitlines() if "norm" in l] == [l for l in many.stdout.splitlines() if "norm" in l]


def _one_and_many(solver_bin, tmp_path, args, ranks, stdin_text=None, env=None):
    (tmp_path / "one").mkdir(), (tmp_path / "many").mkdir()
    env = None if env is None else {**os.environ, **env}
    one = subprocess.run([solver_bin] + args, cwd=tmp_path / "one", capture_output=True, text=True, timeout=300,
                         input=stdin_text, env=env)
    assert one.returncode == 0, one.stdout + one.stderr
    many = subprocess.run([solver_bin] + args + [f"ranks={ranks}"], cwd=tmp_path / "many", capture_output=True,
                          text=True, timeout=300, input=stdin_text, env=env)
    assert many.returncode == 0, many.stdout + many.stderr
    return one, many


@pytest.mark.parametrize("per_launch", ["2", "3"])
def test_narrowest_slabs_and_the_refusal_below_them(per_launch, tmp_path, solver_bin):
    """Slabs of exactly twelve columns (two edge ranges of six, no interior left) give the single-GPU
    record; narrower ones are refused by every rank with the reason, and the program ends non-zero
    instead of waiting for anyone."""
    args = BASE + ["omega=100", "B=0.5", "dt=0.001", "n-harmonics=8", "g-grid=36", "t-max=0.02", "o=out.txt"]
    one, many = _one_and_many(solver_bin, tmp_path, args, 3, env={"SLB_STEPS_PER_LAUNCH": per_launch})
    assert (tmp_path / "one" / "out.txt").read_bytes() == (tmp_path / "many" / "out.txt").read_bytes()
    bad = subprocess.run([solver_bin] + args + ["ranks=4"], cwd=tmp_path / "many", capture_output=True, text=True,
                         timeout=120, env={**os.environ, "SLB_STEPS_PER_LAUNCH": per_launch})
    assert bad.returncode != 0 and "at least 12 columns" in bad.stderr


MODES = ["E_dc=1", "E_omega=0.1", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4", "dt=0.001", "quiet=1", "o=out.txt"]


@pytest.mark.parametrize("ranks,grid", [(2, "37"), (3, "500")])
def test_program_over_ranks_writes_the_single_gpu_movie(ranks, grid, tmp_path, solver_bin):
    """display=7 over ranks: a frame file every 0.01 of time once t > frame-start (single steps
    with a ghost refresh where a frame follows, double steps in between), same names, same bytes."""
    args = ["display=7", "omega=100", "B=0.5", "n-harmonics=6", "g-grid=" + grid, "t-max=0.02", "frame-start=0.03"] + MODES
    one, many = _one_and_many(solver_bin, tmp_path, args, ranks)
    names = sorted(f.name for f in (tmp_path / "one").glob("frame*.data"))
    assert names == sorted(f.name for f in (tmp_path / "many").glob("frame*.data")) and len(names) >= 4
    for n in names:
        assert (tmp_path / "one" / n).read_bytes() == (tmp_path / "many" / n).read_bytes(), n
    pick = lambda s: [l for l in s.splitlines() if l.startswith("Writing") or "norm" in l]
    assert pick(one.stdout) == pick(many.stdout)


def test_program_over_ranks_writes_the_single_gpu_strobes(tmp_path, solver_bin):
    """display=9 over ranks: 101 periods, the strobe image accumulated on rank 0."""
    args = ["display=9", "omega=400", "B=0.5", "n-harmonics=4", "g-grid=29", "t-max=0.01"] + MODES
    one, many = _one_and_many(solver_bin, tmp_path, args, 2)
    names = sorted(f.name for f in (tmp_path / "one").glob("strobe*.data"))
    assert names == sorted(f.name for f in (tmp_path / "many").glob("strobe*.data")) and len(names) == 101
    for n in (names[0], names[1], names[50], names[-1]):
        assert (tmp_path / "one" / n).read_bytes() == (tmp_path / "many" / n).read_bytes(), n


@pytest.mark.parametrize("ranks,grid", [(2, "61"), (4, "1000")])
def test_program_over_ranks_writes_the_single_gpu_dump(ranks, grid, tmp_path, solver_bin):
    """display=3 over ranks: the final distribution next to the equilibrium one, and the norm."""
    args = ["display=3", "omega=100", "B=0.5", "n-harmonics=8", "g-grid=" + grid, "t-max=0.01"] + MODES
    one, many = _one_and_many(solver_bin, tmp_path, args, ranks)
    a, b = (tmp_path / "one" / "out.txt").read_bytes(), (tmp_path / "many" / "out.txt").read_bytes()
    assert len(a.splitlines()) == 629 * (int(grid) - 1) + 1 and a == b
    assert [l for l in one.stdout.splitlines() if "norm" in l] == [l for l in many.stdout.splitlines() if "norm" in l]


@pytest.mark.parametrize("mode", [None, "cpu", "fixed"])
def test_program_over_ranks_goes_on_with_new_parameters(mode, tmp_path, solver_bin):
    """read-from=stdin over ranks: rank 0 reads the new value and tells the others; every
    record equals the single-GPU program's in each reading of what a new value does."""
    args = ["display=4", "omega=100", "B=0.5", "n-harmonics=8", "g-grid=1001", "t-max=0.02", "read-from=stdin"] + MODES
    records = "E_dc 1.5 0.02\nB 0.4 0.013\nomega 50 0.01\nbogus 1 0.005\nexit\n"
    one, many = _one_and_many(solver_bin, tmp_path, args, 3, stdin_text=records,
                              env=None if mode is None else {"SLB_REPARAM": mode})
    a, b = (tmp_path / "one" / "out.txt").read_text(), (tmp_path / "many" / "out.txt").read_text()
    assert len(a.splitlines()) == 15 and a == b


def test_program_over_ranks_refuses_what_it_cannot_do(tmp_path, solver_bin):
    args = ["E_dc=1", "E_omega=0.1", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4", "omega=100", "B=0.5",
            "n-harmonics=8", "g-grid=500", "t-max=0.01", "o=stdout"]
    r = subprocess.run([solver_bin, "display=4"] + args + ["ranks=0"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1 and "Invalid value of ranks=" in r.stderr
    r = subprocess.run([solver_bin, "display=77", "ranks=2"] + [a for a in args if not a.startswith("n-harm")] +
                       ["n-harmonics=1"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "display=77 needs n-harmonics >= 2" in r.stderr


def test_single_rank_loop_in_c_matches_the_oracle(slb, oracle):
    """slb_slabs_* with one rank (no link): double steps with av rows, a last single step,
    av_finish -- lattices and averaging data bit-identical to the CPU oracle."""
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    N, M = 12, 3001
    p = oracle.make_params(N=N, M=M, B=0.7, omega=30.0, E_omega=0.2)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    L = slb.lib()
    orc = oracle.OracleState(p)
    r = L.slb_slabs_create(0, 1, None)
    assert r
    try:
        L.slb_slabs_tiptoe(r)
        s = L.slb_slabs_solver(r).contents
        av_cpu = np.zeros(6, np.float32)
        def cpu_av():
            cur = orc.st.current
            oracle.lib().orc_av(C.byref(p), orc.array("a", cur), orc.array("b", cur), av_cpu, t_k)
        for k in range(11):                      # 11 double steps: more than two batches of the chain
            rows = (C.c_void_p * 2)()
            L.slb_slabs_av_rows(r, 2, rows)
            tt = (C.c_float * 2)(s.t, np.float32(np.float32(s.t) + np.float32(p.dt)))
            L.slb_slabs_double_step(r, rows[0], rows[1])
            L.slb_slabs_av_submit(r, 2, tt, 0)
            for t_k in tt:
                t_k = np.float32(t_k)
                orc.step(1)
                cpu_av()
        for k in range(7):                       # launches of three; of the first only the last two lattices are averaged
            rows = (C.c_void_p * 3)()
            L.slb_slabs_av_rows(r, 3, rows)
            t1 = np.float32(s.t)
            t2 = np.float32(t1 + np.float32(p.dt))
            t3 = np.float32(t2 + np.float32(p.dt))
            L.slb_slabs_multi_step(r, 3, rows)
            skip = 1 if k == 0 else 0
            tt = (C.c_float * (3 - skip))(*[t1, t2, t3][skip:])
            L.slb_slabs_av_submit(r, 3 - skip, tt, skip)
            for i, t_k in enumerate((t1, t2, t3)):
                orc.step(1)
                if i >= skip:
                    cpu_av()
        t_k = np.float32(s.t)
        L.slb_slabs_last_step(r)
        L.slb_slabs_av_submit_current(r, t_k)
        orc.step(1)
        cpu_av()
        av6 = (C.c_float * 6)()
        L.slb_slabs_av_finish(r, C.byref(av6))
        torch.cuda.synchronize()
        assert np.array_equal(np.array(av6, dtype=np.float32).view(np.uint32), av_cpu.view(np.uint32))
        assert np.float32(s.t) == np.float32(orc.st.t)
        total = np.empty((N + 1, M + 3), np.float32)
        for which, arr, idx in (("a", 1 + s.current, orc.st.current), ("b", 5 + s.current, orc.st.current),
                                ("a", 1 + s.current_hs, orc.st.current_hs), ("b", 5 + s.current_hs, orc.st.current_hs)):
            L.slb_slabs_gather_rows(r, arr, 0, N + 1, total.ctypes.data)
            assert np.array_equal(total.view(np.uint32), orc.array(which, idx).view(np.uint32)), which
    finally:
        L.slb_slabs_destroy(r)
        orc.close()


@pytest.mark.parametrize("ranks,plan", [
    (2, "dsdsddsl"),        # a double step right after a single step: the spares are re-copied while the side stream is busy
    (3, "sdsdddssdl"),
    (2, "tttdtdstl"),       # launches of three write a/b[next]; a double step after an odd number of them re-copies the spares
    (4, "ttsttdttl"),
])
def test_c_loop_over_ranks_in_any_order_of_steps(ranks, plan, tmp_path, oracle):
    """slb_slabs_double_step() after slb_slabs_step() on several ranks (sockets, one GPU): the
    spare arrays are copied again on the main stream while the side stream still has halo work,
    and the side stream's next edge launches must wait for those copies.  Every cell of both
    current lattices, gathered over the ranks, equals the single-domain oracle's."""
    import socket
    import sys
    N, M, B, omega = 12, 6007, 0.7, 30.0
    pairs = {}
    for i in range(ranks):
        for j in range(i + 1, ranks):
            pairs[(i, j)] = socket.socketpair()
    procs = []
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slab_c_worker.py")
    for q in range(ranks):
        fds = [-1] * ranks
        for (i, j), (si, sj) in pairs.items():
            if i == q:
                fds[j] = si.fileno()
            elif j == q:
                fds[i] = sj.fileno()
        cmd = [sys.executable, worker, str(q), str(ranks), str(tmp_path), plan, str(N), str(M), str(B), str(omega)]
        procs.append(subprocess.Popen(cmd + [str(f) for f in fds], pass_fds=[f for f in fds if f >= 0],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for si, sj in pairs.values():
        si.close(), sj.close()
    outs = [p_.communicate(timeout=200)[0] for p_ in procs]
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(outs)
    steps = sum({"t": 3, "d": 2}.get(c, 1) for c in plan)
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    orc = oracle.OracleState(p)
    try:
        orc.step(steps)
        got = np.load(tmp_path / "gathered.npz")
        assert np.float32(got["t"]) == np.float32(orc.st.t)
        assert list(got["idx"]) == [orc.st.current, orc.st.current_hs]
        for name, which, idx in (("a_main", "a", orc.st.current), ("b_main", "b", orc.st.current),
                                 ("a_half", "a", orc.st.current_hs), ("b_half", "b", orc.st.current_hs)):
            assert np.array_equal(got[name].view(np.uint32), orc.array(which, idx).view(np.uint32)), name
    finally:
        orc.close()


def test_a_failing_rank_ends_the_run(tmp_path, solver_bin):
    """A rank that ends abnormally takes the run down with a non-zero status instead of leaving
    its neighbours waiting for its messages for good (SLB_FAIL_RANK makes rank 1 end after the
    first half step, the way any of the driver's error paths would)."""
    args = BASE + ["omega=100", "B=0.5", "dt=0.001", "n-harmonics=8", "g-grid=2001", "t-max=0.01", "ranks=3", "o=out.txt"]
    r = subprocess.run([solver_bin] + args, cwd=tmp_path, capture_output=True, text=True, timeout=120,
                       env={**os.environ, "SLB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 ended with status" in r.stderr, r.stderr


def test_rccl_link_moves_a_message_on_this_gpu(slb):
    """The RCCL transport itself, as far as one GPU goes: librccl is found and opened at run
    time, a communicator of one rank comes up from an identifier made here, and a grouped
    ncclSend/ncclRecv pair addressed to the rank itself moves a device buffer on a stream."""
    import torch
    torch.cuda.set_device(0)
    L = slb.lib()
    ident = C.create_string_buffer(128)
    assert L.slb_link_unique_id(ident) == 0
    link = L.slb_link_open_rccl(0, 1, ident.raw)
    assert link
    try:
        assert L.slb_link_ranks_seen(link) == 1 and L.slb_link_transport(link) == b"rccl"
        st = torch.cuda.Stream()
        src = torch.arange(5000, dtype=torch.float32, device="cuda") * 0.5
        dst = torch.zeros_like(src)
        st.wait_stream(torch.cuda.current_stream())
        for n in (5000, 7):
            L.slb_link_begin(link)
            L.slb_link_send(link, 0, src.data_ptr(), n, st.cuda_stream)
            L.slb_link_recv(link, 0, dst.data_ptr(), n, st.cuda_stream)
            L.slb_link_end(link, st.cuda_stream)
            st.synchronize()
            assert torch.equal(dst[:n], src[:n])
            dst.zero_()
            torch.cuda.synchronize()
    finally:
        L.slb_link_close(link)


def test_python_wrapper_of_the_c_loop_steps_like_the_engine(slb, oracle):
    """dist.CSlabRun (what bench.py drives at N > 1) with one rank: tiptoe, double steps and a
    last step leave the oracle's lattices, and torch's current stream is the loop's main stream."""
    import importlib
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    drv = importlib.import_module(slb.__name__ + ".dist")
    N, M = 10, 2999
    p = oracle.make_params(N=N, M=M, B=0.6, omega=20.0)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    run = drv.CSlabRun(slb, 0, 1)
    try:
        assert run.ranks_seen() == 1
        assert torch.cuda.current_stream().cuda_stream == slb.lib().slb_slabs_main_stream(run.r)
        run.tiptoe()
        for _ in range(2):
            run.double_step()
        run.step_fused()                      # an odd count in between: the spares follow
        for _ in range(2):
            run.double_step()
        run.last_step()
        orc.step(10)
        torch.cuda.synchronize()
        s = run.s
        total = np.empty((N + 1, M + 3), np.float32)
        for which, arr, idx in (("a", 1 + s.current, orc.st.current), ("b", 5 + s.current, orc.st.current),
                                ("a", 1 + s.current_hs, orc.st.current_hs), ("b", 5 + s.current_hs, orc.st.current_hs)):
            slb.lib().slb_slabs_gather_rows(run.r, arr, 0, N + 1, total.ctypes.data)
            assert np.array_equal(total.view(np.uint32), orc.array(which, idx).view(np.uint32)), which
    finally:
        run.close()
        orc.close()
