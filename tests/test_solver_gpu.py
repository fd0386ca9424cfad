"""The C host engine and the boltzmann_solver program vs the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ARR = {("a0", 0): 0, **{("a", i): 1 + i for i in range(4)}, **{("b", i): 5 + i for i in range(4)}}


def _compare_state(sol, orc, M):
    s = sol.s
    assert (s.current, s.next, s.current_hs, s.next_hs) == \
           (orc.st.current, orc.st.next, orc.st.current_hs, orc.st.next_hs)
    for (which, i), arr in ARR.items():
        got = sol.rows(arr)[:, :M + 3]
        exp = orc.array(which, i)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (which, i)


@pytest.mark.parametrize("N,M,B,omega,steps", [(32, 4096, 1.0, 1.0, 60), (8, 61, 0.05, 100.0, 25),
                                               (64, 20001, 0.2, 10.0, 12)])
def test_engine_fields_bit_exact(slb, oracle, N, M, B, omega, steps):
    """init (a0, a[0]<-a0), tiptoe and `steps` loop iterations: all nine arrays
    and the ping-pong indices equal the oracle's, bit for bit."""
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    try:
        sol.tiptoe()
        _compare_state(sol, orc, M)
        done = 0
        for chunk in (1, 1, steps - 2):
            for _ in range(chunk):
                sol.step()
            orc.step(chunk)
            done += chunk
            _compare_state(sol, orc, M)
        assert sol.s.t == orc.st.t and sol.s.steps == done
    finally:
        sol.close()
        orc.close()


def _run_both(solver_bin, oracle, tmp_path, args, extra_files=(), stdin_text=None, env=None):
    gdir, odir = tmp_path / "gpu", tmp_path / "orc"
    gdir.mkdir(), odir.mkdir()
    g = subprocess.run([solver_bin] + args + ["o=out.txt"], cwd=gdir, capture_output=True, text=True, timeout=600,
                       input=stdin_text, env=None if env is None else {**os.environ, **env})
    assert g.returncode == 0, g.stdout + g.stderr
    o = oracle.run_cli(args + ["o=out.txt"], cwd=odir, stdin_text=stdin_text, env=env)
    return gdir, odir, g, o


COMMON = ["E_dc=1", "E_omega=0.1", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4", "dt=0.001"]


PER_LAUNCH = pytest.mark.parametrize("per_launch", [None, "3"], ids=["default", "three-per-launch"])


def _per_launch_env(per_launch, env=None):
    """The environment that forces that many time steps into each launch wherever the program's
    schedule allows it (three is otherwise chosen only on wide lattices)."""
    env = dict(env or {})
    if per_launch is not None:
        env["SLB_STEPS_PER_LAUNCH"] = per_launch
    return env or None


def test_program_display4_identical(solver_bin, oracle, tmp_path):
    args = ["display=4", "omega=100", "B=1", "n-harmonics=32", "g-grid=4096", "t-max=0.05"] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args)
    assert (gdir / "out.txt").read_text() == (odir / "out.txt").read_text()
    assert "# t_max = " in g.stdout and "# norm=" in g.stdout


def test_program_output_forms(solver_bin, oracle, tmp_path):
    """o=+file appends to an existing file (reference: src/boltzmann_cli.c:171-173), o=stdout
    writes the record between the log lines, and a file that cannot be opened ends the run
    with the reference's message."""
    args = ["display=4", "omega=100", "B=0.5", "n-harmonics=8", "g-grid=253", "t-max=0.01", "quiet=1"] + COMMON
    for prog_dir in ("gpu", "orc"):
        (tmp_path / prog_dir).mkdir()
        (tmp_path / prog_dir / "log.txt").write_text("# kept line\n")
    for e_dc in ("1", "1.25"):
        a = [x for x in args if not x.startswith("E_dc=")] + ["E_dc=" + e_dc, "o=+log.txt"]
        g = subprocess.run([solver_bin] + a, cwd=tmp_path / "gpu", capture_output=True, text=True, timeout=300)
        assert g.returncode == 0, g.stdout + g.stderr
        oracle.run_cli(a, cwd=tmp_path / "orc")
    got, exp = (tmp_path / "gpu" / "log.txt").read_text(), (tmp_path / "orc" / "log.txt").read_text()
    assert got == exp and got.startswith("# kept line\n") and len(got.splitlines()) == 7
    g = subprocess.run([solver_bin] + args + ["o=stdout"], cwd=tmp_path / "gpu", capture_output=True, text=True, timeout=300)
    assert g.returncode == 0, g.stdout + g.stderr
    lines = g.stdout.splitlines()
    assert all(l in lines for l in exp.splitlines()[1:4])          # the first record, on stdout this time
    bad = subprocess.run([solver_bin] + args + ["o=no_such_dir/x.txt"], cwd=tmp_path / "gpu", capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'ERROR: Faled to open file "no_such_dir/x.txt" for writing.' in bad.stderr


@pytest.mark.parametrize("tmax", ["0.012", "0.011", "0.013"])     # averaging starts on the first / second / third iteration of a launch
@pytest.mark.parametrize("env", [None, {"SLB_FORM": "fused"}, {"SLB_AV": "chain"}, {"SLB_STEPS_PER_LAUNCH": "3"},
                                 {"SLB_STEPS_PER_LAUNCH": "3", "SLB_AV": "tree"}])
def test_program_display4_wide_lattice_parallel_exact_sums(solver_bin, oracle, tmp_path, env, tmax):
    """display=4 on a lattice wide enough for the parallel form of the exact sums, fed from the
    rows the double-step kernel hands out and summed on the side streams: the record still
    equals the oracle's to the last digit."""
    args = ["display=4", "omega=100", "B=0.1", "n-harmonics=8", "g-grid=70001", "t-max=" + tmax] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, env=env)
    if env and env.get("SLB_AV") == "tree":          # (three iterations per launch with the tree sums: the noisy columns aside)
        a, b = (gdir / "out.txt").read_text().splitlines(), (odir / "out.txt").read_text().splitlines()
        ca, cb = a[2].split(), b[2].split()
        assert a[:2] == b[:2] and [ca[i] for i in (0, 1, 2, 3, 4, 6, 7, 8)] == [cb[i] for i in (0, 1, 2, 3, 4, 6, 7, 8)]
        return
    assert (gdir / "out.txt").read_text() == (odir / "out.txt").read_text()


def test_program_display4_tree_sums_within_rounding_noise(solver_bin, oracle, tmp_path):
    """SLB_AV=tree: same record except that the period averages carry the difference between
    a tree sum in double and the CPU's sequential float sum."""
    args = ["display=4", "omega=100", "B=0.1", "n-harmonics=8", "g-grid=70001", "t-max=0.012"] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, env={"SLB_AV": "tree"})
    a, b = (gdir / "out.txt").read_text().splitlines(), (odir / "out.txt").read_text().splitlines()
    assert a[:2] == b[:2]
    ca, cb = a[2].split(), b[2].split()
    assert [ca[i] for i in (0, 1, 2, 3, 4, 6, 7, 8)] == [cb[i] for i in (0, 1, 2, 3, 4, 6, 7, 8)]
    scale = max(abs(float(cb[i])) for i in (9, 10, 11))
    for i in (5, 9, 10, 11, 12):
        assert abs(float(ca[i]) - float(cb[i])) <= 1e-4 * scale + 1e-9, (i, ca[i], cb[i])   # the float sum is that noisy


@PER_LAUNCH
def test_program_display77_identical(solver_bin, oracle, tmp_path, per_launch):
    """display=77 trace on a shrunken config-2 shape: every record identical,
    including the flat-index sums and the in-place rescaling of the averages."""
    args = ["display=77", "omega=100", "B=0.12", "n-harmonics=64", "g-grid=32768", "t-max=0.05"] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, env=_per_launch_env(per_launch))
    got, exp = (gdir / "out.txt").read_text(), (odir / "out.txt").read_text()
    assert len(exp.splitlines()) == 22          # 11 records of header + data
    assert got == exp
    # the "# t=... norm=..." log lines agree as well
    pick = lambda s: [l for l in s.splitlines() if l.startswith("# t=")]
    assert pick(g.stdout) == pick(o.stdout)


def test_program_display77_config2_full_size_identical(solver_bin, oracle, tmp_path):
    """The second benchmark configuration at its full size (display=77, n-harmonics=64, g-grid=262144, B=0.015,
    113 iterations, 11 records; reference: src/boltzmann_c_solver.c:178-186, 298-331): all 22
    lines of the trace and the norm log equal the CPU oracle's, which runs the same lattice on
    the host's cores."""
    args = ["display=77", "omega=100", "B=0.015", "n-harmonics=64", "g-grid=262144", "t-max=0.05"] + COMMON
    gdir, odir = tmp_path / "gpu", tmp_path / "orc"
    gdir.mkdir(), odir.mkdir()
    g = subprocess.run([solver_bin] + args + ["o=out.txt"], cwd=gdir, capture_output=True, text=True, timeout=600)
    assert g.returncode == 0, g.stdout + g.stderr
    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 16
    o = oracle.run_cli(args + ["o=out.txt"], cwd=odir, threads=threads, timeout=900)
    got, exp = (gdir / "out.txt").read_text(), (odir / "out.txt").read_text()
    assert len(exp.splitlines()) == 22
    assert got == exp
    pick = lambda s: [l for l in s.splitlines() if l.startswith("# t=")]
    assert pick(g.stdout) == pick(o.stdout) and len(pick(o.stdout)) == 11


def test_program_display4_config3_shape_averaged_throughout(solver_bin, oracle, tmp_path):
    """The headline benchmark shape as a program (display=4, n-harmonics=128, g-grid=1048576,
    omega=100, t-max=0.001: 64 iterations, all but the first averaged): the double-step kernel
    with the row buffers under the whole-strips-first schedule feeds the parallel exact sums,
    and the record -- instantaneous sums, norm, period averages, absorption -- equals the CPU
    oracle's to the last digit (reference: src/boltzmann_c_solver.c:164-194, 236-268, 413-437).
    The oracle runs the same lattice on the host's cores (about 25 s on 16 threads)."""
    args = ["display=4", "omega=100", "B=0.0038", "n-harmonics=128", "g-grid=1048576", "t-max=0.001"] + COMMON
    odir = tmp_path / "orc"
    odir.mkdir()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 16
    oracle.run_cli(args + ["o=out.txt"], cwd=odir, threads=threads, timeout=900)
    exp = (odir / "out.txt").read_text()
    assert len(exp.splitlines()) == 3
    # the default on this lattice: three iterations per launch (the LDS-ring kernel handing out the rows
    # of all three); then two per launch
    for form in ("", "double"):
        gdir = tmp_path / ("gpu" + form)
        gdir.mkdir()
        g = subprocess.run([solver_bin] + args + ["o=out.txt"], cwd=gdir, capture_output=True, text=True, timeout=600,
                           env={**os.environ, "SLB_TIMING": "1", "SLB_FORM": form, "SLB_SHOW_SCHEDULE": "1"})
        assert g.returncode == 0, g.stdout + g.stderr
        assert "64 iterations" in g.stderr, g.stderr
        assert ("slb schedule: S=3" in g.stderr) == (form == ""), g.stderr
        assert (gdir / "out.txt").read_text() == exp, form


@PER_LAUNCH
def test_program_display8_identical(solver_bin, oracle, tmp_path, per_launch):
    args = ["display=8", "omega=100", "B=0.5", "n-harmonics=8", "g-grid=61", "t-max=0.01"] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, env=_per_launch_env(per_launch))
    got, exp = (gdir / "frame.data").read_text(), (odir / "frame.data").read_text()
    assert len(exp.splitlines()) == 629 * 62 + 2
    assert got == exp


@pytest.mark.parametrize("form,mode", [("double", None), ("fused", None), ("sweeps", None), ("triple", None), ("triple", "fixed"),
                                       ("double", "fixed"), ("double", "cpu"), ("sweeps", "cuda")])
def test_program_read_from_stdin_goes_on_with_new_parameters(solver_bin, oracle, tmp_path, form, mode):
    """read-from=stdin: after each display=4 record one parameter is replaced and the run
    continues from the present state (new E_dc, new B, new omega with the old period still in
    t_max); every record equals the oracle's, in each of the three readings of "replaced"
    (SLB_REPARAM: the reference CUDA host's -- the default: a new B without its derived
    constant --, the reference CPU solver's -- the value is ignored --, and the repaired one).
    No reference output pins this path (the reference holds none): parity unpinned."""
    args = ["display=4", "omega=100", "B=0.5", "n-harmonics=8", "g-grid=253", "t-max=0.02",
            "read-from=stdin"] + COMMON
    records = "E_dc 1.5 0.02\nB 0.4 0.013\nomega 50 0.01\nbogus 1 0.005\nexit\n"
    env = {}
    if form == "triple":
        env["SLB_STEPS_PER_LAUNCH"] = "3"
    elif form != "double":
        env["SLB_FORM"] = form
    if mode is not None:
        env["SLB_REPARAM"] = mode
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, stdin_text=records, env=env or None)
    got, exp = (gdir / "out.txt").read_text(), (odir / "out.txt").read_text()
    assert len(exp.splitlines()) == 15            # five records of three lines
    assert got == exp
    pick = lambda s: [l for l in s.splitlines() if l.startswith("# t_max")]
    assert pick(g.stdout) == pick(o.stdout) and len(pick(o.stdout)) == 5
    rec = [l.split() for l in exp.splitlines() if not l.startswith("#")]
    if mode == "cpu":                             # the new values never show, not even in the record
        assert all(r[0] == rec[0][0] and r[2] == rec[0][2] for r in rec)
    else:
        assert float(rec[1][0]) == 1.5 and float(rec[3][2]) == 50.0


@PER_LAUNCH
def test_program_display7_movie_frames_identical(solver_bin, oracle, tmp_path, per_launch):
    """display=7: a frame file every 0.01 of time once t > frame-start, same names and text."""
    args = ["display=7", "omega=100", "B=0.5", "n-harmonics=6", "g-grid=37", "t-max=0.02", "frame-start=0.03"] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, env=_per_launch_env(per_launch))
    names = sorted(f.name for f in odir.glob("frame*.data"))
    assert names == sorted(f.name for f in gdir.glob("frame*.data")) and len(names) >= 4
    for n in names:
        assert (gdir / n).read_text() == (odir / n).read_text(), n
    assert [l for l in g.stdout.splitlines() if l.startswith("Writing")] == \
           [l for l in o.stdout.splitlines() if l.startswith("Writing")]


@PER_LAUNCH
def test_program_display9_strobe_identical(solver_bin, oracle, tmp_path, per_launch):
    """display=9: 101 periods, one accumulated strobe image per period; the last file
    carries the sum of all of them."""
    args = ["display=9", "omega=100", "B=0.5", "n-harmonics=4", "g-grid=13", "t-max=0.02"] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, env=_per_launch_env(per_launch))
    names = sorted(f.name for f in odir.glob("strobe*.data"))
    assert names == sorted(f.name for f in gdir.glob("strobe*.data")) and len(names) == 101
    for n in (names[0], names[1], names[50], names[-1]):
        assert (gdir / n).read_text() == (odir / n).read_text(), n
    # display=9 keeps averaging as display=4 does, but prints no record
    assert (gdir / "out.txt").read_text() == (odir / "out.txt").read_text() == ""


@PER_LAUNCH
def test_program_display3_dump_identical(solver_bin, oracle, tmp_path, per_launch):
    """display=3: the final distribution and the equilibrium one, columns 1..M-1."""
    args = ["display=3", "omega=100", "B=0.5", "n-harmonics=8", "g-grid=61", "t-max=0.01"] + COMMON
    gdir, odir, g, o = _run_both(solver_bin, oracle, tmp_path, args, env=_per_launch_env(per_launch))
    got, exp = (gdir / "out.txt").read_text(), (odir / "out.txt").read_text()
    assert len(exp.splitlines()) == 629 * 60 + 1
    assert got == exp


@pytest.mark.parametrize("N,M,B,omega,steps,n_chunk", [
    (1, 5, 0.5, 3.0, 3, 0), (2, 61, 0.05, 100.0, 6, 0), (7, 300, 0.4, 10.0, 5, 3),
    (32, 4096, 1.0, 1.0, 20, 0), (33, 1000, 0.5, 30.0, 7, 5), (64, 20001, 0.2, 10.0, 6, 16),
    (16, 247, 0.3, 10.0, 4, 0), (16, 248, 0.3, 10.0, 4, 8), (16, 249, 0.3, 10.0, 4, 0),
    (128, 70000, 0.06, 50.0, 3, 32),
])
def test_fused_step_bit_exact(slb, oracle, N, M, B, omega, steps, n_chunk):
    """The one-launch step (both sweeps fused) leaves all nine arrays identical
    to the oracle's, step after step, including every cell no sweep writes."""
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    try:
        sol.tiptoe()
        for variant in (0, 7):
            L.slb_set_tuning(variant, n_chunk, 0)
            for _ in range(steps):
                sol.step_fused()
            orc.step(steps)
            _compare_state(sol, orc, M)
        # mixing the two forms is allowed at any step
        L.slb_set_tuning(0, 0, 0)
        sol.step(); sol.step_fused(); sol.step()
        orc.step(3)
        _compare_state(sol, orc, M)
    finally:
        L.slb_set_tuning(0, 0, 0)
        sol.close()
        orc.close()


def test_program_config1_matches_recorded_reference_line(solver_bin, tmp_path):
    """The reference's own CPU-runnable case end to end on the GPU (16 281 steps, av every step of the
    last period): the display=4 line equals, digit for digit, the line the
    compiled reference CPU solver printed when it was surveyed."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                       "config1_display4_reference.json")))
    for form in ("double", "fused", "sweeps"):
        r = subprocess.run([solver_bin] + gold["args"].split() + ["o=out.txt"], cwd=tmp_path,
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, SLB_FORM=form))
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"# t_max = {gold['t_max']}" in r.stdout
        lines = (tmp_path / "out.txt").read_text().splitlines()
        assert len(lines) == 3
        v = lines[2].split()
        got = dict(zip(["v_dr/v_p", "A(omega)", "NORM", "v_y/v_p", "m/m_x", "<v_dr/v_p>", "<v_y/v_p>",
                        "<m/m_x>", "Asin"], v[4:]))
        assert got == gold["columns"], form


def _compare_live_state(sol, orc, M):
    """After a double step the lattices at t+dt and t+3dt/2 were never stored, so
    the `next` buffers hold older data than the oracle's; nothing ever reads that
    data again (the next sweep overwrites it).  What must agree: a0, the current
    lattices in full, and the cells of the `next` buffers that no sweep writes."""
    s = sol.s
    assert (s.current, s.next, s.current_hs, s.next_hs) == \
           (orc.st.current, orc.st.next, orc.st.current_hs, orc.st.next_hs)
    N = s.N
    for which, i, arr in (("a0", 0, 0), ("a", s.current, 1 + s.current), ("b", s.current, 5 + s.current),
                          ("a", s.current_hs, 1 + s.current_hs), ("b", s.current_hs, 5 + s.current_hs)):
        got, exp = sol.rows(arr)[:, :M + 3], orc.array(which, i)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (which, i)
    for which, i, arr, hs in (("a", s.next, 1 + s.next, False), ("b", s.next, 5 + s.next, False),
                              ("a", s.next_hs, 1 + s.next_hs, True), ("b", s.next_hs, 5 + s.next_hs, True)):
        got, exp = sol.rows(arr)[:, :M + 3].view(np.uint32), orc.array(which, i).view(np.uint32)
        assert np.array_equal(got[N], exp[N]), (which, i, "row N")
        assert np.array_equal(got[:, 0], exp[:, 0]) and np.array_equal(got[:, M + 2], exp[:, M + 2]), (which, i)
        if hs:
            assert np.array_equal(got[:, M + 1], exp[:, M + 1]), (which, i, "column M+1")
        if which == "b":
            assert np.array_equal(got[0], exp[0]), (which, i, "row 0")


@pytest.mark.parametrize("N,M,B,omega,pairs,n_chunk", [
    (1, 5, 0.5, 3.0, 2, 0), (2, 61, 0.05, 100.0, 3, 0), (3, 61, 0.05, 100.0, 3, 2), (4, 300, 0.4, 10.0, 3, 3),
    (7, 300, 0.4, 10.0, 3, 4), (32, 4096, 1.0, 1.0, 10, 0), (33, 1000, 0.5, 30.0, 4, 5),
    (64, 20001, 0.2, 10.0, 3, 16), (16, 247, 0.3, 10.0, 2, 0), (16, 248, 0.3, 10.0, 2, 8),
    (16, 249, 0.3, 10.0, 2, 1), (128, 70000, 0.06, 50.0, 2, 32),
])
def test_double_step_bit_exact(slb, oracle, N, M, B, omega, pairs, n_chunk):
    """Two loop iterations per launch: after every pair of steps all nine arrays,
    the ping-pong indices and the loop time equal the oracle's, bit for bit."""
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    try:
        sol.tiptoe()
        for variant in (0, 20, 21, 23, 24, 25, 26, 27, 29, 30, 34, 38, 39, 50, 51, 60, 61, 62, 63):
            L.slb_set_tuning(variant, n_chunk, 0)
            for _ in range(pairs):
                sol.double_step()
            orc.step(2 * pairs)
            _compare_live_state(sol, orc, M)
            assert sol.s.t == orc.st.t
        # an odd step in between changes which buffers are current: the spares follow
        L.slb_set_tuning(0, 0, 0)
        sol.step_fused(); sol.double_step(); sol.step(); sol.double_step()
        orc.step(6)
        _compare_live_state(sol, orc, M)
    finally:
        L.slb_set_tuning(0, 0, 0)
        sol.close()
        orc.close()


@pytest.mark.parametrize("S,variant", [(1, 0), (2, 0), (3, 0), (4, 0), (2, 34), (3, 32), (3, 33), (4, 32), (2, 35), (2, 37), (3, 36), (4, 35), (4, 36),
                                       (2, 70), (2, 73), (3, 71), (3, 72), (3, 73), (3, 74), (3, 75),
                                       (2, 80), (2, 82), (2, 84), (3, 80), (3, 81), (3, 83), (2, 90), (2, 91), (3, 90), (3, 1), (3, 2), (3, 3)])
@pytest.mark.parametrize("N,M,B,omega,n_chunk", [
    (1, 5, 0.5, 3.0, 0), (2, 61, 0.05, 100.0, 0), (5, 61, 0.05, 100.0, 2), (7, 300, 0.4, 10.0, 3),
    (32, 4096, 1.0, 1.0, 0), (33, 1000, 0.5, 30.0, 5), (16, 239, 0.3, 10.0, 0), (16, 240, 0.3, 10.0, 8),
    (16, 241, 0.3, 10.0, 1), (64, 20001, 0.2, 10.0, 16), (16, 115, 0.3, 10.0, 0), (16, 116, 0.3, 10.0, 0),
    (16, 117, 0.3, 10.0, 0), (70, 200001, 0.02, 30.0, 0),
])
def test_multi_step_bit_exact(slb, oracle, S, variant, N, M, B, omega, n_chunk):
    """S loop iterations per launch (general kernel, four columns per lane or -- variants
    32..34 -- two): live state equals the oracle's."""
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    try:
        sol.tiptoe()
        L.slb_set_tuning(variant, n_chunk, 0)
        for _ in range(3):
            sol.multi_step(S)
        orc.step(3 * S)
        _compare_live_state(sol, orc, M)
        assert sol.s.t == orc.st.t
        for single in (sol.step, sol.step_fused):
            L.slb_set_tuning(0, n_chunk, 0)          # the variant numbers belong to the multi-step kernel
            single()
            L.slb_set_tuning(variant, n_chunk, 0)
            sol.multi_step(S)
        orc.step(2 + 2 * S)
        _compare_live_state(sol, orc, M)
    finally:
        L.slb_set_tuning(0, 0, 0)
        sol.close()
        orc.close()


@pytest.mark.parametrize("N,M,B,omega", [(1, 5, 0.5, 3.0), (8, 61, 0.05, 100.0), (32, 4096, 1.0, 1.0),
                                         (64, 40001, 0.1, 10.0)])
def test_double_step_hands_out_the_rows_av_reads(slb, oracle, N, M, B, omega):
    """slb_step_fused2_rows: the side buffers hold a rows 0, 1 and b row 1 of the main lattice
    after the first iteration (stored nowhere else) and after the second, bit for bit; the
    exact sums formed from them are the oracle's av() for those two lattices."""
    import ctypes as C
    import torch
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    try:
        sol.tiptoe()
        for _ in range(2):
            sol.double_step()
        orc.step(4)
        pitch = sol.s.pitch
        rows = [torch.zeros((3, pitch), dtype=torch.float32, device="cuda") for _ in range(2)]
        av_dev = torch.zeros(6, dtype=torch.float32, device="cuda")
        tot = [torch.zeros(4, dtype=torch.float32, device="cuda") for _ in range(2)]
        av_cpu = np.zeros(6, np.float32)
        sol.double_step_rows(rows[0].data_ptr(), rows[1].data_ptr())
        slab = slb.Slab(pitch, 0, 1, M)
        a_ptrs = (C.c_void_p * 2)(rows[0].data_ptr(), rows[1].data_ptr())
        b_ptrs = (C.c_void_p * 2)(rows[0].data_ptr() + 4 * pitch, rows[1].data_ptr() + 4 * pitch)
        t_ptrs = (C.c_void_p * 2)(tot[0].data_ptr(), tot[1].data_ptr())
        assert L.slb_av_sums_exact_n(C.byref(slab), 2, a_ptrs, b_ptrs, t_ptrs) == 0
        for k in range(2):
            t_k = np.float32(orc.st.t)
            orc.step(1)
            cur = orc.st.current
            a, b = orc.array("a", cur), orc.array("b", cur)
            got = rows[k].cpu().numpy()
            for r, exp in ((0, a[0]), (1, a[1]), (2, b[1])):
                assert np.array_equal(got[r, 1:M + 2].view(np.uint32), exp[1:M + 2].view(np.uint32)), (k, r)
            oracle.lib().orc_av(C.byref(p), a, b, av_cpu, t_k)
            assert L.slb_av_update_exact(tot[k].data_ptr(), av_dev.data_ptr(), t_k) == 0
        torch.cuda.synchronize()
        assert np.array_equal(av_dev.cpu().numpy().view(np.uint32), av_cpu.view(np.uint32))
        _compare_live_state(sol, orc, M)
    finally:
        sol.close()
        orc.close()


@pytest.mark.parametrize("S", [1, 2, 3])
@pytest.mark.parametrize("N,M,B,omega,n_chunk", [(1, 5, 0.5, 3.0, 0), (2, 61, 0.05, 100.0, 0), (8, 61, 0.05, 100.0, 3),
                                                 (32, 4096, 1.0, 1.0, 0), (64, 40001, 0.1, 10.0, 0),
                                                 (64, 40001, 0.1, 10.0, 1), (33, 70001, 0.1, 10.0, 7)])
def test_multi_step_hands_out_the_rows_av_reads(slb, oracle, S, N, M, B, omega, n_chunk):
    """slb_step_multi_rows (S = 3: the LDS-ring kernel): the side buffers hold a rows 0, 1 and b
    row 1 of the main lattice after EACH of the S iterations (the ones in between are stored
    nowhere else), bit for bit; the exact sums formed from them are the oracle's av() of those
    lattices; the lattices afterwards are the oracle's."""
    import ctypes as C
    import torch
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    try:
        sol.tiptoe()
        sol.multi_step(S)
        sol.step_fused()
        orc.step(S + 1)
        L.slb_set_tuning(0, n_chunk, 0)
        pitch = sol.s.pitch
        rows = [torch.full((3, pitch), 7.0, dtype=torch.float32, device="cuda") for _ in range(S)]
        av_dev = torch.zeros(6, dtype=torch.float32, device="cuda")
        tot = [torch.zeros(4, dtype=torch.float32, device="cuda") for _ in range(S)]
        av_cpu = np.zeros(6, np.float32)
        sol.multi_step_rows(S, [r.data_ptr() for r in rows])
        slab = slb.Slab(pitch, 0, 1, M)
        a_ptrs = (C.c_void_p * S)(*[r.data_ptr() for r in rows])
        b_ptrs = (C.c_void_p * S)(*[r.data_ptr() + 4 * pitch for r in rows])
        t_ptrs = (C.c_void_p * S)(*[t.data_ptr() for t in tot])
        assert L.slb_av_sums_exact_n(C.byref(slab), S, a_ptrs, b_ptrs, t_ptrs) == 0
        for k in range(S):
            t_k = np.float32(orc.st.t)
            orc.step(1)
            cur = orc.st.current
            a, b = orc.array("a", cur), orc.array("b", cur)
            got = rows[k].cpu().numpy()
            for r, exp in ((0, a[0]), (1, a[1]), (2, b[1])):
                assert np.array_equal(got[r, 1:M + 2].view(np.uint32), exp[1:M + 2].view(np.uint32)), (k, r)
            oracle.lib().orc_av(C.byref(p), a, b, av_cpu, t_k)
            assert L.slb_av_update_exact(tot[k].data_ptr(), av_dev.data_ptr(), t_k) == 0
        torch.cuda.synchronize()
        assert np.array_equal(av_dev.cpu().numpy().view(np.uint32), av_cpu.view(np.uint32))
        _compare_live_state(sol, orc, M)
        assert sol.s.t == orc.st.t
    finally:
        L.slb_set_tuning(0, 0, 0)
        sol.close()
        orc.close()


@pytest.mark.parametrize("N,M,expected", [(64, 262144, 2), (128, 262144, 3), (96, 262144, 3), (32, 524288, 3), (16, 2097152, 2),
                                          (512, 65536, 3), (128, 131072, 2), (32, 4096, 2), (128, 1 << 20, 3)])
def test_steps_per_launch_follows_the_measured_rule(slb, N, M, expected, monkeypatch):
    """Three iterations per launch from 32 harmonics up where the lattice has 2^19 columns or 24 million
    points (where the three-step kernel was measured faster), two elsewhere; the environment overrides."""
    monkeypatch.delenv("SLB_STEPS_PER_LAUNCH", raising=False)
    slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    sol = slb.Solver()
    try:
        assert sol.steps_per_launch() == expected
        monkeypatch.setenv("SLB_STEPS_PER_LAUNCH", "2")
        assert sol.steps_per_launch() == 2
        monkeypatch.setenv("SLB_STEPS_PER_LAUNCH", "3")
        assert sol.steps_per_launch() == 3
    finally:
        sol.close()


def _hip_runtime():
    import ctypes
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    hip = ctypes.CDLL(paths[0])
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return hip


def _hostile_field(rng, rows, cols):
    """Values no smooth run holds side by side: magnitudes from the smallest denormal to 1e30 next to
    each other, both zeros, isolated denormals.  (No infinities or NaN: which NaN an invalid operation
    makes is the processor's choice, not the algorithm's.)"""
    x = rng.standard_normal((rows, cols)) * 10.0 ** rng.uniform(-46, 30, (rows, cols))
    x = x.astype(np.float32)
    r = rng.random((rows, cols))
    x[r < 0.02] = 0.0
    x[(r >= 0.02) & (r < 0.04)] = -0.0
    x[(r >= 0.04) & (r < 0.05)] = np.float32(1e-41)
    x[(r >= 0.05) & (r < 0.06)] = np.float32(-1.4e-45)
    return x


@pytest.mark.parametrize("S", [1, 2, 3])
@pytest.mark.parametrize("N,M,B,omega,n_chunk", [(5, 61, 0.05, 100.0, 0), (16, 1000, 0.3, 10.0, 0), (33, 9001, 0.5, 30.0, 5),
                                                 (24, 130001, 0.02, 30.0, 0), (8, 520001, 0.005, 30.0, 0)])
def test_multi_step_bit_exact_on_hostile_fields(slb, oracle, S, N, M, B, omega, n_chunk):
    """The same lattices of wild values (denormals beside 1e30, both zeros) in the engine and in the
    oracle, a0 included, then launches of S iterations against single oracle steps: the point update's
    divisions meet numerators of every size, the halo lanes and the row patches meet data that is not
    smooth.  The last shape is wide enough for the launch schedule of the full-size lattices (the two
    edge strip groups first, then one round of whole column strips, the rest in row chunks)."""
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    hip = _hip_runtime()
    rng = np.random.default_rng(1000 * S + N)
    try:
        sol.tiptoe()
        s = sol.s
        pitch = s.pitch
        for which, i, ptr in [("a0", 0, s.a0)] + [("a", i, s.a[i]) for i in range(4)] + [("b", i, s.b[i]) for i in range(4)]:
            x = _hostile_field(rng, N + 1, M + 3)
            orc.array(which, i)[:] = x
            padded = np.zeros((N + 1, pitch), np.float32)
            padded[:, :M + 3] = x
            assert hip.hipMemcpy(ptr, padded.ctypes.data, padded.nbytes, 1) == 0          # host to device
        L.slb_set_tuning(0, n_chunk, 0)
        for _ in range(2):
            sol.multi_step(S)
        orc.step(2 * S)
        _compare_live_state(sol, orc, M)
        sol.step(); sol.multi_step(S)
        orc.step(1 + S)
        _compare_live_state(sol, orc, M)
        sol.double_step(); sol.step_fused()          # the hand-written two-step kernel and the one-step launch
        orc.step(3)
        _compare_live_state(sol, orc, M)
        for which, i, arr in (("a", s.current, 1 + s.current), ("b", s.current_hs, 5 + s.current_hs)):
            assert np.isfinite(sol.rows(arr)[:, :M + 3]).all(), "the comparison is meant to be between finite values"
    finally:
        L.slb_set_tuning(0, 0, 0)
        sol.close()
        orc.close()


@pytest.mark.parametrize("seed", [1, 20251111])
def test_shared_reciprocal_division_is_the_ieee_division(slb, seed):
    """div2_shared (one refined reciprocal for the two quotients of a point, hardware scaling of
    the numerators only) against the compiler's IEEE division on the GPU: 2^32 pseudo-random
    triples per seed -- numerators over every bit pattern (denormals, zeros, infinities, NaN),
    denominators from 0.5 to 2^100 and of the form nu^2 + x^2 -- not one bit may differ."""
    import torch
    L = slb.lib()
    differ = torch.zeros(1, dtype=torch.int64, device="cuda")
    first = torch.zeros(4, dtype=torch.float32, device="cuda")
    assert L.slb_selftest_division(1 << 32, seed, differ.data_ptr(), first.data_ptr()) == 0
    torch.cuda.synchronize()
    assert int(differ.item()) == 0, f"{int(differ.item())} quotients differ; first: n, d, got, expected = {first.cpu().numpy()}"


@pytest.mark.parametrize("N,M,mu", [(32, 4096, 5.0), (128, 300001, 5.0), (64, 70001, 0.7), (24, 5003, 31.0)])
def test_a0_from_its_factors_is_checked_cell_by_cell(slb, oracle, N, M, mu):
    """The double step forms a0 from a row factor and a column factor only after every cell of
    the lattice was compared with the stored a0 (reference: src/boltzmann_c_solver.c:117-122 is
    a float coefficient times a long-double Gaussian).  Here: the engine's check passed (rows
    whose factor had to move to a neighbouring double are counted), the stored a0 is the
    oracle's, the check kernel does find a factor that is off by parts in 1e9, and the double
    step with a0 loaded (default), formed once per row (50) and formed per stage (51) all leave
    the oracle's lattices."""
    import torch
    p = oracle.make_params(N=N, M=M, B=0.3, omega=20.0, mu=mu)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    L = slb.lib()
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    try:
        s = sol.s
        assert s.a0_row and s.a0_col and 0 <= s.a0_rows_adjusted <= N + 1
        assert np.array_equal(sol.rows(0)[:, :M + 3].view(np.uint32), orc.array("a0").view(np.uint32))
        bad = torch.zeros(N + 1, dtype=torch.int32, device="cuda")
        assert L.slb_a0_factors_check(s.a0, s.pitch, 0, N + 1, M + 3, s.a0_row, s.a0_col, bad.data_ptr()) == 0
        torch.cuda.synchronize()
        assert int(bad.sum()) == 0
        # a row factor that is slightly off is caught
        fac = _device_doubles(torch, s.a0_row, N + 1)
        wrong = fac.clone()
        wrong[1] *= 1.0 + 1e-9
        bad.zero_()
        assert L.slb_a0_factors_check(s.a0, s.pitch, 0, N + 1, M + 3, wrong.data_ptr(), s.a0_col, bad.data_ptr()) == 0
        torch.cuda.synchronize()
        counts = bad.cpu().numpy()
        assert counts[1] > 0 and counts[0] == 0 and counts[2:].sum() == 0
        sol.tiptoe()
        for variant in (50, 51, 0):
            L.slb_set_tuning(variant, 0, 0)
            sol.double_step()
            sol.double_step()
            orc.step(4)
            _compare_live_state(sol, orc, M)
    finally:
        L.slb_set_tuning(0, 0, 0)
        sol.close()
        orc.close()


def _device_doubles(torch, ptr, n):
    """n doubles at device address `ptr` as a new CUDA tensor (copied with the HIP runtime)."""
    import ctypes
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    hip = ctypes.CDLL(paths[0])
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(out.data_ptr(), ptr, 8 * n, 3) == 0          # hipMemcpyDeviceToDevice
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M", [30000, 31000])        # 125 and 130 waves of 240 columns, four chunks of rows each
@pytest.mark.parametrize("variant", [0, 38, 39])
def test_both_sides_of_the_narrow_launch_threshold(slb, oracle, M, variant):
    """Launches of at most 512 wave-chunks go to the one-column-per-lane kernel, wider ones to
    the four-column kernel; which one a lattice gets depends on its width (and the chunking on
    the device's occupancy).  Just under and just over the switch, with the row buffers of the
    averaging period: the default choice, the four-column kernel forced (38) and the
    occupancy-unknown schedule (39) all leave the oracle's bits, lattices and rows alike."""
    import torch
    N = 16
    p = oracle.make_params(N=N, M=M, B=0.3, omega=20.0)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    try:
        sol.tiptoe()
        L.slb_set_tuning(variant, 0, 0)
        rows = [torch.zeros((3, sol.s.pitch), dtype=torch.float32, device="cuda") for _ in range(2)]
        for pair in range(3):
            if pair == 1:
                sol.double_step()
                orc.step(2)
                continue
            sol.double_step_rows(rows[0].data_ptr(), rows[1].data_ptr())
            torch.cuda.synchronize()
            for k in range(2):
                orc.step(1)
                cur = orc.st.current
                a, b = orc.array("a", cur), orc.array("b", cur)
                got = rows[k].cpu().numpy()
                for r, exp in ((0, a[0]), (1, a[1]), (2, b[1])):
                    assert np.array_equal(got[r, 1:M + 1].view(np.uint32), exp[1:M + 1].view(np.uint32)), (pair, k, r)
        _compare_live_state(sol, orc, M)
    finally:
        L.slb_set_tuning(0, 0, 0)
        sol.close()
        orc.close()


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("form", ["sweeps", "fused", "double"])
def test_engine_matches_committed_field_fixtures(slb, form):
    """tests/golden/fields_N8_M61.npz (regression pins written from the CPU oracle): the GPU
    engine reproduces the arrays after the tiptoe and after 1, 2, 10 iterations from the
    files alone, in every step form (live state only after a double step)."""
    fx = np.load(os.path.join(GOLDEN, "fields_N8_M61.npz"))
    N, M = 8, 61
    slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=0.05,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    sol = slb.Solver()
    try:
        sol.tiptoe()
        done = 0
        for steps in (0, 1, 2, 10):
            while done < steps:
                if form == "double" and steps - done >= 2:
                    sol.double_step(); done += 2
                elif form == "sweeps":
                    sol.step(); done += 1
                else:
                    sol.step_fused(); done += 1
            s = sol.s
            assert [s.current, s.next, s.current_hs, s.next_hs] == list(fx[f"idx_{steps}"])
            assert np.float32(s.t) == fx[f"t_{steps}"]
            live = [("a0", 0, 0), ("a", s.current, 1 + s.current), ("b", s.current, 5 + s.current),
                    ("a", s.current_hs, 1 + s.current_hs), ("b", s.current_hs, 5 + s.current_hs)]
            every = [("a0", 0, 0)] + [("a", i, 1 + i) for i in range(4)] + [("b", i, 5 + i) for i in range(4)]
            for w, i, arr in (live if form == "double" and steps >= 2 else every):
                key = f"eq_{steps}" if w == "a0" else f"{w}{i}_{steps}"
                assert np.array_equal(sol.rows(arr)[:, :M + 3].view(np.uint32), fx[key].view(np.uint32)), (key, form)
    finally:
        sol.close()


def test_program_matches_committed_display77_trace(solver_bin, tmp_path):
    args = ["display=77", "omega=100", "B=0.12", "n-harmonics=64", "g-grid=32768", "t-max=0.05"] + COMMON
    g = subprocess.run([solver_bin] + args + ["o=out.txt"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert g.returncode == 0, g.stdout + g.stderr
    assert (tmp_path / "out.txt").read_text() == open(os.path.join(GOLDEN, "display77_shrunken_config2.txt")).read()


def test_program_config2_matches_what_was_observed_on_the_reference(solver_bin, tmp_path):
    """The display=77 run of the full config-2 shape against the observations recorded from
    the reference (tests/golden/config2_display77_reference_observations.json)."""
    import json
    from test_oracle_cpu import _check_config2_observations
    obs = json.load(open(os.path.join(GOLDEN, "config2_display77_reference_observations.json")))
    g = subprocess.run([solver_bin] + obs["args"].split() + ["o=out.txt", "quiet=1"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert g.returncode == 0, g.stdout + g.stderr
    _check_config2_observations((tmp_path / "out.txt").read_text())


def test_unusual_parameters_all_forms(slb, oracle):
    """Nothing in the kernels is tied to the usual +-4 range, mu = 5 or dt = 0.001: an
    asymmetric phi_y range, other field strengths and a different time step give the
    oracle's bits in every step form, and so do av() and the frame values."""
    import ctypes as C
    import torch
    kw = dict(N=12, M=777, E_dc=2.5, E_omega=0.7, omega=13.0, mu=2.3, alpha=0.8, B=1.7,
              PhiYmin=-3.0, PhiYmax=5.0, dt=0.0005)
    p = oracle.make_params(t_start=0.2, **kw)
    slb.configure(**kw)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    L = slb.lib()
    try:
        sol.tiptoe()
        for _ in range(3):
            sol.step()
        orc.step(3)
        _compare_state(sol, orc, kw["M"])
        sol.step_fused(); sol.double_step(); sol.multi_step(3); sol.step_fused()
        orc.step(7)
        _compare_live_state(sol, orc, kw["M"])
        cur = orc.st.current
        av_cpu = np.zeros(6, np.float32)
        av_dev = torch.zeros(6, dtype=torch.float32, device="cuda")
        L.slb_set_av_mode(slb.SLB_AV_EXACT)
        for t in (0.3, 0.3005):
            oracle.lib().orc_av(C.byref(p), orc.array("a", cur), orc.array("b", cur), av_cpu, np.float32(t))
            L.av(0, sol.s.a[sol.s.current], sol.s.b[sol.s.current], av_dev.data_ptr(), np.float32(t))
        torch.cuda.synchronize()
        assert np.array_equal(av_dev.cpu().numpy().view(np.uint32), av_cpu.view(np.uint32))
        phi = oracle.frame_phi_x()[::60]
        exp = oracle.frame_values(p, orc.array("a", cur), orc.array("b", cur), phi)
        assert np.array_equal(sol.frame_values(phi).view(np.uint32), exp.view(np.uint32))
    finally:
        sol.close()
        orc.close()
