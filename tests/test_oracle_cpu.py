"""CPU-only checks of the oracle: the recorded reference line, Bessel values,
step counts, and slab/whole-domain consistency of the oracle's own sweep."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import scipy.special

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_recorded_reference_line(oracle, tmp_path):
    """config 1 (n-harmonics=32 g-grid=4096 t-max=10, 16 281 steps): the nine
    computed columns of the display=4 line equal, digit for digit, the line the
    compiled reference printed when it was surveyed."""
    with open(os.path.join(GOLDEN, "config1_display4_reference.json")) as f:
        gold = json.load(f)
    out = tmp_path / "cfg1.txt"
    r = oracle.run_cli(gold["args"].split() + [f"o={out}"], threads=8, timeout=900)
    assert f"# t_max = {gold['t_max']}" in r.stdout
    lines = out.read_text().splitlines()
    assert len(lines) == 3 and lines[0].startswith("# display=4 E_dc=1.0000")
    v = lines[2].split()
    assert len(v) == 13
    got = dict(zip(["v_dr/v_p", "A(omega)", "NORM", "v_y/v_p", "m/m_x", "<v_dr/v_p>", "<v_y/v_p>",
                    "<m/m_x>", "Asin"], v[4:]))
    assert got == gold["columns"]
    assert re.fullmatch(r"# display=4 .* t-max=10\.0+ dt=0\.0010000000474974\d+ g-grid=4096", lines[0])


def test_bessel_against_scipy(oracle, slb):
    for x in (0.5, 5.0, 20.0):
        for n in (0, 1, 2, 7, 32, 64, 128):
            exp = scipy.special.iv(n, x)
            for fn in (oracle.lib().orc_bessel_In, slb.lib().slb_bessel_In):
                got = fn(n, x)
                # scipy's own error grows to ~1e-14 for orders far above the argument
                tol = 4e-15 if n <= 7 else 2e-13
                assert abs(got - exp) <= tol * exp, (n, x, got, exp)
    # far above the argument the value underflows double; 0 is the float image as well
    assert oracle.lib().orc_bessel_In(512, 5.0) == 0.0
    assert slb.lib().slb_bessel_In(512, 5.0) == 0.0
    assert oracle.lib().orc_bessel_In(1, 5.0) == slb.lib().slb_bessel_In(1, 5.0)


def _count_steps(oracle, dt, t_start, omega):
    p = oracle.make_params(N=1, M=4, dt=dt, t_start=t_start, omega=omega)
    st = oracle.OracleState(p, threads=1)
    t, n = np.float32(0), 0
    while t < np.float32(st.st.t_max):
        t = np.float32(t + np.float32(dt))
        n += 1
    st.close()
    return n


def test_step_counts_from_float_time_accumulation(oracle):
    assert _count_steps(oracle, 0.001, 10.0, 1.0) == 16281
    assert _count_steps(oracle, 0.001, 0.05, 100.0) == 113
    assert _count_steps(oracle, 0.001, 0.001, 100.0) == 64


def test_oracle_sweep_is_slab_invariant(oracle):
    """Updating a window of columns through a copied sub-array with a column
    offset gives the same bits as updating the whole array."""
    N, M = 9, 200
    p = oracle.make_params(N=N, M=M, B=1.3, omega=5.0)
    rng = np.random.default_rng(3)
    rows, pitch = N + 1, M + 3
    arrs = [(rng.standard_normal((rows, pitch)) * 0.1).astype(np.float32) for _ in range(7)]
    a0, ac, bc, a_s, b_s, ao, bo = arrs
    full_a, full_b = ao.copy(), bo.copy()
    oracle.sweep(p, pitch, 0, 1, M + 1, a0, ac, bc, a_s, b_s, full_a, full_b, 0.3, 0.2)
    lo, hi = 37, 121                       # window of owned columns, one ghost each side
    sub = [np.ascontiguousarray(x[:, lo - 1:hi + 2]) for x in arrs]
    oracle.sweep(p, hi - lo + 3, lo - 1, 1, hi - lo + 1, *sub[:5], sub[5], sub[6], 0.3, 0.2)
    assert np.array_equal(sub[5][:, 1:-1].view(np.uint32), full_a[:, lo:hi + 1].view(np.uint32))
    assert np.array_equal(sub[6][:, 1:-1].view(np.uint32), full_b[:, lo:hi + 1].view(np.uint32))
    # untouched cells: row N, column 0 / M+2, row 0 of b
    assert np.array_equal(full_a[N], ao[N]) and np.array_equal(full_b[0], bo[0])
    assert np.array_equal(full_a[:, 0], ao[:, 0]) and np.array_equal(full_a[:, M + 2], ao[:, M + 2])


def test_display77_trace_shape(oracle, tmp_path):
    """Shrunken config 2: 113 steps give 11 records; NORM stays at 1 within 2e-4
    and v_dr rises monotonically from the cold start."""
    out = tmp_path / "t77.txt"
    oracle.run_cli(("display=77 E_dc=1 E_omega=0.1 omega=100 mu=5 alpha=1 B=0.5 PhiYmin=-4 PhiYmax=4 "
                    "n-harmonics=32 g-grid=8192 dt=0.001 t-max=0.05").split() + [f"o={out}"], threads=8)
    rec = [l.split() for l in out.read_text().splitlines() if not l.startswith("#")]
    assert len(rec) == 11 and all(len(r) == 15 for r in rec)
    v_dr = [float(r[4]) for r in rec]
    assert all(b > a for a, b in zip(v_dr, v_dr[1:])) and 0.03 < v_dr[0] < 0.08
    assert all(abs(float(r[6]) - 1) < 2e-4 for r in rec)


def test_frame_values_are_what_the_frame_file_holds(oracle, tmp_path):
    """orc_frame_values (what the device kernel is compared with) and the display=8
    text writer share one loop: the file's third column is those values, printed."""
    args = ["display=8", "omega=100", "B=0.5", "n-harmonics=8", "g-grid=61", "t-max=0.01",
            "E_dc=1", "E_omega=0.1", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4", "dt=0.001"]
    oracle.run_cli(args + ["o=out.txt"], cwd=tmp_path)
    lines = [l for l in (tmp_path / "frame.data").read_text().splitlines() if not l.startswith("#")]
    p = oracle.make_params(N=8, M=61, B=0.5, omega=100.0, t_start=0.01)
    orc = oracle.OracleState(p)
    try:
        while orc.st.t < orc.st.t_max:
            orc.step(1)
        phi = oracle.frame_phi_x()
        cur = orc.st.current
        vals = oracle.frame_values(p, orc.array("a", cur), orc.array("b", cur), phi)
        assert len(lines) == vals.size == 629 * 62
        want = ["%0.5f" % x for x in phi]
        for k in (0, 1, 61, 62, 300 * 62 + 17, 628 * 62 + 61):
            px, py, v = lines[k].split()
            assert px == want[k // 62] and v == "%0.20f" % vals[k // 62, k % 62]
        assert all(l.split()[2] == "%0.20f" % v for l, v in zip(lines, vals.ravel()))
    finally:
        orc.close()


@pytest.mark.parametrize("E_dc", [0.5, 1.0, 2.0])
def test_esaki_tsu_limit_without_magnetic_field(oracle, tmp_path, E_dc):
    """A pin that does not come from the reference's output: with B = 0 and no a/c field the
    model is the one-dimensional superlattice, whose stationary drift velocity is the
    Esaki-Tsu characteristic v_dr/v_p = 2 E/(1 + E^2) whatever mu is, with norm 1 and no
    transverse velocity.  The oracle relaxes to it within the discretisation error."""
    args = ["display=4", f"E_dc={E_dc}", "E_omega=0", "omega=0", "mu=5", "alpha=1", "B=0",
            "PhiYmin=-4", "PhiYmax=4", "n-harmonics=24", "g-grid=256", "dt=0.001", "t-max=12", "o=out.txt"]
    oracle.run_cli(args, cwd=tmp_path, threads=4)
    cols = (tmp_path / "out.txt").read_text().splitlines()[-1].split()
    v_dr, norm, v_y = float(cols[4]), float(cols[6]), float(cols[7])
    assert abs(v_dr - 2 * E_dc / (1 + E_dc ** 2)) < 2e-4
    assert abs(norm - 1) < 1e-4 and abs(v_y) < 1e-6


FIELDS = dict(N=8, M=61, B=0.05, omega=100.0)


def test_oracle_matches_committed_field_fixtures(oracle):
    """tests/golden/fields_N8_M61.npz and av_N8_M61.npz (written by make_fixtures.py from the
    oracle itself: regression pins, not reference outputs) still come out bit for bit."""
    fx = np.load(os.path.join(GOLDEN, "fields_N8_M61.npz"))
    p = oracle.make_params(**FIELDS)
    st = oracle.OracleState(p, threads=3)
    try:
        done = 0
        for steps in (0, 1, 2, 10):
            st.step(steps - done)
            done = steps
            assert [st.st.current, st.st.next, st.st.current_hs, st.st.next_hs] == list(fx[f"idx_{steps}"])
            assert np.float32(st.st.t) == fx[f"t_{steps}"]
            assert np.array_equal(st.array("a0").view(np.uint32), fx[f"eq_{steps}"].view(np.uint32))
            for i in range(4):
                for w in "ab":
                    assert np.array_equal(st.array(w, i).view(np.uint32), fx[f"{w}{i}_{steps}"].view(np.uint32)), (w, i, steps)
        av_fx = np.load(os.path.join(GOLDEN, "av_N8_M61.npz"))
        cur = st.st.current
        av = np.zeros(6, np.float32)
        for k, t in enumerate(av_fx["times"], 1):
            oracle.lib().orc_av(C.byref(p), st.array("a", cur), st.array("b", cur), av, np.float32(t))
            assert np.array_equal(av.view(np.uint32), av_fx[f"after_{k}"].view(np.uint32)), k
    finally:
        st.close()


def test_oracle_matches_committed_display77_trace(oracle, tmp_path):
    args = ["display=77", "omega=100", "B=0.12", "n-harmonics=64", "g-grid=32768", "t-max=0.05", "E_dc=1",
            "E_omega=0.1", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4", "dt=0.001", "o=out.txt"]
    oracle.run_cli(args, cwd=tmp_path)
    assert (tmp_path / "out.txt").read_text() == open(os.path.join(GOLDEN, "display77_shrunken_config2.txt")).read()


def test_step_counts_fixture_holds_the_counts_observed_on_the_reference(oracle):
    """The loop runs while the float t < t_max: 16 281, 113 and 64 iterations are the counts the
    reference itself was seen to take for these parameters; the oracle's clock gives the same."""
    cases = json.load(open(os.path.join(GOLDEN, "step_counts.json")))
    seen = {(c["dt"], c["t-max"], c["omega"]): c["iterations"] for c in cases}
    assert seen[(0.001, 10.0, 1.0)] == 16281 and seen[(0.001, 0.05, 100.0)] == 113 and seen[(0.001, 0.001, 100.0)] == 64
    for c in cases[1:]:
        p = oracle.make_params(N=2, M=13, B=0.05, omega=c["omega"], dt=c["dt"], t_start=c["t-max"])
        st = oracle.OracleState(p, threads=1)
        n = 0
        while st.st.t < st.st.t_max:
            st.step(1)
            n += 1
        assert n == c["iterations"] and "%0.20f" % st.st.t_max == c["t_max"]
        st.close()


def _check_config2_observations(text):
    obs = json.load(open(os.path.join(GOLDEN, "config2_display77_reference_observations.json")))
    lines = text.splitlines()
    assert len(lines) == obs["lines"]
    rows = [l.split() for l in lines if not l.startswith("#")]
    assert len(rows) == obs["records"]
    v_dr = [float(r[4]) for r in rows]
    norm = [float(r[6]) for r in rows]
    t = [float(r[13]) for r in rows]
    assert abs(v_dr[0] - obs["v_dr_first"]) < 5e-5 and abs(v_dr[-1] - obs["v_dr_last"]) < 5e-5
    assert all(x < y for x, y in zip(v_dr, v_dr[1:]))                      # "rises"
    assert abs(t[0] - obs["t_first"]) < 1e-6 and abs(t[-1] - obs["t_last"]) < 1e-6
    assert round(min(norm), 6) == obs["norm_min"] and round(max(norm), 6) == obs["norm_max"]


def test_oracle_reproduces_what_was_observed_on_the_reference_at_config2(oracle, tmp_path):
    """Second anchor from the reference itself: the display=77 run of the config-2 shape was
    seen to give 11 records with v_dr rising 0.0526 -> 0.5004 over t = 0.01 .. 0.11 and the
    norm staying within 0.999876 .. 0.999879."""
    obs = json.load(open(os.path.join(GOLDEN, "config2_display77_reference_observations.json")))
    oracle.run_cli(obs["args"].split() + ["o=out.txt"], cwd=tmp_path, threads=8)
    _check_config2_observations((tmp_path / "out.txt").read_text())


def test_reparameterisation_modes_of_the_oracle(oracle, tmp_path):
    """read-from=stdin in the three readings of the reference (SLB_REPARAM): the CPU solver's
    (reference src/boltzmann_c_solver.c:270-280, new values are never read: the run equals one
    whose records repeat the old values), the CUDA host's (src/boltzmann_solver.c:384-393, the
    default: a new B acts without its derived constant bdt) and the repaired one (differs from
    the CUDA host's only after the record that changes B)."""
    args = ["display=4", "E_dc=1", "E_omega=0.1", "omega=100", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4",
            "B=0.5", "n-harmonics=6", "g-grid=121", "t-max=0.02", "dt=0.001", "read-from=stdin", "o=out.txt"]
    new = "E_dc 1.5 0.02\nB 0.4 0.013\nexit\n"
    same = "E_dc 1 0.02\nB 0.5 0.013\nexit\n"

    def run(name, text, mode):
        d = tmp_path / name
        d.mkdir()
        oracle.run_cli(args, cwd=d, stdin_text=text, threads=2, env={"SLB_REPARAM": mode} if mode else None)
        return (d / "out.txt").read_text().splitlines()

    cpu, unchanged = run("cpu", new, "cpu"), run("same", same, None)
    cuda, default, fixed = run("cuda", new, "cuda"), run("default", new, None), run("fixed", new, "fixed")
    assert len(cpu) == 9 and cpu == unchanged
    assert cuda == default and cuda != cpu
    assert fixed[:6] == cuda[:6] and fixed[6] == cuda[6] and fixed[8] != cuda[8]
