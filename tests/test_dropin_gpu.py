"""The boundary as the reference uses it: at LINK time.  tests/dropin/dropin_host.c is a host
in the reference's shape -- it defines host_E_dc ... PADDED_MSIZE itself (reference:
src/boltzmann_cli.c:20-45, src/boltzmann_solver.c:47-65), owns all device memory and calls
load_data / step_on_grid / step_on_half_grid / av / HandleError -- linked against libslb2d.so
with the documented link line (-lslb2d), with and without -rdynamic.  libslb2d.so carries
definitions of the same globals (for foreign-function callers); the test shows that
load_data() inside the library reads the EXECUTABLE's copies: the executable never touches the
library's, which stay zero, and a library that read its own would refuse to launch.  A second
load_data() after the executable changed host_B (reference: src/boltzmann_solver.c:391) takes
the new value.  Lattices and averaging data are compared bit for bit with the CPU oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_run(oracle, p, B2, steps1, steps2):
    orc = oracle.OracleState(p, threads=4)
    av = np.zeros(6, np.float32)
    try:
        for k in range(steps1 + steps2):
            if k == steps1:
                orc.st.p.B = B2               # the field term's B; bdt keeps its value, as after the CUDA host's load_data()
            t = np.float32(orc.st.t)
            orc.step(1)
            cur = orc.st.current
            oracle.lib().orc_av(C.byref(orc.st.p), orc.array("a", cur), orc.array("b", cur), av, t)
        cur, chs = orc.st.current, orc.st.current_hs
        fields = [orc.array("a", cur).copy(), orc.array("b", cur).copy(),
                  orc.array("a", chs).copy(), orc.array("b", chs).copy()]
        return fields, av, np.float32(orc.st.t)
    finally:
        orc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["dropin_host", "dropin_host_rdynamic"])
@pytest.mark.parametrize("N,M,B,B2,omega,steps1,steps2", [
    (16, 4093, 0.9, 0.4, 30.0, 7, 6),
    (5, 70001, 0.05, 0.08, 100.0, 3, 3),      # wide enough for the parallel form of av()'s sums
])
def test_host_that_defines_the_globals_itself(oracle, tmp_path, binary, N, M, B, B2, omega, steps1, steps2):
    exe = os.path.join(ROOT, "bin", binary)
    assert os.path.exists(exe), "build it with `make` (tests/dropin/dropin_host.c)"
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    a0 = np.zeros((N + 1, M + 3), np.float32)
    oracle.lib().orc_init_a0(C.byref(p), a0)
    a0.tofile(tmp_path / "a0.bin")
    r = subprocess.run([exe, str(N), str(M), repr(B), repr(B2), repr(omega), str(steps1), str(steps2), "a0.bin", "out.bin"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    cells = (N + 1) * (M + 3)
    assert raw.size == 4 * cells + 7
    fields, av, t = _oracle_run(oracle, p, np.float32(B2), steps1, steps2)
    for q, name in enumerate(("a[current]", "b[current]", "a[current_hs]", "b[current_hs]")):
        got = raw[q * cells:(q + 1) * cells].reshape(N + 1, M + 3)
        assert np.array_equal(got.view(np.uint32), fields[q].view(np.uint32)), name
    assert np.array_equal(raw[4 * cells:4 * cells + 6].view(np.uint32), av.view(np.uint32))
    assert av[0] == steps1 + steps2
    assert raw[-1] == t


def test_the_executables_export_their_parameters():
    """What makes the above work: the link editor puts the executable's definitions of the
    parameters into its dynamic symbol table because the library on its link line refers to
    them -- with or without -rdynamic -- so the library's references bind to them."""
    names = ["host_E_dc", "host_E_omega", "host_omega", "host_mu", "host_alpha", "PhiYmin", "PhiYmax", "host_B",
             "t_start", "host_dPhi", "host_dt", "host_bdt", "host_nu_tilde", "host_nu2", "host_nu",
             "host_M", "host_N", "MSIZE", "MP1", "NSIZE", "host_TMSIZE", "PADDED_MSIZE"]
    for binary in ("dropin_host", "dropin_host_rdynamic"):
        exe = os.path.join(ROOT, "bin", binary)
        if not os.path.exists(exe):
            subprocess.run(["make", "-C", ROOT, "bin/" + binary], check=True)
        out = subprocess.run(["nm", "-D", "--defined-only", exe], capture_output=True, text=True, check=True).stdout
        defined = {line.split()[-1] for line in out.splitlines() if line.split()}
        assert not [n for n in names if n not in defined], binary
