"""display=8 frame values formed on the device vs the CPU oracle (bit for bit)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(slb, oracle, N, M, B, omega, steps):
    p = oracle.make_params(N=N, M=M, B=B, omega=omega)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    orc = oracle.OracleState(p)
    sol = slb.Solver()
    sol.tiptoe()
    for _ in range(steps):
        sol.step()
    orc.step(steps)
    return p, orc, sol


@pytest.mark.parametrize("N,M,B,omega,steps,nphi", [
    (8, 61, 0.5, 100.0, 10, None),        # the whole 629-value phi_x sequence
    (1, 5, 0.5, 3.0, 3, 9),               # one harmonic, a ragged last batch of phi_x
    (32, 4096, 1.0, 1.0, 20, 37),
    (128, 20001, 0.2, 10.0, 6, 16),
])
def test_frame_values_bit_exact(slb, oracle, N, M, B, omega, steps, nphi):
    p, orc, sol = _setup(slb, oracle, N, M, B, omega, steps)
    try:
        phi = oracle.frame_phi_x()
        assert len(phi) == 629
        if nphi is not None:
            phi = phi[:: max(1, len(phi) // nphi)][:nphi]
        cur = orc.st.current
        exp = oracle.frame_values(p, orc.array("a", cur), orc.array("b", cur), phi)
        got = sol.frame_values(phi)
        assert got.shape == exp.shape == (len(phi), M + 1)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        assert (got >= 0).all() and got.max() > 0
    finally:
        sol.close()
        orc.close()


def test_equilibrium_frame_values_bit_exact(slb, oracle):
    """source=1: the cosine-only sum over a0 that display=3 prints beside the distribution.
    The oracle's frame loop with b = 0 adds exact zeros to every term, i.e. the same values."""
    p, orc, sol = _setup(slb, oracle, 16, 300, 0.4, 10.0, 2)
    try:
        phi = oracle.frame_phi_x()[::25]
        a0 = orc.array("a0")
        exp = oracle.frame_values(p, a0, np.zeros_like(a0), phi)
        got = sol.frame_values(phi, source=1)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    finally:
        sol.close()
        orc.close()


def test_frame_values_entry_point_checks_its_arguments(slb):
    L = slb.lib()
    ok = 0x10000
    assert L.slb_frame_values(ok, ok, 64, 9, 60, 8, ok, 4, ok) == -1          # columns run off the row
    assert L.slb_frame_values(ok, ok, 64, 9, 1, 8, ok + 8, 4, ok) == -1       # table not 16-byte aligned
    assert L.slb_frame_values(ok, ok, 64, 9, 1, 0, ok, 4, ok) == 0            # nothing to do
