"""ctypes binding of the CPU oracle (oracle/slb_oracle.c).  Tests only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "libslb_oracle.so")
CLI = os.path.join(ORACLE_DIR, "_build", "slb_oracle_solver")


class Params(C.Structure):
    _fields_ = [("N", C.c_int), ("M", C.c_int),
                ("E_dc", C.c_float), ("E_omega", C.c_float), ("omega", C.c_float), ("mu", C.c_float),
                ("alpha", C.c_float), ("B", C.c_float), ("PhiYmin", C.c_float), ("PhiYmax", C.c_float),
                ("dt", C.c_float), ("t_start", C.c_float),
                ("dPhi", C.c_float), ("nu", C.c_float), ("nu2", C.c_float), ("nu_tilde", C.c_float),
                ("bdt", C.c_float), ("MSIZE", C.c_int), ("TMSIZE", C.c_int)]


class State(C.Structure):
    _fields_ = [("p", Params), ("a0", C.POINTER(C.c_float)),
                ("a", C.POINTER(C.c_float) * 4), ("b", C.POINTER(C.c_float) * 4),
                ("av_data", C.c_float * 6),
                ("current", C.c_int), ("next", C.c_int), ("current_hs", C.c_int), ("next_hs", C.c_int),
                ("t", C.c_float), ("t0", C.c_float), ("t_max", C.c_float), ("T", C.c_float),
                ("frame_time", C.c_float), ("steps_done", C.c_long)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not (os.path.exists(LIB) and os.path.exists(CLI)):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True)
        L = C.CDLL(LIB)
        fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
        PP = C.POINTER(Params)
        L.orc_derive.argtypes, L.orc_derive.restype = [PP], None
        L.orc_bessel_In.argtypes, L.orc_bessel_In.restype = [C.c_int, C.c_double], C.c_double
        L.orc_init_a0.argtypes, L.orc_init_a0.restype = [PP, fp], None
        L.orc_sweep.argtypes = [PP, C.c_int, C.c_int, C.c_int, C.c_int] + [fp] * 7 + [C.c_float] * 2
        L.orc_sweep.restype = None
        L.orc_step_on_grid.argtypes = [PP] + [fp] * 7 + [C.c_float] * 2
        L.orc_step_on_grid.restype = None
        L.orc_step_on_half_grid.argtypes = [PP] + [fp] * 7 + [C.c_float] * 2
        L.orc_step_on_half_grid.restype = None
        L.orc_av.argtypes, L.orc_av.restype = [PP, fp, fp, fp, C.c_float], None
        L.orc_eval_norm.argtypes, L.orc_eval_norm.restype = [PP, fp], C.c_float
        L.orc_state_init.argtypes, L.orc_state_init.restype = [C.POINTER(State), PP], C.c_int
        L.orc_state_free.argtypes, L.orc_state_free.restype = [C.POINTER(State)], None
        L.orc_state_step.argtypes, L.orc_state_step.restype = [C.POINTER(State)], None
        L.orc_set_threads.argtypes, L.orc_set_threads.restype = [C.c_int], None
        L.orc_seq_sum.argtypes, L.orc_seq_sum.restype = [fp, C.c_long], C.c_float
        L.orc_frame_values.argtypes, L.orc_frame_values.restype = [PP, fp, fp, fp, C.c_int, fp], None
        _lib = L
    return _lib


def make_params(*, N, M, E_dc=1.0, E_omega=0.1, omega=1.0, mu=5.0, alpha=1.0, B=1.0,
                PhiYmin=-4.0, PhiYmax=4.0, dt=0.001, t_start=1.0):
    p = Params()
    p.N, p.M = N, M
    p.E_dc, p.E_omega, p.omega, p.mu, p.alpha, p.B = E_dc, E_omega, omega, mu, alpha, B
    p.PhiYmin, p.PhiYmax, p.dt, p.t_start = PhiYmin, PhiYmax, dt, t_start
    lib().orc_derive(C.byref(p))
    return p


def sweep(p, pitch, m_offset, j_first, j_last, a0, ac, bc, a_s, b_s, a_out, b_out, c0, c1):
    lib().orc_sweep(C.byref(p), pitch, m_offset, j_first, j_last, a0, ac, bc, a_s, b_s, a_out, b_out, c0, c1)


class OracleState:
    """The oracle's solver state after the tiptoe; step() advances one loop iteration."""

    def __init__(self, p, threads=4):
        lib().orc_set_threads(threads)
        self.st = State()
        if lib().orc_state_init(C.byref(self.st), C.byref(p)) != 0:
            raise MemoryError("orc_state_init")
        self.rows = p.N + 1
        self.pitch = self.st.p.MSIZE

    def step(self, n=1):
        for _ in range(n):
            lib().orc_state_step(C.byref(self.st))

    def array(self, which, idx=0):
        ptr = {"a0": self.st.a0, "a": self.st.a[idx], "b": self.st.b[idx]}[which]
        return np.ctypeslib.as_array(ptr, shape=(self.rows, self.pitch))

    def close(self):
        lib().orc_state_free(C.byref(self.st))


def seq_sum(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return np.float32(lib().orc_seq_sum(x, len(x)))


def frame_phi_x():
    """The reference's phi_x sequence for display=8: a float advanced by 0.01 in double
    (reference: src/boltzmann_solver.c:495)."""
    out, x = [], np.float32(-3.14159265358979323846)
    while x < 3.14159265358979323846:
        out.append(x)
        x = np.float32(np.float64(x) + 0.01)
    return np.array(out, dtype=np.float32)


def frame_values(p, a, b, phi_x):
    phi_x = np.ascontiguousarray(phi_x, dtype=np.float32)
    out = np.empty((len(phi_x), p.M + 1), dtype=np.float32)
    lib().orc_frame_values(C.byref(p), a, b, phi_x, len(phi_x), out)
    return out


def run_cli(args, threads=8, timeout=600, cwd=None, stdin_text=None, env=None):
    lib()
    return subprocess.run([CLI] + list(args) + [f"threads={threads}"], check=True, timeout=timeout,
                          capture_output=True, text=True, cwd=cwd, input=stdin_text,
                          env=None if env is None else {**os.environ, **env})
