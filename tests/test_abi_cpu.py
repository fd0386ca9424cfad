"""CPU-only checks of the boundary: the library loads, exports every symbol the
header declares, and the program's command line behaves like the reference's."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "slb_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    funcs = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)) - {"defined"}
    ext = re.findall(r"extern\s+(?:float|int)\s+([^;]+);", text)
    globs = {g.strip() for decl in ext for g in decl.split(",")}
    return funcs, globs


def test_library_exports_every_declared_symbol(slb):
    funcs, globs = _declared_symbols()
    assert funcs == set(slb.ABI_FUNCTIONS)
    assert globs == set(slb.ABI_GLOBALS)
    L = slb.lib()
    for name in funcs:
        assert getattr(L, name) is not None
    for name in globs:
        C.c_int.in_dll(L, name)
    # the multi-GPU entry points (include/slb_slabs.h) come from the same library
    text = open(os.path.join(ROOT, "include", "slb_slabs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    slab_funcs = set(re.findall(r"\b(slb_(?:slabs|link)_[a-z0-9_]+)\s*\(", text))
    assert len(slab_funcs) >= 30
    for name in slab_funcs:
        assert getattr(L, name) is not None and getattr(L, name).argtypes is not None, name


def test_every_bound_function_declares_its_argument_types(slb):
    """A call through ctypes without argtypes passes Python ints as C ints: a 64-bit device
    pointer would be cut to 32 bits without a word.  Every function of the header has them."""
    L = slb.lib()
    for name in slb.ABI_FUNCTIONS:
        fn = getattr(L, name)
        assert fn.argtypes is not None, name
    assert L.slb_version().startswith(b"slb2d")
    # the host engine's functions (host/slb_host.h: engine object, links, slab loop) as well
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, slb.__name__, "host", "slb_host.h")).read() + \
             open(os.path.join(root, "include", "slb_slabs.h")).read()
    declared = set(re.findall(r"\b(slb_(?:slabs|link|solver)_\w+)\s*\(", header))
    assert len(declared) > 50
    for name in sorted(declared):
        assert getattr(L, name).argtypes is not None, name


def test_half_step_sweep_leaves_the_last_main_column_alone(oracle):
    """The parity target's half-step sweep stops at column M (reference:
    src/boltzmann_c_solver.c:391, `m < TMSIZE`), one column before the reference's CUDA
    kernels do (src/boltzmann_gpu.cu:176): column M+1 of the output arrays keeps what it held."""
    import numpy as np
    N, M = 6, 40
    p = oracle.make_params(N=N, M=M)
    rng = np.random.default_rng(5)
    arr = lambda: rng.standard_normal((N + 1, M + 3)).astype(np.float32)
    a0, ac, bc, a_s, b_s = arr(), arr(), arr(), arr(), arr()
    a_out, b_out = np.full((N + 1, M + 3), 7.0, np.float32), np.full((N + 1, M + 3), 9.0, np.float32)
    oracle.lib().orc_step_on_half_grid(C.byref(p), a0, a_s, b_s, ac, bc, a_out, b_out, 0.5, 0.25)
    assert np.all(a_out[:, M + 1:] == 7.0) and np.all(b_out[:, M + 1:] == 9.0) and np.all(a_out[:, 0] == 7.0)
    assert np.all(a_out[:N, 1:M + 1] != 7.0) and np.all(a_out[N] == 7.0) and np.all(b_out[0] == 9.0)
    oracle.lib().orc_step_on_grid(C.byref(p), a0, ac, bc, a_out, b_out, a_s, b_s, 0.5, 0.25)
    assert np.all(a_out[:N, M + 1] != 7.0) and np.all(a_out[:, M + 2] == 7.0)


def test_derived_constants_follow_the_reference_formulas(slb):
    d = slb.configure(N=32, M=4096, E_dc=1.0, E_omega=0.1, omega=1.0, mu=5.0, alpha=1.0, B=1.0,
                      PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    import numpy as np
    f = np.float32
    assert d["host_dPhi"] == f(8.0) / f(4096)
    assert d["host_nu"] == f(1) + f(0.001) / f(2)
    assert d["host_nu2"] == f(d["host_nu"]) * f(d["host_nu"])
    assert d["host_nu_tilde"] == f(1) - f(0.001) / f(2)
    assert d["host_bdt"] == f(f(1.0) * f(0.001)) / f(f(4) * f(d["host_dPhi"]))
    assert (d["MSIZE"], d["NSIZE"], d["host_TMSIZE"]) == (4099, 33, 4097)
    assert d["PADDED_MSIZE"] == 4128 and d["PADDED_MSIZE"] * 4 % 128 == 0
    # a row count that is already a multiple of 128 bytes is not padded further
    d = slb.configure(N=4, M=3069, E_dc=1.0, E_omega=0.1, omega=1.0, mu=5.0, alpha=1.0, B=1.0,
                      PhiYmin=-4.0, PhiYmax=4.0)
    assert d["PADDED_MSIZE"] == 3072


ARGS = ("display=4 E_dc=1 E_omega=0.1 omega=1 mu=5 alpha=1 n-harmonics=8 PhiYmin=-4 PhiYmax=4 "
        "B=1 t-max=1").split()


def _run(solver_bin, args, **kw):
    return subprocess.run([solver_bin] + args, capture_output=True, text=True, timeout=60, **kw)


def test_missing_required_parameter(solver_bin, tmp_path):
    for drop, name in ((1, "E_dc"), (6, "n-harmonics"), (10, "t-max")):
        args = [a for i, a in enumerate(ARGS) if i != drop] + ["o=stdout"]
        r = _run(solver_bin, args, cwd=tmp_path)
        assert r.returncode == 1
        assert f'ERROR: Parameter "{name}" must be set.' in r.stderr
    r = _run(solver_bin, ARGS[1:] + ["o=stdout"], cwd=tmp_path)
    assert r.returncode == 1 and 'Parameter "display" must be set' in r.stderr


def test_invalid_values(solver_bin, tmp_path):
    r = _run(solver_bin, ["display=5"] + ARGS[1:] + ["o=stdout"], cwd=tmp_path)
    assert r.returncode == 1 and "Invalid value of display= parameter" in r.stderr
    r = _run(solver_bin, ARGS[:-1] + ["t-max=0", "o=stdout"], cwd=tmp_path)
    assert r.returncode == 1 and "Invalid value of t-max= parameter" in r.stderr
    r = _run(solver_bin, ARGS + ["read-from=file.txt", "o=stdout"], cwd=tmp_path)
    assert r.returncode == 1 and "Invalid value of read-from=" in r.stderr
    r = _run(solver_bin, ARGS + ["o=/nonexistent-dir/x.txt"], cwd=tmp_path)
    assert r.returncode == 1 and 'Faled to open file "/nonexistent-dir/x.txt" for writing.' in r.stderr
    # parsing stops at the first token without '=': everything after it is ignored
    r = _run(solver_bin, ARGS[:3] + ["stop"] + ARGS[3:] + ["o=stdout"], cwd=tmp_path)
    assert r.returncode == 1 and 'Parameter "omega" must be set' in r.stderr


def test_argument_errors_are_reported_without_a_launch(slb, capfd):
    """Bad geometry, misaligned arrays, aliasing outputs and empty ranges are handled
    on the host: -1 (or 0 for "nothing to do") comes back before any kernel launch."""
    L = slb.lib()
    slb.configure(N=8, M=61, E_dc=1.0, E_omega=0.1, omega=1.0, mu=5.0, alpha=1.0, B=1.0,
                  PhiYmin=-4.0, PhiYmax=4.0)
    ok = [0x1000 * (i + 1) for i in range(13)]             # fake, 16-byte aligned device addresses
    f2 = (C.c_float * 2)(0.5, 0.5)

    # two-sweep form
    assert L.slb_sweep(C.byref(slb.Slab(64, 0, 0, 10)), *ok[:7], 0.5, 0.5) == -1      # j_first < 1
    assert L.slb_sweep(C.byref(slb.Slab(64, 0, 1, 63)), *ok[:7], 0.5, 0.5) == -1      # j_last > pitch-2
    assert L.slb_sweep(C.byref(slb.Slab(64, 0, 5, 4)), *ok[:7], 0.5, 0.5) == 0        # empty range
    L.slb_set_tuning(2, 0, 0)
    assert L.slb_sweep(C.byref(slb.Slab(67, 0, 1, 60)), *ok[:7], 0.5, 0.5) == -1      # 16-byte kernel, odd pitch
    bad = list(ok[:7]); bad[3] += 4
    assert L.slb_sweep(C.byref(slb.Slab(64, 0, 1, 60)), *bad, 0.5, 0.5) == -1         # misaligned array
    L.slb_set_tuning(0, 0, 96)
    assert L.slb_sweep(C.byref(slb.Slab(64, 0, 1, 60)), *ok[:7], 0.5, 0.5) == -1      # block not a multiple of 64
    L.slb_set_tuning(0, 0, 0)

    # one launch per step
    c4 = (C.c_float * 4)(1, 1, 1, 1)
    g = slb.StepGeom(66, 0, 1, 62, 1, 61)
    assert L.slb_step_fused(C.byref(g), *ok[:9], C.byref(c4)) == -1                   # pitch not a multiple of 4
    g = slb.StepGeom(64, 0, 5, 4, 5, 4)
    assert L.slb_step_fused(C.byref(g), *ok[:9], C.byref(c4)) == 0                    # empty

    # halo movers and reductions
    assert L.slb_halo_pack(ok[0], ok[1], 64, 9, 63, 2, ok[2]) == -1                   # columns run off the row
    assert L.slb_halo_unpack(ok[0], ok[1], 64, 0, 1, 1, ok[2]) == -1                  # no rows
    err = capfd.readouterr().err
    assert "bad geometry" in err and "16-byte aligned" in err and "block must be" in err


def test_display77_needs_the_rows_its_sums_read(solver_bin, tmp_path):
    r = _run(solver_bin, ["display=77"] + [a for a in ARGS[1:] if not a.startswith("n-harmonics")] +
             ["n-harmonics=1", "o=stdout"], cwd=tmp_path)
    assert r.returncode == 1 and "display=77 needs n-harmonics >= 2" in r.stderr
