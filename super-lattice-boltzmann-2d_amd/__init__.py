"""Thin ctypes binding of libslb2d.so (HIP kernels + C host for gfx950).

The product is the C library and the ``boltzmann_solver`` program; this module
only loads the library for tests, the benchmark and the one-process-per-GPU
driver.  It never computes anything itself and fails loudly when the library
has not been built (``make lib`` at the repository root).

Entry points bound here are the ones declared in ``include/slb_abi.h`` (the
reference boundary: reference src/boltzmann_gpu.h:4-29) and in
``host/slb_host.h`` (the host engine: reference src/boltzmann_solver.c:97-253).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLB_LIB: another build of the same library (A/B timing of kernel variants); default: in-tree
LIB_PATH = os.environ.get("SLB_LIB") or os.path.join(_HERE, "libslb2d.so")

_FLOAT_GLOBALS = ("host_E_dc", "host_E_omega", "host_omega", "host_mu", "host_alpha",
                  "PhiYmin", "PhiYmax", "host_B", "t_start", "host_dPhi", "host_dt",
                  "host_bdt", "host_nu_tilde", "host_nu2", "host_nu", "frame_start", "t_max")
_INT_GLOBALS = ("host_M", "host_N", "MSIZE", "MP1", "NSIZE", "host_TMSIZE", "PADDED_MSIZE",
                "display", "quiet", "device")

# every symbol include/slb_abi.h declares
ABI_FUNCTIONS = ("load_data", "step_on_grid", "step_on_half_grid", "av", "HandleError",
                 "slb_sweep", "slb_step_fused", "slb_step_fused2", "slb_step_fused2_rows", "slb_step_multi", "slb_step_multi_rows",
                 "slb_av_sums_exact", "slb_av_sums_exact_n", "slb_av_sums_exact_from_n", "slb_av_update_exact", "slb_halo_pack", "slb_halo_unpack", "slb_halo_pack_segs", "slb_halo_unpack_segs", "slb_av_partial",
                 "slb_frame_values", "slb_seq_sum", "slb_seq_sum_stats",
                 "slb_av_update", "slb_set_av_mode", "slb_set_stream", "slb_get_stream", "slb_push_stream", "slb_pop_stream",
                 "slb_set_tuning", "slb_version", "slb_static_init", "slb_av_static",
                 "slb_a0_factors_check", "slb_step_fused2_factored", "slb_selftest_division", "slb_debug_schedule")
ABI_GLOBALS = ("host_E_dc", "host_E_omega", "host_omega", "host_mu", "host_alpha",
               "PhiYmin", "PhiYmax", "host_B", "t_start", "host_dPhi", "host_dt",
               "host_bdt", "host_nu_tilde", "host_nu2", "host_nu",
               "host_M", "host_N", "MSIZE", "MP1", "NSIZE", "host_TMSIZE", "PADDED_MSIZE")

SLB_AV_EXACT, SLB_AV_TREE, SLB_AV_CHAIN = 0, 1, 2


class Slab(C.Structure):
    _fields_ = [("pitch", C.c_int), ("m_offset", C.c_int), ("j_first", C.c_int), ("j_last", C.c_int)]


class StepGeom(C.Structure):
    _fields_ = [("pitch", C.c_int), ("m_offset", C.c_int), ("j1_first", C.c_int), ("j1_last", C.c_int),
                ("j2_first", C.c_int), ("j2_last", C.c_int)]


class Step2Geom(C.Structure):
    _fields_ = [("pitch", C.c_int), ("m_offset", C.c_int), ("j_first", C.c_int * 4), ("j_last", C.c_int * 4)]


class MultiGeom(C.Structure):
    _fields_ = [("pitch", C.c_int), ("m_offset", C.c_int), ("steps", C.c_int),
                ("j_first", C.c_int * 8), ("j_last", C.c_int * 8)]


class HaloSeg(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("col", C.c_int), ("ncols", C.c_int),
                ("buf", C.c_void_p)]


class SolverStruct(C.Structure):
    """Mirror of ``slb_solver`` in host/slb_host.h."""
    _fields_ = [("N", C.c_int), ("M", C.c_int), ("rank", C.c_int), ("nranks", C.c_int),
                ("W", C.c_int), ("pitch", C.c_int), ("m_offset", C.c_int),
                ("main_slab", Slab), ("half_slab", Slab), ("av_slab", Slab),
                ("a0", C.c_void_p), ("a", C.c_void_p * 4), ("b", C.c_void_p * 4),
                ("av_data", C.c_void_p), ("sums", C.c_void_p), ("halo_buf", C.c_void_p),
                ("current", C.c_int), ("next", C.c_int), ("current_hs", C.c_int), ("next_hs", C.c_int),
                ("t", C.c_float), ("frame_time", C.c_float), ("steps", C.c_long), ("fused", C.c_int),
                ("G", C.c_int),
                ("spare_a", C.c_void_p), ("spare_b", C.c_void_p), ("spare_ah", C.c_void_p),
                ("spare_bh", C.c_void_p), ("spare_of", C.c_int), ("spare_of_hs", C.c_int),
                ("stream", C.c_void_p), ("has_stream", C.c_int),
                ("a0_row", C.c_void_p), ("a0_col", C.c_void_p), ("a0_rows_adjusted", C.c_int)]


_lib = None


def lib():
    """The loaded library; raises if it is missing (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make lib` (hipcc --offload-arch=gfx950). "
                "There is no CPU or eager fallback for the stepping loop.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        fp, vp, i, f = C.c_void_p, C.c_void_p, C.c_int, C.c_float
        L.load_data.argtypes, L.load_data.restype = [], None
        L.step_on_grid.argtypes = [i] + [fp] * 7 + [f] * 4
        L.step_on_grid.restype = None
        L.step_on_half_grid.argtypes = [i] + [fp] * 9 + [f] * 4
        L.step_on_half_grid.restype = None
        L.av.argtypes, L.av.restype = [i, fp, fp, fp, f], None
        L.slb_sweep.argtypes = [C.POINTER(Slab)] + [fp] * 7 + [f, f]
        L.slb_sweep.restype = i
        L.slb_step_fused.argtypes = [C.POINTER(StepGeom)] + [fp] * 9 + [C.POINTER(C.c_float * 4)]
        L.slb_step_fused.restype = i
        L.HandleError.argtypes, L.HandleError.restype = [i, C.c_char_p, i], None
        L.slb_step_fused2.argtypes = [C.POINTER(Step2Geom)] + [fp] * 13 + [C.POINTER(C.c_float * 8)]
        L.slb_step_fused2.restype = i
        L.slb_step_fused2_rows.argtypes = [C.POINTER(Step2Geom)] + [fp] * 13 + [C.POINTER(C.c_float * 8), fp, fp]
        L.slb_step_fused2_rows.restype = i
        L.slb_step_fused2_factored.argtypes = [C.POINTER(Step2Geom)] + [fp] * 15 + [C.POINTER(C.c_float * 8), fp, fp]
        L.slb_step_fused2_factored.restype = i
        L.slb_a0_factors_check.argtypes, L.slb_a0_factors_check.restype = [fp, i, i, i, i, fp, fp, fp], i
        L.slb_selftest_division.argtypes = [C.c_ulonglong, C.c_uint, fp, fp]
        L.slb_selftest_division.restype = i
        L.slb_debug_schedule.argtypes = [i] * 8 + [C.POINTER(C.c_int)] * 2 + [i]
        L.slb_debug_schedule.restype = i
        L.slb_step_multi.argtypes = [C.POINTER(MultiGeom), fp, C.POINTER(C.c_void_p * 4), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p * 4), C.POINTER(C.c_float)]
        L.slb_step_multi.restype = i
        L.slb_step_multi_rows.argtypes = [C.POINTER(MultiGeom), fp, C.POINTER(C.c_void_p * 4), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p * 4), C.POINTER(C.c_float),
                                          C.POINTER(C.c_void_p)]
        L.slb_step_multi_rows.restype = i
        L.slb_static_init.argtypes, L.slb_static_init.restype = [], i
        L.slb_av_static.argtypes, L.slb_av_static.restype = [fp, fp, fp, f], i
        L.slb_halo_pack.argtypes = [fp, fp, i, i, i, i, fp]
        L.slb_halo_pack.restype = i
        L.slb_halo_unpack.argtypes = [fp, fp, i, i, i, i, fp]
        L.slb_halo_unpack.restype = i
        L.slb_av_partial.argtypes = [C.POINTER(Slab), fp, fp, fp]
        L.slb_av_partial.restype = i
        L.slb_av_update.argtypes = [fp, fp, f]
        L.slb_av_update.restype = i
        L.slb_set_av_mode.argtypes, L.slb_set_av_mode.restype = [i], None
        L.slb_seq_sum.argtypes, L.slb_seq_sum.restype = [fp, i, fp], i
        L.slb_seq_sum_stats.argtypes, L.slb_seq_sum_stats.restype = [fp, i, fp, fp], i
        L.slb_set_stream.argtypes, L.slb_set_stream.restype = [vp], None
        L.slb_get_stream.argtypes, L.slb_get_stream.restype = [], vp
        L.slb_push_stream.argtypes, L.slb_push_stream.restype = [vp], None
        L.slb_pop_stream.argtypes, L.slb_pop_stream.restype = [], None
        L.slb_set_tuning.argtypes, L.slb_set_tuning.restype = [i, i, i], None
        L.slb_version.argtypes, L.slb_version.restype = [], C.c_char_p
        # host engine
        L.slb_derive_and_load.argtypes, L.slb_derive_and_load.restype = [], None
        L.slb_bessel_In.argtypes, L.slb_bessel_In.restype = [i, C.c_double], C.c_double
        L.slb_solver_create.argtypes, L.slb_solver_create.restype = [i, i], C.POINTER(SolverStruct)
        L.slb_solver_destroy.argtypes, L.slb_solver_destroy.restype = [C.POINTER(SolverStruct)], None
        L.slb_solver_tiptoe.argtypes, L.slb_solver_tiptoe.restype = [C.POINTER(SolverStruct)], None
        L.slb_solver_sweep_main.argtypes = [C.POINTER(SolverStruct), f, f]
        L.slb_solver_sweep_main.restype = None
        L.slb_solver_sweep_half.argtypes = [C.POINTER(SolverStruct), f, f]
        L.slb_solver_sweep_half.restype = None
        L.slb_solver_swap.argtypes, L.slb_solver_swap.restype = [C.POINTER(SolverStruct)], None
        L.slb_solver_sweep_range.argtypes = [C.POINTER(SolverStruct), i, i, i, f, f]
        L.slb_solver_sweep_range.restype = None
        L.slb_solver_advance.argtypes, L.slb_solver_advance.restype = [C.POINTER(SolverStruct)], None
        L.slb_step_cosines.argtypes, L.slb_step_cosines.restype = [f, C.POINTER(C.c_float * 4)], None
        L.slb_solver_step.argtypes, L.slb_solver_step.restype = [C.POINTER(SolverStruct)], None
        L.slb_solver_step_fused.argtypes, L.slb_solver_step_fused.restype = [C.POINTER(SolverStruct)], None
        L.slb_solver_double_step.argtypes, L.slb_solver_double_step.restype = [C.POINTER(SolverStruct)], None
        L.slb_solver_prepare_double_step.argtypes = [C.POINTER(SolverStruct)]
        L.slb_solver_prepare_double_step.restype = None
        L.slb_solver_double_step_rows.argtypes = [C.POINTER(SolverStruct), fp, fp]
        L.slb_solver_double_step_rows.restype = None
        L.slb_av_sums_exact.argtypes, L.slb_av_sums_exact.restype = [C.POINTER(Slab), fp, fp, fp], i
        L.slb_av_update_exact.argtypes, L.slb_av_update_exact.restype = [fp, fp, f], i
        pp = C.POINTER(C.c_void_p)
        L.slb_av_sums_exact_from_n.argtypes = [C.POINTER(Slab), i, pp, pp, pp, pp]
        L.slb_av_sums_exact_from_n.restype = i
        L.slb_av_sums_exact_n.argtypes, L.slb_av_sums_exact_n.restype = [C.POINTER(Slab), i, pp, pp, pp], i
        L.slb_solver_copy_av_rows.argtypes, L.slb_solver_copy_av_rows.restype = [C.POINTER(SolverStruct), fp], i
        L.slb_solver_double_step_range.argtypes = [C.POINTER(SolverStruct), i, i, C.POINTER(C.c_float * 8)]
        L.slb_solver_double_step_range.restype = None
        L.slb_solver_double_step_range_rows.argtypes = [C.POINTER(SolverStruct), i, i, C.POINTER(C.c_float * 8), fp, fp]
        L.slb_solver_double_step_range_rows.restype = None
        L.slb_solver_double_step_finish.argtypes = [C.POINTER(SolverStruct)]
        L.slb_solver_double_step_finish.restype = None
        L.slb_solver_multi_step.argtypes, L.slb_solver_multi_step.restype = [C.POINTER(SolverStruct), i], None
        L.slb_solver_multi_step_range.argtypes = [C.POINTER(SolverStruct), i, i, i, C.POINTER(C.c_float)]
        L.slb_solver_multi_step_range.restype = None
        L.slb_solver_multi_step_finish.argtypes = [C.POINTER(SolverStruct), i]
        L.slb_solver_multi_step_finish.restype = None
        L.slb_steps_cosines.argtypes, L.slb_steps_cosines.restype = [f, i, C.POINTER(C.c_float)], None
        L.slb_solver_multi_step_rows.argtypes = [C.POINTER(SolverStruct), i, C.POINTER(C.c_void_p)]
        L.slb_solver_multi_step_rows.restype = None
        L.slb_solver_multi_step_range_rows.argtypes = [C.POINTER(SolverStruct), i, i, i, C.POINTER(C.c_float), C.POINTER(C.c_void_p)]
        L.slb_solver_multi_step_range_rows.restype = None
        L.slb_solver_steps_per_launch.argtypes, L.slb_solver_steps_per_launch.restype = [C.POINTER(SolverStruct)], i
        L.slb_step2_cosines.argtypes, L.slb_step2_cosines.restype = [f, C.POINTER(C.c_float * 8)], None
        L.slb_solver_step_fused_range.argtypes = [C.POINTER(SolverStruct), i, i, C.POINTER(C.c_float * 4)]
        L.slb_solver_step_fused_range.restype = None
        L.slb_solver_run_timed.argtypes = [C.POINTER(SolverStruct), i, C.POINTER(C.c_float)]
        L.slb_solver_run_timed.restype = i
        L.slb_solver_halo_pack.argtypes = [C.POINTER(SolverStruct), i, i, i, fp]
        L.slb_solver_halo_pack.restype = i
        L.slb_solver_halo_unpack.argtypes = [C.POINTER(SolverStruct), i, i, i, fp]
        L.slb_solver_halo_unpack.restype = i
        L.slb_solver_av_partial.argtypes, L.slb_solver_av_partial.restype = [C.POINTER(SolverStruct), fp], i
        L.slb_halo_pack_segs.argtypes = [C.POINTER(HaloSeg), i, i, i]
        L.slb_halo_pack_segs.restype = i
        L.slb_halo_unpack_segs.argtypes = [C.POINTER(HaloSeg), i, i, i]
        L.slb_halo_unpack_segs.restype = i
        L.slb_solver_halo_pack_double.argtypes = [C.POINTER(SolverStruct), fp, fp]
        L.slb_solver_halo_pack_double.restype = i
        L.slb_solver_halo_unpack_double.argtypes = [C.POINTER(SolverStruct), fp, fp]
        L.slb_solver_halo_unpack_double.restype = i
        L.slb_frame_values.argtypes, L.slb_frame_values.restype = [fp, fp, i, i, i, i, fp, i, fp], i
        L.slb_solver_frame_values.argtypes = [C.POINTER(SolverStruct), i, fp, i, fp]
        L.slb_solver_frame_values.restype = i
        L.slb_solver_get_rows.argtypes = [C.POINTER(SolverStruct), i, i, i, fp]
        L.slb_solver_get_rows.restype = i
        L.slb_solver_get_window.argtypes = [C.POINTER(SolverStruct), i, i, i, fp]
        L.slb_solver_get_window.restype = i
        # several GPUs: links between processes and the slab time loop (host/slb_link.c, slb_slabs.c)
        sz = C.c_size_t
        L.slb_link_unique_id.argtypes, L.slb_link_unique_id.restype = [C.c_char_p], i
        L.slb_link_open_rccl.argtypes, L.slb_link_open_rccl.restype = [i, i, C.c_char_p], vp
        L.slb_link_open_fds.argtypes, L.slb_link_open_fds.restype = [i, i, C.POINTER(C.c_int)], vp
        L.slb_link_ranks_seen.argtypes, L.slb_link_ranks_seen.restype = [vp], i
        L.slb_link_transport.argtypes, L.slb_link_transport.restype = [vp], C.c_char_p
        L.slb_link_begin.argtypes, L.slb_link_begin.restype = [vp], None
        L.slb_link_send.argtypes, L.slb_link_send.restype = [vp, i, fp, sz, vp], None
        L.slb_link_recv.argtypes, L.slb_link_recv.restype = [vp, i, fp, sz, vp], None
        L.slb_link_end.argtypes, L.slb_link_end.restype = [vp, vp], None
        L.slb_link_close.argtypes, L.slb_link_close.restype = [vp], None
        L.slb_slabs_create.argtypes, L.slb_slabs_create.restype = [i, i, vp], vp
        L.slb_slabs_create_loopback.argtypes, L.slb_slabs_create_loopback.restype = [i, i, vp], vp
        L.slb_slabs_destroy.argtypes, L.slb_slabs_destroy.restype = [vp], None
        L.slb_slabs_solver.argtypes, L.slb_slabs_solver.restype = [vp], C.POINTER(SolverStruct)
        L.slb_slabs_main_stream.argtypes, L.slb_slabs_main_stream.restype = [vp], vp
        L.slb_slabs_ranks_seen.argtypes, L.slb_slabs_ranks_seen.restype = [vp], i
        L.slb_slabs_tiptoe.argtypes, L.slb_slabs_tiptoe.restype = [vp], None
        L.slb_slabs_double_step.argtypes, L.slb_slabs_double_step.restype = [vp, fp, fp], None
        L.slb_slabs_multi_step.argtypes, L.slb_slabs_multi_step.restype = [vp, i, C.POINTER(C.c_void_p)], None
        L.slb_solver_halo_pack_multi.argtypes = [C.POINTER(SolverStruct), i, fp, fp]
        L.slb_solver_halo_pack_multi.restype = i
        L.slb_solver_halo_unpack_multi.argtypes = [C.POINTER(SolverStruct), i, fp, fp]
        L.slb_solver_halo_unpack_multi.restype = i
        L.slb_slabs_last_step.argtypes, L.slb_slabs_last_step.restype = [vp], None
        L.slb_slabs_step.argtypes, L.slb_slabs_step.restype = [vp], None
        L.slb_link_rank.argtypes, L.slb_link_rank.restype = [vp], i
        L.slb_link_nranks.argtypes, L.slb_link_nranks.restype = [vp], i
        L.slb_solver_set_stream.argtypes, L.slb_solver_set_stream.restype = [C.POINTER(SolverStruct), vp], None
        L.slb_solver_frame_values_device.argtypes = [C.POINTER(SolverStruct), i, C.POINTER(C.c_float), i, fp]
        L.slb_solver_frame_values_device.restype = i
        L.slb_slabs_av_set.argtypes, L.slb_slabs_av_set.restype = [vp, C.POINTER(C.c_float * 6)], None
        L.slb_slabs_av_data.argtypes, L.slb_slabs_av_data.restype = [vp], vp
        L.slb_slabs_write_frame.argtypes, L.slb_slabs_write_frame.restype = [vp, C.c_char_p, f], i
        L.slb_slabs_join.argtypes, L.slb_slabs_join.restype = [vp], None
        L.slb_slabs_av_rows.argtypes, L.slb_slabs_av_rows.restype = [vp, i, C.POINTER(C.c_void_p)], None
        L.slb_slabs_av_submit.argtypes, L.slb_slabs_av_submit.restype = [vp, i, C.POINTER(C.c_float), i], None
        L.slb_slabs_av_submit_current.argtypes, L.slb_slabs_av_submit_current.restype = [vp, f], None
        L.slb_slabs_av_finish.argtypes, L.slb_slabs_av_finish.restype = [vp, C.POINTER(C.c_float * 6)], None
        L.slb_slabs_gather_rows.argtypes, L.slb_slabs_gather_rows.restype = [vp, i, i, i, fp], None
        _lib = L
    return _lib


def set_globals(**kw):
    """Set host-owned globals by their C names (host_E_dc=..., host_M=..., ...)."""
    L = lib()
    for k, v in kw.items():
        if k in _FLOAT_GLOBALS:
            C.c_float.in_dll(L, k).value = v
        elif k in _INT_GLOBALS:
            C.c_int.in_dll(L, k).value = v
        else:
            raise KeyError(k)


def get_global(name):
    L = lib()
    if name in _FLOAT_GLOBALS:
        return C.c_float.in_dll(L, name).value
    if name in _INT_GLOBALS:
        return C.c_int.in_dll(L, name).value
    raise KeyError(name)


def configure(*, N, M, E_dc, E_omega, omega, mu, alpha, B, PhiYmin, PhiYmax, dt=0.001, t_start=1.0):
    """What the program's command line plus slb_derive_and_load() do: set the
    run parameters, derive dPhi / nu / bdt / pitch, snapshot them for launches."""
    set_globals(host_N=N, host_M=M, host_E_dc=E_dc, host_E_omega=E_omega, host_omega=omega,
                host_mu=mu, host_alpha=alpha, host_B=B, PhiYmin=PhiYmin, PhiYmax=PhiYmax,
                host_dt=dt, t_start=t_start)
    lib().slb_derive_and_load()
    return {k: get_global(k) for k in ("host_dPhi", "host_nu", "host_nu2", "host_nu_tilde",
                                       "host_bdt", "MSIZE", "PADDED_MSIZE", "NSIZE", "host_TMSIZE")}


class Solver:
    """One phi_y slab of the lattice on the current HIP device (host/slb_solver.c)."""

    def __init__(self, rank=0, nranks=1):
        self._p = lib().slb_solver_create(rank, nranks)
        if not self._p:
            raise RuntimeError("slb_solver_create failed")

    @property
    def s(self):
        return self._p.contents

    def close(self):
        if self._p:
            lib().slb_solver_destroy(self._p)
            self._p = None

    def tiptoe(self):
        lib().slb_solver_tiptoe(self._p)

    def step(self):
        lib().slb_solver_step(self._p)

    def step_fused(self):
        lib().slb_solver_step_fused(self._p)

    def double_step(self):
        lib().slb_solver_double_step(self._p)

    def multi_step(self, S):
        lib().slb_solver_multi_step(self._p, S)

    def multi_step_rows(self, S, rows):
        """rows: S device pointers to [3][pitch] buffers that receive the rows av() reads of the
        main lattice after each of the S iterations."""
        r = (C.c_void_p * S)(*rows)
        lib().slb_solver_multi_step_rows(self._p, S, r)

    def steps_per_launch(self):
        return lib().slb_solver_steps_per_launch(self._p)

    def multi_step_range(self, S, j_first, j_last, cosines, rows=None):
        c = (C.c_float * (4 * S))(*cosines)
        if rows is None:
            lib().slb_solver_multi_step_range(self._p, S, j_first, j_last, c)
        else:
            r = (C.c_void_p * S)(*rows)
            lib().slb_solver_multi_step_range_rows(self._p, S, j_first, j_last, c, r)

    def multi_step_finish(self, S):
        lib().slb_solver_multi_step_finish(self._p, S)

    def double_step_range(self, j_first, j_last, cosines8, rows=None):
        """rows: optional pair of device pointers to [3][pitch] buffers that receive the rows
        av() reads of the main lattices after the first and the second iteration."""
        c = (C.c_float * 8)(*cosines8)
        if rows is None:
            lib().slb_solver_double_step_range(self._p, j_first, j_last, C.byref(c))
        else:
            lib().slb_solver_double_step_range_rows(self._p, j_first, j_last, C.byref(c), rows[0], rows[1])

    def double_step_finish(self):
        lib().slb_solver_double_step_finish(self._p)

    def set_fused(self, on):
        """0/False: two launches per step; 1/True: one launch per step; 2: one launch per two steps;
        3: one launch per three steps (the rest of a run in twos and ones); 10 + S: slb_step_multi with S."""
        self._p.contents.fused = int(on)

    def sweep_main(self, c0, c1):
        lib().slb_solver_sweep_main(self._p, c0, c1)

    def sweep_half(self, c0, c1):
        lib().slb_solver_sweep_half(self._p, c0, c1)

    def swap(self):
        lib().slb_solver_swap(self._p)

    def sweep_range(self, lattice, j_first, j_last, c0, c1):
        lib().slb_solver_sweep_range(self._p, lattice, j_first, j_last, c0, c1)

    def advance(self):
        lib().slb_solver_advance(self._p)

    def run_timed(self, steps):
        ms = C.c_float(0)
        rc = lib().slb_solver_run_timed(self._p, steps, C.byref(ms))
        if rc:
            raise RuntimeError(f"slb_solver_run_timed -> {rc}")
        return ms.value

    def step_fused_range(self, j2_first, j2_last, cosines):
        c = (C.c_float * 4)(*cosines)
        lib().slb_solver_step_fused_range(self._p, j2_first, j2_last, C.byref(c))

    def halo_pack(self, which, side, ncols, buf_ptr):
        rc = lib().slb_solver_halo_pack(self._p, which, side, ncols, buf_ptr)
        if rc:
            raise RuntimeError(f"slb_solver_halo_pack -> {rc}")

    def halo_unpack(self, which, side, ncols, buf_ptr):
        rc = lib().slb_solver_halo_unpack(self._p, which, side, ncols, buf_ptr)
        if rc:
            raise RuntimeError(f"slb_solver_halo_unpack -> {rc}")

    def halo_pack_double(self, left_ptr, right_ptr):
        """Both new lattices' edge columns after a double step's edge ranges, one launch;
        a pointer of None means no neighbour on that side."""
        rc = lib().slb_solver_halo_pack_double(self._p, left_ptr, right_ptr)
        if rc:
            raise RuntimeError(f"slb_solver_halo_pack_double -> {rc}")

    def halo_unpack_double(self, left_ptr, right_ptr):
        rc = lib().slb_solver_halo_unpack_double(self._p, left_ptr, right_ptr)
        if rc:
            raise RuntimeError(f"slb_solver_halo_unpack_double -> {rc}")

    def prepare_double_step(self):
        lib().slb_solver_prepare_double_step(self._p)

    def double_step_rows(self, rows_first_ptr, rows_second_ptr):
        """double_step() that also writes a rows 0, 1 and b row 1 of both new main lattices
        into two [3][pitch] device buffers."""
        lib().slb_solver_double_step_rows(self._p, rows_first_ptr, rows_second_ptr)

    def copy_av_rows(self, rows_ptr):
        """a rows 0, 1 and b row 1 of the current main lattice -> a [3][pitch] device buffer."""
        rc = lib().slb_solver_copy_av_rows(self._p, rows_ptr)
        if rc:
            raise RuntimeError(f"slb_solver_copy_av_rows -> {rc}")

    def av_partial(self, sums_ptr):
        rc = lib().slb_solver_av_partial(self._p, sums_ptr)
        if rc:
            raise RuntimeError(f"slb_solver_av_partial -> {rc}")

    def frame_values(self, phi_x, source=0):
        """display=8 values of the current main lattice (source=0) or of the equilibrium
        coefficients a0 (source=1) for the given phi_x values: a (len(phi_x), columns)
        float32 array, columns = the slab's main-lattice columns."""
        import numpy as np
        phi_x = np.ascontiguousarray(phi_x, dtype=np.float32)
        ms = self.s.main_slab
        out = np.empty((len(phi_x), ms.j_last - ms.j_first + 1), dtype=np.float32)
        rc = lib().slb_solver_frame_values(self._p, source, phi_x.ctypes.data, len(phi_x), out.ctypes.data)
        if rc:
            raise RuntimeError(f"slb_solver_frame_values -> {rc}")
        return out

    def rows(self, arr, row0=0, nrows=None):
        """Rows of device array `arr` (0=a0, 1..4=a[0..3], 5..8=b[0..3]) as a numpy
        array of shape (nrows, pitch)."""
        import numpy as np
        s = self.s
        nrows = s.N + 1 - row0 if nrows is None else nrows
        out = np.empty((nrows, s.pitch), dtype=np.float32)
        rc = lib().slb_solver_get_rows(self._p, arr, row0, nrows, out.ctypes.data)
        if rc:
            raise RuntimeError(f"slb_solver_get_rows -> {rc}")
        return out


    def window(self, arr, col0, ncols):
        """Columns [col0, col0+ncols) of every row of device array `arr`."""
        import numpy as np
        out = np.empty((self.s.N + 1, ncols), dtype=np.float32)
        rc = lib().slb_solver_get_window(self._p, arr, col0, ncols, out.ctypes.data)
        if rc:
            raise RuntimeError(f"slb_solver_get_window -> {rc}")
        return out


def steps_cosines(t, S):
    c = (C.c_float * (4 * S))()
    lib().slb_steps_cosines(t, S, c)
    return tuple(c)


def step2_cosines(t):
    c = (C.c_float * 8)()
    lib().slb_step2_cosines(t, C.byref(c))
    return tuple(c)


def step_cosines(t):
    c = (C.c_float * 4)()
    lib().slb_step_cosines(t, C.byref(c))
    return tuple(c)
