"""The parallel form of a sequential float sum (slb_seq_sum, and through it av()'s
exact mode) vs the plain loop of the CPU oracle: same bits for any input."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(2024)
    f32 = np.float32
    out = {}
    for n in (0, 1, 2, 63, 64, 65, 1023, 1024, 1025, 5000, 100003):
        out[f"normal_{n}"] = (rng.standard_normal(n) * 0.05).astype(f32)
    phi = np.linspace(-4, 4, 1 << 20).astype(f32)
    gauss = np.exp(-2.5 * phi.astype(np.float64) ** 2)
    out["gaussian_2^20"] = (gauss * (8.0 / (1 << 20))).astype(f32)                 # grows through ~60 binades
    out["odd_gaussian_2^20"] = (gauss * phi * (8.0 / (1 << 20))).astype(f32)       # cancels back to ~0
    out["all_ties"] = np.concatenate([[f32(1.0)], np.full(5000, 2.0 ** -24, f32)])  # every addition is a tie
    out["ties_odd"] = np.concatenate([[f32(1.0 + 2.0 ** -23)], np.full(3000, 2.0 ** -24, f32)])
    out["count_past_2^24"] = np.ones((1 << 24) + 5000, f32)                        # stalls at 2^24
    big = np.tile(np.array([1e10, 1.0, -1e10, 1.0], f32), 2000)
    out["absorb_and_cancel"] = big
    out["denormals"] = np.full(10000, 1e-41, f32)
    out["denormal_to_normal"] = np.full(100000, 3e-39, f32)
    out["inf"] = np.concatenate([rng.standard_normal(700).astype(f32), [f32(np.inf)],
                                 rng.standard_normal(700).astype(f32)])
    out["wide_exponents"] = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-30, 10, 200000)).astype(f32)
    out["integer_walk"] = rng.integers(-3, 4, 300000).astype(f32)                  # passes through exact zeros
    out["half_steps"] = (rng.integers(-5, 6, 100000) * 0.5).astype(f32)
    out["negative_growth"] = -np.abs(rng.standard_normal(50000)).astype(f32)
    x = rng.standard_normal(70000).astype(f32)
    x[rng.random(70000) < 0.3] = 0.0
    out["with_zeros"] = x
    return out


CASES = _cases()


@pytest.mark.parametrize("name", list(CASES))
def test_seq_sum_bits(slb, oracle, name):
    import torch
    x = CASES[name]
    exp = oracle.seq_sum(x)
    dev = torch.from_numpy(np.ascontiguousarray(x)).cuda() if len(x) else torch.zeros(1, device="cuda")
    out = torch.full((1,), 123.0, dtype=torch.float32, device="cuda")
    slb.lib().slb_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert slb.lib().slb_seq_sum(dev.data_ptr(), len(x), out.data_ptr()) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.view(np.uint32)[0] == np.array([exp], np.float32).view(np.uint32)[0], (name, got[0], exp)


@pytest.mark.parametrize("M", [61, 1021, 40000, 1 << 20])
def test_av_modes_agree_with_cpu(slb, oracle, M):
    """av(): the parallel exact mode and the one-lane chain both give the CPU's bits."""
    import torch
    L = slb.lib()
    N = 2
    p = oracle.make_params(N=N, M=M, B=0.5, omega=2.0)
    d = slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                      B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    pitch = d["PADDED_MSIZE"]
    rng = np.random.default_rng(M)
    phi = np.linspace(-4, 4, pitch)
    env = np.exp(-2.5 * phi ** 2)
    a = (env[None, :] * (1 + 0.1 * rng.standard_normal((N + 1, pitch)))).astype(np.float32)
    b = (env[None, :] * 0.3 * rng.standard_normal((N + 1, pitch))).astype(np.float32)
    a_cpu, b_cpu = np.ascontiguousarray(a[:, :M + 3]), np.ascontiguousarray(b[:, :M + 3])
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    L.slb_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    av_cpu = np.zeros(6, np.float32)
    for t in (0.2, 0.201):
        oracle.lib().orc_av(C.byref(p), a_cpu, b_cpu, av_cpu, np.float32(t))
    try:
        for mode in (slb.SLB_AV_EXACT, slb.SLB_AV_CHAIN):
            L.slb_set_av_mode(mode)
            av_dev = torch.zeros(6, dtype=torch.float32, device="cuda")
            for t in (0.2, 0.201):
                L.av(0, da.data_ptr(), db.data_ptr(), av_dev.data_ptr(), np.float32(t))
            torch.cuda.synchronize()
            assert np.array_equal(av_dev.cpu().numpy().view(np.uint32), av_cpu.view(np.uint32)), mode
    finally:
        L.slb_set_av_mode(slb.SLB_AV_EXACT)


@pytest.mark.parametrize("M,cuts", [(1021, (1, 400, 401)), (40000, (13, 20000)), (150000, (40000, 100000)),
                                    (1 << 20, (300000, 700001))])
def test_av_sums_carried_over_column_ranges(slb, oracle, M, cuts):
    """The exact sums of a lattice cut into column ranges, each range starting from the totals
    of the one before (one-lane form for narrow ranges, parallel form for wide ones, two
    lattices per launch): the totals after the last range are the CPU's sums over all columns."""
    import torch
    L = slb.lib()
    N = 2
    p = oracle.make_params(N=N, M=M, B=0.5, omega=2.0)
    d = slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                      B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    pitch = d["PADDED_MSIZE"]
    rng = np.random.default_rng(M + 1)
    phi = np.linspace(-4, 4, pitch)
    env = np.exp(-2.5 * phi ** 2)
    lattices = []
    for k in range(2):
        a = (env[None, :] * (1 + 0.1 * rng.standard_normal((N + 1, pitch)))).astype(np.float32)
        b = (env[None, :] * 0.3 * rng.standard_normal((N + 1, pitch))).astype(np.float32)
        lattices.append((a, b))
    dev = [(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) for a, b in lattices]
    L.slb_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    vp = C.c_void_p
    tot = torch.full((2, 3), 7.0, dtype=torch.float32, device="cuda")
    a_arr = (vp * 2)(dev[0][0].data_ptr(), dev[1][0].data_ptr())
    b_arr = (vp * 2)(dev[0][1].data_ptr(), dev[1][1].data_ptr())
    t_arr = (vp * 2)(tot.data_ptr(), tot.data_ptr() + 12)
    edges = [1] + [c + 1 for c in cuts] + [M + 1]           # columns 1..M
    for r, (lo, hi) in enumerate(zip(edges[:-1], edges[1:])):
        slab = slb.Slab(pitch, 0, lo, hi - 1)
        rc = L.slb_av_sums_exact_from_n(C.byref(slab), 2, a_arr, b_arr, None if r == 0 else t_arr, t_arr)
        assert rc == 0
    torch.cuda.synchronize()
    got = tot.cpu().numpy()
    f = np.float32
    for k, (a, b) in enumerate(lattices):
        ph = (f(p.PhiYmin) + f(p.dPhi) * np.arange(0, M, dtype=np.float32)).astype(np.float32)
        terms = (b[1, 1:M + 1] * f(p.dPhi), (a[0, 1:M + 1] * ph) * f(p.dPhi), a[1, 1:M + 1] * f(p.dPhi))
        exp = np.array([oracle.seq_sum(np.ascontiguousarray(x, dtype=np.float32)) for x in terms], np.float32)
        assert np.array_equal(got[k].view(np.uint32), exp.view(np.uint32)), (k, got[k], exp)


def test_seq_sum_fuzz(slb, oracle):
    """Many random term streams of assorted shapes: magnitudes spread over a few to sixty
    binades, runs of equal terms, sign patterns, sparse large terms, cancellation."""
    import torch
    rng = np.random.default_rng(77)
    L = slb.lib()
    L.slb_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    out = torch.zeros(1, dtype=torch.float32, device="cuda")
    for case in range(120):
        n = int(rng.choice([1, 7, 64, 1000, 4095, 4096, 4097, 9000, 33000, 70001]))
        kind = case % 6
        if kind == 0:
            x = rng.standard_normal(n) * 2.0 ** rng.uniform(-20, 20)
        elif kind == 1:
            x = rng.standard_normal(n) * 2.0 ** rng.uniform(-60, 0, n)
        elif kind == 2:                                   # multiples of a power of two: ties everywhere
            x = rng.integers(-64, 65, n) * 2.0 ** float(rng.integers(-30, 5))
        elif kind == 3:                                   # smooth bump, like the solver's rows
            t = np.linspace(-4, 4, n)
            x = np.exp(-rng.uniform(0.5, 5) * t * t) * rng.choice([1.0, -1.0]) * (8.0 / n)
            if case % 12 == 3:
                x = x * t
        elif kind == 4:                                   # mostly tiny, a few huge
            x = rng.standard_normal(n) * 1e-6
            x[rng.integers(0, n, max(1, n // 500))] = rng.standard_normal(max(1, n // 500)) * 1e6
        else:                                             # pairs that cancel exactly, shuffled a little
            h = rng.standard_normal((n + 1) // 2)
            x = np.concatenate([h, -h])[:n]
            rng.shuffle(x[: n // 3])
        x = np.ascontiguousarray(x, dtype=np.float32)
        exp = oracle.seq_sum(x)
        dev = torch.from_numpy(x).cuda()
        assert L.slb_seq_sum(dev.data_ptr(), n, out.data_ptr()) == 0
        got = out.cpu().numpy()
        assert got.view(np.uint32)[0] == np.array([exp], np.float32).view(np.uint32)[0], (case, kind, n, got[0], exp)
