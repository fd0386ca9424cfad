"""Parity at the full shapes of the named configurations.  The oracle cannot run whole lattices
of this size in seconds, but it can run any WINDOW of columns (it takes a
column offset), so each step of the GPU run is checked on windows at the left
boundary, the centre and the right boundary, bit for bit; plus size-independent
properties: the norm stays 1, untouched cells stay untouched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL = list(range(9))         # 0 = a0, 1..4 = a[0..3], 5..8 = b[0..3]


def _check_step_on_window(sol, slb, oracle, p, c0, c1):
    """Take columns [c0, c1] of all nine arrays before a step, run the step on
    the GPU, replay both sweeps of that step on the window with the oracle."""
    s = sol.s
    M, ncols = p.M, c1 - c0 + 1
    before = [sol.window(k, c0, ncols) for k in ALL]
    cur, nxt, chs, nhs = s.current, s.next, s.current_hs, s.next_hs
    cos = slb.step_cosines(s.t)
    sol.step()
    a0 = before[0]
    A = lambda i: before[1 + i]
    B = lambda i: before[5 + i]
    # main sweep: writable global columns 1..M+1, and the window's own edge columns lack a neighbour
    g_lo, g_hi = max(c0 + 1, 1), min(c1 - 1, M + 1)
    oracle.sweep(p, ncols, c0, g_lo - c0, g_hi - c0, a0, A(cur), B(cur), A(chs), B(chs), A(nxt), B(nxt),
                 cos[0], cos[1])
    # half sweep reads the main sweep's output one column to each side
    h_lo, h_hi = max(g_lo + 1, 1) if c0 > 0 else 1, min(g_hi - 1, M)
    oracle.sweep(p, ncols, c0, h_lo - c0, h_hi - c0, a0, A(chs), B(chs), A(nxt), B(nxt), A(nhs), B(nhs),
                 cos[2], cos[3])
    for idx, lo, hi in ((nxt, g_lo, g_hi), (nhs, h_lo, h_hi)):
        for base, name in ((1, "a"), (5, "b")):
            got = sol.window(base + idx, lo, hi - lo + 1)
            exp = before[base + idx][:, lo - c0:hi - c0 + 1]
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (name, idx, lo, hi)


@pytest.mark.parametrize("N,M,steps", [
    (128, 1 << 20, 3),       # config 3
    (512, 1 << 20, 2),       # config 5: harmonic-heavy, I_n underflows to 0 for large n
    (128, 1 << 23, 2),       # config 4 on a single GPU (39 GB)
    (256, 1 << 20, 2),       # between the named shapes: another row-chunk length
    (130, 300001, 2),        # odd sizes: partial last chunk, ragged last strip
    (64, 1 << 25, 2),        # 8.7 GB per array (the reference's int byte counts end at 2 GiB): offsets beyond 2^31 floats
])
def test_full_shape_windows_bit_exact(slb, oracle, N, M, steps):
    p = oracle.make_params(N=N, M=M, B=4000.0 / M, omega=100.0)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    sol = slb.Solver()
    try:
        sol.tiptoe()
        sol.step()
        windows = [(0, 600), (M // 2 - 300, M // 2 + 300), (M // 2 + M // 8, M // 2 + M // 8 + 257),
                   (M - 500, M + 2)]
        for k in range(steps):
            c0, c1 = windows[k % len(windows)]
            _check_step_on_window(sol, slb, oracle, p, c0, c1)
        for c0, c1 in windows[steps % len(windows):][:2]:
            _check_step_on_window(sol, slb, oracle, p, c0, c1)
        s = sol.s
        # norm = 2 pi sqrt(alpha) sum_m a_0(m) dPhi stays 1 (double sum of the float terms)
        row0 = sol.rows(1 + s.current, 0, 1)[0]
        norm = 2 * np.pi * np.sqrt(p.alpha) * np.sum(row0[1:M + 1].astype(np.float64) * p.dPhi)
        assert abs(norm - 1) < 2e-4
        # cells no sweep may write: row N of a zero-initialised buffer, row 0 of b
        assert not sol.rows(1 + 1, N, 1).any()
        assert not sol.rows(5 + s.current, 0, 1).any()
    finally:
        sol.close()


def _stage_ranges(c0, c1, M, stages):
    """Columns each of `stages` consecutive sweeps can form from a window [c0, c1] of the
    lattices before the first of them: a sweep needs its stencil one column to each side, so
    the range shrinks by a column per sweep at an edge of the window that is not the edge of
    the lattice; at the lattice's own edges the sweeps' ranges are 1..M+1 (main) and 1..M."""
    out, lo, hi = [], c0, c1
    for q in range(stages):
        top = M + 1 if q % 2 == 0 else M
        lo = 1 if c0 == 0 else lo + 1
        hi = top if c1 == M + 2 else min(hi - 1, top)
        out.append((lo, hi))
    return out


def _check_double_step_on_window(sol, slb, oracle, p, c0, c1, rows=None):
    """The default form -- two iterations per launch -- held to the oracle directly: columns
    [c0, c1] of all nine arrays before the launch, the launch on the GPU, then its four sweeps
    replayed on the window with the oracle (reference: src/boltzmann_c_solver.c:164-194,
    355-411).  Each in-between lattice starts from the buffer the reference's ping-pong would
    have put it in, the final two from the current lattices (the spare arrays are copies of
    them), so every cell no sweep writes is what the reference finds there."""
    s = sol.s
    M, ncols = p.M, c1 - c0 + 1
    before = [sol.window(k, c0, ncols) for k in ALL]
    cur, nxt, chs, nhs = s.current, s.next, s.current_hs, s.next_hs
    cos = slb.step2_cosines(s.t)
    if rows is None:
        sol.double_step()
    else:
        sol.double_step_rows(rows[0].data_ptr(), rows[1].data_ptr())
    assert (s.current, s.current_hs) == (cur, chs)
    a0 = before[0]
    A = lambda i: before[1 + i]
    B = lambda i: before[5 + i]
    rng = _stage_ranges(c0, c1, M, 4)
    o1a, o1b = A(nxt).copy(), B(nxt).copy()
    o2a, o2b = A(nhs).copy(), B(nhs).copy()
    o3a, o3b = A(cur).copy(), B(cur).copy()
    o4a, o4b = A(chs).copy(), B(chs).copy()
    sw = lambda q, ac, bc, a_s, b_s, ao, bo: oracle.sweep(p, ncols, c0, rng[q][0] - c0, rng[q][1] - c0, a0, ac, bc,
                                                         a_s, b_s, ao, bo, cos[2 * q], cos[2 * q + 1])
    sw(0, A(cur), B(cur), A(chs), B(chs), o1a, o1b)
    sw(1, A(chs), B(chs), o1a, o1b, o2a, o2b)
    sw(2, o1a, o1b, o2a, o2b, o3a, o3b)
    sw(3, o2a, o2b, o3a, o3b, o4a, o4b)
    for idx, (lo, hi), ea, eb in ((cur, rng[2], o3a, o3b), (chs, rng[3], o4a, o4b)):
        for base, exp, name in ((1, ea, "a"), (5, eb, "b")):
            got = sol.window(base + idx, lo, hi - lo + 1)
            assert np.array_equal(got.view(np.uint32), exp[:, lo - c0:hi - c0 + 1].view(np.uint32)), (name, idx, lo, hi)
    if rows is not None:
        import torch
        torch.cuda.synchronize()
        # the three rows av() reads of the lattice after the first iteration (never stored
        # itself) and after the second
        for buf, (lo, hi), ea, eb in ((rows[0], rng[0], o1a, o1b), (rows[1], rng[2], o3a, o3b)):
            hi = min(hi, M)                          # av() sums columns 1..M
            got = buf[:, lo:hi + 1].cpu().numpy()
            exp = np.stack([ea[0, lo - c0:hi - c0 + 1], ea[1, lo - c0:hi - c0 + 1], eb[1, lo - c0:hi - c0 + 1]])
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), ("rows", lo, hi)


@pytest.mark.parametrize("with_rows", [False, True])
@pytest.mark.parametrize("N,M", [
    (128, 1 << 20),          # config 3: 1.4 rounds of workgroups, whole strips first, the rest in chunks
    (512, 1 << 20),          # config 5
    (128, 1 << 23),          # config 4 on a single GPU: eleven rounds
    (128, 800003),           # just over one round of workgroups
    (64, 1 << 18),           # config 2's shape: every strip cut into short chunks
    (64, 1 << 25),           # 8.7 GB per array
])
def test_full_shape_double_step_windows_bit_exact(slb, oracle, N, M, with_rows):
    """The program's and the benchmark's default kernel under the launch schedule those shapes
    get (whole column strips first from one round of workgroups up), with and without the row
    buffers of the averaging period, against the oracle's four sweeps -- no other HIP kernel
    in between.  Windows: both ends of the lattice, the centre, and the columns where the
    strips taken whole end and the strips cut into chunks begin (768 workgroups of 960 columns
    on a device that holds three workgroups on each of 256 compute units)."""
    import torch
    p = oracle.make_params(N=N, M=M, B=4000.0 / M, omega=100.0)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    sol = slb.Solver()
    try:
        sol.tiptoe()
        sol.step()
        rows = None
        if with_rows:
            rows = (torch.zeros((3, sol.s.pitch), dtype=torch.float32, device="cuda"),
                    torch.zeros((3, sol.s.pitch), dtype=torch.float32, device="cuda"))
        windows = [(0, 600), (M - 500, M + 2), (M // 2 - 300, M // 2 + 300)]
        seam = 768 * 960
        for k in range(1, 12):
            if k * seam + 400 < M:
                windows.append((k * seam - 400, k * seam + 400))
        windows = windows[:6]
        for c0, c1 in windows:
            _check_double_step_on_window(sol, slb, oracle, p, c0, c1, rows)
        row0 = sol.rows(1 + sol.s.current, 0, 1)[0]
        norm = 2 * np.pi * np.sqrt(p.alpha) * np.sum(row0[1:M + 1].astype(np.float64) * p.dPhi)
        assert abs(norm - 1) < 2e-4
    finally:
        sol.close()


def _check_multi_step_on_window(sol, slb, oracle, p, c0, c1, S, rows=None):
    """S iterations per launch held to the oracle directly: columns [c0, c1] of all nine arrays
    before the launch, the launch on the GPU, then its 2S sweeps replayed on the window with the
    oracle.  The lattice a sweep writes starts from the buffer the reference's ping-pong would
    have put it in (a/b[next], a/b[next_hs], a/b[current], a/b[current_hs], a/b[next], ...), so
    every cell no sweep writes is what the reference finds there."""
    s = sol.s
    M, ncols, Q = p.M, c1 - c0 + 1, 2 * S
    before = [sol.window(k, c0, ncols) for k in ALL]
    cur, nxt, chs, nhs = s.current, s.next, s.current_hs, s.next_hs
    t_before = s.t
    cos = slb.steps_cosines(s.t, S)
    if rows is None:
        sol.multi_step(S)
    else:
        sol.multi_step_rows(S, [r.data_ptr() for r in rows])
    a0 = before[0]
    A = lambda i: before[1 + i]
    B = lambda i: before[5 + i]
    rng = _stage_ranges(c0, c1, M, Q)
    outs, where = [], []
    for q in range(Q):
        into_next = (q // 2) % 2 == 0
        idx = (nxt if into_next else cur) if q % 2 == 0 else (nhs if into_next else chs)
        oa, ob = A(idx).copy(), B(idx).copy()
        ca, cb = (A(cur), B(cur)) if q == 0 else ((A(chs), B(chs)) if q == 1 else outs[q - 2])
        sa, sb = (A(chs), B(chs)) if q == 0 else outs[q - 1]
        oracle.sweep(p, ncols, c0, rng[q][0] - c0, rng[q][1] - c0, a0, ca, cb, sa, sb, oa, ob, cos[2 * q], cos[2 * q + 1])
        outs.append((oa, ob))
        where.append(idx)
    assert (s.current, s.current_hs) == (where[Q - 2], where[Q - 1])
    assert s.t != t_before
    for q in (Q - 2, Q - 1):
        lo, hi = rng[q]
        for base, exp, name in ((1, outs[q][0], "a"), (5, outs[q][1], "b")):
            got = sol.window(base + where[q], lo, hi - lo + 1)
            assert np.array_equal(got.view(np.uint32), exp[:, lo - c0:hi - c0 + 1].view(np.uint32)), (name, q, lo, hi)
    if rows is not None:
        import torch
        torch.cuda.synchronize()
        for i in range(S):                        # the three rows av() reads of the lattice after iteration i
            lo, hi = rng[2 * i]
            hi = min(hi, M)                       # av() sums columns 1..M
            ea, eb = outs[2 * i]
            got = rows[i][:, lo:hi + 1].cpu().numpy()
            exp = np.stack([ea[0, lo - c0:hi - c0 + 1], ea[1, lo - c0:hi - c0 + 1], eb[1, lo - c0:hi - c0 + 1]])
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), ("rows", i, lo, hi)


@pytest.mark.parametrize("with_rows", [False, True])
@pytest.mark.parametrize("N,M", [
    (128, 1 << 20),          # config 3: 2.1 rounds of workgroups of the ring kernel
    (512, 1 << 20),          # config 5
    (128, 1 << 23),          # config 4 on a single GPU
    (128, 800003),
    (64, 1 << 19),           # the narrowest lattice the program runs in threes
    (33, 600011),            # odd row count, ragged right edge
    (64, 1 << 25),           # 8.7 GB per array: row offsets beyond 2^31 floats, 34 rounds of workgroups
])
def test_full_shape_triple_step_windows_bit_exact(slb, oracle, N, M, with_rows):
    """The default of the program and of the benchmark on wide lattices -- three iterations per
    launch, ring_step_kernel: stage outputs in an LDS ring, branch-free steady passes with the
    next pass's rows requested ahead, edge strip groups first in short chunks, whole strips next,
    the rest in chunks -- against the oracle's six sweeps, no other HIP kernel in between
    (reference: src/boltzmann_c_solver.c:164-194, 355-411).  Windows: both ends of the lattice
    (the edge groups), the centre, the seam between the first edge group and the whole strips,
    and the columns where a round of 512 workgroups of 960 columns ends."""
    import torch
    p = oracle.make_params(N=N, M=M, B=4000.0 / M, omega=100.0)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    sol = slb.Solver()
    try:
        assert sol.steps_per_launch() == 3
        sol.tiptoe()
        sol.step()
        rows = None
        if with_rows:
            rows = [torch.zeros((3, sol.s.pitch), dtype=torch.float32, device="cuda") for _ in range(3)]
        windows = [(0, 600), (M - 500, M + 2), (M // 2 - 300, M // 2 + 300), (960 - 300, 960 + 300)]
        seam = 512 * 960
        for k in range(1, 18):
            if 960 + k * seam + 400 < M:
                windows.append((960 + k * seam - 400, 960 + k * seam + 400))
        windows = windows[:7]
        for c0, c1 in windows:
            _check_multi_step_on_window(sol, slb, oracle, p, c0, c1, 3, rows)
        _check_multi_step_on_window(sol, slb, oracle, p, 0, 600, 2)       # a double step in between: the spares follow
        _check_multi_step_on_window(sol, slb, oracle, p, M - 500, M + 2, 3, rows)
        row0 = sol.rows(1 + sol.s.current, 0, 1)[0]
        norm = 2 * np.pi * np.sqrt(p.alpha) * np.sum(row0[1:M + 1].astype(np.float64) * p.dPhi)
        assert abs(norm - 1) < 2e-4
    finally:
        sol.close()


def _rows_to_compare(N, M):
    """All rows of a lattice that fits comfortably, otherwise the rows around every place a
    workgroup's chunk of rows may begin or end, the first and last rows, and every 16th."""
    if (N + 1) * M <= 140 * (1 << 20):
        return list(range(N + 1))
    rows = set(range(4)) | set(range(N - 3, N + 1)) | set(range(0, N + 1, 16))
    for chunk in ((7 * N + 15) // 16, (N + 2) // 3, N // 2, 32, 64):
        for edge in range(chunk, N, chunk):
            rows |= set(range(max(edge - 4, 0), min(edge + 4, N) + 1))
    return sorted(rows)


@pytest.mark.parametrize("N,M", [
    (128, 1 << 20),          # config 3
    (512, 1 << 20),          # config 5
    (128, 1 << 23),          # config 4 on a single GPU
    (256, 1 << 20),
    (130, 300001),           # odd sizes: partial last chunk, ragged last strip
    (128, 800003),           # just over one round of workgroups: few strips left over, cut into many short chunks
    (65, 1000001),           # odd row count under the whole-strips-first schedule
    (200, 737290),           # a handful of strips beyond one round
])
def test_full_shape_fused_forms_equal_the_two_sweep_form(slb, N, M):
    """At full size the default form (two iterations per launch, rows per workgroup chosen by
    the library) and the one-launch-per-iteration form leave the same bits in both current
    lattices as the two-sweep form, which the test above holds to the oracle: four iterations
    each, row by row."""
    import torch
    kw = dict(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
              PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    slb.configure(**kw)
    one, two, fus = slb.Solver(), slb.Solver(), slb.Solver()
    try:
        one.tiptoe(); two.tiptoe(); fus.tiptoe()
        for _ in range(4):
            one.step()
            fus.step_fused()                 # one launch per iteration
        two.double_step(); two.double_step()
        torch.cuda.synchronize()
        s1 = one.s
        for other in (two, fus):
            s2 = other.s
            assert (s1.current, s1.current_hs, s1.t) == (s2.current, s2.current_hs, s2.t)
        for base, idx in ((1, s1.current), (5, s1.current), (1, s1.current_hs), (5, s1.current_hs)):
            for r in _rows_to_compare(N, M):
                x = one.rows(base + idx, r, 1).view(np.uint32)
                assert np.array_equal(x, two.rows(base + idx, r, 1).view(np.uint32)), ("double", base, idx, r)
                assert np.array_equal(x, fus.rows(base + idx, r, 1).view(np.uint32)), ("fused", base, idx, r)
    finally:
        one.close()
        two.close()
        fus.close()
