"""GPU sweeps vs the CPU oracle, bit for bit, on seeded random fields."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields(rng, rows, pitch):
    # magnitudes like the solver's (O(1e-2)), both signs, plus a few exact zeros and denormals
    out = []
    for _ in range(7):
        x = (rng.standard_normal((rows, pitch)) * 0.05).astype(np.float32)
        x[rng.random((rows, pitch)) < 0.01] = 0.0
        x[rng.random((rows, pitch)) < 0.01] = np.float32(1e-41)
        out.append(x)
    return out


CASES = [
    # N, M, pitch (0 = M+3 padded to 32), variant, n_chunk
    (1, 5, 0, 0, 0),
    (2, 61, 0, 0, 0),
    (3, 61, 64, 2, 0),
    (8, 61, 67, 1, 0),          # odd pitch: one column per lane only
    (8, 253, 0, 3, 3),          # chunks that start on odd rows
    (9, 255, 0, 3, 4),
    (32, 4096, 0, 1, 0),
    (32, 4096, 0, 2, 0),
    (32, 4096, 0, 3, 0),
    (33, 1000, 0, 3, 7),
    (64, 8191, 0, 3, 16),
    (5, 20000, 0, 3, 0),
    (2, 61, 0, 4, 0),
    (9, 255, 0, 4, 4),
    (33, 1000, 0, 4, 7),
    (64, 8191, 0, 5, 16),
    (32, 4096, 0, 5, 0),
]


@pytest.mark.parametrize("N,M,pitch,variant,n_chunk", CASES)
def test_sweep_matches_oracle(slb, oracle, N, M, pitch, variant, n_chunk):
    import torch
    L = slb.lib()
    rng = np.random.default_rng(1234 + N * 7 + M)
    rows = N + 1
    pitch = pitch or (M + 3 + 31) // 32 * 32
    p = oracle.make_params(N=N, M=M, B=4000.0 / M if M > 100 else 0.7, omega=3.0)
    slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                  B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    for name in ("dPhi", "nu", "nu2", "nu_tilde", "bdt"):
        assert slb.get_global("host_" + name) == getattr(p, name), name

    a0, ac, bc, a_s, b_s, ao, bo = _fields(rng, rows, pitch)
    c0, c1 = np.float32(0.8775826), np.float32(0.8770)
    for j_first, j_last, m_off in ((1, M + 1, 0), (1, M, 0), (3, max(3, M // 2), 17)):
        exp_a, exp_b = ao.copy(), bo.copy()
        oracle.sweep(p, pitch, m_off, j_first, j_last, a0, ac, bc, a_s, b_s, exp_a, exp_b, c0, c1)

        dev = [torch.from_numpy(x).cuda() for x in (a0, ac, bc, a_s, b_s, ao, bo)]
        slab = slb.Slab(pitch, m_off, j_first, j_last)
        L.slb_set_tuning(variant, n_chunk, 0)
        rc = L.slb_sweep(C.byref(slab), *[d.data_ptr() for d in dev], c0, c1)
        L.slb_set_tuning(0, 0, 0)
        assert rc == 0
        torch.cuda.synchronize()
        got_a, got_b = dev[5].cpu().numpy(), dev[6].cpu().numpy()
        # bit-exact, including cells the sweep must leave untouched
        assert np.array_equal(got_a.view(np.uint32), exp_a.view(np.uint32))
        assert np.array_equal(got_b.view(np.uint32), exp_b.view(np.uint32))
        # inputs are not modified
        for d, h in zip(dev[:5], (a0, ac, bc, a_s, b_s)):
            assert np.array_equal(d.cpu().numpy().view(np.uint32), h.view(np.uint32))


def test_reference_entry_points(slb, oracle):
    """step_on_grid / step_on_half_grid / av through the host globals + load_data()."""
    import torch
    L = slb.lib()
    N, M = 16, 1021
    p = oracle.make_params(N=N, M=M, B=2.0, omega=2.0)
    d = slb.configure(N=N, M=M, E_dc=p.E_dc, E_omega=p.E_omega, omega=p.omega, mu=p.mu, alpha=p.alpha,
                      B=p.B, PhiYmin=p.PhiYmin, PhiYmax=p.PhiYmax, dt=p.dt)
    pitch, rows = d["PADDED_MSIZE"], N + 1
    assert pitch % 32 == 0 and pitch >= M + 3
    rng = np.random.default_rng(7)
    host = _fields(rng, rows, pitch) + _fields(rng, rows, pitch)[:2]   # a0, ac, bc, an, bn, ahs, bhs, anhs, bnhs
    a0, ac, bc, an, bn, ahs, bhs, anhs, bnhs = host
    dev = [torch.from_numpy(x).cuda() for x in host]
    ptr = [x.data_ptr() for x in dev]
    c0, c1 = np.float32(0.5), np.float32(0.25)

    # the oracle works on the unpadded CPU layout
    cpu = [np.ascontiguousarray(x[:, :M + 3]) for x in host]
    oracle.lib().orc_step_on_grid(C.byref(p), cpu[0], cpu[1], cpu[2], cpu[3], cpu[4], cpu[5], cpu[6], c0, c1)
    L.step_on_grid(0, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], 0.0, 0.0, c0, c1)
    oracle.lib().orc_step_on_half_grid(C.byref(p), cpu[0], cpu[3], cpu[4], cpu[5], cpu[6], cpu[7], cpu[8], c0, c1)
    L.step_on_half_grid(0, *ptr[:9], 0.0, 0.0, c0, c1)
    torch.cuda.synchronize()
    for k in (3, 4, 7, 8):
        got = dev[k].cpu().numpy()
        assert np.array_equal(got[:, :M + 3].view(np.uint32), cpu[k].view(np.uint32)), k
        assert np.array_equal(got[:, M + 3:], host[k][:, M + 3:])       # padding untouched

    # av, exact mode: bit-identical running averages over several calls
    av_cpu = np.zeros(6, dtype=np.float32)
    av_dev = torch.zeros(6, dtype=torch.float32, device="cuda")
    L.slb_set_av_mode(slb.SLB_AV_EXACT)
    for k, t in enumerate((0.1, 0.101, 0.102)):
        oracle.lib().orc_av(C.byref(p), cpu[3], cpu[4], av_cpu, np.float32(t))
        L.av(0, ptr[3], ptr[4], av_dev.data_ptr(), np.float32(t))
    torch.cuda.synchronize()
    assert np.array_equal(av_dev.cpu().numpy().view(np.uint32), av_cpu.view(np.uint32))

    # av, tree mode: same quantities up to the rounding noise of a float running sum
    av_tree = torch.zeros(6, dtype=torch.float32, device="cuda")
    L.slb_set_av_mode(slb.SLB_AV_TREE)
    for t in (0.1, 0.101, 0.102):
        L.av(0, ptr[3], ptr[4], av_tree.data_ptr(), np.float32(t))
    torch.cuda.synchronize()
    L.slb_set_av_mode(slb.SLB_AV_EXACT)
    got = av_tree.cpu().numpy()
    assert got[0] == 3.0
    scale = np.abs(av_cpu[1:]).max()
    assert np.all(np.abs(got[1:] - av_cpu[1:]) <= 2e-5 * scale + 1e-9)
