"""slb_av_static(): av() with its reduction workspace in a statically allocated device array
(for callers that cannot allocate).  Same kernels, so av_data must come out bit for bit as
from av(), on two calls in a row, the second on a smaller lattice, in both av modes; and a
lattice beyond the stated limit is refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lattice(torch, slb, N, M, seed):
    d = slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=3.0, mu=5.0, alpha=1.0, B=0.5,
                      PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    pitch = d["PADDED_MSIZE"]
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.rand((N + 1, pitch), generator=g) - 0.3).to(torch.float32).cuda()
    b = (torch.rand((N + 1, pitch), generator=g) - 0.5).to(torch.float32).cuda()
    return a, b


@pytest.mark.parametrize("mode", ["exact", "tree"])
def test_static_workspace_av_equals_av(slb, mode):
    import torch
    torch.cuda.set_device(0)
    L = slb.lib()
    L.slb_set_stream(None)
    L.slb_set_av_mode(slb.SLB_AV_TREE if mode == "tree" else slb.SLB_AV_EXACT)
    assert L.slb_static_init() == 0
    # the second lattice is smaller: fewer blocks of partials, stale ones from the first call
    # stay in the workspace behind them
    for N, M, seed in ((4, 300000, 1), (4, 70001, 2), (3, 999, 3)):
        a, b = _lattice(torch, slb, N, M, seed)
        ref = torch.tensor([2.0, 0.25, -0.5, 0.125, 0.01, -0.02], dtype=torch.float32).cuda()
        got = ref.clone()
        for k in range(3):
            t = 0.37 + 0.001 * k
            L.av(0, a.data_ptr(), b.data_ptr(), ref.data_ptr(), t)
            assert L.slb_av_static(a.data_ptr(), b.data_ptr(), got.data_ptr(), t) == 0
        torch.cuda.synchronize()
        assert np.array_equal(ref.cpu().numpy().view(np.uint32), got.cpu().numpy().view(np.uint32)), (mode, M)
        assert ref.cpu().numpy()[0] == 5.0


def test_static_workspace_av_refuses_what_it_cannot_hold(slb, capfd):
    import torch
    torch.cuda.set_device(0)
    L = slb.lib()
    slb.configure(N=1, M=(1 << 21) + 1, E_dc=1.0, E_omega=0.1, omega=3.0, mu=5.0, alpha=1.0, B=0.5,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    x = torch.zeros(8, dtype=torch.float32).cuda()
    assert L.slb_av_static(x.data_ptr(), x.data_ptr(), x.data_ptr(), 0.5) == -1      # refused before any launch
    assert "beyond the 2097152 columns" in capfd.readouterr().err
    slb.configure(N=1, M=1 << 21, E_dc=1.0, E_omega=0.1, omega=3.0, mu=5.0, alpha=1.0, B=0.5,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
    pitch = slb.get_global("PADDED_MSIZE")
    a = torch.full((2, pitch), 0.001, dtype=torch.float32).cuda()
    av1, av2 = torch.zeros(6).cuda(), torch.zeros(6).cuda()
    L.av(0, a.data_ptr(), a.data_ptr(), av1.data_ptr(), 0.5)
    assert L.slb_av_static(a.data_ptr(), a.data_ptr(), av2.data_ptr(), 0.5) == 0     # the limit itself is fine
    torch.cuda.synchronize()
    assert np.array_equal(av1.cpu().numpy().view(np.uint32), av2.cpu().numpy().view(np.uint32))
