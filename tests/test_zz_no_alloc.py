"""slb_av_static(): av() with its reduction workspace in a statically allocated device array
(for callers that cannot allocate).  Same kernels, so av_data must come out bit for bit as
from av(), on two calls in a row, the second on a smaller lattice, in both av modes; the
workspace does not grow with the lattice, so wide lattices work too; and the call records
into a HIP stream capture, where any allocation would be an error."""
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


def test_static_workspace_av_takes_any_width(slb, capfd):
    """The static array holds at most 1024 block partials whatever the width: beyond 2^21
    columns the blocks of the tree sums take several passes, the exact sums use no partials."""
    import torch
    torch.cuda.set_device(0)
    L = slb.lib()
    for mode in (slb.SLB_AV_EXACT, slb.SLB_AV_TREE):
        L.slb_set_av_mode(mode)
        for M in (1 << 21, (1 << 22) + 12345):
            slb.configure(N=1, M=M, E_dc=1.0, E_omega=0.1, omega=3.0, mu=5.0, alpha=1.0, B=0.5,
                          PhiYmin=-4.0, PhiYmax=4.0, dt=0.001)
            pitch = slb.get_global("PADDED_MSIZE")
            g = torch.Generator(device="cpu").manual_seed(M)
            a = ((torch.rand((2, pitch), generator=g) - 0.4) * 1e-3).to(torch.float32).cuda()
            av1, av2 = torch.zeros(6).cuda(), torch.zeros(6).cuda()
            L.av(0, a.data_ptr(), a.data_ptr(), av1.data_ptr(), 0.5)
            assert L.slb_av_static(a.data_ptr(), a.data_ptr(), av2.data_ptr(), 0.5) == 0
            torch.cuda.synchronize()
            assert np.array_equal(av1.cpu().numpy().view(np.uint32), av2.cpu().numpy().view(np.uint32)), (mode, M)
    # what it does refuse: parameters never loaded is not reachable here (configure() loads
    # them), a null array is
    x = torch.zeros(8, dtype=torch.float32).cuda()
    assert L.slb_av_static(None, x.data_ptr(), x.data_ptr(), 0.5) == -1
    assert "null array" in capfd.readouterr().err


def _hip_runtime():
    """The HIP runtime this process has already loaded (the one libslb2d.so is bound to)."""
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert paths, "no HIP runtime loaded"
    return C.CDLL(sorted(paths)[0])


@pytest.mark.parametrize("mode", ["exact", "tree"])
def test_static_workspace_av_records_into_a_stream_capture(slb, mode):
    """What "allocates nothing" means to a caller: after slb_static_init() the whole call can
    be recorded between hipStreamBeginCapture and hipStreamEndCapture (global capture mode,
    in which hipMalloc or hipFree by this thread fail and invalidate the capture), and the
    graph, launched twice, does to av_data what two calls of av() do."""
    import torch
    torch.cuda.set_device(0)
    L = slb.lib()
    hip = _hip_runtime()
    vp = C.c_void_p
    hip.hipStreamCreate.argtypes = [C.POINTER(vp)]
    hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipStreamSynchronize.argtypes = [vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipStreamDestroy.argtypes = [vp]
    hip.hipGraphGetNodes.argtypes = [vp, vp, C.POINTER(C.c_size_t)]

    L.slb_set_av_mode(slb.SLB_AV_TREE if mode == "tree" else slb.SLB_AV_EXACT)
    a, b = _lattice(torch, slb, 3, 150001, 7)
    ref = torch.tensor([1.0, 0.5, -0.25, 0.75, 0.02, -0.01], dtype=torch.float32).cuda()
    got = ref.clone()
    t = 0.123
    assert L.slb_static_init() == 0
    torch.cuda.synchronize()
    stream, graph, graph_exec = vp(), vp(), vp()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    L.slb_set_stream(stream)
    try:
        assert hip.hipStreamBeginCapture(stream, 0) == 0          # hipStreamCaptureModeGlobal
        rc = L.slb_av_static(a.data_ptr(), b.data_ptr(), got.data_ptr(), t)
        end = hip.hipStreamEndCapture(stream, C.byref(graph))
        assert rc == 0 and end == 0 and graph.value, (rc, end)
        n_nodes = C.c_size_t(0)
        assert hip.hipGraphGetNodes(graph, None, C.byref(n_nodes)) == 0
        assert n_nodes.value >= 2                                  # the sums and the update, kernels only
        torch.cuda.synchronize()
        assert torch.equal(got, ref)                               # recorded, not run
        assert hip.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, 0) == 0
        for _ in range(2):
            assert hip.hipGraphLaunch(graph_exec, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
    finally:
        L.slb_set_stream(None)
        if graph_exec.value:
            hip.hipGraphExecDestroy(graph_exec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        hip.hipStreamDestroy(stream)
    for _ in range(2):
        L.av(0, a.data_ptr(), b.data_ptr(), ref.data_ptr(), t)
    torch.cuda.synchronize()
    assert np.array_equal(ref.cpu().numpy().view(np.uint32), got.cpu().numpy().view(np.uint32))
    assert got.cpu().numpy()[0] == 3.0
