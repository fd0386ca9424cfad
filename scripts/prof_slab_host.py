import cProfile, pstats, sys, os, importlib, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/scripts")
sys.argv = ["slab_overhead.py"]
import slab_overhead as so
import torch
slb, drv = so.slb, so.drv
world, per_gpu, N = 3, 1 << 20, 128
M = per_gpu * world
torch.zeros(1, device="cuda")
slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M, PhiYmin=-4.0, PhiYmax=4.0, dt=0.001, t_start=1.0)
slb.lib().slb_set_stream(torch.cuda.current_stream().cuda_stream)
for rank in (1, 0, 1):
    run = so.Loopback(slb, rank, world)
    run.tiptoe()
    for _ in range(5): run.double_step()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for _ in range(100): run.double_step()
    ti = time.perf_counter() - t0
    pr.disable()
    torch.cuda.synchronize()
    print("rank", rank, "issue ms", ti * 10, "total ms", (time.perf_counter() - t0) * 10)
    if rank == 0:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    run.close()
