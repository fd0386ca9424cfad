"""GPU-side cost of the slab driver around the two-steps-per-launch kernel, on ONE GPU:
a middle slab (neighbours on both sides) whose halo messages are looped back locally
(send buffer -> receive buffer copy) instead of travelling.  Times edge launches, packs,
unpacks and the spare-array bookkeeping; the transport itself is not in it.  The lattice
values are not physical (the ghost columns receive the slab's own edge data)."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
slb = importlib.import_module("super-lattice-boltzmann-2d_amd")
drv = importlib.import_module("super-lattice-boltzmann-2d_amd.dist")


class Loopback(drv.SlabRun):
    def _post(self, send, recv):
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is not None:
                recv[side].copy_(send[side], non_blocking=True)
        return [], []


class NcclSelf(drv.SlabRun):
    """The messages go through torch.distributed + RCCL for real, addressed to this rank
    itself (one process, backend nccl): the library's launch and host costs without xGMI."""
    def _post(self, send, recv):
        import torch.distributed as dist
        ops = []
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is not None:
                ops.append(dist.P2POp(dist.isend, send[side], 0))
                ops.append(dist.P2POp(dist.irecv, recv[side], 0))
        return (dist.batch_isend_irecv(ops) if ops else []), []


def main():
    nccl_self = "--nccl-self" in sys.argv
    if nccl_self:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    world, per_gpu, N = 3, 1 << 20, 128
    M = per_gpu * world
    torch.zeros(1, device="cuda")
    slb.configure(N=N, M=M, E_dc=1.0, E_omega=0.1, omega=100.0, mu=5.0, alpha=1.0, B=4000.0 / M,
                  PhiYmin=-4.0, PhiYmax=4.0, dt=0.001, t_start=1.0)
    if "--own-stream" in sys.argv:              # not the legacy null stream
        torch.cuda.set_stream(torch.cuda.Stream())
    slb.lib().slb_set_stream(torch.cuda.current_stream().cuda_stream)
    for rank, label in ((1, "middle slab (two neighbours)"), (0, "first slab (one neighbour)")):
        run = (NcclSelf if nccl_self else Loopback)(slb, rank, world)
        run.tiptoe()
        for _ in range(5):
            run.double_step()
        torch.cuda.synchronize()
        K = 100
        t0 = time.perf_counter()
        for _ in range(K):
            run.double_step()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{label}: {dt / K * 1e3:.3f} ms per double step "
              f"({N * per_gpu * 2 * K / dt:.3e} updates/s), host issue time {t_issue / K * 1e3:.3f} ms")
        run.close()


if __name__ == "__main__":
    main()
