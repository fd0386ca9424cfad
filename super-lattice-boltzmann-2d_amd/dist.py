"""phi_y slab partitioning across GPUs: one process per GPU, halo columns moved
with torch.distributed point-to-point operations (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" for CPU rehearsals).

The reference is single-GPU; this driver is additive.  Coupling along phi_y is
+-1 column on the stencil lattice of each sweep (reference:
src/boltzmann_c_solver.c:370-372), so before a sweep each slab needs one ghost
column per side of the lattice that sweep reads as stencil -- which is the
lattice the PREVIOUS sweep wrote.  Each sweep therefore runs as

    edge columns -> pack -> start send/recv -> interior columns -> wait -> unpack

so the exchange of the freshly written edge columns travels while the interior
is being updated.  Ranks form a non-periodic chain: rank r talks to r-1 and r+1.
All arithmetic stays in the engine (the HIP kernels); this file only orders
launches and moves 2*(N+1) floats per side.
"""
import torch
import torch.distributed as dist


class HipEngine:
    """The C host engine (host/slb_solver.c) on the current HIP device."""

    def __init__(self, slb, rank, world):
        self.slb = slb
        self.sol = slb.Solver(rank, world)
        s = self.sol.s
        self.rank, self.world = rank, world
        self.N, self.W = s.N, s.W
        self.j_last_main, self.j_last_half = s.main_slab.j_last, s.half_slab.j_last
        self.device = torch.device("cuda", torch.cuda.current_device())

    def new_buffer(self):
        return torch.empty(2 * (self.N + 1), dtype=torch.float32, device=self.device)

    def tiptoe(self):
        self.sol.tiptoe()

    def cosines(self):
        return self.slb.step_cosines(self.sol.s.t)

    def sweep_range(self, lattice, j_first, j_last, c0, c1):
        self.sol.sweep_range(lattice, j_first, j_last, c0, c1)

    def pack(self, which, side, buf):
        self.sol.halo_pack(which, side, buf.data_ptr())

    def unpack(self, which, side, buf):
        self.sol.halo_unpack(which, side, buf.data_ptr())

    def swap(self):
        self.sol.swap()

    def advance(self):
        self.sol.advance()

    def close(self):
        self.sol.close()


class SlabRun:
    """Time stepping of one slab with halo exchange.  `engine` defaults to the
    HIP engine; tests may pass an object with the same methods."""

    def __init__(self, slb, rank, world, engine=None, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.eng = engine if engine is not None else HipEngine(slb, rank, world)
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.send = [self.eng.new_buffer(), self.eng.new_buffer()]
        self.recv = [self.eng.new_buffer(), self.eng.new_buffer()]
        self.cpu_staging = dist.is_initialized() and dist.get_backend(group) == "gloo"

    # -- halo exchange of lattice `which` (see slb_solver_halo_pack) ----------
    def _start_exchange(self, which):
        ops, staged = [], []
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is None:
                continue
            self.eng.pack(which, side, self.send[side])
            if self.cpu_staging and self.send[side].is_cuda:
                s_buf, r_buf = self.send[side].cpu(), torch.empty_like(self.recv[side], device="cpu")
                staged.append((side, r_buf))
            else:
                s_buf, r_buf = self.send[side], self.recv[side]
            ops.append(dist.P2POp(dist.isend, s_buf, peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, r_buf, peer, group=self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return reqs, staged

    def _finish_exchange(self, which, pending):
        reqs, staged = pending
        for r in reqs:
            r.wait()
        for side, r_buf in staged:
            self.recv[side].copy_(r_buf)
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is not None:
                self.eng.unpack(which, side, self.recv[side])

    def exchange(self, which):
        self._finish_exchange(which, self._start_exchange(which))

    # -- one sweep: edges first, exchange its output while the interior runs --
    def _sweep(self, lattice, which_out, c0, c1):
        e = self.eng
        j_last = e.j_last_main if lattice == 0 else e.j_last_half
        lo, hi = 1, j_last
        if self.left is not None:
            e.sweep_range(lattice, 1, 1, c0, c1)
            lo = 2
        if self.right is not None:
            e.sweep_range(lattice, e.W, e.W, c0, c1)
            hi = e.W - 1
        pending = self._start_exchange(which_out)
        e.sweep_range(lattice, lo, hi, c0, c1)
        self._finish_exchange(which_out, pending)

    def tiptoe(self):
        """First half step; afterwards the half-step lattice's ghost columns are
        brought up to date so that step() can start with a main sweep."""
        self.eng.tiptoe()
        self.exchange(0)

    def step(self):
        c = self.eng.cosines()
        self._sweep(0, 1, c[0], c[1])      # main lattice: writes a/b[next], ships its edges
        self._sweep(1, 2, c[2], c[3])      # half-step lattice: writes a/b[next_hs], ships its edges
        self.eng.swap()
        self.eng.advance()

    def close(self):
        self.eng.close()
