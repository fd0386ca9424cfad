"""phi_y slab partitioning across GPUs: one process per GPU, halo columns moved
with torch.distributed point-to-point operations (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" for CPU rehearsals).

The reference is single-GPU; this driver is additive.  Coupling along phi_y is
+-1 column on the lattice a sweep reads as stencil (reference:
src/boltzmann_c_solver.c:370-372), and that lattice is the one the PREVIOUS
sweep wrote.  Ranks form a non-periodic chain: rank r talks to r-1 and r+1.
A slab keeps G ghost columns per side (the engine's choice: 6 in host/slb_solver.c, so that the C slab
loop can take three steps per launch; the forms of this driver read at most 4 deep); owned columns are
G .. G+W-1.

Three step forms, bit-identical to each other and to a single-domain run:

* ``step()``  -- the reference's two sweeps, each as
      edge columns -> pack 1 column -> send/recv -> interior columns -> wait -> unpack
  (two exchanges per step).
* ``step_fused()`` -- both sweeps in one launch per column range.  A slab forms
  the main sweep one column into its ghost zone itself (same inputs, same bits as
  the owner), so only the NEW half-step lattice travels: 2 columns per side,
  ONE exchange per step:
      edge ranges -> pack 2 columns -> send/recv -> interior range -> wait -> unpack

* ``double_step()`` -- two loop iterations (four sweeps) in one launch per column
  range.  A slab recomputes the three earlier sweeps up to three columns into
  its ghost zone; afterwards the new main lattice (3 columns) and the new
  half-step lattice (4 columns) travel, ONE message per neighbour per TWO steps.
  On the GPU the two kinds of work go to two streams that run side by side:
      side stream:  edge ranges -> pack -> send/recv -> wait -> unpack
      main stream:  interior range
  Each waits for the other's PREVIOUS double step only (the interior reads what the
  last edge ranges wrote and the other way round; this step's results go to the
  spare arrays), so the interior launches follow each other without a gap and
  everything about the halo hides behind them.

Either way the few-KB messages travel while the interior is being updated.
After every step form all G ghost columns of both current lattices are valid,
so the forms can be mixed freely.
All arithmetic stays in the engine (the HIP kernels); this file only orders
launches and moves 2*ncols*(N+1) floats per side.
"""
import collections
import contextlib
import os
import ctypes as C

import torch
import torch.distributed as dist


class HipEngine:
    """The C host engine (host/slb_solver.c) on the current HIP device."""

    def __init__(self, slb, rank, world):
        self.slb = slb
        self.sol = slb.Solver(rank, world)
        s = self.sol.s
        self.rank, self.world = rank, world
        self.N, self.W, self.G = s.N, s.W, s.G
        self.j_last_main, self.j_last_half = s.main_slab.j_last, s.half_slab.j_last
        self.device = torch.device("cuda", torch.cuda.current_device())

    def new_buffer(self, ncols):
        return torch.empty(2 * ncols * (self.N + 1), dtype=torch.float32, device=self.device)

    def tiptoe(self):
        self.sol.tiptoe()

    def cosines(self):
        return self.slb.step_cosines(self.sol.s.t)

    def sweep_range(self, lattice, j_first, j_last, c0, c1):
        self.sol.sweep_range(lattice, j_first, j_last, c0, c1)

    def step_fused_range(self, j2_first, j2_last, cosines):
        self.sol.step_fused_range(j2_first, j2_last, cosines)

    def cosines2(self):
        return self.slb.step2_cosines(self.sol.s.t)

    def double_step_range(self, j_first, j_last, cosines8, rows=None):
        self.sol.double_step_range(j_first, j_last, cosines8,
                                   None if rows is None else (rows[0].data_ptr(), rows[1].data_ptr()))

    def new_rows(self):
        """A [3][pitch] buffer for the rows av() reads (a rows 0, 1 and b row 1)."""
        return torch.zeros((3, self.sol.s.pitch), dtype=torch.float32, device=self.device)

    def av_partial_rows(self, rows):
        sums = torch.zeros(4, dtype=torch.float64, device=self.device)
        rc = self.slb.lib().slb_av_partial(C.byref(self.sol.s.av_slab), rows.data_ptr(),
                                           rows.data_ptr() + 4 * self.sol.s.pitch, sums.data_ptr())
        if rc:
            raise RuntimeError(f"slb_av_partial -> {rc}")
        return sums

    def double_step_finish(self):
        self.sol.double_step_finish()

    # -- the pieces ExactAvChain needs ------------------------------------------
    def new_chain_slot(self):
        slot = _ChainSlot()
        slot.rows = (self.new_rows(), self.new_rows())
        # running sums of up to two iterations: where they start (from the slab before) and
        # where they end, on the device and in pinned host memory for the messages
        slot.init, slot.tot = (torch.zeros(6, dtype=torch.float32).pin_memory() for _ in range(2))
        slot.init_dev, slot.tot_dev = (torch.zeros(6, dtype=torch.float32, device=self.device) for _ in range(2))
        slot.done = None
        return slot

    def new_av_data(self):
        return torch.zeros(6, dtype=torch.float32, device=self.device)

    def copy_av_rows(self, rows):
        self.sol.copy_av_rows(rows.data_ptr())

    def mark(self):
        """An event after everything queued so far on the current stream."""
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def chain_launch(self, items, av_data, update):
        """Queue, on the stream of the sums, ONE launch with the exact-order sums of the row
        buffers of `items` (each: .slot, .n row buffers, .use_init, .events to wait for,
        .times), every set starting from its slot.init (else zero); then either the updates
        of av_data in item order (update: last slab) or the copies of the totals to slot.tot."""
        lib = self.slb.lib()
        if getattr(self, "av_stream", None) is None:
            self.av_stream = torch.cuda.Stream()
        st = self.av_stream
        vp = C.c_void_p
        pitch = self.sol.s.pitch
        a, b, ini, tot = [], [], [], []
        for it in items:
            for ev in it.events:
                st.wait_event(ev)
            for k in range(it.n):
                rows = it.slot.rows[k].data_ptr()
                a.append(rows)
                b.append(rows + 4 * pitch)                       # b + pitch floats = row 2
                ini.append(it.slot.init_dev.data_ptr() + 12 * k if it.use_init else None)
                tot.append(it.slot.tot_dev.data_ptr() + 12 * k)
        n = len(a)
        before = lib.slb_get_stream()
        try:
            with torch.cuda.stream(st):
                lib.slb_set_stream(st.cuda_stream)
                for it in items:
                    if it.use_init:
                        it.slot.init_dev.copy_(it.slot.init, non_blocking=True)
                rc = lib.slb_av_sums_exact_from_n(C.byref(self.sol.s.av_slab), n, (vp * n)(*a), (vp * n)(*b),
                                                  (vp * n)(*ini), (vp * n)(*tot))
                if rc:
                    raise RuntimeError(f"slb_av_sums_exact_from_n -> {rc}")
                for it in items:
                    if not update:
                        it.slot.tot.copy_(it.slot.tot_dev, non_blocking=True)
                        continue
                    for k, t in enumerate(it.times):
                        rc = lib.slb_av_update_exact(it.slot.tot_dev.data_ptr() + 12 * k, av_data.data_ptr(), t)
                        if rc:
                            raise RuntimeError(f"slb_av_update_exact -> {rc}")
                done = torch.cuda.Event()
                done.record(st)
                for it in items:
                    it.slot.done = done
        finally:
            lib.slb_set_stream(before)

    def chain_done(self, slot, block):
        if block:
            slot.done.synchronize()
            return True
        return slot.done.query()

    def av_data_host(self, av_data):
        if getattr(self, "av_stream", None) is not None:
            self.av_stream.synchronize()
        return av_data.cpu()

    def av_partial(self):
        sums = torch.zeros(4, dtype=torch.float64, device=self.device)
        self.sol.av_partial(sums.data_ptr())
        return sums

    def av_update(self, sums, av_data, t):
        rc = self.slb.lib().slb_av_update(sums.data_ptr(), av_data.data_ptr(), t)
        if rc:
            raise RuntimeError(f"slb_av_update -> {rc}")

    def pack(self, which, side, ncols, buf):
        self.sol.halo_pack(which, side, ncols, buf.data_ptr())

    def pack_double(self, left, right):
        self.sol.halo_pack_double(None if left is None else left.data_ptr(),
                                  None if right is None else right.data_ptr())

    def unpack_double(self, left, right):
        self.sol.halo_unpack_double(None if left is None else left.data_ptr(),
                                    None if right is None else right.data_ptr())

    def unpack(self, which, side, ncols, buf):
        self.sol.halo_unpack(which, side, ncols, buf.data_ptr())

    def swap(self):
        self.sol.swap()

    def advance(self):
        self.sol.advance()

    def close(self):
        self.sol.close()


class _ChainSlot:
    pass


class SlabRun:
    """Time stepping of one slab with halo exchange.  `engine` defaults to the
    HIP engine; tests may pass an object with the same methods."""

    def __init__(self, slb, rank, world, engine=None, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.eng = engine if engine is not None else HipEngine(slb, rank, world)
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        # [ncols][side] send / receive buffers
        self.send = {n: [self.eng.new_buffer(n), self.eng.new_buffer(n)] for n in (1, 2, 3, 4)}
        self.recv = {n: [self.eng.new_buffer(n), self.eng.new_buffer(n)] for n in (1, 2, 3, 4)}
        # a double step's message per side: 3 main-lattice columns, then 4 half-step columns
        self.send2 = [self.eng.new_buffer(7), self.eng.new_buffer(7)]
        self.recv2 = [self.eng.new_buffer(7), self.eng.new_buffer(7)]
        self.cpu_staging = dist.is_initialized() and dist.get_backend(group) == "gloo"
        # second stream for the halo work of double steps (HIP engine with neighbours only)
        self.side = None
        if isinstance(self.eng, HipEngine) and (self.left is not None or self.right is not None):
            self.main = torch.cuda.current_stream()
            # default priority: a high-priority side stream was seen to stall runs in which
            # several processes share one GPU (SLB_SIDE_PRIORITY=-1 asks for one, to look into it)
            self.side = torch.cuda.Stream(priority=int(os.environ.get("SLB_SIDE_PRIORITY", "0")))
            self.ev_main = torch.cuda.Event()       # the last interior launch
            self.ev_side = torch.cuda.Event()       # the last halo chain
            self.side_busy = False

    # -- transport: one message per neighbour, all of them in one batch ---------
    def _post(self, send, recv):
        """Send send[side] to / receive recv[side] from each neighbour; returns what
        _complete() needs.  The messages travel while later launches run."""
        ops, staged = [], []
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is None:
                continue
            s_buf, r_buf = send[side], recv[side]
            if self.cpu_staging and s_buf.is_cuda:
                s_buf, r_buf = s_buf.cpu(), torch.empty_like(r_buf, device="cpu")
                staged.append((recv[side], r_buf))
            ops.append(dist.P2POp(dist.isend, s_buf, peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, r_buf, peer, group=self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return reqs, staged

    @staticmethod
    def _complete(posted):
        reqs, staged = posted
        for r in reqs:
            r.wait()
        for r_dev, r_buf in staged:
            r_dev.copy_(r_buf)

    # -- halo exchange of lattice `which` (see slb_solver_halo_pack) ----------
    def _start_exchange(self, which, ncols):
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is not None:
                self.eng.pack(which, side, ncols, self.send[ncols][side])
        return self._post(self.send[ncols], self.recv[ncols]), ncols

    def _finish_exchange(self, which, pending):
        posted, ncols = pending
        self._complete(posted)
        for side, peer in ((0, self.left), (1, self.right)):
            if peer is not None:
                self.eng.unpack(which, side, ncols, self.recv[ncols][side])

    # -- both lattices of a double step in one message per side -----------------
    def _double_halo(self, bufs, pack):
        e = self.eng
        left = bufs[0] if self.left is not None else None
        right = bufs[1] if self.right is not None else None
        if hasattr(e, "pack_double"):                 # one launch for everything
            (e.pack_double if pack else e.unpack_double)(left, right)
            return
        split = 2 * 3 * (e.N + 1)
        for side, buf in ((0, left), (1, right)):
            if buf is not None:
                move = e.pack if pack else e.unpack
                move(4, side, 3, buf[:split])         # spare main lattice
                move(5, side, 4, buf[split:])         # spare half-step lattice

    def exchange(self, which, ncols):
        self._finish_exchange(which, self._start_exchange(which, ncols))

    # -- sweep-by-sweep form ---------------------------------------------------
    def _sweep(self, lattice, which_out, ncols, c0, c1):
        e = self.eng
        first, last_owned = e.G, e.G + e.W - 1
        j_last = e.j_last_main if lattice == 0 else e.j_last_half
        lo, hi = first, j_last
        if self.left is not None:
            e.sweep_range(lattice, first, first + ncols - 1, c0, c1)
            lo = first + ncols
        if self.right is not None:
            e.sweep_range(lattice, last_owned - ncols + 1, last_owned, c0, c1)
            hi = last_owned - ncols
        pending = self._start_exchange(which_out, ncols)
        e.sweep_range(lattice, lo, hi, c0, c1)
        self._finish_exchange(which_out, pending)

    def _refresh_ghosts(self):
        """All ghost columns of the current main lattice (3 are ever read) and of
        the current half-step lattice (4), so that any step form may follow."""
        self.exchange(3, 3)
        self.exchange(0, 4)

    def tiptoe(self):
        """First half step; afterwards the half-step lattice's ghost columns are
        brought up to date (the main lattice's are from the initial state)."""
        self.eng.tiptoe()
        self.exchange(0, 4)

    def step(self):
        self.join()
        c = self.eng.cosines()
        self._sweep(0, 1, 1, c[0], c[1])   # main lattice: writes a/b[next], ships 1 edge column
        self._sweep(1, 2, 1, c[2], c[3])   # half-step lattice: writes a/b[next_hs]
        self.eng.swap()
        self.eng.advance()
        self._refresh_ghosts()

    # -- one-launch form -------------------------------------------------------
    def step_fused(self):
        self.join()
        e = self.eng
        c = e.cosines()
        first, last_owned = e.G, e.G + e.W - 1
        lo, hi = first, last_owned
        if self.left is not None:
            e.step_fused_range(first, first + 1, c)
            lo = first + 2
        if self.right is not None:
            e.step_fused_range(last_owned - 1, last_owned, c)
            hi = last_owned - 2
        pending = self._start_exchange(2, 2)         # new half-step lattice, 2 columns per side
        e.step_fused_range(lo, hi, c)
        self._finish_exchange(2, pending)
        e.swap()
        e.advance()
        self._refresh_ghosts()

    # -- two iterations per launch ----------------------------------------------
    def double_step(self, rows=None):
        """rows: optional pair of buffers from engine.new_rows(); they receive the rows av()
        reads of the main lattice after the first and after the second iteration (own
        columns), for av_rows()."""
        kw = {} if rows is None else {"rows": rows}
        if self.side is not None:
            return self._double_step_two_streams(kw)
        e = self.eng
        c = e.cosines2()
        first, last_owned = e.G, e.G + e.W - 1
        lo, hi = first, last_owned
        if self.left is not None:
            e.double_step_range(first, first + 3, c, **kw)
            lo = first + 4
        if self.right is not None:
            e.double_step_range(last_owned - 3, last_owned, c, **kw)
            hi = last_owned - 4
        # the edge columns of both new lattices sit in the spare arrays
        self._double_halo(self.send2, pack=True)
        posted = self._post(self.send2, self.recv2)
        e.double_step_range(lo, hi, c, **kw)
        self._complete(posted)
        self._double_halo(self.recv2, pack=False)
        e.double_step_finish()                       # the spares become the current lattices

    @contextlib.contextmanager
    def _on_side(self):
        """Launches of the engine and of torch.distributed go to the side stream."""
        L = self.eng.slb.lib()
        with torch.cuda.stream(self.side):
            L.slb_push_stream(self.side.cuda_stream)     # the calling thread's launches, for the duration
            try:
                yield
            finally:
                L.slb_pop_stream()

    def _double_step_two_streams(self, kw):
        e = self.eng
        c = e.cosines2()
        first, last_owned = e.G, e.G + e.W - 1
        lo = first + 4 if self.left is not None else first
        hi = last_owned - 4 if self.right is not None else last_owned
        e.slb.lib().slb_set_stream(self.main.cuda_stream)
        e.sol.prepare_double_step()                  # spare arrays, if stale, on the main stream
        # halo chain: reads what the last interior launch wrote, writes the spare arrays'
        # edge and ghost columns (which that launch read)
        self.side.wait_event(self.ev_main) if self.side_busy else self.side.wait_stream(self.main)
        with self._on_side():
            if self.left is not None:
                e.double_step_range(first, first + 3, c, **kw)
            if self.right is not None:
                e.double_step_range(last_owned - 3, last_owned, c, **kw)
            self._double_halo(self.send2, pack=True)
            posted = self._post(self.send2, self.recv2)
        # interior: reads what the last halo chain wrote, writes the spare arrays' interior
        # (which that chain read)
        if self.side_busy:
            self.main.wait_event(self.ev_side)
        e.double_step_range(lo, hi, c, **kw)
        self.ev_main = torch.cuda.Event()
        self.ev_main.record(self.main)
        with self._on_side():
            self._complete(posted)
            self._double_halo(self.recv2, pack=False)
            self.ev_side = torch.cuda.Event()
            self.ev_side.record(self.side)
        self.side_busy = True
        e.double_step_finish()                       # the spares become the current lattices

    def join(self):
        """Make the main stream wait for the side stream's halo work (before anything but
        another double step touches the lattices)."""
        if self.side is not None and self.side_busy:
            self.main.wait_event(self.ev_side)
            self.side_busy = False

    # -- harmonic reductions over all slabs ---------------------------------------
    def av(self, av_data, t):
        """The reference's av() for the lattice the last step produced: every slab sums
        its own columns (wave64 tree, double), the three sums are added across slabs,
        and every rank applies the same update to its copy of av_data (6 floats on
        the device).  `t` is the loop time of the step just taken."""
        self.join()
        sums = self.eng.av_partial()
        if self.world > 1:
            if self.cpu_staging and sums.is_cuda:
                host = sums.cpu()
                dist.all_reduce(host, group=self.group)
                sums.copy_(host)
            else:
                dist.all_reduce(sums, group=self.group)
        self.eng.av_update(sums, av_data, t)

    def av_rows(self, rows, av_data, t):
        """av() from a rows buffer that a double_step(rows=...) filled; t = the loop time of
        the iteration the buffer belongs to."""
        self.join()
        sums = self.eng.av_partial_rows(rows)
        if self.world > 1:
            if self.cpu_staging and sums.is_cuda:
                host = sums.cpu()
                dist.all_reduce(host, group=self.group)
                sums.copy_(host)
            else:
                dist.all_reduce(sums, group=self.group)
        self.eng.av_update(sums, av_data, t)

    def rows_written(self):
        """Events after which the row buffers of the last double_step(rows=...) are complete."""
        if self.side is not None and self.side_busy:
            return [self.ev_main, self.ev_side]      # interior launch; edge launches and halo chain
        return [self.eng.mark()]

    def close(self):
        self.join()
        if self.side is not None:
            torch.cuda.synchronize()
        self.eng.close()


class ExactAvChain:
    """av() over phi_y slabs with the bits of the reference's av(): its three sums are float
    running sums over ALL columns in order (reference: src/boltzmann_c_solver.c:413-437), so
    slab r has to start from the totals slab r-1 ended with.  The totals of one iteration
    (3 floats; 6 for a double step) travel down the chain of ranks as host messages; the last
    slab applies the running-average updates in step order, and finish() hands av_data to
    every rank.  The sums of different iterations do not depend on each other, so nothing
    here blocks the time loop: an iteration's rows wait in one of `slots` buffers until the
    totals from the left have arrived, and the loop only waits when it needs a buffer back.

    torch's current stream has to be the library's launch stream (slb_set_stream), as in
    slab_program.py: the events that order the sums after the steps are recorded on it.

        slot = chain.acquire()
        run.double_step(rows=slot.rows)
        chain.submit(slot, [t, t + dt])          # or [t] / only_second=True, see submit()
        ...
        av6 = chain.finish()
    """

    def __init__(self, run, slots=48, lockstep=None, max_batch=4):
        self.run, self.eng = run, run.eng
        self.rank, self.world = run.rank, run.world
        # With halos staged through host memory (gloo rehearsals) an exchange blocks the host
        # until the neighbours have posted theirs; a rank waiting here for totals from a rank
        # that waits there for its halo would never be served.  There every submit() sees its
        # iteration through (ranks then move in step); over RCCL, where exchanges only block
        # streams, iterations are left in flight.
        self.lockstep = run.cpu_staging if lockstep is None else lockstep
        self.is_first, self.is_last = self.rank == 0, self.rank == self.world - 1
        self.group = None
        if self.world > 1:
            # few-byte host messages on a gloo group of their own, so that they are never
            # matched with halo messages between the same two ranks
            self.group = dist.new_group(backend="gloo")
        self.max_batch = max_batch                     # iterations per launch of the sums: 2 lattices each, 8 per launch
        # slots: a hop of the chain takes one launch of the sums (about 2 ms next to the time
        # loop) plus the wait for the next poll (one double step, about 1.1 ms); eight slabs
        # are then some twenty double steps behind rank 0, which 48 buffers (25 MB each at
        # 2^20 columns) cover with room to spare
        self.free = [self.eng.new_chain_slot() for _ in range(slots)]
        self.queue = collections.deque()
        self.av_data = self.eng.new_av_data()
        self.seq = 1

    def acquire(self):
        while not self.free:
            self._advance(block=True)
        return self.free.pop()

    def submit(self, slot, times, only_second=False):
        """The double step (or copy) that fills slot.rows has been queued.  times: loop times
        of the iterations to average, in order: both iterations of the double step, or one --
        the first row buffer, or with only_second the second (averaging starts between the
        two iterations)."""
        n = len(times)
        if only_second:
            slot.rows = (slot.rows[1], slot.rows[0])
        item = _ChainSlot()
        item.slot, item.n, item.times, item.stage = slot, n, [float(t) for t in times], 0
        item.use_init = not self.is_first
        item.events = self.run.rows_written()
        item.tag = self.seq
        self.seq += 1
        item.recv = item.send = None
        if not self.is_first:
            item.recv = dist.irecv(slot.init[:3 * n], src=self.rank - 1, group=self.group, tag=item.tag)
        self.queue.append(item)
        if self.lockstep:
            while self.queue:
                self._advance(block=True)
        else:
            self._advance(block=False)

    def submit_current(self, t):
        """av() of the current main lattice (after a single step)."""
        slot = self.acquire()
        self.eng.copy_av_rows(slot.rows[0])
        self.submit(slot, [t])

    def _advance(self, block):
        """Move every queued iteration as far as it can go without waiting; with block, wait
        for the oldest one to finish.  Iterations whose start values are here go to the
        device together, up to max_batch per launch (the sums of one lattice are a chain of
        dependent passes that keeps three workgroups busy for about as long as a double step
        keeps the whole device busy; the sums of several lattices side by side take no
        longer): at once when nothing is running, else when a full batch has gathered."""
        q = self.queue
        ready = []
        for it in q:
            if it.stage != 0:
                continue
            if it.recv is not None:
                if block and it is q[0]:
                    it.recv.wait()
                elif not it.recv.is_completed():
                    break
            ready.append(it)
            if len(ready) == self.max_batch:
                break
        running = any(it.stage == 1 and not self.eng.chain_done(it.slot, False) for it in q)
        if ready and (block or not running or len(ready) == self.max_batch):
            self.eng.chain_launch(ready, self.av_data, self.is_last)
            for it in ready:
                it.stage = 1
        for k, it in enumerate(q):
            wait = block and k == 0
            if it.stage == 1:                              # sums running
                if not self.eng.chain_done(it.slot, wait):
                    continue
                if self.is_last:
                    it.stage = 3
                else:
                    it.send = dist.isend(it.slot.tot[:3 * it.n], dst=self.rank + 1, group=self.group, tag=it.tag)
                    it.stage = 2
            if it.stage == 2:                              # totals on their way to the right slab
                if wait:
                    it.send.wait()
                if wait or it.send.is_completed():
                    it.stage = 3
        while q and q[0].stage == 3:
            self.free.append(q.popleft().slot)

    def finish(self):
        """Wait for everything submitted; returns av_data (6 floats, numpy) on every rank."""
        while self.queue:
            self._advance(block=True)
        host = self.eng.av_data_host(self.av_data)
        if self.world > 1:
            dist.broadcast(host, src=self.world - 1, group=self.group)
        return host.numpy().copy()


class CSlabRun:
    """The slab time loop of host/slb_slabs.c -- launch order, halo messages and the av chain
    in C, RCCL called from C -- for a process that torch.distributed.run started: the same
    tiptoe() / double_step() / close() as SlabRun.  torch.distributed only hands rank 0's RCCL
    identifier to the other ranks (backend "nccl"; with any other backend the messages go over
    sockets between the ranks, for rehearsals with several ranks on one GPU).  torch's current stream becomes the loop's main stream, so
    events recorded by the caller bracket the interior launches."""

    def __init__(self, slb, rank, world, backend="nccl"):
        self.slb, self.rank, self.world = slb, rank, world
        L = slb.lib()
        self.link = None
        if world > 1 and backend != "nccl":
            # rehearsal where RCCL cannot run (several ranks on one GPU): the same C loop with its
            # messages staged over sockets between the ranks, as `boltzmann_solver ranks=N` does
            fds = self._socket_mesh(rank, world)
            self.link = L.slb_link_open_fds(rank, world, (C.c_int * world)(*fds))
            if not self.link:
                raise RuntimeError("slb_link_open_fds failed")
        elif world > 1:
            ident = C.create_string_buffer(128)
            ok = rank != 0 or L.slb_link_unique_id(ident) == 0
            box = [ident.raw if ok else None]
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:                      # every rank learns of it and can fall back together
                raise RuntimeError("ncclGetUniqueId failed on rank 0 (is librccl.so.1 on the library path?)")
            self.link = L.slb_link_open_rccl(rank, world, box[0])
            if not self.link:
                raise RuntimeError("slb_link_open_rccl failed")
        self.r = L.slb_slabs_create(rank, world, self.link)
        if not self.r:
            raise RuntimeError("slb_slabs_create failed")
        self.stream = torch.cuda.ExternalStream(L.slb_slabs_main_stream(self.r))
        torch.cuda.set_stream(self.stream)

    @staticmethod
    def _socket_mesh(rank, world):
        """A TCP connection on the loopback interface to every other rank (ports handed round
        with torch.distributed); returns the descriptors, -1 for this rank itself.  The link
        owns them afterwards."""
        import socket
        import struct
        srv = socket.socket()
        srv.bind(("127.0.0.1", 0))
        srv.listen(world)
        ports = [None] * world
        dist.all_gather_object(ports, srv.getsockname()[1])
        socks = [None] * world
        for q in range(rank + 1, world):              # the lower rank connects, the higher one accepts
            c = socket.create_connection(("127.0.0.1", ports[q]))
            c.sendall(struct.pack("i", rank))
            socks[q] = c
        for _ in range(rank):
            c, _addr = srv.accept()
            hello = b""
            while len(hello) < 4:
                hello += c.recv(4 - len(hello))
            socks[struct.unpack("i", hello)[0]] = c
        srv.close()
        fds = []
        for q, c in enumerate(socks):
            if c is None:
                fds.append(-1)
                continue
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.setblocking(True)
            fds.append(c.detach())
        return fds

    def transport(self):
        return self.slb.lib().slb_link_transport(self.link).decode() if self.link else "none"

    @property
    def s(self):
        return self.slb.lib().slb_slabs_solver(self.r).contents

    def ranks_seen(self):
        return self.slb.lib().slb_slabs_ranks_seen(self.r)

    def tiptoe(self):
        self.slb.lib().slb_slabs_tiptoe(self.r)

    def double_step(self):
        self.slb.lib().slb_slabs_double_step(self.r, None, None)

    def multi_step(self, S):
        """S = 2 or 3 iterations per launch (3: ring_step_kernel); 11 halo columns per neighbour either way."""
        self.slb.lib().slb_slabs_multi_step(self.r, S, None)

    def step_fused(self):
        """One iteration (one launch per range) with the ghost columns brought up to date:
        what an odd number of steps ends on."""
        self.slb.lib().slb_slabs_step(self.r)

    step = step_fused

    def last_step(self):
        self.slb.lib().slb_slabs_last_step(self.r)

    def join(self):
        self.slb.lib().slb_slabs_join(self.r)

    def close(self):
        L = self.slb.lib()
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())
        L.slb_slabs_destroy(self.r)
        if self.link:
            L.slb_link_close(self.link)
        self.r = self.link = None
