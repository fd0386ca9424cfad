"""The slab driver (super-lattice-boltzmann-2d_amd/dist.py) with world_size 2 and 3
over gloo on CPU: partition + halo exchange + edge-first ordering must
reproduce the single-domain oracle bit for bit.  The arithmetic here is the
CPU oracle's (the test checks the driver's host logic, not the kernels)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_binding as ob          # noqa: E402
import slab_helpers as sh            # noqa: E402

PHYS = dict(N=6, M=101, B=0.6, omega=40.0, E_omega=0.3)   # 3 slabs of 33-34 columns
STEPS = 7


def _worker(rank, world, port, outdir, mode):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    slb, drv = sh.import_pkg()
    p = ob.make_params(**PHYS)
    eng = sh.OracleEngine(ob, p, rank, world)
    run = drv.SlabRun(slb, rank, world, engine=eng)
    run.tiptoe()
    k = 0
    while k < STEPS:
        # the lattices in between two fused iterations are never stored, so finish on a
        # single step: then every buffer, not only the current ones, equals the oracle's
        if mode == "double" and k + 2 < STEPS:
            run.double_step()
            k += 2
            continue
        if mode == "mixed" and k % 4 == 1 and k + 2 < STEPS:
            run.double_step()
            k += 2
            continue
        if mode in ("fused", "double") or (mode == "mixed" and k % 2 == 0):
            run.step_fused()
        else:
            run.step()
        k += 1
    np.savez(os.path.join(outdir, f"r{rank}.npz"), t=eng.t,
             idx=np.array([eng.cur, eng.nxt, eng.cur_hs, eng.nxt_hs]),
             **{f"{w}{i}": eng.owned(w, i) for w in "ab" for i in range(4)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "sweeps"), (3, "sweeps"), (2, "fused"), (3, "fused"), (2, "double"), (3, "double"), (3, "mixed"),
                                        (4, "mixed"), (8, "double")])      # 8 slabs of 12-13 columns
def test_slab_driver_matches_single_domain(world, mode, tmp_path):
    mp.spawn(_worker, args=(world, sh.free_port(), str(tmp_path), mode), nprocs=world, join=True)
    p = ob.make_params(**PHYS)
    ref = ob.OracleState(p, threads=1)
    ref.step(STEPS)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert all(np.float32(x["t"]) == np.float32(ref.st.t) for x in parts)
    assert list(parts[0]["idx"]) == [ref.st.current, ref.st.next, ref.st.current_hs, ref.st.next_hs]
    for w in "ab":
        for i in range(4):
            got = np.concatenate([x[f"{w}{i}"] for x in parts], axis=1)
            exp = ref.array(w, i)
            assert got.shape == exp.shape
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (w, i)
    ref.close()


def _av_worker(rank, world, port, outdir, lockstep):
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    slb, drv = sh.import_pkg()
    p = ob.make_params(**PHYS)
    eng = sh.OracleEngine(ob, p, rank, world)
    run = drv.SlabRun(slb, rank, world, engine=eng)
    chain = drv.ExactAvChain(run, slots=8, lockstep=lockstep)
    run.tiptoe()
    av = sh.run_av_plan(run, chain, lambda: eng.t, np.float32(p.dt))
    np.save(os.path.join(outdir, f"av{rank}.npy"), av)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lockstep", [(1, False), (2, False), (3, False), (3, True), (8, False)])
def test_exact_av_chain_over_slabs_has_the_single_domain_bits(world, lockstep, tmp_path):
    """The running sums of av() handed from slab to slab: av_data on every rank equals the
    oracle's av() over the whole lattice, bit for bit, whatever mix of steps fed it."""
    if world == 1:
        _av_worker(0, 1, 0, str(tmp_path), lockstep)
    else:
        mp.spawn(_av_worker, args=(world, sh.free_port(), str(tmp_path), lockstep), nprocs=world, join=True)
    av_ref = sh.reference_av_plan(ob, ob.make_params(**PHYS))
    for r in range(world):
        got = np.load(tmp_path / f"av{r}.npy")
        assert np.array_equal(got.view(np.uint32), av_ref.view(np.uint32)), (r, got, av_ref)


def test_slab_columns_cover_the_grid():
    for M in (7, 100, 101, 4096):
        for world in (1, 2, 3, 8):
            if M < world:
                continue
            nxt = 1
            for r in range(world):
                W, first = sh.slab_columns(M, r, world)
                assert first == nxt and W >= 1
                nxt += W
            assert nxt == M + 1
