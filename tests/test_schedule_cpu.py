"""Launch schedules of the multi-step kernels, checked on the host: whatever the lattice shape and
the number of workgroups the device holds, every row of every column-strip group must be formed by
exactly one workgroup.  Uses the library's own schedule builder and the very function its kernels
place their workgroups with (slb_debug_schedule runs both on the host; no GPU involved)."""
import ctypes as C

import numpy as np
import pytest

import slab_helpers as sh


@pytest.fixture(scope="module")
def lib():
    slb, _ = sh.import_pkg()
    return slb.lib()


def _schedule(lib, kind, steps, N, col_first, col_last, block, n_chunk, slots):
    info = (C.c_int * 4)()
    groups_max = (col_last + 64) // (60 * 4 * (block // 64)) + 2
    cover = np.full(groups_max * N, -7, dtype=np.int32)
    total = lib.slb_debug_schedule(kind, steps, N, col_first, col_last, block, n_chunk, slots, info,
                                   cover.ctypes.data_as(C.POINTER(C.c_int)), cover.size)
    assert total > 0, (total, kind, steps, N, col_first, col_last, block, n_chunk, slots)
    groups, phases, per_group, jj0 = info[0], info[1], info[2], info[3]
    return total, groups, phases, per_group, jj0, cover[:groups * N].reshape(groups, N)


def _check(lib, kind, steps, N, col_first, col_last, block, n_chunk, slots):
    total, groups, phases, per_group, jj0, cover = _schedule(lib, kind, steps, N, col_first, col_last, block, n_chunk, slots)
    what = (kind, steps, N, col_first, col_last, block, n_chunk, slots)
    assert (cover == 1).all(), (what, np.argwhere(cover != 1)[:5].tolist())
    # the strip groups reach from at or before the first column to the last one, none beyond
    assert jj0 <= col_first and jj0 + groups * per_group > col_last >= jj0 + (groups - 1) * per_group, what
    assert 1 <= phases <= 3 and total >= groups, what
    return total, groups, phases


@pytest.mark.parametrize("kind,steps", [(2, 2), (3, 2), (3, 3), (4, 2), (4, 3), (5, 3)])
def test_every_row_of_every_strip_group_once_random_shapes(lib, kind, steps):
    rng = np.random.default_rng(100 * kind + steps)
    seen_phases = set()
    for _ in range(1500):
        N = int(rng.choice([1, 2, 3, 5, 7, 8, 15, 16, 17, 23, 24, 25, 31, 32, 33, 63, 64, 100, 127, 128, 129, 255, 512]))
        block = 256 if kind != 2 or rng.random() < 0.7 else int(rng.choice([64, 128, 192]))
        per_group = 240 * (block // 64)
        groups_wanted = int(rng.choice([1, 2, 3, 7, 8, 9, 63, 64, 65, 300, 511, 512, 513, 514, 515, 600, 1023, 1024, 1025, 1026, 1093,
                                        1500, 2047, 2048, 2049, 2050, 3000, 4371]))
        col_first = int(rng.choice([1, 2, 4, 6, 31, 32, 33, 100]))
        col_last = col_first + groups_wanted * per_group - int(rng.integers(0, per_group))
        col_last = max(col_last, col_first)
        slots = int(rng.choice([0, 1, 2, 8, 256, 300, 511, 512, 513, 768, 1024, 2048]))
        n_chunk = int(rng.choice([0, 0, 0, 1, 3, 16, 43, 64, 200]))
        _, _, phases = _check(lib, kind, steps, N, col_first, col_last, block, n_chunk, slots)
        seen_phases.add(phases)
    assert seen_phases >= ({1, 2, 3} if kind in (3, 5) else {1, 2}), seen_phases


@pytest.mark.parametrize("N,cols,slots,expect", [
    (128, 1 << 20, 512, (3, 3)),       # the benchmark shape on MI355X: edge groups, two rounds of whole strips, the rest in chunks
    (512, 1 << 20, 512, (3, 3)),
    (128, 1 << 23, 512, (3, 3)),
    (64, 1 << 18, 512, (3, 1)),        # narrower than one round of workgroups: chunks throughout
    (128, 1 << 20, 0, (3, 1)),         # occupancy unknown: chunks throughout
])
def test_schedules_of_the_full_size_shapes(lib, N, cols, slots, expect):
    kind, want_phases = expect
    total, groups, phases = _check(lib, kind, 3, N, 1, cols + 1, 256, 0, slots)
    assert phases == want_phases and groups == (cols + 1 + 960) // 960
    total2, groups2, _ = _check(lib, 2, 2, N, 1, cols + 1, 256, 0, slots)
    assert groups2 == groups


def test_bad_arguments_are_refused(lib):
    info = (C.c_int * 4)()
    cover = (C.c_int * 16)()
    assert lib.slb_debug_schedule(3, 3, 128, 1, 1 << 20, 256, 0, 512, info, cover, 16) == -3     # cover too short
    assert lib.slb_debug_schedule(6, 3, 8, 1, 100, 256, 0, 0, info, cover, 16) == -1
    assert lib.slb_debug_schedule(3, 4, 8, 1, 100, 256, 0, 0, info, cover, 16) == -1
    assert lib.slb_debug_schedule(3, 3, 0, 1, 100, 256, 0, 0, info, cover, 16) == -1
