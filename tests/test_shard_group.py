"""CPU: the N > 1 host path of the sharded prover — the shared-memory exchange board (csrc/shard_group.hpp) — with 2 and 4
processes: every rank sees every rank's record for every exchange, in order, including when the ring wraps; a missing rank
is a timeout at open, not a hang."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(name, world, rank, rounds, q):
    sys.path.insert(0, ROOT)
    from jolt_atlas_amd import sharded
    try:
        g = sharded.ShardGroup(name, world, rank)
        ok = True
        for k in range(rounds):
            mine = np.array([rank * 1000003 + k, k * k + rank, 7, rank], dtype=np.uint64)
            allv = g.allgather(mine)
            for r in range(world):
                ok &= bool(np.array_equal(allv[r], np.array([r * 1000003 + k, k * k + r, 7, r], dtype=np.uint64)))
        big = g.allgather(np.full(62, rank, dtype=np.uint64))            # 496 bytes: the largest record
        ok &= all(int(big[r][0]) == r and int(big[r][61]) == r for r in range(world))
        g.close()
        q.put((rank, ok))
    except Exception as e:             # noqa: BLE001
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 4])
def test_allgather_over_shared_memory(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/atlas_test_{os.getpid()}_{world}"
    ps = [ctx.Process(target=_worker, args=(name, world, r, 50, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps: p.join(timeout=30)
    assert sorted(r for r, _ in res) == list(range(world))
    assert all(ok is True for _, ok in res), res


def _crash_worker(name, world, rank, q):
    """joins the board and exits WITHOUT closing it: the segment stays behind, complete (joined == world), as after a crashed run"""
    sys.path.insert(0, ROOT)
    from jolt_atlas_amd import sharded
    g = sharded.ShardGroup(name, world, rank)
    g.allgather(np.array([rank], dtype=np.uint64))
    q.put(rank)
    q.close(); q.join_thread()         # flush the feeder thread, then leave without any cleanup
    os._exit(0)


def _late_rank0_worker(name, world, rank, delay, q):
    import time
    time.sleep(delay)
    _worker(name, world, rank, 5, q)


def test_stale_segment_of_a_crashed_run_is_not_joined():
    """a rank that starts BEFORE rank 0 has replaced a complete segment left under the same name must not map the old one (its epoch and
    slots are dead): it waits for the fresh board, and the exchange works"""
    ctx = mp.get_context("spawn")
    name = f"/atlas_test_stale_{os.getpid()}"
    q = ctx.Queue()
    ps = [ctx.Process(target=_crash_worker, args=(name, 2, r, q)) for r in range(2)]
    for p in ps: p.start()
    for _ in ps: q.get(timeout=60)
    for p in ps: p.join(timeout=30)
    assert os.path.exists("/dev/shm" + name), "the crashed run must leave its segment behind for this test to mean anything"
    q2 = ctx.Queue()
    early = ctx.Process(target=_worker, args=(name, 2, 1, 5, q2))                    # rank 1 first: it finds the stale segment
    late = ctx.Process(target=_late_rank0_worker, args=(name, 2, 0, 1.0, q2))        # rank 0 a second later
    early.start(); late.start()
    res = [q2.get(timeout=120) for _ in range(2)]
    early.join(timeout=30); late.join(timeout=30)
    assert all(ok is True for _, ok in res), res
    assert not os.path.exists("/dev/shm" + name)                                     # rank 0's close unlinked the fresh board


def test_bad_arguments():
    sys.path.insert(0, ROOT)
    import jolt_atlas_amd as A
    from jolt_atlas_amd import sharded
    with pytest.raises(A.AtlasError):
        sharded.ShardGroup("/atlas_test_bad", 3, 0)          # not a power of two
    with pytest.raises(A.AtlasError):
        sharded.ShardGroup("/atlas_test_bad", 2, 2)          # rank out of range


def _giving_up_worker(name, world, rank, q):
    """ranks exchange three records; rank 1 then gives up INSTEAD of the fourth exchange; afterwards every rank exchanges once more"""
    sys.path.insert(0, ROOT)
    import time
    from jolt_atlas_amd import sharded
    try:
        g = sharded.ShardGroup(name, world, rank)
        for k in range(3): g.allgather(np.array([rank, k], dtype=np.uint64))
        seen = None
        if rank == 1:
            g.fail_exchange(-5)
        else:
            t0 = time.time()
            try:
                g.allgather(np.array([rank, 3], dtype=np.uint64))
                seen = "no error"
            except Exception:              # noqa: BLE001
                seen = g.remote_failed() + (time.time() - t0 < 5.0,)
        after = g.allgather(np.array([rank, 99], dtype=np.uint64))          # the exchange numbers still agree
        ok_after = all(int(after[r][0]) == r and int(after[r][1]) == 99 for r in range(world))
        g.close()
        q.put((rank, seen, ok_after))
    except Exception as e:             # noqa: BLE001
        q.put((rank, repr(e), False))


@pytest.mark.parametrize("world", [2, 4])
def test_a_rank_that_gives_up_is_learnt_at_once_and_the_board_stays_in_step(world):
    """the failure handshake (csrc/shard_group.hpp: fail_exchange): no rank waits for the board's 30 s timeout, each learns who gave up and
    why, and all of them have made the same number of exchanges"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/atlas_test_fail_{os.getpid()}_{world}"
    ps = [ctx.Process(target=_giving_up_worker, args=(name, world, r, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps: p.join(timeout=30)
    for rank, seen, ok_after in res:
        assert ok_after is True, res
        if rank == 1: assert seen is None
        else: assert seen == (1, -5, True), res
