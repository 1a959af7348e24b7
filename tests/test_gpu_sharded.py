"""GPU: one sumcheck instance and one MSM sharded over 2 and 4 ranks (processes sharing the
single GPU of the test box, gloo for the exchange — the same code path runs over RCCL on a
multi-GPU node).  Every rank's proof must equal the oracle's single-instance proof."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n", [(2, 12), (4, 14), (2, 3)])
def test_sharded_sumcheck_and_msm(tmp_path, world, n):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, ctypes as C
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch, torch.distributed as dist
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        import jolt_atlas_amd as A
        from jolt_atlas_amd import sharded
        from oracle import orc
        A.init(0)
        n = {n}
        L = orc.random_fr(1 << n, 11); R = orc.random_fr(1 << n, 12)
        t = A.Blake2bTranscript(b"sharded")
        proof, ch, fin, claim = sharded.prove_dot_sharded(dist, sharded.strided_shard(L, rank, world),
                                                          sharded.strided_shard(R, rank, world), t)
        t_o = orc.new_transcript(b"sharded")
        want_claim = orc.dot_claim(L, R)
        proof_o, ch_o, fin_o = orc.sumcheck_dot_prove(L, R, want_claim, t_o)
        assert np.array_equal(claim, want_claim[0])
        assert ch == ch_o and np.array_equal(proof, proof_o) and np.array_equal(fin, fin_o)
        assert t.state == t_o.state_bytes() and t.n_rounds == t_o.n_rounds
        # MSM split by point range
        m = 1 << 10
        tau = orc.random_fr(1, 5)[0]
        srs_full = orc.srs_powers(tau, m)
        sc = orc.random_fr(m, 6)
        lo, hi = rank * m // world, (rank + 1) * m // world
        srs = A.SRS.upload(srs_full[lo:hi])
        got = sharded.msm_sharded(dist, srs, sc[lo:hi])
        assert orc.g1_eq(got, orc.msm(srs_full, sc))
        dist.barrier()
        if rank == 0:
            print("SHARDED_OK", world)
        dist.destroy_process_group()
    """))
    port = 29620 + world + n
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert f"SHARDED_OK {world}" in out.stdout


@pytest.mark.parametrize("world,n", [(2, 14), (4, 16), (8, 13), (2, 3)])
def test_sharded_over_shared_memory_board(tmp_path, world, n):
    """atlas_sumcheck_prove_dot_sharded: one call per rank, round channel + shared-memory board, no collective; every rank's
    proof equals the oracle's proof of the whole instance.  Processes share the test box's GPU."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        rank, world = int(sys.argv[1]), {world}
        import jolt_atlas_amd as A
        from jolt_atlas_amd import sharded
        from oracle import orc
        A.init(0)
        n = {n}
        L = orc.random_fr(1 << n, 11); R = orc.random_fr(1 << n, 12)
        grp = sharded.ShardGroup(sys.argv[2], world, rank)
        t = A.Blake2bTranscript(b"sharded")
        proof, ch, fin, claim = sharded.prove_dot_sharded_shm(grp, sharded.strided_shard(L, rank, world), sharded.strided_shard(R, rank, world), t)
        t_o = orc.new_transcript(b"sharded")
        want_claim = orc.dot_claim(L, R)
        proof_o, ch_o, fin_o = orc.sumcheck_dot_prove(L, R, want_claim, t_o)
        assert np.array_equal(claim, want_claim[0])
        assert ch == ch_o and np.array_equal(proof, proof_o) and np.array_equal(fin, fin_o)
        assert t.state == t_o.state_bytes() and t.n_rounds == t_o.n_rounds
        m = 1 << 10
        tau = orc.random_fr(1, 5)[0]
        srs_full = orc.srs_powers(tau, m)
        sc = orc.random_fr(m, 6)
        lo, hi = rank * m // world, (rank + 1) * m // world
        srs = A.SRS.upload(srs_full[lo:hi])
        assert orc.g1_eq(sharded.msm_sharded_shm(grp, srs, sc[lo:hi]), orc.msm(srs_full, sc))
        grp.close()
        print("SHM_SHARDED_OK", rank)
    """))
    name = f"/atlas_gpu_{os.getpid()}_{world}_{n}"
    procs = [subprocess.Popen([sys.executable, str(script), str(r), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, e[-3000:]
        assert f"SHM_SHARDED_OK {r}" in o
