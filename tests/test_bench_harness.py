"""CPU: bench.py's timing contract on a 2-process gloo group (the N>1 path): barrier on
both sides, max over ranks, whole-job aggregation; plus the analytic work counters."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_field_op_counts():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.field_ops(1) == 14 and bench.field_ops(22) == 14 * ((1 << 22) - 1)


def test_timed_steps_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time, json
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        import bench
        dist.init_process_group("gloo")
        rank = dist.get_rank()
        calls = []
        def step(i):
            calls.append(i); time.sleep(0.01 * (1 + rank))      # rank 1 is the slow one
        def allreduce_max(x):
            t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); return float(t.item())
        dt = bench.timed_steps(step, 5, 2, lambda: None, dist.barrier, allreduce_max)
        assert calls == list(range(7)), calls
        assert dt >= 5 * 0.02 * 0.9, dt                          # max over ranks: the slow rank's time
        if rank == 0:
            print(json.dumps({{"dt": dt, "world": dist.get_world_size()}}))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert '"world": 2' in out.stdout
