#!/usr/bin/env python3
"""What one small exchange between the ranks of a sharded proof costs, three ways (DESIGN §6 "the collective"):

  board   the all-gather of the POSIX shared-memory board (csrc/shard_group.hpp) in C++ (tools/exp_board.cpp: 2, 4 and 8 forked
          processes, 64-byte records, no Python between the calls)
  gloo    torch.distributed.all_gather of the same record, gloo backend, 2 processes (host sockets)
  rccl    torch.distributed.all_gather of the same record on the device through the nccl (= RCCL) backend with ONE rank: a
          one-GPU box cannot hold two RCCL ranks (duplicate GPU), so this is the FLOOR of an RCCL collective — its kernel launch
          and the stream synchronisation the host transcript needs before it can read the result — without any link traffic

usage (GPU box): python tools/exp_collective.py          prints one JSON line"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 20000


def board_worker(name, world, rank):
    import numpy as np
    from jolt_atlas_amd import sharded
    grp = sharded.ShardGroup(name, world, rank)
    rec = np.arange(8, dtype=np.uint64) + rank
    for _ in range(1000):
        grp.allgather(rec)
    t0 = time.perf_counter()
    for _ in range(N):
        grp.allgather(rec)
    dt = (time.perf_counter() - t0) / N
    grp.close()
    if rank == 0:
        print(json.dumps({"us": dt * 1e6}))


def gloo_worker(port, world, rank):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    t = torch.arange(8, dtype=torch.int64)
    out = [torch.empty_like(t) for _ in range(world)]
    for _ in range(200):
        dist.all_gather(out, t)
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        dist.all_gather(out, t)
    dt = (time.perf_counter() - t0) / n
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"us": dt * 1e6}))


def rccl_floor(port):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    t = torch.arange(8, dtype=torch.int64, device="cuda")
    out = [torch.empty_like(t)]
    for _ in range(200):
        dist.all_gather(out, t); torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        dist.all_gather(out, t)
        torch.cuda.synchronize()                 # the host transcript reads the gathered sums: it has to wait for the stream
    dt = (time.perf_counter() - t0) / n
    host = torch.empty(8, dtype=torch.int64).pin_memory()
    t0 = time.perf_counter()
    for _ in range(n):
        dist.all_gather(out, t)
        host.copy_(out[0], non_blocking=True)
        torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / n
    dist.destroy_process_group()
    print(json.dumps({"us": dt * 1e6, "us_with_copy_to_host": dt2 * 1e6}))


def spawn(mode, world, extra):
    procs = [subprocess.Popen([sys.executable, __file__, mode, extra, str(world), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    if any(p.returncode for p in procs):
        return {"error": outs[0][1][-300:]}
    lines = [l for l in outs[0][0].splitlines() if l.strip().startswith("{")]        # the runtime may print lines of its own
    return json.loads(lines[-1]) if lines else {"error": "no result line", "stdout": outs[0][0][-300:], "stderr": outs[0][1][-300:]}


if __name__ == "__main__":
    if len(sys.argv) > 1:
        mode, extra, world, rank = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        if mode == "board":
            board_worker(extra, world, rank)
        elif mode == "gloo":
            gloo_worker(int(extra), world, rank)
        elif mode == "rccl":
            rccl_floor(int(extra))
        sys.exit(0)
    res = {"record_bytes": 64}
    exe = f"/tmp/exp_board_{os.getpid()}"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", ROOT, os.path.join(ROOT, "tools", "exp_board.cpp"), "-o", exe, "-lrt", "-lpthread"], check=True)
    for line in subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout.splitlines():
        w, us = line.split(":")[0].split()[1], float(line.split(":")[1].split()[0])
        res[f"board_world{w}"] = {"us": us}
    os.unlink(exe)
    res["gloo_world2"] = spawn("gloo", 2, str(29600 + os.getpid() % 300))
    res["rccl_world1_floor"] = spawn("rccl", 1, str(29950 + os.getpid() % 40))
    print(json.dumps(res))
