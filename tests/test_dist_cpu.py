"""CPU (gloo) coverage of the N>1 plumbing: torch.distributed process group -> engine communicator, and the
engine's inter-rank transport primitives (ring exchange, all-gather, reductions) between 2 real processes."""
import os
import subprocess
import sys
import textwrap
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import ctypes, os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from lammps_le_amd import library_path, lammps
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["MASTER_PORT"],
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    # the launcher-side broadcast used for the RCCL unique id (bytes travel through the process group)
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.arange(128, dtype=torch.uint8)
    dist.broadcast(t, src=0)
    assert t.tolist() == list(range(128))
    # the session name for the test transport is agreed on the same way
    sess = [os.environ["LE_SESSION"]] if rank == 0 else [None]
    dist.broadcast_object_list(sess, src=0)
    lib = ctypes.CDLL(library_path())
    rc = lib.lammps_le_comm_selftest(sess[0].encode(), rank, world)
    assert rc == 0, rc
    # an engine instance joins the group; without a GPU the slab run itself must fail loudly, not fall back
    lmp = lammps(cmdargs=["-screen", "none"])
    try:
        lmp.comm_init("shm", rank, world, session=sess[0] + "b")
        ok = torch.cuda.is_available()
    except Exception as e:
        ok = "No HIP device" in str(e)
    assert ok
    dist.barrier()
    print("rank", rank, "ok")
""") % ROOT


def test_two_rank_gloo_plumbing(tmp_path):
    port = str(29500 + (os.getpid() % 2000))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   LE_SESSION=uuid.uuid4().hex[:10])
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("ok" in o for o in outs)


def test_bench_self_launch_relays_failure():
    """`python bench.py --gpus 2` with no launcher starts its own ranks.  Without a GPU every rank stops
    with the loud no-device message; the launcher must come back with a non-zero code instead of hanging or hiding it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                        "--workload", "chain32k"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    import torch
    if torch.cuda.is_available():
        return                      # on a GPU box the launch is covered by tests/test_gpu_dd.py
    assert p.returncode != 0
    assert b"needs a HIP device" in p.stderr and b"rank" in p.stderr
    assert p.stdout.strip() == b""
