#!/usr/bin/env python3
"""bench.py — MD timesteps/s of the bead-spring + loop-extrusion hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).  A "step" is one
velocity-Verlet timestep of the whole system (pair lj/cut + bond fene + fix nve + fix langevin + the three
USER-LE fixes on their firing steps, reneighboring included), inputs resident in HBM when timing starts.

Default workload = BASELINE.json configs[3] at N=1 (the configuration the north-star target is quoted on):
a 1M-bead single chain at melt density (strong scaling over N GPUs as z-slabs), barrier beads every 200, `extrusion 1000` / `ex_load 1000 prob 0.01` /
`ex_unload 1000 prob 0.01` (the dense LE parameter set BASELINE.md measured the reference with), so that two
firings of every LE fix fall inside the default 2000-step timed window.

Extra objects on the JSON line:
  roofline     — the fused step kernel k_step: algorithmic bytes (144*N + 4*F, F = stored full-list entries;
                 DESIGN.md §3) / mean kernel duration from HIP events recorded on the launch
                 stream inside the engine over the timed region, vs the 8 TB/s HBM3E peak.
  cpu_baseline — the CPU oracle (oracle/le_oracle.c, a serial port of the reference path) on 1 host core, on a
                 bounded sample (first steps of the same system from the same state); reported, not the target.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (beads, chains, barrier_every, n1, nload, pload, tp)
    "chain1m": (1000000, 1, 200, 1000, 1000, 0.01, 0.5),
    "chains10x100k": (1000000, 10, 200, 1000, 1000, 0.01, 0.5),
    "chain100k": (100000, 1, 0, 17500, 7000, 0.001, 1.0),     # README.md:17,33-34 parameters
    "chain32k": (32000, 1, 0, 1000, 1000, 0.01, 1.0),
    "chain250k": (250000, 1, 200, 1000, 1000, 0.01, 0.5),
    "chain500k": (500000, 1, 200, 1000, 1000, 0.01, 0.5),
    "chain8m": (8000000, 1, 200, 1000, 1000, 0.01, 0.5),        # per-GPU size of the 8 x 1M weak-scaling config, on one GPU
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default="chain1m", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-steps", type=int, default=-1, help="oracle sample length (0 = skip the CPU baseline)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU fallback)")
    # rehearsal of the N > 1 path on a one-GPU box (every rank on device 0, file-mailbox transport instead of RCCL,
    # which refuses two ranks on one device): LAMMPS_LE_BENCH_SHM=1.  Never used for reported numbers.
    shm_rehearsal = world > 1 and os.environ.get("LAMMPS_LE_BENCH_SHM") == "1"
    if shm_rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if shm_rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    os.environ["LAMMPS_LE_KERNEL_TIMING"] = "1"
    from lammps_le_amd import lammps
    from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data

    nbeads, nchains, bar, n1, nload, pload, tp = WORKLOADS[args.workload]
    # N > 1: the SAME system is decomposed into N z-slabs (strong scaling); every rank builds the identical input
    sysd = lattice_chains(nbeads, nchains=nchains, seed=1, barrier_every=bar)
    ntypes = sysd["ntypes"]
    tmp = tempfile.mkdtemp(prefix="le_bench_")
    data = os.path.join(tmp, "data.r%d" % rank)
    write_data(data, sysd)
    left, right, lr = (2, 3, "4") if ntypes == 4 else (1, 1, "")
    script = CHAIN_INPUT.format(data=data, n1=n1, left=left, right=right, tp=tp, lr=lr, nload=nload, pload=pload)

    lmp = lammps(cmdargs=["-screen", "none"])
    if world > 1 and shm_rehearsal:
        lmp.comm_init("shm", rank, world, session="bench%s" % os.environ.get("MASTER_PORT", "0"))
    elif world > 1:
        from lammps_le_amd import init_from_torch_distributed
        init_from_torch_distributed(lmp)     # engine's own RCCL communicator (unique id broadcast by torch)
    for ln in script.split("\n"):
        lmp.command(ln)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed warm-up of W steps: upload, melt from the lattice, first LE firings.  The state the CPU baseline starts
    # from is copied out 50 steps before its end: the copy is ~0.1 s of host work during which the GPU idles and its
    # clocks drop (the first 100 steps after such a pause were measured 2x slower); the last 50 warm-up steps bring
    # them back before the timed region starts.
    tail = 50 if args.warmup >= 100 else 0
    lmp.command("run %d" % (args.warmup - tail))
    if world == 1:
        x_state, v_state = lmp.gather("x"), lmp.gather("v")
    if tail:
        lmp.command("run %d" % tail)
    barrier()
    t0 = time.perf_counter()
    lmp.command("run %d" % args.steps)             # `run` = Verlet::setup + K steps, synchronised at the end
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if shm_rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (k_force: pair lj/cut + bonds, one launch per step) ----
    kms = lmp.stat("pair_kernel_ms")
    full_entries = lmp.stat("neigh_pairs")          # stored full-list entries (all ranks) = 2 * half pairs
    nbonds = lmp.get_thermo("bonds")
    nlocal = lmp.stat("nlocal")
    # k_step moves, per owned bead: pos 32 r + pos' 32 w + v 24 r + 24 w + tag 4 + draws 12 + bond table 12 +
    # numneigh 4 = 144 B, plus 4 B per stored neighbor entry; this rank's share when decomposed
    alg_bytes = 144.0 * nlocal + 4.0 * full_entries * (nlocal / nbeads)
    achieved = alg_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload, {}).get("k_step_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_step (pair lj/cut + bond fene + langevin + nve final/initial, fused)",
                "achieved": round(achieved, 1),
                "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(kms, 5),
                "launches_timed": int(lmp.stat("pair_kernel_launches"))}

    # ---- CPU baseline: the oracle on a bounded sample, rank 0 at N=1 only ----
    cpu = None
    cpu_steps = args.cpu_steps
    if cpu_steps < 0:
        cpu_steps = max(10, int(4.0e6 * 15 / nbeads)) if nbeads >= 100000 else 2000
    if rank == 0 and world == 1 and cpu_steps > 0:
        from systems import OracleScript
        s2 = dict(sysd)
        s2["x"], s2["v"] = x_state, v_state          # the state the timed GPU window started from
        osc = OracleScript(s2)
        for ln in script.split("\n"):
            if ln.startswith("thermo_style"):
                continue
            osc.line(ln)
        tc = time.perf_counter()
        osc.o.run(cpu_steps)
        wall = time.perf_counter() - tc
        tm = osc.o.timers()
        cpu = {"value": round(cpu_steps / tm["total"], 3), "unit": "timesteps/s", "cores": 1, "kind": "port",
               "sample": "%d steps of the same %d-bead system from the state near the end of the warm-up (setup excluded, as the "
                         "reference's Loop time); wall incl. setup %.1f s" % (cpu_steps, nbeads, wall),
               "split_pct": {k: round(100 * tm[k] / tm["total"], 1) for k in ("pair", "bond", "neigh", "modify")}}

    if rank == 0:
        out = {
            "metric": "MD timesteps/sec, bead-spring LJ+FENE chain with loop extrusion",
            "value": round(args.steps / elapsed, 2),
            "unit": "timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d beads, %d chain(s), lj/cut 1.12 + fene + nve + langevin + extrusion %d / "
                                   "ex_load %d prob %g / ex_unload %d prob %g" % (args.workload, nbeads, nchains, n1, nload,
                                                                                 pload, nload, pload),
                       "beads_total": nbeads,
                       "parallelism": "1 GPU" if world == 1 else
                       "%d z-slabs, one rank per GPU, halo + migration over %s, replicated extruder table"
                       % (world, "a file mailbox on ONE shared GPU (rehearsal, not a result)" if shm_rehearsal else "RCCL")},
            "roofline": roofline, "cpu_baseline": cpu,
            # LJ units: the reference prints tau/day instead of ns/day (src/finish.cpp:124-145); timestep 0.005 tau
            "tau_per_day": round(args.steps / elapsed * 0.005 * 86400.0, 1),
            "engine_loop_time_s": round(lmp.stat("loop_time"), 5), "neigh_builds": int(lmp.stat("neigh_builds")),
            "extruders": int(nbonds - (nbeads - nchains)), "fene_warnings": int(lmp.stat("fene_warnings")),
        }
        print(json.dumps(out))
    lmp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
