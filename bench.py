#!/usr/bin/env python3
"""bench.py — MD timesteps/s of the bead-spring + loop-extrusion hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).  A "step" is one
velocity-Verlet timestep of the whole system (pair lj/cut + bond fene + fix nve + fix langevin + the three
USER-LE fixes on their firing steps, reneighboring included), inputs resident in HBM when timing starts.

Default workload `walk1m` = BASELINE.json configs[3] at N=1 (the configuration the north-star target is quoted on):
a 1M-bead single chain at melt density from the SCRAMBLED start (a random Hamiltonian path inside every 10^3-site
block: chain order is not memory order, as in a melt and as the reference's own generator tools/chain.f makes them;
`--workload chain1m` is the serpentine-lattice start, which flatters every tag-indexed gather), strong scaling over N
GPUs as z-slabs, barrier beads every 200, `extrusion 1000` / `ex_load 1000 prob 0.002` / `ex_unload 1000 prob 0.05`.
BASELINE.md timed the reference with the dense set 0.01 / 0.01 over 2-4k steps; that set ends in `Bad FENE bond` in
BOTH engines on long runs, so the line says `deviates_from_baseline` and `--workload chain1m_dense`
keeps the dense set for comparison.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (one child process
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, nothing touches HIP in the parent), relays rank 0's JSON
line and exits non-zero if any rank did; under `torch.distributed.run` it is one of the ranks as before.

Order of a run (whatever K and W are, so that a 20-step line and a 2000-step line measure the same thing):
  1. untimed PRE-ROLL to a state that has left the start lattice and carries extruders (default: to step 3010, i.e.
     past three firings of every LE fix) — `pre_roll_steps` on the JSON line;
  2. W untimed warm-up steps;
  3. barrier + synchronize, `run K`, barrier + synchronize.  `value` = K / Loop time of that run, the metric exactly
     as the reference prints it (src/finish.cpp:124-145: the timer starts after Verlet::setup, src/run.cpp:178-186),
     MAX over ranks; the wall clock around the whole `run K` (setup included) is reported as `wall_s` / `value_wall`;
  4. the three LE firings timed on their own (`le_firing`): the steps ...001-...010 of the next period (extrusion at
     ...001, ex_unload at ...002, ex_load at ...003, each followed by the reneighboring it forces) against ten
     ordinary steps, so a short window that holds no firing still reports what a firing costs;
  5. only now is the state copied out for the CPU baseline (the copy idles the GPU; nothing timed follows it).

Extra objects on the JSON line:
  roofline     — the fused step kernel k_step: algorithmic bytes (132*N + 4*(F + 2*B), F = stored full-list entries, B = bonds;
                 DESIGN.md §3) / mean kernel duration from HIP events recorded on the launch stream inside the
                 engine over the timed region (every launch when K <= 64, every 16th otherwise), vs the 8 TB/s peak.
  cpu_baseline — the CPU oracle (oracle/le_oracle.c, a serial port of the reference path) on 1 host core, on a
                 bounded sample of the same system from the state (positions, velocities, bond topology with its
                 extruders) the GPU run ended in; reported, not the target.
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

# LE parameter set of the default workloads: extrusion every 1000 steps, ex_load prob 0.002, ex_unload prob 0.05.
# Round 1 used the "dense" set 0.01 / 0.01 (the one BASELINE.md timed the reference with over 2-4k steps): it loads
# 1 % of ~0.35 N candidate pairs per firing and unloads 1 % of the extruders, i.e. it heads for extruders on a third of
# the beads; loops then grow faster than the melt relaxes, extruder bonds over-stretch and BOTH engines end in the
# reference's `Bad FENE bond` (oracle at 100k beads: 200 warnings by step 100k; product at 1M: abort before step 80k;
# tests/soak_onset.py replays the first warning of a 250k-bead product run in the oracle: same bond, same mechanism).
# The set below saturates near 0.4 % loaded beads and was run for 300 000 steps at 1M beads (2 warnings, no abort) and
# 120 000 steps in the oracle at 100k beads (0 warnings): tests/soak_1m.py, tests/soak_compare.py.
WORKLOADS = {
    # name: (beads, chains, barrier_every, n1, nload, pload, tp, generator, punload)
    "chain1m": (1000000, 1, 200, 1000, 1000, 0.002, 0.5, "lattice", 0.05),
    "chains10x100k": (1000000, 10, 200, 1000, 1000, 0.002, 0.5, "lattice", 0.05),
    "chain100k": (100000, 1, 0, 17500, 7000, 0.001, 1.0, "lattice", 0.001),     # README.md:17,33-34 parameters
    "chain32k": (32000, 1, 0, 1000, 1000, 0.002, 1.0, "lattice", 0.05),
    "chain250k": (250000, 1, 200, 1000, 1000, 0.002, 0.5, "lattice", 0.05),
    "chain500k": (500000, 1, 200, 1000, 1000, 0.002, 0.5, "lattice", 0.05),
    # BASELINE configs[4] names its load: dense (prob 0.01, N1 = 1000); the per-GPU size of the 8 x 1M weak-scaling run
    "chain8m": (8000000, 1, 200, 1000, 1000, 0.01, 0.5, "lattice", 0.01),
    "chain1m_dense": (1000000, 1, 200, 1000, 1000, 0.01, 0.5, "lattice", 0.01),  # round-1 default, for comparison
    # scrambled start (lammps_le_amd.synth.scrambled_chains): chain order is NOT memory order, which is what a melt looks
    # like to every tag-indexed gather (the serpentine lattice start flatters them)
    "walk1m": (1000000, 1, 200, 1000, 1000, 0.002, 0.5, "walk", 0.05),
    "walk100k": (100000, 1, 200, 1000, 1000, 0.002, 0.5, "walk", 0.05),
    "walk8m": (8000000, 1, 200, 1000, 1000, 0.01, 0.5, "walk", 0.01),        # chain8m from the scrambled start (the one that survives long runs)
    # semiflexible chain: walk1m + angle_style cosine on every backbone triple, ex_load ... atype 2 (the angle
    # kernel writes its forces right before the fused step kernel, which adds them: `roofline` is that variant of k_step)
    "walk1m_angles": (1000000, 1, 200, 1000, 1000, 0.002, 0.5, "walk", 0.05),
    # anchored beads: walk1m with fix nve / fix langevin on `group mobile type 1` - the barrier beads (every 200th) neither move
    # nor are thermostatted; the step kernel's group variant (`roofline` is that variant; LAMMPS_LE_NO_FUSED_GROUPS=1: unfused)
    "walk1m_anchors": (1000000, 1, 200, 1000, 1000, 0.002, 0.5, "walk", 0.05),
}


def bonds_by_tag(lmp):
    """(type, lo, hi) rows of every stored bond, once, in ascending (lo, slot) order (vectorised bond_set)."""
    nb, bt, ba = lmp.gather("num_bond"), lmp.gather("bond_type"), lmp.gather("bond_atom")
    n, w = bt.shape
    own = np.repeat(np.arange(1, n + 1, dtype=np.int64)[:, None], w, axis=1)
    live = np.arange(w)[None, :] < nb[:, None]
    keep = live & (own < ba)                      # newton_bond off: both ends store the bond, keep the lower end's copy
    return np.stack([bt[keep], own[keep], ba[keep]], axis=1).astype(np.int32)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: be the launcher.  N children, one per GPU, each a fresh process (the
    parent never imports torch or touches HIP, and no process that has is ever re-exec'ed); rank 0 inherits stdout, so its
    one JSON line is this command's output.  A rank that fails takes the others down (bounded: the engine's own
    communicator time-out ends them, the parent only kills what is still there after a grace period).  Exit code =
    first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, t_fail = 0, None
    live = list(range(n))
    while live:
        time.sleep(0.2)
        for r in list(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.remove(r)
            if c != 0 and rc == 0:
                rc, t_fail = c, time.time()
                print("bench.py: rank %d exited with code %d" % (r, c), file=sys.stderr)
        if t_fail is not None and live and time.time() - t_fail > 60.0:
            for r in live:
                procs[r].kill()           # exactly the children started above
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default="walk1m", choices=sorted(WORKLOADS))
    ap.add_argument("--pre-roll", type=int, default=-1,
                    help="untimed steps before the warm-up (default: past three firings of every LE fix)")
    ap.add_argument("--cpu-steps", type=int, default=-1, help="oracle sample length (0 = skip the CPU baseline)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    # stdout carries exactly ONE line, the JSON: everything else a library may print there (gloo's connection banner, RCCL
    # notices) goes to stderr - fd 1 is pointed at fd 2 and the result is written to a private copy of the real stdout
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)"
                         % (args.gpus, world, args.gpus))
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

    nbeads, nchains, bar, n1, nload, pload, tp, gen = WORKLOADS[args.workload][:8]
    punload = WORKLOADS[args.workload][8] if len(WORKLOADS[args.workload]) > 8 else pload
    # N > 1: the SAME system is decomposed into N z-slabs (strong scaling); every rank builds the identical input
    if gen == "walk":
        from lammps_le_amd.synth import scrambled_chains
        sysd = scrambled_chains(nbeads, nchains=nchains, seed=1, barrier_every=bar)
    else:
        sysd = lattice_chains(nbeads, nchains=nchains, seed=1, barrier_every=bar)
    angles = args.workload.endswith("_angles")
    if angles:
        from lammps_le_amd.synth import add_backbone_angles
        add_backbone_angles(sysd, nchains=nchains)
    ntypes = sysd["ntypes"]
    tmp = tempfile.mkdtemp(prefix="le_bench_")
    data = os.path.join(tmp, "data.r%d" % rank)
    write_data(data, sysd)
    left, right, lr = (2, 3, "4") if ntypes == 4 else (1, 1, "")
    script = CHAIN_INPUT.format(data=data, n1=n1, left=left, right=right, tp=tp, lr=lr, nload=nload, pload=pload, punload=punload)
    if angles:
        script = script.replace("atom_style bond", "atom_style molecular") \
            .replace("pair_style lj/cut", "angle_style cosine\nangle_coeff * 2.0\npair_style lj/cut") \
            .replace("iparam 1 1 jparam 1 1", "iparam 1 1 jparam 1 1 atype 2")

    if args.workload.endswith("_anchors"):
        script = script.replace("fix 1 all nve", "group mobile type 1\nfix 1 mobile nve").replace("fix 2 all langevin", "fix 2 mobile langevin")

    lmp = lammps(cmdargs=["-screen", "none"])
    rccl_nranks = 1
    if world > 1 and shm_rehearsal:
        lmp.comm_init("shm", rank, world, session="bench%s" % os.environ.get("MASTER_PORT", "0"))
    elif world > 1:
        from lammps_le_amd import init_from_torch_distributed
        init_from_torch_distributed(lmp)     # engine's own RCCL communicator (unique id broadcast by torch)
        # self-check of the first real multi-GPU run: the engine's communicator must span exactly N ranks
        rccl_nranks = int(lmp.stat("comm_nranks"))
        if rccl_nranks != world:
            print("bench.py: engine communicator has %d ranks, expected %d" % (rccl_nranks, world), file=sys.stderr)
            sys.exit(3)
    for ln in script.split("\n"):
        lmp.command(ln)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(val):
        if world == 1:
            return float(val)
        t = torch.tensor([val], dtype=torch.float64, device="cpu" if shm_rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    period = max(n1, nload)
    # N > 1: the per-step halo can go through peer windows (the step kernel stores into the neighbour's IPC-mapped buffer,
    # kernels_dd.hip) instead of a grouped RCCL exchange.  That path has only been validated between processes on ONE
    # GPU, so the pre-roll runs it in verify mode - every window halo is also sent through RCCL and compared, RCCL's copy
    # wins - and the timed run uses the windows only if every rank saw them used and not one value differ.
    halo_mode = "rccl"
    try_windows = world > 1 and os.environ.get("LAMMPS_LE_FAST_HALO", "1") != "0"
    if try_windows:
        os.environ["LAMMPS_LE_FAST_HALO"] = "1"
        os.environ["LAMMPS_LE_FAST_HALO_VERIFY"] = "1"
    # ---- 1. pre-roll (untimed): off the start lattice, extruders on the chain ----
    pre = args.pre_roll if args.pre_roll >= 0 else 3 * period + 10
    if try_windows:
        pre = max(pre, 200)
    if pre > 0:
        lmp.command("run %d" % pre)
    if try_windows:
        used = -max_over_ranks(-lmp.stat("halo_window_exchanges"))          # min over ranks
        bad = max_over_ranks(lmp.stat("halo_window_mismatches"))
        os.environ["LAMMPS_LE_FAST_HALO_VERIFY"] = "0"
        if used > 0 and bad == 0:
            halo_mode = "peer windows, %s (verified against the transport over %d exchanges of the pre-roll)" % (
                "one launch per exchange" if lmp.stat("halo_fused") else "counter kernel + copy kernel (a neighbour shares this GPU)",
                int(used))
        else:
            os.environ["LAMMPS_LE_FAST_HALO"] = "0"
            halo_mode = "rccl (peer windows %s)" % ("not available" if used <= 0 else "REJECTED: %d values differed" % int(bad))
    if shm_rehearsal:
        halo_mode = halo_mode.replace("rccl", "file mailbox")
    # ---- 2. warm-up (untimed) ----
    if args.warmup > 0:
        lmp.command("run %d" % args.warmup)
    # ---- 3. the timed region ----
    barrier()
    t0 = time.perf_counter()
    lmp.command("run %d" % args.steps)             # `run` = Verlet::setup + K steps, synchronised at the end
    barrier()
    wall = max_over_ranks(time.perf_counter() - t0)
    my_loop = lmp.stat("loop_time")
    loop = max_over_ranks(my_loop)                 # the reference's Loop time: K steps, setup excluded
    step_first = int(lmp.get_thermo("step")) - args.steps + 1
    kms = lmp.stat("pair_kernel_ms")
    klaunches = int(lmp.stat("pair_kernel_launches"))
    builds = int(lmp.stat("neigh_builds"))
    full_entries = lmp.stat("neigh_pairs")          # stored full-list entries (all ranks) = 2 * half pairs
    nbonds = lmp.get_thermo("bonds")
    nlocal = lmp.stat("nlocal")
    sections = {k: round(lmp.stat("time_" + k), 6) for k in ("pair", "neigh", "comm", "modify", "output", "other")}

    # ---- 4. what a firing of the three LE fixes costs (untimed for `value`) ----
    le_firing = None
    if period >= 100:
        now = int(lmp.get_thermo("step"))
        to_boundary = (period - now % period) % period
        if to_boundary:
            lmp.command("run %d" % to_boundary)
        barrier()
        lmp.command("run 10")                       # steps ...001 (extrusion), ...002 (ex_unload), ...003 (ex_load) + 7
        fire_loop = max_over_ranks(lmp.stat("loop_time"))
        fire_builds = int(lmp.stat("neigh_builds"))
        lmp.command("run 30")                       # settle back to the ordinary rhythm
        lmp.command("run 10")
        plain_loop = max_over_ranks(lmp.stat("loop_time"))
        plain_builds = int(lmp.stat("neigh_builds"))
        extra_ms = 1e3 * (fire_loop - plain_loop)
        le_firing = {"ms_per_period": round(extra_ms, 4), "period_steps": period,
                     "ten_firing_steps_ms": round(1e3 * fire_loop, 4), "ten_plain_steps_ms": round(1e3 * plain_loop, 4),
                     "neigh_builds": [fire_builds, plain_builds],
                     "share_of_step_time": round(extra_ms / period / (1e3 * loop / args.steps), 5) if loop > 0 else None}
    extruders = int(lmp.get_thermo("bonds") - (nbeads - nchains))

    # ---- roofline of the dominant kernel (k_step: pair lj/cut + bonds + langevin + nve, one launch per step) ----
    # k_step moves, per owned bead: pos 32 r + pos' 32 w + v 24 r + 24 w + tag 4 + draws 12 + numneigh 4 = 132 B, plus
    # 4 B per list entry: the stored neighbor entries and the bead's bonds, which open its list (one entry per bond end:
    # 2 per bond) - DESIGN.md §3; this rank's share when decomposed
    alg_bytes = 132.0 * nlocal + 4.0 * (full_entries + 2.0 * nbonds) * (nlocal / nbeads)
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
                "launches_timed": klaunches}

    # ---- 5. CPU baseline: the oracle on a bounded sample, rank 0 at N=1 only ----
    cpu = None
    cpu_steps = args.cpu_steps
    if cpu_steps < 0:
        cpu_steps = max(10, int(4.0e6 * 15 / nbeads)) if nbeads >= 100000 else 2000
    if rank == 0 and world == 1 and cpu_steps > 0:
        from systems import OracleScript
        s2 = dict(sysd)
        s2["x"], s2["v"], s2["image"] = lmp.gather("x"), lmp.gather("v"), lmp.gather("image")
        s2["type"] = lmp.gather("type").astype(np.int32)
        s2["bonds"] = bonds_by_tag(lmp)              # backbone + the extruders the GPU run has loaded
        osc = OracleScript(s2)
        for ln in script.split("\n"):
            if ln.startswith("thermo_style"):
                continue
            osc.line(ln)
        # same phase of the LE firing periods as the GPU run (a firing right after a firing moves extruder bonds that
        # have not relaxed yet: an oracle restarted at step 0 aborts with `Bad FENE bond` at 1M beads)
        osc.line("reset_timestep %d" % int(lmp.get_thermo("step")))
        tc = time.perf_counter()
        osc.o.run(cpu_steps)
        cwall = time.perf_counter() - tc
        tm = osc.o.timers()
        cpu = {"value": round(cpu_steps / tm["total"], 3), "unit": "timesteps/s", "cores": 1, "kind": "port",
               "sample": "%d steps of the same %d-bead system from the state the GPU run ended in (x, v, bond topology "
                         "with %d extruders; setup excluded, as the reference's Loop time); wall incl. setup %.1f s"
                         % (cpu_steps, nbeads, extruders, cwall),
               "split_pct": {k: round(100 * tm[k] / tm["total"], 1) for k in ("pair", "bond", "neigh", "modify")}}

    # one record per rank: what it owned and how long its own loop and step kernel took (the line's value is the max)
    mine = {"rank": rank, "owned_beads": int(nlocal), "ghost_beads": int(lmp.stat("nghost")),
            "us_per_step": round(1e6 * my_loop / args.steps, 3), "k_step_us": round(1e3 * kms, 3),
            "halo": halo_mode if world > 1 else None, "device": torch.cuda.current_device()}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    dense = (pload == 0.01 and punload == 0.01)

    if rank == 0:
        out = {
            "metric": "MD timesteps/sec, bead-spring LJ+FENE chain with loop extrusion",
            "value": round(args.steps / loop, 2),
            "unit": "timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * loop / args.steps, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d beads, %d chain(s), %s start, lj/cut 1.12 + fene + nve + langevin + extrusion %d / "
                                   "ex_load %d prob %g / ex_unload %d prob %g" % (args.workload, nbeads, nchains, gen, n1, nload,
                                                                                 pload, nload, punload),
                       "beads_total": nbeads,
                       "parallelism": "1 GPU" if world == 1 else
                       "%d z-slabs, one rank per GPU, halo + migration over %s, replicated extruder table"
                       % (world, "a file mailbox on ONE shared GPU (rehearsal, not a result)" if shm_rehearsal else "RCCL")},
            "roofline": roofline, "cpu_baseline": cpu,
            # LJ units: the reference prints tau/day instead of ns/day (src/finish.cpp:124-145); timestep 0.005 tau
            "tau_per_day": round(args.steps / loop * 0.005 * 86400.0, 1),
            "timing": "value = steps / Loop time (setup excluded, src/finish.cpp:124-145), max over ranks",
            "engine_loop_time_s": round(loop, 6), "wall_s": round(wall, 6), "value_wall": round(args.steps / wall, 2),
            "pre_roll_steps": pre, "timed_steps": [step_first, step_first + args.steps - 1],
            "neigh_builds": builds, "extruders": extruders, "le_firing": le_firing,
            "loop_sections_s": sections, "rccl_nranks": rccl_nranks, "halo": halo_mode if world > 1 else None,
            "fene_warnings": int(lmp.stat("fene_warnings")),
            "per_rank": per_rank,
            # BASELINE.md timed the reference with ex_load 0.01 / ex_unload 0.01 over 2-4k steps from the lattice start
            "deviates_from_baseline": None if (dense and gen == "lattice") else
            ", ".join(x for x in ("" if dense else "LE probabilities %g / %g instead of 0.01 / 0.01 (the dense set aborts with "
                                  "`Bad FENE bond` on long runs in both engines, DESIGN.md 4)" % (pload, punload),
                                  "" if gen == "lattice" else "scrambled (melt-like) start instead of the serpentine lattice") if x),
        }
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    lmp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
