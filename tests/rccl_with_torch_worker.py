"""The engine's own RCCL communicator next to a live torch.distributed NCCL(=RCCL) process group in ONE process, as in
`bench.py --gpus N` (test helper of test_gpu_dd.py; one rank, one GPU).  Prints which RCCL files the process mapped."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from lammps_le_amd import library_path, use_torch_rccl

torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[1], rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
if sys.argv[2] == "shared":
    use_torch_rccl()
lib = ctypes.CDLL(library_path())
rc = lib.lammps_le_rccl_selftest()
dist.all_reduce(t)              # torch's communicator still works afterwards
torch.cuda.synchronize()
maps = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
print("RCCL_FILES", len(maps), " ".join(maps))
dist.destroy_process_group()
sys.exit(rc)
