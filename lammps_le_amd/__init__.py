"""lammps_le_amd — MI355X-native engine for the bead-spring + loop-extrusion hot path of
polly-code/lammps_le, behind the reference's own interfaces.

`lammps` mirrors the subset of the reference's ctypes wrapper (python/lammps.py) that drives this
path: same method names and argument meaning, on top of the C-ABI in include/lammps_le.h
(liblammps_le.so, hand-written HIP for gfx950).  There is NO CPU fallback: `run` fails loudly
without a HIP device, and importing fails loudly if the shared library has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblammps_le.so")
if os.environ.get("LAMMPS_LE_LIBRARY"):      # development only: A/B runs of two builds of the engine (scripts/run_ab.sh)
    _SO = os.path.abspath(os.environ["LAMMPS_LE_LIBRARY"])

__all__ = ["lammps", "LammpsError", "library_path", "comm_unique_id", "init_from_torch_distributed"]


class LammpsError(Exception):
    pass


def library_path():
    return _SO


def _load():
    if not os.path.exists(_SO):
        raise ImportError(
            "lammps_le_amd: %s is missing — build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C lammps_le_amd/csrc)" % _SO)
    lib = C.CDLL(_SO, mode=C.RTLD_GLOBAL)
    lib.lammps_open_no_mpi.restype = C.c_void_p
    lib.lammps_open_no_mpi.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p]
    lib.lammps_close.argtypes = [C.c_void_p]
    lib.lammps_file.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_command.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_command.restype = C.c_char_p
    lib.lammps_commands_string.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_get_natoms.argtypes = [C.c_void_p]
    lib.lammps_get_natoms.restype = C.c_double
    lib.lammps_get_thermo.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_get_thermo.restype = C.c_double
    lib.lammps_extract_setting.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_extract_setting.restype = C.c_int
    lib.lammps_extract_global.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_extract_global.restype = C.c_void_p
    lib.lammps_extract_atom.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_extract_atom.restype = C.c_void_p
    lib.lammps_extract_fix.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.lammps_extract_fix.restype = C.c_void_p
    lib.lammps_gather_atoms.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    lib.lammps_scatter_atoms.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    lib.lammps_extract_box.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    lib.lammps_free.argtypes = [C.c_void_p]
    lib.lammps_has_error.argtypes = [C.c_void_p]
    lib.lammps_get_last_error_message.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.lammps_has_style.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.lammps_le_stat.argtypes = [C.c_void_p, C.c_char_p]
    lib.lammps_le_stat.restype = C.c_double
    return lib


class lammps(object):
    """Counterpart of python/lammps.py `lammps` for this path (1 rank per process, one GPU per rank)."""

    def __init__(self, name="", cmdargs=None, ptr=None, comm=None):
        self.lib = _load()
        args = ["lammps"] + list(cmdargs or [])
        argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
        self.lmp = C.c_void_p(self.lib.lammps_open_no_mpi(len(args), argv, None))
        if not self.lmp:
            raise LammpsError("could not create engine instance")
        self.opened = 1

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "opened", 0):
            self.lib.lammps_close(self.lmp)
            self.opened = 0
            self.lmp = None

    # -- error plumbing: python/lammps.py raises on lammps_has_error when built with exceptions --
    def _check(self):
        if self.lib.lammps_has_error(self.lmp):
            buf = C.create_string_buffer(512)
            self.lib.lammps_get_last_error_message(self.lmp, buf, 512)
            raise LammpsError(buf.value.decode())

    def version(self):
        return self.lib.lammps_version(self.lmp)

    def file(self, path):
        self.lib.lammps_file(self.lmp, path.encode())
        self._check()

    def command(self, cmd):
        self.lib.lammps_command(self.lmp, cmd.encode())
        self._check()

    def commands_list(self, cmdlist):
        for c in cmdlist:
            self.command(c)

    def commands_string(self, multicmd):
        self.lib.lammps_commands_string(self.lmp, multicmd.encode())
        self._check()

    def get_natoms(self):
        return int(self.lib.lammps_get_natoms(self.lmp))

    def get_thermo(self, name):
        v = self.lib.lammps_get_thermo(self.lmp, name.encode())
        self._check()
        return v

    def extract_setting(self, name):
        return self.lib.lammps_extract_setting(self.lmp, name.encode())

    def extract_box(self):
        lo = (C.c_double * 3)()
        hi = (C.c_double * 3)()
        xy, yz, xz = C.c_double(), C.c_double(), C.c_double()
        pf = (C.c_int * 3)()
        bf = C.c_int()
        self.lib.lammps_extract_box(self.lmp, lo, hi, C.byref(xy), C.byref(yz), C.byref(xz), pf, C.byref(bf))
        return list(lo), list(hi), xy.value, yz.value, xz.value, list(pf), bf.value

    def extract_fix(self, fid, style, type, nrow=0, ncol=0):
        p = self.lib.lammps_extract_fix(self.lmp, fid.encode(), style, type, nrow, ncol)
        self._check()
        if not p:
            return None
        val = C.cast(p, C.POINTER(C.c_double))[0]
        self.lib.lammps_free(p)
        return val

    def gather_atoms(self, name, type, count):
        """As python/lammps.py: returns a ctypes array of natoms*count values ordered by atom ID."""
        n = self.get_natoms()
        data = ((C.c_double if type == 1 else C.c_int) * (n * count))()
        self.lib.lammps_gather_atoms(self.lmp, name.encode(), type, count, data)
        self._check()
        return data

    def scatter_atoms(self, name, type, count, data):
        self.lib.lammps_scatter_atoms(self.lmp, name.encode(), type, count, data)
        self._check()

    # -- numpy conveniences (python/lammps.py has the same idea in lammps.numpy) --
    def gather(self, name):
        ints = {"type": 1, "id": 1, "mask": 1, "molecule": 1, "image": 3, "num_bond": 1, "nspecial": 3, "num_angle": 1}
        n = self.get_natoms()
        if name in ("x", "v", "f"):
            return np.ctypeslib.as_array(self.gather_atoms(name, 1, 3)).reshape(n, 3).copy()
        if name in ("bond_type", "bond_atom"):
            w = self.extract_setting("bond_per_atom")
        elif name == "special":
            w = self.extract_setting("maxspecial")
        elif name in ("angle_type", "angle_atom1", "angle_atom2", "angle_atom3"):
            w = self.extract_setting("angle_per_atom")
        else:
            w = ints[name]
        a = np.ctypeslib.as_array(self.gather_atoms(name, 0, w)).reshape(n, w).copy()
        return a[:, 0] if w == 1 else a

    def scatter(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64 if name in ("x", "v", "f") else np.int32)
        self.scatter_atoms(name, 1 if arr.dtype == np.float64 else 0, arr.shape[1] if arr.ndim > 1 else 1,
                           arr.ctypes.data_as(C.c_void_p))

    def bond_set(self):
        """Set of (type, lo_tag, hi_tag) over all stored bonds."""
        nb, bt, ba = self.gather("num_bond"), self.gather("bond_type"), self.gather("bond_atom")
        out = set()
        for i in np.nonzero(nb)[0]:
            for m in range(nb[i]):
                a, b = int(i) + 1, int(ba[i, m])
                out.add((int(bt[i, m]), min(a, b), max(a, b)))
        return out

    def angle_set(self):
        """{(type, a1, a2, a3)} of the copies the CENTRAL atoms store (one per angle), ends ordered."""
        na, at = self.gather("num_angle"), self.gather("angle_type")
        a1, a2, a3 = self.gather("angle_atom1"), self.gather("angle_atom2"), self.gather("angle_atom3")
        if at.ndim == 1:
            at, a1, a2, a3 = (v.reshape(-1, 1) for v in (at, a1, a2, a3))
        out = set()
        for i in np.nonzero(na)[0]:
            for m in range(na[i]):
                if a2[i, m] == i + 1:
                    out.add((int(at[i, m]), int(min(a1[i, m], a3[i, m])), int(a2[i, m]), int(max(a1[i, m], a3[i, m]))))
        return out

    def has_style(self, category, name):
        return self.lib.lammps_has_style(self.lmp, category.encode(), name.encode()) != 0

    def stat(self, name):
        return self.lib.lammps_le_stat(self.lmp, name.encode())

    def thermo_history(self):
        """Every thermo line printed so far: array of rows (step, temp, epair, emol, etotal, press, bonds)."""
        import numpy as np
        self.lib.lammps_le_thermo_log.restype = C.c_int
        self.lib.lammps_le_thermo_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        n = self.lib.lammps_le_thermo_log(self.lmp, None, 0)
        out = np.zeros((n, 7))
        if n:
            self.lib.lammps_le_thermo_log(self.lmp, out.ctypes.data_as(C.c_void_p), n)
        return out

    # -- ranks: one process per GPU, z-slab spatial decomposition (engine extension) --
    def comm_init(self, backend, rank, world, unique_id=b"", session="default"):
        """Join a group of `world` engine instances.  backend "rccl": unique_id = the 128-byte ncclUniqueId
        created on rank 0 by `comm_unique_id()` and broadcast by the launcher; backend "shm": file mailbox under
        /dev/shm (test transport, one process per rank); backend "local": the ranks are instances driven by threads
        of this process (development transport).  Must be called before the first run; every rank then issues the same commands."""
        self.lib.lammps_le_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        buf = C.create_string_buffer(bytes(unique_id).ljust(128, b"\0"), 128)
        self.lib.lammps_le_comm_init(self.lmp, backend.encode(), rank, world, buf, session.encode())
        self._check()


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the launcher broadcasts it)."""
    lib = _load()
    buf = C.create_string_buffer(128)
    if lib.lammps_le_comm_unique_id(buf) != 0:
        raise LammpsError("could not create an RCCL unique id")
    return buf.raw


def use_torch_rccl():
    """Point the engine at the RCCL copy PyTorch has already loaded (torch/lib/librccl.so), so that the process runs one
    RCCL instance.  No effect when LAMMPS_LE_RCCL_LIB is set by the user or torch ships no copy."""
    if os.environ.get("LAMMPS_LE_RCCL_LIB"):
        return
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(cand):
        os.environ["LAMMPS_LE_RCCL_LIB"] = cand


def init_from_torch_distributed(lmp):
    """Give every rank of an initialised torch.distributed group the same RCCL unique id and join the engine's
    own communicator (RANK / WORLD_SIZE from the process group).  Used by bench.py."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        return
    use_torch_rccl()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        t = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).to(dev)
    dist.broadcast(t, src=0)
    lmp.comm_init("rccl", rank, world, bytes(t.cpu().numpy().tobytes()))
