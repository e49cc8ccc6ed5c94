#!/bin/bash
# The engine's host code (script parser, groups, velocity, data / restart files, C API) under AddressSanitizer + UBSan:
# builds lammps_le_amd/csrc/asan/liblammps_le_asan.so (host sources instrumented, device code as always) and runs the CPU
# suite against it.  No GPU needed; GPU sanitizers are not available on the pool.
set -e
cd "$(dirname "$0")/.."
make -s -C lammps_le_amd/csrc asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  LAMMPS_LE_LIBRARY=$PWD/lammps_le_amd/csrc/asan/liblammps_le_asan.so python3 -m pytest tests -x -q -m "not gpu"
rm -rf lammps_le_amd/csrc/asan      # (12 MB of objects: not worth a place in the snapshots sent to the GPU box)
