// main.cpp — `lmp_le -in script` command-line driver over the C-ABI (counterpart of src/main.cpp)
#include <cstdio>
#include <cstring>

#include "../../include/lammps_le.h"

int main(int argc, char **argv) {
  const char *in = nullptr;
  for (int i = 1; i < argc - 1; i++)
    if (!strcmp(argv[i], "-in") || !strcmp(argv[i], "-i")) in = argv[i + 1];
  void *h = lammps_open_no_mpi(argc, argv, nullptr);
  if (!h) return 1;
  if (in) lammps_file(h, in);
  else {
    char line[4096];
    while (fgets(line, sizeof line, stdin)) {
      lammps_command(h, line);
      if (lammps_has_error(h)) break;
    }
  }
  int rc = lammps_has_error(h);
  lammps_close(h);
  return rc;
}
