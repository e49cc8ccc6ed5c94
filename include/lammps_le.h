/* lammps_le.h — C-ABI of the MI355X bead-spring/loop-extrusion engine.
 *
 * Drop-in boundary for the hot path of polly-code/lammps_le: the entry points are the subset of
 * the reference's C library interface (/root/reference/src/library.h) that a driver of this path
 * binds (src/main.cpp, python/lammps.py, unittest/c-library).  Same names, argument meaning and
 * ownership rules; plain pointers and sizes only.  Each declaration cites the reference line it
 * replaces.  Errors never exit() across the boundary: they set the error flag
 * (lammps_has_error / lammps_get_last_error_message, as a LAMMPS_EXCEPTIONS build does).
 *
 * The handle owns one GPU (LOCAL_RANK selects it) and is single-threaded like the reference.
 */
#ifndef LAMMPS_LE_H
#define LAMMPS_LE_H
#ifdef __cplusplus
extern "C" {
#endif

/* create / destroy — library.h:93, :95.  argv accepts -screen none|file, -log none|file, -echo, -var */
void *lammps_open_no_mpi(int argc, char **argv, void **ptr);
void  lammps_close(void *handle);

/* commands — library.h:104, :106, :107, :108.  lammps_command returns a BORROWED pointer to the
 * parsed command name (src/library.cpp:410-426 -> Input::one), NULL on error or empty line. */
void  lammps_file(void *handle, const char *file);
char *lammps_command(void *handle, const char *cmd);
void  lammps_commands_list(void *handle, int ncmd, const char **cmds);
void  lammps_commands_string(void *handle, const char *str);

/* queries — library.h:114, :115, :116-118, :122, :124 */
double lammps_get_natoms(void *handle);
double lammps_get_thermo(void *handle, const char *keyword);
void   lammps_extract_box(void *handle, double *boxlo, double *boxhi, double *xy, double *yz, double *xz,
                          int *pflags, int *boxflag);
int    lammps_extract_setting(void *handle, const char *keyword);
/* names: dt ntimestep atime atimestep boxlo boxhi natoms nbonds ntypes boltz units (src/library.cpp:1230-1420) */
void  *lammps_extract_global(void *handle, const char *name);

/* per-atom data — library.h:135.  Borrowed pointers into engine memory in TAG order (row t-1 is
 * atom ID t), valid until the next command: "x","v","f" -> double** ; "type","id","mask" -> int* ;
 * "image" -> int* (encoded as lammps_encode_image_flags) ; "mass" -> double* (index = type) */
void  *lammps_extract_atom(void *handle, const char *name);

/* fix data — library.h:142.  style 0 = global, type 0/1/2 = scalar/vector/array; returns a malloc'ed
 * double the caller releases with lammps_free (as the reference does for global fix data) */
void  *lammps_extract_fix(void *handle, char *id, int style, int type, int nrow, int ncol);

/* gather / scatter ordered by atom ID — library.h:150, :153.  type 0 = int, 1 = double; count = values/atom.
 * names: x v f (count 3), type id mask (count 1), image (count 1 or 3),
 * plus topology vectors of this path: num_bond (1), bond_type, bond_atom (count = bond_per_atom) */
void lammps_gather_atoms(void *handle, char *name, int type, int count, void *data);
void lammps_scatter_atoms(void *handle, char *name, int type, int count, void *data);

/* utility — library.h:163, :206-207, :231, :233, :236-237 */
int  lammps_version(void *handle);
int  lammps_encode_image_flags(int ix, int iy, int iz);
void lammps_decode_image_flags(int image, int *flags);
void lammps_free(void *ptr);
int  lammps_is_running(void *handle);
int  lammps_has_error(void *handle);
int  lammps_get_last_error_message(void *handle, char *buffer, int buf_size);
int  lammps_config_has_exceptions(void);
int  lammps_has_style(void *handle, const char *category, const char *name);

/* engine-specific introspection (no reference counterpart; used by bench.py / tests):
 * name = "loop_time", "neigh_builds", "pair_kernel_ms" (mean HIP-event duration of the force kernel
 * over the last run), "pair_kernel_launches", "neigh_pairs" (stored full-list entries), "maxneigh", "nlocal", "nghost",
 * "fene_warnings", "time_pair" / "_bond" / "_neigh" / "_comm" / "_output" / "_modify" / "_other" (the reference's loop
 * sections, seconds), "special_asym" (1 once some bead's 1-2 list lost an entry its partner still has: the list build
 * then asks which end stores each pair); with ranks: "comm_nranks", "comm_bytes_allgather", "comm_bytes_allreduce",
 * "halo_window_exchanges", "halo_window_mismatches", "halo_fused", "rng_segments", "rng_segments_held",
 * "rng_late_generations".  An unknown name returns 0. */
double lammps_le_stat(void *handle, const char *name);
/* the thermo lines printed so far as numbers: rows of 7 doubles (step, temp, epair, emol, etotal, press, bonds); returns the
 * number of rows, writes at most max_rows (the reference's counterpart is parsing its log file) */
int    lammps_le_thermo_log(void *handle, double *out, int max_rows);

/* ranks: one process per GPU.  Replaces the MPI_Comm argument of the reference's `lammps_open` entry point
 * (library.h:91; there is no MPI here, so that entry point itself is not exported): the launcher creates a 128-byte RCCL unique
 * id on rank 0, distributes it, and every rank joins before its first `run`.  backend = "rccl" (xGMI) or "shm"
 * (file mailbox; tests).  Afterwards every rank issues the same commands, as MPI ranks of the reference do. */
int  lammps_le_comm_unique_id(char *out128);
void lammps_le_comm_init(void *handle, const char *backend, int rank, int world, const char *unique_id,
                         const char *session);

#ifdef __cplusplus
}
#endif
#endif
