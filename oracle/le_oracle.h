/* le_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the per-timestep hot path of polly-code/lammps_le
 * (stock LAMMPS 29Oct2020 + USER-LE) at the canonical configuration:
 * 1 MPI rank, newton_bond off.  Every function cites the reference file:line
 * it restates.  Nothing under lammps_le_amd/ (the product) may include, link
 * or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / reported CPU baseline.
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *  - pinned by reference-owned vectors: LJ (unittest mol-pair-lj_cut.yaml),
 *    FENE / harmonic / hybrid(+morse) bonds (bond-fene.yaml, bond-harmonic.yaml, bond-hybrid.yaml),
 *    harmonic / cosine angles (angle-harmonic.yaml, angle-cosine.yaml), fix nve on a group with
 *    pair + bond + angle forces (fix-timestep-nve.yaml), NVE + RanMars +
 *    Langevin + Atom::sort + thermo (bench/log.6Oct16.chain.fixed.icc.1:48-49).
 *  - USER-LE fixes (extrusion / ex_load / ex_unload): PARITY UNPINNED — the
 *    reference holds no test, example or log for them and the reference cannot
 *    be built under this round's rules (needs generated style_*.h headers).
 */
#ifndef LE_ORACLE_H
#define LE_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct leo leo_t;

/* ---- RanMars (src/random_mars.cpp:29-95) ---- */
typedef struct {
  double u[98];
  int i97, j97;
  double c, cd, cm;
} leo_ranmars;
void   leo_ranmars_init(leo_ranmars *r, int seed);
double leo_ranmars_uniform(leo_ranmars *r);
/* fill out[0..n) with successive uniform() values of RanMars(seed) */
void   leo_ranmars_stream(int seed, int n, double *out);

/* ---- system construction ---- */
leo_t *leo_new(int natoms, int ntypes, int nbondtypes, int bond_per_atom, int maxspecial);
void   leo_free(leo_t *s);
/* units: 0 = lj, 1 = real  (src/update.cpp:132-200) */
void   leo_units(leo_t *s, int units);
void   leo_set_box(leo_t *s, const double lo[3], const double hi[3]);
void   leo_set_mass(leo_t *s, int type, double m);
/* atoms in data-file order; x,v are [n][3]; image [n][3] */
void   leo_set_atoms(leo_t *s, const int *tag, const int *type, const double *x,
                     const double *v, const int *image);
/* bonds as in the data file "Bonds" section; stored on BOTH atoms (newton_bond off,
   src/atom.cpp:1235-1282) in section order */
void   leo_set_bonds(leo_t *s, int nbonds, const int *btype, const int *a1, const int *a2);
/* special_bonds lj w1 w2 w3 ; builds 1-2/1-3/1-4 lists (src/special.cpp:55-) */
void   leo_special_build(leo_t *s, double w1, double w2, double w3);
/* Coulomb weights of special_bonds (src/force.cpp:748-826; defaults 0 0 0).  No Coulomb on this path, but they decide
   with the lj weights which special levels are dropped from / flagged in the pair list and which lists get built;
   call BEFORE leo_special_build */
void   leo_special_coul(leo_t *s, double c1, double c2, double c3);

/* ---- force field ---- */
/* pair_style lj/cut rc ; mix: 0 geometric, 1 arithmetic ; shift: pair_modify shift yes */
void   leo_pair_lj_cut(leo_t *s, double cut_global, int shift, int mix);
void   leo_pair_coeff(leo_t *s, int i, int j, double eps, double sigma, double cut /* <0: global */);
/* bond styles per bond type: 1 = fene (K R0 eps sigma), 2 = harmonic (K r0), 3 = morse (D alpha r0) */
void   leo_bond_coeff(leo_t *s, int btype, int style, double p0, double p1, double p2, double p3);

/* angles (newton_bond off: stored on all three atoms): Angles section rows + "extra angle per atom"; call after leo_set_atoms */
void   leo_set_angles(leo_t *s, int nangletypes, int nangles, const int *atype, const int *a1, const int *a2, const int *a3,
                      int extra_angle);
/* angle_coeff: style 1 = harmonic (K, theta0 in degrees), 2 = cosine (K) */
void   leo_angle_coeff(leo_t *s, int type, int style, double k, double theta0_deg);
/* fix ID group nve with a group other than all: flag_by_tag[t-1] = 1 for integrated atoms */
/* the group of fix number fix_index (LE fixes and their src/MC parents: both atoms of a bond / candidate pair must be members) */
void   leo_fix_group(leo_t *s, int fix_index, const int *flag_by_tag);
void   leo_nve_group(leo_t *s, const int *flag_by_tag);
/* fix ID group langevin with a group other than all: only members draw (and feel drag / noise), in local order */
void   leo_langevin_group(leo_t *s, const int *flag_by_tag);
/* fix langevin keywords `scale itype ratio` and `zero yes|no`, for the most recently defined fix langevin */
void   leo_langevin_scale(leo_t *s, int itype, double ratio);
void   leo_langevin_zero(leo_t *s, int flag);
/* fix ex_load ... atype N (fix_ex_load.cpp:855-954): angles of type N are created around every new bond */
void   leo_ex_load_atype(leo_t *s, int fix_index, int atype);
long   leo_nangles(leo_t *s);
int    leo_angle_per_atom(leo_t *s);
void   leo_get_angles(leo_t *s, int *num_angle, int *angle_type, int *a1, int *a2, int *a3);   /* tag order, [n], [n*apa] */
double leo_angle_energy(leo_t *s);
void   leo_angle_virial(leo_t *s, double *out6);

/* ---- settings ---- */
void   leo_timestep(leo_t *s, double dt);
void   leo_neighbor(leo_t *s, double skin, int every, int delay, int check);
void   leo_atom_sort(leo_t *s, int sortfreq);        /* atom_modify sort N 0 */
int    leo_run_style_respa_angle(leo_t *s, int level_angle);     /* keyword `angle L` (0 = default: the bond level) */
int    leo_run_style_respa(leo_t *s, int nlevels, const int *loops, int level_bond, int level_pair);   /* 0 levels = verlet; level args 1-based, 0 = default */
void   leo_newton_pair(leo_t *s, int on);            /* newton on|off for pairs: which end stores a pair (visit order of ex_load) */
void   leo_reset_timestep(leo_t *s, long step);
void   leo_thermo_every(leo_t *s, int n);            /* thermo N : record a thermo snapshot every N steps */

/* ---- fixes (called in script order) ---- */
void   leo_fix_nve(leo_t *s);
void   leo_fix_langevin(leo_t *s, double t_start, double t_stop, double damp, int seed);
void   leo_fix_extrusion(leo_t *s, int nevery, int neutral, int ctcf_left, int ctcf_right,
                         double through_prob, int btype, int ctcf_left_right /* -1 if absent */);
void   leo_fix_ex_load(leo_t *s, int nevery, int iatomtype, int jatomtype, double cutoff,
                       int btype, int imaxbond, int inewtype, int jmaxbond, int jnewtype,
                       double fraction, int seed);
void   leo_fix_ex_unload(leo_t *s, int nevery, int btype, double cutoff, double fraction, int seed);
void   leo_fix_bond_create(leo_t *s, int nevery, int it, int jt, double cutoff, int btype, int imax, int inew,
                           int jmax, int jnew, double fraction, int seed);
void   leo_fix_bond_break(leo_t *s, int nevery, int btype, double cutoff, double fraction, int seed);

/* ---- running ---- */
/* run N steps exactly as Run::command + Verlet::setup/run (src/run.cpp, src/verlet.cpp).
   returns 0 on success, nonzero on a reference error (message via leo_error) */
int    leo_run(leo_t *s, int nsteps);
const char *leo_error(leo_t *s);
/* single calls for state-in/state-out tests (no integration): */
int    leo_setup_forces(leo_t *s);              /* pbc+neighbor+pair+bond with eflag/vflag; no fixes */
int    leo_fire_fix(leo_t *s, int fix_index);   /* call post_integrate body of LE fix regardless of step */

/* ---- queries ---- */
long   leo_ntimestep(leo_t *s);
int    leo_natoms(leo_t *s);
long   leo_nbonds(leo_t *s);
/* thermo: out[0]=temp out[1]=epair out[2]=emol out[3]=toteng out[4]=press out[5]=ke (same
   normalisation as the reference's thermo line: /natoms in lj units) ; also raw: out[6]=evdwl
   out[7]=ebond out[8..13]=virial(xx,yy,zz,xy,xz,yz) total pair+bond (not normalised) */
void   leo_thermo(leo_t *s, double *out);
/* thermo history of all leo_run calls: entry = {step, out[0..13] as above, nbonds} */
int    leo_thermo_count(leo_t *s);
void   leo_thermo_get(leo_t *s, int idx, double *out16);
int    leo_bond_per_atom(leo_t *s);
int    leo_maxspecial(leo_t *s);
void   leo_pair_virial(leo_t *s, double *out6);
void   leo_bond_virial(leo_t *s, double *out6);
/* per-atom state gathered in TAG order (like lammps_gather_atoms) */
void   leo_get_x(leo_t *s, double *out);
void   leo_get_v(leo_t *s, double *out);
void   leo_get_f(leo_t *s, double *out);
void   leo_get_type(leo_t *s, int *out);
void   leo_get_image(leo_t *s, int *out);
void   leo_get_local_order(leo_t *s, int *out_tags);   /* tag of local index i */
void   leo_set_x(leo_t *s, const double *x_by_tag);
void   leo_set_v(leo_t *s, const double *v_by_tag);
/* topology, tag order: num_bond[n], bond_type[n*bpa], bond_atom[n*bpa] */
void   leo_get_bonds(leo_t *s, int *num_bond, int *bond_type, int *bond_atom);
void   leo_get_special(leo_t *s, int *nspecial3, int *special);
/* counters: fix vector f_ID[1], f_ID[2] */
void   leo_fix_vector(leo_t *s, int fix_index, double *out2);
long   leo_neigh_builds(leo_t *s);
long   leo_neigh_pairs(leo_t *s);       /* half-list pairs of the last build */
long   leo_fene_warnings(leo_t *s);
/* section timers of the last leo_run (seconds): pair,bond,neigh,modify,other,total */
void   leo_timers(leo_t *s, double *out6);

#ifdef __cplusplus
}
#endif
#endif
