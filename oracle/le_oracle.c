/* le_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see le_oracle.h).
 *
 * Serial plain-C restatement of the reference hot path at 1 rank / newton_bond off.
 * Paths below are relative to /root/reference/.
 */
#define _POSIX_C_SOURCE 200809L
#include "le_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define BIG 1.0e20
#define TWO_1_3 1.2599210498948732  /* src/math_const.h MY_CUBEROOT2 */
#define MAXFIX 16

enum { FIX_NVE = 1, FIX_LANGEVIN, FIX_EXTRUSION, FIX_EX_LOAD, FIX_EX_UNLOAD, FIX_BOND_CREATE };

typedef struct {
  int kind;
  /* langevin */
  double t_start, t_stop, t_period; int seed;
  double *gfactor1, *gfactor2; double tsqrt;
  double *ratio; int zeroflag;   /* keywords `scale itype ratio`, `zero yes` (src/fix_langevin.cpp:135-141, 148-153) */
  leo_ranmars rng;
  /* LE common */
  int nevery, btype;
  int atype;                 /* ex_load: angle type created with every new bond (0 = none), fix_ex_load.cpp:855-954 */
  int phase;                 /* fires when ntimestep % nevery == phase */
  long next_reneighbor;
  double cutsq, fraction;
  /* extrusion */
  int neutral, ctcf_left, ctcf_right, ctcf_lr; double through_prob;
  /* ex_load */
  int iatomtype, jatomtype, imaxbond, inewtype, jmaxbond, jnewtype;
  /* bond/create: bonds of btype per atom, by tag, counted at the first setup and then only incremented (stock rule) */
  int *bc_tag; int counted;
  /* counters */
  int lastcount; long totalcount;
  int *gmask;               /* fix nve / fix langevin on a group: by tag, 1 = member (NULL: group all) */
} leo_fix;

struct leo {
  int n, ntypes, nbondtypes, extra_bond, extra_special, bpa, maxspecial;
  int units;
  double boltz, mvv2e, ftm2v, nktv2p, dt;
  double lo[3], hi[3], prd[3];
  /* per-atom (local index order) */
  int *tag, *type, *img;
  double *x, *v, *f;
  int *num_bond, *bond_type, *bond_atom;
  int *nspecial, *special;
  int maxtag, *map;
  double *mass;
  long nbonds;
  /* pair */
  int pair_on, shift, mix; double cut_global;
  double *eps, *sig, *cut, *lj1, *lj2, *lj3, *lj4, *offset, *cutsq; int *setflag;
  double special_lj[4], special_coul[4];   /* src/force.cpp:47-49: coul defaults 0 0 0 (no Coulomb here; only the list flags use it) */
  /* bonds */
  int *bstyle; double *bp0, *bp1, *bp2, *bp3;
  /* angles (newton_bond off: each of the three atoms stores the angle, src/atom.cpp:1290-1353) */
  int nangletypes, apa;                 /* angle types, angles per atom */
  int *num_angle, *angle_type, *angle_a1, *angle_a2, *angle_a3;      /* [n], [n * apa] local-index order, atoms as tags */
  int *astyle; double *ak, *atheta0;    /* per type: 1 harmonic (k, theta0 in radians), 2 cosine (k) */
  long nangles;
  int unload_angleflag;                 /* FixExUnload::init: angles exist -> broken bonds take their angles with them */
  int nanglelist, maxanglelist; int *al_i; signed char *al_s;        /* anglelist: 4 ints (i1 i2 i3 type) + 9 image shifts per entry */
  double eangle, vangle[6];
  /* neighbor */
  double skin, cutneighmax, triggersq; int every, delay, check, ago;
  int newton_pair;        /* `newton on|off [bond]`: who stores an owned-owned pair (half/bin/newton vs newtoff); bonds are always newton off */
  long nbuilds, ndanger;
  int brute;              /* image-enumerating path for small boxes */
  long npairs, maxpairs, noneside;
  int *firstneigh;        /* CSR [n+1] (cell path) */
  int *pj;                /* partner local index | special bits << 30 */
  int *pi_;               /* brute path: explicit i */
  signed char *pshift;    /* brute path: 3 shifts per pair */
  double *xhold;
  /* bond list (src/ntopo_bond_all.cpp:39-86) */
  int nbondlist, maxbondlist; int *bl_i, *bl_j, *bl_t; signed char *bl_s;
  /* fixes */
  int nfix; leo_fix fix[MAXFIX];
  /* sort */
  int sortfreq; long nextsort;
  /* run_style respa (src/respa.cpp): 0 levels = run_style verlet */
  int respa_levels, respa_loop[8], respa_level_bond, respa_level_pair, respa_level_angle;
  double respa_step[8];
  double *flevel[8];      /* FixRespa's per-level force arrays (src/fix_respa.cpp), kept BY TAG: immune to pbc / sort permutations */
  /* run state */
  long ntimestep, beginstep, endstep;
  int thermo_every;
  /* energies / virial of last eflag evaluation */
  double evdwl, ebond, vpair[6], vbond[6];
  int nthermo, maxthermo; double *thermo_hist;   /* 16 doubles per entry: step + 14 */
  long fene_warn;
  char err[256]; int errflag;
  double t_pair, t_bond, t_neigh, t_modify, t_total;
  /* scratch for LE fixes */
  int *bondcount, *ia, *ib, *ic, *id, *ie; double *da, *db;
  int *copy;
};

static double now(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static int seterr(leo_t *s, const char *m) {
  if (!s->errflag) { snprintf(s->err, sizeof s->err, "%s", m); s->errflag = 1; }
  return 1;
}

/* ===================== RanMars: src/random_mars.cpp:29-95 ===================== */
void leo_ranmars_init(leo_ranmars *r, int seed) {
  int ij, kl, i, j, k, l, ii, jj, m; double s, t;
  memset(r->u, 0, sizeof r->u);
  ij = (seed - 1) / 30082;
  kl = (seed - 1) - 30082 * ij;
  i = (ij / 177) % 177 + 2;
  j = ij % 177 + 2;
  k = (kl / 169) % 178 + 1;
  l = kl % 169;
  for (ii = 1; ii <= 97; ii++) {
    s = 0.0; t = 0.5;
    for (jj = 1; jj <= 24; jj++) {
      m = ((i * j) % 179) * k % 179;
      i = j; j = k; k = m;
      l = (53 * l + 1) % 169;
      if ((l * m) % 64 >= 32) s = s + t;
      t = 0.5 * t;
    }
    r->u[ii] = s;
  }
  r->c = 362436.0 / 16777216.0;
  r->cd = 7654321.0 / 16777216.0;
  r->cm = 16777213.0 / 16777216.0;
  r->i97 = 97; r->j97 = 33;
  leo_ranmars_uniform(r);
}
double leo_ranmars_uniform(leo_ranmars *r) {
  double uni = r->u[r->i97] - r->u[r->j97];
  if (uni < 0.0) uni += 1.0;
  r->u[r->i97] = uni;
  r->i97--; if (r->i97 == 0) r->i97 = 97;
  r->j97--; if (r->j97 == 0) r->j97 = 97;
  r->c -= r->cd; if (r->c < 0.0) r->c += r->cm;
  uni -= r->c; if (uni < 0.0) uni += 1.0;
  return uni;
}
void leo_ranmars_stream(int seed, int n, double *out) {
  leo_ranmars r; leo_ranmars_init(&r, seed);
  for (int i = 0; i < n; i++) out[i] = leo_ranmars_uniform(&r);
}

/* ===================== construction ===================== */
leo_t *leo_new(int natoms, int ntypes, int nbondtypes, int extra_bond, int extra_special) {
  leo_t *s = calloc(1, sizeof *s);
  s->n = natoms; s->ntypes = ntypes; s->nbondtypes = nbondtypes;
  s->extra_bond = extra_bond; s->extra_special = extra_special;
  int n = natoms, nt = ntypes + 1;
  s->tag = calloc(n, sizeof(int)); s->type = calloc(n, sizeof(int)); s->img = calloc(3 * n, sizeof(int));
  s->x = calloc(3 * n, sizeof(double)); s->v = calloc(3 * n, sizeof(double)); s->f = calloc(3 * n, sizeof(double));
  s->xhold = calloc(3 * n, sizeof(double));
  s->mass = calloc(nt, sizeof(double));
  s->eps = calloc(nt * nt, sizeof(double)); s->sig = calloc(nt * nt, sizeof(double)); s->cut = calloc(nt * nt, sizeof(double));
  s->lj1 = calloc(nt * nt, sizeof(double)); s->lj2 = calloc(nt * nt, sizeof(double));
  s->lj3 = calloc(nt * nt, sizeof(double)); s->lj4 = calloc(nt * nt, sizeof(double));
  s->offset = calloc(nt * nt, sizeof(double)); s->cutsq = calloc(nt * nt, sizeof(double));
  s->setflag = calloc(nt * nt, sizeof(int));
  int nb = nbondtypes + 1;
  s->bstyle = calloc(nb, sizeof(int));
  s->bp0 = calloc(nb, sizeof(double)); s->bp1 = calloc(nb, sizeof(double));
  s->bp2 = calloc(nb, sizeof(double)); s->bp3 = calloc(nb, sizeof(double));
  s->special_lj[0] = 1.0; s->special_lj[1] = s->special_lj[2] = s->special_lj[3] = 0.0;
  s->special_coul[0] = 1.0; s->special_coul[1] = s->special_coul[2] = s->special_coul[3] = 0.0;
  /* defaults: src/neighbor.cpp:87-89, src/atom.cpp:89-91, src/update.cpp (units lj) */
  s->every = 1; s->delay = 10; s->check = 1; s->sortfreq = 1000;
  s->thermo_every = 0;
  leo_units(s, 0);
  s->skin = 0.3;
  s->bondcount = calloc(n, sizeof(int));
  s->ia = calloc(n, sizeof(int)); s->ib = calloc(n, sizeof(int)); s->ic = calloc(n, sizeof(int));
  s->id = calloc(n, sizeof(int)); s->ie = calloc(n, sizeof(int));
  s->da = calloc(n, sizeof(double)); s->db = calloc(n, sizeof(double));
  return s;
}
void leo_free(leo_t *s) {
  if (!s) return;
  free(s->tag); free(s->type); free(s->img); free(s->x); free(s->v); free(s->f); free(s->xhold);
  for (int l = 0; l < 8; l++) free(s->flevel[l]);
  free(s->num_bond); free(s->bond_type); free(s->bond_atom); free(s->nspecial); free(s->special);
  free(s->map); free(s->mass);
  free(s->eps); free(s->sig); free(s->cut); free(s->lj1); free(s->lj2); free(s->lj3); free(s->lj4);
  free(s->offset); free(s->cutsq); free(s->setflag);
  free(s->bstyle); free(s->bp0); free(s->bp1); free(s->bp2); free(s->bp3);
  free(s->firstneigh); free(s->pj); free(s->pi_); free(s->pshift);
  free(s->bl_i); free(s->bl_j); free(s->bl_t); free(s->bl_s);
  for (int i = 0; i < s->nfix; i++) { free(s->fix[i].gfactor1); free(s->fix[i].gfactor2); free(s->fix[i].bc_tag); }
  free(s->thermo_hist);
  free(s->bondcount); free(s->ia); free(s->ib); free(s->ic); free(s->id); free(s->ie);
  free(s->da); free(s->db); free(s->copy);
  free(s->num_angle); free(s->angle_type); free(s->angle_a1); free(s->angle_a2); free(s->angle_a3);
  free(s->astyle); free(s->ak); free(s->atheta0); free(s->al_i); free(s->al_s); for (int k = 0; k < s->nfix; k++) free(s->fix[k].gmask);
  free(s);
}
/* src/update.cpp:132-200 (set_units) */
void leo_units(leo_t *s, int units) {
  s->units = units;
  if (units == 0) {
    s->boltz = 1.0; s->mvv2e = 1.0; s->ftm2v = 1.0; s->nktv2p = 1.0; s->dt = 0.005; s->skin = 0.3;
  } else {
    s->boltz = 0.0019872067; s->mvv2e = 48.88821291 * 48.88821291;
    s->ftm2v = 1.0 / 48.88821291 / 48.88821291; s->nktv2p = 68568.415; s->dt = 1.0; s->skin = 2.0;
  }
}
void leo_set_box(leo_t *s, const double lo[3], const double hi[3]) {
  for (int d = 0; d < 3; d++) { s->lo[d] = lo[d]; s->hi[d] = hi[d]; s->prd[d] = hi[d] - lo[d]; }
}
void leo_set_mass(leo_t *s, int type, double m) { s->mass[type] = m; }
void leo_set_atoms(leo_t *s, const int *tag, const int *type, const double *x, const double *v, const int *image) {
  int n = s->n;
  memcpy(s->tag, tag, n * sizeof(int)); memcpy(s->type, type, n * sizeof(int));
  memcpy(s->x, x, 3 * n * sizeof(double));
  if (v) memcpy(s->v, v, 3 * n * sizeof(double));
  if (image) memcpy(s->img, image, 3 * n * sizeof(int));
  s->maxtag = 0;
  for (int i = 0; i < n; i++) if (tag[i] > s->maxtag) s->maxtag = tag[i];
  free(s->map); s->map = malloc((s->maxtag + 2) * sizeof(int));
  for (int t = 0; t <= s->maxtag + 1; t++) s->map[t] = -1;
  for (int i = 0; i < n; i++) s->map[tag[i]] = i;
}
static inline int map_(const leo_t *s, int t) { return (t < 0 || t > s->maxtag) ? -1 : s->map[t]; }

/* src/atom.cpp:1235-1282 with newton_bond == 0 */
void leo_set_bonds(leo_t *s, int nbonds, const int *btype, const int *a1, const int *a2) {
  int n = s->n;
  int *cnt = calloc(n, sizeof(int));
  for (int b = 0; b < nbonds; b++) { cnt[s->map[a1[b]]]++; cnt[s->map[a2[b]]]++; }
  int mx = 0; for (int i = 0; i < n; i++) if (cnt[i] > mx) mx = cnt[i];
  s->bpa = mx + s->extra_bond;
  if (s->bpa < 1) s->bpa = 1;
  free(s->num_bond); free(s->bond_type); free(s->bond_atom);
  s->num_bond = calloc(n, sizeof(int));
  s->bond_type = calloc((size_t)n * s->bpa, sizeof(int));
  s->bond_atom = calloc((size_t)n * s->bpa, sizeof(int));
  for (int b = 0; b < nbonds; b++) {
    int m = s->map[a1[b]];
    s->bond_type[m * s->bpa + s->num_bond[m]] = btype[b]; s->bond_atom[m * s->bpa + s->num_bond[m]] = a2[b]; s->num_bond[m]++;
    m = s->map[a2[b]];
    s->bond_type[m * s->bpa + s->num_bond[m]] = btype[b]; s->bond_atom[m * s->bpa + s->num_bond[m]] = a1[b]; s->num_bond[m]++;
  }
  s->nbonds = nbonds;
  free(cnt);
}

/* src/special.cpp:55- (build), :533 dedup, combine: set semantics, generation order */
static int append_unique(int *list, int n, int self, int val, const int *a, int na, const int *b, int nb) {
  if (val == self) return n;
  for (int k = 0; k < n; k++) if (list[k] == val) return n;
  for (int k = 0; k < na; k++) if (a[k] == val) return n;
  for (int k = 0; k < nb; k++) if (b[k] == val) return n;
  list[n] = val; return n + 1;
}
/* special_bonds ... coul c1 c2 c3 (src/force.cpp:748-826); call before leo_special_build */
void leo_special_coul(leo_t *s, double c1, double c2, double c3) {
  s->special_coul[0] = 1.0; s->special_coul[1] = c1; s->special_coul[2] = c2; s->special_coul[3] = c3;
}
void leo_special_build(leo_t *s, double w1, double w2, double w3) {
  int n = s->n;
  s->special_lj[0] = 1.0; s->special_lj[1] = w1; s->special_lj[2] = w2; s->special_lj[3] = w3;
  const double *c = s->special_coul;
  /* src/special.cpp:97-131: 1-3 (1-4) lists are built unless BOTH the lj and the coul weights of the levels beyond are 1.0 */
  int do13 = !(w2 == 1.0 && c[2] == 1.0 && w3 == 1.0 && c[3] == 1.0), do14 = do13 && !(w3 == 1.0 && c[3] == 1.0);
  int cap = 4096;
  int *l12 = malloc(cap * sizeof(int)), *l13 = malloc(cap * sizeof(int)), *l14 = malloc(cap * sizeof(int));
  /* two passes: size then fill */
  int *n12 = calloc(n, sizeof(int)), *n13 = calloc(n, sizeof(int)), *n14 = calloc(n, sizeof(int));
  int **tmp = calloc(n, sizeof(int *));
  int maxall = 0;
  for (int i = 0; i < n; i++) {
    int a = 0, b = 0, c = 0, self = s->tag[i];
    for (int m = 0; m < s->num_bond[i]; m++) a = append_unique(l12, a, self, s->bond_atom[i * s->bpa + m], NULL, 0, NULL, 0);
    if (do13)
      for (int j = 0; j < a; j++) {
        int jj = s->map[l12[j]];
        for (int m = 0; m < s->num_bond[jj]; m++) b = append_unique(l13, b, self, s->bond_atom[jj * s->bpa + m], l12, a, NULL, 0);
      }
    if (do14)
      for (int j = 0; j < b; j++) {
        int jj = s->map[l13[j]];
        for (int m = 0; m < s->num_bond[jj]; m++) c = append_unique(l14, c, self, s->bond_atom[jj * s->bpa + m], l12, a, l13, b);
      }
    n12[i] = a; n13[i] = b; n14[i] = c;
    tmp[i] = malloc((a + b + c + 1) * sizeof(int));
    memcpy(tmp[i], l12, a * sizeof(int)); memcpy(tmp[i] + a, l13, b * sizeof(int)); memcpy(tmp[i] + a + b, l14, c * sizeof(int));
    if (a + b + c > maxall) maxall = a + b + c;
  }
  s->maxspecial = maxall + s->extra_special;
  if (s->maxspecial < 1) s->maxspecial = 1;
  free(s->nspecial); free(s->special);
  s->nspecial = calloc(3 * (size_t)n, sizeof(int));
  s->special = calloc((size_t)n * s->maxspecial, sizeof(int));
  for (int i = 0; i < n; i++) {
    s->nspecial[3 * i] = n12[i]; s->nspecial[3 * i + 1] = n12[i] + n13[i]; s->nspecial[3 * i + 2] = n12[i] + n13[i] + n14[i];
    memcpy(s->special + (size_t)i * s->maxspecial, tmp[i], (n12[i] + n13[i] + n14[i]) * sizeof(int));
    free(tmp[i]);
  }
  free(tmp); free(n12); free(n13); free(n14); free(l12); free(l13); free(l14);
  free(s->copy); s->copy = malloc(((size_t)s->maxspecial * s->maxspecial + s->maxspecial + 8) * sizeof(int));
}

/* ===================== force field ===================== */
void leo_pair_lj_cut(leo_t *s, double cut_global, int shift, int mix) {
  s->pair_on = 1; s->cut_global = cut_global; s->shift = shift; s->mix = mix;
}
void leo_pair_coeff(leo_t *s, int i, int j, double eps, double sigma, double cut) {
  int nt = s->ntypes + 1;
  if (i > j) { int t = i; i = j; j = t; }
  s->eps[i * nt + j] = eps; s->sig[i * nt + j] = sigma; s->cut[i * nt + j] = (cut < 0) ? s->cut_global : cut;
  s->setflag[i * nt + j] = 1;
}
void leo_bond_coeff(leo_t *s, int bt, int style, double p0, double p1, double p2, double p3) {
  s->bstyle[bt] = style; s->bp0[bt] = p0; s->bp1[bt] = p1; s->bp2[bt] = p2; s->bp3[bt] = p3;
}
/* src/pair_lj_cut.cpp:512-569 init_one ; mixing src/pair.cpp mix_energy/mix_distance */
/* Angles section of a data file (src/atom.cpp:1290-1353, newton_bond off): stored with atom2 first, then atom1, then atom3.
   apa = the largest count over the atoms + extra_angle ("extra angle per atom").  Call after leo_set_atoms. */
void leo_set_angles(leo_t *s, int nangletypes, int nangles, const int *atype, const int *a1, const int *a2, const int *a3,
                    int extra_angle) {
  int n = s->n;
  int *count = calloc(n, sizeof(int)), apa = 0;
  for (int k = 0; k < nangles; k++) { count[map_(s, a1[k])]++; count[map_(s, a2[k])]++; count[map_(s, a3[k])]++; }
  for (int i = 0; i < n; i++) if (count[i] > apa) apa = count[i];
  free(count);
  apa += extra_angle;
  if (apa < 1) apa = 1;
  s->apa = apa; s->nangletypes = nangletypes; s->nangles = nangles;
  free(s->num_angle); free(s->angle_type); free(s->angle_a1); free(s->angle_a2); free(s->angle_a3);
  s->num_angle = calloc(n, sizeof(int)); s->angle_type = calloc((size_t)n * apa, sizeof(int));
  s->angle_a1 = calloc((size_t)n * apa, sizeof(int)); s->angle_a2 = calloc((size_t)n * apa, sizeof(int));
  s->angle_a3 = calloc((size_t)n * apa, sizeof(int));
  free(s->astyle); free(s->ak); free(s->atheta0);
  s->astyle = calloc(nangletypes + 1, sizeof(int)); s->ak = calloc(nangletypes + 1, sizeof(double));
  s->atheta0 = calloc(nangletypes + 1, sizeof(double));
  for (int k = 0; k < nangles; k++) {
    const int order[3] = { a2[k], a1[k], a3[k] };
    for (int q = 0; q < 3; q++) {
      int m = map_(s, order[q]), c = s->num_angle[m]++;
      s->angle_type[(size_t)m * apa + c] = atype[k];
      s->angle_a1[(size_t)m * apa + c] = a1[k]; s->angle_a2[(size_t)m * apa + c] = a2[k]; s->angle_a3[(size_t)m * apa + c] = a3[k];
    }
  }
}
/* angle_coeff: style 1 = harmonic (K theta0[degrees], src/MOLECULE/angle_harmonic.cpp:170-196), 2 = cosine (K, angle_cosine.cpp:124-146) */
void leo_angle_coeff(leo_t *s, int type, int style, double k, double theta0_deg) {
  s->astyle[type] = style; s->ak[type] = k; s->atheta0[type] = theta0_deg / 180.0 * 3.14159265358979323846;
}
/* fix nve / fix langevin on a group: flag by tag (1 = member) for the most recently defined fix of that style.
   The reference's group mask, fix_nve.cpp:82, fix_langevin.cpp:661 (only members draw, in local order) */
static void set_fix_group(leo_t *s, int kind, const int *flag_by_tag) {
  for (int k = s->nfix - 1; k >= 0; k--)
    if (s->fix[k].kind == kind) {
      free(s->fix[k].gmask);
      s->fix[k].gmask = malloc(((size_t)s->maxtag + 1) * sizeof(int));
      s->fix[k].gmask[0] = 0;
      memcpy(s->fix[k].gmask + 1, flag_by_tag, (size_t)s->maxtag * sizeof(int));
      return;
    }
}
/* any fix by index: the LE fixes test both atoms of a bond / candidate pair against their group
   (fix_extrusion.cpp:373-376, fix_ex_load.cpp:435,450, fix_ex_unload.cpp:228-229, and their src/MC parents) */
void leo_fix_group(leo_t *s, int fix_index, const int *flag_by_tag) {
  leo_fix *f = &s->fix[fix_index];
  free(f->gmask);
  f->gmask = malloc(((size_t)s->maxtag + 1) * sizeof(int));
  f->gmask[0] = 0;
  memcpy(f->gmask + 1, flag_by_tag, (size_t)s->maxtag * sizeof(int));
}
#define IN_GROUP(s, fx, i) (!(fx)->gmask || (fx)->gmask[(s)->tag[i]])
void leo_nve_group(leo_t *s, const int *flag_by_tag) { set_fix_group(s, FIX_NVE, flag_by_tag); }
void leo_langevin_group(leo_t *s, const int *flag_by_tag) { set_fix_group(s, FIX_LANGEVIN, flag_by_tag); }
void leo_ex_load_atype(leo_t *s, int fix_index, int atype) { s->fix[fix_index].atype = atype; }
long leo_nangles(leo_t *s) { return s->nangles; }
int leo_angle_per_atom(leo_t *s) { return s->apa; }
void leo_get_angles(leo_t *s, int *na, int *at, int *a1, int *a2, int *a3) {      /* tag order */
  for (int i = 0; i < s->n; i++) {
    size_t t = (size_t)(s->tag[i] - 1);
    na[t] = s->num_angle ? s->num_angle[i] : 0;
    for (int m = 0; m < s->apa; m++) {
      at[t * s->apa + m] = s->angle_type[(size_t)i * s->apa + m]; a1[t * s->apa + m] = s->angle_a1[(size_t)i * s->apa + m];
      a2[t * s->apa + m] = s->angle_a2[(size_t)i * s->apa + m]; a3[t * s->apa + m] = s->angle_a3[(size_t)i * s->apa + m];
    }
  }
}
double leo_angle_energy(leo_t *s) { return s->eangle; }
void leo_angle_virial(leo_t *s, double *o) { memcpy(o, s->vangle, sizeof s->vangle); }

static void pair_init(leo_t *s) {
  int nt = s->ntypes + 1;
  s->cutneighmax = 0.0;
  if (!s->pair_on) return;
  for (int i = 1; i <= s->ntypes; i++)
    for (int j = i; j <= s->ntypes; j++) {
      int ij = i * nt + j, ji = j * nt + i, ii = i * nt + i, jj = j * nt + j;
      if (!s->setflag[ij]) {
        s->eps[ij] = sqrt(s->eps[ii] * s->eps[jj]);
        s->sig[ij] = s->mix ? 0.5 * (s->sig[ii] + s->sig[jj]) : sqrt(s->sig[ii] * s->sig[jj]);
        s->cut[ij] = s->mix ? 0.5 * (s->cut[ii] + s->cut[jj]) : sqrt(s->cut[ii] * s->cut[jj]);
      }
      s->lj1[ij] = 48.0 * s->eps[ij] * pow(s->sig[ij], 12.0);
      s->lj2[ij] = 24.0 * s->eps[ij] * pow(s->sig[ij], 6.0);
      s->lj3[ij] = 4.0 * s->eps[ij] * pow(s->sig[ij], 12.0);
      s->lj4[ij] = 4.0 * s->eps[ij] * pow(s->sig[ij], 6.0);
      if (s->shift && s->cut[ij] > 0.0) {
        double ratio = s->sig[ij] / s->cut[ij];
        s->offset[ij] = 4.0 * s->eps[ij] * (pow(ratio, 12.0) - pow(ratio, 6.0));
      } else s->offset[ij] = 0.0;
      s->cutsq[ij] = s->cut[ij] * s->cut[ij];
      s->lj1[ji] = s->lj1[ij]; s->lj2[ji] = s->lj2[ij]; s->lj3[ji] = s->lj3[ij]; s->lj4[ji] = s->lj4[ij];
      s->offset[ji] = s->offset[ij]; s->cutsq[ji] = s->cutsq[ij]; s->cut[ji] = s->cut[ij];
      if (s->cut[ij] > s->cutneighmax) s->cutneighmax = s->cut[ij];
    }
  s->cutneighmax += s->skin;
}

void leo_timestep(leo_t *s, double dt) { s->dt = dt; }
void leo_neighbor(leo_t *s, double skin, int every, int delay, int check) {
  if (skin >= 0) s->skin = skin;
  if (every > 0) s->every = every;
  if (delay >= 0) s->delay = delay;
  if (check >= 0) s->check = check;
}
void leo_atom_sort(leo_t *s, int sortfreq) { s->sortfreq = sortfreq; }
/* run_style respa N loop_1 .. loop_{N-1} [bond L] [pair L] (src/respa.cpp:47-262: levels are 1-based in the script; defaults
   bond -> innermost, pair -> outermost; `inner/middle/outer`, `hybrid`, angle.. are outside this path).  nlevels = 0: verlet */
int leo_run_style_respa(leo_t *s, int nlevels, const int *loops, int level_bond, int level_pair) {
  if (nlevels < 0 || nlevels > 8) return seterr(s, "Respa levels must be >= 1");
  s->respa_levels = nlevels;
  for (int l = 0; l + 1 < nlevels; l++) { if (loops[l] <= 0) return seterr(s, "Illegal run_style respa command"); s->respa_loop[l] = loops[l]; }
  if (nlevels) s->respa_loop[nlevels - 1] = 1;
  s->respa_level_bond = level_bond > 0 ? level_bond - 1 : 0;                      /* :169 */
  s->respa_level_pair = level_pair > 0 ? level_pair - 1 : nlevels - 1;             /* :174-175 */
  if (nlevels && s->respa_level_pair < s->respa_level_bond) return seterr(s, "Invalid order of forces within respa levels");   /* :221-224 */
  s->respa_level_angle = s->respa_level_bond;                                      /* :172 */
  return 0;
}
/* keyword `angle L` (1-based; 0 = the default, the bond level): after leo_run_style_respa.  src/respa.cpp:85-88, 172, 217-224 */
int leo_run_style_respa_angle(leo_t *s, int level_angle) {
  s->respa_level_angle = level_angle > 0 ? level_angle - 1 : s->respa_level_bond;
  if (s->respa_levels && (s->respa_level_angle < s->respa_level_bond || s->respa_level_pair < s->respa_level_angle))
    return seterr(s, "Invalid order of forces within respa levels");
  return 0;
}
void leo_newton_pair(leo_t *s, int on) { s->newton_pair = on; }
void leo_reset_timestep(leo_t *s, long step) { s->ntimestep = step; }
void leo_thermo_every(leo_t *s, int n) { s->thermo_every = n; }

/* ===================== fixes: registration ===================== */
void leo_fix_nve(leo_t *s) { leo_fix *f = &s->fix[s->nfix++]; memset(f, 0, sizeof *f); f->kind = FIX_NVE; f->next_reneighbor = -1; }
void leo_fix_langevin(leo_t *s, double t_start, double t_stop, double damp, int seed) {
  leo_fix *f = &s->fix[s->nfix++]; memset(f, 0, sizeof *f);
  f->kind = FIX_LANGEVIN; f->t_start = t_start; f->t_stop = t_stop; f->t_period = damp; f->seed = seed;
  f->next_reneighbor = -1;
  leo_ranmars_init(&f->rng, seed);                          /* src/fix_langevin.cpp:86 (me = 0) */
  f->gfactor1 = calloc(s->ntypes + 1, sizeof(double)); f->gfactor2 = calloc(s->ntypes + 1, sizeof(double));
  f->ratio = calloc(s->ntypes + 1, sizeof(double));
  for (int t = 0; t <= s->ntypes; t++) f->ratio[t] = 1.0;   /* :96 */
}
/* optional keywords of the most recently defined fix langevin: `scale itype ratio` (:135-141), `zero yes|no` (:148-153) */
void leo_langevin_scale(leo_t *s, int itype, double ratio) {
  for (int k = s->nfix - 1; k >= 0; k--) if (s->fix[k].kind == FIX_LANGEVIN) { s->fix[k].ratio[itype] = ratio; return; }
}
void leo_langevin_zero(leo_t *s, int flag) {
  for (int k = s->nfix - 1; k >= 0; k--) if (s->fix[k].kind == FIX_LANGEVIN) { s->fix[k].zeroflag = flag; return; }
}
void leo_fix_extrusion(leo_t *s, int nevery, int neutral, int l, int r, double tp, int btype, int lr) {
  leo_fix *f = &s->fix[s->nfix++]; memset(f, 0, sizeof *f);
  f->kind = FIX_EXTRUSION; f->phase = 1; f->nevery = nevery; f->neutral = neutral; f->ctcf_left = l; f->ctcf_right = r;
  f->through_prob = tp; f->btype = btype; f->ctcf_lr = lr; f->next_reneighbor = -1;
  leo_ranmars_init(&f->rng, 12345);                          /* src/USER-LE/fix_extrusion.cpp:98-99 */
}
void leo_fix_ex_load(leo_t *s, int nevery, int it, int jt, double cutoff, int btype, int imax, int inew,
                     int jmax, int jnew, double fraction, int seed) {
  leo_fix *f = &s->fix[s->nfix++]; memset(f, 0, sizeof *f);
  f->kind = FIX_EX_LOAD; f->phase = 3; f->nevery = nevery; f->iatomtype = it; f->jatomtype = jt; f->cutsq = cutoff * cutoff;
  f->btype = btype; f->imaxbond = imax; f->inewtype = inew; f->jmaxbond = jmax; f->jnewtype = jnew;
  f->fraction = fraction; f->seed = seed; f->next_reneighbor = -1;
  leo_ranmars_init(&f->rng, seed);                           /* src/USER-LE/fix_ex_load.cpp:137 */
}
void leo_fix_ex_unload(leo_t *s, int nevery, int btype, double cutoff, double fraction, int seed) {
  leo_fix *f = &s->fix[s->nfix++]; memset(f, 0, sizeof *f);
  f->kind = FIX_EX_UNLOAD; f->phase = 2; f->nevery = nevery; f->btype = btype; f->cutsq = cutoff * cutoff;
  f->fraction = fraction; f->seed = seed; f->next_reneighbor = -1;
  leo_ranmars_init(&f->rng, seed);                           /* src/USER-LE/fix_ex_unload.cpp:88 */
}
/* stock fix bond/create (src/MC/fix_bond_create.cpp): the parent of ex_load; same arguments */
void leo_fix_bond_create(leo_t *s, int nevery, int it, int jt, double cutoff, int btype, int imax, int inew,
                         int jmax, int jnew, double fraction, int seed) {
  leo_fix_ex_load(s, nevery, it, jt, cutoff, btype, imax, inew, jmax, jnew, fraction, seed);
  leo_fix *f = &s->fix[s->nfix - 1];
  f->kind = FIX_BOND_CREATE; f->phase = 0;
  f->bc_tag = calloc((size_t)s->n + 2, sizeof(int)); f->counted = 0;
}
/* stock fix bond/break: src/MC/fix_bond_break.cpp is the text of fix_ex_unload.cpp but for the firing step (:178) */
void leo_fix_bond_break(leo_t *s, int nevery, int btype, double cutoff, double fraction, int seed) {
  leo_fix_ex_unload(s, nevery, btype, cutoff, fraction, seed);
  s->fix[s->nfix - 1].phase = 0;
}

/* ===================== domain ===================== */
/* src/domain.cpp:528-645 Domain::pbc (orthogonal, fully periodic) */
static void pbc(leo_t *s) {
  for (int i = 0; i < s->n; i++)
    for (int d = 0; d < 3; d++) {
      double *x = &s->x[3 * i + d];
      if (*x < s->lo[d]) { *x += s->prd[d]; s->img[3 * i + d]--; }
      if (*x >= s->hi[d]) { *x -= s->prd[d]; if (*x < s->lo[d]) *x = s->lo[d]; s->img[3 * i + d]++; }
    }
}
static inline void minimg(const leo_t *s, double *d, signed char *sh) {
  for (int k = 0; k < 3; k++) {
    double h = 0.5 * s->prd[k]; sh[k] = 0;
    /* src/domain.cpp closest_image picks the nearest stored image */
    if (d[k] > h) { d[k] -= s->prd[k]; sh[k] = 1; }       /* partner image shifted by +prd */
    else if (d[k] < -h) { d[k] += s->prd[k]; sh[k] = -1; }
  }
}

/* src/atom.cpp:2003-2094 Atom::sort + :2100-2208 setup_sort_bins (1 rank: sub-domain = box) */
static void swap_perm_d(double *a, const int *perm, int n, int w) {
  double *t = malloc((size_t)n * w * sizeof(double));
  for (int i = 0; i < n; i++) memcpy(t + (size_t)i * w, a + (size_t)perm[i] * w, w * sizeof(double));
  memcpy(a, t, (size_t)n * w * sizeof(double)); free(t);
}
static void swap_perm_i(int *a, const int *perm, int n, int w) {
  int *t = malloc((size_t)n * w * sizeof(int));
  for (int i = 0; i < n; i++) memcpy(t + (size_t)i * w, a + (size_t)perm[i] * w, w * sizeof(int));
  memcpy(a, t, (size_t)n * w * sizeof(int)); free(t);
}
static void atom_sort(leo_t *s) {
  s->nextsort = (s->ntimestep / s->sortfreq) * s->sortfreq + s->sortfreq;
  double binsize = 0.5 * s->cutneighmax;
  if (binsize == 0.0) return;
  double bininv = 1.0 / binsize;
  int nb[3]; double binv[3];
  for (int d = 0; d < 3; d++) {
    nb[d] = (int)((s->hi[d] - s->lo[d]) * bininv); if (nb[d] == 0) nb[d] = 1;
    binv[d] = nb[d] / (s->hi[d] - s->lo[d]);
  }
  long nbins = (long)nb[0] * nb[1] * nb[2];
  if (nbins == 1) return;
  int n = s->n;
  int *binhead = malloc(nbins * sizeof(int)), *next = malloc(n * sizeof(int)), *perm = malloc(n * sizeof(int));
  for (long b = 0; b < nbins; b++) binhead[b] = -1;
  for (int i = n - 1; i >= 0; i--) {
    int c[3];
    for (int d = 0; d < 3; d++) {
      c[d] = (int)((s->x[3 * i + d] - s->lo[d]) * binv[d]);
      if (c[d] < 0) c[d] = 0; if (c[d] > nb[d] - 1) c[d] = nb[d] - 1;
    }
    long ibin = (long)c[2] * nb[1] * nb[0] + (long)c[1] * nb[0] + c[0];
    next[i] = binhead[ibin]; binhead[ibin] = i;
  }
  int k = 0;
  for (long m = 0; m < nbins; m++) for (int i = binhead[m]; i >= 0; i = next[i]) perm[k++] = i;
  swap_perm_i(s->tag, perm, n, 1); swap_perm_i(s->type, perm, n, 1); swap_perm_i(s->img, perm, n, 3);
  swap_perm_d(s->x, perm, n, 3); swap_perm_d(s->v, perm, n, 3); swap_perm_d(s->f, perm, n, 3);
  swap_perm_i(s->num_bond, perm, n, 1); swap_perm_i(s->bond_type, perm, n, s->bpa); swap_perm_i(s->bond_atom, perm, n, s->bpa);
  swap_perm_i(s->nspecial, perm, n, 3); swap_perm_i(s->special, perm, n, s->maxspecial);
  if (s->apa) {
    swap_perm_i(s->num_angle, perm, n, 1); swap_perm_i(s->angle_type, perm, n, s->apa);
    swap_perm_i(s->angle_a1, perm, n, s->apa); swap_perm_i(s->angle_a2, perm, n, s->apa); swap_perm_i(s->angle_a3, perm, n, s->apa);
  }
  for (int i = 0; i < n; i++) s->map[s->tag[i]] = i;
  free(binhead); free(next); free(perm);
}

/* ===================== neighbor ===================== */
/* src/npair.h:112-136 find_special + flags src/neighbor.cpp:360-376 */
static inline int find_special(const leo_t *s, int i, int tagj) {
  const int *list = s->special + (size_t)i * s->maxspecial;
  int n1 = s->nspecial[3 * i], n2 = s->nspecial[3 * i + 1], n3 = s->nspecial[3 * i + 2];
  for (int k = 0; k < n3; k++)
    if (list[k] == tagj) {
      int lev = (k < n1) ? 1 : (k < n2) ? 2 : 3;
      double w = s->special_lj[lev], c = s->special_coul[lev];
      if (w == 0.0 && c == 0.0) return -1;      /* special_flag 0: pair dropped from the list */
      if (w == 1.0 && c == 1.0) return 0;       /* special_flag 1: ordinary entry */
      return lev;                               /* special_flag 2: entry carries the level, factor_lj = special_lj[lev] */
    }
  return 0;
}
/* list entry: index | ONESIDE << 29 | which << 30.  ONESIDE: an owned-ghost entry of a newton_pair-off list - a pair that
   interacts through a periodic image is stored by BOTH owned ends (each with the ghost image of the other, its own special
   status, force and half the energy / virial to the owned end only: pair_lj_cut.cpp:118-122, pair.cpp ev_tally) */
#define PJ_INDEX(e) ((int)((unsigned)(e) & 0x1FFFFFFFu))
#define PJ_ONESIDE(e) ((int)(((unsigned)(e) >> 29) & 1u))
#define PJ_WHICH(e) ((int)(((unsigned)(e) >> 30) & 3u))
static void push_pair(leo_t *s, int i, int j, int which, const signed char *sh, int oneside) {
  if (s->npairs == s->maxpairs) {
    s->maxpairs = s->maxpairs ? 2 * s->maxpairs : 1024;
    s->pj = realloc(s->pj, s->maxpairs * sizeof(int));
    if (s->brute) { s->pi_ = realloc(s->pi_, s->maxpairs * sizeof(int)); s->pshift = realloc(s->pshift, 3 * s->maxpairs); }
  }
  s->pj[s->npairs] = (int)((unsigned)j | ((unsigned)oneside << 29) | ((unsigned)which << 30));
  if (oneside) s->noneside++;
  if (s->brute) { s->pi_[s->npairs] = i; memcpy(s->pshift + 3 * s->npairs, sh, 3); }
  s->npairs++;
}
/* src/ntopo_bond_all.cpp:39-86 with newton_bond off: listed from i if i < atom1, where atom1 is the
   closest stored image (a ghost index >= nlocal when the bond straddles a periodic face) */
static void build_bondlist(leo_t *s) {
  s->nbondlist = 0;
  for (int i = 0; i < s->n; i++)
    for (int m = 0; m < s->num_bond[i]; m++) {
      int j = map_(s, s->bond_atom[i * s->bpa + m]);
      if (j < 0) { seterr(s, "Bond atoms missing"); return; }
      double d[3] = { s->x[3 * i] - s->x[3 * j], s->x[3 * i + 1] - s->x[3 * j + 1], s->x[3 * i + 2] - s->x[3 * j + 2] };
      signed char sh[3]; minimg(s, d, sh);
      int ghost = sh[0] || sh[1] || sh[2];
      if (ghost || i < j) {
        if (s->nbondlist == s->maxbondlist) {
          s->maxbondlist = s->maxbondlist ? 2 * s->maxbondlist : 1024;
          s->bl_i = realloc(s->bl_i, s->maxbondlist * sizeof(int)); s->bl_j = realloc(s->bl_j, s->maxbondlist * sizeof(int));
          s->bl_t = realloc(s->bl_t, s->maxbondlist * sizeof(int)); s->bl_s = realloc(s->bl_s, 3 * s->maxbondlist);
        }
        int k = s->nbondlist++;
        s->bl_i[k] = i; s->bl_j[k] = j; s->bl_t[k] = s->bond_type[i * s->bpa + m]; memcpy(s->bl_s + 3 * k, sh, 3);
      }
    }
}
/* src/ntopo_angle_all.cpp:37-93 with newton_bond off: the copy stored on atom i is listed if i <= every one of its three
   atoms, each taken as the image closest to i (Domain::closest_image; a periodic image is a ghost, index >= nlocal, so it
   never blocks the listing): an angle inside the box is listed once, from its lowest local index; one that straddles a
   face is listed from every atom that sees the others as images, and each listing only moves its owned atoms. */
static void build_anglelist(leo_t *s) {
  s->nanglelist = 0;
  if (!s->apa || !s->num_angle) return;
  for (int i = 0; i < s->n; i++)
    for (int m = 0; m < s->num_angle[i]; m++) {
      const int tg[3] = { s->angle_a1[(size_t)i * s->apa + m], s->angle_a2[(size_t)i * s->apa + m], s->angle_a3[(size_t)i * s->apa + m] };
      int idx[3]; signed char sh[9]; int listed = 1;
      for (int q = 0; q < 3; q++) {
        idx[q] = map_(s, tg[q]);
        if (idx[q] < 0) { seterr(s, "Angle atoms missing"); return; }
        double d[3] = { s->x[3 * i] - s->x[3 * idx[q]], s->x[3 * i + 1] - s->x[3 * idx[q] + 1], s->x[3 * i + 2] - s->x[3 * idx[q] + 2] };
        minimg(s, d, sh + 3 * q);
        int ghost = sh[3 * q] || sh[3 * q + 1] || sh[3 * q + 2];
        if (!ghost && idx[q] < i) listed = 0;
      }
      if (!listed) continue;
      if (s->nanglelist == s->maxanglelist) {
        s->maxanglelist = s->maxanglelist ? 2 * s->maxanglelist : 1024;
        s->al_i = realloc(s->al_i, 4 * (size_t)s->maxanglelist * sizeof(int)); s->al_s = realloc(s->al_s, 9 * (size_t)s->maxanglelist);
      }
      int k = s->nanglelist++;
      s->al_i[4 * k] = idx[0]; s->al_i[4 * k + 1] = idx[1]; s->al_i[4 * k + 2] = idx[2]; s->al_i[4 * k + 3] = s->angle_type[(size_t)i * s->apa + m];
      memcpy(s->al_s + 9 * k, sh, 9);
    }
}
/* src/MOLECULE/angle_harmonic.cpp:53-147, angle_cosine.cpp:49-121; tallies as Angle::ev_tally (src/angle.cpp:164-250) with
   newton_bond off: a third of the energy / virial per OWNED atom of the listing */
#define ANGLE_SMALL 0.001
static int angle_compute(leo_t *s, int eflag) {
  if (eflag) { s->eangle = 0.0; memset(s->vangle, 0, sizeof s->vangle); }
  const double *x = s->x; double *f = s->f;
  for (int n = 0; n < s->nanglelist; n++) {
    const int i1 = s->al_i[4 * n], i2 = s->al_i[4 * n + 1], i3 = s->al_i[4 * n + 2], type = s->al_i[4 * n + 3];
    const signed char *sh = s->al_s + 9 * n;
    if (type <= 0 || s->astyle[type] == 0) continue;
    double p1[3], p2[3], p3[3];
    for (int d = 0; d < 3; d++) {
      p1[d] = x[3 * i1 + d] + sh[d] * s->prd[d]; p2[d] = x[3 * i2 + d] + sh[3 + d] * s->prd[d]; p3[d] = x[3 * i3 + d] + sh[6 + d] * s->prd[d];
    }
    const int own1 = !(sh[0] || sh[1] || sh[2]), own2 = !(sh[3] || sh[4] || sh[5]), own3 = !(sh[6] || sh[7] || sh[8]);
    double delx1 = p1[0] - p2[0], dely1 = p1[1] - p2[1], delz1 = p1[2] - p2[2];
    double rsq1 = delx1 * delx1 + dely1 * dely1 + delz1 * delz1, r1 = sqrt(rsq1);
    double delx2 = p3[0] - p2[0], dely2 = p3[1] - p2[1], delz2 = p3[2] - p2[2];
    double rsq2 = delx2 * delx2 + dely2 * dely2 + delz2 * delz2, r2 = sqrt(rsq2);
    double c = delx1 * delx2 + dely1 * dely2 + delz1 * delz2;
    c /= r1 * r2;
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    double a, eangle = 0.0;
    if (s->astyle[type] == 1) {
      double sn = sqrt(1.0 - c * c);
      if (sn < ANGLE_SMALL) sn = ANGLE_SMALL;
      sn = 1.0 / sn;
      double dtheta = acos(c) - s->atheta0[type], tk = s->ak[type] * dtheta;
      if (eflag) eangle = tk * dtheta;
      a = -2.0 * tk * sn;
    } else {
      if (eflag) eangle = s->ak[type] * (1.0 + c);
      a = s->ak[type];
    }
    double a11 = a * c / rsq1, a12 = -a / (r1 * r2), a22 = a * c / rsq2;
    double f1[3] = { a11 * delx1 + a12 * delx2, a11 * dely1 + a12 * dely2, a11 * delz1 + a12 * delz2 };
    double f3[3] = { a22 * delx2 + a12 * delx1, a22 * dely2 + a12 * dely1, a22 * delz2 + a12 * delz1 };
    if (own1) { f[3 * i1] += f1[0]; f[3 * i1 + 1] += f1[1]; f[3 * i1 + 2] += f1[2]; }
    if (own2) { f[3 * i2] -= f1[0] + f3[0]; f[3 * i2 + 1] -= f1[1] + f3[1]; f[3 * i2 + 2] -= f1[2] + f3[2]; }
    if (own3) { f[3 * i3] += f3[0]; f[3 * i3 + 1] += f3[1]; f[3 * i3 + 2] += f3[2]; }
    if (eflag) {
      const double third = 1.0 / 3.0;
      double v[6] = { delx1 * f1[0] + delx2 * f3[0], dely1 * f1[1] + dely2 * f3[1], delz1 * f1[2] + delz2 * f3[2],
                      delx1 * f1[1] + delx2 * f3[1], delx1 * f1[2] + delx2 * f3[2], dely1 * f1[2] + dely2 * f3[2] };
      for (int o = 0; o < 3; o++) {
        if (!(o == 0 ? own1 : o == 1 ? own2 : own3)) continue;
        s->eangle += third * eangle;
        for (int q = 0; q < 6; q++) s->vangle[q] += third * v[q];
      }
    }
  }
  return 0;
}
/* FixExLoad::create_angles (fix_ex_load.cpp:855-954, newton_bond off) for local atom m; `created`: this firing's new bonds */
static int create_angles(leo_t *s, leo_fix *fx, int m, int ncreate, const int *created, long *nangles) {
  const int apa = s->apa, ms = s->maxspecial;
  int num = s->num_angle[m];
  int *at = s->angle_type + (size_t)m * apa, *a1 = s->angle_a1 + (size_t)m * apa, *a2 = s->angle_a2 + (size_t)m * apa, *a3 = s->angle_a3 + (size_t)m * apa;
  int overflow = 0;
#define NEWBOND(I1, I2, I3) ({ int n_; for (n_ = 0; n_ < ncreate; n_++) { \
    if (created[2 * n_] == (I1) && created[2 * n_ + 1] == (I2)) break; if (created[2 * n_] == (I2) && created[2 * n_ + 1] == (I1)) break; \
    if (created[2 * n_] == (I2) && created[2 * n_ + 1] == (I3)) break; if (created[2 * n_] == (I3) && created[2 * n_ + 1] == (I2)) break; } n_ < ncreate; })
  /* atom M central: pairs of its 1-2 neighbours */
  int i2 = s->tag[m], n2 = s->nspecial[3 * m];
  const int *s2list = s->special + (size_t)m * ms;
  for (int i = 0; i < n2; i++)
    for (int j = i + 1; j < n2; j++) {
      int i1 = s2list[i], i3 = s2list[j];
      if (!NEWBOND(i1, i2, i3)) continue;
      if (num < apa) { at[num] = fx->atype; a1[num] = i1; a2[num] = i2; a3[num] = i3; num++; (*nangles)++; }
      else overflow = 1;
    }
  /* atom M as atom 1 of the angle */
  int i1 = s->tag[m], n1 = s->nspecial[3 * m];
  const int *s1list = s->special + (size_t)m * ms;
  for (int i = 0; i < n1; i++) {
    i2 = s1list[i];
    int i2local = map_(s, i2);
    if (i2local < 0) return seterr(s, "Fix ex_load needs ghost atoms from further away");
    const int *sl2 = s->special + (size_t)i2local * ms;
    n2 = s->nspecial[3 * i2local];
    for (int j = 0; j < n2; j++) {
      int i3 = sl2[j];
      if (i3 == i1) continue;
      if (!NEWBOND(i1, i2, i3)) continue;
      if (num < apa) { at[num] = fx->atype; a1[num] = i1; a2[num] = i2; a3[num] = i3; num++; (*nangles)++; }
      else overflow = 1;
    }
  }
#undef NEWBOND
  s->num_angle[m] = num;
  return overflow ? seterr(s, "Fix ex_load induced too many angles/dihedrals/impropers per atom") : 0;
}
/* FixExUnload::break_angles (fix_ex_unload.cpp:551-582) */
static void break_angles(leo_t *s, int m, int id1, int id2, long *nangles) {
  const int apa = s->apa;
  int num = s->num_angle[m];
  int *at = s->angle_type + (size_t)m * apa, *a1 = s->angle_a1 + (size_t)m * apa, *a2 = s->angle_a2 + (size_t)m * apa, *a3 = s->angle_a3 + (size_t)m * apa;
  int i = 0;
  while (i < num) {
    int found = 0;
    if (a1[i] == id1 && a2[i] == id2) found = 1;
    else if (a2[i] == id1 && a3[i] == id2) found = 1;
    else if (a1[i] == id2 && a2[i] == id1) found = 1;
    else if (a2[i] == id2 && a3[i] == id1) found = 1;
    if (!found) i++;
    else {
      for (int j = i; j < num - 1; j++) { at[j] = at[j + 1]; a1[j] = a1[j + 1]; a2[j] = a2[j + 1]; a3[j] = a3[j + 1]; }
      num--; (*nangles)++;
    }
  }
  s->num_angle[m] = num;
}

/* Which end of an owned-owned pair stores it in the half list (positions = those of the build, s->xhold).
   newton_pair off, npair_half_bin_newtoff.cpp:90: the lower local index.
   newton_pair on,  npair_half_bin_newton.cpp:84-149: atoms of one bin -> the one earlier in the bin's list (= lower local
   index, bins are filled in reverse order: nbin_standard.cpp:192-232); different bins -> the pair is found from the atom
   whose bin precedes the other's in (z, y, x) order, because the stencil holds only the "upper half" offsets
   (nstencil_half_bin_3d_newton.cpp: k > 0 || j > 0 || (j == 0 && i > 0)).  Bins: nbin_standard.cpp:53-186 with
   binsize_optimal = cutneighmax / 2, coord2bin of an owned atom src/nbin.cpp:120-152. */
static void ref_bin(const leo_t *s, const double *x, int *b) {
  double binsizeinv = 1.0 / (0.5 * s->cutneighmax);
  for (int d = 0; d < 3; d++) {
    int nb = (int)(s->prd[d] * binsizeinv); if (nb == 0) nb = 1;
    double bininv = 1.0 / (s->prd[d] / nb);
    int i = (int)((x[d] - s->lo[d]) * bininv); if (i > nb - 1) i = nb - 1;
    b[d] = i;
  }
}
/* NBin::coord2bin for any coordinate (src/nbin.cpp:120-152): a ghost image lies outside the box, in the bins beyond it */
static void ref_bin_any(const leo_t *s, const double *x, int *b) {
  double binsizeinv = 1.0 / (0.5 * s->cutneighmax);
  for (int d = 0; d < 3; d++) {
    int nb = (int)(s->prd[d] * binsizeinv); if (nb == 0) nb = 1;
    double bininv = 1.0 / (s->prd[d] / nb), hi = s->lo[d] + s->prd[d];
    if (x[d] >= hi) b[d] = (int)((x[d] - hi) * bininv) + nb;
    else if (x[d] >= s->lo[d]) { int i = (int)((x[d] - s->lo[d]) * bininv); b[d] = i > nb - 1 ? nb - 1 : i; }
    else b[d] = (int)((x[d] - s->lo[d]) * bininv) - 1;
  }
}
/* newton_pair on, owned i and the GHOST image of j at x_j + sh * prd: i stores the pair iff the ghost's bin is one of the
   upper-half stencil bins of i's (npair_half_bin_newton.cpp:120-149), or - same bin - the ghost is not below / behind /
   left of i (:85-91); otherwise the owner of j stores it with the ghost image of i */
static int stores_image_pair(const leo_t *s, int i, int j, const signed char *sh) {
  double g[3]; int bi[3], bj[3];
  for (int d = 0; d < 3; d++) g[d] = s->xhold[3 * j + d] + sh[d] * s->prd[d];
  ref_bin_any(s, s->xhold + 3 * i, bi); ref_bin_any(s, g, bj);
  for (int d = 2; d >= 0; d--) if (bi[d] != bj[d]) return bi[d] < bj[d];
  const double *xi = s->xhold + 3 * i;
  if (g[2] < xi[2]) return 0;
  if (g[2] == xi[2]) { if (g[1] < xi[1]) return 0; if (g[1] == xi[1] && g[0] < xi[0]) return 0; }
  return 1;
}
static int stores_pair(const leo_t *s, int i, int j) {          /* 1: i is the storing end of (i, j) */
  if (!s->newton_pair) return i < j;
  int bi[3], bj[3];
  ref_bin(s, s->xhold + 3 * i, bi); ref_bin(s, s->xhold + 3 * j, bj);
  for (int d = 2; d >= 0; d--) if (bi[d] != bj[d]) return bi[d] < bj[d];
  return i < j;
}
/* src/neighbor.cpp:2022-2101 build: xhold, half list (all owned; stored under the end stores_pair names), topology */
static void neigh_build(leo_t *s) {
  int n = s->n;
  s->ago = 0; s->nbuilds++;
  memcpy(s->xhold, s->x, 3 * n * sizeof(double));
  s->npairs = 0; s->noneside = 0;
  double cutneighsq = s->cutneighmax * s->cutneighmax;
  int nc[3]; s->brute = 0;
  if (!s->pair_on || s->cutneighmax <= 0.0) { build_bondlist(s); build_anglelist(s); return; }
  for (int d = 0; d < 3; d++) {
    nc[d] = (int)(s->prd[d] / s->cutneighmax);
    if (nc[d] < 3) s->brute = 1;
  }
  if (s->brute) {
    /* small periodic boxes (unit-test systems): enumerate the 27 images explicitly, like ghost atoms */
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        for (int sx = -1; sx <= 1; sx++) for (int sy = -1; sy <= 1; sy++) for (int sz = -1; sz <= 1; sz++) {
          if (j == i) continue;
          signed char sh[3] = { (signed char)sx, (signed char)sy, (signed char)sz };
          const int image = sx || sy || sz;
          int oneside = 0;
          if (!image) { if (!stores_pair(s, i, j)) continue; }
          else if (!s->newton_pair) oneside = 1;
          else if (!stores_image_pair(s, i, j, sh)) continue;
          double dx = s->x[3 * i] - (s->x[3 * j] + sx * s->prd[0]);
          double dy = s->x[3 * i + 1] - (s->x[3 * j + 1] + sy * s->prd[1]);
          double dz = s->x[3 * i + 2] - (s->x[3 * j + 2] + sz * s->prd[2]);
          double rsq = dx * dx + dy * dy + dz * dz;
          if (rsq > cutneighsq) continue;
          int which = find_special(s, i, s->tag[j]);
          /* src/domain.h:156-161 minimum_image_check */
          if (which != 0 && (fabs(dx) > 0.5 * s->prd[0] || fabs(dy) > 0.5 * s->prd[1] || fabs(dz) > 0.5 * s->prd[2])) which = 0;
          if (which < 0) continue;
          push_pair(s, i, j, which, sh, oneside);
        }
    build_bondlist(s); build_anglelist(s);
    return;
  }
  /* cell list, cells >= cutneigh, 27-cell stencil, minimum image (box >= 3 cells per dim) */
  double cinv[3]; for (int d = 0; d < 3; d++) cinv[d] = nc[d] / s->prd[d];
  long ncell = (long)nc[0] * nc[1] * nc[2];
  int *head = malloc((ncell + 1) * sizeof(int)), *cellof = malloc(n * sizeof(int)), *order = malloc(n * sizeof(int));
  memset(head, 0, (ncell + 1) * sizeof(int));
  for (int i = 0; i < n; i++) {
    int c[3];
    for (int d = 0; d < 3; d++) {
      c[d] = (int)((s->x[3 * i + d] - s->lo[d]) * cinv[d]);
      if (c[d] < 0) c[d] = 0; if (c[d] >= nc[d]) c[d] = nc[d] - 1;
    }
    cellof[i] = (c[2] * nc[1] + c[1]) * nc[0] + c[0];
    head[cellof[i] + 1]++;
  }
  for (long c = 0; c < ncell; c++) head[c + 1] += head[c];
  int *fill = malloc(ncell * sizeof(int)); memcpy(fill, head, ncell * sizeof(int));
  for (int i = 0; i < n; i++) order[fill[cellof[i]]++] = i;
  free(fill);
  s->firstneigh = realloc(s->firstneigh, (n + 1) * sizeof(int));
  for (int i = 0; i < n; i++) {
    s->firstneigh[i] = (int)s->npairs;
    int ci = cellof[i];
    int cx = ci % nc[0], cy = (ci / nc[0]) % nc[1], cz = ci / (nc[0] * nc[1]);
    double xi = s->x[3 * i], yi = s->x[3 * i + 1], zi = s->x[3 * i + 2];
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
      int ax = (cx + dx + nc[0]) % nc[0], ay = (cy + dy + nc[1]) % nc[1], az = (cz + dz + nc[2]) % nc[2];
      int c = (az * nc[1] + ay) * nc[0] + ax;
      for (int p = head[c]; p < head[c + 1]; p++) {
        int j = order[p];
        if (j == i) continue;
        double d[3] = { xi - s->x[3 * j], yi - s->x[3 * j + 1], zi - s->x[3 * j + 2] };
        signed char sh[3]; minimg(s, d, sh);
        /* who stores the pair: an owned-owned pair one end (stores_pair); a pair through a periodic image is an owned-ghost
           pair - both ends with newton_pair off, the end whose stencil reaches the ghost with newton_pair on */
        int oneside = 0;
        if (!(sh[0] || sh[1] || sh[2])) { if (!stores_pair(s, i, j)) continue; }
        else if (!s->newton_pair) oneside = 1;
        else if (!stores_image_pair(s, i, j, sh)) continue;
        double rsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (rsq > cutneighsq) continue;
        int which = find_special(s, i, s->tag[j]);
        if (which < 0) continue;
        push_pair(s, i, j, which, sh, oneside);
      }
    }
  }
  s->firstneigh[n] = (int)s->npairs;
  free(head); free(cellof); free(order);
  build_bondlist(s); build_anglelist(s);
}

/* src/neighbor.cpp:1933-1948 decide + :1962-2014 check_distance */
static int neigh_decide(leo_t *s) {
  for (int k = 0; k < s->nfix; k++)
    if (s->fix[k].kind >= FIX_EXTRUSION && s->ntimestep == s->fix[k].next_reneighbor) return 1;
  s->ago++;
  if (s->ago >= s->delay && s->ago % s->every == 0) {
    if (!s->check) return 1;
    int flag = 0;
    for (int i = 0; i < s->n; i++) {
      double dx = s->x[3 * i] - s->xhold[3 * i], dy = s->x[3 * i + 1] - s->xhold[3 * i + 1], dz = s->x[3 * i + 2] - s->xhold[3 * i + 2];
      if (dx * dx + dy * dy + dz * dz > s->triggersq) flag = 1;
    }
    if (flag && s->ago == (s->every > s->delay ? s->every : s->delay)) s->ndanger++;
    return flag;
  }
  return 0;
}

/* ===================== forces ===================== */
/* src/pair_lj_cut.cpp:68-140 ; energy/virial src/pair.cpp:920- ev_tally (totals only) */
static void pair_compute(leo_t *s, int eflag) {
  int nt = s->ntypes + 1;
  if (eflag) { s->evdwl = 0.0; memset(s->vpair, 0, sizeof s->vpair); }
  if (!s->pair_on) return;
  const double *x = s->x; double *f = s->f;
  if (s->brute) {
    for (long p = 0; p < s->npairs; p++) {
      int i = s->pi_[p], j = s->pj[p]; double factor = s->special_lj[PJ_WHICH(j)]; const int one = PJ_ONESIDE(j); j = PJ_INDEX(j);
      const double w = one ? 0.5 : 1.0;
      const signed char *sh = s->pshift + 3 * p;
      double dx = x[3 * i] - (x[3 * j] + sh[0] * s->prd[0]);
      double dy = x[3 * i + 1] - (x[3 * j + 1] + sh[1] * s->prd[1]);
      double dz = x[3 * i + 2] - (x[3 * j + 2] + sh[2] * s->prd[2]);
      double rsq = dx * dx + dy * dy + dz * dz;
      int ij = s->type[i] * nt + s->type[j];
      if (rsq < s->cutsq[ij]) {
        double r2inv = 1.0 / rsq, r6inv = r2inv * r2inv * r2inv;
        double forcelj = r6inv * (s->lj1[ij] * r6inv - s->lj2[ij]);
        double fpair = factor * forcelj * r2inv;
        f[3 * i] += dx * fpair; f[3 * i + 1] += dy * fpair; f[3 * i + 2] += dz * fpair;
        if (!one) { f[3 * j] -= dx * fpair; f[3 * j + 1] -= dy * fpair; f[3 * j + 2] -= dz * fpair; }
        if (eflag) {
          double e = r6inv * (s->lj3[ij] * r6inv - s->lj4[ij]) - s->offset[ij];
          s->evdwl += w * factor * e;
          s->vpair[0] += w * dx * dx * fpair; s->vpair[1] += w * dy * dy * fpair; s->vpair[2] += w * dz * dz * fpair;
          s->vpair[3] += w * dx * dy * fpair; s->vpair[4] += w * dx * dz * fpair; s->vpair[5] += w * dy * dz * fpair;
        }
      }
    }
    return;
  }
  double hx = 0.5 * s->prd[0], hy = 0.5 * s->prd[1], hz = 0.5 * s->prd[2];
  for (int i = 0; i < s->n; i++) {
    double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    int ti = s->type[i];
    double fx = 0, fy = 0, fz = 0;
    for (int p = s->firstneigh[i]; p < s->firstneigh[i + 1]; p++) {
      int j = s->pj[p]; double factor = s->special_lj[PJ_WHICH(j)]; const int one = PJ_ONESIDE(j); j = PJ_INDEX(j);
      const double w = one ? 0.5 : 1.0;
      double dx = xi - x[3 * j], dy = yi - x[3 * j + 1], dz = zi - x[3 * j + 2];
      if (dx > hx) dx -= s->prd[0]; else if (dx < -hx) dx += s->prd[0];
      if (dy > hy) dy -= s->prd[1]; else if (dy < -hy) dy += s->prd[1];
      if (dz > hz) dz -= s->prd[2]; else if (dz < -hz) dz += s->prd[2];
      double rsq = dx * dx + dy * dy + dz * dz;
      int ij = ti * nt + s->type[j];
      if (rsq < s->cutsq[ij]) {
        double r2inv = 1.0 / rsq, r6inv = r2inv * r2inv * r2inv;
        double forcelj = r6inv * (s->lj1[ij] * r6inv - s->lj2[ij]);
        double fpair = factor * forcelj * r2inv;
        fx += dx * fpair; fy += dy * fpair; fz += dz * fpair;
        if (!one) { f[3 * j] -= dx * fpair; f[3 * j + 1] -= dy * fpair; f[3 * j + 2] -= dz * fpair; }
        if (eflag) {
          double e = r6inv * (s->lj3[ij] * r6inv - s->lj4[ij]) - s->offset[ij];
          s->evdwl += w * factor * e;
          s->vpair[0] += w * dx * dx * fpair; s->vpair[1] += w * dy * dy * fpair; s->vpair[2] += w * dz * dz * fpair;
          s->vpair[3] += w * dx * dy * fpair; s->vpair[4] += w * dx * dz * fpair; s->vpair[5] += w * dy * dz * fpair;
        }
      }
    }
    f[3 * i] += fx; f[3 * i + 1] += fy; f[3 * i + 2] += fz;
  }
}
/* src/MOLECULE/bond_fene.cpp:52-128, bond_harmonic.cpp:48-101, bond_morse.cpp:50-115, dispatch as bond_hybrid.cpp:66-152.
   A straddling bond is listed from both owned ends; each listing only updates its owned end
   (newton_bond off, i2 >= nlocal) and tallies half the energy/virial (src/bond.cpp ev_tally). */
static int bond_compute(leo_t *s, int eflag) {
  if (eflag) { s->ebond = 0.0; memset(s->vbond, 0, sizeof s->vbond); }
  const double *x = s->x; double *f = s->f;
  for (int k = 0; k < s->nbondlist; k++) {
    int i1 = s->bl_i[k], i2 = s->bl_j[k], type = s->bl_t[k];
    const signed char *sh = s->bl_s + 3 * k;
    int ghost = sh[0] || sh[1] || sh[2];
    if (type <= 0 || s->bstyle[type] == 0) continue;   /* bond_style zero / none */
    double dx = x[3 * i1] - (x[3 * i2] + sh[0] * s->prd[0]);
    double dy = x[3 * i1 + 1] - (x[3 * i2 + 1] + sh[1] * s->prd[1]);
    double dz = x[3 * i1 + 2] - (x[3 * i2 + 2] + sh[2] * s->prd[2]);
    double rsq = dx * dx + dy * dy + dz * dz, fbond, ebond = 0.0;
    if (s->bstyle[type] == 1) {
      double K = s->bp0[type], R0 = s->bp1[type], epsb = s->bp2[type], sigb = s->bp3[type];
      double r0sq = R0 * R0, rlogarg = 1.0 - rsq / r0sq, sr6 = 0.0;
      if (rlogarg < 0.1) {
        s->fene_warn++;
        if (rlogarg <= -3.0) return seterr(s, "Bad FENE bond");
        rlogarg = 0.1;
      }
      fbond = -K / rlogarg;
      if (rsq < TWO_1_3 * sigb * sigb) {
        double sr2 = sigb * sigb / rsq; sr6 = sr2 * sr2 * sr2;
        fbond += 48.0 * epsb * sr6 * (sr6 - 0.5) / rsq;
      }
      if (eflag) {
        ebond = -0.5 * K * r0sq * log(rlogarg);
        if (rsq < TWO_1_3 * sigb * sigb) ebond += 4.0 * epsb * sr6 * (sr6 - 1.0) + epsb;
      }
    } else if (s->bstyle[type] == 2) {
      double r = sqrt(rsq), dr = r - s->bp1[type], rk = s->bp0[type] * dr;
      fbond = (r > 0.0) ? -2.0 * rk / r : 0.0;
      if (eflag) ebond = rk * dr;
    } else if (s->bstyle[type] == 3) {                 /* src/MOLECULE/bond_morse.cpp:50-115: D alpha r0 */
      double r = sqrt(rsq), dr = r - s->bp2[type], ralpha = exp(-s->bp1[type] * dr);
      fbond = (r > 0.0) ? -2.0 * s->bp0[type] * s->bp1[type] * (1 - ralpha) * ralpha / r : 0.0;
      if (eflag) ebond = s->bp0[type] * (1 - ralpha) * (1 - ralpha);
    } else return seterr(s, "Bond coeffs not set");
    f[3 * i1] += dx * fbond; f[3 * i1 + 1] += dy * fbond; f[3 * i1 + 2] += dz * fbond;
    if (!ghost) { f[3 * i2] -= dx * fbond; f[3 * i2 + 1] -= dy * fbond; f[3 * i2 + 2] -= dz * fbond; }
    if (eflag) {
      double w = ghost ? 0.5 : 1.0;
      s->ebond += w * ebond;
      s->vbond[0] += w * dx * dx * fbond; s->vbond[1] += w * dy * dy * fbond; s->vbond[2] += w * dz * dz * fbond;
      s->vbond[3] += w * dx * dy * fbond; s->vbond[4] += w * dx * dz * fbond; s->vbond[5] += w * dy * dz * fbond;
    }
  }
  return 0;
}

/* ===================== fix nve / langevin ===================== */
/* src/fix_nve.cpp:64-104, :108-141 ; dtv = dt, dtf = 0.5*dt*ftm2v (:51-58) */
static void nve_initial(leo_t *s, const leo_fix *fx) {
  double dtv = s->dt, dtf = 0.5 * s->dt * s->ftm2v;
  for (int i = 0; i < s->n; i++) {
    if (fx->gmask && !fx->gmask[s->tag[i]]) continue;          /* group mask, fix_nve.cpp:82 */
    double dtfm = dtf / s->mass[s->type[i]];
    for (int d = 0; d < 3; d++) { s->v[3 * i + d] += dtfm * s->f[3 * i + d]; s->x[3 * i + d] += dtv * s->v[3 * i + d]; }
  }
}
static void nve_final(leo_t *s, const leo_fix *fx) {
  double dtf = 0.5 * s->dt * s->ftm2v;
  for (int i = 0; i < s->n; i++) {
    if (fx->gmask && !fx->gmask[s->tag[i]]) continue;
    double dtfm = dtf / s->mass[s->type[i]];
    for (int d = 0; d < 3; d++) s->v[3 * i + d] += dtfm * s->f[3 * i + d];
  }
}
/* src/fix_langevin.cpp:296-310 init gfactors */
static void langevin_init(leo_t *s, leo_fix *fx) {
  for (int t = 1; t <= s->ntypes; t++) {
    fx->gfactor1[t] = -s->mass[t] / fx->t_period / s->ftm2v;
    fx->gfactor2[t] = sqrt(s->mass[t]) * sqrt(24.0 * s->boltz / fx->t_period / s->dt / s->mvv2e) / s->ftm2v;
    fx->gfactor1[t] *= 1.0 / fx->ratio[t];
    fx->gfactor2[t] *= 1.0 / sqrt(fx->ratio[t]);
  }
}
/* src/fix_langevin.cpp:585-778 post_force_templated<0,0,0,0,0,Tp_ZERO>; compute_target :784-797 */
static void langevin_post_force(leo_t *s, leo_fix *fx) {
  double fsum[3] = {0.0, 0.0, 0.0};
  long count = 0;                                              /* group->count(igroup), :630-635 */
  if (fx->zeroflag) {
    long members = 0;
    for (int i = 0; i < s->n; i++) if (!fx->gmask || fx->gmask[s->tag[i]]) members++;
    if (members == 0) { seterr(s, "Cannot zero Langevin force of 0 atoms"); return; }
  }
  double delta = (double)(s->ntimestep - s->beginstep);
  if (delta != 0.0) delta /= (double)(s->endstep - s->beginstep);
  double t_target = fx->t_start + delta * (fx->t_stop - fx->t_start);
  fx->tsqrt = sqrt(t_target);
  for (int i = 0; i < s->n; i++) {
    if (fx->gmask && !fx->gmask[s->tag[i]]) continue;          /* mask[i] & groupbit, :661 */
    double gamma1 = fx->gfactor1[s->type[i]], gamma2 = fx->gfactor2[s->type[i]] * fx->tsqrt;
    double fran[3], fdrag[3];
    fran[0] = gamma2 * (leo_ranmars_uniform(&fx->rng) - 0.5);
    fran[1] = gamma2 * (leo_ranmars_uniform(&fx->rng) - 0.5);
    fran[2] = gamma2 * (leo_ranmars_uniform(&fx->rng) - 0.5);
    fdrag[0] = gamma1 * s->v[3 * i]; fdrag[1] = gamma1 * s->v[3 * i + 1]; fdrag[2] = gamma1 * s->v[3 * i + 2];
    s->f[3 * i] += fdrag[0] + fran[0]; s->f[3 * i + 1] += fdrag[1] + fran[1]; s->f[3 * i + 2] += fdrag[2] + fran[2];
    fsum[0] += fran[0]; fsum[1] += fran[1]; fsum[2] += fran[2];  /* Tp_ZERO, :725-729 */
    count++;
  }
  if (fx->zeroflag) {                                          /* set total force to zero, :752-772 */
    for (int k = 0; k < 3; k++) fsum[k] /= (double)count;
    for (int i = 0; i < s->n; i++) {
      if (fx->gmask && !fx->gmask[s->tag[i]]) continue;
      s->f[3 * i] -= fsum[0]; s->f[3 * i + 1] -= fsum[1]; s->f[3 * i + 2] -= fsum[2];
    }
  }
}

/* ===================== USER-LE helpers ===================== */
/* dedup: src/USER-LE/fix_extrusion.cpp:1116-1135 (identical copies in ex_load/ex_unload) */
static int le_dedup(int nstart, int nstop, int *copy) {
  int i, m = nstart;
  while (m < nstop) {
    for (i = 0; i < m; i++)
      if (copy[i] == copy[m]) { copy[m] = copy[nstop - 1]; nstop--; break; }
    if (i == m) m++;
  }
  return nstop;
}
/* rebuild_special_one: src/USER-LE/fix_extrusion.cpp:1045-1108 */
static int rebuild_special_one(leo_t *s, int m) {
  int ms = s->maxspecial; int *copy = s->copy;
  int *slist = s->special + (size_t)m * ms;
  int n1 = s->nspecial[3 * m], cn1 = 0, cn2, cn3;
  for (int i = 0; i < n1; i++) copy[cn1++] = slist[i];
  cn2 = cn1;
  for (int i = 0; i < cn1; i++) {
    int n = map_(s, copy[i]);
    if (n < 0) return seterr(s, "Fix bond/create needs ghost atoms from further away");
    const int *sl = s->special + (size_t)n * ms; int nn1 = s->nspecial[3 * n];
    for (int j = 0; j < nn1; j++) if (sl[j] != s->tag[m]) copy[cn2++] = sl[j];
  }
  cn2 = le_dedup(cn1, cn2, copy);
  if (cn2 > ms) return seterr(s, "Special list size exceeded in fix bond/create");
  cn3 = cn2;
  for (int i = cn1; i < cn2; i++) {
    int n = map_(s, copy[i]);
    if (n < 0) return seterr(s, "Fix bond/create needs ghost atoms from further away");
    const int *sl = s->special + (size_t)n * ms; int nn1 = s->nspecial[3 * n];
    for (int j = 0; j < nn1; j++) if (sl[j] != s->tag[m]) copy[cn3++] = sl[j];
  }
  cn3 = le_dedup(cn2, cn3, copy);
  if (cn3 > ms) return seterr(s, "Special list size exceeded in fix bond/create");
  s->nspecial[3 * m] = cn1; s->nspecial[3 * m + 1] = cn2; s->nspecial[3 * m + 2] = cn3;
  memcpy(slist, copy, cn3 * sizeof(int));
  return 0;
}
/* influence rule for broken bonds: fix_extrusion.cpp:940-969 / fix_ex_unload.cpp:417-483 */
/* angles: 1 = FixExUnload::update_topology (break_angles for every broken bond that influences the atom, :445-460; the
   angle count goes down by a third of the removed copies, :466-472); 0 = FixExtrusion::update_topology, which leaves angles alone */
static int topo_broken(leo_t *s, int nbreak, const int *broken, int angles) {
  long nang = 0;
  for (int i = 0; i < s->n; i++) {
    int influenced = 0; const int *slist = s->special + (size_t)i * s->maxspecial;
    for (int j = 0; j < nbreak && (angles || !influenced); j++) {
      int id1 = broken[2 * j], id2 = broken[2 * j + 1], influence = 0;
      if (s->tag[i] == id1 || s->tag[i] == id2) influence = 1;
      else {
        int n = s->nspecial[3 * i + 2], found = 0;
        for (int k = 0; k < n; k++) if (slist[k] == id1 || slist[k] == id2) found++;
        if (found == 2) influence = 1;
      }
      if (!influence) continue;
      influenced = 1;
      if (angles) break_angles(s, i, id1, id2, &nang);
    }
    if (influenced && rebuild_special_one(s, i)) return 1;
  }
  s->nangles -= nang / 3;
  return 0;
}
/* influence rule for created bonds: fix_extrusion.cpp:971-1001 / fix_ex_load.cpp:720-753 */
static int create_angles(leo_t *s, leo_fix *fx, int m, int ncreate, const int *created, long *nangles);
/* fx != NULL with fx->atype and an angle style: FixExLoad::update_topology creates the angles the new bonds induce */
static int topo_created(leo_t *s, int ncreate, const int *created, leo_fix *fx) {
  long nang = 0;
  const int angleflag = fx && fx->atype > 0 && s->apa > 0 && s->astyle;
  for (int i = 0; i < s->n; i++) {
    int influenced = 0; const int *slist = s->special + (size_t)i * s->maxspecial;
    for (int j = 0; j < ncreate && !influenced; j++) {
      int id1 = created[2 * j], id2 = created[2 * j + 1];
      if (s->tag[i] == id1 || s->tag[i] == id2) influenced = 1;
      else {
        int n = s->nspecial[3 * i + 1];
        for (int k = 0; k < n; k++) if (slist[k] == id1 || slist[k] == id2) { influenced = 1; break; }
      }
    }
    if (influenced && rebuild_special_one(s, i)) return 1;
    if (influenced && angleflag && create_angles(s, fx, i, ncreate, created, &nang)) return 1;
  }
  s->nangles += nang / 3;
  return 0;
}
/* delete bond to `partner` from atom i by shifting: fix_extrusion.cpp:656-668 */
static void delete_bond(leo_t *s, int i, int partner) {
  int b = s->bpa;
  for (int m = 0; m < s->num_bond[i]; m++)
    if (s->bond_atom[i * b + m] == partner) {
      for (int k = m; k < s->num_bond[i] - 1; k++) {
        s->bond_atom[i * b + k] = s->bond_atom[i * b + k + 1]; s->bond_type[i * b + k] = s->bond_type[i * b + k + 1];
      }
      s->num_bond[i]--; break;
    }
}
/* remove partner from 1-2 specials: fix_extrusion.cpp:673-683 */
static void special_remove12(leo_t *s, int i, int partner) {
  int *slist = s->special + (size_t)i * s->maxspecial;
  int n1 = s->nspecial[3 * i], m;
  for (m = 0; m < n1; m++) if (slist[m] == partner) break;
  int n3 = s->nspecial[3 * i + 2];
  for (; m < n3 - 1; m++) slist[m] = slist[m + 1];
  s->nspecial[3 * i]--; s->nspecial[3 * i + 1]--; s->nspecial[3 * i + 2]--;
}
/* insert partner as 1-2 special: fix_extrusion.cpp:748-771 */
static int special_insert12(leo_t *s, int i, int partner, const char *errmsg) {
  int *slist = s->special + (size_t)i * s->maxspecial;
  int n1 = s->nspecial[3 * i], n2 = s->nspecial[3 * i + 1], n3 = s->nspecial[3 * i + 2], m, n;
  for (m = n1; m < n3; m++) if (slist[m] == partner) break;
  if (m < n3) {
    for (n = m; n < n3 - 1; n++) slist[n] = slist[n + 1];
    n3--; if (m < n2) n2--;
  }
  if (n3 == s->maxspecial) return seterr(s, errmsg);
  for (m = n3; m > n1; m--) slist[m] = slist[m - 1];
  slist[n1] = partner;
  s->nspecial[3 * i] = n1 + 1; s->nspecial[3 * i + 1] = n2 + 1; s->nspecial[3 * i + 2] = n3 + 1;
  return 0;
}
static void recount_bondcount(leo_t *s, int btype, int *over) {
  for (int i = 0; i < s->n; i++) {
    s->bondcount[i] = 0;
    for (int j = 0; j < s->num_bond[i]; j++)
      if (s->bond_type[i * s->bpa + j] == btype) { s->bondcount[i]++; if (over && s->bondcount[i] > 1) *over = 1; }
  }
}

/* ===================== fix extrusion: src/USER-LE/fix_extrusion.cpp:256-872 ===================== */
static int ext_can(leo_t *s, leo_fix *fx, int X, int blk) {
  if (X < 0) return 0;   /* reference reads out of bounds; unreachable for valid chains */
  if (!(s->num_bond[X] - s->bondcount[X] == 2 && s->bondcount[X] == 0)) return 0;
  int t = s->type[X];
  if (!(t == fx->ctcf_left || t == fx->ctcf_right || t == fx->ctcf_lr || t == fx->neutral)) return 0;
  if (!(t != blk || fx->through_prob > leo_ranmars_uniform(&fx->rng))) return 0;
  if (!(t != fx->ctcf_lr || fx->through_prob > leo_ranmars_uniform(&fx->rng))) return 0;
  return 1;
}
static inline double d2raw(const leo_t *s, int a, int b) {
  double dx = s->x[3 * a] - s->x[3 * b], dy = s->x[3 * a + 1] - s->x[3 * b + 1], dz = s->x[3 * a + 2] - s->x[3 * b + 2];
  return dx * dx + dy * dy + dz * dz;
}
static int fire_extrusion(leo_t *s, leo_fix *fx) {
  int n = s->n, over = 0;
  int *to_remove = s->ia, *to_add = s->ib, *final_to_remove = s->ic, *final_to_add = s->id;
  double *dc = s->da;
  const int *tag = s->tag;
  fx->lastcount = 0;
  recount_bondcount(s, fx->btype, &over);
  if (over) return seterr(s, "Fix extrusion, more than one bond type 2");
  for (int i = 0; i < n; i++) { to_remove[i] = to_add[i] = final_to_remove[i] = final_to_add[i] = 0; dc[i] = BIG; }

  /* Phase 1 (:368-516): loop over the bond list of the last reneighbor */
  for (int k = 0; k < s->nbondlist; k++) {
    int i1 = s->bl_i[k], i2 = s->bl_j[k];
    if (!IN_GROUP(s, fx, i1) || !IN_GROUP(s, fx, i2)) continue;     /* :373-376 */
    if (s->bl_t[k] != fx->btype) continue;
    if (tag[i1] > tag[i2]) { int t = i1; i1 = i2; i2 = t; }
    else if (tag[i1] == tag[i2]) return seterr(s, "Fix extrusion, bond i-i exists");
    if (s->num_bond[i1] == 1 || s->num_bond[i2] == 1 || s->num_bond[i1] == 0 || s->num_bond[i2] == 0 ||
        s->bondcount[i1] != 1 || s->bondcount[i2] != 1) continue;
    int L = map_(s, tag[i1] - 1), R = map_(s, tag[i2] + 1);
    if (ext_can(s, fx, L, fx->ctcf_left)) {
      if (ext_can(s, fx, R, fx->ctcf_right)) {
        double rsq = d2raw(s, L, R);
        if (rsq >= dc[L] && rsq >= dc[R]) continue;
        if (rsq < dc[L]) { dc[L] = rsq; to_add[L] = tag[R]; }
        if (rsq < dc[R]) { dc[R] = rsq; to_add[R] = tag[L]; }
        to_remove[i1] = tag[i2]; to_remove[i2] = tag[i1];
      } else {
        double rsq = d2raw(s, L, i2);
        if (rsq >= dc[L]) continue;
        dc[L] = rsq; to_add[L] = tag[i2];
        if (dc[i2] == BIG) { dc[i2] = rsq; to_add[i2] = tag[L]; }
        to_remove[i1] = tag[i2]; to_remove[i2] = tag[i1];
      }
    } else if (ext_can(s, fx, R, fx->ctcf_right)) {
      double rsq = d2raw(s, i1, R);
      if (rsq >= dc[R]) continue;
      if (dc[i1] == BIG) { dc[i1] = rsq; to_add[i1] = tag[R]; }
      if (rsq < dc[R]) { dc[R] = rsq; to_add[R] = tag[i1]; }
      to_remove[i1] = tag[i2]; to_remove[i2] = tag[i1];
    }
  }
  /* Phase 2 (:517-599): losers cancel the removal of their extruder */
  {
    int iisleft = 0, lb = 0, rb = 0;
    for (int i = 0; i < n; i++) {
      if (to_add[i] == 0) continue;
      int j = map_(s, to_add[i]);
      if (to_add[j] != tag[i]) {
        if (tag[i] < tag[j]) { lb = map_(s, tag[i] + 1); rb = map_(s, tag[j] - 1); iisleft = 1; }
        else if (tag[i] > tag[j]) { lb = map_(s, tag[j] + 1); rb = map_(s, tag[i] - 1); iisleft = 0; }
        if (iisleft) {
          if (tag[lb] == to_remove[rb] && to_remove[lb] == tag[rb]) { to_remove[lb] = 0; to_remove[rb] = 0; }
          else if (tag[i] == to_remove[rb] && to_remove[i] == tag[rb]) { to_remove[i] = 0; to_remove[rb] = 0; }
          else if (tag[lb] == to_remove[j] && to_remove[lb] == tag[j]) { to_remove[lb] = 0; to_remove[j] = 0; }
          else if (tag[i] == to_remove[j] && tag[j] == to_remove[i]) { to_remove[i] = 0; to_remove[j] = 0; }
        } else {
          if (tag[lb] == to_remove[rb] && to_remove[lb] == tag[rb]) { to_remove[lb] = 0; to_remove[rb] = 0; }
          else if (tag[i] == to_remove[lb] && to_remove[i] == tag[lb]) { to_remove[i] = 0; to_remove[lb] = 0; }
          else if (tag[rb] == to_remove[j] && to_remove[rb] == tag[j]) { to_remove[rb] = 0; to_remove[j] = 0; }
          else if (tag[i] == to_remove[j] && tag[j] == to_remove[i]) { to_remove[i] = 0; to_remove[j] = 0; }
        }
      }
    }
  }
  /* Phase 3 (:618-692): removals */
  int nbreak = 0;
#define TA(t) (map_(s, (t)) < 0 ? 0 : to_add[map_(s, (t))])
#define TASET(t, v) do { int m_ = map_(s, (t)); if (m_ >= 0) to_add[m_] = (v); } while (0)
  for (int i = 0; i < n; i++) {
    if (to_remove[i] == 0) continue;
    int j = map_(s, to_remove[i]);
    if (to_remove[j] != tag[i]) continue;
    int lb, rb;
    if (to_remove[i] < tag[i]) { lb = to_remove[i]; rb = tag[i]; } else { lb = tag[i]; rb = to_remove[i]; }
    if (TA(lb - 1) == rb && TA(rb) == lb - 1 && TA(lb) == rb + 1 && TA(rb + 1) == lb) {
      TASET(lb - 1, rb + 1); TASET(rb + 1, lb - 1); TASET(lb, 0); TASET(rb, 0);
    }
    if ((TA(lb - 1) == rb && TA(rb) == lb - 1) || (TA(lb - 1) == rb + 1 && TA(rb + 1) == lb - 1) ||
        (TA(lb) == rb + 1 && TA(rb + 1) == lb)) {
      delete_bond(s, i, to_remove[i]);
      special_remove12(s, i, to_remove[i]);
      final_to_remove[i] = tag[j]; final_to_remove[j] = tag[i];
      if (tag[i] < tag[j]) nbreak++;
    }
  }
#undef TA
#undef TASET
  /* Phase 4 (:699-786): creations */
  int ncreate = 0;
#define TR(t) (map_(s, (t)) < 0 ? 0 : to_remove[map_(s, (t))])
  for (int i = 0; i < n; i++) {
    if (to_add[i] == 0) continue;
    int j = map_(s, to_add[i]);
    if (to_add[j] != tag[i]) continue;
    if (s->num_bond[i] == s->bpa) continue;
    int lb, rb;
    if (to_add[i] < tag[i]) { lb = to_add[i]; rb = tag[i]; } else { lb = tag[i]; rb = to_add[i]; }
    if ((TR(lb + 1) == rb && TR(rb) == lb + 1) || (TR(lb + 1) == rb - 1 && TR(rb - 1) == lb + 1) ||
        (TR(lb) == rb - 1 && TR(rb - 1) == lb)) {
      s->bond_type[i * s->bpa + s->num_bond[i]] = fx->btype; s->bond_atom[i * s->bpa + s->num_bond[i]] = tag[j]; s->num_bond[i]++;
      if (special_insert12(s, i, tag[j], "New bond exceeded special list size in fix extrusion")) return 1;
      s->bondcount[i]++;
      final_to_add[i] = tag[j]; final_to_add[j] = tag[i];
      if (tag[i] < tag[j]) ncreate++;
    }
  }
#undef TR
  fx->lastcount = nbreak;                        /* compute_vector [1] = breakcount (:1496-1501) */
  if (!nbreak && !ncreate) return 0;
  if (nbreak != ncreate) return seterr(s, "Numbers of created and broken bonds are not equal");
  fx->next_reneighbor = s->ntimestep;
  /* broken[] / created[] over owned atoms (:823-863), then update_topology (:924-1002) */
  int *broken = malloc(2 * (size_t)(nbreak + 1) * sizeof(int)), *created = malloc(2 * (size_t)(ncreate + 1) * sizeof(int));
  int nb = 0, ncr = 0;
  for (int i = 0; i < n; i++) {
    if (final_to_remove[i] == 0) continue;
    int j = map_(s, final_to_remove[i]);
    if (j < 0 || tag[i] < tag[j]) { broken[2 * nb] = tag[i]; broken[2 * nb + 1] = final_to_remove[i]; nb++; }
  }
  for (int i = 0; i < n; i++) {
    if (final_to_add[i] == 0) continue;
    int j = map_(s, final_to_add[i]);
    if (j < 0 || tag[i] < tag[j]) { created[2 * ncr] = tag[i]; created[2 * ncr + 1] = final_to_add[i]; ncr++; }
  }
  int rc = topo_broken(s, nb, broken, 0) || topo_created(s, ncr, created, NULL);
  free(broken); free(created);
  return rc;
}

/* ===================== fix ex_load: src/USER-LE/fix_ex_load.cpp:329-655 ===================== */
/* Is (i, j), i < j local and both owned, an entry of the pair list of the last reneighbor?  fix_ex_load.cpp:427-451 walks
   list->firstneigh of an occasional list that NPairCopy (src/npair_copy.cpp) aliases to the pair list; the pair is stored
   under the smaller local index (npair_half_bin_newtoff.cpp:90) unless find_special said "weight 0" (:103-112) or it was
   farther than the list cutoff at build time.  The owned-owned entry is the unshifted one. */
static int in_pair_list(const leo_t *s, int i, int j) {
  if (!s->pair_on) return 0;
  if (s->brute) {
    for (long p = 0; p < s->npairs; p++)
      if (s->pi_[p] == i && PJ_INDEX(s->pj[p]) == j && !s->pshift[3 * p] && !s->pshift[3 * p + 1] && !s->pshift[3 * p + 2]) return 1;
    return 0;
  }
  for (int p = s->firstneigh[i]; p < s->firstneigh[i + 1]; p++) if (PJ_INDEX(s->pj[p]) == j && !PJ_ONESIDE(s->pj[p])) return 1;
  return 0;
}
static int fire_ex_load(leo_t *s, leo_fix *fx) {
  int n = s->n; const int *tag = s->tag;
  int *partner = s->ia, *finalpartner = s->ib; double *distsq = s->da;
  fx->lastcount = 0;
  recount_bondcount(s, fx->btype, NULL);
  for (int i = 0; i < n; i++) { partner[i] = 0; finalpartner[i] = 0; distsq[i] = BIG; }
  /* candidate scan (:433-505).  Half list half/bin/newtoff: pair stored under the smaller local index,
     visited for ascending i.  Only |tag_i - tag_j| == 2 pairs survive; ghost entries (periodic images)
     are skipped de facto by num_bond[ghost] != 2, so the test uses stored (raw) coordinates. */
  for (int i = 0; i < n; i++) {
    int cand[2], ncand = 0;
    int j1 = map_(s, tag[i] - 2), j2 = map_(s, tag[i] + 2);
    if (j1 >= 0 && stores_pair(s, i, j1)) cand[ncand++] = j1;      /* the pairs the scan meets in i's list */
    if (j2 >= 0 && stores_pair(s, i, j2)) cand[ncand++] = j2;
    if (ncand == 2 && cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
    if (!IN_GROUP(s, fx, i)) continue;                               /* :435 */
    for (int c = 0; c < ncand; c++) {
      int j = cand[c];
      if (!IN_GROUP(s, fx, j)) continue;                             /* :450 */
      int itype = s->type[i], jtype = s->type[j], possible = 0;
      if (itype == fx->iatomtype && jtype == fx->jatomtype) {
        if ((fx->imaxbond == 0 || s->bondcount[i] < fx->imaxbond) && (fx->jmaxbond == 0 || s->bondcount[j] < fx->jmaxbond)) possible = 1;
      } else if (itype == fx->jatomtype && jtype == fx->iatomtype) {
        if ((fx->jmaxbond == 0 || s->bondcount[i] < fx->jmaxbond) && (fx->imaxbond == 0 || s->bondcount[j] < fx->imaxbond)) possible = 1;
      }
      if (!possible) continue;
      int mid = map_(s, (tag[i] < tag[j] ? tag[i] : tag[j]) + 1);
      if (partner[mid] != 0) continue;
      if (s->num_bond[i] != 2) continue;
      if (s->num_bond[j] != 2) continue;
      if (s->num_bond[mid] != 2) continue;
      if (partner[mid] != 0) continue;
      const int *sl = s->special + (size_t)i * s->maxspecial;
      for (int k = 0; k < s->nspecial[3 * i]; k++) if (sl[k] == tag[j]) possible = 0;
      if (!possible) continue;
      double rsq = d2raw(s, i, j);
      if (rsq >= fx->cutsq) continue;
      /* the scan only ever sees pairs that ARE entries of the list (:437-441 loop over jlist).  Checked last here: every
         test above is a `continue` without side effects, so their order relative to this one cannot change the result */
      if (!in_pair_list(s, i, j)) continue;
      if (rsq < distsq[i]) { partner[i] = tag[j]; distsq[i] = rsq; }
      if (rsq < distsq[j]) { partner[j] = tag[i]; distsq[j] = rsq; }
    }
  }
  double *probability = distsq;
  if (fx->fraction < 1.0)
    for (int i = 0; i < n; i++) if (partner[i]) probability[i] = leo_ranmars_uniform(&fx->rng);
  int ncreate = 0;
  for (int i = 0; i < n; i++) {
    if (partner[i] == 0) continue;
    int j = map_(s, partner[i]);
    if (partner[j] != tag[i]) continue;
    if (fx->fraction < 1.0) {
      if (tag[i] < tag[j]) { if (probability[i] >= fx->fraction) continue; }
      else { if (probability[j] >= fx->fraction) continue; }
    }
    if (s->num_bond[i] == s->bpa) return seterr(s, "New bond exceeded bonds per atom in fix ex_load");
    s->bond_type[i * s->bpa + s->num_bond[i]] = fx->btype; s->bond_atom[i * s->bpa + s->num_bond[i]] = tag[j]; s->num_bond[i]++;
    if (special_insert12(s, i, tag[j], "New bond exceeded special list size in fix ex_load")) return 1;
    s->bondcount[i]++;
    if (s->type[i] == fx->iatomtype) { if (s->bondcount[i] == fx->imaxbond) s->type[i] = fx->inewtype; }
    else { if (s->bondcount[i] == fx->jmaxbond) s->type[i] = fx->jnewtype; }
    finalpartner[i] = tag[j]; finalpartner[j] = tag[i];
    if (tag[i] < tag[j]) ncreate++;
  }
  fx->lastcount = ncreate; fx->totalcount += ncreate; s->nbonds += ncreate;
  if (!ncreate) return 0;
  fx->next_reneighbor = s->ntimestep;
  int *created = malloc(2 * (size_t)ncreate * sizeof(int)), nc = 0;
  for (int i = 0; i < n; i++) {
    if (finalpartner[i] == 0) continue;
    int j = map_(s, finalpartner[i]);
    if (j < 0 || tag[i] < tag[j]) { created[2 * nc] = tag[i]; created[2 * nc + 1] = finalpartner[i]; nc++; }
  }
  int rc = topo_created(s, nc, created, fx);
  free(created);
  return rc;
}

/* ===================== fix bond/create: src/MC/fix_bond_create.cpp:302-345 setup, :349-640 post_integrate =====================
   Candidates are ALL pairs of the pair neighbor list (the reference rebuilds an occasional list at the firing step;
   here the list of the last reneighboring is walked, which holds every pair now within the fix cutoff <= pair cutoff;
   a pair whose special status changed earlier in the same step is still seen with its old status).  Exact ties of two
   candidate distances are broken by visit order, which is the order of this list, not the reference's bin order. */
static void bond_create_setup(leo_t *s, leo_fix *fx) {            /* :302-345, once ("countflag") */
  if (fx->counted) return;
  fx->counted = 1;
  for (int i = 0; i < s->n; i++) {
    int c = 0;
    for (int j = 0; j < s->num_bond[i]; j++) if (s->bond_type[i * s->bpa + j] == fx->btype) c++;
    fx->bc_tag[s->tag[i]] = c;
  }
}
static int fire_bond_create(leo_t *s, leo_fix *fx) {
  int n = s->n; const int *tag = s->tag;
  int *partner = s->ia, *finalpartner = s->ib; double *distsq = s->da;
  fx->lastcount = 0;
  if (s->brute) return seterr(s, "oracle: fix bond/create needs a box of at least 3 neighbor cutoffs per dimension");
  for (int i = 0; i < n; i++) { partner[i] = 0; finalpartner[i] = 0; distsq[i] = BIG; s->bondcount[i] = fx->bc_tag[tag[i]]; }
  if (s->pair_on)
    for (int i = 0; i < n; i++) {
      int itype = s->type[i];
      if (!IN_GROUP(s, fx, i)) continue;                             /* fix_bond_create.cpp:421 */
      for (int p = s->firstneigh[i]; p < s->firstneigh[i + 1]; p++) {
        int j = PJ_INDEX(s->pj[p]);
        if (!IN_GROUP(s, fx, j)) continue;                           /* :432 */
        int jtype = s->type[j], possible = 0;
        if (itype == fx->iatomtype && jtype == fx->jatomtype) {
          if ((fx->imaxbond == 0 || s->bondcount[i] < fx->imaxbond) && (fx->jmaxbond == 0 || s->bondcount[j] < fx->jmaxbond)) possible = 1;
        } else if (itype == fx->jatomtype && jtype == fx->iatomtype) {
          if ((fx->jmaxbond == 0 || s->bondcount[i] < fx->jmaxbond) && (fx->imaxbond == 0 || s->bondcount[j] < fx->imaxbond)) possible = 1;
        }
        if (!possible) continue;
        const int *sl = s->special + (size_t)i * s->maxspecial;                     /* :455-458 no duplicate bond */
        for (int k = 0; k < s->nspecial[3 * i]; k++) if (sl[k] == tag[j]) possible = 0;
        if (!possible) continue;
        double d[3] = { s->x[3 * i] - s->x[3 * j], s->x[3 * i + 1] - s->x[3 * j + 1], s->x[3 * i + 2] - s->x[3 * j + 2] };
        signed char sh[3]; minimg(s, d, sh);
        double rsq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (rsq >= fx->cutsq) continue;
        if (rsq < distsq[i]) { partner[i] = tag[j]; distsq[i] = rsq; }
        if (rsq < distsq[j]) { partner[j] = tag[i]; distsq[j] = rsq; }
      }
    }
  double *probability = distsq;
  if (fx->fraction < 1.0)
    for (int i = 0; i < n; i++) if (partner[i]) probability[i] = leo_ranmars_uniform(&fx->rng);
  int ncreate = 0;
  for (int i = 0; i < n; i++) {
    if (partner[i] == 0) continue;
    int j = map_(s, partner[i]);
    if (partner[j] != tag[i]) continue;
    if (fx->fraction < 1.0) {
      if (tag[i] < tag[j]) { if (probability[i] >= fx->fraction) continue; }
      else { if (probability[j] >= fx->fraction) continue; }
    }
    if (s->num_bond[i] == s->bpa) return seterr(s, "New bond exceeded bonds per atom in fix bond/create");
    s->bond_type[i * s->bpa + s->num_bond[i]] = fx->btype; s->bond_atom[i * s->bpa + s->num_bond[i]] = tag[j]; s->num_bond[i]++;
    if (special_insert12(s, i, tag[j], "New bond exceeded special list size in fix bond/create")) return 1;
    s->bondcount[i]++; fx->bc_tag[tag[i]]++;
    if (s->type[i] == fx->iatomtype) { if (s->bondcount[i] == fx->imaxbond) s->type[i] = fx->inewtype; }
    else { if (s->bondcount[i] == fx->jmaxbond) s->type[i] = fx->jnewtype; }
    finalpartner[i] = tag[j]; finalpartner[j] = tag[i];
    if (tag[i] < tag[j]) ncreate++;
  }
  fx->lastcount = ncreate; fx->totalcount += ncreate; s->nbonds += ncreate;
  if (!ncreate) return 0;
  fx->next_reneighbor = s->ntimestep;
  int *created = malloc(2 * (size_t)ncreate * sizeof(int)), nc = 0;
  for (int i = 0; i < n; i++) {
    if (finalpartner[i] == 0) continue;
    int j = map_(s, finalpartner[i]);
    if (j < 0 || tag[i] < tag[j]) { created[2 * nc] = tag[i]; created[2 * nc + 1] = finalpartner[i]; nc++; }
  }
  int rc = topo_created(s, nc, created, fx);
  free(created);
  return rc;
}

/* ===================== fix ex_unload: src/USER-LE/fix_ex_unload.cpp:172-372 ===================== */
static int fire_ex_unload(leo_t *s, leo_fix *fx) {
  int n = s->n; const int *tag = s->tag;
  int *partner = s->ia, *finalpartner = s->ib; double *distsq = s->da;
  fx->lastcount = 0;
  for (int i = 0; i < n; i++) { partner[i] = 0; finalpartner[i] = 0; distsq[i] = 0.0; }
  for (int k = 0; k < s->nbondlist; k++) {
    int i1 = s->bl_i[k], i2 = s->bl_j[k];
    if (!IN_GROUP(s, fx, i1) || !IN_GROUP(s, fx, i2)) continue;     /* fix_ex_unload.cpp:228-229 */
    if (s->bl_t[k] != fx->btype) continue;
    const signed char *sh = s->bl_s + 3 * k;
    int ghost = sh[0] || sh[1] || sh[2];
    double dx = s->x[3 * i1] - (s->x[3 * i2] + sh[0] * s->prd[0]);
    double dy = s->x[3 * i1 + 1] - (s->x[3 * i2 + 1] + sh[1] * s->prd[1]);
    double dz = s->x[3 * i1 + 2] - (s->x[3 * i2 + 2] + sh[2] * s->prd[2]);
    double rsq = dx * dx + dy * dy + dz * dz;
    if (rsq <= fx->cutsq) continue;
    if (rsq > distsq[i1]) { partner[i1] = tag[i2]; distsq[i1] = rsq; }
    /* i2 is a ghost slot when the listing straddles a face: the owned copy is written by the other listing */
    if (!ghost && rsq > distsq[i2]) { partner[i2] = tag[i1]; distsq[i2] = rsq; }
  }
  double *probability = distsq;
  if (fx->fraction < 1.0)
    for (int i = 0; i < n; i++) if (partner[i]) probability[i] = leo_ranmars_uniform(&fx->rng);
  int nbreak = 0;
  for (int i = 0; i < n; i++) {
    if (partner[i] == 0) continue;
    int j = map_(s, partner[i]);
    if (partner[j] != tag[i]) continue;
    if (fx->fraction < 1.0) {
      if (tag[i] < tag[j]) { if (probability[i] >= fx->fraction) continue; }
      else { if (probability[j] >= fx->fraction) continue; }
    }
    delete_bond(s, i, partner[i]);
    special_remove12(s, i, partner[i]);
    finalpartner[i] = tag[j]; finalpartner[j] = tag[i];
    if (tag[i] < tag[j]) nbreak++;
  }
  fx->lastcount = nbreak; fx->totalcount += nbreak; s->nbonds -= nbreak;
  if (!nbreak) return 0;
  fx->next_reneighbor = s->ntimestep;
  int *broken = malloc(2 * (size_t)nbreak * sizeof(int)), nb = 0;
  for (int i = 0; i < n; i++) {
    if (finalpartner[i] == 0) continue;
    int j = map_(s, finalpartner[i]);
    if (j < 0 || tag[i] < tag[j]) { broken[2 * nb] = tag[i]; broken[2 * nb + 1] = finalpartner[i]; nb++; }
  }
  int rc = topo_broken(s, nb, broken, s->unload_angleflag);     /* fix_ex_unload.cpp:149-152: set in init() */
  free(broken);
  return rc;
}

static int fire_fix(leo_t *s, leo_fix *fx) {
  switch (fx->kind) {
    case FIX_EXTRUSION: return fire_extrusion(s, fx);
    case FIX_EX_LOAD: return fire_ex_load(s, fx);
    case FIX_EX_UNLOAD: return fire_ex_unload(s, fx);
    case FIX_BOND_CREATE: return fire_bond_create(s, fx);
  }
  return 0;
}
int leo_fire_fix(leo_t *s, int k) { return fire_fix(s, &s->fix[k]); }

/* ===================== thermo ===================== */
static void thermo_eval(leo_t *s, double *out) {
  int n = s->n;
  double t = 0.0;
  for (int i = 0; i < n; i++)
    t += (s->v[3 * i] * s->v[3 * i] + s->v[3 * i + 1] * s->v[3 * i + 1] + s->v[3 * i + 2] * s->v[3 * i + 2]) * s->mass[s->type[i]];
  double dof = 3.0 * n - 3.0;                                   /* src/compute_temp.cpp:60-68 */
  double tfactor = dof > 0 ? s->mvv2e / (dof * s->boltz) : 0.0;
  double temp = t * tfactor;
  double norm = (s->units == 0) ? (double)n : 1.0;               /* thermo_modify norm default: lj yes */
  double ke = temp * 0.5 * dof * s->boltz;                        /* src/thermo.cpp compute_ke */
  double vol = s->prd[0] * s->prd[1] * s->prd[2];
  double vir = s->vpair[0] + s->vpair[1] + s->vpair[2] + s->vbond[0] + s->vbond[1] + s->vbond[2] + s->vangle[0] + s->vangle[1] + s->vangle[2];
  const double emol = s->ebond + s->eangle;                       /* thermo keyword emol = ebond + eangle (+ ...) src/thermo.cpp */
  double press = (dof * s->boltz * temp + vir) / 3.0 / vol * s->nktv2p;   /* src/compute_pressure.cpp:228-236 */
  out[0] = temp; out[1] = s->evdwl / norm; out[2] = emol / norm; out[3] = (ke + s->evdwl + emol) / norm;
  out[4] = press; out[5] = ke / norm; out[6] = s->evdwl; out[7] = s->ebond;
  for (int k = 0; k < 6; k++) out[8 + k] = s->vpair[k] + s->vbond[k] + s->vangle[k];
}
static void thermo_record(leo_t *s) {
  if (s->nthermo == s->maxthermo) {
    s->maxthermo = s->maxthermo ? 2 * s->maxthermo : 16;
    s->thermo_hist = realloc(s->thermo_hist, (size_t)s->maxthermo * 16 * sizeof(double));
  }
  double *e = s->thermo_hist + (size_t)s->nthermo * 16;
  e[0] = (double)s->ntimestep; thermo_eval(s, e + 1); e[15] = (double)s->nbonds;
  s->nthermo++;
}
void leo_thermo(leo_t *s, double *out) { thermo_eval(s, out); }
int leo_thermo_count(leo_t *s) { return s->nthermo; }
void leo_thermo_get(leo_t *s, int idx, double *out16) { memcpy(out16, s->thermo_hist + (size_t)idx * 16, 16 * sizeof(double)); }
void leo_pair_virial(leo_t *s, double *o) { memcpy(o, s->vpair, sizeof s->vpair); }
void leo_bond_virial(leo_t *s, double *o) { memcpy(o, s->vbond, sizeof s->vbond); }

/* ===================== run: src/run.cpp:38-188, src/verlet.cpp:87-156 + :223-354 ===================== */
static int run_init(leo_t *s) {
  pair_init(s);
  s->triggersq = 0.25 * s->skin * s->skin;                        /* src/neighbor.cpp:240- init */
  for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_LANGEVIN) langevin_init(s, &s->fix[k]);
  s->unload_angleflag = s->nangles > 0;
  if (!s->num_bond) { int z = 0; leo_set_bonds(s, 0, &z, &z, &z); }
  if (!s->special) leo_special_build(s, s->special_lj[1], s->special_lj[2], s->special_lj[3]);
  return 0;
}
static int verlet_setup(leo_t *s) {
  pbc(s);
  if (s->sortfreq > 0) atom_sort(s);
  double t0 = now();
  neigh_build(s); s->nbuilds = 0;
  s->t_neigh += now() - t0;
  if (s->errflag) return 1;
  memset(s->f, 0, 3 * (size_t)s->n * sizeof(double));
  pair_compute(s, 1);
  if (bond_compute(s, 1)) return 1;
  angle_compute(s, 1);
  /* modify->setup: FixLangevin::setup -> post_force (src/fix_langevin.cpp:372-373) */
  for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_LANGEVIN) langevin_post_force(s, &s->fix[k]);
  if (s->errflag) return 1;
  for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_BOND_CREATE) bond_create_setup(s, &s->fix[k]);
  thermo_record(s);
  return 0;
}
int leo_setup_forces(leo_t *s) {
  run_init(s);
  pbc(s);
  neigh_build(s);
  if (s->errflag) return 1;
  memset(s->f, 0, 3 * (size_t)s->n * sizeof(double));
  pair_compute(s, 1);
  if (bond_compute(s, 1)) return 1;
  return angle_compute(s, 1);
}
/* ===================== run_style respa (src/respa.cpp) ===================== */
static void copy_flevel_f(leo_t *s, int l) {                       /* :793-815 */
  for (int i = 0; i < s->n; i++) memcpy(s->f + 3 * i, s->flevel[l] + 3 * (size_t)(s->tag[i] - 1), 3 * sizeof(double));
}
static void copy_f_flevel(leo_t *s, int l) {                       /* :769-791 */
  for (int i = 0; i < s->n; i++) memcpy(s->flevel[l] + 3 * (size_t)(s->tag[i] - 1), s->f + 3 * i, 3 * sizeof(double));
}
static void sum_flevel_f(leo_t *s) {                               /* :817-843: level 0 copied, the others added in order */
  copy_flevel_f(s, 0);
  for (int l = 1; l < s->respa_levels; l++)
    for (int i = 0; i < s->n; i++)
      for (int d = 0; d < 3; d++) s->f[3 * i + d] += s->flevel[l][3 * (size_t)(s->tag[i] - 1) + d];
}
/* FixNVE::initial_integrate_respa / final_integrate_respa (src/fix_nve.cpp:145-163): the innermost level moves x and v,
   every other level only kicks v, each with its own level's step */
static void nve_kick(leo_t *s, const leo_fix *fx, double step, int drift) {
  double dtv = step, dtf = 0.5 * step * s->ftm2v;
  for (int i = 0; i < s->n; i++) {
    if (fx->gmask && !fx->gmask[s->tag[i]]) continue;          /* (the respa variants call the plain ones: same group mask) */
    double dtfm = dtf / s->mass[s->type[i]];
    for (int d = 0; d < 3; d++) {
      s->v[3 * i + d] += dtfm * s->f[3 * i + d];
      if (drift) s->x[3 * i + d] += dtv * s->v[3 * i + d];
    }
  }
}
static int respa_level_forces(leo_t *s, int l, int eflag) {          /* :673-713 (same order as Verlet: pair, then bond) */
  memset(s->f, 0, 3 * (size_t)s->n * sizeof(double));
  double t0 = now();
  if (s->respa_level_pair == l) pair_compute(s, eflag);
  s->t_pair += now() - t0; t0 = now();
  if (s->respa_level_bond == l && bond_compute(s, eflag)) return 1;
  if (s->respa_level_angle == l && angle_compute(s, eflag)) return 1;      /* :707-710 */
  s->t_bond += now() - t0;
  return 0;
}
static int respa_setup(leo_t *s) {                                 /* Respa::setup :369-468 */
  const int top = s->respa_levels - 1;
  s->respa_step[top] = s->dt;                                      /* Respa::init :340-344 */
  for (int l = top - 1; l >= 0; l--) s->respa_step[l] = s->respa_step[l + 1] / s->respa_loop[l];
  for (int l = 0; l <= top; l++) {
    free(s->flevel[l]);
    s->flevel[l] = (double *)calloc(3 * (size_t)(s->maxtag + 1), sizeof(double));
  }
  pbc(s);
  if (s->sortfreq > 0) atom_sort(s);
  neigh_build(s); s->nbuilds = 0;
  if (s->errflag) return 1;
  for (int l = 0; l <= top; l++) {
    if (respa_level_forces(s, l, 1)) return 1;
    copy_f_flevel(s, l);
  }
  sum_flevel_f(s);
  /* modify->setup: FixLangevin::setup, respa branch (src/fix_langevin.cpp:372-378): the thermostat force goes into the
     outermost level's array */
  for (int k = 0; k < s->nfix; k++)
    if (s->fix[k].kind == FIX_LANGEVIN) { copy_flevel_f(s, top); langevin_post_force(s, &s->fix[k]); copy_f_flevel(s, top); }
  if (s->errflag) return 1;
  for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_BOND_CREATE) bond_create_setup(s, &s->fix[k]);
  thermo_record(s);
  return 0;
}
static int respa_recurse(leo_t *s, int l, int eflag) {              /* Respa::recurse :600-741 */
  const int top = s->respa_levels - 1;
  copy_flevel_f(s, l);
  for (int iloop = 0; iloop < s->respa_loop[l]; iloop++) {
    double t0 = now();
    for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_NVE) nve_kick(s, &s->fix[k], s->respa_step[l], l == 0);
    /* post_integrate_respa: the LE fixes act at the outermost level only (fix_extrusion.cpp:1139-1143, fix_ex_load.cpp:1283-1286,
       fix_ex_unload.cpp:694-697, MC/fix_bond_create.cpp:1249-1252, MC/fix_bond_break.cpp:693-696:
       `if (ilevel == nlevels_respa-1) post_integrate()`) */
    if (l == top)
      for (int k = 0; k < s->nfix; k++) {
        leo_fix *fx = &s->fix[k];
        if (fx->kind < FIX_EXTRUSION) continue;
        if (s->ntimestep % fx->nevery - fx->phase) continue;
        if (fire_fix(s, fx)) return 1;
      }
    s->t_modify += now() - t0;
    if (l == top && neigh_decide(s)) {                              /* :616-651: decided BEFORE the inner levels move x */
      t0 = now();
      pbc(s);
      if (s->sortfreq > 0 && s->ntimestep >= s->nextsort) atom_sort(s);
      neigh_build(s);
      s->t_neigh += now() - t0;
      if (s->errflag) return 1;
    }
    if (l && respa_recurse(s, l - 1, eflag)) return 1;
    if (respa_level_forces(s, l, eflag)) return 1;
    t0 = now();
    if (l == top)                                                   /* FixLangevin::post_force_respa (src/fix_langevin.cpp:576-579) */
      for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_LANGEVIN) langevin_post_force(s, &s->fix[k]);
    for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_NVE) nve_kick(s, &s->fix[k], s->respa_step[l], 0);
    s->t_modify += now() - t0;
  }
  copy_f_flevel(s, l);
  return 0;
}
static int respa_run(leo_t *s, int nsteps) {                       /* Respa::run :544-575 */
  if (respa_setup(s)) return 1;
  double tstart = now();
  for (int it = 0; it < nsteps; it++) {
    s->ntimestep++;
    int eflag = (s->ntimestep == s->endstep) || (s->thermo_every > 0 && s->ntimestep % s->thermo_every == 0);
    if (respa_recurse(s, s->respa_levels - 1, eflag)) return 1;
    sum_flevel_f(s);
    if (eflag) thermo_record(s);
  }
  s->t_total = now() - tstart;
  return 0;
}

int leo_run(leo_t *s, int nsteps) {
  if (s->errflag) return 1;
  run_init(s);
  if (s->respa_levels > 0) {
    s->beginstep = s->ntimestep; s->endstep = s->ntimestep + nsteps;
    s->t_pair = s->t_bond = s->t_neigh = s->t_modify = 0.0;
    return respa_run(s, nsteps);
  }
  s->beginstep = s->ntimestep; s->endstep = s->ntimestep + nsteps;
  s->t_pair = s->t_bond = s->t_neigh = s->t_modify = 0.0;
  if (verlet_setup(s)) return 1;
  double tstart = now(), t0;
  for (int it = 0; it < nsteps; it++) {
    s->ntimestep++;
    int eflag = (s->ntimestep == s->endstep) || (s->thermo_every > 0 && s->ntimestep % s->thermo_every == 0);
    t0 = now();
    for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_NVE) nve_initial(s, &s->fix[k]);
    for (int k = 0; k < s->nfix; k++) {
      leo_fix *fx = &s->fix[k];
      if (fx->kind < FIX_EXTRUSION) continue;
      if (s->ntimestep % fx->nevery - fx->phase) continue;               /* fix_extrusion.cpp:265 etc. */
      if (fire_fix(s, fx)) return 1;
    }
    s->t_modify += now() - t0;
    int nflag = neigh_decide(s);
    if (nflag) {
      t0 = now();
      pbc(s);
      if (s->sortfreq > 0 && s->ntimestep >= s->nextsort) atom_sort(s);
      neigh_build(s);
      s->t_neigh += now() - t0;
      if (s->errflag) return 1;
    }
    memset(s->f, 0, 3 * (size_t)s->n * sizeof(double));
    t0 = now(); pair_compute(s, eflag); s->t_pair += now() - t0;
    t0 = now(); if (bond_compute(s, eflag)) return 1; angle_compute(s, eflag); s->t_bond += now() - t0;
    t0 = now();
    for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_LANGEVIN) langevin_post_force(s, &s->fix[k]);
    for (int k = 0; k < s->nfix; k++) if (s->fix[k].kind == FIX_NVE) nve_final(s, &s->fix[k]);
    s->t_modify += now() - t0;
    if (eflag) thermo_record(s);
  }
  s->t_total = now() - tstart;
  return 0;
}
const char *leo_error(leo_t *s) { return s->err; }

/* ===================== queries ===================== */
long leo_ntimestep(leo_t *s) { return s->ntimestep; }
int leo_natoms(leo_t *s) { return s->n; }
long leo_nbonds(leo_t *s) { return s->nbonds; }
#define BYTAG(dst, src, w, T) for (int i = 0; i < s->n; i++) memcpy((dst) + (size_t)(s->tag[i] - 1) * (w), (src) + (size_t)i * (w), (w) * sizeof(T))
void leo_get_x(leo_t *s, double *o) { BYTAG(o, s->x, 3, double); }
void leo_get_v(leo_t *s, double *o) { BYTAG(o, s->v, 3, double); }
void leo_get_f(leo_t *s, double *o) { BYTAG(o, s->f, 3, double); }
void leo_get_type(leo_t *s, int *o) { BYTAG(o, s->type, 1, int); }
void leo_get_image(leo_t *s, int *o) { BYTAG(o, s->img, 3, int); }
void leo_get_local_order(leo_t *s, int *o) { memcpy(o, s->tag, s->n * sizeof(int)); }
void leo_set_x(leo_t *s, const double *x) { for (int i = 0; i < s->n; i++) memcpy(s->x + 3 * i, x + 3 * (size_t)(s->tag[i] - 1), 3 * sizeof(double)); }
void leo_set_v(leo_t *s, const double *v) { for (int i = 0; i < s->n; i++) memcpy(s->v + 3 * i, v + 3 * (size_t)(s->tag[i] - 1), 3 * sizeof(double)); }
void leo_get_bonds(leo_t *s, int *nb, int *bt, int *ba) {
  BYTAG(nb, s->num_bond, 1, int); BYTAG(bt, s->bond_type, s->bpa, int); BYTAG(ba, s->bond_atom, s->bpa, int);
}
void leo_get_special(leo_t *s, int *ns, int *sp) { BYTAG(ns, s->nspecial, 3, int); BYTAG(sp, s->special, s->maxspecial, int); }
int leo_bond_per_atom(leo_t *s) { return s->bpa; }
int leo_maxspecial(leo_t *s) { return s->maxspecial; }
void leo_fix_vector(leo_t *s, int k, double *o) {
  o[0] = s->fix[k].lastcount;
  o[1] = (s->fix[k].kind == FIX_EXTRUSION) ? 0.0 : (double)s->fix[k].totalcount;  /* fix_extrusion.cpp:1496-1501 */
}
long leo_neigh_builds(leo_t *s) { return s->nbuilds; }
long leo_neigh_pairs(leo_t *s) { return s->npairs - s->noneside / 2; }   /* (an owned-ghost pair is listed by both of its owned ends) */
long leo_fene_warnings(leo_t *s) { return s->fene_warn; }
void leo_timers(leo_t *s, double *o) {
  o[0] = s->t_pair; o[1] = s->t_bond; o[2] = s->t_neigh; o[3] = s->t_modify;
  o[4] = s->t_total - s->t_pair - s->t_bond - s->t_neigh - s->t_modify; o[5] = s->t_total;
}

/* debug (tests only): scratch arrays of the last LE firing, local index order */
void leo_debug_scratch(leo_t *s, int *ia, int *ib, double *da) {
  if (ia) memcpy(ia, s->ia, s->n * sizeof(int));
  if (ib) memcpy(ib, s->ib, s->n * sizeof(int));
  if (da) memcpy(da, s->da, s->n * sizeof(double));
}
