/* oracle/mmd_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see mmd_oracle.h).
 *
 * Restates, function by function, the algorithm of Mantevo/miniMD `ref/` (citations are
 * path:line under /root/reference).  Serial, strict evaluation order, no FMA contraction, so that it
 * is bit-comparable with the reference run with one thread.  Parity is PINNED by tests/test_oracle_*.py
 * against the reference's published logs, rows printed by the reference built in this container and
 * per-atom arrays dumped from the reference objects (tests/golden/).
 */
#define _GNU_SOURCE
#include "mmd_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef orc_real real;
#define PAD 3
#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

static char g_err[512];
const char* orc_last_error(void) { return g_err; }
static void set_err(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static double wall(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------------
 * data
 * ---------------------------------------------------------------------------------------------- */

enum { FORCE_LJ = 0, FORCE_EAM = 1 };
enum { UNITS_LJ = 0, UNITS_METAL = 1 };

typedef struct {            /* ref/ljs.h:37-51 */
  int nx, ny, nz;
  real t_request, rho;
  int units, forcetype;
  real epsilon, sigma;
  char datafile[1000];
  int has_datafile;
  int ntimes;
  real dt;
  int neigh_every;
  real force_cut, neigh_cut;
  int thermo_nstat;
} deck_t;

typedef struct {            /* ref/atom.h:40-45 */
  real xprd, yprd, zprd, xlo, xhi, ylo, yhi, zlo, zhi;
} box_t;

typedef struct {            /* ref/atom.h:47-106 */
  int natoms, nlocal, nghost, nmax;
  real *x, *v, *f;
  int *type, *tag;
  real mass;
  box_t box;
  real *x_copy, *v_copy;
  int *type_copy, *tag_copy;
  int copy_size;
  struct random_data rnd;   /* per-rank glibc rand() stream: srand(5413) (ref/ljs.cpp:110, atom.cpp:97) */
  char rnd_state[128];
} atom_t;

typedef struct {            /* ref/neighbor.h:39-90 */
  int every, nbinx, nbiny, nbinz;
  real cutneigh, *cutneighsq;
  int *numneigh, *neighbors, maxneighs, halfneigh, ghost_newton, nmax;
  int *bincount, *bins, mbins, atoms_per_bin;
  real xprd, yprd, zprd;
  int nstencil, *stencil;
  int mbinx, mbiny, mbinz, mbinxlo, mbinylo, mbinzlo;
  real binsizex, binsizey, binsizez, bininvx, bininvy, bininvz;
  int ncalls;
} neigh_t;

typedef struct {            /* ref/comm.h:39-102 */
  int me, nswap, maxswap;
  int *pbc_any, *pbc_flagx, *pbc_flagy, *pbc_flagz;
  int *sendnum, *recvnum, *sendproc, *recvproc, *firstrecv;
  int **sendlist, *maxsendlist;
  real *buf_send, *buf_recv;
  int *tag_send, tag_cap;            /* oracle-only: lattice ids of the atoms in an exchange buffer */
  int maxsend, maxrecv;
  int nsend_now, nrecv_now;          /* lock-step handshake values of the current swap */
  int procneigh[3][2], procgrid[3], myloc[3], need[3];
  real *slablo, *slabhi;
} comm_t;

typedef struct {            /* ref/force.h:40-69, force_eam.h:46-108 */
  int style, ntypes, evflag, use_oldcompute;
  real cutforce, *cutforcesq, *epsilon, *sigma, *sigma6;
  real eng_vdwl, virial, mass;
  /* EAM */
  int nrho, nr, nrho_tot, nr_tot;
  real dr, rdr, drho, rdrho, cutmax;
  real *frho, *rhor, *z2r, *rhor_spline, *frho_spline, *z2r_spline, *rho, *fp;
  int eam_nmax;
  int fl_nrho, fl_nr;
  double fl_drho, fl_dr, fl_cut, fl_mass;
  real *fl_frho, *fl_rhor, *fl_zr;
} force_t;

typedef struct {            /* ref/thermo.h:47-71 */
  int nstat, ntimes;
  real t_act, p_act, e_act, t_scale, e_scale, p_scale, mvv2e, dof_boltz, rho;
} thermo_t;

typedef struct {
  atom_t atom;
  neigh_t neigh;
  comm_t comm;
  force_t force;
} rank_t;

struct orc_world {
  int nprocs, quiet, num_threads, ntypes, halfneigh, ghost_newton, sort_flag, sort_every, yaml_output;
  int safe_exchange;         /* Comm::do_safeexchange (ref/comm.h:87): Comm::exchange -> exchange_all (ref/comm.cpp:366-367) */
  deck_t in;
  char input_file[1024];
  rank_t* r;
  thermo_t thermo;
  real dt, dtforce;          /* ref/integrate.h:44-45 */
  int ntimes, run_started;
  double timer[5], t_prev, t_total_start;
  int nrows, maxrows;
  int* row_step;
  double *row_t, *row_u, *row_p;
};

enum { T_TOTAL = 0, T_COMM = 1, T_FORCE = 2, T_NEIGH = 3, T_TEST = 4 };

/* ------------------------------------------------------------------------------------------------
 * Atom storage (ref/atom.cpp:71-100)
 * ---------------------------------------------------------------------------------------------- */

#define ATOM_DELTA 20000

static void atom_grow(atom_t* a)
{
  a->nmax += ATOM_DELTA;
  a->x = (real*)realloc(a->x, (size_t)a->nmax * PAD * sizeof(real));
  a->v = (real*)realloc(a->v, (size_t)a->nmax * PAD * sizeof(real));
  a->f = (real*)realloc(a->f, (size_t)a->nmax * PAD * sizeof(real));
  a->type = (int*)realloc(a->type, (size_t)a->nmax * sizeof(int));
  a->tag = (int*)realloc(a->tag, (size_t)a->nmax * sizeof(int));
}

static void atom_add(atom_t* a, int ntypes, real x, real y, real z, real vx, real vy, real vz, int tag)
{
  if(a->nlocal == a->nmax) atom_grow(a);
  const int n = a->nlocal;
  a->x[n * PAD + 0] = x; a->x[n * PAD + 1] = y; a->x[n * PAD + 2] = z;
  a->v[n * PAD + 0] = vx; a->v[n * PAD + 1] = vy; a->v[n * PAD + 2] = vz;
  int32_t rv;
  random_r(&a->rnd, &rv);                /* == rand() after srand(5413) in glibc */
  a->type[n] = rv % ntypes;              /* ref/atom.cpp:97 */
  a->tag[n] = tag;
  a->nlocal++;
}

/* ref/atom.cpp:106-122 */
static void atom_pbc(atom_t* a)
{
  real* x = a->x;
  for(int i = 0; i < a->nlocal; i++) {
    if(x[i * PAD + 0] < 0.0) x[i * PAD + 0] += a->box.xprd;
    if(x[i * PAD + 0] >= a->box.xprd) x[i * PAD + 0] -= a->box.xprd;
    if(x[i * PAD + 1] < 0.0) x[i * PAD + 1] += a->box.yprd;
    if(x[i * PAD + 1] >= a->box.yprd) x[i * PAD + 1] -= a->box.yprd;
    if(x[i * PAD + 2] < 0.0) x[i * PAD + 2] += a->box.zprd;
    if(x[i * PAD + 2] >= a->box.zprd) x[i * PAD + 2] -= a->box.zprd;
  }
}

static void atom_copy(atom_t* a, int from, int to)   /* ref/atom.cpp:124-133 (+tag) */
{
  for(int d = 0; d < 3; d++) {
    a->x[to * PAD + d] = a->x[from * PAD + d];
    a->v[to * PAD + d] = a->v[from * PAD + d];
  }
  a->type[to] = a->type[from];
  a->tag[to] = a->tag[from];
}

/* ------------------------------------------------------------------------------------------------
 * input deck (ref/input.cpp:48-187) — 14 physical lines, lines 1-2 ignored
 * ---------------------------------------------------------------------------------------------- */

static real parse_real(const char* s, char** end)
{
#if MMD_PRECISION == 1
  return strtof(s, end);
#else
  return strtod(s, end);
#endif
}

static int read_deck(deck_t* in, const char* filename)
{
  char line[256];
  FILE* fp = fopen(filename, "r");
  if(!fp) { set_err("ERROR: Cannot open %s", filename); return 1; }
  char* tok; char* e;
  if(!fgets(line, 256, fp) || !fgets(line, 256, fp) || !fgets(line, 256, fp)) goto bad;
  tok = strtok(line, " \t\n");
  if(tok && strcmp(tok, "lj") == 0) in->units = UNITS_LJ;
  else if(tok && strcmp(tok, "metal") == 0) in->units = UNITS_METAL;
  else { set_err("Unknown units option in file at line 3 ('%s')", tok ? tok : ""); fclose(fp); return 1; }
  if(!fgets(line, 256, fp)) goto bad;
  tok = strtok(line, " \t\n");
  if(tok && strcmp(tok, "none") == 0) in->has_datafile = 0;
  else { in->has_datafile = 1; strncpy(in->datafile, tok ? tok : "", 999); }
  if(!fgets(line, 256, fp)) goto bad;
  tok = strtok(line, " \t\n");
  if(tok && strcmp(tok, "lj") == 0) in->forcetype = FORCE_LJ;
  else if(tok && strcmp(tok, "eam") == 0) in->forcetype = FORCE_EAM;
  else { set_err("Unknown forcetype option in file at line 5 ('%s')", tok ? tok : ""); fclose(fp); return 1; }
  if(!fgets(line, 256, fp)) goto bad;
  in->epsilon = parse_real(line, &e); in->sigma = parse_real(e, &e);
  if(!fgets(line, 256, fp)) goto bad;
  sscanf(line, "%d %d %d", &in->nx, &in->ny, &in->nz);
  if(!fgets(line, 256, fp)) goto bad;
  sscanf(line, "%d", &in->ntimes);
  if(!fgets(line, 256, fp)) goto bad;
  in->dt = parse_real(line, &e);
  if(!fgets(line, 256, fp)) goto bad;
  in->t_request = parse_real(line, &e);
  if(!fgets(line, 256, fp)) goto bad;
  in->rho = parse_real(line, &e);
  if(!fgets(line, 256, fp)) goto bad;
  sscanf(line, "%d", &in->neigh_every);
  if(!fgets(line, 256, fp)) goto bad;
  in->force_cut = parse_real(line, &e); in->neigh_cut = parse_real(e, &e);
  if(!fgets(line, 256, fp)) goto bad;
  sscanf(line, "%d", &in->thermo_nstat);
  fclose(fp);
  in->neigh_cut += in->force_cut;        /* ref/input.cpp:183: the deck holds the skin */
  return 0;
bad:
  fclose(fp);
  set_err("ERROR: input deck %s is shorter than 14 lines", filename);
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Comm::setup (ref/comm.cpp:60-272)
 * ---------------------------------------------------------------------------------------------- */

static int cart_rank(const int pg[3], int c0, int c1, int c2)
{
  /* MPI_Cart_create(reorder=0), periodic: row-major rank, coordinates wrap */
  c0 = (c0 % pg[0] + pg[0]) % pg[0];
  c1 = (c1 % pg[1] + pg[1]) % pg[1];
  c2 = (c2 % pg[2] + pg[2]) % pg[2];
  return (c0 * pg[1] + c1) * pg[2] + c2;
}

static int comm_setup(comm_t* c, real cutneigh, atom_t* atom, int me, int nprocs)
{
  real prd[3] = {atom->box.xprd, atom->box.yprd, atom->box.zprd};
  real area[3] = {prd[0] * prd[1], prd[0] * prd[2], prd[1] * prd[2]};
  real bestsurf = 2.0 * (area[0] + area[1] + area[2]);
  c->me = me;
  c->procgrid[0] = c->procgrid[1] = c->procgrid[2] = 0;
  /* every factorisation ipx*ipy*ipz = nprocs; keep the one with the smallest sub-domain surface
     (strict <, so the first minimum in (ipx, ipy) scan order wins)  ref/comm.cpp:80-120 */
  for(int ipx = 1; ipx <= nprocs; ipx++) {
    if(nprocs % ipx) continue;
    const int nremain = nprocs / ipx;
    for(int ipy = 1; ipy <= nremain; ipy++) {
      if(nremain % ipy) continue;
      const int ipz = nremain / ipy;
      const real surf = area[0] / ipx / ipy + area[1] / ipx / ipz + area[2] / ipy / ipz;
      if(surf < bestsurf) {
        bestsurf = surf;
        c->procgrid[0] = ipx; c->procgrid[1] = ipy; c->procgrid[2] = ipz;
      }
    }
  }
  if(c->procgrid[0] * c->procgrid[1] * c->procgrid[2] != nprocs) { set_err("ERROR: Bad grid of processors"); return 1; }
  const int* pg = c->procgrid;
  c->myloc[0] = me / (pg[1] * pg[2]);
  c->myloc[1] = (me / pg[2]) % pg[1];
  c->myloc[2] = me % pg[2];
  for(int d = 0; d < 3; d++) {
    int lo[3] = {c->myloc[0], c->myloc[1], c->myloc[2]}, hi[3] = {c->myloc[0], c->myloc[1], c->myloc[2]};
    lo[d] -= 1; hi[d] += 1;
    c->procneigh[d][0] = cart_rank(pg, lo[0], lo[1], lo[2]);
    c->procneigh[d][1] = cart_rank(pg, hi[0], hi[1], hi[2]);
  }
  atom->box.xlo = c->myloc[0] * prd[0] / pg[0];
  atom->box.xhi = (c->myloc[0] + 1) * prd[0] / pg[0];
  atom->box.ylo = c->myloc[1] * prd[1] / pg[1];
  atom->box.yhi = (c->myloc[1] + 1) * prd[1] / pg[1];
  atom->box.zlo = c->myloc[2] * prd[2] / pg[2];
  atom->box.zhi = (c->myloc[2] + 1) * prd[2] / pg[2];
  for(int d = 0; d < 3; d++) c->need[d] = (int)(cutneigh * pg[d] / prd[d] + 1);

  c->maxswap = 2 * (c->need[0] + c->need[1] + c->need[2]);
  const int ms = c->maxswap;
  c->slablo = (real*)calloc(ms, sizeof(real)); c->slabhi = (real*)calloc(ms, sizeof(real));
  c->pbc_any = (int*)calloc(ms, sizeof(int)); c->pbc_flagx = (int*)calloc(ms, sizeof(int));
  c->pbc_flagy = (int*)calloc(ms, sizeof(int)); c->pbc_flagz = (int*)calloc(ms, sizeof(int));
  c->sendproc = (int*)calloc(ms, sizeof(int)); c->recvproc = (int*)calloc(ms, sizeof(int));
  c->sendnum = (int*)calloc(ms, sizeof(int)); c->recvnum = (int*)calloc(ms, sizeof(int));
  c->firstrecv = (int*)calloc(ms, sizeof(int)); c->maxsendlist = (int*)calloc(ms, sizeof(int));
  c->sendlist = (int**)calloc(ms, sizeof(int*));
  for(int i = 0; i < ms; i++) { c->maxsendlist[i] = 1000; c->sendlist[i] = (int*)malloc(1000 * sizeof(int)); }
  c->maxsend = 1000; c->buf_send = (real*)malloc((c->maxsend + 1000) * sizeof(real));
  c->maxrecv = 1000; c->buf_recv = (real*)malloc(c->maxrecv * sizeof(real));

  const real boxlo[3] = {atom->box.xlo, atom->box.ylo, atom->box.zlo};
  const real boxhi[3] = {atom->box.xhi, atom->box.yhi, atom->box.zhi};
  int ns = 0;
  for(int d = 0; d < 3; d++) {
    for(int ineed = 0; ineed < 2 * c->need[d]; ineed++) {
      int* flag = d == 0 ? c->pbc_flagx : (d == 1 ? c->pbc_flagy : c->pbc_flagz);
      real lo, hi;
      if(ineed % 2 == 0) {                 /* towards the low neighbor; atoms originate in box myloc+ineed/2 */
        c->sendproc[ns] = c->procneigh[d][0];
        c->recvproc[ns] = c->procneigh[d][1];
        const int nbox = c->myloc[d] + ineed / 2;
        lo = nbox * prd[d] / pg[d];
        hi = boxlo[d] + cutneigh;
        hi = ORC_MIN(hi, (nbox + 1) * prd[d] / pg[d]);
        if(c->myloc[d] == 0) { c->pbc_any[ns] = 1; flag[ns] = 1; }
      } else {                             /* towards the high neighbor */
        c->sendproc[ns] = c->procneigh[d][1];
        c->recvproc[ns] = c->procneigh[d][0];
        const int nbox = c->myloc[d] - ineed / 2;
        hi = (nbox + 1) * prd[d] / pg[d];
        lo = boxhi[d] - cutneigh;
        lo = ORC_MAX(lo, nbox * prd[d] / pg[d]);
        if(c->myloc[d] == pg[d] - 1) { c->pbc_any[ns] = 1; flag[ns] = -1; }
      }
      c->slablo[ns] = lo; c->slabhi[ns] = hi;
      ns++;
    }
  }
  c->nswap = ns;
  return 0;
}

static void comm_growsend(comm_t* c, int n)  /* ref/comm.cpp:887-891 */
{
  c->maxsend = (int)(1.5 * n);
  c->buf_send = (real*)realloc(c->buf_send, ((size_t)c->maxsend + 100) * sizeof(real));
}
static void comm_growrecv(comm_t* c, int n)  /* ref/comm.cpp:895-900 */
{
  c->maxrecv = (int)(1.5 * n);
  free(c->buf_recv);
  c->buf_recv = (real*)malloc((size_t)c->maxrecv * sizeof(real));
}
static void comm_growlist(comm_t* c, int iswap, int n)  /* ref/comm.cpp:904-909 */
{
  c->maxsendlist[iswap] = (int)(1.5 * n);
  c->sendlist[iswap] = (int*)realloc(c->sendlist[iswap], (size_t)c->maxsendlist[iswap] * sizeof(int));
}

/* ------------------------------------------------------------------------------------------------
 * Neighbor::setup, coord2bin, bindist, binatoms, build (ref/neighbor.cpp)
 * ---------------------------------------------------------------------------------------------- */

static real nb_bindist(const neigh_t* nb, int i, int j, int k)   /* ref/neighbor.cpp:456-482 */
{
  real dx = 0.0, dy = 0.0, dz = 0.0;
  if(i > 0) dx = (i - 1) * nb->binsizex; else if(i < 0) dx = (i + 1) * nb->binsizex;
  if(j > 0) dy = (j - 1) * nb->binsizey; else if(j < 0) dy = (j + 1) * nb->binsizey;
  if(k > 0) dz = (k - 1) * nb->binsizez; else if(k < 0) dz = (k + 1) * nb->binsizez;
  return dx * dx + dy * dy + dz * dz;
}

static int nb_setup(neigh_t* nb, const atom_t* atom, int ntypes)   /* ref/neighbor.cpp:318-452 */
{
  const real small = 1.0e-6, factor = 0.999;
  for(int i = 0; i < ntypes * ntypes; i++) nb->cutneighsq[i] = nb->cutneigh * nb->cutneigh;
  nb->xprd = atom->box.xprd; nb->yprd = atom->box.yprd; nb->zprd = atom->box.zprd;
  nb->binsizex = nb->xprd / nb->nbinx; nb->binsizey = nb->yprd / nb->nbiny; nb->binsizez = nb->zprd / nb->nbinz;
  nb->bininvx = 1.0 / nb->binsizex; nb->bininvy = 1.0 / nb->binsizey; nb->bininvz = 1.0 / nb->binsizez;

  real coord;
  int hi;
  coord = atom->box.xlo - nb->cutneigh - small * nb->xprd;
  nb->mbinxlo = (int)(coord * nb->bininvx); if(coord < 0.0) nb->mbinxlo -= 1;
  coord = atom->box.xhi + nb->cutneigh + small * nb->xprd;
  hi = (int)(coord * nb->bininvx);
  nb->mbinxlo -= 1; hi += 1; nb->mbinx = hi - nb->mbinxlo + 1;

  coord = atom->box.ylo - nb->cutneigh - small * nb->yprd;
  nb->mbinylo = (int)(coord * nb->bininvy); if(coord < 0.0) nb->mbinylo -= 1;
  coord = atom->box.yhi + nb->cutneigh + small * nb->yprd;
  hi = (int)(coord * nb->bininvy);
  nb->mbinylo -= 1; hi += 1; nb->mbiny = hi - nb->mbinylo + 1;

  coord = atom->box.zlo - nb->cutneigh - small * nb->zprd;
  nb->mbinzlo = (int)(coord * nb->bininvz); if(coord < 0.0) nb->mbinzlo -= 1;
  coord = atom->box.zhi + nb->cutneigh + small * nb->zprd;
  hi = (int)(coord * nb->bininvz);
  nb->mbinzlo -= 1; hi += 1; nb->mbinz = hi - nb->mbinzlo + 1;

  int nextx = (int)(nb->cutneigh * nb->bininvx); if(nextx * nb->binsizex < factor * nb->cutneigh) nextx++;
  int nexty = (int)(nb->cutneigh * nb->bininvy); if(nexty * nb->binsizey < factor * nb->cutneigh) nexty++;
  int nextz = (int)(nb->cutneigh * nb->bininvz); if(nextz * nb->binsizez < factor * nb->cutneigh) nextz++;

  free(nb->stencil);
  nb->stencil = (int*)malloc((size_t)(2 * nextz + 1) * (2 * nexty + 1) * (2 * nextx + 1) * sizeof(int));
  nb->nstencil = 0;
  const int newton_half = nb->halfneigh && nb->ghost_newton;
  int kstart = -nextz;
  if(newton_half) { kstart = 0; nb->stencil[nb->nstencil++] = 0; }   /* own bin first, then "upper right" bins */
  for(int k = kstart; k <= nextz; k++)
    for(int j = -nexty; j <= nexty; j++)
      for(int i = -nextx; i <= nextx; i++) {
        if(newton_half && !(k > 0 || j > 0 || (j == 0 && i > 0))) continue;
        if(nb_bindist(nb, i, j, k) < nb->cutneighsq[0])
          nb->stencil[nb->nstencil++] = k * nb->mbiny * nb->mbinx + j * nb->mbinx + i;
      }
  nb->mbins = nb->mbinx * nb->mbiny * nb->mbinz;
  free(nb->bincount); free(nb->bins);
  /* +1 slack bin: coord2bin's stray "+1" (ref/neighbor.cpp:299) can index one past the last bin */
  nb->bincount = (int*)calloc((size_t)nb->mbins + 1, sizeof(int));
  nb->bins = (int*)malloc(((size_t)nb->mbins + 1) * nb->atoms_per_bin * sizeof(int));
  return 0;
}

static inline int nb_coord2bin(const neigh_t* nb, real x, real y, real z)   /* ref/neighbor.cpp:274-300 */
{
  int ix, iy, iz;
  if(x >= nb->xprd) ix = (int)((x - nb->xprd) * nb->bininvx) + nb->nbinx - nb->mbinxlo;
  else if(x >= 0.0) ix = (int)(x * nb->bininvx) - nb->mbinxlo;
  else ix = (int)(x * nb->bininvx) - nb->mbinxlo - 1;
  if(y >= nb->yprd) iy = (int)((y - nb->yprd) * nb->bininvy) + nb->nbiny - nb->mbinylo;
  else if(y >= 0.0) iy = (int)(y * nb->bininvy) - nb->mbinylo;
  else iy = (int)(y * nb->bininvy) - nb->mbinylo - 1;
  if(z >= nb->zprd) iz = (int)((z - nb->zprd) * nb->bininvz) + nb->nbinz - nb->mbinzlo;
  else if(z >= 0.0) iz = (int)(z * nb->bininvz) - nb->mbinzlo;
  else iz = (int)(z * nb->bininvz) - nb->mbinzlo - 1;
  return iz * nb->mbiny * nb->mbinx + iy * nb->mbinx + ix + 1;
}

static void nb_binatoms(neigh_t* nb, const atom_t* atom, int count)   /* ref/neighbor.cpp:215-268 */
{
  const int nall = count < 0 ? atom->nlocal + atom->nghost : count;
  const real* x = atom->x;
  nb->xprd = atom->box.xprd; nb->yprd = atom->box.yprd; nb->zprd = atom->box.zprd;
  int resize = 1;
  while(resize) {
    resize = 0;
    memset(nb->bincount, 0, ((size_t)nb->mbins + 1) * sizeof(int));
    for(int i = 0; i < nall; i++) {
      const int ibin = nb_coord2bin(nb, x[i * PAD + 0], x[i * PAD + 1], x[i * PAD + 2]);
      if(nb->bincount[ibin] < nb->atoms_per_bin) {
        const int slot = nb->bincount[ibin]++;
        nb->bins[(size_t)ibin * nb->atoms_per_bin + slot] = i;
      } else resize = 1;
    }
    if(resize) {
      free(nb->bins);
      nb->atoms_per_bin *= 2;
      nb->bins = (int*)malloc(((size_t)nb->mbins + 1) * nb->atoms_per_bin * sizeof(int));
    }
  }
}

static void nb_build(neigh_t* nb, const atom_t* atom, int ntypes)   /* ref/neighbor.cpp:79-213 */
{
  nb->ncalls++;
  const int nlocal = atom->nlocal, nall = atom->nlocal + atom->nghost;
  if(nall > nb->nmax) {
    nb->nmax = nall;
    free(nb->numneigh); free(nb->neighbors);
    nb->numneigh = (int*)malloc((size_t)nb->nmax * sizeof(int));
    nb->neighbors = (int*)malloc((size_t)nb->nmax * nb->maxneighs * sizeof(int));
  }
  nb_binatoms(nb, atom, -1);
  const real* x = atom->x;
  const int* type = atom->type;
  const int half = nb->halfneigh, gn = nb->ghost_newton;
  int resize = 1;
  while(resize) {
    int new_max = nb->maxneighs;
    resize = 0;
    for(int i = 0; i < nlocal; i++) {
      int* row = &nb->neighbors[(size_t)i * nb->maxneighs];
      int n = 0;
      const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
      const int ti = type[i];
      const int ibin = nb_coord2bin(nb, xi, yi, zi);
      for(int s = 0; s < nb->nstencil; s++) {
        const int jbin = ibin + nb->stencil[s];
        const int* slots = &nb->bins[(size_t)jbin * nb->atoms_per_bin];
        const int cnt = nb->bincount[jbin];
        for(int m = 0; m < cnt; m++) {
          const int j = slots[m];
          if(jbin == ibin) {
            /* own bin: never self; half lists keep j>i only; with ghost-newton a ghost j is kept only
               if it lies "above" i in (z,y,x) lexicographic order  (ref/neighbor.cpp:153-157) */
            if(j == i) continue;
            if(half && !gn && j < i) continue;
            if(half && gn) {
              if(j < i) continue;
              if(j >= nlocal) {
                const real xj = x[j * PAD + 0], yj = x[j * PAD + 1], zj = x[j * PAD + 2];
                if(zj < zi || (zj == zi && yj < yi) || (zj == zi && yj == yi && xj < xi)) continue;
              }
            }
          } else {
            if(half && !gn && j < i) continue;     /* ref/neighbor.cpp:171 */
          }
          const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
          const real rsq = dx * dx + dy * dy + dz * dz;
          if(rsq <= nb->cutneighsq[ti * ntypes + type[j]]) {
            if(n < nb->maxneighs) row[n] = j;      /* store guarded (the reference overruns, then retries) */
            n++;
          }
        }
      }
      nb->numneigh[i] = n;
      if(n >= nb->maxneighs) { resize = 1; if(n >= new_max) new_max = n; }
    }
    if(resize) {
      nb->maxneighs = new_max * 1.2;
      free(nb->neighbors);
      nb->neighbors = (int*)malloc((size_t)nb->nmax * nb->maxneighs * sizeof(int));
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * ForceLJ (ref/force_lj.cpp)
 * ---------------------------------------------------------------------------------------------- */

void orc_lj_force_full(const real* x, const int* type, int nlocal, const int* neighbors, const int* numneigh,
                       int maxneighs, int ntypes, const real* cutforcesq, const real* sigma6,
                       const real* epsilon, int evflag, real* f, real* eng_vdwl, real* virial)
{
  /* ref/force_lj.cpp:366-449 */
  real t_eng = 0, t_vir = 0;
  for(int i = 0; i < nlocal; i++) { f[i * PAD + 0] = 0.0; f[i * PAD + 1] = 0.0; f[i * PAD + 2] = 0.0; }
  for(int i = 0; i < nlocal; i++) {
    const int* row = &neighbors[(size_t)i * maxneighs];
    const int nn = numneigh[i];
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    const int ti = type[i];
    real fix = 0, fiy = 0, fiz = 0;
    for(int k = 0; k < nn; k++) {
      const int j = row[k];
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = ti * ntypes + type[j];
      if(rsq < cutforcesq[tij]) {
        const real sr2 = 1.0 / rsq;
        const real sr6 = sr2 * sr2 * sr2 * sigma6[tij];
        const real force = 48.0 * sr6 * (sr6 - 0.5) * sr2 * epsilon[tij];
        fix += dx * force; fiy += dy * force; fiz += dz * force;
        if(evflag) {
          t_eng += sr6 * (sr6 - 1.0) * epsilon[tij];
          t_vir += (dx * dx + dy * dy + dz * dz) * force;
        }
      }
    }
    f[i * PAD + 0] += fix; f[i * PAD + 1] += fiy; f[i * PAD + 2] += fiz;
  }
  t_eng *= 4.0;        /* both i-j and j-i were visited: thermo's e_scale=0.5 halves it (thermo.cpp:62) */
  t_vir *= 0.5;
  *eng_vdwl += t_eng;
  *virial += t_vir;
}

void orc_lj_force_half(const real* x, const int* type, int nlocal, int nall, const int* neighbors,
                       const int* numneigh, int maxneighs, int ntypes, const real* cutforcesq,
                       const real* sigma6, const real* epsilon, int evflag, int ghost_newton,
                       real* f, real* eng_vdwl, real* virial)
{
  /* ref/force_lj.cpp:185-263 (serial) == :271-357 (threaded) arithmetic */
  for(int i = 0; i < nall; i++) { f[i * PAD + 0] = 0.0; f[i * PAD + 1] = 0.0; f[i * PAD + 2] = 0.0; }
  real t_energy = 0, t_virial = 0;
  for(int i = 0; i < nlocal; i++) {
    const int* row = &neighbors[(size_t)i * maxneighs];
    const int nn = numneigh[i];
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    const int ti = type[i];
    real fix = 0.0, fiy = 0.0, fiz = 0.0;
    for(int k = 0; k < nn; k++) {
      const int j = row[k];
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = ti * ntypes + type[j];
      if(rsq < cutforcesq[tij]) {
        const real sr2 = 1.0 / rsq;
        const real sr6 = sr2 * sr2 * sr2 * sigma6[tij];
        const real force = 48.0 * sr6 * (sr6 - 0.5) * sr2 * epsilon[tij];
        fix += dx * force; fiy += dy * force; fiz += dz * force;
        const int owned = ghost_newton || j < nlocal;
        if(owned) { f[j * PAD + 0] -= dx * force; f[j * PAD + 1] -= dy * force; f[j * PAD + 2] -= dz * force; }
        if(evflag) {
          const real scale = owned ? 1.0 : 0.5;
          t_energy += scale * (4.0 * sr6 * (sr6 - 1.0)) * epsilon[tij];
          t_virial += scale * (dx * dx + dy * dy + dz * dz) * force;
        }
      }
    }
    f[i * PAD + 0] += fix; f[i * PAD + 1] += fiy; f[i * PAD + 2] += fiz;
  }
  *eng_vdwl += t_energy;
  *virial += t_virial;
}

static void lj_force_original(const real* x, const int* type, int nlocal, int nall, const int* neighbors,
                              const int* numneigh, int maxneighs, int ntypes, const real* cutforcesq,
                              const real* sigma6, const real* epsilon, int evflag, real* f, real* eng, real* vir)
{
  /* ref/force_lj.cpp:118-176 (--half_neigh -1) */
  for(int i = 0; i < nall * PAD; i++) f[i] = 0.0;
  for(int i = 0; i < nlocal; i++) {
    const int* row = &neighbors[(size_t)i * maxneighs];
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    for(int k = 0; k < numneigh[i]; k++) {
      const int j = row[k];
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = type[i] * ntypes + type[j];
      if(rsq < cutforcesq[tij]) {
        const real sr2 = 1.0 / rsq;
        const real sr6 = sr2 * sr2 * sr2 * sigma6[tij];
        const real force = 48.0 * sr6 * (sr6 - 0.5) * sr2 * epsilon[tij];
        f[i * PAD + 0] += dx * force; f[i * PAD + 1] += dy * force; f[i * PAD + 2] += dz * force;
        f[j * PAD + 0] -= dx * force; f[j * PAD + 1] -= dy * force; f[j * PAD + 2] -= dz * force;
        if(evflag) {
          *eng += (4.0 * sr6 * (sr6 - 1.0)) * epsilon[tij];
          *vir += (dx * dx + dy * dy + dz * dz) * force;
        }
      }
    }
  }
}

int orc_neighbor_brute_full(const real* x, int nlocal, int nall, real cutneighsq, int maxneighs,
                            int* neighbors, int* numneigh)
{
  int maxfound = 0;
  for(int i = 0; i < nlocal; i++) {
    int n = 0;
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    for(int j = 0; j < nall; j++) {
      if(j == i) continue;
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      if(rsq <= cutneighsq) { if(n < maxneighs) neighbors[(size_t)i * maxneighs + n] = j; n++; }
    }
    numneigh[i] = n;
    if(n > maxfound) maxfound = n;
  }
  return maxfound;
}

/* ------------------------------------------------------------------------------------------------
 * ForceEAM: funcfl file -> arrays -> splines (ref/force_eam.cpp:505-793)
 * ---------------------------------------------------------------------------------------------- */

static int eam_grab(FILE* fp, int n, real* list)   /* ref/force_eam.cpp:801-814 */
{
  char line[1024];
  int i = 0;
  while(i < n) {
    if(!fgets(line, 1024, fp)) return 1;
    for(char* p = strtok(line, " \t\n\r\f"); p; p = strtok(NULL, " \t\n\r\f")) list[i++] = atof(p);
  }
  return 0;
}

static int eam_read_file(force_t* F, const char* filename)   /* ref/force_eam.cpp:505-582 */
{
  char line[1024];
  FILE* fp = fopen(filename, "r");
  if(!fp) { set_err("Can't open EAM Potential file: %s", filename); return 1; }
  int tmp;
  if(!fgets(line, 1024, fp) || !fgets(line, 1024, fp)) { fclose(fp); return 1; }
  sscanf(line, "%d %lg", &tmp, &F->fl_mass);
  if(!fgets(line, 1024, fp)) { fclose(fp); return 1; }
  sscanf(line, "%d %lg %d %lg %lg", &F->fl_nrho, &F->fl_drho, &F->fl_nr, &F->fl_dr, &F->fl_cut);
  F->mass = F->fl_mass;
  F->fl_frho = (real*)calloc(F->fl_nrho + 1, sizeof(real));
  F->fl_rhor = (real*)calloc(F->fl_nr + 1, sizeof(real));
  F->fl_zr = (real*)calloc(F->fl_nr + 1, sizeof(real));
  /* file order: F(rho), Z(r), rho(r); then shift everything to 1-based (ref :575-579) */
  if(eam_grab(fp, F->fl_nrho, F->fl_frho) || eam_grab(fp, F->fl_nr, F->fl_zr) || eam_grab(fp, F->fl_nr, F->fl_rhor)) {
    fclose(fp); set_err("EAM potential file %s is truncated", filename); return 1;
  }
  fclose(fp);
  for(int i = F->fl_nrho; i > 0; i--) F->fl_frho[i] = F->fl_frho[i - 1];
  for(int i = F->fl_nr; i > 0; i--) F->fl_rhor[i] = F->fl_rhor[i - 1];
  for(int i = F->fl_nr; i > 0; i--) F->fl_zr[i] = F->fl_zr[i - 1];
  return 0;
}

/* 4-point Lagrange re-grid of a 1-based table `src` (n entries, spacing dsrc) at abscissa r */
static double eam_lagrange4(const real* src, int n, double dsrc, double r)
{
  const double sixth = 1.0 / 6.0;
  double p = r / dsrc + 1.0;
  int k = (int)p;
  k = ORC_MIN(k, n - 2);
  k = ORC_MAX(k, 2);
  p -= k;
  p = ORC_MIN(p, 2.0);
  const double c1 = -sixth * p * (p - 1.0) * (p - 2.0);
  const double c2 = 0.5 * (p * p - 1.0) * (p - 2.0);
  const double c3 = -0.5 * p * (p + 1.0) * (p - 2.0);
  const double c4 = sixth * p * (p * p - 1.0);
  return c1 * src[k - 1] + c2 * src[k] + c3 * src[k + 1] + c4 * src[k + 2];
}

static void eam_file2array(force_t* F)   /* ref/force_eam.cpp:589-728 */
{
  F->dr = F->fl_dr > 0.0 ? F->fl_dr : 0.0;
  F->drho = F->fl_drho > 0.0 ? F->fl_drho : 0.0;
  const double rmax = ORC_MAX(0.0, (F->fl_nr - 1) * F->fl_dr);
  const double rhomax = ORC_MAX(0.0, (F->fl_nrho - 1) * F->fl_drho);
  F->nr = (int)(rmax / F->dr + 0.5);
  F->nrho = (int)(rhomax / F->drho + 0.5);
  F->frho = (real*)calloc(F->nrho + 1, sizeof(real));
  F->rhor = (real*)calloc(F->nr + 1, sizeof(real));
  F->z2r = (real*)calloc(F->nr + 1, sizeof(real));
  for(int m = 1; m <= F->nrho; m++) {
    const double r = (m - 1) * F->drho;
    F->frho[m] = eam_lagrange4(F->fl_frho, F->fl_nrho, F->fl_drho, r);
  }
  for(int m = 1; m <= F->nr; m++) {
    const double r = (m - 1) * F->dr;
    F->rhor[m] = eam_lagrange4(F->fl_rhor, F->fl_nr, F->fl_dr, r);
  }
  for(int m = 1; m <= F->nr; m++) {
    const double r = (m - 1) * F->dr;
    const double zri = eam_lagrange4(F->fl_zr, F->fl_nr, F->fl_dr, r);
    const double zrj = eam_lagrange4(F->fl_zr, F->fl_nr, F->fl_dr, r);
    F->z2r[m] = 27.2 * 0.529 * zri * zrj;       /* Hartree*Bohr -> eV*Angstrom, as the reference spells it */
  }
}

static void eam_interpolate(int n, real delta, const real* f, real* s)   /* ref/force_eam.cpp:765-793 */
{
  for(int m = 1; m <= n; m++) s[m * 7 + 6] = f[m];
  s[1 * 7 + 5] = s[2 * 7 + 6] - s[1 * 7 + 6];
  s[2 * 7 + 5] = 0.5 * (s[3 * 7 + 6] - s[1 * 7 + 6]);
  s[(n - 1) * 7 + 5] = 0.5 * (s[n * 7 + 6] - s[(n - 2) * 7 + 6]);
  s[n * 7 + 5] = s[n * 7 + 6] - s[(n - 1) * 7 + 6];
  for(int m = 3; m <= n - 2; m++)
    s[m * 7 + 5] = ((s[(m - 2) * 7 + 6] - s[(m + 2) * 7 + 6]) + 8.0 * (s[(m + 1) * 7 + 6] - s[(m - 1) * 7 + 6])) / 12.0;
  for(int m = 1; m <= n - 1; m++) {
    s[m * 7 + 4] = 3.0 * (s[(m + 1) * 7 + 6] - s[m * 7 + 6]) - 2.0 * s[m * 7 + 5] - s[(m + 1) * 7 + 5];
    s[m * 7 + 3] = s[m * 7 + 5] + s[(m + 1) * 7 + 5] - 2.0 * (s[(m + 1) * 7 + 6] - s[m * 7 + 6]);
  }
  s[n * 7 + 4] = 0.0;
  s[n * 7 + 3] = 0.0;
  for(int m = 1; m <= n; m++) {
    s[m * 7 + 2] = s[m * 7 + 5] / delta;
    s[m * 7 + 1] = 2.0 * s[m * 7 + 4] / delta;
    s[m * 7 + 0] = 3.0 * s[m * 7 + 3] / delta;
  }
}

static void eam_array2spline(force_t* F)   /* ref/force_eam.cpp:732-761 */
{
  const int nt2 = F->ntypes * F->ntypes;
  F->rdr = 1.0 / F->dr;
  F->rdrho = 1.0 / F->drho;
  F->nrho_tot = (F->nrho + 1) * 7 + 64; F->nrho_tot -= F->nrho_tot % 64;
  F->nr_tot = (F->nr + 1) * 7 + 64; F->nr_tot -= F->nr_tot % 64;
  F->frho_spline = (real*)calloc((size_t)nt2 * F->nrho_tot, sizeof(real));
  F->rhor_spline = (real*)calloc((size_t)nt2 * F->nr_tot, sizeof(real));
  F->z2r_spline = (real*)calloc((size_t)nt2 * F->nr_tot, sizeof(real));
  eam_interpolate(F->nrho, F->drho, F->frho, F->frho_spline);
  eam_interpolate(F->nr, F->dr, F->rhor, F->rhor_spline);
  eam_interpolate(F->nr, F->dr, F->z2r, F->z2r_spline);
  for(int t = 1; t < nt2; t++) {
    memcpy(&F->frho_spline[(size_t)t * F->nrho_tot], F->frho_spline, F->nrho_tot * sizeof(real));
    memcpy(&F->rhor_spline[(size_t)t * F->nr_tot], F->rhor_spline, F->nr_tot * sizeof(real));
    memcpy(&F->z2r_spline[(size_t)t * F->nr_tot], F->z2r_spline, F->nr_tot * sizeof(real));
  }
}

static int eam_setup(force_t* F)   /* ref/force_eam.cpp:74-79, 460-496 */
{
  if(eam_read_file(F, "Cu_u6.eam")) return 1;
  F->cutmax = F->fl_cut;
  for(int i = 0; i < F->ntypes * F->ntypes; i++) F->cutforcesq[i] = F->cutmax * F->cutmax;
  eam_file2array(F);
  eam_array2spline(F);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * world helpers
 * ---------------------------------------------------------------------------------------------- */

static void stamp(orc_world* w) { w->t_prev = wall(); }
static void stamp_to(orc_world* w, int which)
{
  const double t = wall();
  w->timer[which] += t - w->t_prev;
  w->t_prev = t;
}

/* ------------------------------------------------------------------------------------------------
 * Comm::communicate / reverse_communicate (ref/comm.cpp:276-355) on virtual ranks
 * ---------------------------------------------------------------------------------------------- */

void orc_communicate(orc_world* w)
{
  const int nswap = w->r[0].comm.nswap;
  for(int s = 0; s < nswap; s++) {
    for(int p = 0; p < w->nprocs; p++) {          /* Atom::pack_comm  ref/atom.cpp:135-158 */
      rank_t* R = &w->r[p];
      const comm_t* c = &R->comm;
      const real* x = R->atom.x;
      const int* list = c->sendlist[s];
      real* buf = c->buf_send;
      if(!c->pbc_any[s]) {
        for(int i = 0; i < c->sendnum[s]; i++) {
          const int j = list[i];
          buf[3 * i] = x[j * PAD + 0]; buf[3 * i + 1] = x[j * PAD + 1]; buf[3 * i + 2] = x[j * PAD + 2];
        }
      } else {
        for(int i = 0; i < c->sendnum[s]; i++) {
          const int j = list[i];
          buf[3 * i] = x[j * PAD + 0] + c->pbc_flagx[s] * R->atom.box.xprd;
          buf[3 * i + 1] = x[j * PAD + 1] + c->pbc_flagy[s] * R->atom.box.yprd;
          buf[3 * i + 2] = x[j * PAD + 2] + c->pbc_flagz[s] * R->atom.box.zprd;
        }
      }
    }
    for(int p = 0; p < w->nprocs; p++) {          /* MPI_Sendrecv + Atom::unpack_comm  ref/atom.cpp:160-170 */
      rank_t* R = &w->r[p];
      const comm_t* c = &R->comm;
      const real* buf = w->r[c->recvproc[s]].comm.buf_send;   /* what my recv partner packed for me */
      real* x = R->atom.x;
      const int first = c->firstrecv[s];
      for(int i = 0; i < c->recvnum[s]; i++) {
        x[(first + i) * PAD + 0] = buf[3 * i]; x[(first + i) * PAD + 1] = buf[3 * i + 1]; x[(first + i) * PAD + 2] = buf[3 * i + 2];
      }
    }
  }
}

void orc_reverse_communicate(orc_world* w)
{
  const int nswap = w->r[0].comm.nswap;
  for(int s = nswap - 1; s >= 0; s--) {
    for(int p = 0; p < w->nprocs; p++) {          /* Atom::pack_reverse  ref/atom.cpp:172-182 */
      rank_t* R = &w->r[p];
      const comm_t* c = &R->comm;
      const real* f = R->atom.f;
      const int first = c->firstrecv[s];
      for(int i = 0; i < c->recvnum[s]; i++) {
        c->buf_send[3 * i] = f[(first + i) * PAD + 0];
        c->buf_send[3 * i + 1] = f[(first + i) * PAD + 1];
        c->buf_send[3 * i + 2] = f[(first + i) * PAD + 2];
      }
    }
    for(int p = 0; p < w->nprocs; p++) {          /* sent to recvproc, received from sendproc; unpack_reverse :184-195 */
      rank_t* R = &w->r[p];
      const comm_t* c = &R->comm;
      const real* buf = w->r[c->sendproc[s]].comm.buf_send;
      real* f = R->atom.f;
      const int* list = c->sendlist[s];
      for(int i = 0; i < c->sendnum[s]; i++) {
        const int j = list[i];
        f[j * PAD + 0] += buf[3 * i]; f[j * PAD + 1] += buf[3 * i + 1]; f[j * PAD + 2] += buf[3 * i + 2];
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Comm::exchange (ref/comm.cpp:364-597), single-thread semantics, on virtual ranks
 * ---------------------------------------------------------------------------------------------- */

static void orc_exchange_all(orc_world* w);

void orc_exchange(orc_world* w)
{
  if(w->safe_exchange) { orc_exchange_all(w); return; }          /* ref/comm.cpp:366-367 */
  for(int p = 0; p < w->nprocs; p++) atom_pbc(&w->r[p].atom);
  for(int d = 0; d < 3; d++) {
    if(w->r[0].comm.procgrid[d] == 1) continue;
    /* phase 1: every rank removes its leavers (packing 7 values each) and closes the holes */
    for(int p = 0; p < w->nprocs; p++) {
      rank_t* R = &w->r[p];
      atom_t* a = &R->atom;
      comm_t* c = &R->comm;
      const real lo = d == 0 ? a->box.xlo : (d == 1 ? a->box.ylo : a->box.zlo);
      const real hi = d == 0 ? a->box.xhi : (d == 1 ? a->box.yhi : a->box.zhi);
      const int nlocal = a->nlocal;
      int* stays = (int*)malloc(((size_t)nlocal + 1) * sizeof(int));
      int* leavers = (int*)malloc(((size_t)nlocal + 1) * sizeof(int));
      int nsend = 0;
      for(int i = 0; i < nlocal; i++) {
        const real v = a->x[i * PAD + d];
        if(v < lo || v >= hi) { leavers[nsend++] = i; stays[i] = 0; } else stays[i] = 1;
      }
      if(nsend * 7 > c->maxsend) comm_growsend(c, nsend * 7);
      if(nsend > c->tag_cap) { c->tag_cap = nsend + 1024; c->tag_send = (int*)realloc(c->tag_send, c->tag_cap * sizeof(int)); }
      /* stayers of the tail [nlocal-nsend, nlocal) fill, in ascending order, the holes left below it */
      int j = nlocal - nsend;
      for(int k = 0; k < nsend; k++) {
        const int i = leavers[k];
        real* b = &c->buf_send[k * 7];                  /* Atom::pack_exchange  ref/atom.cpp:228-239 */
        b[0] = a->x[i * PAD + 0]; b[1] = a->x[i * PAD + 1]; b[2] = a->x[i * PAD + 2];
        b[3] = a->v[i * PAD + 0]; b[4] = a->v[i * PAD + 1]; b[5] = a->v[i * PAD + 2];
        b[6] = a->type[i];
        c->tag_send[k] = a->tag[i];
        if(i < nlocal - nsend) {
          while(!stays[j]) j++;
          atom_copy(a, j++, i);
        }
      }
      a->nlocal = nlocal - nsend;
      c->nsend_now = nsend;
      free(stays); free(leavers);
    }
    /* phase 2: receive from the +1 neighbor what it sent towards -1 (and, if the grid is wider
       than 2, from the -1 neighbor too); keep the atoms that fall inside my [lo,hi) */
    for(int p = 0; p < w->nprocs; p++) {
      rank_t* R = &w->r[p];
      atom_t* a = &R->atom;
      comm_t* c = &R->comm;
      const real lo = d == 0 ? a->box.xlo : (d == 1 ? a->box.ylo : a->box.zlo);
      const real hi = d == 0 ? a->box.xhi : (d == 1 ? a->box.yhi : a->box.zhi);
      const int nsrc = c->procgrid[d] > 2 ? 2 : 1;
      for(int s = 0; s < nsrc; s++) {
        const comm_t* from = &w->r[c->procneigh[d][s == 0 ? 1 : 0]].comm;
        for(int i = 0; i < from->nsend_now; i++) {
          const real* b = &from->buf_send[i * 7];
          if(b[d] >= lo && b[d] < hi) {                 /* Atom::unpack_exchange  ref/atom.cpp:241-254 */
            if(a->nlocal == a->nmax) atom_grow(a);
            const int n = a->nlocal++;
            a->x[n * PAD + 0] = b[0]; a->x[n * PAD + 1] = b[1]; a->x[n * PAD + 2] = b[2];
            a->v[n * PAD + 0] = b[3]; a->v[n * PAD + 1] = b[4]; a->v[n * PAD + 2] = b[5];
            a->type[n] = b[6];
            a->tag[n] = from->tag_send[i];
          }
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Comm::exchange_all (ref/comm.cpp:599-689): "safe exchange" — the atoms that left my box in dimension d are offered to EVERY rank
 * within `need[d]` sub-domains in that dimension, nearest first, alternating -i / +i (sendproc_exc / recvproc_exc of
 * MPI_Cart_shift(cartesian, d, i), ref/comm.cpp:176-180: even swaps send to me-i and receive from me+i, odd ones the other way);
 * a swap runs only while `ineed < procgrid[d] - 1` (:656), so no rank is offered the same buffer twice. Every receiver keeps what
 * falls inside its [lo, hi) of that dimension (:675-681). The send buffer is packed once per dimension (:625-637).
 * ---------------------------------------------------------------------------------------------- */
static void orc_exchange_all(orc_world* w)
{
  for(int p = 0; p < w->nprocs; p++) atom_pbc(&w->r[p].atom);
  for(int d = 0; d < 3; d++) {
    const int pgd = w->r[0].comm.procgrid[d];
    if(pgd == 1) continue;
    for(int p = 0; p < w->nprocs; p++) {                /* pack the leavers, close the holes: as orc_exchange */
      rank_t* R = &w->r[p];
      atom_t* a = &R->atom;
      comm_t* c = &R->comm;
      const real lo = d == 0 ? a->box.xlo : (d == 1 ? a->box.ylo : a->box.zlo);
      const real hi = d == 0 ? a->box.xhi : (d == 1 ? a->box.yhi : a->box.zhi);
      const int nlocal = a->nlocal;
      int* stays = (int*)malloc(((size_t)nlocal + 1) * sizeof(int));
      int* leavers = (int*)malloc(((size_t)nlocal + 1) * sizeof(int));
      int nsend = 0;
      for(int i = 0; i < nlocal; i++) {
        const real v = a->x[i * PAD + d];
        if(v < lo || v >= hi) { leavers[nsend++] = i; stays[i] = 0; } else stays[i] = 1;
      }
      if(nsend * 7 > c->maxsend) comm_growsend(c, nsend * 7);
      if(nsend > c->tag_cap) { c->tag_cap = nsend + 1024; c->tag_send = (int*)realloc(c->tag_send, c->tag_cap * sizeof(int)); }
      int j = nlocal - nsend;
      for(int k = 0; k < nsend; k++) {
        const int i = leavers[k];
        real* b = &c->buf_send[k * 7];
        b[0] = a->x[i * PAD + 0]; b[1] = a->x[i * PAD + 1]; b[2] = a->x[i * PAD + 2];
        b[3] = a->v[i * PAD + 0]; b[4] = a->v[i * PAD + 1]; b[5] = a->v[i * PAD + 2];
        b[6] = a->type[i];
        c->tag_send[k] = a->tag[i];
        if(i < nlocal - nsend) {
          while(!stays[j]) j++;
          atom_copy(a, j++, i);
        }
      }
      a->nlocal = nlocal - nsend;
      c->nsend_now = nsend;
      free(stays); free(leavers);
    }
    for(int ineed = 0; ineed < 2 * w->r[0].comm.need[d]; ineed++) {
      if(!(ineed < pgd - 1)) continue;                  /* ref/comm.cpp:656 */
      const int dist = ineed / 2 + 1;
      for(int p = 0; p < w->nprocs; p++) {
        rank_t* R = &w->r[p];
        atom_t* a = &R->atom;
        comm_t* c = &R->comm;
        const real lo = d == 0 ? a->box.xlo : (d == 1 ? a->box.ylo : a->box.zlo);
        const real hi = d == 0 ? a->box.xhi : (d == 1 ? a->box.yhi : a->box.zhi);
        int loc[3] = {c->myloc[0], c->myloc[1], c->myloc[2]};
        loc[d] += (ineed % 2 == 0) ? dist : -dist;      /* recvproc_exc: me+i for even swaps, me-i for odd ones */
        const comm_t* from = &w->r[cart_rank(c->procgrid, loc[0], loc[1], loc[2])].comm;
        for(int i = 0; i < from->nsend_now; i++) {
          const real* b = &from->buf_send[i * 7];
          if(b[d] >= lo && b[d] < hi) {
            if(a->nlocal == a->nmax) atom_grow(a);
            const int n = a->nlocal++;
            a->x[n * PAD + 0] = b[0]; a->x[n * PAD + 1] = b[1]; a->x[n * PAD + 2] = b[2];
            a->v[n * PAD + 0] = b[3]; a->v[n * PAD + 1] = b[4]; a->v[n * PAD + 2] = b[5];
            a->type[n] = b[6];
            a->tag[n] = from->tag_send[i];
          }
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Comm::borders (ref/comm.cpp:700-883)
 * ---------------------------------------------------------------------------------------------- */

void orc_borders(orc_world* w)
{
  for(int p = 0; p < w->nprocs; p++) w->r[p].atom.nghost = 0;
  int* nfirst = (int*)calloc(w->nprocs, sizeof(int));
  int* nlast = (int*)calloc(w->nprocs, sizeof(int));
  int iswap = 0;
  for(int d = 0; d < 3; d++) {
    for(int p = 0; p < w->nprocs; p++) nlast[p] = 0;
    for(int ineed = 0; ineed < 2 * w->r[0].comm.need[d]; ineed++) {
      for(int p = 0; p < w->nprocs; p++) {
        rank_t* R = &w->r[p];
        atom_t* a = &R->atom;
        comm_t* c = &R->comm;
        /* first swap of a pair scans owned + all ghosts so far; later pairs only the newly arrived */
        if(ineed % 2 == 0) { nfirst[p] = nlast[p]; nlast[p] = a->nlocal + a->nghost; }
        const real lo = c->slablo[iswap], hi = c->slabhi[iswap];
        int nsend = 0;
        for(int i = nfirst[p]; i < nlast[p]; i++) {
          const real v = a->x[i * PAD + d];
          if(v >= lo && v <= hi) {
            if(nsend >= c->maxsendlist[iswap]) comm_growlist(c, iswap, nsend + 1);
            c->sendlist[iswap][nsend++] = i;
          }
        }
        if(nsend * 4 > c->maxsend) comm_growsend(c, nsend * 4);
        for(int k = 0; k < nsend; k++) {                 /* Atom::pack_border  ref/atom.cpp:197-214 */
          const int i = c->sendlist[iswap][k];
          real* b = &c->buf_send[k * 4];
          if(!c->pbc_any[iswap]) {
            b[0] = a->x[i * PAD + 0]; b[1] = a->x[i * PAD + 1]; b[2] = a->x[i * PAD + 2];
          } else {
            b[0] = a->x[i * PAD + 0] + c->pbc_flagx[iswap] * a->box.xprd;
            b[1] = a->x[i * PAD + 1] + c->pbc_flagy[iswap] * a->box.yprd;
            b[2] = a->x[i * PAD + 2] + c->pbc_flagz[iswap] * a->box.zprd;
          }
          b[3] = a->type[i];
        }
        c->nsend_now = nsend;
      }
      for(int p = 0; p < w->nprocs; p++) {
        rank_t* R = &w->r[p];
        atom_t* a = &R->atom;
        comm_t* c = &R->comm;
        const comm_t* from = &w->r[c->recvproc[iswap]].comm;
        const int nrecv = from->nsend_now;
        const int n = a->nlocal + a->nghost;
        for(int i = 0; i < nrecv; i++) {                 /* Atom::unpack_border  ref/atom.cpp:216-226 */
          while(n + i >= a->nmax) atom_grow(a);
          const real* b = &from->buf_send[i * 4];
          a->x[(n + i) * PAD + 0] = b[0]; a->x[(n + i) * PAD + 1] = b[1]; a->x[(n + i) * PAD + 2] = b[2];
          a->type[n + i] = b[3];
        }
        c->nrecv_now = nrecv;
      }
      for(int p = 0; p < w->nprocs; p++) {
        rank_t* R = &w->r[p];
        comm_t* c = &R->comm;
        c->sendnum[iswap] = c->nsend_now;
        c->recvnum[iswap] = c->nrecv_now;
        c->firstrecv[iswap] = R->atom.nlocal + R->atom.nghost;
        R->atom.nghost += c->nrecv_now;
      }
      iswap++;
    }
  }
  /* buffers must also hold a full reverse swap (3 values per atom)  ref/comm.cpp:870-882 */
  for(int p = 0; p < w->nprocs; p++) {
    comm_t* c = &w->r[p].comm;
    int m = 0;
    for(int s = 0; s < c->nswap; s++) { m = ORC_MAX(m, 3 * c->recvnum[s]); m = ORC_MAX(m, 3 * c->sendnum[s]); }
    if(m > c->maxsend) comm_growsend(c, m);
  }
  free(nfirst); free(nlast);
}

/* ------------------------------------------------------------------------------------------------
 * Atom::sort (ref/atom.cpp:355-421): counting sort of the owned atoms by bin, x/v/type only
 * ---------------------------------------------------------------------------------------------- */

void orc_sort(orc_world* w)
{
  for(int p = 0; p < w->nprocs; p++) {
    rank_t* R = &w->r[p];
    atom_t* a = &R->atom;
    neigh_t* nb = &R->neigh;
    nb_binatoms(nb, a, a->nlocal);
    if(a->copy_size < a->nmax) {
      free(a->x_copy); free(a->v_copy); free(a->type_copy); free(a->tag_copy);
      a->x_copy = (real*)malloc((size_t)a->nmax * PAD * sizeof(real));
      a->v_copy = (real*)malloc((size_t)a->nmax * PAD * sizeof(real));
      a->type_copy = (int*)malloc((size_t)a->nmax * sizeof(int));
      a->tag_copy = (int*)malloc((size_t)a->nmax * sizeof(int));
      a->copy_size = a->nmax;
    }
    int dst = 0;
    for(int b = 0; b < nb->mbins; b++) {
      const int* slots = &nb->bins[(size_t)b * nb->atoms_per_bin];
      for(int k = 0; k < nb->bincount[b]; k++, dst++) {
        const int src = slots[k];
        for(int d = 0; d < 3; d++) {
          a->x_copy[dst * PAD + d] = a->x[src * PAD + d];
          a->v_copy[dst * PAD + d] = a->v[src * PAD + d];
        }
        a->type_copy[dst] = a->type[src];
        a->tag_copy[dst] = a->tag[src];
      }
    }
    real* t;
    int* ti;
    t = a->x; a->x = a->x_copy; a->x_copy = t;
    t = a->v; a->v = a->v_copy; a->v_copy = t;
    ti = a->type; a->type = a->type_copy; a->type_copy = ti;
    ti = a->tag; a->tag = a->tag_copy; a->tag_copy = ti;
  }
}

void orc_neighbor_build(orc_world* w)
{
  for(int p = 0; p < w->nprocs; p++) nb_build(&w->r[p].neigh, &w->r[p].atom, w->ntypes);
}

/* ------------------------------------------------------------------------------------------------
 * ForceEAM::compute (ref/force_eam.cpp:94-449) with its own fp halo (:851-913)
 * ---------------------------------------------------------------------------------------------- */

static void eam_ensure(force_t* F, const atom_t* a)
{
  if(a->nmax > F->eam_nmax) {
    F->eam_nmax = a->nmax;
    free(F->rho); free(F->fp);
    F->rho = (real*)calloc(F->eam_nmax, sizeof(real));
    F->fp = (real*)calloc(F->eam_nmax, sizeof(real));
  }
}

static inline void eam_knot(const force_t* F, real r, int* m_out, real* p_out)
{
  real p = r * F->rdr + 1.0;
  int m = (int)p;
  m = m < F->nr - 1 ? m : F->nr - 1;
  p -= m;
  p = p < 1.0 ? p : 1.0;
  *m_out = m; *p_out = p;
}

static void eam_fp_halo(orc_world* w)   /* ref/force_eam.cpp:851-913 */
{
  const int nswap = w->r[0].comm.nswap;
  for(int s = 0; s < nswap; s++) {
    for(int p = 0; p < w->nprocs; p++) {
      rank_t* R = &w->r[p];
      const comm_t* c = &R->comm;
      for(int i = 0; i < c->sendnum[s]; i++) c->buf_send[i] = R->force.fp[c->sendlist[s][i]];
    }
    for(int p = 0; p < w->nprocs; p++) {
      rank_t* R = &w->r[p];
      const comm_t* c = &R->comm;
      const real* buf = w->r[c->recvproc[s]].comm.buf_send;
      for(int i = 0; i < c->recvnum[s]; i++) R->force.fp[c->firstrecv[s] + i] = buf[i];
    }
  }
}

/* sweep 1 of the full-list path: density, embedding derivative fp, embedding energy (ref :304-349) */
static void eam_full_density(rank_t* R, int ntypes, real* evdwl)
{
  force_t* F = &R->force;
  const atom_t* a = &R->atom;
  const neigh_t* nb = &R->neigh;
  const real* x = a->x;
  for(int i = 0; i < a->nlocal; i++) {
    const int* row = &nb->neighbors[(size_t)i * nb->maxneighs];
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    const int ti = a->type[i];
    real rhoi = 0;
    for(int k = 0; k < nb->numneigh[i]; k++) {
      const int j = row[k];
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = ti * ntypes + a->type[j];
      if(rsq < F->cutforcesq[tij]) {
        int m; real p;
        eam_knot(F, sqrt(rsq), &m, &p);
        const real* c = &F->rhor_spline[(size_t)tij * F->nr_tot + m * 7];
        rhoi += ((c[3] * p + c[4]) * p + c[5]) * p + c[6];
      }
    }
    const int tii = ti * ti;                       /* sic: ref/force_eam.cpp:337 */
    real p = 1.0 * rhoi * F->rdrho + 1.0;
    int m = (int)p;
    m = ORC_MAX(1, ORC_MIN(m, F->nrho - 1));
    p -= m;
    p = ORC_MIN(p, 1.0);
    const real* c = &F->frho_spline[(size_t)tii * F->nrho_tot + m * 7];
    F->fp[i] = (c[0] * p + c[1]) * p + c[2];
    if(F->evflag) *evdwl += ((c[3] * p + c[4]) * p + c[5]) * p + c[6];
  }
}

/* sweep 2 of the full-list path: pair force (ref :368-441) */
static void eam_full_force(rank_t* R, int ntypes, real* evdwl, real* t_virial)
{
  force_t* F = &R->force;
  atom_t* a = &R->atom;
  const neigh_t* nb = &R->neigh;
  const real* x = a->x;
  for(int i = 0; i < a->nlocal; i++) {
    const int* row = &nb->neighbors[(size_t)i * nb->maxneighs];
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    const int ti = a->type[i];
    real fx = 0.0, fy = 0.0, fz = 0.0;
    for(int k = 0; k < nb->numneigh[i]; k++) {
      const int j = row[k];
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = ti * ntypes + a->type[j];
      if(rsq < F->cutforcesq[tij]) {
        const real r = sqrt(rsq);
        int m; real p;
        eam_knot(F, r, &m, &p);
        const real* cr = &F->rhor_spline[(size_t)tij * F->nr_tot + m * 7];
        const real* cz = &F->z2r_spline[(size_t)tij * F->nr_tot + m * 7];
        const real rhoip = (cr[0] * p + cr[1]) * p + cr[2];
        const real z2p = (cz[0] * p + cz[1]) * p + cz[2];
        const real z2 = ((cz[3] * p + cz[4]) * p + cz[5]) * p + cz[6];
        const real recip = 1.0 / r;
        const real phi = z2 * recip;
        const real phip = z2p * recip - phi * recip;
        const real psip = F->fp[i] * rhoip + F->fp[j] * rhoip + phip;
        real fpair = -psip * recip;
        fx += dx * fpair; fy += dy * fpair; fz += dz * fpair;
        fpair *= 0.5;
        if(F->evflag) {
          *t_virial += dx * dx * fpair + dy * dy * fpair + dz * dz * fpair;
          *evdwl += 0.5 * phi;
        }
      }
    }
    a->f[i * PAD + 0] = fx; a->f[i * PAD + 1] = fy; a->f[i * PAD + 2] = fz;
  }
}

/* half-list path (serial only in the reference): ref/force_eam.cpp:94-270 */
static void eam_half_density(rank_t* R, int ntypes, real* evdwl)
{
  force_t* F = &R->force;
  atom_t* a = &R->atom;
  const neigh_t* nb = &R->neigh;
  const real* x = a->x;
  const int nlocal = a->nlocal;
  for(int i = 0; i < (nlocal + a->nghost) * PAD; i++) a->f[i] = 0;
  for(int i = 0; i < nlocal; i++) F->rho[i] = 0.0;
  for(int i = 0; i < nlocal; i++) {
    const int* row = &nb->neighbors[(size_t)i * nb->maxneighs];
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    const int ti = a->type[i];
    real rhoi = 0.0;
    for(int k = 0; k < nb->numneigh[i]; k++) {
      const int j = row[k];
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = ti * ntypes + a->type[j];
      if(rsq < F->cutforcesq[tij]) {
        int m; real p;
        eam_knot(F, sqrt(rsq), &m, &p);
        const real* c = &F->rhor_spline[(size_t)tij * F->nr_tot + m * 7];
        rhoi += ((c[3] * p + c[4]) * p + c[5]) * p + c[6];
        if(j < nlocal) F->rho[j] += ((c[3] * p + c[4]) * p + c[5]) * p + c[6];
      }
    }
    F->rho[i] += rhoi;
  }
  for(int i = 0; i < nlocal; i++) {
    real p = 1.0 * F->rho[i] * F->rdrho + 1.0;
    int m = (int)p;
    const int tii = a->type[i] * a->type[i];
    m = ORC_MAX(1, ORC_MIN(m, F->nrho - 1));
    p -= m;
    p = ORC_MIN(p, 1.0);
    const real* c = &F->frho_spline[(size_t)tii * F->nrho_tot + m * 7];
    F->fp[i] = (c[0] * p + c[1]) * p + c[2];
    if(F->evflag) *evdwl += ((c[3] * p + c[4]) * p + c[5]) * p + c[6];
  }
}

static void eam_half_force(rank_t* R, int ntypes, real* evdwl)
{
  force_t* F = &R->force;
  atom_t* a = &R->atom;
  const neigh_t* nb = &R->neigh;
  const real* x = a->x;
  const int nlocal = a->nlocal;
  for(int i = 0; i < nlocal; i++) {
    const int* row = &nb->neighbors[(size_t)i * nb->maxneighs];
    const real xi = x[i * PAD + 0], yi = x[i * PAD + 1], zi = x[i * PAD + 2];
    const int ti = a->type[i];
    real fx = 0, fy = 0, fz = 0;
    for(int k = 0; k < nb->numneigh[i]; k++) {
      const int j = row[k];
      const real dx = xi - x[j * PAD + 0], dy = yi - x[j * PAD + 1], dz = zi - x[j * PAD + 2];
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = ti * ntypes + a->type[j];
      if(rsq < F->cutforcesq[tij]) {
        const real r = sqrt(rsq);
        int m; real p;
        eam_knot(F, r, &m, &p);
        const real* cr = &F->rhor_spline[(size_t)tij * F->nr_tot + m * 7];
        const real* cz = &F->z2r_spline[(size_t)tij * F->nr_tot + m * 7];
        const real rhoip = (cr[0] * p + cr[1]) * p + cr[2];
        const real z2p = (cz[0] * p + cz[1]) * p + cz[2];
        const real z2 = ((cz[3] * p + cz[4]) * p + cz[5]) * p + cz[6];
        const real recip = 1.0 / r;
        const real phi = z2 * recip;
        const real phip = z2p * recip - phi * recip;
        const real psip = F->fp[i] * rhoip + F->fp[j] * rhoip + phip;
        real fpair = -psip * recip;
        fx += dx * fpair; fy += dy * fpair; fz += dz * fpair;
        if(j < nlocal) {
          a->f[j * PAD + 0] -= dx * fpair; a->f[j * PAD + 1] -= dy * fpair; a->f[j * PAD + 2] -= dz * fpair;
        } else fpair *= 0.5;
        if(F->evflag) F->virial += dx * dx * fpair + dy * dy * fpair + dz * dz * fpair;
        if(j < nlocal) *evdwl += phi; else *evdwl += 0.5 * phi;
      }
    }
    a->f[i * PAD + 0] += fx; a->f[i * PAD + 1] += fy; a->f[i * PAD + 2] += fz;
  }
}

void orc_force_compute(orc_world* w, int evflag)
{
  const int nt = w->ntypes;
  if(w->in.forcetype == FORCE_LJ) {
    for(int p = 0; p < w->nprocs; p++) {
      rank_t* R = &w->r[p];
      force_t* F = &R->force;
      const atom_t* a = &R->atom;
      const neigh_t* nb = &R->neigh;
      F->evflag = evflag;
      F->eng_vdwl = 0; F->virial = 0;       /* ref/force_lj.cpp:74-75 */
      if(F->use_oldcompute)
        lj_force_original(a->x, a->type, a->nlocal, a->nlocal + a->nghost, nb->neighbors, nb->numneigh, nb->maxneighs,
                          nt, F->cutforcesq, F->sigma6, F->epsilon, evflag, a->f, &F->eng_vdwl, &F->virial);
      else if(nb->halfneigh)
        orc_lj_force_half(a->x, a->type, a->nlocal, a->nlocal + a->nghost, nb->neighbors, nb->numneigh, nb->maxneighs,
                          nt, F->cutforcesq, F->sigma6, F->epsilon, evflag, nb->ghost_newton, a->f, &F->eng_vdwl, &F->virial);
      else
        orc_lj_force_full(a->x, a->type, a->nlocal, nb->neighbors, nb->numneigh, nb->maxneighs,
                          nt, F->cutforcesq, F->sigma6, F->epsilon, evflag, a->f, &F->eng_vdwl, &F->virial);
    }
    return;
  }
  /* EAM: sweep 1 on every rank, fp halo, sweep 2 on every rank */
  real* evdwl = (real*)calloc(w->nprocs, sizeof(real));
  real* tvir = (real*)calloc(w->nprocs, sizeof(real));
  const int half = w->r[0].neigh.halfneigh;
  for(int p = 0; p < w->nprocs; p++) {
    rank_t* R = &w->r[p];
    R->force.evflag = evflag;
    eam_ensure(&R->force, &R->atom);
    if(half) { R->force.virial = 0; eam_half_density(R, nt, &evdwl[p]); }
    else { R->force.eng_vdwl = 0; R->force.virial = 0; eam_full_density(R, nt, &evdwl[p]); }
  }
  eam_fp_halo(w);
  for(int p = 0; p < w->nprocs; p++) {
    rank_t* R = &w->r[p];
    if(half) { eam_half_force(R, nt, &evdwl[p]); R->force.eng_vdwl = evdwl[p]; }
    else {
      eam_full_force(R, nt, &evdwl[p], &tvir[p]);
      R->force.virial += tvir[p];
      R->force.eng_vdwl += 2.0 * evdwl[p];    /* ref/force_eam.cpp:446 */
    }
  }
  free(evdwl); free(tvir);
}

/* ------------------------------------------------------------------------------------------------
 * Integrate (ref/integrate.cpp:41-68) and Thermo (ref/thermo.cpp)
 * ---------------------------------------------------------------------------------------------- */

void orc_initial_integrate(orc_world* w)
{
  const real dt = w->dt, dtf = w->dtforce;
  for(int p = 0; p < w->nprocs; p++) {
    atom_t* a = &w->r[p].atom;
    real *x = a->x, *v = a->v;
    const real* f = a->f;
    for(int i = 0; i < a->nlocal; i++) {
      v[i * PAD + 0] += dtf * f[i * PAD + 0];
      v[i * PAD + 1] += dtf * f[i * PAD + 1];
      v[i * PAD + 2] += dtf * f[i * PAD + 2];
      x[i * PAD + 0] += dt * v[i * PAD + 0];
      x[i * PAD + 1] += dt * v[i * PAD + 1];
      x[i * PAD + 2] += dt * v[i * PAD + 2];
    }
  }
}

void orc_final_integrate(orc_world* w)
{
  const real dtf = w->dtforce;
  for(int p = 0; p < w->nprocs; p++) {
    atom_t* a = &w->r[p].atom;
    real* v = a->v;
    const real* f = a->f;
    for(int i = 0; i < a->nlocal; i++) {
      v[i * PAD + 0] += dtf * f[i * PAD + 0];
      v[i * PAD + 1] += dtf * f[i * PAD + 1];
      v[i * PAD + 2] += dtf * f[i * PAD + 2];
    }
  }
}

static real thermo_temperature(orc_world* w)   /* ref/thermo.cpp:140-174 */
{
  real t_all = 0;
  for(int p = 0; p < w->nprocs; p++) {
    const atom_t* a = &w->r[p].atom;
    real t = 0.0;
    for(int i = 0; i < a->nlocal; i++) {
      const real vx = a->v[i * PAD + 0], vy = a->v[i * PAD + 1], vz = a->v[i * PAD + 2];
      t += (vx * vx + vy * vy + vz * vz) * a->mass;
    }
    t_all += t;
  }
  return t_all * w->thermo.t_scale;
}

void orc_thermo(orc_world* w, double* t_out, double* u_out, double* p_out)
{
  const thermo_t* th = &w->thermo;
  const real t = thermo_temperature(w);
  real eng = 0, vir = 0;
  for(int p = 0; p < w->nprocs; p++) {          /* ref/thermo.cpp:119-136, 181-194 */
    real e_act = w->r[p].force.eng_vdwl;
    if(w->r[p].neigh.halfneigh) e_act *= 2.0;
    e_act *= th->e_scale;
    eng += e_act;
    vir += w->r[p].force.virial;
  }
  const real u = eng / w->r[0].atom.natoms;
  const real pr = (t * th->dof_boltz + vir) * th->p_scale;
  *t_out = t; *u_out = u; *p_out = pr;
}

static void record_row(orc_world* w, int step)
{
  double t, u, p;
  orc_thermo(w, &t, &u, &p);
  if(w->nrows == w->maxrows) {
    w->maxrows = w->maxrows ? 2 * w->maxrows : 64;
    w->row_step = (int*)realloc(w->row_step, w->maxrows * sizeof(int));
    w->row_t = (double*)realloc(w->row_t, w->maxrows * sizeof(double));
    w->row_u = (double*)realloc(w->row_u, w->maxrows * sizeof(double));
    w->row_p = (double*)realloc(w->row_p, w->maxrows * sizeof(double));
  }
  w->row_step[w->nrows] = step; w->row_t[w->nrows] = t; w->row_u[w->nrows] = u; w->row_p[w->nrows] = p;
  w->nrows++;
  if(!w->quiet) {
    const double elapsed = step == 0 ? 0.0 : wall() - w->t_total_start;
    fprintf(stdout, "%i %e %e %e %6.3lf\n", step, t, u, p, elapsed);   /* ref/thermo.cpp:110 */
    fflush(stdout);
  }
}

/* Thermo::compute gating (ref/thermo.cpp:78-80) */
static void thermo_compute(orc_world* w, int iflag)
{
  const int nstat = w->thermo.nstat;
  if(iflag > 0 && iflag % nstat) return;
  if(iflag == -1 && nstat > 0 && w->ntimes % nstat == 0) return;
  record_row(w, iflag == -1 ? w->ntimes : iflag);
}

/* ------------------------------------------------------------------------------------------------
 * setup: create_box / create_atoms / create_velocity (ref/setup.cpp:305-517), Thermo::setup
 * ---------------------------------------------------------------------------------------------- */

static double pm_random(int* idum)   /* Park-Miller minimal standard, Schrage form (ref/setup.cpp:505-517) */
{
  const int k = *idum / 127773;
  *idum = 16807 * (*idum - k * 127773) - 2836 * k;
  if(*idum < 0) *idum += 2147483647;
  return (1.0 / 2147483647) * (*idum);
}

static int create_atoms(orc_world* w)
{
  const int nx = w->in.nx, ny = w->in.ny, nz = w->in.nz;
  const double rho = w->in.rho;
  const double alat = pow(4.0 / rho, 1.0 / 3.0);
  int total = 0;
  for(int p = 0; p < w->nprocs; p++) {
    atom_t* a = &w->r[p].atom;
    a->natoms = 4 * nx * ny * nz;
    a->nlocal = 0;
    int ilo = (int)(a->box.xlo / (0.5 * alat) - 1), ihi = (int)(a->box.xhi / (0.5 * alat) + 1);
    int jlo = (int)(a->box.ylo / (0.5 * alat) - 1), jhi = (int)(a->box.yhi / (0.5 * alat) + 1);
    int klo = (int)(a->box.zlo / (0.5 * alat) - 1), khi = (int)(a->box.zhi / (0.5 * alat) + 1);
    ilo = ORC_MAX(ilo, 0); ihi = ORC_MIN(ihi, 2 * nx - 1);
    jlo = ORC_MAX(jlo, 0); jhi = ORC_MIN(jhi, 2 * ny - 1);
    klo = ORC_MAX(klo, 0); khi = ORC_MIN(khi, 2 * nz - 1);
    /* visit the half-lattice in 8x8x8 tiles (tile index x fastest, then y, then z; inside a tile x
       fastest too) so that consecutively created atoms are spatially close  (ref/setup.cpp:355-422) */
    const int T = 8;
    for(int oz = 0; oz * T <= khi; oz++)
      for(int oy = 0; oy * T <= jhi; oy++)
        for(int ox = 0; ox * T <= ihi; ox++)
          for(int sz = 0; sz < T; sz++)
            for(int sy = 0; sy < T; sy++)
              for(int sx = 0; sx < T; sx++) {
                const int i = ox * T + sx, j = oy * T + sy, k = oz * T + sz;
                if((i + j + k) % 2) continue;
                if(i < ilo || i > ihi || j < jlo || j > jhi || k < klo || k > khi) continue;
                const double xt = 0.5 * alat * i, yt = 0.5 * alat * j, zt = 0.5 * alat * k;
                if(xt >= a->box.xlo && xt < a->box.xhi && yt >= a->box.ylo && yt < a->box.yhi &&
                   zt >= a->box.zlo && zt < a->box.zhi) {
                  /* velocity depends only on the global lattice index: 3 x (5 warm-ups + 1 draw) */
                  int n = k * (2 * ny) * (2 * nx) + j * (2 * nx) + i + 1;
                  const int lattice_id = n;
                  double vel[3];
                  for(int c = 0; c < 3; c++) {
                    for(int m = 0; m < 5; m++) pm_random(&n);
                    vel[c] = pm_random(&n);
                  }
                  atom_add(a, w->ntypes, xt, yt, zt, vel[0], vel[1], vel[2], lattice_id);
                }
              }
    total += a->nlocal;
  }
  if(total != w->r[0].atom.natoms) { set_err("Created incorrect # of atoms"); return 1; }
  return 0;
}

static void thermo_setup(orc_world* w)   /* ref/thermo.cpp:42-72 */
{
  thermo_t* th = &w->thermo;
  const atom_t* a = &w->r[0].atom;
  th->rho = w->in.rho;
  th->ntimes = w->ntimes;
  if(w->in.units == UNITS_LJ) {
    th->mvv2e = 1.0;
    th->dof_boltz = (a->natoms * 3 - 3);
    th->t_scale = th->mvv2e / th->dof_boltz;
    th->p_scale = 1.0 / 3 / a->box.xprd / a->box.yprd / a->box.zprd;
    th->e_scale = 0.5;
  } else {
    th->mvv2e = 1.036427e-04;
    th->dof_boltz = (a->natoms * 3 - 3) * 8.617343e-05;
    th->t_scale = th->mvv2e / th->dof_boltz;
    th->p_scale = 1.602176e+06 / 3 / a->box.xprd / a->box.yprd / a->box.zprd;
    th->e_scale = 524287.985533;
    w->dtforce /= th->mvv2e;
  }
}

static void create_velocity(orc_world* w)   /* ref/setup.cpp:454-494 */
{
  double vtot[3] = {0.0, 0.0, 0.0};
  for(int d = 0; d < 3; d++) {
    double all = 0.0;
    for(int p = 0; p < w->nprocs; p++) {
      const atom_t* a = &w->r[p].atom;
      double s = 0.0;
      for(int i = 0; i < a->nlocal; i++) s += a->v[i * PAD + d];
      all += s;
    }
    vtot[d] = all / w->r[0].atom.natoms;
  }
  for(int p = 0; p < w->nprocs; p++) {
    atom_t* a = &w->r[p].atom;
    for(int i = 0; i < a->nlocal; i++)
      for(int d = 0; d < 3; d++) a->v[i * PAD + d] -= vtot[d];
  }
  const double t = thermo_temperature(w);
  const double factor = sqrt(w->in.t_request / t);
  for(int p = 0; p < w->nprocs; p++) {
    atom_t* a = &w->r[p].atom;
    for(int i = 0; i < a->nlocal * PAD; i++) a->v[i] *= factor;
  }
}

/* ------------------------------------------------------------------------------------------------
 * orc_create: CLI (ref/ljs.cpp:87-408)
 * ---------------------------------------------------------------------------------------------- */

static int arg_is(const char* a, const char* s1, const char* s2)
{
  return strcmp(a, s1) == 0 || (s2 && strcmp(a, s2) == 0);
}

orc_world* orc_create(int argc, char** argv, int nprocs, int quiet)
{
  orc_world* w = (orc_world*)calloc(1, sizeof(orc_world));
  w->nprocs = nprocs > 0 ? nprocs : 1;
  w->quiet = quiet;
  w->num_threads = 1;
  w->ntypes = 4;
  w->halfneigh = 1;
  w->ghost_newton = 1;
  w->sort_flag = -1;
  strcpy(w->input_file, "in.lj.miniMD");
  int num_steps = -1, system_size = -1, nx = -1, ny = -1, nz = -1, neighbor_size = -1;
  int units_override = -1, force_override = -1;
  for(int i = 0; i < argc; i++)
    if(arg_is(argv[i], "-i", "--input_file") && i + 1 < argc) strncpy(w->input_file, argv[++i], 1023);
  if(read_deck(&w->in, w->input_file)) { free(w); return NULL; }
  for(int i = 0; i < argc; i++) {
    const char* a = argv[i];
    const int has = i + 1 < argc;
    if(arg_is(a, "-t", "--num_threads") && has) w->num_threads = atoi(argv[++i]);
    else if(arg_is(a, "--teams", NULL) && has) ++i;
    else if(arg_is(a, "-n", "--nsteps") && has) num_steps = atoi(argv[++i]);
    else if(arg_is(a, "-s", "--size") && has) system_size = atoi(argv[++i]);
    else if(arg_is(a, "-nx", NULL) && has) nx = atoi(argv[++i]);
    else if(arg_is(a, "-ny", NULL) && has) ny = atoi(argv[++i]);
    else if(arg_is(a, "-nz", NULL) && has) nz = atoi(argv[++i]);
    else if(arg_is(a, "--ntypes", NULL) && has) w->ntypes = atoi(argv[++i]);
    else if(arg_is(a, "-b", "--neigh_bins") && has) neighbor_size = atoi(argv[++i]);
    else if(arg_is(a, "--half_neigh", NULL) && has) w->halfneigh = atoi(argv[++i]);
    else if(arg_is(a, "-sse", NULL) && has) ++i;
    else if(arg_is(a, "--sort", NULL) && has) w->sort_flag = atoi(argv[++i]);
    else if(arg_is(a, "-o", "--yaml_output") && has) w->yaml_output = atoi(argv[++i]);
    else if(arg_is(a, "-f", "--data_file") && has) { w->in.has_datafile = 1; strncpy(w->in.datafile, argv[++i], 999); }
    else if(arg_is(a, "-u", "--units") && has) units_override = strcmp(argv[++i], "metal") == 0 ? UNITS_METAL : UNITS_LJ;
    else if(arg_is(a, "-p", "--force") && has) force_override = strcmp(argv[++i], "eam") == 0 ? FORCE_EAM : FORCE_LJ;
    else if(arg_is(a, "-gn", "--ghost_newton") && has) w->ghost_newton = atoi(argv[++i]);
    else if(arg_is(a, "--safe_exchange", NULL)) w->safe_exchange = 1;     /* listed by the reference's --help (ref/ljs.cpp:251), parsed here */
  }
  if(units_override >= 0) w->in.units = units_override;
  if(force_override >= 0) w->in.forcetype = force_override;
  if(w->in.has_datafile) { set_err("LAMMPS data files are not supported by the oracle"); free(w); return NULL; }
  if(w->in.forcetype == FORCE_EAM && w->ghost_newton == 1) {
    if(!quiet) printf("# EAM currently requires '--ghost_newton 0'; Changing setting now.\n");
    w->ghost_newton = 0;
  }
  if(num_steps > 0) w->in.ntimes = num_steps;
  if(system_size > 0) { w->in.nx = w->in.ny = w->in.nz = system_size; }
  if(nx > 0) {
    w->in.nx = nx;
    if(ny > 0) w->in.ny = ny; else if(system_size < 0) w->in.ny = nx;
    if(nz > 0) w->in.nz = nz; else if(system_size < 0) w->in.nz = nx;
  }
  int nb[3];
  if(neighbor_size > 0) nb[0] = nb[1] = nb[2] = neighbor_size;
  else {
    const real neighscale = 5.0 / 6.0;      /* computed in MMD_float: ref/ljs.cpp:357-362 */
    nb[0] = neighscale * w->in.nx; nb[1] = neighscale * w->in.ny; nb[2] = neighscale * w->in.nz;
  }
  for(int d = 0; d < 3; d++) if(nb[d] == 0) nb[d] = 1;

  w->ntimes = w->in.ntimes;
  w->dt = w->in.dt;
  w->sort_every = w->sort_flag > 0 ? w->sort_flag : (w->sort_flag < 0 ? w->in.neigh_every : 0);
  w->thermo.nstat = w->in.thermo_nstat;

  if(!quiet) printf("# Create System:\n");
  w->r = (rank_t*)calloc(w->nprocs, sizeof(rank_t));
  const double lattice = pow(4.0 / w->in.rho, 1.0 / 3.0);
  for(int p = 0; p < w->nprocs; p++) {
    rank_t* R = &w->r[p];
    memset(&R->atom.rnd, 0, sizeof(R->atom.rnd));
    initstate_r(1, R->atom.rnd_state, sizeof(R->atom.rnd_state), &R->atom.rnd);
    srandom_r(5413, &R->atom.rnd);
    R->atom.mass = 1;
    R->atom.box.xprd = w->in.nx * lattice;      /* create_box  ref/setup.cpp:305-311 */
    R->atom.box.yprd = w->in.ny * lattice;
    R->atom.box.zprd = w->in.nz * lattice;
    neigh_t* N = &R->neigh;
    N->maxneighs = 100; N->atoms_per_bin = 8;   /* ref/neighbor.cpp:41-58 */
    N->cutneighsq = (real*)calloc(w->ntypes * w->ntypes, sizeof(real));
    N->halfneigh = w->halfneigh; N->ghost_newton = w->ghost_newton;
    N->nbinx = nb[0]; N->nbiny = nb[1]; N->nbinz = nb[2];
    N->every = w->in.neigh_every; N->cutneigh = w->in.neigh_cut;
    force_t* F = &R->force;
    const int nt2 = w->ntypes * w->ntypes;
    F->style = w->in.forcetype; F->ntypes = w->ntypes;
    F->use_oldcompute = w->halfneigh < 0;
    F->cutforce = w->in.force_cut;
    F->cutforcesq = (real*)calloc(nt2, sizeof(real));
    F->epsilon = (real*)calloc(nt2, sizeof(real)); F->sigma = (real*)calloc(nt2, sizeof(real));
    F->sigma6 = (real*)calloc(nt2, sizeof(real));
    for(int i = 0; i < nt2; i++) {               /* ref/ljs.cpp:299-305 */
      const real s = w->in.sigma;
      F->epsilon[i] = w->in.epsilon; F->sigma[i] = s; F->sigma6[i] = s * s * s * s * s * s;
    }
    if(comm_setup(&R->comm, N->cutneigh, &R->atom, p, w->nprocs)) { orc_destroy(w); return NULL; }
    nb_setup(N, &R->atom, w->ntypes);
  }
  w->dtforce = 0.5 * w->dt;                      /* Integrate::setup  ref/integrate.cpp:41-44 */
  for(int p = 0; p < w->nprocs; p++) {
    force_t* F = &w->r[p].force;
    if(F->style == FORCE_LJ) {                   /* ForceLJ::setup  ref/force_lj.cpp:65-69 */
      for(int i = 0; i < w->ntypes * w->ntypes; i++) F->cutforcesq[i] = F->cutforce * F->cutforce;
    } else {
      if(eam_setup(F)) { orc_destroy(w); return NULL; }
      w->r[p].atom.mass = F->mass;               /* ref/ljs.cpp:403 */
    }
  }
  if(create_atoms(w)) { orc_destroy(w); return NULL; }
  thermo_setup(w);
  create_velocity(w);
  if(!quiet) {
    const rank_t* R = &w->r[0];
    printf("# Done .... \n");
    printf("# miniMD-oracle (plain-C restatement of miniMD-Reference 2.0) output ...\n");
    printf("# Run Settings: \n");
    printf("\t# MPI processes: %i\n", w->nprocs);
    printf("\t# OpenMP threads: %i\n", w->num_threads);
    printf("\t# Inputfile: %s\n", w->input_file);
    printf("\t# Datafile: %s\n", "None");
    printf("# Physics Settings: \n");
    printf("\t# ForceStyle: %s\n", w->in.forcetype == FORCE_LJ ? "LJ" : "EAM");
    printf("\t# Force Parameters: %2.2lf %2.2lf\n", (double)w->in.epsilon, (double)w->in.sigma);
    printf("\t# Units: %s\n", w->in.units == 0 ? "LJ" : "METAL");
    printf("\t# Atoms: %i\n", R->atom.natoms);
    printf("\t# Atom types: %i\n", w->ntypes);
    printf("\t# System size: %2.2lf %2.2lf %2.2lf (unit cells: %i %i %i)\n", (double)R->atom.box.xprd,
           (double)R->atom.box.yprd, (double)R->atom.box.zprd, w->in.nx, w->in.ny, w->in.nz);
    printf("\t# Density: %lf\n", (double)w->in.rho);
    printf("\t# Force cutoff: %lf\n", (double)R->force.cutforce);
    printf("\t# Timestep size: %lf\n", (double)w->dt);
    printf("# Technical Settings: \n");
    printf("\t# Neigh cutoff: %lf\n", (double)R->neigh.cutneigh);
    printf("\t# Half neighborlists: %i\n", w->halfneigh);
    printf("\t# Neighbor bins: %i %i %i\n", nb[0], nb[1], nb[2]);
    printf("\t# Neighbor frequency: %i\n", R->neigh.every);
    printf("\t# Sorting frequency: %i\n", w->sort_every);
    printf("\t# Thermo frequency: %i\n", w->thermo.nstat);
    printf("\t# Ghost Newton: %i\n", w->ghost_newton);
    printf("\t# Use intrinsics: %i\n", 0);
    printf("\t# Do safe exchange: %i\n", w->safe_exchange);
    printf("\t# Size of float: %i\n\n", (int)sizeof(real));
  }
  return w;
}

void orc_destroy(orc_world* w)
{
  if(!w) return;
  /* test infrastructure: process-lifetime objects, only the big arrays are returned */
  if(w->r) {
    for(int p = 0; p < w->nprocs; p++) {
      rank_t* R = &w->r[p];
      free(R->atom.x); free(R->atom.v); free(R->atom.f); free(R->atom.type); free(R->atom.tag);
      free(R->atom.x_copy); free(R->atom.v_copy); free(R->atom.type_copy); free(R->atom.tag_copy);
      free(R->neigh.numneigh); free(R->neigh.neighbors); free(R->neigh.bins); free(R->neigh.bincount);
      free(R->neigh.stencil); free(R->neigh.cutneighsq);
      free(R->comm.buf_send); free(R->comm.buf_recv);
      if(R->comm.sendlist) for(int s = 0; s < R->comm.maxswap; s++) free(R->comm.sendlist[s]);
      free(R->comm.sendlist);
      free(R->force.rho); free(R->force.fp);
      free(R->force.rhor_spline); free(R->force.frho_spline); free(R->force.z2r_spline);
    }
    free(w->r);
  }
  free(w->row_step); free(w->row_t); free(w->row_u); free(w->row_p);
  free(w);
}

/* ------------------------------------------------------------------------------------------------
 * driver: ref/ljs.cpp:445-495 and Integrate::run (ref/integrate.cpp:70-207)
 * ---------------------------------------------------------------------------------------------- */

int orc_initial(orc_world* w)
{
  orc_exchange(w);
  if(w->sort_flag > 0) orc_sort(w);
  orc_borders(w);
  orc_neighbor_build(w);
  orc_force_compute(w, 1);
  if(w->halfneigh && w->ghost_newton) orc_reverse_communicate(w);
  if(!w->quiet) { printf("# Starting dynamics ...\n"); printf("# Timestep T U P Time\n"); }
  w->nrows = 0;
  thermo_compute(w, 0);
  return 0;
}

int orc_run(orc_world* w)
{
  const int every = w->r[0].neigh.every;
  const int nstat = w->thermo.nstat;
  for(int i = 0; i < 5; i++) w->timer[i] = 0.0;
  w->t_total_start = wall();
  if(!w->run_started) {                       /* ref/integrate.cpp:80-81 (first and only run) */
    w->dtforce = w->dtforce / w->r[0].atom.mass;
    w->run_started = 1;
  }
  int next_sort = w->sort_every > 0 ? w->sort_every : w->ntimes + 1;
  for(int n = 0; n < w->ntimes; n++) {
    orc_initial_integrate(w);
    stamp(w);
    if((n + 1) % every) {
      orc_communicate(w);
      stamp_to(w, T_COMM);
    } else {
      const double t0 = wall();
      orc_exchange(w);
      if(n + 1 >= next_sort) { orc_sort(w); next_sort += w->sort_every; }
      orc_borders(w);
      w->timer[T_TEST] += wall() - t0;
      stamp_to(w, T_COMM);
      orc_neighbor_build(w);
      stamp_to(w, T_NEIGH);
    }
    orc_force_compute(w, nstat ? ((n + 1) % nstat == 0) : 0);
    stamp_to(w, T_FORCE);
    if(w->halfneigh && w->ghost_newton) {
      orc_reverse_communicate(w);
      stamp_to(w, T_COMM);
    }
    orc_final_integrate(w);
    if(nstat) thermo_compute(w, n + 1);
  }
  w->timer[T_TOTAL] = wall() - w->t_total_start;
  /* ref/ljs.cpp:477-483 */
  orc_force_compute(w, 1);
  if(w->halfneigh && w->ghost_newton) orc_reverse_communicate(w);
  thermo_compute(w, -1);
  return 0;
}

void orc_print_perf(orc_world* w)
{
  const double* t = w->timer;
  const int natoms = w->r[0].atom.natoms;
  const double other = t[T_TOTAL] - t[T_FORCE] - t[T_NEIGH] - t[T_COMM];
  printf("\n\n# Performance Summary:\n");
  printf("# MPI_proc OMP_threads nsteps natoms t_total t_force t_neigh t_comm t_other performance perf/thread grep_string t_extra\n");
  printf("%i %i %i %i %lf %lf %lf %lf %lf %lf %lf PERF_SUMMARY %lf\n\n\n", w->nprocs, w->num_threads, w->ntimes, natoms,
         t[T_TOTAL], t[T_FORCE], t[T_NEIGH], t[T_COMM], other, 1.0 * natoms * w->ntimes / t[T_TOTAL],
         1.0 * natoms * w->ntimes / t[T_TOTAL] / w->nprocs / w->num_threads, t[T_TEST]);
}

/* ------------------------------------------------------------------------------------------------
 * accessors
 * ---------------------------------------------------------------------------------------------- */

int orc_nprocs(const orc_world* w) { return w->nprocs; }
int orc_natoms(const orc_world* w) { return w->r[0].atom.natoms; }
int orc_ntypes(const orc_world* w) { return w->ntypes; }
int orc_nlocal(const orc_world* w, int p) { return w->r[p].atom.nlocal; }
int orc_nghost(const orc_world* w, int p) { return w->r[p].atom.nghost; }
real* orc_x(orc_world* w, int p) { return w->r[p].atom.x; }
real* orc_v(orc_world* w, int p) { return w->r[p].atom.v; }
real* orc_f(orc_world* w, int p) { return w->r[p].atom.f; }
int* orc_type(orc_world* w, int p) { return w->r[p].atom.type; }
int* orc_tag(orc_world* w, int p) { return w->r[p].atom.tag; }
int* orc_numneigh(orc_world* w, int p) { return w->r[p].neigh.numneigh; }
int* orc_neighbors(orc_world* w, int p) { return w->r[p].neigh.neighbors; }
int orc_maxneighs(const orc_world* w, int p) { return w->r[p].neigh.maxneighs; }
real* orc_eam_fp(orc_world* w, int p) { return w->r[p].force.fp; }
double orc_eng_vdwl(const orc_world* w, int p) { return w->r[p].force.eng_vdwl; }
double orc_virial(const orc_world* w, int p) { return w->r[p].force.virial; }
void orc_box(const orc_world* w, int p, double o[9])
{
  const box_t* b = &w->r[p].atom.box;
  o[0] = b->xprd; o[1] = b->yprd; o[2] = b->zprd; o[3] = b->xlo; o[4] = b->xhi; o[5] = b->ylo; o[6] = b->yhi; o[7] = b->zlo; o[8] = b->zhi;
}
void orc_procgrid(const orc_world* w, int o[3]) { for(int d = 0; d < 3; d++) o[d] = w->r[0].comm.procgrid[d]; }
int orc_nswap(const orc_world* w, int p) { return w->r[p].comm.nswap; }
int* orc_sendnum(orc_world* w, int p) { return w->r[p].comm.sendnum; }
int* orc_recvnum(orc_world* w, int p) { return w->r[p].comm.recvnum; }
int* orc_firstrecv(orc_world* w, int p) { return w->r[p].comm.firstrecv; }
int* orc_sendlist(orc_world* w, int p, int s) { return w->r[p].comm.sendlist[s]; }
void orc_swap_info(const orc_world* w, int p, int s, double o2[2], int o6[6])
{
  const comm_t* c = &w->r[p].comm;
  o2[0] = c->slablo[s]; o2[1] = c->slabhi[s];
  o6[0] = c->pbc_any[s]; o6[1] = c->pbc_flagx[s]; o6[2] = c->pbc_flagy[s]; o6[3] = c->pbc_flagz[s];
  o6[4] = c->sendproc[s]; o6[5] = c->recvproc[s];
}
void orc_nbins(const orc_world* w, int o[3]) { o[0] = w->r[0].neigh.nbinx; o[1] = w->r[0].neigh.nbiny; o[2] = w->r[0].neigh.nbinz; }
void orc_bin_geometry(const orc_world* w, int p, int mb[3], int lo[3], int* nst)
{
  const neigh_t* n = &w->r[p].neigh;
  mb[0] = n->mbinx; mb[1] = n->mbiny; mb[2] = n->mbinz; lo[0] = n->mbinxlo; lo[1] = n->mbinylo; lo[2] = n->mbinzlo;
  *nst = n->nstencil;
}
int orc_nrows(const orc_world* w) { return w->nrows; }
void orc_row(const orc_world* w, int i, int* step, double* t, double* u, double* p)
{
  *step = w->row_step[i]; *t = w->row_t[i]; *u = w->row_u[i]; *p = w->row_p[i];
}
double orc_param(const orc_world* w, const char* k)
{
  const rank_t* R = &w->r[0];
  if(!strcmp(k, "dt")) return w->dt;
  if(!strcmp(k, "dtforce")) return w->dtforce;
  if(!strcmp(k, "mass")) return R->atom.mass;
  if(!strcmp(k, "cutneigh")) return R->neigh.cutneigh;
  if(!strcmp(k, "cutforce")) return R->force.cutforce;
  if(!strcmp(k, "t_scale")) return w->thermo.t_scale;
  if(!strcmp(k, "e_scale")) return w->thermo.e_scale;
  if(!strcmp(k, "p_scale")) return w->thermo.p_scale;
  if(!strcmp(k, "dof_boltz")) return w->thermo.dof_boltz;
  if(!strcmp(k, "mvv2e")) return w->thermo.mvv2e;
  if(!strcmp(k, "ntimes")) return w->ntimes;
  if(!strcmp(k, "nstat")) return w->thermo.nstat;
  if(!strcmp(k, "neigh_every")) return R->neigh.every;
  if(!strcmp(k, "halfneigh")) return w->halfneigh;
  if(!strcmp(k, "ghost_newton")) return w->ghost_newton;
  if(!strcmp(k, "forcetype")) return w->in.forcetype;
  if(!strcmp(k, "units")) return w->in.units;
  if(!strcmp(k, "rho")) return w->in.rho;
  if(!strcmp(k, "t_request")) return w->in.t_request;
  if(!strcmp(k, "nx")) return w->in.nx;
  if(!strcmp(k, "ny")) return w->in.ny;
  if(!strcmp(k, "nz")) return w->in.nz;
  if(!strcmp(k, "sort_every")) return w->sort_every;
  if(!strcmp(k, "atoms_per_bin")) return R->neigh.atoms_per_bin;
  if(!strcmp(k, "eam_nr")) return R->force.nr;
  if(!strcmp(k, "eam_nrho")) return R->force.nrho;
  if(!strcmp(k, "eam_nr_tot")) return R->force.nr_tot;
  if(!strcmp(k, "eam_nrho_tot")) return R->force.nrho_tot;
  if(!strcmp(k, "eam_rdr")) return R->force.rdr;
  if(!strcmp(k, "eam_rdrho")) return R->force.rdrho;
  if(!strcmp(k, "eam_cutmax")) return R->force.cutmax;
  return NAN;
}
real* orc_cutforcesq(orc_world* w) { return w->r[0].force.cutforcesq; }
real* orc_lj_epsilon(orc_world* w) { return w->r[0].force.epsilon; }
real* orc_lj_sigma6(orc_world* w) { return w->r[0].force.sigma6; }
real* orc_eam_rhor_spline(orc_world* w) { return w->r[0].force.rhor_spline; }
real* orc_eam_z2r_spline(orc_world* w) { return w->r[0].force.z2r_spline; }
real* orc_eam_frho_spline(orc_world* w) { return w->r[0].force.frho_spline; }
void orc_timers(const orc_world* w, double o[5]) { for(int i = 0; i < 5; i++) o[i] = w->timer[i]; }

#ifdef MMD_ORACLE_MAIN
/* CLI twin of the reference executable:  mmd_oracle_dp [--nprocs P] <reference flags>  */
int main(int argc, char** argv)
{
  int nprocs = 1;
  for(int i = 1; i + 1 < argc; i++)
    if(strcmp(argv[i], "--nprocs") == 0) nprocs = atoi(argv[i + 1]);
  orc_world* w = orc_create(argc, argv, nprocs, 0);
  if(!w) { printf("%s\n", orc_last_error()); return 0; }
  orc_initial(w);
  orc_run(w);
  orc_print_perf(w);
  orc_destroy(w);
  return 0;
}
#endif
