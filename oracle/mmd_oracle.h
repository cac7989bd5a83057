/* oracle/mmd_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's (Mantevo/miniMD `ref/`) per-timestep hot path, written to be
 * bit-comparable with the reference run single-threaded:  same loop order, same arithmetic order,
 * no FMA contraction (built with -ffp-contract=off), so thermo rows and per-atom arrays agree to the
 * last bit with `oracle/_ref/miniMD_ref_*` (the unmodified reference) — see tests/test_oracle_*.py,
 * which pin it against (a) the reference's published logs tests/reference_output/{*.lj,*.eam},
 * (b) rows printed by the reference built here, (c) per-atom arrays dumped from the reference objects.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (minimd_amd/, include/mmd.h) never links, imports or calls it.
 *
 * Multi-rank runs are modelled by "virtual ranks": P rank states inside one process stepping in
 * lock-step, MPI_Sendrecv being a memcpy between two rank states (same swap order, same buffers as
 * ref/comm.cpp), MPI_Allreduce a sum in rank order.
 *
 * Precision is a compile-time choice like the reference's -DPRECISION (ref/types.h:61-72):
 *   -DMMD_PRECISION=2 -> double (libmmd_oracle_dp.so),  =1 -> float (libmmd_oracle_sp.so).
 */
#ifndef MMD_ORACLE_H
#define MMD_ORACLE_H

#ifndef MMD_PRECISION
#define MMD_PRECISION 2
#endif
#if MMD_PRECISION == 1
typedef float orc_real;
#else
typedef double orc_real;
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_world orc_world;

/* ---- whole-program level (ref/ljs.cpp main) -------------------------------------------------- */

/* Parse the reference CLI (ref/ljs.cpp:87-261) + input deck (ref/input.cpp:48-187) and build the
 * system on `nprocs` virtual ranks (create_box, Comm::setup, Neighbor::setup, Integrate::setup,
 * Force::setup, create_atoms, Thermo::setup, create_velocity — ref/ljs.cpp:393-408).
 * `quiet`!=0 suppresses all stdout.  Returns NULL on error (message via orc_last_error). */
orc_world* orc_create(int argc, char** argv, int nprocs, int quiet);
void orc_destroy(orc_world*);
const char* orc_last_error(void);

/* ref/ljs.cpp:445-468: exchange, [sort], borders, neighbor build, force(evflag=1), reverse comm,
 * thermo row 0. */
int orc_initial(orc_world*);
/* ref/ljs.cpp:470-483: Integrate::run for the configured ntimes, then the final force+thermo. */
int orc_run(orc_world*);
/* prints "# Performance Summary" + PERF_SUMMARY row (ref/ljs.cpp:485-495) */
void orc_print_perf(orc_world*);

/* ---- step-level pieces, callable one by one (all virtual ranks in lock-step) ------------------ */
void orc_initial_integrate(orc_world*);        /* ref/integrate.cpp:46-57  */
void orc_final_integrate(orc_world*);          /* ref/integrate.cpp:59-68  */
void orc_communicate(orc_world*);              /* ref/comm.cpp:276-317     */
void orc_reverse_communicate(orc_world*);      /* ref/comm.cpp:321-355     */
void orc_exchange(orc_world*);                 /* ref/comm.cpp:364-597     */
void orc_borders(orc_world*);                  /* ref/comm.cpp:700-883     */
void orc_sort(orc_world*);                     /* ref/atom.cpp:355-421     */
void orc_neighbor_build(orc_world*);           /* ref/neighbor.cpp:79-213  */
void orc_force_compute(orc_world*, int evflag);/* ref/force_lj.cpp:72-113, ref/force_eam.cpp:82-91 */
/* T,U,P exactly as Thermo::compute would print them (ref/thermo.cpp:74-194) */
void orc_thermo(orc_world*, double* t, double* u, double* p);

/* ---- accessors -------------------------------------------------------------------------------- */
int orc_nprocs(const orc_world*);
int orc_natoms(const orc_world*);
int orc_ntypes(const orc_world*);
int orc_nlocal(const orc_world*, int rank);
int orc_nghost(const orc_world*, int rank);
orc_real* orc_x(orc_world*, int rank);      /* [(nlocal+nghost)*3] */
orc_real* orc_v(orc_world*, int rank);      /* [nlocal*3] */
orc_real* orc_f(orc_world*, int rank);      /* [(nlocal or nlocal+nghost)*3] */
int* orc_type(orc_world*, int rank);        /* [nlocal+nghost] */
int* orc_tag(orc_world*, int rank);         /* [nlocal] global lattice id n = k*(2ny)(2nx)+j*(2nx)+i+1 (ref/setup.cpp:378), oracle-only bookkeeping */
int* orc_numneigh(orc_world*, int rank);    /* [nlocal] */
int* orc_neighbors(orc_world*, int rank);   /* row-major [nlocal*maxneighs] (ref/neighbor.cpp:128) */
int orc_maxneighs(const orc_world*, int rank);
orc_real* orc_eam_fp(orc_world*, int rank); /* [nlocal+nghost] after an EAM force compute */
double orc_eng_vdwl(const orc_world*, int rank);
double orc_virial(const orc_world*, int rank);
void orc_box(const orc_world*, int rank, double out9[9]); /* xprd yprd zprd xlo xhi ylo yhi zlo zhi */
void orc_procgrid(const orc_world*, int out3[3]);
int orc_nswap(const orc_world*, int rank);
int* orc_sendnum(orc_world*, int rank);
int* orc_recvnum(orc_world*, int rank);
int* orc_firstrecv(orc_world*, int rank);
int* orc_sendlist(orc_world*, int rank, int iswap);
/* comm geometry of `rank` (ref/comm.cpp:208-269): per swap slablo, slabhi, pbc_any, pbc_flagx/y/z, sendproc, recvproc */
void orc_swap_info(const orc_world*, int rank, int iswap, double out2[2], int out6[6]);
void orc_nbins(const orc_world*, int out3[3]);
void orc_bin_geometry(const orc_world*, int rank, int out_mbin[3], int out_mbinlo[3], int* nstencil);
/* thermo history recorded by orc_initial/orc_run: rows of (step, T, U, P) */
int orc_nrows(const orc_world*);
void orc_row(const orc_world*, int i, int* step, double* t, double* u, double* p);
/* scalar parameters (for feeding the product's kernel-level hooks with identical inputs) */
double orc_param(const orc_world*, const char* name);
/* LJ tables [ntypes*ntypes] */
orc_real* orc_cutforcesq(orc_world*);
orc_real* orc_lj_epsilon(orc_world*);
orc_real* orc_lj_sigma6(orc_world*);
/* EAM tables (ref/force_eam.cpp:732-793): 7 coeffs per knot, stride nr_tot / nrho_tot per type pair */
orc_real* orc_eam_rhor_spline(orc_world*);
orc_real* orc_eam_z2r_spline(orc_world*);
orc_real* orc_eam_frho_spline(orc_world*);
void orc_timers(const orc_world*, double out5[5]); /* total comm force neigh extra */

/* ---- kernel-level pure functions on raw arrays (array-for-array diff against the HIP kernels) -- */

/* ref/force_lj.cpp:366-449 (compute_fullneigh<EV>): f[0..nlocal) overwritten. */
void orc_lj_force_full(const orc_real* x, const int* type, int nlocal, const int* neighbors, const int* numneigh,
                       int maxneighs, int ntypes, const orc_real* cutforcesq, const orc_real* sigma6,
                       const orc_real* epsilon, int evflag, orc_real* f, orc_real* eng_vdwl, orc_real* virial);
/* ref/force_lj.cpp:185-263 (compute_halfneigh<EV,GN>): f[0..nall) zeroed then accumulated. */
void orc_lj_force_half(const orc_real* x, const int* type, int nlocal, int nall, const int* neighbors,
                       const int* numneigh, int maxneighs, int ntypes, const orc_real* cutforcesq,
                       const orc_real* sigma6, const orc_real* epsilon, int evflag, int ghost_newton,
                       orc_real* f, orc_real* eng_vdwl, orc_real* virial);
/* Brute-force O(N*Nall) full neighbor sets {j != i : rsq <= cutneighsq}, ascending j: an independent
 * check of any binned build (rows of fixed stride maxneighs; returns max row length found). */
int orc_neighbor_brute_full(const orc_real* x, int nlocal, int nall, orc_real cutneighsq, int maxneighs,
                            int* neighbors, int* numneigh);

#ifdef __cplusplus
}
#endif
#endif
