/* include/mmd.h — C-ABI of the MI355X-native miniMD hot path (libmmd_hip_{dp,sp}.so).
 *
 * Drop-in boundary for Mantevo/miniMD `ref/`: every entry point below replaces one method of the
 * reference's per-timestep classes (citations are path:line under the reference checkout). All device
 * state lives behind an opaque handle (one per GPU / rank); host arrays are borrowed only for the
 * duration of an upload/download call and use the REFERENCE's layouts (AoS stride PAD=3 for x/v/f —
 * ref/types.h:77-81; row-major fixed-stride neighbor rows — ref/neighbor.cpp:128), so a reference
 * object can hand over its own pointers.
 *
 * Conventions
 *   - plain C, no C++/torch/HIP types in signatures;
 *   - every function returns 0 on success, <0 on error (mmd_last_error() holds the text); the library
 *     never calls exit();  a missing/unsuitable GPU is an error, there is NO CPU fallback;
 *   - precision is a build-time choice like the reference's -DPRECISION (ref/types.h:61-72):
 *     libmmd_hip_dp.so (MMD_PRECISION=2, mmd_float=double), libmmd_hip_sp.so (=1, float);
 *   - calls on one handle are not re-entrant; work is enqueued on the handle's HIP streams and
 *     functions that return data synchronise.
 */
#ifndef MMD_H
#define MMD_H

#ifndef MMD_PRECISION
#define MMD_PRECISION 2
#endif
#if MMD_PRECISION == 1
typedef float mmd_float;
#else
typedef double mmd_float;
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmd_handle mmd_handle;

/* ---------------------------------------------------------------------------------------------
 * lifecycle
 * ------------------------------------------------------------------------------------------- */
/* device: HIP device ordinal, -1 = the launcher's local rank modulo the visible devices (mmd_launch_env; default 0), -2 = host-only handle (geometry functions
 * only: mmd_atom_set_box, mmd_comm_setup/info, mmd_neighbor_setup/geometry — usable without a GPU). */
int mmd_create(int device, mmd_handle** out);
int mmd_destroy(mmd_handle* h);
int mmd_device_count(void);               /* HIP devices this process sees (0 without a GPU) */
const char* mmd_last_error(void);
int mmd_float_size(void);                 /* sizeof(MMD_float): "# Size of float" (ref/ljs.cpp:442) */
const char* mmd_variant_string(void);     /* counterpart of VARIANT_STRING (ref/variant.h:33) */
int mmd_device_info(mmd_handle* h, char* name, int name_len, int* cu_count, double* hbm_gib);

/* ---------------------------------------------------------------------------------------------
 * Atom  (ref/atom.h:47-106, ref/atom.cpp)
 * ------------------------------------------------------------------------------------------- */
/* Box: prd = global periodic lengths, lo/hi = this rank's sub-box (struct Box, ref/atom.h:40-45). */
int mmd_atom_get_box(mmd_handle* h, mmd_float prd[3], mmd_float lo[3], mmd_float hi[3]);
int mmd_atom_set_box(mmd_handle* h, const mmd_float prd[3], const mmd_float lo[3], const mmd_float hi[3]);
int mmd_atom_set_mass(mmd_handle* h, mmd_float mass);
/* Upload owned (+ optionally ghost) atoms. x: (nlocal+nghost)*3, v: nlocal*3 (NULL = zeros),
 * type: nlocal+nghost, tag: nlocal global ids (NULL = 0..nlocal-1). Replaces Atom::addatom/growarray
 * (ref/atom.cpp:71-100) for a whole array at once. */
int mmd_atom_upload(mmd_handle* h, const mmd_float* x, const mmd_float* v, const int* type, const int* tag,
                    int nlocal, int nghost);
/* Download in reference layout; any pointer may be NULL. x: (nlocal+nghost)*3, v: nlocal*3,
 * f: nf*3 with nf = nlocal (full lists) or nlocal+nghost (half lists), type: nlocal+nghost, tag: nlocal. */
/* positions only (x[3*(nlocal+nghost)], same atoms in the same order as the last mmd_atom_upload): what a reference Atom looks like after
 * initialIntegrate + Comm::communicate of a step without re-neighboring (ref/integrate.cpp:94-105). Types, velocities and the neighbor
 * list (built or uploaded) stay valid — a plugin uploads the list only when the reference's Neighbor::build has run. */
int mmd_atom_upload_x(mmd_handle* h, const mmd_float* x, int nall);
int mmd_atom_download(mmd_handle* h, mmd_float* x, mmd_float* v, mmd_float* f, int* type, int* tag);
int mmd_atom_upload_f(mmd_handle* h, const mmd_float* f, int n);       /* set atom.f (n atoms) */
int mmd_atom_counts(mmd_handle* h, int* nlocal, int* nghost, int* nmax);
int mmd_atom_pbc(mmd_handle* h);                                       /* Atom::pbc  ref/atom.cpp:106-122 */
int mmd_atom_sort(mmd_handle* h);                                      /* Atom::sort ref/atom.cpp:355-421 */

/* ---------------------------------------------------------------------------------------------
 * Neighbor  (ref/neighbor.h:39-90, ref/neighbor.cpp)
 * ------------------------------------------------------------------------------------------- */
/* Neighbor::setup (ref/neighbor.cpp:318-452): bin geometry from the box set by mmd_atom_set_box;
 * nbin = global bins per dimension (-b / 5/6*n, ref/ljs.cpp:351-371). */
int mmd_neighbor_setup(mmd_handle* h, const int nbin[3], mmd_float cutneigh, int halfneigh, int ghost_newton,
                       int ntypes);
/* Neighbor::build (ref/neighbor.cpp:79-213): bin owned+ghost atoms, build per-atom rows; grows
 * maxneighs and retries on overflow exactly like the reference (:186-208). */
int mmd_neighbor_build(mmd_handle* h);
/* bin grid incl. ghost margins (mbin, mbinlo as ref/neighbor.cpp:381-391), 2x2x2-bin blocks, stencil reach in blocks */
int mmd_neighbor_geometry(mmd_handle* h, int mbin[3], int mbinlo[3], int nblk[3], int reach[3]);
/* maxneighs (row stride), number of device bins, sum over rows, max row length of the last build */
int mmd_neighbor_info(mmd_handle* h, int* maxneighs, int* mbins, long long* total_neigh, int* max_row);
/* diagnostics of the device tile form of the list (full or half): out = {tiles, largest candidate union, sum of
 * candidate unions, sum of padded row counts, sum of atoms in tiles, longest padded row}; zeros when no tiles exist */
int mmd_neighbor_tile_stats(mmd_handle* h, long long out[6]);
/* diagnostics: histograms (nb bins of `width`, the last one open-ended) of the tiles' candidate-union sizes and padded row counts */
int mmd_neighbor_tile_histogram(mmd_handle* h, int nb, int width, long long* hist_ncand, long long* hist_rows);
/* diagnostic: raw tile form of one tile (padded rows of 16-bit LDS record offsets [k][64], the atoms of its 64 lanes, its candidate union) */
int mmd_neighbor_tile_rows(mmd_handle* h, int tile, unsigned short* rows, int rows_cap, int* kmax, int* atoms64, int* cand, int cand_cap, int* ncand);
/* rows in REFERENCE layout neighbors[i*maxneighs + k] (ref/neighbor.cpp:128); maxneighs = caller's stride.
 * Download: valid directly after mmd_neighbor_build (half lists with ghost newton are re-derived with the reference's partition rule from the
 * current positions). Upload: the rows are also turned into the library's tile form where they fit it (every entry within the cutoff the bins
 * were set up for), so the tile force kernels serve an uploaded list like one built here; otherwise the general row kernels do. */
int mmd_neighbor_download(mmd_handle* h, int* neighbors, int maxneighs, int* numneigh);
int mmd_neighbor_upload(mmd_handle* h, const int* neighbors, int maxneighs, const int* numneigh, int nlocal);

/* ---------------------------------------------------------------------------------------------
 * Force  (abstract class Force ref/force.h:40-69; ForceLJ ref/force_lj.cpp; ForceEAM ref/force_eam.cpp)
 * ------------------------------------------------------------------------------------------- */
/* ForceLJ::ForceLJ/setup (ref/force_lj.cpp:41-69) + table fill (ref/ljs.cpp:299-305); tables are
 * ntypes*ntypes, indexed type_i*ntypes+type_j. */
int mmd_force_lj_setup(mmd_handle* h, int ntypes, const mmd_float* cutforcesq, const mmd_float* sigma6,
                       const mmd_float* epsilon);
/* ForceEAM::setup result (ref/force_eam.cpp:732-761): 7-coefficient splines with row stride
 * nr_tot / nrho_tot per type pair. Host-side table construction: mmd_eam_tables_from_file(). */
int mmd_force_eam_setup(mmd_handle* h, int ntypes, int nr, int nrho, int nr_tot, int nrho_tot, mmd_float rdr,
                        mmd_float rdrho, const mmd_float* rhor_spline, const mmd_float* frho_spline,
                        const mmd_float* z2r_spline, const mmd_float* cutforcesq);
/* Force::compute (ref/force_lj.cpp:72-113, ref/force_eam.cpp:82-91). evflag as Force::evflag;
 * eng_vdwl / virial (may be NULL) are written only when evflag != 0, with the reference's own
 * conventions (full lists: 4*sum over both directions / 0.5*virial — ref/force_lj.cpp:441-442). */
int mmd_force_compute(mmd_handle* h, int evflag, double* eng_vdwl, double* virial);
int mmd_force_eam_download_fp(mmd_handle* h, mmd_float* fp);           /* fp[nlocal+nghost] after EAM compute */
/* ForceEAM::communicate (ref/force_eam.cpp:851-887: the halo of fp = F'(rho) between the two sweeps of ForceEAM::compute) as a
 * customisation point for callers whose ghost atoms were NOT made by this library's Comm::borders (mmd_atom_upload of a reference
 * Atom incl. its ghosts): between the sweeps the library hands `fp` to the callback as a host array of nlocal + nghost values with the
 * owned part filled in; the callback fills the ghost part (the reference plugin calls ITS ForceEAM::communicate on its Comm's send lists,
 * tests/integration/force_eam_hip.h) and returns 0. fn = NULL removes it (then the handle's own Comm serves the halo). */
typedef int (*mmd_fp_halo_fn)(void* ctx, mmd_float* fp, int nlocal, int nghost);
int mmd_force_eam_set_fp_halo(mmd_handle* h, mmd_fp_halo_fn fn, void* ctx);

/* ---------------------------------------------------------------------------------------------
 * Comm  (ref/comm.h:39-102, ref/comm.cpp)
 * ------------------------------------------------------------------------------------------- */
/* Comm::setup (ref/comm.cpp:60-272): processor grid, neighbors, sub-box bounds (written back to the
 * handle's box), ghost-layer counts and the 2*sum(need) swaps. prd must already be set. */
int mmd_comm_setup(mmd_handle* h, mmd_float cutneigh, int me, int nprocs);
/* geometry read-back (host-only; no GPU needed) */
int mmd_comm_info(mmd_handle* h, int procgrid[3], int myloc[3], int procneigh[6], int need[3], int* nswap);
int mmd_comm_swap_info(mmd_handle* h, int iswap, double slab[2], int pbc[4], int procs[2], int counts[3]);
/* attach the multi-GPU transport: RCCL communicator built from a 128-byte ncclUniqueId */
int mmd_comm_unique_id(unsigned char id[128]);
int mmd_comm_init_rccl(mmd_handle* h, const unsigned char id[128], int rank, int nranks);
/* transport in use: kind 1 = RCCL (nranks / rank from ncclCommCount / ncclCommUserRank), 2 = host-staged callbacks, 0 = none */
int mmd_comm_transport_info(mmd_handle* h, int* kind, int* nranks, int* rank);
/* host-staged transport for tests / fallback (e.g. torch.distributed gloo): the callback must
 * exchange nsend bytes to `dest` and receive up to nrecv_max bytes from `src`, returning bytes received */
typedef long long (*mmd_sendrecv_fn)(void* ctx, const void* sendbuf, long long nsend, int dest, void* recvbuf,
                                     long long nrecv_max, int src);
typedef int (*mmd_allreduce_fn)(void* ctx, double* vals, int n);
int mmd_comm_set_host_transport(mmd_handle* h, mmd_sendrecv_fn sr, mmd_allreduce_fn ar, void* ctx);
int mmd_comm_exchange(mmd_handle* h);              /* Comm::exchange            ref/comm.cpp:364-597 */
int mmd_comm_borders(mmd_handle* h);               /* Comm::borders             ref/comm.cpp:700-883 */
int mmd_comm_communicate(mmd_handle* h);           /* Comm::communicate         ref/comm.cpp:276-317 */
int mmd_comm_reverse_communicate(mmd_handle* h);   /* Comm::reverse_communicate ref/comm.cpp:321-355 */
int mmd_comm_download_lists(mmd_handle* h, int iswap, int* sendlist);  /* sendlist[iswap][0..sendnum) */

/* ---------------------------------------------------------------------------------------------
 * Integrate / Thermo  (ref/integrate.cpp, ref/thermo.cpp)
 * ------------------------------------------------------------------------------------------- */
/* dt and the fully scaled dtforce (0.5*dt [/mvv2e] /mass — ref/integrate.cpp:43,80-81, thermo.cpp:69) */
int mmd_integrate_setup(mmd_handle* h, mmd_float dt, mmd_float dtforce, int neigh_every, int sort_every);
int mmd_integrate_initial(mmd_handle* h);          /* Integrate::initialIntegrate ref/integrate.cpp:46-57 */
int mmd_integrate_final(mmd_handle* h);            /* Integrate::finalIntegrate   ref/integrate.cpp:59-68 */
/* --check_exchange (ref/integrate.cpp:112-151, 168-169): mark the owned atoms' positions (the reference's xold copy
 * after borders) / largest distance any owned atom moved since the mark, with the reference's +-prd correction.
 * mmd_integrate_run does both and prints the reference's warning when mmd_set_option(h, "check_exchange", 1). */
int mmd_integrate_mark_positions(mmd_handle* h);
int mmd_integrate_max_move(mmd_handle* h, double* d_max);
/* sum_i m v_i^2 over owned atoms (the loop of Thermo::temperature, ref/thermo.cpp:151-157) */
int mmd_thermo_temperature(mmd_handle* h, double* sum_mv2);
/* Called on thermo steps with globally reduced raw sums; the host applies the unit scales
 * (ref/thermo.cpp:119-194) and prints the row. */
typedef void (*mmd_thermo_fn)(void* ctx, int step, double sum_mv2, double eng_vdwl, double virial);
/* Integrate::run (ref/integrate.cpp:70-207): ntimes steps starting at step number `first_step`
 * (thermo fires when (n+1) % nstat == 0, n counted from first_step). */
int mmd_integrate_run(mmd_handle* h, int first_step, int ntimes, int thermo_nstat, mmd_thermo_fn cb, void* ctx);
/* wall-clock buckets of the last run: TOTAL, COMM, FORCE, NEIGH, TEST(extra) (ref/timer.h:35-40),
 * then GPU-event time of Force::compute (sum ms over the TIMED calls) and the number of timed calls: the clock is read on
 * every 7th call of a run (option "time_force_sample": a dispatch that carries the event pair costs ~11 us of gaps around it); FORCE in
 * out5 is their mean times the number of calls */
int mmd_timers(mmd_handle* h, double out5[5], double* force_kernel_ms, int* force_kernel_launches);
/* diagnostics of the last mmd_integrate_run on this rank: how often the host blocked on the GPU stream (count handshakes of
 * exchange / borders, list-size read-backs, thermo rows) and how many bytes it sent to OTHER ranks (halos, migrating atoms,
 * ghost lists, handshakes). transport_syncs counts, apart from host_syncs, the waits of the host-staged TEST transport (staging a message
 * through host memory; RCCL has none). bench.py reports them per step so that a multi-GPU line is diagnosable from the record alone. */
int mmd_run_stats(mmd_handle* h, long long* host_syncs, long long* bytes_sent, long long* transport_syncs);
/* read-only diagnostic counters of a handle (since mmd_create): "exchange_fast" / "exchange_overflows" = Comm::exchange calls served by the
 * handshake-free path / finished by the count-handshake path after a fixed-size message overflowed; "borders_fast" / "borders_general" =
 * Comm::borders calls served by the device-resident path / the swap-by-swap path; "device_bins_coarser" = 1 when the device bins coarser than the
 * reference's `-b` grid (mmd_neighbor_setup); "tiles_ready" / "rows_uploaded" = state of the current neighbor list; "spec_runs" / "spec_fails" = force launches
 * issued behind a neighbor build / of those the build's verdict turned into no-ops; "bin_reuses" = build binnings that placed the ghosts only. No reference counterpart. */
int mmd_get_counter(mmd_handle* h, const char* name, long long* value);
/* time `nrep` launches of one hot kernel with hipEvents on the handle's compute stream.
 * which: 0 = force (current style, evflag=0), 1 = neighbor build, 2 = initial integrate, 3 = final integrate */
int mmd_profile_kernel(mmd_handle* h, int which, int nrep, double* avg_ms);
int mmd_set_option(mmd_handle* h, const char* name, int value);       /* tuning knobs: the table of DESIGN_HISTORY.md section 4.5 (defaults are the measured best) */
int mmd_sync(mmd_handle* h);

/* ---------------------------------------------------------------------------------------------
 * Host-side setup shared by the `miniMD` executable and the Python mirror (no GPU needed)
 * ------------------------------------------------------------------------------------------- */
typedef struct {                          /* struct In, ref/ljs.h:37-51 */
  int nx, ny, nz;
  mmd_float t_request, rho;
  int units;                              /* 0 = lj, 1 = metal */
  int forcetype;                          /* 0 = lj, 1 = eam   */
  mmd_float epsilon, sigma;
  char datafile[1000];
  int has_datafile;
  int ntimes;
  mmd_float dt;
  int neigh_every;
  mmd_float force_cut, neigh_cut;
  int thermo_nstat;
} mmd_input;
/* input() (ref/input.cpp:48-187): returns 0 ok, 1 = cannot open / malformed */
int mmd_input_read(mmd_input* in, const char* filename);
/* create_box (ref/setup.cpp:305-311) */
int mmd_create_box(int nx, int ny, int nz, double rho, mmd_float prd[3]);
/* create_atoms (ref/setup.cpp:315-450) for the sub-box lo/hi: two-call protocol, first with
 * x=v=type=tag=NULL to get the count. Types follow rand()%ntypes after srand(5413) (ref/atom.cpp:97). */
int mmd_create_atoms(int nx, int ny, int nz, double rho, const mmd_float lo[3], const mmd_float hi[3], int ntypes,
                     mmd_float* x, mmd_float* v, int* type, int* tag, int* nlocal);
/* read_lammps_data (ref/setup.cpp:55-301), parsing part: header ("<n> atoms", "xlo xhi" ...; the box starts at 0 like the
 * reference assumes) and the Atoms / Velocities / Masses sections. Two-call protocol: with x = v = NULL only natoms,
 * prd and mass (-1 when the file has no Masses section) are returned; then x, v (3*natoms each) indexed by file id-1. */
int mmd_lammps_data_read(const char* file, int* natoms, mmd_float prd[3], mmd_float* mass, mmd_float* x, mmd_float* v);
/* the atoms of sub-box [lo,hi) in file order with rand()%ntypes types (ref/setup.cpp:281-286, ref/atom.cpp:86-100);
 * two-call protocol like mmd_create_atoms; tag = file id */
int mmd_lammps_data_select(int natoms, const mmd_float* x_all, const mmd_float* v_all, const mmd_float lo[3], const mmd_float hi[3],
                           int ntypes, mmd_float* x, mmd_float* v, int* type, int* tag, int* nlocal);
/* EAM funcfl file -> spline tables (ref/force_eam.cpp:505-793). Two-call protocol: with the three
 * spline pointers NULL only the sizes/scalars are returned. */
int mmd_eam_tables_from_file(const char* filename, int ntypes, int* nr, int* nrho, int* nr_tot, int* nrho_tot,
                             mmd_float* rdr, mmd_float* rdrho, mmd_float* cutmax, mmd_float* mass,
                             mmd_float* rhor_spline, mmd_float* frho_spline, mmd_float* z2r_spline);

/* ---------------------------------------------------------------------------------------------
 * Whole-program twin of ref/ljs.cpp main(): same CLI, same stdout grammar
 * ------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * Process-level contract of a multi-rank run. The reference asks MPI for rank and size (MPI_Comm_rank / MPI_Comm_size,
 * ref/ljs.cpp:63-68) and its harness starts `${MPISTART} -np N ./miniMD ...` (ref/run_one_test:50). Without linking MPI the same
 * launchers work through what they export to their children:
 *   torchrun RANK / WORLD_SIZE / LOCAL_RANK — Open MPI OMPI_COMM_WORLD_{RANK,SIZE,LOCAL_RANK} — MPICH / Intel MPI (hydra) PMI_RANK /
 *   PMI_SIZE / MPI_LOCALRANKID — PMIx PMIX_RANK — Slurm SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID   (first match in this order; none: 1 rank).
 * mmd_launch_rendezvous: MASTER_ADDR / MASTER_PORT when exported, else 127.0.0.1 and a port derived from the launcher's job id (or the
 * parent's pid), the same on every rank of the job. mmd_mesh_*: the TCP streams the ranks meet on (rank 0 listens on port + 17); the mesh
 * carries the RCCL id and, when the ranks of a node outnumber its GPUs, is itself the (debug, host-staged) transport: mmd_mesh_sendrecv /
 * mmd_mesh_allreduce are mmd_sendrecv_fn / mmd_allreduce_fn with the mesh as ctx (MPI_Sendrecv, ref/comm.cpp:291-297; MPI_Allreduce SUM,
 * ref/thermo.cpp:131-133 — summed in rank order, the same bits on every rank). Host code only: usable without a GPU.
 * ------------------------------------------------------------------------------------------- */
int mmd_launch_env(int* rank, int* nranks, int* local_rank, int* local_size, char* launcher, int launcher_len);
int mmd_launch_rendezvous(char* addr, int addr_len, int* port);
typedef struct mmd_mesh mmd_mesh;
int mmd_mesh_create(int rank, int nranks, const char* addr, int port, mmd_mesh** out);
int mmd_mesh_destroy(mmd_mesh* m);
int mmd_mesh_allgather(mmd_mesh* m, const void* mine, int nbytes, void* all);
long long mmd_mesh_sendrecv(void* mesh, const void* sendbuf, long long nsend, int dest, void* recvbuf, long long nrecv_max, int src);
int mmd_mesh_allreduce(void* mesh, double* vals, int n);
int mmd_mesh_info(mmd_mesh* m, int* rank, int* nranks, long long* bytes_sent, long long* messages);

typedef struct mmd_sim mmd_sim;
/* parses argv (ref/ljs.cpp:87-261) + deck, builds the system and uploads it (ref/ljs.cpp:264-443).
 * Several ranks: one process per rank, rank / size from mmd_launch_env. Transport, first match: the host callbacks of
 * mmd_sim_set_host_transport; RCCL with the id of mmd_sim_set_unique_id; otherwise the ranks meet on the mesh and use RCCL when every
 * rank of every node has a GPU of its own, else (or with MMD_TRANSPORT=tcp) the mesh itself — named in the banner ("# Transport: ..."). */
int mmd_sim_set_unique_id(const unsigned char id[128]);
/* use a host-staged transport (see mmd_comm_set_host_transport) for the sims created afterwards */
int mmd_sim_set_host_transport(mmd_sendrecv_fn sr, mmd_allreduce_fn ar, void* ctx);
int mmd_sim_create(int argc, char** argv, int quiet, mmd_sim** out);
int mmd_sim_initial(mmd_sim* s);                  /* ref/ljs.cpp:445-468 */
int mmd_sim_run(mmd_sim* s);                      /* ref/ljs.cpp:470-483 */
int mmd_sim_run_steps(mmd_sim* s, int nsteps, double* seconds);   /* timed re-entrant slice for bench.py */
int mmd_sim_print_perf(mmd_sim* s);               /* ref/ljs.cpp:485-495 */
/* YAML report of -o / --yaml_output and --yaml_screen: output() + stats() (ref/output.cpp:48-547) */
int mmd_sim_output(mmd_sim* s, int screen_yaml);
int mmd_sim_wants_yaml(mmd_sim* s, int* screen_yaml);   /* returns the -o level parsed from the command line */
int mmd_sim_rows(mmd_sim* s, int* nrows, int* steps, double* t, double* u, double* p, int maxrows);
int mmd_sim_natoms(mmd_sim* s);
mmd_handle* mmd_sim_handle(mmd_sim* s);
int mmd_sim_destroy(mmd_sim* s);

#ifdef __cplusplus
}
#endif
#endif
