// minimd_amd/csrc/mmd_internal.hpp — internal state of one device handle (not part of the C-ABI).
//
// Data layout in HBM (gfx950, one handle per GPU):
//   x      real4[nmax+1]   position + type packed in .w ((real)type): ONE 32-byte (DP) / 16-byte (SP)
//                          aligned gather per neighbor in the force kernels; slot [nlocal+nghost] is the
//                          far-away "dummy" atom that pads neighbor rows (never inside any cutoff)
//   v, f   real[3*nmax]    AoS stride 3 (streamed only, never gathered)
//   type, tag int[nmax]
//   neigh  int[nwaves*maxneighs*64]  wave-interleaved neighbor rows: entry k of atom i lives at
//                          ((i>>6)*maxneighs + k)*64 + (i&63) so the 64 lanes of a wavefront read one
//                          coalesced 256-byte line per k;  rows are padded with the dummy index up to the
//                          wavefront's longest row (rounded up to the unroll factor)
//   bins   bin_start[mbins+1], binned[nall]: counting-sorted atom indices, bins numbered block-major
//                          (2x2x2 bins = one block = ~one wavefront of atoms) for L1/L2 locality
#pragma once
#include <functional>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mmd.h"

typedef mmd_float real;
#if MMD_PRECISION == 1
typedef float4 real4;
#else
typedef double4 real4;
#endif

#define MMD_WAVE 64
#define MMD_UNROLL 8            // neighbor rows are padded to a multiple of this (>= every kernel's unroll factor)
#define MMD_BLOCK 256
#define FCLK_SLOTS 256            // launches of a run whose device-clock stamps are kept (the rest of a longer run is not stamped)
#define FCLK_TAIL 2048            // last-dispatched workgroups of a launch that store their end time (a slot of its own each: no same-address atomics)
#define FCLK_STRIDE (FCLK_TAIL + 8)
// record of the k-th stamped launch of a run: the first FCLK_SLOTS / 2 launches keep theirs, later ones share a ring of FCLK_SLOTS / 2 (the run's last launches survive)
static inline int fclk_slot(long long k) { return k < FCLK_SLOTS / 2 ? (int)k : FCLK_SLOTS / 2 + (int)((k - FCLK_SLOTS / 2) % (FCLK_SLOTS / 2)); }

// profiling-only phase switches of the tile kernels / the tile build ("ablate": results invalid). The shipped library is built
// WITHOUT -DMMD_PROFILE: every switch folds to 0 at compile time and mmd_set_option("ablate") is refused; tools/build_variant.sh
// builds the profiling variant (tools/prof_force.py ABLATE=1 uses it).
#ifdef MMD_PROFILE
#define MMD_ABLATE(a) (a)
#else
#define MMD_ABLATE(a) 0
#endif

void mmd_set_error(const char* fmt, ...);
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if(_e != hipSuccess) {                                                                         \
      mmd_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));     \
      return -1;                                                                                   \
    }                                                                                              \
  } while(0)
#define MMD_TRY(expr)                  \
  do {                                 \
    int _r = (expr);                   \
    if(_r < 0) return _r;              \
  } while(0)

// growable device array
template <typename T>
struct DevArr {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n, bool preserve, hipStream_t s, size_t preserve_count = (size_t)-1)
  {
    if(n <= cap) return 0;
    size_t ncap = n + n / 8 + 1024;
#ifdef MMD_DEBUG_REALLOC          // (tools/realloc_probe.py: which arrays still grow inside a run)
    fprintf(stderr, "ensure: %zu x %zu B -> %zu (had %zu) %s\n", n, sizeof(T), ncap, cap, __PRETTY_FUNCTION__);
#endif
    T* q = nullptr;
    HIP_TRY(hipMalloc((void**)&q, ncap * sizeof(T)));
    if(preserve && p && cap) {
      size_t cnt = preserve_count == (size_t)-1 ? cap : (preserve_count < cap ? preserve_count : cap);
      if(cnt) HIP_TRY(hipMemcpyAsync(q, p, cnt * sizeof(T), hipMemcpyDeviceToDevice, s));
    }
    if(p) {
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(hipFree(p));
    }
    p = q;
    cap = ncap;
    return 0;
  }
  void release()
  {
    if(p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct LJParams {            // uniform-table fast path (all type pairs identical, checked at setup)
  real cutforcesq, sigma6, epsilon;
};

struct BinGeom {             // Neighbor::setup result (ref/neighbor.cpp:318-452) + block-major numbering
  real prd[3], bininv[3], binsize[3];
  real sublo[3], subhi[3];   // this rank's sub-box (the whole box on one rank)
  int nbin[3], mbinlo[3], mbin[3];
  int blkshift[3];           // parity shift so that the periodic box edge falls on a block boundary
  int nblk[3];               // 2x2x2-bin blocks per dimension
  int reach[3];              // stencil reach in BLOCKS (covers ref's next{x,y,z} bins)
  int mbins;                 // nblk[0]*nblk[1]*nblk[2]*8
};

struct Swap {                // one of the 2*sum(need) swaps of Comm::setup (ref/comm.cpp:208-269)
  int dim;
  real slablo, slabhi;
  int pbc_any, pbc[3];
  int sendproc, recvproc;
  int sendnum = 0, recvnum = 0, firstrecv = 0;
  DevArr<int> sendlist;
};

struct EventPair { hipEvent_t a, b; int kind = 0; };

// A force launch enqueued behind a neighbor build whose result words the host has not read yet (Integrate::run, one rank): the kernel
// takes the build's verdict (`gate`, written by k_publish_flags: 1 = every list fits what this launch was sized for), the tile count and
// the ghost count from device memory. gate == nullptr: an ordinary launch.
// clk != nullptr: the launch stamps the device's wall clock — clk[0] the start of its first workgroup, clk[8 + q] the end of its q-th last
// workgroup (q < FCLK_TAIL; the host takes the latest) (force-kernel time of EVERY launch of a run, without an event packet on the stream: mmd_get_counter "force_clock_ns")
struct SpecLaunch { const int* gate; const int* ntiles_dev; const int* nghost_dev; unsigned long long* clk; };

// Direct halo (several ranks, need = 1 in every dimension): the forward communication of a step as ONE exchange with the up to 26 neighbours instead
// of three dependent rounds (one per dimension, the later ones forwarding what the earlier ones received). The ghosts of a rank are 26 lists — one
// per combination (x swap | none, y swap | none, z swap | none) — and list L holds, in ascending index order, the OWNED atoms of the rank at grid
// offset -dir(L) that lie in all slabs of L (comm.hip: BrdList); the swap-by-swap order of ref/comm.cpp:364-597 is these lists laid end to end. Once
// per re-neighboring every rank compacts its 26 send lists and exchanges their lengths; per step it packs once and receives straight into the ghost slots.
struct DirectHalo {
  bool ready = false, pending = false;
  int opt = 1;
  int ns[26], nr[26], soff[27], rbase[27], target[26], source[26];
  // one message per distinct partner: the lists that go to (come from) the same rank travel end to end, in list order
  int sdst[26], rsrc[26];     // offset of list l inside the send / receive buffer (entries); -1: the list stays on this rank
  int npeer_s = 0, npeer_r = 0, peer_s[26], peer_r[26], peer_soff[27], peer_roff[27];
  int lps[26], lpr[26];       // list l travels in outgoing message lps[l] / arrives in incoming message lpr[l] (-1: it stays on this rank)
  int code[26];               // periodic-image code the ghosts of list l carry (image_add over the list's swaps, as pack_border accumulates it)
  int total_recv = 0;         // entries that arrive in the receive buffer
  int nsrc = 0, src_rank[26]; // RCCL: the distinct ranks whose 26 list lengths arrived (block k of the pinned copy)
  real shift[26][3];
  int total_send = 0;
  DevArr<int> idx;            // the 26 send lists, end to end
  DevArr<int> counts;         // device: [0..25] lengths of my send lists, [32..57] of the lists I receive
  DevArr<int> scratch;
  int* h_counts = nullptr;    // pinned copy of `counts` (32 x 30 ints)
  int* h_counts_dev = nullptr; // device view of h_counts (direct borders: k_db_unpack writes the lengths there itself)
  // Direct borders (comm.hip, borders_direct): the plan of the previous re-neighboring sizes this one's fixed messages
  bool prev_valid = false;
  int ns_prev[26], nr_prev[26];
  int opt_borders = 1;        // 1: Comm::borders on several ranks as ONE exchange of the 26 lists where a previous plan exists; 0: swap by swap
  int opt_recv = 3;           // per-step halo: 1 = one message per partner + k_dh_unpack; 3 = the same, and where the tile kernels can follow (LJ full lists) the partners' messages
                              // land in the position buffer itself, behind the ghost slots, and the boundary tiles read them there: no k_dh_unpack on the step
                              // (2, round 5: every list a message of its own received straight into its ghost slots — 52 p2p operations per group, 6890 against 8710 Matom-steps/s: removed)
  // halo_recv 3: the position buffers are longer than owned + ghosts; entry R + o is record o of the per-step receive layout (one message per partner, in partner
  // order). gmap[g] (written by k_db_unpack) = where ghost g's position arrives: R + o, or its own slot nlocal + g for a list that stays on this rank; the build
  // writes the boundary tiles' candidate lists once more with every ghost named by gmap (tile_cand_src), and a step's force kernel stages from there.
  DevArr<int> gmap;
  int R = 0;
  bool gmap_live = false;     // gmap describes the ghosts of the current borders (set when they were made by the direct borders)
  bool x_unpack_pending = false;   // the last position halo was received behind the ghost slots and has not been copied into them (mmd_ghosts_refresh does)
};

struct mmd_handle {
  int device = 0;
  bool host_only = false;    // geometry-only handle (mmd_create(-2)): no GPU, host functions only
  hipStream_t stream = nullptr, comm_stream = nullptr;
  hipDeviceProp_t prop;
  // ---- Atom
  real prd[3] = {0, 0, 0}, lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  real mass = 1;
  int nlocal = 0, nghost = 0, nmax = 0;
  DevArr<real4> x, x_alt;
  DevArr<real4> xold;          // positions marked at the last re-neighboring (--check_exchange, ref/integrate.cpp:168-169)
  int xold_n = -1;
  DevArr<real> v, v_alt, f;
  DevArr<real> x_stage;        // AoS-3 staging of mmd_atom_upload_x
  DevArr<int> type, type_alt, tag, tag_alt;
  // ---- Neighbor
  bool neigh_ready = false;
  real cutneigh = 0, cutneighsq = 0;
  int halfneigh = 0, ghost_newton = 0, ntypes = 1;
  BinGeom bg;                // device bins (= the reference's unless those are too fine for the build kernels, mmd_neighbor_setup)
  BinGeom bg_ref;            // the reference's bins (Neighbor::setup)
  DevArr<int> bin_count, bin_start, bin_start_alt, binned, scan_tmp, atom_bin;
  DevArr<int> atom_rank;       // arrival rank of each atom inside its bin (k_bin_count)
  int maxneighs = 100;       // row stride (multiple of MMD_UNROLL)
  DevArr<int> neigh, numneigh, wave_max;
  int neigh_nlocal = 0;      // nlocal the list was built for
  long long total_neigh = 0;
  int max_row = 0;
  DevArr<int> ghost_image;   // per ghost: packed periodic-image code (half lists with ghost newton)
  bool ghost_chain_ok = false;
  DevArr<int> ghost_root;    // per ghost: owned atom it is an image of (valid when every swap is a self swap)
  // block-local ("tile") neighbor list used by the LDS force kernels: a tile = up to 64 consecutive
  // entries of binned[] inside one block; nl16[(tile*maxneighs + k)*64 + lane] = slot of the neighbor in
  // the block's candidate sequence (the order in which k_build walks the surrounding blocks)
  bool tiles_ready = false;
  bool rows_uploaded = false;            // the current list came through mmd_neighbor_upload (its rows are what a download returns, whatever the mode)
  bool rows_ready = false;               // the wave-interleaved 32-bit rows (`neigh`, wave_max) are materialised
  int ntiles = 0, tile_tmax = 0;          // tile_tmax: largest candidate count of any block (LDS sizing)
  DevArr<int> tile_of_block, tile_block, tile_first, tile_max;
  DevArr<unsigned> pencil_lohi;           // per pencil: first / last (+1) bin holding an owned atom, collected by k_bin_sort when a build asks for it
  bool pencil_lohi_req = false, pencil_lohi_ready = false;
  DevArr<int> pencil_range;               // per pencil (row of blocks along x): [first, last) entry of binned[] between its first and last owned bin
  DevArr<int> tile_cand, tile_ncand, tile_cnt;
  DevArr<int> tile_cand_src;              // one-rank runs: tile_cand with every ghost named by its owner + image code (GhostResolve, tile_lds.hpp)
  bool cand_src_ready = false;
  bool cand_src_halo = false;            // tile_cand_src names the ghosts by where the per-step halo delivers them (DirectHalo::gmap), not by owner + image code
  bool halo_in_x_allow = false;          // transient (Integrate::run): this step's position halo may stay behind the ghost slots (the force launch that follows reads it there)
  DevArr<real> box_dev;                   // the box lengths in device memory (ghost_shifted fetches them inside its rare branch)
  bool box_dev_valid = false;
  DevArr<unsigned short> tile_self;       // half lists: union slot of each tile atom itself (0xffff: not in the union)
  // Rows in two parts (full lists, one rank): pairs closer than `radius` = cutforce + margin at the build come first ("core"), the rest of
  // the skin behind them; tile_kcore = padded length of the core part. A pair of the rest can only come inside the force cutoff after
  // its two atoms have moved `margin` towards each other, so for as long as no atom has moved further than margin/2 since the build a
  // force kernel may stop after the core part — and computes exactly the same forces. The fused integrator of the tile force kernels
  // tracks the largest displacement since the build (xbuild) in `words` (3 sets of 64 unsigned = float bits of d^2: the kernel of step n
  // writes set n%3, the kernel of step n+1 reads it, set (n+1)%3 is zeroed meanwhile).
  struct CoreRows {
    real radius = 0;             // 0: off
    real margin = 0;
    bool rows_built = false;     // the current tile lists are in two parts
    int mode_now = 0;            // set by Integrate::run around a force call: 0 whole rows, 1 core part (positions are those of the build),
                                 // 2 core part if the displacement tracked by the previous launch allows it
    bool tracked_last = false;   // the last force call advanced the atoms inside the kernel and recorded their displacement
    unsigned step = 0;           // launches with tracking so far (selects the word sets)
  } core;
  DevArr<int> tile_kcore;
  DevArr<real4> xbuild;
  DevArr<unsigned> core_words;
  DevArr<unsigned> tile_words;            // scratch of k_build_rows: per tile the hit words of its candidate groups [group][lane]
  DevArr<int> tile_rowmax, tile_rowsum;   // per tile: longest row / sum of the row lengths (reduced by k_tile_reduce)
  DevArr<int> tile_ghost, tile_order;     // per tile: references a ghost atom?; tiles ordered interior-first
  int ntiles_interior = 0;
  hipEvent_t ev_x_ready = nullptr, ev_halo_done = nullptr;
  // several ranks: the forward halo of a step on the communication stream under the interior tiles, the boundary tiles behind it. 1 on, 0 off, 2 on (one
  // rank too); -1 (default) = decided by measurement: after the first re-neighboring of a run a few steps are timed in each form, the times are summed
  // over the ranks, and every rank keeps the faster form (overlap_choice; the split costs two launches per step + the tile order per build, and only a real
  // transfer can pay for it — in RCCL loop-back on one GPU it does not)
  int opt_overlap = -1;
  int overlap_choice = -1;           // -1 not decided yet, 0 / 1
  bool trial_armed = false;          // a re-neighboring has run since the last trial attempt: the next plain stretch of 2 x 8 steps (in this or the next mmd_integrate_run) is timed
  // RCCL bring-up self-check (mmd_comm_init_rccl): distinct partners exchanged with, seconds it took, PCI bus id of this rank's device
  int rccl_check_partners = 0;
  double rccl_check_s = 0;
  char pci[32] = "?";
  // LJ full lists, overlapped step: 1 = the boundary tiles are launched on the COMMUNICATION stream right behind the transfer (they run under the tail of the
  // interior tiles, the compute stream only joins at the end of the step); 0 = on the compute stream behind a wait for the halo (round 4)
  long long overlap_join_steps = 0;
  double overlap_trial_s[2] = {0, 0}; // per step, summed over the ranks: [0] without, [1] with overlap
  hipEvent_t ev_trial[3] = {nullptr, nullptr, nullptr};
  int tile_cstride = 0, tile_cmax = 0;
  DevArr<unsigned short> nl16;
  int opt_tiles = 1;
  int opt_force_transport = 0;   // testing: route self-swaps through the transport too (RCCL loop-back on one GPU)
  int opt_safe_exchange = 0;                      // Comm::do_safeexchange (ref/comm.h:87): Comm::exchange offers leavers to every rank within `need` sub-domains
  int opt_check_exchange = 0;                     // --check_exchange: warn when an atom moved further than a sub-domain
  int opt_fuse = 2;          // >=1: fused final+initial integrate, single-kernel ghost update on one rank; 2: integrator inside the LJ tile kernel
  int fuse_now = 0;          // transient: the next tile launch carries the integrator
  bool halo_pending = false; // transient: this step's position halo is in flight on the communication stream (ev_halo_done behind it)
  const void* xalt_dummy_ptr[2] = {nullptr, nullptr}; int xalt_dummy_slot[2] = {-1, -1}; int xalt_dummy_next = 0;
  int opt_ablate = 0;        // profiling only: 1 = skip LDS staging, 2 = skip the neighbor loop (results invalid)
  // ---- Force
  int style = 0;             // 0 LJ, 1 EAM
  bool lj_uniform = true;
  LJParams lj;
  DevArr<real> lj_tables;    // [3][ntypes*ntypes]: cutforcesq, sigma6, epsilon
  std::vector<real> h_cutforcesq;
  // EAM
  int nr = 0, nrho = 0, nr_tot = 0, nrho_tot = 0;
  real rdr = 0, rdrho = 0;
  bool eam_uniform = true;
  int opt_eam_mlo = -1;      // first spline knot kept in LDS by the EAM tile kernels (-1: 0.3 x cutoff)
  int eam_diag[4] = {0, 0, 0, 0};   // last full-list tile launch: LDS bytes of the two sweeps, workgroups per CU of their persistent grids
  bool eam_attr_set = false; // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for this handle's device
  DevArr<real> rhor_spline, frho_spline, z2r_spline, fp, rho;
  // ---- Comm
  int me = 0, nprocs = 1;
  int procgrid[3] = {1, 1, 1}, myloc[3] = {0, 0, 0}, procneigh[3][2] = {{0, 0}, {0, 0}, {0, 0}}, need[3] = {1, 1, 1};
  std::vector<Swap> swaps;
  DevArr<real> buf_send, buf_recv;
  void* rccl = nullptr;      // ncclComm_t
  mmd_sendrecv_fn host_sr = nullptr;
  mmd_allreduce_fn host_ar = nullptr;
  void* host_ctx = nullptr;
  std::vector<char> stage_send, stage_recv;
  DevArr<int> flag_tmp, bnd_list, bstate;
  DirectHalo dh;
  DevArr<unsigned char> brd_bits;      // one-rank borders in three launches: per owned atom, which of the six send slabs hold it
  DevArr<unsigned char> ghost_bits;    // direct borders: the same bits of every ghost, as its owner computed them (the send lists of the swaps are derived from them on demand)
  bool sendlists_stale = false;        // the ghosts came from the direct borders: swaps[q].sendlist is built when somebody asks (mmd_comm_sendlists_ensure)
  long long borders_direct_runs = 0, halo_in_x_steps = 0;
  bool borders_direct_pending = false; // the borders in flight are the direct form (borders_fast_finish)
  DevArr<int> est, ex_list;            // handshake-free Comm::exchange (comm.hip): device-resident counts / leaver list
  int ex_prev_send[3] = {0, 0, 0}, ex_prev_recv[3][2] = {{0, 0}, {0, 0}, {0, 0}};     // migration counts of the last exchange (size the fixed messages)
  bool ex_prev_valid = false;
  int opt_exchange_cap = 0;            // > 0: record capacity of the fixed-size exchange messages (tests force the overflow protocol); 0: 4 x previous + 4096
  long long ex_fast = 0, borders_fast_runs = 0, borders_general_runs = 0;      // how often each path ran (mmd_get_counter)
  int ex_overflows = 0;                // exchanges whose fixed-size messages overflowed and were finished by the handshake path (diagnostic, tests)
  bool borders_general_done = false;   // a swap-by-swap Comm::borders has run (several ranks: the collective condition for the fixed-size-message path)
  bool big_bins = false;       // some bin holds more than NB_BIGBIN atoms: binning runs the grid-wide rank sort too
  bool in_reneighbor = false;  // inside Integrate::run's re-neighboring: Comm::borders follows Atom::sort, ghosts need not ride along
  // one-rank LJ full-list steps: the tile kernel stages ghosts from their owners, no per-step Comm::communicate. 1 = where it pays: whenever the
  // build left the candidate lists with the ghosts named by owner + image code (tile_cand_src: no look-up in front of the position load: +0.9 % at
  // -s 80, +4.6 % at -s 32), otherwise on small systems only (the look-up costs the boundary tiles a round trip); 2 = always; 0 = never
  int opt_ghost_resolve = 1;
  // one-rank half-list LJ steps with ghost newton: the tile kernel adds a ghost's share to its owner (no Comm::reverse_communicate)
  int opt_fold_reverse = 1;
  bool fold_reverse_now = false;
                                       // (1: small systems, where the saved launch shows — no difference at -s 64; 2: always; 0: never)
  bool fp_ghosts_stale = false;
  bool ghosts_uploaded = false;          // the ghost atoms came through mmd_atom_upload, not from this handle's Comm::borders (no send lists, no ghost_root)
  mmd_fp_halo_fn fp_halo_fn = nullptr;   // caller-supplied ForceEAM::communicate (mmd_force_eam_set_fp_halo)
  void* fp_halo_ctx = nullptr;
  std::vector<real> fp_stage;
  bool eam_half_attr_set = false;
  const int* nghost_dev = nullptr;     // != nullptr: one-rank borders enqueued, ghost count still on the device (nghost holds a bound)
  int bf_est_nb = 0;
  bool pbc_defer = false, pbc_pending = false;     // Atom::pbc folded into the binning pass of the Atom::sort that follows
  // the binning Atom::sort did inside this re-neighboring left the owned atoms in bin order and their counts in the histogram: the build's binning places the ghosts only (mmd_bin_atoms)
  bool bin_owned_valid = false; int bin_owned_n = 0, bin_owned_mbins = 0; long long bin_reuses = 0;
  int bin_count_clean = -1;            // mbins for which bin_count is known to be all zero (k_bin_sort leaves it so)
  int opt_core_pct = 30;               // EAM full lists on one rank: rows in two parts, the core part ends this many per cent into the skin (0: off)
  bool zero_f_in_integrate = false;    // Integrate::run, one-rank half lists: k_final_initial_integrate clears f behind itself
  int f_zeroed_n = 0;                  // f[0 .. 3*f_zeroed_n) is known to be zero (consumed by the next half-list Force::compute)
  int opt_lj_original = 0;             // --half_neigh -1: ForceLJ::compute_original (ref/force_lj.cpp:118-176) = the row kernel k_lj_half, not the tile kernel
  int ntiles_hint = 0;
  int opt_time_sample = 0;             // force-kernel clock on every n-th Force::compute of a run (0: every 7th)
  int force_calls = 0;                 // Force::compute calls of the current run that went through the sampled clock
  int run_ntimes = 0;                  // length of the current mmd_integrate_run
  // device-clock stamps of the LJ full-list tile launches of the current run (FCLK_SLOTS records of FCLK_STRIDE words, harvested when the run ends)
  DevArr<unsigned long long> fclk;
  int fclk_n = 0;                      // launches stamped in this run
  double fclk_ms = 0;                  // sum of their durations (harvested)
  int fclk_launches = 0;
  bool fclk_harvested = true;
  bool fclk_sampled[FCLK_SLOTS] = {false};   // which of the stamped launches also carried an event pair
  double fclk_ms_sampled = 0; int fclk_launches_sampled = 0;
  double fclk_gap_ms = 0; int fclk_gaps = 0;     // idle time between stamped launches that follow each other directly
  double fclk_first_ms = 0, fclk_median_ms = 0, fclk_last_ms = 0;      // mean span of the run's first <= 100 stamped launches, median of all kept, mean of its last <= 100
  bool spec_clk_redo = false;          // the launch behind the build was cancelled: the launch that replaces it takes its clock slot
  long long force_sample_ctr = 0;      // the same, never reset: call number modulo the sampling period decides which launches carry the clock
  bool resolve_now = false, ghosts_stale = false;
  hipEvent_t launch_ev_a = nullptr, launch_ev_b = nullptr;     // event pair the next tile-kernel launch attaches to its dispatch
  int opt_borders_est = 150;    // device-resident borders: 0 off, 1 swap by swap (count / scatter pair per dimension), 2 + the three-launch form where every swap is a self swap; its sizing estimate in per cent of the previous counts
  int prev_nb = 0, prev_nghost = 0;   // counts of the last Comm::borders (size the device-resident one-rank path of the next one)
  // ---- Integrate
  real dt = 0, dtforce = 0;
  int neigh_every = 20, sort_every = 20;
  int next_sort = -1;        // global step number of the next Atom::sort (persists across mmd_integrate_run slices)
  // ---- reductions
  DevArr<double> partials;   // per-block partial sums
  double* h_result = nullptr;  // pinned host: [0..7]
  double* d_result = nullptr;
  int* h_flags = nullptr;      // pinned host ints
  int* h_flags_dev = nullptr;  // device view of h_flags (k_publish_flags)
  int flag_seq = 0;
  int clk_slot = -1;           // >= 0: the next k_bin_count stamps the device wall clock into result word pair 56 + 2 * clk_slot (phase clocks of a re-neighboring)
  int clk_written = 0;         // bit s: slot s was stamped in this re-neighboring; bit 2: the build published its words (incl. its own stamp)
  long long clk_last = 0;      // last stamp seen (stamps of one re-neighboring must lie behind it and ascend)
  double clk_rate_hz = 0;      // hipDeviceAttributeWallClockRate
  // Force::compute of a re-neighboring step launched BEHIND the build, before its result words have reached the host (no idle GPU while the
  // host reads them and enqueues the kernel): Integrate::run leaves the launch as a closure, mmd_neighbor_build calls it between publishing
  // the words and polling for them; the kernel does nothing unless the build's verdict on the device says the lists fit (SpecLaunch)
  int opt_spec = 16;                   // 0: off; > 0: on, the launch provides LDS for the previous build's largest candidate union + this many atoms
  std::function<int()> spec_fn;        // transient: how to launch this step's Force::compute
  SpecLaunch spec = {nullptr, nullptr, nullptr, nullptr};   // transient: set around spec_fn
  bool spec_fused = false;             // the gated launch carried the integrator (it wrote the second position buffer's dummy atom itself)
  bool spec_done = false;              // the build launched this step's force kernel and its verdict was "go"
  int spec_cmax = 0;                   // largest candidate union the speculative launch provides LDS for
  long long spec_launches = 0, spec_runs = 0, spec_fails = 0;
  bool in_run = false;         // inside mmd_integrate_run
  int* h_flags_big = nullptr;  // pinned host ints (64): read-back of the device-resident borders state
  std::vector<int> h_bstate;
  int* d_flags = nullptr;
  // ---- timers (ref/timer.h:35-40) + GPU events around the force kernel
  double timer[5] = {0, 0, 0, 0, 0};
  double timer_raw[5] = {0, 0, 0, 0, 0};      // the same before mmd_integrate_run made them partition the wall clock
  std::vector<EventPair> ev_pool;
  size_t ev_used = 0;
  double force_ms = 0, comm_ms = 0;
  double halo_ms = 0;                  // forward halos of the sampled steps (every 4th)
  hipEvent_t ovf_a = nullptr, ovf_b = nullptr;   // bracket of an overlapped step's two force launches
  bool ovf_open = false;
  int force_launches = 0;
  double force_ms_all = 0;             // Force::compute calls timed on every step (overlapped multi-rank steps), not sampled
  int force_launches_all = 0;
  // diagnostics of the last Integrate::run (mmd_run_stats): host synchronisations (hipStreamSynchronize / blocking copies issued by
  // the step loop and everything below it) and bytes this rank sent to OTHER ranks (halo, exchange, borders payloads + handshakes)
  long long host_syncs = 0, halo_bytes = 0;
  long long transport_syncs = 0;       // waits that belong to the host-staged TEST transport (device<->host staging of a message), not to the algorithm
  // ---- options
  int opt_force_block = MMD_BLOCK;
};

// ---- shared device/host helpers implemented across the .hip files
int mmd_ensure_atoms(mmd_handle* h, int n, bool preserve);
int mmd_set_dummy(mmd_handle* h);
int mmd_box_dev(mmd_handle* h);
int mmd_comm_sendlists_ensure(mmd_handle* h);      // the six send lists of ref/comm.cpp:700-883 where the direct borders left them to be derived
int mmd_ghosts_refresh(mmd_handle* h);             // ghost slots of x in step with their owners again (after steps whose force kernels read the ghosts elsewhere)
int mmd_dh_exchange(mmd_handle* h, int what);      // direct halo of a step: 0 positions (x), 1 EAM fp; only when h->dh.ready
int mmd_borders_deferred_finish(mmd_handle* h);
int mmd_run_reserve(mmd_handle* h);            // buffers the re-neighborings of a run will ask for, before its clock starts (comm.hip)
int mmd_borders_deferred_resolve(mmd_handle* h);
int mmd_exclusive_scan(mmd_handle* h, int* data, int n, int* total_host);   // in-place, returns total
int mmd_exclusive_scan_from(mmd_handle* h, const int* src, int* data, int n, int* total_host);   // src -> data (may alias)
int mmd_bin_atoms(mmd_handle* h, int count);
int mmd_lj_compute(mmd_handle* h, int evflag, double* eng, double* vir);
int mmd_eam_compute(mmd_handle* h, int evflag, double* eng, double* vir);
int mmd_zero_forces(mmd_handle* h, int n);
int mmd_lj_tiles_available(mmd_handle* h);
int mmd_lj_half_tiles_available(mmd_handle* h);
int mmd_lj_can_fuse_integrate(mmd_handle* h);
int mmd_eam_can_fuse_integrate(mmd_handle* h);
int mmd_prepare_x_alt(mmd_handle* h);        // second position buffer (capacity + dummy atom) for the fused force+integrate kernel
int mmd_lj_compute_tiles_split(mmd_handle* h, int evflag, int part);   // part 0: interior tiles, 1: boundary tiles + energy sum
int mmd_order_tiles(mmd_handle* h);
int mmd_ensure_rows(mmd_handle* h);       // materialise `neigh` from the tile form when a kernel needs it
int mmd_transport_sendrecv(mmd_handle* h, const void* dsend, size_t nsend, int dest, void* drecv, size_t nrecv, int src);
int mmd_transport_sendrecv_counts(mmd_handle* h, int nsend, int dest, int* nrecv, int src);
int mmd_transport_sendrecv_counts_pair(mmd_handle* h, const int nsend[2], const int dest[2], int nrecv[2], const int src[2]);
int mmd_transport_sendrecv_pair(mmd_handle* h, const void* const dsend[2], const size_t nsend[2], const int dest[2], void* const drecv[2],
                                const size_t nrecv[2], const int src[2]);
int mmd_transport_allreduce(mmd_handle* h, double* vals, int n);
double mmd_wall();

// every blocking wait on the handle's stream goes through here, so that a run can report how often the host stalled the GPU
static inline hipError_t mmd_stream_sync(mmd_handle* h) { h->host_syncs++; return hipStreamSynchronize(h->stream); }
// The caller's wait for the END of a run (Integrate::run's last kernel, mmd_sync): polled. hipStreamSynchronize parks the thread, and waking it
// costs 20-40 us that the run's own wall clock (TIME_TOTAL, ref/integrate.cpp:84-207 is timed by the host) would count on every call — 1 % of a
// 20-step slice at -s 80, 7 % at -s 32. A wait that outlasts `spin_s` falls back to the blocking form (nobody burns a core on a long queue).
static inline hipError_t mmd_stream_wait_polled(hipStream_t st, double spin_s = 0.05)
{
  const double t0 = mmd_wall();
  for(unsigned long spins = 1;; spins++) {
    const hipError_t e = hipStreamQuery(st);
    if(e != hipErrorNotReady) return e;
    (void)hipGetLastError();                   // ("not ready" is reported through the sticky error too)
    if((spins & 0xff) == 0 && mmd_wall() - t0 > spin_s) return hipStreamSynchronize(st);
  }
}
// (waits of the host-staged test transport — staging a message through host memory — are counted apart: RCCL has none of them)
static inline hipError_t mmd_stream_sync_transport(mmd_handle* h) { h->transport_syncs++; return hipStreamSynchronize(h->stream); }

static inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }
