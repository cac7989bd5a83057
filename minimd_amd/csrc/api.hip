// minimd_amd/csrc/api.hip — handle lifecycle, Force::compute dispatch, Integrate::run (ref/integrate.cpp:70-207),
// timers and profiling hooks of the C-ABI (include/mmd.h).
#include "device_utils.hpp"
#include "mmd_internal.hpp"

int mmd_temperature_async(mmd_handle* h, int slot);
int mmd_integrate_final_initial(mmd_handle* h);

extern "C" int mmd_float_size(void) { return (int)sizeof(mmd_float); }
extern "C" const char* mmd_variant_string(void) { return "miniMD-HIP 1.0 (MI355X gfx950, HIP+RCCL)"; }

extern "C" int mmd_create(int device, mmd_handle** out)
{
  if(!out) { mmd_set_error("mmd_create: out is NULL"); return -1; }
  *out = nullptr;
  if(device == -2) {            // host-only handle: box / Comm::setup / Neighbor::setup geometry without a GPU
    mmd_handle* g = new mmd_handle();
    g->host_only = true;
    g->device = -1;
    *out = g;
    return 0;
  }
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    mmd_set_error("mmd_create: no HIP device is visible — this library has no CPU fallback");
    return -2;
  }
  if(device < 0) {                // the rank's own device: local rank as the launcher names it (torchrun, mpirun, srun: launch.cpp)
    int lr = 0;
    if(mmd_launch_env(nullptr, nullptr, &lr, nullptr, nullptr, 0) < 0) lr = 0;
    device = lr % ndev;
  }
  if(device >= ndev) { mmd_set_error("mmd_create: device %d out of range (%d visible)", device, ndev); return -1; }
  HIP_TRY(hipSetDevice(device));
  mmd_handle* h = new mmd_handle();
  h->device = device;
  HIP_TRY(hipGetDeviceProperties(&h->prop, device));
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  {
    // the communication stream carries a step's halo (pack, RCCL p2p kernel, unpack: a few workgroups) while the compute stream floods the CUs with the
    // interior tiles: at the highest priority its workgroups are dispatched as soon as slots free up instead of queueing behind 30 k tiles
    int lo_p = 0, hi_p = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    HIP_TRY(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi_p));
  }
  HIP_TRY(hipHostMalloc((void**)&h->h_result, 32 * sizeof(double), hipHostMallocDefault));
  HIP_TRY(hipMalloc((void**)&h->d_result, 32 * sizeof(double)));
  HIP_TRY(hipHostMalloc((void**)&h->h_flags, 64 * sizeof(int), hipHostMallocDefault));
  { int khz = 0; if(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) == hipSuccess && khz > 0) h->clk_rate_hz = 1.0e3 * khz; (void)hipGetLastError(); }
  HIP_TRY(hipHostMalloc((void**)&h->h_flags_big, 64 * sizeof(int), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&h->dh.h_counts, 32 * 30 * sizeof(int), hipHostMallocDefault));
  memset(h->dh.h_counts, 0, 32 * 30 * sizeof(int));
  HIP_TRY(hipHostGetDevicePointer((void**)&h->dh.h_counts_dev, h->dh.h_counts, 0));
  HIP_TRY(hipMalloc((void**)&h->d_flags, 64 * sizeof(int)));
  HIP_TRY(hipMemset(h->d_result, 0, 32 * sizeof(double)));
  HIP_TRY(hipMemset(h->d_flags, 0, 64 * sizeof(int)));
  // (pinned memory comes back from the runtime's pool as the previous owner left it: the sequence word the host polls — h_flags[62],
  //  flags_publish_and_wait — must not start with somebody else's number)
  memset(h->h_flags, 0, 64 * sizeof(int)); memset(h->h_flags_big, 0, 64 * sizeof(int)); memset(h->h_result, 0, 32 * sizeof(double));
  *out = h;
  return 0;
}

extern "C" int mmd_device_count(void)
{
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return ndev;
}

extern "C" int mmd_destroy(mmd_handle* h)
{
  if(!h) return 0;
  if(h->host_only) { delete h; return 0; }
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  h->x.release(); h->x_alt.release(); h->v.release(); h->v_alt.release(); h->f.release(); h->x_stage.release();
  h->type.release(); h->type_alt.release(); h->tag.release(); h->tag_alt.release();
  h->bin_count.release(); h->bin_start.release(); h->bin_start_alt.release(); h->binned.release(); h->scan_tmp.release(); h->atom_bin.release(); h->atom_rank.release();
  h->neigh.release(); h->numneigh.release(); h->wave_max.release(); h->ghost_image.release(); h->ghost_root.release();
  h->tile_of_block.release(); h->pencil_range.release(); h->pencil_lohi.release(); h->tile_block.release(); h->tile_first.release(); h->tile_max.release();
  h->tile_cand.release(); h->tile_cand_src.release(); h->box_dev.release(); h->tile_ncand.release(); h->tile_cnt.release(); h->nl16.release();
  h->lj_tables.release(); h->rhor_spline.release(); h->frho_spline.release(); h->z2r_spline.release(); h->fp.release(); h->rho.release();
  for(auto& s : h->swaps) s.sendlist.release();
  h->buf_send.release(); h->buf_recv.release(); h->est.release(); h->ex_list.release(); h->flag_tmp.release(); h->bnd_list.release(); h->bstate.release(); h->brd_bits.release(); h->ghost_bits.release(); h->partials.release();
  for(auto& e : h->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if(h->ev_x_ready) { (void)hipEventDestroy(h->ev_x_ready); (void)hipEventDestroy(h->ev_halo_done); }
  for(int e = 0; e < 3; e++) if(h->ev_trial[e]) (void)hipEventDestroy(h->ev_trial[e]);
  h->tile_ghost.release(); h->tile_order.release(); h->tile_self.release(); h->tile_rowmax.release(); h->tile_rowsum.release(); h->tile_words.release(); h->tile_kcore.release(); h->xbuild.release(); h->core_words.release();
  if(h->h_result) (void)hipHostFree(h->h_result);
  if(h->d_result) (void)hipFree(h->d_result);
  if(h->h_flags) (void)hipHostFree(h->h_flags);
  if(h->h_flags_big) (void)hipHostFree(h->h_flags_big);
  if(h->dh.h_counts) (void)hipHostFree(h->dh.h_counts);
  h->dh.idx.release(); h->dh.counts.release(); h->dh.scratch.release(); h->dh.gmap.release(); h->fclk.release();
  if(h->d_flags) (void)hipFree(h->d_flags);
  if(h->stream) (void)hipStreamDestroy(h->stream);
  if(h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  delete h;
  return 0;
}

extern "C" int mmd_device_info(mmd_handle* h, char* name, int name_len, int* cu_count, double* hbm_gib)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(name && name_len > 0) { strncpy(name, h->prop.name, name_len - 1); name[name_len - 1] = 0; }
  if(cu_count) *cu_count = h->prop.multiProcessorCount;
  if(hbm_gib) *hbm_gib = h->prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0);
  return 0;
}

extern "C" int mmd_sync(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  HIP_TRY(mmd_stream_wait_polled(h->stream));      // (the caller's own wait: not one of the run's host synchronisations, mmd_run_stats)
  return 0;
}

extern "C" int mmd_set_option(mmd_handle* h, const char* name, int value)
{
  if(!h || !name) { mmd_set_error("mmd_set_option: bad arguments"); return -1; }
  else if(!strcmp(name, "tiles")) h->opt_tiles = value;
  else if(!strcmp(name, "eam_mlo")) h->opt_eam_mlo = value;
  else if(!strcmp(name, "ghost_resolve")) h->opt_ghost_resolve = value;
  else if(!strcmp(name, "time_force_sample")) h->opt_time_sample = value;
  else if(!strcmp(name, "fold_reverse")) h->opt_fold_reverse = value;
  else if(!strcmp(name, "lj_original")) h->opt_lj_original = value;
  else if(!strcmp(name, "core_pct")) h->opt_core_pct = value;
  else if(!strcmp(name, "borders_est")) h->opt_borders_est = value;
  else if(!strcmp(name, "spec")) h->opt_spec = value;
  else if(!strcmp(name, "direct_halo")) h->dh.opt = value;
  else if(!strcmp(name, "direct_borders")) h->dh.opt_borders = value;
  else if(!strcmp(name, "halo_recv")) { if(value != 1 && value != 3) { mmd_set_error("mmd_set_option: halo_recv is 1 or 3"); return -1; } h->dh.opt_recv = value; }
  else if(!strcmp(name, "exchange_cap")) h->opt_exchange_cap = value;
  else if(!strcmp(name, "force_transport")) h->opt_force_transport = value;
  else if(!strcmp(name, "ablate")) {
#ifdef MMD_PROFILE
    h->opt_ablate = value;
#else
    if(value) { mmd_set_error("mmd_set_option: 'ablate' needs a library built with -DMMD_PROFILE (tools/build_variant.sh)"); return -1; }
#endif
  }
  else if(!strcmp(name, "fuse")) h->opt_fuse = value;
  else if(!strcmp(name, "overlap")) h->opt_overlap = value;
  else if(!strcmp(name, "check_exchange")) h->opt_check_exchange = value;
  else if(!strcmp(name, "safe_exchange")) h->opt_safe_exchange = value;
  else if(!strcmp(name, "maxneighs")) h->maxneighs = (value + MMD_UNROLL - 1) / MMD_UNROLL * MMD_UNROLL;
  else { mmd_set_error("mmd_set_option: unknown option '%s'", name); return -1; }
  return 0;
}

// ---- event pairs around the force kernel (GPU time of the kernel itself, for the roofline) ------------
// kind 0: Force::compute (TIME_FORCE), kind 1: Comm::communicate / reverse_communicate (added to TIME_COMM like
// ref/integrate.cpp:103-105,190-195). Pairs are recorded on the stream the work is enqueued on (h->stream at that moment).
static int ev_begin(mmd_handle* h, int kind = 0)
{
  if(h->ev_used == h->ev_pool.size()) {
    EventPair p;
    HIP_TRY(hipEventCreate(&p.a));
    HIP_TRY(hipEventCreate(&p.b));
    h->ev_pool.push_back(p);
  }
  h->ev_pool[h->ev_used].kind = kind;
  HIP_TRY(hipEventRecord(h->ev_pool[h->ev_used].a, h->stream));
  return 0;
}
static int ev_end(mmd_handle* h)
{
  if(h->ev_used >= h->ev_pool.size()) { mmd_set_error("internal: event bracket closed without having been opened"); return -1; }
  HIP_TRY(hipEventRecord(h->ev_pool[h->ev_used].b, h->stream));
  h->ev_used++;
  return 0;
}
// the bracket around the two force launches of an overlapped step stays open while the step's halo (with its own pair) is enqueued: a pair of its own,
// read when the next one is about to be recorded (four steps later: long complete) or at the end of the run
static int ovf_harvest(mmd_handle* h)
{
  if(!h->ovf_open) return 0;
  HIP_TRY(hipEventSynchronize(h->ovf_b));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, h->ovf_a, h->ovf_b));
  h->force_ms_all += ms; h->force_launches_all++;
  h->ovf_open = false;
  return 0;
}
static int ovf_begin(mmd_handle* h)
{
  MMD_TRY(ovf_harvest(h));
  if(!h->ovf_a) { HIP_TRY(hipEventCreate(&h->ovf_a)); HIP_TRY(hipEventCreate(&h->ovf_b)); }
  HIP_TRY(hipEventRecord(h->ovf_a, h->stream));
  return 0;
}
static int ovf_end(mmd_handle* h)
{
  HIP_TRY(hipEventRecord(h->ovf_b, h->stream));
  h->ovf_open = true;
  return 0;
}
// kinds 2 / 3: the Comm (exchange + sort + borders) and Neighbor::build phases of a re-neighboring (TIME_COMM + TIME_TEST / TIME_NEIGH).
// sync = false collects only the pairs that have already completed (no host wait) and keeps the others in the pool.
static int ev_collect(mmd_handle* h, bool sync = true)
{
  if(sync) {
    HIP_TRY(mmd_stream_wait_polled(h->stream));
    if(h->comm_stream) HIP_TRY(mmd_stream_wait_polled(h->comm_stream));
  }
  size_t keep = 0;
  for(size_t i = 0; i < h->ev_used; i++) {
    if(!sync && hipEventQuery(h->ev_pool[i].b) != hipSuccess) {      // still in flight: keep the pair
      std::swap(h->ev_pool[keep], h->ev_pool[i]);
      keep++;
      continue;
    }
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b));
    switch(h->ev_pool[i].kind) {
      case 0: h->force_ms += ms; h->force_launches++; break;
      case 4: h->force_ms_all += ms; h->force_launches_all++; break;      // Force::compute of overlapped steps (its two launches bracketed; every 7th such step)
      case 5: h->halo_ms += ms; break;                                    // forward halo of a step (every 7th step: an event pair costs the stream two marker packets)
      case 1: h->comm_ms += ms; break;
      case 2: h->timer[1] += ms * 1e-3; h->timer[4] += ms * 1e-3; break;     // ref/integrate.cpp:155-166
      default: h->timer[3] += ms * 1e-3; break;
    }
  }
  h->ev_used = keep;
  (void)hipGetLastError();                     // (hipEventQuery reports "not ready" through the sticky error too)
  return 0;
}

// Force::compute (virtual dispatch of ref/force.h:57 -> ForceLJ / ForceEAM)
static int force_compute_async(mmd_handle* h, int evflag, double* eng, double* vir, bool timed)
{
  if(timed) {
    // the kernel clock is read on every `time_force_sample`-th call of a run (3 or 7: coprime with the re-neighboring and thermo
    // periods, so every kind of step is sampled in proportion); TIME_FORCE = mean of the timed calls x number of calls
    // (time_force_sample 0 = automatic = every 7th call, coprime with both periods too: a dispatch that carries the pair sits 6.6 us behind its
    //  predecessor and 4.5 us in front of its successor, which otherwise follow each other without a gap — its completion signal and time
    //  stamps; tools/steps_probe.sh. On every call that is 5 % of a step at -s 80 and 40 % at -s 32, on every 7th 0.7 % and 6 %)
    const int every = h->opt_time_sample > 0 ? h->opt_time_sample : 7;
    // (the sampling counter runs on ACROSS mmd_integrate_run calls: a run cut into 20-step slices has every position of the slice — the launch behind
    //  k_initial_integrate, the plain ones, the launch behind the neighbor build, which runs 5-20 % longer — sampled in proportion over time, instead of
    //  the same three positions in every slice)
    timed = every <= 1 || h->force_sample_ctr % every == 0 || (h->force_calls == 0 && h->run_ntimes < every);      // (a run shorter than the period still has its first launch clocked)
    h->force_sample_ctr++;
    h->force_calls++;
  }
  if(timed && h->style == 0 && !h->halfneigh && !h->halo_pending && mmd_lj_tiles_available(h)) {
    // LJ over full lists in tile form is ONE launch: the pair is attached to that dispatch instead of bracketing it
    if(h->ev_used == h->ev_pool.size()) {
      EventPair p;
      HIP_TRY(hipEventCreate(&p.a));
      HIP_TRY(hipEventCreate(&p.b));
      h->ev_pool.push_back(p);
    }
    h->ev_pool[h->ev_used].kind = 0;
    h->launch_ev_a = h->ev_pool[h->ev_used].a; h->launch_ev_b = h->ev_pool[h->ev_used].b;
    const int r = mmd_lj_compute(h, evflag, eng, vir);
    if(h->launch_ev_a == nullptr) h->ev_used++;              // (consumed by the launch)
    h->launch_ev_a = h->launch_ev_b = nullptr;
    return r;
  }
  if(timed) MMD_TRY(ev_begin(h));
  int r = h->style == 0 ? mmd_lj_compute(h, evflag, eng, vir) : mmd_eam_compute(h, evflag, eng, vir);
  if(timed) MMD_TRY(ev_end(h));
  return r;
}

extern "C" int mmd_force_compute(mmd_handle* h, int evflag, double* eng_vdwl, double* virial)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  double e = 0, v = 0;
  if(h->ghosts_stale) { MMD_TRY(mmd_ghosts_refresh(h)); h->ghosts_stale = false; }
  MMD_TRY(force_compute_async(h, evflag, &e, &v, false));
  HIP_TRY(mmd_stream_sync(h));
  if(evflag) { if(eng_vdwl) *eng_vdwl = e; if(virial) *virial = v; }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Integrate::run (ref/integrate.cpp:70-207)
// ---------------------------------------------------------------------------------------------------
extern "C" int mmd_integrate_run(mmd_handle* h, int first_step, int ntimes, int thermo_nstat, mmd_thermo_fn cb, void* ctx)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  for(int i = 0; i < 5; i++) h->timer[i] = 0;
  h->force_ms = 0; h->comm_ms = 0; h->force_launches = 0; h->force_calls = 0; h->ev_used = 0;
  h->force_ms_all = 0; h->force_launches_all = 0; h->halo_ms = 0;
  h->run_ntimes = ntimes;
  h->fclk_n = 0; h->fclk_ms = 0; h->fclk_launches = 0; h->fclk_harvested = true;
  if(h->clk_rate_hz > 0 && h->style == 0 && !h->halfneigh) {
    MMD_TRY(h->fclk.ensure((size_t)FCLK_STRIDE * FCLK_SLOTS, false, h->stream));
    HIP_TRY(hipMemsetAsync(h->fclk.p, 0, (size_t)FCLK_STRIDE * std::min(FCLK_SLOTS, ntimes + 2) * sizeof(unsigned long long), h->stream));
  }
  long long halo_calls = 0, halo_timed = 0, ovf_calls = 0;
  h->host_syncs = 0; h->halo_bytes = 0; h->transport_syncs = 0;
  // the step loop steers the kernels through transient flags of the handle; whatever way this function is left (an overflowing
  // build, a transport error), they are cleared, so a later call on the same handle never waits on a stale event or skips a halo
  struct TransientGuard {
    mmd_handle* h;
    ~TransientGuard() {
      h->fuse_now = 0; h->resolve_now = false; h->fold_reverse_now = false; h->core.mode_now = 0; h->zero_f_in_integrate = false;
      h->halo_pending = false; h->in_reneighbor = false; h->pbc_defer = false; h->launch_ev_a = h->launch_ev_b = nullptr; h->halo_in_x_allow = false;
      h->in_run = false; h->bin_owned_valid = false;
      h->spec_fn = nullptr; h->spec = SpecLaunch{nullptr, nullptr, nullptr, nullptr}; h->spec_done = false;
      h->ovf_open = false;
    }
  } transient_guard{h};
  h->in_run = true;
  // (a previous run that failed mid-step may have left the ghosts one step behind their owners)
  if(h->ghosts_stale) { MMD_TRY(mmd_ghosts_refresh(h)); h->ghosts_stale = false; }
  if(first_step == 0) MMD_TRY(mmd_run_reserve(h));      // (no allocation inside the first re-neighborings)
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double t_start = mmd_wall();
  // host-side phase clocks need the device drained at phase boundaries only when a phase is to be
  // attributed; kernels are enqueued asynchronously, so the COMM/NEIGH/FORCE buckets are sampled
  // with a stream sync on re-neighbor steps (where the host must read counts anyway) and on thermo steps.
  // the re-neighbor / sort schedule follows the GLOBAL step number (first_step + n + 1), so a run cut into slices of any
  // length re-neighbors on exactly the steps one uninterrupted run does (ref/integrate.cpp:84,101,159-160)
  if(first_step == 0 || h->next_sort < 0) h->next_sort = h->sort_every > 0 ? h->sort_every : 0x7fffffff;
  const bool reverse = h->halfneigh && h->ghost_newton;
  bool initial_done = false;        // initialIntegrate of this step already ran fused with the previous finalIntegrate
  // multi-rank (or forced-transport) runs with the LJ tile path overlap the forward halo with the interior tiles
  const bool multi = h->nprocs > 1 || h->opt_force_transport != 0;
  const bool overlap_auto = h->opt_overlap < 0 && multi;          // (-1: the form is chosen by measurement, collectively: mmd_internal.hpp)
  bool overlap = (h->opt_overlap > 0 && (multi || h->opt_overlap >= 2)) || (overlap_auto && h->overlap_choice == 1);     // (2: also on one rank — the ghost update under the interior tiles)
  // the trial of the automatic choice: B steps with, B steps without overlap behind a re-neighboring, no thermo step and no re-neighboring among them
  // (armed by a re-neighboring, kept in the handle: a caller that drives the run in slices ending on the re-neighboring step — bench.py's 20-step slices — still gets its trial,
  //  in the slice that follows; round-5 advisor)
  int trial_phase = 0, trial_left = 0;
  const int trial_B = std::min(8, (h->neigh_every - 2) / 2);
  bool halo_pending = false, collect_pending = false, ovf_timed_now = false, joined = false;
  int core_next = 0;                 // CoreRows: what the next force call may assume about the displacement since the build
  // the per-step halos are timed (into TIME_COMM) only where they are more than one tiny kernel: an event pair costs the stream
  // two markers, ~5 us per step that a -s 32 run would notice
  // one rank, half lists with ghost newton in tile form: a ghost's share of a pair goes straight to its owner
  const bool fold = reverse && h->opt_fold_reverse && h->nprocs == 1 && !h->opt_force_transport && h->opt_fuse && h->style == 0;
  const bool time_halo = (h->nprocs > 1 || h->opt_force_transport || (reverse && !fold));
  int evflag_pending = 0;
  const bool fuse_force = h->opt_fuse >= 2 && !reverse && !h->halfneigh;
  // one rank, LJ over full lists in tile form: no per-step ghost update at all (the tile kernel resolves ghosts itself)
  const bool resolve = h->opt_ghost_resolve && !overlap && h->ghost_chain_ok && h->opt_fuse && !h->opt_force_transport &&
                       h->style == 0 && !h->halfneigh && h->nprocs == 1;
  // EAM over full lists on one rank: the same, where the build left the ghosts named by owner + image code (both sweeps stage them from their
  // owners, the force sweep reads their fp through the owners: no Comm::communicate, no ForceEAM::communicate launch on such a step)
  const bool resolve_eam = h->opt_ghost_resolve && !overlap && h->opt_fuse && !h->opt_force_transport && h->style == 1 && !h->halfneigh && h->nprocs == 1;
  // LJ over half lists on one rank: the same; with ghost newton only where the reverse communication is folded into the kernel
  const bool resolve_half = h->opt_ghost_resolve && !overlap && h->opt_fuse && !h->opt_force_transport && h->style == 0 && h->halfneigh && h->nprocs == 1 &&
                            !h->opt_lj_original && (!h->ghost_newton || fold);
  bool fused_force = false;          // this step's force launch carries finalIntegrate + the next initialIntegrate
  bool final_fused = false;          // this (last) step's force launch carries finalIntegrate
  if((overlap || overlap_auto) && !h->ev_x_ready) {
    HIP_TRY(hipEventCreateWithFlags(&h->ev_x_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_halo_done, hipEventDisableTiming));
  }
  if(h->opt_check_exchange && h->xold_n != h->nlocal) MMD_TRY(mmd_integrate_mark_positions(h));
  bool folded = false;
  // Force::compute of step n (not overlapped with a halo): which kernel form, which transient switches — one place, because on a
  // re-neighboring step the neighbor build may issue this launch itself, behind its own kernels (opt_spec, mmd_internal.hpp)
  auto launch_force = [&](int n, int evflag) -> int {
    fused_force = fuse_force && !evflag && n + 1 < ntimes && (h->style == 0 ? mmd_lj_can_fuse_integrate(h) : mmd_eam_can_fuse_integrate(h));
    // the last step of a run (LJ tile kernel): finalIntegrate inside the force launch — no k_final_integrate pass over f and v behind it
    final_fused = fuse_force && !evflag && n + 1 == ntimes && h->style == 0 && mmd_lj_can_fuse_integrate(h);
    if(fused_force) MMD_TRY(mmd_prepare_x_alt(h));
    h->fuse_now = fused_force ? 1 : (final_fused ? 2 : 0);
    h->resolve_now = h->ghosts_stale;
    h->fold_reverse_now = folded = fold && h->ghost_chain_ok && mmd_lj_half_tiles_available(h);
    h->core.mode_now = core_next;                // rows in two parts (CoreRows): which part this call may walk
    h->core.tracked_last = false;
    const int rc = force_compute_async(h, evflag, nullptr, nullptr, true);
    h->fuse_now = 0;
    h->resolve_now = false;
    h->fold_reverse_now = false;
    h->core.mode_now = 0;
    core_next = h->core.tracked_last ? 2 : 0;
    return rc;
  };
  // the launch behind the build: LJ over full lists in tile form on one rank, no halo overlap, lists of the previous build to size it from
  // (several ranks: the same, when the step's halo is not overlapped — the ghosts Comm::borders has just made are what this launch reads, the verdict covers the
  //  max-reduced overflow flag of the direct borders; the direct-halo plan is finished behind it, when the words have arrived)
  const bool spec_static = h->opt_spec > 0 && h->style == 0 && !h->halfneigh &&
                           h->opt_tiles && !h->opt_check_exchange && h->lj_uniform;
  for(int n = 0; n < ntimes; n++) {
    if(overlap_auto && h->overlap_choice < 0) {
      if(trial_phase == 0 && h->trial_armed) {
        h->trial_armed = false;
        // (nothing rank-local may enter this condition: the trial ends in a collective, every rank has to run it on the same steps — a rank whose lists are
        //  not in tile form simply does not overlap during its trial steps)
        bool ok = trial_B >= 2 && n + 2 * trial_B <= ntimes;
        for(int k = 0; k < 2 * trial_B && ok; k++) {
          const int sk = first_step + n + 1 + k;
          if(sk % h->neigh_every == 0 || (thermo_nstat > 0 && sk % thermo_nstat == 0)) ok = false;
        }
        if(ok) {
          for(int e = 0; e < 3; e++) if(!h->ev_trial[e]) HIP_TRY(hipEventCreate(&h->ev_trial[e]));
          HIP_TRY(hipEventRecord(h->ev_trial[0], h->stream));
          trial_phase = 1; trial_left = trial_B;
        }
      }
      overlap = trial_phase == 1;
    }
    if(!initial_done) MMD_TRY(mmd_integrate_initial(h));
    initial_done = false;
    if((first_step + n + 1) % h->neigh_every) {
      const int step_now = first_step + n + 1;
      const int ev_now = thermo_nstat > 0 && (step_now % thermo_nstat == 0);
      if(overlap && (h->style == 0 ? (mmd_lj_tiles_available(h) || mmd_lj_half_tiles_available(h)) : mmd_eam_can_fuse_integrate(h))) {
        // halo of this step on the communication stream, interior tiles (no ghost among their candidates)
        // concurrently on the compute stream
        // (the clocks of such a step — halo on its stream, the two force launches on theirs — are read on every 7th one: each event pair costs its stream
        //  two marker packets, ~5 us of idle GPU apiece, and a step has two pairs)
        const bool timed_step = time_halo && (halo_calls % 7 == 0);
        halo_calls++;
        if(timed_step) halo_timed++;
        HIP_TRY(hipEventRecord(h->ev_x_ready, h->stream));
        HIP_TRY(hipStreamWaitEvent(h->comm_stream, h->ev_x_ready, 0));
        const bool lj_full = h->style == 0 && !h->halfneigh;
        ovf_timed_now = false;
        if(lj_full) {
          // the interior tiles go onto the compute stream FIRST: enqueuing the halo (an ncclGroup costs the host ~15 us) must not hold them up
          ovf_calls++;
          ovf_timed_now = timed_step;
          if(ovf_timed_now) MMD_TRY(ovf_begin(h));
          fused_force = fuse_force && !ev_now && n + 1 < ntimes && mmd_lj_can_fuse_integrate(h);
          if(fused_force) MMD_TRY(mmd_prepare_x_alt(h));
          h->fuse_now = fused_force;
          const int rc0 = mmd_lj_compute_tiles_split(h, ev_now, 0);
          h->fuse_now = 0;
          MMD_TRY(rc0);
        }
        std::swap(h->stream, h->comm_stream);
        int rc = timed_step ? ev_begin(h, 5) : 0;
        // (round 5) the boundary tiles go onto the communication stream, right behind the transfer: they start the moment the ghosts are there, under the tail of the
        // interior tiles, and read the received records where they landed (halo_recv 3: no unpack kernel either); the compute stream joins at the end of the step.
        // A thermo step keeps the round-4 form (its energy sum needs both launches finished).
        joined = lj_full && !ev_now;
        if(joined) h->halo_in_x_allow = h->opt_fuse && h->opt_ghost_resolve;
        if(rc >= 0) rc = mmd_comm_communicate(h);
        h->halo_in_x_allow = false;
        if(rc >= 0 && timed_step) rc = ev_end(h);
        if(rc >= 0 && joined) {
          h->fuse_now = fused_force;
          h->resolve_now = h->dh.x_unpack_pending;
          rc = mmd_lj_compute_tiles_split(h, 0, 1);
          h->fuse_now = 0; h->resolve_now = false;
          if(h->dh.x_unpack_pending) h->ghosts_stale = true;
          h->overlap_join_steps++;
        }
        std::swap(h->stream, h->comm_stream);
        MMD_TRY(rc);
        HIP_TRY(hipEventRecord(h->ev_halo_done, h->comm_stream));
        if(lj_full) {
          halo_pending = true;
          evflag_pending = ev_now;
        } else
          h->halo_pending = true;                // ForceEAM::compute / the half-list LJ dispatch split their launches themselves
      } else if(resolve && mmd_lj_tiles_available(h) && (h->opt_ghost_resolve >= 2 || h->cand_src_ready || h->ntiles <= 8192)) {
        h->ghosts_stale = true;                  // this step's force kernel reads the ghosts through their owners (tile_lds.hpp)
      } else if(resolve_eam && h->ghost_chain_ok && h->cand_src_ready && mmd_eam_can_fuse_integrate(h)) {
        h->ghosts_stale = true;                  // (both EAM sweeps stage the ghosts from their owners)
      } else if(resolve_half && h->ghost_chain_ok && h->cand_src_ready && mmd_lj_half_tiles_available(h)) {
        h->ghosts_stale = true;                  // (the half-list tile kernel stages the ghosts from their owners; their shares go to the owners)
      } else {
        const bool timed_step = time_halo && (halo_calls % 7 == 0);
        halo_calls++;
        if(timed_step) { halo_timed++; MMD_TRY(ev_begin(h, 5)); }
        // several ranks, LJ over full lists in tile form: the partners' messages may stay where they land — behind the ghost slots of the position buffer —
        // and this step's force launch stages the ghosts from there (halo_recv 3: no k_dh_unpack between the transfer and the force kernel)
        h->halo_in_x_allow = h->style == 0 && !h->halfneigh && h->opt_fuse && h->opt_ghost_resolve && mmd_lj_tiles_available(h);
        const int rch = mmd_comm_communicate(h);
        h->halo_in_x_allow = false;
        MMD_TRY(rch);
        if(h->dh.x_unpack_pending) h->ghosts_stale = true;
        if(timed_step) MMD_TRY(ev_end(h));
      }
    } else {
      h->ghosts_stale = false;                   // (borders rebuilds every ghost)
      h->trial_armed = true;
      const bool had_tiles = h->tiles_ready && h->neigh_nlocal == h->nlocal && h->nlocal > 0;
      if(h->opt_check_exchange && h->xold_n == h->nlocal) {    // ref/integrate.cpp:112-151 (warning text as there)
        double d_max = 0;
        MMD_TRY(mmd_integrate_max_move(h, &d_max));
        const double sx = h->hi[0] - h->lo[0], sy = h->hi[1] - h->lo[1], sz = h->hi[2] - h->lo[2];
        if(d_max > sx || d_max > sy || d_max > sz)
          printf("Warning: Atoms move further than your subdomain size, which will eventually cause lost atoms.\n"
                 "Increase reneighboring frequency or choose a different processor grid\n"
                 "Maximum move distance: %lf; Subdomain dimensions: %lf %lf %lf\n", d_max, sx, sy, sz);
      }
      // phase clocks are event pairs on the stream (no host synchronisation just to read a clock); the pairs that have
      // completed are folded into the timers after the build, whose own count read-back has drained the stream anyway
      h->in_reneighbor = true;
      const bool sort_now = first_step + n + 1 >= h->next_sort;
      // phase clocks: one rank with Atom::sort in the window and the production build — the first kernel of each phase (k_bin_count of the sort,
      // k_bin_count of the build) and k_tile_reduce stamp the device's wall clock into the build's result words; otherwise event pairs on the stream
      // (each record costs the stream a marker packet, ~5 us of idle GPU)
      const bool dev_clock = h->clk_rate_hz > 0 && h->nprocs == 1 && !h->opt_force_transport && sort_now &&
                             h->nlocal > 0 && h->neigh_ready && h->opt_tiles && h->tiles_ready && !h->opt_check_exchange;
      h->clk_written = 0;
      h->clk_slot = dev_clock ? 0 : -1;
      int rc = dev_clock ? 0 : ev_begin(h, 2);
      // one rank: Comm::exchange is Atom::pbc alone; when Atom::sort follows, its binning pass wraps the atoms on the way
      h->pbc_defer = sort_now && h->nprocs == 1 && h->nlocal > 0 && h->neigh_ready;
      if(rc >= 0) rc = mmd_comm_exchange(h);
      h->pbc_defer = false;
      if(rc >= 0 && sort_now) { rc = mmd_atom_sort(h); h->next_sort += h->sort_every; }
      if(h->pbc_pending) { mmd_set_error("Integrate::run: deferred Atom::pbc was not applied"); rc = -1; h->pbc_pending = false; }
      if(rc >= 0) rc = mmd_comm_borders(h);
      h->in_reneighbor = false;
      MMD_TRY(rc);
      if(h->opt_check_exchange) MMD_TRY(mmd_integrate_mark_positions(h));
      if(!dev_clock) { MMD_TRY(ev_end(h)); MMD_TRY(ev_begin(h, 3)); }
      else h->clk_slot = 1;
      core_next = 1;                               // the atoms are where the build saw them
      h->spec_done = false;
      {
        const int ev_rb = thermo_nstat > 0 && ((first_step + n + 1) % thermo_nstat == 0);
        // (only with the device-side phase clocks: an event bracket around the build is open here, and the launch would attach its own pair inside it)
        // (several ranks: the phase clocks are event pairs — the Neighbor::build bracket is closed in front of the launch, so that it does not take the force kernel in)
        bool neigh_bracket_closed = false;
        if(spec_static && !overlap && had_tiles && !ev_rb && dev_clock) h->spec_fn = [&launch_force, n]() { return launch_force(n, 0); };
        else if(spec_static && !overlap && had_tiles && !ev_rb && multi && (h->opt_overlap >= 0 || h->overlap_choice == 0))
          h->spec_fn = [&launch_force, &neigh_bracket_closed, h, n]() { if(!neigh_bracket_closed) { MMD_TRY(ev_end(h)); neigh_bracket_closed = true; } return launch_force(n, 0); };
        const int rcb = mmd_neighbor_build(h);
        h->spec_fn = nullptr;
        MMD_TRY(rcb);
        if(!dev_clock && !neigh_bracket_closed) MMD_TRY(ev_end(h));
      }
      h->clk_slot = -1;
      if(!dev_clock) {}
      else if(h->clk_written == 7) {
        // (a build that went through a fall-back or ran twice leaves stamps that do not line up: that re-neighboring is not clocked)
        long long c[3];
        memcpy(c, h->h_flags + 56, sizeof(c));
        if(c[0] > h->clk_last && c[1] >= c[0] && c[2] >= c[1]) {
          const double t_comm = (double)(c[1] - c[0]) / h->clk_rate_hz, t_neigh = (double)(c[2] - c[1]) / h->clk_rate_hz;
          h->timer[1] += t_comm; h->timer[4] += t_comm; h->timer[3] += t_neigh;       // ref/integrate.cpp:155-166
          h->clk_last = c[2];
        }
      }
      collect_pending = true;                  // (folded into the timers once this step's force kernel is in flight)
    }
    const int step = first_step + n + 1;
    const int evflag = thermo_nstat > 0 && (step % thermo_nstat == 0);
    folded = false;
    if(halo_pending) {
      // overlapped step: interior tiles ran under the halo; now wait for the ghosts and finish the boundary tiles
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_halo_done, 0));
      if(!joined) {
        h->fuse_now = fused_force;
        const int rc1 = mmd_lj_compute_tiles_split(h, evflag_pending, 1);
        h->fuse_now = 0;
        MMD_TRY(rc1);
      }
      joined = false;
      if(ovf_timed_now) MMD_TRY(ovf_end(h));
      ovf_timed_now = false;
      halo_pending = false;
    } else if(h->spec_done) {
      h->spec_done = false;                        // (the neighbor build issued this step's launch behind its own kernels, and its verdict let it run)
    } else
      MMD_TRY(launch_force(n, evflag));
    if(reverse && !folded) {
      if(time_halo) MMD_TRY(ev_begin(h, 1));
      MMD_TRY(mmd_comm_reverse_communicate(h));
      if(time_halo) MMD_TRY(ev_end(h));
    }
    if(fused_force) {
      std::swap(h->x, h->x_alt);                 // the tile kernel wrote v and the next positions of every owned atom
      initial_done = true;
      fused_force = false;
    } else if(h->opt_fuse && !evflag && n + 1 < ntimes) {
      h->zero_f_in_integrate = folded;           // (the next step's half-list force call starts from zeros: no separate fill)
      const int rci = mmd_integrate_final_initial(h);
      h->zero_f_in_integrate = false;
      MMD_TRY(rci);
      initial_done = true;
    } else if(final_fused) final_fused = false;      // (the launch did it)
    else MMD_TRY(mmd_integrate_final(h));
    if(collect_pending) { MMD_TRY(ev_collect(h, false)); collect_pending = false; }      // host work under the force kernel
    if(trial_phase == 1 && --trial_left == 0) { HIP_TRY(hipEventRecord(h->ev_trial[1], h->stream)); trial_phase = 2; trial_left = trial_B; }
    else if(trial_phase == 2 && --trial_left == 0) {
      // both forms have run: the sums over the ranks decide, the same way on every rank (one host synchronisation, once per handle)
      HIP_TRY(hipEventRecord(h->ev_trial[2], h->stream));
      HIP_TRY(hipEventSynchronize(h->ev_trial[2]));
      h->host_syncs++;
      float ms_on = 0, ms_off = 0;
      HIP_TRY(hipEventElapsedTime(&ms_on, h->ev_trial[0], h->ev_trial[1]));
      HIP_TRY(hipEventElapsedTime(&ms_off, h->ev_trial[1], h->ev_trial[2]));
      double t[2] = {ms_off * 1e-3 / trial_B, ms_on * 1e-3 / trial_B};
      MMD_TRY(mmd_transport_allreduce(h, t, 2));
      h->overlap_trial_s[0] = t[0]; h->overlap_trial_s[1] = t[1];
      h->overlap_choice = t[1] < t[0] ? 1 : 0;
      overlap = h->overlap_choice == 1;
      trial_phase = 3;
    }
    if(evflag) {
      MMD_TRY(mmd_temperature_async(h, 2));
      HIP_TRY(hipMemcpyAsync(h->h_result, h->d_result, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(mmd_stream_sync(h));
      double vals[3] = {h->h_result[2], h->h_result[0], h->h_result[1]};     // mv2, eng, virial
      MMD_TRY(mmd_transport_allreduce(h, vals, 3));
      if(cb) cb(ctx, step, vals[0], vals[1], vals[2]);
    }
  }
  if(h->ghosts_stale) { MMD_TRY(mmd_ghosts_refresh(h)); h->ghosts_stale = false; }     // leave x consistent for the caller
  MMD_TRY(ev_collect(h));
  MMD_TRY(ovf_harvest(h));
  h->timer[0] = mmd_wall() - t_start;
  h->fclk_harvested = false;            // (the stamps stay on the device until somebody asks: mmd_get_counter)
  // TIME_FORCE: GPU time between the events around Force::compute (scaled from the sampled calls to all of them)
  // force_calls / force_launches count the SAMPLED path only (force_compute_async); calls that are bracketed every time add their time as it is
  h->timer[2] = h->force_ms * 1e-3 * (h->force_launches > 0 && h->force_calls > h->force_launches ? (double)h->force_calls / h->force_launches : 1.0) +
                h->force_ms_all * 1e-3 * (h->force_launches_all > 0 && ovf_calls > h->force_launches_all ? (double)ovf_calls / h->force_launches_all : 1.0);
  // TIME_COMM also counts the per-step halos (GPU time between their events; the forward halos scaled from the sampled steps to all of them)
  h->timer[1] += h->comm_ms * 1e-3 + h->halo_ms * 1e-3 * (halo_timed > 0 ? (double)halo_calls / (double)halo_timed : 1.0);
  // The reference's buckets are host wall-clock intervals that follow each other: they PARTITION t_total, and its t_other = t_total - t_force - t_neigh - t_comm
  // (ref/integrate.cpp:100-107, 155-207; ref/ljs.cpp:485-495) is never negative. Here FORCE / NEIGH / COMM are GPU times between events, and on several ranks they
  // can overlap each other: the halo of an overlapped step runs on the communication stream UNDER the interior tiles; sampled pairs are scaled to all calls; ranks that share a
  // GPU over the debug transport see each other's kernels inside their own pairs. The raw sums are kept (timer_raw, counters "timer_raw_*_us"); what is reported is their
  // exclusive share of the wall clock: time counted twice is taken off COMM first (communication hidden under Force::compute is by definition not on the rank's critical
  // path; what stays is the EXPOSED part), then — only if the compute buckets alone exceed the wall clock (shared GPU) — off FORCE and NEIGH in proportion.
  for(int i = 0; i < 5; i++) h->timer_raw[i] = h->timer[i];
  {
    double& tot = h->timer[0]; double& comm = h->timer[1]; double& force = h->timer[2]; double& neigh = h->timer[3]; double& extra = h->timer[4];
    double excess = comm + force + neigh - tot;
    if(excess > 0) {
      const double take = std::min(excess, comm);
      comm -= take; excess -= take;
      if(excess > 0 && force + neigh > 0) { const double sc = (tot - comm) / (force + neigh); force *= sc; neigh *= sc; }
      if(extra > comm) extra = comm;             // (TIME_TEST is a part of TIME_COMM, ref/integrate.cpp:155-166)
    }
  }
  return 0;
}

extern "C" int mmd_timers(mmd_handle* h, double out5[5], double* force_kernel_ms, int* force_kernel_launches)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(out5) for(int i = 0; i < 5; i++) out5[i] = h->timer[i];
  if(force_kernel_ms) *force_kernel_ms = h->force_ms + h->force_ms_all;
  if(force_kernel_launches) *force_kernel_launches = h->force_launches + h->force_launches_all;
  return 0;
}

extern "C" int mmd_run_stats(mmd_handle* h, long long* host_syncs, long long* bytes_sent, long long* transport_syncs)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(host_syncs) *host_syncs = h->host_syncs;
  if(bytes_sent) *bytes_sent = h->halo_bytes;
  if(transport_syncs) *transport_syncs = h->transport_syncs;
  return 0;
}

// device-clock stamps of the last run's LJ full-list tile launches -> fclk_ms / fclk_launches (on demand: not inside anybody's timed region)
static int fclk_harvest(mmd_handle* h)
{
  if(h->fclk_harvested) return 0;
  h->fclk_harvested = true;
  h->fclk_ms = 0; h->fclk_launches = 0; h->fclk_ms_sampled = 0; h->fclk_launches_sampled = 0; h->fclk_gap_ms = 0; h->fclk_gaps = 0;
  h->fclk_first_ms = h->fclk_median_ms = h->fclk_last_ms = 0;
  if(h->fclk_n > 0 && h->fclk.p) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    static thread_local std::vector<unsigned long long> hc;
    const int nkept = std::min(h->fclk_n, FCLK_SLOTS);
    hc.resize((size_t)FCLK_STRIDE * nkept);
    HIP_TRY(hipMemcpy(hc.data(), h->fclk.p, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // launch numbers whose records survive, in launch order: all of a short run; the first half + the last half (ring) of a long one
    std::vector<long long> kept;
    if(h->fclk_n <= FCLK_SLOTS) for(int k = 0; k < h->fclk_n; k++) kept.push_back(k);
    else { for(int k = 0; k < FCLK_SLOTS / 2; k++) kept.push_back(k); for(long long k = (long long)h->fclk_n - FCLK_SLOTS / 2; k < h->fclk_n; k++) kept.push_back(k); }
    std::vector<double> spans;
    unsigned long long prev_end = 0;
    long long prev_k = -2;
    for(long long k : kept) {
      const int slot = fclk_slot(k);
      const unsigned long long* c = hc.data() + (size_t)FCLK_STRIDE * slot;
      unsigned long long e = 0;
      for(int q = 0; q < FCLK_TAIL; q++) e = std::max(e, c[8 + q]);
      // two launches that follow each other directly (plain steps on one rank): the time between the last workgroup of one and the first of the next is
      // the completion of the first + the dispatch of the second — what a profiler's per-kernel duration contains beyond the span
      if(k == prev_k + 1 && prev_end != 0 && c[0] > prev_end && (double)(c[0] - prev_end) / h->clk_rate_hz < 30.0e-6) { h->fclk_gap_ms += (double)(c[0] - prev_end) / h->clk_rate_hz * 1e3; h->fclk_gaps++; }
      prev_end = e; prev_k = k;
      if(c[0] != 0 && e > c[0]) {
        const double ms = (double)(e - c[0]) / h->clk_rate_hz * 1e3;
        h->fclk_ms += ms; h->fclk_launches++;
        spans.push_back(ms);
        if(h->fclk_sampled[slot]) { h->fclk_ms_sampled += ms; h->fclk_launches_sampled++; }
      }
    }
    if(!spans.empty()) {
      const size_t m = std::min<size_t>(100, spans.size());
      for(size_t i = 0; i < m; i++) { h->fclk_first_ms += spans[i] / m; h->fclk_last_ms += spans[spans.size() - 1 - i] / m; }
      std::vector<double> sorted = spans;
      std::sort(sorted.begin(), sorted.end());
      h->fclk_median_ms = sorted[sorted.size() / 2];
    }
  }
  return 0;
}

extern "C" int mmd_get_counter(mmd_handle* h, const char* name, long long* value)
{
  if(!h || !name || !value) { mmd_set_error("mmd_get_counter: bad arguments"); return -1; }
  if(!strcmp(name, "exchange_overflows")) *value = h->ex_overflows;
  else if(!strcmp(name, "exchange_fast")) *value = h->ex_fast;
  else if(!strcmp(name, "borders_fast")) *value = h->borders_fast_runs;
  else if(!strcmp(name, "borders_general")) *value = h->borders_general_runs;
  else if(!strcmp(name, "borders_direct")) *value = h->borders_direct_runs;
  else if(!strcmp(name, "force_clock_ns")) { MMD_TRY(fclk_harvest(h)); *value = (long long)(h->fclk_ms * 1e6); }       // last run: device-clock time of ALL its LJ full-list tile launches ...
  else if(!strcmp(name, "force_clock_gap_ns")) { MMD_TRY(fclk_harvest(h)); *value = (long long)(h->fclk_gap_ms * 1e6); }       // ... idle time between launches that follow each other directly
  else if(!strcmp(name, "force_clock_gaps")) { MMD_TRY(fclk_harvest(h)); *value = h->fclk_gaps; }
  else if(!strcmp(name, "force_clock_sampled_ns")) { MMD_TRY(fclk_harvest(h)); *value = (long long)(h->fclk_ms_sampled * 1e6); }       // ... the same over the launches that also carried an event pair
  else if(!strcmp(name, "force_clock_sampled_launches")) { MMD_TRY(fclk_harvest(h)); *value = h->fclk_launches_sampled; }
  else if(!strcmp(name, "force_clock_first_ns")) { MMD_TRY(fclk_harvest(h)); *value = (long long)(h->fclk_first_ms * 1e6); }       // ... mean span of the run's first (up to) 100 launches,
  else if(!strcmp(name, "force_clock_median_ns")) { MMD_TRY(fclk_harvest(h)); *value = (long long)(h->fclk_median_ms * 1e6); }     // median of the kept ones (a long run keeps its first and last 128),
  else if(!strcmp(name, "force_clock_last_ns")) { MMD_TRY(fclk_harvest(h)); *value = (long long)(h->fclk_last_ms * 1e6); }         // mean of its last (up to) 100
  else if(!strcmp(name, "force_clock_launches")) { MMD_TRY(fclk_harvest(h)); *value = h->fclk_launches; }            // ... and how many there were
  else if(!strcmp(name, "overlap_choice")) *value = h->overlap_choice;        // halo overlap chosen by measurement: -1 undecided, 0 without, 1 with (option overlap = -1)
  else if(!strcmp(name, "overlap_trial_off_ns")) *value = (long long)(h->overlap_trial_s[0] * 1e9);      // per step, summed over the ranks
  else if(!strcmp(name, "overlap_trial_on_ns")) *value = (long long)(h->overlap_trial_s[1] * 1e9);
  else if(!strcmp(name, "timer_raw_comm_us")) *value = (long long)(h->timer_raw[1] * 1e6);      // last run: GPU time of the buckets BEFORE they were made to partition the wall clock
  else if(!strcmp(name, "timer_raw_force_us")) *value = (long long)(h->timer_raw[2] * 1e6);
  else if(!strcmp(name, "timer_raw_neigh_us")) *value = (long long)(h->timer_raw[3] * 1e6);
  else if(!strcmp(name, "rccl_check_partners")) *value = h->rccl_check_partners;      // RCCL bring-up self-check: partners a verified pattern was exchanged with ...
  else if(!strcmp(name, "rccl_check_us")) *value = (long long)(h->rccl_check_s * 1e6);      // ... and what it took
  else if(!strcmp(name, "dh_total_recv")) *value = h->dh.total_recv;
  else if(!strcmp(name, "dh_R")) *value = h->dh.R;
  else if(!strcmp(name, "dh_gmap_live")) *value = h->dh.gmap_live ? 1 : 0;
  else if(!strcmp(name, "dh_ready")) *value = h->dh.ready ? 1 : 0;
  else if(!strcmp(name, "cand_src_halo")) *value = (h->cand_src_ready ? 1 : 0) + (h->cand_src_halo ? 2 : 0);
  else if(!strcmp(name, "overlap_join_steps")) *value = h->overlap_join_steps;      // overlapped steps whose boundary tiles ran on the communication stream behind the transfer
  else if(!strcmp(name, "halo_in_x_steps")) *value = h->halo_in_x_steps;      // steps whose position halo stayed behind the ghost slots (no k_dh_unpack)
  else if(!strcmp(name, "device_bins_coarser")) *value = h->neigh_ready && (h->bg.nbin[0] != h->bg_ref.nbin[0] || h->bg.nbin[1] != h->bg_ref.nbin[1] || h->bg.nbin[2] != h->bg_ref.nbin[2]) ? 1 : 0;
  else if(!strcmp(name, "tiles_ready")) *value = h->tiles_ready ? 1 : 0;
  else if(!strcmp(name, "eam_lds_density")) *value = h->eam_diag[0];
  else if(!strcmp(name, "eam_lds_force")) *value = h->eam_diag[1];
  else if(!strcmp(name, "eam_wg_density")) *value = h->eam_diag[2];
  else if(!strcmp(name, "eam_wg_force")) *value = h->eam_diag[3];
  else if(!strcmp(name, "tile_cmax")) *value = h->tile_cmax;
  else if(!strcmp(name, "rows_uploaded")) *value = h->rows_uploaded ? 1 : 0;
  else if(!strcmp(name, "bin_reuses")) *value = h->bin_reuses;        // build binnings that placed the ghosts only (owned atoms taken from the binning of Atom::sort)
  else if(!strcmp(name, "spec_runs")) *value = h->spec_runs;          // Force::compute launches issued behind a neighbor build ...
  else if(!strcmp(name, "spec_fails")) *value = h->spec_fails;        // ... and how many of them the build's verdict turned into no-ops
  else { mmd_set_error("mmd_get_counter: unknown counter '%s'", name); return -1; }
  return 0;
}

extern "C" int mmd_profile_kernel(mmd_handle* h, int which, int nrep, double* avg_ms)
{
  if(!h || nrep < 1 || !avg_ms) { mmd_set_error("mmd_profile_kernel: bad arguments"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  DevArr<real> vsave;
  DevArr<real4> xsave;
  if(which == 2 || which == 3) {      // integrators mutate state: work on a saved copy and restore
    MMD_TRY(vsave.ensure((size_t)3 * h->nlocal + 1, false, h->stream));
    MMD_TRY(xsave.ensure((size_t)h->nlocal + 1, false, h->stream));
    HIP_TRY(hipMemcpyAsync(vsave.p, h->v.p, (size_t)3 * h->nlocal * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(xsave.p, h->x.p, (size_t)h->nlocal * sizeof(real4), hipMemcpyDeviceToDevice, h->stream));
  }
  // one untimed launch first
  auto once = [&]() -> int {
    switch(which) {
      case 0: return force_compute_async(h, 0, nullptr, nullptr, false);
      case 1: return mmd_neighbor_build(h);
      case 2: return mmd_integrate_initial(h);
      case 3: return mmd_integrate_final(h);
      case 4: return mmd_comm_communicate(h);
      default: mmd_set_error("mmd_profile_kernel: unknown kernel id %d", which); return -1;
    }
  };
  MMD_TRY(once());
  HIP_TRY(mmd_stream_sync(h));
  HIP_TRY(hipEventRecord(a, h->stream));
  for(int r = 0; r < nrep; r++) MMD_TRY(once());
  HIP_TRY(hipEventRecord(b, h->stream));
  HIP_TRY(hipEventSynchronize(b));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, a, b));
  *avg_ms = ms / nrep;
  // half lists with ghost newton: Force::compute alone leaves the ghosts' shares on the ghosts; the step loop expects f AFTER
  // Comm::reverse_communicate (it continues with initialIntegrate). One rank: complete it here, so that profiling between two slices
  // of a run leaves the run untouched (up to the order of the sums). Several ranks: the caller has to (it is a collective).
  if(which == 0 && h->halfneigh && h->ghost_newton && h->nprocs == 1 && !h->opt_force_transport) MMD_TRY(mmd_comm_reverse_communicate(h));
  if(which == 2 || which == 3) {
    HIP_TRY(hipMemcpyAsync(h->v.p, vsave.p, (size_t)3 * h->nlocal * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->x.p, xsave.p, (size_t)h->nlocal * sizeof(real4), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    vsave.release(); xsave.release();
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return 0;
}
