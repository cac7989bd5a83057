// minimd_amd/csrc/sim.cpp — whole-program twin of ref/ljs.cpp main(): same command line, same input deck,
// same stdout grammar (banner, "# Timestep T U P Time" rows, PERF_SUMMARY), driving the device handle
// through the C-ABI of include/mmd.h. One process per GPU; rank and size come from the launcher's environment (torchrun, mpirun / mpiexec of
// Open MPI or MPICH, srun: launch.cpp), the ranks meet over TCP and talk RCCL — or, when they have to share GPUs, the TCP mesh itself (debug transport).
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mmd_internal.hpp"

struct ThermoScales {      // Thermo::setup (ref/thermo.cpp:42-72)
  mmd_float t_scale, e_scale, p_scale, mvv2e, dof_boltz;
};

struct mmd_sim {
  mmd_handle* h = nullptr;
  mmd_input in;
  std::string input_file = "in.lj.miniMD";
  int me = 0, nprocs = 1, quiet = 0;
  int local_rank = 0, local_size = 1;
  char launcher[16] = "single";
  mmd_mesh* mesh = nullptr;      // rendezvous of the ranks; stays open as the transport when they share GPUs
  int transport = 0;             // 0 none (one rank), 1 RCCL, 2 caller's host callbacks, 3 built-in TCP mesh (debug)
  int ngpu_node = 0;
  int num_threads = 1, ntypes = 4, halfneigh = 1, ghost_newton = 1, sort = -1, yaml_output = 0, yaml_screen = 0, check_exchange = 0, safe_exchange = 0;
  int nbin[3] = {1, 1, 1};
  int natoms = 0;
  int sort_every = 0;
  int dev_half = 0;          // neighbor-list style used on the device (differs from the requested one only under --eam_half_full)
  int eam_half_full = 0;
  mmd_float prd[3], mass = 1, dt = 0, dtforce = 0;
  ThermoScales th;
  int steps_done = 0;        // steps integrated so far (bench slices)
  double t_run_start = 0;
  // thermo history
  std::vector<int> row_step;
  std::vector<double> row_t, row_u, row_p;
  double last_timers[5] = {0, 0, 0, 0, 0};
};

static bool g_have_id = false;
static unsigned char g_id[128];
static mmd_sendrecv_fn g_host_sr = nullptr;
static mmd_allreduce_fn g_host_ar = nullptr;
static void* g_host_ctx = nullptr;

// host-staged transport for every sim created afterwards (tests: gloo; also an MPI bridge) instead of RCCL
extern "C" int mmd_sim_set_host_transport(mmd_sendrecv_fn sr, mmd_allreduce_fn ar, void* ctx)
{
  g_host_sr = sr; g_host_ar = ar; g_host_ctx = ctx;
  return 0;
}

extern "C" int mmd_sim_set_unique_id(const unsigned char id[128])
{
  memcpy(g_id, id, 128);
  g_have_id = true;
  return 0;
}

static bool is_flag(const char* a, const char* s1, const char* s2 = nullptr) { return !strcmp(a, s1) || (s2 && !strcmp(a, s2)); }

static void print_help()
{
  printf("\n%s\n\n", mmd_variant_string());
  printf("miniMD-HIP: the per-timestep path of Mantevo miniMD (LJ/EAM forces over neighbor lists, binned neighbor\n"
         "build, velocity-Verlet) as native MI355X HIP kernels. Same options and input deck as miniMD-Reference:\n\n");
  printf("  -i / --input_file <file>   input deck (default in.lj.miniMD)\n  -n / --nsteps <int>        number of timesteps\n"
         "  -s / --size <int>          unit cells per dimension;  -nx/-ny/-nz <int> per dimension\n"
         "  --ntypes <int>             number of atom types (default 4)\n  -b / --neigh_bins <int>    bins per dimension\n"
         "  --half_neigh <int>         1 half neighbor lists (default), 0 full lists\n"
         "  -gn / --ghost_newton <int> Newton's third law across ghosts (half lists; default 1, EAM forces 0)\n"
         "  --sort <n>                 re-sort atoms every n steps (default: every re-neighboring, 0 never)\n"
         "  -u / --units <lj|metal>    -p / --force <lj|eam>    -t / --num_threads <int> (accepted, unused)\n"
         "  -o / --yaml_output <int>   --yaml_screen   -f / --data_file <file>   --check_exchange   --safe_exchange   -h / --help\n\n");
}

static void thermo_setup(mmd_sim* s)
{
  ThermoScales& t = s->th;
  if(s->in.units == 0) {
    t.mvv2e = 1.0;
    t.dof_boltz = (s->natoms * 3 - 3);
    t.t_scale = t.mvv2e / t.dof_boltz;
    t.p_scale = 1.0 / 3 / s->prd[0] / s->prd[1] / s->prd[2];
    t.e_scale = 0.5;
  } else {
    t.mvv2e = 1.036427e-04;
    t.dof_boltz = (s->natoms * 3 - 3) * 8.617343e-05;
    t.t_scale = t.mvv2e / t.dof_boltz;
    t.p_scale = 1.602176e+06 / 3 / s->prd[0] / s->prd[1] / s->prd[2];
    t.e_scale = 524287.985533;
    s->dtforce /= t.mvv2e;                        // ref/thermo.cpp:69
  }
}

// Thermo::compute row (ref/thermo.cpp:74-194) from globally reduced raw sums
static void thermo_row(void* ctx, int step, double sum_mv2, double eng_vdwl, double virial)
{
  mmd_sim* s = (mmd_sim*)ctx;
  const ThermoScales& th = s->th;
  const mmd_float t = (mmd_float)sum_mv2 * th.t_scale;
  mmd_float e_act = (mmd_float)eng_vdwl;
  if(s->halfneigh && !s->dev_half) e_act *= 0.5;       // full-list kernels standing in for a half-list request (EAM)
  if(s->halfneigh) e_act *= 2.0;
  e_act *= th.e_scale;
  const mmd_float eng = e_act / s->natoms;
  const mmd_float p = (t * th.dof_boltz + (mmd_float)virial) * th.p_scale;
  s->row_step.push_back(step); s->row_t.push_back(t); s->row_u.push_back(eng); s->row_p.push_back(p);
  if(s->me == 0 && !s->quiet) {
    fprintf(stdout, "%i %e %e %e %6.3lf\n", step, (double)t, (double)eng, (double)p, step == 0 ? 0.0 : mmd_wall() - s->t_run_start);
    fflush(stdout);
  }
}

extern "C" int mmd_sim_create(int argc, char** argv, int quiet, mmd_sim** out)
{
  if(!out) { mmd_set_error("mmd_sim_create: out is NULL"); return -1; }
  *out = nullptr;
  mmd_sim* s = new mmd_sim();
  s->quiet = quiet;
  // rank / size as the launcher describes them (the reference asks MPI, ref/ljs.cpp:63-68)
  if(mmd_launch_env(&s->me, &s->nprocs, &s->local_rank, &s->local_size, s->launcher, (int)sizeof(s->launcher)) < 0) {
    printf("ERROR: %s\n", mmd_last_error());
    delete s;
    return -1;
  }
  for(int i = 0; i < argc; i++)
    if(is_flag(argv[i], "-i", "--input_file") && i + 1 < argc) s->input_file = argv[++i];
  if(mmd_input_read(&s->in, s->input_file.c_str())) {
    if(s->me == 0 && !quiet) printf("%s\n", mmd_last_error());
    delete s;
    return -1;
  }
  int num_steps = -1, system_size = -1, nx = -1, ny = -1, nz = -1, neighbor_size = -1;
  for(int i = 0; i < argc; i++) {
    const char* a = argv[i];
    const bool has = i + 1 < argc;
    if(is_flag(a, "-t", "--num_threads") && has) s->num_threads = atoi(argv[++i]);
    else if(is_flag(a, "--teams") && has) ++i;
    else if(is_flag(a, "-n", "--nsteps") && has) num_steps = atoi(argv[++i]);
    else if(is_flag(a, "-s", "--size") && has) system_size = atoi(argv[++i]);
    else if(is_flag(a, "-nx") && has) nx = atoi(argv[++i]);
    else if(is_flag(a, "-ny") && has) ny = atoi(argv[++i]);
    else if(is_flag(a, "-nz") && has) nz = atoi(argv[++i]);
    else if(is_flag(a, "--ntypes") && has) s->ntypes = atoi(argv[++i]);
    else if(is_flag(a, "-b", "--neigh_bins") && has) neighbor_size = atoi(argv[++i]);
    else if(is_flag(a, "--half_neigh") && has) s->halfneigh = atoi(argv[++i]);
    else if(is_flag(a, "-sse") && has) ++i;
    else if(is_flag(a, "--sort") && has) s->sort = atoi(argv[++i]);
    else if(is_flag(a, "-o", "--yaml_output") && has) s->yaml_output = atoi(argv[++i]);
    else if(is_flag(a, "--yaml_screen")) s->yaml_screen = 1;
    else if(is_flag(a, "--check_exchange")) s->check_exchange = 1;
    else if(is_flag(a, "--safe_exchange")) s->safe_exchange = 1;        // (ref/ljs.cpp:251 lists it; Comm::do_safeexchange, ref/comm.cpp:366-367)
    else if(is_flag(a, "-f", "--data_file") && has) { s->in.has_datafile = 1; strncpy(s->in.datafile, argv[++i], sizeof(s->in.datafile) - 1); }
    else if(is_flag(a, "-u", "--units") && has) s->in.units = strcmp(argv[++i], "metal") == 0 ? 1 : 0;
    else if(is_flag(a, "-p", "--force") && has) s->in.forcetype = strcmp(argv[++i], "eam") == 0 ? 1 : 0;
    else if(is_flag(a, "-gn", "--ghost_newton") && has) s->ghost_newton = atoi(argv[++i]);
    else if(is_flag(a, "--eam_half_full")) s->eam_half_full = 1;
    else if(is_flag(a, "-h", "--help")) { if(s->me == 0) print_help(); delete s; return 1; }
    // unknown flags are ignored, like the reference (run_one_test passes -dm)
  }
  // --half_neigh -1 ("original miniMD force", ref/force_lj.cpp:118-176: half list, force on both partners, reverse halo): the
  // half-list + ghost-newton lists with the un-tiled force kernel (k_lj_half: one atom per lane walks its row, the partner is
  // updated with one atomic per pair and component) — the direct rendering of that loop, kept as a path of its own
  if(s->halfneigh < 0) s->ghost_newton = s->in.forcetype == 1 ? 0 : 1;
  if(s->in.forcetype == 1 && s->ghost_newton == 1) {
    if(s->me == 0 && !quiet) printf("# EAM currently requires '--ghost_newton 0'; Changing setting now.\n");
    s->ghost_newton = 0;
  }
  if(num_steps > 0) s->in.ntimes = num_steps;
  if(system_size > 0) s->in.nx = s->in.ny = s->in.nz = system_size;
  if(nx > 0) {
    s->in.nx = nx;
    if(ny > 0) s->in.ny = ny; else if(system_size < 0) s->in.ny = nx;
    if(nz > 0) s->in.nz = nz; else if(system_size < 0) s->in.nz = nx;
  }
  // LAMMPS data file (ref/ljs.cpp:385-391, ref/setup.cpp:218-301): every rank parses the whole file, like the reference
  std::vector<mmd_float> file_x, file_v;
  int file_natoms = 0;
  mmd_float file_mass = -1;
  if(s->in.has_datafile) {
    int rc = mmd_lammps_data_read(s->in.datafile, &file_natoms, s->prd, &file_mass, nullptr, nullptr);
    if(rc == 0) {
      file_x.resize((size_t)3 * file_natoms); file_v.resize((size_t)3 * file_natoms);
      rc = mmd_lammps_data_read(s->in.datafile, &file_natoms, s->prd, &file_mass, file_x.data(), file_v.data());
    }
    if(rc) { if(s->me == 0 && !quiet) printf("ERROR: %s\n", mmd_last_error()); delete s; return -1; }
  }
  if(neighbor_size > 0) s->nbin[0] = s->nbin[1] = s->nbin[2] = neighbor_size;
  else if(s->in.has_datafile) {                   // bins from the number density (ref/setup.cpp:229-236)
    const mmd_float volume = s->prd[0] * s->prd[1] * s->prd[2];
    const mmd_float rho = 1.0 * file_natoms / volume;
    const mmd_float neigh_bin_size = std::pow(rho * 16, mmd_float(1.0 / 3.0));
    for(int d = 0; d < 3; d++) s->nbin[d] = s->prd[d] / neigh_bin_size;
  } else {
    const mmd_float neighscale = 5.0 / 6.0;       // evaluated in MMD_float (ref/ljs.cpp:357-362)
    s->nbin[0] = neighscale * s->in.nx; s->nbin[1] = neighscale * s->in.ny; s->nbin[2] = neighscale * s->in.nz;
  }
  for(int d = 0; d < 3; d++) if(s->nbin[d] == 0) s->nbin[d] = 1;
  s->sort_every = s->sort > 0 ? s->sort : (s->sort < 0 ? s->in.neigh_every : 0);
  s->dt = s->in.dt;

  if(s->me == 0 && !quiet) printf("# Create System:\n");
  {
    int ndev = mmd_device_count();
    s->ngpu_node = ndev;
    if(mmd_create(ndev > 0 ? s->local_rank % ndev : -1, &s->h)) { if(s->me == 0 && !quiet) printf("ERROR: %s\n", mmd_last_error()); delete s; return -1; }
  }
  mmd_handle* h = s->h;
#define SIM_TRY(expr) do { if((expr) < 0) { if(s->me == 0 && !quiet) printf("ERROR: %s\n", mmd_last_error()); mmd_sim_destroy(s); return -1; } } while(0)
  if(s->in.has_datafile) {                         // ref/ljs.cpp:387-388
    const mmd_float volume = s->prd[0] * s->prd[1] * s->prd[2];
    s->in.rho = 1.0 * file_natoms / volume;
    if(file_mass >= 0) s->mass = file_mass;        // "Masses" section (ref/setup.cpp:267-276); EAM overrides it below
  } else mmd_create_box(s->in.nx, s->in.ny, s->in.nz, s->in.rho, s->prd);
  { const mmd_float zero[3] = {0, 0, 0}; SIM_TRY(mmd_atom_set_box(h, s->prd, zero, s->prd)); }
  SIM_TRY(mmd_comm_setup(h, s->in.neigh_cut, s->me, s->nprocs));
  if(s->nprocs > 1 && g_host_sr) {
    SIM_TRY(mmd_comm_set_host_transport(h, g_host_sr, g_host_ar, g_host_ctx));
    s->transport = 2;
  } else if(s->nprocs > 1 && g_have_id) {
    SIM_TRY(mmd_comm_init_rccl(h, g_id, s->me, s->nprocs));
    s->transport = 1;
  } else if(s->nprocs > 1) {
    // the ranks meet (launch.cpp) and settle the transport TOGETHER: RCCL when every rank of every node has a device of its own, otherwise — or on
    // request, MMD_TRANSPORT=tcp — the mesh itself carries the messages, staged through host memory (a debug transport: the reference's np = 3 / 8
    // validation runs on a one-GPU box). MMD_TRANSPORT=rccl insists on RCCL and fails where it cannot be had.
    char addr[256];
    int port = 0;
    SIM_TRY(mmd_launch_rendezvous(addr, (int)sizeof(addr), &port));
    SIM_TRY(mmd_mesh_create(s->me, s->nprocs, addr, port, &s->mesh));
    struct Card { int ndev, local_size, want; unsigned host; } mine, *all;
    char hn[256] = {0};
    gethostname(hn, sizeof(hn) - 1);
    unsigned hh = 2166136261u;
    for(const char* c = hn; *c; c++) { hh ^= (unsigned char)*c; hh *= 16777619u; }
    const char* want = getenv("MMD_TRANSPORT");
    mine = Card{s->ngpu_node, s->local_size, want && !strcmp(want, "tcp") ? 2 : (want && !strcmp(want, "rccl") ? 1 : 0), hh};
    std::vector<Card> cards(s->nprocs);
    all = cards.data();
    SIM_TRY(mmd_mesh_allgather(s->mesh, &mine, (int)sizeof(Card), all));
    bool gpu_each = true, any_tcp = false, any_rccl = false;
    for(int r = 0; r < s->nprocs; r++) {
      int on_host = 0;
      for(int q = 0; q < s->nprocs; q++) on_host += all[q].host == all[r].host ? 1 : 0;
      if(on_host > all[r].ndev) gpu_each = false;
      any_tcp = any_tcp || all[r].want == 2;
      any_rccl = any_rccl || all[r].want == 1;
    }
    if(any_rccl && !gpu_each) { mmd_set_error("MMD_TRANSPORT=rccl, but the ranks of a node outnumber its GPUs (%d ranks, %d visible here)", s->nprocs, s->ngpu_node); SIM_TRY(-1); }
    if(gpu_each && !any_tcp) {
      std::vector<unsigned char> ids((size_t)128 * s->nprocs, 0);
      unsigned char id[128] = {0};
      if(s->me == 0) SIM_TRY(mmd_comm_unique_id(id));
      SIM_TRY(mmd_mesh_allgather(s->mesh, id, 128, ids.data()));
      SIM_TRY(mmd_comm_init_rccl(h, ids.data(), s->me, s->nprocs));
      mmd_mesh_destroy(s->mesh);
      s->mesh = nullptr;
      s->transport = 1;
    } else {
      SIM_TRY(mmd_comm_set_host_transport(h, mmd_mesh_sendrecv, mmd_mesh_allreduce, s->mesh));
      s->transport = 3;
    }
  }
  // Device list style: as requested. EAM with half lists is ForceEAM::compute_halfneigh (ref/force_eam.cpp:94-270, third-law
  // scatter with atomics). `--eam_half_full` (ours; the reference ignores unknown flags) serves such a request on the faster
  // full-list kernels instead: forces are identical and eng_vdwl is converted to the half-list convention (full lists report
  // 2x, ref/force_eam.cpp:446 vs :269), so thermo rows are unchanged.
  s->dev_half = s->halfneigh != 0 ? 1 : 0;
  if(s->in.forcetype == 1 && s->eam_half_full) s->dev_half = 0;
  SIM_TRY(mmd_neighbor_setup(h, s->nbin, s->in.neigh_cut, s->dev_half, s->ghost_newton, s->ntypes));
  if(s->halfneigh < 0 && s->in.forcetype == 0) SIM_TRY(mmd_set_option(h, "lj_original", 1));      // ForceLJ::compute_original
  s->dtforce = 0.5 * s->dt;                        // Integrate::setup (ref/integrate.cpp:41-44)
  const int nt2 = s->ntypes * s->ntypes;
  if(s->in.forcetype == 0) {
    std::vector<mmd_float> cut(nt2), s6(nt2), eps(nt2);
    for(int i = 0; i < nt2; i++) {                 // ref/ljs.cpp:299-305, ref/force_lj.cpp:65-69
      const mmd_float sg = s->in.sigma;
      eps[i] = s->in.epsilon; s6[i] = sg * sg * sg * sg * sg * sg; cut[i] = s->in.force_cut * s->in.force_cut;
    }
    SIM_TRY(mmd_force_lj_setup(h, s->ntypes, cut.data(), s6.data(), eps.data()));
  } else {
    int nr, nrho, nr_tot, nrho_tot;
    mmd_float rdr, rdrho, cutmax, mass;
    SIM_TRY(mmd_eam_tables_from_file("Cu_u6.eam", s->ntypes, &nr, &nrho, &nr_tot, &nrho_tot, &rdr, &rdrho, &cutmax, &mass, nullptr, nullptr, nullptr));
    std::vector<mmd_float> rhor((size_t)nt2 * nr_tot), frho((size_t)nt2 * nrho_tot), z2r((size_t)nt2 * nr_tot), cut(nt2, cutmax * cutmax);
    SIM_TRY(mmd_eam_tables_from_file("Cu_u6.eam", s->ntypes, &nr, &nrho, &nr_tot, &nrho_tot, &rdr, &rdrho, &cutmax, &mass, rhor.data(), frho.data(), z2r.data()));
    SIM_TRY(mmd_force_eam_setup(h, s->ntypes, nr, nrho, nr_tot, nrho_tot, rdr, rdrho, rhor.data(), frho.data(), z2r.data(), cut.data()));
    s->mass = mass;                                // ref/ljs.cpp:403
    s->in.force_cut = cutmax;
  }
  SIM_TRY(mmd_atom_set_mass(h, s->mass));
  int nlocal = 0;
  std::vector<mmd_float> x, v;
  std::vector<int> type, tag;
  if(s->in.has_datafile) {
    // atoms of my sub-box in file order, velocities as read (no create_velocity) — ref/setup.cpp:281-297
    SIM_TRY(mmd_lammps_data_select(file_natoms, file_x.data(), file_v.data(), h->lo, h->hi, s->ntypes, nullptr, nullptr, nullptr, nullptr, &nlocal));
    x.resize((size_t)3 * nlocal + 3); v.resize((size_t)3 * nlocal + 3); type.resize(nlocal + 1); tag.resize(nlocal + 1);
    SIM_TRY(mmd_lammps_data_select(file_natoms, file_x.data(), file_v.data(), h->lo, h->hi, s->ntypes, x.data(), v.data(), type.data(), tag.data(), &nlocal));
    s->natoms = file_natoms;
    double cnt = nlocal;
    SIM_TRY(mmd_transport_allreduce(h, &cnt, 1));
    if((long long)cnt != s->natoms && s->me == 0 && !quiet) printf("Created incorrect # of atoms\n");   // the reference goes on
    thermo_setup(s);
    file_x.clear(); file_x.shrink_to_fit(); file_v.clear(); file_v.shrink_to_fit();
  } else {
  // create_atoms (ref/setup.cpp:315-450) for my sub-box
  SIM_TRY(mmd_create_atoms(s->in.nx, s->in.ny, s->in.nz, s->in.rho, h->lo, h->hi, s->ntypes, nullptr, nullptr, nullptr, nullptr, &nlocal));
  x.resize((size_t)3 * nlocal + 3); v.resize((size_t)3 * nlocal + 3); type.resize(nlocal + 1); tag.resize(nlocal + 1);
  SIM_TRY(mmd_create_atoms(s->in.nx, s->in.ny, s->in.nz, s->in.rho, h->lo, h->hi, s->ntypes, x.data(), v.data(), type.data(), tag.data(), &nlocal));
  s->natoms = 4 * s->in.nx * s->in.ny * s->in.nz;
  { double cnt = nlocal; SIM_TRY(mmd_transport_allreduce(h, &cnt, 1));
    if((long long)cnt != s->natoms) { mmd_set_error("Created incorrect # of atoms"); if(s->me == 0 && !quiet) printf("%s\n", mmd_last_error()); mmd_sim_destroy(s); return -1; } }
  thermo_setup(s);
  // create_velocity (ref/setup.cpp:454-494): remove centre-of-mass motion, rescale to t_request
  {
    double vtot[3] = {0, 0, 0};
    for(int i = 0; i < nlocal; i++) for(int d = 0; d < 3; d++) vtot[d] += v[3 * (size_t)i + d];
    SIM_TRY(mmd_transport_allreduce(h, vtot, 3));
    for(int d = 0; d < 3; d++) vtot[d] /= s->natoms;
    for(int i = 0; i < nlocal; i++) for(int d = 0; d < 3; d++) v[3 * (size_t)i + d] -= vtot[d];
    mmd_float t_act = 0;
    for(int i = 0; i < nlocal; i++) {
      const mmd_float vx = v[3 * (size_t)i], vy = v[3 * (size_t)i + 1], vz = v[3 * (size_t)i + 2];
      t_act += (vx * vx + vy * vy + vz * vz) * s->mass;
    }
    double tsum = t_act;
    SIM_TRY(mmd_transport_allreduce(h, &tsum, 1));
    const double t = (mmd_float)tsum * s->th.t_scale;
    const double factor = sqrt(s->in.t_request / t);
    for(size_t i = 0; i < (size_t)3 * nlocal; i++) v[i] *= factor;
  }
  }
  SIM_TRY(mmd_atom_upload(h, x.data(), v.data(), type.data(), tag.data(), nlocal, 0));
  // dtforce chain: 0.5*dt [/mvv2e] /mass (ref/integrate.cpp:43,80-81; ref/thermo.cpp:69)
  SIM_TRY(mmd_integrate_setup(h, s->dt, s->dtforce / s->mass, s->in.neigh_every, s->sort_every));
  if(s->check_exchange) SIM_TRY(mmd_set_option(h, "check_exchange", 1));
  if(s->safe_exchange) SIM_TRY(mmd_set_option(h, "safe_exchange", 1));
#undef SIM_TRY
  if(s->me == 0 && !quiet) {
    printf("# Done .... \n");
    fprintf(stdout, "# %s output ...\n", mmd_variant_string());
    fprintf(stdout, "# Run Settings: \n");
    fprintf(stdout, "\t# MPI processes: %i\n", s->nprocs);
    if(s->nprocs > 1)
      fprintf(stdout, "\t# Transport: %s (launcher: %s)\n", s->transport == 1 ? "RCCL point-to-point, one GPU per rank" :
              (s->transport == 3 ? "TCP mesh staged through host memory — DEBUG transport, the ranks share GPUs" : "host callbacks of the caller"), s->launcher);
    fprintf(stdout, "\t# OpenMP threads: %i\n", s->num_threads);
    fprintf(stdout, "\t# Inputfile: %s\n", s->input_file.c_str());
    fprintf(stdout, "\t# Datafile: %s\n", s->in.has_datafile ? s->in.datafile : "None");
    fprintf(stdout, "# Physics Settings: \n");
    fprintf(stdout, "\t# ForceStyle: %s\n", s->in.forcetype == 0 ? "LJ" : "EAM");
    fprintf(stdout, "\t# Force Parameters: %2.2lf %2.2lf\n", (double)s->in.epsilon, (double)s->in.sigma);
    fprintf(stdout, "\t# Units: %s\n", s->in.units == 0 ? "LJ" : "METAL");
    fprintf(stdout, "\t# Atoms: %i\n", s->natoms);
    fprintf(stdout, "\t# Atom types: %i\n", s->ntypes);
    fprintf(stdout, "\t# System size: %2.2lf %2.2lf %2.2lf (unit cells: %i %i %i)\n", (double)s->prd[0], (double)s->prd[1], (double)s->prd[2], s->in.nx, s->in.ny, s->in.nz);
    fprintf(stdout, "\t# Density: %lf\n", (double)s->in.rho);
    fprintf(stdout, "\t# Force cutoff: %lf\n", (double)s->in.force_cut);
    fprintf(stdout, "\t# Timestep size: %lf\n", (double)s->dt);
    fprintf(stdout, "# Technical Settings: \n");
    fprintf(stdout, "\t# Neigh cutoff: %lf\n", (double)s->in.neigh_cut);
    fprintf(stdout, "\t# Half neighborlists: %i\n", s->halfneigh);
    fprintf(stdout, "\t# Neighbor bins: %i %i %i\n", s->nbin[0], s->nbin[1], s->nbin[2]);
    fprintf(stdout, "\t# Neighbor frequency: %i\n", s->in.neigh_every);
    fprintf(stdout, "\t# Sorting frequency: %i\n", s->sort_every);
    fprintf(stdout, "\t# Thermo frequency: %i\n", s->in.thermo_nstat);
    fprintf(stdout, "\t# Ghost Newton: %i\n", s->ghost_newton);
    fprintf(stdout, "\t# Use intrinsics: %i\n", 0);
    fprintf(stdout, "\t# Do safe exchange: %i\n", s->safe_exchange);
    fprintf(stdout, "\t# Size of float: %i\n\n", (int)sizeof(mmd_float));
  }
  *out = s;
  return 0;
}

// force (evflag=1) + thermo row for `step`, globally reduced
static int force_and_row(mmd_sim* s, int step)
{
  mmd_handle* h = s->h;
  double eng = 0, vir = 0, mv2 = 0;
  MMD_TRY(mmd_force_compute(h, 1, &eng, &vir));
  if(s->dev_half && s->ghost_newton) MMD_TRY(mmd_comm_reverse_communicate(h));
  MMD_TRY(mmd_thermo_temperature(h, &mv2));
  double vals[3] = {mv2, eng, vir};
  MMD_TRY(mmd_transport_allreduce(h, vals, 3));
  thermo_row(s, step, vals[0], vals[1], vals[2]);
  return 0;
}

extern "C" int mmd_sim_initial(mmd_sim* s)
{
  if(!s) { mmd_set_error("null sim"); return -1; }
  mmd_handle* h = s->h;
  MMD_TRY(mmd_comm_exchange(h));
  if(s->sort > 0) MMD_TRY(mmd_atom_sort(h));
  MMD_TRY(mmd_comm_borders(h));
  MMD_TRY(mmd_neighbor_build(h));
  if(s->me == 0 && !s->quiet) { printf("# Starting dynamics ...\n"); printf("# Timestep T U P Time\n"); }
  s->row_step.clear(); s->row_t.clear(); s->row_u.clear(); s->row_p.clear();
  MMD_TRY(force_and_row(s, 0));
  s->steps_done = 0;
  return 0;
}

extern "C" int mmd_sim_run(mmd_sim* s)
{
  if(!s) { mmd_set_error("null sim"); return -1; }
  mmd_handle* h = s->h;
  const int nstat = s->in.thermo_nstat;
  s->t_run_start = mmd_wall();
  MMD_TRY(mmd_integrate_run(h, s->steps_done, s->in.ntimes, nstat, thermo_row, s));
  MMD_TRY(mmd_timers(h, s->last_timers, nullptr, nullptr));
  s->steps_done += s->in.ntimes;
  // ref/ljs.cpp:477-483 + Thermo::compute(-1) gating (ref/thermo.cpp:80)
  if(!(nstat > 0 && s->in.ntimes % nstat == 0)) MMD_TRY(force_and_row(s, s->in.ntimes));
  else { MMD_TRY(mmd_force_compute(h, 1, nullptr, nullptr)); if(s->dev_half && s->ghost_newton) MMD_TRY(mmd_comm_reverse_communicate(h)); }
  return 0;
}

extern "C" int mmd_sim_run_steps(mmd_sim* s, int nsteps, double* seconds)
{
  if(!s || nsteps < 0) { mmd_set_error("mmd_sim_run_steps: bad arguments"); return -1; }
  mmd_handle* h = s->h;
  MMD_TRY(mmd_sync(h));
  const double t0 = mmd_wall();
  s->t_run_start = t0;
  MMD_TRY(mmd_integrate_run(h, s->steps_done, nsteps, s->in.thermo_nstat, thermo_row, s));
  MMD_TRY(mmd_sync(h));
  if(seconds) *seconds = mmd_wall() - t0;
  MMD_TRY(mmd_timers(h, s->last_timers, nullptr, nullptr));
  s->steps_done += nsteps;
  return 0;
}

extern "C" int mmd_sim_print_perf(mmd_sim* s)
{
  if(!s) { mmd_set_error("null sim"); return -1; }
  if(s->me != 0 || s->quiet) return 0;
  const double* t = s->last_timers;
  const double other = t[0] - t[2] - t[3] - t[1];
  printf("\n\n");
  printf("# Performance Summary:\n");
  printf("# MPI_proc OMP_threads nsteps natoms t_total t_force t_neigh t_comm t_other performance perf/thread grep_string t_extra\n");
  printf("%i %i %i %i %lf %lf %lf %lf %lf %lf %lf PERF_SUMMARY %lf\n\n\n", s->nprocs, s->num_threads, s->in.ntimes, s->natoms, t[0], t[2],
         t[3], t[1], other, 1.0 * s->natoms * s->in.ntimes / t[0], 1.0 * s->natoms * s->in.ntimes / t[0] / s->nprocs / s->num_threads, t[4]);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// YAML report of `-o 1` / `--yaml_screen` — counterpart of output() + stats() (ref/output.cpp:48-547): same keys,
// same sections (run_configuration, thermodynamic_output incl. the energy-conservation ratio, time, per-rank
// histograms of timings and of Nlocal / Nghost / Nswaps / Neighs, total neighbor count).
// ---------------------------------------------------------------------------------------------------
namespace {
struct Emit {                       // every line goes to the file and, when asked, to stdout as well
  FILE* fp; bool screen;
  void operator()(const char* fmt, ...) const
  {
    va_list ap;
    if(fp) { va_start(ap, fmt); vfprintf(fp, fmt, ap); va_end(ap); }
    if(screen) { va_start(ap, fmt); vfprintf(stdout, fmt, ap); va_end(ap); }
  }
};
// all ranks' values of one scalar on every rank (sum-allreduce of a one-hot vector, 16 ranks per call)
int gather_all(mmd_sim* s, double mine, std::vector<double>& all)
{
  all.assign(s->nprocs, 0.0);
  for(int base = 0; base < s->nprocs; base += 16) {
    double buf[16] = {0};
    const int n = std::min(16, s->nprocs - base);
    if(s->me >= base && s->me < base + n) buf[s->me - base] = mine;
    MMD_TRY(mmd_transport_allreduce(s->h, buf, n));
    for(int i = 0; i < n; i++) all[base + i] = buf[i];
  }
  return 0;
}
// ave / max / min + 10-bin histogram over ranks (stats(), ref/output.cpp:496-547)
void histogram(const std::vector<double>& v, double* ave, double* mx, double* mn, int histo[10])
{
  *mn = 1.0e20; *mx = -1.0e20; *ave = 0.0;
  for(double d : v) { *ave += d; if(d < *mn) *mn = d; if(d > *mx) *mx = d; }
  *ave /= (double)v.size();
  for(int i = 0; i < 10; i++) histo[i] = 0;
  const double del = *mx - *mn;
  for(double d : v) {
    int m = del == 0.0 ? 0 : static_cast<int>((d - *mn) / del * 10);
    if(m > 9) m = 9;
    histo[m]++;
  }
}
}  // namespace

extern "C" int mmd_sim_output(mmd_sim* s, int screen_yaml)
{
  if(!s) { mmd_set_error("null sim"); return -1; }
  mmd_handle* h = s->h;
  // enforce PBC, then check for lost atoms (ref/output.cpp:60-85)
  MMD_TRY(mmd_atom_pbc(h));
  int nlocal = 0, nghost = 0;
  MMD_TRY(mmd_atom_counts(h, &nlocal, &nghost, nullptr));
  std::vector<mmd_float> x((size_t)3 * (nlocal + nghost) + 3);
  MMD_TRY(mmd_atom_download(h, x.data(), nullptr, nullptr, nullptr, nullptr));
  double counts[2] = {(double)nlocal, 0.0};
  for(int i = 0; i < nlocal; i++)
    for(int d = 0; d < 3; d++)
      if(x[3 * (size_t)i + d] < 0.0 || x[3 * (size_t)i + d] >= s->prd[d]) { counts[1] += 1; break; }
  MMD_TRY(mmd_transport_allreduce(h, counts, 2));
  if((long long)counts[0] != s->natoms || counts[1] > 0) {
    if(s->me == 0) { printf("Atom counts = %d %d %d\n", (int)counts[1], (int)counts[0], s->natoms); printf("ERROR: Incorrect number of atoms\n"); }
    return 0;
  }
  FILE* fp = nullptr;
  if(s->me == 0) {
    time_t now = time(NULL);
    struct tm lt = *localtime(&now);
    char name[256];
    snprintf(name, sizeof(name), "miniMD-%4d-%02d-%02d-%02d-%02d-%02d.yaml", lt.tm_year + 1900, lt.tm_mon + 1, lt.tm_mday, lt.tm_hour, lt.tm_min, lt.tm_sec);
    fp = fopen(name, "w");
  }
  Emit out{fp, s->me == 0 && screen_yaml != 0};
  out("run_configuration: \n");
  out("  variant: %s\n", mmd_variant_string());
  out("  mpi_processes: %i\n", s->nprocs);
  out("  thread_teams: %i\n", 1);
  out("  threads: %i\n", s->num_threads);
  out("  datafile: %s\n", s->in.has_datafile ? s->in.datafile : "None");
  out("  units: %s\n", s->in.units == 0 ? "LJ" : "METAL");
  out("  atoms: %i\n", s->natoms);
  out("  atom_types: %i\n", s->ntypes);
  out("  system_size: %2.2lf %2.2lf %2.2lf\n", (double)s->prd[0], (double)s->prd[1], (double)s->prd[2]);
  out("  unit_cells: %i %i %i\n", s->in.nx, s->in.ny, s->in.nz);
  out("  density: %lf\n", (double)s->in.rho);
  out("  force_type: %s\n", s->in.forcetype == 0 ? "LJ" : "EAM");
  out("  force_cutoff: %lf\n", (double)s->in.force_cut);
  out("  force_params: %2.2lf %2.2lf\n", (double)s->in.epsilon, (double)s->in.sigma);
  out("  neighbor_cutoff: %lf\n", (double)s->in.neigh_cut);
  out("  neighbor_type: %i\n", s->halfneigh);
  out("  neighbor_bins: %i %i %i\n", s->nbin[0], s->nbin[1], s->nbin[2]);
  out("  neighbor_frequency: %i\n", s->in.neigh_every);
  out("  sort_frequency: %i\n", s->sort_every);
  out("  timestep_size: %lf\n", (double)s->dt);
  out("  thermo_frequency: %i\n", s->in.thermo_nstat);
  out("  ghost_newton: %i\n", s->ghost_newton);
  out("  use_intrinsics: %i\n", 0);
  out("  safe_exchange: %i\n", s->safe_exchange);
  out("  float_size: %i\n\n", (int)sizeof(mmd_float));
  out("\n\nthermodynamic_output:\n");
  for(size_t i = 0; i < s->row_step.size(); i++) {
    const double conserve = (1.5 * s->row_t[i] + s->row_u[i]) / (1.5 * s->row_t[0] + s->row_u[0]);
    out("  timestep: %d \n", s->row_step[i]);
    out("      T*:           %15.10g \n", s->row_t[i]);
    out("      U*:           %15.10g \n", s->row_u[i]);
    out("      P*:           %15.10g \n", s->row_p[i]);
    out("      Conservation: %15.10g \n", conserve);
  }
  if(s->me == 0) { fprintf(stdout, "\n\n"); if(fp) fprintf(fp, "\n\n"); }
  // timings averaged over ranks
  const double* t = s->last_timers;
  double tv[4] = {t[0], t[2], t[3], t[1]};                 // total, force, neigh, comm
  MMD_TRY(mmd_transport_allreduce(h, tv, 4));
  for(double& v : tv) v /= s->nprocs;
  double time_total = tv[0];
  out("time:\n  total:\n");
  out("    time: %g \n", time_total);
  out("    performance: %10.5e \n", (double)s->natoms * s->in.ntimes / time_total);
  out("    performance_proc: %10.5e \n", (double)s->natoms * s->in.ntimes / time_total / s->nprocs / s->num_threads);
  if(time_total == 0.0) time_total = 1.0;
  out("  force: %g\n", tv[1]);
  out("  neigh: %g\n", tv[2]);
  out("  comm:  %g\n", tv[3]);
  out("  other: %g\n", tv[0] - (tv[1] + tv[2] + tv[3]));
  out("\n");
  // per-rank histograms
  long long nswaps = 0, neighs = 0;
  { int ns = 0; MMD_TRY(mmd_comm_info(h, nullptr, nullptr, nullptr, nullptr, &ns));
    for(int i = 0; i < ns; i++) { int c[3]; MMD_TRY(mmd_comm_swap_info(h, i, nullptr, nullptr, nullptr, c)); nswaps += c[0]; } }
  MMD_TRY(mmd_neighbor_info(h, nullptr, nullptr, &neighs, nullptr));
  const struct { const char* label; double v; } items[] = {
    {"# Force time:", t[2]}, {"# Neigh time:", t[3]}, {"# Comm  time:", t[1]}, {"# Other time:", t[0] - (t[2] + t[3] + t[1])},
    {"# Nlocal:    ", (double)nlocal}, {"# Nghost:    ", (double)nghost}, {"# Nswaps:    ", (double)nswaps}, {"# Neighs:    ", (double)neighs}};
  double total_neigh = 0;
  for(int k = 0; k < 8; k++) {
    std::vector<double> all;
    MMD_TRY(gather_all(s, items[k].v, all));
    double ave, mx, mn; int histo[10];
    histogram(all, &ave, &mx, &mn, histo);
    if(k == 0) out("# Timing histograms \n");
    if(k == 4 && s->me == 0) { fprintf(stdout, "\n"); if(fp) fprintf(fp, "\n"); }
    out("%s %g ave %g max %g min\n", items[k].label, ave, mx, mn);
    out("# Histogram:");
    for(int i = 0; i < 10; i++) out(" %d", histo[i]);
    out("\n");
    if(k == 7) for(double d : all) total_neigh += d;
  }
  out("# Total # of neighbors = %g\n", total_neigh);
  out("\n");
  if(fp) fclose(fp);
  return 0;
}

extern "C" int mmd_sim_rows(mmd_sim* s, int* nrows, int* steps, double* t, double* u, double* p, int maxrows)
{
  if(!s || !nrows) { mmd_set_error("mmd_sim_rows: bad arguments"); return -1; }
  *nrows = (int)s->row_step.size();
  for(int i = 0; i < *nrows && i < maxrows; i++) {
    if(steps) steps[i] = s->row_step[i];
    if(t) t[i] = s->row_t[i];
    if(u) u[i] = s->row_u[i];
    if(p) p[i] = s->row_p[i];
  }
  return 0;
}

extern "C" int mmd_sim_natoms(mmd_sim* s) { return s ? s->natoms : -1; }
extern "C" int mmd_sim_wants_yaml(mmd_sim* s, int* screen) { if(!s) return 0; if(screen) *screen = s->yaml_screen; return s->yaml_output; }
extern "C" mmd_handle* mmd_sim_handle(mmd_sim* s) { return s ? s->h : nullptr; }

extern "C" int mmd_sim_destroy(mmd_sim* s)
{
  if(!s) return 0;
  if(s->h) mmd_destroy(s->h);
  if(s->mesh) mmd_mesh_destroy(s->mesh);
  delete s;
  return 0;
}
