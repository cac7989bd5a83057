// oracle/ref_dump.cpp — TEST INFRASTRUCTURE ONLY.
//
// A small driver of OUR OWN that links against the *unmodified* reference objects
// (/root/reference/ref/{atom,neighbor,force_lj,force_eam,comm,integrate,thermo,timer,setup,input}.cpp,
// compiled where they lie by oracle/Makefile target `refdump`) and dumps per-atom arrays the reference
// computes, so tests/golden/make_golden.py can turn them into committed golden vectors
// (tests/golden/*.npz).  It exists only in this container: /root/reference does not travel.
//
// It calls the reference's public entry points in the order ref/ljs.cpp:393-483 wires them
// (create_box, Comm::setup, Neighbor::setup, Integrate::setup, Force::setup, create_atoms,
//  Thermo::setup, create_velocity, Comm::exchange, Comm::borders, Neighbor::build, Force::compute,
//  Integrate::run) — nothing of the reference is re-implemented here.
//
// usage: ref_dump <deck> <size> <half_neigh> <ghost_newton> <nsteps> <out.bin> [ntypes=4] [nbins=-1]
//
// Output: a flat little-endian record stream:  [name:16 bytes][dtype:'i'|'d'|'f':1 byte pad to 8][count:int64][payload]

#define protected public   // test driver only: lets us read ForceEAM::fp (layout is unaffected)
#include "ljs.h"
#include "atom.h"
#include "neighbor.h"
#include "integrate.h"
#include "thermo.h"
#include "comm.h"
#include "timer.h"
#include "threadData.h"
#include "force.h"
#include "force_lj.h"
#include "force_eam.h"
#undef protected

#include <mpi.h>
#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

int input(In&, const char*);
void create_box(Atom&, int, int, int, double);
int create_atoms(Atom&, int, int, int, double);
void create_velocity(double, Atom&, Thermo&);

static FILE* g_out = nullptr;

static void rec(const char* name, char dtype, int64_t count, const void* data, size_t elsize)
{
  char hdr[24];
  memset(hdr, 0, sizeof(hdr));
  strncpy(hdr, name, 15);
  hdr[16] = dtype;
  fwrite(hdr, 1, 24, g_out);
  fwrite(&count, sizeof(int64_t), 1, g_out);
  if(count) fwrite(data, elsize, (size_t)count, g_out);
}
static void rec_i(const char* name, const int* p, int64_t n) { rec(name, 'i', n, p, sizeof(int)); }
static void rec_r(const char* name, const MMD_float* p, int64_t n)
{
  rec(name, sizeof(MMD_float) == 8 ? 'd' : 'f', n, p, sizeof(MMD_float));
}
static void rec_d1(const char* name, double v) { rec(name, 'd', 1, &v, sizeof(double)); }
static void rec_i1(const char* name, int v) { rec(name, 'i', 1, &v, sizeof(int)); }

static void dump_state(const char* tag, Atom& atom, Neighbor& nb, Force* force, int with_lists)
{
  char nm[32];
  const int nall = atom.nlocal + atom.nghost;
  snprintf(nm, 32, "%s.nlocal", tag); rec_i1(nm, atom.nlocal);
  snprintf(nm, 32, "%s.nghost", tag); rec_i1(nm, atom.nghost);
  snprintf(nm, 32, "%s.x", tag);      rec_r(nm, atom.x, (int64_t)nall * PAD);
  snprintf(nm, 32, "%s.v", tag);      rec_r(nm, atom.v, (int64_t)atom.nlocal * PAD);
  snprintf(nm, 32, "%s.f", tag);      rec_r(nm, atom.f, (int64_t)(nb.halfneigh ? nall : atom.nlocal) * PAD);
  snprintf(nm, 32, "%s.type", tag);   rec_i(nm, atom.type, nall);
  snprintf(nm, 32, "%s.eng_vdwl", tag); rec_d1(nm, force->eng_vdwl);
  snprintf(nm, 32, "%s.virial", tag);   rec_d1(nm, force->virial);
  if(with_lists) {
    snprintf(nm, 32, "%s.maxneighs", tag); rec_i1(nm, nb.maxneighs);
    snprintf(nm, 32, "%s.numneigh", tag);  rec_i(nm, nb.numneigh, atom.nlocal);
    // pack the valid part of every fixed-stride row back to back
    std::vector<int> flat;
    for(int i = 0; i < atom.nlocal; i++)
      for(int k = 0; k < nb.numneigh[i]; k++) flat.push_back(nb.neighbors[(size_t)i * nb.maxneighs + k]);
    snprintf(nm, 32, "%s.neighbors", tag); rec_i(nm, flat.data(), (int64_t)flat.size());
  }
  if(force->style == FORCEEAM) {
    ForceEAM* e = (ForceEAM*)force;
    snprintf(nm, 32, "%s.fp", tag); rec_r(nm, e->fp, nall);
  }
}

int main(int argc, char** argv)
{
  if(argc < 7) {
    fprintf(stderr, "usage: %s deck size half_neigh ghost_newton nsteps out.bin [ntypes] [nbins]\n", argv[0]);
    return 2;
  }
  MPI_Init(&argc, &argv);
  const char* deck = argv[1];
  const int size = atoi(argv[2]);
  const int halfneigh = atoi(argv[3]);
  int ghost_newton = atoi(argv[4]);
  const int nsteps = atoi(argv[5]);
  const int ntypes = argc > 7 ? atoi(argv[7]) : 4;
  const int nbins = argc > 8 ? atoi(argv[8]) : -1;
  g_out = fopen(argv[6], "wb");
  if(!g_out) { perror("out"); return 2; }

  In in;
  in.datafile = NULL;
  if(input(in, deck)) return 1;
  srand(5413);
  in.nx = in.ny = in.nz = size;
  in.ntimes = nsteps;

  Atom atom(ntypes);
  Neighbor neighbor(ntypes);
  Integrate integrate;
  Thermo thermo;
  Comm comm;
  Timer timer;
  ThreadData threads;
  Force* force;
  if(in.forcetype == FORCEEAM) { force = (Force*) new ForceEAM(ntypes); ghost_newton = 0; }
  else force = (Force*) new ForceLJ(ntypes);

  threads.mpi_me = 0; threads.mpi_num_threads = 1; threads.omp_me = 0; threads.omp_num_threads = 1;
  atom.threads = comm.threads = force->threads = integrate.threads = neighbor.threads = thermo.threads = &threads;
  if(in.forcetype == FORCELJ)
    for(int i = 0; i < ntypes * ntypes; i++) {
      force->epsilon[i] = in.epsilon;
      force->sigma[i] = in.sigma;
      force->sigma6[i] = in.sigma * in.sigma * in.sigma * in.sigma * in.sigma * in.sigma;
    }
  neighbor.ghost_newton = ghost_newton;
  omp_set_num_threads(1);
  neighbor.timer = &timer; force->timer = &timer;
  comm.check_safeexchange = 0; comm.do_safeexchange = 0;
  force->use_sse = 0;
  neighbor.halfneigh = halfneigh;
  if(nbins > 0) neighbor.nbinx = neighbor.nbiny = neighbor.nbinz = nbins;
  else {
    MMD_float neighscale = 5.0 / 6.0;
    neighbor.nbinx = neighscale * in.nx; neighbor.nbiny = neighscale * in.ny; neighbor.nbinz = neighscale * in.nz;
  }
  if(neighbor.nbinx == 0) neighbor.nbinx = 1;
  if(neighbor.nbiny == 0) neighbor.nbiny = 1;
  if(neighbor.nbinz == 0) neighbor.nbinz = 1;
  integrate.ntimes = in.ntimes; integrate.dt = in.dt; integrate.sort_every = in.neigh_every;
  neighbor.every = in.neigh_every; neighbor.cutneigh = in.neigh_cut;
  force->cutforce = in.force_cut; thermo.nstat = in.thermo_nstat;

  create_box(atom, in.nx, in.ny, in.nz, in.rho);
  comm.setup(neighbor.cutneigh, atom);
  neighbor.setup(atom);
  integrate.setup();
  force->setup();
  if(in.forcetype == FORCEEAM) atom.mass = force->mass;
  create_atoms(atom, in.nx, in.ny, in.nz, in.rho);
  thermo.setup(in.rho, integrate, atom, in.units);
  create_velocity(in.t_request, atom, thermo);

  rec_i1("natoms", atom.natoms);
  rec_i1("ntypes", ntypes);
  rec_i1("halfneigh", halfneigh);
  rec_i1("ghost_newton", ghost_newton);
  rec_i1("nsteps", nsteps);
  rec_i1("size", size);
  rec_i1("floatsize", (int)sizeof(MMD_float));
  { int nb[3] = {neighbor.nbinx, neighbor.nbiny, neighbor.nbinz}; rec_i("nbin", nb, 3); }
  { double b[3] = {(double)atom.box.xprd, (double)atom.box.yprd, (double)atom.box.zprd}; rec("prd", 'd', 3, b, 8); }
  rec_d1("cutneigh", neighbor.cutneigh);
  rec_d1("cutforce", force->cutforce);
  rec_d1("dt", integrate.dt);
  rec_d1("mass", atom.mass);
  rec_d1("t_scale", thermo.t_scale); rec_d1("e_scale", thermo.e_scale); rec_d1("p_scale", thermo.p_scale);
  rec_d1("dof_boltz", thermo.dof_boltz); rec_d1("mvv2e", thermo.mvv2e);
  rec_r("created.x", atom.x, (int64_t)atom.nlocal * PAD);
  rec_r("created.v", atom.v, (int64_t)atom.nlocal * PAD);
  rec_i("created.type", atom.type, atom.nlocal);

  comm.exchange(atom);
  comm.borders(atom);
  rec_i1("nswap", comm.nswap);
  rec_i("sendnum", comm.sendnum, comm.nswap);
  rec_i("recvnum", comm.recvnum, comm.nswap);
  rec_i("firstrecv", comm.firstrecv, comm.nswap);
  force->evflag = 1;
  #pragma omp parallel
  {
    neighbor.build(atom);
    force->compute(atom, neighbor, comm, 0);
  }
  dump_state("s0pre", atom, neighbor, force, 1);   // forces BEFORE reverse communication (ghost f intact)
  if(neighbor.halfneigh && neighbor.ghost_newton) comm.reverse_communicate(atom);
  dump_state("s0", atom, neighbor, force, 0);
  rec_d1("s0.T", thermo.temperature(atom));

  if(nsteps > 0) {
    // keep the thermo rows quiet: the reference prints them itself to stdout
    timer.barrier_start(TIME_TOTAL);
    integrate.run(atom, force, neighbor, comm, thermo, timer);
    timer.barrier_stop(TIME_TOTAL);
    force->evflag = 1;
    force->compute(atom, neighbor, comm, 0);
    dump_state("s1pre", atom, neighbor, force, 1);
    if(neighbor.halfneigh && neighbor.ghost_newton) comm.reverse_communicate(atom);
    dump_state("s1", atom, neighbor, force, 0);
    rec_d1("s1.T", thermo.temperature(atom));
  }
  fclose(g_out);
  MPI_Finalize();
  return 0;
}
