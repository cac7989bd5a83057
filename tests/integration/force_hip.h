// tests/integration/force_hip.h — the binding INTEGRATION.md §2 documents: a Force plugin for the reference's ref/ tree
// (abstract class Force, ref/force.h:40-69; selected where ref/ljs.cpp:274-285 picks ForceLJ / ForceEAM) that forwards
// Force::setup / Force::compute to the C-ABI of include/mmd.h. It is NOT part of the product: __graft_entry__.build()
// compiles it against the reference's own headers (-fsyntax-only) whenever /root/reference is present, and oracle/Makefile's
// `ref_hip` links it into the reference program (oracle/_ref/ref_hip_{dp,sp}), which the -m gpu tests run.
#ifndef FORCE_HIP_H_
#define FORCE_HIP_H_

#include <cstdio>
#include <cstdlib>
#include "force.h"                      // the reference's header (ref/force.h)

#ifndef MMD_PRECISION
#define MMD_PRECISION PRECISION         // ref/types.h:61-72: PRECISION 1 = float, 2 = double — same convention
#endif
extern "C" {
#include "mmd.h"
}

// What both plugins do per Force::compute: hand the reference's Atom and Neighbor over. The neighbor rows (and types, ghost counts) cross the
// boundary only when the reference has re-neighbored since the last call (Neighbor::ncalls, ref/neighbor.cpp:81) — the library turns them into
// its tile form once (mmd_neighbor_upload) and the tile force kernels serve them for the next `every` steps; on the steps between only the
// positions travel (mmd_atom_upload_x: Atom::x after initialIntegrate + Comm::communicate).
struct HipListCache {
  int ncalls, nlocal, nghost;
  HipListCache() : ncalls(-1), nlocal(-1), nghost(-1) {}
  int sync(mmd_handle* h, Atom &atom, Neighbor &neighbor, int ntypes)
  {
    if(neighbor.ncalls == ncalls && atom.nlocal == nlocal && atom.nghost == nghost)
      return mmd_atom_upload_x(h, atom.x, atom.nlocal + atom.nghost);
    MMD_float prd[3] = {atom.box.xprd, atom.box.yprd, atom.box.zprd};
    MMD_float lo[3] = {atom.box.xlo, atom.box.ylo, atom.box.zlo}, hi[3] = {atom.box.xhi, atom.box.yhi, atom.box.zhi};
    const int nbin[3] = {neighbor.nbinx, neighbor.nbiny, neighbor.nbinz};
    if(mmd_atom_set_box(h, prd, lo, hi) != 0) return -1;
    if(mmd_atom_upload(h, atom.x, atom.v, atom.type, 0, atom.nlocal, atom.nghost) != 0) return -1;          // PAD=3 arrays as they are
    if(mmd_neighbor_setup(h, nbin, neighbor.cutneigh, neighbor.halfneigh, neighbor.ghost_newton, ntypes) != 0) return -1;
    if(mmd_neighbor_upload(h, neighbor.neighbors, neighbor.maxneighs, neighbor.numneigh, atom.nlocal) != 0) return -1;   // ref rows as they are
    ncalls = neighbor.ncalls; nlocal = atom.nlocal; nghost = atom.nghost;
    return 0;
  }
};

class ForceHIP : public Force
{
  public:
    mmd_handle* h;
    HipListCache lists;

    explicit ForceHIP(int ntypes_) : h(0)
    {
      // members as ForceLJ::ForceLJ sets them (ref/force_lj.cpp:41-57)
      cutforce = 0.0;
      use_oldcompute = 0;
      reneigh = 1;
      style = FORCELJ;
      ntypes = ntypes_;
      cutforcesq = new MMD_float[ntypes * ntypes];
      epsilon = new MMD_float[ntypes * ntypes];
      sigma6 = new MMD_float[ntypes * ntypes];
      sigma = new MMD_float[ntypes * ntypes];
      for(int i = 0; i < ntypes * ntypes; i++) { cutforcesq[i] = 0.0; epsilon[i] = 1.0; sigma6[i] = 1.0; sigma[i] = 1.0; }
      if(mmd_create(-1, &h) != 0) {                // no GPU: fail loudly, there is no CPU fallback behind this plugin
        fprintf(stderr, "ForceHIP: %s\n", mmd_last_error());
        exit(1);
      }
    }
    virtual ~ForceHIP()
    {
      mmd_destroy(h);
      delete[] cutforcesq; delete[] epsilon; delete[] sigma6; delete[] sigma;
    }

    void setup()                                     // ForceLJ::setup, ref/force_lj.cpp:65-69
    {
      for(int i = 0; i < ntypes * ntypes; i++) cutforcesq[i] = cutforce * cutforce;
      mmd_force_lj_setup(h, ntypes, cutforcesq, sigma6, epsilon);
    }

    // Force::compute(Atom&, Neighbor&, Comm&, int): called by every OpenMP thread from Integrate::run
    // (ref/integrate.cpp:183, inside the parallel region opened at :90) and once outside it (ref/ljs.cpp:455,478)
    void compute(Atom &atom, Neighbor &neighbor, Comm &, int)
    {
      #pragma omp master
      {
        double e = 0, v = 0;
        if(lists.sync(h, atom, neighbor, ntypes) != 0 || mmd_force_compute(h, evflag, &e, &v) != 0) fprintf(stderr, "ForceHIP: %s\n", mmd_last_error());
        mmd_atom_download(h, 0, 0, atom.f, 0, 0);     // half lists: owned + ghost rows (the reference's Comm::reverse_communicate folds them)
        if(evflag) { eng_vdwl = e; virial = v; }      // same conventions as ref/force_lj.cpp:441-447
      }
      #pragma omp barrier
    }
};

#endif
