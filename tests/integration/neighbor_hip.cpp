// tests/integration/neighbor_hip.cpp — the Neighbor plug point of INTEGRATION.md §2: a replacement BODY for the reference's
// `void Neighbor::build(Atom &)` (declared ref/neighbor.h:59, defined ref/neighbor.cpp:79-213, called at ref/ljs.cpp:453 and
// ref/integrate.cpp:174). Neighbor::build is not virtual and `Neighbor neighbor(ntypes)` is a stack object (ref/ljs.cpp:265), so a
// maintainer swaps the definition: oracle/Makefile's `ref_hipnb` compiles ref/neighbor.cpp with -Dbuild=build_reference (its own build
// keeps existing under another name, untouched) and links this file's Neighbor::build instead. Everything else — Neighbor::setup, binatoms
// (Atom::sort uses it), the class layout, the reference's ForceLJ / ForceEAM reading neighbors[i*maxneighs+k] and numneigh[i] — is the
// reference's. Rows equal the reference's as sets (the order inside a row follows the device's candidate order).
// Not part of the product (test infrastructure).
#include <cstdio>
#include <cstdlib>
#include "neighbor.h"                   // the reference's header (ref/neighbor.h)

#ifndef MMD_PRECISION
#define MMD_PRECISION PRECISION
#endif
extern "C" {
#include "mmd.h"
}

static mmd_handle* nb_handle()
{
  static mmd_handle* h = 0;
  if(!h && mmd_create(-1, &h) != 0) {              // no GPU: fail loudly, there is no CPU fallback behind this plugin
    fprintf(stderr, "NeighborHIP: %s\n", mmd_last_error());
    exit(1);
  }
  return h;
}

void Neighbor::build(Atom &atom)
{
  #pragma omp master
  {
    ncalls++;
    mmd_handle* h = nb_handle();
    const int nlocal = atom.nlocal, nall = atom.nlocal + atom.nghost;
    MMD_float prd[3] = {atom.box.xprd, atom.box.yprd, atom.box.zprd};
    MMD_float lo[3] = {atom.box.xlo, atom.box.ylo, atom.box.zlo}, hi[3] = {atom.box.xhi, atom.box.yhi, atom.box.zhi};
    const int nbin[3] = {nbinx, nbiny, nbinz};
    int ok = mmd_atom_set_box(h, prd, lo, hi) == 0 && mmd_atom_upload(h, atom.x, 0, atom.type, 0, atom.nlocal, atom.nghost) == 0 &&
             mmd_neighbor_setup(h, nbin, cutneigh, halfneigh, ghost_newton, atom.ntypes) == 0 && mmd_neighbor_build(h) == 0;
    // the reference's arrays, grown by the reference's rules: nmax follows the atoms (ref/neighbor.cpp:86-104), maxneighs = 1.2 x the longest
    // row when a row does not fit (:186-208). Row lengths first (with ghost newton the rows that cross the boundary follow the
    // reference's own partition of the pairs, whose longest row differs from the device list's), then the rows.
    if(ok && nall > nmax) {
      nmax = nall;
      if(numneigh) free(numneigh);
      if(neighbors) free(neighbors);
      numneigh = (int*) malloc(nmax * sizeof(int));
      neighbors = (int*) malloc((size_t)nmax * maxneighs * sizeof(int));
    }
    if(ok) ok = mmd_neighbor_download(h, 0, 0, numneigh) == 0;
    int max_row = 0;
    for(int i = 0; ok && i < nlocal; i++) if(numneigh[i] > max_row) max_row = numneigh[i];
    if(ok && max_row >= maxneighs) {
      maxneighs = (int)(max_row * 1.2);
      free(neighbors);
      neighbors = (int*) malloc((size_t)nmax * maxneighs * sizeof(int));
    }
    if(ok) ok = mmd_neighbor_download(h, neighbors, maxneighs, numneigh) == 0;
    if(!ok) { fprintf(stderr, "NeighborHIP: %s\n", mmd_last_error()); exit(1); }
  }
  #pragma omp barrier
}
