// tests/integration/force_eam_hip.h — the EAM plugin of INTEGRATION.md §2: where ref/ljs.cpp:274-283 does
//   force = (Force*) new ForceEAM(ntypes);       a maintainer writes       force = (Force*) new ForceEAMHIP(ntypes);
// ForceEAMHIP IS a ForceEAM: the base class's setup() still reads Cu_u6.eam and builds the spline tables (ref/force_eam.cpp:74-79,
// 505-793) — they are handed to the library as they are (mmd_force_eam_setup takes the reference's arrays and strides) — and the halo of
// fp = F'(rho) between the two sweeps stays the reference's own ForceEAM::communicate on ITS Comm's send lists (ref/force_eam.cpp:851-913),
// called back by the library between its density and force kernels (mmd_force_eam_set_fp_halo). Only compute() is replaced.
// Not part of the product (test infrastructure, linked into oracle/_ref/ref_hip_{dp,sp} by oracle/Makefile).
#ifndef FORCE_EAM_HIP_H_
#define FORCE_EAM_HIP_H_

#include "force_hip.h"                  // mmd.h, HipListCache
#include "force_eam.h"                  // the reference's ForceEAM (ref/force_eam.h)

class ForceEAMHIP : public ForceEAM
{
  // `class ForceEAM : Force` inherits PRIVATELY (ref/force_eam.h:47), so the Force members (evflag, eng_vdwl, cutforcesq, ...) are not
  // accessible by name from here; the program itself reaches them through its Force* (ref/ljs.cpp:274), and so does the plugin
  ::Force* base() { return (::Force*)this; }
  Atom* halo_atom;
  Comm* halo_comm;
  HipListCache lists;

  // ForceEAM::communicate for the library: owned fp in, ghost fp out (fp and communicate are protected members of the base class)
  static int fp_halo(void* ctx, MMD_float* buf, int nlocal, int nghost)
  {
    ForceEAMHIP* self = (ForceEAMHIP*)ctx;
    for(int i = 0; i < nlocal; i++) self->fp[i] = buf[i];
    self->communicate(*self->halo_atom, *self->halo_comm);
    for(int i = nlocal; i < nlocal + nghost; i++) buf[i] = self->fp[i];
    return 0;
  }

  public:
    mmd_handle* h;

    explicit ForceEAMHIP(int ntypes_) : ForceEAM(ntypes_), halo_atom(0), halo_comm(0), h(0)
    {
      if(mmd_create(-1, &h) != 0) {                // no GPU: fail loudly, there is no CPU fallback behind this plugin
        fprintf(stderr, "ForceEAMHIP: %s\n", mmd_last_error());
        exit(1);
      }
    }
    virtual ~ForceEAMHIP() { mmd_destroy(h); }

    void setup()
    {
      ForceEAM::setup();                            // coeff("Cu_u6.eam") + init_style(): rhor/frho/z2r splines, cutforcesq, mass
      ::Force* f = base();
      if(mmd_force_eam_setup(h, f->ntypes, nr, nrho, nr_tot, nrho_tot, rdr, rdrho, rhor_spline, frho_spline, z2r_spline, f->cutforcesq) != 0 ||
         mmd_force_eam_set_fp_halo(h, &ForceEAMHIP::fp_halo, this) != 0)
        fprintf(stderr, "ForceEAMHIP: %s\n", mmd_last_error());
    }

    void compute(Atom &atom, Neighbor &neighbor, Comm &comm, int)
    {
      #pragma omp master
      {
        ::Force* f = base();
        if(atom.nmax > nmax) {                       // fp sized like ref/force_eam.cpp:104-112 (the halo callback works on it)
          nmax = atom.nmax;
          delete[] rho; delete[] fp;
          rho = new MMD_float[nmax]; fp = new MMD_float[nmax];
        }
        halo_atom = &atom; halo_comm = &comm;
        double e = 0, v = 0;
        if(lists.sync(h, atom, neighbor, f->ntypes) != 0 || mmd_force_compute(h, f->evflag, &e, &v) != 0) fprintf(stderr, "ForceEAMHIP: %s\n", mmd_last_error());
        mmd_atom_download(h, 0, 0, atom.f, 0, 0);
        if(f->evflag) { f->eng_vdwl = e; f->virial = v; }      // conventions of ref/force_eam.cpp:268, 440-443
      }
      #pragma omp barrier
    }
};

#endif
