// oracle/ref_hip_main.cpp — TEST INFRASTRUCTURE ONLY (built into git-ignored oracle/_ref/ref_hip_{dp,sp} and ref_hipnb_dp; never part of the product).
//
// The UNMODIFIED reference program with TWO substitutions: where ref/ljs.cpp:285 constructs `new ForceLJ(ntypes)` it constructs the
// plugin of tests/integration/force_hip.h, and where ref/ljs.cpp:275 constructs `new ForceEAM(ntypes)` the one of force_eam_hip.h
// (Force::setup / Force::compute forwarded to the C-ABI of include/mmd.h). The reference's main() is compiled from where it lies
// (#include of ref/ljs.cpp through -I$(REFERENCE)/ref; nothing is copied), the rest of the program are the reference's own objects: its
// Atom, Neighbor::build, Comm, Thermo — and its Integrate::run calling force->compute(atom, neighbor, comm, me) through the vtable
// (ref/integrate.cpp:183) on the lists ITS Neighbor built. A run of this binary therefore exercises the in-process plugin point of
// SURVEY.md §8(b) for real: rows must equal the reference's.
// With -DREF_HIP_NEIGHBOR_ONLY the forces stay the reference's and only Neighbor::build is the library's (tests/integration/neighbor_hip.cpp,
// linked in place of the reference's definition by the `ref_hipnb` target).
#ifndef REF_HIP_NEIGHBOR_ONLY
#include "../tests/integration/force_hip.h"      // ForceHIP (includes the reference's force.h and include/mmd.h)
#include "../tests/integration/force_eam_hip.h"  // ForceEAMHIP : ForceEAM
#include "force_lj.h"                            // the reference's ForceLJ declaration, before the name is redirected
#define ForceLJ ForceHIP                         // ref/ljs.cpp:285  force = (Force*) new ForceLJ(ntypes);
#define ForceEAM ForceEAMHIP                     // ref/ljs.cpp:275  force = (Force*) new ForceEAM(ntypes);
#endif
#include "ljs.cpp"                               // the reference's main(), from $(REFERENCE)/ref
