// compile-only check of tests/integration/force_hip.h and force_eam_hip.h against the reference headers (see those files); also shows the
// selection site: where ref/ljs.cpp:274-285 does `force = (Force*) new ForceEAM(ntypes)` / `new ForceLJ(ntypes)` a maintainer writes the lines below.
#include <cstdio>
#include "force_hip.h"
#include "force_eam_hip.h"

Force* make_force_hip(int ntypes, MMD_float cutforce)
{
  ForceHIP* f = new ForceHIP(ntypes);
  f->cutforce = cutforce;
  return (Force*)f;
}

Force* make_force_eam_hip(int ntypes)
{
  return (Force*) new ForceEAMHIP(ntypes);
}
