// compile-only check of tests/integration/force_hip.h against the reference headers (see that file); also shows the
// selection site: where ref/ljs.cpp:274-285 does `force = (Force*) new ForceLJ(ntypes)` a maintainer writes the line below.
#include <cstdio>
#include "force_hip.h"

Force* make_force_hip(int ntypes, MMD_float cutforce)
{
  ForceHIP* f = new ForceHIP(ntypes);
  f->cutforce = cutforce;
  return (Force*)f;
}
