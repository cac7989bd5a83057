// minimd_amd/csrc/main.cpp — the `miniMD` executable: drop-in for ref/ljs.cpp's main (same CLI, decks, stdout).
#include <cstdio>

#include "../../include/mmd.h"

int main(int argc, char** argv)
{
  mmd_sim* sim = nullptr;
  const int rc = mmd_sim_create(argc, argv, 0, &sim);
  if(rc != 0) return 0;                 // errors were printed; the reference also exits with status 0
  if(mmd_sim_initial(sim) < 0 || mmd_sim_run(sim) < 0) {
    printf("ERROR: %s\n", mmd_last_error());
    mmd_sim_destroy(sim);
    return 1;
  }
  mmd_sim_print_perf(sim);
  int screen = 0;
  if(mmd_sim_wants_yaml(sim, &screen)) mmd_sim_output(sim, screen);          // ref/ljs.cpp:497-498
  mmd_sim_destroy(sim);
  return 0;
}
