// Probe: throughput of global float/double atomic adds shaped like the half-list flush (k_lj_half_tile: every workgroup adds ~350 x 3 accumulators, walked in
// memory order = runs of consecutive addresses; every address is hit by 4-5 workgroups) at AGENT scope (sc1: performed beyond the XCD's L2) against WORKGROUP
// scope (performed in the XCD's own L2 — only legal for addresses no other XCD touches during the kernel).
// hipcc -O2 --offload-arch=gfx950 tools/probes/atomic_scope_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
template <typename T, int SCOPE>
__global__ __launch_bounds__(128) void k_flush(T* __restrict__ f, int n_atoms, int per_tile, int natom_tile, int ntiles)
{
  // XCD-contiguous tile order: workgroup b -> XCD b % 8 -> tile (b % 8) * (ntiles / 8) + b / 8
  const int b = blockIdx.x, tile = (b & 7) * (ntiles >> 3) + (b >> 3);
  // the tile's accumulators: a window of per_tile atoms starting natom_tile * tile - per_tile / 2 (neighbouring tiles overlap ~5x)
  long long first = (long long)tile * natom_tile - per_tile / 2;
  for(int e = threadIdx.x; e < 3 * per_tile; e += 128) {
    long long a = (first + e / 3) % n_atoms; if(a < 0) a += n_atoms;
    T* p = f + 3 * a + e % 3;
    if(SCOPE == 0) __hip_atomic_fetch_add(p, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if(SCOPE == 1) __hip_atomic_fetch_add(p, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else *p += (T)1;        // (plain read-modify-write: the data-path floor, wrong sums)
  }
}
template <typename T, int SCOPE> float run(T* f, int n_atoms, int per_tile, int natom_tile, int ntiles)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipMemset(f, 0, sizeof(T) * 3 * (size_t)n_atoms);
  for(int w = 0; w < 2; w++) hipLaunchKernelGGL((k_flush<T, SCOPE>), dim3(ntiles), dim3(128), 0, 0, f, n_atoms, per_tile, natom_tile, ntiles);
  hipEventRecord(a);
  for(int w = 0; w < 10; w++) hipLaunchKernelGGL((k_flush<T, SCOPE>), dim3(ntiles), dim3(128), 0, 0, f, n_atoms, per_tile, natom_tile, ntiles);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 10;
}
int main()
{
  const int natom_tile = 63, per_tile = 346;
  for(int n_atoms : {2048000, 16384000}) {
    const int ntiles = (n_atoms / natom_tile) & ~7;
    float* ff; double* fd;
    hipMalloc(&ff, sizeof(float) * 3 * (size_t)n_atoms); hipMalloc(&fd, sizeof(double) * 3 * (size_t)n_atoms);
    printf("%d atoms, %d tiles x %d accumulators x 3 (%.0f M atomics per launch)\n", n_atoms, ntiles, per_tile, 3.0e-6 * per_tile * ntiles);
    printf("  float  agent %.3f ms   workgroup %.3f ms   plain rmw %.3f ms\n", run<float, 0>(ff, n_atoms, per_tile, natom_tile, ntiles), run<float, 1>(ff, n_atoms, per_tile, natom_tile, ntiles), run<float, 2>(ff, n_atoms, per_tile, natom_tile, ntiles));
    printf("  double agent %.3f ms   workgroup %.3f ms   plain rmw %.3f ms\n", run<double, 0>(fd, n_atoms, per_tile, natom_tile, ntiles), run<double, 1>(fd, n_atoms, per_tile, natom_tile, ntiles), run<double, 2>(fd, n_atoms, per_tile, natom_tile, ntiles));
    hipFree(ff); hipFree(fd);
  }
  return 0;
}
