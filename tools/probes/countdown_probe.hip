// Probe (round 6): would a per-atom COUNTDOWN in the half-list flush pay for the separate integrator pass it removes?
// Config E (in.lj.miniMD -s 160, SP, half lists + ghost newton on one rank): k_lj_half_tile ends every tile by adding its ~346 LDS accumulators to f[] with one
// 64-bit fixed-point atomic (x, y) + one 32-bit float atomic (z) per accumulator; k_final_initial_integrate then reads f, v, x and writes v, x (68 B/atom, ~0.2 ms).
// Round 5's verdict proposed: count, at the build, how many tiles hold each atom as a partner; let every flush decrement that count and let the thread that brings
// it to zero integrate the atom — no separate pass. The flush already runs at the L2 atomic unit's rate (tools/probes/atomic_scope_probe.hip), so the question is
// what the extra (returning) atomic costs. Variants, same access shape as the flush (runs of consecutive atoms, every atom in ~5.5 tiles, XCD-contiguous tile order):
//   A  today's flush: 64-bit + 32-bit atomic per accumulator                                   (+ the separate integrator pass, timed on its own)
//   B  A + a 32-bit RETURNING countdown atomic per accumulator; the last one integrates the atom (reads the three sums back, v, x; writes v, x)
//   C  z share and countdown packed into ONE 64-bit returning atomic (z as 32-bit fixed point in the low word, the count in the high word): still two atomics per
//      accumulator, one of them returning; the last one integrates
// B and C leave out the release/acquire ordering a real kernel needs between an atom's (x, y) atomic and its countdown (a wait for the first atomic's completion in
// front of the second): they are LOWER bounds on the cost of the idea.
// hipcc -O2 --offload-arch=gfx950 tools/probes/countdown_probe.hip -o /tmp/countdown_probe && /tmp/countdown_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"

struct Args {
  unsigned long long* fxy; float* fz; unsigned long long* fzc; int* cnt;
  float4* x; float* v; float4* xout;
  int n_atoms, per_tile, natom_tile, ntiles;
};

__device__ __forceinline__ long long atom_of(const Args& A, int tile, int e)
{
  long long a = ((long long)tile * A.natom_tile - A.per_tile / 2 + e) % A.n_atoms;
  return a < 0 ? a + A.n_atoms : a;
}
__device__ __forceinline__ int tile_of_block(const Args& A) { const int b = blockIdx.x; return (b & 7) * (A.ntiles >> 3) + (b >> 3); }

__global__ __launch_bounds__(128) void k_count(Args A)      // the build's part: how many tiles hold each atom (not timed)
{
  const int tile = tile_of_block(A);
  for(int e = threadIdx.x; e < A.per_tile; e += 128) atomicAdd(&A.cnt[atom_of(A, tile, e)], 1);
}

__device__ __forceinline__ void integrate(const Args& A, long long a, float fx, float fy, float fz)
{
  const float dtf = 0.0025f, dt = 0.005f;
  float4 p = A.x[a];
  float vx = A.v[3 * a] + dtf * fx, vy = A.v[3 * a + 1] + dtf * fy, vz = A.v[3 * a + 2] + dtf * fz;
  vx += dtf * fx; vy += dtf * fy; vz += dtf * fz;
  A.v[3 * a] = vx; A.v[3 * a + 1] = vy; A.v[3 * a + 2] = vz;
  p.x += dt * vx; p.y += dt * vy; p.z += dt * vz;
  A.xout[a] = p;
}

template <int VAR>
__global__ __launch_bounds__(128) void k_flush(Args A)
{
  const int tile = tile_of_block(A);
  for(int e = threadIdx.x; e < A.per_tile; e += 128) {
    const long long a = atom_of(A, tile, e);
    __hip_atomic_fetch_add(&A.fxy[a], 0x0000000100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(VAR == 0) __hip_atomic_fetch_add(&A.fz[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if(VAR == 1) {
      __hip_atomic_fetch_add(&A.fz[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int old = __hip_atomic_fetch_add(&A.cnt[a], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if(old == 1) {
        const unsigned long long xy = __hip_atomic_load(&A.fxy[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float z = __hip_atomic_load(&A.fz[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        integrate(A, a, (float)(int)(unsigned)xy * 9.5367e-7f, (float)(int)(unsigned)(xy >> 32) * 9.5367e-7f, z);
      }
    } else {
      const unsigned long long old = __hip_atomic_fetch_add(&A.fzc[a], 0xffffffff00000001ull /* count - 1, z + 1 */, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if((unsigned)(old >> 32) == 1u) {
        const unsigned long long xy = __hip_atomic_load(&A.fxy[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        integrate(A, a, (float)(int)(unsigned)xy * 9.5367e-7f, (float)(int)(unsigned)(xy >> 32) * 9.5367e-7f, (float)(int)((unsigned)old + 1u) * 9.5367e-7f);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_integrate(Args A)   // the separate pass of today: f, v, x in; v, x out
{
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  if(a >= A.n_atoms) return;
  const unsigned long long xy = A.fxy[a];
  integrate(A, a, (float)(int)(unsigned)xy * 9.5367e-7f, (float)(int)(unsigned)(xy >> 32) * 9.5367e-7f, A.fz[a]);
}
__global__ __launch_bounds__(256) void k_reset(Args A, int var)
{
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  if(a >= A.n_atoms) return;
  if(var == 2) A.fzc[a] = (unsigned long long)(unsigned)A.cnt[a + A.n_atoms] << 32;
  else if(var == 1) A.cnt[a] = A.cnt[a + A.n_atoms];
}

template <int VAR> float run(Args A)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float total = 0;
  const int reps = 6;
  for(int w = -1; w < reps; w++) {
    k_reset<<<(A.n_atoms + 255) / 256, 256>>>(A, VAR);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_flush<VAR>), dim3(A.ntiles), dim3(128), 0, 0, A);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if(w >= 0) total += ms;
  }
  return total / reps;
}

int main()
{
  for(int n_atoms : {2048000, 16384000}) {
    Args A;
    A.n_atoms = n_atoms; A.natom_tile = 63; A.per_tile = 346; A.ntiles = (n_atoms / A.natom_tile) & ~7;
    hipMalloc(&A.fxy, 8 * (size_t)n_atoms); hipMalloc(&A.fz, 4 * (size_t)n_atoms); hipMalloc(&A.fzc, 8 * (size_t)n_atoms); hipMalloc(&A.cnt, 2 * 4 * (size_t)n_atoms);
    hipMalloc(&A.x, 16 * (size_t)n_atoms); hipMalloc(&A.xout, 16 * (size_t)n_atoms); hipMalloc(&A.v, 12 * (size_t)n_atoms);
    hipMemset(A.fxy, 0, 8 * (size_t)n_atoms); hipMemset(A.fz, 0, 4 * (size_t)n_atoms); hipMemset(A.cnt, 0, 8 * (size_t)n_atoms);
    hipMemset(A.x, 0, 16 * (size_t)n_atoms); hipMemset(A.v, 0, 12 * (size_t)n_atoms);
    hipLaunchKernelGGL(k_count, dim3(A.ntiles), dim3(128), 0, 0, A);
    hipMemcpy(A.cnt + n_atoms, A.cnt, 4 * (size_t)n_atoms, hipMemcpyDeviceToDevice);      // the counts of the build, kept behind the working copy
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for(int w = 0; w < 2; w++) k_integrate<<<(n_atoms + 255) / 256, 256>>>(A);
    hipEventRecord(a);
    for(int w = 0; w < 10; w++) k_integrate<<<(n_atoms + 255) / 256, 256>>>(A);
    hipEventRecord(b); hipEventSynchronize(b);
    float ims; hipEventElapsedTime(&ims, a, b); ims /= 10;
    const float fa = run<0>(A), fb = run<1>(A), fc = run<2>(A);
    printf("%d atoms, %d tiles x %d accumulators\n", n_atoms, A.ntiles, A.per_tile);
    printf("  A  flush (64-bit + 32-bit atomic)                    %.3f ms  + separate integrator pass %.3f ms = %.3f ms\n", fa, ims, fa + ims);
    printf("  B  flush + returning 32-bit countdown, last integrates %.3f ms  (%+.3f ms against A + pass)\n", fb, fb - fa - ims);
    printf("  C  z + countdown in one returning 64-bit atomic        %.3f ms  (%+.3f ms against A + pass)\n", fc, fc - fa - ims);
    hipFree(A.fxy); hipFree(A.fz); hipFree(A.fzc); hipFree(A.cnt); hipFree(A.x); hipFree(A.xout); hipFree(A.v);
  }
  return 0;
}
