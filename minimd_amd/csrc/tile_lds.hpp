// minimd_amd/csrc/tile_lds.hpp — LDS addressing shared by the tile force kernels (force_lj.hip, force_eam.hip).
// The tile kernels keep the {x,y,z} records of a tile's candidate union at the START of their dynamic LDS segment and
// declare no static __shared__, so the segment begins at LDS address 0 and the 16-bit values of nl16 (slot * 3 reals, in
// bytes) ARE the ds_read addresses of the records: no per-pair base add.
#pragma once
#include "mmd_internal.hpp"

typedef __attribute__((address_space(3))) const real lds_creal;
typedef __attribute__((address_space(3))) const volatile real lds_cvreal;

// read one {x,y,z} record at LDS byte address `a`; RD=1 keeps the three reads separate (volatile: not fused into ds_read2)
template <int RD>
__device__ __forceinline__ void lds_read3(unsigned a, real& qx, real& qy, real& qz)
{
  if(RD == 1) { lds_cvreal* q = (lds_cvreal*)(size_t)a; qx = q[0]; qy = q[1]; qz = q[2]; }
  else        { lds_creal* q = (lds_creal*)(size_t)a; qx = q[0]; qy = q[1]; qz = q[2]; }
}

// explicit fused multiply-add in the build precision: the tile kernels are compiled with -ffp-contract=fast, and a sum of
// products written with * and + would leave the compiler free to choose WHICH product it fuses, differently per template
// instantiation; with fma_r every instantiation rounds alike
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// v += dtf*f ; x += dt*v with separately rounded multiply and add, like integrate.hip (built without contraction).
// The force files are compiled with -ffp-contract=fast, which lets the backend fuse across a `#pragma clang fp contract(off)`
// (seen for float); the empty asm makes the product opaque, so no fma can be formed.
__device__ __forceinline__ real mul_add_unfused(real a, real b, real c)
{
  real p = a * b;
  asm volatile("" : "+v"(p));
  return p + c;
}


// Every load issued so far has landed (s_waitcnt vmcnt(0), as an intrinsic the compiler's wait-count pass sees). Placed in front of a
// pair loop whose body prefetches the next trip's slots: without it the loop carries the "my atom's position may still be in flight"
// state of its pre-header around the back edge, and since that load was issued AFTER the first trip's slot loads the only safe wait
// for it is vmcnt(0) — in every trip, right behind the prefetch it was meant to overlap with.
#ifndef MMD_NO_DRAIN
__device__ __forceinline__ void drain_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), expcnt / lgkmcnt untouched
#else
__device__ __forceinline__ void drain_loads() {}
#endif

// A load of data that is read ONCE per launch (a tile's rows, its candidate list): non-temporal, so that the stream does not push the
// atom positions — which neighbouring tiles re-read — out of the XCD's L2 (LJ at -s 80: rows + lists are 420 MB per launch against 74 MB
// of positions that were fetched three times over; +2.3 % for both LJ tile kernels). NOT for the EAM sweeps: the second sweep re-reads
// the rows of the first (187 MB at -s 64, they survive in the 256 MB MALL), and a non-temporal load does not leave them there (-1 % / -3.6 %).
#ifndef MMD_NO_NT
template <typename T> __device__ __forceinline__ T stream_load(const T* p) { return __builtin_nontemporal_load(p); }
#else
template <typename T> __device__ __forceinline__ T stream_load(const T* p) { return *p; }
#endif

// A store of data the NEXT launch reads (the fused integrator's v and new positions, the forces): every XCD writes its dirty L2 lines back when the kernel
// ends (the XCDs' L2s are not coherent with each other), and that write-back is serial time between two launches — ~10 us behind a launch that leaves
// 100+ MB of output. -DMMD_NT_OUT=1 marks these stores non-temporal (streamed towards memory while the kernel still runs).
#ifndef MMD_NT_OUT
#define MMD_NT_OUT 0
#endif
template <typename T> __device__ __forceinline__ void out_store(T* p, T v)
{
#if MMD_NT_OUT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// One-rank runs: every ghost is a periodic image of an owned atom (Comm::borders recorded its root and image vector), so a tile
// kernel can stage a ghost candidate straight from the owner's CURRENT position plus the box shift — the per-step
// Comm::communicate (k_ghost_update, ref/comm.cpp:276-317 with self swaps) and its launch gap disappear from the step.
// root == nullptr: off (ghost slots of x are read as usual). The shift is applied one box length at a time, exactly like
// the swap-by-swap packing does (pack_comm adds pbc*prd once per swap).
struct GhostResolve { const int* root; const int* image; const int* tile_ghost; real prd[3]; const int* cand_src; };
// cand_src (k_build_rows, one-rank runs): the tile's candidate list once more, a ghost entry replaced by its OWNER and its image code
// (owner | (code + 1) << 25; owned atoms and the dummy: the plain index) — staging a boundary tile then needs no root / image look-up
// in front of the position load (one dependent round trip less than ghost_resolved)
#define MMD_SRC_BITS 25
#define MMD_SRC_MASK ((1 << MMD_SRC_BITS) - 1)
__device__ __forceinline__ real4 ghost_shifted(real4 p, int src, const real* __restrict__ prd)
{
  const int c1 = (int)((unsigned)src >> MMD_SRC_BITS);
  if(c1 != 0) {                                    // (the box lengths are fetched inside the branch: nothing of them lives in registers on the common path)
    const int code = c1 - 1;
    const int sx = code % 5 - 2, sy = (code / 5) % 5 - 2, sz = code / 25 - 2;
    const real bx = prd[0], by = prd[1], bz = prd[2];
    for(int q = 0; q < (sx < 0 ? -sx : sx); q++) p.x += sx < 0 ? -bx : bx;
    for(int q = 0; q < (sy < 0 ? -sy : sy); q++) p.y += sy < 0 ? -by : by;
    for(int q = 0; q < (sz < 0 ? -sz : sz); q++) p.z += sz < 0 ? -bz : bz;
  }
  return p;
}
__device__ __forceinline__ real4 ghost_shifted(real4 p, int src, const GhostResolve& G) { return ghost_shifted(p, src, G.prd); }

__device__ __forceinline__ real4 ghost_resolved(const real4* __restrict__ x, int j, int nlocal, int nall, const GhostResolve& G)
{
  if(j < nlocal || j >= nall) return x[j];
  const int g = j - nlocal;
  real4 p = x[G.root[g]];
  const int code = G.image[g];
  const int sx = code % 5 - 2, sy = (code / 5) % 5 - 2, sz = code / 25 - 2;
  for(int q = 0; q < (sx < 0 ? -sx : sx); q++) p.x += sx < 0 ? -G.prd[0] : G.prd[0];
  for(int q = 0; q < (sy < 0 ? -sy : sy); q++) p.y += sy < 0 ? -G.prd[1] : G.prd[1];
  for(int q = 0; q < (sz < 0 ? -sz : sz); q++) p.z += sz < 0 ? -G.prd[2] : G.prd[2];
  return p;
}
