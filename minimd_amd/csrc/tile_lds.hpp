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

