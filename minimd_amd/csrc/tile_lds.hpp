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
