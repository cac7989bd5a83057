// minimd_amd/csrc/device_utils.hpp — wavefront (64-lane) and workgroup reduction / scan idioms for gfx950.
#pragma once
#include <hip/hip_runtime.h>

// sum over the 64 lanes of a wavefront (butterfly through DPP/ds_swizzle-backed __shfl_xor)
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) {
    int t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
// inclusive prefix sum over a wavefront
__device__ __forceinline__ int wave_incl_scan(int v)
{
  const int lane = threadIdx.x & 63;
#pragma unroll
  for(int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if(lane >= o) v += t;
  }
  return v;
}

// workgroup sum of doubles (blockDim multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* lds /* >= 16 doubles */)
{
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if(lane == 0) lds[w] = v;
  __syncthreads();
  double r = 0;
  if(threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for(int i = 0; i < nw; i++) r += lds[i];    // fixed order: deterministic
  }
  __syncthreads();
  return r;
}

// workgroup inclusive scan of ints (blockDim multiple of 64, <= 1024); returns inclusive value, total in *total
__device__ __forceinline__ int block_incl_scan(int v, int* lds /* >= 17 ints */, int* total)
{
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int s = wave_incl_scan(v);
  if(lane == 63) lds[w] = s;
  __syncthreads();
  if(threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    int acc = 0;
    for(int i = 0; i < nw; i++) { int t = lds[i]; lds[i] = acc; acc += t; }
    lds[16] = acc;
  }
  __syncthreads();
  s += lds[w];
  *total = lds[16];
  __syncthreads();
  return s;
}
