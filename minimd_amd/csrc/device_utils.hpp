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
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
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
// wavefront min / max of a float through DPP row shifts + row broadcasts (6 VALU ops, no LDS crossbar):
// the classic GCN/CDNA reduction — the result is valid in lane 63 and returned wave-uniform.
#define MMD_DPP_STEP(OP, ctrl, rmask)                                                                              \
  v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), \
                                                                   ctrl, rmask, 0xf, false)))
__device__ __forceinline__ float wave_min_f(float v)
{
  MMD_DPP_STEP(fminf, 0x111, 0xf);   // row_shr:1
  MMD_DPP_STEP(fminf, 0x112, 0xf);   // row_shr:2
  MMD_DPP_STEP(fminf, 0x114, 0xf);   // row_shr:4
  MMD_DPP_STEP(fminf, 0x118, 0xf);   // row_shr:8
  MMD_DPP_STEP(fminf, 0x142, 0xa);   // row_bcast:15 -> rows 1,3
  MMD_DPP_STEP(fminf, 0x143, 0xc);   // row_bcast:31 -> rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_f(float v)
{
  MMD_DPP_STEP(fmaxf, 0x111, 0xf);
  MMD_DPP_STEP(fmaxf, 0x112, 0xf);
  MMD_DPP_STEP(fmaxf, 0x114, 0xf);
  MMD_DPP_STEP(fmaxf, 0x118, 0xf);
  MMD_DPP_STEP(fmaxf, 0x142, 0xa);
  MMD_DPP_STEP(fmaxf, 0x143, 0xc);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// the same reductions on unsigned keys: integer min/max need no NaN canonicalisation, so every step is ONE
// v_min_u32 / v_max_u32 with the DPP modifier folded in (the float versions cost ~5 instructions per step).
// float_key() is the usual order-preserving map float -> unsigned (flip the sign bit of non-negative values, all
// bits of negative ones); key_float() inverts it.
__device__ __forceinline__ unsigned float_key(float f)
{
  const unsigned b = __builtin_bit_cast(unsigned, f);
  return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k)
{
  return __builtin_bit_cast(float, k ^ ((k & 0x80000000u) ? 0x80000000u : 0xffffffffu));
}
// (lanes a step does not feed receive the operation's identity: that is the form the compiler folds into *_dpp)
#define MMD_DPP_STEP_U(OP, IDENT, ctrl, rmask)                                                                      \
  { const unsigned t_ = (unsigned)__builtin_amdgcn_update_dpp((int)(IDENT), (int)v, ctrl, rmask, 0xf, false); v = OP(v, t_); }
__device__ __forceinline__ unsigned wave_min_u(unsigned v)
{
#define MMD_UMIN(a, b) ((a) < (b) ? (a) : (b))
  MMD_DPP_STEP_U(MMD_UMIN, 0xffffffffu, 0x111, 0xf); MMD_DPP_STEP_U(MMD_UMIN, 0xffffffffu, 0x112, 0xf); MMD_DPP_STEP_U(MMD_UMIN, 0xffffffffu, 0x114, 0xf);
  MMD_DPP_STEP_U(MMD_UMIN, 0xffffffffu, 0x118, 0xf); MMD_DPP_STEP_U(MMD_UMIN, 0xffffffffu, 0x142, 0xa); MMD_DPP_STEP_U(MMD_UMIN, 0xffffffffu, 0x143, 0xc);
#undef MMD_UMIN
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v)
{
#define MMD_UMAX(a, b) ((a) > (b) ? (a) : (b))
  MMD_DPP_STEP_U(MMD_UMAX, 0u, 0x111, 0xf); MMD_DPP_STEP_U(MMD_UMAX, 0u, 0x112, 0xf); MMD_DPP_STEP_U(MMD_UMAX, 0u, 0x114, 0xf);
  MMD_DPP_STEP_U(MMD_UMAX, 0u, 0x118, 0xf); MMD_DPP_STEP_U(MMD_UMAX, 0u, 0x142, 0xa); MMD_DPP_STEP_U(MMD_UMAX, 0u, 0x143, 0xc);
#undef MMD_UMAX
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// bitwise OR over the 64 lanes (same DPP ladder; identity 0)
__device__ __forceinline__ unsigned wave_or_u(unsigned v)
{
#define MMD_UOR(a, b) ((a) | (b))
  MMD_DPP_STEP_U(MMD_UOR, 0u, 0x111, 0xf); MMD_DPP_STEP_U(MMD_UOR, 0u, 0x112, 0xf); MMD_DPP_STEP_U(MMD_UOR, 0u, 0x114, 0xf);
  MMD_DPP_STEP_U(MMD_UOR, 0u, 0x118, 0xf); MMD_DPP_STEP_U(MMD_UOR, 0u, 0x142, 0xa); MMD_DPP_STEP_U(MMD_UOR, 0u, 0x143, 0xc);
#undef MMD_UOR
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
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

// sum of cnt[0..upto) by the whole workgroup (256 threads); every thread gets the result
__device__ __forceinline__ int block_prefix_total(const int* __restrict__ cnt, int upto, int* lds /* >= 17 */)
{
  int v = 0;
  for(int t = threadIdx.x; t < upto; t += 256) v += cnt[t];
  int tot;
  block_incl_scan(v, lds, &tot);
  return tot;
}

// XCD-aware work order. Workgroup b of a launch is observed to run on XCD b % 8 (MI355X_MICROARCH.md, "for speed
// only"); mapping b -> (b % 8) * ceil(n/8) + b / 8 hands every XCD one CONTIGUOUS eighth of a spatially sorted
// work list, so the atoms its workgroups gather stay in that XCD's private 4 MiB L2 instead of being streamed by
// all eight. Launch ceil(n/8)*8 workgroups; the function returns -1 for the padding ones. Results never depend on
// the placement.
__device__ __forceinline__ int xcd_work_item(int n)
{
  const int per = gridDim.x >> 3;
  const int w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  return w < n ? w : -1;
}
// the same for a launch whose grid was sized for MORE than n items (n known on the device only): every XCD still takes a contiguous eighth of n
__device__ __forceinline__ int xcd_work_item_of(int n)
{
  const int per = (n + 7) >> 3;
  const int w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  return ((int)(blockIdx.x >> 3) < per && w < n) ? w : -1;
}
static inline int xcd_grid(int n) { return ((n + 7) / 8) * 8; }

// a count that may still be on its way to the host: `bound` = nlocal + the capacity the arrays were sized for
__device__ __forceinline__ int deferred_count(int bound, int nlocal, const int* __restrict__ nghost_dev)
{
  return nghost_dev ? nlocal + min(*nghost_dev, bound - nlocal) : bound;
}

// broadcast lane `l` (wave-uniform index) of a value to the whole wavefront through v_readlane (SGPR result),
// not through the LDS crossbar
__device__ __forceinline__ double readlane_d(double v, int l)
{
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float readlane_d(float v, int l)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
