// minimd_amd/csrc/force_lj.hip — ForceLJ::compute (ref/force_lj.cpp:72-113) as CDNA4 HIP kernels.
//
// One owned atom per lane; a wavefront walks its 64 wave-interleaved neighbor rows in lock-step:
//   index load  neigh[(w*maxneighs + k)*64 + lane]   -> one coalesced 256 B line per k
//   gather      x[j] (real4: xyz + type)             -> one aligned 32 B (DP) / 16 B (SP) load per pair
// rows are padded with the far-away dummy atom up to the wavefront's longest row, so the loop is
// wave-uniform and branch-free (the cutoff test is a select). Energy/virial are reduced with wavefront
// shuffles, one partial per workgroup, summed in fixed order by k_sum_partials => deterministic.
// This file is compiled WITH FMA contraction; parity against the (uncontracted) oracle is to ~1e-13.
#include "device_utils.hpp"
#include "mmd_internal.hpp"

// 1/r^2: v_rcp_f64 + two Newton steps (<= 1 ulp class) instead of the 11-instruction IEEE divide;
// exact division available through mmd_set_option("exact_div", 1) for verification.
template <bool EXACT>
__device__ __forceinline__ double recip(double a)
{
  if(EXACT) return 1.0 / a;
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(e, r, r);
  e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(e, r, r);
  return r;
}
template <bool EXACT>
__device__ __forceinline__ float recip(float a)
{
  if(EXACT) return 1.0f / a;
  float r = __builtin_amdgcn_rcpf(a);
  const float e = __builtin_fmaf(-a, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

struct LJTables {       // general (non-uniform) case: per type-pair tables staged in LDS by the kernel
  const real* cutforcesq;
  const real* sigma6;
  const real* epsilon;
  int ntypes;
};

#define LJ_MAX_TYPES2 64

// ---- full neighbor list: compute_fullneigh<EVFLAG> (ref/force_lj.cpp:366-449) -------------------------
template <int EV, int UNIFORM, bool EXACT>
__global__ __launch_bounds__(MMD_BLOCK) void k_lj_full(const real4* __restrict__ x, const int* __restrict__ neigh,
                                                       const int* __restrict__ wave_max, int nlocal, int maxneighs,
                                                       LJParams P, LJTables T, real* __restrict__ f,
                                                       double* __restrict__ partials)
{
  __shared__ real s_cut[UNIFORM ? 1 : LJ_MAX_TYPES2], s_s6[UNIFORM ? 1 : LJ_MAX_TYPES2], s_eps[UNIFORM ? 1 : LJ_MAX_TYPES2];
  __shared__ double s_red[16];
  if(!UNIFORM) {
    for(int t = threadIdx.x; t < T.ntypes * T.ntypes; t += blockDim.x) { s_cut[t] = T.cutforcesq[t]; s_s6[t] = T.sigma6[t]; s_eps[t] = T.epsilon[t]; }
    __syncthreads();
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = i >> 6;
  const int lane = threadIdx.x & 63;
  const bool owned = i < nlocal;
  const real4 xi = x[owned ? i : nlocal - 1];
  const int ti = UNIFORM ? 0 : (int)xi.w * T.ntypes;
  const int nwaves = (nlocal + 63) >> 6;
  const int kmax = w < nwaves ? __builtin_amdgcn_readfirstlane(wave_max[w]) : 0;
  const int* __restrict__ np = neigh + ((size_t)w * maxneighs) * 64 + lane;

  real fx = 0, fy = 0, fz = 0;
  double e_acc = 0, v_acc = 0;
  const real c48 = (real)48.0 * P.epsilon;

  for(int k = 0; k < kmax; k += MMD_UNROLL) {
    int j[MMD_UNROLL];
    real4 xj[MMD_UNROLL];
#pragma unroll
    for(int u = 0; u < MMD_UNROLL; u++) j[u] = np[(size_t)(k + u) * 64];
#pragma unroll
    for(int u = 0; u < MMD_UNROLL; u++) xj[u] = x[j[u]];
#pragma unroll
    for(int u = 0; u < MMD_UNROLL; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      real cut, s6, c48e, eps;
      if(UNIFORM) { cut = P.cutforcesq; s6 = P.sigma6; c48e = c48; eps = P.epsilon; }
      else { const int tij = ti + (int)xj[u].w; cut = s_cut[tij]; s6 = s_s6[tij]; eps = s_eps[tij]; c48e = (real)48.0 * eps; }
      const real sr2 = recip<EXACT>(rsq);
      const real sr6 = sr2 * sr2 * sr2 * s6;
      real force = c48e * sr6 * (sr6 - (real)0.5) * sr2;
      const bool in = rsq < cut;
      force = in ? force : (real)0;
      fx += dx * force; fy += dy * force; fz += dz * force;
      if(EV) {
        const real en = in ? sr6 * (sr6 - (real)1.0) * eps : (real)0;
        e_acc += (double)en;
        v_acc += (double)(rsq * force);
      }
    }
  }
  if(owned) { f[3 * (size_t)i + 0] = fx; f[3 * (size_t)i + 1] = fy; f[3 * (size_t)i + 2] = fz; }
  if(EV) {
    const double es = block_sum(e_acc, s_red);
    const double vs = block_sum(v_acc, s_red);
    if(threadIdx.x == 0) { partials[2 * (size_t)blockIdx.x] = es; partials[2 * (size_t)blockIdx.x + 1] = vs; }
  }
}

// ---- half neighbor list: compute_halfneigh_threaded<EVFLAG,GHOST_NEWTON> (ref/force_lj.cpp:271-357) ------
// f was zeroed over owned+ghost atoms beforehand; f_j is scattered with native FP atomics.
template <int EV, int GN, int UNIFORM, bool EXACT>
__global__ __launch_bounds__(MMD_BLOCK) void k_lj_half(const real4* __restrict__ x, const int* __restrict__ neigh,
                                                       const int* __restrict__ wave_max, int nlocal, int maxneighs,
                                                       LJParams P, LJTables T, real* __restrict__ f,
                                                       double* __restrict__ partials)
{
  __shared__ real s_cut[UNIFORM ? 1 : LJ_MAX_TYPES2], s_s6[UNIFORM ? 1 : LJ_MAX_TYPES2], s_eps[UNIFORM ? 1 : LJ_MAX_TYPES2];
  __shared__ double s_red[16];
  if(!UNIFORM) {
    for(int t = threadIdx.x; t < T.ntypes * T.ntypes; t += blockDim.x) { s_cut[t] = T.cutforcesq[t]; s_s6[t] = T.sigma6[t]; s_eps[t] = T.epsilon[t]; }
    __syncthreads();
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = i >> 6;
  const int lane = threadIdx.x & 63;
  const bool owned = i < nlocal;
  const real4 xi = x[owned ? i : nlocal - 1];
  const int ti = UNIFORM ? 0 : (int)xi.w * T.ntypes;
  const int nwaves = (nlocal + 63) >> 6;
  const int kmax = w < nwaves ? __builtin_amdgcn_readfirstlane(wave_max[w]) : 0;
  const int* __restrict__ np = neigh + ((size_t)w * maxneighs) * 64 + lane;

  real fx = 0, fy = 0, fz = 0;
  double e_acc = 0, v_acc = 0;
  for(int k = 0; k < kmax; k += MMD_UNROLL) {
    int j[MMD_UNROLL];
    real4 xj[MMD_UNROLL];
#pragma unroll
    for(int u = 0; u < MMD_UNROLL; u++) j[u] = np[(size_t)(k + u) * 64];
#pragma unroll
    for(int u = 0; u < MMD_UNROLL; u++) xj[u] = x[j[u]];
#pragma unroll
    for(int u = 0; u < MMD_UNROLL; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      real cut, s6, eps;
      if(UNIFORM) { cut = P.cutforcesq; s6 = P.sigma6; eps = P.epsilon; }
      else { const int tij = ti + (int)xj[u].w; cut = s_cut[tij]; s6 = s_s6[tij]; eps = s_eps[tij]; }
      if(rsq < cut) {
        const real sr2 = recip<EXACT>(rsq);
        const real sr6 = sr2 * sr2 * sr2 * s6;
        const real force = (real)48.0 * sr6 * (sr6 - (real)0.5) * sr2 * eps;
        fx += dx * force; fy += dy * force; fz += dz * force;
        const bool mine = GN || j[u] < nlocal;
        if(mine) {
          real* fj = f + 3 * (size_t)j[u];
          unsafeAtomicAdd(fj + 0, -dx * force);
          unsafeAtomicAdd(fj + 1, -dy * force);
          unsafeAtomicAdd(fj + 2, -dz * force);
        }
        if(EV) {
          const real scale = mine ? (real)1.0 : (real)0.5;
          e_acc += (double)(scale * ((real)4.0 * sr6 * (sr6 - (real)1.0)) * eps);
          v_acc += (double)(scale * rsq * force);
        }
      }
    }
  }
  if(owned) {
    real* fi = f + 3 * (size_t)i;
    unsafeAtomicAdd(fi + 0, fx); unsafeAtomicAdd(fi + 1, fy); unsafeAtomicAdd(fi + 2, fz);
  }
  if(EV) {
    const double es = block_sum(e_acc, s_red);
    const double vs = block_sum(v_acc, s_red);
    if(threadIdx.x == 0) { partials[2 * (size_t)blockIdx.x] = es; partials[2 * (size_t)blockIdx.x + 1] = vs; }
  }
}

// fixed-order sum of the per-workgroup partials -> out[0..nval)
__global__ __launch_bounds__(1024) void k_sum_partials(const double* __restrict__ partials, int nblocks, int nval, double* __restrict__ out,
                                                       double scale0, double scale1)
{
  __shared__ double s_red[16];
  for(int c = 0; c < nval; c++) {
    double s = 0;
    for(int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[(size_t)nval * b + c];
    const double t = block_sum(s, s_red);
    if(threadIdx.x == 0) out[c] = t * (c == 0 ? scale0 : scale1);
  }
}

int mmd_zero_forces(mmd_handle* h, int n)
{
  if(n) HIP_TRY(hipMemsetAsync(h->f.p, 0, (size_t)3 * n * sizeof(real), h->stream));
  return 0;
}

extern "C" int mmd_force_lj_setup(mmd_handle* h, int ntypes, const mmd_float* cutforcesq, const mmd_float* sigma6,
                                  const mmd_float* epsilon)
{
  if(!h || ntypes < 1 || !cutforcesq || !sigma6 || !epsilon) { mmd_set_error("mmd_force_lj_setup: bad arguments"); return -1; }
  if(ntypes * ntypes > LJ_MAX_TYPES2) { mmd_set_error("mmd_force_lj_setup: at most 8 atom types are supported"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  h->style = 0;
  h->ntypes = ntypes;
  const int n2 = ntypes * ntypes;
  h->lj_uniform = true;
  for(int i = 1; i < n2; i++)
    if(cutforcesq[i] != cutforcesq[0] || sigma6[i] != sigma6[0] || epsilon[i] != epsilon[0]) h->lj_uniform = false;
  h->lj.cutforcesq = cutforcesq[0]; h->lj.sigma6 = sigma6[0]; h->lj.epsilon = epsilon[0];
  h->h_cutforcesq.assign(cutforcesq, cutforcesq + n2);
  MMD_TRY(h->lj_tables.ensure((size_t)3 * n2, false, h->stream));
  HIP_TRY(hipMemcpyAsync(h->lj_tables.p, cutforcesq, n2 * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->lj_tables.p + n2, sigma6, n2 * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->lj_tables.p + 2 * n2, epsilon, n2 * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

template <int EV, int UNIFORM, bool EXACT>
static void launch_full(mmd_handle* h, int nblocks, const LJTables& T)
{
  hipLaunchKernelGGL((k_lj_full<EV, UNIFORM, EXACT>), dim3(nblocks), dim3(MMD_BLOCK), 0, h->stream, h->x.p, h->neigh.p,
                     h->wave_max.p, h->nlocal, h->maxneighs, h->lj, T, h->f.p, h->partials.p);
}
template <int EV, int GN, int UNIFORM, bool EXACT>
static void launch_half(mmd_handle* h, int nblocks, const LJTables& T)
{
  hipLaunchKernelGGL((k_lj_half<EV, GN, UNIFORM, EXACT>), dim3(nblocks), dim3(MMD_BLOCK), 0, h->stream, h->x.p, h->neigh.p,
                     h->wave_max.p, h->nlocal, h->maxneighs, h->lj, T, h->f.p, h->partials.p);
}

// ForceLJ::compute dispatch (ref/force_lj.cpp:72-113); eng_vdwl/virial (reference conventions) land in
// h->d_result[0..1] when evflag
int mmd_lj_compute(mmd_handle* h, int evflag, double* eng, double* vir)
{
  if(h->neigh_nlocal != h->nlocal) { mmd_set_error("mmd_force_compute: neighbor list is stale (build or upload one first)"); return -1; }
  const int nlocal = h->nlocal;
  const int nblocks = div_up(nlocal, MMD_BLOCK);
  const int n2 = h->ntypes * h->ntypes;
  LJTables T{h->lj_tables.p, h->lj_tables.p + n2, h->lj_tables.p + 2 * n2, h->ntypes};
  MMD_TRY(h->partials.ensure((size_t)2 * nblocks + 8, false, h->stream));
  if(nlocal == 0) { if(eng) *eng = 0; if(vir) *vir = 0; return 0; }
  const int ev = evflag ? 1 : 0, uni = h->lj_uniform ? 1 : 0, ex = h->opt_exact_div ? 1 : 0;
  if(!h->halfneigh) {
#define F(EVv, Uv, Xv) if(ev == EVv && uni == Uv && ex == Xv) launch_full<EVv, Uv, (Xv != 0)>(h, nblocks, T)
    F(0, 0, 0); F(0, 0, 1); F(0, 1, 0); F(0, 1, 1); F(1, 0, 0); F(1, 0, 1); F(1, 1, 0); F(1, 1, 1);
#undef F
  } else {
    MMD_TRY(mmd_zero_forces(h, nlocal + h->nghost));     // ref/force_lj.cpp:286-291
    const int gn = h->ghost_newton ? 1 : 0;
#define H(EVv, Gv, Uv, Xv) if(ev == EVv && gn == Gv && uni == Uv && ex == Xv) launch_half<EVv, Gv, Uv, (Xv != 0)>(h, nblocks, T)
    H(0, 0, 0, 0); H(0, 0, 0, 1); H(0, 0, 1, 0); H(0, 0, 1, 1); H(0, 1, 0, 0); H(0, 1, 0, 1); H(0, 1, 1, 0); H(0, 1, 1, 1);
    H(1, 0, 0, 0); H(1, 0, 0, 1); H(1, 0, 1, 0); H(1, 0, 1, 1); H(1, 1, 0, 0); H(1, 1, 0, 1); H(1, 1, 1, 0); H(1, 1, 1, 1);
#undef H
  }
  HIP_TRY(hipGetLastError());
  if(evflag) {
    // reference conventions: full lists visit both directions, then eng*4 and virial*0.5 (force_lj.cpp:441-442)
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, h->stream, h->partials.p, nblocks, 2, h->d_result,
                       h->halfneigh ? 1.0 : 4.0, h->halfneigh ? 1.0 : 0.5);
    HIP_TRY(hipGetLastError());
    if(eng || vir) {
      HIP_TRY(hipMemcpyAsync(h->h_result, h->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      if(eng) *eng = h->h_result[0];
      if(vir) *vir = h->h_result[1];
    }
  }
  return 0;
}
