// minimd_amd/csrc/force_eam.hip — ForceEAM::compute_fullneigh (ref/force_eam.cpp:274-449) as two HIP sweeps
// over the wave-interleaved neighbor rows, with the derivative-of-embedding halo (ForceEAM::communicate,
// ref/force_eam.cpp:851-913) between them.
//
// (production: the tile forms k_eam_density_tile / k_eam_force_tile further down; the kernels on wave-interleaved rows
//  are the fallback for uploaded lists and non-uniform tables)
//   sweep 1  k_eam_density : rho_i = sum_j rho(r_ij) (cubic spline), fp_i = F'(rho_i), [EV] E_embed
//   halo     fp of owned atoms -> ghosts (same send lists as Comm::communicate, 1 scalar per atom)
//   sweep 2  k_eam_force   : f_i = -sum_j (fp_i rho' + fp_j rho' + phi') / r * del ; [EV] phi/2, virial
//
// The spline knots used inside the pair loops are re-packed per sweep and staged in LDS
// (sweep 1: 4 coeffs/knot = 16 KB DP; sweep 2: 3 rho' + 7 z2r coeffs padded to 12 = 48 KB DP), since every
// lane looks up a different knot each iteration; the per-atom embedding lookup reads HBM/L2 directly.
// All type pairs share one table in miniMD (ref/force_eam.cpp:753-760) — verified at setup; otherwise the
// general kernels index the per-pair tables in global memory.
#include <type_traits>
#include "device_utils.hpp"
#include "mmd_internal.hpp"
#include "tile_lds.hpp"

#define EAM_MAX_KNOTS 1024
#define EAM_UNR 4

// ---- sweep 1 --------------------------------------------------------------------------------------
template <int EV, int UNIFORM>
__global__ __launch_bounds__(MMD_BLOCK) void k_eam_density(const real4* __restrict__ x, const int* __restrict__ neigh,
                                                           const int* __restrict__ wave_max, int nlocal, int maxneighs,
                                                           const real* __restrict__ rhor_spline, const real* __restrict__ frho_spline,
                                                           const real* __restrict__ cutforcesq, int ntypes, int nr, int nrho,
                                                           int nr_tot, int nrho_tot, real rdr, real rdrho,
                                                           real* __restrict__ fp, double* __restrict__ partials)
{
  extern __shared__ __align__(16) unsigned char s_raw[];
  real* s_tab = (real*)s_raw;                           // [knot][4] : coeffs 3..6 of rhor_spline (UNIFORM only)
  __shared__ double s_red[16];
  if(UNIFORM) {
    for(int t = threadIdx.x; t < (nr + 1) * 4; t += blockDim.x) s_tab[t] = rhor_spline[(t >> 2) * 7 + 3 + (t & 3)];
    __syncthreads();
  }
  const int wg = xcd_work_item((nlocal + MMD_BLOCK - 1) / MMD_BLOCK);
  if(wg < 0) return;
  const int i = wg * MMD_BLOCK + threadIdx.x;
  const int w = i >> 6, lane = threadIdx.x & 63;
  const bool owned = i < nlocal;
  const real4 xi = x[owned ? i : nlocal - 1];
  const int ti = (int)xi.w;
  const int nwaves = (nlocal + 63) >> 6;
  const int kmax = w < nwaves ? __builtin_amdgcn_readfirstlane(wave_max[w]) : 0;
  const int* __restrict__ np = neigh + ((size_t)w * maxneighs) * 64 + lane;
  const real cut0 = cutforcesq[0];
  real rhoi = 0;
  for(int k = 0; k < kmax; k += EAM_UNR) {
    int j[EAM_UNR];
    real4 xj[EAM_UNR];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) j[u] = np[(size_t)(k + u) * 64];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) xj[u] = x[j[u]];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = UNIFORM ? 0 : ti * ntypes + (int)xj[u].w;
      const real cut = UNIFORM ? cut0 : cutforcesq[tij];
      if(rsq < cut) {
        real p = sqrt(rsq) * rdr + (real)1.0;
        int m = (int)p;
        m = m < nr - 1 ? m : nr - 1;
        p -= m;
        p = p < (real)1.0 ? p : (real)1.0;
        real c3, c4, c5, c6;
        if(UNIFORM) { const real* c = &s_tab[m * 4]; c3 = c[0]; c4 = c[1]; c5 = c[2]; c6 = c[3]; }
        else { const real* c = &rhor_spline[(size_t)tij * nr_tot + m * 7]; c3 = c[3]; c4 = c[4]; c5 = c[5]; c6 = c[6]; }
        rhoi += ((c3 * p + c4) * p + c5) * p + c6;
      }
    }
  }
  double e_acc = 0;
  if(owned) {
    const int tii = UNIFORM ? 0 : ti * ti;              // sic (ref/force_eam.cpp:337)
    real p = (real)1.0 * rhoi * rdrho + (real)1.0;
    int m = (int)p;
    m = max(1, min(m, nrho - 1));
    p -= m;
    p = p < (real)1.0 ? p : (real)1.0;
    const real* c = &frho_spline[(size_t)tii * nrho_tot + m * 7];
    fp[i] = (c[0] * p + c[1]) * p + c[2];
    if(EV) e_acc = (double)(((c[3] * p + c[4]) * p + c[5]) * p + c[6]);
  }
  if(EV) {
    const double es = block_sum(e_acc, s_red);
    if(threadIdx.x == 0) partials[3 * (size_t)wg] = es;
  }
}

// ---- sweep 2 --------------------------------------------------------------------------------------
template <int EV, int UNIFORM>
__global__ __launch_bounds__(MMD_BLOCK) void k_eam_force(const real4* __restrict__ x, const int* __restrict__ neigh,
                                                         const int* __restrict__ wave_max, int nlocal, int maxneighs,
                                                         const real* __restrict__ rhor_spline, const real* __restrict__ z2r_spline,
                                                         const real* __restrict__ cutforcesq, int ntypes, int nr, int nr_tot, real rdr,
                                                         const real* __restrict__ fp, real* __restrict__ f, double* __restrict__ partials)
{
  extern __shared__ __align__(16) unsigned char s_raw[];
  real* s_tab = (real*)s_raw;                           // [knot][12]: rhor 0..2, z2r 0..6, pad (UNIFORM only)
  __shared__ double s_red[16];
  if(UNIFORM) {
    for(int t = threadIdx.x; t < (nr + 1) * 12; t += blockDim.x) {
      const int m = t / 12, c = t % 12;
      s_tab[t] = c < 3 ? rhor_spline[m * 7 + c] : (c < 10 ? z2r_spline[m * 7 + (c - 3)] : (real)0);
    }
    __syncthreads();
  }
  const int wg = xcd_work_item((nlocal + MMD_BLOCK - 1) / MMD_BLOCK);
  if(wg < 0) return;
  const int i = wg * MMD_BLOCK + threadIdx.x;
  const int w = i >> 6, lane = threadIdx.x & 63;
  const bool owned = i < nlocal;
  const real4 xi = x[owned ? i : nlocal - 1];
  const real fpi = fp[owned ? i : nlocal - 1];
  const int ti = (int)xi.w;
  const int nwaves = (nlocal + 63) >> 6;
  const int kmax = w < nwaves ? __builtin_amdgcn_readfirstlane(wave_max[w]) : 0;
  const int* __restrict__ np = neigh + ((size_t)w * maxneighs) * 64 + lane;
  const real cut0 = cutforcesq[0];
  real fx = 0, fy = 0, fz = 0;
  double e_acc = 0, v_acc = 0;
  for(int k = 0; k < kmax; k += EAM_UNR) {
    int j[EAM_UNR];
    real4 xj[EAM_UNR];
    real fpj[EAM_UNR];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) j[u] = np[(size_t)(k + u) * 64];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) { xj[u] = x[j[u]]; fpj[u] = fp[j[u]]; }
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = UNIFORM ? 0 : ti * ntypes + (int)xj[u].w;
      const real cut = UNIFORM ? cut0 : cutforcesq[tij];
      if(rsq < cut) {
        const real r = sqrt(rsq);
        real p = r * rdr + (real)1.0;
        int m = (int)p;
        m = m < nr - 1 ? m : nr - 1;
        p -= m;
        p = p < (real)1.0 ? p : (real)1.0;
        real r0, r1, r2, z0, z1, z2c, z3, z4, z5, z6;
        if(UNIFORM) {
          const real* c = &s_tab[m * 12];
          r0 = c[0]; r1 = c[1]; r2 = c[2]; z0 = c[3]; z1 = c[4]; z2c = c[5]; z3 = c[6]; z4 = c[7]; z5 = c[8]; z6 = c[9];
        } else {
          const real* cr = &rhor_spline[(size_t)tij * nr_tot + m * 7];
          const real* cz = &z2r_spline[(size_t)tij * nr_tot + m * 7];
          r0 = cr[0]; r1 = cr[1]; r2 = cr[2]; z0 = cz[0]; z1 = cz[1]; z2c = cz[2]; z3 = cz[3]; z4 = cz[4]; z5 = cz[5]; z6 = cz[6];
        }
        const real rhoip = (r0 * p + r1) * p + r2;
        const real z2p = (z0 * p + z1) * p + z2c;
        const real z2 = ((z3 * p + z4) * p + z5) * p + z6;
        const real recip = (real)1.0 / r;
        const real phi = z2 * recip;
        const real phip = z2p * recip - phi * recip;
        const real psip = fpi * rhoip + fpj[u] * rhoip + phip;
        real fpair = -psip * recip;
        fx += dx * fpair; fy += dy * fpair; fz += dz * fpair;
        if(EV) {
          fpair *= (real)0.5;
          v_acc += (double)(dx * dx * fpair + dy * dy * fpair + dz * dz * fpair);
          e_acc += (double)((real)0.5 * phi);
        }
      }
    }
  }
  if(owned) { f[3 * (size_t)i + 0] = fx; f[3 * (size_t)i + 1] = fy; f[3 * (size_t)i + 2] = fz; }
  if(EV) {
    const double es = block_sum(e_acc, s_red);
    const double vs = block_sum(v_acc, s_red);
    if(threadIdx.x == 0) { partials[3 * (size_t)wg + 1] = es; partials[3 * (size_t)wg + 2] = vs; }
  }
}


// ---------------------------------------------------------------------------------------------------
// Half neighbor lists: ForceEAM::compute_halfneigh (ref/force_eam.cpp:94-270). Every stored pair (i, j) is visited once:
//   sweep 1  k_eam_half_density : rho_i += rho(r) in registers, rho_j += rho(r) by atomic when j is owned (:152-156)
//   embed    k_eam_half_fp      : fp_i = F'(rho_i), [EV] E_embed                                        (:165-179)
//   halo     fp of owned atoms -> ghosts
//   sweep 2  k_eam_half_force   : f_i += pair force in registers, f_j -= by atomics when j is owned; a ghost partner gets no
//                                 force and the pair counts half in energy and virial              (:244-257)
// The list is the reference's half list without ghost newton (owned j > i, every ghost). rho and f are zeroed beforehand
// (f over owned + ghost atoms like :112-116). Floating-point atomics make the summation order run-dependent.
// ---------------------------------------------------------------------------------------------------
template <int UNIFORM>
__global__ __launch_bounds__(MMD_BLOCK) void k_eam_half_density(const real4* __restrict__ x, const int* __restrict__ neigh,
                                                                const int* __restrict__ wave_max, int nlocal, int maxneighs,
                                                                const real* __restrict__ rhor_spline, const real* __restrict__ cutforcesq,
                                                                int ntypes, int nr, int nr_tot, real rdr, real* __restrict__ rho)
{
  extern __shared__ __align__(16) unsigned char s_raw[];
  real* s_tab = (real*)s_raw;                           // [knot][4] : coeffs 3..6 of rhor_spline (UNIFORM only)
  if(UNIFORM) {
    for(int t = threadIdx.x; t < (nr + 1) * 4; t += blockDim.x) s_tab[t] = rhor_spline[(t >> 2) * 7 + 3 + (t & 3)];
    __syncthreads();
  }
  const int wg = xcd_work_item((nlocal + MMD_BLOCK - 1) / MMD_BLOCK);
  if(wg < 0) return;
  const int i = wg * MMD_BLOCK + threadIdx.x;
  const int w = i >> 6, lane = threadIdx.x & 63;
  const bool owned = i < nlocal;
  const real4 xi = x[owned ? i : nlocal - 1];
  const int ti = (int)xi.w;
  const int nwaves = (nlocal + 63) >> 6;
  const int kmax = w < nwaves ? __builtin_amdgcn_readfirstlane(wave_max[w]) : 0;
  const int* __restrict__ np = neigh + ((size_t)w * maxneighs) * 64 + lane;
  const real cut0 = cutforcesq[0];
  real rhoi = 0;
  for(int k = 0; k < kmax; k += EAM_UNR) {
    int j[EAM_UNR];
    real4 xj[EAM_UNR];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) j[u] = np[(size_t)(k + u) * 64];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) xj[u] = x[j[u]];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = UNIFORM ? 0 : ti * ntypes + (int)xj[u].w;
      const real cut = UNIFORM ? cut0 : cutforcesq[tij];
      if(rsq < cut) {
        real p = sqrt(rsq) * rdr + (real)1.0;
        int m = (int)p;
        m = m < nr - 1 ? m : nr - 1;
        p -= m;
        p = p < (real)1.0 ? p : (real)1.0;
        real c3, c4, c5, c6;
        if(UNIFORM) { const real* c = &s_tab[m * 4]; c3 = c[0]; c4 = c[1]; c5 = c[2]; c6 = c[3]; }
        else { const real* c = &rhor_spline[(size_t)tij * nr_tot + m * 7]; c3 = c[3]; c4 = c[4]; c5 = c[5]; c6 = c[6]; }
        const real term = ((c3 * p + c4) * p + c5) * p + c6;
        rhoi += term;
        if(j[u] < nlocal) unsafeAtomicAdd(rho + j[u], term);
      }
    }
  }
  if(owned) unsafeAtomicAdd(rho + i, rhoi);
}

template <int EV>
__global__ __launch_bounds__(MMD_BLOCK) void k_eam_half_fp(const real4* __restrict__ x, const real* __restrict__ rho, int nlocal,
                                                           const real* __restrict__ frho_spline, int uniform, int nrho, int nrho_tot,
                                                           real rdrho, real* __restrict__ fp, double* __restrict__ partials)
{
  __shared__ double s_red[16];
  const int i = blockIdx.x * MMD_BLOCK + threadIdx.x;
  double e_acc = 0;
  if(i < nlocal) {
    const int ti = (int)x[i].w;
    const int tii = uniform ? 0 : ti * ti;                // sic (ref/force_eam.cpp:168)
    real p = (real)1.0 * rho[i] * rdrho + (real)1.0;
    int m = (int)p;
    m = max(1, min(m, nrho - 1));
    p -= m;
    p = p < (real)1.0 ? p : (real)1.0;
    const real* c = &frho_spline[(size_t)tii * nrho_tot + m * 7];
    fp[i] = (c[0] * p + c[1]) * p + c[2];
    if(EV) e_acc = (double)(((c[3] * p + c[4]) * p + c[5]) * p + c[6]);
  }
  if(EV) {
    const double es = block_sum(e_acc, s_red);
    if(threadIdx.x == 0) partials[blockIdx.x] = es;
  }
}

template <int EV, int UNIFORM>
__global__ __launch_bounds__(MMD_BLOCK) void k_eam_half_force(const real4* __restrict__ x, const int* __restrict__ neigh,
                                                              const int* __restrict__ wave_max, int nlocal, int maxneighs,
                                                              const real* __restrict__ rhor_spline, const real* __restrict__ z2r_spline,
                                                              const real* __restrict__ cutforcesq, int ntypes, int nr, int nr_tot, real rdr,
                                                              const real* __restrict__ fp, real* __restrict__ f, double* __restrict__ partials)
{
  extern __shared__ __align__(16) unsigned char s_raw[];
  real* s_tab = (real*)s_raw;                           // [knot][12]: rhor 0..2, z2r 0..6, pad (UNIFORM only)
  __shared__ double s_red[16];
  if(UNIFORM) {
    for(int t = threadIdx.x; t < (nr + 1) * 12; t += blockDim.x) {
      const int m = t / 12, c = t % 12;
      s_tab[t] = c < 3 ? rhor_spline[m * 7 + c] : (c < 10 ? z2r_spline[m * 7 + (c - 3)] : (real)0);
    }
    __syncthreads();
  }
  const int wg = xcd_work_item((nlocal + MMD_BLOCK - 1) / MMD_BLOCK);
  if(wg < 0) return;
  const int i = wg * MMD_BLOCK + threadIdx.x;
  const int w = i >> 6, lane = threadIdx.x & 63;
  const bool owned = i < nlocal;
  const real4 xi = x[owned ? i : nlocal - 1];
  const real fpi = fp[owned ? i : nlocal - 1];
  const int ti = (int)xi.w;
  const int nwaves = (nlocal + 63) >> 6;
  const int kmax = w < nwaves ? __builtin_amdgcn_readfirstlane(wave_max[w]) : 0;
  const int* __restrict__ np = neigh + ((size_t)w * maxneighs) * 64 + lane;
  const real cut0 = cutforcesq[0];
  real fx = 0, fy = 0, fz = 0;
  double e_acc = 0, v_acc = 0;
  for(int k = 0; k < kmax; k += EAM_UNR) {
    int j[EAM_UNR];
    real4 xj[EAM_UNR];
    real fpj[EAM_UNR];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) j[u] = np[(size_t)(k + u) * 64];
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) { xj[u] = x[j[u]]; fpj[u] = fp[j[u]]; }
#pragma unroll
    for(int u = 0; u < EAM_UNR; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      const int tij = UNIFORM ? 0 : ti * ntypes + (int)xj[u].w;
      const real cut = UNIFORM ? cut0 : cutforcesq[tij];
      if(rsq < cut) {
        const real r = sqrt(rsq);
        real p = r * rdr + (real)1.0;
        int m = (int)p;
        m = m < nr - 1 ? m : nr - 1;
        p -= m;
        p = p < (real)1.0 ? p : (real)1.0;
        real r0, r1, r2, z0, z1, z2c, z3, z4, z5, z6;
        if(UNIFORM) {
          const real* c = &s_tab[m * 12];
          r0 = c[0]; r1 = c[1]; r2 = c[2]; z0 = c[3]; z1 = c[4]; z2c = c[5]; z3 = c[6]; z4 = c[7]; z5 = c[8]; z6 = c[9];
        } else {
          const real* cr = &rhor_spline[(size_t)tij * nr_tot + m * 7];
          const real* cz = &z2r_spline[(size_t)tij * nr_tot + m * 7];
          r0 = cr[0]; r1 = cr[1]; r2 = cr[2]; z0 = cz[0]; z1 = cz[1]; z2c = cz[2]; z3 = cz[3]; z4 = cz[4]; z5 = cz[5]; z6 = cz[6];
        }
        const real rhoip = (r0 * p + r1) * p + r2;
        const real z2p = (z0 * p + z1) * p + z2c;
        const real z2 = ((z3 * p + z4) * p + z5) * p + z6;
        const real recip = (real)1.0 / r;
        const real phi = z2 * recip;
        const real phip = z2p * recip - phi * recip;
        const real psip = fpi * rhoip + fpj[u] * rhoip + phip;
        real fpair = -psip * recip;
        fx += dx * fpair; fy += dy * fpair; fz += dz * fpair;
        const bool jown = j[u] < nlocal;
        if(jown) {
          real* fj = f + 3 * (size_t)j[u];
          unsafeAtomicAdd(fj + 0, -dx * fpair); unsafeAtomicAdd(fj + 1, -dy * fpair); unsafeAtomicAdd(fj + 2, -dz * fpair);
        } else fpair *= (real)0.5;
        if(EV) {
          v_acc += (double)(dx * dx * fpair + dy * dy * fpair + dz * dz * fpair);
          e_acc += (double)(jown ? phi : (real)0.5 * phi);
        }
      }
    }
  }
  if(owned) { real* fi = f + 3 * (size_t)i; unsafeAtomicAdd(fi + 0, fx); unsafeAtomicAdd(fi + 1, fy); unsafeAtomicAdd(fi + 2, fz); }
  if(EV) {
    const double es = block_sum(e_acc, s_red);
    const double vs = block_sum(v_acc, s_red);
    if(threadIdx.x == 0) { partials[2 * (size_t)wg] = es; partials[2 * (size_t)wg + 1] = vs; }
  }
}

// eng_vdwl = E_embed + sum phi (ref/force_eam.cpp:269), virial = sum
__global__ __launch_bounds__(1024) void k_eam_half_sum(const double* __restrict__ embed, int nb_embed, const double* __restrict__ pair, int nb_pair,
                                                       double* __restrict__ out)
{
  __shared__ double s_red[16];
  double a = 0, b = 0, c = 0;
  for(int k = threadIdx.x; k < nb_embed; k += blockDim.x) a += embed[k];
  for(int k = threadIdx.x; k < nb_pair; k += blockDim.x) { b += pair[2 * (size_t)k]; c += pair[2 * (size_t)k + 1]; }
  const double ta = block_sum(a, s_red), tb = block_sum(b, s_red), tc = block_sum(c, s_red);
  if(threadIdx.x == 0) { out[0] = ta + tb; out[1] = tc; }
}

// 1/sqrt(a) for a in the pair range (0.1 .. 100): v_rsq + one Newton step with the second-order term
// (y(1 + e/2 + 3e^2/8), e = 1 - a y^2) => ~1 ulp, instead of the ~25-instruction IEEE sqrt and the ~13-instruction
// IEEE divide the compiler expands sqrt(rsq) and 1.0/r into; r = a * rsqrt(a).
__device__ __forceinline__ double rsqrt_fast(double a)
{
  const double y = __builtin_amdgcn_rsq(a);
  const double e = __builtin_fma(-(a * y), y, 1.0);
  return __builtin_fma(y, e * __builtin_fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ float rsqrt_fast(float a)
{
  const float y = __builtin_amdgcn_rsqf(a);
  const float e = __builtin_fmaf(-(a * y), y, 1.0f);
  return __builtin_fmaf(y, e * __builtin_fmaf(0.375f, e, 0.5f), y);
}

// ---------------------------------------------------------------------------------------------------
// Tile forms of the two sweeps (same data structure as k_lj_full_tile, force_lj.hip): the tile's candidate
// union is staged in LDS ({x,y,z} records, and fp for sweep 2) next to the re-packed spline knots; every pair
// gathers from LDS only. EAM_TW wavefronts per tile split the k range of the same 64 atoms.
// ---------------------------------------------------------------------------------------------------
#define EAM_TW 4
#ifndef EAM_FW
#define EAM_FW 4              // wavefronts per tile in the force sweep
#endif
#define EAM_TU 4              // row padding granularity of the tile lists (NB_ROW_PAD) = pairs per trip of the density sweep
#define EAM_DSTRIDE 6         // reals per knot record of the density sweep's LDS table
#ifndef EAM_FSTRIDE
#define EAM_FSTRIDE 10        // reals per knot record of the force sweep's LDS table (7 are used; 7 = packed records, read as ds_read_b64: tuning variant)
#endif
#ifndef EAM_LDS_BUDGET
#define EAM_LDS_BUDGET (150 * 1024)     // LDS of a CU the persistent grids are sized for
#endif
#ifndef EAM_FU
#define EAM_FU 4              // pairs per trip of the force sweep (+ one trip of EAM_TU where 4 rows remain)
#endif
#ifndef EAM_STAGE
#define EAM_STAGE 2              // candidates per thread and staging round: unions hold 300-500 candidates, a workgroup 256 threads (4: two all-dummy loads per thread and tile, +7 %)
#endif
#ifndef EAM_FWAVES
#define EAM_FWAVES 3            // wavefronts per SIMD the force sweep is compiled for (<= 168 VGPRs): its DP variants need 162-182 registers, and the knot table lets
#endif                        // 3 workgroups = 3 wavefronts per SIMD onto a CU anyway — a variant that ends at 170 would run with 2 (measured: 0.29 -> 0.37 ms per step)
// rows / candidate list in the force sweep of full lists: non-temporal (their last use in the step; the density sweep before it loads them
// normally so that they wait in the MALL). +1.2 %; half lists -1.6 %, so not there.
#define EAM_F_LOAD(p) (HALF ? *(p) : stream_load(p))
#ifndef EAM_RD
#define EAM_RD (MMD_PRECISION == 2 ? 1 : 0)      // DP: the three position reads of a pair stay separate ds_read_b64 (ds_read2_b64 runs at half the LDS rate: -1.7 %)
#endif

// dynamic LDS of both kernels (nothing static precedes it, see tile_lds.hpp):
//   [{x,y,z} records of the candidates: eam_pos_bytes(cmax)] [force sweep only: fp of the candidates] [spline knots] [partials] [16 doubles]
__host__ __device__ constexpr size_t eam_pos_bytes(int cmax) { return (((size_t)3 * (cmax + 2) * sizeof(real)) + 15) & ~(size_t)15; }
__host__ __device__ constexpr size_t eam_fp_bytes(int cmax) { return (((size_t)(cmax + 2) * sizeof(real)) + 15) & ~(size_t)15; }
// half lists: n double accumulators per candidate (1: rho, 3: f) — doubles in both precisions, see k_lj_half_tile
__host__ __device__ constexpr size_t eam_acc_bytes(int cmax, int n) { return (((size_t)n * (cmax + 2) * sizeof(double)) + 15) & ~(size_t)15; }

// Rows in two parts (CoreRows, mmd_internal.hpp): which part a launch walks, and the displacement tracking of the fused integrator
struct EamCore {
  const int* tile_kcore;        // nullptr: whole rows always
  int mode;                     // 0 whole rows, 1 core part, 2 core part if the displacement read from words_read allows it
  const unsigned* words_read;   // 64 words: float bits of the largest squared displacement since the build (written by the previous launch)
  unsigned* words_write;        // FUSE launches: this launch's set / the set to clear for the next launch
  unsigned* words_zero;
  float thr_d2;                 // (margin / 2)^2
  const real4* xbuild;          // positions at the build
  int ablate;                   // profiling only (results invalid): 1 no staging, 2 no pair loop
};
__device__ __forceinline__ bool eam_use_core(const EamCore& C, int lane)
{
  if(C.tile_kcore == nullptr || C.mode == 0) return false;
  if(C.mode == 1) return true;
  const unsigned w = wave_max_u(C.words_read[lane & 63]);
  return __uint_as_float(w) <= C.thr_d2;
}

// HALF=1: ForceEAM::compute_halfneigh's first loop (ref/force_eam.cpp:131-160) on a half-list tile: a pair's term goes to the atom
// in registers and to its PARTNER through an LDS accumulator per candidate (ds_add_f64); at the end of the tile the accumulators
// of the owned candidates are flushed to rho[] with one global atomic each, in memory order (whole lines per wave instruction,
// see k_lj_half_tile). rho was zeroed beforehand; fp = F'(rho) is a separate pass (k_eam_half_fp) once every tile has flushed.
template <int EV, int HALF, int SRC = 0>
__global__ __launch_bounds__(64 * EAM_TW) void k_eam_density_tile(
    const real4* __restrict__ x, const int* __restrict__ binned, const int* __restrict__ tile_first, const int* __restrict__ tile_cnt,
    const int* __restrict__ tile_max, const int* __restrict__ tile_cand, const int* __restrict__ tile_ncand, int cstride, int ntiles,
    const int* __restrict__ tile_list,
    const unsigned short* __restrict__ nl16, int nlocal, int nall, int maxneighs, const real* __restrict__ rhor_spline,
    const real* __restrict__ frho_spline, real cutforcesq, int nr, int nrho, int cmax, real rdr, real rdrho, real* __restrict__ fp,
    double* __restrict__ partials, int mlo, const unsigned short* __restrict__ tile_self, real* __restrict__ rho, EamCore C,
    const int* __restrict__ cand_src, const real* __restrict__ box_dev, const int* __restrict__ tile_ghost)
{
  extern __shared__ __align__(16) unsigned char s_raw[];
  constexpr int NT = 64 * EAM_TW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: the k loop runs on the scalar unit
  real* s_pos = (real*)s_raw;
  double* s_racc = (double*)(s_raw + eam_pos_bytes(cmax));           // HALF: one density accumulator per candidate
  // knot window [mlo, nr]: pairs closer than r(mlo) (never seen in a sane system) read their knot from global memory
  // records of EAM_DSTRIDE reals (48 B): a stride of 12 words spreads random knots over 16 bank offsets, 32-byte records over 8
  real* s_tab = (real*)(s_raw + eam_pos_bytes(cmax) + (HALF ? eam_acc_bytes(cmax, 1) : 0));     // [knot - mlo][EAM_DSTRIDE]: coeffs 3..6 of rhor_spline, 2 unused
  real* s_part = s_tab + (size_t)(nr + 1 - mlo) * EAM_DSTRIDE;
  double* s_red = (double*)(((size_t)(s_part + 64 * (EAM_TW - 1)) + 7) & ~(size_t)7);
  int* s_idx = (int*)(s_red + 16);                                   // HALF: the candidates' atom indices, for the flush (see k_lj_half_tile)
  for(int t = tid; t < (nr + 1 - mlo) * 4; t += NT) s_tab[(t >> 2) * EAM_DSTRIDE + (t & 3)] = rhor_spline[((t >> 2) + mlo) * 7 + 3 + (t & 3)];
  // persistent workgroups: the knots are staged once, then the workgroup walks its share of the tiles of "its" XCD
  // (workgroup b runs on XCD b % 8; XCD e owns the contiguous tile range [e*per, (e+1)*per))
  const int per_xcd = (ntiles + 7) >> 3, wg_per_xcd = gridDim.x >> 3;
  const bool use_core = eam_use_core(C, lane);          // (uniform over the whole launch)
  for(int tq = blockIdx.x >> 3; tq < per_xcd; tq += wg_per_xcd) {
  const int witem = (blockIdx.x & 7) * per_xcd + tq;          // ntiles = length of the work list (all tiles, or one part of them)
  if(witem >= ntiles) break;
  const int tile = tile_list ? tile_list[witem] : witem;
  // Full lists: no barrier here. Everybody passed the barrier behind the pair loop, after which only wave 0 reads LDS (s_part), and the
  // next writes to s_part lie behind the next tile's pre-loop barrier: waves 1.. start loading the next tile under wave 0's epilogue.
  if(HALF) __syncthreads();                             // (half lists: the flush loop of the previous tile reads s_racc / s_idx)
  // The tile's loads in three round trips instead of six: header scalars; then candidate indices, own atom index and first slots
  // together; then the positions. (2-3 workgroups per CU — the knot table bounds the occupancy — hide little of a longer chain.)
  const int ncand = tile_ncand[tile], cnt = tile_cnt[tile], first = tile_first[tile];
  const int kmax = (MMD_ABLATE(C.ablate) & 2) ? 0 : (use_core ? C.tile_kcore[tile] : tile_max[tile]);
  // one rank, full lists (G.cand_src): ghosts are staged from their OWNERS' current positions + the box shift of their image code — the step has
  // no Comm::communicate launch (tile_lds.hpp: GhostResolve); for owned atoms and the dummy the two lists hold the same index
  constexpr bool packed = SRC != 0 && !HALF;      // (EAM: the build writes the second list for every tile, so the choice costs the sweeps no registers)
  const int* __restrict__ cl = (packed ? cand_src : tile_cand) + (size_t)tile * cstride;
  const bool stage = !(MMD_ABLATE(C.ablate) & 1);
  int tt[EAM_STAGE], jj[EAM_STAGE];
#pragma unroll
  for(int u = 0; u < EAM_STAGE; u++) { tt[u] = min(u * NT + tid, ncand); jj[u] = cl[tt[u]]; }     // branch-free: cl[ncand] holds the dummy atom's index
  int i = lane < cnt ? binned[first + lane] : -1;
  const int per = ((kmax / EAM_TU + EAM_TW - 1) / EAM_TW) * EAM_TU;
  const int k0 = min(wv * per, kmax), k1 = min(k0 + per, kmax);
  const unsigned short* __restrict__ np = nl16 + ((size_t)tile * maxneighs + k0) * 64 + lane;
  int sl[EAM_TU];
#pragma unroll
  for(int u = 0; u < EAM_TU; u++) sl[u] = 0;
  if(k0 < k1) {
#pragma unroll
    for(int u = 0; u < EAM_TU; u++) sl[u] = np[u * 64];
  }
  real4 pp[EAM_STAGE];
#pragma unroll
  for(int u = 0; u < EAM_STAGE; u++) pp[u] = x[packed ? (jj[u] & MMD_SRC_MASK) : jj[u]];
  if(i >= nlocal) i = -1;
  const real4 xi = x[i >= 0 ? i : 0];
  if(stage) {
    if(packed) {
#pragma unroll
      for(int u = 0; u < EAM_STAGE; u++) pp[u] = ghost_shifted(pp[u], jj[u], box_dev);
    }
#pragma unroll
    for(int u = 0; u < EAM_STAGE; u++) {
      s_pos[3 * tt[u]] = pp[u].x; s_pos[3 * tt[u] + 1] = pp[u].y; s_pos[3 * tt[u] + 2] = pp[u].z;
      if(HALF) { s_racc[tt[u]] = 0; s_idx[tt[u]] = jj[u]; }
    }
    for(int t0 = EAM_STAGE * NT; t0 <= ncand; t0 += NT) {          // (a union beyond EAM_STAGE * NT candidates: rare)
      const int t = min(t0 + tid, ncand), j = cl[t];
      const real4 p = packed ? ghost_shifted(x[j & MMD_SRC_MASK], j, box_dev) : x[j];
      s_pos[3 * t] = p.x; s_pos[3 * t + 1] = p.y; s_pos[3 * t + 2] = p.z;
      if(HALF) { s_racc[t] = 0; s_idx[t] = j; }
    }
  }
  __syncthreads();
  drain_loads();                                      // (see tile_lds.hpp: lets the slot prefetch of the pair loop really overlap)
  real rhoi = 0;
  for(int k = k0; k < k1; k += EAM_TU) {
    real xj[EAM_TU], yj[EAM_TU], zj[EAM_TU];
    unsigned sc[EAM_TU];
#pragma unroll
    for(int u = 0; u < EAM_TU; u++) { sc[u] = (unsigned)sl[u]; lds_read3<EAM_RD>((unsigned)sl[u], xj[u], yj[u], zj[u]); }
    np += EAM_TU * 64;
    if(k + EAM_TU < k1) {                               // the next trip's slots travel under this trip's arithmetic
#pragma unroll
      for(int u = 0; u < EAM_TU; u++) sl[u] = np[u * 64];
    }
#pragma unroll
    for(int u = 0; u < EAM_TU; u++) {
      const real dx = xi.x - xj[u], dy = xi.y - yj[u], dz = xi.z - zj[u];
      const real rsq = fma_r(dz, dz, fma_r(dy, dy, dx * dx));
      if(rsq < cutforcesq) {
        real p = fma_r(rsq * rsqrt_fast(rsq), rdr, (real)1.0);
        int m = (int)p;
        m = m < nr - 1 ? m : nr - 1;
        p -= m;
        p = p < (real)1.0 ? p : (real)1.0;
        const real* c = &s_tab[(m > mlo ? m - mlo : 0) * EAM_DSTRIDE];
        real c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
        if(__builtin_amdgcn_ballot_w64(m < mlo) != 0ull) {            // (rare: a pair closer than the window's first knot)
          if(m < mlo) { const real* cg = &rhor_spline[m * 7 + 3]; c0 = cg[0]; c1 = cg[1]; c2 = cg[2]; c3 = cg[3]; }
        }
        const real term = fma_r(fma_r(fma_r(c0, p, c1), p, c2), p, c3);
        rhoi += term;
        // the partner's share (a ghost partner's accumulator is simply not flushed: `if (j < nlocal)`, ref :155)
        if(HALF) unsafeAtomicAdd(&s_racc[sc[u] / (3u * (unsigned)sizeof(real))], (double)term);
      }
    }
  }
  if(wv > 0) s_part[64 * (wv - 1) + lane] = rhoi;
  __syncthreads();
  double e_acc = 0;
  if(HALF) {
    if(wv == 0 && i >= 0) {
#pragma unroll
      for(int q = 0; q < EAM_TW - 1; q++) rhoi += s_part[64 * q + lane];
      const unsigned own = tile_self[(size_t)tile * 64 + lane];
      if(own != 0xffffu) s_racc[own] += (double)rhoi;            // the atom is one of the tile's candidates: one flush for both
      else unsafeAtomicAdd(rho + i, rhoi);
    }
    __syncthreads();
    for(int t = tid; t < ncand; t += NT) {
      const int j = s_idx[t];
      const double a = s_racc[t];
      if(j < nlocal && a != 0) unsafeAtomicAdd(rho + j, (real)a);
    }
  } else if(wv == 0 && i >= 0) {
#pragma unroll
    for(int q = 0; q < EAM_TW - 1; q++) rhoi += s_part[64 * q + lane];
    real p = (real)1.0 * rhoi * rdrho + (real)1.0;
    int m = (int)p;
    m = max(1, min(m, nrho - 1));
    p -= m;
    p = p < (real)1.0 ? p : (real)1.0;
    const real* c = &frho_spline[m * 7];
    fp[i] = (c[0] * p + c[1]) * p + c[2];
    if(EV) e_acc = (double)(((c[3] * p + c[4]) * p + c[5]) * p + c[6]);
  }
  if(EV && !HALF) {
    const double es = block_sum(e_acc, s_red);
    if(tid == 0) partials[3 * (size_t)tile] = es;
  }
  }   // tile loop
}

// FUSE=1: wave 0 also applies finalIntegrate(n) + initialIntegrate(n+1) to the tile's atoms (see k_lj_full_tile)
// HALF=1: the third loop of ForceEAM::compute_halfneigh (ref/force_eam.cpp:190-267) on a half-list tile: the partner's share of a
// pair goes to three LDS accumulators per candidate, flushed for the owned candidates at the end of the tile (see
// k_eam_density_tile); a ghost partner gets no force and the pair counts half in energy and virial (:244-257). f was zeroed
// beforehand; partials = {sum phi, virial} per tile.
template <int EV, int FUSE, int HALF, int SRC = 0>
__global__ __launch_bounds__(64 * EAM_FW) __attribute__((amdgpu_waves_per_eu(HALF ? 2 : EAM_FWAVES))) void k_eam_force_tile(
    const real4* __restrict__ x, const int* __restrict__ binned, const int* __restrict__ tile_first, const int* __restrict__ tile_cnt,
    const int* __restrict__ tile_max, const int* __restrict__ tile_cand, const int* __restrict__ tile_ncand, int cstride, int ntiles,
    const int* __restrict__ tile_list,
    const unsigned short* __restrict__ nl16, int nlocal, int nall, int maxneighs, const real* __restrict__ rhor_spline,
    const real* __restrict__ z2r_spline, real cutforcesq, int nr, int cmax, real rdr, const real* __restrict__ fp, real* __restrict__ f,
    double* __restrict__ partials, real* __restrict__ v, real4* __restrict__ xnew, real dt, real dtforce, int mlo,
    const unsigned short* __restrict__ tile_self, EamCore C, const int* __restrict__ fp_root,
    const int* __restrict__ cand_src, const real* __restrict__ box_dev, const int* __restrict__ tile_ghost)
{
  extern __shared__ __align__(16) unsigned char s_raw[];
  constexpr int NT = 64 * EAM_FW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  real* s_pos = (real*)s_raw;
  real* s_fp = (real*)(s_raw + eam_pos_bytes(cmax));                 // fp of the candidates, indexed by slot
  // knot window [mlo, nr], records {rho' 0..2, z2r 3..6, -} of EAM_FSTRIDE reals (80 B: 20 words spread random knots over 16 bank
  // offsets, a 64-byte stride over 4): four aligned ds_read_b128 per pair. The derivative
  // coefficients of z2r are multiples of its value coefficients (array2spline, ref/force_eam.cpp:785-789:
  // [0] = 3 [3] / delta, [1] = 2 [4] / delta, [2] = [5] / delta), so they are not stored: 56 instead of 80 gathered bytes.
  double* s_acc = (double*)(s_raw + eam_pos_bytes(cmax) + eam_fp_bytes(cmax));       // HALF: force accumulators of the candidates
  real* s_tab = (real*)(s_raw + eam_pos_bytes(cmax) + eam_fp_bytes(cmax) + (HALF ? eam_acc_bytes(cmax, 3) : 0));
  real* s_f = s_tab + (size_t)(nr + 1 - mlo) * EAM_FSTRIDE;
  double* s_red = (double*)(((size_t)(s_f + 3 * 64 * (EAM_FW - 1)) + 7) & ~(size_t)7);
  int* s_idx = (int*)(s_red + 16);                                     // HALF: the candidates' atom indices, for the flush
  unsigned char* s_gh = (unsigned char*)(s_idx + ((cmax + 2 + 3) & ~3)); // HALF && EV: candidate is a ghost
  for(int t = tid; t < (nr + 1 - mlo) * 7; t += NT) {
    const int q = (int)(((unsigned)t * 74899u) >> 19);      // t / 7 (exact below 57343 = EAM_MAX_STAGE; eam_*tiles_available keep the window shorter)
    const int m = q + mlo, c = t - 7 * q;
    s_tab[q * EAM_FSTRIDE + c] = c < 3 ? rhor_spline[m * 7 + c] : z2r_spline[m * 7 + c];
  }
  const int per_xcd = (ntiles + 7) >> 3, wg_per_xcd = gridDim.x >> 3;     // persistent workgroups, see k_eam_density_tile
  const bool use_core = eam_use_core(C, lane);
  if(FUSE && C.words_zero != nullptr && blockIdx.x == 0 && tid < 64) C.words_zero[tid] = 0;       // the set the NEXT launch will write
  if(FUSE && blockIdx.x == 0 && tid == 0) xnew[nall] = real4{(real)1.0e15, (real)1.0e15, (real)1.0e15, (real)0};      // (the dummy atom of the buffer this launch fills: no k_set_dummy launch after a re-neighboring)
  float d2max = 0;                                    // FUSE: largest squared displacement since the build over this workgroup's atoms
  for(int tq = blockIdx.x >> 3; tq < per_xcd; tq += wg_per_xcd) {
  const int witem = (blockIdx.x & 7) * per_xcd + tq;          // ntiles = length of the work list (all tiles, or one part of them)
  if(witem >= ntiles) break;
  const int tile = tile_list ? tile_list[witem] : witem;
  if(HALF) __syncthreads();                             // (see k_eam_density_tile: full lists need no barrier here)
  // three round trips (see k_eam_density_tile): header; indices + own atom + first slots; positions + fp
  const int ncand = tile_ncand[tile], cnt = tile_cnt[tile], first = tile_first[tile];
  const int kmax = (MMD_ABLATE(C.ablate) & 2) ? 0 : (use_core ? C.tile_kcore[tile] : tile_max[tile]);
  constexpr bool packed = SRC != 0 && !HALF;       // (see k_eam_density_tile: ghosts named by owner + image code)
  const int* __restrict__ cl = (packed ? cand_src : tile_cand) + (size_t)tile * cstride;
  const bool stage = !(MMD_ABLATE(C.ablate) & 1);
  // one rank: a ghost is an image of an owned atom, its fp is its owner's (ForceEAM::communicate, ref/force_eam.cpp:851-913, folded
  // into the staging: no fp halo launch between the two sweeps)
  auto fp_index = [&](int j) { return packed ? (j & MMD_SRC_MASK) : ((fp_root != nullptr && j >= nlocal && j < nall) ? fp_root[j - nlocal] : j); };
  int tt[EAM_STAGE], jj[EAM_STAGE];
#pragma unroll
  for(int u = 0; u < EAM_STAGE; u++) { tt[u] = min(u * NT + tid, ncand); jj[u] = EAM_F_LOAD(cl + tt[u]); }
  int i = lane < cnt ? binned[first + lane] : -1;
  // the rows (a multiple of 4) are dealt to the wavefronts two at a time: 52 rows = 14,14,12,12 instead of 16,16,16,4 — the slowest
  // wavefront is the tile's critical path
  const int hq = kmax >> 1, hbase = hq / EAM_FW, hrem = hq - hbase * EAM_FW;
  const int k0 = 2 * (wv * hbase + min(wv, hrem)), k1 = k0 + 2 * (hbase + (wv < hrem ? 1 : 0));
  // (the lane offset is made opaque per tile: hoisted out of the tile loop, `nl16 + lane` and its +512-byte twin are two 64-bit register pairs that
  //  live through the pair loop — in the variant that stages ghosts from their owners they were spilled and reloaded behind a full vmcnt(0))
  int lane_o = lane;
  asm volatile("" : "+v"(lane_o));
  const unsigned short* __restrict__ np = nl16 + ((size_t)tile * maxneighs + k0) * 64 + lane_o;
  int sl[EAM_FU];
#pragma unroll
  for(int u = 0; u < EAM_FU; u++) sl[u] = 0;
  if(k0 < k1) {                                   // (a slice of 4 rows reads 4 entries past it: in bounds — nl16 ends with 16 spare rows — and unused)
#pragma unroll
    for(int u = 0; u < EAM_FU; u++) sl[u] = EAM_F_LOAD(np + u * 64);
  }
  real4 pp[EAM_STAGE];
  real ff[EAM_STAGE];
#pragma unroll
  for(int u = 0; u < EAM_STAGE; u++) { pp[u] = x[packed ? (jj[u] & MMD_SRC_MASK) : jj[u]]; ff[u] = fp[fp_index(jj[u])]; }
  if(i >= nlocal) i = -1;
  const real4 xi = x[i >= 0 ? i : 0];
  const real fpi = fp[i >= 0 ? i : 0];
  if(stage) {
    if(packed) {
#pragma unroll
      for(int u = 0; u < EAM_STAGE; u++) pp[u] = ghost_shifted(pp[u], jj[u], box_dev);
    }
#pragma unroll
    for(int u = 0; u < EAM_STAGE; u++) {
      s_pos[3 * tt[u]] = pp[u].x; s_pos[3 * tt[u] + 1] = pp[u].y; s_pos[3 * tt[u] + 2] = pp[u].z; s_fp[tt[u]] = ff[u];
      if(HALF) { s_acc[3 * tt[u]] = 0; s_acc[3 * tt[u] + 1] = 0; s_acc[3 * tt[u] + 2] = 0; s_idx[tt[u]] = jj[u]; }
      if(HALF && EV) s_gh[tt[u]] = jj[u] >= nlocal ? 1 : 0;
    }
    for(int t0 = EAM_STAGE * NT; t0 <= ncand; t0 += NT) {          // (a union beyond EAM_STAGE * NT candidates: rare)
      const int t = min(t0 + tid, ncand), j = cl[t];
      const real4 p = packed ? ghost_shifted(x[j & MMD_SRC_MASK], j, box_dev) : x[j];
      const real fj = fp[fp_index(j)];
      s_pos[3 * t] = p.x; s_pos[3 * t + 1] = p.y; s_pos[3 * t + 2] = p.z; s_fp[t] = fj;
      if(HALF) { s_acc[3 * t] = 0; s_acc[3 * t + 1] = 0; s_acc[3 * t + 2] = 0; s_idx[t] = j; }
      if(HALF && EV) s_gh[t] = j >= nlocal ? 1 : 0;
    }
  }
  __syncthreads();
  drain_loads();
  real fx = 0, fy = 0, fz = 0;
  double e_acc = 0, v_acc = 0;
  // one trip = U pairs of every lane, written without branches so that the U independent chains (position read -> 1/r -> knot ->
  // coefficient gather -> Horner forms) interleave: with 2-3 wavefronts per SIMD (the knot table limits the occupancy) the LDS
  // round trips must be hidden inside a wavefront. Out-of-range pairs and padding look up the last knot and are zeroed.
  auto trip = [&](auto nu, int k) {
    constexpr int U = decltype(nu)::value;
    real xj[U], yj[U], zj[U], fpj[U];
    unsigned sc[U];
#pragma unroll
    for(int u = 0; u < U; u++) {
      lds_read3<EAM_RD>((unsigned)sl[u], xj[u], yj[u], zj[u]);
      sc[u] = (unsigned)sl[u] / (3u * (unsigned)sizeof(real));
      fpj[u] = s_fp[sc[u]];
    }
    np += U * 64;
    if(k + U < k1) {
#pragma unroll
      for(int u = 0; u < EAM_FU; u++) sl[u] = EAM_F_LOAD(np + u * 64);
    }
    real dx[U], dy[U], dz[U], rsq[U], recip[U], p[U];
    const real* c[U];
    bool low = false;
#pragma unroll
    for(int u = 0; u < U; u++) {
      dx[u] = xi.x - xj[u]; dy[u] = xi.y - yj[u]; dz[u] = xi.z - zj[u];
      rsq[u] = fma_r(dz[u], dz[u], fma_r(dy[u], dy[u], dx[u] * dx[u]));      // explicit fma throughout: see tile_lds.hpp
      recip[u] = rsqrt_fast(rsq[u]);
      real pp = fma_r(rsq[u] * recip[u], rdr, (real)1.0);
      pp = pp < (real)nr ? pp : (real)nr;                        // (the dummy atom: keeps the conversion in range)
      int m = (int)pp;
      m = m < nr - 1 ? m : nr - 1;
      pp -= m;
      p[u] = pp < (real)1.0 ? pp : (real)1.0;
      low = low || (m < mlo && rsq[u] < cutforcesq);
      c[u] = &s_tab[(m > mlo ? m - mlo : 0) * EAM_FSTRIDE];
    }
    real r0[U], r1[U], r2[U], z3[U], z4[U], z5[U], z6[U];
#pragma unroll
    for(int u = 0; u < U; u++) { r0[u] = c[u][0]; r1[u] = c[u][1]; r2[u] = c[u][2]; z3[u] = c[u][3]; z4[u] = c[u][4]; z5[u] = c[u][5]; z6[u] = c[u][6]; }
    if(__builtin_amdgcn_ballot_w64(low) != 0ull) {                // (rare: a pair closer than the window's first knot reads global memory)
#pragma unroll
      for(int u = 0; u < U; u++) {
        int m = (int)fma_r(rsq[u] * recip[u], rdr, (real)1.0);
        if(m < mlo && rsq[u] < cutforcesq) {
          const real* cr = &rhor_spline[m * 7]; const real* cz = &z2r_spline[m * 7];
          r0[u] = cr[0]; r1[u] = cr[1]; r2[u] = cr[2]; z3[u] = cz[3]; z4[u] = cz[4]; z5[u] = cz[5]; z6[u] = cz[6];
        }
      }
    }
#pragma unroll
    for(int u = 0; u < U; u++) {
      const bool in = rsq[u] < cutforcesq;
      const real rhoip = fma_r(fma_r(r0[u], p[u], r1[u]), p[u], r2[u]);
      const real z2p = fma_r(fma_r(z3[u] * (real)3.0, p[u], z4[u] + z4[u]), p[u], z5[u]) * rdr;
      const real z2 = fma_r(fma_r(fma_r(z3[u], p[u], z4[u]), p[u], z5[u]), p[u], z6[u]);
      const real phi = z2 * recip[u];
      const real phip = fma_r(z2p, recip[u], -(phi * recip[u]));
      const real psip = fma_r(fpi, rhoip, fma_r(fpj[u], rhoip, phip));
      real fpair = in ? -psip * recip[u] : (real)0;
      if(HALF) {
        const real px = dx[u] * fpair, py = dy[u] * fpair, pz = dz[u] * fpair;
        fx += px; fy += py; fz += pz;
        if(in) {
          double* a = s_acc + 3 * sc[u];
          unsafeAtomicAdd(a + 0, -(double)px); unsafeAtomicAdd(a + 1, -(double)py); unsafeAtomicAdd(a + 2, -(double)pz);
        }
      } else {
        fx = fma_r(dx[u], fpair, fx); fy = fma_r(dy[u], fpair, fy); fz = fma_r(dz[u], fpair, fz);
      }
      if(EV) {
        const real scale = HALF ? (s_gh[sc[u]] ? (real)0.5 : (real)1.0) : (real)0.5;
        v_acc += (double)(rsq[u] * (fpair * scale));
        e_acc += (double)(in ? scale * phi : (real)0);
      }
    }
  };
  {
    int k = k0;
    for(; k + EAM_FU <= k1; k += EAM_FU) trip(std::integral_constant<int, EAM_FU>{}, k);
    if(k < k1) trip(std::integral_constant<int, 2>{}, k);            // (slices are even)
  }
  if(wv > 0) { real* d = s_f + 3 * 64 * (wv - 1); d[lane] = fx; d[64 + lane] = fy; d[128 + lane] = fz; }
  real vx = 0, vy = 0, vz = 0;                        // FUSE: what wave 0's epilogue reads travels under the wait for the other waves
  real4 xb = xi;
  if(FUSE && wv == 0 && i >= 0) {
    vx = v[3 * (size_t)i + 0]; vy = v[3 * (size_t)i + 1]; vz = v[3 * (size_t)i + 2];
    if(C.words_write != nullptr) xb = C.xbuild[i];
  }
  __syncthreads();
  if(wv == 0 && i >= 0) {
#pragma unroll
    for(int q = 0; q < EAM_FW - 1; q++) { const real* d = s_f + 3 * 64 * q; fx += d[lane]; fy += d[64 + lane]; fz += d[128 + lane]; }
    if(HALF) {
      const unsigned own = tile_self[(size_t)tile * 64 + lane];
      if(own != 0xffffu) { s_acc[3 * own] += (double)fx; s_acc[3 * own + 1] += (double)fy; s_acc[3 * own + 2] += (double)fz; }
      else { real* fi = f + 3 * (size_t)i; unsafeAtomicAdd(fi + 0, fx); unsafeAtomicAdd(fi + 1, fy); unsafeAtomicAdd(fi + 2, fz); }
    }
    if(!FUSE && !HALF) { f[3 * (size_t)i + 0] = fx; f[3 * (size_t)i + 1] = fy; f[3 * (size_t)i + 2] = fz; }
    if(FUSE) {          // same operations, same order as k_final_initial_integrate (integrate.hip)
      vx = mul_add_unfused(dtforce, fx, vx); vy = mul_add_unfused(dtforce, fy, vy); vz = mul_add_unfused(dtforce, fz, vz);
      vx = mul_add_unfused(dtforce, fx, vx); vy = mul_add_unfused(dtforce, fy, vy); vz = mul_add_unfused(dtforce, fz, vz);
      v[3 * (size_t)i + 0] = vx; v[3 * (size_t)i + 1] = vy; v[3 * (size_t)i + 2] = vz;
      const real4 xn = real4{mul_add_unfused(dt, vx, xi.x), mul_add_unfused(dt, vy, xi.y), mul_add_unfused(dt, vz, xi.z), xi.w};
      xnew[i] = xn;
      if(C.words_write != nullptr) {                  // how far from its position at the build (rounded up: it gates the core rows)
        const real ex = xn.x - xb.x, ey = xn.y - xb.y, ez = xn.z - xb.z;
        d2max = fmaxf(d2max, (float)(ex * ex + ey * ey + ez * ez) * 1.000001f + 1.0e-30f);
      }
    }
  }
  if(HALF) {
    __syncthreads();
    for(int e = tid; e < 3 * ncand; e += NT) {             // memory order: 3 doubles per candidate, candidates are runs of consecutive atoms
      const int t = (int)(((unsigned)e * 43691u) >> 17);     // e / 3 (exact below 98304)
      const int j = s_idx[t];
      const double a = s_acc[e];
      if(j < nlocal && a != 0) unsafeAtomicAdd(f + 3 * (size_t)j + (e - 3 * t), (real)a);
    }
  }
  if(EV) {
    if(i < 0) { e_acc = 0; v_acc = 0; }
    const double es = block_sum(e_acc, s_red);
    const double vs = block_sum(v_acc, s_red);
    if(tid == 0) {
      if(HALF) { partials[2 * (size_t)tile] = es; partials[2 * (size_t)tile + 1] = vs; }
      else { partials[3 * (size_t)tile + 1] = es; partials[3 * (size_t)tile + 2] = vs; }
    }
  }
  }   // tile loop
  if(FUSE && C.words_write != nullptr && wv == 0) {
    const unsigned m = wave_max_u(__float_as_uint(d2max));       // (d2 >= 0: float bits order like the values)
    if(lane == 0 && m != 0u) atomicMax(&C.words_write[blockIdx.x & 63], m);
  }
}

// eng_vdwl = 2*(E_embed + sum phi/2) (ref/force_eam.cpp:446), virial = sum
__global__ __launch_bounds__(1024) void k_eam_sum(const double* __restrict__ partials, int nblocks, double* __restrict__ out)
{
  __shared__ double s_red[16];
  double a = 0, b = 0, c = 0;
  for(int k = threadIdx.x; k < nblocks; k += blockDim.x) { a += partials[3 * (size_t)k]; b += partials[3 * (size_t)k + 1]; c += partials[3 * (size_t)k + 2]; }
  const double ta = block_sum(a, s_red), tb = block_sum(b, s_red), tc = block_sum(c, s_red);
  if(threadIdx.x == 0) { out[0] = 2.0 * (ta + tb); out[1] = tc; }
}

__global__ __launch_bounds__(256) void k_fp_self(real* __restrict__ fp, const int* __restrict__ list, int n, int first)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) fp[first + i] = fp[list[i]];
}
__global__ __launch_bounds__(256) void k_fp_pack(const real* __restrict__ fp, const int* __restrict__ list, int n, real* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) out[i] = fp[list[i]];
}

// one rank: every ghost is an image of an owned atom (Comm::borders recorded the owner): all six self swaps in one launch
__global__ __launch_bounds__(256) void k_fp_ghosts(real* __restrict__ fp, const int* __restrict__ root, int nlocal, int nghost)
{
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if(g < nghost) fp[nlocal + g] = fp[root[g]];
}

// ForceEAM::communicate (ref/force_eam.cpp:851-887): one scalar per send-list atom, swap by swap
static int eam_fp_halo(mmd_handle* h)
{
  if(h->fp_halo_fn) {
    // the caller's ForceEAM::communicate (ghosts that this handle's Comm did not make): owned fp to the host, ghost fp back
    const size_t nall = (size_t)h->nlocal + h->nghost;
    if(h->fp_stage.size() < nall + 8) h->fp_stage.resize(nall + 8);
    if(h->nlocal) HIP_TRY(hipMemcpyAsync(h->fp_stage.data(), h->fp.p, (size_t)h->nlocal * sizeof(real), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    if(h->fp_halo_fn(h->fp_halo_ctx, h->fp_stage.data(), h->nlocal, h->nghost) != 0) { mmd_set_error("ForceEAM::communicate callback failed"); return -1; }
    if(h->nghost) HIP_TRY(hipMemcpyAsync(h->fp.p + h->nlocal, h->fp_stage.data() + h->nlocal, (size_t)h->nghost * sizeof(real), hipMemcpyHostToDevice, h->stream));
    return 0;
  }
  if(h->ghosts_uploaded) {
    mmd_set_error("ForceEAM::communicate: the ghost atoms were uploaded (mmd_atom_upload), this handle has no send lists for them; "
                  "rebuild them with mmd_comm_borders or install the caller's halo with mmd_force_eam_set_fp_halo");
    return -1;
  }
  if(h->ghost_chain_ok && h->opt_fuse && !h->opt_force_transport) {
    if(h->nghost) hipLaunchKernelGGL(k_fp_ghosts, dim3(div_up(h->nghost, 256)), dim3(256), 0, h->stream, h->fp.p, h->ghost_root.p, h->nlocal, h->nghost);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if(h->dh.ready) return mmd_dh_exchange(h, 1);           // several ranks: one exchange with the up to 26 neighbours (DirectHalo, mmd_internal.hpp)
  MMD_TRY(mmd_comm_sendlists_ensure(h));
  for(auto& s : h->swaps) {
    if(s.sendproc == h->me && !h->opt_force_transport) {
      if(s.sendnum) hipLaunchKernelGGL(k_fp_self, dim3(div_up(s.sendnum, 256)), dim3(256), 0, h->stream, h->fp.p, s.sendlist.p, s.sendnum, s.firstrecv);
    } else {
      MMD_TRY(h->buf_send.ensure((size_t)s.sendnum + 8, false, h->stream));
      if(s.sendnum) hipLaunchKernelGGL(k_fp_pack, dim3(div_up(s.sendnum, 256)), dim3(256), 0, h->stream, h->fp.p, s.sendlist.p, s.sendnum, h->buf_send.p);
      MMD_TRY(mmd_transport_sendrecv(h, h->buf_send.p, (size_t)s.sendnum * sizeof(real), s.sendproc, h->fp.p + s.firstrecv,
                                     (size_t)s.recvnum * sizeof(real), s.recvproc));
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int mmd_force_eam_set_fp_halo(mmd_handle* h, mmd_fp_halo_fn fn, void* ctx)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  h->fp_halo_fn = fn; h->fp_halo_ctx = ctx;
  return 0;
}

extern "C" int mmd_force_eam_setup(mmd_handle* h, int ntypes, int nr, int nrho, int nr_tot, int nrho_tot, mmd_float rdr,
                                   mmd_float rdrho, const mmd_float* rhor_spline, const mmd_float* frho_spline,
                                   const mmd_float* z2r_spline, const mmd_float* cutforcesq)
{
  if(!h || ntypes < 1 || !rhor_spline || !frho_spline || !z2r_spline || !cutforcesq) { mmd_set_error("mmd_force_eam_setup: bad arguments"); return -1; }
  if((nr + 1) * 7 > nr_tot || (nrho + 1) * 7 > nrho_tot) { mmd_set_error("mmd_force_eam_setup: table strides too small"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  h->style = 1;
  h->ntypes = ntypes;
  h->nr = nr; h->nrho = nrho; h->nr_tot = nr_tot; h->nrho_tot = nrho_tot; h->rdr = rdr; h->rdrho = rdrho;
  const int nt2 = ntypes * ntypes;
  bool uni = nr + 1 <= EAM_MAX_KNOTS;
  for(int t = 1; t < nt2 && uni; t++) {
    if(cutforcesq[t] != cutforcesq[0]) uni = false;
    if(memcmp(rhor_spline + (size_t)t * nr_tot, rhor_spline, sizeof(real) * (nr + 1) * 7)) uni = false;
    if(memcmp(z2r_spline + (size_t)t * nr_tot, z2r_spline, sizeof(real) * (nr + 1) * 7)) uni = false;
    if(memcmp(frho_spline + (size_t)t * nrho_tot, frho_spline, sizeof(real) * (nrho + 1) * 7)) uni = false;
  }
  h->eam_uniform = uni;
  h->h_cutforcesq.assign(cutforcesq, cutforcesq + nt2);
  MMD_TRY(h->rhor_spline.ensure((size_t)nt2 * nr_tot, false, h->stream));
  MMD_TRY(h->z2r_spline.ensure((size_t)nt2 * nr_tot, false, h->stream));
  MMD_TRY(h->frho_spline.ensure((size_t)nt2 * nrho_tot, false, h->stream));
  MMD_TRY(h->lj_tables.ensure((size_t)nt2 + 8, false, h->stream));
  HIP_TRY(hipMemcpyAsync(h->rhor_spline.p, rhor_spline, sizeof(real) * (size_t)nt2 * nr_tot, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->z2r_spline.p, z2r_spline, sizeof(real) * (size_t)nt2 * nr_tot, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->frho_spline.p, frho_spline, sizeof(real) * (size_t)nt2 * nrho_tot, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->lj_tables.p, cutforcesq, sizeof(real) * nt2, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  return 0;
}

// first knot kept in LDS: r = 0.3 x cutoff (1.5 A for Cu_u6; closer pairs read global memory)
static int eam_mlo(const mmd_handle* h)
{
  int m = (int)(0.3 * sqrt((double)h->h_cutforcesq[0]) * (double)h->rdr);
  if(h->opt_eam_mlo >= 0) m = h->opt_eam_mlo;               // (tests: push pairs onto the global-memory path)
  return m < 1 ? 1 : (m > h->nr - 1 ? h->nr - 1 : m);
}
static size_t eam_tile_lds_density(const mmd_handle* h)
{
  return eam_pos_bytes(h->tile_cmax) + (size_t)(h->nr + 1 - eam_mlo(h)) * EAM_DSTRIDE * sizeof(real) + (size_t)64 * (EAM_TW - 1) * sizeof(real) + 8 + 16 * sizeof(double);
}
static size_t eam_tile_lds_force(const mmd_handle* h)
{
  return eam_pos_bytes(h->tile_cmax) + eam_fp_bytes(h->tile_cmax) + (size_t)(h->nr + 1 - eam_mlo(h)) * EAM_FSTRIDE * sizeof(real) +
         (size_t)3 * 64 * (EAM_FW - 1) * sizeof(real) + 8 + 16 * sizeof(double);
}
static size_t eam_tile_lds_density_half(const mmd_handle* h) { return eam_tile_lds_density(h) + eam_acc_bytes(h->tile_cmax, 1) + (size_t)4 * (h->tile_cmax + 8); }
static size_t eam_tile_lds_force_half(const mmd_handle* h) { return eam_tile_lds_force(h) + eam_acc_bytes(h->tile_cmax, 3) + (size_t)5 * (h->tile_cmax + 8) + 16; }
// half lists (without ghost newton) in tile form: third-law scatter through LDS accumulators
#define EAM_MAX_STAGE 57343        // table words the force sweep may stage: its division by 7 is a multiply + shift that is exact below this
static bool eam_stage_ok(const mmd_handle* h) { return (long long)(h->nr + 1 - eam_mlo(h)) * 7 < EAM_MAX_STAGE; }
static bool eam_half_tiles_available(const mmd_handle* h)
{
  return h->style == 1 && eam_stage_ok(h) && h->halfneigh && !h->ghost_newton && h->tiles_ready && h->opt_tiles && h->eam_uniform &&
         eam_tile_lds_force_half(h) <= 144 * 1024 && h->neigh_nlocal == h->nlocal;
}
static bool eam_tiles_available(const mmd_handle* h)
{
  return h->style == 1 && eam_stage_ok(h) && !h->halfneigh && h->tiles_ready && h->opt_tiles && h->eam_uniform && eam_tile_lds_force(h) <= 144 * 1024 &&
         h->neigh_nlocal == h->nlocal;
}
// the force sweep of the tile path can carry finalIntegrate(n) + initialIntegrate(n+1) (no energy/virial on that step)
int mmd_eam_can_fuse_integrate(mmd_handle* h) { return eam_tiles_available(h) ? 1 : 0; }

int mmd_eam_compute(mmd_handle* h, int evflag, double* eng, double* vir)
{
  if(h->neigh_nlocal != h->nlocal) { mmd_set_error("mmd_force_compute: neighbor list is stale (build or upload one first)"); return -1; }
  if(h->halfneigh && h->ghost_newton) { mmd_set_error("EAM needs half lists WITHOUT ghost newton (ref/ljs.cpp:219-223 forces -gn 0)"); return -1; }
  const int nlocal = h->nlocal, nall = nlocal + h->nghost;
  if(nlocal == 0) {
    // a rank without atoms still takes part in the fp halo (ForceEAM::communicate: its neighbours send to it and receive from it)
    if(h->halo_pending) { HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_halo_done, 0)); h->halo_pending = false; }
    MMD_TRY(h->fp.ensure((size_t)nall + 64, false, h->stream));
    MMD_TRY(eam_fp_halo(h));
    h->fp_ghosts_stale = false;
    if(evflag) HIP_TRY(hipMemsetAsync(h->d_result, 0, 2 * sizeof(double), h->stream));
    if(eng) *eng = 0;
    if(vir) *vir = 0;
    return 0;
  }
  MMD_TRY(h->fp.ensure((size_t)nall + 64, false, h->stream));
  h->fp_ghosts_stale = false;
  if(eam_half_tiles_available(h)) {
    // ---- ForceEAM::compute_halfneigh (ref/force_eam.cpp:94-270) on the tile lists: the partner's share of every pair is summed in
    // LDS, one global atomic per owned candidate and tile (k_eam_density_tile<.,1>, k_eam_force_tile<.,0,1>)
    const int nt = h->ntiles, mlo = eam_mlo(h);
    const int nblocks = div_up(nlocal, MMD_BLOCK);
    MMD_TRY(h->rho.ensure((size_t)nall + 64, false, h->stream));
    MMD_TRY(h->partials.ensure((size_t)nblocks + 2 * (size_t)nt + 32, false, h->stream));
    double* p_embed = h->partials.p;
    double* p_pair = h->partials.p + nblocks + 8;
    HIP_TRY(hipMemsetAsync(h->rho.p, 0, (size_t)nlocal * sizeof(real), h->stream));
    MMD_TRY(mmd_zero_forces(h, nall));
    const size_t tl1 = eam_tile_lds_density_half(h), tl2 = eam_tile_lds_force_half(h);
    const int cus = h->prop.multiProcessorCount;
    const int pgrid1 = std::max(8, (int)(cus * std::max<size_t>(1, std::min<size_t>(8, EAM_LDS_BUDGET / (tl1 + 512)))) / 8 * 8);
    const int pgrid2 = std::max(8, (int)(cus * std::max<size_t>(1, std::min<size_t>(8, EAM_LDS_BUDGET / (tl2 + 512)))) / 8 * 8);
    if(!h->eam_half_attr_set) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_density_tile<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      h->eam_half_attr_set = true;
    }
    hipLaunchKernelGGL((k_eam_density_tile<0, 1>), dim3(pgrid1), dim3(64 * EAM_TW), tl1, h->stream, h->x.p, h->binned.p, h->tile_first.p,
                       h->tile_cnt.p, h->tile_max.p, h->tile_cand.p, h->tile_ncand.p, h->tile_cstride, nt, (const int*)nullptr, h->nl16.p, nlocal, nall,
                       h->maxneighs, h->rhor_spline.p, h->frho_spline.p, h->h_cutforcesq[0], h->nr, h->nrho, h->tile_cmax, h->rdr, h->rdrho,
                       h->fp.p, (double*)nullptr, mlo, h->tile_self.p, h->rho.p, EamCore{}, (const int*)nullptr, (const real*)nullptr, (const int*)nullptr);
    if(evflag) hipLaunchKernelGGL((k_eam_half_fp<1>), dim3(nblocks), dim3(MMD_BLOCK), 0, h->stream, h->x.p, h->rho.p, nlocal, h->frho_spline.p, 1, h->nrho,
                                  h->nrho_tot, h->rdrho, h->fp.p, p_embed);
    else hipLaunchKernelGGL((k_eam_half_fp<0>), dim3(nblocks), dim3(MMD_BLOCK), 0, h->stream, h->x.p, h->rho.p, nlocal, h->frho_spline.p, 1, h->nrho,
                            h->nrho_tot, h->rdrho, h->fp.p, p_embed);
    HIP_TRY(hipGetLastError());
    MMD_TRY(eam_fp_halo(h));
#define FH(EVv) hipLaunchKernelGGL((k_eam_force_tile<EVv, 0, 1>), dim3(pgrid2), dim3(64 * EAM_FW), tl2, h->stream, h->x.p, h->binned.p, h->tile_first.p, \
                       h->tile_cnt.p, h->tile_max.p, h->tile_cand.p, h->tile_ncand.p, h->tile_cstride, nt, (const int*)nullptr, h->nl16.p, nlocal, nall,   \
                       h->maxneighs, h->rhor_spline.p, h->z2r_spline.p, h->h_cutforcesq[0], h->nr, h->tile_cmax, h->rdr, h->fp.p, h->f.p, p_pair,          \
                       h->v.p, h->x_alt.p, h->dt, h->dtforce, mlo, h->tile_self.p, EamCore{}, (const int*)nullptr, (const int*)nullptr, (const real*)nullptr, (const int*)nullptr)
    if(evflag) FH(1); else FH(0);
#undef FH
    HIP_TRY(hipGetLastError());
    if(evflag) {
      hipLaunchKernelGGL(k_eam_half_sum, dim3(1), dim3(1024), 0, h->stream, p_embed, nblocks, p_pair, nt, h->d_result);
      HIP_TRY(hipGetLastError());
      if(eng || vir) {
        HIP_TRY(hipMemcpyAsync(h->h_result, h->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(mmd_stream_sync(h));
        if(eng) *eng = h->h_result[0];
        if(vir) *vir = h->h_result[1];
      }
    }
    return 0;
  }
  if(h->halfneigh) {
    // ---- the same over 32-bit rows with one global floating-point atomic per pair and component (non-uniform tables, uploaded lists)
    MMD_TRY(mmd_ensure_rows(h));
    MMD_TRY(h->rho.ensure((size_t)nall + 64, false, h->stream));
    const int nblocks = div_up(nlocal, MMD_BLOCK);
    MMD_TRY(h->partials.ensure((size_t)3 * nblocks + 16, false, h->stream));
    double* p_embed = h->partials.p;
    double* p_pair = h->partials.p + nblocks + 8;
    HIP_TRY(hipMemsetAsync(h->rho.p, 0, (size_t)nlocal * sizeof(real), h->stream));
    MMD_TRY(mmd_zero_forces(h, nall));
    const bool uni = h->eam_uniform, ev = evflag != 0;
    const size_t lds1 = uni ? (size_t)(h->nr + 1) * 4 * sizeof(real) : 0;
    const size_t lds2 = uni ? (size_t)(h->nr + 1) * 12 * sizeof(real) : 0;
#define HD(Uv) hipLaunchKernelGGL((k_eam_half_density<Uv>), dim3(xcd_grid(nblocks)), dim3(MMD_BLOCK), lds1, h->stream, h->x.p, h->neigh.p, h->wave_max.p,  \
                                  nlocal, h->maxneighs, h->rhor_spline.p, h->lj_tables.p, h->ntypes, h->nr, h->nr_tot, h->rdr, h->rho.p)
#define HP(EVv) hipLaunchKernelGGL((k_eam_half_fp<EVv>), dim3(nblocks), dim3(MMD_BLOCK), 0, h->stream, h->x.p, h->rho.p, nlocal, h->frho_spline.p,         \
                                   uni ? 1 : 0, h->nrho, h->nrho_tot, h->rdrho, h->fp.p, p_embed)
#define HF(EVv, Uv) hipLaunchKernelGGL((k_eam_half_force<EVv, Uv>), dim3(xcd_grid(nblocks)), dim3(MMD_BLOCK), lds2, h->stream, h->x.p, h->neigh.p,           \
                                       h->wave_max.p, nlocal, h->maxneighs, h->rhor_spline.p, h->z2r_spline.p, h->lj_tables.p, h->ntypes, h->nr,            \
                                       h->nr_tot, h->rdr, h->fp.p, h->f.p, p_pair)
    if(uni) HD(1); else HD(0);
    if(ev) HP(1); else HP(0);
    HIP_TRY(hipGetLastError());
    MMD_TRY(eam_fp_halo(h));
    if(ev && uni) HF(1, 1); else if(ev) HF(1, 0); else if(uni) HF(0, 1); else HF(0, 0);
#undef HD
#undef HP
#undef HF
    HIP_TRY(hipGetLastError());
    if(evflag) {
      hipLaunchKernelGGL(k_eam_half_sum, dim3(1), dim3(1024), 0, h->stream, p_embed, nblocks, p_pair, nblocks, h->d_result);
      HIP_TRY(hipGetLastError());
      if(eng || vir) {
        HIP_TRY(hipMemcpyAsync(h->h_result, h->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(mmd_stream_sync(h));
        if(eng) *eng = h->h_result[0];
        if(vir) *vir = h->h_result[1];
      }
    }
    return 0;
  }
  // ---- tile path: LDS-staged candidates + knots (uniform tables, device-built list)
  const size_t tl1 = eam_tile_lds_density(h), tl2 = eam_tile_lds_force(h);
  if(eam_tiles_available(h)) {
    const int nt = h->ntiles, mlo = eam_mlo(h);
    MMD_TRY(h->partials.ensure((size_t)3 * nt + 8, false, h->stream));
    // persistent grids: as many workgroups as fit the LDS budget of every CU (multiple of 8 for the XCD split)
    const int cus = h->prop.multiProcessorCount;
    const int pgrid1 = std::max(8, (int)(cus * std::max<size_t>(1, std::min<size_t>(8, EAM_LDS_BUDGET / (tl1 + 512)))) / 8 * 8);
    const int pgrid2 = std::max(8, (int)(cus * std::max<size_t>(1, std::min<size_t>(8, EAM_LDS_BUDGET / (tl2 + 512)))) / 8 * 8);
    h->eam_diag[0] = (int)tl1; h->eam_diag[1] = (int)tl2; h->eam_diag[2] = pgrid1 / cus; h->eam_diag[3] = pgrid2 / cus;
    if(!h->eam_attr_set) {     // > 64 KiB of dynamic LDS needs the opt-in (per device: the flag lives in the handle)
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_density_tile<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_density_tile<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<0, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<0, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_force_tile<1, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_density_tile<0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_eam_density_tile<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      h->eam_attr_set = true;
    }
    // rows in two parts (CoreRows): the run loop says which part this call may walk; a fused launch tracks the displacement for the next
    EamCore core{};
    core.ablate = h->opt_ablate;
    if(h->core.rows_built && h->core_words.p) {
      core.tile_kcore = h->tile_kcore.p;
      core.mode = (MMD_ABLATE(h->opt_ablate) & 4) ? 1 : h->core.mode_now;          // (ablate 4, profiling: the core part whatever the displacement)
      core.words_read = h->core_words.p + 64 * ((h->core.step + 2) % 3);       // written by the previous tracked launch
      core.thr_d2 = (float)(0.25 * (double)h->core.margin * (double)h->core.margin * (1.0 - 1.0e-6));
      core.xbuild = h->xbuild.p;
      if(h->fuse_now && !h->halo_pending) {
        core.words_write = h->core_words.p + 64 * (h->core.step % 3);
        core.words_zero = h->core_words.p + 64 * ((h->core.step + 1) % 3);
      }
    }
#define DT(EVv, Sv, LIST, CNT) hipLaunchKernelGGL((k_eam_density_tile<EVv, 0, Sv>), dim3(pgrid1), dim3(64 * EAM_TW), tl1, h->stream, h->x.p, h->binned.p,     \
                                   h->tile_first.p, h->tile_cnt.p, h->tile_max.p, h->tile_cand.p, h->tile_ncand.p, h->tile_cstride, CNT, LIST,   \
                                   h->nl16.p, nlocal, nall, h->maxneighs, h->rhor_spline.p, h->frho_spline.p, h->h_cutforcesq[0], h->nr,  \
                                   h->nrho, h->tile_cmax, h->rdr, h->rdrho, h->fp.p, h->partials.p, mlo, (const unsigned short*)nullptr, (real*)nullptr, core, src_p, box_p, (const int*)h->tile_ghost.p)
#define FT(EVv, Fv, Sv, LIST, CNT) hipLaunchKernelGGL((k_eam_force_tile<EVv, Fv, 0, Sv>), dim3(pgrid2), dim3(64 * EAM_FW), tl2, h->stream, h->x.p, h->binned.p,       \
                                   h->tile_first.p, h->tile_cnt.p, h->tile_max.p, h->tile_cand.p, h->tile_ncand.p, h->tile_cstride, CNT, LIST,   \
                                   h->nl16.p, nlocal, nall, h->maxneighs, h->rhor_spline.p, h->z2r_spline.p, h->h_cutforcesq[0], h->nr,   \
                                   h->tile_cmax, h->rdr, h->fp.p, h->f.p, h->partials.p, h->v.p, h->x_alt.p, h->dt, h->dtforce, mlo, (const unsigned short*)nullptr, core, fp_root, src_p, box_p, (const int*)h->tile_ghost.p)
    // one rank: the force sweep reads a ghost's fp through its owner (no fp halo launch); mmd_force_eam_download_fp completes the array
    // one rank, ghosts named by owner + image code in the candidate lists (Integrate::run sets resolve_now on steps without re-neighboring):
    // both sweeps stage the ghosts from their owners — no Comm::communicate launch in front, no fp halo between them
    const int* src_p = (h->resolve_now && h->cand_src_ready && !h->halo_pending) ? (const int*)h->tile_cand_src.p : (const int*)nullptr;
    const real* box_p = nullptr;
    if(src_p != nullptr) {
      MMD_TRY(mmd_box_dev(h));
      box_p = h->box_dev.p;
    }
    const bool fold_fp = src_p != nullptr ||
                         (nt <= 8192 && !h->halo_pending && h->ghost_chain_ok && h->opt_fuse && !h->opt_force_transport);
    const int* fp_root = (fold_fp && src_p == nullptr) ? (const int*)h->ghost_root.p : (const int*)nullptr;
    auto density = [&](const int* list, int cnt) {
      if(cnt <= 0) return;
      if(src_p) { if(evflag) DT(1, 1, list, cnt); else DT(0, 1, list, cnt); }
      else { if(evflag) DT(1, 0, list, cnt); else DT(0, 0, list, cnt); }
    };
    auto force = [&](const int* list, int cnt) {
      if(cnt <= 0) return;
      if(src_p) { if(evflag) FT(1, 0, 1, list, cnt); else if(h->fuse_now) FT(0, 1, 1, list, cnt); else FT(0, 0, 1, list, cnt); }
      else { if(evflag) FT(1, 0, 0, list, cnt); else if(h->fuse_now) FT(0, 1, 0, list, cnt); else FT(0, 0, 0, list, cnt); }
    };
    if(h->halo_pending) {
      // overlapped step (several ranks): the position halo of this step is in flight on the communication stream (the caller
      // recorded ev_halo_done behind it). Interior tiles — no ghost among their candidates — run under it, the boundary tiles
      // after it; the fp halo (ForceEAM::communicate, ref/force_eam.cpp:851-887) then travels under the interior force sweep.
      MMD_TRY(mmd_order_tiles(h));
      const int n_int = h->ntiles_interior, n_bnd = nt - n_int;
      density(h->tile_order.p, n_int);
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_halo_done, 0));
      density(h->tile_order.p + n_int, n_bnd);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(h->ev_x_ready, h->stream));                // (reused: "fp of the owned atoms is complete")
      HIP_TRY(hipStreamWaitEvent(h->comm_stream, h->ev_x_ready, 0));
      std::swap(h->stream, h->comm_stream);
      const int rc = eam_fp_halo(h);
      std::swap(h->stream, h->comm_stream);
      MMD_TRY(rc);
      HIP_TRY(hipEventRecord(h->ev_halo_done, h->comm_stream));
      force(h->tile_order.p, n_int);
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_halo_done, 0));
      force(h->tile_order.p + n_int, n_bnd);
      h->halo_pending = false;
    } else {
      density(nullptr, nt);
      HIP_TRY(hipGetLastError());
      if(!fold_fp) MMD_TRY(eam_fp_halo(h));
      h->fp_ghosts_stale = fold_fp;
      force(nullptr, nt);
    }
    h->core.tracked_last = core.words_write != nullptr && !evflag;       // (the launch that just went out advanced the atoms and recorded how far they are from the build)
    if(h->core.tracked_last) h->core.step++;
#undef DT
#undef FT
    HIP_TRY(hipGetLastError());
    if(evflag) {
      hipLaunchKernelGGL(k_eam_sum, dim3(1), dim3(1024), 0, h->stream, h->partials.p, nt, h->d_result);
      HIP_TRY(hipGetLastError());
      if(eng || vir) {
        HIP_TRY(hipMemcpyAsync(h->h_result, h->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(mmd_stream_sync(h));
        if(eng) *eng = h->h_result[0];
        if(vir) *vir = h->h_result[1];
      }
    }
    return 0;
  }
  MMD_TRY(mmd_ensure_rows(h));
  const int nblocks = div_up(nlocal, MMD_BLOCK);
  MMD_TRY(h->partials.ensure((size_t)3 * nblocks + 8, false, h->stream));
  const size_t lds1 = h->eam_uniform ? (size_t)(h->nr + 1) * 4 * sizeof(real) : 0;
  const size_t lds2 = h->eam_uniform ? (size_t)(h->nr + 1) * 12 * sizeof(real) : 0;
#define D(EVv, Uv)                                                                                                        \
  hipLaunchKernelGGL((k_eam_density<EVv, Uv>), dim3(xcd_grid(nblocks)), dim3(MMD_BLOCK), lds1, h->stream, h->x.p, h->neigh.p,       \
                     h->wave_max.p, nlocal, h->maxneighs, h->rhor_spline.p, h->frho_spline.p, h->lj_tables.p, h->ntypes,  \
                     h->nr, h->nrho, h->nr_tot, h->nrho_tot, h->rdr, h->rdrho, h->fp.p, h->partials.p)
#define FK(EVv, Uv)                                                                                                       \
  hipLaunchKernelGGL((k_eam_force<EVv, Uv>), dim3(xcd_grid(nblocks)), dim3(MMD_BLOCK), lds2, h->stream, h->x.p, h->neigh.p,         \
                     h->wave_max.p, nlocal, h->maxneighs, h->rhor_spline.p, h->z2r_spline.p, h->lj_tables.p, h->ntypes,   \
                     h->nr, h->nr_tot, h->rdr, h->fp.p, h->f.p, h->partials.p)
  const bool ev = evflag != 0, uni = h->eam_uniform;
  if(ev && uni) D(1, 1); else if(ev) D(1, 0); else if(uni) D(0, 1); else D(0, 0);
  HIP_TRY(hipGetLastError());
  MMD_TRY(eam_fp_halo(h));
  if(ev && uni) FK(1, 1); else if(ev) FK(1, 0); else if(uni) FK(0, 1); else FK(0, 0);
#undef D
#undef FK
  HIP_TRY(hipGetLastError());
  if(evflag) {
    hipLaunchKernelGGL(k_eam_sum, dim3(1), dim3(1024), 0, h->stream, h->partials.p, nblocks, h->d_result);
    HIP_TRY(hipGetLastError());
    if(eng || vir) {
      HIP_TRY(hipMemcpyAsync(h->h_result, h->d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(mmd_stream_sync(h));
      if(eng) *eng = h->h_result[0];
      if(vir) *vir = h->h_result[1];
    }
  }
  return 0;
}

extern "C" int mmd_force_eam_download_fp(mmd_handle* h, mmd_float* fp)
{
  if(!h || !fp) { mmd_set_error("mmd_force_eam_download_fp: bad arguments"); return -1; }
  const int nall = h->nlocal + h->nghost;
  if(h->fp.cap < (size_t)nall) { mmd_set_error("mmd_force_eam_download_fp: no EAM force has been computed"); return -1; }
  if(h->fp_ghosts_stale) { MMD_TRY(eam_fp_halo(h)); h->fp_ghosts_stale = false; }      // (the force sweep read the ghosts' fp through their owners)
  HIP_TRY(hipMemcpyAsync(fp, h->fp.p, (size_t)nall * sizeof(real), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  return 0;
}
