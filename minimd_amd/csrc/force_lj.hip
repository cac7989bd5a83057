// minimd_amd/csrc/force_lj.hip — ForceLJ::compute (ref/force_lj.cpp:72-113) as CDNA4 HIP kernels.
//
// Production kernels work on the TILE form of the neighbor list (neighbor.hip: <= 64 atoms of one 2x2x2-bin block, the
// positions of the ~450 atoms their rows reference staged in LDS, 16-bit row entries that are LDS addresses):
//   k_lj_full_tile   full lists (compute_fullneigh); can carry finalIntegrate(n) + initialIntegrate(n+1)
//   k_lj_half_tile   half lists (compute_halfneigh_threaded): Newton's third law scattered into LDS accumulators
// General fallbacks on the reference-style 32-bit rows (one owned atom per lane, wave-interleaved rows
// neigh[(w*maxneighs + k)*64 + lane], one x[j] gather per pair) for uploaded lists and non-uniform type tables:
//   k_lj_full, k_lj_half (global FP atomics)
// Rows are padded with a far-away dummy atom, so the pair loops are wave-uniform and the cutoff is a select.
// Energy/virial: wavefront shuffles -> one partial per workgroup -> fixed-order sum (k_sum_partials) => deterministic
// for full lists. This file is compiled WITH FMA contraction; parity against the (uncontracted) oracle is ~1e-13.
#include <type_traits>
#include "device_utils.hpp"
#include "mmd_internal.hpp"
#include <hip/hip_ext.h>
#include "tile_lds.hpp"

// 1/r^2: v_rcp_f64 + two Newton steps (<= 1 ulp class) instead of the 11-instruction IEEE divide
__device__ __forceinline__ double recip(double a)
{
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(e, r, r);
  e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(e, r, r);
  return r;
}
// one Newton step + the residual folded in:  r1 = r0 + r0*e, e = 1 - a*r0;  second-order term r0*e*e added
// by a single extra fma => same accuracy class as two full steps with one multiply-add less
__device__ __forceinline__ double recip_fast(double a)
{
  const double r = __builtin_amdgcn_rcp(a);
  const double e = __builtin_fma(-a, r, 1.0);
  const double t = __builtin_fma(e, e, e);       // e + e^2
  return __builtin_fma(r, t, r);                 // r (1 + e + e^2)
}
__device__ __forceinline__ float recip_fast(float a)
{
  const float r = __builtin_amdgcn_rcpf(a);
  const float e = __builtin_fmaf(-a, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float recip(float a)
{
  float r = __builtin_amdgcn_rcpf(a);
  const float e = __builtin_fmaf(-a, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

// f(integral_constant<0>), f(integral_constant<STEP>), ... below N: a compile-time loop (the group index feeds an s_waitcnt immediate)
template <int N, int STEP, int I = 0, class F>
__device__ __forceinline__ void static_for_groups(F&& f)
{
  if constexpr(I < N) { f(std::integral_constant<int, I>{}); static_for_groups<N, STEP, I + STEP>(f); }
}

struct LJTables {       // general (non-uniform) case: per type-pair tables staged in LDS by the kernel
  const real* cutforcesq;
  const real* sigma6;
  const real* epsilon;
  int ntypes;
};

#define LJ_MAX_TYPES2 256            // 16 types with DIFFERENT parameters (tables in LDS); uniform tables — all the reference's CLI produces — have no limit
#ifndef LJH_RD
#define LJH_RD (MMD_PRECISION == 2 ? 1 : 0)      // half-list tile kernel, DP: three separate ds_read_b64 per pair (ds_read2_b64 runs at half the LDS rate: -2.5 %)
#endif
#define LJ_UNR 4                 // unroll of the global-gather kernels (rows are padded to MMD_UNROLL >= this)

// ---- full neighbor list: compute_fullneigh<EVFLAG> (ref/force_lj.cpp:366-449) -------------------------
template <int EV, int UNIFORM>
__global__ __launch_bounds__(MMD_BLOCK) void k_lj_full(const real4* __restrict__ x, const int* __restrict__ neigh,
                                                       const int* __restrict__ wave_max, int nlocal, int maxneighs,
                                                       LJParams P, LJTables T, real* __restrict__ f,
                                                       double* __restrict__ partials, int ablate_arg)
{
  const int ablate = MMD_ABLATE(ablate_arg);     // profiling switches: compiled out of the shipped library (mmd_internal.hpp)
  __shared__ real s_cut[UNIFORM ? 1 : LJ_MAX_TYPES2], s_s6[UNIFORM ? 1 : LJ_MAX_TYPES2], s_eps[UNIFORM ? 1 : LJ_MAX_TYPES2];
  __shared__ double s_red[16];
  if(!UNIFORM) {
    for(int t = threadIdx.x; t < T.ntypes * T.ntypes; t += blockDim.x) { s_cut[t] = T.cutforcesq[t]; s_s6[t] = T.sigma6[t]; s_eps[t] = T.epsilon[t]; }
    __syncthreads();
  }
  const int wg = xcd_work_item((nlocal + MMD_BLOCK - 1) / MMD_BLOCK);
  if(wg < 0) return;
  const int i = wg * MMD_BLOCK + threadIdx.x;
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

  for(int k = 0; k < kmax; k += LJ_UNR) {
    int j[LJ_UNR];
    real4 xj[LJ_UNR];
#pragma unroll
    for(int u = 0; u < LJ_UNR; u++) j[u] = np[(size_t)(k + u) * 64];
    if(ablate) {                               // profiling only (results invalid)
#pragma unroll
      for(int u = 0; u < LJ_UNR; u++) {
        if(ablate & 4) j[u] = __builtin_amdgcn_readfirstlane(j[u]);                   // 1 line per gather
        if(ablate & 8) j[u] = __builtin_amdgcn_readfirstlane(j[u]) + (lane >> 2);     // 16 lines, 4 lanes each
        if(ablate & 16) j[u] = __builtin_amdgcn_readfirstlane(j[u]) + lane;           // 16 lines contiguous (2 KB)
        if(ablate & 32) j[u] = (i + 64 * (k + u)) % nlocal;                            // no index dependency, contiguous
      }
    }
#pragma unroll
    for(int u = 0; u < LJ_UNR; u++) xj[u] = x[j[u]];
#pragma unroll
    for(int u = 0; u < LJ_UNR; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      real cut, s6, c48e, eps;
      if(UNIFORM) { cut = P.cutforcesq; s6 = P.sigma6; c48e = c48; eps = P.epsilon; }
      else { const int tij = ti + (int)xj[u].w; cut = s_cut[tij]; s6 = s_s6[tij]; eps = s_eps[tij]; c48e = (real)48.0 * eps; }
      const real sr2 = recip(rsq);
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
    if(threadIdx.x == 0) { partials[2 * (size_t)wg] = es; partials[2 * (size_t)wg + 1] = vs; }
  }
}


// ---- full neighbor list, block-local ("tile") form: positions of the block's candidate atoms are staged
// in LDS once per workgroup and every pair gathers from LDS instead of the L1/TA path -------------------
// One workgroup = one tile (<= 64 owned atoms of one 2x2x2-bin block) x LJ_TILE_WAVES wavefronts; wave w
// handles the neighbor-row slice k in [w*kmax/W, (w+1)*kmax/W) of the same 64 atoms, partial forces are
// combined through LDS. The candidate sequence is rebuilt exactly like k_build's (same bin_start/binned
// arrays, valid until the next re-neighboring) so the 16-bit slots of nl16 index straight into LDS.

// keep `v` when `in`, otherwise clear its high word only: the result is then a denormal whose square underflows
// to exactly 0, so every product built from it (A = sr2^3, the force, the energy term) is an exact zero —
// one v_cndmask instead of the two a 64-bit select costs
__device__ __forceinline__ double keep_if(bool in, double v)
{
  const long long b = __double_as_longlong(v);
  return __hiloint2double(in ? (int)(b >> 32) : 0, (int)b);
}
__device__ __forceinline__ float keep_if(bool in, float v) { return in ? v : 0.0f; }

// dynamic LDS of the tile kernel: [positions: pos_bytes][wave-slice forces: 3*64*(W-1) reals][16 doubles].
// Nothing static precedes it, so the 16-bit values of nl16 ARE the ds_read addresses of the records.
__host__ __device__ constexpr int lj_tile_sf_bytes(int waves) { return 3 * 64 * (waves - 1) * (int)sizeof(real); }

// FUSE=1 appends finalIntegrate of this step and initialIntegrate of the next one (ref/integrate.cpp:46-68) for the
// tile's atoms: v and the NEW positions go to v / xnew (a second position buffer: other tiles still read the old x),
// the caller swaps the buffers. Same operations in the same order as k_final_initial_integrate => same bits.
// The shape is fixed (round 6: the A/B knobs tile_waves / tile_unroll / tile_read / exact_div are gone, their numbers are in DESIGN_HISTORY.md): two wavefronts per tile, trips of
// eight pairs, one reciprocal per FOUR pairs in double precision (one per pair in float), the three coordinates of a pair as three separate LDS reads in double precision
// (the fused ds_read2_b64 runs at half the LDS rate) and as they come in float.
constexpr int LJ_TILE_WAVES = 2, LJ_TILE_UNR = 8, LJ_TILE_RD = MMD_PRECISION == 2 ? 3 : 2;
template <int EV, int FUSE>
__global__ __launch_bounds__(64 * LJ_TILE_WAVES) void k_lj_full_tile(
    const real4* __restrict__ x, const int* __restrict__ binned, const int* __restrict__ tile_first,
    const int* __restrict__ tile_cnt, const int* __restrict__ tile_max, const int* __restrict__ tile_cand, const int* __restrict__ tile_ncand, int cstride, int ntiles,
    const int* __restrict__ tile_list,
    const unsigned short* __restrict__ nl16, int nlocal, int nall, int maxneighs, int pos_bytes, LJParams P, real* __restrict__ f,
    double* __restrict__ partials, int ablate_arg, real* __restrict__ v, real4* __restrict__ xnew, real dt, real dtforce, GhostResolve G, SpecLaunch SP)
{
  const int ablate = MMD_ABLATE(ablate_arg);     // profiling switches: compiled out of the shipped library (mmd_internal.hpp)
  constexpr int UNR = LJ_TILE_UNR;
  // A launch enqueued BEHIND a neighbor build whose results the host has not seen yet (mmd_internal.hpp: SpecLaunch): the build's own
  // verdict decides on the device whether this launch does anything at all, and the tile / ghost counts come from device memory
  if(SP.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) SP.clk[0] = wall_clock64();
  if(SP.gate != nullptr) {
    if(*(const volatile int*)SP.gate == 0) return;
    ntiles = min(ntiles, *SP.ntiles_dev);
    nall = deferred_count(nall, nlocal, SP.nghost_dev);
  }
  extern __shared__ __align__(16) unsigned char s_raw[];
  real* s_f = (real*)(s_raw + pos_bytes);
  double* s_red = (double*)(s_raw + pos_bytes + lj_tile_sf_bytes(LJ_TILE_WAVES));
  constexpr int LJ_TILE_THREADS = 64 * LJ_TILE_WAVES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform => the k loop runs on the scalar unit
  // XCD-aware order (device_utils.hpp): neighbouring tiles share most of their candidate atoms
  const int witem = SP.gate != nullptr ? xcd_work_item_of(ntiles) : xcd_work_item(ntiles);
  if(witem < 0) return;                          // (grid is padded to a multiple of 8)
  const int tile = tile_list ? tile_list[witem] : witem;
  // (the dummy atom of the position buffer this launch fills sits behind the last ghost — behind a build only the device knows where that is; every
  //  fused launch writes it, so no k_set_dummy launch stands between two force kernels after a re-neighboring has moved it: 5 + 5 us)
  if(FUSE == 1 && witem == 0 && tid == 0) xnew[nall] = real4{(real)1.0e15, (real)1.0e15, (real)1.0e15, (real)0};
  // The tile's loads are issued as three round trips — header scalars; candidate indices + own atom index + first slots; positions —
  // not as the six a straight reading of the steps below would make (indices -> positions -> LDS, then atom index -> position, then slots).
  const int ncand = tile_ncand[tile], cnt = tile_cnt[tile], first = tile_first[tile];     // a tile never straddles pencils
  const int kmax = (ablate & 2) ? 0 : tile_max[tile];
  const bool ghosted = G.root != nullptr && G.tile_ghost[tile] != 0;       // (workgroup-uniform) boundary tile of a one-rank run: ghosts from their owners
  // ---- stage the positions of the tile's candidate union (+1 dummy slot) into LDS: {x,y,z} records of
  // 3 reals (stride 3 is coprime with the bank count: random slots spread over all banks, one address per pair)
  real* sp = (real*)s_raw;
  const bool packed = ghosted && G.cand_src != nullptr;      // (boundary tile whose ghosts are named by owner + image code: no look-up in front of the position load)
  const int* __restrict__ cl = (packed ? G.cand_src : tile_cand) + (size_t)tile * cstride;
  // Branch-free: the build stores the dummy atom's index at cl[ncand], lanes past the end clamp to that entry
  // and (re)write the same dummy record, so one pass of 512 records covers almost every tile (unions hold ~450 atoms at LJ liquid density).
  constexpr int STG = 512 / LJ_TILE_THREADS;
  int tt[STG], jj[STG];
#pragma unroll
  for(int u = 0; u < STG; u++) { tt[u] = min(u * LJ_TILE_THREADS + tid, ncand); jj[u] = stream_load(cl + tt[u]); }
  // ---- my atom and my slice of its neighbor row (wave w takes k in [k0,k1))
  int i = lane < cnt ? binned[first + lane] : -1;
  // rows are padded to a multiple of 4 (NB_ROW_PAD): the wave slices are multiples of 4, run as trips of UNR pairs plus,
  // where 4 rows remain, one half trip
  constexpr int QR = 4;
  const int per = ((kmax / QR + LJ_TILE_WAVES - 1) / LJ_TILE_WAVES) * QR;
  const int k0 = min(wv * per, kmax), k1 = min(k0 + per, kmax);
  // (profiling build, ablate 4: every tile walks the rows of one of 64 tiles — the slot stream comes from the L2 instead of the HBM; results invalid)
  const unsigned short* __restrict__ np = nl16 + ((size_t)((ablate & 4) ? (tile & 63) : tile) * maxneighs + k0) * 64 + lane;
  int s[UNR];
#pragma unroll
  for(int u = 0; u < UNR; u++) s[u] = 0;
  if(k0 < k1) {                                   // (a slice of 4 rows reads 4 rows of padding / of the next slice: in bounds, unused)
#pragma unroll
    for(int u = 0; u < UNR; u++) s[u] = stream_load(np + u * 64);
  }
  real4 pp[STG];
  if(packed) {
#pragma unroll
    for(int u = 0; u < STG; u++) pp[u] = x[jj[u] & MMD_SRC_MASK];
  } else if(ghosted) {
#pragma unroll
    for(int u = 0; u < STG; u++) pp[u] = ghost_resolved(x, jj[u], nlocal, nall, G);
  } else {
#pragma unroll
    for(int u = 0; u < STG; u++) pp[u] = x[jj[u]];
  }
  if(i >= nlocal) i = -1;
  const real4 xi = x[i >= 0 ? i : 0];
  if(!(ablate & 1)) {
    if(packed) {
#pragma unroll
      for(int u = 0; u < STG; u++) pp[u] = ghost_shifted(pp[u], jj[u], G);
    }
#pragma unroll
    for(int u = 0; u < STG; u++) { sp[3 * tt[u]] = pp[u].x; sp[3 * tt[u] + 1] = pp[u].y; sp[3 * tt[u] + 2] = pp[u].z; }
    for(int t0 = 512; t0 <= ncand; t0 += LJ_TILE_THREADS) {        // (a union beyond 512 candidates: 2 % of the tiles at -s 80)
      const int t = min(t0 + tid, ncand), j = cl[t];
      const real4 p = packed ? ghost_shifted(x[j & MMD_SRC_MASK], j, G) : (ghosted ? ghost_resolved(x, j, nlocal, nall, G) : x[j]);
      sp[3 * t] = p.x; sp[3 * t + 1] = p.y; sp[3 * t + 2] = p.z;
    }
  }
  real vx0 = 0, vy0 = 0, vz0 = 0;          // FUSE: the velocity travels under the pair loop (wave 0 integrates)
  if(FUSE && wv == 0 && i >= 0) { vx0 = v[3 * (size_t)i + 0]; vy0 = v[3 * (size_t)i + 1]; vz0 = v[3 * (size_t)i + 2]; }
  __syncthreads();
  drain_loads();

  real fx = 0, fy = 0, fz = 0;
  double e_acc = 0, v_acc = 0;
  // force = 48 eps sr6 (sr6 - 1/2) sr2 with sr6 = s6 A, A = sr2^3  ==  [48 eps s6] * (A sr2) * (s6 A - 1/2):
  // the bracket is uniform and applied once after the loop
  const real c_out = (real)48.0 * P.epsilon * P.sigma6;
  // one trip: U pairs of every lane. Positions come from LDS (s[u] IS the record's address); once the addresses are
  // consumed the same registers receive the slots of the next trip, which travel under this trip's arithmetic
  auto trip = [&](auto nu, int k) {
    constexpr int U = decltype(nu)::value;
    real xj[U], yj[U], zj[U];
    if(ablate & 8) {        // (profiling build: a slot pattern without LDS bank conflicts — 32 consecutive records per lane group; results invalid)
#pragma unroll
      for(int u = 0; u < U; u++) s[u] = (int)(((unsigned)(lane + 7 * (k + u)) & 255u) * (unsigned)(3 * sizeof(real)));
    }
#pragma unroll
    for(int u = 0; u < U; u++) lds_read3<(LJ_TILE_RD == 3 ? 1 : 0)>((unsigned)s[u], xj[u], yj[u], zj[u]);
    np += U * 64;
    if(k + U < k1) {
#pragma unroll
      for(int u = 0; u < UNR; u++) s[u] = stream_load(np + u * 64);
    }
    // every multiply-add is written as an explicit fma: with -ffp-contract=fast the compiler would otherwise be
    // free to pick WHICH product of a sum it fuses, and the instantiations of this template must round alike
    // pairs are worked off in groups of four (register pressure: the second group's positions wait in their LDS-read registers)
    constexpr bool BATCH = (U % 4) == 0 && sizeof(real) == 8;
    constexpr int GRP = BATCH ? 4 : 1;
    static_for_groups<U, GRP>([&](auto g0c) {
      constexpr int g0 = decltype(g0c)::value;
      real dx[GRP], dy[GRP], dz[GRP], rsq[GRP], sr2[GRP];
#pragma unroll
      for(int q = 0; q < GRP; q++) {
        const int u = g0 + q;
        dx[q] = xi.x - xj[u]; dy[q] = xi.y - yj[u]; dz[q] = xi.z - zj[u];
        rsq[q] = fma_r(dz[q], dz[q], fma_r(dy[q], dy[q], dx[q] * dx[q]));
      }
      if(BATCH) {
        // ONE reciprocal per FOUR pairs (the quarter-rate v_rcp_f64 and its Newton step are the costliest part of a pair):
        // r = 1/(abcd) refined once, then 1/a = b (cd r), 1/b = a (cd r), 1/c = d (ab r), 1/d = c (ab r). The product stays far
        // inside the double range even when all four entries are the dummy atom (rsq 3e30); not so in float, which keeps
        // one reciprocal per pair.
        const real p = rsq[0] * rsq[GRP > 1 ? 1 : 0], q = rsq[GRP > 2 ? 2 : 0] * rsq[GRP > 3 ? 3 : 0];
        const real r = recip_fast(p * q);
        const real rp = q * r, rq = p * r;                 // 1/(ab), 1/(cd)
        sr2[0] = rsq[GRP > 1 ? 1 : 0] * rp; sr2[GRP > 1 ? 1 : 0] = rsq[0] * rp;
        sr2[GRP > 2 ? 2 : 0] = rsq[GRP > 3 ? 3 : 0] * rq; sr2[GRP > 3 ? 3 : 0] = rsq[GRP > 2 ? 2 : 0] * rq;
      } else {
#pragma unroll
        for(int q = 0; q < GRP; q++) sr2[q] = recip_fast(rsq[q]);
      }
#pragma unroll
      for(int q = 0; q < GRP; q++) {
        const bool in = rsq[q] < P.cutforcesq;
        const real s2 = keep_if(in, sr2[q]);         // out of range => everything below is 0
        const real A = (s2 * s2) * s2;
        const real t = fma_r(A, P.sigma6, (real)-0.5);
        const real fs = (A * s2) * t;               // force / c_out
        fx = fma_r(dx[q], fs, fx); fy = fma_r(dy[q], fs, fy); fz = fma_r(dz[q], fs, fz);
        if(EV) {
          const real sr6 = A * P.sigma6;
          e_acc = __builtin_fma((double)(sr6 * (sr6 - (real)1.0)), (double)P.epsilon, e_acc);
          v_acc = __builtin_fma((double)rsq[q], (double)fs, v_acc);
        }
      }
    });
  };
  int k = k0;
  for(; k + UNR <= k1; k += UNR) trip(std::integral_constant<int, UNR>{}, k);
  if(UNR > QR && k < k1) trip(std::integral_constant<int, QR>{}, k);
  v_acc *= (double)c_out;
  // combine the wave slices, THEN apply the folded constant: (a + b) * c leaves the compiler no multiply-add to
  // contract, so every instantiation of this kernel rounds the force identically
  if(wv > 0) { real* d = s_f + 3 * 64 * (wv - 1); d[lane] = fx; d[64 + lane] = fy; d[128 + lane] = fz; }
  __syncthreads();
  if(wv == 0 && i >= 0) {
#pragma unroll
    for(int q = 0; q < LJ_TILE_WAVES - 1; q++) { const real* d = s_f + 3 * 64 * q; fx += d[lane]; fy += d[64 + lane]; fz += d[128 + lane]; }
    fx *= c_out; fy *= c_out; fz *= c_out;
    // (a fused step consumes the force here; f[] is next read after the unfused thermo / last step, which stores it)
    // (FUSE = 2, the LAST step of a run: finalIntegrate only — the state the caller gets back is that of a whole step, forces included)
    if(FUSE != 1) { out_store(f + 3 * (size_t)i + 0, fx); out_store(f + 3 * (size_t)i + 1, fy); out_store(f + 3 * (size_t)i + 2, fz); }
    if(FUSE) {
      real vx = vx0, vy = vy0, vz = vz0;
      vx = mul_add_unfused(dtforce, fx, vx); vy = mul_add_unfused(dtforce, fy, vy); vz = mul_add_unfused(dtforce, fz, vz);
      if(FUSE == 1) { vx = mul_add_unfused(dtforce, fx, vx); vy = mul_add_unfused(dtforce, fy, vy); vz = mul_add_unfused(dtforce, fz, vz); }
      out_store(v + 3 * (size_t)i + 0, vx); out_store(v + 3 * (size_t)i + 1, vy); out_store(v + 3 * (size_t)i + 2, vz);
      if(FUSE == 1) {
        real* xo = (real*)(xnew + i);
        out_store(xo + 0, mul_add_unfused(dt, vx, xi.x)); out_store(xo + 1, mul_add_unfused(dt, vy, xi.y)); out_store(xo + 2, mul_add_unfused(dt, vz, xi.z)); out_store(xo + 3, xi.w);
      }
    }
  }
  if(EV) {
    if(i < 0) { e_acc = 0; v_acc = 0; }
    const double es = block_sum(e_acc, s_red);
    const double vs = block_sum(v_acc, s_red);
    if(tid == 0) { partials[2 * (size_t)tile] = es; partials[2 * (size_t)tile + 1] = vs; }
  }
  // (workgroups are dispatched in index order: the launch ends with one of its last few thousand)
  // (wv == 0 && lane == 0, not tid == 0: `tid` is dead by now, and keeping it alive through the pair loop cost the fused instantiation 4 VGPRs — 100 instead of
  //  96, one wavefront per SIMD less: -2.4 % on the step)
  if(SP.clk != nullptr && blockIdx.x + (unsigned)FCLK_TAIL >= gridDim.x && wv == 0 && lane == 0) SP.clk[8 + (gridDim.x - 1 - blockIdx.x)] = (unsigned long long)wall_clock64();
}

// ---- half neighbor list: compute_halfneigh_threaded<EVFLAG,GHOST_NEWTON> (ref/force_lj.cpp:271-357) ------
// f was zeroed over owned+ghost atoms beforehand; f_j is scattered with native FP atomics.
template <int EV, int GN, int UNIFORM>
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
  const int wg = xcd_work_item((nlocal + MMD_BLOCK - 1) / MMD_BLOCK);
  if(wg < 0) return;
  const int i = wg * MMD_BLOCK + threadIdx.x;
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
  for(int k = 0; k < kmax; k += LJ_UNR) {
    int j[LJ_UNR];
    real4 xj[LJ_UNR];
#pragma unroll
    for(int u = 0; u < LJ_UNR; u++) j[u] = np[(size_t)(k + u) * 64];
#pragma unroll
    for(int u = 0; u < LJ_UNR; u++) xj[u] = x[j[u]];
#pragma unroll
    for(int u = 0; u < LJ_UNR; u++) {
      const real dx = xi.x - xj[u].x, dy = xi.y - xj[u].y, dz = xi.z - xj[u].z;
      const real rsq = dx * dx + dy * dy + dz * dz;
      real cut, s6, eps;
      if(UNIFORM) { cut = P.cutforcesq; s6 = P.sigma6; eps = P.epsilon; }
      else { const int tij = ti + (int)xj[u].w; cut = s_cut[tij]; s6 = s_s6[tij]; eps = s_eps[tij]; }
      if(rsq < cut) {
        const real sr2 = recip(rsq);
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
    if(threadIdx.x == 0) { partials[2 * (size_t)wg] = es; partials[2 * (size_t)wg + 1] = vs; }
  }
}

// ---- half neighbor list, tile form -------------------------------------------------------------------------------------
// Newton's third law with the scatter kept ON CHIP: next to the {x,y,z} records of the tile's candidate union the LDS holds
// one force accumulator per candidate; every in-range pair adds to its own atom in registers and to the partner's
// accumulator with a native LDS floating-point atomic (ds_add_f64 / ds_add_f32), and only at the end of the tile do the
// ~450 accumulators (and the 64 owned atoms) go to global memory with one atomic each: ~1.5 k global atomics per tile
// instead of the ~6.5 k of k_lj_half. Same physics/conventions as compute_halfneigh_threaded<EVFLAG,GHOST_NEWTON>
// (ref/force_lj.cpp:271-357): without ghost newton the partner gets no force when it is a ghost and the pair counts half
// in energy and virial. f was zeroed over owned+ghost atoms beforehand. Dynamic LDS:
//   [positions: pos_bytes][accumulators: pos_bytes][wave-slice forces: 3*64 reals][16 doubles][EV && !GN: ghost flag per slot]
// Single precision (round 4): the LDS atomic path, not the arithmetic, is what the half-list kernel waits for (ds_add_f64 / ds_add_u64 /
// ds_add_u32 all retire ~5 lanes per clock and CU, ds_add_f32 an eighth of that), so a pair's x and y shares travel in ONE 64-bit integer
// atomic: fixed point with 20 fractional bits, x in the high and y in the low 32 bits of V = qx * 2^32 + qy (as a two's-complement sum the
// fields add independently as long as neither leaves int32: |sum| < 2048). z keeps its double accumulator: 2 instead of 3 LDS atomics per
// pair. A candidate collects from at most the 64 atoms of the tile, so shares below 2^4 (+ the own force, below 1000) can never overflow a field; a share beyond that
// (a pair closer than ~0.8 sigma: far up the repulsive wall) goes to global memory directly. Resolution 2^-20 of force / c_out per
// share, rounded to nearest (unbiased): ~1e-6 after a row of adds, against ~1e-7 for float sums — far inside the SP parity rule.
#ifndef LJH_PACK
#define LJH_PACK (MMD_PRECISION == 1)
#endif
#define LJH_FIX_SCALE 1048576.0f          // 2^20
// a field holds |sum| < 2048: at most 64 shares below 16 (each rounded to 2^-20: <= 1024 together) + the own force below 1000 stay inside it
#define LJH_FIX_SHARE 16.0f               // largest |share| that takes the packed path
#define LJH_FIX_SUM 1000.0f               // largest |own force / c_out| folded into a packed accumulator
__device__ __forceinline__ unsigned long long ljh_pack_xy(float px, float py)
{
  const long long qx = (long long)__float2int_rn(px * LJH_FIX_SCALE), qy = (long long)__float2int_rn(py * LJH_FIX_SCALE);
  return (unsigned long long)((qx << 32) + qy);
}
__device__ __forceinline__ void ljh_unpack_xy(unsigned long long v, double& x, double& y)
{
  const long long V = (long long)v;
  const long long lo = (long long)(int)(unsigned)v;                 // sign-extended low field
  x = (double)((V - lo) >> 32) * (1.0 / (double)LJH_FIX_SCALE);
  y = (double)lo * (1.0 / (double)LJH_FIX_SCALE);
}

// SRC=1 (one rank): the candidates come from the second list the build wrote, a ghost named by its owner + image code (GhostResolve, tile_lds.hpp):
// it is staged from the owner's current position (no Comm::communicate launch on the step) and, with ghost newton, its share goes to the owner
// at the flush without a look-up; without ghost newton it is flagged and gets none.
template <int EV, int GN, int SRC = 0>
__global__ __launch_bounds__(128) void k_lj_half_tile(
    const real4* __restrict__ x, const int* __restrict__ binned, const int* __restrict__ tile_first,
    const int* __restrict__ tile_cnt, const int* __restrict__ tile_max, const int* __restrict__ tile_cand, const int* __restrict__ tile_ncand, int cstride, int ntiles,
    const int* __restrict__ tile_list,
    const unsigned short* __restrict__ nl16, const unsigned short* __restrict__ tile_self, int nlocal, int nall, int maxneighs, int pos_bytes,
    LJParams P, real* __restrict__ f, double* __restrict__ partials, int ablate_arg, const int* __restrict__ ghost_root,
    const int* __restrict__ cand_src, const real* __restrict__ box_dev, const int* __restrict__ tile_ghost)
{
  const int ablate = MMD_ABLATE(ablate_arg);     // profiling switches: compiled out of the shipped library (mmd_internal.hpp)
  constexpr int UNR = 8, NT = 128, STG = 4;
  extern __shared__ __align__(16) unsigned char s_raw[];
  real* sp = (real*)s_raw;
  // accumulators are doubles in both precisions: ds_add_f64 runs ~8x faster than ds_add_f32 on this part (measured: 0.10 vs
  // 0.81 ms of LDS atomics at -s 80), and the per-candidate sums lose nothing before the single rounding at the flush
  double* s_acc = (double*)(s_raw + pos_bytes);
  const int acc_bytes = pos_bytes * (int)(sizeof(double) / sizeof(real));
  real* s_f = (real*)(s_raw + (size_t)pos_bytes + acc_bytes);
  double* s_red = (double*)(s_raw + (size_t)pos_bytes + acc_bytes + lj_tile_sf_bytes(2));
  // the candidates' atom indices wait in LDS for the flush: read from global memory there, every atomic of a lane would queue
  // behind a load round trip (s_waitcnt vmcnt counts loads and atomics in one queue)
  int* s_idx = (int*)(s_red + 16);
  unsigned char* s_ghost = (unsigned char*)(s_idx + (((size_t)pos_bytes / (3 * sizeof(real)) + 3) & ~(size_t)3));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int witem = xcd_work_item(ntiles);
  if(witem < 0) return;
  const int tile = tile_list ? tile_list[witem] : witem;
  // three round trips (see k_lj_full_tile): header scalars; candidate indices + own atom index + first slots; positions
  const int ncand = tile_ncand[tile], cnt = tile_cnt[tile], first = tile_first[tile], kmax = tile_max[tile];
  const bool packed = SRC != 0 && tile_ghost[tile] != 0;      // (the second list exists for tiles with a ghost candidate only; workgroup-uniform)
  const int* __restrict__ cl = (packed ? cand_src : tile_cand) + (size_t)tile * cstride;
  // what the flush needs of a candidate: the atom whose force collects its share (packed: the owner; a ghost without ghost newton: nobody = nall)
  auto flush_index = [&](int j) { return !packed ? j : ((GN || ((unsigned)j >> MMD_SRC_BITS) == 0u) ? (j & MMD_SRC_MASK) : nall); };
  auto is_ghost = [&](int j) { return packed ? ((unsigned)j >> MMD_SRC_BITS) != 0u : j >= nlocal; };
  int tt[STG], jj[STG];
#pragma unroll
  for(int u = 0; u < STG; u++) { tt[u] = min(u * NT + tid, ncand); jj[u] = stream_load(cl + tt[u]); }
  int i = lane < cnt ? binned[first + lane] : -1;
  constexpr int QR = 4;                                     // rows are padded to 4: trips of UNR pairs + one half trip
  const int per = ((kmax / QR + 1) / 2) * QR;
  const int k0 = min(wv * per, kmax), k1 = min(k0 + per, kmax);
  const unsigned short* __restrict__ np = nl16 + ((size_t)tile * maxneighs + k0) * 64 + lane;
  int s[UNR];
#pragma unroll
  for(int u = 0; u < UNR; u++) s[u] = 0;
  if(k0 < k1) {
#pragma unroll
    for(int u = 0; u < UNR; u++) s[u] = stream_load(np + u * 64);
  }
  real4 pp[STG];
#pragma unroll
  for(int u = 0; u < STG; u++) pp[u] = x[packed ? (jj[u] & MMD_SRC_MASK) : jj[u]];
  if(i >= nlocal) i = -1;
  const real4 xi = x[i >= 0 ? i : 0];
  if(packed) {
#pragma unroll
    for(int u = 0; u < STG; u++) pp[u] = ghost_shifted(pp[u], jj[u], box_dev);
  }
#pragma unroll
  for(int u = 0; u < STG; u++) {                              // positions in, accumulators cleared
    sp[3 * tt[u]] = pp[u].x; sp[3 * tt[u] + 1] = pp[u].y; sp[3 * tt[u] + 2] = pp[u].z;
    s_acc[3 * tt[u]] = 0; s_acc[3 * tt[u] + 1] = 0; s_acc[3 * tt[u] + 2] = 0;
    s_idx[tt[u]] = flush_index(jj[u]);
    if(EV && !GN) s_ghost[tt[u]] = is_ghost(jj[u]) ? 1 : 0;
  }
  for(int t0 = STG * NT; t0 <= ncand; t0 += NT) {             // (a union beyond STG * NT candidates: rare)
    const int t = min(t0 + tid, ncand), j = cl[t];
    const real4 p = packed ? ghost_shifted(x[j & MMD_SRC_MASK], j, box_dev) : x[j];
    sp[3 * t] = p.x; sp[3 * t + 1] = p.y; sp[3 * t + 2] = p.z;
    s_acc[3 * t] = 0; s_acc[3 * t + 1] = 0; s_acc[3 * t + 2] = 0;
    s_idx[t] = flush_index(j);
    if(EV && !GN) s_ghost[t] = is_ghost(j) ? 1 : 0;
  }
  __syncthreads();
  drain_loads();

  real fx = 0, fy = 0, fz = 0;
  double e_acc = 0, v_acc = 0;
  const real c_out = (real)48.0 * P.epsilon * P.sigma6;      // folded constant, applied when the sums leave the chip
  auto trip = [&](auto nu, int k) {
    constexpr int U = decltype(nu)::value;
    real xj[U], yj[U], zj[U];
    int sc[U];
    unsigned farbits = 0;            // LJH_PACK: pairs of this trip whose share does not fit the packed accumulator's field
#pragma unroll
    for(int u = 0; u < U; u++) { sc[u] = s[u]; lds_read3<LJH_RD>((unsigned)s[u], xj[u], yj[u], zj[u]); }
    np += U * 64;
    if(k + U < k1) {
#pragma unroll
      for(int u = 0; u < UNR; u++) s[u] = stream_load(np + u * 64);
    }
    // groups of four pairs share ONE reciprocal in double precision (see k_lj_full_tile)
    constexpr bool BATCH = sizeof(real) == 8;
    constexpr int GRP = BATCH ? 4 : 1;
#pragma unroll
    for(int g0 = 0; g0 < U; g0 += GRP) {
      real dx[GRP], dy[GRP], dz[GRP], rsq[GRP], sr2v[GRP];
#pragma unroll
      for(int q = 0; q < GRP; q++) {
        const int u = g0 + q;
        dx[q] = xi.x - xj[u]; dy[q] = xi.y - yj[u]; dz[q] = xi.z - zj[u];
        rsq[q] = fma_r(dz[q], dz[q], fma_r(dy[q], dy[q], dx[q] * dx[q]));
      }
      if(BATCH) {
        const real p = rsq[0] * rsq[GRP > 1 ? 1 : 0], q2 = rsq[GRP > 2 ? 2 : 0] * rsq[GRP > 3 ? 3 : 0];
        const real r = recip_fast(p * q2);
        const real rp = q2 * r, rq = p * r;
        sr2v[0] = rsq[GRP > 1 ? 1 : 0] * rp; sr2v[GRP > 1 ? 1 : 0] = rsq[0] * rp;
        sr2v[GRP > 2 ? 2 : 0] = rsq[GRP > 3 ? 3 : 0] * rq; sr2v[GRP > 3 ? 3 : 0] = rsq[GRP > 2 ? 2 : 0] * rq;
      } else {
#pragma unroll
        for(int q = 0; q < GRP; q++) sr2v[q] = recip_fast(rsq[q]);
      }
#pragma unroll
      for(int q = 0; q < GRP; q++) {
        const int u = g0 + q;
        // Round 6: the pair's arithmetic is branch-free (an out-of-range pair gets sr2 = 0: every product below is an exact zero) and only the LDS atomics sit in an
        // exec region — one, not the three nested ones of `if(in) { ...; if(share fits) {...} else {...} }`, which cost ~19 scalar instructions per pair.
        const bool in = rsq[q] < P.cutforcesq;                // (the atomics below stay off out-of-range pairs: keeps the padded lanes off the dummy slot's accumulator)
        const real sr2 = keep_if(in, sr2v[q]);
        const real A = (sr2 * sr2) * sr2;
        const real fs = (A * sr2) * fma_r(A, P.sigma6, (real)-0.5);          // force / c_out
        const real px = dx[q] * fs, py = dy[q] * fs, pz = dz[q] * fs;
        fx += px; fy += py; fz += pz;
        // the partner's accumulator collects +p, negated at the flush (sc = slot * 3 reals in bytes -> slot * 3 doubles)
        double* a = (double*)((unsigned char*)s_acc + sc[u] * (int)(sizeof(double) / sizeof(real)));
        if(ablate & 4) {            // (profiling only: integer LDS atomics of the same width / of 32 bits)
          if(in) { atomicAdd((unsigned long long*)a + 0, (unsigned long long)__double_as_longlong((double)px)); atomicAdd((unsigned long long*)a + 1, (unsigned long long)__double_as_longlong((double)py));
                   atomicAdd((unsigned long long*)a + 2, (unsigned long long)__double_as_longlong((double)pz)); }
        } else if(ablate & 8) {
          if(in) {
            if(ablate & 16) { unsafeAtomicAdd((float*)a + 0, (float)px); unsafeAtomicAdd((float*)a + 2, (float)py); unsafeAtomicAdd((float*)a + 4, (float)pz); }
            else { atomicAdd((unsigned*)a + 0, (unsigned)__float_as_int((float)px)); atomicAdd((unsigned*)a + 2, (unsigned)__float_as_int((float)py)); atomicAdd((unsigned*)a + 4, (unsigned)__float_as_int((float)pz)); }
          }
        } else
        if(LJH_PACK && !(ablate & 1)) {
          // (a share that does not fit its fixed-point field — a pair far up the repulsive wall, practically never — is only MARKED here and leaves the chip behind the trip)
          const bool fits = fmaxf(fabsf((float)px), fabsf((float)py)) < LJH_FIX_SHARE;
          if(in && fits) {
            atomicAdd((unsigned long long*)a, ljh_pack_xy((float)px, (float)py));
            if(!(ablate & 32)) unsafeAtomicAdd(a + 1, (double)pz);          // (ablate: profiling only)
          }
          farbits |= (unsigned)(in && !fits) << u;
        } else
        if(!(ablate & 1)) { if(in) { unsafeAtomicAdd(a + 0, (double)px); unsafeAtomicAdd(a + 1, (double)py); unsafeAtomicAdd(a + 2, (double)pz); } }   // (ablate: profiling only)
        if(EV) {
          real scale = (real)1.0;
          if(!GN) scale = s_ghost[(unsigned)sc[u] / (3u * (unsigned)sizeof(real))] ? (real)0.5 : (real)1.0;
          const real sr6 = A * P.sigma6;
          e_acc += (double)(scale * ((real)4.0 * sr6 * (sr6 - (real)1.0)) * P.epsilon);
          v_acc += (double)(scale * rsq[q] * fs);
        }
      }
    }
    if(LJH_PACK && __builtin_amdgcn_ballot_w64(farbits != 0u) != 0ull) {
      // the marked pairs, again, one by one (same operations: the same share): straight to the partner's force in global memory
#pragma unroll
      for(int u = 0; u < U; u++) {
        if((farbits >> u) & 1u) {
          const real dx = xi.x - xj[u], dy = xi.y - yj[u], dz = xi.z - zj[u];
          const real rsq = fma_r(dz, dz, fma_r(dy, dy, dx * dx));
          const real sr2 = recip_fast(rsq);
          const real A = (sr2 * sr2) * sr2;
          const real fs = (A * sr2) * fma_r(A, P.sigma6, (real)-0.5);
          const real px = dx * fs, py = dy * fs, pz = dz * fs;
          int j = s_idx[(unsigned)sc[u] / (3u * (unsigned)sizeof(real))];
          if(GN && ghost_root != nullptr && j >= nlocal) j = ghost_root[j - nlocal];
          if(GN || j < nlocal) {
            unsafeAtomicAdd(f + 3 * (size_t)j + 0, (real)(-(px * c_out))); unsafeAtomicAdd(f + 3 * (size_t)j + 1, (real)(-(py * c_out)));
            unsafeAtomicAdd(f + 3 * (size_t)j + 2, (real)(-(pz * c_out)));
          }
        }
      }
    }
  };
  int k = k0;
  for(; k + UNR <= k1; k += UNR) trip(std::integral_constant<int, UNR>{}, k);
  if(k < k1) trip(std::integral_constant<int, QR>{}, k);
  v_acc *= (double)c_out;
  if(wv > 0) { s_f[lane] = fx; s_f[64 + lane] = fy; s_f[128 + lane] = fz; }
  __syncthreads();                                          // every pair of the tile has been accumulated
  if(wv == 0 && i >= 0) {
    fx += s_f[lane]; fy += s_f[64 + lane]; fz += s_f[128 + lane];
    const unsigned own = tile_self[(size_t)tile * 64 + lane];
    if(LJH_PACK && own != 0xffffu && fmaxf(fabsf((float)fx), fabsf((float)fy)) < LJH_FIX_SUM) {
      ((unsigned long long*)s_acc)[3 * own] += ljh_pack_xy(-(float)fx, -(float)fy);
      s_acc[3 * own + 1] -= (double)fz;
    } else
    if(!LJH_PACK && own != 0xffffu) {          // the atom is one of the tile's candidates: its own force joins that accumulator (sign: see flush)
      s_acc[3 * own] -= (double)fx; s_acc[3 * own + 1] -= (double)fy; s_acc[3 * own + 2] -= (double)fz;
    } else {
      real* fi = f + 3 * (size_t)i;
      unsafeAtomicAdd(fi + 0, fx * c_out); unsafeAtomicAdd(fi + 1, fy * c_out); unsafeAtomicAdd(fi + 2, fz * c_out);
    }
  }
  __syncthreads();
  // partners: ONE global atomic per component and candidate. Lanes walk the accumulators in memory order (3 doubles per candidate):
  // candidates are runs of consecutive atoms, so one wave instruction covers a few whole lines of f instead of a 24-byte stride
  for(int e = tid; e < 3 * ncand && !(ablate & 2); e += NT) {
    const int t = (int)(((unsigned)e * 43691u) >> 17);      // e / 3 (exact below 98304)
    int j = s_idx[t];
    double a = s_acc[e];
    if(LJH_PACK) {                       // record = {x and y packed, z as a double, unused}
      const int c = e - 3 * t;
      double ax, ay;
      ljh_unpack_xy(((const unsigned long long*)s_acc)[3 * t], ax, ay);
      a = c == 0 ? ax : (c == 1 ? ay : s_acc[3 * t + 1]);
    }
    // one rank: a ghost is an image of an owned atom, its share goes straight to the owner (Comm::reverse_communicate folded in)
    if(GN && ghost_root != nullptr && j >= nlocal) j = ghost_root[j - nlocal];
    if((GN || j < nlocal) && a != 0) unsafeAtomicAdd(f + 3 * (size_t)j + (e - 3 * t), (real)(-(a * (double)c_out)));
  }
  if(EV) {
    if(i < 0) { e_acc = 0; v_acc = 0; }
    const double es = block_sum(e_acc, s_red);
    const double vs = block_sum(v_acc, s_red);
    if(tid == 0) { partials[2 * (size_t)tile] = es; partials[2 * (size_t)tile + 1] = vs; }
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
  HIP_TRY(hipSetDevice(h->device));
  h->style = 0;
  h->ntypes = ntypes;
  const int n2 = ntypes * ntypes;
  h->lj_uniform = true;
  for(int i = 1; i < n2; i++)
    if(cutforcesq[i] != cutforcesq[0] || sigma6[i] != sigma6[0] || epsilon[i] != epsilon[0]) h->lj_uniform = false;
  if(!h->lj_uniform && n2 > LJ_MAX_TYPES2) { mmd_set_error("mmd_force_lj_setup: at most 16 atom types with different parameters are supported"); return -1; }
  h->lj.cutforcesq = cutforcesq[0]; h->lj.sigma6 = sigma6[0]; h->lj.epsilon = epsilon[0];
  h->h_cutforcesq.assign(cutforcesq, cutforcesq + n2);
  MMD_TRY(h->lj_tables.ensure((size_t)3 * n2, false, h->stream));
  HIP_TRY(hipMemcpyAsync(h->lj_tables.p, cutforcesq, n2 * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->lj_tables.p + n2, sigma6, n2 * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->lj_tables.p + 2 * n2, epsilon, n2 * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  return 0;
}

template <int EV, int UNIFORM>
static void launch_full(mmd_handle* h, int nblocks, const LJTables& T)
{
  hipLaunchKernelGGL((k_lj_full<EV, UNIFORM>), dim3(xcd_grid(nblocks)), dim3(MMD_BLOCK), 0, h->stream, h->x.p, h->neigh.p,
                     h->wave_max.p, h->nlocal, h->maxneighs, h->lj, T, h->f.p, h->partials.p, h->opt_ablate);
}
template <int EV, int GN, int UNIFORM>
static void launch_half(mmd_handle* h, int nblocks, const LJTables& T)
{
  hipLaunchKernelGGL((k_lj_half<EV, GN, UNIFORM>), dim3(xcd_grid(nblocks)), dim3(MMD_BLOCK), 0, h->stream, h->x.p, h->neigh.p,
                     h->wave_max.p, h->nlocal, h->maxneighs, h->lj, T, h->f.p, h->partials.p);
}

// bytes of the position records in the tile kernel's LDS (rounded to 16)
static size_t lj_tile_pos_bytes(const mmd_handle* h) { return (((size_t)3 * (h->tile_cmax + 2) * sizeof(real)) + 15) & ~(size_t)15; }

// dynamic LDS of k_lj_half_tile: positions, double accumulators, wave-slice forces, reduction scratch, candidate indices, ghost flags
static size_t lj_half_tile_lds(const mmd_handle* h)
{
  const size_t pos_bytes = lj_tile_pos_bytes(h);
  return pos_bytes + pos_bytes * (sizeof(double) / sizeof(real)) + lj_tile_sf_bytes(2) + 16 * sizeof(double) +
         (pos_bytes / (3 * sizeof(real)) + 4) * sizeof(int) + (size_t)(h->tile_cmax + 2) + 16;
}

int mmd_lj_tiles_available(mmd_handle* h)
{
  return h->style == 0 && !h->halfneigh && h->tiles_ready && h->opt_tiles && h->lj_uniform && lj_tile_pos_bytes(h) <= 60 * 1024 && h->neigh_nlocal == h->nlocal;
}

// half lists in tile form (k_lj_half_tile): uniform type tables, device-built list, positions + accumulators fit 64 KB of LDS
int mmd_lj_half_tiles_available(mmd_handle* h)
{
  return h->style == 0 && h->halfneigh && h->tiles_ready && h->opt_tiles && h->lj_uniform && !h->opt_lj_original &&
         lj_half_tile_lds(h) <= 64 * 1024 && h->neigh_nlocal == h->nlocal;
}

// the production tile kernel can carry finalIntegrate(n) + initialIntegrate(n+1) (no energy/virial on that step)
int mmd_lj_can_fuse_integrate(mmd_handle* h)
{
  return mmd_lj_tiles_available(h);
}

// launch the tile kernel over `count` tiles: tile ids from `list` (device) or 0..count-1
static int launch_tiles(mmd_handle* h, int evflag, const int* list, int count)
{
  if(count <= 0) return 0;
  const size_t pos_bytes = lj_tile_pos_bytes(h);
  const int nlocal = h->nlocal;
  const int ev = evflag ? 1 : 0;
  bool launched = false;
  const int fz = h->fuse_now;                    // 0: force only, 1: + finalIntegrate + the next initialIntegrate, 2: + finalIntegrate (last step of a run)
  // the caller's event pair rides ON the dispatch (start/stop stamps of the kernel itself): timing a step's force kernel costs the
  // stream no marker packets (a bracketing hipEventRecord pair costs ~6 us per step, 15 % of a -s 32 step)
  hipEvent_t kev_a = h->launch_ev_a, kev_b = h->launch_ev_b;
  h->launch_ev_a = h->launch_ev_b = nullptr;
  GhostResolve G{nullptr, nullptr, nullptr, {h->prd[0], h->prd[1], h->prd[2]}, nullptr};
  if(h->resolve_now) { G.root = h->ghost_root.p; G.image = h->ghost_image.p; G.tile_ghost = h->tile_ghost.p; G.cand_src = h->cand_src_ready ? h->tile_cand_src.p : nullptr; }
  SpecLaunch SP = list == nullptr ? h->spec : SpecLaunch{nullptr, nullptr, nullptr, nullptr};
  // the launch's own clock (whole-list launches inside a run; a launch cancelled by the build's verdict is stamped again by the one that replaces it)
  SP.clk = nullptr;
  if(list == nullptr && h->in_run && h->fclk.p != nullptr) {
    // (a run longer than FCLK_SLOTS launches keeps the stamps of its first FCLK_SLOTS / 2 and, in a ring, of its LAST FCLK_SLOTS / 2 launches: every word of a record is
    //  a plain store of a later time than the one it replaces, so a re-used record needs no clearing)
    if(h->spec_clk_redo) { h->fclk_n--; h->spec_clk_redo = false; HIP_TRY(hipMemsetAsync(h->fclk.p + (size_t)FCLK_STRIDE * fclk_slot(h->fclk_n), 0, FCLK_STRIDE * sizeof(unsigned long long), h->stream)); }
    const int slot = fclk_slot(h->fclk_n);
    SP.clk = h->fclk.p + (size_t)FCLK_STRIDE * slot;
    h->fclk_sampled[slot] = kev_a != nullptr;          // (this launch also carries the event pair of the sampled clock)
    h->fclk_n++;
  }
  if(SP.gate != nullptr) { h->spec_launches++; h->spec_fused = fz == 1; }
#define TK(EVv, Fv) if(!launched && ev == EVv && fz == Fv) { launched = true;                                                       \
    hipExtLaunchKernelGGL((k_lj_full_tile<EVv, Fv>), dim3(xcd_grid(count)), dim3(64 * LJ_TILE_WAVES),                                \
                       pos_bytes + lj_tile_sf_bytes(LJ_TILE_WAVES) + 16 * sizeof(double), h->stream, kev_a, kev_b, 0, h->x.p,        \
                       h->binned.p, h->tile_first.p, h->tile_cnt.p, h->tile_max.p, h->tile_cand.p, h->tile_ncand.p, h->tile_cstride, \
                       count, list, h->nl16.p, nlocal, nlocal + h->nghost, h->maxneighs, (int)pos_bytes, h->lj, h->f.p,              \
                       h->partials.p, h->opt_ablate, h->v.p, h->x_alt.p, h->dt, h->dtforce, G, SP); }
  TK(0, 1); TK(0, 2); TK(0, 0); TK(1, 0);          // force + integrator of the next step / + finalIntegrate (last step of a run) / force only / force + energy and virial
#undef TK
  if(!launched) { mmd_set_error("tile force kernel: no instantiation for evflag %d with fused integrator %d", ev, fz); return -1; }
  HIP_TRY(hipGetLastError());
  return 0;
}

// two-part launch for the halo overlap: part 0 = interior tiles (no ghost candidates), part 1 = boundary tiles and,
// when evflag, the energy/virial sum over all tiles
int mmd_lj_compute_tiles_split(mmd_handle* h, int evflag, int part)
{
  MMD_TRY(mmd_order_tiles(h));
  MMD_TRY(h->partials.ensure((size_t)2 * h->ntiles + 8, false, h->stream));
  if(part == 0) return launch_tiles(h, evflag, h->tile_order.p, h->ntiles_interior);
  MMD_TRY(launch_tiles(h, evflag, h->tile_order.p + h->ntiles_interior, h->ntiles - h->ntiles_interior));
  if(evflag) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, h->stream, h->partials.p, h->ntiles, 2, h->d_result, 4.0, 0.5);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// ForceLJ::compute dispatch (ref/force_lj.cpp:72-113); eng_vdwl/virial (reference conventions) land in
// h->d_result[0..1] when evflag
static int lj_compute(mmd_handle* h, int evflag, double* eng, double* vir);
int mmd_lj_compute(mmd_handle* h, int evflag, double* eng, double* vir)
{
  const int rc = lj_compute(h, evflag, eng, vir);
  h->f_zeroed_n = 0;                    // (whatever was known about f is consumed)
  return rc;
}
static int lj_compute(mmd_handle* h, int evflag, double* eng, double* vir)
{
  if(h->neigh_nlocal != h->nlocal) { mmd_set_error("mmd_force_compute: neighbor list is stale (build or upload one first)"); return -1; }
  const int nlocal = h->nlocal;
  const int nblocks = div_up(nlocal, MMD_BLOCK);
  const int n2 = h->ntypes * h->ntypes;
  LJTables T{h->lj_tables.p, h->lj_tables.p + n2, h->lj_tables.p + 2 * n2, h->ntypes};
  MMD_TRY(h->partials.ensure((size_t)2 * nblocks + 8, false, h->stream));
  if(nlocal == 0) {
    // a rank without atoms (tiny boxes on several ranks) computes nothing but still takes part in the step: the forces of its ghosts travel
    // back through Comm::reverse_communicate (half lists with ghost newton) and must be zeros, its share of energy and virial is zero,
    // and a halo launched under this call has to be waited for
    if(h->halo_pending) { HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_halo_done, 0)); h->halo_pending = false; }
    if(h->halfneigh) MMD_TRY(mmd_zero_forces(h, h->nghost));
    if(evflag) HIP_TRY(hipMemsetAsync(h->d_result, 0, 2 * sizeof(double), h->stream));
    if(eng) *eng = 0;
    if(vir) *vir = 0;
    return 0;
  }
  const int ev = evflag ? 1 : 0, uni = h->lj_uniform ? 1 : 0;
  int nsum = nblocks;
  if(h->spec.gate != nullptr && (!mmd_lj_tiles_available(h) || evflag)) { mmd_set_error("Force::compute behind the build: only the gated tile launch may run there"); return -1; }
  if(mmd_lj_tiles_available(h)) {
    nsum = h->ntiles;
    MMD_TRY(h->partials.ensure((size_t)2 * nsum + 8, false, h->stream));
    MMD_TRY(launch_tiles(h, evflag, nullptr, h->ntiles));
  } else if(!h->halfneigh) {
    MMD_TRY(mmd_ensure_rows(h));
    MMD_TRY(h->partials.ensure((size_t)2 * nblocks + 8, false, h->stream));
#define F(EVv, Uv) if(ev == EVv && uni == Uv) launch_full<EVv, Uv>(h, nblocks, T)
    F(0, 0); F(0, 1); F(1, 0); F(1, 1);
#undef F
  } else if(mmd_lj_half_tiles_available(h)) {
    // half lists in tile form: on-chip scatter (k_lj_half_tile)
    // ref/force_lj.cpp:286-291 clears f first; inside Integrate::run on one rank the integrator has done that behind itself, and no
    // ghost receives a force (their shares go to the owners in the flush)
    if(!(h->fold_reverse_now && h->f_zeroed_n == nlocal)) MMD_TRY(mmd_zero_forces(h, nlocal + h->nghost));
    nsum = h->ntiles;
    MMD_TRY(h->partials.ensure((size_t)2 * nsum + 8, false, h->stream));
    const size_t pos_bytes = lj_tile_pos_bytes(h);
    const size_t lds = lj_half_tile_lds(h);
    const int gn = h->ghost_newton ? 1 : 0;
    // one rank, ghosts named by owner + image code (Integrate::run sets resolve_now on steps without re-neighboring; with ghost newton only together
    // with the folded reverse communication: the shares of the ghosts then go to their owners)
    const int* src_p = (h->resolve_now && h->cand_src_ready && !h->halo_pending && (!gn || h->fold_reverse_now)) ? (const int*)h->tile_cand_src.p : (const int*)nullptr;
    const real* box_p = nullptr;
    if(src_p != nullptr) { MMD_TRY(mmd_box_dev(h)); box_p = h->box_dev.p; }
#define HTS(EVv, Gv, Sv, LIST, CNT)                                                                                                   \
      hipLaunchKernelGGL((k_lj_half_tile<EVv, Gv, Sv>), dim3(xcd_grid(CNT)), dim3(128), lds, h->stream, h->x.p, h->binned.p,          \
                         h->tile_first.p, h->tile_cnt.p, h->tile_max.p, h->tile_cand.p, h->tile_ncand.p, h->tile_cstride, CNT, LIST,  \
                         h->nl16.p, h->tile_self.p, nlocal, nlocal + h->nghost, h->maxneighs, (int)pos_bytes, h->lj, h->f.p,            \
                         h->partials.p, h->opt_ablate, h->fold_reverse_now ? (const int*)h->ghost_root.p : (const int*)nullptr, src_p, box_p, (const int*)h->tile_ghost.p)
#define HT(EVv, Gv, LIST, CNT) if(ev == EVv && gn == Gv && (CNT) > 0) { if(src_p) HTS(EVv, Gv, 1, LIST, CNT); else HTS(EVv, Gv, 0, LIST, CNT); }
#define HT4(LIST, CNT) { HT(0, 0, LIST, CNT); HT(0, 1, LIST, CNT); HT(1, 0, LIST, CNT); HT(1, 1, LIST, CNT); }
    if(h->halo_pending) {
      // overlapped step (several ranks): interior tiles (no ghost among their candidates) run while the position halo is in
      // flight on the communication stream, the boundary tiles behind ev_halo_done
      MMD_TRY(mmd_order_tiles(h));
      const int n_int = h->ntiles_interior;
      HT4(h->tile_order.p, n_int);
      HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_halo_done, 0));
      HT4(h->tile_order.p + n_int, h->ntiles - n_int);
      h->halo_pending = false;
    } else
      HT4((const int*)nullptr, h->ntiles);
#undef HT4
#undef HT
#undef HTS
  } else {
    MMD_TRY(mmd_ensure_rows(h));
    MMD_TRY(mmd_zero_forces(h, nlocal + h->nghost));     // ref/force_lj.cpp:286-291
    const int gn = h->ghost_newton ? 1 : 0;
#define H(EVv, Gv, Uv) if(ev == EVv && gn == Gv && uni == Uv) launch_half<EVv, Gv, Uv>(h, nblocks, T)
    H(0, 0, 0); H(0, 0, 1); H(0, 1, 0); H(0, 1, 1); H(1, 0, 0); H(1, 0, 1); H(1, 1, 0); H(1, 1, 1);
#undef H
  }
  HIP_TRY(hipGetLastError());
  if(evflag) {
    // reference conventions: full lists visit both directions, then eng*4 and virial*0.5 (force_lj.cpp:441-442)
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, h->stream, h->partials.p, nsum, 2, h->d_result,
                       h->halfneigh ? 1.0 : 4.0, h->halfneigh ? 1.0 : 0.5);
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
