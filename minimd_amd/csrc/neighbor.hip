// minimd_amd/csrc/neighbor.hip — Neighbor::setup / binatoms / build (ref/neighbor.cpp) as HIP kernels.
//
// Design (MI355X-first, not a translation of the reference's per-atom stencil walk):
//  * bins keep the reference's geometry (binsize = prd/nbin, ghost margins, ref/neighbor.cpp:349-391) but
//    are numbered block-major: 2x2x2 bins form a block (~57 atoms at LJ liquid density = one wavefront)
//    and the bins of a block are consecutive, so a counting sort by bin id makes every block a contiguous
//    slice of `binned[]`, and the blocks of one x-row contiguous too.
//  * binning = atomic histogram that also hands out in-bin ranks (one atomic per run of equal bins in a
//    wavefront) + exclusive scan + plain scatter + in-bin index sort (=> deterministic order, identical to
//    the reference run single-threaded inside each bin).
//  * build = one wavefront per block: the candidate atoms of the (2R+1)^3 surrounding blocks (contiguous
//    slices of `binned`, coalesced loads) are held transposed in registers and every owned atom of the
//    block is tested against 64 candidates per VALU pass; hits are appended in candidate order with
//    ballot/mbcnt (see k_build). Bin slices of a block are located through a small LDS table.
//    Distance test is evaluated exactly like the reference (no FMA contraction: this file is compiled
//    with -ffp-contract=off; `rsq <= cutneighsq`, ref/neighbor.cpp:165,179), so rows equal the
//    reference's as sets; the stencil always covers the full cutoff sphere.
//  * overflow protocol as the reference's (ref/neighbor.cpp:184-208): rows count past maxneighs but store
//    guarded; the host reads the maximum, grows maxneighs to 1.2*max and relaunches.
#include "device_utils.hpp"
#include <algorithm>
#include "mmd_internal.hpp"
#include "tile_lds.hpp"
#include <vector>

#define NB_SMALL 1.0e-6
#ifndef NB_XF
#define NB_XF 4                // x-slices per reference bin (see fine_x_of)
#endif
#define NB_SUB (8 * NB_XF)     // device bins per block of 2x2x2 reference bins
#define NB_MAX_ROWS 64          // rows of blocks (pencils) around a tile that the production build holds: (2 reach_y + 1)(2 reach_z + 1)

// ---------------------------------------------------------------------------------------------------
// Neighbor::setup (ref/neighbor.cpp:318-452)
// ---------------------------------------------------------------------------------------------------
extern "C" int mmd_neighbor_setup(mmd_handle* h, const int nbin[3], mmd_float cutneigh, int halfneigh, int ghost_newton,
                                  int ntypes)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(nbin[0] < 1 || nbin[1] < 1 || nbin[2] < 1 || !(cutneigh > 0)) { mmd_set_error("mmd_neighbor_setup: bad bins/cutoff"); return -1; }
  if(!(h->prd[0] > 0)) { mmd_set_error("mmd_neighbor_setup: box not set"); return -1; }
  h->cutneigh = cutneigh;
  h->cutneighsq = cutneigh * cutneigh;
  h->halfneigh = halfneigh;
  h->ghost_newton = ghost_newton;
  h->ntypes = ntypes;
  auto geometry = [&](const int nb[3], BinGeom& g) {
    for(int d = 0; d < 3; d++) {
      g.prd[d] = h->prd[d];
      g.sublo[d] = h->lo[d]; g.subhi[d] = h->hi[d];
      g.nbin[d] = nb[d];
      g.binsize[d] = h->prd[d] / nb[d];
      g.bininv[d] = 1.0 / g.binsize[d];
      real coord = h->lo[d] - cutneigh - NB_SMALL * h->prd[d];
      int lo = static_cast<int>(coord * g.bininv[d]);
      if(coord < 0.0) lo -= 1;
      coord = h->hi[d] + cutneigh + NB_SMALL * h->prd[d];
      int hi = static_cast<int>(coord * g.bininv[d]);
      lo -= 1; hi += 1;                      // one extra bin each side, as the reference
      g.mbinlo[d] = lo;
      g.mbin[d] = hi - lo + 1;
      g.blkshift[d] = (-g.mbinlo[d]) & 1;          // bin of coordinate 0 gets an even shifted index
      g.nblk[d] = (g.mbin[d] + g.blkshift[d] + 1) >> 1;
      int next = static_cast<int>(cutneigh * g.bininv[d]);
      if(next * g.binsize[d] < cutneigh) next++;   // full coverage (the reference shaves 0.1% here, :405-415)
      g.reach[d] = (next + 1) >> 1;
    }
  };
  // the reference's bins (reported by mmd_neighbor_geometry, compared by the reference-rule half build k_build<3>) ...
  geometry(nbin, h->bg_ref);
  h->bg = h->bg_ref;
  // ... are also the device's, unless they are so fine that a tile would have to look at more rows of blocks than the build kernels hold
  // (NB_MAX_ROWS = 64: a reach of 3 blocks in y and z, bins finer than cutneigh / 6; `-b` is free to ask for that): the device then bins at about
  // cutneigh / 2 — the lists do not depend on the bins
  if((2 * h->bg.reach[1] + 1) * (2 * h->bg.reach[2] + 1) > NB_MAX_ROWS || h->bg.reach[0] > 5) {
    int nb[3];
    for(int d = 0; d < 3; d++) nb[d] = h->bg.reach[d] > 2 ? std::max(1, std::min(nbin[d], (int)(2.0 * (double)h->prd[d] / (double)cutneigh))) : nbin[d];
    geometry(nb, h->bg);
  }
  BinGeom& g = h->bg;
  const long long mb = (long long)NB_SUB * g.nblk[0] * g.nblk[1] * g.nblk[2];
  if(mb > 2000000000LL) { mmd_set_error("mmd_neighbor_setup: too many bins"); return -1; }
  g.mbins = (int)mb;
  h->neigh_ready = true;
  h->neigh_nlocal = 0;
  if(h->host_only) return 0;
  MMD_TRY(h->bin_count.ensure((size_t)g.mbins + 2, false, h->stream));
  h->bin_count_clean = -1;
  MMD_TRY(h->bin_start.ensure((size_t)g.mbins + 2, false, h->stream));
  h->neigh_ready = true;
  h->neigh_nlocal = 0;
  return 0;
}

// coord -> (ix,iy,iz) exactly as Neighbor::coord2bin (ref/neighbor.cpp:274-297), packed 10 bits each (the reference-rule half
// build k_build<3> compares bins of the REFERENCE grid; mbin <= 1023 per dimension is checked by its caller)
__device__ __forceinline__ int ref_bin3(const BinGeom& g, real x, real y, real z)
{
  int ix, iy, iz;
  if(x >= g.prd[0]) ix = (int)((x - g.prd[0]) * g.bininv[0]) + g.nbin[0] - g.mbinlo[0];
  else if(x >= (real)0.0) ix = (int)(x * g.bininv[0]) - g.mbinlo[0];
  else ix = (int)(x * g.bininv[0]) - g.mbinlo[0] - 1;
  if(y >= g.prd[1]) iy = (int)((y - g.prd[1]) * g.bininv[1]) + g.nbin[1] - g.mbinlo[1];
  else if(y >= (real)0.0) iy = (int)(y * g.bininv[1]) - g.mbinlo[1];
  else iy = (int)(y * g.bininv[1]) - g.mbinlo[1] - 1;
  if(z >= g.prd[2]) iz = (int)((z - g.prd[2]) * g.bininv[2]) + g.nbin[2] - g.mbinlo[2];
  else if(z >= (real)0.0) iz = (int)(z * g.bininv[2]) - g.mbinlo[2];
  else iz = (int)(z * g.bininv[2]) - g.mbinlo[2] - 1;
  return (ix & 1023) | ((iy & 1023) << 10) | ((iz & 1023) << 20);
}

// The device orders atoms finer than the reference's bins along x: every reference bin is cut into NB_XF slices, a BLOCK = 2x2x2
// reference bins = NB_SUB = 2*NB_XF x-slices of 2x2 (y,z) bins, numbered slice-major, so that along a row of blocks (a "pencil" of
// 2x2 bins cross-section) `binned` is sorted by x to a quarter of a bin. The production tiles are the 64-atom pieces of a pencil
// (k_pencil_tiles): full wavefronts of atoms in a compact x-range, instead of one 57-atom block per tile.

// fine x index along a pencil: (shifted reference bin) * NB_XF + slice of the bin; monotone in x, the reference bin as coord2bin has it
__device__ __forceinline__ int fine_x_of(const BinGeom& g, real x)
{
  int ix;
  real y;                                          // position in units of bins, inside the branch coord2bin takes
  if(x >= g.prd[0]) { y = (x - g.prd[0]) * g.bininv[0]; ix = (int)y + g.nbin[0] - g.mbinlo[0]; }
  else if(x >= (real)0.0) { y = x * g.bininv[0]; ix = (int)y - g.mbinlo[0]; }
  else { y = x * g.bininv[0]; ix = (int)y - g.mbinlo[0] - 1; }
  real fr = y - (real)(int)y;                      // y >= 0: the fraction of the bin; y < 0: the bin is [trunc(y) - 1, trunc(y))
  if(x < (real)0.0) fr += (real)1.0;
  const int q = min(max((int)(fr * (real)NB_XF), 0), NB_XF - 1);
  ix = min(max(ix, 0), g.mbin[0] - 1) + g.blkshift[0];
  return ix * NB_XF + q;
}

// coord -> (ix,iy,iz) exactly as Neighbor::coord2bin (ref/neighbor.cpp:274-297), then block-major id
__device__ __forceinline__ int bin_of(const BinGeom& g, real x, real y, real z)
{
  int ix, iy, iz;
  if(x >= g.prd[0]) ix = (int)((x - g.prd[0]) * g.bininv[0]) + g.nbin[0] - g.mbinlo[0];
  else if(x >= (real)0.0) ix = (int)(x * g.bininv[0]) - g.mbinlo[0];
  else ix = (int)(x * g.bininv[0]) - g.mbinlo[0] - 1;
  if(y >= g.prd[1]) iy = (int)((y - g.prd[1]) * g.bininv[1]) + g.nbin[1] - g.mbinlo[1];
  else if(y >= (real)0.0) iy = (int)(y * g.bininv[1]) - g.mbinlo[1];
  else iy = (int)(y * g.bininv[1]) - g.mbinlo[1] - 1;
  if(z >= g.prd[2]) iz = (int)((z - g.prd[2]) * g.bininv[2]) + g.nbin[2] - g.mbinlo[2];
  else if(z >= (real)0.0) iz = (int)(z * g.bininv[2]) - g.mbinlo[2];
  else iz = (int)(z * g.bininv[2]) - g.mbinlo[2] - 1;
  ix = min(max(ix, 0), g.mbin[0] - 1);
  iy = min(max(iy, 0), g.mbin[1] - 1);
  iz = min(max(iz, 0), g.mbin[2] - 1);
  ix += g.blkshift[0]; iy += g.blkshift[1]; iz += g.blkshift[2];
  const int blk = ((iz >> 1) * g.nblk[1] + (iy >> 1)) * g.nblk[0] + (ix >> 1);
  const int fx = fine_x_of(g, x);                  // = ix * NB_XF + slice (same clamped, shifted ix)
  return blk * NB_SUB + (fx & (2 * NB_XF - 1)) * 4 + ((iz & 1) << 1 | (iy & 1));
}

// ---------------------------------------------------------------------------------------------------
// Neighbor::binatoms (ref/neighbor.cpp:215-268): histogram, scan, fill, in-bin sort
// ---------------------------------------------------------------------------------------------------
// the histogram atomic also hands every atom its arrival rank inside the bin, so the fill pass needs neither a
// second round of atomics nor a zeroed cursor array (the arrival order is made deterministic by k_bin_sort).
// Atoms arrive (nearly) sorted by bin, so the lanes of a wavefront form runs of equal bins: the first lane of a run
// adds the run length once and the others derive their rank from it — ~7x fewer atomics on a sorted system.
// nghost_dev != nullptr: the ghost count of a one-rank Comm::borders is still on its way to the host; n = owned atoms + the
// capacity the arrays were sized for, the kernel clamps to owned + *nghost_dev (deferred_count, device_utils.hpp)
// pbc: Atom::pbc (ref/atom.cpp:106-122, same tests in the same order as k_pbc) is applied on the way — one-rank re-neighborings that sort
// wrap and bin the owned atoms in one pass.
__global__ __launch_bounds__(256) void k_bin_count(real4* __restrict__ x, int n, BinGeom g, int* __restrict__ atom_bin, int* __restrict__ atom_rank,
                                                   int* __restrict__ bin_count, int nlocal, const int* __restrict__ nghost_dev, int pbc,
                                                   real xprd, real yprd, real zprd, long long* __restrict__ clk, int first)
{
  // (phase clocks of a re-neighboring: the first kernel of a phase stamps the device's constant-rate wall clock into a result word that returns
  //  with the build's flags — no event packets on the stream)
  if(clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *clk = wall_clock64();
  n = deferred_count(n, nlocal, nghost_dev);
  // first > 0: the atoms below it are counted already (the owned atoms, in the bin order Atom::sort has just given them: mmd_bin_atoms)
  const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < n;
  int b = -1 - lane;                                          // lanes past the end never join a run
  if(valid) {
    real4 p = x[i];
    if(pbc) {
      const real4 q = p;
      if(p.x < (real)0.0) p.x += xprd;
      if(p.x >= xprd) p.x -= xprd;
      if(p.y < (real)0.0) p.y += yprd;
      if(p.y >= yprd) p.y -= yprd;
      if(p.z < (real)0.0) p.z += zprd;
      if(p.z >= zprd) p.z -= zprd;
      if(p.x != q.x || p.y != q.y || p.z != q.z) x[i] = p;         // (a few atoms in a thousand cross a face between two re-neighborings: the others are not written back)
    }
    b = bin_of(g, p.x, p.y, p.z);
  }
  const int prev = __shfl_up(b, 1, 64);
  const bool head = lane == 0 || b != prev;
  const unsigned long long heads = __builtin_amdgcn_ballot_w64(head);
  const unsigned long long upto = heads & (~0ull >> (63 - lane));              // run heads at or below my lane
  const int start = 63 - __clzll(upto);                                         // first lane of my run
  const unsigned long long after = lane == 63 ? 0ull : heads >> (lane + 1);     // run heads above my lane
  const int len = (after ? lane + 1 + __ffsll((long long)after) - 1 : 64) - start;   // (meaningful on the head lane)
  int base = 0;
  if(head && valid) base = atomicAdd(&bin_count[b], len);
  base = __shfl(base, start, 64);
  if(valid) { atom_bin[i] = b; atom_rank[i] = base + (lane - start); }
}

__global__ void k_bin_fill(const int* __restrict__ atom_bin, const int* __restrict__ atom_rank, int n, const int* __restrict__ bin_start,
                           int* __restrict__ binned, int* __restrict__ big_flag, int nlocal, const int* __restrict__ nghost_dev,
                           unsigned* __restrict__ pencil_lohi, int npencils, int first)
{
  n = deferred_count(n, nlocal, nghost_dev);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t == 0) *big_flag = 0;                                    // (set by k_bin_sort, the next kernel on the stream)
  // (first / last owned bin of every pencil: collected by k_bin_sort, the next kernel on the stream — see there)
  if(pencil_lohi != nullptr) for(int q = t; q < 2 * npencils; q += gridDim.x * blockDim.x) pencil_lohi[q] = (q & 1) ? 0u : 0xffffffffu;
  const int i = first + t;                                     // (first > 0: the atoms below it are placed by k_bin_sort, see there)
  if(i >= n) return;
  binned[bin_start[atom_bin[i]] + atom_rank[i]] = i;
}

// one thread per bin: insertion sort of its (short) slice -> ascending atom index, run-to-run identical.
// Bins longer than NB_BIGBIN (e.g. `-b 1`: every atom in one bin) are left to k_bin_sort_big.
#define NB_BIGBIN 96
// pencil_lohi != nullptr (the binning of a neighbor build inside a run): the pass also collects, per pencil (row of blocks along x = bins_per_pencil
// consecutive bins), the first and the last bin (+1) that holds an owned atom — what k_pencil_count used to find in a launch of its own. A wavefront's 64
// bins nearly always lie in one pencil: it reduces them and issues ONE atomic pair (1 k atomics on the same two words retire one at a time).
// owned_start != nullptr (the binning of a build right behind Atom::sort, mmd_bin_atoms): the owned atoms are in bin order — bin b owns the atoms
// owned_start[b] .. owned_start[b + 1] - 1 —, were counted by the sort's own binning (keep_counts there: the histogram is left holding them instead of
// zeros) and are written here, in front of the bin's ghosts, which k_bin_count / k_bin_fill have placed behind them and which are sorted as usual.
__global__ void k_bin_sort(const int* __restrict__ bin_start, int mbins, int* __restrict__ binned, int* __restrict__ big_flag,
                           int* __restrict__ bin_count, unsigned* __restrict__ pencil_lohi, int bins_per_pencil, int nlocal,
                           const int* __restrict__ owned_start, int keep_counts)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  bool owned = false;
  if(b <= mbins) bin_count[b] = keep_counts && b < mbins ? bin_start[b + 1] - bin_start[b] : 0;     // the histogram has been scanned: zeroed (or the owned counts) for the next binning
  if(b < mbins) {
    const int s = bin_start[b], e = bin_start[b + 1];
    if(e - s > NB_BIGBIN) *big_flag = 1;                      // (left to k_bin_rank_big; the build is redone with that pass switched on)
    else if(owned_start != nullptr) {
      const int o0 = owned_start[b], c1 = owned_start[b + 1] - o0;
      for(int k = 0; k < c1; k++) binned[s + k] = o0 + k;
      for(int a = s + c1 + 1; a < e; a++) {
        const int key = binned[a];
        int c = a - 1;
        while(c >= s + c1 && binned[c] > key) { binned[c + 1] = binned[c]; c--; }
        binned[c + 1] = key;
      }
      owned = c1 > 0;
    } else {
      int kmin = e > s ? binned[s] : 0x7fffffff;              // (smallest index of the bin, kept in a register: no load behind the sort's stores)
      for(int a = s + 1; a < e; a++) {
        const int key = binned[a];
        kmin = min(kmin, key);
        int c = a - 1;
        while(c >= s && binned[c] > key) { binned[c + 1] = binned[c]; c--; }
        binned[c + 1] = key;
      }
      owned = kmin < nlocal;                                  // (a bin's entries ascend now: owned atoms first)
    }
  }
  if(pencil_lohi == nullptr) return;
  // (a pencil has more than 64 bins: a wavefront's 64 consecutive bins lie in at most two pencils, p0 and p0 + 1 — one reduction and one atomic pair each)
  const int p = b < mbins ? b / bins_per_pencil : -1, q = b < mbins ? b - p * bins_per_pencil : 0;
  const int p0 = __builtin_amdgcn_readfirstlane(p);
  if(p0 < 0) return;
#pragma unroll
  for(int seg = 0; seg < 2; seg++) {
    const int ps = p0 + seg;
    const bool mine = owned && p == ps;
    if(__builtin_amdgcn_ballot_w64(mine) == 0ull) continue;
    const unsigned lo = wave_min_u(mine ? (unsigned)q : 0xffffffffu), hi = wave_max_u(mine ? (unsigned)q + 1u : 0u);
    if((threadIdx.x & 63) == 0) { atomicMin(&pencil_lohi[2 * ps], lo); atomicMax(&pencil_lohi[2 * ps + 1], hi); }
  }
  if(owned && p > p0 + 1) { atomicMin(&pencil_lohi[2 * p], (unsigned)q); atomicMax(&pencil_lohi[2 * p + 1], (unsigned)q + 1u); }     // (pencils shorter than 64 bins: tiny boxes)
}
// long bins: every entry finds its rank by counting the smaller entries of its bin (O(n^2) compares spread over the whole
// grid instead of one thread's insertion sort); `scratch` holds as many ints as `binned`; a second launch copies the
// ranked entries back. Both exit at once when no bin is long.
__global__ __launch_bounds__(256) void k_bin_rank_big(const int* __restrict__ bin_start, int mbins, const int* __restrict__ binned,
                                                      int* __restrict__ scratch, const int* __restrict__ big_flag)
{
  if(*big_flag == 0) return;
  const int stride = gridDim.x * blockDim.x;
  for(int b = 0; b < mbins; b++) {
    const int s = bin_start[b], n = bin_start[b + 1] - s;
    if(n <= NB_BIGBIN) continue;
    for(int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
      const int key = binned[s + t];
      int rank = 0;
      for(int u = 0; u < n; u++) rank += binned[s + u] < key ? 1 : 0;
      scratch[s + rank] = key;
    }
  }
}
__global__ __launch_bounds__(256) void k_bin_copy_big(const int* __restrict__ bin_start, int mbins, int* __restrict__ binned,
                                                      const int* __restrict__ scratch, const int* __restrict__ big_flag)
{
  if(*big_flag == 0) return;
  const int stride = gridDim.x * blockDim.x;
  for(int b = 0; b < mbins; b++) {
    const int s = bin_start[b], n = bin_start[b + 1] - s;
    if(n <= NB_BIGBIN) continue;
    for(int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) binned[s + t] = scratch[s + t];
  }
}

int mmd_bin_atoms(mmd_handle* h, int count)
{
  const int n = count < 0 ? h->nlocal + h->nghost : count;
  const BinGeom& g = h->bg;
  MMD_TRY(h->atom_bin.ensure((size_t)n + 1, false, h->stream));
  MMD_TRY(h->atom_rank.ensure((size_t)n + 1, false, h->stream));
  MMD_TRY(h->binned.ensure((size_t)n + 1, false, h->stream));
  // A re-neighboring of a run bins twice: Atom::sort the owned atoms, the build all atoms. After the sort the owned atoms ARE in bin order and their
  // counts are known, so the second pass counts and places the ghosts only (`reuse`; the first pass leaves the owned counts in the histogram: `keep`)
  // and k_bin_sort writes the owned part of every bin from the first pass's starts — the same `binned`, 18 + 9 us less at -s 80.
  const bool reuse = count < 0 && h->bin_owned_valid && h->bin_owned_n == h->nlocal && h->bin_owned_mbins == g.mbins && !h->big_bins && h->nlocal > 0 && n > h->nlocal;
  const bool keep = count >= 0 && count == h->nlocal && h->in_reneighbor && !h->big_bins && count > 0;
  h->bin_owned_valid = false;
  if(!reuse && h->bin_count_clean != g.mbins)            // (first use / new geometry / owned counts nobody used; afterwards k_bin_sort leaves the histogram zeroed)
    HIP_TRY(hipMemsetAsync(h->bin_count.p, 0, ((size_t)g.mbins + 1) * sizeof(int), h->stream));
  h->bin_count_clean = -1;
  const int first = reuse ? h->nlocal : 0;
  if(reuse) MMD_TRY(h->bin_start_alt.ensure((size_t)g.mbins + 8, false, h->stream));
  int* const starts = reuse ? h->bin_start_alt.p : h->bin_start.p;
  // (dense bins, e.g. `-b 1`: the long-bin rank sort is on from the first binning, not only once a build has seen such a bin)
  if(!h->big_bins && (long long)n > 32LL * g.mbin[0] * g.mbin[1] * g.mbin[2]) h->big_bins = true;     // (this rank's bins, not the global grid)
  if(n) hipLaunchKernelGGL(k_bin_count, dim3(div_up(n - first, 256)), dim3(256), 0, h->stream, h->x.p, n, g, h->atom_bin.p, h->atom_rank.p, h->bin_count.p, h->nlocal, count < 0 ? h->nghost_dev : (const int*)nullptr,
                           h->pbc_pending ? 1 : 0, h->prd[0], h->prd[1], h->prd[2], h->clk_slot >= 0 ? (long long*)(h->d_flags + 56 + 2 * h->clk_slot) : (long long*)nullptr, first);
  if(n && h->clk_slot >= 0) { h->clk_written |= 1 << h->clk_slot; h->clk_slot = -1; }
  h->pbc_pending = false;
  MMD_TRY(mmd_exclusive_scan_from(h, h->bin_count.p, starts, g.mbins, nullptr));
  // (a neighbor build that folds k_pencil_count into this pass asked for it: pencil_lohi_req, cleared here)
  unsigned* lohi = nullptr;
  const int npencils = g.nblk[1] * g.nblk[2];
  if(h->pencil_lohi_req && count < 0 && !h->big_bins) {
    MMD_TRY(h->pencil_lohi.ensure((size_t)2 * npencils + 2, false, h->stream));
    lohi = h->pencil_lohi.p;
  }
  h->pencil_lohi_req = false;
  h->pencil_lohi_ready = lohi != nullptr;
  hipLaunchKernelGGL(k_bin_fill, dim3(div_up(n - first > 0 ? n - first : 1, 256)), dim3(256), 0, h->stream, h->atom_bin.p, h->atom_rank.p, n, (const int*)starts, h->binned.p, h->d_flags + 12, h->nlocal, count < 0 ? h->nghost_dev : (const int*)nullptr,
                     lohi, npencils, first);
  hipLaunchKernelGGL(k_bin_sort, dim3(div_up(g.mbins + 1, 256)), dim3(256), 0, h->stream, (const int*)starts, g.mbins, h->binned.p, h->d_flags + 12, h->bin_count.p,
                     lohi, g.nblk[0] * NB_SUB, h->nlocal, reuse ? (const int*)h->bin_start.p : (const int*)nullptr, keep ? 1 : 0);
  if(reuse) { std::swap(h->bin_start, h->bin_start_alt); h->bin_reuses++; }
  if(keep) { h->bin_owned_valid = true; h->bin_owned_n = count; h->bin_owned_mbins = g.mbins; }
  // bins longer than NB_BIGBIN are ordered by the grid-wide rank count (atom_bin is free again after the fill: its scratch).
  // The two launches are skipped while no such bin has been seen: the neighbor build reads the flag k_bin_sort raises
  // (with its own result flags) and then switches them on and bins again.
  if(h->big_bins) {
    hipLaunchKernelGGL(k_bin_rank_big, dim3(128), dim3(256), 0, h->stream, h->bin_start.p, g.mbins, h->binned.p, h->atom_bin.p, h->d_flags + 12);
    hipLaunchKernelGGL(k_bin_copy_big, dim3(128), dim3(256), 0, h->stream, h->bin_start.p, g.mbins, h->binned.p, h->atom_bin.p, h->d_flags + 12);
  }
  HIP_TRY(hipGetLastError());
  h->bin_count_clean = keep ? -1 : g.mbins;          // (keep: it holds the owned counts — a binning that does not reuse them zeroes it first)
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Neighbor::build (ref/neighbor.cpp:79-213)
// ---------------------------------------------------------------------------------------------------
// MODE 0: full list (every j != i).  MODE 1: half, no ghost newton (keep j > i; ghosts always, ref :171).
// MODE 2: half with ghost newton: every pair stored once globally — owned j: j > i; ghost j (periodic image or another
//         rank's atom alike): (z,y,x) lexicographic order of the positions as ref/neighbor.cpp:155-157.
// MODE 3: half with ghost newton, the REFERENCE's own partition (ref/neighbor.cpp:143-182 with the half stencil of :424-441): a
//         partner in the atom's own bin is kept when j > i (owned) / when it is not below the atom in (z,y,x) order (ghost, :150-157);
//         a partner in another bin is kept when that bin lies in the upper half of the stencil (dk > 0, or dk == 0 and (dj > 0 or
//         (dj == 0 and di > 0))), whoever owns it. Used by mmd_neighbor_download only: the force kernels work on the positional
//         partition of the tile build, a downloaded list is the reference's list (rows as sets).
//
// Work decomposition (one wavefront per 2x2x2-bin block):
//   * the ~27 blocks x ~57 atoms of candidates are loaded ONCE into REGISTERS, transposed: lane l holds
//     candidates l, l+64, l+128, ... (NB_CHUNKS chunks; independent coalesced loads -> deep memory parallelism);
//   * the block's owned atoms are then visited one at a time (wave-uniform: position arrives through the
//     scalar cache), every lane testing its own candidate of each chunk; the hits of a chunk are appended
//     IN CANDIDATE ORDER with ballot + mbcnt prefix (no atomics, deterministic), 64 tests per VALU pass.
//   No LDS traffic in the hot loop, ~2 waves/SIMD (DP) instead of 3 waves/CU for the LDS-broadcast form.
#if MMD_PRECISION == 1
#define NB_CHUNKS 32
#else
#define NB_CHUNKS 28
#endif
#define NB_SLOT_BYTES (3 * (int)sizeof(real))   // nl16 holds slot * NB_SLOT_BYTES (<= 43008 DP / 24576 SP)
#define NB_ROW_PAD 4           // padding granularity of the tile rows (tile_max)
#define NB_FASTR 9             // slices handled by the branch-free slot addressing (3x3 rows of blocks)
#define NB_MAXA 512            // owned atoms of one block handled per pass (counts live in LDS)
#define NB_IDX_MASK 0x1FFFFFFF // candidate word = index | info << 29

template <int MODE>
__global__ __launch_bounds__(64, 2) void k_build(const real4* __restrict__ x, const int* __restrict__ binned,
                                                 const int* __restrict__ bin_start, const int* __restrict__ ghost_image,
                                                 BinGeom g, int nlocal, real cutneighsq, int maxneighs,
                                                 int* __restrict__ neigh, int* __restrict__ numneigh, int* __restrict__ flags, BinGeom gref)
{
  __shared__ int rng_start[128], rng_pref[130];
  __shared__ int cnt[NB_MAXA];
  const int lane = threadIdx.x;
  const int b = xcd_work_item(g.nblk[0] * g.nblk[1] * g.nblk[2]);
  if(b < 0) return;
  const int a0 = bin_start[b * NB_SUB], a1 = bin_start[b * NB_SUB + NB_SUB];
  if(a0 == a1) return;                                      // empty block (uniform exit)
  const int bx = b % g.nblk[0], by = (b / g.nblk[0]) % g.nblk[1], bz = b / (g.nblk[0] * g.nblk[1]);

  // candidate slices: for every (dz,dy) one contiguous run of blocks [bx-R, bx+R] (clamped to the grid);
  // lane r handles row r (<= 128 rows: two passes of the wavefront), lengths prefix-summed by a wave scan
  const int ny = 2 * g.reach[1] + 1, nz = 2 * g.reach[2] + 1;
  const int nr = min(ny * nz, 128);
  int carry = 0;
  for(int r0 = 0; r0 < nr; r0 += 64) {
    const int r = r0 + lane;
    int len = 0, start = 0;
    if(r < nr) {
      const int z = bz + r / ny - g.reach[2], y = by + r % ny - g.reach[1];
      if(z >= 0 && z < g.nblk[2] && y >= 0 && y < g.nblk[1]) {
        const int x0 = max(bx - g.reach[0], 0), x1 = min(bx + g.reach[0], g.nblk[0] - 1);
        const int row = (z * g.nblk[1] + y) * g.nblk[0];
        start = bin_start[(row + x0) * NB_SUB];
        len = bin_start[(row + x1) * NB_SUB + NB_SUB] - start;
      }
    }
    const int incl = wave_incl_scan(len);
    if(r < nr) { rng_start[r] = start; rng_pref[r] = carry + incl - len; }
    carry += __shfl(incl, 63, 64);
  }
  if(lane == 0) rng_pref[nr] = carry;
  __syncthreads();
  const int total = rng_pref[nr];
  if(lane == 0) atomicMax(&flags[1], total);

  for(int ab = a0; ab < a1; ab += NB_MAXA) {                // (one pass unless a block holds > NB_MAXA atoms)
    const int ae = min(ab + NB_MAXA, a1);
    for(int t = lane; t < ae - ab; t += 64) cnt[t] = 0;
    __syncthreads();

    for(int t0 = 0; t0 < total; t0 += NB_CHUNKS * 64) {     // (one pass unless > NB_CHUNKS*64 candidates)
      // ---- transpose-load the candidates of this pass into registers
      real cx[NB_CHUNKS], cy[NB_CHUNKS], cz[NB_CHUNKS];
      unsigned cw[NB_CHUNKS];
      int r = 0;                               // slots grow with c: the slice search never restarts
#pragma unroll
      for(int c = 0; c < NB_CHUNKS; c++) {
        const int gt = t0 + c * 64 + lane;
        cx[c] = (real)1.0e15; cy[c] = (real)1.0e15; cz[c] = (real)1.0e15; cw[c] = NB_IDX_MASK;   // never a hit
        if(gt < total) {
          while(r + 1 < nr && rng_pref[r + 1] <= gt) r++;
          const int j = binned[rng_start[r] + (gt - rng_pref[r])];
          const real4 p = x[j];
          cx[c] = p.x; cy[c] = p.y; cz[c] = p.z;
          unsigned info = 1;
          if(MODE == 1) info = j >= nlocal ? 1 : 3;
          if(MODE == 2 || MODE == 3) {
            info = j < nlocal ? 3 : 2;                       // ghosts: (z,y,x) order of the positions decides (ref/neighbor.cpp:155-157)
          }
          cw[c] = (unsigned)j | (info << 29);
        }
      }
      const int nchunks = min(NB_CHUNKS, (total - t0 + 63) >> 6);

      // ---- owned atoms of the block, one per iteration (wave-uniform)
      for(int a = ab; a < ae; a++) {
        const int i = __builtin_amdgcn_readfirstlane(binned[a]);
        if(i >= nlocal) continue;                            // ghosts get no row
        const real4 xi = x[i];                               // uniform address: scalar load
        const real xix = xi.x, xiy = xi.y, xiz = xi.z;
        const int bin_i = MODE == 3 ? ref_bin3(gref, xix, xiy, xiz) : 0;
        int n = cnt[a - ab];
        const size_t rowbase = ((size_t)(i >> 6) * maxneighs) * 64 + (i & 63);
#pragma unroll
        for(int c = 0; c < NB_CHUNKS; c++) {
          if(c < nchunks) {
            const real dx = xix - cx[c], dy = xiy - cy[c], dz = xiz - cz[c];
            const real rsq = dx * dx + dy * dy + dz * dz;
            const int j = (int)(cw[c] & NB_IDX_MASK);
            bool keep = rsq <= cutneighsq && j != i;
            if(MODE != 0 && MODE != 3) {
              const unsigned info = cw[c] >> 29;
              bool ok = info == 1;
              if(info == 3) ok = j > i;
              if(MODE == 2 && info == 2)
                ok = !(cz[c] < xiz || (cz[c] == xiz && cy[c] < xiy) || (cz[c] == xiz && cy[c] == xiy && cx[c] < xix));
              keep = keep && ok;
            }
            if(MODE == 3 && keep) {                          // (only the few candidates inside the cutoff pay for their bin)
              const int bin_j = ref_bin3(gref, cx[c], cy[c], cz[c]);
              bool ok;
              if(bin_j == bin_i) {
                ok = (cw[c] >> 29) == 3 ? j > i
                                        : !(cz[c] < xiz || (cz[c] == xiz && cy[c] < xiy) || (cz[c] == xiz && cy[c] == xiy && cx[c] < xix));
              } else {
                const int di = (bin_j & 1023) - (bin_i & 1023), dj = ((bin_j >> 10) & 1023) - ((bin_i >> 10) & 1023), dk = (bin_j >> 20) - (bin_i >> 20);
                ok = dk > 0 || (dk == 0 && (dj > 0 || (dj == 0 && di > 0)));
              }
              keep = ok;
            }
            const unsigned long long m = __ballot(keep);
            if(m) {
              const int pos = n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
              if(keep && pos < maxneighs) neigh[rowbase + (size_t)pos * 64] = j;
              n += __popcll(m);
            }
          }
        }
        if(lane == 0) cnt[a - ab] = n;
      }
      __syncthreads();
    }
    // ---- row lengths
    int wmax = 0;
    for(int t = lane; t < ae - ab; t += 64) {
      const int i = binned[ab + t];
      if(i < nlocal) { numneigh[i] = cnt[t]; wmax = max(wmax, cnt[t]); }
    }
    wmax = wave_max_i(wmax);
    if(lane == 0) atomicMax(&flags[0], wmax);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// Tiles ("pencil tiles", k_build_rows): a pencil = one row of blocks along x (2x2 reference bins in cross-section); its
// entries of `binned` are sorted by x to a quarter of a bin (NB_XF). The stretch from the pencil's first to its last bin that holds an
// owned atom is cut into pieces of 64 entries: every tile but the last of a pencil is a FULL wavefront of atoms within ~1.2 block
// lengths of x (at LJ liquid density a block holds 57 atoms: one-block tiles leave 11 % of the lanes empty).
// One wavefront per pencil; ntile_of_pencil is scanned by the caller, tile_block[] holds the pencil's first block.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pencil_count(const int* __restrict__ binned, const int* __restrict__ bin_start, int npencils, int nblk0,
                                                     int nlocal, int* __restrict__ ntile_of_pencil, int* __restrict__ pencil_range)
{
  const int p = blockIdx.x, lane = threadIdx.x;
  if(p >= npencils) return;
  const int b0 = p * nblk0 * NB_SUB, nb = nblk0 * NB_SUB;
  unsigned lo = 0xffffffffu, hi = 0u;                      // first / last bin (+1) of the pencil that holds an owned atom
  for(int q = lane; q < nb; q += 64) {
    const int s0 = bin_start[b0 + q], s1 = bin_start[b0 + q + 1];
    if(s1 > s0 && binned[s0] < nlocal) { lo = min(lo, (unsigned)q); hi = max(hi, (unsigned)q + 1u); }     // (a bin's entries ascend: owned atoms first)
  }
  lo = wave_min_u(lo); hi = wave_max_u(hi);
  if(lane == 0) {
    int a0 = 0, a1 = 0;
    if(hi > 0u) { a0 = bin_start[b0 + (int)lo]; a1 = bin_start[b0 + (int)hi]; }
    ntile_of_pencil[p] = (a1 - a0 + 63) >> 6;
    pencil_range[2 * p] = a0; pencil_range[2 * p + 1] = a1;
  }
}
__global__ void k_pencil_fill(const int* __restrict__ pencil_range, int npencils, int nblk0, const int* __restrict__ tile_of_pencil,
                              int* __restrict__ tile_block, int* __restrict__ tile_first, int* __restrict__ tile_cnt, int* __restrict__ flags, int cap,
                              real4* __restrict__ x, int nlocal, int ghost_cap, const int* __restrict__ nghost_dev)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if(p == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0; flags[4] = 0; flags[5] = 0; flags[7] = 0; }     // (result flags of the build kernel and accumulators of k_tile_reduce, which follow on the stream)
  if(p == 0 && nghost_dev) x[nlocal + min(*nghost_dev, ghost_cap)] = real4{(real)1.0e15, (real)1.0e15, (real)1.0e15, (real)0};    // the dummy atom (far outside any cutoff, see k_set_dummy) behind the last ghost, whose number only the device knows yet
  if(p >= npencils) return;
  const int t0 = tile_of_pencil[p], t1 = tile_of_pencil[p + 1];
  const int a0 = pencil_range[2 * p], a1 = pencil_range[2 * p + 1];
  for(int t = t0; t < t1 && t < cap; t++) {
    tile_block[t] = p * nblk0;
    tile_first[t] = a0 + (t - t0) * 64;
    tile_cnt[t] = min(64, a1 - tile_first[t]);
  }
}

// the same with the scan of the per-pencil tile counts folded in (re-neighborings inside a run, where nobody needs the count on the host before the
// build has run): every workgroup sums the counts in front of it itself, the last one stores the total behind the counts (ntiles_dev of the build)
// pencil_lohi != nullptr: no k_pencil_count ran — every pencil's entry range and tile count come from the first / last owned bin k_bin_sort collected
__global__ __launch_bounds__(256) void k_pencil_fill_scan(const int* __restrict__ pencil_range, int npencils, int nblk0, int* __restrict__ ntile_of_pencil,
                                                          int* __restrict__ tile_block, int* __restrict__ tile_first, int* __restrict__ tile_cnt, int* __restrict__ flags,
                                                          int cap, real4* __restrict__ x, int nlocal, int ghost_cap, const int* __restrict__ nghost_dev,
                                                          const unsigned* __restrict__ pencil_lohi, const int* __restrict__ bin_start)
{
  __shared__ int lds[17];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if(p == 0) { flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0; flags[4] = 0; flags[5] = 0; flags[7] = 0; }
  if(p == 0 && nghost_dev) x[nlocal + min(*nghost_dev, ghost_cap)] = real4{(real)1.0e15, (real)1.0e15, (real)1.0e15, (real)0};    // the dummy atom (far outside any cutoff, see k_set_dummy) behind the last ghost, whose number only the device knows yet
  auto range_of = [&](int pp, int& a0, int& a1) {
    a0 = 0; a1 = 0;
    if(pencil_lohi != nullptr) {
      const unsigned lo = pencil_lohi[2 * pp], hi = pencil_lohi[2 * pp + 1];
      const int b0 = pp * nblk0 * NB_SUB;
      if(hi > 0u) { a0 = bin_start[b0 + (int)lo]; a1 = bin_start[b0 + (int)hi]; }
    } else { a0 = pencil_range[2 * pp]; a1 = pencil_range[2 * pp + 1]; }
  };
  int before;
  if(pencil_lohi != nullptr) {
    int v = 0;
    const int nprev = (int)blockIdx.x * 256;
    for(int t0 = threadIdx.x; t0 < nprev; t0 += 4 * 256) {      // (four pencils per trip: their two dependent loads each overlap)
      int a0[4], a1[4];
#pragma unroll
      for(int u = 0; u < 4; u++) { a0[u] = 0; a1[u] = 0; if(t0 + u * 256 < nprev) range_of(t0 + u * 256, a0[u], a1[u]); }
#pragma unroll
      for(int u = 0; u < 4; u++) v += (a1[u] - a0[u] + 63) >> 6;
    }
    int tot0;
    block_incl_scan(v, lds, &tot0);
    before = tot0;
  } else before = block_prefix_total(ntile_of_pencil, blockIdx.x * 256, lds);
  int ma0 = 0, ma1 = 0;
  if(p < npencils) range_of(p, ma0, ma1);
  const int mine = p < npencils ? (pencil_lohi != nullptr ? (ma1 - ma0 + 63) >> 6 : ntile_of_pencil[p]) : 0;
  int tot;
  const int inc = block_incl_scan(mine, lds, &tot);
  const int t0 = before + inc - mine, t1 = t0 + mine;
  if(p < npencils) {
    const int a0 = ma0, a1 = ma1;
    for(int t = t0; t < t1 && t < cap; t++) {
      tile_block[t] = p * nblk0;
      tile_first[t] = a0 + (t - t0) * 64;
      tile_cnt[t] = min(64, a1 - tile_first[t]);
    }
  }
  // (every workgroup has read the counts in front of it before the last one finishes? No: the total goes to slot [npencils], which nobody sums)
  if(blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) ntile_of_pencil[npencils] = before + tot;
}

// ---------------------------------------------------------------------------------------------------
// Tile build, one OWNED ATOM PER LANE (production): a single wavefront builds the rows of one tile.
//   phase 1 (cull)  the candidates of the surrounding blocks are streamed through registers in batches (coalesced,
//                   transposed: lane l looks at candidate c*64+l); those within the cutoff of the tile's bounding box
//                   survive and are appended, in candidate order (ballot/mbcnt), to a small LDS buffer of positions;
//   phase 2 (test)  whenever the buffer fills, every lane tests ITS atom against each buffered candidate (position
//                   broadcast from LDS, 8 candidates per 6 ds_read_b128): the v_cmp mask of a test is shifted into a
//                   per-lane bit word with ONE v_addc_co_u32 — no ballot/mbcnt/append per test;
//   expansion       after 32 candidates the OR of the lanes' words tells which of them belong to the tile's candidate UNION
//                   (referenced by at least one row): those get the next slots and go to tile_cand[]; the set bits of a lane
//                   become row entries (16-bit LDS offsets of the slots), written straight to the lane's column of nl16.
// Nothing is shared between wavefronts: no barriers, ~8 KB of LDS, < 100 VGPRs.
// Double precision (PF): the buffered positions are FLOATS relative to the tile's corner and phase 2 is a conservative
// float pre-test with two thresholds (cutneighsq -/+ eps, eps = 4x the worst-case float error of rsq): below the lower
// one a pair is a hit, above the upper one it is not; the ~0.1 pairs per tile in between are re-tested exactly
// (double, unfused, `rsq <= cutneighsq` as ref/neighbor.cpp:165,179) from the global positions — rows are bit-identical
// to an all-double build at half the VALU cycles and a third of the LDS. Single precision tests exactly in float.
// MODE as in k_build. Half modes store a pair on the atom BELOW it ((z,y,x) order of the exact positions, the rule the
// reference applies to ghosts, ref/neighbor.cpp:155-157; without ghost newton a ghost partner is kept by whoever sees it):
// balanced rows, and candidates below every tile atom are dropped in the cull, so the union is the upper half shell only.
// mmd_neighbor_download re-homes the owned pairs to the reference's `j > i` partition (k_rows_to_ref_half).
// ---------------------------------------------------------------------------------------------------
#ifndef NB2_BATCH
#define NB2_BATCH 4            // chunks of 64 candidates in flight per batch of loads
#endif
#ifndef NB2_BUF
#define NB2_BUF 448            // LDS candidate buffer (slots); flushed between batches when fewer than 64*NB2_BATCH are free
#endif
#define NB2_NE (4 * NB2_BUF / 64)   // list entries (non-empty hit words) per lane that fit the recycled candidate buffer
#define NB2_NG 40              // groups of 32 tested candidates a tile may produce (more: the global-row build takes over)
static_assert(NB2_NG <= 64, "a lane keeps its parked groups in a 64-bit mask");
#define NB2_PF (MMD_PRECISION == 2)
// Double precision, round 4: the float pre-test is evaluated as |a|^2 + |b|^2 - 2 a.b on coordinates relative to the tile's CENTRE: the buffer
// holds -2b and |b|^2 of every candidate, a lane keeps its atom a and thr = cutneighsq - |a|^2, so d = rsq - cutneighsq costs three packed
// fma + one packed subtract per TWO candidates (the difference form: three subtracts + three fma), the hit bit is the SIGN of d shifted into
// the lane's word by one v_alignbit_b32 (no v_cmp / v_addc pair), and a running min3 of |d| says at the end of a group of 32 whether any pair
// of the lane came closer to the threshold than the error bound of this arithmetic — only then (about one lane-group in five tiles at LJ
// density) are the candidates of the group looked at one by one and the pairs inside the band re-tested exactly in double. 5 instead of
// 8.3 VALU instructions per candidate; rows are still those of an all-double build. -DNB2_DOT=0: the difference form with two thresholds.
#ifndef NB2_DOT
#define NB2_DOT 1
#endif
#ifndef NB2_DOT_SP
#define NB2_DOT_SP 1                           // single precision: the same pre-test; the exact re-test then is the reference's own float expression
#endif
// Half lists: the z-order rule ("partner above me") travels the same way — zz = 2 (z_a - z_b) from the buffered -2 z_b, two sign words for
// zz < -ztol ("surely above") and zz < +ztol ("possibly above"); a pair that is a distance hit and lies between the two is decided exactly.
// (A running min of |zz| as the ambiguity measure does NOT work: every tile atom is a candidate of its own tile and has zz = 0 exactly against
// itself, which sent two or three groups per tile through the one-by-one walk: k_build_rows<2,0> 527 -> 820 us at -s 80. The two words name the
// candidates, so the atom itself is masked out with its buffer position like in the full-list kernel.)
// Built, parity-green, and NOT faster than the difference form for half lists (same-box A/B at -s 80 / EAM -s 64 / -s 160 SP: 5293 vs 5290,
// 2326 vs 2304, 7677 vs 7697 Matom-steps/s): half-list tiles test only the upper half shell (~310 candidates), their build is dominated by the cull
// and the expansion. Off by default (tools/build_variant.sh <name> neighbor -DNB2_DOT_HALF=1).
#ifndef NB2_DOT_HALF
#define NB2_DOT_HALF 0
#endif
#define NB2_DOTF(MODE) (NB2_DOT && ((MODE) == 0 || NB2_DOT_HALF) && (NB2_PF || NB2_DOT_SP))

// bits = (bits << 1) | (my bit of m): one VALU instruction (carry-in = the compare mask)
__device__ __forceinline__ unsigned nb2_shift_in(unsigned bits, unsigned long long m)
{
  unsigned long long carry_out;
  asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(carry_out) : "s"(m));
  return bits;
}

typedef float nb2_f2 __attribute__((ext_vector_type(2)));
// bits = (bits << 1) | sign(d): one v_alignbit_b32
__device__ __forceinline__ unsigned nb2_shift_sign(unsigned bits, float d) { return __builtin_amdgcn_alignbit(bits, __float_as_uint(d), 31); }
// min(acc, |a|, |b|): one v_min3_f32 with source modifiers
__device__ __forceinline__ float nb2_min3_abs(float acc, float a, float b)
{
  asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}

// Round 5: the pre-test on the MATRIX cores (full lists). |a|^2 + |b|^2 - 2 a.b - cutneighsq of 32 candidates x 32 tile atoms is ONE
// v_mfma_f32_32x32x16_f16: every fp32 quantity of the packed-VALU form travels as a pair of halves (hi = RN(v), lo = RN(v - hi): 22 bits; a product of
// two halves is exact in the fp32 accumulator) and the 16 k-slots hold
//   k 0..7  (lanes 0..31)  A = m_hi.x m_hi.y m_hi.z bb_hi m_lo.x m_lo.y m_lo.z bb_lo     B = a_hi.x a_hi.y a_hi.z 1 a_hi.x a_hi.y a_hi.z 1
//   k 8..15 (lanes 32..63) A = m_hi.x m_hi.y m_hi.z 1     1      .      .      .         B = a_lo.x a_lo.y a_lo.z -thr_hi -thr_lo 0 0 0
// (m = -2 b, bb = |b|^2 of the candidate, a / thr = cutneighsq - |a|^2 of the tile atom; a_lo . m_lo is dropped into the error bound). The candidate
// buffer holds ONE 16-byte record {m_hi.xy | m_hi.z bb_hi | m_lo.xy | m_lo.z bb_lo} per candidate = the k 0..7 operand as it stands; the upper half of
// the wavefront reads the same record and patches two halves to 1.0 (two v_bfi_b32). Two MFMAs (tile atoms 0..31, 32..63) test 32 candidates against
// the whole tile; a lane finds in its 16 + 16 accumulators the values of atom (lane % 32) / (32 + lane % 32) for half of the candidates — candidate row
// 8 (i / 4) + 4 (lane / 32) + i % 4 in register i (layout checked by tools/probes/mfma_f16_probe.hip) —, shifts their SIGNS into two 16-bit words
// (v_alignbit_b32) and one v_permlane32_swap hands every lane the two words of ITS atom: bit 31 - i = the lower-half row of register i, bit 15 - i the
// upper-half one (mf_bit). 1 LDS read + 2 MFMA + 54 VALU per group of 32 candidates instead of 32 LDS reads + 112 VALU.
// Error band, per ATOM: the hardware's accumulation order is unknown — taken as 12 additions (13 non-zero terms) that each lose up to 2^-23 of the
// sum of the |terms| (measured on gfx950: <= 5.3 x 2^-24 of it over 16 terms, f16 denormals honoured: mfma_f16_probe) = 24 units of 2^-24 — plus, per
// quantity, the splits (4 units each for bb, thr, m, a), the dropped a_lo . m_lo (4), the fp32 |b|^2 chain (3), thr through float (1) and the float
// rounding of the local coordinates (<= 4 B_i): <= 36 x 2^-24 x T_i, T_i = sum of the |terms| for a candidate within the cutoff (+ margin) of atom i:
// B_i + |thr_i| + P_i, s_c = |a_c| + cut', B_i = sum s_c^2, P_i = 2 sum s_c |a_c|. E_i = 40 x 2^-24 x (T_i + cutneighsq). A pair whose exact d = rsq - cutneighsq exceeds the TILE-wide bound (the same with La, Lb) has the right sign whatever its magnitudes; one
// below it lies within cut' of atom i, so E_i holds for it: the sign of an accumulator is trusted iff |value| >= E_i. A group in which some lane sees a
// smaller |value| is decided again for ALL its pairs from the global positions, exactly (`rsq <= cutneighsq`, ref/neighbor.cpp:165,179) — 0.3 groups
// per tile at LJ density. Tiles whose extent would take |b|^2 near the f16 range leave the build to the row kernel (flags[3]).
#ifndef NB2_MFMA
#define NB2_MFMA 1
#endif
// Round 5: a lane's hit words wait for the expansion in REGISTERS (one per group, written through a switch on the wave-uniform group number) instead of a
// global scratch list — 125 MB per build at -s 80 that were written and read back, and a memory round trip at the start of every tile's expansion. The
// non-empty ones are compacted into the LDS list the expansion walks when the last group is done. Not for the two-list core/rest rows (56 registers).
#ifndef NB2_REGPARK
#define NB2_REGPARK 1
#endif
#ifndef NB2_MFMA_CORE
#define NB2_MFMA_CORE 1
#endif
typedef _Float16 nb2_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 nb2_h8 __attribute__((ext_vector_type(8)));
typedef float nb2_f16v __attribute__((ext_vector_type(16)));
// bit of the lane's hit word that candidate c (0..31) of a group lands in
__device__ __forceinline__ int nb2_mf_bit(int c) { return ((c & 4) ? 15 : 31) - (((c >> 3) << 2) | (c & 3)); }
__device__ __forceinline__ unsigned nb2_pack_h2(float lo, float hi)      // {RN f16(lo), RN f16(hi)}: one v_cvt_pk_f16_f32
{
  return __builtin_bit_cast(unsigned, __builtin_convertvector(nb2_f2{lo, hi}, nb2_h2));
}
__device__ __forceinline__ nb2_f2 nb2_unpack_h2(unsigned v) { return __builtin_convertvector(__builtin_bit_cast(nb2_h2, v), nb2_f2); }
__device__ __forceinline__ double nb2_readlane(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float nb2_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <int MODE, int CORE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_build_rows(const real4* __restrict__ x, const int* __restrict__ binned,
                                                   const int* __restrict__ bin_start, const int* __restrict__ ghost_image,
                                                   BinGeom g, int ntiles, int nlocal, int nall, real cutneigh, real cutneighsq, int maxneighs, int cstride,
                                                   const int* __restrict__ tile_block, const int* __restrict__ tile_first,
                                                   const int* __restrict__ tile_cnt, int* __restrict__ numneigh,
                                                   unsigned short* __restrict__ nl16, int* __restrict__ tile_cand,
                                                   int* __restrict__ tile_ncand, int* __restrict__ tile_max, int* __restrict__ tile_ghost,
                                                   unsigned short* __restrict__ tile_self, int* __restrict__ tile_rowmax,
                                                   int* __restrict__ tile_rowsum, unsigned* __restrict__ tile_words, int* __restrict__ flags, int ablate_arg,
                                                   const int* __restrict__ ntiles_dev, const int* __restrict__ nghost_dev,
                                                   float core_thr, real4* __restrict__ xbuild, int* __restrict__ tile_kcore,
                                                   int* __restrict__ cand_src, const int* __restrict__ ghost_root, int cand_src_all)
{
  const int ablate = MMD_ABLATE(ablate_arg);     // profiling switches: compiled out of the shipped library (mmd_internal.hpp)
  nall = deferred_count(nall, nlocal, nghost_dev);
  const bool cand_src_wanted = cand_src != nullptr;
  // chunk table of phase 1 (round 6): the candidate slices cut into 64-entry pieces, {first index into binned[], entries} of up to 64 pieces at a time
  __shared__ __align__(16) int s_cstart[64], s_clen[64];
  static_assert(NB_MAX_ROWS <= 64, "one lane per slice");
  // candidate buffer: x | y | z (floats; PF: relative to the tile's corner) | atom index as bit pattern. Once the last
  // buffer has been tested the same 7 KB hold the lanes' hit-word lists for the lock-step expansion (s_ew).
  constexpr bool DOTK = NB2_DOTF(MODE);
  constexpr bool MFK = NB2_MFMA && DOTK && MODE == 0 && (CORE == 0 || NB2_MFMA_CORE);      // the pre-test on the matrix cores (see above); CORE: two more MFMAs against the core radius (a classification: no band)
  constexpr int NB2_NARR = DOTK ? 5 : 4;             // arrays of the candidate buffer: DOT -2x | -2y | -2z | |b|^2 | index, otherwise x | y | z | index
  constexpr int NB2_IDX = (NB2_NARR - 1) * NB2_BUF;  // the atom index (bit pattern) of a buffered candidate
  __shared__ __align__(16) float s_buf[NB2_NARR * NB2_BUF];
  // (the group number of a lane's list entries is not stored: bit g of the lane's gmask says "I parked a word for group g", entries are in group order)
  __shared__ unsigned char s_own[MODE != 0 ? NB2_BUF : 8];      // half lists: which tile atom the candidate is (0xff: none)
  __shared__ unsigned short s_selfpos[64];            // buffer position of each tile atom's own candidate record (0xffff: not in this buffer)
  __shared__ uint2 s_gSU[NB2_NG];                     // per group of 32 buffered candidates: {first slot of its union members, which of the 32 are in the union}
  __shared__ unsigned short s_self[MODE != 0 ? 64 : 4];         // half lists: final slot of each tile atom itself (0xffff: not in the union)
  const int lane = threadIdx.x;
  // the tile count may still be on its way to the host (ntiles = capacity of the arrays, *ntiles_dev = the count)
  const int tile = xcd_work_item(ntiles_dev ? min(ntiles, *ntiles_dev) : ntiles);
  if(tile < 0) return;
  const int b = tile_block[tile];                  // first block of the tile's pencil (pencil tiles) / the tile's block
  const int ta = tile_first[tile], tcn = tile_cnt[tile];
  const int by = (b / g.nblk[0]) % g.nblk[1], bz = b / (g.nblk[0] * g.nblk[1]);
  // ---- my atom
  const int ii = lane < tcn ? binned[ta + lane] : -1;
  const bool owned = ii >= 0 && ii < nlocal;
  const real4 pme = x[ii >= 0 ? ii : 0];
  const unsigned long long own_mask = __builtin_amdgcn_ballot_w64(owned);
  // ---- candidate slices: for every (dz,dy) the stretch of that pencil whose x-slices can hold an atom within the cutoff of the
  // tile's owned atoms — ONE contiguous run of binned[] (a pencil is sorted by x-slice): [slice(xmin - cutneigh), slice(xmax + cutneigh)]
  const int ny = 2 * g.reach[1] + 1, nz = 2 * g.reach[2] + 1;
  const int nr = min(ny * nz, NB_MAX_ROWS);          // (mmd_neighbor_setup keeps ny * nz <= NB_MAX_ROWS)
  // x-range of the tile's owned atoms (float keys, DPP ladder: the same reduction the bounding box below uses; the float rounding of
  // a coordinate is far inside the margin of reach_x)
  const unsigned kx = float_key((float)pme.x);
  const float bx0 = key_float(wave_min_u(owned ? kx : 0xffffffffu)), bx1 = key_float(wave_max_u(owned ? kx : 0u));
  int sl_start = 0, sl_len = 0;              // lane r: slice r of the candidate pencils (lanes >= nr: empty)
  {
    const real xlo = (real)bx0, xhi = (real)bx1;
    const real reach_x = cutneigh * (real)1.0005 + (real)1.0e-4 * g.binsize[0] + (real)1.0e-5 * (fabs((real)bx0) + fabs((real)bx1));
    const int fmaxx = 2 * NB_XF * g.nblk[0] - 1;
    const int f0 = min(max(fine_x_of(g, xlo - reach_x), 0), fmaxx), f1 = min(max(fine_x_of(g, xhi + reach_x), 0), fmaxx);
    {
      const int r = lane;
      if(r < nr) {
        const int z = bz + r / ny - g.reach[2], y = by + r % ny - g.reach[1];
        if(z >= 0 && z < g.nblk[2] && y >= 0 && y < g.nblk[1] && own_mask != 0ull) {
          const int row = (z * g.nblk[1] + y) * g.nblk[0] * NB_SUB;          // first bin of that pencil; slice f starts at bin row + 4 f
          sl_start = bin_start[row + 4 * f0];
          sl_len = bin_start[row + 4 * f1 + 4] - sl_start;
        }
      }
    }
  }
  if(MODE != 0) s_self[lane] = (unsigned short)0xffff;
  s_selfpos[lane] = (unsigned short)0xffff;
  __syncthreads();
  unsigned short* __restrict__ rowp = nl16 + ((size_t)tile * maxneighs) * 64 + lane;
  const size_t cbase = (size_t)tile * cstride;
  if(own_mask == 0ull || (ablate & 32)) {                        // (second tile of a block that holds only ghosts)
    if(lane == 0) { tile_max[tile] = 0; tile_ncand[tile] = 0; tile_cand[cbase] = nall; if(cand_src != nullptr) cand_src[cbase] = nall; tile_ghost[tile] = 0; tile_rowmax[tile] = 0; tile_rowsum[tile] = 0; }
    if(MODE != 0) tile_self[(size_t)tile * 64 + lane] = (unsigned short)0xffff;
    return;
  }
  // bounding box of the tile's owned atoms (float, conservative through the margin of `cull`)
  const unsigned ky = float_key((float)pme.y), kz = float_key((float)pme.z);
  const float by0 = key_float(wave_min_u(owned ? ky : 0xffffffffu)), by1 = key_float(wave_max_u(owned ? ky : 0u));
  const float bz0 = key_float(wave_min_u(owned ? kz : 0xffffffffu)), bz1 = key_float(wave_max_u(owned ? kz : 0u));
  const float cull = (float)cutneighsq * 1.001f + 1.0e-4f;
  // the second candidate list (ghosts named by owner + image code) is written only for tiles that can have a ghost among their candidates: those
  // whose atoms come within the cutoff of a face of the (one-rank) box — a superset of the tiles tile_ghost will flag; the tile kernels read it for those only
  if(cand_src != nullptr && !cand_src_all) {             // (EAM asks for every tile: its sweeps then need no per-tile choice, which costs them registers they do not have)
    const float m = 1.001f * (float)cutneigh + 1.0e-3f * (float)g.prd[0] * 1.0e-3f + 1.0e-4f;
    // (faces of this rank's sub-box: the whole box on one rank)
    const bool near_face = bx0 - m < (float)g.sublo[0] || bx1 + m > (float)g.subhi[0] || by0 - m < (float)g.sublo[1] || by1 + m > (float)g.subhi[1] ||
                           bz0 - m < (float)g.sublo[2] || bz1 + m > (float)g.subhi[2];
    if(!near_face) cand_src = nullptr;
  }
  // PF: local origin = the box's lower corner (exact in `real`); |local coordinate| of my atom and of every candidate that
  // survives the cull is <= Lmax, which bounds the float error of the pre-test's rsq
  // (DOT: the box's centre — halves the magnitudes that enter the products)
  const real ox = DOTK ? (real)(0.5f * (bx0 + bx1)) : (NB2_PF ? (real)bx0 : (real)0), oy = DOTK ? (real)(0.5f * (by0 + by1)) : (NB2_PF ? (real)by0 : (real)0),
             oz = DOTK ? (real)(0.5f * (bz0 + bz1)) : (NB2_PF ? (real)bz0 : (real)0);
  const float Lmax = fmaxf(fmaxf(bx1 - bx0, by1 - by0), bz1 - bz0) + 1.01f * (float)cutneigh + 0.01f;
  // DOT error model: |a_c| <= La, |b_c| <= Lb per coordinate (a = tile atom, b = a candidate that survived the cull). Worst-case absolute error
  // of d against the exact rsq - cutneighsq of the double positions, in units of 2^-24: rounding of the local coordinates to float
  // 6 cut (La + Lb), |b|^2 by three operations of magnitude <= 3 Lb^2 (9 Lb^2), the three fma of the chain with partial sums
  // <= 3 Lb^2 + 6 La Lb, thr = float(cutneighsq - |a|^2) (|thr| <= cutneighsq + 3 La^2). The band is twice that.
  const float La = 0.5f * fmaxf(fmaxf(bx1 - bx0, by1 - by0), bz1 - bz0) + 1.0e-3f, Lb = La + 1.01f * (float)cutneigh + 0.01f;
  const float eps_dot = 2.0f * 5.96046448e-08f /* 2^-24 */ *
                        (6.0f * (float)cutneigh * (La + Lb) + 18.0f * Lb * Lb + 18.0f * La * Lb + (float)cutneighsq + 3.0f * La * La +
                         // single precision: what has to be reproduced is the reference's FLOAT rsq (three products, two sums: <= 4 roundings of
                         // magnitude cutneigh^2 away from the exact rsq of the float positions) — the band covers that too
                         (NB2_PF ? 0.0f : 8.0f * (float)cutneighsq));
  const float eps = 4.76837158e-07f /* 2^-21 */ * (3.0f * (float)cutneigh * Lmax + (float)cutneighsq);
  const float cut_lo = (float)cutneighsq - eps, cut_hi = (float)cutneighsq + eps;
  // half lists: a pair is kept by the atom BELOW it — partner above in (z,y,x) order on the exact positions, the rule of
  // ref/neighbor.cpp:155-157 for ghosts, applied here to every partner: each atom keeps the neighbors of its upper half
  // sphere, so the rows of a tile have about the same length wherever the atom sits in its block (with `j > i` on the
  // block-ordered indices the rows of a tile range from 0 to all neighbors: 52 % lane efficiency in the force kernels), and
  // the tile's union is still only the upper half shell. PF: the pre-test decides z with a tolerance of 4x the float error;
  // the pairs in the band are re-tested exactly together with the borderline distances.
  const float ztol = 4.76837158e-07f * Lmax;
  const float zcull = bz0 - fabsf(bz0) * 2.4e-7f - 1.0e-30f;      // a candidate below every tile atom is nobody's upper partner
  // lanes without an owned atom sit far away on the other side of the padding candidates: never a hit
  const float ztol2 = 4.0f * 5.96046448e-08f * 2.0f * (La + Lb);      // band of zz = 2 (z_a - z_b) in float (half lists): 4x its worst-case error
  // (DOT: such lanes sit at the centre with a threshold of -1e30: d = |b|^2 + 1e30 > 0 for every candidate)
  const float far_i = DOTK ? 0.0f : -1.0e15f;
  const float fxi = owned ? (float)(pme.x - ox) : far_i, fyi = owned ? (float)(pme.y - oy) : far_i, fzi = owned ? (float)(pme.z - oz) : far_i;
  const float zlo = fzi + ztol, zhi = fzi - ztol;
  const double aa_d = (double)fxi * (double)fxi + (double)fyi * (double)fyi + (double)fzi * (double)fzi;
  const float thr_i = owned ? (float)((double)cutneighsq - aa_d) : -1.0e30f;              // d = (|b|^2 - 2 a.b) - thr_i = rsq - cutneighsq
  const float thrc_i = owned ? (float)((double)core_thr - aa_d) : -1.0e30f;               // CORE: the same against the core radius
  const float twofz = 2.0f * fzi;
  // MFK: the atoms' operands of the two MFMAs, the per-atom error bounds that go with them, the masks that patch the upper half's candidate operand
  nb2_h8 mfB1 = {}, mfB2 = {}, mfB1c = {}, mfB2c = {};
  float mfEA = 0.0f, mfEB = 0.0f;
  const unsigned mf_my = lane < 32 ? 0xffffffffu : 0x0000ffffu, mf_mz = lane < 32 ? 0xffffffffu : 0xffff0000u;
  bool mf_range_bad = false;
  if constexpr(MFK) {
    mf_range_bad = 3.0f * Lb * Lb > 30000.0f;                   // |b|^2 must stay a finite half
    {
      // soundness of the per-atom band (round-5 advisor): a pair whose exact d exceeds the TILE-wide bound has the right sign whatever the magnitudes; one below it must lie
      // within cut' of its atom for E_i to cover it — i.e. the tile-wide bound itself has to stay below cut'^2 - cutneighsq. Long tiles of near-empty pencils
      // (La beyond ~55 at cutneigh 2.8) do not: the row kernel builds those
      const float cutp_t = 1.001f * (float)cutneigh + 0.01f, st = La + cutp_t;
      const float T_tile = 3.0f * st * st + ((float)cutneighsq + 3.0f * La * La) + 6.0f * st * La + (float)cutneighsq;
      mf_range_bad = mf_range_bad || 40.0f * 5.96046448e-08f * T_tile > cutp_t * cutp_t - (float)cutneighsq;
    }
    const unsigned h0 = nb2_pack_h2(fxi, fyi), h1 = nb2_pack_h2(fzi, 1.0f);
    const nb2_f2 b0 = nb2_unpack_h2(h0), b1 = nb2_unpack_h2(h1);
    const float thf = owned ? (float)((double)cutneighsq - aa_d) : -60000.0f;
    const float th = (float)(_Float16)thf;
    const unsigned l0 = nb2_pack_h2(fxi - b0.x, fyi - b0.y), l1 = nb2_pack_h2(fzi - b1.x, -th), l2 = nb2_pack_h2(-(thf - th), 0.0f);
    const unsigned hi4[4] = {h0, h1, h0, h1}, lo4[4] = {l0, l1, l2, 0u};
    unsigned o1[4], o2[4];
#pragma unroll
    for(int j = 0; j < 4; j++) { const auto r = __builtin_amdgcn_permlane32_swap(hi4[j], lo4[j], false, false); o1[j] = r[0]; o2[j] = r[1]; }
    mfB1 = __builtin_bit_cast(nb2_h8, uint4{o1[0], o1[1], o1[2], o1[3]});
    mfB2 = __builtin_bit_cast(nb2_h8, uint4{o2[0], o2[1], o2[2], o2[3]});
    const float cutp = 1.001f * (float)cutneigh + 0.01f;
    const float sx = fabsf(fxi) + cutp, sy = fabsf(fyi) + cutp, sz = fabsf(fzi) + cutp;
    const float Ti = sx * sx + sy * sy + sz * sz + fabsf(thf) + 2.0f * (sx * fabsf(fxi) + sy * fabsf(fyi) + sz * fabsf(fzi)) + (float)cutneighsq;
#ifdef NB2_MF_NOWALK        // (timing probe only: no group is ever decided exactly)
    const float Ei = 0.0f * Ti;
#else
    const float Ei = owned ? 40.0f * 5.96046448e-08f * Ti : 0.0f;
#endif
    if constexpr(CORE) {                  // the same operands with the core radius' threshold in k 11, 12
      // (a pair inside the core radius at the build MUST land in the core part — CoreRows, mmd_internal.hpp —, one outside may: the threshold is raised by the
      //  atom's error bound, so the accumulator's sign can only err towards "core")
      const float tcf = owned ? (float)((double)core_thr + (double)Ei - aa_d) : -60000.0f;
      const float tch = (float)(_Float16)tcf;
      const unsigned c1 = nb2_pack_h2(fzi - b1.x, -tch), c2 = nb2_pack_h2(-(tcf - tch), 0.0f);
      const auto r1 = __builtin_amdgcn_permlane32_swap(h1, c1, false, false);
      const auto r2 = __builtin_amdgcn_permlane32_swap(h0, c2, false, false);
      mfB1c = __builtin_bit_cast(nb2_h8, uint4{o1[0], r1[0], r2[0], o1[3]});
      mfB2c = __builtin_bit_cast(nb2_h8, uint4{o2[0], r1[1], r2[1], o2[3]});
    }
    const auto re = __builtin_amdgcn_permlane32_swap(__float_as_uint(Ei), __float_as_uint(Ei), false, false);
    mfEA = __uint_as_float(re[0]); mfEB = __uint_as_float(re[1]);
  }
  if(MFK && mf_range_bad) {                // (a tile stretched over a near-empty pencil: the row kernel builds this list)
    if(lane == 0) { tile_max[tile] = 0; tile_ncand[tile] = 0; tile_cand[cbase] = nall; if(cand_src != nullptr) cand_src[cbase] = nall; tile_ghost[tile] = 0; tile_rowmax[tile] = 0; tile_rowsum[tile] = 0; atomicMax(&flags[3], 1); }
    return;
  }
  int S = 0, fill = 0;                     // size of the union so far / culled candidates waiting in the buffer (wave-uniform)
  int gcount = 0;                          // groups tested so far
  int cnt = 0;                             // my non-empty hit words so far
  constexpr bool REGP = NB2_REGPARK && !CORE;
  // REGP: my hit word of every group, in ONE register tuple written through the wave-uniform group number (s_set_gpr_idx + v_mov: round 6 — the switch over
  // 28 named registers it replaces compiled to a tree of scalar compares that copied half the tuple around per group: ~30 SALU + ~25 VALU instructions a group)
  typedef unsigned nb2_pwv __attribute__((ext_vector_type(32)));
  static_assert(NB2_NE <= 32, "one register per group");
  nb2_pwv pw = {};
  // scratch of this tile: NB2_NG x 64 words; CORE: a second list of the same shape
  constexpr int WSTRIDE = NB2_NG * 64 * (CORE ? 2 : 1);
  unsigned* __restrict__ ent_w = tile_words + (size_t)tile * WSTRIDE + lane;
  unsigned long long gmask = 0, gmask2 = 0;       // groups I parked a word for (core part / rest)
  int n = 0;                               // my row length
  // CORE: the row is written in two parts, the entries closer than core_thr at the build first ("core"), the rest of the skin behind
  // them: a force kernel may stop after the core part for as long as no atom has moved further than half the margin between the
  // core radius and the force cutoff since the build (mmd_internal.hpp: CoreRows). The lists above hold the core part, these the rest.
  unsigned* __restrict__ ent2_w = ent_w + NB2_NG * 64;
  int cnt2 = 0, n2 = 0;
  bool any_ghost = false;
  int ghost_top = -1;                      // MFK: largest atom index this lane gave a slot (any_ghost = one compare + ballot at the end instead of mask arithmetic per group)

  // ---- phase 2 + expansion over the buffered candidates
  auto flush = [&]() {
    const int fill8 = MFK ? (fill + 31) & ~31 : (fill + 7) & ~7;      // (MFK: whole groups of 32)
    if constexpr(MFK) {
      // pass 2 of the cull: {l, index} of the kept candidates -> the MFMA operand record {m_hi.xy | m_hi.z bb_hi | m_lo.xy | m_lo.z bb_lo} (m = -2 l, bb = |l|^2) in place,
      // the index into its own array (entries past `fill` of the last trip convert whatever the buffer held: nobody reads them, the padding below overwrites its share)
      for(int p = lane; p < fill; p += 64) {
        const float4 r = ((const float4*)s_buf)[p];
        const float mx = -2.0f * r.x, my = -2.0f * r.y, mz = -2.0f * r.z, bb = __builtin_fmaf(r.z, r.z, __builtin_fmaf(r.y, r.y, r.x * r.x));
        const unsigned h0 = nb2_pack_h2(mx, my), h1 = nb2_pack_h2(mz, bb);
        const nb2_f2 b0 = nb2_unpack_h2(h0), b1 = nb2_unpack_h2(h1);
        ((uint4*)s_buf)[p] = uint4{h0, h1, nb2_pack_h2(mx - b0.x, my - b0.y), nb2_pack_h2(mz - b1.x, bb - b1.y)};
        s_buf[NB2_IDX + p] = r.w;
      }
    }
    if(lane < fill8 - fill) {
      if(MFK) {                       // m = 0, |b|^2 = 60000: d > 0 against every atom
        ((uint4*)s_buf)[fill + lane] = uint4{0u, (unsigned)__builtin_bit_cast(unsigned short, (_Float16)60000.0f) << 16, 0u, 0u};
      } else
      if(DOTK) {                      // -2 b and |b|^2 of a candidate at (1e15, 1e15, 1e15)
        s_buf[fill + lane] = -2.0e15f; s_buf[NB2_BUF + fill + lane] = -2.0e15f; s_buf[2 * NB2_BUF + fill + lane] = -2.0e15f;
        s_buf[3 * NB2_BUF + fill + lane] = 3.0e30f;
      } else { s_buf[fill + lane] = 1.0e15f; s_buf[NB2_BUF + fill + lane] = 1.0e15f; s_buf[2 * NB2_BUF + fill + lane] = 1.0e15f; }
      s_buf[NB2_IDX + fill + lane] = __int_as_float((int)0x80000000);      // (an owned index as far as MODE 1 is concerned)
      if(MODE != 0) s_own[fill + lane] = (unsigned char)0xff;
    }
    __syncthreads();
    const int selfpos = (MODE == 0 || DOTK) ? (int)s_selfpos[lane] : -1;
    // MFK: group and bit of the lane's own candidate record, once per buffer (the atom itself is a hit of its own lane: dropped in its group)
    const int self_g = selfpos >> 5;                                   // (0xffff: group 2047, never reached)
    const unsigned self_keep = ~(1u << nb2_mf_bit(selfpos & 31));
    // MFK: the candidates' operand record of the NEXT group is read while this one is tested (one LDS round trip less in every group's chain; 4 VGPRs)
    uint4 rec_next = uint4{0u, 0u, 0u, 0u};
    if constexpr(MFK) { if(fill8 > 0) rec_next = ((const uint4*)s_buf)[lane & 31]; }
    for(int gq = 0; gq < fill8 && !(ablate & 2); gq += 32) {
      const int G = min(32, fill8 - gq);
      unsigned bits = 0, bits_hi = 0, bits_c = 0;
      if constexpr(MFK) {
        // (issuing a group's MFMAs one group ahead of the reads of their accumulators was built and measured 1.5 % slower: 119 instead of 108 VGPRs)
        const uint4 rec = rec_next;
        rec_next = ((const uint4*)s_buf)[min(gq + 32, fill8 - 32) + (lane & 31)];
        uint4 av = rec;
        av.y = (rec.y & mf_my) | (0x3c003c00u & ~mf_my);             // upper half: {m_hi.z, 1.0}
        av.z = (rec.z & mf_mz) | (0x3c003c00u & ~mf_mz);             //             {1.0, .}
        const nb2_h8 A = __builtin_bit_cast(nb2_h8, av);
        const nb2_f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const nb2_f16v d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, mfB1, zero, 0, 0, 0);
        const nb2_f16v d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, mfB2, zero, 0, 0, 0);
        unsigned w1 = 0, w2 = 0;
        float m1 = 3.0e38f, m2 = 3.0e38f;
#pragma unroll
        for(int i = 0; i < 16; i++) w1 = nb2_shift_sign(w1, d1[i]);
#pragma unroll
        for(int i = 0; i < 16; i += 2) m1 = __builtin_fminf(__builtin_fminf(m1, __builtin_fabsf(d1[i])), __builtin_fabsf(d1[i + 1]));      // (v_min3_f32 with |.| modifiers; NOT the inline-asm form: the compiler must see these reads of the MFMA's registers to keep the wait states in front of them)
#pragma unroll
        for(int i = 0; i < 16; i++) w2 = nb2_shift_sign(w2, d2[i]);
#pragma unroll
        for(int i = 0; i < 16; i += 2) m2 = __builtin_fminf(__builtin_fminf(m2, __builtin_fabsf(d2[i])), __builtin_fabsf(d2[i + 1]));
        const auto rw = __builtin_amdgcn_permlane32_swap(w1, w2, false, false);       // -> the two words of MY atom
        bits = (rw[0] << 16) | rw[1];
        if constexpr(CORE) {
          const nb2_f16v c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, mfB1c, zero, 0, 0, 0);
          const nb2_f16v c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, mfB2c, zero, 0, 0, 0);
          unsigned v1 = 0, v2 = 0;
#pragma unroll
          for(int i = 0; i < 16; i++) v1 = nb2_shift_sign(v1, c1[i]);
#pragma unroll
          for(int i = 0; i < 16; i++) v2 = nb2_shift_sign(v2, c2[i]);
          const auto rc = __builtin_amdgcn_permlane32_swap(v1, v2, false, false);
          bits_c = (rc[0] << 16) | rc[1];
        }
        const unsigned long long ambm = __builtin_amdgcn_fcmpf(m1, mfEA, 4 /* ordered < */) | __builtin_amdgcn_fcmpf(m2, mfEB, 4);      // (lane masks straight from the compares)
        if(ambm != 0ull) {
          // some accumulator of the group is inside its atom's band: the pairs of that HALF of the group's candidates (the lower 32 lanes hold rows
          // 0-3, 8-11, ... = the upper 16 bits of a word, the upper 32 lanes the rest) with all 64 atoms are decided exactly from the global positions
          const int jq = __float_as_int(s_buf[NB2_IDX + gq + (lane & 31)]);           // (padding: negative)
          const real4 pq = x[jq >= 0 ? jq : 0];
          const bool walk0 = (unsigned)ambm != 0u, walk1 = (unsigned)(ambm >> 32) != 0u;
          const unsigned wmask = (walk0 ? 0xffff0000u : 0u) | (walk1 ? 0x0000ffffu : 0u);
          unsigned ex = 0, exc = 0;
          for(int c = 0; c < 32; c++) {
            if(!((c & 4) ? walk1 : walk0)) continue;
            const real qx = nb2_readlane(pq.x, c), qy = nb2_readlane(pq.y, c), qz = nb2_readlane(pq.z, c);
            const int sj = __builtin_amdgcn_readlane(jq, c);
            const real dx = pme.x - qx, dy = pme.y - qy, dz = pme.z - qz;
            const real rsq = dx * dx + dy * dy + dz * dz;
            if(sj >= 0 && rsq <= cutneighsq) ex |= 1u << nb2_mf_bit(c);
            if(CORE && sj >= 0 && rsq <= (real)core_thr) exc |= 1u << nb2_mf_bit(c);         // (core_thr carries its own float margin)
          }
          bits = (bits & ~wmask) | (owned ? ex : 0u);
          if(CORE) bits_c = (bits_c & ~wmask) | (owned ? exc : 0u);
        }
      } else
      if constexpr(DOTK) {
        float acc = 3.0e38f;                         // smallest |d| of this lane in the group
        unsigned bits_zs = 0, bits_zp = 0;           // half lists: partner surely / possibly above me in z
        const nb2_f2 AX2 = {fxi, fxi}, AY2 = {fyi, fyi}, AZ2 = {fzi, fzi}, THR2 = {thr_i, thr_i}, THRC2 = {thrc_i, thrc_i};
        const nb2_f2 ZS2 = {twofz + ztol2, twofz + ztol2}, ZP2 = {twofz - ztol2, twofz - ztol2};
        for(int q = 0; q < G; q += 8) {
          const float4* vx = (const float4*)&s_buf[gq + q];
          const float4* vy = (const float4*)&s_buf[NB2_BUF + gq + q];
          const float4* vz = (const float4*)&s_buf[2 * NB2_BUF + gq + q];
          const float4* vb = (const float4*)&s_buf[3 * NB2_BUF + gq + q];
          nb2_f2 mx2[4], my2[4], mz2[4], bb2[4];
#pragma unroll
          for(int u = 0; u < 2; u++) {
            const float4 tx = vx[u], ty = vy[u], tz = vz[u], tb = vb[u];
            mx2[2 * u] = nb2_f2{tx.x, tx.y}; mx2[2 * u + 1] = nb2_f2{tx.z, tx.w};
            my2[2 * u] = nb2_f2{ty.x, ty.y}; my2[2 * u + 1] = nb2_f2{ty.z, ty.w};
            mz2[2 * u] = nb2_f2{tz.x, tz.y}; mz2[2 * u + 1] = nb2_f2{tz.z, tz.w};
            bb2[2 * u] = nb2_f2{tb.x, tb.y}; bb2[2 * u + 1] = nb2_f2{tb.z, tb.w};
          }
#pragma unroll
          for(int u2 = 0; u2 < 4; u2++) {
            const nb2_f2 t2 = __builtin_elementwise_fma(AZ2, mz2[u2], __builtin_elementwise_fma(AY2, my2[u2], __builtin_elementwise_fma(AX2, mx2[u2], bb2[u2])));
            const nb2_f2 d2 = t2 - THR2;
            bits = nb2_shift_sign(bits, d2.x); bits = nb2_shift_sign(bits, d2.y);
            acc = nb2_min3_abs(acc, d2.x, d2.y);
            if(CORE) { const nb2_f2 c2 = t2 - THRC2; bits_c = nb2_shift_sign(bits_c, c2.x); bits_c = nb2_shift_sign(bits_c, c2.y); }
            if(MODE != 0) {                          // zz = 2 (z_a - z_b) = twofz + (-2 z_b): above me <=> zz < 0
              const nb2_f2 zs = mz2[u2] + ZS2, zp = mz2[u2] + ZP2;
              bits_zs = nb2_shift_sign(bits_zs, zs.x); bits_zs = nb2_shift_sign(bits_zs, zs.y);
              bits_zp = nb2_shift_sign(bits_zp, zp.x); bits_zp = nb2_shift_sign(bits_zp, zp.y);
            }
          }
        }
        // candidate q of the group sits at bit G-1-q
        unsigned zamb = 0;                           // half lists: distance hits whose z order the float test cannot decide
        if(MODE != 0) {
          if(MODE == 1) {                            // without ghost newton a ghost partner is kept by whoever sees it
            const unsigned long long gm = __builtin_amdgcn_ballot_w64(lane < G && __float_as_int(s_buf[NB2_IDX + gq + lane]) >= nlocal);
            const unsigned gb = __brev((unsigned)gm) >> (32 - G);
            bits_zs |= gb; bits_zp |= gb;
          }
          zamb = bits & bits_zp & ~bits_zs;
          const unsigned sp = (unsigned)(selfpos - gq);      // the atom itself: a distance hit with zz = 0 — never a partner
          if(sp < (unsigned)G) zamb &= ~(1u << (G - 1 - sp));
          bits &= bits_zs;
        }
        const bool amb = acc < eps_dot || zamb != 0u;
        if(__builtin_amdgcn_ballot_w64(amb) != 0ull) {
          // some lane has a pair inside the error band: walk the group once, re-evaluate d per candidate (the same operations: the same
          // value) and decide the pairs inside the band exactly, in double, from the global positions (ref/neighbor.cpp:165,179)
          for(int qq = 0; qq < G; qq++) {
            const float mx = s_buf[gq + qq], my = s_buf[NB2_BUF + gq + qq], mz = s_buf[2 * NB2_BUF + gq + qq], bb = s_buf[3 * NB2_BUF + gq + qq];
            const float d = __builtin_fmaf(fzi, mz, __builtin_fmaf(fyi, my, __builtin_fmaf(fxi, mx, bb))) - thr_i;
            const unsigned bm = 1u << (G - 1 - qq);
            bool need = amb && fabsf(d) < eps_dot;
            if(MODE != 0) {
              // (a pair inside the distance band matters only where the z rule may keep it; the atom itself is never tested)
              need = (need && ((bits_zp & bm) != 0u) && (unsigned)(selfpos - gq) != (unsigned)qq) || ((zamb & bm) != 0u);
            }
            if(need) {
              const int jx = __float_as_int(s_buf[NB2_IDX + gq + qq]);
              const real4 pj = x[jx];
              const real dx = pme.x - pj.x, dy = pme.y - pj.y, dz = pme.z - pj.z;
              const real rsq = dx * dx + dy * dy + dz * dz;
              bool ok = rsq <= cutneighsq;
              if(MODE != 0 && !(MODE == 1 && jx >= nlocal))
                ok = ok && (pj.z > pme.z || (pj.z == pme.z && (pj.y > pme.y || (pj.y == pme.y && pj.x > pme.x))));
              bits = ok ? (bits | bm) : (bits & ~bm);
            }
          }
        }
      } else {
      for(int q = 0; q < G; q += 8) {
        // 8 buffered candidates per trip: 6 ds_read_b128 (uniform addresses). On this part a VALU instruction costs a
        // wavefront the same issue slot whether it is 32 or 64 bits wide, but v_pk_*_f32 handles TWO floats per lane: the
        // pre-test's rsq of two candidates takes 6 packed instructions (DP: a conservative filter, so fma is welcome).
        const float4* vx = (const float4*)&s_buf[gq + q];
        const float4* vy = (const float4*)&s_buf[NB2_BUF + gq + q];
        const float4* vz = (const float4*)&s_buf[2 * NB2_BUF + gq + q];
        nb2_f2 cx2[4], cy2[4], cz2[4];
#pragma unroll
        for(int u = 0; u < 2; u++) {
          const float4 tx = vx[u], ty = vy[u], tz = vz[u];
          cx2[2 * u] = nb2_f2{tx.x, tx.y}; cx2[2 * u + 1] = nb2_f2{tx.z, tx.w};
          cy2[2 * u] = nb2_f2{ty.x, ty.y}; cy2[2 * u + 1] = nb2_f2{ty.z, ty.w};
          cz2[2 * u] = nb2_f2{tz.x, tz.y}; cz2[2 * u + 1] = nb2_f2{tz.z, tz.w};
        }
        const nb2_f2 X2 = {fxi, fxi}, Y2 = {fyi, fyi}, Z2 = {fzi, fzi};
#pragma unroll
        for(int u2 = 0; u2 < 4; u2++) {
          const nb2_f2 dx = X2 - cx2[u2], dy = Y2 - cy2[u2], dz = Z2 - cz2[u2];
          nb2_f2 rsq2;
          if(NB2_PF) rsq2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
          else rsq2 = dx * dx + dy * dy + dz * dz;                 // (no contraction in this file: rounds like the reference)
#pragma unroll
          for(int hlf = 0; hlf < 2; hlf++) {
            const int u = 2 * u2 + hlf;
            const float rsq = hlf ? rsq2.y : rsq2.x;
            unsigned long long m, mh = 0;
            if(NB2_PF) {
              m = __builtin_amdgcn_fcmpf(rsq, cut_lo, 5 /* ordered <= */);
              mh = __builtin_amdgcn_fcmpf(rsq, cut_hi, 5);
            } else
              m = __builtin_amdgcn_fcmpf(rsq, (float)cutneighsq, 5 /* ordered <= : ref/neighbor.cpp:165,179 */);
            if(MODE != 0) {
              const float czv = hlf ? cz2[u2].y : cz2[u2].x;
              unsigned long long rule, rule_hi;
              if(NB2_PF) {
                rule = __builtin_amdgcn_fcmpf(czv, zlo, 2 /* ordered > */);
                rule_hi = __builtin_amdgcn_fcmpf(czv, zhi, 3 /* ordered >= */);
              } else {
                rule = __builtin_amdgcn_fcmpf(czv, fzi, 2);
                const unsigned long long zeq = __builtin_amdgcn_fcmpf(czv, fzi, 1 /* ordered == */);
                if(zeq) {                                            // (a lattice: whole planes share z)
                  const float cyv = hlf ? cy2[u2].y : cy2[u2].x, cxv = hlf ? cx2[u2].y : cx2[u2].x;
                  const unsigned long long ygt = __builtin_amdgcn_fcmpf(cyv, fyi, 2), yeq = __builtin_amdgcn_fcmpf(cyv, fyi, 1);
                  rule |= zeq & (ygt | (yeq & __builtin_amdgcn_fcmpf(cxv, fxi, 2)));
                }
                rule_hi = rule;
              }
              if(MODE == 1) {                                        // without ghost newton a ghost partner is kept by whoever sees it
                const int cju = __builtin_amdgcn_readfirstlane(__float_as_int(s_buf[NB2_IDX + gq + q + u]));
                if(cju >= nlocal) { rule = ~0ull; rule_hi = ~0ull; }
              }
              m &= rule; mh &= rule_hi;
            }
            bits = nb2_shift_in(bits, m);
            if(NB2_PF) bits_hi = nb2_shift_in(bits_hi, mh);
            if(CORE) bits_c = nb2_shift_in(bits_c, __builtin_amdgcn_fcmpf(rsq, core_thr, 5));     // (a classification, not a decision: float is enough)
          }
        }
      }
      // pass q of the group sits at bit G-1-q
      if(NB2_PF) {
        unsigned amb = bits_hi & ~bits;                            // between the thresholds: decide exactly
        if(__builtin_amdgcn_ballot_w64(amb != 0u) != 0ull) {
          while(amb) {
            const int bq = __builtin_ctz(amb);
            amb &= amb - 1;
            const int jx = __float_as_int(s_buf[NB2_IDX + gq + (G - 1 - bq)]);
            const real4 pj = x[jx];
            const real dx = pme.x - pj.x, dy = pme.y - pj.y, dz = pme.z - pj.z;
            const real rsq = dx * dx + dy * dy + dz * dz;
            bool ok = rsq <= cutneighsq;
            if(MODE != 0 && !(MODE == 1 && jx >= nlocal))
              ok = ok && (pj.z > pme.z || (pj.z == pme.z && (pj.y > pme.y || (pj.y == pme.y && pj.x > pme.x))));
            if(ok) bits |= 1u << bq;
          }
        }
      }
      }       // (difference form)
      // full lists: the atom itself (rsq = 0) is a hit of its own lane: dropped here
      if(MFK) bits &= (gq >> 5) == self_g ? self_keep : ~0u;
      else if(MODE == 0) { const unsigned sp = (unsigned)(selfpos - gq); if(sp < (unsigned)G) bits &= ~(1u << (G - 1 - (int)sp)); }
      // ---- the candidates some lane keeps form the tile's union: they get the next slots, in candidate order
      const unsigned used = wave_or_u(bits);
      const int bq_mine = MFK ? nb2_mf_bit(lane & 31) : G - 1 - lane;      // the bit candidate `lane` of the group sits at
      if(lane < G) {
        const int bq = bq_mine;
        if((used >> bq) & 1u) {
          const int slot = S + __popc(used >> 1 >> bq);           // used candidates before mine
          if(slot < cstride - 1) {
            const int cj = __float_as_int(s_buf[NB2_IDX + gq + lane]);
            tile_cand[cbase + slot] = cj;
            if(MFK) ghost_top = max(ghost_top, cj);
            // (one rank: the same list with a ghost named by its owner and image code, for tile kernels that stage ghosts from their owners)
            // (several ranks, ghost_image == nullptr: ghost_root is DirectHalo::gmap — the entry of the position buffer the per-step halo delivers the ghost to)
            if(cand_src != nullptr) cand_src[cbase + slot] = cj >= nlocal && cj < nall ? (ghost_image != nullptr ? (ghost_root[cj - nlocal] | ((ghost_image[cj - nlocal] + 1) << MMD_SRC_BITS)) : ghost_root[cj - nlocal]) : cj;
          }
          if(MODE != 0) { const unsigned own = s_own[gq + lane]; if(own != 0xffu) s_self[own] = (unsigned short)slot; }
        }
      }
      if(!MFK) any_ghost = any_ghost || __builtin_amdgcn_ballot_w64(lane < G && ((used >> (bq_mine & 31)) & 1u) && __float_as_int(s_buf[NB2_IDX + gq + lane]) >= nlocal) != 0ull;
      // ---- a lane's NON-EMPTY hit words wait, with their group numbers, in a scratch list (lane-interleaved, read back by
      // the same lane) for the lock-step expansion at the end of the tile
      if(gcount < NB2_NG) {
        if(CORE) {
          bits_c &= bits;
          const unsigned rest = bits & ~bits_c;
          if(bits_c != 0u) { ent_w[(unsigned)cnt * 64u] = bits_c; gmask |= 1ull << gcount; cnt++; }
          if(rest != 0u) { ent2_w[(unsigned)cnt2 * 64u] = rest; gmask2 |= 1ull << gcount; cnt2++; }
          n2 += __popc(rest);
          bits = bits_c;                                         // (n counts the core part below)
        } else
        if(REGP) {
          pw[min(gcount, 31)] = bits;        // (unconditional: a guarded write makes the compiler copy the tuple and select 32 registers; groups >= NB2_NE raise maxcnt_over, their words are never read)
        } else
        if(bits != 0u) { ent_w[(unsigned)cnt * 64u] = bits; gmask |= 1ull << gcount; cnt++; }
        if(lane == 0) s_gSU[gcount] = uint2{(unsigned)S, used};
      }
      gcount++;
      n += __popc(bits);
      S += __popc(used);
    }
    if(MODE == 0 || DOTK) s_selfpos[lane] = (unsigned short)0xffff;
    __syncthreads();
    fill = 0;
  };

  // ---- phase 1: stream the candidates, slice by slice (a slice = one contiguous run of binned[]: no per-element address
  // search; chunks are the 64-entry pieces of the slices, the last piece of a slice is partly empty)
  // Round 6: no scalar cursor over the slices (it cost ~10 scalar instructions per piece and ~25 at every slice boundary): every slice knows how many 64-entry
  // pieces it has, a prefix sum places them, and a window of 64 pieces at a time sits in LDS as {start, entries}; a batch reads its four descriptors with two
  // broadcast ds_read_b128 and every lane forms its own address.
  static_assert(NB2_BATCH == 4, "a batch reads its descriptors as two int4");
  const int sl_pieces = (sl_len + 63) >> 6;
  const int sl_first = wave_incl_scan(sl_pieces) - sl_pieces;
  const int npieces = (ablate & 8) ? 0 : __builtin_amdgcn_readlane(sl_first + sl_pieces, 63);
  // MFK: half extents of the owned atoms' box about the local origin (its centre up to float rounding)
  const float hx = fmaxf(bx1 - (float)ox, (float)ox - bx0) * 1.000001f + 1.0e-30f, hy = fmaxf(by1 - (float)oy, (float)oy - by0) * 1.000001f + 1.0e-30f,
              hz = fmaxf(bz1 - (float)oz, (float)oz - bz0) * 1.000001f + 1.0e-30f;
  for(int pbase = 0; pbase < npieces; pbase += 64) {
    __syncthreads();
    s_clen[lane] = 0;
    __syncthreads();
    for(int k = 0; k < sl_pieces; k++) {
      const unsigned c = (unsigned)(sl_first + k - pbase);
      if(c < 64u) { s_cstart[c] = sl_start + 64 * k; s_clen[c] = min(64, sl_len - 64 * k); }
    }
    __syncthreads();
    const int nwin = min(64, npieces - pbase);
  for(int pc = 0; pc < nwin; pc += NB2_BATCH) {
    int jj[NB2_BATCH], aa[NB2_BATCH];
    {
      const int4 st = *(const int4*)&s_cstart[pc], ln = *(const int4*)&s_clen[pc];
      const int st4[4] = {st.x, st.y, st.z, st.w}, ln4[4] = {ln.x, ln.y, ln.z, ln.w};
#pragma unroll
      for(int u = 0; u < NB2_BATCH; u++) {
        const bool ok = lane < ln4[u];
        const int a = st4[u] + lane;
        const int jv = binned[ok ? a : 0];
        aa[u] = ok ? a : -1; jj[u] = ok ? jv : -1;
      }
    }
    real4 pp[NB2_BATCH];
#pragma unroll
    for(int u = 0; u < NB2_BATCH; u++) pp[u] = x[jj[u] >= 0 ? jj[u] : 0];
#pragma unroll
    for(int u = 0; u < NB2_BATCH; u++) {
      const int j = jj[u];
      bool keep = j >= 0 && !(ablate & 16);
      const int cjv = j;
      float lx = 0, ly = 0, lz = 0;
      unsigned long long mk = 0ull;
      if constexpr(MFK) {
        // coordinates about the tile's centre serve the cull AND (pass 2, in flush) the candidate's record: distance to the box = |l| - half extent per dimension
        lx = (float)(pp[u].x - ox); ly = (float)(pp[u].y - oy); lz = (float)(pp[u].z - oz);
        const float ddx = fmaxf(fabsf(lx) - hx, 0.0f), ddy = fmaxf(fabsf(ly) - hy, 0.0f), ddz = fmaxf(fabsf(lz) - hz, 0.0f);
        const float dd2 = ddx * ddx + ddy * ddy + ddz * ddz;
        keep = keep && (dd2 <= cull);
        // (the lane mask straight from the two compares: ballot(bool) of a combined condition compiles to v_cndmask + v_cmp_ne on top of them)
        mk = __builtin_amdgcn_sicmp(j, -1, 38 /* signed > */) & __builtin_amdgcn_fcmpf(dd2, cull, 5 /* ordered <= */);
        if(ablate & 16) mk = 0ull;
      } else {
        const float fx = (float)pp[u].x, fy = (float)pp[u].y, fz = (float)pp[u].z;
        const float ddx = fmaxf(fmaxf(bx0 - fx, fx - bx1), 0.0f);
        const float ddy = fmaxf(fmaxf(by0 - fy, fy - by1), 0.0f);
        const float ddz = fmaxf(fmaxf(bz0 - fz, fz - bz1), 0.0f);
        keep = keep && (ddx * ddx + ddy * ddy + ddz * ddz <= cull);
        if(MODE != 0) keep = keep && (fz >= zcull || (MODE == 1 && j >= nlocal));
      }
      const unsigned long long m = MFK ? mk : __builtin_amdgcn_ballot_w64(keep);
      if(m) {
        const int pos = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if(keep) {
          if constexpr(MFK) {
            // pass 1 parks {l, index}; the f16 hi / lo record is made from it in flush(), 64 KEPT candidates per trip instead of 64 streamed ones
            ((float4*)s_buf)[pos] = float4{lx, ly, lz, __int_as_float(cjv)};
          } else {
          lx = (float)(pp[u].x - ox); ly = (float)(pp[u].y - oy); lz = (float)(pp[u].z - oz);
          if(DOTK) {
            s_buf[pos] = -2.0f * lx; s_buf[NB2_BUF + pos] = -2.0f * ly; s_buf[2 * NB2_BUF + pos] = -2.0f * lz;
            s_buf[3 * NB2_BUF + pos] = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
          } else { s_buf[pos] = lx; s_buf[NB2_BUF + pos] = ly; s_buf[2 * NB2_BUF + pos] = lz; }
          s_buf[NB2_IDX + pos] = __int_as_float(cjv);
          }
          const unsigned own = (unsigned)(aa[u] - ta);                   // the tile's own atoms are binned[ta .. ta+63]
          if(MODE != 0) s_own[pos] = own < 64u ? (unsigned char)own : (unsigned char)0xff;
          if((MODE == 0 || DOTK) && own < 64u) s_selfpos[own] = (unsigned short)pos;
        }
        fill += __popcll(m);
      }
    }
    if(fill > NB2_BUF - 64 * NB2_BATCH) flush();        // (one copy of the test code: flushed between batches only)
  }
  }
  flush();

  // ---- lock-step expansion: in round k every lane emits the k-th entry of its row (the slot of its next set bit, or the
  // dummy slot = S, staged by the force kernels behind the candidates, once its bits are used up), so every store is one whole
  // 128-byte line of nl16 and the padding comes for free. A lane's words are walked in group order; s_gS / s_gU give the slot
  // base and the union mask of a group.
  // (CORE: n = core entries, n2 = the rest; otherwise n = the row)
  if(MFK) any_ghost = __builtin_amdgcn_ballot_w64(ghost_top >= nlocal) != 0ull;
  const int maxn = (int)wave_max_u((unsigned)(n + n2));
  const int kc = min(((int)wave_max_u((unsigned)n) + NB_ROW_PAD - 1) / NB_ROW_PAD * NB_ROW_PAD, maxneighs);
  const int kr = CORE ? min(((int)wave_max_u((unsigned)n2) + NB_ROW_PAD - 1) / NB_ROW_PAD * NB_ROW_PAD, maxneighs - kc) : 0;
  const int kneed = ((int)wave_max_u((unsigned)n) + NB_ROW_PAD - 1) / NB_ROW_PAD * NB_ROW_PAD +
                    (CORE ? ((int)wave_max_u((unsigned)n2) + NB_ROW_PAD - 1) / NB_ROW_PAD * NB_ROW_PAD : 0);     // rows the tile wants (may exceed maxneighs)
  const int kmax = kc + kr;
  const unsigned short dummy = (unsigned short)(S * NB_SLOT_BYTES);
  bool maxcnt_over = false;
  // one part of the rows: entries of the lanes' list (src_w/src_g, mycnt of them) go to rows [kfirst, kfirst + krows)
  auto expand = [&](const unsigned* __restrict__ src_w, unsigned long long groups, int mycnt, int kfirst, int krows) {
    __syncthreads();
    // the lanes' lists come back from the scratch into LDS (the candidate buffer is free now), a few loads in flight at a time:
    // inside the loop a global load would put a full memory round trip into every round (s_waitcnt vmcnt(0) also waits for the
    // row stores), LDS reads do not
    unsigned* s_ew = (unsigned*)s_buf;
    if(REGP) {
      // the non-empty words go to consecutive entries of the lane's column, their groups into the mask: an empty word's store is overwritten by the next one
      int c = 0;
      unsigned long long gm = 0;
#pragma unroll
      for(int i = 0; i < NB2_NE; i++) {
        if(i >= gcount) break;                                   // (wave-uniform)
        const unsigned w = pw[i];
        s_ew[c * 64 + lane] = w;
        c += w != 0u ? 1 : 0;
        gm |= w != 0u ? 1ull << i : 0ull;
      }
      mycnt = c; groups = gm;
      maxcnt_over = maxcnt_over || gcount > NB2_NE;             // (more groups than registers: the row kernel builds the lists)
    }
    const int wmax = (int)wave_max_u((unsigned)mycnt);
    maxcnt_over = maxcnt_over || wmax > NB2_NE;               // (a lane with more non-empty words than the LDS list holds)
    const int maxcnt = min(wmax, NB2_NE);
    for(int e0 = 0; e0 < maxcnt && !REGP; e0 += 4) {
      unsigned tw[4];
#pragma unroll
      for(int u = 0; u < 4; u++) { tw[u] = 0; if(e0 + u < mycnt) tw[u] = src_w[(unsigned)(e0 + u) * 64u]; }
#pragma unroll
      for(int u = 0; u < 4; u++) if(e0 + u < NB2_NE) s_ew[(e0 + u) * 64 + lane] = tw[u];
    }
    __syncthreads();
    const int cn = min(mycnt, NB2_NE);
    unsigned w0 = 0, b0 = 0, u0 = 0;
    int e = 0;                                   // list entry of w0
    // the group of the next entry = the lowest set bit of `groups` (entries were parked in group order)
    if(cn > 0) { w0 = s_ew[lane]; const uint2 su = s_gSU[__builtin_ctzll(groups)]; groups &= groups - 1; b0 = su.x; u0 = su.y; }
    for(int k = 0; k < krows && !(ablate & 1); k++) {
      if((int)(w0 == 0u) & (int)(e + 1 < cn)) {  // (entries are non-empty: one hop always lands on a set bit; `&`: ONE exec region, not two nested ones)
        e++;
        w0 = s_ew[e * 64 + lane];
        const uint2 su = s_gSU[__builtin_ctzll(groups)];
        groups &= groups - 1;
        b0 = su.x; u0 = su.y;
      }
      const bool v = w0 != 0u;
      const int bq = __builtin_ctz(w0 | 0x80000000u);
      w0 &= w0 - 1;
      const unsigned slot = b0 + (unsigned)__popc(u0 >> 1 >> bq);
      // (arithmetic select: the compiler wrapped `v ? slot : dummy` into an exec region of its own, three scalar instructions + a branch per round)
      const unsigned dm = (unsigned)dummy;
      rowp[(unsigned)(kfirst + k) * 64u] = (unsigned short)(dm + ((slot * NB_SLOT_BYTES - dm) & (0u - (unsigned)v)));
    }
  };
  expand(ent_w, gmask, cnt, 0, kc);
  if(CORE) expand(ent2_w, gmask2, cnt2, kc, kr);
  n += n2;
  if(CORE && owned) xbuild[ii] = pme;
  if(owned) numneigh[ii] = n;
  if(MODE != 0) tile_self[(size_t)tile * 64 + lane] = owned ? s_self[lane] : (unsigned short)0xffff;
  const int tsum = wave_sum(n);
  if(lane == 0) {
    tile_max[tile] = kmax;
    if(CORE) tile_kcore[tile] = kc;
    if(kneed > maxneighs) atomicMax(&flags[7], kneed);       // (rare) the two padded parts do not fit the row capacity: the host grows it
    tile_ncand[tile] = S;
    tile_cand[cbase + min(S, cstride - 1)] = nall;          // the dummy atom closes the list
    if(cand_src != nullptr) cand_src[cbase + min(S, cstride - 1)] = nall;
    tile_ghost[tile] = any_ghost ? 1 : 0;
    if(any_ghost && cand_src == nullptr && cand_src_wanted) atomicMax(&flags[3], 1);      // (cannot happen: a ghost candidate without a face in reach — the row build takes over)
    // per-tile results, reduced by k_tile_reduce: 36 k workgroups hammering three global words with atomics cost
    // 0.8 ms at -s 80 (same-address atomics retire one at a time, ~10 ns each)
    tile_rowmax[tile] = maxn;
    tile_rowsum[tile] = tsum;
    if(S > cstride - 2 || (S + 1) * NB_SLOT_BYTES > 65535 || gcount > NB2_NG || maxcnt_over) atomicMax(&flags[3], 1);   // (rare) union does not fit the 16-bit slot offsets / the word scratch
  }
}

// flags[0] = longest row, flags[2] = largest union, *total_out = sum of the row lengths
__global__ __launch_bounds__(1024) void k_tile_reduce(const int* __restrict__ tile_rowmax, const int* __restrict__ tile_rowsum,
                                                      const int* __restrict__ tile_ncand, int ntiles, int* __restrict__ flags,
                                                      unsigned long long* __restrict__ total_out, const int* __restrict__ ntiles_dev,
                                                      const int* __restrict__ bst)
{
  // a few workgroups, one slice of the tiles each, three atomics per workgroup into words k_pencil_fill zeroed
  // (one workgroup walking all 32 k tiles took 20 us)
  __shared__ int s_a[16], s_b[16];
  if(blockIdx.x == 0 && threadIdx.x == 0) *(long long*)(flags + 60) = wall_clock64();                   // end of the build phase (see k_bin_count)
  if(blockIdx.x == 0 && bst && threadIdx.x < 40) flags[16 + threadIdx.x] = bst[threadIdx.x];          // deferred one-rank borders: its counts travel with the flags
  if(ntiles_dev) { if(blockIdx.x == 0 && threadIdx.x == 0) flags[6] = *ntiles_dev; ntiles = min(ntiles, *ntiles_dev); }     // flags[6]: the count for the host
  __shared__ long long s_c[16];
  int a = 0, b = 0;
  long long c = 0;
  for(int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) { a = max(a, tile_rowmax[t]); b = max(b, tile_ncand[t]); c += tile_rowsum[t]; }
  a = wave_max_i(a); b = wave_max_i(b); c = wave_sum(c);
  if((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; s_c[threadIdx.x >> 6] = c; }
  __syncthreads();
  if(threadIdx.x == 0) {
    for(int w = 1; w < (int)(blockDim.x >> 6); w++) { a = max(a, s_a[w]); b = max(b, s_b[w]); c += s_c[w]; }
    atomicMax(&flags[0], a); atomicMax(&flags[2], b); atomicAdd(total_out, (unsigned long long)c);
  }
}

// interior tiles (no ghost among their candidates: they can run while this step's halo is in flight) first, boundary tiles behind
// them, each group in ascending order: count / scatter pair over blocks of 1024 tiles whose prefix every scatter block sums itself
// (no scan launch). The number of interior tiles goes to flags[13] and returns to the host with the build's other results.
__global__ __launch_bounds__(256) void k_tile_order_count(const int* __restrict__ tile_ghost, int ntiles, const int* __restrict__ ntiles_dev,
                                                          int* __restrict__ cnt)
{
  __shared__ int lds[17];
  if(ntiles_dev) ntiles = min(ntiles, *ntiles_dev);
  const int base = blockIdx.x * 1024 + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) if(base + k < ntiles) c += tile_ghost[base + k] == 0 ? 1 : 0;
  int tot;
  block_incl_scan(c, lds, &tot);
  if(threadIdx.x == 0) cnt[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_tile_order_scatter(const int* __restrict__ tile_ghost, int ntiles, const int* __restrict__ ntiles_dev,
                                                            const int* __restrict__ cnt, int* __restrict__ order, int* __restrict__ flags)
{
  __shared__ int lds[17];
  if(ntiles_dev) ntiles = min(ntiles, *ntiles_dev);
  int before = 0, total = 0;
  for(int t = threadIdx.x; t < (int)gridDim.x; t += 256) { const int v = cnt[t]; total += v; if(t < (int)blockIdx.x) before += v; }
  int tb, tt;
  block_incl_scan(before, lds, &tb);
  block_incl_scan(total, lds, &tt);
  const int base = blockIdx.x * 1024 + threadIdx.x * 4;
  bool in[4];
  int c = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) { in[k] = base + k < ntiles && tile_ghost[base + k] == 0; c += in[k] ? 1 : 0; }
  int tot;
  const int inc = block_incl_scan(c, lds, &tot);
  int pos_int = tb + inc - c;                       // interior tiles in front of this thread's first item
#pragma unroll
  for(int k = 0; k < 4; k++) {
    const int t = base + k;
    if(t < ntiles) {
      if(in[k]) order[pos_int++] = t;
      else order[tt + (t - pos_int)] = t;           // boundary tiles in front of t = t - (interior tiles in front of t)
    }
  }
  if(blockIdx.x == 0 && threadIdx.x == 0) flags[13] = tt;
}

// reference-style rows from the tile form: neigh[((i>>6)*maxneighs + k)*64 + (i&63)] = tile_cand[slot]
__global__ __launch_bounds__(64) void k_tiles_to_rows(int nlocal, int maxneighs, int cstride, const int* __restrict__ binned,
                                                      const int* __restrict__ tile_first, const int* __restrict__ tile_cnt,
                                                      const int* __restrict__ tile_cand, const unsigned short* __restrict__ nl16,
                                                      const int* __restrict__ numneigh, int* __restrict__ neigh,
                                                      const int* __restrict__ tile_max, const int* __restrict__ tile_ncand)
{
  const int tile = blockIdx.x, lane = threadIdx.x;
  int i = lane < tile_cnt[tile] ? binned[tile_first[tile] + lane] : -1;
  if(i >= nlocal) i = -1;
  const int n = i >= 0 ? min(numneigh[i], maxneighs) : 0;
  const unsigned short* in = nl16 + ((size_t)tile * maxneighs) * 64 + lane;
  const int* cl = tile_cand + (size_t)tile * cstride;
  const size_t rowbase = i >= 0 ? ((size_t)(i >> 6) * maxneighs) * 64 + (i & 63) : 0;
  // (rows written in two padded parts hold dummy entries between them: skipped)
  const int kmax = tile_max[tile], dummy = tile_ncand[tile];
  int m = 0;
  for(int k = 0; k < kmax && m < n; k++) {
    const int slot = in[(size_t)k * 64] / NB_SLOT_BYTES;
    if(slot != dummy) { neigh[rowbase + (size_t)m * 64] = cl[slot]; m++; }
  }
}

// pad every row with the dummy atom up to its wavefront's longest row (rounded up to the unroll factor)
// so the force kernels run a wave-uniform, branch-free neighbor loop.
__global__ __launch_bounds__(256) void k_pad_rows(int nlocal, int nwaves, int maxneighs, int dummy,
                                                  int* __restrict__ neigh, const int* __restrict__ numneigh,
                                                  int* __restrict__ wave_max, unsigned long long* __restrict__ total)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = i >> 6;
  if(w >= nwaves) return;
  const int n = i < nlocal ? numneigh[i] : 0;
  int m = wave_max_i(n);
  m = (m + MMD_UNROLL - 1) / MMD_UNROLL * MMD_UNROLL;
  if(m > maxneighs) m = maxneighs;
  const size_t rowbase = ((size_t)w * maxneighs) * 64 + (i & 63);
  for(int k = n; k < m; k++) neigh[rowbase + (size_t)k * 64] = dummy;
  if((i & 63) == 0) wave_max[w] = m;
  const long long s = wave_sum((long long)n);
  if((i & 63) == 0 && s) atomicAdd(total, (unsigned long long)s);
}

// pad + wave_max + statistics for the 32-bit rows
static int finish_rows(mmd_handle* h, bool count_total)
{
  const int nlocal = h->nlocal, nall = h->nlocal + h->nghost, nwaves = div_up(nlocal, 64);
  if(count_total) HIP_TRY(hipMemsetAsync(h->d_result, 0, sizeof(double), h->stream));
  if(nwaves)
    hipLaunchKernelGGL(k_pad_rows, dim3(div_up(nwaves * 64, 256)), dim3(256), 0, h->stream, nlocal, nwaves, h->maxneighs, nall,
                       h->neigh.p, h->numneigh.p, h->wave_max.p, (unsigned long long*)(count_total ? h->d_result : h->d_result + 16));
  HIP_TRY(hipGetLastError());
  if(count_total) {
    HIP_TRY(hipMemcpyAsync(h->h_result, h->d_result, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    unsigned long long tot;
    memcpy(&tot, h->h_result, sizeof(tot));
    h->total_neigh = (long long)tot;
  }
  h->rows_ready = true;
  return 0;
}

int mmd_ensure_rows(mmd_handle* h)
{
  if(h->rows_ready) return 0;
  if(!h->tiles_ready) { mmd_set_error("no neighbor list has been built"); return -1; }
  const int nwaves = div_up(h->nlocal, 64);
  MMD_TRY(h->neigh.ensure((size_t)nwaves * h->maxneighs * 64 + 64, false, h->stream));
  MMD_TRY(h->wave_max.ensure((size_t)nwaves + 1, false, h->stream));
  if(h->ntiles)
    hipLaunchKernelGGL(k_tiles_to_rows, dim3(h->ntiles), dim3(64), 0, h->stream, h->nlocal, h->maxneighs, h->tile_cstride, h->binned.p,
                       h->tile_first.p, h->tile_cnt.p, h->tile_cand.p, h->nl16.p, h->numneigh.p, h->neigh.p, h->tile_max.p, h->tile_ncand.p);
  HIP_TRY(hipGetLastError());
  return finish_rows(h, false);
}

// The build's result words travel to the host as stores of a kernel into pinned memory, closed by a sequence number with system scope;
// the host spins on that word (no runtime call in the loop). After two seconds without it the blocking wait takes over (and reports
// whatever went wrong on the stream).
// The verdict a force launch enqueued behind the build waits on (SpecLaunch, mmd_internal.hpp) is formed here, from the same words and by the
// same rules the host applies after reading them (mmd_neighbor_build below): 1 = rows, unions, tile count and the deferred borders all fit
// what that launch was sized for. Word 15 of the flags, published with the others.
struct BuildVerdict { int on, maxneighs, ntiles_cap, nt_async, cmax, has_bst, est_nb, big_bins, core_rows; };
#define NB_GATE_WORD 15
// Small systems (<= 4096 tiles): this one wavefront also does k_tile_reduce's work (one dependent launch less in a window that is nothing but
// dependent 5 us launches; the words it accumulates into were zeroed by k_pencil_fill_scan)
struct TileReduceArgs { int on; const int* rowmax; const int* rowsum; const int* ncand; int ntiles; const int* ntiles_dev; const int* bst; };
__global__ __launch_bounds__(64) void k_publish_flags(int* __restrict__ src, int* __restrict__ dst, int n, int seq, BuildVerdict V, TileReduceArgs R)
{
  const int t = threadIdx.x;
  if(R.on) {
    int nt = R.ntiles;
    if(R.ntiles_dev) nt = min(nt, *R.ntiles_dev);
    if(t == 0) *(long long*)(src + 60) = wall_clock64();                     // end of the build phase (see k_bin_count)
    if(R.bst && t < 40) src[16 + t] = R.bst[t];                              // deferred one-rank borders: its counts travel with the flags
    int a = 0, b = 0;
    long long c = 0;
    for(int q = t; q < nt; q += 64) { a = max(a, R.rowmax[q]); b = max(b, R.ncand[q]); c += R.rowsum[q]; }
    a = wave_max_i(a); b = wave_max_i(b); c = wave_sum(c);
    if(t == 0) {
      if(R.ntiles_dev) src[6] = *R.ntiles_dev;
      src[0] = max(src[0], a); src[2] = max(src[2], b);
      *(unsigned long long*)(src + 4) += (unsigned long long)c;
    }
    __syncthreads();
  }
  if(V.on && t == 0) {
    const int maxn = src[0], need = V.core_rows ? max(maxn, src[7]) : maxn;
    bool ok = !(maxn >= V.maxneighs || need > V.maxneighs) && src[3] == 0 && !(src[12] != 0 && !V.big_bins) && src[2] <= V.cmax;
    if(V.nt_async) ok = ok && src[6] <= V.ntiles_cap;
    if(V.has_bst) ok = ok && src[16 + 1] == 0 && src[16 + 0] <= V.est_nb;        // BST_OVF, BST_NB of the deferred one-rank borders (comm.hip)
    src[NB_GATE_WORD] = ok ? 1 : 0;
  }
  __syncthreads();
  if(t < n) dst[t] = src[t];
  __threadfence_system();
  __syncthreads();
  if(t == 0) __hip_atomic_store(dst + 62, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (Publishing from the last workgroup of k_tile_reduce instead — one launch less — was built and measured: the 1024-thread workgroup's system-scope
//  fence makes that kernel 14 instead of 6 us and bench.py --size 32 loses 2.7 %: 4480 against 4600 Matom-steps/s. A one-wavefront kernel it stays.)
static int flags_publish(mmd_handle* h, int n, const BuildVerdict& V, const TileReduceArgs& R)
{
  if(!h->h_flags_dev) HIP_TRY(hipHostGetDevicePointer((void**)&h->h_flags_dev, h->h_flags, 0));
  const int seq = ++h->flag_seq;
  hipLaunchKernelGGL(k_publish_flags, dim3(1), dim3(64), 0, h->stream, h->d_flags, h->h_flags_dev, n, seq, V, R);
  HIP_TRY(hipGetLastError());
  return 0;
}
static int flags_wait(mmd_handle* h)
{
  const int seq = h->flag_seq;
  h->host_syncs++;
  const double t0 = mmd_wall();
  unsigned long spins = 0;
  while(__atomic_load_n(&h->h_flags[62], __ATOMIC_ACQUIRE) != seq) {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
    __builtin_ia32_pause();
#endif
    if((++spins & 0x3fff) == 0 && mmd_wall() - t0 > 2.0) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if(__atomic_load_n(&h->h_flags[62], __ATOMIC_ACQUIRE) != seq) { mmd_set_error("neighbor build: the result words did not reach the host"); return -1; }
      break;
    }
  }
  return 0;
}

extern "C" int mmd_neighbor_build(mmd_handle* h)
{
  if(!h || !h->neigh_ready) { mmd_set_error("mmd_neighbor_build: call mmd_neighbor_setup first"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  if(h->nghost_dev && !(h->opt_tiles && h->nlocal > 0)) {       // only k_build_rows reads the deferred count
    const int rc = mmd_borders_deferred_resolve(h);
    if(rc < 0) return rc;
  }
  const int nlocal = h->nlocal, nall = h->nlocal + h->nghost;
  const int nwaves = div_up(nlocal, 64);
  const BinGeom& g = h->bg;
  // (re-neighborings inside a run with the production build: the binning pass also collects what k_pencil_count would — one dependent launch less)
  h->pencil_lohi_req = h->opt_tiles && nlocal > 0 && h->ntiles_hint > 0;
  MMD_TRY(mmd_bin_atoms(h, -1));
  MMD_TRY(h->numneigh.ensure((size_t)nlocal + 64, false, h->stream));
  const int nblocks = g.nblk[0] * g.nblk[1] * g.nblk[2];
  h->tiles_ready = false;
  h->cand_src_ready = false;
  h->rows_ready = false;
  h->rows_uploaded = false;
  h->neigh_nlocal = 0;
  // ---- tile form: 16-bit rows of LDS record offsets + per-tile candidate union, see k_build_rows
  bool want_tiles = h->opt_tiles && nlocal > 0;
  if(want_tiles) {
    // pencil tiles: 64-atom pieces of a row of blocks
    const int nunits = g.nblk[1] * g.nblk[2];
    MMD_TRY(h->tile_of_block.ensure((size_t)nblocks + 2, false, h->stream));
    const bool lohi = h->pencil_lohi_ready && h->ntiles_hint > 0;           // (k_bin_sort collected the pencils' owned bins: no k_pencil_count)
    h->pencil_lohi_ready = false;
    MMD_TRY(h->pencil_range.ensure((size_t)2 * nunits + 2, false, h->stream));
    if(!lohi) hipLaunchKernelGGL(k_pencil_count, dim3(nunits), dim3(64), 0, h->stream, h->binned.p, h->bin_start.p, nunits, g.nblk[0], nlocal, h->tile_of_block.p, h->pencil_range.p);
    // the tile count sizes the lists. Once a build has succeeded the previous count (+3 %) does that and the count itself
    // comes back with the build's result flags: one host synchronisation less per re-neighboring
    int nt = 0;
    const bool nt_async = h->ntiles_hint > 0;
    const bool fill_scans = nt_async;           // (the fill kernel sums the counts itself: no scan launch)
    if(!fill_scans) MMD_TRY(mmd_exclusive_scan(h, h->tile_of_block.p, nunits, nt_async ? nullptr : &nt));
    if(nt_async) nt = h->ntiles_hint + h->ntiles_hint / 32 + 64;
    const int* nt_dev = nt_async ? h->tile_of_block.p + nunits : nullptr;
    h->ntiles = nt;
    MMD_TRY(h->tile_block.ensure((size_t)nt + 2, false, h->stream));
    MMD_TRY(h->tile_first.ensure((size_t)nt + 2, false, h->stream));
    MMD_TRY(h->tile_max.ensure((size_t)nt + 2, false, h->stream));
    MMD_TRY(h->tile_ncand.ensure((size_t)nt + 2, false, h->stream));
    MMD_TRY(h->tile_cnt.ensure((size_t)nt + 2, false, h->stream));
    MMD_TRY(h->tile_ghost.ensure((size_t)nt + 2, false, h->stream));
    MMD_TRY(h->tile_rowmax.ensure((size_t)nt + 2, false, h->stream));
    MMD_TRY(h->tile_rowsum.ensure((size_t)nt + 2, false, h->stream));
    // rows in two parts (core first, rest of the skin behind): full lists of a force style that asked for it (CoreRows, mmd_internal.hpp)
    h->core.radius = 0; h->core.margin = 0;
    if(h->opt_core_pct > 0 && h->style == 1 && h->eam_uniform && h->nprocs == 1 && !h->opt_force_transport && !h->h_cutforcesq.empty()) {
      // EAM decks carry a generous skin (1 A for a solid whose atoms move ~0.15 A between re-neighborings): the core part ends
      // opt_core_pct per cent into it
      const double cutforce = sqrt((double)h->h_cutforcesq[0]);
      h->core.margin = (real)((double)h->opt_core_pct * 0.01 * ((double)h->cutneigh - cutforce));
      if(h->core.margin > 0) h->core.radius = (real)(cutforce + (double)h->core.margin);
    }
    const bool core_rows = !h->halfneigh && h->core.radius > 0;
    float core_thr = 0;
    if(core_rows) {
      // classification threshold in the frame of the build's float pre-test (positions relative to the tile corner): 2^-19 of
      // relative margin covers its rounding, so every pair closer than the core radius is classified core
      core_thr = (float)((double)h->core.radius * h->core.radius) * (1.0f + 1.9e-6f) + 1.0e-6f;
      MMD_TRY(h->tile_kcore.ensure((size_t)nt + 2, false, h->stream));
      MMD_TRY(h->xbuild.ensure((size_t)h->nmax + 2, false, h->stream));
      if(!h->core_words.p) {
        MMD_TRY(h->core_words.ensure(192, false, h->stream));
        HIP_TRY(hipMemsetAsync(h->core_words.p, 0, 192 * sizeof(unsigned), h->stream));
      }
    }
    h->core.rows_built = false;
    // (the hit words of a tile wait in registers; only the two-list core/rest rows of EAM still park them in this scratch)
    MMD_TRY(h->tile_words.ensure((NB2_REGPARK && !core_rows) ? (size_t)64 : (size_t)nt * NB2_NG * 64 * 2 + 64, false, h->stream));
    if(h->halfneigh) MMD_TRY(h->tile_self.ensure((size_t)nt * 64 + 64, false, h->stream));
    h->tile_cstride = NB_CHUNKS * 64 + 64;      // + room for the closing dummy entry, rows stay 256-byte aligned
    MMD_TRY(h->tile_cand.ensure((size_t)nt * h->tile_cstride + 64, false, h->stream));
    // one rank, full lists (LJ, EAM with one table set): the candidate lists once more with every ghost named by its owner + image code (the tile kernels can
    // then stage ghosts from their owners' current positions without a look-up: GhostResolve, tile_lds.hpp)
    int* cand_src_p = nullptr;
    h->cand_src_ready = false;
    if(h->opt_ghost_resolve && (h->style == 0 || (h->eam_uniform && !h->halfneigh)) && h->nprocs == 1 && !h->opt_force_transport && (h->nghost_dev != nullptr || h->ghost_chain_ok) &&
       !h->ghosts_uploaded && nlocal + h->nghost < (1 << MMD_SRC_BITS) && h->ghost_root.p != nullptr) {
      MMD_TRY(h->tile_cand_src.ensure((size_t)nt * h->tile_cstride + 64, false, h->stream));
      cand_src_p = h->tile_cand_src.p;
    }
    // several ranks, LJ over full lists: the same second list with every ghost named by the entry of the position buffer its per-step halo message delivers it to
    // (DirectHalo::gmap, written by the direct borders): the step's force kernel needs no k_dh_unpack in front of it
    const int* src_root = (const int*)h->ghost_root.p;
    const int* src_image = (const int*)h->ghost_image.p;
    h->cand_src_halo = false;
    if(cand_src_p == nullptr && h->dh.gmap_live && h->dh.opt_recv == 3 && h->opt_ghost_resolve && h->style == 0 && !h->halfneigh && !h->ghosts_uploaded) {
      MMD_TRY(h->tile_cand_src.ensure((size_t)nt * h->tile_cstride + 64, false, h->stream));
      cand_src_p = h->tile_cand_src.p;
      src_root = (const int*)h->dh.gmap.p; src_image = nullptr;
      h->cand_src_halo = true;
    }
    if(fill_scans)
      hipLaunchKernelGGL(k_pencil_fill_scan, dim3(div_up(nunits, 256)), dim3(256), 0, h->stream, h->pencil_range.p, nunits, g.nblk[0], h->tile_of_block.p, h->tile_block.p, h->tile_first.p,
                         h->tile_cnt.p, h->d_flags, nt, h->x.p, nlocal, h->nghost, h->nghost_dev, lohi ? (const unsigned*)h->pencil_lohi.p : (const unsigned*)nullptr, (const int*)h->bin_start.p);
    else
      hipLaunchKernelGGL(k_pencil_fill, dim3(div_up(nunits, 256)), dim3(256), 0, h->stream, h->pencil_range.p, nunits, g.nblk[0], h->tile_of_block.p, h->tile_block.p, h->tile_first.p,
                         h->tile_cnt.p, h->d_flags, nt, h->x.p, nlocal, h->nghost, h->nghost_dev);
    HIP_TRY(hipGetLastError());
    bool order_here = false, reduce_in_publish = false;
    for(int attempt = 0; attempt < 8 && want_tiles; attempt++) {
      MMD_TRY(h->nl16.ensure((size_t)h->ntiles * h->maxneighs * 64 + 16 * 64, false, h->stream));   // (+ prefetch overrun of the last tile)
      if(attempt > 0) HIP_TRY(hipMemsetAsync(h->d_flags, 0, 8 * sizeof(int), h->stream));     // (a relaunch with longer rows: k_pencil_fill's zeroes are used up; the first launch needs no zeroing)
      const int tmode = !h->halfneigh ? 0 : (h->ghost_newton ? 2 : 1);
#define LAUNCH_ROWS(M) LAUNCH_ROWS2(M, 0)
#define LAUNCH_ROWS2(M, CR)                                                                                                             \
  hipLaunchKernelGGL((k_build_rows<M, CR>), dim3(xcd_grid(h->ntiles)), dim3(64), 0, h->stream, h->x.p, h->binned.p, h->bin_start.p,     \
                     src_image, g, h->ntiles, nlocal, nlocal + h->nghost, h->cutneigh, h->cutneighsq, h->maxneighs, h->tile_cstride, \
                     h->tile_block.p, h->tile_first.p, h->tile_cnt.p, h->numneigh.p, h->nl16.p, h->tile_cand.p, h->tile_ncand.p,         \
                     h->tile_max.p, h->tile_ghost.p, h->tile_self.p, h->tile_rowmax.p, h->tile_rowsum.p, h->tile_words.p, h->d_flags, h->opt_ablate, nt_dev, h->nghost_dev, \
                     core_thr, h->xbuild.p, h->tile_kcore.p, cand_src_p, src_root, h->style == 1 ? 1 : 0)
      {
        if(tmode == 0 && core_rows) LAUNCH_ROWS2(0, 1); else if(tmode == 0) LAUNCH_ROWS(0); else if(tmode == 1) LAUNCH_ROWS(1); else LAUNCH_ROWS(2);
        reduce_in_publish = h->in_run && h->ntiles <= 4096;
        if(!reduce_in_publish)
        hipLaunchKernelGGL(k_tile_reduce, dim3(std::min(32, std::max(1, div_up(h->ntiles, 1024)))), dim3(1024), 0, h->stream, h->tile_rowmax.p, h->tile_rowsum.p, h->tile_ncand.p, h->ntiles,
                           h->d_flags, (unsigned long long*)(h->d_flags + 4), nt_dev, h->nghost_dev ? (const int*)h->bstate.p : (const int*)nullptr);
        order_here = (h->opt_overlap > 0 || (h->opt_overlap < 0 && h->overlap_choice != 0)) && (h->nprocs > 1 || h->opt_force_transport || h->opt_overlap >= 2) && h->ntiles > 0;
        if(order_here) {                  // several ranks: the interior-first order of the halo overlap, no extra host synchronisation
          const int nb_o = div_up(h->ntiles, 1024);
          MMD_TRY(h->tile_order.ensure((size_t)h->ntiles + 8, false, h->stream));
          MMD_TRY(h->flag_tmp.ensure((size_t)nb_o + 8, false, h->stream));
          hipLaunchKernelGGL(k_tile_order_count, dim3(nb_o), dim3(256), 0, h->stream, h->tile_ghost.p, h->ntiles, nt_dev, h->flag_tmp.p);
          hipLaunchKernelGGL(k_tile_order_scatter, dim3(nb_o), dim3(256), 0, h->stream, h->tile_ghost.p, h->ntiles, nt_dev, h->flag_tmp.p, h->tile_order.p, h->d_flags);
        }
      }
#undef LAUNCH_ROWS
#undef LAUNCH_ROWS2
      HIP_TRY(hipGetLastError());
      // [0..3] results, [4..5] total, [6] tile count, [12] long-bin flag, [16..55] bst of a deferred one-rank borders
      int spec_cmax_used = 0, spec_calls_before = 0;
      size_t spec_ev_before = 0;
      long long spec_ctr_before = 0;
      int verdict = 0;                    // 1: this step's Force::compute was launched behind the build and the build's verdict let it run
      bool spec_go = false;
      if(h->in_run) {
        // inside Integrate::run the build's results are PUBLISHED into pinned host memory by a one-wavefront kernel and the host polls that
        // memory: a blocking stream wait costs the wake-up of a sleeping thread (~40 us between the copy and the next force kernel)
        // Force::compute of this step goes onto the stream BEFORE the host has the words (SpecLaunch, mmd_internal.hpp): sized for the
        // tile capacity of this build and the previous build's largest union (+ opt_spec slots), gated on the device by the verdict the
        // publishing kernel forms. One attempt per re-neighboring: after a "no" the lists are rebuilt and the step loop launches as usual.
        std::function<int()> fn;
        fn.swap(h->spec_fn);
        BuildVerdict V{};
        const int spec_cmax = h->tile_cmax + std::max(h->opt_spec, 0);
        spec_cmax_used = spec_cmax;
        bool spec = h->opt_spec > 0 && (bool)fn && attempt == 0 && tmode == 0 && nt_async && !order_here && h->tile_cmax > 0 && h->ntiles > 0;
        const int save_cmax = h->tile_cmax;
        if(spec) {
          h->tile_cmax = spec_cmax; h->tiles_ready = true; h->neigh_nlocal = nlocal;
          spec = mmd_lj_tiles_available(h) != 0;
          if(!spec) { h->tile_cmax = save_cmax; h->tiles_ready = false; h->neigh_nlocal = 0; }
        }
        if(spec) V = BuildVerdict{1, h->maxneighs, h->ntiles, 1, spec_cmax, h->nghost_dev != nullptr ? 1 : 0, h->bf_est_nb, h->big_bins ? 1 : 0, core_rows ? 1 : 0};
        TileReduceArgs R{};
        if(reduce_in_publish) R = TileReduceArgs{1, h->tile_rowmax.p, h->tile_rowsum.p, h->tile_ncand.p, h->ntiles, nt_dev, h->nghost_dev ? (const int*)h->bstate.p : (const int*)nullptr};
        MMD_TRY(flags_publish(h, 62, V, R));
        if(spec) {
          h->spec = SpecLaunch{h->d_flags + NB_GATE_WORD, nt_dev, h->nghost_dev, nullptr};
          h->spec_fused = false;
          const long long before = h->spec_launches;
          spec_calls_before = h->force_calls; spec_ctr_before = h->force_sample_ctr; spec_ev_before = h->ev_used;
          const int rc = fn();
          h->spec = SpecLaunch{nullptr, nullptr, nullptr, nullptr};
          h->tile_cmax = save_cmax; h->tiles_ready = false; h->neigh_nlocal = 0;
          if(rc < 0) return rc;
          if(h->spec_launches != before + 1) { mmd_set_error("neighbor build: the force launch behind the build did not take the gated tile path"); return -1; }
          spec_go = true;
          h->spec_runs++;
        }
        MMD_TRY(flags_wait(h));
        h->clk_written |= 4;
        if(spec_go) {
          verdict = h->h_flags[NB_GATE_WORD];
          // a "no": the gated launch did nothing and the step loop launches this step's Force::compute again — the call counter of the kernel clock and
          // the event pair the no-op may have carried (a 0 ms sample) go back to where they were
          if(!verdict) { h->spec_fails++; h->spec_clk_redo = h->fclk_n > 0; h->force_calls = spec_calls_before; h->force_sample_ctr = spec_ctr_before; h->ev_used = spec_ev_before; }
        }
      } else {
      HIP_TRY(hipMemcpyAsync(h->h_flags, h->d_flags, (h->nghost_dev ? 56 : 16) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(mmd_stream_sync(h));
      }
      memcpy(h->h_result, h->h_flags + 4, sizeof(double));
      // (a verdict of 1 means the gated force kernel is running on these lists: every rule below that sends the build around again is part
      //  of the verdict, so none of them can fire then — if one does, the two have come apart and the run must not go on)
#define NB_REDO_GUARD() if(verdict) { mmd_set_error("neighbor build: verdict of the device and rules of the host disagree"); return -1; }
      if(h->nghost_dev) {                   // the ghost counts of the deferred one-rank borders arrived with the flags
        memcpy(h->h_flags_big, h->h_flags + 16, 40 * sizeof(int));
        const int rc = mmd_borders_deferred_finish(h);
        if(rc < 0) return rc;
        if(rc == 0) { NB_REDO_GUARD(); return mmd_neighbor_build(h); }       // (estimates too small: borders were redone swap by swap; build again)
      }
      if(nt_async) {
        if(h->h_flags[6] > h->ntiles) { NB_REDO_GUARD(); h->ntiles_hint = 0; return mmd_neighbor_build(h); }     // more tiles than provided for: size from the count
        h->ntiles = h->h_flags[6];
      }
      if(h->h_flags[12] && !h->big_bins) {                       // a bin longer than NB_BIGBIN showed up: bin again with the rank sort on
        NB_REDO_GUARD();
        h->big_bins = true;
        return mmd_neighbor_build(h);
      }
      if(h->h_flags[3]) { NB_REDO_GUARD(); want_tiles = false; break; }           // a block has too many candidates: global-row build below
      const int maxn = h->h_flags[0];
      h->max_row = maxn;
      const int need = core_rows ? std::max(maxn, h->h_flags[7]) : maxn;      // (two padded parts per row need a little more room)
      if(maxn >= h->maxneighs || need > h->maxneighs) {            // ref/neighbor.cpp:186-208
        NB_REDO_GUARD();
        int m = (int)(std::max(maxn, need) * 1.2);
        h->maxneighs = (m + MMD_UNROLL - 1) / MMD_UNROLL * MMD_UNROLL;
        continue;
      }
      if(verdict && h->h_flags[2] > spec_cmax_used) { mmd_set_error("neighbor build: verdict of the device and rules of the host disagree (union)"); return -1; }
#undef NB_REDO_GUARD
      h->core.rows_built = core_rows;
      unsigned long long tot;
      memcpy(&tot, h->h_result, sizeof(tot));
      h->total_neigh = (long long)tot;
      h->tile_cmax = h->h_flags[2];
      h->tiles_ready = true;
      h->ntiles_hint = h->ntiles;
      h->neigh_nlocal = nlocal;
      h->ntiles_interior = order_here ? h->h_flags[13] : -1;    // (-1: the interior/boundary order is derived on demand, mmd_order_tiles)
      h->cand_src_ready = cand_src_p != nullptr && (h->ghost_chain_ok || h->cand_src_halo);
      h->spec_done = verdict != 0;
      if(verdict && h->spec_fused)           // the gated kernel wrote the dummy atom of the second position buffer behind the last ghost
        for(int k = 0; k < 2; k++) if(h->xalt_dummy_ptr[k] == (const void*)h->x_alt.p) h->xalt_dummy_slot[k] = nlocal + h->nghost;
      return 0;
    }
    if(want_tiles) { mmd_set_error("mmd_neighbor_build: neighbor rows keep overflowing (maxneighs=%d)", h->maxneighs); return -1; }
  }
  // ---- global 32-bit rows (half lists, or blocks with too many candidates for the tile form)
  MMD_TRY(h->wave_max.ensure((size_t)nwaves + 1, false, h->stream));
  for(int attempt = 0; attempt < 8; attempt++) {
    MMD_TRY(h->neigh.ensure((size_t)nwaves * h->maxneighs * 64 + 64, false, h->stream));
    HIP_TRY(hipMemsetAsync(h->d_flags, 0, 8 * sizeof(int), h->stream));
    const int mode = !h->halfneigh ? 0 : (h->ghost_newton ? 2 : 1);
#define LAUNCH_BUILD(M)                                                                                              \
  hipLaunchKernelGGL(k_build<M>, dim3(xcd_grid(nblocks)), dim3(64), 0, h->stream, h->x.p, h->binned.p, h->bin_start.p,        \
                     h->ghost_image.p, g, nlocal, h->cutneighsq, h->maxneighs, h->neigh.p, h->numneigh.p, h->d_flags, h->bg_ref)
    if(nlocal) {
      if(mode == 0) LAUNCH_BUILD(0);
      else if(mode == 1) LAUNCH_BUILD(1);
      else LAUNCH_BUILD(2);
    }
#undef LAUNCH_BUILD
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->h_flags, h->d_flags, 16 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    if(h->h_flags[12] && !h->big_bins) { h->big_bins = true; return mmd_neighbor_build(h); }
    const int maxn = h->h_flags[0];
    h->max_row = maxn;
    if(maxn >= h->maxneighs) {                      // ref/neighbor.cpp:186-208
      int m = (int)(maxn * 1.2);
      m = (m + MMD_UNROLL - 1) / MMD_UNROLL * MMD_UNROLL;
      h->maxneighs = m;
      continue;
    }
    MMD_TRY(finish_rows(h, true));
    h->neigh_nlocal = nlocal;
    (void)nall;
    return 0;
  }
  mmd_set_error("mmd_neighbor_build: neighbor rows keep overflowing (maxneighs=%d)", h->maxneighs);
  return -1;
}

extern "C" int mmd_neighbor_geometry(mmd_handle* h, int mbin[3], int mbinlo[3], int nblk[3], int reach[3])
{
  if(!h || !h->neigh_ready) { mmd_set_error("mmd_neighbor_geometry: call mmd_neighbor_setup first"); return -1; }
  for(int d = 0; d < 3; d++) {
    // one self-consistent grid: the REFERENCE's bins and the blocks / reach that belong to them. (When `-b` asks for bins finer than the
    // build kernels' reach the device bins coarser, mmd_neighbor_setup; mmd_get_counter "device_bins_coarser" says so — lists do not depend on it.)
    if(mbin) mbin[d] = h->bg_ref.mbin[d];
    if(mbinlo) mbinlo[d] = h->bg_ref.mbinlo[d];
    if(nblk) nblk[d] = h->bg_ref.nblk[d];
    if(reach) reach[d] = h->bg_ref.reach[d];
  }
  return 0;
}

extern "C" int mmd_neighbor_info(mmd_handle* h, int* maxneighs, int* mbins, long long* total_neigh, int* max_row)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(maxneighs) *maxneighs = h->maxneighs;
  if(mbins) *mbins = h->bg.mbins;
  if(total_neigh) *total_neigh = h->total_neigh;
  if(max_row) *max_row = h->max_row;
  return 0;
}

// diagnostics of the tile form: {ntiles, max candidates, sum candidates, sum padded rows, sum atoms, longest padded row}
extern "C" int mmd_neighbor_tile_stats(mmd_handle* h, long long out[6])
{
  if(!h || !out) { mmd_set_error("mmd_neighbor_tile_stats: bad arguments"); return -1; }
  for(int q = 0; q < 6; q++) out[q] = 0;
  if(!h->tiles_ready || h->ntiles <= 0) return 0;
  HIP_TRY(hipSetDevice(h->device));
  std::vector<int> nc(h->ntiles), mx(h->ntiles), ct(h->ntiles);
  HIP_TRY(mmd_stream_sync(h));
  HIP_TRY(hipMemcpy(nc.data(), h->tile_ncand.p, sizeof(int) * h->ntiles, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(mx.data(), h->tile_max.p, sizeof(int) * h->ntiles, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ct.data(), h->tile_cnt.p, sizeof(int) * h->ntiles, hipMemcpyDeviceToHost));
  out[0] = h->ntiles;
  for(int t = 0; t < h->ntiles; t++) {
    if(nc[t] > out[1]) out[1] = nc[t];
    out[2] += nc[t]; out[3] += mx[t]; out[4] += ct[t];
    if(mx[t] > out[5]) out[5] = mx[t];
  }
  return 0;
}

// diagnostics: the raw tile form of ONE tile — its padded rows (16-bit LDS offsets of the candidates' records, [k][64 lanes], kmax rows),
// the atoms of its lanes, its candidate union (tools/lds_conflicts.py prices the LDS bank conflicts of the force kernels' gathers with it)
extern "C" int mmd_neighbor_tile_rows(mmd_handle* h, int tile, unsigned short* rows, int rows_cap, int* kmax, int* atoms64, int* cand, int cand_cap, int* ncand)
{
  if(!h || !rows || !kmax || !atoms64 || !cand || !ncand) { mmd_set_error("mmd_neighbor_tile_rows: bad arguments"); return -1; }
  if(!h->tiles_ready || tile < 0 || tile >= h->ntiles) { mmd_set_error("mmd_neighbor_tile_rows: no such tile"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(mmd_stream_sync(h));
  int km = 0, nc = 0, first = 0, cnt = 0;
  HIP_TRY(hipMemcpy(&km, h->tile_max.p + tile, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&nc, h->tile_ncand.p + tile, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&first, h->tile_first.p + tile, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&cnt, h->tile_cnt.p + tile, sizeof(int), hipMemcpyDeviceToHost));
  if(km * 64 > rows_cap || nc > cand_cap) { mmd_set_error("mmd_neighbor_tile_rows: buffers too small (%d rows, %d candidates)", km, nc); return -1; }
  if(km) HIP_TRY(hipMemcpy(rows, h->nl16.p + (size_t)tile * h->maxneighs * 64, sizeof(unsigned short) * 64 * km, hipMemcpyDeviceToHost));
  if(nc) HIP_TRY(hipMemcpy(cand, h->tile_cand.p + (size_t)tile * h->tile_cstride, sizeof(int) * nc, hipMemcpyDeviceToHost));
  for(int l = 0; l < 64; l++) atoms64[l] = -1;
  if(cnt) HIP_TRY(hipMemcpy(atoms64, h->binned.p + first, sizeof(int) * std::min(cnt, 64), hipMemcpyDeviceToHost));
  *kmax = km; *ncand = nc;
  return 0;
}

// diagnostics: histograms (bins of `width`, `nb` bins, the last one open-ended) of the tiles' union sizes and padded row counts
extern "C" int mmd_neighbor_tile_histogram(mmd_handle* h, int nb, int width, long long* hist_ncand, long long* hist_rows)
{
  if(!h || nb < 1 || width < 1 || !hist_ncand || !hist_rows) { mmd_set_error("mmd_neighbor_tile_histogram: bad arguments"); return -1; }
  for(int q = 0; q < nb; q++) hist_ncand[q] = hist_rows[q] = 0;
  if(!h->tiles_ready || h->ntiles <= 0) return 0;
  HIP_TRY(hipSetDevice(h->device));
  std::vector<int> nc(h->ntiles), mx(h->ntiles);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(nc.data(), h->tile_ncand.p, sizeof(int) * h->ntiles, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(mx.data(), h->tile_max.p, sizeof(int) * h->ntiles, hipMemcpyDeviceToHost));
  for(int t = 0; t < h->ntiles; t++) {
    hist_ncand[std::min(nc[t] / width, nb - 1)]++;
    hist_rows[std::min(mx[t] / width, nb - 1)]++;
  }
  return 0;
}

// ---- layout conversion to/from the reference's row-major rows (ref/neighbor.cpp:128) ----------------
__global__ void k_rows_to_ref(const int* __restrict__ neigh, const int* __restrict__ numneigh, int nlocal, int stride_dev,
                              int* __restrict__ out, int stride_ref)
{
  const int i = blockIdx.x;
  if(i >= nlocal) return;
  const int n = min(numneigh[i], stride_ref);
  const size_t rowbase = ((size_t)(i >> 6) * stride_dev) * 64 + (i & 63);
  for(int k = threadIdx.x; k < n; k += blockDim.x) out[(size_t)i * stride_ref + k] = neigh[rowbase + (size_t)k * 64];
}
// half lists: the device-built rows partition the pairs by the (z,y,x) order of the two positions (k_build_rows); the
// reference keeps an owned-owned pair on the atom with the SMALLER index (ref/neighbor.cpp:171: j <= i skipped). Re-home those
// entries on the way out, so a downloaded list is the reference's list (rows as sets). Ghost partners stay where they are
// (same rule in both). cnt[] must be zero; out == nullptr: counts only.
__global__ void k_rows_to_ref_half(const int* __restrict__ neigh, const int* __restrict__ numneigh, int nlocal, int stride_dev,
                                   int* __restrict__ out, int stride_ref, int* __restrict__ cnt)
{
  const int i = blockIdx.x;
  if(i >= nlocal) return;
  const int n = numneigh[i];
  const size_t rowbase = ((size_t)(i >> 6) * stride_dev) * 64 + (i & 63);
  for(int k = threadIdx.x; k < n; k += blockDim.x) {
    const int j = neigh[rowbase + (size_t)k * 64];
    const int home = (j >= nlocal || j > i) ? i : j;
    const int pos = atomicAdd(&cnt[home], 1);
    if(out && pos < stride_ref) out[(size_t)home * stride_ref + pos] = home == i ? j : i;
  }
}
__global__ void k_rows_from_ref(const int* __restrict__ in, const int* __restrict__ numneigh, int nlocal, int stride_ref,
                                int* __restrict__ neigh, int stride_dev)
{
  const int i = blockIdx.x;
  if(i >= nlocal) return;
  const int n = numneigh[i];
  const size_t rowbase = ((size_t)(i >> 6) * stride_dev) * 64 + (i & 63);
  for(int k = threadIdx.x; k < n; k += blockDim.x) neigh[rowbase + (size_t)k * 64] = in[(size_t)i * stride_ref + k];
}

extern "C" int mmd_neighbor_download(mmd_handle* h, int* neighbors, int maxneighs, int* numneigh)
{
  if(!h || h->neigh_nlocal != h->nlocal) { mmd_set_error("mmd_neighbor_download: no neighbor list for the current atoms"); return -1; }
  const int n = h->nlocal;
  if(h->halfneigh && h->ghost_newton && n && h->tiles_ready && !h->rows_uploaded) {
    // (valid directly after mmd_neighbor_build: the rows are re-derived from the CURRENT positions and the bins of the last build; ghosts that
    //  a run left one step behind their owners are brought up to date first)
    if(h->ghosts_stale) { MMD_TRY(mmd_ghosts_refresh(h)); h->ghosts_stale = false; }
    // half lists with ghost newton: the device list partitions the pairs by the (z,y,x) order of the two positions, the reference by
    // its half stencil of bins + the same-bin rules (ref/neighbor.cpp:143-182, :424-441). What crosses the boundary is the REFERENCE's
    // list: rebuilt here from the same binned atoms with the reference's rule (k_build<3>), rows equal the oracle's as sets.
    const BinGeom& g = h->bg;
    if(h->bg_ref.mbin[0] > 1023 || h->bg_ref.mbin[1] > 1023 || h->bg_ref.mbin[2] > 1023) { mmd_set_error("mmd_neighbor_download: more than 1023 bins per dimension"); return -1; }
    const int nwaves = div_up(n, 64), nblocks = g.nblk[0] * g.nblk[1] * g.nblk[2];
    DevArr<int> rows, cnt, tmp;
    MMD_TRY(cnt.ensure((size_t)n + 64, false, h->stream));
    int stride = 2 * h->maxneighs;
    for(int attempt = 0; ; attempt++) {
      MMD_TRY(rows.ensure((size_t)nwaves * stride * 64 + 64, false, h->stream));
      HIP_TRY(hipMemsetAsync(h->d_flags, 0, 8 * sizeof(int), h->stream));
      hipLaunchKernelGGL(k_build<3>, dim3(xcd_grid(nblocks)), dim3(64), 0, h->stream, h->x.p, h->binned.p, h->bin_start.p, h->ghost_image.p, g, n,
                         h->cutneighsq, stride, rows.p, cnt.p, h->d_flags, h->bg_ref);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(h->h_flags, h->d_flags, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(mmd_stream_sync(h));
      if(h->h_flags[0] < stride) break;
      if(attempt == 4) { mmd_set_error("mmd_neighbor_download: rows keep overflowing"); return -1; }
      stride = (int)(h->h_flags[0] * 1.2) + 8;
    }
    if(neighbors && h->h_flags[0] > maxneighs) { mmd_set_error("mmd_neighbor_download: a row holds %d entries, the stride is %d", h->h_flags[0], maxneighs); return -1; }
    if(numneigh) HIP_TRY(hipMemcpyAsync(numneigh, cnt.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if(neighbors) {
      MMD_TRY(tmp.ensure((size_t)n * maxneighs, false, h->stream));
      hipLaunchKernelGGL(k_rows_to_ref, dim3(n), dim3(64), 0, h->stream, rows.p, cnt.p, n, stride, tmp.p, maxneighs);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(neighbors, tmp.p, (size_t)n * maxneighs * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(mmd_stream_sync(h));
    rows.release(); cnt.release(); tmp.release();
    return 0;
  }
  MMD_TRY(mmd_ensure_rows(h));
  if(h->halfneigh && n) {
    DevArr<int> tmp, cnt;
    MMD_TRY(cnt.ensure((size_t)n + 1, false, h->stream));
    if(neighbors) MMD_TRY(tmp.ensure((size_t)n * maxneighs, false, h->stream));
    HIP_TRY(hipMemsetAsync(cnt.p, 0, (size_t)n * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_rows_to_ref_half, dim3(n), dim3(64), 0, h->stream, h->neigh.p, h->numneigh.p, n, h->maxneighs,
                       neighbors ? tmp.p : (int*)nullptr, maxneighs, cnt.p);
    HIP_TRY(hipGetLastError());
    std::vector<int> hc((size_t)n);
    HIP_TRY(hipMemcpyAsync(hc.data(), cnt.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if(neighbors) HIP_TRY(hipMemcpyAsync(neighbors, tmp.p, (size_t)n * maxneighs * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    tmp.release(); cnt.release();
    int worst = 0;
    for(int i = 0; i < n; i++) worst = hc[i] > worst ? hc[i] : worst;
    if(neighbors && worst > maxneighs) { mmd_set_error("mmd_neighbor_download: a row holds %d entries, the stride is %d", worst, maxneighs); return -1; }
    if(numneigh) memcpy(numneigh, hc.data(), (size_t)n * sizeof(int));
    return 0;
  }
  if(numneigh) HIP_TRY(hipMemcpyAsync(numneigh, h->numneigh.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if(neighbors && n) {
    DevArr<int> tmp;
    MMD_TRY(tmp.ensure((size_t)n * maxneighs, false, h->stream));
    hipLaunchKernelGGL(k_rows_to_ref, dim3(n), dim3(64), 0, h->stream, h->neigh.p, h->numneigh.p, n, h->maxneighs, tmp.p, maxneighs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(neighbors, tmp.p, (size_t)n * maxneighs * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    tmp.release();
  }
  HIP_TRY(mmd_stream_sync(h));
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Uploaded rows -> tile form (the inverse of k_tiles_to_rows): a list built elsewhere — the reference's own Neighbor::build handed
// over by mmd_neighbor_upload — is served by the tile force kernels like a list built here. The atoms are binned and cut into pencil
// tiles exactly as for a build; one wavefront per tile then
//   1. lays out the tile's candidate runs (the x-stretches of the surrounding pencils, as k_build_rows does) and marks, in an LDS
//      bitmap over those positions, every atom some row of the tile refers to (binned_inv: atom index -> position in binned[]);
//   2. numbers the marked positions in order (= the tile's union, written to tile_cand as atom indices);
//   3. rewrites every row entry as the 16-bit LDS offset of its union slot, k-major, padded with the dummy slot.
// An entry that lies outside the runs (a list older than the positions, a partner beyond the cutoff the bins were set up for), a
// union beyond the 16-bit offsets or runs longer than the bitmap raise flags[3]: the rows then stay on the general row kernels.
// ---------------------------------------------------------------------------------------------------
#define R2T_MAXC 16384          // candidate positions a tile's runs may span (bitmap: 2 KB of LDS)
__global__ void k_invert_binned(const int* __restrict__ binned, int n, int* __restrict__ inv)
{
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if(a < n) inv[binned[a]] = a;
}

template <int HALF>
__global__ __launch_bounds__(64) void k_rows_to_tiles(const real4* __restrict__ x, const int* __restrict__ binned, const int* __restrict__ bin_start,
                                                      const int* __restrict__ binned_inv, BinGeom g, int ntiles, int nlocal, int nall, real cutneigh,
                                                      int maxneighs, int cstride, const int* __restrict__ tile_block, const int* __restrict__ tile_first,
                                                      const int* __restrict__ tile_cnt, const int* __restrict__ numneigh, const int* __restrict__ rows,
                                                      unsigned short* __restrict__ nl16, int* __restrict__ tile_cand, int* __restrict__ tile_ncand,
                                                      int* __restrict__ tile_max, int* __restrict__ tile_ghost, unsigned short* __restrict__ tile_self,
                                                      int* __restrict__ tile_rowmax, int* __restrict__ tile_rowsum, int* __restrict__ flags)
{
  __shared__ int rng_start[NB_MAX_ROWS], rng_len[NB_MAX_ROWS], rng_off[NB_MAX_ROWS + 1];
  __shared__ unsigned s_bits[R2T_MAXC / 32];
  __shared__ int s_pre[R2T_MAXC / 32];
  __shared__ unsigned short s_self[64];
  const int lane = threadIdx.x;
  const int tile = xcd_work_item(ntiles);
  if(tile < 0) return;
  const int b = tile_block[tile];
  const int ta = tile_first[tile], tcn = tile_cnt[tile];
  const int by = (b / g.nblk[0]) % g.nblk[1], bz = b / (g.nblk[0] * g.nblk[1]);
  const int ii = lane < tcn ? binned[ta + lane] : -1;
  const bool owned = ii >= 0 && ii < nlocal;
  const real4 pme = x[ii >= 0 ? ii : 0];
  const unsigned long long own_mask = __builtin_amdgcn_ballot_w64(owned);
  const int ny = 2 * g.reach[1] + 1, nz = 2 * g.reach[2] + 1;
  const int nr = min(ny * nz, NB_MAX_ROWS);
  const unsigned kx = float_key((float)pme.x);
  const float bx0 = key_float(wave_min_u(owned ? kx : 0xffffffffu)), bx1 = key_float(wave_max_u(owned ? kx : 0u));
  {
    const real xlo = (real)bx0, xhi = (real)bx1;
    const real reach_x = cutneigh * (real)1.0005 + (real)1.0e-4 * g.binsize[0] + (real)1.0e-5 * (fabs((real)bx0) + fabs((real)bx1));
    const int fmaxx = 2 * NB_XF * g.nblk[0] - 1;
    const int f0 = min(max(fine_x_of(g, xlo - reach_x), 0), fmaxx), f1 = min(max(fine_x_of(g, xhi + reach_x), 0), fmaxx);
    if(lane < nr) {                                    // (NB_MAX_ROWS = 64: one lane per run)
      int len = 0, start = 0;
      const int z = bz + lane / ny - g.reach[2], y = by + lane % ny - g.reach[1];
      if(z >= 0 && z < g.nblk[2] && y >= 0 && y < g.nblk[1] && own_mask != 0ull) {
        const int row = (z * g.nblk[1] + y) * g.nblk[0] * NB_SUB;
        start = bin_start[row + 4 * f0];
        len = bin_start[row + 4 * f1 + 4] - start;
      }
      rng_start[lane] = start; rng_len[lane] = len;
    }
    const int lmine = lane < nr ? rng_len[lane] : 0;
    const int incl = wave_incl_scan(lmine);
    if(lane < nr) rng_off[lane] = incl - lmine;
    if(lane == 63) rng_off[NB_MAX_ROWS] = incl;
  }
  for(int w = lane; w < R2T_MAXC / 32; w += 64) s_bits[w] = 0u;
  s_self[lane] = (unsigned short)0xffff;
  __syncthreads();
  const int ctot = rng_off[NB_MAX_ROWS];
  unsigned short* __restrict__ rowp = nl16 + ((size_t)tile * maxneighs) * 64 + lane;
  const size_t cbase = (size_t)tile * cstride;
  if(own_mask == 0ull) {
    if(lane == 0) { tile_max[tile] = 0; tile_ncand[tile] = 0; tile_cand[cbase] = nall; tile_ghost[tile] = 0; tile_rowmax[tile] = 0; tile_rowsum[tile] = 0; }
    if(HALF) tile_self[(size_t)tile * 64 + lane] = (unsigned short)0xffff;
    return;
  }
  bool bad = ctot > R2T_MAXC;
  const int n = owned ? numneigh[ii] : 0;
  const int nmax_w = (int)wave_max_u((unsigned)n);
  const size_t rbase = owned ? ((size_t)(ii >> 6) * maxneighs) * 64 + (ii & 63) : 0;
  // ---- 1. mark; the candidate position of every entry is parked in its own nl16 cell for step 3
  if(!bad) {
    for(int k = 0; k < nmax_w && k < maxneighs; k++) {
      if(k < n) {
        const int j = rows[rbase + (size_t)k * 64];
        const int a = binned_inv[j];
        int c = -1;
        for(int r = 0; r < nr; r++) { const int d = a - rng_start[r]; if((unsigned)d < (unsigned)rng_len[r]) c = rng_off[r] + d; }
        if(c < 0) bad = true;
        else { atomicOr(&s_bits[c >> 5], 1u << (c & 31)); rowp[(unsigned)k * 64u] = (unsigned short)c; }
      }
    }
  }
  bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
  __syncthreads();
  // ---- 2. number the marked positions: lane l owns words [8 l, 8 l + 8)
  constexpr int WPL = R2T_MAXC / 32 / 64;
  int mine = 0;
#pragma unroll
  for(int q = 0; q < WPL; q++) mine += __popc(s_bits[lane * WPL + q]);
  const int incl = wave_incl_scan(mine);
  int run = incl - mine;
  const int S = __shfl(incl, 63, 64);
  bool any_ghost = false;
  if(!bad) {
#pragma unroll
    for(int q = 0; q < WPL; q++) {
      const int w = lane * WPL + q;
      unsigned word = s_bits[w];
      s_pre[w] = run;
      while(word) {
        const int bq = __builtin_ctz(word);
        word &= word - 1;
        const int c = w * 32 + bq;
        int a = 0;
        for(int r = 0; r < nr; r++) { const int d = c - rng_off[r]; if((unsigned)d < (unsigned)rng_len[r]) a = rng_start[r] + d; }
        const int atom = binned[a];
        if(run < cstride - 1) tile_cand[cbase + run] = atom;
        any_ghost = any_ghost || atom >= nlocal;
        if(HALF && (unsigned)(a - ta) < 64u) s_self[a - ta] = (unsigned short)run;
        run++;
      }
    }
  }
  any_ghost = __builtin_amdgcn_ballot_w64(any_ghost) != 0ull;
  __syncthreads();
  // ---- 3. rows as LDS offsets of the union slots, padded with the dummy slot (= S) to the tile's longest row (multiple of NB_ROW_PAD)
  const int kc = min((nmax_w + NB_ROW_PAD - 1) / NB_ROW_PAD * NB_ROW_PAD, maxneighs);
  const unsigned short dummy = (unsigned short)(S * NB_SLOT_BYTES);
  if(!bad) {
    for(int k = 0; k < kc; k++) {
      unsigned short v = dummy;
      if(k < n) {
        const int c = rowp[(unsigned)k * 64u];
        const int slot = s_pre[c >> 5] + __popc(s_bits[c >> 5] & ((1u << (c & 31)) - 1u));
        v = (unsigned short)(slot * NB_SLOT_BYTES);
      }
      rowp[(unsigned)k * 64u] = v;
    }
  }
  if(HALF) tile_self[(size_t)tile * 64 + lane] = owned ? s_self[lane] : (unsigned short)0xffff;
  const int tsum = wave_sum(n);
  if(lane == 0) {
    tile_max[tile] = bad ? 0 : kc;
    tile_ncand[tile] = bad ? 0 : S;
    tile_cand[cbase + (bad ? 0 : min(S, cstride - 1))] = nall;
    tile_ghost[tile] = any_ghost ? 1 : 0;
    tile_rowmax[tile] = nmax_w;
    tile_rowsum[tile] = tsum;
    if(bad || S > cstride - 2 || (S + 1) * NB_SLOT_BYTES > 65535 || nmax_w > maxneighs) atomicMax(&flags[3], 1);
  }
}

// tile form of the uploaded rows (h->neigh / numneigh, wave-interleaved): 1 = tiles ready, 0 = not applicable / a tile did not fit
// (the row kernels serve the list), < 0 error
static int tiles_from_rows(mmd_handle* h)
{
  h->tiles_ready = false;
  h->cand_src_ready = false;
  const int nlocal = h->nlocal, nall = h->nlocal + h->nghost;
  if(!h->opt_tiles || !h->neigh_ready || nlocal == 0) return 0;
  const BinGeom& g = h->bg;
  if((2 * g.reach[1] + 1) * (2 * g.reach[2] + 1) > NB_MAX_ROWS) return 0;
  MMD_TRY(mmd_bin_atoms(h, -1));
  int* inv = h->atom_rank.p;                              // (free again after the fill pass of the binning)
  hipLaunchKernelGGL(k_invert_binned, dim3(div_up(nall, 256)), dim3(256), 0, h->stream, h->binned.p, nall, inv);
  const int nblocks = g.nblk[0] * g.nblk[1] * g.nblk[2], nunits = g.nblk[1] * g.nblk[2];
  MMD_TRY(h->tile_of_block.ensure((size_t)nblocks + 2, false, h->stream));
  MMD_TRY(h->pencil_range.ensure((size_t)2 * nunits + 2, false, h->stream));
  hipLaunchKernelGGL(k_pencil_count, dim3(nunits), dim3(64), 0, h->stream, h->binned.p, h->bin_start.p, nunits, g.nblk[0], nlocal, h->tile_of_block.p, h->pencil_range.p);
  int nt = 0;
  MMD_TRY(mmd_exclusive_scan(h, h->tile_of_block.p, nunits, &nt));
  h->ntiles = nt;
  MMD_TRY(h->tile_block.ensure((size_t)nt + 2, false, h->stream));
  MMD_TRY(h->tile_first.ensure((size_t)nt + 2, false, h->stream));
  MMD_TRY(h->tile_max.ensure((size_t)nt + 2, false, h->stream));
  MMD_TRY(h->tile_ncand.ensure((size_t)nt + 2, false, h->stream));
  MMD_TRY(h->tile_cnt.ensure((size_t)nt + 2, false, h->stream));
  MMD_TRY(h->tile_ghost.ensure((size_t)nt + 2, false, h->stream));
  MMD_TRY(h->tile_rowmax.ensure((size_t)nt + 2, false, h->stream));
  MMD_TRY(h->tile_rowsum.ensure((size_t)nt + 2, false, h->stream));
  if(h->halfneigh) MMD_TRY(h->tile_self.ensure((size_t)nt * 64 + 64, false, h->stream));
  h->tile_cstride = NB_CHUNKS * 64 + 64;
  MMD_TRY(h->tile_cand.ensure((size_t)nt * h->tile_cstride + 64, false, h->stream));
  MMD_TRY(h->nl16.ensure((size_t)nt * h->maxneighs * 64 + 16 * 64, false, h->stream));
  h->core.rows_built = false;
  hipLaunchKernelGGL(k_pencil_fill, dim3(div_up(nunits, 256)), dim3(256), 0, h->stream, h->pencil_range.p, nunits, g.nblk[0], h->tile_of_block.p, h->tile_block.p, h->tile_first.p,
                     h->tile_cnt.p, h->d_flags, nt, h->x.p, nlocal, h->nghost, (const int*)nullptr);
  if(nt) {
#define R2T(HF) hipLaunchKernelGGL((k_rows_to_tiles<HF>), dim3(xcd_grid(nt)), dim3(64), 0, h->stream, h->x.p, h->binned.p, h->bin_start.p, (const int*)inv, g, nt, nlocal, \
                                   nall, h->cutneigh, h->maxneighs, h->tile_cstride, h->tile_block.p, h->tile_first.p, h->tile_cnt.p, h->numneigh.p, h->neigh.p,  \
                                   h->nl16.p, h->tile_cand.p, h->tile_ncand.p, h->tile_max.p, h->tile_ghost.p, h->tile_self.p, h->tile_rowmax.p,                  \
                                   h->tile_rowsum.p, h->d_flags)
    if(h->halfneigh) R2T(1); else R2T(0);
#undef R2T
    hipLaunchKernelGGL(k_tile_reduce, dim3(std::min(32, std::max(1, div_up(nt, 1024)))), dim3(1024), 0, h->stream, h->tile_rowmax.p, h->tile_rowsum.p, h->tile_ncand.p, nt,
                       h->d_flags, (unsigned long long*)(h->d_flags + 4), (const int*)nullptr, (const int*)nullptr);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h->h_flags, h->d_flags, 16 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  if(h->h_flags[12] && !h->big_bins) { h->big_bins = true; return tiles_from_rows(h); }       // (a bin longer than NB_BIGBIN: bin again with the rank sort on)
  if(h->h_flags[3] || nt == 0) return 0;
  h->tile_cmax = h->h_flags[2];
  h->ntiles_interior = -1;
  h->tiles_ready = true;
  return 1;
}

extern "C" int mmd_neighbor_upload(mmd_handle* h, const int* neighbors, int maxneighs, const int* numneigh, int nlocal)
{
  if(!h || !neighbors || !numneigh || nlocal != h->nlocal) { mmd_set_error("mmd_neighbor_upload: bad arguments (nlocal mismatch?)"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  const int nwaves = div_up(nlocal, 64);
  int maxn = 0;
  for(int i = 0; i < nlocal; i++) maxn = numneigh[i] > maxn ? numneigh[i] : maxn;
  if(maxn > maxneighs) { mmd_set_error("mmd_neighbor_upload: a row is longer than the stride"); return -1; }
  int m = (maxn + MMD_UNROLL) / MMD_UNROLL * MMD_UNROLL;
  if(m > h->maxneighs) h->maxneighs = m;
  MMD_TRY(h->numneigh.ensure((size_t)nlocal + 64, false, h->stream));
  MMD_TRY(h->wave_max.ensure((size_t)nwaves + 1, false, h->stream));
  MMD_TRY(h->neigh.ensure((size_t)nwaves * h->maxneighs * 64 + 64, false, h->stream));
  DevArr<int> tmp;
  MMD_TRY(tmp.ensure((size_t)nlocal * maxneighs + 1, false, h->stream));
  HIP_TRY(hipMemcpyAsync(tmp.p, neighbors, (size_t)nlocal * maxneighs * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->numneigh.p, numneigh, (size_t)nlocal * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if(nlocal) {
    hipLaunchKernelGGL(k_rows_from_ref, dim3(nlocal), dim3(64), 0, h->stream, tmp.p, h->numneigh.p, nlocal, maxneighs, h->neigh.p, h->maxneighs);
    HIP_TRY(hipMemsetAsync(h->d_result, 0, sizeof(double), h->stream));
    hipLaunchKernelGGL(k_pad_rows, dim3(div_up(nwaves * 64, 256)), dim3(256), 0, h->stream, nlocal, nwaves, h->maxneighs,
                       h->nlocal + h->nghost, h->neigh.p, h->numneigh.p, h->wave_max.p, (unsigned long long*)h->d_result);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(mmd_stream_sync(h));
  tmp.release();
  h->max_row = maxn;
  h->rows_ready = true;
  h->rows_uploaded = true;
  h->neigh_nlocal = 0;
  // the tile form of these rows, where they fit it: the list is then served by the tile force kernels like one built here
  const int rt = tiles_from_rows(h);
  if(rt < 0) return rt;
  h->neigh_nlocal = nlocal;
  return 0;
}
