// minimd_amd/csrc/atom.hip — device-resident Atom (ref/atom.h:47-106): upload/download in the reference's
// AoS-3 layout, PBC wrap, spatial sort.
#include "device_utils.hpp"
#include "mmd_internal.hpp"

// ---- layout conversion: ref AoS stride 3 (+ int type)  <->  real4 {x,y,z,(real)type} -----------------
__global__ void k_pack_x4(const real* __restrict__ x3, const int* __restrict__ type, real4* __restrict__ x4, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  x4[i] = real4{x3[3 * i + 0], x3[3 * i + 1], x3[3 * i + 2], (real)type[i]};
}
__global__ void k_unpack_x4(const real4* __restrict__ x4, real* __restrict__ x3, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  const real4 p = x4[i];
  x3[3 * i + 0] = p.x; x3[3 * i + 1] = p.y; x3[3 * i + 2] = p.z;
}
__global__ void k_iota(int* __restrict__ a, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i < n) a[i] = i;
}

extern "C" int mmd_atom_set_box(mmd_handle* h, const mmd_float prd[3], const mmd_float lo[3], const mmd_float hi[3])
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  for(int d = 0; d < 3; d++) { h->prd[d] = prd[d]; h->lo[d] = lo[d]; h->hi[d] = hi[d]; }
  h->box_dev_valid = false;
  return 0;
}

extern "C" int mmd_atom_get_box(mmd_handle* h, mmd_float prd[3], mmd_float lo[3], mmd_float hi[3])
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  for(int d = 0; d < 3; d++) { if(prd) prd[d] = h->prd[d]; if(lo) lo[d] = h->lo[d]; if(hi) hi[d] = h->hi[d]; }
  return 0;
}

extern "C" int mmd_atom_set_mass(mmd_handle* h, mmd_float mass)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  h->mass = mass;
  return 0;
}

extern "C" int mmd_atom_upload(mmd_handle* h, const mmd_float* x, const mmd_float* v, const int* type, const int* tag,
                               int nlocal, int nghost)
{
  if(!h || !x || !type || nlocal < 0 || nghost < 0) { mmd_set_error("mmd_atom_upload: bad arguments"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  const int nall = nlocal + nghost;
  h->nlocal = 0; h->nghost = 0;
  MMD_TRY(mmd_ensure_atoms(h, nall + nall / 4 + 4096, false));
  DevArr<real> tmp;
  MMD_TRY(tmp.ensure((size_t)3 * nall + 1, false, h->stream));
  HIP_TRY(hipMemcpyAsync(tmp.p, x, (size_t)3 * nall * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->type.p, type, (size_t)nall * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if(nall) hipLaunchKernelGGL(k_pack_x4, dim3(div_up(nall, 256)), dim3(256), 0, h->stream, tmp.p, h->type.p, h->x.p, nall);
  if(v) HIP_TRY(hipMemcpyAsync(h->v.p, v, (size_t)3 * nlocal * sizeof(real), hipMemcpyHostToDevice, h->stream));
  else HIP_TRY(hipMemsetAsync(h->v.p, 0, (size_t)3 * nlocal * sizeof(real), h->stream));
  HIP_TRY(hipMemsetAsync(h->f.p, 0, (size_t)3 * nall * sizeof(real), h->stream));
  if(tag) HIP_TRY(hipMemcpyAsync(h->tag.p, tag, (size_t)nlocal * sizeof(int), hipMemcpyHostToDevice, h->stream));
  else if(nlocal) hipLaunchKernelGGL(k_iota, dim3(div_up(nlocal, 256)), dim3(256), 0, h->stream, h->tag.p, nlocal);
  HIP_TRY(hipGetLastError());
  h->nlocal = nlocal;
  h->nghost = nghost;
  h->ghosts_uploaded = nghost > 0;         // (whatever Comm::borders recorded about the previous ghosts — send lists, owners — is stale now)
  h->ghost_chain_ok = false;
  MMD_TRY(mmd_set_dummy(h));
  HIP_TRY(mmd_stream_sync(h));
  tmp.release();
  h->neigh_nlocal = 0;
  return 0;
}

// positions only, atom for atom (the state a reference Atom is in after initialIntegrate + Comm::communicate of a step without
// re-neighboring, ref/integrate.cpp:94-105): types, velocities, tags, ghost bookkeeping and the neighbor list stay valid
extern "C" int mmd_atom_upload_x(mmd_handle* h, const mmd_float* x, int nall)
{
  if(!h || !x || nall != h->nlocal + h->nghost) { mmd_set_error("mmd_atom_upload_x: bad arguments (%d atoms here, %d given)", h ? h->nlocal + h->nghost : 0, nall); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  if(nall == 0) return 0;
  MMD_TRY(h->x_stage.ensure((size_t)3 * nall + 1, false, h->stream));
  HIP_TRY(hipMemcpyAsync(h->x_stage.p, x, (size_t)3 * nall * sizeof(real), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_pack_x4, dim3(div_up(nall, 256)), dim3(256), 0, h->stream, h->x_stage.p, h->type.p, h->x.p, nall);
  HIP_TRY(hipGetLastError());
  HIP_TRY(mmd_stream_sync(h));                    // (the caller's array is borrowed for the duration of the call only)
  h->ghosts_stale = false;
  return 0;
}

extern "C" int mmd_atom_download(mmd_handle* h, mmd_float* x, mmd_float* v, mmd_float* f, int* type, int* tag)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  const int nall = h->nlocal + h->nghost;
  if(x && nall) {
    DevArr<real> tmp;
    MMD_TRY(tmp.ensure((size_t)3 * nall, false, h->stream));
    hipLaunchKernelGGL(k_unpack_x4, dim3(div_up(nall, 256)), dim3(256), 0, h->stream, h->x.p, tmp.p, nall);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(x, tmp.p, (size_t)3 * nall * sizeof(real), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    tmp.release();
  }
  if(v) HIP_TRY(hipMemcpyAsync(v, h->v.p, (size_t)3 * h->nlocal * sizeof(real), hipMemcpyDeviceToHost, h->stream));
  if(f) {
    const int nf = h->halfneigh ? nall : h->nlocal;
    HIP_TRY(hipMemcpyAsync(f, h->f.p, (size_t)3 * nf * sizeof(real), hipMemcpyDeviceToHost, h->stream));
  }
  if(type) HIP_TRY(hipMemcpyAsync(type, h->type.p, (size_t)nall * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if(tag) HIP_TRY(hipMemcpyAsync(tag, h->tag.p, (size_t)h->nlocal * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  return 0;
}

extern "C" int mmd_atom_upload_f(mmd_handle* h, const mmd_float* f, int n)
{
  if(!h || !f || n > h->nmax) { mmd_set_error("mmd_atom_upload_f: bad arguments"); return -1; }
  HIP_TRY(hipMemcpyAsync(h->f.p, f, (size_t)3 * n * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  return 0;
}

extern "C" int mmd_atom_counts(mmd_handle* h, int* nlocal, int* nghost, int* nmax)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(nlocal) *nlocal = h->nlocal;
  if(nghost) *nghost = h->nghost;
  if(nmax) *nmax = h->nmax;
  return 0;
}

// ---- Atom::pbc (ref/atom.cpp:106-122): the two tests per dimension are ordered so that
//      lo <= coord < hi holds even when (coord +/- eps) +/- period == bound ---------------------------
__global__ void k_pbc(real4* __restrict__ x, int n, real xprd, real yprd, real zprd)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  real4 p = x[i];
  if(p.x < (real)0.0) p.x += xprd;
  if(p.x >= xprd) p.x -= xprd;
  if(p.y < (real)0.0) p.y += yprd;
  if(p.y >= yprd) p.y -= yprd;
  if(p.z < (real)0.0) p.z += zprd;
  if(p.z >= zprd) p.z -= zprd;
  x[i] = p;
}

extern "C" int mmd_atom_pbc(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(h->pbc_defer) { h->pbc_pending = true; return 0; }        // (Atom::sort follows: its binning pass wraps the atoms on the way)
  if(h->nlocal)
    hipLaunchKernelGGL(k_pbc, dim3(div_up(h->nlocal, 256)), dim3(256), 0, h->stream, h->x.p, h->nlocal, h->prd[0], h->prd[1], h->prd[2]);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- Atom::sort (ref/atom.cpp:355-421): counting sort of the owned atoms by bin; x, v, type (and our
//      tag) are permuted, f is not (it is recomputed before its next use, as in the reference).
//      Bins are numbered block-major (2x2x2 bins per block) so 64 consecutive atoms = one compact cube.
__global__ void k_sort_permute(const int* __restrict__ binned, int n, const real4* __restrict__ x, const real* __restrict__ v,
                               const int* __restrict__ type, const int* __restrict__ tag, real4* __restrict__ xo,
                               real* __restrict__ vo, int* __restrict__ typeo, int* __restrict__ tago)
{
  const int dst = blockIdx.x * blockDim.x + threadIdx.x;
  if(dst >= n) return;
  const int src = binned[dst];
  xo[dst] = x[src];
  vo[3 * dst + 0] = v[3 * src + 0]; vo[3 * dst + 1] = v[3 * src + 1]; vo[3 * dst + 2] = v[3 * src + 2];
  typeo[dst] = type[src];
  tago[dst] = tag[src];
}

extern "C" int mmd_atom_sort(mmd_handle* h)
{
  if(!h || !h->neigh_ready) { mmd_set_error("mmd_atom_sort: neighbor bins are not set up"); return -1; }
  const int n = h->nlocal;
  if(n == 0) return 0;
  MMD_TRY(mmd_bin_atoms(h, n));               // Neighbor::binatoms(atom, nlocal)
  // bins longer than the one-thread sort handles are ordered by the grid-wide rank count, which is skipped until such a bin has
  // been seen. Inside Integrate::run the neighbor build that follows reads the flag with its own results; a bare Atom::sort
  // (C-ABI users) reads it here, so its permutation never depends on the arrival order of the histogram atomics
  if(!h->big_bins && !h->in_reneighbor) {
    HIP_TRY(hipMemcpyAsync(h->h_flags + 12, h->d_flags + 12, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    if(h->h_flags[12]) { h->big_bins = true; MMD_TRY(mmd_bin_atoms(h, n)); }
  }
  // the copies only need to hold the current atoms (+ dummy slot); sizing them by the live arrays' capacity
  // made the two buffers leap-frog each other and re-allocate on every sort
  const size_t need = (size_t)h->nmax + 1;
  MMD_TRY(h->x_alt.ensure(need, false, h->stream));
  MMD_TRY(h->v_alt.ensure(3 * need, false, h->stream));
  MMD_TRY(h->type_alt.ensure(need, false, h->stream));
  MMD_TRY(h->tag_alt.ensure(need, false, h->stream));
  hipLaunchKernelGGL(k_sort_permute, dim3(div_up(n, 256)), dim3(256), 0, h->stream, h->binned.p, n, h->x.p, h->v.p, h->type.p,
                     h->tag.p, h->x_alt.p, h->v_alt.p, h->type_alt.p, h->tag_alt.p);
  HIP_TRY(hipGetLastError());
  // ghosts (+ dummy slot) ride along unchanged — except inside Integrate::run, where Comm::borders rebuilds them right after
  if(!h->in_reneighbor) {
    HIP_TRY(hipMemcpyAsync(h->x_alt.p + n, h->x.p + n, ((size_t)h->nghost + 1) * sizeof(real4), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->type_alt.p + n, h->type.p + n, (size_t)h->nghost * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  }
  std::swap(h->x, h->x_alt);
  std::swap(h->v, h->v_alt);
  std::swap(h->type, h->type_alt);
  std::swap(h->tag, h->tag_alt);
  h->neigh_nlocal = 0;            // atom indices changed: any neighbor list is stale
  h->dh.ready = false;
  h->tiles_ready = false;
  h->cand_src_ready = false;
  return 0;
}
