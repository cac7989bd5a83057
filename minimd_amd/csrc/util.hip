// minimd_amd/csrc/util.hip — error text, wall clock, device exclusive scan, capacity management.
#include <time.h>

#include "device_utils.hpp"
#include "mmd_internal.hpp"

static thread_local char g_err[1024] = "";
void mmd_set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mmd_last_error(void) { return g_err; }

double mmd_wall()
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

// ---------------------------------------------------------------------------------------------------
// in-place exclusive scan of int[n]: (1) per-tile totals, (2) one workgroup scans the tile totals,
// (3) per-tile scan + offset.  Tiles of 1024 elements, 256 threads x 4 items.
// ---------------------------------------------------------------------------------------------------
#define SCAN_TILE 1024

__global__ __launch_bounds__(256) void k_scan_tile_sums(const int* __restrict__ d, int n, int* __restrict__ sums)
{
  __shared__ int lds[17];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) if(base + k < n) s += d[base + k];
  int tot;
  block_incl_scan(s, lds, &tot);
  if(threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_scan_sums(const int* src, int* dst, int ntiles, int* total)      // (src may be dst)
{
  __shared__ int lds[17];
  int carry = 0;
  for(int b = 0; b < ntiles; b += 1024) {
    const int i = b + threadIdx.x;
    const int v = i < ntiles ? src[i] : 0;
    int tot;
    const int inc = block_incl_scan(v, lds, &tot);
    if(i < ntiles) dst[i] = carry + inc - v;
    carry += tot;
  }
  if(threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const int* src, int* d, int n, const int* __restrict__ sums)   // (src may be d)
{
  __shared__ int lds[17];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int v[4];
  int s = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) { v[k] = base + k < n ? src[base + k] : 0; s += v[k]; }
  int tot;
  const int inc = block_incl_scan(s, lds, &tot);
  int run = sums[blockIdx.x] + inc - s;
#pragma unroll
  for(int k = 0; k < 4; k++) {
    if(base + k < n) d[base + k] = run;
    run += v[k];
  }
}

// second launch of the two-launch scan: every workgroup sums the tile totals in front of it itself (a few hundred ints: no launch for a
// scan of the totals), then scans its tile; the last workgroup also stores the grand total
__global__ __launch_bounds__(256) void k_scan_apply_self(const int* src, int* d, int n, const int* __restrict__ sums, int* total)   // (src may be d)
{
  __shared__ int lds[17];
  const int before = block_prefix_total(sums, blockIdx.x, lds);
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  int v[4];
  int s = 0;
#pragma unroll
  for(int k = 0; k < 4; k++) { v[k] = base + k < n ? src[base + k] : 0; s += v[k]; }
  int tot;
  const int inc = block_incl_scan(s, lds, &tot);
  int run = before + inc - s;
#pragma unroll
  for(int k = 0; k < 4; k++) {
    if(base + k < n) d[base + k] = run;
    run += v[k];
  }
  if(blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = before + tot;
}

// exclusive scan of src[n] into data[n] (src == data: in place). data[n] must be writable: the grand total is also stored there
// (bin_start[mbins] convention)
int mmd_exclusive_scan_from(mmd_handle* h, const int* src, int* data, int n, int* total_host)
{
  if(n <= 32768) {            // short arrays (per-tile counts of the compactions, bins of small boxes): one workgroup
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, h->stream, src, data, n, data + n);
  } else {
    const int ntiles = div_up(n, SCAN_TILE);
    MMD_TRY(h->scan_tmp.ensure((size_t)ntiles + 8, false, h->stream));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(256), 0, h->stream, src, n, h->scan_tmp.p);
    if(ntiles <= 8192) {      // two launches: each workgroup of the second sums the totals in front of it (<= 32 per thread)
      hipLaunchKernelGGL(k_scan_apply_self, dim3(ntiles), dim3(256), 0, h->stream, src, data, n, (const int*)h->scan_tmp.p, data + n);
    } else {
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, h->stream, (const int*)h->scan_tmp.p, h->scan_tmp.p, ntiles, data + n);
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(256), 0, h->stream, src, data, n, h->scan_tmp.p);
    }
  }
  HIP_TRY(hipGetLastError());
  if(total_host) {
    HIP_TRY(hipMemcpyAsync(h->h_flags, data + n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(mmd_stream_sync(h));
    *total_host = h->h_flags[0];
  }
  return 0;
}
int mmd_exclusive_scan(mmd_handle* h, int* data, int n, int* total_host) { return mmd_exclusive_scan_from(h, data, data, n, total_host); }

// ---------------------------------------------------------------------------------------------------
// per-atom capacity (Atom::growarray, ref/atom.cpp:71-84) — +1 slot for the dummy atom
// ---------------------------------------------------------------------------------------------------
int mmd_ensure_atoms(mmd_handle* h, int n, bool preserve)
{
  if(n + 1 <= h->nmax) return 0;
  const size_t keep = (size_t)h->nlocal + h->nghost + 1;
  const size_t want = (size_t)n + 1;
  MMD_TRY(h->x.ensure(want, preserve, h->stream, keep));
  MMD_TRY(h->v.ensure(3 * want, preserve, h->stream, 3 * keep));
  MMD_TRY(h->f.ensure(3 * want, preserve, h->stream, 3 * keep));
  MMD_TRY(h->type.ensure(want, preserve, h->stream, keep));
  MMD_TRY(h->tag.ensure(want, preserve, h->stream, keep));
  size_t c = h->x.cap;
  if(h->v.cap / 3 < c) c = h->v.cap / 3;
  if(h->f.cap / 3 < c) c = h->f.cap / 3;
  if(h->type.cap < c) c = h->type.cap;
  if(h->tag.cap < c) c = h->tag.cap;
  h->nmax = (int)c - 1;
  return 0;
}

__global__ void k_set_dummy(real4* x, int slot)
{
  // far outside any cutoff, finite in float and double after squaring and summing
  x[slot] = real4{(real)1.0e15, (real)1.0e15, (real)1.0e15, (real)0};
}

int mmd_prepare_x_alt(mmd_handle* h)
{
  MMD_TRY(h->x_alt.ensure((size_t)h->nmax + 1, false, h->stream));
  const int slot = h->nlocal + h->nghost;
  // the two position buffers alternate every fused step; each needs its dummy atom written once per re-neighboring
  int q = -1;
  for(int k = 0; k < 2; k++) if(h->xalt_dummy_ptr[k] == (const void*)h->x_alt.p) q = k;
  if(q < 0) { q = h->xalt_dummy_next; h->xalt_dummy_next ^= 1; h->xalt_dummy_ptr[q] = h->x_alt.p; h->xalt_dummy_slot[q] = -1; }
  // (a launch behind the neighbor build: the ghost count is still on the device, the gated kernel writes the dummy atom itself)
  if(h->spec.gate != nullptr) { h->xalt_dummy_slot[q] = -1; return 0; }
  h->xalt_dummy_slot[q] = slot;             // (the fused force kernels write the dummy atom with the rest of the buffer they fill)
  return 0;
}

// the box lengths in device memory (ghost_shifted, tile_lds.hpp)
int mmd_box_dev(mmd_handle* h)
{
  if(h->box_dev_valid) return 0;
  MMD_TRY(h->box_dev.ensure(4, false, h->stream));
  HIP_TRY(hipMemcpyAsync(h->box_dev.p, h->prd, 3 * sizeof(real), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  h->box_dev_valid = true;
  return 0;
}

int mmd_set_dummy(mmd_handle* h)
{
  hipLaunchKernelGGL(k_set_dummy, dim3(1), dim3(1), 0, h->stream, h->x.p, h->nlocal + h->nghost);
  HIP_TRY(hipGetLastError());
  for(int k = 0; k < 2; k++) if(h->xalt_dummy_ptr[k] == (const void*)h->x.p) h->xalt_dummy_slot[k] = h->nlocal + h->nghost;
  return 0;
}
