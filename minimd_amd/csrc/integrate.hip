// minimd_amd/csrc/integrate.hip — Integrate::initialIntegrate/finalIntegrate (ref/integrate.cpp:46-68) and the
// kinetic-energy sum of Thermo::temperature (ref/thermo.cpp:151-157). Pure HBM streams; compiled with
// -ffp-contract=off so v += dtforce*f ; x += dt*v round exactly like the reference.
#include "device_utils.hpp"
#include "mmd_internal.hpp"

__global__ __launch_bounds__(256) void k_initial_integrate(real4* __restrict__ x, real* __restrict__ v, const real* __restrict__ f,
                                                           int n, real dt, real dtforce)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  real4 p = x[i];
  real vx = v[3 * (size_t)i + 0], vy = v[3 * (size_t)i + 1], vz = v[3 * (size_t)i + 2];
  vx += dtforce * f[3 * (size_t)i + 0];
  vy += dtforce * f[3 * (size_t)i + 1];
  vz += dtforce * f[3 * (size_t)i + 2];
  p.x += dt * vx; p.y += dt * vy; p.z += dt * vz;
  v[3 * (size_t)i + 0] = vx; v[3 * (size_t)i + 1] = vy; v[3 * (size_t)i + 2] = vz;
  x[i] = p;
}

__global__ __launch_bounds__(256) void k_final_integrate(real* __restrict__ v, const real* __restrict__ f, int n, real dtforce)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= 3 * n) return;
  v[i] += dtforce * f[i];
}

// finalIntegrate of step n fused with initialIntegrate of step n+1 (same f, same operation order:
// v += dtf*f ; v += dtf*f ; x += dt*v) — 136 instead of 208 bytes per atom; used when step n is not a thermo step
// zero_f: the forces are cleared on the way out (half lists accumulate into f: the next Force::compute then needs no separate fill pass)
__global__ __launch_bounds__(256) void k_final_initial_integrate(real4* __restrict__ x, real* __restrict__ v, real* __restrict__ f,
                                                                 int n, real dt, real dtforce, int zero_f)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if(i >= n) return;
  real4 p = x[i];
  real vx = v[3 * (size_t)i + 0], vy = v[3 * (size_t)i + 1], vz = v[3 * (size_t)i + 2];
  const real fx = f[3 * (size_t)i + 0], fy = f[3 * (size_t)i + 1], fz = f[3 * (size_t)i + 2];
  vx += dtforce * fx; vy += dtforce * fy; vz += dtforce * fz;
  vx += dtforce * fx; vy += dtforce * fy; vz += dtforce * fz;
  p.x += dt * vx; p.y += dt * vy; p.z += dt * vz;
  v[3 * (size_t)i + 0] = vx; v[3 * (size_t)i + 1] = vy; v[3 * (size_t)i + 2] = vz;
  x[i] = p;
  if(zero_f) { f[3 * (size_t)i + 0] = 0; f[3 * (size_t)i + 1] = 0; f[3 * (size_t)i + 2] = 0; }
}

int mmd_integrate_final_initial(mmd_handle* h)
{
  const int zero_f = h->zero_f_in_integrate ? 1 : 0;
  if(h->nlocal)
    hipLaunchKernelGGL(k_final_initial_integrate, dim3(div_up(h->nlocal, 256)), dim3(256), 0, h->stream, h->x.p, h->v.p, h->f.p, h->nlocal, h->dt, h->dtforce, zero_f);
  HIP_TRY(hipGetLastError());
  h->f_zeroed_n = zero_f ? h->nlocal : 0;
  return 0;
}

__global__ __launch_bounds__(256) void k_temperature(const real* __restrict__ v, int n, real mass, double* __restrict__ partials)
{
  __shared__ double s_red[16];
  double t = 0;
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const real vx = v[3 * (size_t)i + 0], vy = v[3 * (size_t)i + 1], vz = v[3 * (size_t)i + 2];
    t += (double)((vx * vx + vy * vy + vz * vz) * mass);
  }
  const double s = block_sum(t, s_red);
  if(threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_sum1(const double* __restrict__ partials, int nblocks, double* __restrict__ out)
{
  __shared__ double s_red[16];
  double s = 0;
  for(int b = threadIdx.x; b < nblocks; b += blockDim.x) s += partials[b];
  const double t = block_sum(s, s_red);
  if(threadIdx.x == 0) *out = t;
}

extern "C" int mmd_integrate_setup(mmd_handle* h, mmd_float dt, mmd_float dtforce, int neigh_every, int sort_every)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(neigh_every < 1) { mmd_set_error("mmd_integrate_setup: neigh_every must be >= 1"); return -1; }
  h->dt = dt; h->dtforce = dtforce; h->neigh_every = neigh_every; h->sort_every = sort_every;
  h->next_sort = -1;
  return 0;
}

extern "C" int mmd_integrate_initial(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(h->nlocal)
    hipLaunchKernelGGL(k_initial_integrate, dim3(div_up(h->nlocal, 256)), dim3(256), 0, h->stream, h->x.p, h->v.p, h->f.p, h->nlocal, h->dt, h->dtforce);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int mmd_integrate_final(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  if(h->nlocal)
    hipLaunchKernelGGL(k_final_integrate, dim3(div_up(3LL * h->nlocal, 256)), dim3(256), 0, h->stream, h->v.p, h->f.p, h->nlocal, h->dtforce);
  HIP_TRY(hipGetLastError());
  return 0;
}

// enqueue the reduction; result lands in h->d_result[slot]
int mmd_temperature_async(mmd_handle* h, int slot)
{
  const int nb = 1024;
  MMD_TRY(h->partials.ensure((size_t)nb + 8, false, h->stream));
  hipLaunchKernelGGL(k_temperature, dim3(nb), dim3(256), 0, h->stream, h->v.p, h->nlocal, h->mass, h->partials.p);
  hipLaunchKernelGGL(k_sum1, dim3(1), dim3(1024), 0, h->stream, h->partials.p, nb, h->d_result + slot);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- --check_exchange (ref/integrate.cpp:112-151): largest move of an owned atom since the last re-neighboring ----
// per-atom distance with the reference's single +-prd correction; squared maxima of non-negative doubles are
// combined with an integer atomicMax on their bit patterns (order preserving)
__global__ __launch_bounds__(256) void k_max_move(const real4* __restrict__ x, const real4* __restrict__ xold, int n, real px, real py, real pz,
                                                  unsigned long long* __restrict__ out)
{
  __shared__ double s_red[4];
  double m = 0;
  for(int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const real4 a = x[i], b = xold[i];
    double dx = (double)a.x - (double)b.x, dy = (double)a.y - (double)b.y, dz = (double)a.z - (double)b.z;
    if(dx > px) dx -= px;
    if(dx < -px) dx += px;
    if(dy > py) dy -= py;
    if(dy < -py) dy += py;
    if(dz > pz) dz -= pz;
    if(dz < -pz) dz += pz;
    const double d = dx * dx + dy * dy + dz * dz;
    m = d > m ? d : m;
  }
  m = wave_max_d(m);
  if((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  if(threadIdx.x == 0) {
    for(int w = 1; w < (int)(blockDim.x >> 6); w++) m = s_red[w] > m ? s_red[w] : m;
    atomicMax(out, (unsigned long long)__double_as_longlong(m));
  }
}

extern "C" int mmd_integrate_mark_positions(mmd_handle* h)
{
  if(!h) { mmd_set_error("null handle"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  MMD_TRY(h->xold.ensure((size_t)h->nlocal + 1, false, h->stream));
  if(h->nlocal) HIP_TRY(hipMemcpyAsync(h->xold.p, h->x.p, (size_t)h->nlocal * sizeof(real4), hipMemcpyDeviceToDevice, h->stream));
  h->xold_n = h->nlocal;
  return 0;
}

extern "C" int mmd_integrate_max_move(mmd_handle* h, double* d_max)
{
  if(!h || !d_max) { mmd_set_error("mmd_integrate_max_move: bad arguments"); return -1; }
  if(h->xold_n != h->nlocal) { mmd_set_error("mmd_integrate_max_move: no marked positions for the current atoms"); return -1; }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemsetAsync(h->d_result + 8, 0, sizeof(double), h->stream));
  if(h->nlocal) hipLaunchKernelGGL(k_max_move, dim3(512), dim3(256), 0, h->stream, h->x.p, h->xold.p, h->nlocal, h->prd[0], h->prd[1], h->prd[2],
                                   (unsigned long long*)(h->d_result + 8));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h->h_result + 8, h->d_result + 8, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  *d_max = sqrt(h->h_result[8]);
  return 0;
}

extern "C" int mmd_thermo_temperature(mmd_handle* h, double* sum_mv2)
{
  if(!h || !sum_mv2) { mmd_set_error("mmd_thermo_temperature: bad arguments"); return -1; }
  MMD_TRY(mmd_temperature_async(h, 2));
  HIP_TRY(hipMemcpyAsync(h->h_result + 2, h->d_result + 2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(mmd_stream_sync(h));
  *sum_mv2 = h->h_result[2];
  return 0;
}
