// Probe for the neighbor build's MFMA pre-test: v_mfma_f32_32x32x16_f16 on gfx950 —
//   (1) operand / result layout (lane -> row, column, k), (2) are f16 DENORMAL inputs honoured, (3) accumulation error against the exact sum
//       in units of 2^-24 * sum|terms| (the error model of the pre-test's band assumes <= 15 roundings of 2^-23 relative to sum|terms|).
// build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 tools/probes/mfma_f16_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_probe(const _Float16* __restrict__ A /* [32][16] */, const _Float16* __restrict__ B /* [16][32] as [col][k] */, float* __restrict__ D /* [64][16] */)
{
  const int l = threadIdx.x;
  h8 a, b;
  for(int j = 0; j < 8; j++) { a[j] = A[(l % 32) * 16 + 8 * (l / 32) + j]; b[j] = B[(l % 32) * 16 + 8 * (l / 32) + j]; }
  const f16v c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for(int i = 0; i < 16; i++) D[l * 16 + i] = d[i];
}

static double frand() { return (double)rand() / RAND_MAX; }

int main()
{
  std::vector<_Float16> A(32 * 16), B(32 * 16);
  std::vector<float> D(64 * 16);
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
  double worst = 0, worst_den = 0;
  int layout_bad = 0;
  for(int trial = 0; trial < 200; trial++) {
    const int mode = trial % 4;     // 0: O(1) values, 1: mixed magnitudes with cancellation, 2: denormal f16 in A, 3: denormals in both
    for(int i = 0; i < 32 * 16; i++) {
      double va = (frand() - 0.5) * 16, vb = (frand() - 0.5) * 10;
      if(mode == 1) { va *= (i % 3 == 0) ? 12.0 : 0.01; vb *= (i % 5 == 0) ? 8.0 : 0.02; }
      if(mode >= 2 && (i % 2) == 0) va = (frand() - 0.5) * 1.0e-4;       // |x| < 6.1e-5: f16 denormal
      if(mode == 3 && (i % 3) == 0) vb = (frand() - 0.5) * 1.0e-4;
      A[i] = (_Float16)va; B[i] = (_Float16)vb;
    }
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    for(int l = 0; l < 64; l++)
      for(int i = 0; i < 16; i++) {
        const int col = l % 32, row = 8 * (i / 4) + 4 * (l / 32) + i % 4;      // assumed result layout
        double ex = 0, sa = 0;
        for(int k = 0; k < 16; k++) { const double p = (double)A[row * 16 + k] * (double)B[col * 16 + k]; ex += p; sa += fabs(p); }
        const double err = fabs((double)D[l * 16 + i] - ex) / (sa * 5.9604644775390625e-08 + 1e-300);
        if(err > 64) layout_bad++;
        else if(mode >= 2) worst_den = fmax(worst_den, err); else worst = fmax(worst, err);
      }
  }
  printf("mfma_f32_32x32x16_f16: results outside 64 x 2^-24 x sum|terms| of the assumed layout: %d of %d\n", layout_bad, 200 * 1024);
  printf("worst error, normal inputs      : %.3f x 2^-24 x sum|terms|\n", worst);
  printf("worst error, denormal f16 inputs: %.3f x 2^-24 x sum|terms|   (flushed denormals would show as errors >> 1 here: the denormal terms are most of the sum in a third of the rows)\n", worst_den);
  // a direct denormal check: A row = one denormal, B = 1.0 -> D must equal the denormal
  for(int i = 0; i < 32 * 16; i++) { A[i] = (_Float16)0; B[i] = (_Float16)0; }
  for(int r = 0; r < 32; r++) { A[r * 16 + (r % 16)] = (_Float16)3.0e-6; B[r * 16 + (r % 16)] = (_Float16)1.0; }
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("denormal 3.0e-6 (f16 %.9g) x 1.0 on the diagonal: D[0][0] = %.9g (0 would mean flushed)\n", (double)(_Float16)3.0e-6, (double)D[0]);
  return 0;
}
