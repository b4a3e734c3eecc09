// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on this engine's access patterns (known byte counts):
//   rec_copy : every lane reads one 128-byte record with 8 x 16 B loads and writes it back with 8 x 16 B stores
//   row_write: a 64-lane block writes one 274-float row with 4 B/lane coalesced stores (the observation row)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void rec_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 t[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) t[k] = src[(size_t)i * 8 + k];
#pragma unroll
  for (int k = 0; k < 8; ++k) { t[k].x += 1; dst[(size_t)i * 8 + k] = t[k]; }
}
__global__ void row_write(float* __restrict__ dst, int D) {
  float* row = dst + (size_t)blockIdx.x * D;
  for (int k = threadIdx.x; k < D; k += blockDim.x) row[k] = (float)k;
}
int main() {
  const int n = 4096 * 17, rows = 4096, D = 274;
  uint4 *a, *b; float* o;
  hipMalloc(&a, (size_t)n * 128); hipMalloc(&b, (size_t)n * 128); hipMalloc(&o, (size_t)rows * D * 4);
  hipMemset(a, 1, (size_t)n * 128);
  for (int it = 0; it < 20; ++it) {
    hipLaunchKernelGGL(rec_copy, dim3((n + 63) / 64), dim3(64), 0, 0, a, b, n);
    hipLaunchKernelGGL(row_write, dim3(rows), dim3(64), 0, 0, o, D);
  }
  hipDeviceSynchronize();
  printf("rec_copy: read %d B, write %d B per launch; row_write: write %d B per launch\n", n * 128, n * 128, rows * D * 4);
  return 0;
}
