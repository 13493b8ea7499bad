// Microbenchmark: v_mfma_f64_16x16x4_f64 vs v_fma_f64 issue rates on gfx950, and whether
// the two pipes overlap across waves of one SIMD.  Informs the kernel design.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_valu(double* out, int iters) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512 threads: waves 0-3 MFMA (4 acc), waves 4-7 VALU; each SIMD hosts one of each
__global__ __launch_bounds__(512) void k_mixed(double* out, int iters_m, int iters_v) {
  const int wave = threadIdx.x >> 6;
  double s = 0;
  if (wave < 4) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
  double* out; hipMalloc(&out, sizeof(double) * 512 * 256 * 8);
  const int blocks = 256;  // one per CU
  const int iters = 200000;
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device %s CUs %d clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  float t1 = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
  float t4 = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, iters / 4); });
  float tv = timeit([&] { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, out, iters); });
  float t8w = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks * 2), dim3(256), 0, 0, out, iters / 4); });
  float tm = timeit([&] { hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(512), 0, 0, out, iters / 4, iters); });
  // per-SIMD: one wave per SIMD; MFMAs per wave = iters
  printf("mfma dep chain (1 acc): %.3f ms -> %.1f ns/MFMA (= %.1f cyc @2.4GHz)\n", t1, t1 * 1e6 / iters, t1 * 1e6 / iters * 2.4);
  printf("mfma 4 acc           : %.3f ms -> %.1f ns/MFMA (= %.1f cyc @2.4GHz), %.1f TFLOP/s chip\n", t4, t4 * 1e6 / iters,
         t4 * 1e6 / iters * 2.4, 2.0 * 1024 * iters * 4.0 * blocks / (t4 * 1e-3) / 1e12);
  printf("mfma 4 acc, 2 waves/SIMD: %.3f ms (x%.2f of 1 wave)\n", t8w, t8w / t4);
  printf("valu fma f64 x8      : %.3f ms -> %.2f ns/FMA-instr (= %.2f cyc), %.1f TFLOP/s chip\n", tv, tv * 1e6 / (iters * 8.0),
         tv * 1e6 / (iters * 8.0) * 2.4, 2.0 * 64 * iters * 8.0 * 4 * blocks / (tv * 1e-3) / 1e12);
  printf("mixed (mfma wave + valu wave per SIMD): %.3f ms vs mfma alone %.3f, valu alone %.3f, sum %.3f\n", tm, t4, tv, t4 + tv);
  return 0;
}
