// Checks the cross-lane primitives the sparse kernels rely on (gfx950):
//  - DPP row_ror all-reduce within 16-lane rows
//  - v_permlane16_swap / v_permlane32_swap as xor-16 / xor-32 sum
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ double row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum(double v) {
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}
__device__ __forceinline__ double sum_g(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
  l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  double v = 1.0 + lane * 0.5 + (lane * lane) * 0.001;
  out[lane] = row_sum(v);
  out[64 + lane] = sum_g(v);
}
int main() {
  double* d; hipMalloc(&d, 128 * 8); double h[128];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  auto val = [](int l) { return 1.0 + l * 0.5 + (l * l) * 0.001; };
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    double rs = 0, gs = 0;
    for (int c = 0; c < 16; ++c) rs += val((l & ~15) + c);
    for (int g = 0; g < 4; ++g) gs += val((l & 15) + 16 * g);
    if (fabs(h[l] - rs) > 1e-9) { ++bad; if (bad < 5) printf("row_sum lane %d got %f want %f\n", l, h[l], rs); }
    if (fabs(h[64 + l] - gs) > 1e-9) { ++bad; if (bad < 9) printf("sum_g lane %d got %f want %f\n", l, h[64 + l], gs); }
  }
  printf(bad ? "XLANE CHECK FAILED (%d)\n" : "XLANE CHECK OK\n", bad);
  return bad != 0;
}
