// Microbenchmark (DESIGN.md 4.8 / profiles/r04_notes.md section 9): what bounds the record stores of the per-wave d = 40 kernels?
// 2000 "series", each one wave writing 1000 records of 1640 doubles (13 120 B):
//   pattern 0: as k_filter_w48's store_record -- 18 instructions of 16 bytes per lane at the column-major tile addresses (64 separate pieces each)
//   pattern 1: 13 instructions of 16 bytes per lane, lane L of instruction k -> bytes [16 (64 k + L), +16): one contiguous KB each
// with the LDS per workgroup chosen so that 1, 2 or 4 waves share a SIMD.   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x, double y) {
  const u4 v = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x), (unsigned)__double2loint(y), (unsigned)__double2hiint(y)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}
template <int PATTERN, int YLOAD>
__global__ __launch_bounds__(64) void k_store(double* out, int T, int d, double seed, const double* __restrict__ y) {
  extern __shared__ double pad[];
  const int n = blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int rec = d + d * d, recb = rec * 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)n * (T + 1) * rec, 0, (int)((size_t)(T + 1) * recb), 0x00020000);
  const int OOB = 0x7ffffff0;
  double v = seed + lane;
  if (lane == 0) pad[0] = v;
  const double* yn = y + (size_t)n * T * 20;
  double ynext = (YLOAD && lane < 20) ? yn[lane] : 0.0;
  for (int t = 0; t <= T; ++t) {
    const int so = t * recb;
    if (YLOAD) {   // the next observation, requested one step ahead and consumed here: its wait also waits for every store issued before it
      const double ycur = ynext;
      ynext = (lane < 20 && t + 1 < T) ? yn[(size_t)(t + 1) * 20 + lane] : 0.0;
      v += ycur * 1e-9;
    }
    if (PATTERN == 0) {
#pragma unroll
      for (int aa = 0; aa < 3; ++aa)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int rr = 0; rr < 4; rr += 2) {
            const int i = 16 * aa + 4 * (rr + (g & 1)) + (g & 2), col = 16 * b + c;
            st16(r, (i < d && col < d) ? (d + col * d + i) * 8 : OOB, so, v, v + 1.0);
          }
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const u4 dummy = {0, 0, 0, 0}; (void)dummy;
        __builtin_amdgcn_raw_buffer_store_b64((unsigned __attribute__((ext_vector_type(2)))){(unsigned)__double2loint(v), (unsigned)__double2hiint(v)}, r, (g == 0 && 16 * b + c < d) ? (16 * b + c) * 8 : OOB, so, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 13; ++k) {
        const int q = 64 * k + lane;
        st16(r, q * 16 < recb ? q * 16 : OOB, so, v, v + 1.0);
      }
    }
    v += 1e-3;
  }
}
int main(int argc, char** argv) {
  const int N = 2000, T = 1000, d = 40, rec = d + d * d;
  double* out;
  const size_t bytes = (size_t)N * (T + 1) * rec * 8;
  if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  double* y; hipMalloc(&y, (size_t)N * T * 20 * 8); hipMemset(y, 0, (size_t)N * T * 20 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4}) {
    const size_t lds = wps == 1 ? 40 * 1024 : (wps == 2 ? 20 * 1024 : 10 * 1024);   // 160 KB / (4 SIMDs x wps) per one-wave workgroup
    for (int yl = 0; yl < 2; ++yl)
    for (int pat = 0; pat < 2; ++pat) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (pat == 0 && !yl) hipLaunchKernelGGL((k_store<0, 0>), dim3(N), dim3(64), lds, 0, out, T, d, 1.0 + rep, (const double*)y);
        else if (pat == 1 && !yl) hipLaunchKernelGGL((k_store<1, 0>), dim3(N), dim3(64), lds, 0, out, T, d, 1.0 + rep, (const double*)y);
        else if (pat == 0) hipLaunchKernelGGL((k_store<0, 1>), dim3(N), dim3(64), lds, 0, out, T, d, 1.0 + rep, (const double*)y);
        else hipLaunchKernelGGL((k_store<1, 1>), dim3(N), dim3(64), lds, 0, out, T, d, 1.0 + rep, (const double*)y);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("waves/SIMD %d %s pattern %s: %.3f ms = %.2f TB/s\n", wps, yl ? "with a dependent 160-byte load per step" : "stores only", pat ? "linear (13 x 1 KB)" : "tiles  (18 x 64 pieces)", best, bytes / (best * 1e-3) / 1e12);
    }
  }
  hipFree(out);
  return 0;
}
