// What HBM rate can the record access patterns sustain, with no compute at all?
// One wave per series (10 000 series x 1001 records of 1456 B), exactly like the engine:
//  A: 8 B/lane, 4 instructions of 13-of-16 lanes (the std-layout pattern), masked by OOB buffer offsets
//  B: 16 B/lane, 2 fully contiguous instructions per record (64 + 27 lanes)
// modes: write-only (forward pass), read+write (backward pass).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int OOB = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs(const void* p, size_t b) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)b, 0x00020000); }

template <int PAT, int RW, int PF>
__global__ __launch_bounds__(256) void k(const double* in, double* out, int N, int T1) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (n >= N) return;
  const int d = 13, rec = 182, recb = rec * 8;
  const __amdgpu_buffer_rsrc_t rin = rs(in + (size_t)n * T1 * rec, (size_t)T1 * recb);
  const __amdgpu_buffer_rsrc_t rout = rs(out + (size_t)n * T1 * rec, (size_t)T1 * recb);
  const int g = lane >> 4, c = lane & 15;
  int off[5];
  for (int r = 0; r < 4; ++r) { const int i = 4 * r + g; off[r] = (i < d && c < d) ? (d + i * d + c) * 8 : OOB; }
  off[4] = (g == 0 && c < d) ? c * 8 : OOB;
  const int vo0 = lane * 16, vo1 = (1024 + lane * 16 < recb) ? 1024 + lane * 16 : OOB;
  double acc = 0.0;
  if (PAT == 0) {
    u2 v[PF][5];
    for (int q = 0; q < PF; ++q) for (int r = 0; r < 5; ++r) v[q][r] = u2{0u, 0u};
    if (RW) for (int q = 0; q < PF; ++q) for (int r = 0; r < 5; ++r) v[q][r] = __builtin_amdgcn_raw_buffer_load_b64(rin, off[r], (T1 - 1 - q) * recb, 0);
    for (int t = T1 - 1; t >= 0; --t) {
      u2 cur[5];
      for (int r = 0; r < 5; ++r) cur[r] = v[0][r];
      for (int q = 0; q + 1 < PF; ++q) for (int r = 0; r < 5; ++r) v[q][r] = v[q + 1][r];
      if (RW) { const int tp = t - PF >= 0 ? t - PF : 0; for (int r = 0; r < 5; ++r) v[PF - 1][r] = __builtin_amdgcn_raw_buffer_load_b64(rin, off[r], tp * recb, 0); }
      for (int r = 0; r < 5; ++r) { cur[r][0] += (unsigned)t; __builtin_amdgcn_raw_buffer_store_b64(cur[r], rout, off[r], t * recb, 0); }
    }
  } else {
    u4 v[PF][2];
    for (int q = 0; q < PF; ++q) { v[q][0] = u4{0, 0, 0, 0}; v[q][1] = u4{0, 0, 0, 0}; }
    if (RW) for (int q = 0; q < PF; ++q) { v[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rin, vo0, (T1 - 1 - q) * recb, 0); v[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rin, vo1, (T1 - 1 - q) * recb, 0); }
    for (int t = T1 - 1; t >= 0; --t) {
      u4 c0 = v[0][0], c1 = v[0][1];
      for (int q = 0; q + 1 < PF; ++q) { v[q][0] = v[q + 1][0]; v[q][1] = v[q + 1][1]; }
      if (RW) { const int tp = t - PF >= 0 ? t - PF : 0; v[PF - 1][0] = __builtin_amdgcn_raw_buffer_load_b128(rin, vo0, tp * recb, 0); v[PF - 1][1] = __builtin_amdgcn_raw_buffer_load_b128(rin, vo1, tp * recb, 0); }
      c0[0] += (unsigned)t; c1[0] += (unsigned)t;
      __builtin_amdgcn_raw_buffer_store_b128(c0, rout, vo0, t * recb, 0);
      __builtin_amdgcn_raw_buffer_store_b128(c1, rout, vo1, t * recb, 0);
    }
  }
  if (acc == 1.2345) out[0] = acc;
}

template <int PAT, int RW, int PF>
void run(const char* name, const double* in, double* out, int N, int T1) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<PAT, RW, PF>), dim3((N + 3) / 4), dim3(256), 0, 0, in, out, N, T1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, RW, PF>), dim3((N + 3) / 4), dim3(256), 0, 0, in, out, N, T1);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
  const double bytes = (double)N * T1 * 1456.0 * (RW ? 2 : 1);
  printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const int N = 10000, T1 = 1001;
  double *in, *out;
  hipMalloc(&in, (size_t)N * T1 * 1456); hipMalloc(&out, (size_t)N * T1 * 1456);
  hipMemset(in, 0, (size_t)N * T1 * 1456);
  run<0, 0, 1>("A 8B/lane rows, write-only", in, out, N, T1);
  run<1, 0, 1>("B 16B/lane contiguous, write-only", in, out, N, T1);
  run<0, 1, 1>("A 8B/lane rows, read+write, prefetch 1", in, out, N, T1);
  run<0, 1, 2>("A 8B/lane rows, read+write, prefetch 2", in, out, N, T1);
  run<0, 1, 4>("A 8B/lane rows, read+write, prefetch 4", in, out, N, T1);
  run<1, 1, 1>("B 16B/lane contiguous, read+write, prefetch 1", in, out, N, T1);
  run<1, 1, 2>("B 16B/lane contiguous, read+write, prefetch 2", in, out, N, T1);
  run<1, 1, 4>("B 16B/lane contiguous, read+write, prefetch 4", in, out, N, T1);
  return 0;
}
