// Which record LAYOUT lets the two record streams of the C2 pass run closer to the copy rate?
// tools/io_pattern_bench.hip showed that one wave per series on series-major records sustains 5.5 TB/s write-only and
// 4.85 TB/s read+write whatever the access width; a plain copy reaches 6.3.  Candidates for the gap: 10 000 concurrent
// streams, each on its own DRAM rows / TLB pages.  This bench keeps the engine's schedule (one wave per series, one
// 1456-byte record per step, backward in time) and varies only where record (n, t) lives:
//   L0  series-major      [N][T1][rec]              (the engine's layout)
//   L1  time-major        [T1][N][rec]
//   L2  blocked           [N/B][T1][B][rec], B = 64 or 256 series (neighbouring workgroups)
//   L3  series-major with records padded to 1536 B (whole 128-byte lines)
//   C   a plain 16 B/lane grid-stride copy of the same bytes (the ceiling on THIS box)
// modes: write-only (forward pass), read+write (backward pass).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int LAY, int RW, int B>
__global__ __launch_bounds__(256) void k(const char* in, char* out, int N, int T1, int recb, int padb) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (n >= N) return;
  const int vo0 = lane * 16, vo1 = 1024 + lane * 16;
  const bool a0 = vo0 < recb, a1 = vo1 < recb;
  u4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  for (int t = T1 - 1; t >= 0; --t) {
    size_t off;
    if (LAY == 0) off = ((size_t)n * T1 + t) * recb;
    else if (LAY == 1) off = ((size_t)t * N + n) * recb;
    else if (LAY == 2) off = (((size_t)(n / B) * T1 + t) * B + (n % B)) * recb;
    else off = ((size_t)n * T1 + t) * padb;
    off = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)off) |
          ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(off >> 32)) << 32);   // wave-uniform: scalar address
    if (RW) {
      if (a0) c0 = *(const u4*)(in + off + vo0);
      if (a1) c1 = *(const u4*)(in + off + vo1);
    }
    c0[0] += (unsigned)t; c1[0] += (unsigned)t;
    if (a0) *(u4*)(out + off + vo0) = c0;
    if (a1) *(u4*)(out + off + vo1) = c1;
  }
}

template <int RW>
__global__ __launch_bounds__(256) void kcopy(const u4* in, u4* out, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    u4 v = {1, 2, 3, 4};
    if (RW) v = in[i];
    v[0] += 1u;
    out[i] = v;
  }
}

template <int LAY, int RW, int B>
void run(const char* name, const char* in, char* out, int N, int T1) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<LAY, RW, B>), dim3((N + 3) / 4), dim3(256), 0, 0, in, out, N, T1, 1456, 1536);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LAY, RW, B>), dim3((N + 3) / 4), dim3(256), 0, 0, in, out, N, T1, 1456, 1536);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
  const double bytes = (double)N * T1 * 1456.0 * (RW ? 2 : 1);
  printf("%-52s N=%5d %7.3f ms  %6.2f TB/s\n", name, N, ms, bytes / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

template <int RW>
void run_copy(const char* name, const char* in, char* out, size_t bytes1) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const size_t n16 = bytes1 / 16;
  hipLaunchKernelGGL((kcopy<RW>), dim3(256 * 32), dim3(256), 0, 0, (const u4*)in, (u4*)out, n16);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kcopy<RW>), dim3(256 * 32), dim3(256), 0, 0, (const u4*)in, (u4*)out, n16);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
  const double bytes = (double)bytes1 * (RW ? 2 : 1);
  printf("%-52s         %7.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  const int N = 10000, T1 = 1001;
  char *in, *out;
  const size_t bytes = (size_t)N * T1 * 1536;
  if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(in, 0, bytes); hipMemset(out, 0, bytes);
  run_copy<0>("C  plain 16 B/lane fill (write-only)", in, out, (size_t)N * T1 * 1456);
  run_copy<1>("C  plain 16 B/lane copy (read+write)", in, out, (size_t)N * T1 * 1456);
  run<0, 0, 1>("L0 series-major, write-only", in, out, N, T1);
  run<1, 0, 1>("L1 time-major, write-only", in, out, N, T1);
  run<2, 0, 64>("L2 blocked B=64, write-only", in, out, N, T1);
  run<2, 0, 256>("L2 blocked B=256, write-only", in, out, N, T1);
  run<3, 0, 1>("L3 series-major padded 1536, write-only", in, out, N, T1);
  run<0, 1, 1>("L0 series-major, read+write", in, out, N, T1);
  run<1, 1, 1>("L1 time-major, read+write", in, out, N, T1);
  run<2, 1, 64>("L2 blocked B=64, read+write", in, out, N, T1);
  run<2, 1, 256>("L2 blocked B=256, read+write", in, out, N, T1);
  run<3, 1, 1>("L3 series-major padded 1536, read+write", in, out, N, T1);
  for (int n : {5000, 2500, 1250}) {
    run<0, 1, 1>("L0 series-major, read+write", in, out, n, T1);
    run<1, 1, 1>("L1 time-major, read+write", in, out, n, T1);
  }
  return 0;
}
