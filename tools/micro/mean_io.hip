// Microbenchmark: the memory traffic of k_mean_rts16 without its arithmetic -- 2500 waves of four series, per step the four
// smoothed records (4 x 1456 B) as six 16-byte-per-lane store instructions, optionally the table row by LDS DMA from L2 (1456 B, two steps
// ahead) and the four 128-byte heads of the filter records from HBM (eight steps ahead), with the kernel's counted waits.
//   hipcc --offload-arch=gfx950 -O3 mean_io.hip -o mean_io
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4 rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i4 r = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)),
          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  return r;
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma(const i4& rs, unsigned lds_addr, int voff, int soff, bool on) {
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  if (on) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// MODE 0: stores only; 1: + table row DMA; 2: + means DMA; 3: both
template <int MODE, int DEPTH>
__global__ __launch_bounds__(64, 2) void k_io(double* out, const double* tab, const double* filt, int T, int N) {
  __shared__ __attribute__((aligned(16))) double lds[8 * 256 + 8 * 64];
  constexpr int DESC = 1;
  const int lane = threadIdx.x, n0 = 4 * blockIdx.x;
  const int rec = 182, recb = rec * 8, npc = rec / 2;
  const size_t sbytes = (size_t)(T + 1) * recb;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((char*)out + (size_t)n0 * sbytes, 0, (int)(4 * sbytes), 0x00020000);
  const i4 rtab = rsrc_words(tab, (unsigned)sbytes);
  const i4 rmean = rsrc_words((const char*)filt + (size_t)n0 * sbytes, (unsigned)(4 * sbytes));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int OOB = 0x7ffffff0;
  int pdst[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { const int q = 64 * k + lane, sj = q / npc, pp = q - sj * npc; pdst[k] = sj < 4 ? (int)((size_t)sj * sbytes) + pp * 16 : OOB; }
  const int mvoff = lane < 32 ? (int)((size_t)(lane >> 3) * sbytes) + (lane & 7) * 16 : OOB;
  double v = 1.0 + lane;
  for (int s = 0; s <= T; ++s) {
    const int t = DESC ? T - s : s;
    // request the inputs of step s + DEPTH - 1 (the first DEPTH - 1 steps' inputs are simply not waited for: same traffic, same counts)
    const int ta = DESC ? (t - (DEPTH - 1) > 0 ? t - (DEPTH - 1) : 0) : t;
    if (MODE & 2) dma(rmean, lds0 + 16384 + (s & 7) * 512, mvoff, ta * recb, lane < 32);
    if (MODE & 1) { dma(rtab, lds0 + (s & 7) * 2048, lane * 16, ta * recb, true); dma(rtab, lds0 + (s & 7) * 2048 + 1024, lane * 16 + 1024, ta * recb, lane + 64 < npc); }
    // operations younger than the request issued DEPTH - 1 steps ago: (DEPTH - 1) steps of (6 stores + this mode's requests), minus nothing: that request's own
    // younger requests of its step are counted too (conservative by at most 2)
    constexpr int PER = 6 + ((MODE & 2) ? 1 : 0) + ((MODE & 1) ? 2 : 0);
    constexpr int W = (DEPTH - 1) * PER + 6 > 63 ? 63 : (DEPTH - 1) * PER + 6;
    if (MODE) vm_wait<W>();
    if (MODE) v += lds[(s & 7) * 256 + lane] * 1e-9;
    const int so = t * recb;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const u4 w = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v), (unsigned)k, (unsigned)lane};
      __builtin_amdgcn_raw_buffer_store_b128(w, rout, pdst[k], so, 0);
    }
    v += 1e-3;
  }
  vm_wait<0>();
}
int main() {
  const int N = 10000, T = 1000, rec = 182;
  const size_t bytes = (size_t)N * (T + 1) * rec * 8;
  double *out, *filt, *tab;
  if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&filt, bytes) != hipSuccess || hipMalloc(&tab, (size_t)(T + 1) * rec * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(filt, 0, bytes); (void)hipMemset(tab, 0, (size_t)(T + 1) * rec * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[4] = {"stores only", "+ table row from L2 (LDS DMA)", "+ 4 x 128 B means from HBM (LDS DMA)", "+ both"};
  for (int depth = 1; depth <= 7; ++depth)
  for (int mode = 3; mode < 4; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
#define L(M, D) hipLaunchKernelGGL((k_io<M, D>), dim3(N / 4), dim3(64), 0, 0, out, (const double*)tab, (const double*)filt, T, N)
      switch (depth) { case 1: L(3, 1); break; case 2: L(3, 2); break; case 3: L(3, 3); break; case 4: L(3, 4); break; case 5: L(3, 5); break; case 6: L(3, 6); break; default: L(3, 7); }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("inputs requested %d step(s) ahead, %s: %.3f ms = %.2f TB/s written\n", depth - 1, names[mode], best, bytes / (best * 1e-3) / 1e12);
  }
  return 0;
}
