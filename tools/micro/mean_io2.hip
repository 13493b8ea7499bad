// Microbenchmark 2: variants of k_mean_rts16's traffic (profiles/r04_notes.md section 9).  10 000 series x 1001 records of 1456 B written by 2500 waves of four series;
//   means:  0 none | 1 the 128-byte heads of the filter records (4 scattered pieces per step) | 2 a compact stream (512 contiguous bytes per wave and step)
//   table:  0 none | 1 one 1456-byte row per wave and step by LDS DMA | 2 one row per workgroup of four waves (s_barrier per step)
//   stores: cache-policy bits of the record stores (aux of raw_buffer_store: 0 default, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1, 3 sc0 nt, 19 all)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4 rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i4 r = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)), __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  return r;
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma(const i4& rs, unsigned lds_addr, int voff, int soff, bool on) {
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  if (on) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int MEANS, int TABLE, int AUX, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_io(double* out, const double* tab, const double* filt, const double* cmp, int T, int N) {
  __shared__ __attribute__((aligned(16))) double lds[WPB * (2 * 256 + 8 * 64)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n0 = 4 * (blockIdx.x * WPB + wave);
  if (n0 >= N) return;
  const int rec = 182, recb = rec * 8, npc = rec / 2;
  const size_t sbytes = (size_t)(T + 1) * recb;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((char*)out + (size_t)n0 * sbytes, 0, (int)(4 * sbytes), 0x00020000);
  const i4 rtab = rsrc_words(tab, (unsigned)sbytes);
  const i4 rmean = MEANS == 2 ? rsrc_words((const char*)cmp + (size_t)(n0 / 4) * (T + 1) * 512, (unsigned)((size_t)(T + 1) * 512)) : rsrc_words((const char*)filt + (size_t)n0 * sbytes, (unsigned)(4 * sbytes));
  double* my = lds + (TABLE == 2 ? 0 : wave * (2 * 256 + 8 * 64));
  double* mym = lds + WPB * 2 * 256 + wave * 8 * 64;
  const unsigned ldst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my, ldsm = (unsigned)(size_t)(__attribute__((address_space(3))) char*)mym;
  const int OOB = 0x7ffffff0;
  int pdst[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { const int q = 64 * k + lane, sj = q / npc, pp = q - sj * npc; pdst[k] = sj < 4 ? (int)((size_t)sj * sbytes) + pp * 16 : OOB; }
  const int mvoff = MEANS == 2 ? lane * 16 : (lane < 32 ? (int)((size_t)(lane >> 3) * sbytes) + (lane & 7) * 16 : OOB);
  double v = 1.0 + lane;
  for (int s = 0; s <= T; ++s) {
    const int t = T - s;
    if (MEANS) dma(rmean, ldsm + (s & 7) * 512, mvoff, t * (MEANS == 2 ? 512 : recb), lane < 32);
    if (TABLE == 1 || (TABLE == 2 && wave == 0)) { dma(rtab, ldst + (s & 1) * 2048, lane * 16, t * recb, true); dma(rtab, ldst + (s & 1) * 2048 + 1024, lane * 16 + 1024, t * recb, lane + 64 < npc); }
    constexpr int PER = 6 + (MEANS ? 1 : 0) + (TABLE ? 2 : 0);
    if ((MEANS || TABLE) && !(AUX & 256)) vm_wait<PER + 6>();
    if (TABLE == 2) __builtin_amdgcn_s_barrier();
    if ((MEANS || TABLE) && !(AUX & 256)) v += my[(s & 1) * 256 + lane] * 1e-9 + mym[(s & 7) * 64 + lane] * 1e-9;
    const int so = t * recb;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const u4 w = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v), (unsigned)k, (unsigned)lane};
      __builtin_amdgcn_raw_buffer_store_b128(w, rout, pdst[k], so, AUX & 255);
    }
    v += 1e-3;
  }
  vm_wait<0>();
}
int main() {
  const int N = 10000, T = 1000, rec = 182;
  const size_t bytes = (size_t)N * (T + 1) * rec * 8;
  double *out, *filt, *tab, *cmp;
  if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&filt, bytes) != hipSuccess || hipMalloc(&tab, (size_t)(T + 1) * rec * 8) != hipSuccess || hipMalloc(&cmp, (size_t)(N / 4) * (T + 1) * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(filt, 0, bytes); (void)hipMemset(tab, 0, (size_t)(T + 1) * rec * 8); (void)hipMemset(cmp, 0, (size_t)(N / 4) * (T + 1) * 512);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("%-78s %.3f ms\n", name, best);
  };
#define L(M, TB, AUX, WPB) [&]() { hipLaunchKernelGGL((k_io<M, TB, AUX, WPB>), dim3((N / 4 + WPB - 1) / WPB), dim3(64 * WPB), 0, 0, out, (const double*)tab, (const double*)filt, (const double*)cmp, T, N); }
  run("stores only", L(0, 0, 0, 1));
  run("scattered means + row per wave (k_mean_rts16 today)", L(1, 1, 0, 1));
  run("compact means + row per wave", L(2, 1, 0, 1));
  run("compact means, no row", L(2, 0, 0, 1));
  run("scattered means, no row", L(1, 0, 0, 1));
  run("no means, row per wave", L(0, 1, 0, 1));
  run("no means, row per workgroup of 4 waves (barrier per step)", L(0, 2, 0, 4));
  run("compact means + row per workgroup of 4 waves", L(2, 2, 0, 4));
  run("scattered means + row per wave, stores nt", L(1, 1, 2, 1));
  run("scattered means + row per wave, stores sc1", L(1, 1, 16, 1));
  run("scattered means + row per wave, stores sc0 sc1", L(1, 1, 17, 1));
  run("scattered means + row per wave, stores sc0 sc1 nt", L(1, 1, 19, 1));
  run("compact means + row per wave, stores nt", L(2, 1, 2, 1));
  run("stores only, nt", L(0, 0, 2, 1));
  run("compact means requested but never waited for", L(2, 0, 256, 1));
  run("scattered means + row requested but never waited for", L(1, 1, 256, 1));
  run("scattered means + row requested but never waited for, stores nt", L(1, 1, 258, 1));
  return 0;
}
