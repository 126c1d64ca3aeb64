// What does a transform wave get next to a v_mfma_f32_16x16x4_f32 stream on the same SIMD?  (r05, sr_wino4.hip's wave-specialised
// form: the T waves take 6.4 k clocks per slab next to the M waves' MFMAs and 1.5 k alone.)  One 8-wave workgroup per CU: waves
// 0-3 stream MFMAs (two accumulators interleaved, like the kernel), waves 4-7 a stream of ONE kind of instruction.  Reported:
// clocks per MFMA of the M waves, clocks per instruction of the other waves -- alone and together -- for 16x16x4 and 32x32x2.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma16_overlap mfma16_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND, int MF>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, int roleLo, int roleHi, int itA, int itB, float seed, int prio) {
  __shared__ float lds[8192];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = wave < 4 ? roleLo : roleHi;
  f4 c0 = {seed, 0, 0, 0}, c1 = {0, seed, 0, 0}, c2 = {0, 0, seed, 0}, c3 = {0, 0, 0, seed};
  f16v d0 = {}, d1 = {};
  float v[8] = {seed, seed + 1, seed + 2, seed + 3, seed + 4, seed + 5, seed + 6, seed + 7};
  f2 p[4] = {{seed, 1}, {seed, 2}, {seed, 3}, {seed, 4}};
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = seed;
  __syncthreads();
  const unsigned la = (threadIdx.x & 63) * 4u;
  if (prio && role == 2) __builtin_amdgcn_s_setprio(3);
  if (prio == 2 && role == 1) __builtin_amdgcn_s_setprio(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 1) {
    for (int it = 0; it < itA; ++it) {
      if (MF == 16) { REP16(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c1, 0, 0, 0);) }
      else if (MF == 165) { REP16(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c0, 0, 0, 0); asm volatile("s_nop 6");
                                  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c1, 0, 0, 0); asm volatile("s_nop 6");) }
      else if (MF == 166) { REP16(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c0, 0, 0, 0); asm volatile("s_nop 4");
                                  c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c1, 0, 0, 0); asm volatile("s_nop 4");) }
      else if (MF == 164) { REP4(REP4(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c1, 0, 0, 0);
                                      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c3, 0, 0, 0);)
                                 REP4(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c1, 0, 0, 0);
                                      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c3, 0, 0, 0);)) }
      else { REP16(d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, d1, 0, 0, 0);) }
    }
  } else if (role == 2) {
    for (int it = 0; it < itB; ++it) {
      if (KIND == 0) { REP4(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));) }
      else if (KIND == 1) { REP4(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                                              "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3"
                                              : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));) }
      else if (KIND == 2) { float r0, r1, r2, r3, r4, r5, r6, r7;
        REP4(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                          "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(la) : "memory"); v[0] += r0 + r7;) }
      else if (KIND == 3) { REP4(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                                              "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:1280\n ds_write_b32 %0, %1 offset:1536\n ds_write_b32 %0, %1 offset:1792\n s_waitcnt lgkmcnt(0)"
                                              : : "v"(la), "v"(v[0]) : "memory");) }
      else if (KIND == 4) { float r0, r1, r2, r3, r4, r5, r6, r7;   // ds_read2_b32: 4 instructions = 8 dwords
        REP4(asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:20\n ds_read2_b32 %1, %4 offset0:40 offset1:60\n ds_read2_b32 %2, %4 offset0:80 offset1:100\n ds_read2_b32 %3, %4 offset0:120 offset1:140\n s_waitcnt lgkmcnt(0)"
                          : "=v"(*(f2*)&r0), "=v"(*(f2*)&r2), "=v"(*(f2*)&r4), "=v"(*(f2*)&r6) : "v"(la) : "memory"); v[0] += r0 + r7;
             asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:20\n ds_read2_b32 %1, %4 offset0:40 offset1:60\n ds_read2_b32 %2, %4 offset0:80 offset1:100\n ds_read2_b32 %3, %4 offset0:120 offset1:140\n s_waitcnt lgkmcnt(0)"
                          : "=v"(*(f2*)&r0), "=v"(*(f2*)&r2), "=v"(*(f2*)&r4), "=v"(*(f2*)&r6) : "v"(la) : "memory"); v[1] += r1 + r6;) }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + p[0][0] + p[1][1] + p[2][0] + p[3][1];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) | ((unsigned long long)(s == 12345.f) << 62);
}

static void report(const char* what, unsigned long long* d, double nA, double nB) {
  hipDeviceSynchronize();
  static unsigned long long h[8 * 256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double lo = 0, hi = 0;
  for (int b = 0; b < 256; ++b) {
    for (int w = 0; w < 4; ++w) lo += (double)(h[b * 8 + w] & 0xffffffffffffULL);
    for (int w = 4; w < 8; ++w) hi += (double)(h[b * 8 + w] & 0xffffffffffffULL);
  }
  lo /= 1024; hi /= 1024;
  printf("%-58s", what);
  if (nA > 0) printf(" M waves %6.1f clk/mfma", lo / nA);
  if (nB > 0) printf("   other waves %6.2f clk/instr", hi / nB);
  printf("\n");
}

template <int KIND, int MF>
static void run(const char* kname, unsigned long long* d) {
  const int itA = 64, itB = 512;
  const double nA = itA * 32.0, nB = itB * 32.0;
  char buf[160];
  hipLaunchKernelGGL((k<KIND, MF>), dim3(256), dim3(512), 0, 0, d, 0, 2, itA, itB, 1.0f, 0);
  snprintf(buf, sizeof buf, "[%s] alone", kname); report(buf, d, 0, nB);
  const double mf = MF == 164 ? 128.0 : 32.0;
  for (int prio = 0; prio < 2; ++prio) {
    hipLaunchKernelGGL((k<KIND, MF>), dim3(256), dim3(512), 0, 0, d, 1, 2, itA * 8, itB, 1.0f, prio);
    snprintf(buf, sizeof buf, "[%s] next to a %s stream%s", kname, MF == 16 ? "16x16x4 (2 acc)" : MF == 164 ? "16x16x4 (4 acc)" : MF == 165 ? "16x16x4 + s_nop 6" : MF == 166 ? "16x16x4 + s_nop 4" : "32x32x2 (2 acc)",
             prio ? ", s_setprio 3" : ""); report(buf, d, itA * 8 * mf, nB);
  }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 8 * 256 * 8);
  hipLaunchKernelGGL((k<0, 16>), dim3(256), dim3(512), 0, 0, d, 1, 0, 64, 0, 1.0f, 0); report("16x16x4 MFMA stream alone", d, 64 * 32.0, 0);
  hipLaunchKernelGGL((k<0, 32>), dim3(256), dim3(512), 0, 0, d, 1, 0, 64, 0, 1.0f, 0); report("32x32x2 MFMA stream alone", d, 64 * 32.0, 0);
  run<0, 16>("v_fma_f32", d); run<0, 165>("v_fma_f32", d); run<0, 166>("v_fma_f32", d); run<2, 165>("ds_read_b32 x8 + wait", d); run<0, 164>("v_fma_f32", d); run<0, 32>("v_fma_f32", d);
  run<2, 164>("ds_read_b32 x8 + wait", d); run<3, 164>("ds_write_b32 x8 + wait", d); run<4, 164>("ds_read2_b32 x4 + wait", d);
  run<1, 16>("v_pk_fma_f32", d); run<1, 32>("v_pk_fma_f32", d);
  run<2, 16>("ds_read_b32 x8 + wait", d); run<2, 32>("ds_read_b32 x8 + wait", d);
  run<3, 16>("ds_write_b32 x8 + wait", d); run<3, 32>("ds_write_b32 x8 + wait", d);
  run<4, 16>("ds_read2_b32 x4 + wait", d); run<4, 32>("ds_read2_b32 x4 + wait", d);
  return 0;
}
