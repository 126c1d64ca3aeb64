// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of plain / packed fp32 ops at 1, 2, 4 waves
// per SIMD (s_memtime around an unrolled, dependency-free stream).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(unsigned long long* out, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed};
  f2 p4 = {seed + 4, seed}, p5 = {seed + 5, seed}, p6 = {seed + 6, seed}, p7 = {seed + 7, seed};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; ++it) {
    if (OP == 0) {
      REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (OP == 1) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                         "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (OP == 2) {
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n"
                         "v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %5, %5, %5\n v_pk_mul_f32 %6, %6, %6\n v_pk_mul_f32 %7, %7, %7"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (OP == 3) {
      REP16(asm volatile("v_mul_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                         "v_mul_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (OP == 4) {
      REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
    } else if (OP == 5) {
      REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (OP == 6) {
      REP16(asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n"
                         "v_pk_add_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (OP == 7) {
      REP16(asm volatile("v_floor_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_floor_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n"
                         "v_floor_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_floor_f32 %6, %6\n v_cvt_i32_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y + p7.y;
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) | ((unsigned long long)(s == 12345.f) << 63);
}

template <int OP>
void run(const char* name, unsigned long long* d) {
  for (int waves = 4; waves <= 16; waves *= 2) {   // waves per workgroup = waves per CU (1 WG per CU): 1, 2, 4 per SIMD
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(256), dim3(64 * waves), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = (double)(h[0] & 0x7FFFFFFFFFFFFFFFull);
    const double ninst = 64.0 * 16 * 8;
    printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction (per wave), %.2f per SIMD-issue\n", name, waves / 4,
           cyc / ninst, cyc / ninst / (waves / 4));
  }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 1 << 20);
  run<0>("v_fma_f32", d);
  run<1>("v_pk_fma_f32", d);
  run<2>("v_pk_mul_f32", d);
  run<6>("v_pk_add_f32", d);
  run<3>("v_mul_f32/v_add_f32", d);
  run<4>("v_cmp+v_cndmask", d);
  run<5>("v_rcp_f32", d);
  run<7>("v_floor/v_cvt_i32", d);
  return 0;
}
