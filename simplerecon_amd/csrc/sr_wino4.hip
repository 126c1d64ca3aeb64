// sr_wino4.hip -- 3x3 / stride-1 convolutions through Winograd F(4x4, 3x3) on the fp32 matrix cores (gfx950), r05; the
// wave-specialised form every F(4x4) launch takes since r06 is described at sr_wino4ws_kernel below.
//
// Same operator as sr_conv3x3_wino_nhwc_fwd (conv + bias + residual + LeakyReLU of the reference's BasicBlock,
// modules/layers.py:24-85), same arithmetic class (fp32 products, fp32 accumulation).  F(4x4, 3x3) needs 36 multiplies per
// 4x4 output tile and (input, output) channel pair -- 2.25 per output instead of F(2x2)'s 4 and the direct algorithm's 9 --
// so the full-resolution 64-channel layers of the UNet++ decoder (179 of the conv stack's 327 GFLOP per frame, SURVEY.md
// appendix B) issue 1.78x fewer MFMAs than sr_wino_kernel.  Interpolation points (0, +-1/2, +-2, inf): every entry of B^T and
// A^T is a dyadic rational (exact in fp32); measured fp32 error 1.3e-6 of the output range on a 64-channel layer (points
// 0, +-1, +-2: 2.4e-6; F(2x2): 3e-7), far inside the 1e-4 parity bar -- tests/test_gpu_wino4.py holds it against fp64.
//
// Work item = 4 x 4 Winograd tiles (16 x 16 output pixels) x 64 output channels; a workgroup walks items persistently.  In the
// first form (sr_wino4_kernel: 4 waves, two workgroups per CU; kept for A/B and as the bit-identity reference), per 16-channel
// slab of the input:
//   S  the 18 x 18 pixel patch goes global -> registers -> LDS (`raw`, 20 floats per pixel); the loads of slab s + 1 are
//      issued before the transform of slab s and have a whole slab of lead,
//   T  thread (tile, ci) applies V = B^T d B to its 6 x 6 patch: 36 ds_read_b32, 144 VALU, 36 ds_write_b32 into
//      V[36 frequencies][16 tiles][16 ci] (the channel quads of a tile row XOR-swizzled so that the MFMA phase's
//      ds_read_b128 are conflict-free),
//   M  wave w owns output channels [16 w, 16 w + 16) for ALL 36 frequencies: v_mfma_f32_16x16x4_f32 with
//      A = U_f[co][ci] (one 16-byte load per lane and frequency from the packed weights, L2-resident) and
//      B = V_f[ci][tile] (one ds_read_b128 per lane and frequency), D[co][tile] -- 144 accumulator registers.
// Because a lane holds all 36 frequencies of (4 consecutive output channels) x (one tile), the output transform
// Y = A^T M A runs entirely in registers, IN PLACE over the accumulators (no LDS exchange, no barrier): a lane ends up with
// 16 pixels x 4 channels = 16 float4: bias + residual + activation, 16-byte stores into the consumer's concat slice.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "sr_common.h"

namespace {

typedef float w4_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int w4_u4 __attribute__((ext_vector_type(4)));
#define W4_RSRC_FLAGS 0x00020000
#define W4_OOB 0x7fffffffu

constexpr int W4_PS = 18;                                // patch side: 16 output pixels + 2
constexpr int W4_RS = 20;                                // floats per staged pixel (16 channels + 4 pad: tiles 4 px apart -> other banks)
constexpr int W4_RAW_FLOATS = W4_PS * W4_PS * W4_RS;     // 6480
constexpr int W4_V_FLOATS = 36 * 16 * 16;                // 9216
constexpr int W4_LDS_BYTES = (W4_RAW_FLOATS + W4_V_FLOATS) * 4;   // 62 784: two workgroups per CU
constexpr int W4_STAGE = 6;                              // float4 loads per thread and slab (1296 of 1536 slots)
#ifndef SR_W4_NA
#define SR_W4_NA 3
#define SR_W4_PD 2
#endif
constexpr int W4_NA = SR_W4_NA, W4_PD = SR_W4_PD;
// Issue priority (s_setprio) of the non-MFMA phases.  scripts/micro/mfma16_overlap.hip: next to a wave that streams
// v_mfma_f32_16x16x4_f32 back to back, a second wave on the same SIMD gets ONE instruction issued per MFMA (37.6 clk per
// v_fma_f32 -- or per v_pk_fma_f32 -- instead of 5.7) at equal priority: the arbiter serves the MFMA wave first although its
// next MFMA cannot start before the matrix pipe is free.  With the transform / staging / epilogue code at a higher priority
// those instructions issue when they are ready and the MFMAs take the slots in between (one per 32 clocks is all they need).
#ifndef SR_W4_PRIO
#define SR_W4_PRIO 2
#endif
__device__ __forceinline__ void w4_prio_other() { if (SR_W4_PRIO) __builtin_amdgcn_s_setprio(SR_W4_PRIO); }
__device__ __forceinline__ void w4_prio_mfma() { if (SR_W4_PRIO) __builtin_amdgcn_s_setprio(0); }                      // weight-fragment register sets / prefetch distance (frequency pairs)

// Phase ablations (timing experiments only, results are wrong): -DSR_W4_ABL=<bits>  1: no transform, 2: no MFMA phase,
// 4: no epilogue, 8: every weight fragment from one cached address, 16: no staging loads / stores, 32: no operand loads in
// the MFMA phase (bare MFMAs).  0 in the product build.
#ifndef SR_W4_ABL
#define SR_W4_ABL 0
#endif

struct SrWino4Params {
  const float* in; int64_t in_sb; int in_sp;
  const float* wu;                   // packed U: [36][S][4 kq][Co_pad][4]
  const float* bias;
  const float* res; int64_t res_sb; int res_sp;
  float* out; int64_t out_sb; int out_sp;
  int B, H, W, Cin, Cout, Co_pad, S;
  int regions_x, regions_y, co_blocks, total;
  float slope;
#ifdef SR_W4_TRACE
  unsigned long long* trace;   // [blocks][2 groups][W4_TR_N] shader-clock stamps (debug builds only)
#endif
};
#ifdef SR_W4_TRACE
#define W4_TR_N 128   // the last 4 slots of group 0: shader clock / 100-MHz wall clock at the start and the end of the workgroup
// (the scheduling barriers keep the compiler from moving arithmetic across the clock read)
#define W4_TR(code)                                                                                              \
  do {                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    const unsigned long long tr_now = clock64();                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    if (tid == 0 && tr_n < W4_TR_N - 4 && blockIdx.x < 16)                                                          \
      p.trace[((size_t)blockIdx.x * 2 + grp) * W4_TR_N + tr_n++] = ((unsigned long long)(code) << 56) | (tr_now & 0xffffffffffffffull); \
  } while (0)
#else
#define W4_TR(code) do {} while (0)
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, W4_RSRC_FLAGS);
}
__device__ __forceinline__ w4_f4 w4_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(w4_f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// cache-policy probes for the wave-specialised kernel's streams (gfx940+ aux bits: 1 = sc0, 2 = nt, 16 = sc1); product: 0
#ifndef SR_W4WS_DEPHASE_UNIT
#define SR_W4WS_DEPHASE_UNIT 1
#endif
#ifndef SR_W4WS_AUX_PATCH
#define SR_W4WS_AUX_PATCH 0
#endif
#ifndef SR_W4WS_AUX_RES
#define SR_W4WS_AUX_RES 0
#endif
#ifndef SR_W4WS_AUX_OUT
#define SR_W4WS_AUX_OUT 0
#endif
template <int AUX>
__device__ __forceinline__ w4_f4 w4_load_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(w4_f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void w4_store_aux(w4_f4 v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w4_u4, v), r, (int)voff, 0, AUX);
}
// (the delta rides in the LANE offset: a 16-byte buffer store with an SGPR offset operand followed by a VALU write to its
// data registers stores the new contents on gfx950 -- sr_wino.h, r04)
__device__ __forceinline__ void w4_store(w4_f4 v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w4_u4, v), r, (int)voff, 0, 0);
}

// B^T for the points (0, 1/2, -1/2, 2, -2, inf):
//   [1 0 -17/4 0 1 0; 0 -2 -4 1/2 1 0; 0 2 -4 -1/2 1 0; 0 -1/2 -1/4 2 1 0; 0 1/2 -1/4 -2 1 0; 0 1 0 -17/4 0 1]
__device__ __forceinline__ void w4_bt(float d0, float d1, float d2, float d3, float d4, float d5, float& t0, float& t1,
                                      float& t2, float& t3, float& t4, float& t5) {
  const float a = fmaf(-4.0f, d2, d4), b = fmaf(-4.0f, d1, d3);
  const float c = fmaf(-0.25f, d2, d4), e = fmaf(-0.25f, d1, d3);
  t0 = fmaf(-4.25f, d2, d0) + d4;
  t1 = fmaf(0.5f, b, a);
  t2 = fmaf(-0.5f, b, a);
  t3 = fmaf(2.0f, e, c);
  t4 = fmaf(-2.0f, e, c);
  t5 = fmaf(-4.25f, d3, d1) + d5;
}
// A^T = [1 1 1 1 1 0; 0 1/2 -1/2 2 -2 0; 0 1/4 1/4 4 4 0; 0 1/8 -1/8 8 -8 1] on four channels at once
__device__ __forceinline__ void w4_at(w4_f4 m0, w4_f4 m1, w4_f4 m2, w4_f4 m3, w4_f4 m4, w4_f4 m5, w4_f4& s0, w4_f4& s1,
                                      w4_f4& s2, w4_f4& s3) {
  const w4_f4 p = m1 + m2, q = m1 - m2, u = m3 + m4, v = m3 - m4;
  s0 = (m0 + p) + u;
  s1 = 0.5f * q + 2.0f * v;
  s2 = 0.25f * p + 4.0f * u;
  s3 = 0.125f * q + (8.0f * v + m5);
}

struct W4Item { int b, oy0, ox0, co0; };
__device__ __forceinline__ W4Item w4_decode(const SrWino4Params& p, int work) {
  W4Item it;
  int rem = work;
  const int rx = rem % p.regions_x; rem /= p.regions_x;
  const int ry = rem % p.regions_y; rem /= p.regions_y;
  it.b = rem % p.B;
  it.co0 = (rem / p.B) * 64;
  it.oy0 = ry * 16;
  it.ox0 = rx * 16;
  return it;
}

// `off` if `ok`, else an out-of-range offset (the load returns 0, the store is dropped).  The empty asm materialises the
// offset first: left alone the compiler turns the select into a branch around the address arithmetic (and then waits for
// every load in flight at each of those branches).
__device__ __forceinline__ unsigned w4_sel(bool ok, unsigned off) {
  asm volatile("" : "+v"(off));
  return ok ? off : W4_OOB;
}

// loads of slab `s` of the patch of item `it` into st[]: out-of-image pixels and channels past Cin read 0
__device__ __forceinline__ void w4_stage(const SrWino4Params& p, const W4Item& it, int s, int st_q, int st_pp0,
                                         w4_f4 (&st)[W4_STAGE]) {
  const unsigned in_img_bytes = (unsigned)(((int64_t)(p.H * p.W - 1) * p.in_sp + p.Cin) * 4);
  const __amdgpu_buffer_rsrc_t rs_in = w4_rsrc(p.in + (int64_t)it.b * p.in_sb, in_img_bytes);
  const bool chan_ok = 16 * s + 4 * st_q + 4 <= p.Cin;
  const bool interior = (it.oy0 >= 1) & (it.ox0 >= 1) & (it.oy0 + 17 <= p.H) & (it.ox0 + 17 <= p.W);   // uniform
  const unsigned q_off = (unsigned)(16 * s + 4 * st_q) * 4u;
  if (interior) {
    const unsigned base = (unsigned)(((it.oy0 - 1) * p.W + (it.ox0 - 1)) * p.in_sp) * 4u;   // scalar offset operand
#pragma unroll
    for (int j = 0; j < W4_STAGE; ++j) {
      const int pp = st_pp0 + 64 * j;
      const int r = (pp * 3641) >> 16, c = pp - 18 * r;   // (recomputed per slab: six live registers less)
      const bool ok = j < W4_STAGE - 1 ? chan_ok : (chan_ok & (pp < W4_PS * W4_PS));
      st[j] = w4_load(rs_in, w4_sel(ok, (unsigned)((r * p.W + c) * p.in_sp) * 4u + q_off), base);
    }
  } else {
#pragma unroll
    for (int j = 0; j < W4_STAGE; ++j) {
      const int pp = st_pp0 + 64 * j;
      const int r = (pp * 3641) >> 16, c = pp - 18 * r;
      const int y = it.oy0 - 1 + r, x = it.ox0 - 1 + c;
      const bool ok = chan_ok & (pp < W4_PS * W4_PS) & (y >= 0) & (y < p.H) & (x >= 0) & (x < p.W);
      st[j] = w4_load(rs_in, w4_sel(ok, (unsigned)((y * p.W + x) * p.in_sp) * 4u + q_off), 0u);
    }
  }
}

// epilogue of one item: Y = A^T M A in place over the accumulators (acc[6 i + j][r] = M[i][j] of output channel
// cq + r and this lane's tile), then per output row: + bias + residual, activation, 16-byte stores.  FULL: every pixel of
// the 16 x 16 region is inside the image (no per-lane predicate).
template <bool FULL, bool RES>
__device__ __forceinline__ void w4_epilogue(const SrWino4Params& p, const W4Item& it, w4_f4 (&acc)[36], int cq, int m_j) {
#pragma unroll
  for (int j = 0; j < 6; ++j)
    w4_at(acc[j], acc[6 + j], acc[12 + j], acc[18 + j], acc[24 + j], acc[30 + j], acc[j], acc[6 + j], acc[12 + j],
          acc[18 + j]);
  const int oy = it.oy0 + 4 * (m_j >> 2), ox = it.ox0 + 4 * (m_j & 3);
  const bool cq_ok = cq < p.Cout;
  const unsigned out_img_bytes = (unsigned)(((int64_t)(p.H * p.W - 1) * p.out_sp + p.Cout) * 4);
  const __amdgpu_buffer_rsrc_t rs_out = w4_rsrc(p.out + (int64_t)it.b * p.out_sb, out_img_bytes);
  const unsigned res_img_bytes = RES ? (unsigned)(((int64_t)(p.H * p.W - 1) * p.res_sp + p.Cout) * 4) : 0u;
  const __amdgpu_buffer_rsrc_t rs_res = w4_rsrc(RES ? p.res + (int64_t)it.b * p.res_sb : p.wu, res_img_bytes);
  // lane offsets of the tile's first pixel; a lane whose channel quad is past Cout is switched off for good
  const unsigned o0 = w4_sel(cq_ok, (unsigned)((oy * p.W + ox) * p.out_sp + cq) * 4u);
  const unsigned r0 = RES ? w4_sel(cq_ok, (unsigned)((oy * p.W + ox) * p.res_sp + cq) * 4u) : 0u;
  w4_f4 bv = w4_f4{0.0f, 0.0f, 0.0f, 0.0f};
  if (p.bias) bv = __builtin_bit_cast(w4_f4, __builtin_amdgcn_raw_buffer_load_b128(w4_rsrc(p.bias, (int64_t)p.Cout * 4),
                                                                                   (int)w4_sel(cq_ok, (unsigned)cq * 4u), 0, 0));
  const float slope = sr_uniform(p.slope);
  w4_f4 rv[2][4];
  if (RES) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const bool ok = FULL ? true : ((oy < p.H) & (ox + l < p.W));
      rv[0][l] = w4_load(rs_res, FULL ? r0 + (unsigned)(l * p.res_sp) * 4u : w4_sel(ok, r0 + (unsigned)(l * p.res_sp) * 4u), 0u);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (RES && k + 1 < 4) {
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const unsigned d = (unsigned)(((k + 1) * p.W + l) * p.res_sp) * 4u;
        const bool ok = FULL ? true : ((oy + k + 1 < p.H) & (ox + l < p.W));
        rv[(k + 1) & 1][l] = w4_load(rs_res, FULL ? r0 + d : w4_sel(ok, r0 + d), 0u);
      }
    }
    w4_at(acc[6 * k], acc[6 * k + 1], acc[6 * k + 2], acc[6 * k + 3], acc[6 * k + 4], acc[6 * k + 5], acc[6 * k],
          acc[6 * k + 1], acc[6 * k + 2], acc[6 * k + 3]);
    float o16[16];   // the row's 4 pixels x 4 channels: ONE activation-code test per row (a per-quad test is 5 scalar branches)
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      w4_f4 y = acc[6 * k + l] + bv;
      if (RES) y = y + rv[k & 1][l];
#pragma unroll
      for (int r = 0; r < 4; ++r) o16[4 * l + r] = y[r];
    }
    sr_activate_group(o16, slope);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const unsigned d = (unsigned)((k * p.W + l) * p.out_sp) * 4u;
      const bool ok = FULL ? true : ((oy + k < p.H) & (ox + l < p.W));
      w4_store(w4_f4{o16[4 * l], o16[4 * l + 1], o16[4 * l + 2], o16[4 * l + 3]}, rs_out, FULL ? o0 + d : w4_sel(ok, o0 + d));
    }
  }
}

// V = B^T d B of this thread's (tile, ci) patch: raw -> V, in two halves of the vertical frequencies (18 instead of 36 live
// temporaries; the 6 x 6 patch is read twice -- LDS reads are not what the phase waits for)
__device__ __forceinline__ void w4_transform(const float* t_rd, float* t_wr) {
#pragma unroll
  for (int half = 0; half < ((SR_W4_ABL & 1) ? 0 : 2); ++half) {
    float tt[3][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      float d[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) d[r] = t_rd[(r * W4_PS + c) * W4_RS];
      if (half == 0) {
        const float a = fmaf(-4.0f, d[2], d[4]), b = fmaf(-4.0f, d[1], d[3]);
        tt[0][c] = fmaf(-4.25f, d[2], d[0]) + d[4];
        tt[1][c] = fmaf(0.5f, b, a);
        tt[2][c] = fmaf(-0.5f, b, a);
      } else {
        const float cc = fmaf(-0.25f, d[2], d[4]), e = fmaf(-0.25f, d[1], d[3]);
        tt[0][c] = fmaf(2.0f, e, cc);
        tt[1][c] = fmaf(-2.0f, e, cc);
        tt[2][c] = fmaf(-4.25f, d[3], d[1]) + d[5];
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float v[6];
      w4_bt(tt[i][0], tt[i][1], tt[i][2], tt[i][3], tt[i][4], tt[i][5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
      for (int j = 0; j < 6; ++j) t_wr[((3 * half + i) * 6 + j) * 256] = v[j];
    }
  }
}

// the first W4_PD frequency pairs' weight fragments of a slab (issued a barrier ahead of the MFMA tick that uses them)
template <int NA = W4_NA, int PD = W4_PD>
__device__ __forceinline__ void w4_u_prefetch(__amdgpu_buffer_rsrc_t rs_u, unsigned u_voff, unsigned u_slab, unsigned u_fstride,
                                              w4_f4 (&ua)[NA][2]) {
  if (SR_W4_ABL & (2 | 32)) return;
#pragma unroll
  for (int fp = 0; fp < PD; ++fp) {
    ua[fp][0] = w4_load(rs_u, u_voff, u_slab + (unsigned)(2 * fp) * u_fstride);
    ua[fp][1] = w4_load(rs_u, u_voff, u_slab + (unsigned)(2 * fp + 1) * u_fstride);
  }
}

// 36 frequencies x (16 co x 16 tiles x 16 ci): pairs of frequencies interleaved (dependent MFMAs 64 clk apart); the first
// W4_PD pairs of weight fragments are already in ua[]
// FIRST: the accumulators start at zero (an item's first slab): the first MFMA of every frequency takes the constant 0 as its C
// operand instead of 144 registers that somebody had to clear.
#ifndef SR_W4WS_YIELD
#define SR_W4WS_YIELD 0
#endif
#ifndef SR_W4WS_YIELD_PERIOD
#define SR_W4WS_YIELD_PERIOD 1
#endif
constexpr int W4WS_YIELD = SR_W4WS_YIELD;   // wait states the M waves idle behind every W4WS_YIELD_PERIOD-th MFMA (w4_mfma_tick)
constexpr int W4WS_YIELD_PERIOD = SR_W4WS_YIELD_PERIOD;
// YIELD > 0 (wave-specialised form): `s_nop YIELD - 1` behind every MFMA.  An fp32 MFMA holds the wave's issue for its 32 clocks and
// the M wave re-arbitrates with its next MFMA at once: the T wave on the SIMD gets a slot only every second or third MFMA
// (profiles/r06_w4ws_trace.txt: its 51-instruction column pass takes the whole 4.9-k-clock stream) and does the rest of its tick
// AFTER the stream while the M waves idle at the barrier.  A few wait states per MFMA are issue slots the T wave takes; they
// cost the M wave their full length, so only as many as the T wave's work needs.
template <int N>
__device__ __forceinline__ void w4_yield() {
  if (N > 0) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(N > 16 ? 15 : (N > 0 ? N - 1 : 0)));
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <bool FIRST = false, int NA = W4_NA, int PD = W4_PD, int YIELD = 0>
__device__ __forceinline__ void w4_mfma_tick(__amdgpu_buffer_rsrc_t rs_u, unsigned u_voff, unsigned u_slab, unsigned u_fstride,
                                             const float* m_rd, w4_f4 (&ua)[NA][2], w4_f4 (&acc)[36], int lane) {
  if (SR_W4_ABL & 2) return;
  w4_prio_mfma();
  w4_f4 vb[2][2];
  if (SR_W4_ABL & 32) {
#pragma unroll
    for (int a = 0; a < NA; ++a) ua[a][0] = ua[a][1] = w4_f4{1.0f, 2.0f, 3.0f, (float)u_slab};
    vb[0][0] = vb[0][1] = vb[1][0] = vb[1][1] = w4_f4{1.0f, 0.5f, 0.25f, (float)lane};
  } else {
    vb[0][0] = *reinterpret_cast<const w4_f4*>(m_rd);
    vb[0][1] = *reinterpret_cast<const w4_f4*>(m_rd + 256);
  }
#pragma unroll
  for (int fp = 0; fp < 18; ++fp) {
    if (!(SR_W4_ABL & 32) && fp + PD < 18) {
      ua[(fp + PD) % NA][0] = w4_load(rs_u, u_voff, u_slab + (unsigned)(2 * (fp + PD)) * u_fstride);
      ua[(fp + PD) % NA][1] = w4_load(rs_u, u_voff, u_slab + (unsigned)(2 * (fp + PD) + 1) * u_fstride);
    }
    if (!(SR_W4_ABL & 32) && fp + 1 < 18) {
      vb[(fp + 1) & 1][0] = *reinterpret_cast<const w4_f4*>(m_rd + (2 * fp + 2) * 256);
      vb[(fp + 1) & 1][1] = *reinterpret_cast<const w4_f4*>(m_rd + (2 * fp + 3) * 256);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const w4_f4 zero = w4_f4{0.0f, 0.0f, 0.0f, 0.0f};
      acc[2 * fp] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[fp % NA][0][e], vb[fp & 1][0][e], (FIRST && e == 0) ? zero : acc[2 * fp], 0, 0, 0);
      if ((8 * fp + 2 * e) % W4WS_YIELD_PERIOD == 0) w4_yield<YIELD>();
      acc[2 * fp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[fp % NA][1][e], vb[fp & 1][1][e], (FIRST && e == 0) ? zero : acc[2 * fp + 1], 0, 0, 0);
      if ((8 * fp + 2 * e + 1) % W4WS_YIELD_PERIOD == 0) w4_yield<YIELD>();
      __builtin_amdgcn_sched_barrier(0);   // keep the two accumulators interleaved (left alone hipcc issues 4 dependent MFMAs in a row)
    }
  }
  w4_prio_other();
}

__device__ __forceinline__ void w4_finish(const SrWino4Params& p, const W4Item& it, w4_f4 (&acc)[36], int wave, int m_kq,
                                          int m_j, int tid) {
  if (SR_W4_ABL & 4) {   // keep the accumulators alive without an epilogue
    w4_f4 sum = acc[0];
#pragma unroll
    for (int f = 1; f < 36; ++f) sum = sum + acc[f];
    if (sum[0] + sum[1] + sum[2] + sum[3] == 1.2345e33f) p.out[tid] = sum[0];
    return;
  }
  const int cq = it.co0 + 16 * wave + 4 * m_kq;                       // first of this lane's 4 output channels
  const bool full = (it.oy0 + 16 <= p.H) & (it.ox0 + 16 <= p.W);      // uniform
  if (p.res != nullptr) {
    if (full) w4_epilogue<true, true>(p, it, acc, cq, m_j);
    else w4_epilogue<false, true>(p, it, acc, cq, m_j);
  } else {
    if (full) w4_epilogue<true, false>(p, it, acc, cq, m_j);
    else w4_epilogue<false, false>(p, it, acc, cq, m_j);
  }
}

// ---- the first form: two independent 4-wave workgroups per CU, each slab = stage / barrier / T / barrier / M in sequence.
// Kept as `variant` 1 for A/B measurements; same operations in the same order as the ping-pong kernel below.
__global__ __launch_bounds__(256, 2) void sr_wino4_kernel(SrWino4Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  w4_prio_other();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t_ci = tid & 15, t_tile = tid >> 4;
  const int t_sig = (0x1230 >> (t_tile & 12)) & 3;   // sigma(tile >> 2) = (0, 3, 2, 1)
  const float* t_rd = lds + ((4 * (t_tile >> 2)) * W4_PS + 4 * (t_tile & 3)) * W4_RS + t_ci;
  float* t_wr = lds + W4_RAW_FLOATS + t_tile * 16 + 4 * ((t_ci >> 2) ^ t_sig) + (t_ci & 3);
  const int m_j = lane & 15, m_kq = lane >> 4;
  const int m_sig = (0x1230 >> (m_j & 12)) & 3;
  const float* m_rd = lds + W4_RAW_FLOATS + m_j * 16 + 4 * (m_kq ^ m_sig);
  const unsigned u_voff = (unsigned)(m_kq * p.Co_pad + 16 * wave + m_j) * 16u;
  const unsigned u_fstride = (SR_W4_ABL & 8) ? 0u : (unsigned)p.S * 4u * (unsigned)p.Co_pad * 16u;   // bytes between two frequencies
  const unsigned u_sstride = (SR_W4_ABL & 8) ? 0u : 4u * (unsigned)p.Co_pad * 16u;                   // ... two slabs
  const int st_q = tid & 3, st_pp0 = tid >> 2;
  float* st_wr = lds + st_pp0 * W4_RS + 4 * st_q;

  int work = blockIdx.x;
  if (work >= p.total) return;
  W4Item it = w4_decode(p, work);
  w4_f4 st[W4_STAGE];
  w4_stage(p, it, 0, st_q, st_pp0, st);
  const __amdgpu_buffer_rsrc_t rs_u = w4_rsrc(p.wu, (int64_t)36 * p.S * 4 * p.Co_pad * 16);
  for (;;) {
    const int next_work = work + (int)gridDim.x;
    const bool has_next = next_work < p.total;
    const W4Item nxt = w4_decode(p, has_next ? next_work : work);
    w4_f4 acc[36];
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = w4_f4{0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned u_item = (SR_W4_ABL & 8) ? 0u : (unsigned)it.co0 * 16u;
    for (int s = 0; s < p.S; ++s) {
      if (!(SR_W4_ABL & 16)) {   // this slab's patch registers -> LDS; then the next slab's (or the next item's first) loads
#pragma unroll
        for (int j = 0; j < W4_STAGE - 1; ++j) *reinterpret_cast<w4_f4*>(st_wr + j * 64 * W4_RS) = st[j];
        if (tid < 16) *reinterpret_cast<w4_f4*>(st_wr + (W4_STAGE - 1) * 64 * W4_RS) = st[W4_STAGE - 1];
        if (s + 1 < p.S) w4_stage(p, it, s + 1, st_q, st_pp0, st);
        else if (has_next) w4_stage(p, nxt, 0, st_q, st_pp0, st);
      }
      __syncthreads();   // raw visible; every wave is past the previous slab's V reads
      w4_transform(t_rd, t_wr);
      __syncthreads();   // V visible (and raw free for the next slab's store)
      w4_f4 ua[W4_NA][2];
      w4_u_prefetch(rs_u, u_voff, u_item + (unsigned)s * u_sstride, u_fstride, ua);
      w4_mfma_tick(rs_u, u_voff, u_item + (unsigned)s * u_sstride, u_fstride, m_rd, ua, acc, lane);
    }
    w4_finish(p, it, acc, wave, m_kq, m_j, tid);
    if (!has_next) break;
    work = next_work;
    it = nxt;
  }
}

// ---- the ping-pong form (default): ONE 8-wave workgroup per CU = two 4-wave groups, each walking its own list of work items
// with the slab pipeline of sr_wino4_kernel, one workgroup barrier per TICK, and the two groups one tick apart: while group A
// transforms (LDS + VALU + its epilogue's memory traffic) group B streams MFMAs, then they swap.  Two independent workgroups
// per CU that start together run their phases IN step (r04 found the same on sr_wino_kernel; r05 ablations on the 4-wave form:
// MFMA 72 us + transform 19 + operand loads 45 + staging 32 + epilogue 64 = the measured 218 us of a 64 -> 64 layer at
// 8 x 240 x 320, i.e. nothing overlapped); here the complementary phases are the construction, not an accident.
//   T tick of a group:  [tail tick: epilogue of the item that just finished]  T(s): raw -> V;  first weight fragments of M(s)
//   M tick:             M(s): 144 MFMAs per wave;  patch registers of slab s + 1 -> raw;  loads of slab s + 2
// (raw is read in T ticks and written at the end of M ticks, V is written in T ticks and read in M ticks: the tick barrier
// orders both).  An item takes 2 S ticks.  Every wave executes the same NUMBER of barriers: group 1 starts one barrier late,
// whoever finishes first pads.
__global__ __launch_bounds__(512, 2) void sr_wino4pp_kernel(SrWino4Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  w4_prio_other();
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  float* lds = lds_all + grp * (W4_RAW_FLOATS + W4_V_FLOATS);
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int t_ci = tid & 15, t_tile = tid >> 4;
  const int t_sig = (0x1230 >> (t_tile & 12)) & 3;
  const float* t_rd = lds + ((4 * (t_tile >> 2)) * W4_PS + 4 * (t_tile & 3)) * W4_RS + t_ci;
  float* t_wr = lds + W4_RAW_FLOATS + t_tile * 16 + 4 * ((t_ci >> 2) ^ t_sig) + (t_ci & 3);
  const int m_j = lane & 15, m_kq = lane >> 4;
  const int m_sig = (0x1230 >> (m_j & 12)) & 3;
  const float* m_rd = lds + W4_RAW_FLOATS + m_j * 16 + 4 * (m_kq ^ m_sig);
  const unsigned u_voff = (unsigned)(m_kq * p.Co_pad + 16 * wave + m_j) * 16u;
  const unsigned u_fstride = (SR_W4_ABL & 8) ? 0u : (unsigned)p.S * 4u * (unsigned)p.Co_pad * 16u;
  const unsigned u_sstride = (SR_W4_ABL & 8) ? 0u : 4u * (unsigned)p.Co_pad * 16u;   // bytes between two slabs
  const int st_q = tid & 3, st_pp0 = tid >> 2;
  float* st_wr = lds + st_pp0 * W4_RS + 4 * st_q;

  // this group's items: slot, slot + stride, ...; barriers: grp + 1 + 2 S per item, padded to the longer group's count
  const int stride = 2 * (int)gridDim.x;
  const int slot = 2 * (int)blockIdx.x + grp;
  const int n_mine = slot < p.total ? (p.total - slot + stride - 1) / stride : 0;
  const int n_a = (p.total - 2 * (int)blockIdx.x + stride - 1) / stride;   // group 0 (the grid never exceeds ceil(total / 2))
  const int n_b = 2 * (int)blockIdx.x + 1 < p.total ? (p.total - 2 * (int)blockIdx.x - 1 + stride - 1) / stride : 0;
  const int bars_a = 1 + 2 * p.S * n_a, bars_b = n_b > 0 ? 2 + 2 * p.S * n_b : 0;
  const int bars_all = bars_a > bars_b ? bars_a : bars_b;
  int bars = 0;   // barriers this wave has executed
#ifdef SR_W4_TRACE
  int tr_n = 0;
#endif

  if (n_mine > 0) {
    int work = slot;
    W4Item it = w4_decode(p, work);
    w4_f4 st[W4_STAGE];
    w4_f4 acc[36];
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = w4_f4{0.0f, 0.0f, 0.0f, 0.0f};
    w4_f4 ua[W4_NA][2];
    const __amdgpu_buffer_rsrc_t rs_u = w4_rsrc(p.wu, (int64_t)36 * p.S * 4 * p.Co_pad * 16);
    if (grp == 1) { __syncthreads(); ++bars; }   // one tick late
    // first patch: slab 0 -> raw (this group's own buffer, nobody else reads it: no barrier needed before its own T ... but the
    // writers are 4 waves and the readers other threads of the group: the tick barrier below comes AFTER T, so synchronise the
    // group here through a workgroup barrier that the other group matches with one of its own ticks)
    w4_stage(p, it, 0, st_q, st_pp0, st);
    if (!(SR_W4_ABL & 16)) {
#pragma unroll
      for (int j = 0; j < W4_STAGE - 1; ++j) *reinterpret_cast<w4_f4*>(st_wr + j * 64 * W4_RS) = st[j];
      if (tid < 16) *reinterpret_cast<w4_f4*>(st_wr + (W4_STAGE - 1) * 64 * W4_RS) = st[W4_STAGE - 1];
    }
    {
      const bool more = p.S > 1 || n_mine > 1;
      if (!(SR_W4_ABL & 16) && more) {
        if (p.S > 1) w4_stage(p, it, 1, st_q, st_pp0, st);
        else w4_stage(p, w4_decode(p, work + stride), 0, st_q, st_pp0, st);
      }
    }
    __syncthreads(); ++bars;   // raw(slab 0) visible to the group
    w4_transform(t_rd, t_wr);
    w4_u_prefetch(rs_u, u_voff, (SR_W4_ABL & 8) ? 0u : (unsigned)it.co0 * 16u, u_fstride, ua);
    __syncthreads(); ++bars;   // ---- end of the first T tick

    for (int done = 0; done < n_mine; ++done) {
      const bool has_next = done + 1 < n_mine;
      const W4Item nxt = w4_decode(p, has_next ? work + stride : work);
      const unsigned u_item = (SR_W4_ABL & 8) ? 0u : (unsigned)it.co0 * 16u;
      for (int s = 0; s < p.S; ++s) {
        // ---- M tick: MFMAs of slab s; then the patch in st[] (slab s + 1 / the next item's slab 0) -> raw and the loads of
        // the patch after that
        W4_TR(1);
        w4_mfma_tick(rs_u, u_voff, u_item + (unsigned)s * u_sstride, u_fstride, m_rd, ua, acc, lane);
        W4_TR(2);
        const bool last = s + 1 == p.S;
        if (!(SR_W4_ABL & 16) && (!last || has_next)) {
#pragma unroll
          for (int j = 0; j < W4_STAGE - 1; ++j) *reinterpret_cast<w4_f4*>(st_wr + j * 64 * W4_RS) = st[j];
          if (tid < 16) *reinterpret_cast<w4_f4*>(st_wr + (W4_STAGE - 1) * 64 * W4_RS) = st[W4_STAGE - 1];
          if (s + 2 < p.S) w4_stage(p, it, s + 2, st_q, st_pp0, st);
          else if (!last && has_next) w4_stage(p, nxt, 0, st_q, st_pp0, st);                  // s + 2 == S: the next item's slab 0
          else if (last && has_next) {                                                         // the next item's slab 1 / its successor's slab 0
            if (p.S > 1) w4_stage(p, nxt, 1, st_q, st_pp0, st);
            else if (done + 2 < n_mine) w4_stage(p, w4_decode(p, work + 2 * stride), 0, st_q, st_pp0, st);
          }
        }
        W4_TR(3);
        __syncthreads(); ++bars;
        if (!last) {
          // ---- T tick of slab s + 1
          W4_TR(4);
          w4_transform(t_rd, t_wr);
          w4_u_prefetch(rs_u, u_voff, u_item + (unsigned)(s + 1) * u_sstride, u_fstride, ua);
          W4_TR(5);
          __syncthreads(); ++bars;
        }
      }
      // ---- tail tick: this item's epilogue, then T of the next item's slab 0
      W4_TR(6);
      w4_finish(p, it, acc, wave, m_kq, m_j, tid);
      W4_TR(7);
#pragma unroll
      for (int f = 0; f < 36; ++f) acc[f] = w4_f4{0.0f, 0.0f, 0.0f, 0.0f};
      if (has_next) {
        w4_transform(t_rd, t_wr);
        w4_u_prefetch(rs_u, u_voff, (SR_W4_ABL & 8) ? 0u : (unsigned)nxt.co0 * 16u, u_fstride, ua);
        W4_TR(8);
        __syncthreads(); ++bars;
      }
      work += stride;
      it = nxt;
    }
  }
  for (; bars < bars_all; ++bars) __syncthreads();   // (the other group is still working)
}

// packed fp32 arithmetic on channel PAIRS (v_pk_fma_f32 / v_pk_add_f32: two IEEE fmas per instruction, component-wise the very
// operations of w4_bt -- bit-identical)
typedef float w4_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ w4_f2 w4_fma2(float a, w4_f2 b, w4_f2 c) { return __builtin_elementwise_fma(w4_f2{a, a}, b, c); }
__device__ __forceinline__ void w4_bt2(w4_f2 d0, w4_f2 d1, w4_f2 d2, w4_f2 d3, w4_f2 d4, w4_f2 d5, w4_f2& t0, w4_f2& t1, w4_f2& t2,
                                       w4_f2& t3, w4_f2& t4, w4_f2& t5) {
  const w4_f2 a = w4_fma2(-4.0f, d2, d4), b = w4_fma2(-4.0f, d1, d3);
  const w4_f2 c = w4_fma2(-0.25f, d2, d4), e = w4_fma2(-0.25f, d1, d3);
  t0 = w4_fma2(-4.25f, d2, d0) + d4;
  t1 = w4_fma2(0.5f, b, a);
  t2 = w4_fma2(-0.5f, b, a);
  t3 = w4_fma2(2.0f, e, c);
  t4 = w4_fma2(-2.0f, e, c);
  t5 = w4_fma2(-4.25f, d3, d1) + d5;
}
// ---- the wave-specialised form (variant 3; rewritten in r06): ONE 8-wave workgroup per CU; waves 0-3 ("M") do nothing but
// stream the MFMAs (weight fragments from L2, V fragments from LDS) and, when an item's last slab is done, run the output
// transform in place and hand the finished 16 x 16 x 64 tile over through LDS; waves 4-7 ("T") own BOTH memory sides: they move
// the input patches (global -> registers -> LDS), transform them into V, and turn the finished tiles into stores (residual +
// bias + activation, whole 256-byte pixel records).
// What r06 measured on the r05 form (profiles/r06_w4ws_trace.txt; 64 -> 64 at 8 x 240 x 320: 206 us, 143 without the output
// side, 170 without the patch movement) and what this form does about it:
//   * the r05 M waves also loaded the patches, in the same in-order vmcnt queue as their weight fragments: the first MFMA of a tick
//     waited for the L2 misses of a patch that was not needed for another tick (the last slab of an item ran 5-10 k clocks
//     instead of 4.9 k).  Here the M waves' only memory traffic is the weight stream, prefetched four frequency pairs ahead;
//   * a wave next to an MFMA stream issues at most ONE instruction per MFMA (scripts/micro/mfma16_overlap.hip; in this kernel
//     one per two or three), so every T-wave instruction is worth 40-90 clocks while the M waves stream.  The r05 T waves executed
//     ~1 550 instructions per 64 -> 64 item (576 MFMA slots): the item decode's integer divisions on every tick (~110), both
//     sums and a select per value around a uniform `has_res`, a branch per store; the M waves idled at the barrier behind them.
//     Here: ~120 per tick + ~250 per item; the item cursor advances by additions, residual / activation class are template
//     parameters, the transform needs no lane swaps;
//   * the M waves' output transform runs on channel pairs (240 v_pk_* instead of 432 instructions), the accumulators are not
//     cleared (the first MFMA of a frequency takes C = 0);
//   * tried and dropped: patches straight into the T waves' registers in transform layout (18 8-byte loads per lane, no staging
//     buffer, whole tile in LDS, no hand-over barriers) -- correct, but each such wave-load touches eight 64-byte segments and
//     the patch is fetched 2.25x: the texture-address unit needed ~2 k clocks per tick for them next to the weight stream's 2.3 k
//     (of a 4.9-k tick) and the T waves became the critical path (202 us).
// Per TICK (one workgroup barrier F), k = 0 .. K over the slabs of all the workgroup's items in a row:
//   T waves: [tile half 0 of the item that closed in tick k - 1: LDS -> registers; barriers B2, B3]
//            [residual loads of the item the M waves are closing in this tick: consumed a tick and a half later]
//            V[k & 1] = B^T d B of raw[k & 1]   (first: it needs nothing from memory, the patch loads in flight get the whole tick)
//            [the closed item's output side: half 0, then half 1 from LDS]
//            patch k + 1 (in registers since tick k - 1) -> raw[(k + 1) & 1]; loads of patch k + 2
//   M waves: MFMAs of slab k - 1 on V[(k - 1) & 1]; if that closes an item: Y = A^T M A in place, rows 0-1 of every tile -> LDS,
//            F, B2 (the T waves have read them), rows 2-3 -> LDS, B3.
// raw and V are double-buffered, the hand-over buffer holds half a tile: 158 336 of the CU's 163 840 bytes.
// Weight-fragment register sets / prefetch distance (frequency pairs) of the M waves: the CU's vector-memory pipeline returns
// data in order ACROSS waves, so a weight fragment (an L2 hit) requested behind the T waves' patch / residual loads (L2 misses)
// arrives when THEY do; the M waves have the registers for a longer lead (256 clocks per pair).
#ifndef SR_W4WS_NA
#define SR_W4WS_NA 5
#define SR_W4WS_PD 4
#endif
constexpr int W4WS_NA = SR_W4WS_NA, W4WS_PD = SR_W4WS_PD;

constexpr int W4_WS_RAW2 = 2 * W4_RAW_FLOATS;                          // V buffers start behind the two raw buffers
constexpr int W4_WS_OUT = W4_WS_RAW2 + 2 * W4_V_FLOATS;                // the output hand-over buffer (half a region: 16 tiles x 8 pixels x 64 channels)
constexpr int W4_WS_OUT_FLOATS = 16 * 8 * 64;
constexpr int W4_WS_LDS_BYTES = (W4_WS_OUT + W4_WS_OUT_FLOATS) * 4;    // 158 336 of the CU's 163 840: one workgroup per CU

// A^T on channel pairs (v_pk_*: the M waves' output transform is issue-bound, half the instructions of the float4 form).  The
// scalings are powers of two (exact), so the value of each output equals w4_at's bit for bit.
__device__ __forceinline__ void w4_at2(w4_f2 m0, w4_f2 m1, w4_f2 m2, w4_f2 m3, w4_f2 m4, w4_f2 m5, w4_f2& s0, w4_f2& s1, w4_f2& s2,
                                       w4_f2& s3) {
  const w4_f2 p = m1 + m2, q = m1 - m2, u = m3 + m4, v = m3 - m4;
  s0 = (m0 + p) + u;
  s1 = w4_fma2(0.5f, q, 2.0f * v);
  s2 = w4_fma2(0.25f, p, 4.0f * u);
  s3 = w4_fma2(0.125f, q, w4_fma2(8.0f, v, m5));
}
#ifndef SR_W4WS_XCD
#define SR_W4WS_XCD 1   // (A/B: 0 = items dealt round-robin to the workgroups)
#endif
#ifndef SR_W4WS_TUNE
#define SR_W4WS_TUNE 0   // A/B builds: 4: scalar output transform (384 v_* instead of 240 v_pk_*: same time, r06)
#endif
__device__ __forceinline__ float w4_opaque(float x) { asm volatile("" : "+v"(x)); return x; }   // (keeps the SLP vectoriser from pairing)
__device__ __forceinline__ void w4_at_pk(w4_f4& a0, w4_f4& a1, w4_f4& a2, w4_f4& a3, const w4_f4& a4, const w4_f4& a5) {
  if (SR_W4_ABL & 64) { a0 = a0 + a4; a1 = a1 + a5; return; }   // (timing experiments: nearly no arithmetic)
  if (SR_W4WS_TUNE & 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m0 = a0[e], m1 = a1[e], m2 = a2[e], m3 = a3[e], m4 = a4[e], m5 = a5[e];
      const float p = w4_opaque(m1 + m2), q = w4_opaque(m1 - m2), u = w4_opaque(m3 + m4), v = w4_opaque(m3 - m4);
      a0[e] = w4_opaque(w4_opaque(m0 + p) + u);
      a1[e] = w4_opaque(fmaf(0.5f, q, w4_opaque(2.0f * v)));
      a2[e] = w4_opaque(fmaf(0.25f, p, w4_opaque(4.0f * u)));
      a3[e] = w4_opaque(fmaf(0.125f, q, w4_opaque(fmaf(8.0f, v, m5))));
    }
    return;
  }
  w4_f2 lo[4], hi[4];
  w4_at2(w4_f2{a0[0], a0[1]}, w4_f2{a1[0], a1[1]}, w4_f2{a2[0], a2[1]}, w4_f2{a3[0], a3[1]}, w4_f2{a4[0], a4[1]}, w4_f2{a5[0], a5[1]},
         lo[0], lo[1], lo[2], lo[3]);
  w4_at2(w4_f2{a0[2], a0[3]}, w4_f2{a1[2], a1[3]}, w4_f2{a2[2], a2[3]}, w4_f2{a3[2], a3[3]}, w4_f2{a4[2], a4[3]}, w4_f2{a5[2], a5[3]},
         hi[0], hi[1], hi[2], hi[3]);
  a0 = w4_f4{lo[0][0], lo[0][1], hi[0][0], hi[0][1]};
  a1 = w4_f4{lo[1][0], lo[1][1], hi[1][0], hi[1][1]};
  a2 = w4_f4{lo[2][0], lo[2][1], hi[2][0], hi[2][1]};
  a3 = w4_f4{lo[3][0], lo[3][1], hi[3][0], hi[3][1]};
}

// The work-item cursor of a role: (region column, region row, image, output-channel block) of item `work`, advanced by the grid
// stride with additions and carries (scalar unit) -- the divisions of w4_decode happen twice per launch, not once per tick.
struct W4Cursor { int rx, ry, b, cb; };
__device__ __forceinline__ W4Cursor w4_cursor(const SrWino4Params& p, int work) {
  W4Cursor c;
  int rem = work;
  c.rx = rem % p.regions_x; rem /= p.regions_x;
  c.ry = rem % p.regions_y; rem /= p.regions_y;
  c.b = rem % p.B;
  c.cb = rem / p.B;
  return c;
}
__device__ __forceinline__ void w4_cursor_advance(const SrWino4Params& p, W4Cursor& c, const W4Cursor& step) {
  c.rx += step.rx;
  int carry = c.rx >= p.regions_x; c.rx -= carry ? p.regions_x : 0;
  c.ry += step.ry + carry;
  carry = c.ry >= p.regions_y; c.ry -= carry ? p.regions_y : 0;
  c.b += step.b + carry;
  carry = c.b >= p.B; c.b -= carry ? p.B : 0;
  c.cb += step.cb + carry;
}
__device__ __forceinline__ W4Item w4_item(const W4Cursor& c) { return W4Item{c.b, 16 * c.ry, 16 * c.rx, 64 * c.cb}; }

// The T waves' transform: a lane takes one (tile, channel pair) and HALF of the vertical frequencies -- waves 4, 5 rows i = 0..2 of
// T = B^T d (the t0, t1, t2 outputs of w4_bt: 6 packed operations per patch column, patch rows 0-4), waves 6, 7 rows 3..5 (t3, t4,
// t5: 6 more, patch rows 1-5) --, then its three rows of V = T B.  Per slab and wave: 15 ds_read2_b64, 72 packed operations,
// 9 ds_write2st64_b64 -- 96 instructions.  (r05 / early r06: the two halves of ONE wave shared an item, each took three patch
// COLUMNS through all six vertical frequencies and 18 v_permlane32_swap handed the halves their rows: 129 instructions; every
// instruction of these waves costs a 37-clock slot next to the MFMA stream.)  Component-wise the very operations of w4_bt in the
// same order: bit-identical to the other kernel forms.  `upper` is wave-uniform; t_wr carries the half's offset (18 frequencies).
#ifdef SR_W4_TRACE
#define W4_TF_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); tf_stamp[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ void w4_transform36_half(const float* t_rd, float* t_wr, bool upper, unsigned long long (&tf_stamp)[2]) {
#else
#define W4_TF_STAMP(i) do {} while (0)
__device__ __forceinline__ void w4_transform36_half(const float* t_rd, float* t_wr, bool upper) {
#endif
  if (SR_W4_ABL & 1) return;
  w4_f2 T[3][6];
  if (!upper) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      w4_f2 d[5];
#pragma unroll
      for (int r = 0; r < 5; ++r) d[r] = *reinterpret_cast<const w4_f2*>(t_rd + (r * W4_PS + k) * W4_RS);
      const w4_f2 a = w4_fma2(-4.0f, d[2], d[4]), b = w4_fma2(-4.0f, d[1], d[3]);
      T[0][k] = w4_fma2(-4.25f, d[2], d[0]) + d[4];
      T[1][k] = w4_fma2(0.5f, b, a);
      T[2][k] = w4_fma2(-0.5f, b, a);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      w4_f2 d[6];
#pragma unroll
      for (int r = 1; r < 6; ++r) d[r] = *reinterpret_cast<const w4_f2*>(t_rd + (r * W4_PS + k) * W4_RS);
      const w4_f2 c = w4_fma2(-0.25f, d[2], d[4]), e = w4_fma2(-0.25f, d[1], d[3]);
      T[0][k] = w4_fma2(2.0f, e, c);
      T[1][k] = w4_fma2(-2.0f, e, c);
      T[2][k] = w4_fma2(-4.25f, d[3], d[1]) + d[5];
    }
  }
  W4_TF_STAMP(0);
#pragma unroll
  for (int ii = 0; ii < 3; ++ii) {
    w4_f2 v[6];
    w4_bt2(T[ii][0], T[ii][1], T[ii][2], T[ii][3], T[ii][4], T[ii][5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<w4_f2*>(t_wr + (ii * 6 + j) * 256) = v[j];
  }
}

// patch loads of the T waves: interior regions take the precomputed lane offsets (no address arithmetic per slab).  `valid` false
// (uniform; past the workgroup's last slab): the same six instructions with every lane switched off -- a load that is skipped on
// SOME path makes the compiler's s_waitcnt for the loads in flight wait for everything.
__device__ __forceinline__ void w4ws_stage(const SrWino4Params& p, const W4Item& it, int s, bool valid, int st_q, int st_pp0,
                                           const unsigned (&st_off)[W4_STAGE], w4_f4 (&st)[W4_STAGE]) {
  if (SR_W4_ABL & 16) return;
  const bool interior = (it.oy0 >= 1) & (it.ox0 >= 1) & (it.oy0 + 17 <= p.H) & (it.ox0 + 17 <= p.W);   // uniform
  const unsigned in_img_bytes = (unsigned)(((int64_t)(p.H * p.W - 1) * p.in_sp + p.Cin) * 4);
  const __amdgpu_buffer_rsrc_t rs_in = w4_rsrc(p.in + (int64_t)it.b * p.in_sb, in_img_bytes);
  const bool chan_ok = valid & (16 * s + 4 * st_q + 4 <= p.Cin);
  if (interior) {
    const unsigned base = (unsigned)(((it.oy0 - 1) * p.W + (it.ox0 - 1)) * p.in_sp + 16 * s) * 4u;   // scalar offset operand
    if (valid && 16 * s + 16 <= p.Cin) {   // (uniform) every channel quad of the slab exists
#pragma unroll
      for (int j = 0; j < W4_STAGE; ++j) st[j] = w4_load_aux<SR_W4WS_AUX_PATCH>(rs_in, st_off[j], base);
    } else {
#pragma unroll
      for (int j = 0; j < W4_STAGE; ++j) st[j] = w4_load_aux<SR_W4WS_AUX_PATCH>(rs_in, chan_ok ? st_off[j] : W4_OOB, base);
    }
  } else {
    const unsigned q_off = (unsigned)(16 * s + 4 * st_q) * 4u;
#pragma unroll
    for (int j = 0; j < W4_STAGE; ++j) {
      const int pp = st_pp0 + 64 * j;
      const int r = (pp * 3641) >> 16, c = pp - 18 * r;
      const int y = it.oy0 - 1 + r, x = it.ox0 - 1 + c;
      const bool ok = chan_ok & (pp < W4_PS * W4_PS) & (y >= 0) & (y < p.H) & (x >= 0) & (x < p.W);
      st[j] = w4_load(rs_in, w4_sel(ok, (unsigned)((y * p.W + x) * p.in_sp) * 4u + q_off), 0u);
    }
  }
}

// GENERIC_ACT false: LeakyReLU with 0 <= slope <= 1 (max(v, slope v)) or no activation -- every 3x3 convolution of the reference's
// BasicBlock stack; true: any activation code (sr_activate_group: SiLU of the image-prior encoder's blocks).  Two kernels instead
// of three code paths per output batch in one (the SiLU path alone is 420 instructions per 32 values).
// RES: the layer adds a residual (a load on SOME path spoils the wait counts of everything behind it, see w4ws_stage).
template <bool GENERIC_ACT, bool RES>
__global__ __launch_bounds__(512, 2) void sr_wino4ws_kernel(SrWino4Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  w4_prio_other();
  const int role_t = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // 0: MFMA waves, 1: transform / memory waves
  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's items: first, first + item_stride, ...; K = n S slabs in a row; slab k - 1 closes an item iff k % S == 0
#if SR_W4WS_XCD
  // XCD-aware work order: workgroup b runs on XCD b % 8 (observed dispatch order; speed only) and every XCD has its own L2, so
  // each XCD walks a CONTIGUOUS eighth of the items (a band of region rows of one image): the 2-pixel halo a patch shares with its
  // neighbours is fetched through the fabric once per XCD instead of once per region (round-robin: 1.31 x the algorithmic bytes).
  const int xcd = (int)blockIdx.x & 7, n_x = ((int)gridDim.x - xcd + 7) >> 3;   // workgroups of this XCD
  const int x_chunk = ((int)p.total + 7) >> 3;
  const int first = xcd * x_chunk + ((int)blockIdx.x >> 3), x_end = min((int)p.total, (xcd + 1) * x_chunk);
  const int n_items = first < x_end ? (x_end - first + n_x - 1) / n_x : 0;
  const int item_stride = n_x;
  if (n_items == 0) return;   // (uniform: the whole workgroup, before any barrier)
#else
  const int n_items = ((int)p.total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int first = (int)blockIdx.x, item_stride = (int)gridDim.x;
#endif
#ifdef SR_W4WS_DEPHASE   // (probe builds: workgroups start a fraction of a tick apart -- profiles/r06_w4ws_trace.txt section 11)
  for (int i = (((int)blockIdx.x >> 3) * SR_W4WS_DEPHASE) & 63; i > 0; --i) __builtin_amdgcn_s_sleep(SR_W4WS_DEPHASE_UNIT);
#endif
  const int K = n_items * p.S;
  const bool one_slab = p.S == 1;   // an item closes on EVERY tick: half 1 of a tile is read in the tick that writes half 0 of the next
  const W4Cursor step = w4_cursor(p, item_stride);
#ifdef SR_W4_TRACE
  int tr_n = 0;
  const int grp = role_t;
  const unsigned long long tr_c0 = clock64(), tr_w0 = wall_clock64();
#endif

  if (role_t) {
    // ================= T waves
    // transform role: work item (tile, channel pair) = 64 * (wave & 1) + lane, vertical frequencies 3 (wave >> 1) .. + 2
    const int t_item = 64 * (wave & 1) + lane, t_half = wave >> 1;   // (t_half is wave-uniform)
    const int t_ci = 2 * (t_item & 7), t_tile = t_item >> 3;
    const int t_sig = (0x1230 >> (t_tile & 12)) & 3;
    const int t_rd_off = ((4 * (t_tile >> 2)) * W4_PS + 4 * (t_tile & 3)) * W4_RS + t_ci;
    const int t_wr_off = W4_WS_RAW2 + t_half * 18 * 256 + t_tile * 16 + 4 * ((t_ci >> 2) ^ t_sig) + (t_ci & 3);
    // patch role: channel quad st_q of the patch pixels st_pp0 + 64 j
    const int st_q = tid & 3, st_pp0 = tid >> 2;
    const int st_wr_off = st_pp0 * W4_RS + 4 * st_q;
    unsigned st_off[W4_STAGE];
#pragma unroll
    for (int j = 0; j < W4_STAGE; ++j) {
      const int pp = st_pp0 + 64 * j;
      const int r = (pp * 3641) >> 16, c = pp - 18 * r;
      st_off[j] = pp < W4_PS * W4_PS ? (unsigned)((r * p.W + c) * p.in_sp + 4 * st_q) * 4u : W4_OOB;
    }
    // output role: channel quad o_c of the pixels pxl_i = (tid >> 4) + 16 i, i = 0 .. 7, of a half tile (16 tiles x 2 x 4 pixels:
    // pxl = 8 tile + 4 kk + l).  With t0 = tid >> 7, kk = (tid >> 6) & 1, l = (tid >> 4) & 3:  tile_i = t0 + 2 i, so pixel i of half h
    // sits at row 4 (i >> 1) + 2 h + kk, column 4 t0 + 8 (i & 1) + l of the region: ONE lane offset per tensor (row kk, column
    // 4 t0 + l, channel quad) plus a scalar per (i, h) -- no address arithmetic per access; a wave covers four neighbouring pixels
    // x 64 channels per access (whole 256-byte records).
    const int o_c = tid & 15, o_t0 = tid >> 7, o_kk = (tid >> 6) & 1, o_l = (tid >> 4) & 3;
    const unsigned o_lane = (unsigned)((o_kk * p.W + 4 * o_t0 + o_l) * p.out_sp + 4 * o_c) * 4u;
    const unsigned r_lane = (unsigned)((o_kk * p.W + 4 * o_t0 + o_l) * p.res_sp + 4 * o_c) * 4u;
    const char* const OUT = reinterpret_cast<const char*>(lds + W4_WS_OUT) + ((tid >> 4) * 64) * 4;   // pixel 0's record
    const int o_cx = 16 * (o_c ^ o_t0);                     // its 16-byte chunk; pixel i: chunk o_c ^ tile_i = (o_c ^ t0) ^ 2 i
    const float slope = sr_uniform(p.slope);
    constexpr bool has_res = RES && !(SR_W4_ABL & 4);
    const unsigned out_img_bytes = (unsigned)(((int64_t)(p.H * p.W - 1) * p.out_sp + p.Cout) * 4);
    const unsigned res_img_bytes = (unsigned)(((int64_t)(p.H * p.W - 1) * p.res_sp + p.Cout) * 4);

    W4Cursor ld = w4_cursor(p, first);   // the patch to load next: slab ld_s of item ld
    int ld_s = 0;
    W4Cursor ep = ld;                              // the item whose output tile comes next
    w4_f4 st[W4_STAGE];
    w4_f4 rv[16];
    w4_f4 bv;

    auto load_next = [&](bool valid) {
      w4ws_stage(p, w4_item(ld), ld_s, valid, st_q, st_pp0, st_off, st);
      if (++ld_s == p.S) { ld_s = 0; w4_cursor_advance(p, ld, step); }
    };
    auto store_patch = [&](int par) {
      if (SR_W4_ABL & 16) return;
      float* wr = lds + par * W4_RAW_FLOATS + st_wr_off;
#pragma unroll
      for (int j = 0; j < W4_STAGE - 1; ++j) *reinterpret_cast<w4_f4*>(wr + j * 64 * W4_RS) = st[j];
      if (tid < 16) *reinterpret_cast<w4_f4*>(wr + (W4_STAGE - 1) * 64 * W4_RS) = st[W4_STAGE - 1];
    };
    // residual of the whole 16 x 16 x 64 tile of item `eit` + its bias quad (an empty buffer reads 0): 17 loads
    auto residual_loads = [&](const W4Item& eit) {
      if (SR_W4_ABL & 4) return;
      const bool cq_ok = eit.co0 + 4 * o_c < p.Cout;
      bv = __builtin_bit_cast(w4_f4, __builtin_amdgcn_raw_buffer_load_b128(w4_rsrc(p.bias ? p.bias : p.wu, p.bias ? (int64_t)p.Cout * 4 : 0),
                                                                          (int)(cq_ok ? (unsigned)(eit.co0 + 4 * o_c) * 4u : W4_OOB), 0, 0));
      if (!has_res) return;
      const bool full = (eit.oy0 + 16 <= p.H) & (eit.ox0 + 16 <= p.W);   // uniform
      const __amdgpu_buffer_rsrc_t rs_res = w4_rsrc(p.res + (int64_t)eit.b * p.res_sb, res_img_bytes);
      const unsigned rbase = (unsigned)((eit.oy0 * p.W + eit.ox0) * p.res_sp + eit.co0) * 4u;   // scalar
      const unsigned rl = cq_ok ? r_lane : W4_OOB;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned d = (unsigned)(((4 * (i >> 1) + 2 * h) * p.W + 8 * (i & 1)) * p.res_sp) * 4u;   // scalar
          if (full) {
            rv[8 * h + i] = w4_load_aux<SR_W4WS_AUX_RES>(rs_res, rl, rbase + d);
          } else {
            const bool ok = (eit.oy0 + 4 * (i >> 1) + 2 * h + o_kk < p.H) & (eit.ox0 + 4 * o_t0 + 8 * (i & 1) + o_l < p.W);
            rv[8 * h + i] = w4_load_aux<SR_W4WS_AUX_RES>(rs_res, ok ? rl : W4_OOB, rbase + d);
          }
        }
    };
    auto read_half = [&](w4_f4 (&y)[8]) {
      if (SR_W4_ABL & 4) return;
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = *reinterpret_cast<const w4_f4*>(OUT + i * 16 * 64 * 4 + (o_cx ^ (32 * i)));
    };
    // half h of the tile of item `eit`: + bias + residual, activation, stores
    auto output_half = [&](const W4Item& eit, int h, const w4_f4 (&y)[8]) {
      if (SR_W4_ABL & 4) return;
      const bool cq_ok = eit.co0 + 4 * o_c < p.Cout;
      const bool full = (eit.oy0 + 16 <= p.H) & (eit.ox0 + 16 <= p.W);   // uniform
      const __amdgpu_buffer_rsrc_t rs_out = w4_rsrc(p.out + (int64_t)eit.b * p.out_sb, out_img_bytes);
      const unsigned obase = (unsigned)((eit.oy0 * p.W + eit.ox0) * p.out_sp + eit.co0) * 4u;   // scalar
      const unsigned ol = cq_ok ? o_lane : W4_OOB;
      float o[32];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        w4_f4 v = y[i] + bv;
        if (has_res) v = v + rv[8 * h + i];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * i + e] = v[e];
      }
      if (GENERIC_ACT) {
        sr_activate_group(o, slope);
      } else if (slope >= 0.0f) {   // (uniform) LeakyReLU, 0 <= slope <= 1: max(v, slope v) -- v_pk_mul_f32 + 2 v_max_f32 per pair
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const w4_f2 t = slope * w4_f2{o[2 * i], o[2 * i + 1]};
          o[2 * i] = sr_vmax(o[2 * i], t[0]);
          o[2 * i + 1] = sr_vmax(o[2 * i + 1], t[1]);
        }
      }
      if (full) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          w4_store_aux<SR_W4WS_AUX_OUT>(w4_f4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]}, rs_out,
                                        ol + obase + (unsigned)(((4 * (i >> 1) + 2 * h) * p.W + 8 * (i & 1)) * p.out_sp) * 4u);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool ok = (eit.oy0 + 4 * (i >> 1) + 2 * h + o_kk < p.H) & (eit.ox0 + 4 * o_t0 + 8 * (i & 1) + o_l < p.W);
          const unsigned voff = ol + obase + (unsigned)(((4 * (i >> 1) + 2 * h) * p.W + 8 * (i & 1)) * p.out_sp) * 4u;
          w4_store_aux<SR_W4WS_AUX_OUT>(w4_f4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]}, rs_out, ok ? voff : W4_OOB);
        }
      }
    };

    load_next(true);                  // patch 0 -> raw[0]; patch 1 -> registers
    store_patch(0);
    load_next(1 < K);
    __syncthreads();   // raw[0] visible
    int cs = 0;        // k % S of the tick about to run
    for (int k = 0; k <= K; ++k) {
      const bool ep_now = k >= 2 && cs == (one_slab ? 0 : 1);   // (uniform) an item closed in tick k - 1: half 0 of its tile is in LDS
      const bool closing = k >= 1 && cs == 0;                  // (uniform) the M waves close an item in THIS tick
      cs = cs + 1 == p.S ? 0 : cs + 1;
      const W4Item eit = w4_item(ep);
      w4_f4 y[8];
      W4_TR(1);
      if (ep_now) {
        read_half(y);
        __syncthreads();   // B2: half 0 is in registers
        __syncthreads();   // B3: the M waves have written half 1
      }
      if (closing && !one_slab) residual_loads(eit);   // consumed in the next tick (a tick and a half of lead: L2 misses, ~3 us under load)
      // the transform FIRST: it needs nothing from memory (raw[k & 1] is in LDS since the last tick), so the patch loads requested at
      // the end of the last tick have this whole tick to land before store_patch waits for them
#ifdef SR_W4_TRACE
      unsigned long long tf_stamp[2] = {0, 0};
      if (k < K) w4_transform36_half(lds + (k & 1) * W4_RAW_FLOATS + t_rd_off, lds + (k & 1) * W4_V_FLOATS + t_wr_off, t_half != 0, tf_stamp);
      if (tid == 0 && tr_n < W4_TR_N - 4 && blockIdx.x < 16 && k < K)
        p.trace[((size_t)blockIdx.x * 2 + grp) * W4_TR_N + tr_n++] = (9ull << 56) | (tf_stamp[0] & 0xffffffffffffffull);
#else
      if (k < K) w4_transform36_half(lds + (k & 1) * W4_RAW_FLOATS + t_rd_off, lds + (k & 1) * W4_V_FLOATS + t_wr_off, t_half != 0);
#endif
      W4_TR(2);
      if (ep_now) {
        output_half(eit, 0, y);
        read_half(y);
        output_half(eit, 1, y);
        w4_cursor_advance(p, ep, step);
        W4_TR(4);
        if (one_slab) __syncthreads();   // B4: half 1 read -- the M waves may write half 0 of the next tile
      }
      if (closing && one_slab) residual_loads(w4_item(ep));   // (an item closes on every tick: only now are rv / bv free)
      if (k + 1 < K) store_patch((k + 1) & 1);
      load_next(k + 2 < K);
      W4_TR(5);
      __syncthreads();   // ---- F: end of tick k
      W4_TR(3);
    }
    {   // the last item closed in tick K
      const W4Item eit = w4_item(ep);
      w4_f4 y[8];
      read_half(y);
      __syncthreads();   // B2
      __syncthreads();   // B3
      output_half(eit, 0, y);
      read_half(y);
      output_half(eit, 1, y);
    }
  } else {
    // ================= M waves: weight stream + MFMAs + output transform
    const int m_j = lane & 15, m_kq = lane >> 4;
    const int m_sig = (0x1230 >> (m_j & 12)) & 3;
    const int m_rd_off = W4_WS_RAW2 + m_j * 16 + 4 * (m_kq ^ m_sig);
    const unsigned u_voff = (unsigned)(m_kq * p.Co_pad + 16 * wave + m_j) * 16u;
    const unsigned u_fstride = (SR_W4_ABL & 8) ? 0u : (unsigned)p.S * 4u * (unsigned)p.Co_pad * 16u;
    const unsigned u_sstride = (SR_W4_ABL & 8) ? 0u : 4u * (unsigned)p.Co_pad * 16u;
    const __amdgpu_buffer_rsrc_t rs_u = w4_rsrc(p.wu, (int64_t)36 * p.S * 4 * p.Co_pad * 16);
    // this lane's 16-byte chunk (output channels 16 wave + 4 m_kq ...) of the 8 pixel records of its tile's half
    float* const OUT = lds + W4_WS_OUT + m_j * 8 * 64 + 4 * ((4 * wave + m_kq) ^ m_j);
    w4_f4 acc[36];
    w4_f4 ua[W4WS_NA][2];
    W4Cursor cur = w4_cursor(p, first);
    int s = 0;
    w4_u_prefetch<W4WS_NA, W4WS_PD>(rs_u, u_voff, (SR_W4_ABL & 8) ? 0u : (unsigned)(64 * cur.cb) * 16u, u_fstride, ua);
    __syncthreads();   // (raw[0] visible to the T waves)
    __syncthreads();   // ---- F: end of tick 0
    for (int k = 1; k <= K; ++k) {
      W4_TR(1);
      const unsigned u_item = (SR_W4_ABL & 8) ? 0u : (unsigned)(64 * cur.cb) * 16u;
      const float* vrd = lds + ((k - 1) & 1) * W4_V_FLOATS + m_rd_off;
      if (s == 0) w4_mfma_tick<true, W4WS_NA, W4WS_PD, W4WS_YIELD>(rs_u, u_voff, u_item, u_fstride, vrd, ua, acc, lane);
      else w4_mfma_tick<false, W4WS_NA, W4WS_PD, W4WS_YIELD>(rs_u, u_voff, u_item + (unsigned)s * u_sstride, u_fstride, vrd, ua, acc, lane);
      W4_TR(2);
      if (++s == p.S) {   // (uniform) slab k - 1 closed an item
        s = 0;
        w4_cursor_advance(p, cur, step);
        if (k < K) w4_u_prefetch<W4WS_NA, W4WS_PD>(rs_u, u_voff, (SR_W4_ABL & 8) ? 0u : (unsigned)(64 * cur.cb) * 16u, u_fstride, ua);
        if (SR_W4_ABL & 4) {
          w4_f4 sum = acc[0];
#pragma unroll
          for (int f = 1; f < 36; ++f) sum = sum + acc[f];
          if (sum[0] + sum[1] + sum[2] + sum[3] == 1.2345e33f) p.out[tid] = sum[0];
          if (one_slab && k >= 2) __syncthreads();
          __syncthreads(); __syncthreads(); __syncthreads();
        } else {
          // Y = A^T M A in place: columns, then output rows 0-1 -> LDS, [F], rows 2-3 while the T waves fetch half 0, [B2], -> LDS, [B3]
          W4_TR(8);
#pragma unroll
          for (int j = 0; j < 6; ++j) w4_at_pk(acc[j], acc[6 + j], acc[12 + j], acc[18 + j], acc[24 + j], acc[30 + j]);
#pragma unroll
          for (int r = 0; r < 2; ++r) w4_at_pk(acc[6 * r], acc[6 * r + 1], acc[6 * r + 2], acc[6 * r + 3], acc[6 * r + 4], acc[6 * r + 5]);
          W4_TR(6);
          if (one_slab && k >= 2) __syncthreads();   // B4: the T waves have read half 1 of the previous tile
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int l = 0; l < 4; ++l) if (!(SR_W4_ABL & 128) || acc[0][0] == 1.2345e33f) *reinterpret_cast<w4_f4*>(OUT + (4 * r + l) * 64) = acc[6 * r + l];
          W4_TR(7);
          __syncthreads();   // ---- F: end of tick k (half 0 visible)
          W4_TR(12);
#pragma unroll
          for (int r = 2; r < 4; ++r) w4_at_pk(acc[6 * r], acc[6 * r + 1], acc[6 * r + 2], acc[6 * r + 3], acc[6 * r + 4], acc[6 * r + 5]);
          W4_TR(9);
          __syncthreads();   // B2: the T waves hold half 0
          W4_TR(10);
#pragma unroll
          for (int r = 2; r < 4; ++r)
#pragma unroll
            for (int l = 0; l < 4; ++l) if (!(SR_W4_ABL & 128) || acc[0][0] == 1.2345e33f) *reinterpret_cast<w4_f4*>(OUT + (4 * (r - 2) + l) * 64) = acc[6 * r + l];
          __syncthreads();   // B3: half 1 visible
          W4_TR(11);
        }
      } else {
        if (k < K) w4_u_prefetch<W4WS_NA, W4WS_PD>(rs_u, u_voff, u_item + (unsigned)s * u_sstride, u_fstride, ua);
        W4_TR(7);
        __syncthreads();   // ---- F: end of tick k
      }
    }
#ifdef SR_W4_TRACE
    if (tid == 0 && blockIdx.x < 16) {
      unsigned long long* t = p.trace + ((size_t)blockIdx.x * 2) * W4_TR_N + (W4_TR_N - 4);
      t[0] = tr_c0; t[1] = tr_w0; t[2] = clock64(); t[3] = wall_clock64();
    }
#endif
  }
}

// U = G g G^T per (co, ci) for the points (0, 1/2, -1/2, 2, -2, inf), in double, rounded once; stored in MFMA A-fragment
// order: element (f, s, kq, co, e) = U_f[co][16 s + 4 kq + e], f = 6 i + j (i vertical, j horizontal frequency).
__global__ void sr_wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ wu, int Co, int Ci, int S, int Co_pad) {
  const double G[6][3] = {{1.0, 0.0, 0.0},
                          {-8.0 / 15.0, -4.0 / 15.0, -2.0 / 15.0},
                          {-8.0 / 15.0, 4.0 / 15.0, -2.0 / 15.0},
                          {1.0 / 30.0, 1.0 / 15.0, 2.0 / 15.0},
                          {1.0 / 30.0, -1.0 / 15.0, 2.0 / 15.0},
                          {0.0, 0.0, 1.0}};
  const int64_t total = (int64_t)36 * S * 4 * Co_pad * 4;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 3);
    int64_t r = idx >> 2;
    const int co = (int)(r % Co_pad); r /= Co_pad;
    const int kq = (int)(r & 3); r >>= 2;
    const int s = (int)(r % S);
    const int f = (int)(r / S);
    const int ci = 16 * s + 4 * kq + e;
    double v = 0.0;
    if (co < Co && ci < Ci) {
      const float* g = w + ((int64_t)co * Ci + ci) * 9;
      const int i = f / 6, j = f % 6;
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) v += G[i][a] * (double)g[a * 3 + c] * G[j][c];
    }
    wu[idx] = (float)v;
  }
}

// CU count of the CURRENT device (cached per device id; -1 = not asked yet)
constexpr int W4_MAX_DEVICES = 64;
int w4_current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < W4_MAX_DEVICES ? dev : 0;
}
int w4_num_cus() { return sr_device_cus(); }
// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device) instead of once per launch; a device that refuses the
// size answers SR_ERR_UNSUPPORTED (the caller falls back to the F(2x2) kernel) -- ADVICE r05.
int w4_allow_lds(const void* kernel, int slot, int bytes) {
  static std::atomic<int> state[8][W4_MAX_DEVICES];   // 0: not set, 1: ok, 2: refused
  const int dev = w4_current_device();
  int st = state[slot][dev].load(std::memory_order_acquire);
  if (st == 0) {
    st = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : 2;
    if (st == 2) (void)hipGetLastError();
    state[slot][dev].store(st, std::memory_order_release);
  }
  return st == 1 ? SR_OK : SR_ERR_UNSUPPORTED;
}

inline bool w4_al16(const void* ptr) { return (((uintptr_t)ptr) & 15) == 0; }

}  // namespace

extern "C" size_t sr_wino4_packed_weight_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const size_t S = (size_t)(Cin + 15) / 16, Co_pad = (size_t)((Cout + 63) / 64) * 64;
  return 36 * S * 4 * Co_pad * 4;
}

extern "C" int sr_wino4_pack_weights(const float* weight, int Cout, int Cin, float* packed, void* stream_) {
  if (!weight || !packed || Cout <= 0 || Cin <= 0) return SR_ERR_INVALID_ARGUMENT;
  const int S = (Cin + 15) / 16, Co_pad = ((Cout + 63) / 64) * 64;
  hipLaunchKernelGGL(sr_wino4_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream_, weight, packed, Cout, Cin, S, Co_pad);
  return sr_hip_rc(hipGetLastError());
}

// Which F(4x4, 3x3) kernel form, if any, is expected to beat the F(2x2) kernel for this 3x3 / stride-1 convolution: 0 = none,
// 1 = two 4-wave workgroups per CU, 3 = the wave-specialised 8-wave workgroup.  Fitted on scripts/wino4_shape_sweep.py
// (profiles/r05_wino4_shape_sweep.txt: every 3x3 shape of the hero conv stack at batch 8 / 1 and the matching encoder's
// layer1 at 64 images, all three kernels):
//   * many work items (>= 512, the last round >= 80 % full of 256 workgroups, <= 10 % region padding): F(4x4) wins by
//     10-30 %; r05: the wave-specialised form on short slab chains (Cin <= 64), the 4-wave form on long ones; since the r06
//     rewrite the wave-specialised form at every slab count (3-5 % ahead of the 4-wave form at Cin = 128 / 192, step -0.05 ms);
//   * one round of 150-256 items (60x80 / 30x40 levels at batch 8): the wave-specialised form wins by ~16 % (one workgroup
//     per CU has no second-round tail; F(2x2)'s 8x16 regions leave more of the chip idle there);
//   * everything else (two partial rounds, < 150 items, 15x20 maps, batch 1): F(2x2) with its split-K plans stays.
// (Measured and rejected at the end of r05, after the transform waves' split transform: widening the rule to what the isolated
// sweep then favoured -- the wave-specialised form also for Cin 65..128 and for the 320-item 60x80 layers -- made the STEP slower,
// 27.17-27.19 against 26.98-27.09 ms on one box: a 158-KB workgroup shares its CU with nothing, and the decoder's three branch
// streams fill each other's tails only with the smaller kernels.)
// `mode` 0: never, 1: this rule, 2: the 4-wave form wherever the kernel applies (tests).
extern "C" int sr_conv_prefers_wino4(int B, int H, int W, int Cin, int Cout, int mode) {
  if (mode == 0 || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 4 != 0 || Cout % 4 != 0) return 0;
  if (mode == 2) return 1;
  const long regions = (long)((H + 15) / 16) * ((W + 15) / 16);
  const double util = (double)H * W / (double)(regions * 256);
  const int co_blocks = (Cout + 63) / 64;
  const double co_util = (double)Cout / (double)(co_blocks * 64);
  const long items = regions * B * co_blocks;
  const long cus = w4_num_cus();
  if (co_util < 0.99 || Cin < 16) return 0;
  if (items >= 2 * cus) {
    const long rounds = (items + cus - 1) / cus;
    const double fill = (double)items / (double)(rounds * cus);
    if (util < 0.9 || fill < 0.8) return 0;
    return 3;   // (r05: the 4-wave form for Cin > 64; the r06 wave-specialised form wins at every slab count -- profiles/r06_wino4_shape_sweep.txt)
  }
  if (items * 10 >= cus * 6 && items <= cus && util >= 0.75) return 3;   // one round on >= 60 % of the CUs
  return 0;
}

#ifdef SR_W4_TRACE
static unsigned long long* w4_trace_buf = nullptr;
static int w4_trace_launches = 0;
static void w4_trace_begin(SrWino4Params& p, hipStream_t stream) {
  const size_t trace_n = (size_t)16 * 2 * W4_TR_N;
  if (!w4_trace_buf) (void)hipMalloc((void**)&w4_trace_buf, trace_n * 8);
  (void)hipMemsetAsync(w4_trace_buf, 0, trace_n * 8, stream);
  p.trace = w4_trace_buf;
}
static void w4_trace_end(int blocks, hipStream_t stream) {
  const size_t trace_n = (size_t)16 * 2 * W4_TR_N;
  const char* at = getenv("SR_W4_TRACE_LAUNCH");
  if (++w4_trace_launches != (at ? atoi(at) : 10)) return;
  (void)hipStreamSynchronize(stream);
  unsigned long long* host = (unsigned long long*)malloc(trace_n * 8);
  (void)hipMemcpy(host, w4_trace_buf, trace_n * 8, hipMemcpyDeviceToHost);
  for (int b = 0; b < 16 && b < blocks; b += 5)
    for (int g = 0; g < 2; ++g) {
      fprintf(stderr, "W4TRACE block %d group %d:", b, g);
      unsigned long long t0 = host[((size_t)b * 2 + 0) * W4_TR_N] & 0xffffffffffffffull, prev = 0;
      for (int i = 0; i < W4_TR_N; ++i) {
        const unsigned long long v = host[((size_t)b * 2 + g) * W4_TR_N + i];
        if (!v) break;
        const unsigned long long t = (v & 0xffffffffffffffull) - t0;
        fprintf(stderr, " %d@%llu(+%llu)", (int)(v >> 56), t, t - prev);
        prev = t;
      }
      fprintf(stderr, "\n");
    }
  {   // shader clock of workgroup 0 over its whole run: clock64 ticks per wall_clock64 tick (100 MHz)
    const unsigned long long* t = host + (W4_TR_N - 4);
    if (t[3] > t[1]) fprintf(stderr, "W4CLOCK %.3f GHz over %.1f us\n", 0.1 * (double)(t[2] - t[0]) / (double)(t[3] - t[1]),
                             (double)(t[3] - t[1]) * 0.01);
  }
  free(host);
}
#endif

static int w4_run(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_u, const float* bias,
                  const float* residual, int64_t res_batch_stride, int res_pix_stride, float* out, int64_t out_batch_stride,
                  int out_pix_stride, int B, int H, int W, int Cin, int Cout, float leaky_slope, int variant, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !packed_u || !out) return SR_ERR_INVALID_ARGUMENT;
  // 16-byte channel quads everywhere (the vector staging / epilogue is the only instantiation)
  if (Cin % 4 != 0 || Cout % 4 != 0 || !w4_al16(in) || !w4_al16(out) || in_pix_stride % 4 != 0 || in_batch_stride % 4 != 0 ||
      out_pix_stride % 4 != 0 || out_batch_stride % 4 != 0 || (bias && !w4_al16(bias)) ||
      (residual && (!w4_al16(residual) || res_pix_stride % 4 != 0 || res_batch_stride % 4 != 0)))
    return SR_ERR_UNSUPPORTED;
  const int64_t lim = (int64_t)1 << 31;   // per-image byte offsets are 32-bit (buffer addressing)
  if (((int64_t)(H * (int64_t)W - 1) * in_pix_stride + Cin) * 4 >= lim || ((int64_t)(H * (int64_t)W - 1) * out_pix_stride + Cout) * 4 >= lim ||
      (residual && ((int64_t)(H * (int64_t)W - 1) * res_pix_stride + Cout) * 4 >= lim))
    return SR_ERR_UNSUPPORTED;
  SrWino4Params p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.wu = packed_u; p.bias = bias;
  p.res = residual; p.res_sb = res_batch_stride; p.res_sp = res_pix_stride;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.Co_pad = ((Cout + 63) / 64) * 64;
  p.S = (Cin + 15) / 16;
  if ((int64_t)36 * p.S * 4 * p.Co_pad * 16 >= lim) return SR_ERR_UNSUPPORTED;
  p.regions_x = (W + 15) / 16;
  p.regions_y = (H + 15) / 16;
  p.co_blocks = p.Co_pad / 64;
  const int64_t total = (int64_t)p.regions_x * p.regions_y * p.co_blocks * B;
  if (total >= lim) return SR_ERR_UNSUPPORTED;
  p.total = (int)total;
  p.slope = leaky_slope;
  if (variant == 3) {   // wave-specialised: 4 MFMA waves + 4 transform waves, one workgroup per CU
    int blocks = w4_num_cus();
#ifdef SR_W4WS_MAXBLOCKS   // (probe builds: how much of an item's time is chip-wide contention -- profiles/r06_w4ws_trace.txt section 11)
    if (blocks > SR_W4WS_MAXBLOCKS) blocks = SR_W4WS_MAXBLOCKS;
#endif
    if (blocks > p.total) blocks = p.total;
    const bool generic_act = !((leaky_slope >= 0.0f && leaky_slope <= 1.0f) || (leaky_slope < 0.0f && leaky_slope > -1.5f));
    auto kernel = generic_act ? (residual ? sr_wino4ws_kernel<true, true> : sr_wino4ws_kernel<true, false>)
                              : (residual ? sr_wino4ws_kernel<false, true> : sr_wino4ws_kernel<false, false>);
    if (int rc = w4_allow_lds((const void*)kernel, (generic_act ? 2 : 0) + (residual ? 1 : 0), W4_WS_LDS_BYTES)) return rc;
#ifdef SR_W4_TRACE
    w4_trace_begin(p, (hipStream_t)stream_);
#endif
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), W4_WS_LDS_BYTES, (hipStream_t)stream_, p);
#ifdef SR_W4_TRACE
    w4_trace_end(blocks, (hipStream_t)stream_);
#endif
    return sr_hip_rc(hipGetLastError());
  }
  if (variant != 2) {   // two independent 4-wave workgroups per CU
    int blocks = 2 * w4_num_cus();
    if (blocks > p.total) blocks = p.total;
    if (int rc = w4_allow_lds((const void*)sr_wino4_kernel, 4, W4_LDS_BYTES)) return rc;
    hipLaunchKernelGGL(sr_wino4_kernel, dim3(blocks), dim3(256), W4_LDS_BYTES, (hipStream_t)stream_, p);
    return sr_hip_rc(hipGetLastError());
  }
  int blocks = w4_num_cus();
  if (blocks > (p.total + 1) / 2) blocks = (p.total + 1) / 2;
#ifdef SR_W4_TRACE
  w4_trace_begin(p, (hipStream_t)stream_);
#endif
  if (int rc = w4_allow_lds((const void*)sr_wino4pp_kernel, 5, 2 * W4_LDS_BYTES)) return rc;
  hipLaunchKernelGGL(sr_wino4pp_kernel, dim3(blocks), dim3(512), 2 * W4_LDS_BYTES, (hipStream_t)stream_, p);
#ifdef SR_W4_TRACE
  w4_trace_end(blocks, (hipStream_t)stream_);
#endif
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_conv3x3_wino4_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_u,
                                         const float* bias, const float* residual, int64_t res_batch_stride,
                                         int res_pix_stride, float* out, int64_t out_batch_stride, int out_pix_stride, int B,
                                         int H, int W, int Cin, int Cout, float leaky_slope, void* stream_) {
  return w4_run(in, in_batch_stride, in_pix_stride, packed_u, bias, residual, res_batch_stride, res_pix_stride, out,
                out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, leaky_slope, 0, stream_);
}

// `variant` 0: the default form (= sr_conv3x3_wino4_nhwc_fwd), 1: two independent 4-wave workgroups per CU, 2: one 8-wave
// ping-pong workgroup per CU.  All forms compute every output with the same operations in the same order: bit-identical
// results (tests/test_gpu_wino4.py).
extern "C" int sr_conv3x3_wino4_variant_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                                 const float* packed_u, const float* bias, const float* residual,
                                                 int64_t res_batch_stride, int res_pix_stride, float* out,
                                                 int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                                 int Cout, float leaky_slope, int variant, void* stream_) {
  if (variant < 0 || variant > 3) return SR_ERR_INVALID_ARGUMENT;
  return w4_run(in, in_batch_stride, in_pix_stride, packed_u, bias, residual, res_batch_stride, res_pix_stride, out,
                out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, leaky_slope, variant, stream_);
}
