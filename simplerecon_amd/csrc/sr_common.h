// sr_common.h -- shared device/host helpers for libsimplerecon_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "simplerecon_hip.h"

// Per-(batch, view) geometry record written by sr_geom_kernel into the workspace:
//   [0,12)  rows 0..2 of P = K_src @ T_src_cur      (reference utils/geometry_utils.py:78-80)
//   [12,15) t = T_cur_src[:3,3]  (source camera centre in the reference frame; cost_volume.py:654-669)
//   [15]    pose_dist  [16] R_measure  [17] t_measure  (geometry_utils.py:178-191)
#define SR_GEOM_STRIDE 20

#define SR_WAVE 64

// value of a run-time switch (include/simplerecon_hip.h, SR_OPT_*): one relaxed atomic load; the environment is read once, at the
// first access in the process (sr_options.hip)
int sr_opt(int id);

// CU count of the CURRENT device, cached per device index (sr_options.hip); 256 when the runtime cannot be asked
int sr_device_cus();

static inline int sr_hip_rc(hipError_t e) { return e == hipSuccess ? SR_OK : SR_ERR_HIP_BASE + (int)e; }

static inline size_t sr_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct SrPlanes {
  const float* ptr;
  int64_t sb, sd, sy, sx;
};

struct SrVolumeOut {
  float* cv;
  int64_t sb, sd, sp;
  float* lowest;   // [B,h,w] or null
  uint8_t* mask;   // [B,h,w] or null
};

// ----------------------------------------------------------------------------------------
// Geometry shared by the dot-product and the MLP sweep.  Written op-by-op in the reference's
// order with FP contraction OFF so that pixel coordinates, z', masks and bilinear weights are
// bit-identical to the fp32 CPU oracle (discontinuous decisions cannot flip).
// ----------------------------------------------------------------------------------------
struct SrSample {
  float zp;            // z' = q_z + 1e-8                    (geometry_utils.py:84)
  float pix_x, pix_y;  // q_x * s, q_y * s                   (geometry_utils.py:85-87)
  float w_nw, w_ne, w_sw, w_se;  // bilinear weights, already zeroed for out-of-image taps
  int o_nw, o_ne, o_sw, o_se;    // texel indices (y*w + x), clamped in-image
};

typedef float sr_f2v __attribute__((ext_vector_type(2)));

// Integer tap origin (floats, unclamped) and the four bilinear weights of the unnormalised sampling position (ix, iy)
// (grid_sample, bilinear, zeros padding, align_corners=False): weights of out-of-image taps are zero.
__device__ __forceinline__ void sr_bilinear_taps(float ix, float iy, int h, int w, float& fx0, float& fy0, float& w_nw,
                                                 float& w_ne, float& w_sw, float& w_se) {
#pragma clang fp contract(off)
  const sr_f2v ixy = {ix, iy};
  fx0 = floorf(ix);
  fy0 = floorf(iy);
  const sr_f2v f0 = {fx0, fy0};
  const sr_f2v f1 = f0 + 1.0f;
  const float fx1 = f1.x, fy1 = f1.y;
  const float wm = (float)(w - 1), hm = (float)(h - 1);
  // (bitwise & on purpose: short-circuit && makes hipcc emit divergent branches that split the
  // scheduling region the callers want to interleave with MFMAs)
  const bool vx0 = (fx0 >= 0.0f) & (fx0 <= wm), vx1 = (fx1 >= 0.0f) & (fx1 <= wm);
  const bool vy0 = (fy0 >= 0.0f) & (fy0 <= hm), vy1 = (fy1 >= 0.0f) & (fy1 <= hm);
  const sr_f2v a1 = f1 - ixy, a0 = ixy - f0;            // (ax1, ay1), (ax0, ay0)
  const sr_f2v wtop = sr_f2v{a1.x, a0.x} * a1.y;        // (ax1*ay1, ax0*ay1)
  const sr_f2v wbot = sr_f2v{a1.x, a0.x} * a0.y;        // (ax1*ay0, ax0*ay0)
  w_nw = (vx0 & vy0) ? wtop.x : 0.0f;
  w_ne = (vx1 & vy0) ? wtop.y : 0.0f;
  w_sw = (vx0 & vy1) ? wbot.x : 0.0f;
  w_se = (vx1 & vy1) ? wbot.y : 0.0f;
}

// Core of the projection: everything up to the bilinear weights, with the UNCLAMPED integer tap origin (fx0, fy0) as
// floats.  sr_project_sample (clamped global texel offsets) and the LDS-staged sweep (tile footprints) both build on it,
// so their geometry is the same instruction sequence.
// x / y components run as packed fp32 pairs (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, FP contraction off, so
// every value is bit-identical to the scalar formulation) -- half the VALU instructions of the projection.
struct SrSampleXY {
  float zp, pix_x, pix_y;
  float ix, iy;                  // unnormalised sampling position (texel units; may be anything, incl. NaN)
  float fx0, fy0;                // its floor = the NW tap
  float w_nw, w_ne, w_sw, w_se;  // bilinear weights, already zeroed for out-of-image taps
};

__device__ __forceinline__ void sr_project_sample_xy(const float* __restrict__ g /*geom record*/,
                                                     float X0, float X1, float X2, int h, int w,
                                                     float inv_w, float inv_h, SrSampleXY& s) {
#pragma clang fp contract(off)
  const float eps = 1e-8f;
  const sr_f2v q01 = sr_f2v{g[0], g[4]} * X0 + sr_f2v{g[1], g[5]} * X1 + sr_f2v{g[2], g[6]} * X2 + sr_f2v{g[3], g[7]};
  const float q2 = g[8] * X0 + g[9] * X1 + g[10] * X2 + g[11];
  s.zp = q2 + eps;
  const float sc = (fabsf(q2) > eps) ? 1.0f / s.zp : 1.0f;
  const sr_f2v pix = q01 * sc;
  s.pix_x = pix.x;
  s.pix_y = pix.y;
  // uv = 2*pix*(1/w,1/h) - 1 (cost_volume.py:199); grid_sample unnormalise, align_corners=False
  const sr_f2v uv = 2.0f * pix * sr_f2v{inv_w, inv_h} - 1.0f;
  const sr_f2v size = {(float)w, (float)h};
  const sr_f2v ixy = ((uv + 1.0f) * size - 1.0f) / 2.0f;
  s.ix = ixy.x;
  s.iy = ixy.y;
  sr_bilinear_taps(s.ix, s.iy, h, w, s.fx0, s.fy0, s.w_nw, s.w_ne, s.w_sw, s.w_se);
}

__device__ __forceinline__ void sr_project_sample(const float* __restrict__ g /*geom record*/,
                                                  float X0, float X1, float X2, int h, int w,
                                                  float inv_w, float inv_h, SrSample& s) {
  SrSampleXY c;
  sr_project_sample_xy(g, X0, X1, X2, h, w, inv_w, inv_h, c);
  s.zp = c.zp; s.pix_x = c.pix_x; s.pix_y = c.pix_y;
  s.w_nw = c.w_nw; s.w_ne = c.w_ne; s.w_sw = c.w_sw; s.w_se = c.w_se;
  const float wm = (float)(w - 1), hm = (float)(h - 1);
  const float fx1 = c.fx0 + 1.0f, fy1 = c.fy0 + 1.0f;
  // clamp (NaN-safe: fmaxf(NaN, 0) = 0) so every tap address is in-image; weight 0 kills it
  const int x0 = (int)fminf(fmaxf(c.fx0, 0.0f), wm), x1 = (int)fminf(fmaxf(fx1, 0.0f), wm);
  const int y0 = (int)fminf(fmaxf(c.fy0, 0.0f), hm), y1 = (int)fminf(fmaxf(fy1, 0.0f), hm);
  s.o_nw = y0 * w + x0;
  s.o_ne = y0 * w + x1;
  s.o_sw = y1 * w + x0;
  s.o_se = y1 * w + x1;
}

// bounds test of get_mask (cost_volume.py:77-97)
__device__ __forceinline__ bool sr_in_bounds(const SrSample& s, int h, int w) {
  return (s.pix_x > 2.0f) & (s.pix_x < (float)(w - 2)) & (s.pix_y > 2.0f) & (s.pix_y < (float)(h - 2));
}

// Activation code carried by the `leaky_slope` argument of the convolution entry points (simplerecon_hip.h): a value
// >= 0 is the LeakyReLU slope (0 = ReLU), SR_ACT_NONE (any value in (-1.5, 0)) the identity, SR_ACT_SILU x * sigmoid(x).
#ifdef __HIPCC__
// max / min as ONE instruction.  fmaxf() compiles to two: hipcc first canonicalises an operand it cannot prove quiet
// (v_max x, x, x) although in IEEE mode the instruction itself already returns a quiet result -- and it folds
// v_med3(a, b, inf) back into the same thing, so this is inline asm.  The compiler's hazard recogniser does not look
// inside inline asm: a (non-MFMA) VALU write needs 2 wait states before an MFMA reads the register (measured: wrong
// volumes in the W2-streaming MLP sweep without them), hence the `s_nop 1` in the _mfma form.  The plain form is for
// values that go to VALU arithmetic / stores.  For ordinary computed values (activations): NaN handling = v_max's.
__device__ __forceinline__ float sr_vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sr_vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sr_vmax_mfma(float a, float b) {   // result may be an MFMA operand
  float r;
  asm("v_max_f32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sr_activate(float v, float slope) {
  if (slope >= 0.0f) return fmaxf(v, 0.0f) + slope * fminf(v, 0.0f);
  if (slope < -1.5f) return v / (1.0f + __expf(-v));
  return v;
}
// A wave-uniform float moved to a scalar register, so that a test on it is ONE s_cmp / s_cbranch instead of a vector
// compare + branch per value it guards.
__device__ __forceinline__ float sr_uniform(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
// sr_activate on a GROUP of values with the activation code tested once per group (r03: per value, the 32 outputs a
// Winograd epilogue thread owns cost 111 conditional branches around the SiLU code per region).  `slope` must be
// wave-uniform (pass it through sr_uniform).  LeakyReLU with 0 <= slope <= 1 is max(v, slope v) -- the same value as
// max(v,0) + slope min(v,0) for every finite v, in 1.5 instead of 3 instructions per value (v_pk_mul_f32 + v_max_f32).
template <int N>
__device__ __forceinline__ void sr_activate_group(float (&v)[N], float slope) {
  if (slope >= 0.0f) {
    if (slope <= 1.0f) {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = sr_vmax(v[i], slope * v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = sr_vmax(v[i], 0.0f) + slope * sr_vmin(v[i], 0.0f);
    }
  } else if (slope < -1.5f) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] / (1.0f + __expf(-v[i]));
  }
}
#endif

// launch parameters of the dot-product sweeps (sr_dot_volume.hip, sr_dot_volume_lds.hip)
struct SrDotParams {
  const float* cur;       // [B,C,h,w]
  const float* src_nhwc;  // [B*K, h*w, C]
  const float* invK;      // [B,16]
  const float* geom;      // [B*K, SR_GEOM_STRIDE]
  SrPlanes planes;
  SrVolumeOut out;
  int B, K, h, w, D;
  float inv_w, inv_h;
};

// workspace carving: [geom records | channels-last source features | per-pixel argmax keys of the LDS-staged sweep]
static inline float* sr_ws_geom(void* workspace) { return (float*)sr_align_up((size_t)workspace, 256); }
static inline float* sr_ws_src_nhwc(void* workspace, int B, int K) {
  return (float*)((char*)sr_ws_geom(workspace) + sr_align_up((size_t)B * K * SR_GEOM_STRIDE * sizeof(float), 256));
}

static inline size_t sr_ws_nhwc_bytes(int B, int K, int C, int h, int w) {
  return sr_align_up((size_t)B * K * h * w * C * sizeof(float), 256);
}
static inline unsigned long long* sr_ws_keys(void* workspace, int B, int K, int C, int h, int w) {
  return (unsigned long long*)((char*)sr_ws_src_nhwc(workspace, B, K) + sr_ws_nhwc_bytes(B, K, C, h, w));
}

// internal launchers shared between translation units
// LDS-staged 16-channel dot-product sweep (sr_dot_volume_lds.hip); SR_ERR_UNSUPPORTED when the shape is outside its
// range (the caller then takes the L1-gather kernels)
int sr_launch_dot_volume_lds(const SrDotParams& p, unsigned long long* keys, hipStream_t stream);
int sr_launch_geom(const float* K_src, const float* T_src_cur, const float* T_cur_src, float* geom,
                   int n, hipStream_t stream);
int sr_launch_argmax_planes(const float* cv, int64_t sb, int64_t sd, int64_t sp, SrPlanes planes, int B, int h, int w,
                            int D, float* lowest, hipStream_t stream);
// out = act(sum_k part[k * part_stride + ...] + bias + res): finish of a split-K convolution (partials dense
// channels-last [B, HW, Cout], Cout % 4 == 0, 16-byte aligned operands); defined in sr_wino.hip
int sr_launch_splitk_reduce(const float* part, int ksplit, int64_t part_stride, const float* bias, const float* res,
                            int64_t res_sb, int res_sp, float* out, int64_t out_sb, int out_sp, int B, int HW, int Cout,
                            float slope, hipStream_t stream);
// [images, npix, C] -> [images, C, npix] (C % 4 == 0); defined in sr_dot_volume_bwd.hip
int sr_launch_unpack_nhwc(const float* src_nhwc, float* dst_nchw, int images, int C, int npix, hipStream_t stream);
int sr_launch_pack_nhwc(const float* src_nchw, float* dst_nhwc, int images, int C, int npix,
                        hipStream_t stream);
