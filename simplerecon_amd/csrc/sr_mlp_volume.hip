// sr_mlp_volume.hip -- fused plane sweep of the metadata-MLP feature volume for gfx950.
//
// Replaces FeatureVolumeManager.build_cost_volume / FastFeatureVolumeManager of the reference
// (modules/cost_volume.py:451-736, 967-1164): per depth plane ~40 ATen launches, a 202-channel
// concat tensor and two [B*N,128] hidden tensors (or, in the "fast" variant, 7.9 GB of them at
// batch 8).  Here a wavefront owns 64 points (64 pixels of one depth plane):
//   * VALU part: homography, 4-tap bilinear gather of the channels-last source features,
//     dot / mask / depth / rays / ray angle -- each lane builds the MLP input vector of ITS point
//     in registers, never in memory;
//   * MFMA part (fp32 in, fp32 accumulate, v_mfma_f32_32x32x2_f32): H1^T = W1 . X^T and
//     H2^T = W2 . H1^T as two 32-point column groups.  One v_permlane32_swap per k-step turns two
//     per-lane feature registers into the B operands of both groups; the layer-1 accumulators ARE
//     the layer-2 B operands (the k order of W2 is permuted to the MFMA C layout on the host side),
//     so activations never touch LDS or HBM.  Layer 3 (128 -> 1) is a per-lane dot + one swap.
//   * W1 (packed in k-step order) lives in LDS for the whole persistent block; W2 streams from
//     L2 with a register prefetch (both do not fit the 160 KB LDS in fp32).
#include <stdlib.h>
#include <string.h>

#include "sr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SR_HID 128          // hidden width of the matching MLP (reference cost_volume.py:402)
// Layer-1 inputs are split into a depth-VARIANT part (per view: 16 warped channels, mask, z', dot, ray
// angle, source ray(3) and one spare slot that carries the plane depth d for view 0) and a depth-INVARIANT
// part (reference features 16, reference ray 3, the constant 1 that carries the bias, 3 pose measures per
// view).  The invariant part is multiplied ONCE per (64-pixel tile, chunk of planes) and re-used as the
// accumulator initialiser of every plane of the chunk (-17 % layer-1 MFMA work).
#define SR_VIEW_SLOTS 24
#define SR_INV_FIXED 20     // cur(16), cur_ray(3), one
#define SR_PLANE_CHUNK 8    // planes per work unit

static inline int sr_mlp_steps_var(int K) { return (SR_VIEW_SLOTS / 2) * K; }
static inline int sr_mlp_steps_inv(int K) { return (SR_INV_FIXED + 3 * K + 1) / 2; }
static inline int sr_mlp_steps1(int K) { return sr_mlp_steps_var(K) + sr_mlp_steps_inv(K); }
#define SR_MLP_STEPS2 65    // 64 hidden pairs + one bias step

// packed parameter block (floats): [W1p: steps1*256 (variant steps first)][W2p: 65*256][w3tab: 128][b3: 1][pad]
static inline size_t sr_mlp_packed_floats(int K) { return (size_t)(sr_mlp_steps1(K) + SR_MLP_STEPS2) * 256 + 128 + 4; }

// ------------------------------------------------------------------ weight packing ----
// W?p[t][lane][mt] = W[row = 32*mt + (lane&31)][column of k-slot (t, half = lane>>5)]
__global__ void sr_mlp_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                   const float* __restrict__ W2, const float* __restrict__ b2,
                                   const float* __restrict__ W3, const float* __restrict__ b3,
                                   float* __restrict__ packed, int K, int C) {
  const int steps_var = (SR_VIEW_SLOTS / 2) * K;
  const int steps1 = steps_var + (SR_INV_FIXED + 3 * K + 1) / 2;
  const int Cin = C * (K + 1) + 10 * K + 4;
  // reference channel offsets (cost_volume.py:709-723)
  const int o_cur = K * C, o_mask = o_cur + C, o_z = o_mask + K, o_d = o_z + K, o_dot = o_d + 1;
  const int o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3, o_pd = o_sray + 3 * K;
  const int o_rm = o_pd + K, o_tm = o_rm + K;
  const int total = (steps1 + SR_MLP_STEPS2) * 256 + 128 + 1;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (e < steps1 * 256) {
      const int t = e >> 8, lane = (e >> 2) & 63, mt = e & 3;
      const int row = 32 * mt + (lane & 31);
      int col = -1;  // -1: zero, -2: bias
      if (t < steps_var) {
        const int slot = 2 * t + (lane >> 5);
        const int k = slot / SR_VIEW_SLOTS, s = slot - k * SR_VIEW_SLOTS;
        if (s < 16) col = k * C + s;
        else if (s == 16) col = o_mask + k;
        else if (s == 17) col = o_z + k;
        else if (s == 18) col = o_dot + k;
        else if (s == 19) col = o_ang + k;
        else if (s < 23) col = o_sray + 3 * k + (s - 20);
        else col = (k == 0) ? o_d : -1;
      } else {
        const int u = 2 * (t - steps_var) + (lane >> 5);
        if (u < 16) col = o_cur + u;
        else if (u < 19) col = o_cray + (u - 16);
        else if (u == 19) col = -2;
        else if (u < SR_INV_FIXED + 3 * K) {
          const int k = (u - SR_INV_FIXED) / 3, i = (u - SR_INV_FIXED) - 3 * k;
          col = (i == 0 ? o_pd : (i == 1 ? o_rm : o_tm)) + k;
        }
      }
      v = (col >= 0) ? W1[(size_t)row * Cin + col] : (col == -2 ? b1[row] : 0.0f);
    } else if (e < (steps1 + SR_MLP_STEPS2) * 256) {
      const int e2 = e - steps1 * 256;
      const int t = e2 >> 8, lane = (e2 >> 2) & 63, mt = e2 & 3;
      const int row = 32 * mt + (lane & 31);
      if (t < 64) {
        // k-step t = (m, r): hidden feature held by this half in accumulator register r of tile m
        const int m = t >> 4, r = t & 15;
        const int f = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        v = W2[(size_t)row * SR_HID + f];
      } else {
        v = (lane < 32) ? b2[row] : 0.0f;  // bias step: B operand is 1 in half 0
      }
    } else if (e < (steps1 + SR_MLP_STEPS2) * 256 + 128) {
      // w3tab[half][mt][r] = W3[feature of accumulator register r of tile mt in that half]
      const int i = e - (steps1 + SR_MLP_STEPS2) * 256;
      const int half = i >> 6, mt = (i >> 4) & 3, r = i & 15;
      v = W3[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half];
    } else {
      v = b3[0];
    }
    packed[e] = v;
  }
}

// ------------------------------------------------------------------ the sweep ---------

struct SrMlpParams {
  const float* cur;       // [B,16,h,w]
  const float* src_nhwc;  // [B*K, h*w, 16]
  const float* invK;      // [B,16]
  const float* geom;      // [B*K, SR_GEOM_STRIDE]
  const float* packed;    // sr_mlp_pack_kernel output
  SrPlanes planes;
  SrVolumeOut out;
  int B, K, h, w, D;
  int tiles;              // ceil(h*w / 64)
  int chunk, chunks;      // planes per work unit, ceil(D / chunk)
  float inv_w, inv_h, slope;
  int debug;              // ablation bits (env SR_MLP_DEBUG), 0 in production
  int xcd_order;          // 1: each XCD (workgroup index mod 8) sweeps its own contiguous eighth of the work units
  int vec_store;          // channels-last volume (plane stride 1): a lane keeps the costs of its unit's planes and
                          // stores them as 16-byte pieces at the end of the unit (instead of one 4-byte store per plane
                          // into 64 different 256-byte rows: r02 PMC counted 6.7x the volume's bytes in WRITE_SIZE)
};

__device__ __forceinline__ void sr_swap_halves(float fa, float fb, float& bP, float& bQ) {
  // vdst lanes 32-63 <-> src lanes 0-31: bP = [fa(0..31) | fb(0..31)], bQ = [fa(32..63) | fb(32..63)]
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);
  bP = __uint_as_float(r[0]);
  bQ = __uint_as_float(r[1]);
}

#define SR_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void sr_l1_step(f32x16 (&acc)[2][4], const float4 wA, float fa, float fb) {
  float bP, bQ;
  sr_swap_halves(fa, fb, bP, bQ);
  acc[0][0] = SR_MFMA(wA.x, bP, acc[0][0]);
  acc[1][0] = SR_MFMA(wA.x, bQ, acc[1][0]);
  acc[0][1] = SR_MFMA(wA.y, bP, acc[0][1]);
  acc[1][1] = SR_MFMA(wA.y, bQ, acc[1][1]);
  acc[0][2] = SR_MFMA(wA.z, bP, acc[0][2]);
  acc[1][2] = SR_MFMA(wA.z, bQ, acc[1][2]);
  acc[0][3] = SR_MFMA(wA.w, bP, acc[0][3]);
  acc[1][3] = SR_MFMA(wA.w, bQ, acc[1][3]);
}

// first k-step of a plane: D = A*B + hc (the hoisted depth-invariant part), no accumulator copy
__device__ __forceinline__ void sr_l1_step_init(f32x16 (&acc)[2][4], const f32x16 (&hc)[2][4], const float4 wA,
                                                float fa, float fb) {
  float bP, bQ;
  sr_swap_halves(fa, fb, bP, bQ);
  acc[0][0] = SR_MFMA(wA.x, bP, hc[0][0]);
  acc[1][0] = SR_MFMA(wA.x, bQ, hc[1][0]);
  acc[0][1] = SR_MFMA(wA.y, bP, hc[0][1]);
  acc[1][1] = SR_MFMA(wA.y, bQ, hc[1][1]);
  acc[0][2] = SR_MFMA(wA.z, bP, hc[0][2]);
  acc[1][2] = SR_MFMA(wA.z, bQ, hc[1][2]);
  acc[0][3] = SR_MFMA(wA.w, bP, hc[0][3]);
  acc[1][3] = SR_MFMA(wA.w, bQ, hc[1][3]);
}

// Ablation switches (env SR_MLP_DEBUG) exist only in -DSR_MLP_ABLATION builds; in production they are compile-time 0.
#ifdef SR_MLP_ABLATION
#define SR_MLP_DBG(bit) (p.debug & (bit))
#else
#define SR_MLP_DBG(bit) 0
#endif

#ifndef SR_MLP_NT_TAPS
#define SR_MLP_NT_TAPS 0
#endif
#define SR_LDS_W3_FLOATS 256  // w3tab (128) + b3 + pad, in front of W1 in LDS

template <bool W1_LDS, bool W2_LDS>
__global__ __launch_bounds__(256, 1) void sr_mlp_volume_kernel(SrMlpParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int C = 16;
  constexpr int VS = SR_VIEW_SLOTS / 2;  // k-steps per view
  const int lane = threadIdx.x & 63;
  const int steps_var = VS * p.K;
  const int steps1 = steps_var + (SR_INV_FIXED + 3 * p.K + 1) / 2;
  const float4* gW1 = reinterpret_cast<const float4*>(p.packed);
  const float4* gW1inv = gW1 + (size_t)steps_var * 64;
  const float4* gW2 = gW1 + (size_t)steps1 * 64;
  const float* gW3 = p.packed + (size_t)(steps1 + SR_MLP_STEPS2) * 256;

  for (int i = threadIdx.x; i < 129; i += blockDim.x) lds[i] = gW3[i];
  // LDS: [w3tab + b3 | W2 (65 KB, when it fits) | depth-variant part of W1 (12*K KB, when it fits)]
  constexpr int W2_FLOATS = W2_LDS ? SR_MLP_STEPS2 * 256 : 0;
  if (W2_LDS) {
    float4* l4 = reinterpret_cast<float4*>(lds + SR_LDS_W3_FLOATS);
    for (int i = threadIdx.x; i < SR_MLP_STEPS2 * 64; i += blockDim.x) l4[i] = gW2[i];
  }
  if (W1_LDS) {
    float4* l4 = reinterpret_cast<float4*>(lds + SR_LDS_W3_FLOATS + W2_FLOATS);
    for (int i = threadIdx.x; i < steps_var * 64; i += blockDim.x) l4[i] = gW1[i];
  }
  __syncthreads();
  const float4* W1p = W1_LDS ? reinterpret_cast<const float4*>(lds + SR_LDS_W3_FLOATS + W2_FLOATS) : gW1;

  const int N = p.h * p.w;
  const long nunits = (long)p.B * p.tiles * p.chunks;
  // XCD-aware work order (r04): workgroup b runs on XCD b % 8 and each XCD has its own 4 MB L2.  With units dealt round-robin
  // every XCD touches every region of every source image (41 MB per keyframe at 15 views) and the tap reads miss its L2
  // again and again (r03 PMC, cfg5: 1.2 GB of fabric traffic per launch for 244 MB of compulsory bytes).  Units are
  // ordered (image, pixel tile, plane chunk): give XCD x the contiguous eighth [x U8, (x + 1) U8) -- a band of pixel tiles
  // whose tap footprints overlap -- and let its waves stride through that band.
  long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  long unit_end = nunits;
  if (p.xcd_order && (gridDim.x & 7) == 0) {
    const long u8 = (nunits + 7) / 8;
    const int xcd = blockIdx.x & 7;
    nwaves = (long)(gridDim.x >> 3) * (blockDim.x >> 6);
    wave0 = xcd * u8 + (long)(blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unit_end = min(nunits, (xcd + 1) * u8);
  }
  const int half = lane >> 5;

  for (long unit = wave0; unit < unit_end; unit += nwaves) {
    const int chunk = (int)(unit % p.chunks);
    const long tb = unit / p.chunks;
    const int tile = (int)(tb % p.tiles);
    const int b = (int)(tb / p.tiles);
    const int j0 = chunk * p.chunk, j1 = min(p.D, j0 + p.chunk);
    const int pix = tile * 64 + lane;
    const bool active = pix < N;
    const int pc = active ? pix : N - 1;
    const int y = pc / p.w, x = pc - y * p.w;
    const float* plane_ptr = p.planes.ptr + b * p.planes.sb + y * p.planes.sy + x * p.planes.sx;
    const float* geom_b = p.geom + (size_t)b * p.K * SR_GEOM_STRIDE;
    const float* src_b = p.src_nhwc + (size_t)b * p.K * N * C;

    float cur[C];
#pragma unroll
    for (int c = 0; c < C; ++c) cur[c] = p.cur[((size_t)b * C + c) * N + pc];

    float r0, r1, r2;
    {
#pragma clang fp contract(off)
      const float* iK = p.invK + 16 * (size_t)b;
      const float px = (float)x + 0.5f, py = (float)y + 0.5f;
      r0 = iK[0] * px + iK[1] * py + iK[2];
      r1 = iK[4] * px + iK[5] * py + iK[6];
      r2 = iK[8] * px + iK[9] * py + iK[10];
    }
    // current-frame ray F.normalize(X) (cost_volume.py:641-651, eps 1e-12): X = d * r, so the ray does not depend
    // on the plane beyond rounding; it is formed from plane 0 for every chunk (so results do not depend on how
    // planes are chunked into work units) and shared by all planes.
    float cr0, cr1, cr2, crn0, crn1, crn2;
    {
#pragma clang fp contract(off)
      const float d0 = plane_ptr[0];
      const float X0 = d0 * r0, X1 = d0 * r1, X2 = d0 * r2;
      const float cden = fmaxf(sqrtf((X0 * X0 + X1 * X1) + X2 * X2), 1e-12f);
      cr0 = X0 / cden; cr1 = X1 / cden; cr2 = X2 / cden;
      const float n1 = fmaxf(sqrtf((cr0 * cr0 + cr1 * cr1) + cr2 * cr2), 1e-5f);
      crn0 = cr0 / n1; crn1 = cr1 / n1; crn2 = cr2 / n1;  // cosine_similarity's first operand
    }

    // ---- depth-invariant part of layer 1 (weights straight from L2, once per unit) ----
    f32x16 hc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) hc[g][m][r] = 0.0f;
    {
      const float4* wi = gW1inv + lane;
#pragma unroll
      for (int t = 0; t < 8; ++t) sr_l1_step(hc, wi[t * 64], cur[2 * t], cur[2 * t + 1]);
      sr_l1_step(hc, wi[8 * 64], cr0, cr1);
      sr_l1_step(hc, wi[9 * 64], cr2, 1.0f);
      const int npose = 3 * p.K;
#pragma unroll 1
      for (int u = 0; u < npose; u += 2) {  // pose measures: [b,k] scalars (geometry_utils.py:178-191)
        const int ka = u / 3, ia = u - 3 * ka, kb = (u + 1) / 3, ib = (u + 1) - 3 * kb;
        const float fa = geom_b[ka * SR_GEOM_STRIDE + 15 + ia];
        const float fb = (u + 1 < npose) ? geom_b[kb * SR_GEOM_STRIDE + 15 + ib] : 0.0f;
        sr_l1_step(hc, wi[(size_t)(10 + (u >> 1)) * 64], fa, fb);
      }
    }

    SrSample smp;
    float4 taps[16];
    float f[SR_VIEW_SLOTS], fn[SR_VIEW_SLOTS];
    float X0, X1, X2, d;
    bool any_depth, any_bounds;
    auto issue_view = [&](int k) {
      sr_project_sample(geom_b + k * SR_GEOM_STRIDE, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, smp);
      const float* img = src_b + (size_t)k * N * C;
      const float4* t_nw = reinterpret_cast<const float4*>(img + (size_t)smp.o_nw * C);
      const float4* t_ne = reinterpret_cast<const float4*>(img + (size_t)smp.o_ne * C);
      const float4* t_sw = reinterpret_cast<const float4*>(img + (size_t)smp.o_sw * C);
      const float4* t_se = reinterpret_cast<const float4*>(img + (size_t)smp.o_se * C);
      if (!SR_MLP_DBG(2)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if SR_MLP_NT_TAPS   // streaming policy for the taps: they should not push the W1 / W2 blocks of a streaming variant out of L2
          typedef float nt_f4 __attribute__((ext_vector_type(4)));
          auto ntl = [](const float4* q) {
            const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(q));
            return make_float4(v.x, v.y, v.z, v.w);
          };
          taps[i] = ntl(&t_nw[i]); taps[4 + i] = ntl(&t_ne[i]); taps[8 + i] = ntl(&t_sw[i]); taps[12 + i] = ntl(&t_se[i]);
#else
          taps[i] = t_nw[i]; taps[4 + i] = t_ne[i]; taps[8 + i] = t_sw[i]; taps[12 + i] = t_se[i];
#endif
        }
      }
    };
    // pieces of the per-view feature assembly (o = fn or f); the ray pieces need no taps
    float rv0, rv1, rv2, rsd, rdot;
#define SR_RAY_A(o, g, kview)                                                                 \
    {                                                                                         \
      _Pragma("clang fp contract(off)")                                                       \
      rv0 = X0 - (g)[12]; rv1 = X1 - (g)[13]; rv2 = X2 - (g)[14];                              \
      rsd = fmaxf(sqrtf((rv0 * rv0 + rv1 * rv1) + rv2 * rv2), 1e-12f);                         \
      (o)[23] = ((kview) == 0) ? d : 0.0f; /* plane depth rides in view 0's spare slot */      \
    }
  // (r03) the source ray F.normalize(X - t_k) (cost_volume.py:654-669) and the ray angle F.cosine_similarity(...)
  // (:683-688) divide three components by one norm each: ONE IEEE reciprocal + three multiplies per normalisation
  // instead of three IEEE divisions (10 VALU instructions apiece beside the MFMAs).  x * (1/n) differs from x / n by at
  // most one ulp; these channels only feed the MLP (parity bar 1e-4, tests: 2e-5 vs the oracle's true divisions).
#define SR_RAY_B(o)                                                                           \
    {                                                                                         \
      _Pragma("clang fp contract(off)")                                                       \
      const float rinv = 1.0f / rsd;                                                          \
      (o)[20] = rv0 * rinv; (o)[21] = rv1 * rinv; (o)[22] = rv2 * rinv;                        \
    }
#define SR_RAY_C(o)                                                                           \
    {                                                                                         \
      _Pragma("clang fp contract(off)")                                                       \
      const float n2 = fmaxf(sqrtf(((o)[20] * (o)[20] + (o)[21] * (o)[21]) + (o)[22] * (o)[22]), 1e-5f); \
      (o)[19] = ((crn0 * (o)[20] + crn1 * (o)[21]) + crn2 * (o)[22]) * (1.0f / n2);            \
    }
#define SR_INTERP2(o, i, lo)  /* channels 4i+2lo, 4i+2lo+1 as one packed fp32 pair (v_pk_fma_f32) */       \
    {                                                                                         \
      const float4 a = taps[i], bq = taps[4 + (i)], c4 = taps[8 + (i)], d4 = taps[12 + (i)];   \
      const sr_f2v ta = (lo) == 0 ? sr_f2v{a.x, a.y} : sr_f2v{a.z, a.w};                       \
      const sr_f2v tb = (lo) == 0 ? sr_f2v{bq.x, bq.y} : sr_f2v{bq.z, bq.w};                   \
      const sr_f2v tc = (lo) == 0 ? sr_f2v{c4.x, c4.y} : sr_f2v{c4.z, c4.w};                   \
      const sr_f2v td = (lo) == 0 ? sr_f2v{d4.x, d4.y} : sr_f2v{d4.z, d4.w};                   \
      const sr_f2v r2 = __builtin_elementwise_fma(sr_f2v{smp.w_se, smp.w_se}, td,              \
                        __builtin_elementwise_fma(sr_f2v{smp.w_sw, smp.w_sw}, tc,              \
                        __builtin_elementwise_fma(sr_f2v{smp.w_ne, smp.w_ne}, tb,              \
                                                  sr_f2v{smp.w_nw, smp.w_nw} * ta)));          \
      (o)[4 * (i) + 2 * (lo) + 0] = r2.x;                                                     \
      (o)[4 * (i) + 2 * (lo) + 1] = r2.y;                                                     \
    }
#define SR_DOT(o)                                                                             \
    {                                                                                         \
      rdot = 0.0f;                                                                            \
      _Pragma("unroll") for (int c = 0; c < 16; ++c) rdot = fmaf((o)[c], cur[c], rdot);        \
      const bool front = smp.zp > 0.0f;                                                       \
      any_depth |= front;                                                                     \
      any_bounds |= sr_in_bounds(smp, p.h, p.w);                                              \
      (o)[16] = front ? 1.0f : 0.0f; /* mask (cost_volume.py:611-612) */                      \
      (o)[17] = smp.zp;              /* z'_k (cost_volume.py:603-609) */                      \
      (o)[18] = front ? rdot : 0.0f; /* dot * mask (cost_volume.py:691-695) */                \
    }
#define SR_SB __builtin_amdgcn_sched_barrier(0);

    // costs of this unit's planes, kept until the 16-byte stores at its end: in registers (select chain), or -- in the
    // variants that stream W1 and have neither registers nor a full LDS -- in a wave-private LDS strip
    constexpr bool CST_LDS = !W1_LDS;
    const int cst_base = SR_LDS_W3_FLOATS + W2_FLOATS +
                         __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (SR_PLANE_CHUNK * 64);   // scalar
    float cst[SR_PLANE_CHUNK];
#pragma unroll
    for (int q = 0; q < SR_PLANE_CHUNK; ++q) cst[q] = 0.0f;
#pragma unroll 1
    for (int j = j0; j < j1; ++j) {
      d = plane_ptr[j * p.planes.sd];
      {
#pragma clang fp contract(off)
        X0 = d * r0; X1 = d * r1; X2 = d * r2;  // geometry_utils.py:56-57
      }
      any_depth = false; any_bounds = false;
      f32x16 acc[2][4];

      {  // view 0, not overlapped
        issue_view(0);
        SR_RAY_A(f, geom_b, 0) SR_RAY_B(f) SR_RAY_C(f)
        SR_INTERP2(f, 0, 0) SR_INTERP2(f, 0, 1) SR_INTERP2(f, 1, 0) SR_INTERP2(f, 1, 1)
        SR_INTERP2(f, 2, 0) SR_INTERP2(f, 2, 1) SR_INTERP2(f, 3, 0) SR_INTERP2(f, 3, 1)
        SR_DOT(f)
      }
      // ---- layer 1, software-pipelined over views: the feature vector of view k+1 is assembled in 12 small
      // VALU pieces, each placed in the shadow of one k-step (8 MFMAs = 512 cycles) of view k; its 16 tap
      // loads are issued before step 0 and first touched at step 3.  sched_barrier(0) pins the placement.
      // view 0's first step takes the hoisted invariant part as its C operand; the loop below handles the rest
#define SR_STEP(t)                                           \
        wN = wk[((t) + 1 < VS ? (t) + 1 : (t)) * 64];       \
        sr_l1_step(acc, wA, f[2 * (t)], f[2 * (t) + 1]);     \
        wA = wN;
#define SR_VIEW_BODY(FIRST)                                                            \
        {                                                                              \
          const int kn = min(k + 1, p.K - 1); /* last iteration re-derives view K-1 */ \
          const float* g = geom_b + kn * SR_GEOM_STRIDE;                               \
          issue_view(kn);                                                              \
          const float4* wk = W1p + (size_t)(VS * k) * 64 + lane;                       \
          float4 wA = wk[0], wN;                                                       \
          SR_SB                                                                        \
          if (FIRST) { wN = wk[64]; sr_l1_step_init(acc, hc, wA, f[0], f[1]); wA = wN; } \
          else { SR_STEP(0) }                                                          \
          SR_RAY_A(fn, g, kn) SR_SB                                                    \
          SR_STEP(1) SR_RAY_B(fn) SR_SB                                                \
          SR_STEP(2) SR_RAY_C(fn) SR_SB                                                \
          SR_STEP(3) SR_SB /* taps not touched before step 7 (>= 3.5k cycles after issue) */ \
          SR_STEP(4) SR_SB                                                             \
          SR_STEP(5) SR_SB                                                             \
          SR_STEP(6) SR_SB                                                             \
          SR_STEP(7) SR_INTERP2(fn, 0, 0) SR_INTERP2(fn, 0, 1) SR_SB                   \
          SR_STEP(8) SR_INTERP2(fn, 1, 0) SR_INTERP2(fn, 1, 1) SR_SB                   \
          SR_STEP(9) SR_INTERP2(fn, 2, 0) SR_INTERP2(fn, 2, 1) SR_SB                   \
          SR_STEP(10) SR_INTERP2(fn, 3, 0) SR_SB                                       \
          SR_STEP(11) SR_INTERP2(fn, 3, 1) SR_DOT(fn) SR_SB                            \
          _Pragma("unroll") for (int t = 0; t < SR_VIEW_SLOTS; ++t) f[t] = fn[t];      \
        }
      {
        const int k = 0;
        SR_VIEW_BODY(true)
      }
#pragma unroll 1
      for (int k = 1; k < p.K; ++k) SR_VIEW_BODY(false)

      // layer 2: the layer-1 accumulators, passed through LeakyReLU(slope) = max(v, slope*v) (networks.py:139,
      // 0 < slope < 1) on the fly, are the B operands; W2 streams from L2 three steps ahead
      // (not zeroed: the first k-step takes the constant 0 as its C operand -- 128 v_mov per plane and wave otherwise)
      f32x16 acc2[2][4];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float4* w2 = (W2_LDS ? reinterpret_cast<const float4*>(lds + SR_LDS_W3_FLOATS) : gW2) + lane;
      if (!W2_LDS) asm volatile("" : "+v"(w2));  // keep hipcc from hoisting 65 loop-invariant 64-bit addresses (spills)
      float4 wn0 = w2[0], wn1 = w2[64], wn2 = w2[128];
#pragma unroll
      for (int t = 0; t < 64; ++t) {
        const float4 wA = wn0;
        wn0 = wn1;
        wn1 = wn2;
        if (!SR_MLP_DBG(1)) wn2 = w2[(size_t)min(t + 3, 64) * 64];
        const float aP = acc[0][t >> 4][t & 15], aQ = acc[1][t >> 4][t & 15];
        const float bP = sr_vmax_mfma(aP, p.slope * aP), bQ = sr_vmax_mfma(aQ, p.slope * aQ);
        acc2[0][0] = SR_MFMA(wA.x, bP, t == 0 ? zero16 : acc2[0][0]);
        acc2[1][0] = SR_MFMA(wA.x, bQ, t == 0 ? zero16 : acc2[1][0]);
        acc2[0][1] = SR_MFMA(wA.y, bP, t == 0 ? zero16 : acc2[0][1]);
        acc2[1][1] = SR_MFMA(wA.y, bQ, t == 0 ? zero16 : acc2[1][1]);
        acc2[0][2] = SR_MFMA(wA.z, bP, t == 0 ? zero16 : acc2[0][2]);
        acc2[1][2] = SR_MFMA(wA.z, bQ, t == 0 ? zero16 : acc2[1][2]);
        acc2[0][3] = SR_MFMA(wA.w, bP, t == 0 ? zero16 : acc2[0][3]);
        acc2[1][3] = SR_MFMA(wA.w, bQ, t == 0 ? zero16 : acc2[1][3]);
      }
      {
        const float4 wA = wn0;  // bias step (t = 64)
        const float one = half ? 0.0f : 1.0f;
        acc2[0][0] = SR_MFMA(wA.x, one, acc2[0][0]);
        acc2[1][0] = SR_MFMA(wA.x, one, acc2[1][0]);
        acc2[0][1] = SR_MFMA(wA.y, one, acc2[0][1]);
        acc2[1][1] = SR_MFMA(wA.y, one, acc2[1][1]);
        acc2[0][2] = SR_MFMA(wA.z, one, acc2[0][2]);
        acc2[1][2] = SR_MFMA(wA.z, one, acc2[1][2]);
        acc2[0][3] = SR_MFMA(wA.w, one, acc2[0][3]);
        acc2[1][3] = SR_MFMA(wA.w, one, acc2[1][3]);
      }

      // layer 3 (128 -> 1, no activation: disable_final_activation=True, cost_volume.py:438); w3tab in LDS
      const float4* w3 = reinterpret_cast<const float4*>(lds + half * 64);
      sr_f2v oPQ = {0.0f, 0.0f};  // the two point groups as one packed fp32 pair (v_pk_mul_f32 / v_pk_fma_f32)
      const sr_f2v slope2 = {p.slope, p.slope};
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 wv = w3[m * 4 + q];
          const float wr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const sr_f2v h2 = {acc2[0][m][4 * q + i], acc2[1][m][4 * q + i]};
            const sr_f2v s2 = slope2 * h2;
            oPQ = __builtin_elementwise_fma(sr_f2v{wr[i], wr[i]}, sr_f2v{sr_vmax(h2.x, s2.x), sr_vmax(h2.y, s2.y)}, oPQ);
          }
        }
      float oP = oPQ.x, oQ = oPQ.y;
      oP += __shfl_xor(oP, 32);
      oQ += __shfl_xor(oQ, 32);
      const float cost = (half ? oQ : oP) + lds[128];

      if (p.vec_store) {
        if (CST_LDS) lds[cst_base + (j - j0) * 64 + lane] = cost;
        else {
#pragma unroll
          for (int q = 0; q < SR_PLANE_CHUNK; ++q) cst[q] = (j - j0 == q) ? cost : cst[q];
        }
      }
      if (active) {
        if (!p.vec_store) p.out.cv[b * p.out.sb + j * p.out.sd + (int64_t)pix * p.out.sp] = cost;
        if (j == p.D - 1 && p.out.mask) p.out.mask[(size_t)b * N + pix] = (uint8_t)(any_depth && any_bounds);
      }
    }
    if (p.vec_store && active) {   // planes j0 .. j1-1 of this pixel are consecutive floats (host checked the alignment)
      float* row = p.out.cv + b * p.out.sb + (int64_t)pix * p.out.sp + j0;
      if (CST_LDS) {
#pragma unroll
        for (int q = 0; q < SR_PLANE_CHUNK; ++q) cst[q] = lds[cst_base + q * 64 + lane];   // this lane's own writes: no barrier
      }
#pragma unroll
      for (int q = 0; q < SR_PLANE_CHUNK; q += 4) {
        if (j0 + q + 4 <= j1) *reinterpret_cast<float4*>(row + q) = make_float4(cst[q], cst[q + 1], cst[q + 2], cst[q + 3]);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (j0 + q + i < j1) row[q + i] = cst[q + i];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ split-precision variant (fenced experiment) ----
// SR_MLP_SPLIT=bf16|f16 (read per call; default off): layers 1 (depth-variant part) and 2 run on the 16-bit matrix pipe
// (v_mfma_f32_32x32x16_{bf16,f16}: 16x the fp32 rate) with every fp32 operand split into two 16-bit pieces,
// x = x_hi + x_lo (round-to-nearest each), and three products per k-step, x_hi w_hi + x_hi w_lo + x_lo w_hi, accumulated in
// fp32 by the MFMA (the dropped x_lo w_lo term is 2^-18 (bf16) / 2^-24 (f16) of the product).  Inputs, outputs, the
// depth-invariant part of layer 1, layer 3 and every accumulator stay fp32.  Same wave-owns-64-points structure as the
// fp32 kernel; a k-step now covers 16 inputs (8 per half-wave), so a view is two steps: its 16 warped channels, then its 8
// metadata slots (upper k-half zero).  The layer-1 accumulators still ARE the layer-2 B operands (after LeakyReLU and
// the split).  Not the default and not what bench.py's headline measures.
typedef unsigned sr_u4v __attribute__((ext_vector_type(4)));

template <int FMT> struct SrSplitFmt;
template <> struct SrSplitFmt<1> {   // two bf16 pieces: 16-17 significant bits, fp32 exponent range
  typedef __bf16 e2 __attribute__((ext_vector_type(2)));
  typedef __bf16 e8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void split(float a, float b, unsigned& hi, unsigned& lo) {
    const sr_f2v v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, e2));
    const sr_f2v v0 = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - v0, e2));
  }
  static __device__ __forceinline__ f32x16 mfma(sr_u4v a, sr_u4v b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(e8, a), __builtin_bit_cast(e8, b), c, 0, 0, 0);
  }
};
template <> struct SrSplitFmt<2> {   // two fp16 pieces: 22-24 significant bits for 2^-14 <= |x| < 65504 (denormal pieces are
  typedef _Float16 e2 __attribute__((ext_vector_type(2)));   // honoured by v_cvt_pk_f16_f32 and the MFMA: measured on gfx950);
  typedef _Float16 e8 __attribute__((ext_vector_type(8)));   // beyond +-65504 the high piece is inf and the result NaN -- loud,
  static __device__ __forceinline__ void split(float a, float b, unsigned& hi, unsigned& lo) {   // not a saturated wrong value
    const sr_f2v v = {a, b};
    const e2 h = __builtin_convertvector(v, e2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - __builtin_convertvector(h, sr_f2v), e2));
  }
  static __device__ __forceinline__ f32x16 mfma(sr_u4v a, sr_u4v b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(e8, a), __builtin_bit_cast(e8, b), c, 0, 0, 0);
  }
};

// LDS / packed layout of the split variant (32-bit words):
//   table  [0,512):   w3tab (128) | b3 (1) | pad | b2tab at 256 (128: C layout, [half][mt][r])
//   W2     8 steps x [mt 4][piece 2][lane 64][4 words]                        = 16384 words
//   W1var  per view: step A [mt 4][piece 2][lane 64][4] (2048) + metadata step [mt 4][piece 2][lane 32][4] (1024)
#ifndef SR_SPL_ABL
#define SR_SPL_ABL 0   // phase ablation (tuning builds only; wrong results): 1 no tap loads, 2 no layer-1 MFMAs, 4 no layer-2 MFMAs
#endif
#define SR_SPL_TAB 512
#define SR_SPL_W2_WORDS 16384
#define SR_SPL_VIEW_WORDS 3072

template <int FMT>
__global__ void sr_mlp_pack_split_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                         const float* __restrict__ W2, const float* __restrict__ b2,
                                         const float* __restrict__ W3, const float* __restrict__ b3,
                                         float* __restrict__ packed, int K, int C) {
  typedef SrSplitFmt<FMT> S;
  const int steps_var = (SR_VIEW_SLOTS / 2) * K;
  const int steps_inv = (SR_INV_FIXED + 3 * K + 1) / 2;
  const int steps1 = steps_var + steps_inv;
  const int Cin = C * (K + 1) + 10 * K + 4;
  const int o_cur = K * C, o_mask = o_cur + C, o_z = o_mask + K, o_d = o_z + K, o_dot = o_d + 1;
  const int o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3, o_pd = o_sray + 3 * K;
  const int o_rm = o_pd + K, o_tm = o_rm + K;
  unsigned* pk = reinterpret_cast<unsigned*>(packed);
  auto var_col = [&](int k, int s) {   // column of W1 for slot s of view k (-1: none)
    if (s < 16) return k * C + s;
    if (s == 16) return o_mask + k;
    if (s == 17) return o_z + k;
    if (s == 18) return o_dot + k;
    if (s == 19) return o_ang + k;
    if (s < 23) return o_sray + 3 * k + (s - 20);
    return (k == 0) ? o_d : -1;
  };
  auto put = [&](size_t word, int piece, float wa, float wb) {
    unsigned hi, lo;
    S::split(wa, wb, hi, lo);
    pk[word] = piece ? lo : hi;
  };
  const int total = (steps1 + SR_MLP_STEPS2) * 256 + 128 + 1;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    if (e < steps_var * 256) {
      const int k = e / SR_SPL_VIEW_WORDS, q = e - k * SR_SPL_VIEW_WORDS;
      int mt, piece, lane, dw, slot0;
      if (q < 2048) { mt = q >> 9; piece = (q >> 8) & 1; lane = (q >> 2) & 63; dw = q & 3; slot0 = 8 * (lane >> 5) + 2 * dw; }
      else { const int q2 = q - 2048; mt = q2 >> 8; piece = (q2 >> 7) & 1; lane = (q2 >> 2) & 31; dw = q2 & 3; slot0 = 16 + 2 * dw; }
      const int row = 32 * mt + (lane & 31);
      const int ca = var_col(k, slot0), cb = var_col(k, slot0 + 1);
      put(e, piece, ca >= 0 ? W1[(size_t)row * Cin + ca] : 0.0f, cb >= 0 ? W1[(size_t)row * Cin + cb] : 0.0f);
    } else if (e < steps1 * 256) {   // depth-invariant part: fp32, the layout of sr_mlp_pack_kernel
      const int e1 = e - steps_var * 256;
      const int t = e1 >> 8, lane = (e1 >> 2) & 63, mt = e1 & 3;
      const int row = 32 * mt + (lane & 31);
      const int u = 2 * t + (lane >> 5);
      int col = -1;
      if (u < 16) col = o_cur + u;
      else if (u < 19) col = o_cray + (u - 16);
      else if (u == 19) col = -2;
      else if (u < SR_INV_FIXED + 3 * K) {
        const int k = (u - SR_INV_FIXED) / 3, i = (u - SR_INV_FIXED) - 3 * k;
        col = (i == 0 ? o_pd : (i == 1 ? o_rm : o_tm)) + k;
      }
      packed[e] = (col >= 0) ? W1[(size_t)row * Cin + col] : (col == -2 ? b1[row] : 0.0f);
    } else if (e < steps1 * 256 + SR_SPL_W2_WORDS) {
      const int q = e - steps1 * 256;
      const int s = q >> 11, mt = (q >> 9) & 3, piece = (q >> 8) & 1, lane = (q >> 2) & 63, dw = q & 3;
      const int row = 32 * mt + (lane & 31);
      // k-slot (s, lane >> 5, i): the hidden feature held in accumulator register r = 8 (s & 1) + i of tile s >> 1
      auto feat = [&](int i) { const int r = 8 * (s & 1) + i; return 32 * (s >> 1) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); };
      put(e, piece, W2[(size_t)row * SR_HID + feat(2 * dw)], W2[(size_t)row * SR_HID + feat(2 * dw + 1)]);
    } else if (e < (steps1 + SR_MLP_STEPS2) * 256) {
      const int i = e - steps1 * 256 - SR_SPL_W2_WORDS;   // 0..255: b2tab in the first 128
      const int half = i >> 6, mt = (i >> 4) & 3, r = i & 15;
      packed[e] = (i < 128) ? b2[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.0f;
    } else if (e < (steps1 + SR_MLP_STEPS2) * 256 + 128) {
      const int i = e - (steps1 + SR_MLP_STEPS2) * 256;
      const int half = i >> 6, mt = (i >> 4) & 3, r = i & 15;
      packed[e] = W3[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * half];
    } else {
      packed[e] = b3[0];
    }
  }
}

__device__ __forceinline__ void sr_swap_halves_u4(const sr_u4v fa, const sr_u4v fb, sr_u4v& bP, sr_u4v& bQ) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    auto r = __builtin_amdgcn_permlane32_swap(fa[i], fb[i], false, false);
    bP[i] = r[0];
    bQ[i] = r[1];
  }
}

// one 16-input k-step of layer 1 or 2: 4 output tiles x 2 point groups x 3 products
// INIT: 0 accumulate, 1 the first products take c0 as their C operand, 2 they take zero
template <int FMT, int INIT>
__device__ __forceinline__ void sr_spl_step(f32x16 (&acc)[2][4], const f32x16 (&c0)[2][4], const sr_u4v* aw, int ls,
                                            const sr_u4v bPh, const sr_u4v bPl, const sr_u4v bQh, const sr_u4v bQl) {
  typedef SrSplitFmt<FMT> S;
  sr_u4v ah[4], al[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) { ah[m] = aw[(2 * m) * ls]; al[m] = aw[(2 * m + 1) * ls]; }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0][m] = S::mfma(ah[m], bPh, INIT == 1 ? c0[0][m] : (INIT == 2 ? zero16 : acc[0][m]));
    acc[1][m] = S::mfma(ah[m], bQh, INIT == 1 ? c0[1][m] : (INIT == 2 ? zero16 : acc[1][m]));
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    acc[0][m] = S::mfma(ah[m], bPl, acc[0][m]);
    acc[1][m] = S::mfma(ah[m], bQl, acc[1][m]);
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    acc[0][m] = S::mfma(al[m], bPh, acc[0][m]);
    acc[1][m] = S::mfma(al[m], bQh, acc[1][m]);
  }
}

template <int FMT>
__global__ __launch_bounds__(256, 1) void sr_mlp_volume_split_kernel(SrMlpParams p) {
  typedef SrSplitFmt<FMT> S;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int C = 16;
  const int lane = threadIdx.x & 63;
  const int steps_var = (SR_VIEW_SLOTS / 2) * p.K;
  const int steps1 = steps_var + (SR_INV_FIXED + 3 * p.K + 1) / 2;
  const float4* gW1inv = reinterpret_cast<const float4*>(p.packed) + (size_t)steps_var * 64;
  const float* gW2 = p.packed + (size_t)steps1 * 256;
  const float* gW3 = p.packed + (size_t)(steps1 + SR_MLP_STEPS2) * 256;

  for (int i = threadIdx.x; i < 129; i += blockDim.x) lds[i] = gW3[i];
  for (int i = threadIdx.x; i < 128; i += blockDim.x) lds[256 + i] = gW2[SR_SPL_W2_WORDS + i];
  {
    float4* l4 = reinterpret_cast<float4*>(lds + SR_SPL_TAB);
    const float4* g4 = reinterpret_cast<const float4*>(gW2);
    for (int i = threadIdx.x; i < SR_SPL_W2_WORDS / 4; i += blockDim.x) l4[i] = g4[i];
    l4 += SR_SPL_W2_WORDS / 4;
    g4 = reinterpret_cast<const float4*>(p.packed);
    for (int i = threadIdx.x; i < p.K * (SR_SPL_VIEW_WORDS / 4); i += blockDim.x) l4[i] = g4[i];
  }
  __syncthreads();
  const sr_u4v* W2s = reinterpret_cast<const sr_u4v*>(lds + SR_SPL_TAB);
  const sr_u4v* W1s = W2s + SR_SPL_W2_WORDS / 4;

  const int N = p.h * p.w;
  const long nunits = (long)p.B * p.tiles * p.chunks;
  long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  long unit_end = nunits;
  if (p.xcd_order && (gridDim.x & 7) == 0) {   // XCD-contiguous unit ranges, as in sr_mlp_volume_kernel
    const long u8 = (nunits + 7) / 8;
    const int xcd = blockIdx.x & 7;
    nwaves = (long)(gridDim.x >> 3) * (blockDim.x >> 6);
    wave0 = xcd * u8 + (long)(blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unit_end = min(nunits, (xcd + 1) * u8);
  }
  const int half = lane >> 5;
  const sr_u4v zero4 = {0u, 0u, 0u, 0u};

  for (long unit = wave0; unit < unit_end; unit += nwaves) {
    const int chunk = (int)(unit % p.chunks);
    const long tb = unit / p.chunks;
    const int tile = (int)(tb % p.tiles);
    const int b = (int)(tb / p.tiles);
    const int j0 = chunk * p.chunk, j1 = min(p.D, j0 + p.chunk);
    const int pix = tile * 64 + lane;
    const bool active = pix < N;
    const int pc = active ? pix : N - 1;
    const int y = pc / p.w, x = pc - y * p.w;
    const float* plane_ptr = p.planes.ptr + b * p.planes.sb + y * p.planes.sy + x * p.planes.sx;
    const float* geom_b = p.geom + (size_t)b * p.K * SR_GEOM_STRIDE;
    const float* src_b = p.src_nhwc + (size_t)b * p.K * N * C;

    float cur[C];
#pragma unroll
    for (int c = 0; c < C; ++c) cur[c] = p.cur[((size_t)b * C + c) * N + pc];

    float r0, r1, r2;
    {
#pragma clang fp contract(off)
      const float* iK = p.invK + 16 * (size_t)b;
      const float px = (float)x + 0.5f, py = (float)y + 0.5f;
      r0 = iK[0] * px + iK[1] * py + iK[2];
      r1 = iK[4] * px + iK[5] * py + iK[6];
      r2 = iK[8] * px + iK[9] * py + iK[10];
    }
    float cr0, cr1, cr2, crn0, crn1, crn2;
    {
#pragma clang fp contract(off)
      const float d0 = plane_ptr[0];
      const float X0 = d0 * r0, X1 = d0 * r1, X2 = d0 * r2;
      const float cden = fmaxf(sqrtf((X0 * X0 + X1 * X1) + X2 * X2), 1e-12f);
      cr0 = X0 / cden; cr1 = X1 / cden; cr2 = X2 / cden;
      const float n1 = fmaxf(sqrtf((cr0 * cr0 + cr1 * cr1) + cr2 * cr2), 1e-5f);
      crn0 = cr0 / n1; crn1 = cr1 / n1; crn2 = cr2 / n1;
    }

    // depth-invariant part of layer 1: fp32 MFMA, once per unit (as in sr_mlp_volume_kernel)
    f32x16 hc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) hc[g][m][r] = 0.0f;
    {
      const float4* wi = gW1inv + lane;
#pragma unroll
      for (int t = 0; t < 8; ++t) sr_l1_step(hc, wi[t * 64], cur[2 * t], cur[2 * t + 1]);
      sr_l1_step(hc, wi[8 * 64], cr0, cr1);
      sr_l1_step(hc, wi[9 * 64], cr2, 1.0f);
      const int npose = 3 * p.K;
#pragma unroll 1
      for (int u = 0; u < npose; u += 2) {
        const int ka = u / 3, ia = u - 3 * ka, kb = (u + 1) / 3, ib = (u + 1) - 3 * kb;
        const float fa = geom_b[ka * SR_GEOM_STRIDE + 15 + ia];
        const float fb = (u + 1 < npose) ? geom_b[kb * SR_GEOM_STRIDE + 15 + ib] : 0.0f;
        sr_l1_step(hc, wi[(size_t)(10 + (u >> 1)) * 64], fa, fb);
      }
    }

    SrSample smp;
    float4 taps[16];
    float fn[SR_VIEW_SLOTS];
    unsigned ph[SR_VIEW_SLOTS / 2], pl[SR_VIEW_SLOTS / 2];   // the assembled view as 12 pairs of 16-bit pieces
    float X0, X1, X2, d;
    bool any_depth, any_bounds;
    auto issue_view = [&](int k) {
      sr_project_sample(geom_b + k * SR_GEOM_STRIDE, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, smp);
      const float* img = src_b + (size_t)k * N * C;
      const float4* t_nw = reinterpret_cast<const float4*>(img + (size_t)smp.o_nw * C);
      const float4* t_ne = reinterpret_cast<const float4*>(img + (size_t)smp.o_ne * C);
      const float4* t_sw = reinterpret_cast<const float4*>(img + (size_t)smp.o_sw * C);
      const float4* t_se = reinterpret_cast<const float4*>(img + (size_t)smp.o_se * C);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (SR_SPL_ABL & 1) { taps[i] = taps[4 + i] = taps[8 + i] = taps[12 + i] = make_float4(smp.w_nw, smp.w_ne, smp.w_sw, smp.w_se); }
        else { taps[i] = t_nw[i]; taps[4 + i] = t_ne[i]; taps[8 + i] = t_sw[i]; taps[12 + i] = t_se[i]; }
      }
    };
    float rv0, rv1, rv2, rsd, rdot;
#define SR_SPLIT_VIEW                                                                     \
    _Pragma("unroll") for (int i = 0; i < SR_VIEW_SLOTS / 2; ++i) S::split(fn[2 * i], fn[2 * i + 1], ph[i], pl[i]);

    float cst[SR_PLANE_CHUNK];
#pragma unroll
    for (int q = 0; q < SR_PLANE_CHUNK; ++q) cst[q] = 0.0f;
#pragma unroll 1
    for (int j = j0; j < j1; ++j) {
      d = plane_ptr[j * p.planes.sd];
      {
#pragma clang fp contract(off)
        X0 = d * r0; X1 = d * r1; X2 = d * r2;
      }
      any_depth = false; any_bounds = false;
      f32x16 acc[2][4];

      {  // view 0, not overlapped
        issue_view(0);
        SR_RAY_A(fn, geom_b, 0) SR_RAY_B(fn) SR_RAY_C(fn)
        SR_INTERP2(fn, 0, 0) SR_INTERP2(fn, 0, 1) SR_INTERP2(fn, 1, 0) SR_INTERP2(fn, 1, 1)
        SR_INTERP2(fn, 2, 0) SR_INTERP2(fn, 2, 1) SR_INTERP2(fn, 3, 0) SR_INTERP2(fn, 3, 1)
        SR_DOT(fn)
        SR_SPLIT_VIEW
      }
      // layer 1 (depth-variant part), software-pipelined over views: view k+1 is assembled (fp32) and split beside the
      // 2 x 24 MFMAs of view k
#define SR_SPL_VIEW_BODY(FIRST)                                                                          \
      {                                                                                                  \
        const int kn = min(k + 1, p.K - 1);                                                              \
        const float* g = geom_b + kn * SR_GEOM_STRIDE;                                                   \
        const sr_u4v* wk = W1s + (size_t)k * (SR_SPL_VIEW_WORDS / 4);                                    \
        sr_u4v bPh, bPl, bQh, bQl;                                                                       \
        sr_swap_halves_u4(sr_u4v{ph[0], ph[1], ph[2], ph[3]}, sr_u4v{ph[4], ph[5], ph[6], ph[7]}, bPh, bQh); \
        sr_swap_halves_u4(sr_u4v{pl[0], pl[1], pl[2], pl[3]}, sr_u4v{pl[4], pl[5], pl[6], pl[7]}, bPl, bQl); \
        sr_u4v mPh, mPl, mQh, mQl;                                                                       \
        sr_swap_halves_u4(sr_u4v{ph[8], ph[9], ph[10], ph[11]}, zero4, mPh, mQh);                        \
        sr_swap_halves_u4(sr_u4v{pl[8], pl[9], pl[10], pl[11]}, zero4, mPl, mQl);                        \
        issue_view(kn);                                                                                  \
        if (!(SR_SPL_ABL & 2) || (FIRST)) sr_spl_step<FMT, (FIRST) ? 1 : 0>(acc, hc, wk + lane, 64, bPh, bPl, bQh, bQl); \
        SR_RAY_A(fn, g, kn) SR_RAY_B(fn) SR_RAY_C(fn)                                                    \
        SR_SB                                                                                            \
        if (!(SR_SPL_ABL & 2)) sr_spl_step<FMT, 0>(acc, hc, wk + 512 + (lane & 31), 32, mPh, mPl, mQh, mQl); \
        SR_INTERP2(fn, 0, 0) SR_INTERP2(fn, 0, 1) SR_INTERP2(fn, 1, 0) SR_INTERP2(fn, 1, 1)              \
        SR_INTERP2(fn, 2, 0) SR_INTERP2(fn, 2, 1) SR_INTERP2(fn, 3, 0) SR_INTERP2(fn, 3, 1)              \
        SR_DOT(fn)                                                                                       \
        SR_SPLIT_VIEW                                                                                    \
        SR_SB                                                                                            \
      }
      {
        const int k = 0;
        SR_SPL_VIEW_BODY(true)
      }
#pragma unroll 1
      for (int k = 1; k < p.K; ++k) SR_SPL_VIEW_BODY(false)

      // layer 2: 8 k-steps; step s takes accumulator registers 8 (s & 1) .. + 7 of tile s >> 1 through LeakyReLU and the split
      f32x16 acc2[2][4];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        sr_u4v bh[2], bl[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float a0 = acc[g][s >> 1][8 * (s & 1) + 2 * i], a1 = acc[g][s >> 1][8 * (s & 1) + 2 * i + 1];
            unsigned hi, lo;
            S::split(sr_vmax(a0, p.slope * a0), sr_vmax(a1, p.slope * a1), hi, lo);
            bh[g][i] = hi;
            bl[g][i] = lo;
          }
        const sr_u4v* w2 = W2s + (size_t)s * 512 + lane;
        if (s == 0) sr_spl_step<FMT, 2>(acc2, acc2, w2, 64, bh[0], bl[0], bh[1], bl[1]);
        else if (!(SR_SPL_ABL & 4)) sr_spl_step<FMT, 0>(acc2, acc2, w2, 64, bh[0], bl[0], bh[1], bl[1]);
      }

      // layer 3 (128 -> 1) on acc2 + b2; w3tab and b2tab in LDS (C layout)
      const float4* w3 = reinterpret_cast<const float4*>(lds + half * 64);
      const float4* b2t = reinterpret_cast<const float4*>(lds + 256 + half * 64);
      sr_f2v oPQ = {0.0f, 0.0f};
      const sr_f2v slope2 = {p.slope, p.slope};
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 wv = w3[m * 4 + q], bv = b2t[m * 4 + q];
          const float wr[4] = {wv.x, wv.y, wv.z, wv.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const sr_f2v h2 = sr_f2v{acc2[0][m][4 * q + i], acc2[1][m][4 * q + i]} + sr_f2v{br[i], br[i]};
            const sr_f2v s2 = slope2 * h2;
            oPQ = __builtin_elementwise_fma(sr_f2v{wr[i], wr[i]}, sr_f2v{sr_vmax(h2.x, s2.x), sr_vmax(h2.y, s2.y)}, oPQ);
          }
        }
      float oP = oPQ.x, oQ = oPQ.y;
      oP += __shfl_xor(oP, 32);
      oQ += __shfl_xor(oQ, 32);
      const float cost = (half ? oQ : oP) + lds[128];

      if (p.vec_store) {
#pragma unroll
        for (int q = 0; q < SR_PLANE_CHUNK; ++q) cst[q] = (j - j0 == q) ? cost : cst[q];
      }
      if (active) {
        if (!p.vec_store) p.out.cv[b * p.out.sb + j * p.out.sd + (int64_t)pix * p.out.sp] = cost;
        if (j == p.D - 1 && p.out.mask) p.out.mask[(size_t)b * N + pix] = (uint8_t)(any_depth && any_bounds);
      }
    }
    if (p.vec_store && active) {
      float* row = p.out.cv + b * p.out.sb + (int64_t)pix * p.out.sp + j0;
#pragma unroll
      for (int q = 0; q < SR_PLANE_CHUNK; q += 4) {
        if (j0 + q + 4 <= j1) *reinterpret_cast<float4*>(row + q) = make_float4(cst[q], cst[q + 1], cst[q + 2], cst[q + 3]);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (j0 + q + i < j1) row[q + i] = cst[q + i];
        }
      }
    }
  }
}

// option SR_OPT_MLP_SPLIT: 0 = fp32 MFMA (the product path), 1 = bf16 pieces, 2 = f16 pieces, -1 = unknown (refused)
static int sr_mlp_split_mode() { return sr_opt(SR_OPT_MLP_SPLIT); }

// lowest_cost = planes[argmax_d volume] (cost_volume.py:338-342, 374-378); first maximum wins
__global__ void sr_argmax_planes_kernel(const float* __restrict__ cv, int64_t sb, int64_t sd, int64_t sp,
                                        SrPlanes planes, int h, int w, int D, float* __restrict__ lowest) {
  const int N = h * w;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= N) return;
  const int y = pix / w, x = pix - y * w;
  const float* c = cv + b * sb + (int64_t)pix * sp;
  float best = c[0];
  int bj = 0;
  int j = 1;
  for (; j + 8 <= D; j += 8) {  // 8 loads in flight per thread (the launch has few threads: latency-bound otherwise)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = c[(int64_t)(j + u) * sd];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v[u] > best || (v[u] != v[u] && best == best)) { best = v[u]; bj = j + u; }  // NaN = max, like torch.argmax
  }
  for (; j < D; ++j) {
    const float v = c[(int64_t)j * sd];
    if (v > best || (v != v && best == best)) { best = v; bj = j; }
  }
  lowest[(size_t)b * N + pix] = planes.ptr[b * planes.sb + bj * planes.sd + y * planes.sy + x * planes.sx];
}

// ------------------------------------------------------------------ C ABI -------------

extern "C" size_t sr_mlp_volume_workspace_bytes(int B, int K, int C, int h, int w, int hidden) {
  (void)hidden;
  if (B < 0 || K < 0) return 0;
  return sr_volume_workspace_bytes(B, K, C, h, w) + sr_align_up(sr_mlp_packed_floats(K) * sizeof(float), 256) + 256;
}

static float* sr_ws_packed(void* workspace, int B, int K, int C, int h, int w) {
  return (float*)sr_align_up((size_t)workspace + sr_volume_workspace_bytes(B, K, C, h, w), 256);
}

extern "C" int sr_mlp_volume_sweep(const float* cur, const float* invK_cur, const float* planes, int64_t ps_b,
                                   int64_t ps_d, int64_t ps_y, int64_t ps_x, float leaky_slope, int B, int K,
                                   int C, int h, int w, int D, float* out_cv, int64_t cv_sb, int64_t cv_sd,
                                   int64_t cv_sp, float* out_lowest, uint8_t* out_mask, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!cur || !invK_cur || !planes || !out_cv || !workspace) return SR_ERR_INVALID_ARGUMENT;
  if (C != 16) return SR_ERR_UNSUPPORTED;
  if (!(leaky_slope > 0.0f && leaky_slope < 1.0f)) return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_mlp_volume_workspace_bytes(B, K, C, h, w, SR_HID)) return SR_ERR_WORKSPACE_TOO_SMALL;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = h * w;

  SrMlpParams p;
  p.cur = cur; p.src_nhwc = sr_ws_src_nhwc(workspace, B, K); p.invK = invK_cur; p.geom = sr_ws_geom(workspace);
  p.packed = sr_ws_packed(workspace, B, K, C, h, w);
  p.planes = {planes, ps_b, ps_d, ps_y, ps_x};
  p.out = {out_cv, cv_sb, cv_sd, cv_sp, out_lowest, out_mask};
  p.B = B; p.K = K; p.h = h; p.w = w; p.D = D;
  p.tiles = (N + 63) / 64;
  p.inv_w = (float)(1.0 / (double)w);
  p.inv_h = (float)(1.0 / (double)h);
  p.slope = leaky_slope;
#ifdef SR_MLP_ABLATION   // (ablation builds only)
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("SR_MLP_DEBUG"); dbg = e ? atoi(e) : 0; } p.debug = dbg; }
#else
  p.debug = 0;
#endif

  // planes per work unit: as many as possible (the hoisted invariant part is paid once per unit) while the
  // units still spread evenly over the 4*CUs persistent waves
  auto plan = [&](int ncu, int& chunk) {
    const long waves = 4L * ncu;
    double best_cost = 1e30;
    chunk = 1;
    for (int c = SR_PLANE_CHUNK; c >= 1; c >>= 1) {
      const long units = (long)B * p.tiles * ((D + c - 1) / c);
      const long rounds = (units + waves - 1) / waves;
      const double cost = (double)rounds * (c * 1192.0 + 168.0);  // MFMAs per unit: c planes + invariant part
      if (cost < best_cost) { best_cost = cost; chunk = c; }
    }
    return best_cost;
  };
  int cus = sr_device_cus(), best = 1;
  const double full_cost = plan(cus, best);
  // SR_MLP_RESERVE_CUS = n: the persistent grid leaves n CUs to other streams (one workgroup owns a CU's LDS and registers, so
  // nothing else starts on a CU of the sweep); the plane chunk is planned for the CUs that remain.  DepthModel sets it around its
  // own call at small batch, where the image-prior encoder's chain of small launches on the side stream is the critical path
  // and would be parked for the whole sweep (batch 1, graph replay: 5.72 -> 5.48 ms per frame although the sweep itself takes
  // 14 % longer).  At batch 8 it costs what it gains (25.4-25.6 vs 25.2-25.5 ms; 27.2 vs 26.5 on the keyframe stream), at
  // 960x736 / 96 planes it loses 2.4 ms of 46: the caller's decision, by batch size.
  {
    const int r = sr_opt(SR_OPT_MLP_RESERVE_CUS);
    if (r > 0 && cus - r >= 8) { cus -= r; (void)plan(cus, best); }
  }
  (void)full_cost;
  p.chunk = best;
  p.chunks = (D + best - 1) / best;
  p.vec_store = (cv_sd == 1) && (p.chunk % 4 == 0) && (cv_sp % 4 == 0) && (cv_sb % 4 == 0) &&
                (((uintptr_t)out_cv & 15) == 0);
  if (sr_opt(SR_OPT_MLP_VEC_STORE) == 0) p.vec_store = 0;   // ablation
  p.xcd_order = sr_opt(SR_OPT_MLP_XCD);
  const long nunits = (long)B * p.tiles * p.chunks;
  const int blocks = (int)((nunits + 3) / 4 < cus ? (nunits + 3) / 4 : cus);
  const size_t w3_bytes = SR_LDS_W3_FLOATS * sizeof(float);
  const size_t w1_bytes = (size_t)sr_mlp_steps_var(K) * 1024;
  const size_t w2_bytes = (size_t)SR_MLP_STEPS2 * 1024;
  const size_t lds_max = 160 * 1024;
  const int split = sr_mlp_split_mode();
  if (split < 0) return SR_ERR_INVALID_ARGUMENT;
  if (split) {   // fenced experiment: both weight blocks must fit the LDS (K <= 7)
    const size_t bytes = ((size_t)SR_SPL_TAB + SR_SPL_W2_WORDS + (size_t)K * SR_SPL_VIEW_WORDS) * 4;
    if (bytes > lds_max) return SR_ERR_UNSUPPORTED;
    const void* fn = split == 1 ? (const void*)sr_mlp_volume_split_kernel<1> : (const void*)sr_mlp_volume_split_kernel<2>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return sr_hip_rc(e);
    if (split == 1) hipLaunchKernelGGL((sr_mlp_volume_split_kernel<1>), dim3(blocks), dim3(256), bytes, stream, p);
    else hipLaunchKernelGGL((sr_mlp_volume_split_kernel<2>), dim3(blocks), dim3(256), bytes, stream, p);
    int rc = sr_hip_rc(hipGetLastError());
    if (rc) return rc;
    if (out_lowest) rc = sr_launch_argmax_planes(out_cv, cv_sb, cv_sd, cv_sp, p.planes, B, h, w, D, out_lowest, stream);
    return rc;
  }
#define SR_MLP_LAUNCH(L1, L2, BYTES)                                                                              \
  {                                                                                                               \
    hipError_t e = hipFuncSetAttribute((const void*)sr_mlp_volume_kernel<L1, L2>,                                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES));                 \
    if (e != hipSuccess) return sr_hip_rc(e);                                                                     \
    hipLaunchKernelGGL((sr_mlp_volume_kernel<L1, L2>), dim3(blocks), dim3(256), (BYTES), stream, p);              \
  }
  if (w1_bytes + w2_bytes + w3_bytes <= lds_max) SR_MLP_LAUNCH(true, true, w1_bytes + w2_bytes + w3_bytes)
  else if (w1_bytes + w3_bytes <= lds_max) SR_MLP_LAUNCH(true, false, w1_bytes + w3_bytes)
  else SR_MLP_LAUNCH(false, true, w2_bytes + w3_bytes + 4 * SR_PLANE_CHUNK * 64 * sizeof(float))
#undef SR_MLP_LAUNCH
  int rc = sr_hip_rc(hipGetLastError());
  if (rc) return rc;
  if (out_lowest) rc = sr_launch_argmax_planes(out_cv, cv_sb, cv_sd, cv_sp, p.planes, B, h, w, D, out_lowest, stream);
  return rc;
}

int sr_launch_argmax_planes(const float* cv, int64_t sb, int64_t sd, int64_t sp, SrPlanes planes, int B, int h, int w,
                            int D, float* lowest, hipStream_t stream) {
  hipLaunchKernelGGL(sr_argmax_planes_kernel, dim3((h * w + 63) / 64, B), dim3(64), 0, stream, cv, sb, sd, sp,
                     planes, h, w, D, lowest);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_mlp_pack_weights(const float* W1, const float* b1, const float* W2, const float* b2,
                                   const float* W3, const float* b3, int hidden, int B, int K, int C, int h, int w,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
  if (!W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !workspace || K <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (hidden != SR_HID || C != 16) return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_mlp_volume_workspace_bytes(B, K, C, h, w, hidden)) return SR_ERR_WORKSPACE_TOO_SMALL;
  const int split = sr_mlp_split_mode();
  if (split < 0) return SR_ERR_INVALID_ARGUMENT;
  float* packed = sr_ws_packed(workspace, B, K, C, h, w);
  if (split == 1) hipLaunchKernelGGL((sr_mlp_pack_split_kernel<1>), dim3(64), dim3(256), 0, (hipStream_t)stream_, W1, b1, W2, b2, W3, b3, packed, K, C);
  else if (split == 2) hipLaunchKernelGGL((sr_mlp_pack_split_kernel<2>), dim3(64), dim3(256), 0, (hipStream_t)stream_, W1, b1, W2, b2, W3, b3, packed, K, C);
  else hipLaunchKernelGGL(sr_mlp_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream_, W1, b1, W2, b2, W3, b3, packed, K, C);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_mlp_volume_fwd(const float* cur, const float* src, const float* K_src, const float* T_src_cur,
                                 const float* T_cur_src, const float* invK_cur, const float* planes, int64_t ps_b,
                                 int64_t ps_d, int64_t ps_y, int64_t ps_x, const float* W1, const float* b1,
                                 const float* W2, const float* b2, const float* W3, const float* b3, int hidden,
                                 float leaky_slope, int B, int K, int C, int h, int w, int D, float* out_cv,
                                 int64_t cv_sb, int64_t cv_sd, int64_t cv_sp, float* out_lowest, uint8_t* out_mask,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!T_cur_src) return SR_ERR_INVALID_ARGUMENT;
  if (hidden != SR_HID || C != 16) return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_mlp_volume_workspace_bytes(B, K, C, h, w, hidden)) return SR_ERR_WORKSPACE_TOO_SMALL;
  int rc = sr_volume_prepare(src, K_src, T_src_cur, T_cur_src, B, K, C, h, w, workspace, workspace_bytes, stream_);
  if (rc) return rc;
  rc = sr_mlp_pack_weights(W1, b1, W2, b2, W3, b3, hidden, B, K, C, h, w, workspace, workspace_bytes, stream_);
  if (rc) return rc;
  return sr_mlp_volume_sweep(cur, invK_cur, planes, ps_b, ps_d, ps_y, ps_x, leaky_slope, B, K, C, h, w, D, out_cv,
                             cv_sb, cv_sd, cv_sp, out_lowest, out_mask, workspace, workspace_bytes, stream_);
}
