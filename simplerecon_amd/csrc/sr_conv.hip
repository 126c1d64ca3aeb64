// sr_conv.hip -- 2-D convolutions of the cost-volume encoder / UNet++ decoder for gfx950.
//
// Replaces the Conv2d + bias + (residual add) + LeakyReLU(0.2) compositions of the reference's
// BasicBlock (modules/layers.py:24-85) as used by CVEncoder / DepthDecoderPP
// (modules/networks.py:20-127), and the bilinear x2 `upsample` (utils/generic_utils.py:96-105).
//
// Implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
// accumulate -- the reference runs these convs in fp32): M = output pixels, N = output
// channels, K = taps x input channels.  Activations are channels-last in HBM.
//   * A workgroup (4 waves) owns TH x 32 output pixels x (32*NT) output channels.  Per 16-channel
//     slab of the input it stages the (TH*S+KS-1) x (32*S+KS-1) halo tile ONCE into LDS
//     (coalesced 64-byte reads per pixel, zero-filled outside the image = the conv padding) and
//     re-uses it for all KS*KS taps and all N tiles: 9x less L2->LDS traffic than im2col.
//   * LDS rows are padded to 20 floats so that the per-lane 16-byte A-fragment reads
//     (ds_read_b128, 4 k-steps per read) are bank-conflict-free.
//   * Weights are pre-packed in MFMA B-fragment order and stream straight from L2 into VGPRs
//     (one coalesced 1 KiB read per wave feeds 4 k-steps x MT M-tiles); they never occupy LDS.
//   * Epilogue in registers: + bias, + residual (identity or projected skip), LeakyReLU, and the
//     store goes directly into a channel slice of the consumer's concat buffer.
#include "sr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SR_CK 16        // input channels per LDS slab
#define SR_LDS_ROW 20   // floats per staged pixel (16 + 4 pad)
#define SR_TW 32        // output columns per workgroup tile (= one 32-row MFMA M-tile per row)

struct SrConvParams {
  const float* in; int64_t in_sb; int in_sp;        // batch stride, pixel stride (elements)
  const float* wp;                                  // packed weights [taps][G][2][Co_pad][4]
  const float* bias;                                // [Cout] or null
  const float* res; int64_t res_sb; int res_sp;     // residual (output geometry) or null
  float* out; int64_t out_sb; int out_sp;
  int H, W, Cin, Ho, Wo, Cout, Co_pad, G;           // G = 8-channel groups (even, zero padded)
  int tiles_x;
  float slope;                                      // < 0: no activation
  int vec4;                                         // input rows 16-byte aligned
};

template <int KS, int S, int MT, int NT>
__global__ __launch_bounds__(256) void sr_conv_kernel(SrConvParams p) {
  constexpr int TH = 4 * MT;                         // output rows per workgroup (MT per wave)
  constexpr int HH = (TH - 1) * S + KS;              // halo rows
  constexpr int HW = (SR_TW - 1) * S + KS;           // halo cols
  constexpr int PAD = KS / 2;
  __shared__ __attribute__((aligned(16))) float tile[HH * HW * SR_LDS_ROW];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int b = blockIdx.z;
  const int co0 = blockIdx.y * (32 * NT);
  const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x - ty * p.tiles_x;
  const int oy0 = ty * TH, ox0 = tx * SR_TW;
  const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

  const float* in_b = p.in + (int64_t)b * p.in_sb;
  // this lane's A-fragment base inside the halo tile (M-tile m = output row wave*MT + m)
  int a_off[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_off[m] = (((wave * MT + m) * S) * HW + i * S) * SR_LDS_ROW + 4 * kk;
  // this lane's B-fragment base inside the packed weights
  const float4* wp4 = reinterpret_cast<const float4*>(p.wp);
  const int w_lane = kk * p.Co_pad + co0 + i;  // float4 index within one (tap, g) record of 2*Co_pad float4

  const int chunks = p.G >> 1;
  for (int ch = 0; ch < chunks; ++ch) {
    const int c0 = ch * SR_CK;
    __syncthreads();  // previous slab fully consumed
    // ---- stage the halo tile of channels [c0, c0+16) ----
    for (int e = threadIdx.x; e < HH * HW * 4; e += 256) {
      const int px = e >> 2, q = e & 3;
      const int hy = px / HW, hx = px - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const int c = c0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin) {
        const float* src = in_b + ((int64_t)iy * p.W + ix) * p.in_sp + c;
        if (p.vec4 && c + 3 < p.Cin) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (c + 1 < p.Cin) v.y = src[1];
          if (c + 2 < p.Cin) v.z = src[2];
          if (c + 3 < p.Cin) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&tile[px * SR_LDS_ROW + 4 * q]) = v;
    }
    __syncthreads();
    // ---- MFMA over taps x 2 channel groups of this slab ----
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int tap = ky * KS + kx;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float4 a[MT], bw[NT];
#pragma unroll
          for (int m = 0; m < MT; ++m)
            a[m] = *reinterpret_cast<const float4*>(&tile[a_off[m] + (ky * HW + kx) * SR_LDS_ROW + 8 * g]);
          const float4* wrec = wp4 + ((int64_t)(tap * p.G + 2 * ch + g) * 2) * p.Co_pad + w_lane;
#pragma unroll
          for (int n = 0; n < NT; ++n) bw[n] = wrec[32 * n];
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, bw[n].x, acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, bw[n].y, acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, bw[n].z, acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, bw[n].w, acc[m][n], 0, 0, 0);
            }
        }
      }
  }

  // ---- epilogue: C[row = pixel column, col = output channel] ----
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int oy = oy0 + wave * MT + m;
    if (oy >= p.Ho) continue;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = co0 + 32 * n + i;
      if (co >= p.Cout) continue;
      const float bv = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (ox >= p.Wo) continue;
        const int64_t opix = (int64_t)oy * p.Wo + ox;
        float v = acc[m][n][r] + bv;
        if (p.res) v += p.res[(int64_t)b * p.res_sb + opix * p.res_sp + co];
        if (p.slope >= 0.0f) v = v > 0.0f ? v : v * p.slope;
        p.out[(int64_t)b * p.out_sb + opix * p.out_sp + co] = v;
      }
    }
  }
}

// [Co, Ci, KS, KS] -> [taps][G][2][Co_pad][4], zero padded: element (tap, g, kk, co, e) =
// W[co][8g + 4kk + e][tap]; this is the per-lane B fragment of 4 consecutive k-steps.
__global__ void sr_conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int taps,
                                    int G, int Co_pad) {
  const int64_t total = (int64_t)taps * G * 2 * Co_pad * 4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int el = (int)(e & 3);
    int64_t r = e >> 2;
    const int co = (int)(r % Co_pad); r /= Co_pad;
    const int kk = (int)(r & 1); r >>= 1;
    const int g = (int)(r % G);
    const int tap = (int)(r / G);
    const int ci = 8 * g + 4 * kk + el;
    wp[e] = (co < Co && ci < Ci) ? w[((int64_t)co * Ci + ci) * taps + tap] : 0.0f;
  }
}

// bilinear x2, align_corners=False (ATen upsample_bilinear2d semantics), channels-last, 4 channels / thread
__global__ void sr_upsample2x_kernel(const float* __restrict__ in, int64_t in_sb, int in_sp, float* __restrict__ out,
                                     int64_t out_sb, int out_sp, int H, int W, int C, int vec4) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int cq = (C + 3) >> 2;
  const int64_t total = (int64_t)Ho * Wo * cq;
  const int b = blockIdx.y;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(e % cq);
    const int64_t opix = e / cq;
    const int oy = (int)(opix / Wo), ox = (int)(opix - (int64_t)oy * Wo);
    const float sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.0f);
    const float sx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.0f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* ib = in + (int64_t)b * in_sb + 4 * q;
    const float* p00 = ib + ((int64_t)y0 * W + x0) * in_sp;
    const float* p01 = ib + ((int64_t)y0 * W + x1) * in_sp;
    const float* p10 = ib + ((int64_t)y1 * W + x0) * in_sp;
    const float* p11 = ib + ((int64_t)y1 * W + x1) * in_sp;
    float* o = out + (int64_t)b * out_sb + opix * out_sp + 4 * q;
    const int nc = min(4, C - 4 * q);
    if (vec4 && nc == 4) {
      const float4 a = *reinterpret_cast<const float4*>(p00), bq = *reinterpret_cast<const float4*>(p01);
      const float4 c = *reinterpret_cast<const float4*>(p10), d = *reinterpret_cast<const float4*>(p11);
      float4 r;
      r.x = hy * (hx * a.x + lx * bq.x) + ly * (hx * c.x + lx * d.x);
      r.y = hy * (hx * a.y + lx * bq.y) + ly * (hx * c.y + lx * d.y);
      r.z = hy * (hx * a.z + lx * bq.z) + ly * (hx * c.z + lx * d.z);
      r.w = hy * (hx * a.w + lx * bq.w) + ly * (hx * c.w + lx * d.w);
      *reinterpret_cast<float4*>(o) = r;
    } else {
      for (int k = 0; k < nc; ++k)
        o[k] = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
    }
  }
}

// ------------------------------------------------------------------ C ABI -------------

extern "C" size_t sr_conv_packed_weight_floats(int Cout, int Cin, int ksize) {
  if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
  const size_t G = (size_t)((Cin + SR_CK - 1) / SR_CK) * 2, Co_pad = (size_t)((Cout + 31) / 32) * 32;
  return (size_t)ksize * ksize * G * 2 * Co_pad * 4;
}

extern "C" int sr_conv_pack_weights(const float* weight, int Cout, int Cin, int ksize, float* packed, void* stream_) {
  if (!weight || !packed || Cout <= 0 || Cin <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (ksize != 1 && ksize != 3) return SR_ERR_UNSUPPORTED;
  const int G = ((Cin + SR_CK - 1) / SR_CK) * 2, Co_pad = ((Cout + 31) / 32) * 32;
  hipLaunchKernelGGL(sr_conv_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream_, weight, packed, Cout, Cin,
                     ksize * ksize, G, Co_pad);
  return sr_hip_rc(hipGetLastError());
}

template <int KS, int S, int MT>
static int sr_conv_dispatch_nt(const SrConvParams& p, int B, hipStream_t stream) {
  const int TH = 4 * MT;
  const int tiles_y = (p.Ho + TH - 1) / TH;
  const int nt = (p.Co_pad % 64 == 0) ? 2 : 1;
  dim3 grid(p.tiles_x * tiles_y, p.Co_pad / (32 * nt), B), block(256);
  if (nt == 2) hipLaunchKernelGGL((sr_conv_kernel<KS, S, MT, 2>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((sr_conv_kernel<KS, S, MT, 1>), grid, block, 0, stream, p);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_conv2d_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                  const float* packed_weight, const float* bias, const float* residual,
                                  int64_t res_batch_stride, int res_pix_stride, float* out,
                                  int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                  int Cout, int ksize, int stride, float leaky_slope, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !packed_weight || !out) return SR_ERR_INVALID_ARGUMENT;
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return SR_ERR_UNSUPPORTED;
  const int pad = ksize / 2;
  SrConvParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.wp = packed_weight; p.bias = bias;
  p.res = residual; p.res_sb = res_batch_stride; p.res_sp = res_pix_stride;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.Ho = (H + 2 * pad - ksize) / stride + 1;
  p.Wo = (W + 2 * pad - ksize) / stride + 1;
  p.Co_pad = ((Cout + 31) / 32) * 32;
  p.G = ((Cin + SR_CK - 1) / SR_CK) * 2;
  p.tiles_x = (p.Wo + SR_TW - 1) / SR_TW;
  p.slope = leaky_slope;
  p.vec4 = (((uintptr_t)in & 15) == 0) && (in_pix_stride % 4 == 0) && (in_batch_stride % 4 == 0);
  hipStream_t stream = (hipStream_t)stream_;
  if (ksize == 3 && stride == 1) return sr_conv_dispatch_nt<3, 1, 2>(p, B, stream);
  if (ksize == 3 && stride == 2) return sr_conv_dispatch_nt<3, 2, 1>(p, B, stream);
  if (ksize == 1 && stride == 1) return sr_conv_dispatch_nt<1, 1, 2>(p, B, stream);
  return sr_conv_dispatch_nt<1, 2, 1>(p, B, stream);
}

extern "C" int sr_upsample2x_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out,
                                      int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C,
                                      void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !out) return SR_ERR_INVALID_ARGUMENT;
  const int vec4 = (((uintptr_t)in & 15) == 0) && (((uintptr_t)out & 15) == 0) && (in_pix_stride % 4 == 0) &&
                   (out_pix_stride % 4 == 0) && (in_batch_stride % 4 == 0) && (out_batch_stride % 4 == 0);
  const int64_t total = (int64_t)4 * H * W * ((C + 3) / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sr_upsample2x_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream_, in, in_batch_stride,
                     in_pix_stride, out, out_batch_stride, out_pix_stride, H, W, C, vec4);
  return sr_hip_rc(hipGetLastError());
}
