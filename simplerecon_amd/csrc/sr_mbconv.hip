// sr_mbconv.hip -- the parts of an MBConv block that are not dense convolutions, for gfx950.
//
// The image-prior encoder of the reference is timm's `tf_efficientnetv2_s` feature pyramid
// (reference modules/depth_model.py:110-116; third-party architecture).  Its stem,
// ConvBnAct / FusedMBConv stages and every 1x1 convolution go through the MFMA kernels of sr_conv.hip
// (SiLU epilogue = SR_ACT_SILU, TF-"SAME" padding = sr_conv2d_padded_nhwc_fwd).  This file holds the rest
// of the inverted-residual blocks of stages 3-5:
//   * sr_dwconv3x3_nhwc_fwd   depthwise 3x3 (stride 1 / 2, explicit padding) + folded BatchNorm + SiLU,
//                             with the squeeze-excite global average pool as a by-product (deterministic
//                             per-block partial sums, no atomics);
//   * sr_se_gate_fwd          squeeze-excite gate  sigmoid(W2 silu(W1 mean + b1) + b2)  per image;
//   * sr_scale_channels_nhwc_fwd   x[b, p, c] *= gate[b, c];
//   * sr_add_nhwc_fwd         a + b (the identity skip of stage 0's ConvBnAct blocks, where the sum follows the
//                             activation and therefore cannot ride in a convolution epilogue).
// All of them are byte shuffling on small maps (<= 30x40 at 640x480 input), bound by load latency rather than
// bandwidth: channels-last, one 16-byte access per lane, many short workgroups, and 8-18 independent loads in
// flight per lane (explicit register arrays) -- see DESIGN.md §3.7 for the measurements.
#include "sr_common.h"

namespace {

constexpr int DW_CH = 64;       // channels per workgroup (16 lanes x float4)
constexpr int DW_XL = 16;       // pixel lanes per workgroup

struct SrDwParams {
  const float* in; int64_t in_sb; int in_sp;
  const float* w;                 // [9][C]: tap-major, BatchNorm scale folded in
  const float* bias;              // [C] (BatchNorm shift) or null
  float* out; int64_t out_sb; int out_sp;
  float* pool;                    // [B][bands][C] partial sums of the activated output, or null
  int H, W, C, Ho, Wo, stride, pad_y, pad_x, rows, bands;
  float slope;
};

__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

__global__ __launch_bounds__(256) void sr_dwconv3x3_kernel(SrDwParams p) {
  __shared__ float4 red[DW_XL][DW_CH / 4];
  const int c4 = threadIdx.x & 15, xl = threadIdx.x >> 4;
  const int c = blockIdx.x * DW_CH + 4 * c4;
  const int band = blockIdx.y, b = blockIdx.z;
  const bool live = c < p.C;
  float4 wt[9], bv = make_float4(0.f, 0.f, 0.f, 0.f), sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 9; ++t)
    wt[t] = live ? *reinterpret_cast<const float4*>(p.w + (int64_t)t * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && p.bias) bv = *reinterpret_cast<const float4*>(p.bias + c);
  const float* __restrict__ inb = p.in + (int64_t)b * p.in_sb + c;
  float* __restrict__ outb = p.out + (int64_t)b * p.out_sb + c;
  const int oy0 = band * p.rows;
  const int npx = (min(p.Ho, oy0 + p.rows) - oy0) * p.Wo;   // output pixels of this band, row-major
  if (live) {
    // two output pixels per pass: 18 independent 16-byte loads in flight (the kernel is latency-bound)
    for (int q0 = xl; q0 < npx; q0 += 2 * DW_XL) {
      float4 v[2][9];
      int opix[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = q0 + u * DW_XL;
        const bool have = q < npx;
        const int oy = oy0 + (have ? q / p.Wo : 0), ox = have ? q % p.Wo : 0;
        opix[u] = have ? oy * p.Wo + ox : -1;
        const int iy0 = oy * p.stride - p.pad_y, ix0 = ox * p.stride - p.pad_x;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int iy = iy0 + ky, ix = ix0 + kx;
            const bool ok = have && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
            const float4 t = *reinterpret_cast<const float4*>(inb + (int64_t)(ok ? iy * p.W + ix : 0) * p.in_sp);
            v[u][ky * 3 + kx] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (opix[u] < 0) continue;
        float4 acc = bv;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = f4fma(v[u][t], wt[t], acc);
        {
          float a4[4] = {acc.x, acc.y, acc.z, acc.w};
          sr_activate_group(a4, sr_uniform(p.slope));
          acc = make_float4(a4[0], a4[1], a4[2], a4[3]);
        }
        *reinterpret_cast<float4*>(outb + (int64_t)opix[u] * p.out_sp) = acc;
        sum.x += acc.x; sum.y += acc.y; sum.z += acc.z; sum.w += acc.w;
      }
    }
  }
  if (p.pool) {  // fixed-order tree over the 16 pixel lanes: run-to-run deterministic
    red[xl][c4] = sum;
    __syncthreads();
    for (int s = DW_XL / 2; s > 0; s >>= 1) {
      if (xl < s) {
        const float4 a = red[xl][c4], o = red[xl + s][c4];
        red[xl][c4] = make_float4(a.x + o.x, a.y + o.y, a.z + o.z, a.w + o.w);
      }
      __syncthreads();
    }
    if (xl == 0 && live)
      *reinterpret_cast<float4*>(p.pool + ((int64_t)b * p.bands + band) * p.C + c) = red[0][c4];
  }
}

struct SrSeParams {
  const float* pool; int bands; float inv_count;
  const float* w1; const float* b1;   // [rd][C], [rd]
  const float* w2; const float* b2;   // [C][rd], [C]
  float* gate;                        // [B][C]
  int C, rd;
};

__global__ __launch_bounds__(256) void sr_se_gate_kernel(SrSeParams p) {
  extern __shared__ float sm[];
  float* mean = sm;            // [C]
  float* hid = sm + p.C;       // [rd]
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float s = 0.f;
    for (int k = 0; k < p.bands; ++k) s += p.pool[((int64_t)b * p.bands + k) * p.C + c];
    mean[c] = s * p.inv_count;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < p.rd; j += 4) {
    float s = 0.f;
    for (int c = lane; c < p.C; c += 64) s = fmaf(p.w1[(int64_t)j * p.C + c], mean[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) hid[j] = sr_activate(s + (p.b1 ? p.b1[j] : 0.f), SR_ACT_SILU);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float s = p.b2 ? p.b2[c] : 0.f;
    for (int j = 0; j < p.rd; ++j) s = fmaf(p.w2[(int64_t)c * p.rd + j], hid[j], s);
    p.gate[(int64_t)b * p.C + c] = 1.0f / (1.0f + __expf(-s));
  }
}

// Squeeze-excite in two short, wide launches (the one-workgroup-per-image sr_se_gate_kernel streams both weight
// matrices through 4 waves: 58 us at C = 1536):
//   sr_se_hidden_kernel   one wave per (image, hidden unit): hid = silu(w1[j] . mean + b1[j]);
//   sr_se_scale_kernel    one workgroup per (image, 64 channels): gates from hid, then scales those channels of
//                         every pixel of the image.
// (Both kernels are pure latency: every loop keeps 8-16 independent loads in flight through explicit register
// arrays -- with run-time trip counts the compiler otherwise waits on each load before issuing the next.)
__global__ __launch_bounds__(256) void sr_se_hidden_kernel(SrSeParams p, float* __restrict__ hidden) {
  extern __shared__ float mean[];   // [C]: all 256 threads finish the average pool, then one wave per hidden unit
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave, b = blockIdx.y;
  const float* __restrict__ pool = p.pool + (int64_t)b * p.bands * p.C;
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float m = 0.f;
    for (int k0 = 0; k0 < p.bands; k0 += 16) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = (k0 + k < p.bands) ? pool[(int64_t)(k0 + k) * p.C + c] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) m += v[k];   // band order: deterministic
    }
    mean[c] = m * p.inv_count;
  }
  __syncthreads();
  if (j >= p.rd) return;
  const float* __restrict__ w = p.w1 + (int64_t)j * p.C;
  float s = 0.f;
  for (int c0 = 0; c0 < p.C; c0 += 64 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int c = c0 + 64 * u + lane; v[u] = c < p.C ? w[c] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int c = c0 + 64 * u + lane; s = fmaf(v[u], c < p.C ? mean[c] : 0.f, s); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) hidden[(int64_t)b * p.rd + j] = sr_activate(s + (p.b1 ? p.b1[j] : 0.f), SR_ACT_SILU);
}

__global__ __launch_bounds__(256) void sr_se_scale_kernel(SrSeParams p, const float* __restrict__ hidden,
                                                          const float* __restrict__ in, int64_t in_sb, int in_sp,
                                                          float* __restrict__ out, int64_t out_sb, int out_sp, int HW) {
  __shared__ float hid[256];
  __shared__ __attribute__((aligned(16))) float g[DW_CH];
  const int b = blockIdx.y, c0 = blockIdx.x * DW_CH;
  for (int j = threadIdx.x; j < p.rd; j += 256) hid[j] = hidden[(int64_t)b * p.rd + j];
  __syncthreads();
  {  // gates of the 64 channels: 4 threads per channel, each a quarter of the hidden units, 8 loads in flight
    const int cl = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int c = c0 + cl;
    float s = 0.f;
    if (c < p.C) {
      const float* __restrict__ w = p.w2 + (int64_t)c * p.rd;
      for (int j0 = q; j0 < p.rd; j0 += 4 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = j0 + 4 * u; v[u] = j < p.rd ? w[j] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = j0 + 4 * u; s = fmaf(v[u], j < p.rd ? hid[j] : 0.f, s); }
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (q == 0) {
      if (c < p.C) {
        s = 1.0f / (1.0f + __expf(-(s + (p.b2 ? p.b2[c] : 0.f))));
        if (p.gate) p.gate[(int64_t)b * p.C + c] = s;
      }
      g[cl] = s;
    }
  }
  __syncthreads();
  const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = c0 + 4 * c4;
  if (c >= p.C) return;
  const float4 gv = *reinterpret_cast<const float4*>(&g[4 * c4]);
  const float* __restrict__ inb = in + (int64_t)b * in_sb + c;
  float* __restrict__ outb = out + (int64_t)b * out_sb + c;
  for (int px0 = pl; px0 < HW; px0 += 16 * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int px = px0 + 16 * u;
      v[u] = px < HW ? *reinterpret_cast<const float4*>(inb + (int64_t)px * in_sp) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int px = px0 + 16 * u;
      if (px < HW)
        *reinterpret_cast<float4*>(outb + (int64_t)px * out_sp) =
            make_float4(v[u].x * gv.x, v[u].y * gv.y, v[u].z * gv.z, v[u].w * gv.w);
    }
  }
}

__global__ __launch_bounds__(256) void sr_scale_channels_kernel(const float* __restrict__ in, int64_t in_sb, int in_sp,
                                                                const float* __restrict__ gate,
                                                                float* __restrict__ out, int64_t out_sb, int out_sp,
                                                                int HW, int C4) {
  const int b = blockIdx.y;
  const int64_t n = (int64_t)HW * C4;
  for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int px = (int)(e / C4), c4 = (int)(e - (int64_t)px * C4);
    const float4 g = *reinterpret_cast<const float4*>(gate + ((int64_t)b * C4 + c4) * 4);
    float4 v = *reinterpret_cast<const float4*>(in + (int64_t)b * in_sb + (int64_t)px * in_sp + 4 * c4);
    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    *reinterpret_cast<float4*>(out + (int64_t)b * out_sb + (int64_t)px * out_sp + 4 * c4) = v;
  }
}

__global__ __launch_bounds__(256) void sr_add_kernel(const float* __restrict__ a, int64_t a_sb, int a_sp,
                                                     const float* __restrict__ b_, int64_t b_sb, int b_sp,
                                                     float* __restrict__ out, int64_t out_sb, int out_sp, int HW,
                                                     int C4) {
  const int b = blockIdx.y;
  const int64_t n = (int64_t)HW * C4;
  for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int px = (int)(e / C4), c4 = (int)(e - (int64_t)px * C4);
    const float4 u = *reinterpret_cast<const float4*>(a + (int64_t)b * a_sb + (int64_t)px * a_sp + 4 * c4);
    const float4 v = *reinterpret_cast<const float4*>(b_ + (int64_t)b * b_sb + (int64_t)px * b_sp + 4 * c4);
    *reinterpret_cast<float4*>(out + (int64_t)b * out_sb + (int64_t)px * out_sp + 4 * c4) =
        make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}

// ---- RGB stem: conv 3x3 / stride 2 on a 3-channel image (any strides) -> COUT channels, channels-last ----
// EfficientNetV2's conv_stem (3 -> 24, TF-"SAME": one row / column of zeros below / right on even images).  As an
// implicit GEMM on the matrix cores its K = 27 pads to 9 x 16 channels and its input rows cannot be read as float4
// (sr_conv_kernel<3,2,1,1,32,false>: 169 us per 8 images, 3 % of the MFMA peak).  It is 0.4 GFLOP on 29 MB in / 59 MB
// out: byte work.  One lane = one output pixel x all COUT channels; the 27 x COUT weights are wave-uniform (scalar
// loads, SGPR operands of the FMAs); lanes of a wave are consecutive pixels of a row, so the 27 input loads are
// contiguous (stride 2) and the 6 float4 stores of a wave cover 6 KB without gaps.
template <int COUT>
__global__ __launch_bounds__(256) void sr_rgb_stem3x3s2_kernel(const float* __restrict__ img, int64_t sb, int64_t sc, int64_t sy,
                                                               int64_t sx, const float* __restrict__ w /*[27][COUT]*/,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int64_t out_sb, int out_sp, int H, int W, int Ho, int Wo,
                                                               int pad_y, int pad_x, float slope) {
  const int b = blockIdx.y;
  const int64_t npix = (int64_t)Ho * Wo;
  for (int64_t pix = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * blockDim.x) {
    const int oy = (int)(pix / Wo), ox = (int)(pix - (int64_t)oy * Wo);
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = bias ? bias[c] : 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - pad_y + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - pad_x + kx;
        const bool ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
        const float* q = img + b * sb + (ok ? iy * sy + ix * sx : 0);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float v = ok ? q[ci * sc] : 0.0f;
          const float* wr = w + ((ky * 3 + kx) * 3 + ci) * COUT;
#pragma unroll
          for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
        }
      }
    }
    sr_activate_group(acc, sr_uniform(slope));
    float* o = out + b * out_sb + pix * out_sp;
#pragma unroll
    for (int c = 0; c < COUT; c += 4) *reinterpret_cast<float4*>(o + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  }
}

inline bool aligned16(const void* ptr) { return (((uintptr_t)ptr) & 15) == 0; }

}  // namespace

extern "C" int sr_dwconv3x3_pool_bands(int Ho) {
  if (Ho <= 0) return 0;
  // output rows per workgroup: short bands = many workgroups (the kernel is latency-bound on these small maps:
  // 15x20 x 1536 channels x 8 images with 4-row bands is only 2 workgroups per CU)
  const int rows = Ho > 64 ? 4 : 2;
  return (Ho + rows - 1) / rows;
}

extern "C" int sr_dwconv3x3_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* weight9c,
                                     const float* bias, float* out, int64_t out_batch_stride, int out_pix_stride,
                                     float* pool_partial, int B, int H, int W, int C, int stride, int pad_top,
                                     int pad_left, int pad_bottom, int pad_right, float leaky_slope, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (stride != 1 && stride != 2) return SR_ERR_UNSUPPORTED;
  if (pad_top < 0 || pad_left < 0 || pad_bottom < 0 || pad_right < 0 || pad_top > 2 || pad_left > 2 ||
      pad_bottom > 2 || pad_right > 2 || H + pad_top + pad_bottom < 3 || W + pad_left + pad_right < 3)
    return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !weight9c || !out) return SR_ERR_INVALID_ARGUMENT;
  if (C % 4 != 0 || in_pix_stride % 4 != 0 || out_pix_stride % 4 != 0 || in_batch_stride % 4 != 0 ||
      out_batch_stride % 4 != 0 || !aligned16(in) || !aligned16(out) || !aligned16(weight9c) ||
      (bias && !aligned16(bias)) || (pool_partial && !aligned16(pool_partial)))
    return SR_ERR_UNSUPPORTED;   // 16-byte channel quads only (every EfficientNetV2 width is a multiple of 8)
  SrDwParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.w = weight9c; p.bias = bias;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.pool = pool_partial;
  p.H = H; p.W = W; p.C = C; p.stride = stride; p.pad_y = pad_top; p.pad_x = pad_left;
  p.Ho = (H + pad_top + pad_bottom - 3) / stride + 1;
  p.Wo = (W + pad_left + pad_right - 3) / stride + 1;
  p.bands = sr_dwconv3x3_pool_bands(p.Ho);
  p.rows = (p.Ho + p.bands - 1) / p.bands;
  p.slope = leaky_slope;
  if ((int64_t)H * W * in_pix_stride >= (1ll << 31) || (int64_t)p.Ho * p.Wo * out_pix_stride >= (1ll << 31))
    return SR_ERR_UNSUPPORTED;
  dim3 grid((C + DW_CH - 1) / DW_CH, p.bands, B);
  hipLaunchKernelGGL(sr_dwconv3x3_kernel, grid, dim3(256), 0, (hipStream_t)stream_, p);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_se_gate_fwd(const float* pool_partial, int bands, int pixels, const float* w_reduce,
                              const float* b_reduce, const float* w_expand, const float* b_expand, float* gate, int B,
                              int C, int rd, void* stream_) {
  if (B < 0 || C <= 0 || rd <= 0 || bands <= 0 || pixels <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!pool_partial || !w_reduce || !w_expand || !gate) return SR_ERR_INVALID_ARGUMENT;
  const size_t lds = (size_t)(C + rd) * sizeof(float);
  if (lds > 64 * 1024) return SR_ERR_UNSUPPORTED;
  SrSeParams p;
  p.pool = pool_partial; p.bands = bands; p.inv_count = 1.0f / (float)pixels;
  p.w1 = w_reduce; p.b1 = b_reduce; p.w2 = w_expand; p.b2 = b_expand; p.gate = gate; p.C = C; p.rd = rd;
  hipLaunchKernelGGL(sr_se_gate_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream_, p);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_scale_channels_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                          const float* gate, float* out, int64_t out_batch_stride, int out_pix_stride,
                                          int B, int H, int W, int C, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !gate || !out) return SR_ERR_INVALID_ARGUMENT;
  if (C % 4 != 0 || in_pix_stride % 4 != 0 || out_pix_stride % 4 != 0 || in_batch_stride % 4 != 0 ||
      out_batch_stride % 4 != 0 || !aligned16(in) || !aligned16(out) || !aligned16(gate))
    return SR_ERR_UNSUPPORTED;
  const int64_t n = (int64_t)H * W * (C / 4);
  const int blocks = (int)((n + 256 * 4 - 1) / (256 * 4));
  dim3 grid(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks), B);
  hipLaunchKernelGGL(sr_scale_channels_kernel, grid, dim3(256), 0, (hipStream_t)stream_, in, in_batch_stride,
                     in_pix_stride, gate, out, out_batch_stride, out_pix_stride, H * W, C / 4);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_add_nhwc_fwd(const float* a, int64_t a_batch_stride, int a_pix_stride, const float* b,
                               int64_t b_batch_stride, int b_pix_stride, float* out, int64_t out_batch_stride,
                               int out_pix_stride, int B, int H, int W, int C, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!a || !b || !out) return SR_ERR_INVALID_ARGUMENT;
  if (C % 4 != 0 || a_pix_stride % 4 != 0 || b_pix_stride % 4 != 0 || out_pix_stride % 4 != 0 ||
      a_batch_stride % 4 != 0 || b_batch_stride % 4 != 0 || out_batch_stride % 4 != 0 || !aligned16(a) ||
      !aligned16(b) || !aligned16(out))
    return SR_ERR_UNSUPPORTED;
  const int64_t n = (int64_t)H * W * (C / 4);
  const int blocks = (int)((n + 256 * 4 - 1) / (256 * 4));
  dim3 grid(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks), B);
  hipLaunchKernelGGL(sr_add_kernel, grid, dim3(256), 0, (hipStream_t)stream_, a, a_batch_stride, a_pix_stride, b,
                     b_batch_stride, b_pix_stride, out, out_batch_stride, out_pix_stride, H * W, C / 4);
  return sr_hip_rc(hipGetLastError());
}

// act(conv3x3 / stride 2 (3 -> Cout) + bias) of an RGB image with explicit top / left zero padding (bottom / right: whatever
// the output size needs); `weight27c` = [ky][kx][ci][Cout] with the eval-mode BatchNorm scale folded in.  Cout = 24 only.
extern "C" int sr_rgb_stem3x3s2_fwd(const float* image, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* weight27c,
                                    const float* bias, float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H,
                                    int W, int Cout, int pad_top, int pad_left, int Ho, int Wo, float act_code, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || pad_top < 0 || pad_left < 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!image || !weight27c || !out) return SR_ERR_INVALID_ARGUMENT;
  if (Cout != 24 || out_pix_stride % 4 != 0 || out_batch_stride % 4 != 0 || !aligned16(out)) return SR_ERR_UNSUPPORTED;
  if (2 * (Ho - 1) - pad_top >= H || 2 * (Wo - 1) - pad_left >= W) return SR_ERR_INVALID_ARGUMENT;   // a window with no pixel
  const int64_t npix = (int64_t)Ho * Wo;
  const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
  hipLaunchKernelGGL((sr_rgb_stem3x3s2_kernel<24>), dim3(blocks, B), dim3(256), 0, (hipStream_t)stream_, image, sb, sc, sy, sx,
                     weight27c, bias, out, out_batch_stride, out_pix_stride, H, W, Ho, Wo, pad_top, pad_left, act_code);
  return sr_hip_rc(hipGetLastError());
}

// Squeeze-excite gates only ([B][C]), for consumers that apply them themselves (sr_pw_conv_nhwc_fwd scales the projection's
// input while it loads it): the two short launches of sr_se_scale_nhwc_fwd without the scaling pass over the map.
extern "C" int sr_se_gate2_fwd(const float* pool_partial, int bands, int pixels, const float* w_reduce,
                               const float* b_reduce, const float* w_expand, const float* b_expand, float* hidden,
                               float* gate, int B, int C, int rd, void* stream_) {
  if (B < 0 || C <= 0 || rd <= 0 || bands <= 0 || pixels <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!pool_partial || !w_reduce || !w_expand || !hidden || !gate) return SR_ERR_INVALID_ARGUMENT;
  if (rd > 256 || C > 16384) return SR_ERR_UNSUPPORTED;
  SrSeParams p;
  p.pool = pool_partial; p.bands = bands; p.inv_count = 1.0f / (float)pixels;
  p.w1 = w_reduce; p.b1 = b_reduce; p.w2 = w_expand; p.b2 = b_expand; p.gate = gate; p.C = C; p.rd = rd;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(sr_se_hidden_kernel, dim3((rd + 3) / 4, B), dim3(256), (size_t)C * sizeof(float), stream, p, hidden);
  hipLaunchKernelGGL(sr_se_scale_kernel, dim3((C + DW_CH - 1) / DW_CH, B), dim3(256), 0, stream, p, (const float*)hidden,
                     (const float*)nullptr, (int64_t)0, 0, (float*)nullptr, (int64_t)0, 0, 0);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_se_scale_nhwc_fwd(const float* pool_partial, int bands, const float* w_reduce, const float* b_reduce,
                                    const float* w_expand, const float* b_expand, float* hidden, const float* in,
                                    int64_t in_batch_stride, int in_pix_stride, float* out, int64_t out_batch_stride,
                                    int out_pix_stride, float* gate, int B, int H, int W, int C, int rd,
                                    void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || rd <= 0 || bands <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!pool_partial || !w_reduce || !w_expand || !hidden || !in || !out) return SR_ERR_INVALID_ARGUMENT;
  if (rd > 256 || C > 16384 || C % 4 != 0 || in_pix_stride % 4 != 0 || out_pix_stride % 4 != 0 || in_batch_stride % 4 != 0 ||
      out_batch_stride % 4 != 0 || !aligned16(in) || !aligned16(out))
    return SR_ERR_UNSUPPORTED;
  SrSeParams p;
  p.pool = pool_partial; p.bands = bands; p.inv_count = 1.0f / (float)((int64_t)H * W);
  p.w1 = w_reduce; p.b1 = b_reduce; p.w2 = w_expand; p.b2 = b_expand; p.gate = gate; p.C = C; p.rd = rd;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(sr_se_hidden_kernel, dim3((rd + 3) / 4, B), dim3(256), (size_t)C * sizeof(float), stream, p, hidden);
  hipLaunchKernelGGL(sr_se_scale_kernel, dim3((C + DW_CH - 1) / DW_CH, B), dim3(256), 0, stream, p,
                     (const float*)hidden, in, in_batch_stride, in_pix_stride, out, out_batch_stride, out_pix_stride,
                     H * W);
  return sr_hip_rc(hipGetLastError());
}
