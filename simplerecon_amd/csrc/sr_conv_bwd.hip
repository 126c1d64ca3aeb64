// sr_conv_bwd.hip -- backward pieces of the conv stack (BasicBlock / CVEncoder / DepthDecoderPP training path; reference
// train.py:126-145 differentiates modules/layers.py:24-85 and modules/networks.py:20-127 through autograd) for gfx950.
//
//   data gradient   = the FORWARD kernels (sr_conv.hip / sr_wino.hip) on the flipped, transposed weight
//                     (sr_conv_flip_transpose_weights); for a stride-2 conv on the zero-stuffed output gradient
//                     (sr_zero_stuff2x_nhwc): dL/dx = conv3x3_s1(stuff(dL/dy), flip(W)^T)
//   weight gradient = sr_conv_wgrad_nhwc: dW[co,ci,ky,kx] = sum_{b,y,x} dL/dy[b,co,y,x] * x[b,ci,s*y+ky-p,s*x+kx-p], a
//                     [Cout] x [pixels] x [Cin*k*k] product on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, M = co,
//                     N = ci, K = 2 output pixels per step).  A workgroup owns a 64 x 64 (co, ci) block and a strided
//                     share of the (image, output row, 32-pixel segment) items: gradient rows and the k input rows of a
//                     segment are staged in LDS, wave (co half, ci half) keeps its k*k accumulator tiles (144 registers
//                     for 3x3) over ALL its items and writes them once as a partial [k*k][64][64] slab; a second kernel
//                     adds the slabs of a block in index order (deterministic; a first version flushed with fp32
//                     atomics: 18.9 M atomics for one 64 -> 64 layer at 2 x 240 x 320 made the backward 13x the forward);
//   bias gradient   = sr_bias_grad_nhwc (column sums of dL/dy);
//   activation      = sr_act_bwd_nhwc: g * act'(y) from the saved OUTPUT (LeakyReLU with slope > 0 preserves the sign);
//   bilinear x2     = sr_upsample2x_bwd_nhwc, the exact adjoint of sr_upsample2x_nhwc_fwd (same clamped taps).
#include "sr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------ weight gradient ----

#define WG_P 32        // output pixels of a segment
#define WG_CT 64       // channels per block side

struct SrWgradParams {
  const float* x; int64_t x_sb; int x_sp;      // input  [B, H, W, Cin]  (channels-last view)
  const float* g; int64_t g_sb; int g_sp;      // dL/dy  [B, Ho, Wo, Cout]
  float* part;                                  // [blocks][wgs_per_block][k*k][64 co][64 ci] partial slabs
  int B, H, W, Cin, Cout, Ho, Wo, stride, pad, pad_x;   // pad = rows above, pad_x = columns left of the image
  int co_blocks, ci_blocks, items, wgs_per_block;
  int vec_x, vec_g;   // input / gradient rows are whole 16-byte aligned channel quads: float4 staging loads
};

// PIPE (stride 1): the staging loads of the NEXT item are issued into registers before the current item's MFMA loop and
// stored to LDS after it (one float4 per slot, slot coordinates fixed per thread -- no divisions, no load -> store ->
// load latency chain per item: the first version issued its 7 + 2 loads per item one at a time, each followed by its
// LDS store, ~7 us of exposed latency against 3.8 us of MFMA work).  Stride 2 keeps the direct form.
template <int KS, bool PIPE>
__global__ __launch_bounds__(256) void sr_conv_wgrad_kernel(SrWgradParams p) {
  constexpr int TAPS = KS * KS;
  // LDS: gradient segment [WG_P][64 co] + input rows [KS][span][64 ci], span = stride*(WG_P-1) + KS (<= 65)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int span = PIPE ? (WG_P - 1 + KS) : p.stride * (WG_P - 1) + KS;
  float* gs = lds;                       // [WG_P][WG_CT]
  float* xs = lds + WG_P * WG_CT;        // [KS][span][WG_CT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int blk = blockIdx.x / p.wgs_per_block, sub = blockIdx.x - blk * p.wgs_per_block;
  const int cob = blk / p.ci_blocks, cib = blk - cob * p.ci_blocks;
  const int co0 = cob * WG_CT, ci0 = cib * WG_CT;
  const int coh = wave & 1, cih = wave >> 1;   // this wave's 32 x 32 quadrant of the block
  const int segs = (p.Wo + WG_P - 1) / WG_P;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // register staging (PIPE): GS gradient slots + XS input slots of one float4 each
  constexpr int SPAN1 = WG_P - 1 + KS;
  constexpr int GS = WG_P * (WG_CT / 4) / 256;                       // 2
  constexpr int XS = PIPE ? (KS * SPAN1 * (WG_CT / 4) + 255) / 256 : 1;   // 7 (3x3), 2 (1x1)
  float4 gq[GS], xq[XS];
  auto fetch = [&](int item) {
    int it = item;
    const int seg = it % segs; it /= segs;
    const int oy = it % p.Ho;
    const int b = it / p.Ho;
    const int ox0 = seg * WG_P;
#pragma unroll
    for (int u = 0; u < GS; ++u) {
      const int e = tid + 256 * u, px = e >> 4, q = e & 15;
      const int ox = ox0 + px, c = co0 + 4 * q;
      const bool ok = ox < p.Wo && c < p.Cout;
      const float* src = p.g + (int64_t)b * p.g_sb + (ok ? (oy * p.Wo + ox) * p.g_sp + c : 0);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.vec_g) { const float4 t = *reinterpret_cast<const float4*>(src); if (ok) v = t; }
      else if (ok) {
        v.x = src[0];
        if (c + 1 < p.Cout) v.y = src[1];
        if (c + 2 < p.Cout) v.z = src[2];
        if (c + 3 < p.Cout) v.w = src[3];
      }
      gq[u] = v;
    }
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      const int e = tid + 256 * u, q = e & 15, pc = e >> 4;
      const int col = pc % SPAN1, ky = pc / SPAN1;   // compile-time divisor
      const int iy = oy + ky - p.pad, ix = ox0 + col - p.pad_x, c = ci0 + 4 * q;
      const bool ok = ky < KS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin;
      const float* src = p.x + (int64_t)b * p.x_sb + (ok ? (iy * p.W + ix) * p.x_sp + c : 0);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.vec_x) { const float4 t = *reinterpret_cast<const float4*>(src); if (ok) v = t; }
      else if (ok) {
        v.x = src[0];
        if (c + 1 < p.Cin) v.y = src[1];
        if (c + 2 < p.Cin) v.z = src[2];
        if (c + 3 < p.Cin) v.w = src[3];
      }
      xq[u] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < GS; ++u) {
      const int e = tid + 256 * u;
      *reinterpret_cast<float4*>(&gs[(e >> 4) * WG_CT + 4 * (e & 15)]) = gq[u];
    }
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      const int e = tid + 256 * u;
      if (e < KS * SPAN1 * (WG_CT / 4)) *reinterpret_cast<float4*>(&xs[(e >> 4) * WG_CT + 4 * (e & 15)]) = xq[u];
    }
  };
  if (PIPE && sub < p.items) fetch(sub);

  for (int item = sub; item < p.items; item += p.wgs_per_block) {
    __syncthreads();  // previous item's fragments are consumed
    if (PIPE) {
      stash();
      __syncthreads();
      if (item + p.wgs_per_block < p.items) fetch(item + p.wgs_per_block);   // in flight under this item's MFMAs
    } else {
    int it = item;
    const int seg = it % segs; it /= segs;
    const int oy = it % p.Ho;
    const int b = it / p.Ho;
    const int ox0 = seg * WG_P;
    // stage dL/dy[b, oy, ox0 .. ox0+31, co0 .. co0+63] (zeros outside the map / the channel range)
    for (int e = tid; e < WG_P * (WG_CT / 4); e += 256) {
      const int px = e >> 4, q = e & 15;
      const int ox = ox0 + px, c = co0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ox < p.Wo && c < p.Cout) {
        const float* src = p.g + (int64_t)b * p.g_sb + ((int64_t)oy * p.Wo + ox) * p.g_sp + c;
        if (p.vec_g) v = *reinterpret_cast<const float4*>(src);   // whole 16-byte aligned quads (uniform flag)
        else {
          v.x = src[0];
          if (c + 1 < p.Cout) v.y = src[1];
          if (c + 2 < p.Cout) v.z = src[2];
          if (c + 3 < p.Cout) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&gs[px * WG_CT + 4 * q]) = v;
    }
    // stage the KS input rows iy = stride*oy + ky - pad, columns stride*ox0 - pad .. (zero padding outside the image)
    for (int e = tid; e < KS * span * (WG_CT / 4); e += 256) {
      const int q = e & 15;
      const int col = (e >> 4) % span, ky = (e >> 4) / span;
      const int iy = p.stride * oy + ky - p.pad, ix = p.stride * ox0 + col - p.pad_x, c = ci0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && c < p.Cin) {
        const float* src = p.x + (int64_t)b * p.x_sb + ((int64_t)iy * p.W + ix) * p.x_sp + c;
        if (p.vec_x) v = *reinterpret_cast<const float4*>(src);
        else {
          v.x = src[0];
          if (c + 1 < p.Cin) v.y = src[1];
          if (c + 2 < p.Cin) v.z = src[2];
          if (c + 3 < p.Cin) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&xs[(ky * span + col) * WG_CT + 4 * q]) = v;
    }
    __syncthreads();
    }
    // K loop: two output pixels per MFMA step (lane half kk picks the pixel)
    const int st = PIPE ? 1 : p.stride;
#pragma unroll 4
    for (int ps = 0; ps < WG_P; ps += 2) {
      const float a = gs[(ps + kk) * WG_CT + 32 * coh + i];          // A[m = co][k = pixel]
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ky = t / KS, kx = t - ky * KS;
        const float bq = xs[(ky * span + st * (ps + kk) + kx) * WG_CT + 32 * cih + i];   // B[k = pixel][n = ci]
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[t], 0, 0, 0);
      }
    }
  }
  // flush: acc[t][r] = dW[co0 + 32*coh + (r&3) + 8*(r>>2) + 4*kk][ci0 + 32*cih + i][tap t] -> this workgroup's slab
  float* slab = p.part + (size_t)blockIdx.x * TAPS * WG_CT * WG_CT;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      slab[(t * WG_CT + 32 * coh + (r & 3) + 8 * (r >> 2) + 4 * kk) * WG_CT + 32 * cih + i] = acc[t][r];
}

// dW[co][ci][t] = sum over the workgroups of a (co, ci) block of their slabs, in index order.  Threads walk the SLAB
// layout ([tap][co][ci], ci fastest: coalesced reads of every partial -- walking dW's layout instead read 4-byte words
// 16 KB apart and cost 12.7 ms of a 160 ms batch-8 training step); the scattered 4-byte writes are dW itself, once.
__global__ __launch_bounds__(256) void sr_conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                  int Cout, int Cin, int taps, int ci_blocks, int per,
                                                                  int blocks) {
  const int64_t slab = (int64_t)taps * WG_CT * WG_CT;
  const int64_t total = (int64_t)blocks * slab;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int blk = (int)(e / slab);
    const int64_t j = e - (int64_t)blk * slab;
    const int ci_l = (int)(j % WG_CT), co_l = (int)((j / WG_CT) % WG_CT), t = (int)(j / (WG_CT * WG_CT));
    const int co = (blk / ci_blocks) * WG_CT + co_l, ci = (blk % ci_blocks) * WG_CT + ci_l;
    if (co >= Cout || ci >= Cin) continue;
    const float* q = part + (size_t)blk * per * slab + j;
    float s = 0.0f;
    int k = 0;
    for (; k + 4 <= per; k += 4) {   // four loads in flight, added in index order
      const float v0 = q[(size_t)k * slab], v1 = q[(size_t)(k + 1) * slab], v2 = q[(size_t)(k + 2) * slab],
                  v3 = q[(size_t)(k + 3) * slab];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; k < per; ++k) s += q[(size_t)k * slab];
    dw[((int64_t)co * Cin + ci) * taps + t] = s;
  }
}

static void sr_wgrad_plan(int B, int Ho, int Wo, int Cin, int Cout, int& blocks, int& per, int& items) {
  blocks = ((Cout + WG_CT - 1) / WG_CT) * ((Cin + WG_CT - 1) / WG_CT);
  items = B * Ho * ((Wo + WG_P - 1) / WG_P);
  per = (2 * 256 + blocks - 1) / blocks;   // ~2 workgroups per CU in total
  if (per > items) per = items;
  if (per < 1) per = 1;
}

static size_t sr_wgrad_ws(int B, int Ho, int Wo, int Cin, int Cout, int ksize) {
  int blocks, per, items;
  sr_wgrad_plan(B, Ho, Wo, Cin, Cout, blocks, per, items);
  return (size_t)blocks * per * ksize * ksize * WG_CT * WG_CT * sizeof(float);
}

extern "C" size_t sr_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2))
    return 0;
  const int pad = ksize / 2;
  return sr_wgrad_ws(B, (H + 2 * pad - ksize) / stride + 1, (W + 2 * pad - ksize) / stride + 1, Cin, Cout, ksize);
}

// explicit-padding form (TF-"SAME" convolutions, the valid convolution behind a replicate pad): Ho x Wo is the
// gradient's size, `pad_top` / `pad_left` the zero rows / columns in front of the image
extern "C" size_t sr_conv_wgrad_padded_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ksize) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
  return sr_wgrad_ws(B, Ho, Wo, Cin, Cout, ksize);
}

extern "C" int sr_conv_wgrad_padded_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* grad_out,
                                         int64_t g_batch_stride, int g_pix_stride, float* d_weight, int B, int H, int W,
                                         int Cin, int Cout, int ksize, int stride, int pad_top, int pad_left, int Ho, int Wo,
                                         void* workspace, size_t workspace_bytes, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || pad_top < 0 || pad_left < 0)
    return SR_ERR_INVALID_ARGUMENT;
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return SR_ERR_UNSUPPORTED;
  if (!d_weight) return SR_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0) return sr_hip_rc(hipMemsetAsync(d_weight, 0, (size_t)Cout * Cin * ksize * ksize * sizeof(float), stream));
  if (!in || !grad_out || !workspace) return SR_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < sr_wgrad_ws(B, Ho, Wo, Cin, Cout, ksize)) return SR_ERR_WORKSPACE_TOO_SMALL;
  SrWgradParams p;
  p.x = in; p.x_sb = in_batch_stride; p.x_sp = in_pix_stride;
  p.g = grad_out; p.g_sb = g_batch_stride; p.g_sp = g_pix_stride;
  p.part = (float*)workspace;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.stride = stride; p.pad = pad_top; p.pad_x = pad_left;
  p.Ho = Ho; p.Wo = Wo;
  p.co_blocks = (Cout + WG_CT - 1) / WG_CT;
  p.ci_blocks = (Cin + WG_CT - 1) / WG_CT;
  int blocks, per;
  sr_wgrad_plan(B, Ho, Wo, Cin, Cout, blocks, per, p.items);
  p.wgs_per_block = per;
  p.vec_x = Cin % 4 == 0 && in_pix_stride % 4 == 0 && in_batch_stride % 4 == 0 && (((uintptr_t)in) & 15) == 0;
  p.vec_g = Cout % 4 == 0 && g_pix_stride % 4 == 0 && g_batch_stride % 4 == 0 && (((uintptr_t)grad_out) & 15) == 0;
  const int span = stride * (WG_P - 1) + ksize;
  const size_t lds = (size_t)(WG_P * WG_CT + ksize * span * WG_CT) * sizeof(float);
  // pixel offsets are 32-bit in the pipelined form
  const bool pipe = stride == 1 && (int64_t)H * W * in_pix_stride < 0x7fffffffLL && (int64_t)Ho * Wo * g_pix_stride < 0x7fffffffLL;
#define SR_WGRAD_LAUNCH(KSV, PV)                                                                                     \
  {                                                                                                                  \
    hipError_t e = hipFuncSetAttribute((const void*)sr_conv_wgrad_kernel<KSV, PV>,                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
    if (e != hipSuccess) return sr_hip_rc(e);                                                                        \
    hipLaunchKernelGGL((sr_conv_wgrad_kernel<KSV, PV>), dim3(blocks * per), dim3(256), lds, stream, p);              \
  }
  if (ksize == 3 && pipe) SR_WGRAD_LAUNCH(3, true)
  else if (ksize == 3) SR_WGRAD_LAUNCH(3, false)
  else if (pipe) SR_WGRAD_LAUNCH(1, true)
  else SR_WGRAD_LAUNCH(1, false)
#undef SR_WGRAD_LAUNCH
  int rc = sr_hip_rc(hipGetLastError());
  if (rc != SR_OK) return rc;
  const long total = (long)blocks * ksize * ksize * WG_CT * WG_CT;
  int rblocks = (int)((total + 255) / 256);
  if (rblocks > 4096) rblocks = 4096;
  hipLaunchKernelGGL(sr_conv_wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, stream, (const float*)workspace, d_weight,
                     Cout, Cin, ksize * ksize, p.ci_blocks, per, blocks);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_conv_wgrad_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* grad_out,
                                  int64_t g_batch_stride, int g_pix_stride, float* d_weight, int B, int H, int W, int Cin,
                                  int Cout, int ksize, int stride, void* workspace, size_t workspace_bytes, void* stream_) {
  if (H <= 0 || W <= 0) return SR_ERR_INVALID_ARGUMENT;
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return SR_ERR_UNSUPPORTED;
  const int pad = ksize / 2;
  return sr_conv_wgrad_padded_nhwc(in, in_batch_stride, in_pix_stride, grad_out, g_batch_stride, g_pix_stride, d_weight, B, H,
                                   W, Cin, Cout, ksize, stride, pad, pad, (H + 2 * pad - ksize) / stride + 1,
                                   (W + 2 * pad - ksize) / stride + 1, workspace, workspace_bytes, stream_);
}

// ------------------------------------------------------------------------------------------ bias gradient ------

__global__ __launch_bounds__(256) void sr_bias_grad_kernel(const float* __restrict__ g, int64_t g_sb, int g_sp, int B,
                                                          int HW, int C, float* __restrict__ db) {
  // block = 64 channels x 4 pixel lanes; grid.x = pixel chunks, grid.y = channel blocks
  __shared__ float red[4][64];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float s = 0.0f;
  if (c < C) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t px = (int64_t)blockIdx.x * 4 + part; px < total; px += (int64_t)gridDim.x * 4) {
      const int b = (int)(px / HW);
      s += g[(int64_t)b * g_sb + (px - (int64_t)b * HW) * g_sp + c];
    }
  }
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && c < C) atomicAdd(db + c, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

extern "C" int sr_bias_grad_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, float* d_bias, int B,
                                 int H, int W, int C, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || !d_bias) return SR_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  hipError_t e = hipMemsetAsync(d_bias, 0, (size_t)C * sizeof(float), stream);
  if (e != hipSuccess) return sr_hip_rc(e);
  if (B == 0) return SR_OK;
  if (!grad_out) return SR_ERR_INVALID_ARGUMENT;
  const long total = (long)B * H * W;
  int chunks = (int)((total + 255) / 256);
  if (chunks > 512) chunks = 512;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(sr_bias_grad_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, stream, grad_out, g_batch_stride,
                     g_pix_stride, B, H * W, C, d_bias);
  return sr_hip_rc(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ elementwise --------

// g_pre = g * act'(y) from the saved output y = act(pre): LeakyReLU slope >= 0 (y > 0 <=> pre > 0 for slope > 0;
// for slope == 0, ReLU, y > 0 is the derivative's support as well).  Dense [n] arrays (same layout for g, y, out).
__global__ __launch_bounds__(256) void sr_act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                        float* __restrict__ out, int64_t n, float slope) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = y[i] > 0.0f ? g[i] : g[i] * slope;
}

extern "C" int sr_act_bwd(const float* grad, const float* out_saved, float* grad_pre, int64_t n, float leaky_slope,
                          void* stream_) {
  if (n < 0 || leaky_slope < 0.0f) return SR_ERR_INVALID_ARGUMENT;
  if (n == 0) return SR_OK;
  if (!grad || !out_saved || !grad_pre) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, grad,
                     out_saved, grad_pre, n, leaky_slope);
  return sr_hip_rc(hipGetLastError());
}

// Fused elementwise backward of "conv + bias + LeakyReLU": one pass over the dense channels-last gradient does
//   grad_pre = grad * act'(saved output)        (skipped when out_saved is null: no activation, grad_pre = grad)
//   d_bias   = sum over pixels of grad_pre      (skipped when d_bias is null)
// with float4 accesses (4 channels per lane) instead of sr_act_bwd's and sr_bias_grad_nhwc's 4-byte ones and without
// their second read of the gradient (r02 training profile: 3.6 + 8.3 ms of a 170 ms batch-8 step).  The bias gradient
// is deterministic: per-block partial sums (pixel lanes added in lane order) go to the workspace [blocks][C] and a
// second kernel adds them in block order.
#define SR_AB_MAX_BLOCKS 1024
__global__ __launch_bounds__(256) void sr_act_bwd_bias_kernel(const float4* __restrict__ g, const float4* __restrict__ y,
                                                             float4* __restrict__ gp, float4* __restrict__ partial,
                                                             int64_t pixels, int CQ, int LQ, int PB, float slope) {
  __shared__ float4 red[256];
  const int q0 = threadIdx.x % LQ, p0 = threadIdx.x / LQ;   // channel quad / pixel lane; lanes with p0 >= PB idle
  // uniform trip count (the body holds workgroup barriers): every thread runs ceil(CQ / LQ) rounds, lanes whose quad is
  // past CQ in the last round only take part in the barriers (C > 1024 with C % 1024 != 0 used to diverge here)
  for (int qb = 0; qb < CQ; qb += LQ) {
    const int q = qb + q0;
    const bool live = q < CQ;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 < PB && live) {
      // 8 pixels per trip, their loads issued together (one load in flight per lane made this pass latency-bound)
      constexpr int U = 8;
      const int64_t step = (int64_t)gridDim.x * PB;
      for (int64_t px0 = (int64_t)blockIdx.x * PB + p0; px0 < pixels; px0 += step * U) {
        float4 v[U], o[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int64_t px = px0 + k * step;
          const bool ok = px < pixels;
          const int64_t e = (ok ? px : px0) * CQ + q;
          v[k] = g[e];
          if (y) o[k] = y[e];
          if (!ok) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int64_t px = px0 + k * step;
          if (y) {
            v[k].x = o[k].x > 0.0f ? v[k].x : v[k].x * slope;
            v[k].y = o[k].y > 0.0f ? v[k].y : v[k].y * slope;
            v[k].z = o[k].z > 0.0f ? v[k].z : v[k].z * slope;
            v[k].w = o[k].w > 0.0f ? v[k].w : v[k].w * slope;
            if (px < pixels) gp[px * CQ + q] = v[k];
          }
          s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w;   // pixel order: deterministic
        }
      }
    }
    if (partial) {   // uniform
      red[threadIdx.x] = s;
      __syncthreads();
      if (p0 == 0 && live) {
        float4 t = red[q0];
        for (int k = 1; k < PB; ++k) { const float4 r = red[k * LQ + q0]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
        partial[(int64_t)blockIdx.x * CQ + q] = t;
      }
      __syncthreads();
    }
  }
}

// a workgroup = 16 channels x 16 block lanes (lane j adds blocks j, j + 16, ... in index order, then the 16 lane sums in
// lane order: deterministic; one workgroup of 4 lanes per channel walked up to 1024 partials per lane in a row)
__global__ __launch_bounds__(256) void sr_bias_partial_reduce_kernel(const float* __restrict__ partial, int blocks, int C,
                                                                    float* __restrict__ db) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, j = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float s = 0.0f;
  if (c < C)
    for (int b = j; b < blocks; b += 16) s += partial[(int64_t)b * C + c];
  red[j][cl] = s;
  __syncthreads();
  if (j == 0 && c < C) {
    float t = red[0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][cl];
    db[c] = t;
  }
}

static int sr_act_bwd_bias_blocks(int64_t pixels, int C) {
  const int CQ = C / 4, LQ = CQ < 256 ? CQ : 256, PB = 256 / LQ;
  int64_t blocks = (pixels + (int64_t)PB * 16 - 1) / ((int64_t)PB * 16);   // >= 16 pixels (two trips) per lane
  if (blocks > SR_AB_MAX_BLOCKS) blocks = SR_AB_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

extern "C" size_t sr_act_bwd_bias_workspace_bytes(int64_t pixels, int C) {
  if (pixels <= 0 || C <= 0 || C % 4 != 0) return 0;
  return (size_t)sr_act_bwd_bias_blocks(pixels, C) * C * sizeof(float);
}

extern "C" int sr_act_bwd_bias_nhwc(const float* grad, const float* out_saved, float* grad_pre, float* d_bias,
                                    int64_t pixels, int C, float leaky_slope, void* workspace, size_t workspace_bytes,
                                    void* stream_) {
  if (pixels < 0 || C <= 0 || (out_saved && leaky_slope < 0.0f)) return SR_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  if (pixels == 0) {
    if (d_bias) { const hipError_t e = hipMemsetAsync(d_bias, 0, (size_t)C * sizeof(float), stream); if (e != hipSuccess) return sr_hip_rc(e); }
    return SR_OK;
  }
  if (!grad || (out_saved && !grad_pre)) return SR_ERR_INVALID_ARGUMENT;
  if (!out_saved && !d_bias) return SR_OK;
  if (C % 4 != 0 || ((uintptr_t)grad & 15) || ((uintptr_t)out_saved & 15) || ((uintptr_t)grad_pre & 15)) return SR_ERR_UNSUPPORTED;
  const int blocks = sr_act_bwd_bias_blocks(pixels, C);
  if (d_bias && (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < (size_t)blocks * C * sizeof(float)))
    return SR_ERR_WORKSPACE_TOO_SMALL;
  const int CQ = C / 4, LQ = CQ < 256 ? CQ : 256, PB = 256 / LQ;
  hipLaunchKernelGGL(sr_act_bwd_bias_kernel, dim3(blocks), dim3(256), 0, stream, (const float4*)grad,
                     (const float4*)out_saved, (float4*)grad_pre, d_bias ? (float4*)workspace : (float4*)nullptr, pixels, CQ,
                     LQ, PB, leaky_slope);
  if (d_bias)
    hipLaunchKernelGGL(sr_bias_partial_reduce_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, (const float*)workspace,
                       blocks, C, d_bias);
  return sr_hip_rc(hipGetLastError());
}

// out[b, 2y, 2x, :] = in[b, y, x, :], zeros elsewhere (out is [B, Hs, Ws, C] with Hs >= 2*H-1, Ws >= 2*W-1): the
// zero-stuffed output gradient on which a stride-1 convolution computes the data gradient of a stride-2 one.
__global__ __launch_bounds__(256) void sr_zero_stuff2x_kernel(const float* __restrict__ in, int64_t in_sb, int in_sp,
                                                             float* __restrict__ out, int H, int W, int Hs, int Ws, int C) {
  const int b = blockIdx.y;
  const int64_t total = (int64_t)Hs * Ws * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t px = e / C;
    const int oy = (int)(px / Ws), ox = (int)(px - (int64_t)oy * Ws);
    float v = 0.0f;
    if (!(oy & 1) && !(ox & 1) && (oy >> 1) < H && (ox >> 1) < W)
      v = in[(int64_t)b * in_sb + ((int64_t)(oy >> 1) * W + (ox >> 1)) * in_sp + c];
    out[((int64_t)b * Hs * Ws + px) * C + c] = v;
  }
}

extern "C" int sr_zero_stuff2x_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out, int B, int H,
                                    int W, int Hs, int Ws, int C, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || Hs < 2 * H - 1 || Ws < 2 * W - 1) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !out) return SR_ERR_INVALID_ARGUMENT;
  const long total = (long)Hs * Ws * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sr_zero_stuff2x_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream_, in, in_batch_stride,
                     in_pix_stride, out, H, W, Hs, Ws, C);
  return sr_hip_rc(hipGetLastError());
}

// Adjoint of sr_upsample2x_nhwc_fwd (bilinear x2, align_corners=False, reference generic_utils.py:96-105): every
// input pixel gathers from the <= 4 x 4 output pixels whose (clamped) taps include it, with the forward's weights.
__device__ __forceinline__ float sr_up_weight(int o, int src, int n) {
  // weight of input index `src` in the interpolation of output index `o` along an axis of input length n
  const float s = fmaxf(((float)o + 0.5f) * 0.5f - 0.5f, 0.0f);
  const int i0 = (int)s, i1 = i0 + (i0 < n - 1 ? 1 : 0);
  const float l = s - (float)i0;
  return (src == i0 ? 1.0f - l : 0.0f) + (src == i1 ? l : 0.0f);
}

__global__ __launch_bounds__(256) void sr_upsample2x_bwd_kernel(const float* __restrict__ g, int64_t g_sb, int g_sp,
                                                               float* __restrict__ out, int64_t out_sb, int out_sp, int H,
                                                               int W, int C) {
  const int b = blockIdx.y;
  const int64_t total = (int64_t)H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const int64_t px = e / C;
    const int y = (int)(px / W), x = (int)(px - (int64_t)y * W);
    float s = 0.0f;
    for (int oy = max(2 * y - 1, 0); oy <= min(2 * y + 2, 2 * H - 1); ++oy) {
      const float wy = sr_up_weight(oy, y, H);
      if (wy == 0.0f) continue;
      float r = 0.0f;
      for (int ox = max(2 * x - 1, 0); ox <= min(2 * x + 2, 2 * W - 1); ++ox) {
        const float wx = sr_up_weight(ox, x, W);
        if (wx != 0.0f) r += wx * g[(int64_t)b * g_sb + ((int64_t)oy * 2 * W + ox) * g_sp + c];
      }
      s += wy * r;
    }
    out[(int64_t)b * out_sb + px * out_sp + c] = s;
  }
}

extern "C" int sr_upsample2x_bwd_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, float* grad_in,
                                      int64_t in_batch_stride, int in_pix_stride, int B, int H, int W, int C,
                                      void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!grad_out || !grad_in) return SR_ERR_INVALID_ARGUMENT;
  const long total = (long)H * W * C;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sr_upsample2x_bwd_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream_, grad_out,
                     g_batch_stride, g_pix_stride, grad_in, in_batch_stride, in_pix_stride, H, W, C);
  return sr_hip_rc(hipGetLastError());
}

// W'[ci, co, ky, kx] = W[co, ci, k-1-ky, k-1-kx]: the weight of the convolution that computes the data gradient.
__global__ __launch_bounds__(256) void sr_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ out, int Co,
                                                               int Ci, int KK, int k) {
  const int64_t total = (int64_t)Co * Ci * KK;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(e % KK);
    const int64_t r = e / KK;
    const int co = (int)(r % Co), ci = (int)(r / Co);   // e indexes out[ci][co][t]
    const int ky = t / k, kx = t - ky * k;
    out[e] = w[((int64_t)co * Ci + ci) * KK + (k - 1 - ky) * k + (k - 1 - kx)];
  }
}

extern "C" int sr_conv_flip_transpose_weights(const float* weight, int Cout, int Cin, int ksize, float* out, void* stream_) {
  if (!weight || !out || Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return SR_ERR_INVALID_ARGUMENT;
  const long total = (long)Cout * Cin * ksize * ksize;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sr_flip_transpose_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, weight, out, Cout, Cin,
                     ksize * ksize, ksize);
  return sr_hip_rc(hipGetLastError());
}

// out = a * b (dense arrays): backward of depth = exp(log_depth) (grad * depth), reference depth_model.py:392-400
__global__ __launch_bounds__(256) void sr_mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

extern "C" int sr_mul_fwd(const float* a, const float* b, float* out, int64_t n, void* stream_) {
  if (n < 0) return SR_ERR_INVALID_ARGUMENT;
  if (n == 0) return SR_OK;
  if (!a || !b || !out) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a, b, out, n);
  return sr_hip_rc(hipGetLastError());
}
