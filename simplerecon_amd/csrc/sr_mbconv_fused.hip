// sr_mbconv_fused.hip -- the front half of an MBConv block in ONE launch (r05).
//
// EfficientNetV2-S stages 3-5 (reference experiment_modules/depth_model.py:110-116: timm tf_efficientnetv2_s)
// are 30 inverted-residual blocks; r04 ran each as five launches (1x1 expansion GEMM, depthwise 3x3, squeeze-excite hidden
// layer, squeeze-excite gates, 1x1 projection GEMM).  The maps are small (30x40 / 15x20 x batch 8), so every launch is a
// 5-50 us kernel that pays its own ramp, tail and dependent-launch gap, and the expanded map makes an HBM / L2 round trip
// between the expansion and the depthwise conv.  Here:
//   sr_mbconv_expand_dw_se_fwd = expansion (1x1 conv + BatchNorm + SiLU) -> depthwise 3x3 (+ BatchNorm + SiLU) -> squeeze-excite
//                                average pool -> squeeze-excite MLP (gates), stride-1 blocks;
//   the projection stays sr_pw_conv_nhwc_fwd with the gates applied to its A operand: 2 launches per block.
// A workgroup owns (image, 16 expanded channels) over the WHOLE map: the depthwise conv needs no halo exchange and the
// channel's average pool is complete inside the workgroup (deterministic, no atomics on data).
//   E  expansion as v_mfma_f32_16x16x4_f32 with M = channels, N = 16-pixel tiles: the 16 x Cin weight slice lives in registers
//      for the whole workgroup (one float4 per lane and 16 input channels), the activations stream from L2 (one float4 per
//      lane and 16 channels, the next tile's loads in flight under the current tile's MFMAs); the SiLU'd tile lands in LDS
//      as [pixel][16] (a lane holds 4 consecutive channels of one pixel: ds_write_b128);
//   D  depthwise 3x3 from LDS: a thread takes (pixel, channel quad), 9 ds_read_b128 + 36 FMA, SiLU, 16-byte store, and
//      accumulates its part of the channel sums; a fixed-order tree over the 64 threads of a quad finishes the pool;
//   S  the LAST workgroup of an image to arrive (one device-scope counter per image; the channel sums travel as device-scope
//      stores / loads, guides/cdna_hip_programming.md G16's fence-free form) computes hidden = silu(W1 mean + b1), gate = sigmoid(W2 hidden + b2)
//      for that image and resets the counter -- no launch of its own.
#include "sr_common.h"

// timing ablations (results wrong): -DSR_MX_ABL=<bits>  1: no output store, 2: no depthwise phase, 4: no expansion MFMAs / loads,
// 8: no squeeze-excite tail.  0 in the product build.
#ifndef SR_MX_ABL
#define SR_MX_ABL 0
#endif

namespace {

typedef float mx_f4 __attribute__((ext_vector_type(4)));
#define MX_RSRC_FLAGS 0x00020000
#define MX_OOB 0x7fffffffu

struct SrMbxParams {
  const float* in; int64_t in_sb; int in_sp;
  const float* w_exp; const float* b_exp;     // [mid][Cin], [mid] (BatchNorm folded)
  const float* w_dw; const float* b_dw;       // [9][mid], [mid]
  float* out; int64_t out_sb; int out_sp;
  float* pool;                                // [B][mid] channel sums of the activated depthwise output
  const float* w1; const float* b1;           // [rd][mid], [rd]
  const float* w2; const float* b2;           // [mid][rd], [mid]
  float* gate;                                // [B][mid]
  unsigned* counter;                          // [B], zero at entry, zero again at exit
  int B, H, W, Cin, mid, rd, slices;
};

__device__ __forceinline__ float mx_silu(float v) { return v / (1.0f + __expf(-v)); }

template <int KJ>   // KJ = Cin / 16
__global__ __launch_bounds__(256, 2) void sr_mbconv_expand_dw_se_kernel(SrMbxParams p) {
  extern __shared__ __attribute__((aligned(16))) float E[];   // [HW][16]; later [mid] means + [rd] hidden (last workgroup)
  __shared__ mx_f4 red[256];
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (image, slice): consecutive workgroups go to consecutive XCDs (blockIdx % 8), so with b = blockIdx % B the
  // workgroups of one image share ONE XCD's L2 at batch 8 -- each L2 then holds its image's input map plus the weights instead of
  // every image's map (which, with the depthwise output streaming through, no longer fit: A loads went to HBM latency)
  const int b = blockIdx.x % p.B, slice = blockIdx.x / p.B;
  const int c0 = slice * 16;
  const int HW = p.H * p.W;
  const int m_i = lane & 15, m_kq = lane >> 4;

  // ---- E: expansion.  A = weights (registers), B = activations of a 16-pixel tile
  mx_f4 wreg[KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j)
    wreg[j] = *reinterpret_cast<const mx_f4*>(p.w_exp + (int64_t)(c0 + m_i) * p.Cin + 16 * j + 4 * m_kq);
  mx_f4 be = mx_f4{0.0f, 0.0f, 0.0f, 0.0f};
  if (p.b_exp) be = *reinterpret_cast<const mx_f4*>(p.b_exp + c0 + 4 * m_kq);
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in + (int64_t)b * p.in_sb), 0, (int)(((int64_t)(HW - 1) * p.in_sp + p.Cin) * 4), MX_RSRC_FLAGS);
  const int tiles = (HW + 15) / 16;
  auto load_tile = [&](int mt, mx_f4 (&xa)[KJ]) {
    const int px = mt * 16 + m_i;
    const unsigned voff = (mt < tiles && px < HW) ? (unsigned)(px * p.in_sp + 4 * m_kq) * 4u : MX_OOB;
#pragma unroll
    for (int j = 0; j < KJ; ++j)
      xa[j] = __builtin_bit_cast(mx_f4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)voff, 64 * j, 0));
  };
  auto compute_tile = [&](int mt, const mx_f4 (&xa)[KJ]) {
    mx_f4 acc = mx_f4{0.0f, 0.0f, 0.0f, 0.0f}, acc2 = mx_f4{0.0f, 0.0f, 0.0f, 0.0f};   // two chains: dependent MFMAs 64 clk apart
#pragma unroll
    for (int j = 0; j < KJ; j += 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][e], xa[j][e], acc, 0, 0, 0);
        if (j + 1 < KJ) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j + 1][e], xa[j + 1][e], acc2, 0, 0, 0);
      }
    }
    const int px = mt * 16 + m_i;
    if (px < HW) {
      mx_f4 v = acc + acc2 + be;
      *reinterpret_cast<mx_f4*>(E + px * 16 + 4 * m_kq) = mx_f4{mx_silu(v[0]), mx_silu(v[1]), mx_silu(v[2]), mx_silu(v[3])};
    }
  };
  {
    mx_f4 xa[KJ], xb[KJ];
    int mt = (SR_MX_ABL & 4) ? tiles : wave;
    load_tile(mt, xa);
    for (; mt < tiles; mt += 8) {
      load_tile(mt + 4, xb);
      compute_tile(mt, xa);
      if (mt + 4 < tiles) {
        load_tile(mt + 8, xa);
        compute_tile(mt + 4, xb);
      }
    }
  }
  __syncthreads();

  // ---- D: depthwise 3x3 (pad 1) + SiLU, store, pool partials.  Thread = (pixel, channel quad q).
  const int q = tid & 3;
  mx_f4 wt[9], bd = mx_f4{0.0f, 0.0f, 0.0f, 0.0f}, sum = mx_f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const mx_f4*>(p.w_dw + (int64_t)t * p.mid + c0 + 4 * q);
  if (p.b_dw) bd = *reinterpret_cast<const mx_f4*>(p.b_dw + c0 + 4 * q);
  float* outb = p.out + (int64_t)b * p.out_sb + c0 + 4 * q;
  for (int px = (SR_MX_ABL & 2) ? HW : (tid >> 2); px < HW; px += 64) {
    const int y = px / p.W, x = px - y * p.W;
    mx_f4 acc = bd;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        const bool ok = (yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W);
        const mx_f4 v = *reinterpret_cast<const mx_f4*>(E + (ok ? (yy * p.W + xx) : px) * 16 + 4 * q);
        acc = acc + (ok ? v : mx_f4{0.0f, 0.0f, 0.0f, 0.0f}) * wt[ky * 3 + kx];
      }
    }
    const mx_f4 o = mx_f4{mx_silu(acc[0]), mx_silu(acc[1]), mx_silu(acc[2]), mx_silu(acc[3])};
    if (!(SR_MX_ABL & 1) || o[0] == 1.2345e33f) *reinterpret_cast<mx_f4*>(outb + (int64_t)px * p.out_sp) = o;
    sum = sum + o;
  }
  // fixed-order tree over the 64 threads of a quad: run-to-run deterministic
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s >= 4; s >>= 1) {
    if (tid < s) red[tid] = red[tid] + red[tid + s];
    __syncthreads();
  }
  // The 16 channel sums are the ONLY data another workgroup reads (the depthwise output goes to the next launch): publish them
  // with device-scope (write-through) stores, wait for their acknowledgement, then arrive on the image's counter; the last
  // workgroup reads every sum with device-scope loads (L1 bypassed).  No release / acquire FENCES: a fence writes back the
  // whole XCD L2's dirty lines -- 77 KB of depthwise output per workgroup here -- and 480 workgroups doing that one after the
  // other cost ~90 us of a 114-us launch (guides/MI355X_MICROARCH.md, price list: "publish-large").
  if (tid < 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      __hip_atomic_store(p.pool + (int64_t)b * p.mid + c0 + 4 * tid + e, red[tid][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) is_last = (atomicAdd(p.counter + b, 1u) == (unsigned)(p.slices - 1)) ? 1 : 0;
  __syncthreads();
  if (!is_last || (SR_MX_ABL & 8)) { if (is_last && tid == 0) p.counter[b] = 0u; return; }
  float* mean = E;             // [mid]
  float* hid = E + p.mid;      // [rd]
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < p.mid; c += 256)
    mean[c] = __hip_atomic_load(p.pool + (int64_t)b * p.mid + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * inv;
  __syncthreads();
  {   // hidden units: 4 threads per unit, each a quarter of the channels through 8 independent float4 loads at a time (the
      // workgroup is alone with two latency-bound matrix-vector products: everything is about loads in flight)
    const int j = tid >> 2, part = tid & 3;
    float s = 0.0f;
    if (j < p.rd) {
      const int per = ((p.mid / 4 + 3) / 4) * 4;            // channels per part, whole float4s
      const int cbeg = part * per, cend = min(p.mid, cbeg + per);
      const float* w = p.w1 + (int64_t)j * p.mid;
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int cc = cbeg; cc < cend; cc += 32) {
        mx_f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (cc + 4 * u < cend) ? *reinterpret_cast<const mx_f4*>(w + cc + 4 * u) : mx_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (cc + 4 * u < cend) {
            const mx_f4 m = *reinterpret_cast<const mx_f4*>(mean + cc + 4 * u);
            a8[u] = fmaf(v[u][0], m[0], fmaf(v[u][1], m[1], fmaf(v[u][2], m[2], fmaf(v[u][3], m[3], a8[u]))));
          }
        }
      }
      s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (j < p.rd && part == 0) hid[j] = mx_silu(s + (p.b1 ? p.b1[j] : 0.0f));
  }
  __syncthreads();
  // gates: thread = (channel, quarter of the hidden units), 4 adjacent lanes per channel; every weight load of a pass is issued
  // before the first is used (the r05 first cut walked a channel's rd weights 16 at a time: 24 dependent round trips to L2)
  for (int cb = 0; cb < p.mid; cb += 64) {
    const int c = cb + (tid >> 2), part = tid & 3;
    const int per = ((p.rd + 3) / 4 + 3) / 4 * 4;             // hidden units per part, whole float4s when rd % 4 == 0
    const int jb = part * per, je = min(p.rd, jb + per);
    float s = 0.0f;
    if (c < p.mid) {
      const float* w = p.w2 + (int64_t)c * p.rd;
      if ((p.rd & 3) == 0) {
        mx_f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (jb + 4 * u < je) ? *reinterpret_cast<const mx_f4*>(w + jb + 4 * u) : mx_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (jb + 4 * u < je)
            s = fmaf(v[u][0], hid[jb + 4 * u], fmaf(v[u][1], hid[jb + 4 * u + 1],
                     fmaf(v[u][2], hid[jb + 4 * u + 2], fmaf(v[u][3], hid[jb + 4 * u + 3], s))));
      } else {
        for (int j = jb; j < je; ++j) s = fmaf(w[j], hid[j], s);
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (c < p.mid && part == 0) p.gate[(int64_t)b * p.mid + c] = 1.0f / (1.0f + __expf(-(s + (p.b2 ? p.b2[c] : 0.0f))));
  }
  if (tid == 0) p.counter[b] = 0u;
}

inline bool mx_al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

}  // namespace

// 1 when sr_mbconv_expand_dw_se_fwd serves the block: stride 1, Cin in {128, 160, 256}, expanded channels in whole groups of
// 16, the map x 16 channels inside the LDS budget.
extern "C" int sr_mbconv_fused_supported(int H, int W, int Cin, int mid, int rd) {
  if (H <= 0 || W <= 0 || (Cin != 128 && Cin != 160 && Cin != 256) || mid % 16 != 0 || rd <= 0 || rd > 64) return 0;
  const size_t lds = (size_t)H * W * 16 * sizeof(float);
  if (lds > 78 * 1024 || (size_t)(mid + rd) * sizeof(float) > lds) return 0;
  return 1;
}

// expansion (1x1, BatchNorm folded, SiLU) -> depthwise 3x3 / stride 1 / pad 1 (BatchNorm folded, SiLU) -> squeeze-excite gates
// of an MBConv block.  `w_expand` [mid][Cin], `w_dw9c` [9][mid] tap-major; `pool` [B][mid] receives the channel sums, `gate`
// [B][mid] the gates; `counters` = B zeroed 32-bit words (left zeroed).  Deterministic.
extern "C" int sr_mbconv_expand_dw_se_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* w_expand,
                                          const float* b_expand, const float* w_dw9c, const float* b_dw, const float* w_reduce,
                                          const float* b_reduce, const float* w_excite, const float* b_excite, float* out,
                                          int64_t out_batch_stride, int out_pix_stride, float* pool, float* gate,
                                          unsigned* counters, int B, int H, int W, int Cin, int mid, int rd, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || mid <= 0 || rd <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !w_expand || !w_dw9c || !w_reduce || !w_excite || !out || !pool || !gate || !counters) return SR_ERR_INVALID_ARGUMENT;
  if (!sr_mbconv_fused_supported(H, W, Cin, mid, rd)) return SR_ERR_UNSUPPORTED;
  if (!mx_al16(in) || in_pix_stride % 4 != 0 || in_batch_stride % 4 != 0 || !mx_al16(out) || out_pix_stride % 4 != 0 ||
      out_batch_stride % 4 != 0 || !mx_al16(w_expand) || !mx_al16(w_dw9c) || (b_expand && !mx_al16(b_expand)) ||
      (b_dw && !mx_al16(b_dw)) || !mx_al16(pool))
    return SR_ERR_UNSUPPORTED;
  if (((int64_t)((int64_t)H * W - 1) * in_pix_stride + Cin) * 4 >= ((int64_t)1 << 31)) return SR_ERR_UNSUPPORTED;
  SrMbxParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.w_exp = w_expand; p.b_exp = b_expand; p.w_dw = w_dw9c; p.b_dw = b_dw;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.pool = pool; p.w1 = w_reduce; p.b1 = b_reduce; p.w2 = w_excite; p.b2 = b_excite; p.gate = gate; p.counter = counters;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.mid = mid; p.rd = rd; p.slices = mid / 16;
  const size_t lds = (size_t)H * W * 16 * sizeof(float);
  const dim3 grid((unsigned)(B * p.slices));
  hipStream_t stream = (hipStream_t)stream_;
#define MX_LAUNCH(KJV)                                                                                                      \
  {                                                                                                                         \
    hipError_t e = hipFuncSetAttribute((const void*)sr_mbconv_expand_dw_se_kernel<KJV>,                                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    if (e != hipSuccess) return sr_hip_rc(e);                                                                               \
    hipLaunchKernelGGL((sr_mbconv_expand_dw_se_kernel<KJV>), grid, dim3(256), lds, stream, p);                              \
  }
  if (Cin == 128) MX_LAUNCH(8)
  else if (Cin == 160) MX_LAUNCH(10)
  else MX_LAUNCH(16)
#undef MX_LAUNCH
  return sr_hip_rc(hipGetLastError());
}
