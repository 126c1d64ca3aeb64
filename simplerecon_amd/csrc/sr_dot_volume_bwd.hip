// sr_dot_volume_bwd.hip -- backward of the dot-product plane sweep for gfx950 (first piece of SURVEY.md §8f "next" #3).
//
// Forward (reference modules/cost_volume.py:237-335, sr_dot_volume.hip):
//   cost[b,j,y,x] = sum_k m_k * sum_c cur[b,c,y,x] * sum_{4 taps t} w_t * src[b,k,c,tap_t]
// with tap positions / weights / masks that depend on the geometry only (poses, intrinsics and depth planes are data:
// the reference's autograd does not differentiate them either).  The volume is bilinear in (cur, src), so with
// g = dL/dcost:
//   d_cur[b,c,y,x]    = sum_j g * sum_k m_k * sum_t w_t * src[b,k,c,tap_t]            (a gather: same reads as forward)
//   d_src[b,k,c,tap] += g * m_k * w_t * cur[b,c,y,x]   over all (j, y, x, t) hitting the texel (a scatter)
// which is what ATen's grid_sampler_2d_backward + the broadcasting mul / sum backward compute in the reference.
//
// One lane per reference pixel; the lane walks all planes and views, recomputing the projection exactly as the forward
// kernel does (sr_project_sample, bit-identical taps), accumulates d_cur in registers and scatters into a channels-last
// d_src image ([B*K, h*w, C], zero-filled by the caller side of this file) with hardware fp32 atomics
// (global_atomic_add_f32; summation order is therefore not fixed, exactly like torch's grid_sample backward).
// A second kernel transposes d_src to the reference layout [B,K,C,h,w].
// Work: 4 taps x C atomics per (pixel, plane, view) -- atomic-throughput-bound; not tuned yet (round 1 groundwork).
#include "sr_common.h"

namespace {

struct SrDotBwdParams {
  const float* grad_cv; int64_t g_sb, g_sd, g_sp;
  const float* cur;        // [B,C,h,w]
  const float* src_nhwc;   // [B*K, h*w, C] (workspace of sr_volume_prepare)
  const float* invK;       // [B,16]
  const float* geom;       // [B*K, SR_GEOM_STRIDE]
  SrPlanes planes;
  float* d_cur;            // [B,C,h,w] or null
  float* d_src_nhwc;       // [B*K, h*w, C] zero-initialised, or null
  int B, K, h, w, D;
  float inv_w, inv_h;
};

template <int C>
__global__ __launch_bounds__(256) void sr_dot_volume_bwd_kernel(SrDotBwdParams p) {
  // wave-private LDS: tap records of the current (plane, view) [64 pixels][4 taps] (texel, weight * g * mask) and the
  // wave's reference features [64][C] -- for the scatter with lane = channel (see below)
  __shared__ int s_off[4][64 * 4];
  __shared__ float s_wt[4][64 * 4];
  __shared__ float s_cur[4][64 * C];
  const int b = blockIdx.y;
  const int N = p.h * p.w;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pix_raw = blockIdx.x * 256 + threadIdx.x;
  const bool active = pix_raw < N;
  const int pix = active ? pix_raw : N - 1;
  const int y = pix / p.w, x = pix - y * p.w;

  float cur[C], dcur[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    cur[c] = p.cur[((size_t)b * C + c) * N + pix];
    dcur[c] = 0.0f;
    s_cur[wave][lane * C + c] = cur[c];
  }
  float r0, r1, r2;
  {
#pragma clang fp contract(off)
    const float* iK = p.invK + 16 * (size_t)b;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;  // geometry_utils.py:34-44
    r0 = iK[0] * px + iK[1] * py + iK[2];
    r1 = iK[4] * px + iK[5] * py + iK[6];
    r2 = iK[8] * px + iK[9] * py + iK[10];
  }
  const float* geom_b = p.geom + (size_t)b * p.K * SR_GEOM_STRIDE;
  const float* src_b = p.src_nhwc + (size_t)b * p.K * N * C;
  float* dsrc_b = p.d_src_nhwc ? p.d_src_nhwc + (size_t)b * p.K * N * C : nullptr;
  const float* planes = p.planes.ptr + b * p.planes.sb + y * p.planes.sy + x * p.planes.sx;
  const float* gcv = p.grad_cv + b * p.g_sb + (int64_t)pix * p.g_sp;
  constexpr int UNITS_PER_INSTR = 64 / C;          // (pixel, tap) units one wave instruction scatters
  const int sc = lane % C, su = lane / C;

  for (int j = 0; j < p.D; ++j) {
    const float d = planes[j * p.planes.sd];
    const float g = active ? gcv[j * p.g_sd] : 0.0f;
    float X0, X1, X2;
    {
#pragma clang fp contract(off)
      X0 = d * r0; X1 = d * r1; X2 = d * r2;  // geometry_utils.py:56-57
    }
#pragma unroll 1
    for (int k = 0; k < p.K; ++k) {
      SrSample s;
      sr_project_sample(geom_b + k * SR_GEOM_STRIDE, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, s);
      const float gm = (s.zp > 0.0f) ? g : 0.0f;   // mask_k (cost_volume.py:231-232)
      const float* img = src_b + (size_t)k * N * C;
      const float wt[4] = {s.w_nw * gm, s.w_ne * gm, s.w_sw * gm, s.w_se * gm};
      const int ot[4] = {s.o_nw, s.o_ne, s.o_sw, s.o_se};
      if (p.d_cur) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (wt[t] == 0.0f) continue;   // out-of-image taps (zero padding), masked views and zero gradients
          const float4* tp = reinterpret_cast<const float4*>(img + (size_t)ot[t] * C);
#pragma unroll
          for (int q = 0; q < C / 4; ++q) {
            const float4 v = tp[q];
            dcur[4 * q + 0] = fmaf(wt[t], v.x, dcur[4 * q + 0]);
            dcur[4 * q + 1] = fmaf(wt[t], v.y, dcur[4 * q + 1]);
            dcur[4 * q + 2] = fmaf(wt[t], v.z, dcur[4 * q + 2]);
            dcur[4 * q + 3] = fmaf(wt[t], v.w, dcur[4 * q + 3]);
          }
        }
      }
      if (dsrc_b) {
        // scatter d_src[tap] += wt * cur with lane = channel: the C lanes of a (pixel, tap) unit add to the contiguous
        // 4*C bytes of one texel, so a wave instruction touches 64 / C texels instead of 64 (the lane = pixel form did;
        // the scatter's atomics are what bounds this kernel).  Tap records travel through wave-private LDS.
        if (__ballot((wt[0] != 0.0f) | (wt[1] != 0.0f) | (wt[2] != 0.0f) | (wt[3] != 0.0f)) == 0) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s_off[wave][lane * 4 + t] = ot[t];
          s_wt[wave][lane * 4 + t] = wt[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* dimg = dsrc_b + (size_t)k * N * C;
#pragma unroll 4
        for (int unit0 = 0; unit0 < 256; unit0 += UNITS_PER_INSTR) {
          const int unit = unit0 + su;                         // (pixel, tap) = (unit >> 2, unit & 3)
          if (su < UNITS_PER_INSTR && unit < 256) {            // (C = 12, 24: the last lanes of a wave sit out)
            const float wq = s_wt[wave][unit];
            if (wq != 0.0f)
              unsafeAtomicAdd(dimg + (size_t)s_off[wave][unit] * C + sc, wq * s_cur[wave][(unit >> 2) * C + sc]);
          }
        }
        __builtin_amdgcn_wave_barrier();   // the records are overwritten by the next (plane, view)
      }
    }
  }
  if (p.d_cur && active) {
#pragma unroll
    for (int c = 0; c < C; ++c) p.d_cur[((size_t)b * C + c) * N + pix] = dcur[c];
  }
}

// [images][N][C] -> [images][C][N]
__global__ __launch_bounds__(256) void sr_unpack_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int C, int N) {
  const int img = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= N) return;
  const float* s = src + ((size_t)img * N + pix) * C;
  float* d = dst + (size_t)img * C * N + pix;
  for (int c = 0; c < C; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(s + c);
    d[(size_t)(c + 0) * N] = v.x;
    d[(size_t)(c + 1) * N] = v.y;
    d[(size_t)(c + 2) * N] = v.z;
    d[(size_t)(c + 3) * N] = v.w;
  }
}

}  // namespace

int sr_launch_unpack_nhwc(const float* src_nhwc, float* dst_nchw, int images, int C, int npix, hipStream_t stream) {
  if (images == 0 || npix == 0) return SR_OK;
  hipLaunchKernelGGL(sr_unpack_nhwc_kernel, dim3((npix + 255) / 256, images), dim3(256), 0, stream, src_nhwc, dst_nchw,
                     C, npix);
  return sr_hip_rc(hipGetLastError());
}

extern "C" size_t sr_dot_volume_bwd_scratch_bytes(int B, int K, int C, int h, int w) {
  if (B < 0 || K < 0 || C < 0 || h < 0 || w < 0) return 0;
  return (size_t)B * K * h * w * C * sizeof(float);
}

extern "C" int sr_dot_volume_bwd(const float* grad_cv, int64_t g_sb, int64_t g_sd, int64_t g_sp, const float* cur,
                                 const float* invK_cur, const float* planes, int64_t ps_b, int64_t ps_d, int64_t ps_y,
                                 int64_t ps_x, int B, int K, int C, int h, int w, int D, float* d_cur, float* d_src,
                                 void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                 void* stream_) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!grad_cv || !cur || !invK_cur || !planes || !workspace || (!d_cur && !d_src)) return SR_ERR_INVALID_ARGUMENT;
  if (C % 4 != 0 || C > 32) return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_volume_workspace_bytes(B, K, C, h, w)) return SR_ERR_WORKSPACE_TOO_SMALL;
  if (d_src && (!scratch || scratch_bytes < sr_dot_volume_bwd_scratch_bytes(B, K, C, h, w)))
    return SR_ERR_WORKSPACE_TOO_SMALL;
  if (d_src && (((uintptr_t)scratch) & 15)) return SR_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = h * w;
  SrDotBwdParams p;
  p.grad_cv = grad_cv; p.g_sb = g_sb; p.g_sd = g_sd; p.g_sp = g_sp;
  p.cur = cur; p.src_nhwc = sr_ws_src_nhwc(workspace, B, K); p.invK = invK_cur; p.geom = sr_ws_geom(workspace);
  p.planes = {planes, ps_b, ps_d, ps_y, ps_x};
  p.d_cur = d_cur; p.d_src_nhwc = d_src ? (float*)scratch : nullptr;
  p.B = B; p.K = K; p.h = h; p.w = w; p.D = D;
  p.inv_w = 1.0f / (float)w; p.inv_h = 1.0f / (float)h;
  if (d_src) {
    hipError_t e = hipMemsetAsync(scratch, 0, sr_dot_volume_bwd_scratch_bytes(B, K, C, h, w), stream);
    if (e != hipSuccess) return sr_hip_rc(e);
  }
  dim3 grid((N + 255) / 256, B), block(256);
  switch (C) {
    case 4: hipLaunchKernelGGL(sr_dot_volume_bwd_kernel<4>, grid, block, 0, stream, p); break;
    case 8: hipLaunchKernelGGL(sr_dot_volume_bwd_kernel<8>, grid, block, 0, stream, p); break;
    case 12: hipLaunchKernelGGL(sr_dot_volume_bwd_kernel<12>, grid, block, 0, stream, p); break;
    case 16: hipLaunchKernelGGL(sr_dot_volume_bwd_kernel<16>, grid, block, 0, stream, p); break;
    case 24: hipLaunchKernelGGL(sr_dot_volume_bwd_kernel<24>, grid, block, 0, stream, p); break;
    case 32: hipLaunchKernelGGL(sr_dot_volume_bwd_kernel<32>, grid, block, 0, stream, p); break;
    default: return SR_ERR_UNSUPPORTED;
  }
  int rc = sr_hip_rc(hipGetLastError());
  if (rc != SR_OK || !d_src) return rc;
  return sr_launch_unpack_nhwc((const float*)scratch, d_src, B * K, C, N, stream);
}
