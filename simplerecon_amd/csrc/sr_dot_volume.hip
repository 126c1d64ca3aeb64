// sr_dot_volume.hip -- fused plane sweep of the dot-product cost volume for gfx950.
//
// Replaces CostVolumeManager.build_cost_volume/forward of the reference
// (modules/cost_volume.py:237-380): per depth plane ~12 ATen launches and a
// [B,K,C,h,w] warped-feature tensor written+read per plane.  Here: one launch per batch;
// a wavefront owns 64 consecutive pixels; the source features live channels-last in HBM
// ([B*K, h*w, C], one bilinear tap = one contiguous 4*C-byte read, neighbouring lanes hit
// neighbouring texels); the reference feature vector stays in VGPRs over all planes; the
// over-views reduction and the running argmax stay in registers; planes are split over the
// waves of a workgroup (more bytes in flight for the L1/TA-bound gather) and merged in LDS.
#include <stdlib.h>

#include "sr_common.h"

// ------------------------------------------------------------------------ prologue ----

// One thread per (b,k).  P = K @ T (rows 0..2), source-camera centre and the DVMVS pose
// measures (geometry_utils.py:178-191).  Op-by-op, contraction off (see sr_common.h).
__global__ void sr_geom_kernel(const float* __restrict__ K_src, const float* __restrict__ T_src_cur,
                               const float* __restrict__ T_cur_src, float* __restrict__ geom, int n) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* Km = K_src + 16 * (size_t)i;
  const float* T = T_src_cur + 16 * (size_t)i;
  float* g = geom + SR_GEOM_STRIDE * (size_t)i;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += Km[r * 4 + k] * T[k * 4 + c];
      g[r * 4 + c] = s;
    }
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, dist = 0.f, rm = 0.f, tm = 0.f;
  if (T_cur_src) {
    const float* Tc = T_cur_src + 16 * (size_t)i;
    t0 = Tc[3]; t1 = Tc[7]; t2 = Tc[11];
    const float tr = (Tc[0] + Tc[5]) + Tc[10];
    rm = sqrtf(2.0f * (1.0f - fminf(3.0f, tr) / 3.0f));
    tm = sqrtf((t0 * t0 + t1 * t1) + t2 * t2);
    dist = sqrtf(tm * tm + rm * rm);
  }
  g[12] = t0; g[13] = t1; g[14] = t2; g[15] = dist; g[16] = rm; g[17] = tm; g[18] = 0.f; g[19] = 0.f;
}

int sr_launch_geom(const float* K_src, const float* T_src_cur, const float* T_cur_src, float* geom,
                   int n, hipStream_t stream) {
  if (n == 0) return SR_OK;
  hipLaunchKernelGGL(sr_geom_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, K_src, T_src_cur, T_cur_src,
                     geom, n);
  return sr_hip_rc(hipGetLastError());
}

// [images, C, npix] -> [images, npix, C] through an LDS tile (coalesced on both sides).
template <int C>
__global__ __launch_bounds__(256) void sr_pack_nhwc_kernel(const float* __restrict__ src,
                                                           float* __restrict__ dst, int npix) {
  __shared__ float tile[64][C + 1];
  const int img = blockIdx.y;
  const int p0 = blockIdx.x * 64;
  const float* s = src + (size_t)img * C * npix;
  float* d = dst + (size_t)img * npix * C;
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int c = e >> 6, p = e & 63;
    tile[p][c] = (p0 + p < npix) ? s[(size_t)c * npix + p0 + p] : 0.0f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * C; e += 256) {
    const int p = e / C, c = e - p * C;
    if (p0 + p < npix) d[(size_t)(p0 + p) * C + c] = tile[p][c];
  }
}

int sr_launch_pack_nhwc(const float* src_nchw, float* dst_nhwc, int images, int C, int npix,
                        hipStream_t stream) {
  if (images == 0 || npix == 0) return SR_OK;
  dim3 grid((npix + 63) / 64, images), block(256);
  switch (C) {
    case 4: hipLaunchKernelGGL(sr_pack_nhwc_kernel<4>, grid, block, 0, stream, src_nchw, dst_nhwc, npix); break;
    case 8: hipLaunchKernelGGL(sr_pack_nhwc_kernel<8>, grid, block, 0, stream, src_nchw, dst_nhwc, npix); break;
    case 12: hipLaunchKernelGGL(sr_pack_nhwc_kernel<12>, grid, block, 0, stream, src_nchw, dst_nhwc, npix); break;
    case 16: hipLaunchKernelGGL(sr_pack_nhwc_kernel<16>, grid, block, 0, stream, src_nchw, dst_nhwc, npix); break;
    case 24: hipLaunchKernelGGL(sr_pack_nhwc_kernel<24>, grid, block, 0, stream, src_nchw, dst_nhwc, npix); break;
    case 32: hipLaunchKernelGGL(sr_pack_nhwc_kernel<32>, grid, block, 0, stream, src_nchw, dst_nhwc, npix); break;
    default: return SR_ERR_UNSUPPORTED;
  }
  return sr_hip_rc(hipGetLastError());
}

// ------------------------------------------------------------------------ the sweep ---

// dot of one C-channel texel with the reference feature vector (C/4 x 16-byte loads)
template <int C>
__device__ __forceinline__ float sr_tap_dot(const float* __restrict__ img, int texel, const float (&cur)[C]) {
  const float4* t = reinterpret_cast<const float4*>(img + (size_t)texel * C);
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < C / 4; ++i) {
    const float4 v = t[i];
    acc = fmaf(v.x, cur[4 * i + 0], acc);
    acc = fmaf(v.y, cur[4 * i + 1], acc);
    acc = fmaf(v.z, cur[4 * i + 2], acc);
    acc = fmaf(v.w, cur[4 * i + 3], acc);
  }
  return acc;
}

template <int C>
__global__ __launch_bounds__(1024) void sr_dot_volume_kernel(SrDotParams p) {
  extern __shared__ float smem[];  // [S][64] best cost, [S][64] best depth, [S][64] valid flag
  const int lane = threadIdx.x & 63;
  // plane groups: waves of one workgroup (LDS argmax merge below), or -- when there are too few pixel tiles to
  // balance 256 CUs -- separate single-wave workgroups along grid z (lowest cost then comes from sr_argmax_planes)
  const int grp = gridDim.z > 1 ? (int)blockIdx.z : (int)(threadIdx.x >> 6);
  const int S = gridDim.z > 1 ? (int)gridDim.z : (int)(blockDim.x >> 6);
  const int b = blockIdx.y;
  const int N = p.h * p.w;
  // XCD-aware tile order: workgroup id % 8 selects the XCD (observed dispatch order; speed only, never
  // correctness), so give each XCD a CONTIGUOUS band of pixel tiles -- its private L2 then holds one band
  // (plus the disparity range) of every source map instead of all of them.  Bijective for any grid size.
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int pix = tile * 64 + lane;
  const bool active = pix < N;
  const int pc = active ? pix : N - 1;
  const int y = pc / p.w, x = pc - y * p.w;

  const int per = (p.D + S - 1) / S;
  const int j0 = grp * per;
  const int j1 = min(p.D, j0 + per);

  float cur[C];
#pragma unroll
  for (int c = 0; c < C; ++c) cur[c] = p.cur[((size_t)b * C + c) * N + pc];

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
  const float* planes = p.planes.ptr + b * p.planes.sb + y * p.planes.sy + x * p.planes.sx;
  float* out = p.out.cv + b * p.out.sb + (int64_t)pc * p.out.sp;

  float best = 0.0f, best_d = 0.0f;
  bool have = false;
  for (int j = j0; j < j1; ++j) {
    const float d = planes[j * p.planes.sd];
    float X0, X1, X2;
    {
#pragma clang fp contract(off)
      X0 = d * r0; X1 = d * r1; X2 = d * r2;  // geometry_utils.py:56-57
    }
    float cost = 0.0f;
    bool any_depth = false, any_bounds = false;
#pragma unroll 1
    for (int k = 0; k < p.K; ++k) {
      SrSample s;
      sr_project_sample(geom_b + k * SR_GEOM_STRIDE, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, s);
      const float* img = src_b + (size_t)k * N * C;
      const float d_nw = sr_tap_dot<C>(img, s.o_nw, cur);
      const float d_ne = sr_tap_dot<C>(img, s.o_ne, cur);
      const float d_sw = sr_tap_dot<C>(img, s.o_sw, cur);
      const float d_se = sr_tap_dot<C>(img, s.o_se, cur);
      // sum_c (sum_taps w_t * tap_c) * cur_c  ==  sum_taps w_t * (tap . cur)   (cost_volume.py:322-326)
      const float dot = fmaf(s.w_se, d_se, fmaf(s.w_sw, d_sw, fmaf(s.w_ne, d_ne, s.w_nw * d_nw)));
      const bool front = s.zp > 0.0f;  // cost_volume.py:231-232
      cost += front ? dot : 0.0f;      // cost_volume.py:329
      any_depth |= front;
      any_bounds |= sr_in_bounds(s, p.h, p.w);
    }
    if (active) out[j * p.out.sd] = cost;
    if (!have || cost > best || (cost != cost && best == best)) { best = cost; best_d = d; have = true; }  // first max wins; NaN = max (torch.argmax)
    if (j == p.D - 1 && p.out.mask && active)
      p.out.mask[(size_t)b * N + pix] = (uint8_t)(any_depth && any_bounds);
  }

  if (p.out.lowest) {
    if (gridDim.z > 1) {
      // merged by the caller (separate launch)
    } else if (S == 1) {
      if (active) p.out.lowest[(size_t)b * N + pix] = best_d;
    } else {
      float* s_best = smem;
      float* s_d = smem + S * 64;
      float* s_have = smem + 2 * S * 64;
      s_best[grp * 64 + lane] = best;
      s_d[grp * 64 + lane] = best_d;
      s_have[grp * 64 + lane] = have ? 1.0f : 0.0f;
      __syncthreads();
      if (grp == 0 && active) {
        float bb = best, bd = best_d;  // group 0 always owns plane 0
        for (int g = 1; g < S; ++g)
          if (s_have[g * 64 + lane] != 0.0f &&
              (s_best[g * 64 + lane] > bb || (s_best[g * 64 + lane] != s_best[g * 64 + lane] && bb == bb))) {
            bb = s_best[g * 64 + lane];
            bd = s_d[g * 64 + lane];
          }
        p.out.lowest[(size_t)b * N + pix] = bd;
      }
    }
  }
}

// 16-channel variant (what SimpleRecon uses) with texel-coalesced taps.  In the generic kernel a lane owns a pixel and
// pulls its four 64-byte texels with 16 float4 loads, so one load instruction touches 64 different texel records
// (32+ cache lines for 1 KB of data): the L1 tag rate, not its bandwidth, bounds the sweep.  Here the projection is
// still computed with lane = pixel (bit-identical geometry), but the taps are fetched in 4 rounds of 16 pixels with
// lane = (pixel, channel quad): the 4 lanes of a pixel read the 64 contiguous bytes of a texel (one float4 each), each
// forms its 4-channel partial dot, and two DPP quad exchanges finish the reduction.  Sample parameters travel from the
// owning lane with ds_bpermute; per-round reference features are loaded once per tile.
typedef float sr_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sr_quad_sum(float v) {
  // sum over the 4 lanes of a quad: quad_perm [1,0,3,2] then [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}

__global__ __launch_bounds__(1024) void sr_dot_volume_kernel16q(SrDotParams p) {
  constexpr int C = 16;
  extern __shared__ float smem[];  // [S][64] best cost, [S][64] best depth, [S][64] valid flag
  const int lane = threadIdx.x & 63;
  // plane groups: waves of one workgroup (LDS argmax merge below), or -- when there are too few pixel tiles to
  // balance 256 CUs -- separate single-wave workgroups along grid z (lowest cost then comes from sr_argmax_planes)
  const int grp = gridDim.z > 1 ? (int)blockIdx.z : (int)(threadIdx.x >> 6);
  const int S = gridDim.z > 1 ? (int)gridDim.z : (int)(blockDim.x >> 6);
  const int b = blockIdx.y;
  const int N = p.h * p.w;
  int tile = blockIdx.x;  // XCD-aware tile order, see sr_dot_volume_kernel
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int pix = tile * 64 + lane;
  const bool active = pix < N;
  const int pc = active ? pix : N - 1;
  const int y = pc / p.w, x = pc - y * p.w;
  const int per = (p.D + S - 1) / S;
  const int j0 = grp * per, j1 = min(p.D, j0 + per);

  // reference features of the pixel this lane serves in round r (pixel 16r + lane/4), channels 4*(lane%4)..+3
  const int quad = lane & 3, slot = lane >> 2;
  float4 curq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int pr = min(tile * 64 + 16 * r + slot, N - 1);
    const float* cp = p.cur + ((size_t)b * C + 4 * quad) * N + pr;
    curq[r] = make_float4(cp[0], cp[(size_t)N], cp[2 * (size_t)N], cp[3 * (size_t)N]);
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
  const float* planes = p.planes.ptr + b * p.planes.sb + y * p.planes.sy + x * p.planes.sx;
  float* out = p.out.cv + b * p.out.sb + (int64_t)pc * p.out.sp;
  const int gather_lane = 4 * (lane & 15);  // where this lane's own pixel sits inside its round

  float best = 0.0f, best_d = 0.0f;
  bool have = false;
  for (int j = j0; j < j1; ++j) {
    const float d = planes[j * p.planes.sd];
    float X0, X1, X2;
    {
#pragma clang fp contract(off)
      X0 = d * r0; X1 = d * r1; X2 = d * r2;  // geometry_utils.py:56-57
    }
    float costr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool any_depth = false, any_bounds = false;
#pragma unroll 1
    for (int k = 0; k < p.K; ++k) {
      SrSample s;
      sr_project_sample(geom_b + k * SR_GEOM_STRIDE, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, s);
      const bool front = s.zp > 0.0f;  // cost_volume.py:231-232: the mask multiplies the view's dot product
      any_depth |= front;
      any_bounds |= sr_in_bounds(s, p.h, p.w);
      const float w_nw = front ? s.w_nw : 0.0f, w_ne = front ? s.w_ne : 0.0f;
      const float w_sw = front ? s.w_sw : 0.0f, w_se = front ? s.w_se : 0.0f;
      const float* img = src_b + (size_t)k * N * C + 4 * quad;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int sl = 16 * r + slot;
        const int o_nw = __shfl(s.o_nw, sl), o_ne = __shfl(s.o_ne, sl), o_sw = __shfl(s.o_sw, sl), o_se = __shfl(s.o_se, sl);
        const float a_nw = __shfl(w_nw, sl), a_ne = __shfl(w_ne, sl), a_sw = __shfl(w_sw, sl), a_se = __shfl(w_se, sl);
        const float4 t_nw = *reinterpret_cast<const float4*>(img + (size_t)o_nw * C);
        const float4 t_ne = *reinterpret_cast<const float4*>(img + (size_t)o_ne * C);
        const float4 t_sw = *reinterpret_cast<const float4*>(img + (size_t)o_sw * C);
        const float4 t_se = *reinterpret_cast<const float4*>(img + (size_t)o_se * C);
        // sum_c (sum_taps w_t * tap_c) * cur_c (cost_volume.py:322-326) on packed fp32 pairs (v_pk_fma_f32: two FMAs
        // per lane and instruction -- the sweep is VALU-bound, 260 VALU instructions per 64 samples before this)
        const sr_f2 cl = {curq[r].x, curq[r].y}, ch = {curq[r].z, curq[r].w};
        const sr_f2 wnw = {a_nw, a_nw}, wne = {a_ne, a_ne}, wsw = {a_sw, a_sw}, wse = {a_se, a_se};
        sr_f2 lo = wnw * sr_f2{t_nw.x, t_nw.y}, hi = wnw * sr_f2{t_nw.z, t_nw.w};
        lo = __builtin_elementwise_fma(wne, sr_f2{t_ne.x, t_ne.y}, lo);
        hi = __builtin_elementwise_fma(wne, sr_f2{t_ne.z, t_ne.w}, hi);
        lo = __builtin_elementwise_fma(wsw, sr_f2{t_sw.x, t_sw.y}, lo);
        hi = __builtin_elementwise_fma(wsw, sr_f2{t_sw.z, t_sw.w}, hi);
        lo = __builtin_elementwise_fma(wse, sr_f2{t_se.x, t_se.y}, lo);
        hi = __builtin_elementwise_fma(wse, sr_f2{t_se.z, t_se.w}, hi);
        const sr_f2 pr = __builtin_elementwise_fma(hi, ch, lo * cl);
        costr[r] += pr.x + pr.y;
      }
    }
    // quad partial sums -> per-pixel cost, back to lane = pixel
    float cost = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = __shfl(sr_quad_sum(costr[r]), gather_lane);
      cost = (lane >> 4) == r ? v : cost;
    }
    if (active) out[j * p.out.sd] = cost;
    if (!have || cost > best || (cost != cost && best == best)) { best = cost; best_d = d; have = true; }  // first max wins; NaN = max (torch.argmax)
    if (j == p.D - 1 && p.out.mask && active)
      p.out.mask[(size_t)b * N + pix] = (uint8_t)(any_depth && any_bounds);
  }

  if (p.out.lowest) {
    if (gridDim.z > 1) {
      // merged by the caller (separate launch)
    } else if (S == 1) {
      if (active) p.out.lowest[(size_t)b * N + pix] = best_d;
    } else {
      float* s_best = smem;
      float* s_d = smem + S * 64;
      float* s_have = smem + 2 * S * 64;
      s_best[grp * 64 + lane] = best;
      s_d[grp * 64 + lane] = best_d;
      s_have[grp * 64 + lane] = have ? 1.0f : 0.0f;
      __syncthreads();
      if (grp == 0 && active) {
        float bb = best, bd = best_d;  // group 0 always owns plane 0
        for (int g = 1; g < S; ++g)
          if (s_have[g * 64 + lane] != 0.0f &&
              (s_best[g * 64 + lane] > bb || (s_best[g * 64 + lane] != s_best[g * 64 + lane] && bb == bb))) {
            bb = s_best[g * 64 + lane];
            bd = s_d[g * 64 + lane];
          }
        p.out.lowest[(size_t)b * N + pix] = bd;
      }
    }
  }
}

// ------------------------------------------------------------------------ C ABI -------

extern "C" int sr_abi_version(void) { return 1; }
extern "C" const char* sr_target_arch(void) { return "gfx950"; }

extern "C" size_t sr_volume_workspace_bytes(int B, int K, int C, int h, int w) {
  if (B < 0 || K < 0 || C < 0 || h < 0 || w < 0) return 0;
  const size_t geom = sr_align_up((size_t)B * K * SR_GEOM_STRIDE * sizeof(float), 256);
  const size_t nhwc = sr_ws_nhwc_bytes(B, K, C, h, w);
  const size_t keys = sr_align_up((size_t)B * h * w * sizeof(unsigned long long), 256);  // LDS-staged sweep's argmax keys
  return geom + nhwc + keys + 256;
}

static int sr_pick_plane_split(int B, int N, int D) {
  // Enough waves to cover 256 CUs several times over; never more groups than planes.
  const long tiles = (long)B * ((N + 63) / 64);
  int S = 1;
  while (S < 16 && tiles * S < 256L * 16 && S * 2 <= D) S *= 2;
  return S;
}

// geometry records + channels-last source features -> workspace (shared by both sweeps)
extern "C" int sr_volume_prepare(const float* src, const float* K_src, const float* T_src_cur,
                                 const float* T_cur_src, int B, int K, int C, int h, int w, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!src || !K_src || !T_src_cur || !workspace) return SR_ERR_INVALID_ARGUMENT;
  if (C % 4 != 0 || C > 32) return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_volume_workspace_bytes(B, K, C, h, w)) return SR_ERR_WORKSPACE_TOO_SMALL;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = sr_launch_geom(K_src, T_src_cur, T_cur_src, sr_ws_geom(workspace), B * K, stream);
  if (rc) return rc;
  return sr_launch_pack_nhwc(src, sr_ws_src_nhwc(workspace, B, K), B * K, C, h * w, stream);
}

extern "C" int sr_dot_volume_sweep(const float* cur, const float* invK_cur, const float* planes, int64_t ps_b,
                                   int64_t ps_d, int64_t ps_y, int64_t ps_x, int B, int K, int C, int h, int w,
                                   int D, float* out_cv, int64_t cv_sb, int64_t cv_sd, int64_t cv_sp,
                                   float* out_lowest, uint8_t* out_mask, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!cur || !invK_cur || !planes || !out_cv || !workspace) return SR_ERR_INVALID_ARGUMENT;
  if (C % 4 != 0 || C > 32) return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_volume_workspace_bytes(B, K, C, h, w)) return SR_ERR_WORKSPACE_TOO_SMALL;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = h * w;

  SrDotParams p;
  p.cur = cur; p.src_nhwc = sr_ws_src_nhwc(workspace, B, K); p.invK = invK_cur; p.geom = sr_ws_geom(workspace);
  p.planes = {planes, ps_b, ps_d, ps_y, ps_x};
  p.out = {out_cv, cv_sb, cv_sd, cv_sp, out_lowest, out_mask};
  p.B = B; p.K = K; p.h = h; p.w = w; p.D = D;
  p.inv_w = (float)(1.0 / (double)w);
  p.inv_h = (float)(1.0 / (double)h);

  if (C == 16) {
    // default: footprints of 8x32-pixel tiles staged in LDS (sr_dot_volume_lds.hip); SR_DOT_LDS=0 selects the
    // L1-gather kernels below (ablation), which also take the shapes the staged kernel refuses
    const int use_lds = sr_opt(SR_OPT_DOT_LDS);
    if (use_lds) {
      const int rc = sr_launch_dot_volume_lds(p, sr_ws_keys(workspace, B, K, C, h, w), stream);
      if (rc != SR_ERR_UNSUPPORTED) return rc;
    }
  }
  const int S = sr_pick_plane_split(B, N, D);
  // With few pixel tiles (batch 1: 300 workgroups of 16 waves on 256 CUs) whole-workgroup granularity leaves CUs
  // idle: spread the plane groups over grid z as single-wave workgroups and take the lowest cost in a second launch.
  const bool spread = S > 1 && (long)B * ((N + 63) / 64) < 4L * 256;
  dim3 grid((N + 63) / 64, B, spread ? S : 1), block(spread ? 64 : 64 * S);
  const size_t lds = spread ? 0 : (size_t)3 * S * 64 * sizeof(float);
  switch (C) {
    case 4: hipLaunchKernelGGL(sr_dot_volume_kernel<4>, grid, block, lds, stream, p); break;
    case 8: hipLaunchKernelGGL(sr_dot_volume_kernel<8>, grid, block, lds, stream, p); break;
    case 12: hipLaunchKernelGGL(sr_dot_volume_kernel<12>, grid, block, lds, stream, p); break;
    case 16: {
      const int quad = sr_opt(SR_OPT_DOT_QUAD);  // 0 selects the generic lane-per-pixel kernel (ablation)
      if (quad) hipLaunchKernelGGL(sr_dot_volume_kernel16q, grid, block, lds, stream, p);
      else hipLaunchKernelGGL(sr_dot_volume_kernel<16>, grid, block, lds, stream, p);
      break;
    }
    case 24: hipLaunchKernelGGL(sr_dot_volume_kernel<24>, grid, block, lds, stream, p); break;
    case 32: hipLaunchKernelGGL(sr_dot_volume_kernel<32>, grid, block, lds, stream, p); break;
    default: return SR_ERR_UNSUPPORTED;
  }
  int rc = sr_hip_rc(hipGetLastError());
  if (rc == SR_OK && spread && out_lowest)
    rc = sr_launch_argmax_planes(out_cv, cv_sb, cv_sd, cv_sp, p.planes, B, h, w, D, out_lowest, stream);
  return rc;
}

extern "C" int sr_dot_volume_fwd(const float* cur, const float* src, const float* K_src,
                                 const float* T_src_cur, const float* invK_cur, const float* planes,
                                 int64_t ps_b, int64_t ps_d, int64_t ps_y, int64_t ps_x, int B, int K, int C,
                                 int h, int w, int D, float* out_cv, int64_t cv_sb, int64_t cv_sd,
                                 int64_t cv_sp, float* out_lowest, uint8_t* out_mask, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
  if (!cur || !src) return B == 0 ? SR_OK : SR_ERR_INVALID_ARGUMENT;
  int rc = sr_volume_prepare(src, K_src, T_src_cur, nullptr, B, K, C, h, w, workspace, workspace_bytes, stream_);
  if (rc) return rc;
  return sr_dot_volume_sweep(cur, invK_cur, planes, ps_b, ps_d, ps_y, ps_x, B, K, C, h, w, D, out_cv, cv_sb,
                             cv_sd, cv_sp, out_lowest, out_mask, workspace, workspace_bytes, stream_);
}

// ------------------------------------------------------------------ warp_features ---------
// Materialising variant of the sweep's front end for the reference's public helper
// CostVolumeManager.warp_features / FastFeatureVolumeManager.warp_features
// (cost_volume.py:139-234, 812-964): one thread per (pixel, b, k, plane).
struct SrWarpParams {
  const float* src_nhwc; const float* invK; const float* geom;
  SrPlanes planes;
  float* world;   // [B,Dp,4,N] or null
  float* depths;  // [B,K,Dp,N]
  float* warped;  // [B,K,Dp,C,N]
  float* mask;    // [B,K,Dp,N]
  float* pix;     // [B,K,Dp,2,N] or null
  int B, K, C, h, w, Dp;
  float inv_w, inv_h;
};

__global__ __launch_bounds__(256) void sr_warp_features_kernel(SrWarpParams p) {
  const int N = p.h * p.w;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= N) return;
  int z = blockIdx.y;
  const int j = z % p.Dp; z /= p.Dp;
  const int k = z % p.K;
  const int b = z / p.K;
  const int y = pix / p.w, x = pix - y * p.w;
  const float d = p.planes.ptr[b * p.planes.sb + j * p.planes.sd + y * p.planes.sy + x * p.planes.sx];
  float X0, X1, X2;
  {
#pragma clang fp contract(off)
    const float* iK = p.invK + 16 * (size_t)b;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    X0 = d * (iK[0] * px + iK[1] * py + iK[2]);
    X1 = d * (iK[4] * px + iK[5] * py + iK[6]);
    X2 = d * (iK[8] * px + iK[9] * py + iK[10]);
  }
  if (p.world && k == 0) {
    float* wp = p.world + ((size_t)b * p.Dp + j) * 4 * N + pix;
    wp[0] = X0; wp[N] = X1; wp[2 * (size_t)N] = X2; wp[3 * (size_t)N] = 1.0f;
  }
  SrSample s;
  sr_project_sample(p.geom + ((size_t)b * p.K + k) * SR_GEOM_STRIDE, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, s);
  const size_t rec = ((size_t)b * p.K + k) * p.Dp + j;
  p.depths[rec * N + pix] = s.zp;
  p.mask[rec * N + pix] = s.zp > 0.0f ? 1.0f : 0.0f;
  if (p.pix) { p.pix[(rec * 2 + 0) * N + pix] = s.pix_x; p.pix[(rec * 2 + 1) * N + pix] = s.pix_y; }
  const float* img = p.src_nhwc + ((size_t)b * p.K + k) * N * p.C;
  float* out = p.warped + rec * p.C * N + pix;
  for (int c = 0; c < p.C; ++c) {
    // same accumulation order as ATen's grid_sampler_2d: nw, ne, sw, se
    float v = img[(size_t)s.o_nw * p.C + c] * s.w_nw;
    v = fmaf(img[(size_t)s.o_ne * p.C + c], s.w_ne, v);
    v = fmaf(img[(size_t)s.o_sw * p.C + c], s.w_sw, v);
    v = fmaf(img[(size_t)s.o_se * p.C + c], s.w_se, v);
    out[(size_t)c * N] = v;
  }
}

extern "C" int sr_warp_features_fwd(const float* src, const float* K_src, const float* T_src_cur,
                                    const float* invK_cur, const float* planes, int64_t ps_b, int64_t ps_d,
                                    int64_t ps_y, int64_t ps_x, int B, int K, int C, int h, int w, int Dp,
                                    float* out_world, float* out_depths, float* out_warped, float* out_mask,
                                    float* out_pix, void* workspace, size_t workspace_bytes, void* stream_) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || Dp <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!invK_cur || !planes || !out_depths || !out_warped || !out_mask) return SR_ERR_INVALID_ARGUMENT;
  int rc = sr_volume_prepare(src, K_src, T_src_cur, nullptr, B, K, C, h, w, workspace, workspace_bytes, stream_);
  if (rc) return rc;
  SrWarpParams p;
  p.src_nhwc = sr_ws_src_nhwc(workspace, B, K); p.invK = invK_cur; p.geom = sr_ws_geom(workspace);
  p.planes = {planes, ps_b, ps_d, ps_y, ps_x};
  p.world = out_world; p.depths = out_depths; p.warped = out_warped; p.mask = out_mask; p.pix = out_pix;
  p.B = B; p.K = K; p.C = C; p.h = h; p.w = w; p.Dp = Dp;
  p.inv_w = (float)(1.0 / (double)w);
  p.inv_h = (float)(1.0 / (double)h);
  const int N = h * w;
  if ((long)B * K * Dp > 65535) return SR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sr_warp_features_kernel, dim3((N + 255) / 256, B * K * Dp), dim3(256), 0, (hipStream_t)stream_, p);
  return sr_hip_rc(hipGetLastError());
}
