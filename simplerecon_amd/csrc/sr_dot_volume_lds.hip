// sr_dot_volume_lds.hip -- dot-product plane sweep with the bilinear taps staged through LDS tiles (gfx950, C = 16).
//
// Same operator as sr_dot_volume.hip (reference modules/cost_volume.py:139-234 warp_features, :237-335
// build_cost_volume, :338-380 forward): cost[b,j,y,x] = sum_k (z'_k > 0) * sum_c bilinear(src_k)[c] * cur[c].
// The L1-gather kernels there pull every tap (64 B) of every (pixel, plane, view) sample through the vector L1:
// 2.2 GB per 640x480 frame against 14.8 MB of compulsory HBM bytes, and the L1 tag/data rate is the bound.  Here a
// workgroup owns a 2-D tile of 8 x 32 reference pixels and a group of 8 consecutive depth planes.  Per source view:
//   1. every lane projects its pixel onto the view for the 8 planes (sr_project_sample_xy: the same instruction
//      sequence as the other sweeps -> identical taps, weights, masks);
//   2. the bounding box of the tile's tap footprint over those planes is reduced (packed u16 min / max, DPP inside a
//      wave, LDS across the 4 waves).  Consecutive planes move the footprint by a fraction of the tile size along the
//      epipolar line, so one box serves all 8 planes: ~0.2 texels are staged per sample instead of 4 gathered;
//   3. the box is copied global -> LDS with coalesced 64-byte texel reads (4 lanes x float4 per texel; rows of the box
//      are contiguous in the channels-last source map).  LDS layout = 4 planes of [texel] float4 (one per channel
//      quad), plane pitch = 32 mod 128 bytes: a lane's four ds_read_b128 of a tap are `base + const`, and 16 lanes
//      reading 16 consecutive texels of one channel quad cover all 64 banks (conflict-free);
//   4. lane = pixel takes its 4 taps x 64 B from LDS (16 ds_read_b128) and accumulates w_t * (tap_t . cur).
// Views / plane groups whose footprint lies wholly outside the source image (or behind the camera) are skipped after
// step 2 -- they contribute exactly 0 (57 % of the (tile, view, plane) units of the synthetic 7-view configuration).
// A box that does not fit the LDS buffer is split (8 -> 4 -> 2 -> 1 planes); a single plane that still does not fit
// (extreme zoom-out homography) takes its taps straight from global memory.
// lowest_cost (argmax over ALL planes, first maximum wins, NaN counts as the maximum like torch.argmax): plane groups
// live in different workgroups, so each folds (orderable(cost) << 32 | ~plane) into a per-pixel 64-bit key with one
// atomic max; a small kernel turns the winning plane index into its depth.
#include <stdlib.h>

#include "sr_common.h"

#define LT_H 8    // tile rows
#define LT_W 32   // tile columns (a wave = 2 rows x 32 columns: every hardware 16-lane LDS group stays inside a row)
#define SR_NOREL 0xFFFFFFFFu

typedef unsigned short sr_us2 __attribute__((ext_vector_type(2)));
typedef float sr_f2 __attribute__((ext_vector_type(2)));
typedef float sr_f4 __attribute__((ext_vector_type(4)));  // (arrays of HIP float4 structs are not promoted to registers)

template <bool MIN>
__device__ __forceinline__ unsigned sr_pk16(unsigned a, unsigned b) {
  const sr_us2 x = __builtin_bit_cast(sr_us2, a), y = __builtin_bit_cast(sr_us2, b);
  return __builtin_bit_cast(unsigned, MIN ? __builtin_elementwise_min(x, y) : __builtin_elementwise_max(x, y));
}

#define SR_DPP(v, ctrl) ((unsigned)__builtin_amdgcn_mov_dpp((int)(v), (ctrl), 0xF, 0xF, true))

// componentwise u16 min / max over the 64 lanes (all lanes must be active); result is wave-uniform
template <bool MIN>
__device__ __forceinline__ unsigned sr_wave_pk16(unsigned v) {
  v = sr_pk16<MIN>(v, SR_DPP(v, 0xB1));   // quad_perm [1,0,3,2]
  v = sr_pk16<MIN>(v, SR_DPP(v, 0x4E));   // quad_perm [2,3,0,1]
  v = sr_pk16<MIN>(v, SR_DPP(v, 0x141));  // row_half_mirror
  v = sr_pk16<MIN>(v, SR_DPP(v, 0x140));  // row_mirror: every lane of a 16-lane row now holds the row's result
  const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
  return sr_pk16<MIN>(sr_pk16<MIN>(r0, r1), sr_pk16<MIN>(r2, r3));
}

__device__ __forceinline__ float sr_dot16(const float4& a, const float4& b, const float4& c, const float4& d,
                                          const float (&cur)[16]) {
  // 16-channel dot on packed fp32 pairs (v_pk_fma_f32): even / odd channel chains, one add at the end
  sr_f2 acc = sr_f2{a.x, a.y} * sr_f2{cur[0], cur[1]};
  acc = __builtin_elementwise_fma(sr_f2{a.z, a.w}, sr_f2{cur[2], cur[3]}, acc);
  acc = __builtin_elementwise_fma(sr_f2{b.x, b.y}, sr_f2{cur[4], cur[5]}, acc);
  acc = __builtin_elementwise_fma(sr_f2{b.z, b.w}, sr_f2{cur[6], cur[7]}, acc);
  acc = __builtin_elementwise_fma(sr_f2{c.x, c.y}, sr_f2{cur[8], cur[9]}, acc);
  acc = __builtin_elementwise_fma(sr_f2{c.z, c.w}, sr_f2{cur[10], cur[11]}, acc);
  acc = __builtin_elementwise_fma(sr_f2{d.x, d.y}, sr_f2{cur[12], cur[13]}, acc);
  acc = __builtin_elementwise_fma(sr_f2{d.z, d.w}, sr_f2{cur[14], cur[15]}, acc);
  return acc.x + acc.y;
}

struct SrLdsCtx {
  float4* tex;           // [4][CAP] float4
  unsigned* sbox;        // [2][8]
  const float* img;      // channels-last source map of the current view
  int w, h;
  int tid, wave, lane;   // wave is held in an SGPR (readfirstlane)
  unsigned round;
};

// 1 / x for two values at once, bit-identical to the IEEE division `1.0f / x` the other sweeps compile to, provided
// 2^-60 <= |x| <= 2^60 (checked by the caller): the AMDGPU f32 division expansion with its scaling steps removed --
// v_div_scale leaves such operands alone, v_div_fmas is then a plain fma and v_div_fixup the identity -- i.e. v_rcp_f32,
// one Newton step, and two residual corrections, here on packed pairs (6 v_pk_fma_f32 for two reciprocals).
__device__ __forceinline__ sr_f2 sr_rcp2_exact(sr_f2 x) {
  const sr_f2 one = {1.0f, 1.0f};
  sr_f2 r = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  const sr_f2 e = __builtin_elementwise_fma(-x, r, one);
  r = __builtin_elementwise_fma(e, r, r);
  sr_f2 rem = __builtin_elementwise_fma(-x, r, one);
  const sr_f2 q = __builtin_elementwise_fma(rem, r, r);
  rem = __builtin_elementwise_fma(-x, q, one);
  return __builtin_elementwise_fma(rem, r, q);
}

// Projection of one reference pixel onto a source view at TWO depth planes: every operation is the one of
// sr_project_sample_xy (sr_common.h), applied to a packed pair (IEEE per component, FP contraction off), so the sampling
// positions are bit-identical to the per-sample code at half the VALU instructions (packed fp32 ops issue at the rate
// of plain ones on gfx950).
struct SrPair {
  sr_f2 zp, pix_x, pix_y, ix, iy;
};

__device__ __forceinline__ void sr_project_pair(const float* __restrict__ g, float r0, float r1, float r2, sr_f2 d,
                                                int h, int w, float inv_w, float inv_h, SrPair& s) {
#pragma clang fp contract(off)
  const float eps = 1e-8f;
  const sr_f2 X0 = d * r0, X1 = d * r1, X2 = d * r2;  // geometry_utils.py:56-57
  const sr_f2 q0 = g[0] * X0 + g[1] * X1 + g[2] * X2 + g[3];
  const sr_f2 q1 = g[4] * X0 + g[5] * X1 + g[6] * X2 + g[7];
  const sr_f2 q2 = g[8] * X0 + g[9] * X1 + g[10] * X2 + g[11];
  s.zp = q2 + eps;
  const bool div0 = fabsf(q2.x) > eps, div1 = fabsf(q2.y) > eps;
  const float az0 = fabsf(s.zp.x), az1 = fabsf(s.zp.y);
  const bool odd = (div0 & !((az0 >= 0x1p-60f) & (az0 <= 0x1p60f))) | (div1 & !((az1 >= 0x1p-60f) & (az1 <= 0x1p60f)));
  sr_f2 sc;
  if (__ballot(odd) == 0) {
    // (lanes that do not divide get a harmless operand)
    const sr_f2 rc = sr_rcp2_exact(sr_f2{div0 ? s.zp.x : 1.0f, div1 ? s.zp.y : 1.0f});
    sc = sr_f2{div0 ? rc.x : 1.0f, div1 ? rc.y : 1.0f};
  } else {  // operands outside the fast path's range somewhere in the wave: the compiler's full division
    sc = sr_f2{div0 ? 1.0f / s.zp.x : 1.0f, div1 ? 1.0f / s.zp.y : 1.0f};
  }
  s.pix_x = q0 * sc;
  s.pix_y = q1 * sc;
  const sr_f2 uvx = 2.0f * s.pix_x * inv_w - 1.0f, uvy = 2.0f * s.pix_y * inv_h - 1.0f;  // cost_volume.py:199
  s.ix = ((uvx + 1.0f) * (float)w - 1.0f) / 2.0f;  // grid_sample unnormalise, align_corners=False
  s.iy = ((uvy + 1.0f) * (float)h - 1.0f) / 2.0f;
}

// Tap origins + weights of the two planes of a pair: the arithmetic of sr_bilinear_taps on packed pairs.  A sample
// contributes (`rel`) iff it is wanted (`ok`: active pixel, existing plane, in front of the source camera) and its NW
// origin lies in [-1, w-1] x [-1, h-1], i.e. -1 <= ix < w and -1 <= iy < h (float compares: a NaN position never
// contributes).  Positions that do not contribute are replaced by (-8, 0) up front, so everything downstream is finite
// and their weights come out as exact zeros without further masking.  Out-of-image taps: instead of selecting each
// product w = valid_x & valid_y ? ax * ay : 0 (sr_bilinear_taps), the FACTORS are selected (ax = valid_x ? ax : 0, ...),
// which yields the same products or exact zeros with half the selects.
// xy = (x0+1) | (y0+1) << 16 of the NW tap, SR_NOREL when the sample contributes nothing.
__device__ __forceinline__ void sr_pair_taps(const SrPair& s, bool ok0, bool ok1, int h, int w, unsigned& xy0,
                                             unsigned& xy1, sr_f2 (&wt)[4]) {
#pragma clang fp contract(off)
  const float fw = (float)w, fh = (float)h;
  const bool rel0 = ok0 & (s.ix.x >= -1.0f) & (s.ix.x < fw) & (s.iy.x >= -1.0f) & (s.iy.x < fh);
  const bool rel1 = ok1 & (s.ix.y >= -1.0f) & (s.ix.y < fw) & (s.iy.y >= -1.0f) & (s.iy.y < fh);
  const sr_f2 ix = {rel0 ? s.ix.x : -8.0f, rel1 ? s.ix.y : -8.0f}, iy = {rel0 ? s.iy.x : 0.0f, rel1 ? s.iy.y : 0.0f};
  const sr_f2 f0x = {floorf(ix.x), floorf(ix.y)}, f0y = {floorf(iy.x), floorf(iy.y)};
  const sr_f2 f1x = f0x + 1.0f, f1y = f0y + 1.0f;
  sr_f2 a1x = f1x - ix, a0x = ix - f0x, a1y = f1y - iy, a0y = iy - f0y;
  const int x00 = (int)f0x.x, x01 = (int)f0x.y, y00 = (int)f0y.x, y01 = (int)f0y.y;
  const unsigned uw = (unsigned)w, uh = (unsigned)h;
  a1x = sr_f2{(unsigned)x00 < uw ? a1x.x : 0.0f, (unsigned)x01 < uw ? a1x.y : 0.0f};              // west column valid
  a0x = sr_f2{(unsigned)(x00 + 1) < uw ? a0x.x : 0.0f, (unsigned)(x01 + 1) < uw ? a0x.y : 0.0f};  // east column
  a1y = sr_f2{(unsigned)y00 < uh ? a1y.x : 0.0f, (unsigned)y01 < uh ? a1y.y : 0.0f};              // north row
  a0y = sr_f2{(unsigned)(y00 + 1) < uh ? a0y.x : 0.0f, (unsigned)(y01 + 1) < uh ? a0y.y : 0.0f};  // south row
  wt[0] = a1x * a1y;  // nw
  wt[1] = a0x * a1y;  // ne
  wt[2] = a1x * a0y;  // sw
  wt[3] = a0x * a0y;  // se
  xy0 = rel0 ? ((unsigned)(x00 + 1) | ((unsigned)(y00 + 1) << 16)) : SR_NOREL;
  xy1 = rel1 ? ((unsigned)(x01 + 1) | ((unsigned)(y01 + 1) << 16)) : SR_NOREL;
}

// Staging of a footprint box, chunk by chunk (compile-time recursion keeps the register array statically indexed):
// chunk u of this wave = chunk (wave + 4u) of the box = 16 consecutive texels of one row; lane = (texel, channel quad).
// row = chunk / nchunk through a 16.16 fixed-point reciprocal (exact for the < 128 chunks of a box): scalar ALU only.
// r06: buffer loads -- the row part of a chunk's address is scalar (soffset = clamped row * row bytes), the lane part is
// clamp(lane column + chunk column) * 64 + channel quad: 3 vector instructions per chunk instead of the 11 of the 64-bit flat
// address (PMC r02-r06: the sweep's issue port is saturated, 2 785 VALU + 1 466 SALU per wave; a fifth of them was this
// address arithmetic).  Same texels, same order, same values.
template <int U, int NU, int U0>
__device__ __forceinline__ void sr_stage_load(const SrLdsCtx& c, sr_f4 (&v)[NU], unsigned rcp_nchunk, int nchunk,
                                              int nchunks, int lane_x0, int miny, __amdgpu_buffer_rsrc_t rs_img) {
  if constexpr (U < NU) {
    const int ch = c.wave + 4 * (U0 + U);
    if (ch < nchunks) {
      const int ry = (int)(((unsigned)ch * rcp_nchunk) >> 16), cx = ch - ry * nchunk;   // (scalar)
      const int gy = min(max(miny - 1 + ry, 0), c.h - 1);                               // (scalar)
      const int gx = min(max(lane_x0 + cx * 16, 0), c.w - 1);
#ifdef SR_DOT_ABL_HOT   // (timing experiment: every staging load hits the same L1-resident texels -- what the load latency costs)
      v[U] = __builtin_bit_cast(sr_f4, __builtin_amdgcn_raw_buffer_load_b128(rs_img, (gx & 1) * 64 + (c.lane & 3) * 16, (gy & 1) * 64, 0));
#else
      v[U] = __builtin_bit_cast(sr_f4, __builtin_amdgcn_raw_buffer_load_b128(rs_img, gx * 64 + (c.lane & 3) * 16, gy * c.w * 64, 0));
#endif
      sr_stage_load<U + 1, NU, U0>(c, v, rcp_nchunk, nchunk, nchunks, lane_x0, miny, rs_img);
    }
  }
}

template <int CAP, int U, int NU, int U0>
__device__ __forceinline__ void sr_stage_store(const SrLdsCtx& c, const sr_f4 (&v)[NU], int nchunks) {
  if constexpr (U < NU) {
    if (c.wave + 4 * (U0 + U) < nchunks) {
      reinterpret_cast<sr_f4*>(c.tex)[(c.lane & 3) * CAP + (c.wave + 4 * (U0 + U)) * 16 + (c.lane >> 2)] = v[U];
      sr_stage_store<CAP, U + 1, NU, U0>(c, v, nchunks);
    }
  }
}

// Planes [LO, LO+N) of the current view: footprint box -> LDS -> taps.  xy / wt: per-lane tap state of the G planes.
template <int CAP, int G, int LO, int N>
__device__ __forceinline__ void sr_lds_unit(SrLdsCtx& c, const unsigned (&xy)[G], const sr_f2 (&wt)[G / 2][4],
                                            const float (&cur)[16], float (&cost)[G]) {
  unsigned mn = SR_NOREL, mx = 0u;
#pragma unroll
  for (int g = LO; g < LO + N; ++g) {
    mn = sr_pk16<true>(mn, xy[g]);
    mx = sr_pk16<false>(mx, xy[g] == SR_NOREL ? 0u : xy[g]);
  }
  mn = sr_wave_pk16<true>(mn);
  mx = sr_wave_pk16<false>(mx);
  unsigned* sb = c.sbox + (c.round & 1u) * 8;
  c.round++;
  if (c.lane == 0) { sb[2 * c.wave] = mn; sb[2 * c.wave + 1] = mx; }
  __syncthreads();  // (A) box parts visible; every wave is done with the taps of the previous unit
  {
    const uint4 a = *reinterpret_cast<const uint4*>(sb), b = *reinterpret_cast<const uint4*>(sb + 4);
    mn = sr_pk16<true>(sr_pk16<true>(a.x, a.z), sr_pk16<true>(b.x, b.z));
    mx = sr_pk16<false>(sr_pk16<false>(a.y, a.w), sr_pk16<false>(b.y, b.w));
  }
  mn = (unsigned)__builtin_amdgcn_readfirstlane((int)mn);
  mx = (unsigned)__builtin_amdgcn_readfirstlane((int)mx);
  if (mn == SR_NOREL) return;  // nothing of this tile lands in the view at these planes
  const int minx = (int)(mn & 0xFFFFu), miny = (int)(mn >> 16);
  const int bw = (int)(mx & 0xFFFFu) - minx + 2, bh = (int)(mx >> 16) - miny + 2;  // NW origins + the E / S taps
  // rows are staged in chunks of 16 texels: the LDS row pitch is a multiple of 16 texels, so a 16-lane group whose
  // pixels straddle two footprint rows still hits 16 different bank groups
  const int nchunk = (bw + 15) >> 4, pitch = nchunk << 4, nchunks = nchunk * bh;
  if (nchunks * 16 <= CAP) {
    // stage rows [miny-1, miny-1+bh) x columns [minx-1, minx-1+pitch) (coordinates clamped into the image: clamped
    // copies are only ever read with weight 0).  A wave takes chunks wave, wave+4, ...; lane = (texel, channel quad):
    // 4 lanes read the 64 contiguous bytes of a texel, a wave 1 KiB; all loads are issued before the first LDS store.
    // floor(ch / nchunk) = (ch * rcp) >> 16 with rcp = floor(65536 / nchunk) + 1, valid for ch * nchunk < 65536
    const unsigned rcp_nchunk = (unsigned)__builtin_amdgcn_readfirstlane((int)(65536.0f / (float)nchunk)) + 1u;
    // two batches (6 + the rest of the <= NU chunks of a wave): 24 staging registers instead of 40 keep the kernel
    // inside the 128-register budget of 4 workgroups per CU; typical boxes (<= 24 chunks) need the first batch only
    constexpr int NU = (CAP / 16 + 3) / 4, NA = NU < 6 ? NU : 6;
    const __amdgpu_buffer_rsrc_t rs_img =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.img), 0, c.h * c.w * 64, 0x00020000);
    const int lane_x0 = minx - 1 + (c.lane >> 2);
    {
      sr_f4 v[NA];
      sr_stage_load<0, NA, 0>(c, v, rcp_nchunk, nchunk, nchunks, lane_x0, miny, rs_img);
      sr_stage_store<CAP, 0, NA, 0>(c, v, nchunks);
    }
    if constexpr (NU > NA) {
      if (c.wave + 4 * NA < nchunks) {
        sr_f4 v[NU - NA];
        sr_stage_load<0, NU - NA, NA>(c, v, rcp_nchunk, nchunk, nchunks, lane_x0, miny, rs_img);
        sr_stage_store<CAP, 0, NU - NA, NA>(c, v, nchunks);
      }
    }
    __syncthreads();  // (B) footprint staged
#pragma unroll
    for (int g = LO; g < LO + N; ++g) {
      const bool rel = xy[g] != SR_NOREL;
      if (__ballot(rel) == 0) continue;  // wave-uniform: none of this wave's 64 pixels lands in the view at plane g
      const int t = rel ? ((int)(xy[g] >> 16) - miny) * pitch + ((int)(xy[g] & 0xFFFFu) - minx) : 0;
      const float4* n = c.tex + t;
      const float4* s = n + pitch;
      const float d_nw = sr_dot16(n[0], n[CAP], n[2 * CAP], n[3 * CAP], cur);
      const float d_ne = sr_dot16(n[1], n[CAP + 1], n[2 * CAP + 1], n[3 * CAP + 1], cur);
      const float d_sw = sr_dot16(s[0], s[CAP], s[2 * CAP], s[3 * CAP], cur);
      const float d_se = sr_dot16(s[1], s[CAP + 1], s[2 * CAP + 1], s[3 * CAP + 1], cur);
      // sum_c (sum_taps w_t * tap_c) * cur_c  ==  sum_taps w_t * (tap . cur)   (cost_volume.py:322-326)
      cost[g] += fmaf(wt[g / 2][3][g & 1], d_se, fmaf(wt[g / 2][2][g & 1], d_sw,
                                                      fmaf(wt[g / 2][1][g & 1], d_ne, wt[g / 2][0][g & 1] * d_nw)));
    }
  } else if constexpr (N > 1) {
    sr_lds_unit<CAP, G, LO, N / 2>(c, xy, wt, cur, cost);
    sr_lds_unit<CAP, G, LO + N / 2, N / 2>(c, xy, wt, cur, cost);
  } else {
    // one plane whose footprint exceeds the LDS buffer: taps from global memory (clamped addresses, weight 0 outside)
    const bool rel = xy[LO] != SR_NOREL;
    const int x0 = rel ? (int)(xy[LO] & 0xFFFFu) - 1 : 0, y0 = rel ? (int)(xy[LO] >> 16) - 1 : 0;
    const int xa = max(x0, 0), xb = min(x0 + 1, c.w - 1), ya = max(y0, 0), yb = min(y0 + 1, c.h - 1);
    const float4* nw = reinterpret_cast<const float4*>(c.img + (size_t)(ya * c.w + xa) * 16);
    const float4* ne = reinterpret_cast<const float4*>(c.img + (size_t)(ya * c.w + xb) * 16);
    const float4* sw = reinterpret_cast<const float4*>(c.img + (size_t)(yb * c.w + xa) * 16);
    const float4* se = reinterpret_cast<const float4*>(c.img + (size_t)(yb * c.w + xb) * 16);
    const float d_nw = sr_dot16(nw[0], nw[1], nw[2], nw[3], cur);
    const float d_ne = sr_dot16(ne[0], ne[1], ne[2], ne[3], cur);
    const float d_sw = sr_dot16(sw[0], sw[1], sw[2], sw[3], cur);
    const float d_se = sr_dot16(se[0], se[1], se[2], se[3], cur);
    cost[LO] += fmaf(wt[LO / 2][3][LO & 1], d_se, fmaf(wt[LO / 2][2][LO & 1], d_sw,
                                                      fmaf(wt[LO / 2][1][LO & 1], d_ne, wt[LO / 2][0][LO & 1] * d_nw)));
  }
}

template <int CAP, int WPS, int G>
__global__ __launch_bounds__(256, WPS) void sr_dot_volume_lds_kernel(SrDotParams p, unsigned long long* __restrict__ keys,
                                                                      int tiles_x, int tiles_y, int groups, int cull) {
  static_assert(G % 2 == 0, "planes are projected in pairs");
  __shared__ float4 tex[4 * CAP];
  __shared__ __attribute__((aligned(16))) unsigned sbox[16];
  SrLdsCtx c;
  c.tex = tex; c.sbox = sbox; c.w = p.w; c.h = p.h; c.round = 0;
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); c.lane = threadIdx.x & 63;

  // XCD-aware order: workgroup id % 8 selects the XCD (observed dispatch order; speed only), so every XCD gets a
  // CONTIGUOUS band of the (batch, tile, plane-group) sequence: its L2 holds one band of rows of each source map.
  int id = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int grp = id % groups;
  id /= groups;
  const int ntile = tiles_x * tiles_y;
  const int tile = id % ntile, b = id / ntile;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int x = tx * LT_W + (c.lane & 31), y = ty * LT_H + 2 * c.wave + (c.lane >> 5);
  const bool active = (x < p.w) & (y < p.h);
  const int xc = min(x, p.w - 1), yc = min(y, p.h - 1);
  const int N = p.h * p.w, pix = yc * p.w + xc;
  const int j0 = grp * G, nj = min(G, p.D - j0);

  float cur[16];
#pragma unroll
  for (int ch = 0; ch < 16; ++ch) cur[ch] = p.cur[((size_t)b * 16 + ch) * N + pix];

  const float* iK = p.invK + 16 * (size_t)b;
  float r0, r1, r2;
  {
#pragma clang fp contract(off)
    const float px = (float)xc + 0.5f, py = (float)yc + 0.5f;  // geometry_utils.py:34-44
    r0 = iK[0] * px + iK[1] * py + iK[2];
    r1 = iK[4] * px + iK[5] * py + iK[6];
    r2 = iK[8] * px + iK[9] * py + iK[10];
  }
  const float* planes = p.planes.ptr + b * p.planes.sb + yc * p.planes.sy + xc * p.planes.sx;
  float d[G], cost[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    d[g] = planes[(int64_t)(j0 + min(g, nj - 1)) * p.planes.sd];
    cost[g] = 0.0f;
  }
  const float* geom_b = p.geom + (size_t)b * p.K * SR_GEOM_STRIDE;
  const float* src_b = p.src_nhwc + (size_t)b * p.K * N * 16;
  const bool last_group = j0 + nj == p.D;
  bool any_depth = false, any_bounds = false;

  // Wave-level culling (planes shared by all pixels only): lanes 0..7 carry the 4 corners of this wave's 2 x 32 pixel
  // strip at the group's first and last plane.  The sampling position is a projective image of (pixel, depth) whose
  // denominator z' is bilinear in them, so when z' > 0 at the 8 corners every sample of the strip lies in the convex
  // hull of the 8 projected corners: if that hull misses the source image (with a margin far above the fp32 error of
  // the positions), or if the whole strip is behind the source camera, no sample of the wave contributes to this view.
  const bool can_cull = cull != 0 && !(last_group && p.out.mask != nullptr) && ty * LT_H + 2 * c.wave < p.h;
  float cr0 = 0.f, cr1 = 0.f, cr2 = 0.f, cd = 1.f;
  if (can_cull) {
#pragma clang fp contract(off)
    const int cxp = (c.lane & 1) ? min(tx * LT_W + LT_W - 1, p.w - 1) : tx * LT_W;
    const int cyp = (c.lane & 2) ? min(ty * LT_H + 2 * c.wave + 1, p.h - 1) : ty * LT_H + 2 * c.wave;
    const float px = (float)cxp + 0.5f, py = (float)cyp + 0.5f;
    cr0 = iK[0] * px + iK[1] * py + iK[2];
    cr1 = iK[4] * px + iK[5] * py + iK[6];
    cr2 = iK[8] * px + iK[9] * py + iK[10];
    cd = (c.lane & 4) ? planes[(int64_t)(j0 + nj - 1) * p.planes.sd] : planes[(int64_t)j0 * p.planes.sd];
  }
  const bool row_oob = ty * LT_H + 2 * c.wave >= p.h;  // the whole strip is below the image: nothing to do

#pragma unroll 1
  for (int k = 0; k < p.K; ++k) {
    unsigned xy[G];
    sr_f2 wt[G / 2][4];
    const float* g_k = geom_b + k * SR_GEOM_STRIDE;
    bool culled = row_oob;
    if (can_cull) {
      float X0, X1, X2;
      {
#pragma clang fp contract(off)
        X0 = cd * cr0; X1 = cd * cr1; X2 = cd * cr2;
      }
      SrSampleXY s;
      sr_project_sample_xy(g_k, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, s);
      const unsigned long long m8 = 0xFFull;
      const bool behind = (__ballot(s.zp < -1e-3f) & m8) == m8;
      const bool front = (__ballot(s.zp > 1e-3f) & m8) == m8;
      const bool left = (__ballot(s.pix_x < -1.0f) & m8) == m8, right = (__ballot(s.pix_x > (float)p.w + 1.0f) & m8) == m8;
      const bool above = (__ballot(s.pix_y < -1.0f) & m8) == m8, below = (__ballot(s.pix_y > (float)p.h + 1.0f) & m8) == m8;
      culled = behind | (front & (left | right | above | below));
    }
    if (culled) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        xy[g] = SR_NOREL;
        wt[g / 2][0] = wt[g / 2][1] = wt[g / 2][2] = wt[g / 2][3] = sr_f2{0.0f, 0.0f};
      }
    } else {
#pragma unroll
      for (int g = 0; g < G; g += 2) {
        SrPair s;
        sr_project_pair(g_k, r0, r1, r2, sr_f2{d[g], d[g + 1]}, p.h, p.w, p.inv_w, p.inv_h, s);
        const bool f0 = s.zp.x > 0.0f, f1 = s.zp.y > 0.0f;  // cost_volume.py:231-232
        if (last_group) {  // plane D-1 (get_mask, cost_volume.py:77-97)
          const bool b0 = (s.pix_x.x > 2.0f) & (s.pix_x.x < (float)(p.w - 2)) & (s.pix_y.x > 2.0f) & (s.pix_y.x < (float)(p.h - 2));
          const bool b1 = (s.pix_x.y > 2.0f) & (s.pix_x.y < (float)(p.w - 2)) & (s.pix_y.y > 2.0f) & (s.pix_y.y < (float)(p.h - 2));
          if (g == nj - 1) { any_depth |= f0; any_bounds |= b0; }
          if (g + 1 == nj - 1) { any_depth |= f1; any_bounds |= b1; }
        }
        sr_pair_taps(s, active & (g < nj) & f0, active & (g + 1 < nj) & f1, p.h, p.w, xy[g], xy[g + 1], wt[g / 2]);
      }
    }
    c.img = src_b + (size_t)k * N * 16;
    sr_lds_unit<CAP, G, 0, G>(c, xy, wt, cur, cost);
  }

  if (active) {
    float* out = p.out.cv + b * p.out.sb + (int64_t)pix * p.out.sp;
    float best = 0.0f, best_d = 0.0f;
    int best_j = 0;
    bool have = false;
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (g < nj) {
        out[(int64_t)(j0 + g) * p.out.sd] = cost[g];
        if (!have || cost[g] > best || (cost[g] != cost[g] && best == best)) {  // first max wins; NaN = max
          best = cost[g]; best_d = d[g]; best_j = j0 + g; have = true;
        }
      }
    if (p.out.lowest) {
      if (groups == 1) {
        p.out.lowest[(size_t)b * N + pix] = best_d;
      } else {
        unsigned u = __float_as_uint(best);
        u = (u >> 31) ? ~u : (u | 0x80000000u);  // order-preserving map of finite floats and infinities
        if (best != best) u = 0xFFFFFFFFu;
        const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)best_j);
        atomicMax(keys + (size_t)b * N + pix, key);
      }
    }
    if (last_group && p.out.mask) p.out.mask[(size_t)b * N + pix] = (uint8_t)(any_depth && any_bounds);
  }
}

__global__ __launch_bounds__(256) void sr_dot_lowest_from_keys_kernel(const unsigned long long* __restrict__ keys,
                                                                     SrPlanes planes, int h, int w, int B,
                                                                     float* __restrict__ lowest) {
  const int N = h * w;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * N) return;
  const int b = (int)(i / N), pix = (int)(i - (size_t)b * N);
  const int y = pix / w, x = pix - y * w;
  const unsigned j = 0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull);
  lowest[i] = planes.ptr[b * planes.sb + (int64_t)j * planes.sd + y * planes.sy + x * planes.sx];
}

template <int CAP, int WPS, int G>
static void sr_launch_lds_variant(const SrDotParams& p, unsigned long long* keys, int tiles_x, int tiles_y, int cull,
                                  hipStream_t stream) {
  const int groups = (p.D + G - 1) / G;
  const long nwg = (long)p.B * tiles_x * tiles_y * groups;
  hipLaunchKernelGGL((sr_dot_volume_lds_kernel<CAP, WPS, G>), dim3((unsigned)nwg), dim3(256), 0, stream, p, keys,
                     tiles_x, tiles_y, groups, cull);
}

int sr_launch_dot_volume_lds(const SrDotParams& p, unsigned long long* keys, hipStream_t stream) {
  if (p.w > 65000 || p.h > 65000) return SR_ERR_UNSUPPORTED;
  // switches (option table): LDS buffer in texels <-> workgroups per CU (634: 4, 770: 3), planes per group (2 / 4 / 8), hull
  // culling on / off
  const int Genv = sr_opt(SR_OPT_DOT_LDS_G);
  const int cull_env = sr_opt(SR_OPT_DOT_LDS_CULL);
  const int cap = sr_opt(SR_OPT_DOT_LDS_CAP);
  const int tiles_x = (p.w + LT_W - 1) / LT_W, tiles_y = (p.h + LT_H - 1) / LT_H;
  int G = Genv;
  if (G != 2 && G != 4 && G != 8)  // small grids (batch 1): 2-plane groups give twice the workgroups to balance 256 CUs
    G = ((long)p.B * tiles_x * tiles_y * ((p.D + 3) / 4) < 2048) ? 2 : 4;
  const int groups = (p.D + G - 1) / G;
  const long nwg = (long)p.B * tiles_x * tiles_y * groups;
  if (nwg > 0x7FFFFFFFL) return SR_ERR_UNSUPPORTED;
  const size_t N = (size_t)p.h * p.w;
  if (p.out.lowest && groups > 1) {
    hipError_t e = hipMemsetAsync(keys, 0, (size_t)p.B * N * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return sr_hip_rc(e);
  }
  const int cull = (cull_env != 0 && p.planes.sy == 0 && p.planes.sx == 0) ? 1 : 0;  // hull argument needs shared planes
#define SR_LDS_CASE(CAP_, WPS_)                                                              \
  switch (G) {                                                                               \
    case 2: sr_launch_lds_variant<CAP_, WPS_, 2>(p, keys, tiles_x, tiles_y, cull, stream); break; \
    case 8: sr_launch_lds_variant<CAP_, WPS_, 8>(p, keys, tiles_x, tiles_y, cull, stream); break; \
    default: sr_launch_lds_variant<CAP_, WPS_, 4>(p, keys, tiles_x, tiles_y, cull, stream); break; \
  }
  switch (cap) {
    case 770: SR_LDS_CASE(770, 3) break;
    case 506: SR_LDS_CASE(506, 5) break;
    default: SR_LDS_CASE(634, 4) break;
  }
#undef SR_LDS_CASE
  int rc = sr_hip_rc(hipGetLastError());
  if (rc == SR_OK && p.out.lowest && groups > 1) {
    const size_t total = (size_t)p.B * N;
    hipLaunchKernelGGL(sr_dot_lowest_from_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, keys,
                       p.planes, p.h, p.w, p.B, p.out.lowest);
    rc = sr_hip_rc(hipGetLastError());
  }
  return rc;
}

// Self-test hook: out_fast[i] = the packed reciprocal of the staged sweep, out_div[i] = the compiler's IEEE 1.0f / x[i]
// (tests/test_gpu_dot_volume.py checks them bit for bit over the range the sweep uses the fast form for).
__global__ void sr_selftest_rcp_kernel(const float* __restrict__ x, float* __restrict__ out_fast,
                                       float* __restrict__ out_div, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const sr_f2 r = sr_rcp2_exact(sr_f2{x[i], x[n - 1 - i]});
  out_fast[i] = r.x;
  out_div[i] = 1.0f / x[i];
}

extern "C" int sr_selftest_rcp(const float* x, float* out_fast, float* out_div, int n, void* stream) {
  if (n < 0 || (n > 0 && (!x || !out_fast || !out_div))) return SR_ERR_INVALID_ARGUMENT;
  if (n == 0) return SR_OK;
  hipLaunchKernelGGL(sr_selftest_rcp_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out_fast,
                     out_div, n);
  return sr_hip_rc(hipGetLastError());
}
