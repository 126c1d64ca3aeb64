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
#include <stdio.h>
#include <stdlib.h>

#include "sr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// tuning knobs: weight prefetch depth with one N-tile, residual prefetch in the last slab
#ifndef SR_NB1
#define SR_NB1 6
#endif
#ifndef SR_RES_PF
#define SR_RES_PF 0
#endif

// input channels per LDS slab: 16 for 3x3 convs (18 k-steps per slab), 64 for 1x1 convs (8 k-steps per
// slab instead of 2: a 1x1 slab of 16 channels is all barrier).  LDS rows carry 4 floats of padding:
// row strides of 20 and 68 floats are both conflict-free for the 16-byte A-fragment reads.
#ifndef SR_CK1
#define SR_CK1 64   // channels per slab of the 1x1 / stride-1 instantiation
#endif
#ifndef SR_CONV_BLOCKS_PER_CU
#define SR_CONV_BLOCKS_PER_CU 2
#endif
__host__ __device__ constexpr int sr_ck(int ks, int stride = 1) { return (ks == 1 && stride == 1) ? SR_CK1 : 16; }

// Phase-ablation switches (env SR_CONV_DEBUG) exist only in -DSR_CONV_ABLATION builds; in production they are compile-time 0,
// which keeps dead branches out of the hot loops (they cost registers: the MLP sweep spilled because of them).
#ifdef SR_CONV_ABLATION
#define SR_CV_DBG(bit) (p.debug & (bit))
#else
#define SR_CV_DBG(bit) 0
#endif

struct SrConvParams {
  const float* in; int64_t in_sb; int in_sp;        // batch stride, pixel stride (elements)
  const float* wp;                                  // packed weights [taps][G][2][Co_pad][4]
  const float* bias;                                // [Cout] or null
  const float* res; int64_t res_sb; int res_sp;     // residual (output geometry) or null
  float* out; int64_t out_sb; int out_sp;
  int H, W, Cin, Ho, Wo, Cout, Co_pad, G;           // G = 8-channel groups (even, zero padded)
  int tiles_x, tiles_y, co_blocks, total_tiles;
  float slope;                                      // >= 0: LeakyReLU slope; SR_ACT_NONE; SR_ACT_SILU
  int pad_y, pad_x;                                 // zero (or replicate) padding above / left of the image
  int vec4;                                         // input rows 16-byte aligned
  int debug;                                        // ablation bits (env SR_CONV_DEBUG), 0 in production
  int replicate;                                    // padding_mode="replicate": halo coordinates clamp to the border
  // split-K (1x1 convs with a long channel chain on few tiles): a work item covers 1/ksplit of the input slabs and
  // stores its raw partial sums (no bias / residual / activation) to part + ks * part_stride, dense channels-last
  // [B, Ho*Wo, Cout]; sr_launch_splitk_reduce() finishes.  ksplit = 1: off.
  int ksplit; float* part; int64_t part_stride;
};

// Stages global -> registers (issued before the MFMA phase of the previous slab) -> LDS (after it):
// the HBM/L2 latency of the next slab hides under the current slab's MFMAs (double-buffered tile).
// An M-tile (the 32 rows of one MFMA) is RM x CM output pixels (RM * CM = 32): 1 x 32 for wide maps,
// 4 x 8 for the low-resolution pyramid levels (less padding waste at widths 20 / 40).  A workgroup
// (4 waves x MT M-tiles) covers TH = 4*MT*RM rows x CM columns.
template <int KS, int S, int MT, int CM>
struct SrConvGeom {
  static constexpr int RM = 32 / CM;
  static constexpr int TH = 4 * MT * RM;
  static constexpr int HH = (TH - 1) * S + KS;
  static constexpr int HW = (CM - 1) * S + KS;
  static constexpr int CK = sr_ck(KS, S);              // channels per slab
  static constexpr int ROW = CK + 4;                   // floats per staged pixel
  static constexpr int Q = CK / 4;                     // float4 per staged pixel
  static constexpr int ELEMS = HH * HW * Q;            // float4 elements per slab
  static constexpr int PER_THREAD = (ELEMS + 255) / 256;
  static constexpr int TILE_FLOATS = HH * HW * ROW;
};

// Per-thread element offsets (in floats, relative to the image base; -1 = outside the image) of the
// slab elements this thread stages; fixed for a tile, so the per-slab work is one add + one load each.
template <int KS, int S, int MT, int CM>
__device__ __forceinline__ void sr_conv_stage_setup(const SrConvParams& p, int iy0, int ix0,
                                                    int (&offs)[SrConvGeom<KS, S, MT, CM>::PER_THREAD]) {
  using G = SrConvGeom<KS, S, MT, CM>;
#pragma unroll
  for (int it = 0; it < G::PER_THREAD; ++it) {
    const int e = threadIdx.x + it * 256;
    const int px = e / G::Q, q = e % G::Q;
    const int hy = px / G::HW, hx = px - hy * G::HW;
    int iy = iy0 + hy, ix = ix0 + hx;
    if (p.replicate) {
      iy = min(max(iy, 0), p.H - 1);
      ix = min(max(ix, 0), p.W - 1);
    }
    const bool ok = (e < G::ELEMS) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W) && !SR_CV_DBG(4);
    offs[it] = ok ? (iy * p.W + ix) * p.in_sp + 4 * q : -1;
  }
}

// VEC4 = true: rows are 16-byte aligned and Cin % 4 == 0 -> branch-free: every lane issues exactly one
// 16-byte load per element (out-of-image / out-of-range lanes read element 0 and are masked to zero).
template <int KS, int S, int MT, int CM, bool VEC4>
__device__ __forceinline__ void sr_conv_stage_load(const SrConvParams& p, const float* __restrict__ in_b, int c0,
                                                   const int (&offs)[SrConvGeom<KS, S, MT, CM>::PER_THREAD],
                                                   float4 (&stg)[SrConvGeom<KS, S, MT, CM>::PER_THREAD]) {
  using G = SrConvGeom<KS, S, MT, CM>;
#pragma unroll
  for (int it = 0; it < G::PER_THREAD; ++it) {
    const int c = c0 + 4 * ((threadIdx.x + it * 256) % G::Q);
    const bool ok = (offs[it] >= 0) && (c < p.Cin);
    const float* src = in_b + (ok ? offs[it] + c0 : 0);
    if (VEC4) {
      const float4 v = *reinterpret_cast<const float4*>(src);  // always a valid address; masked below
      stg[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        v.x = src[0];
        if (c + 1 < p.Cin) v.y = src[1];
        if (c + 2 < p.Cin) v.z = src[2];
        if (c + 3 < p.Cin) v.w = src[3];
      }
      stg[it] = v;
    }
  }
}

template <int KS, int S, int MT, int CM>
__device__ __forceinline__ void sr_conv_stage_store(float* __restrict__ tile,
                                                    const float4 (&stg)[SrConvGeom<KS, S, MT, CM>::PER_THREAD]) {
  using G = SrConvGeom<KS, S, MT, CM>;
#pragma unroll
  for (int it = 0; it < G::PER_THREAD; ++it) {
    const int e = threadIdx.x + it * 256;
    if (e < G::ELEMS) *reinterpret_cast<float4*>(&tile[(e / G::Q) * G::ROW + 4 * (e % G::Q)]) = stg[it];
  }
}

struct SrTileCoord { int b, co0, oy0, ox0, ks; };

template <int TH, int NT, int CM>
__device__ __forceinline__ SrTileCoord sr_conv_tile(const SrConvParams& p, int work) {
  // work = ((b * tiles_y + ty) * tiles_x + tx) * co_blocks + cb: the output-channel blocks of one pixel tile are
  // neighbours in the work order, so they run at about the same time and share the input tile through L2
  SrTileCoord t;
  t.ks = 0;
  if (p.ksplit > 1) { t.ks = work % p.ksplit; work /= p.ksplit; }   // the K-parts of a tile are neighbours too
  const int cb = work % p.co_blocks; work /= p.co_blocks;
  const int tx = work % p.tiles_x; work /= p.tiles_x;
  const int ty = work % p.tiles_y;
  t.b = work / p.tiles_y;
  t.co0 = cb * (32 * NT);
  t.oy0 = ty * TH;
  t.ox0 = tx * CM;
  return t;
}

// Persistent workgroups: each loops over output tiles, and the (tile, slab) sequence is ONE software
// pipeline -- the first slab of the next tile is fetched during the last slab of the current one and
// the epilogue stores drain while the next tile's MFMAs run, so there is no per-tile fill/drain.
template <int KS, int S, int MT, int NT, int CM, bool VEC4>
__global__ __launch_bounds__(256) void sr_conv_kernel(SrConvParams p) {
  using G = SrConvGeom<KS, S, MT, CM>;
  constexpr int TH = G::TH, HW = G::HW, RM = G::RM;
  constexpr int GPS = G::CK / 8;        // 8-channel groups per slab
  constexpr int STEPS = KS * KS * GPS;  // (tap, 8-channel group) steps per slab
  constexpr int ROW = G::ROW;
  __shared__ __attribute__((aligned(16))) float tiles[2][G::TILE_FLOATS];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 31, kk = lane >> 5;

  int a_off[MT];  // this lane's A-fragment base inside the halo tile (M-tile m = output row wave*MT + m)
#pragma unroll
  for (int m = 0; m < MT; ++m)
    a_off[m] = ((((wave * MT + m) * RM + i / CM) * S) * HW + (i % CM) * S) * ROW + 4 * kk;
  const int64_t rec = (int64_t)2 * p.Co_pad;  // float4 per (tap, g) weight record
  const int chunks = p.G / GPS;

  // Weight (B) fragments stream from L2 with a prefetch distance of PD steps through NB rotating
  // register sets.  VMEM returns in order, so the slab staging loads are issued when the next PD
  // steps' weights are already in flight: nothing younger than them is needed for >= PD steps.
  // (deeper with one N-tile, where a buffer is 4 VGPRs: 5 steps ~ 2.5-5k cycles also cover the residual
  // prefetch of the epilogue, which is issued in front of the last slab's weight stream)
  constexpr int NB = (NT == 1) ? (STEPS % 6 == 0 ? SR_NB1 : STEPS % 4 == 0 ? 4 : STEPS % 3 == 0 ? 3 : 2)
                               : (STEPS % 3 == 0 ? 3 : 2);
  constexpr int PD = NB - 1;
  static_assert(STEPS % NB == 0 && PD < STEPS + 1, "rotating buffers must line up across slabs");
  constexpr bool RES_PF = (MT * NT <= 2) && SR_RES_PF;  // residual values of the tile prefetched during its last slab
  float4 b_f[NB][NT], a_f[2][MT], stg[G::PER_THREAD];
  int offs[G::PER_THREAD];

  int work = blockIdx.x;
  if (work >= p.total_tiles) return;
  SrTileCoord t = sr_conv_tile<TH, NT, CM>(p, work);
  const float4* wp4 = reinterpret_cast<const float4*>(p.wp) + (kk * p.Co_pad + t.co0 + i);
  auto load_b = [&](const float4* base, int ch, int s, float4 (&dst)[NT]) {
    const int tap = s / GPS, g = s % GPS;
    const float4* wrec = base + (int64_t)(tap * p.G + GPS * ch + g) * rec;
#pragma unroll
    for (int n = 0; n < NT; ++n) dst[n] = wrec[32 * n];
  };

  // slab range [c_beg, c_end) of this work item (all of them unless split-K)
  int c_beg = t.ks * chunks / p.ksplit, c_end = (t.ks + 1) * chunks / p.ksplit;
  sr_conv_stage_setup<KS, S, MT, CM>(p, t.oy0 * S - p.pad_y, t.ox0 * S - p.pad_x, offs);
  sr_conv_stage_load<KS, S, MT, CM, VEC4>(p, p.in + (int64_t)t.b * p.in_sb, c_beg * G::CK, offs, stg);
  sr_conv_stage_store<KS, S, MT, CM>(tiles[0], stg);
#pragma unroll
  for (int s = 0; s < PD; ++s) load_b(wp4, c_beg, s, b_f[s]);
  __syncthreads();
  int buf = 0;
  const bool partial = p.ksplit > 1;

  while (true) {
    const int next_work = work + gridDim.x;
    const bool have_next = next_work < p.total_tiles;
    SrTileCoord tn = t;
    if (have_next) tn = sr_conv_tile<TH, NT, CM>(p, next_work);
    const float4* wp4n = reinterpret_cast<const float4*>(p.wp) + (kk * p.Co_pad + tn.co0 + i);
    const int cn_beg = tn.ks * chunks / p.ksplit, cn_end = (tn.ks + 1) * chunks / p.ksplit;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    float rpf[RES_PF ? MT : 1][RES_PF ? NT : 1][16];  // prefetched residual values (RES_PF only)
    const float* __restrict__ resp = (p.res && !partial && !SR_CV_DBG(2)) ? p.res + (int64_t)t.b * p.res_sb : nullptr;

    for (int ch = c_beg; ch < c_end; ++ch) {
      const float* tile = tiles[buf];
      const bool last = ch + 1 == c_end;
      const bool more = !last || have_next;        // is there a following slab in the pipeline?
      if (more) {
        if (last) sr_conv_stage_setup<KS, S, MT, CM>(p, tn.oy0 * S - p.pad_y, tn.ox0 * S - p.pad_x, offs);  // next tile
        sr_conv_stage_load<KS, S, MT, CM, VEC4>(p, p.in + (int64_t)(last ? tn.b : t.b) * p.in_sb,
                                            (last ? cn_beg : ch + 1) * G::CK, offs, stg);
      }
      const float4* wnext = last ? wp4n : wp4;
      const int chn = last ? cn_beg : ch + 1;
      if (RES_PF && last) {
        // residual of THIS tile: 16 x MT x NT dword loads, branch-free (masked), 5 k-steps of slack before the
        // in-order VMEM queue is needed again
#pragma unroll
        for (int m = 0; m < (RES_PF ? MT : 0); ++m) {
          const int oyb = t.oy0 + (wave * MT + m) * RM;
          const int pixb = oyb * p.Wo + t.ox0 + 4 * kk;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int co = t.co0 + 32 * n + i;
            const unsigned rb = (unsigned)(pixb * p.res_sp + co);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (8 * (r >> 2)) / CM, colc = (8 * (r >> 2)) % CM + (r & 3);
              const bool ok = (resp != nullptr) & (co < p.Cout) & (oyb + row < p.Ho) & (t.ox0 + colc + 4 * kk < p.Wo);
              const float v = (resp ? resp : p.in)[ok ? rb + (unsigned)((row * p.Wo + colc) * p.res_sp) : 0u];
              rpf[m][n][r] = ok ? v : 0.0f;
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) a_f[0][m] = *reinterpret_cast<const float4*>(&tile[a_off[m]]);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int cb = s % NB, ca = s & 1;
        // request: weights of step s+PD (possibly of the next slab), LDS fragments of step s+1
        if (s + PD < STEPS) load_b(wp4, ch, s + PD, b_f[(s + PD) % NB]);
        else if (more) load_b(wnext, chn, s + PD - STEPS, b_f[(s + PD) % NB]);
        if (s + 1 < STEPS) {
          const int tap = (s + 1) / GPS, g = (s + 1) % GPS;
          const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            a_f[ca ^ 1][m] = *reinterpret_cast<const float4*>(&tile[a_off[m] + (ky * HW + kx) * ROW + 8 * g]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // k-step outer, accumulators inner: consecutive MFMAs hit different accumulators
#define SR_CONV_KSTEP(E)                                                                                       \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                         \
          _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                       \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca][m].E, b_f[cb][n].E, acc[m][n], 0, 0, 0);
        SR_CONV_KSTEP(x) SR_CONV_KSTEP(y) SR_CONV_KSTEP(z) SR_CONV_KSTEP(w)
#undef SR_CONV_KSTEP
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) sr_conv_stage_store<KS, S, MT, CM>(tiles[buf ^ 1], stg);
      __syncthreads();
      buf ^= 1;
    }

    // ---- epilogue of tile t: C[row = pixel j of the M-tile, col = output channel] ----
    // j = (r&3) + 8*(r>>2) + 4*kk  ->  M-tile row (8*(r>>2))/CM, column (8*(r>>2))%CM + (r&3) + 4*kk: only the
    // 4*kk part is per lane, the rest folds to immediates.  32-bit element offsets off one scalar base per
    // tensor; all residual loads of a fragment are issued before its stores (out / residual may alias).
    {
      float* __restrict__ outp = partial ? p.part + t.ks * p.part_stride + (int64_t)t.b * p.Ho * p.Wo * p.Cout
                                         : p.out + (int64_t)t.b * p.out_sb;
      const int osp = partial ? p.Cout : p.out_sp, rsp = p.res_sp;
      const float slope = sr_uniform(partial ? SR_ACT_NONE : p.slope);   // scalar: tested once per 16-value fragment
      const bool no_res = (resp == nullptr);
      // interior tiles (workgroup-uniform test) take a branch-free path
      const bool full = (t.oy0 + TH <= p.Ho) && (t.ox0 + CM <= p.Wo) && (t.co0 + 32 * NT <= p.Cout) && !SR_CV_DBG(1);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oyb = t.oy0 + (wave * MT + m) * RM;
        const int pixb = oyb * p.Wo + t.ox0 + 4 * kk;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = t.co0 + 32 * n + i;
          const bool okc = co < p.Cout;
          const float bv = (p.bias && okc && !partial) ? p.bias[co] : 0.0f;
          const unsigned ob = (unsigned)(pixb * osp + co), rb = (unsigned)(pixb * rsp + co);
          float rv[16];
          if (RES_PF) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = rpf[RES_PF ? m : 0][RES_PF ? n : 0][r];
          }
          if (full) {
            if (RES_PF) {
            } else if (!no_res) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = (8 * (r >> 2)) / CM, colc = (8 * (r >> 2)) % CM + (r & 3);
                rv[r] = resp[rb + (unsigned)((row * p.Wo + colc) * rsp)];
              }
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) rv[r] = 0.0f;
            }
            float o16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) o16[r] = acc[m][n][r] + bv + rv[r];
            sr_activate_group(o16, slope);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (8 * (r >> 2)) / CM, colc = (8 * (r >> 2)) % CM + (r & 3);
              outp[ob + (unsigned)((row * p.Wo + colc) * osp)] = o16[r];
            }
          } else {
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (8 * (r >> 2)) / CM, colc = (8 * (r >> 2)) % CM + (r & 3);
              ok[r] = okc && (oyb + row < p.Ho) && (t.ox0 + colc + 4 * kk < p.Wo);
              if (!RES_PF) rv[r] = (!no_res && ok[r]) ? resp[rb + (unsigned)((row * p.Wo + colc) * rsp)] : 0.0f;
            }
            float o16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) o16[r] = acc[m][n][r] + bv + rv[r];
            sr_activate_group(o16, slope);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (8 * (r >> 2)) / CM, colc = (8 * (r >> 2)) % CM + (r & 3);
              if (ok[r] && (!SR_CV_DBG(1) || o16[r] == 1.2345e33f)) outp[ob + (unsigned)((row * p.Wo + colc) * osp)] = o16[r];
            }
          }
        }
      }
    }
    if (!have_next) break;
    work = next_work;
    t = tn;
    c_beg = cn_beg; c_end = cn_end;
    wp4 = wp4n;
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

// The same operator, one thread per 2 x 2 block of OUTPUT pixels x 4 channels (r05).  Output rows 2y + 1, 2y + 2 and columns
// 2x + 1, 2x + 2 interpolate between the same four input pixels (y, y + 1) x (x, x + 1) with weights 3/4, 1/4 (rows / columns
// 0 and 2H - 1 / 2W - 1: the clamped border, blocks y = -1, x = -1 and y = H - 1, x = W - 1 hold one row / column): 4 loads
// for 4 stores instead of 16, one index computation instead of four.  Per output the source positions and weights come from
// the same expressions as in sr_upsample2x_kernel (same rounding, identical results).  Grid: y = (image, block row), x = the
// (W + 1) * cq threads of a block row; write-bound: 4 bytes written per byte read.
__global__ __launch_bounds__(256) void sr_upsample2x_quad_kernel(const float* __restrict__ in, int64_t in_sb, int in_sp,
                                                                 float* __restrict__ out, int64_t out_sb, int out_sp, int B,
                                                                 int H, int W, int cq) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (W + 1) * cq) return;
  const int q = t % cq, bx = t / cq - 1;
  const int xa = min(max(bx, 0), W - 1), xb = min(xa + 1, W - 1);
  float lx[2], hx[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const float sx = fmaxf(((float)(2 * bx + 1 + r) + 0.5f) * 0.5f - 0.5f, 0.0f);
    lx[r] = sx - (float)xa;
    hx[r] = 1.0f - lx[r];
  }
  for (int row = blockIdx.y; row < B * (H + 1); row += gridDim.y) {
    const int b = row / (H + 1), by = row - b * (H + 1) - 1;   // (scalar)
    const int ya = min(max(by, 0), H - 1), yb = min(ya + 1, H - 1);
    const float* ib = in + (int64_t)b * in_sb + 4 * q;
    const float4 a = *reinterpret_cast<const float4*>(ib + ((int64_t)ya * W + xa) * in_sp);
    const float4 bq = *reinterpret_cast<const float4*>(ib + ((int64_t)ya * W + xb) * in_sp);
    const float4 c = *reinterpret_cast<const float4*>(ib + ((int64_t)yb * W + xa) * in_sp);
    const float4 d = *reinterpret_cast<const float4*>(ib + ((int64_t)yb * W + xb) * in_sp);
    float* ob = out + (int64_t)b * out_sb + 4 * q;
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
      const int oy = 2 * by + 1 + ry;
      if (oy < 0 || oy >= Ho) continue;   // (uniform)
      const float sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.0f);
      const float ly = sy - (float)ya, hy = 1.0f - ly;
#pragma unroll
      for (int rx = 0; rx < 2; ++rx) {
        const int ox = 2 * bx + 1 + rx;
        if (ox < 0 || ox >= Wo) continue;
        float4 r;
        r.x = hy * (hx[rx] * a.x + lx[rx] * bq.x) + ly * (hx[rx] * c.x + lx[rx] * d.x);
        r.y = hy * (hx[rx] * a.y + lx[rx] * bq.y) + ly * (hx[rx] * c.y + lx[rx] * d.y);
        r.z = hy * (hx[rx] * a.z + lx[rx] * bq.z) + ly * (hx[rx] * c.z + lx[rx] * d.z);
        r.w = hy * (hx[rx] * a.w + lx[rx] * bq.w) + ly * (hx[rx] * c.w + lx[rx] * d.w);
        *reinterpret_cast<float4*>(ob + ((int64_t)oy * Wo + ox) * out_sp) = r;
      }
    }
  }
}

// depth = exp(log_depth) (reference depth_model.py:392-400), elementwise
__global__ __launch_bounds__(256) void sr_exp_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = expf(in[i]);
}

// ------------------------------------------------------------------ C ABI -------------

extern "C" int sr_exp_fwd(const float* in, float* out, int64_t n, void* stream_) {
  if (n < 0) return SR_ERR_INVALID_ARGUMENT;
  if (n == 0) return SR_OK;
  if (!in || !out) return SR_ERR_INVALID_ARGUMENT;
  const int64_t blocks = (n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048;
  hipLaunchKernelGGL(sr_exp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, in, out, n);
  return sr_hip_rc(hipGetLastError());
}

extern "C" size_t sr_conv_packed_weight_floats(int Cout, int Cin, int ksize) {
  if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
  const int ck = sr_ck(ksize);
  const size_t G = (size_t)((Cin + ck - 1) / ck) * (ck / 8), Co_pad = (size_t)((Cout + 31) / 32) * 32;
  return (size_t)ksize * ksize * G * 2 * Co_pad * 4;
}

extern "C" int sr_conv_pack_weights(const float* weight, int Cout, int Cin, int ksize, float* packed, void* stream_) {
  if (!weight || !packed || Cout <= 0 || Cin <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (ksize != 1 && ksize != 3) return SR_ERR_UNSUPPORTED;
  const int ck = sr_ck(ksize);
  const int G = ((Cin + ck - 1) / ck) * (ck / 8), Co_pad = ((Cout + 31) / 32) * 32;
  hipLaunchKernelGGL(sr_conv_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream_, weight, packed, Cout, Cin,
                     ksize * ksize, G, Co_pad);
  return sr_hip_rc(hipGetLastError());
}

static int sr_num_cus() { return sr_device_cus(); }

template <int KS, int S, int MT, int CM>
static int sr_conv_launch(SrConvParams& p, int B, int nt, hipStream_t stream) {
  using G = SrConvGeom<KS, S, MT, CM>;
  p.tiles_x = (p.Wo + CM - 1) / CM;
  p.tiles_y = (p.Ho + G::TH - 1) / G::TH;
  p.co_blocks = (p.Co_pad + 32 * nt - 1) / (32 * nt);
  p.total_tiles = p.tiles_x * p.tiles_y * p.co_blocks * B;
  if (p.ksplit > 1) {   // split-K plan (1x1 / stride 1 only, see sr_conv2d_dispatch): fill the 2-per-CU slots once
    const int chunks = p.G / (G::CK / 8);
    int ks = (2 * sr_num_cus()) / p.total_tiles;
    // 3x3 / stride 2: a work item is 9 taps deep and the finish is a second launch -- measured (scripts/conv_s2_sweep.py): 72 items x
    // 16 slabs 56 -> 28 us split 8 ways, 150 items x 4 slabs 19 -> 28 us split in two, 144 items at batch 8 94 -> 102 us split in three
    if (KS == 3 && 2 * p.total_tiles > sr_num_cus()) ks = 1;
    if (ks > chunks / 2) ks = chunks / 2;
    if (ks > p.ksplit) ks = p.ksplit;
    p.ksplit = ks < 2 ? 1 : ks;
    p.total_tiles *= p.ksplit;
  }
  int blocks = sr_num_cus() * (KS == 1 ? SR_CONV_BLOCKS_PER_CU : 2);  // persistent grid: workgroups per CU (register / LDS limit)
  if (blocks > p.total_tiles) blocks = p.total_tiles;
  dim3 grid(blocks), block(256);
  const bool v4 = p.vec4 && (p.Cin % 4 == 0);
  if (nt == 2 && v4) hipLaunchKernelGGL((sr_conv_kernel<KS, S, MT, 2, CM, true>), grid, block, 0, stream, p);
  else if (nt == 2) hipLaunchKernelGGL((sr_conv_kernel<KS, S, MT, 2, CM, false>), grid, block, 0, stream, p);
  else if (v4) hipLaunchKernelGGL((sr_conv_kernel<KS, S, MT, 1, CM, true>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((sr_conv_kernel<KS, S, MT, 1, CM, false>), grid, block, 0, stream, p);
  return sr_hip_rc(hipGetLastError());
}

// Tile-shape choice.  Tiles are indivisible and the MFMA pipes of a CU are shared by its resident
// workgroups, so the launch time is ~ (max tiles on one CU) x (MFMA work of a tile): minimise
//   ceil(tiles / CUs) * (MT * NT + per-tile overhead).
// Candidates: pixels 8x32 (MT=2), 4x32 (MT=1), 16x8 (MT=1, 4x8 M-tiles) x output channels 64 (NT=2) or 32 (NT=1).
// Padding waste (partial tiles) and tail quantisation both show up in the tile count.
struct SrConvCfg { int shape, nt; };
static SrConvCfg sr_conv_pick(const SrConvParams& p, int B, int stride, int ksize) {
  const int cus = sr_num_cus();
  // staging cost per pixel tile relative to its MFMA work: negligible for 3x3 (one slab feeds 9 taps), large for 1x1
  const double stage_w = ksize == 1 ? 0.6 : 0.0;
  const int th[3] = {8, 4, 16}, tw[3] = {32, 32, 8}, mt[3] = {2, 1, 1};
  SrConvCfg best = {1, 1};
  double best_cost = -1;
  for (int c = 0; c < 3; ++c) {
    if (stride == 2 && c == 0) continue;  // the 8x32 stride-2 halo does not fit LDS twice
    for (int nt = 2; nt >= 1; --nt) {
      if (nt == 2 && p.Co_pad % 64 != 0) continue;
      const long tiles = (long)((p.Wo + tw[c] - 1) / tw[c]) * ((p.Ho + th[c] - 1) / th[c]) *
                         ((p.Co_pad + 32 * nt - 1) / (32 * nt)) * B;
      // + 0.04*nt: weight fragments come from L2 (VMEM) per N-tile, A fragments from LDS -- measured: at equal
      // MFMA work an 8x32x32 tile is ~4 % faster than a 4x32x64 tile
      double rounds = (double)((tiles + cus - 1) / cus);
      // 1x1 convs with at most two of the SMALLEST tiles per CU are latency- rather than MFMA-bound: the second
      // resident workgroup of a CU costs ~35 % (measured: 960 -> 160 channels on 9600 pixels, 375 tiles of 4x32x32:
      // 57 us; 190 tiles of 8x32x32: 79 us).  Larger tiles do serialise on the MFMA pipe (128 -> 512 channels: 304
      // tiles of 8x32x64 50 us, 600 tiles of 4x32x64 33 us): they keep the whole-round count.
      if (ksize == 1 && mt[c] * nt == 1 && tiles > cus && tiles <= 2 * cus)
        rounds = 1.0 + 0.35 * (double)(tiles - cus) / cus;
      const double cost = rounds * (mt[c] * nt + 0.12 + 0.04 * nt + stage_w * mt[c]);
      if (best_cost < 0 || cost < best_cost) { best = {c, nt}; best_cost = cost; }
    }
  }
  return best;
}

static SrConvCfg sr_conv_cfg(const SrConvParams& p, int B, int stride, int ksize) {
  SrConvCfg cfg = sr_conv_pick(p, B, stride, ksize);
  const int v = sr_opt(SR_OPT_CONV_TILE);  // ablation override: shape + 10 * nt (nt = 0: keep); 0 = none
  if (v > 0) {
    const int shape = v % 10, nt = v / 10;
    if (shape >= 0 && shape <= 2 && !(stride == 2 && shape == 0)) cfg.shape = shape;
    if (nt == 1 || (nt == 2 && p.Co_pad % 64 == 0)) cfg.nt = nt;
  }
  return cfg;
}

// Split-K is planned for 1x1 / stride-1 convolutions (the deep projections of the image-prior encoder's MBConv
// blocks: 1536 -> 256 channels on 15x20 maps is 24 slabs in a row on a few dozen workgroups) and, since r06, for 3x3 /
// stride-2 convolutions on small maps (256 -> 384 onto a 15x20 map at batch 1: 72 workgroups x 16 slabs in a row, 56 us);
// the finishing kernel works on 16-byte channel quads.
#define SR_CONV_KSPLIT_MAX 8
static bool sr_conv_splitk_eligible(const float* out, int64_t out_sb, int out_sp, const float* bias, const float* res,
                                    int64_t res_sb, int res_sp, int Cin, int Cout, int ksize, int stride) {
  const int on = sr_opt(SR_OPT_CONV_KSPLIT);
  const bool pw = ksize == 1 && stride == 1 && Cin >= 4 * SR_CK1;
  const bool s2 = ksize == 3 && stride == 2 && Cin >= 8 * sr_ck(3);   // r06: the deep down-convolutions of CVEncoder at batch 1
  if (!on || !(pw || s2) || Cout % 4 != 0) return false;
  auto al = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al(out) || out_sp % 4 != 0 || out_sb % 4 != 0 || (bias && !al(bias))) return false;
  if (res && (!al(res) || res_sp % 4 != 0 || res_sb % 4 != 0)) return false;
  return true;
}

static int sr_conv2d_dispatch(const float* in, int64_t in_batch_stride, int in_pix_stride,
                             const float* packed_weight, const float* bias, const float* residual,
                             int64_t res_batch_stride, int res_pix_stride, float* out, int64_t out_batch_stride,
                             int out_pix_stride, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                             float leaky_slope, int replicate, void* stream_, const int* pads = nullptr,
                             void* workspace = nullptr, size_t workspace_bytes = 0) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !packed_weight || !out) return SR_ERR_INVALID_ARGUMENT;
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return SR_ERR_UNSUPPORTED;
  const int pad = ksize / 2;
  // pads = {top, left, bottom, right}; default: ksize / 2 on every side
  const int pt = pads ? pads[0] : pad, pl = pads ? pads[1] : pad, pb = pads ? pads[2] : pad, pr = pads ? pads[3] : pad;
  if (pt < 0 || pl < 0 || pb < 0 || pr < 0 || pt >= ksize || pl >= ksize || pb >= ksize || pr >= ksize ||
      H + pt + pb < ksize || W + pl + pr < ksize)
    return SR_ERR_INVALID_ARGUMENT;
  SrConvParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.wp = packed_weight; p.bias = bias;
  p.res = residual; p.res_sb = res_batch_stride; p.res_sp = res_pix_stride;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.Ho = (H + pt + pb - ksize) / stride + 1;
  p.Wo = (W + pl + pr - ksize) / stride + 1;
  p.pad_y = pt; p.pad_x = pl;
  p.Co_pad = ((Cout + 31) / 32) * 32;
  p.G = ((Cin + sr_ck(ksize) - 1) / sr_ck(ksize)) * (sr_ck(ksize) / 8);
  p.slope = leaky_slope;
  p.replicate = replicate;
  p.ksplit = 1; p.part = nullptr; p.part_stride = 0;
  if (workspace && sr_conv_splitk_eligible(out, out_batch_stride, out_pix_stride, bias, residual, res_batch_stride,
                                           res_pix_stride, Cin, Cout, ksize, stride)) {
    p.part_stride = (int64_t)B * p.Ho * p.Wo * Cout;
    const size_t fit = workspace_bytes / ((size_t)p.part_stride * sizeof(float));
    p.ksplit = fit < 2 ? 1 : (fit < SR_CONV_KSPLIT_MAX ? (int)fit : SR_CONV_KSPLIT_MAX);   // upper bound: the launch plans
    p.part = (float*)workspace;
  }
#ifdef SR_CONV_ABLATION   // (ablation builds only: the product library reads no environment on a launch path)
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("SR_CONV_DEBUG"); dbg = e ? atoi(e) : 0; } p.debug = dbg; }
#else
  p.debug = 0;
#endif
  p.vec4 = (((uintptr_t)in & 15) == 0) && (in_pix_stride % 4 == 0) && (in_batch_stride % 4 == 0);
  hipStream_t stream = (hipStream_t)stream_;
  const SrConvCfg cfg = sr_conv_cfg(p, B, stride, ksize);
  const int c = cfg.shape, nt = cfg.nt;
  int rc;
  if (ksize == 3 && stride == 1) {
    if (c == 0) rc = sr_conv_launch<3, 1, 2, 32>(p, B, nt, stream);
    else if (c == 1) rc = sr_conv_launch<3, 1, 1, 32>(p, B, nt, stream);
    else rc = sr_conv_launch<3, 1, 1, 8>(p, B, nt, stream);
  } else if (ksize == 3 && stride == 2) {
    if (c == 1) rc = sr_conv_launch<3, 2, 1, 32>(p, B, nt, stream);
    else rc = sr_conv_launch<3, 2, 1, 8>(p, B, nt, stream);
  } else if (ksize == 1 && stride == 1) {
    if (c == 0) rc = sr_conv_launch<1, 1, 2, 32>(p, B, nt, stream);
    else if (c == 1) rc = sr_conv_launch<1, 1, 1, 32>(p, B, nt, stream);
    else rc = sr_conv_launch<1, 1, 1, 8>(p, B, nt, stream);
  } else {
    if (c == 1) rc = sr_conv_launch<1, 2, 1, 32>(p, B, nt, stream);
    else rc = sr_conv_launch<1, 2, 1, 8>(p, B, nt, stream);
  }
  if (rc == SR_OK && p.ksplit > 1)   // the launch settled the final split factor in p.ksplit
    rc = sr_launch_splitk_reduce(p.part, p.ksplit, p.part_stride, bias, residual, res_batch_stride, res_pix_stride, out,
                                 out_batch_stride, out_pix_stride, B, p.Ho * p.Wo, Cout, leaky_slope, stream);
  return rc;
}

extern "C" int sr_conv2d_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                  const float* packed_weight, const float* bias, const float* residual,
                                  int64_t res_batch_stride, int res_pix_stride, float* out,
                                  int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                  int Cout, int ksize, int stride, float leaky_slope, void* stream_) {
  return sr_conv2d_dispatch(in, in_batch_stride, in_pix_stride, packed_weight, bias, residual, res_batch_stride,
                            res_pix_stride, out, out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, ksize, stride,
                            leaky_slope, 0, stream_);
}

extern "C" int sr_conv2d_replicate_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                            const float* packed_weight, const float* bias, const float* residual,
                                            int64_t res_batch_stride, int res_pix_stride, float* out,
                                            int64_t out_batch_stride, int out_pix_stride, int B, int H, int W,
                                            int Cin, int Cout, int ksize, int stride, float leaky_slope,
                                            void* stream_) {
  return sr_conv2d_dispatch(in, in_batch_stride, in_pix_stride, packed_weight, bias, residual, res_batch_stride,
                            res_pix_stride, out, out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, ksize, stride,
                            leaky_slope, 1, stream_);
}

extern "C" int sr_conv2d_padded_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                         const float* packed_weight, const float* bias, const float* residual,
                                         int64_t res_batch_stride, int res_pix_stride, float* out,
                                         int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                         int Cout, int ksize, int stride, int pad_top, int pad_left, int pad_bottom,
                                         int pad_right, float leaky_slope, void* stream_) {
  const int pads[4] = {pad_top, pad_left, pad_bottom, pad_right};
  return sr_conv2d_dispatch(in, in_batch_stride, in_pix_stride, packed_weight, bias, residual, res_batch_stride,
                            res_pix_stride, out, out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, ksize, stride,
                            leaky_slope, 0, stream_, pads);
}

extern "C" size_t sr_conv_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout % 4 != 0) return 0;
  const bool pw = ksize == 1 && stride == 1 && Cin >= 4 * SR_CK1;
  const bool s2 = ksize == 3 && stride == 2 && Cin >= 8 * sr_ck(3);
  if (!pw && !s2) return 0;
  // small pixel counts only: with >= 2 work items per CU already (1x1) / one per CU (3x3: a work item is 9x the MFMAs) there is
  // nothing to gain
  const int64_t px = s2 ? (int64_t)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) : (int64_t)B * H * W;
  const int64_t tiles = ((px + 127) / 128) * ((Cout + 31) / 32);
  if (s2 ? 2 * tiles > sr_num_cus() : tiles >= 2 * sr_num_cus()) return 0;
  return (size_t)SR_CONV_KSPLIT_MAX * px * Cout * sizeof(float);
}

extern "C" int sr_conv2d_splitk_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                         const float* packed_weight, const float* bias, const float* residual,
                                         int64_t res_batch_stride, int res_pix_stride, float* out,
                                         int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                         int Cout, int ksize, int stride, float leaky_slope, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
  return sr_conv2d_dispatch(in, in_batch_stride, in_pix_stride, packed_weight, bias, residual, res_batch_stride,
                            res_pix_stride, out, out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, ksize, stride,
                            leaky_slope, 0, stream_, nullptr, workspace, workspace_bytes);
}

// Symbol of the kernel instantiation sr_conv2d_nhwc_fwd picks for these arguments (for profilers / bench).
extern "C" const char* sr_conv_kernel_name(int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                                           int aligned16) {
  static thread_local char buf[96];
  SrConvParams p;
  const int pad = ksize / 2;
  p.Ho = (H + 2 * pad - ksize) / stride + 1;
  p.Wo = (W + 2 * pad - ksize) / stride + 1;
  p.Co_pad = ((Cout + 31) / 32) * 32;
  const SrConvCfg cfg = sr_conv_cfg(p, B, stride, ksize);
  const int mt = cfg.shape == 0 ? 2 : 1, cm = cfg.shape == 2 ? 8 : 32, nt = cfg.nt;
  snprintf(buf, sizeof(buf), "sr_conv_kernel<%d, %d, %d, %d, %d, %s>", ksize, stride, mt, nt, cm,
           (aligned16 && Cin % 4 == 0) ? "true" : "false");
  return buf;
}

extern "C" int sr_upsample2x_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out,
                                      int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C,
                                      void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !out) return SR_ERR_INVALID_ARGUMENT;
  const int vec4 = (((uintptr_t)in & 15) == 0) && (((uintptr_t)out & 15) == 0) && (in_pix_stride % 4 == 0) &&
                   (out_pix_stride % 4 == 0) && (in_batch_stride % 4 == 0) && (out_batch_stride % 4 == 0);
  if (vec4 && C % 4 == 0 && sr_opt(SR_OPT_UPSAMPLE_QUAD)) {
    const int cq = C / 4;
    const int64_t rows = (int64_t)B * (H + 1);
    hipLaunchKernelGGL(sr_upsample2x_quad_kernel, dim3(((W + 1) * cq + 255) / 256, (unsigned)(rows < 65535 ? rows : 65535)),
                       dim3(256), 0, (hipStream_t)stream_, in, in_batch_stride, in_pix_stride, out, out_batch_stride,
                       out_pix_stride, B, H, W, cq);
    return sr_hip_rc(hipGetLastError());
  }
  const int64_t total = (int64_t)4 * H * W * ((C + 3) / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sr_upsample2x_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream_, in, in_batch_stride,
                     in_pix_stride, out, out_batch_stride, out_pix_stride, H, W, C, vec4);
  return sr_hip_rc(hipGetLastError());
}
