// sr_matching.hip -- matching-feature encoder kernels (reference
// modules/networks.py:149-205 ResnetMatchingEncoder): the antialiased ResNet-18 stem
//   conv1 7x7/s2 (+ BatchNorm affine + ReLU)            -> sr_stem_kernel        (fp32 MFMA implicit GEMM)
//   MaxPool2d(2, stride 1) + BlurPool(filt 4, stride 2) -> sr_maxblurpool_kernel (HBM-bound)
// and the InstanceNorm2d (+ LeakyReLU) of the tail       -> sr_inorm_* kernels    (HBM-bound, deterministic).
// layer1 and the tail convolutions run on the conv kernels of sr_conv.hip / sr_wino.hip.
// gfx950 only.
#include "sr_common.h"

typedef float v16f __attribute__((ext_vector_type(16)));
#define SR_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------ stem: conv 7x7 / stride 2 / pad 3, Cin = 3 ---
//
// Implicit GEMM  out[pixel, co] = sum_k patch[pixel, k] * W[k, co]  with k = c*49 + ky*7 + kx -- 147 taps packed
// densely into 74 MFMA k-steps of 2 (the k-lane `half` takes tap 2*step + half; tap 147 is a zero pad).
// A workgroup (4 waves) owns a 16x16 output tile; its 37x37x3 input patch sits in LDS and every A fragment is a
// single ds_read_b32 -- at an immediate offset when the two taps of a step are neighbours in a patch row, through a
// per-lane select in the 11 steps that straddle a row / channel end (stride-2 columns: 64 lanes hit each bank twice =
// the 2-cycle minimum).  The packed weight (37 KB) stays in LDS for the whole (persistent) workgroup:
// [step][half][32 cols][2 N-tiles]; two workgroups share a CU (three fit the LDS but spill registers: measured slower).
// Wave w computes output rows 4w..4w+3 (two 2x16 M-tiles) x 64 channels: 4 MFMAs per 3 LDS reads.
#define SR_STEM_T 16
#define SR_STEM_PR 37
#define SR_STEM_PW 37
#define SR_STEM_STEPS 74
#define SR_STEM_WFLOATS (SR_STEM_STEPS * 128)
#define SR_STEM_PFLOATS (3 * SR_STEM_PR * SR_STEM_PW)

struct SrStemParams {
  const float* in; int64_t sb, sc, sy, sx;   // image strides (elements): batch, channel, row, column
  const float* wp;                           // packed weights
  const float* scale; const float* shift;    // per-channel affine after the conv (BatchNorm eval), may be null
  float slope;                               // LeakyReLU slope (0 = ReLU), < 0: none
  float* out; int64_t out_sb; int out_sp;    // channels-last output
  int H, W, Ho, Wo, tiles_x, tiles_y, total;
};

__global__ void sr_stem_pack_kernel(const float* __restrict__ w /*[64,3,7,7]*/, float* __restrict__ packed) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < SR_STEM_WFLOATS; idx += gridDim.x * blockDim.x) {
    const int step = idx >> 7, r = idx & 127;
    const int half = r >> 6, col = (r & 63) >> 1, nt = r & 1;
    const int tap = 2 * step + half;  // = (c*7 + ky)*7 + kx: the [3,7,7] part of the weight, flattened
    const int co = nt * 32 + col;
    packed[idx] = tap < 147 ? w[co * 147 + tap] : 0.f;
  }
}

// patch offset of tap k (k = 147: the pad, any valid address)
__host__ __device__ constexpr int sr_stem_tap_offset(int k) {
  return k >= 147 ? 0 : ((k / 49) * SR_STEM_PR + (k % 49) / 7) * SR_STEM_PW + (k % 7);
}

__global__ __launch_bounds__(256, 2) void sr_stem_kernel(SrStemParams p) {
  __shared__ __attribute__((aligned(16))) float wl[SR_STEM_WFLOATS];
  __shared__ float patch[SR_STEM_PFLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;

  for (int e = threadIdx.x; e < SR_STEM_WFLOATS / 4; e += 256)
    reinterpret_cast<float4*>(wl)[e] = reinterpret_cast<const float4*>(p.wp)[e];

  // A-fragment bases: M-tile m of this wave = output rows 4*wave + 2*m + {0,1}, 16 columns
  int abase[2];  // without the k-lane term
#pragma unroll
  for (int m = 0; m < 2; ++m) abase[m] = (2 * (4 * wave + 2 * m + (i >> 4))) * SR_STEM_PW + 2 * (i & 15);
  const float2* wl2 = reinterpret_cast<const float2*>(wl) + half * 32 + i;

  float sc[2], sh[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    sc[n] = p.scale ? p.scale[n * 32 + i] : 1.f;
    sh[n] = p.shift ? p.shift[n * 32 + i] : 0.f;
  }

  // The next tile's patch is fetched into registers while the current tile's MFMAs run (one LDS patch buffer:
  // the stores happen between two barriers after the MFMA loop), so the global-load latency is off the critical path.
  constexpr int PF = (SR_STEM_PFLOATS + 255) / 256;
  float pf[PF];
  auto fetch = [&](int tile) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const float* __restrict__ img = p.in + (int64_t)b * p.sb;
#pragma unroll
    for (int it = 0; it < PF; ++it) {
      const int e = threadIdx.x + it * 256;
      const int c = e / (SR_STEM_PR * SR_STEM_PW), rem = e - c * (SR_STEM_PR * SR_STEM_PW);
      const int r = rem / SR_STEM_PW, x = rem - r * SR_STEM_PW;
      const int gy = 2 * ty * SR_STEM_T - 3 + r, gx = 2 * tx * SR_STEM_T - 3 + x;
      const bool ok = (e < SR_STEM_PFLOATS) & (gy >= 0) & (gy < p.H) & (gx >= 0) & (gx < p.W);
      const int64_t off = ok ? c * p.sc + gy * p.sy + gx * p.sx : 0;
      const float v = img[off];
      pf[it] = ok ? v : 0.f;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int it = 0; it < PF; ++it) {
      const int e = threadIdx.x + it * 256;
      if (e < SR_STEM_PFLOATS) patch[e] = pf[it];
    }
  };
  if ((int)blockIdx.x < p.total) fetch(blockIdx.x);
  stash();
  __syncthreads();  // weights + first patch in place

  for (int tile = blockIdx.x; tile < p.total; tile += gridDim.x) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int oy0 = ty * SR_STEM_T, ox0 = tx * SR_STEM_T;
    const int next = tile + gridDim.x;
    if (next < p.total) fetch(next);
    __builtin_amdgcn_sched_barrier(0);

    v16f acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

#pragma unroll
    for (int step = 0; step < SR_STEM_STEPS; ++step) {
      constexpr int dummy = 0; (void)dummy;
      const int o0 = sr_stem_tap_offset(2 * step), o1 = sr_stem_tap_offset(2 * step + 1);
      // neighbours in a patch row: one immediate offset (+ half); otherwise a per-lane select between two offsets
      const int oa = (o1 == o0 + 1) ? o0 + half : (half ? o1 : o0);
      const float a0 = patch[abase[0] + oa], a1 = patch[abase[1] + oa];
      const float2 w = wl2[step * 64];
      acc[0][0] = SR_MFMA(a0, w.x, acc[0][0]);
      acc[1][0] = SR_MFMA(a1, w.x, acc[1][0]);
      acc[0][1] = SR_MFMA(a0, w.y, acc[0][1]);
      acc[1][1] = SR_MFMA(a1, w.y, acc[1][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every wave is done reading this tile's patch
    if (next < p.total) stash();
    __syncthreads();

    float* __restrict__ ob = p.out + (int64_t)b * p.out_sb;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mrow = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oy = oy0 + 4 * wave + 2 * m + (mrow >> 4), ox = ox0 + (mrow & 15);
        if (oy < p.Ho && ox < p.Wo) {
          float* o = ob + ((int64_t)oy * p.Wo + ox) * p.out_sp + i;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            float v = acc[m][n][r] * sc[n] + sh[n];
            if (p.slope >= 0.f) v = v >= 0.f ? v : v * p.slope;
            o[n * 32] = v;
          }
        }
      }
  }
}

// ------------------------------------------------------------------ MaxPool(2, s1) + BlurPool(4, s2, reflect) ---
//
// out[oy,ox] = sum_{i,j<4} f_i f_j MP[refl(2oy-1+i)][refl(2ox-1+j)],  f = (1,3,3,1)/8,  MP[y][x] = max in[y..y+1][x..x+1]
// (MP is (H-1) x (W-1); reflection is on MP's index range, as nn.ReflectionPad2d((1,2,1,2)) applies it to the max-pooled
// map).  One thread per (output pixel, 4 channels).  Interior pixels read their 5x5 window once (25 float4 loads,
// vertical max then horizontal max); border pixels take the generic reflected path.
__device__ __forceinline__ int sr_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ float4 sr_max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 sr_axpy4(float s, float4 a, float4 acc) {
  return make_float4(acc.x + s * a.x, acc.y + s * a.y, acc.z + s * a.z, acc.w + s * a.w);
}

__device__ __forceinline__ float4 sr_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// One output pixel through the generic (reflected) path: 16 max-pooled samples, 4 loads each.
__device__ __forceinline__ float4 sr_maxblur_generic(const float* __restrict__ ib, int in_sp, int W, int Hm, int Wm,
                                                     int oy, int ox, int c4) {
  const float f[4] = {0.125f, 0.375f, 0.375f, 0.125f};
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < 4; ++i) {
    const int my = sr_reflect(2 * oy - 1 + i, Hm);
    float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < 4; ++j) {
      const int mx = sr_reflect(2 * ox - 1 + j, Wm);
      const float* q = ib + ((int64_t)my * W + mx) * in_sp + 4 * c4;
      row = sr_axpy4(f[j], sr_max4(sr_max4(sr_ld4(q), sr_ld4(q + in_sp)),
                                   sr_max4(sr_ld4(q + (int64_t)W * in_sp), sr_ld4(q + (int64_t)W * in_sp + in_sp))), row);
    }
    acc = sr_axpy4(f[i], row, acc);
  }
  return acc;
}

// One thread = a block of 2 x BW output pixels x 4 channels (BW = 2 or 4).  Interior blocks stream their 7 x (2 BW + 3) input
// window row by row (49 float4 loads for 4 outputs, 77 for 8): vertical max with the previous row, horizontal max,
// horizontal blur for the BW output columns, then the row's contribution to the two output rows.  (BW = 4 is an ablation
// switch: measured slower, see sr_maxblurpool_nhwc_fwd.)
template <int BW>
__global__ __launch_bounds__(256) void sr_maxblurpool_kernel(const float* __restrict__ in, int64_t in_sb, int in_sp,
                                                             float* __restrict__ out, int64_t out_sb, int out_sp,
                                                             int H, int W, int Ho, int Wo, int C4, int xcd_order) {
  constexpr int WC = 2 * BW + 3;   // input columns of a block's window
  const int Hm = H - 1, Wm = W - 1;
  const int bx = (Wo + BW - 1) / BW, by = (Ho + 1) / 2;
  const int64_t total = (int64_t)bx * by * C4;
  const float f[4] = {0.125f, 0.375f, 0.375f, 0.125f};
  // XCD-aware work order (r04): workgroups are dispatched round-robin over the 8 XCDs, each with its own L2, and vertically
  // adjacent block rows share 3 of their 7 input rows -- 5 workgroups apart at 120x160, i.e. always in another L2 (r02-r03 PMC:
  // 1.75 x the algorithmic bytes = exactly 7 / 4).  When the grid covers the map without striding, XCD x takes the contiguous
  // eighth [x T / 8, (x + 1) T / 8) of the launch's workgroups (images included), so neighbouring rows meet in one L2
  // (r04 PMC: 2 755 -> 1 612 MB per launch = 1.02 x the algorithmic bytes).
  unsigned bxi = blockIdx.x, byi = blockIdx.y;
  if (xcd_order) {
    const unsigned T = gridDim.x * gridDim.y, L = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned Lp = (L & 7u) * (T >> 3) + (L >> 3);
    byi = Lp / gridDim.x;
    bxi = Lp - byi * gridDim.x;
  }
  const float* __restrict__ ib = in + (int64_t)byi * in_sb;
  float* __restrict__ ob = out + (int64_t)byi * out_sb;
  for (int64_t idx = (int64_t)bxi * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const int blk = (int)(idx / C4);
    const int oy0 = 2 * (blk / bx), ox0 = BW * (blk % bx);
    const int y0 = 2 * oy0 - 1, x0 = 2 * ox0 - 1;  // first max-pooled row / column of the window
    if (y0 >= 0 && y0 + 5 < Hm && x0 >= 0 && x0 + 2 * BW + 1 < Wm && oy0 + 1 < Ho && ox0 + BW - 1 < Wo) {
      const float* q = ib + ((int64_t)y0 * W + x0) * in_sp + 4 * c4;
      float4 prev[WC], acc[2][BW];
#pragma unroll
      for (int j = 0; j < WC; ++j) prev[j] = sr_ld4(q + (int64_t)j * in_sp);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < BW; ++b2) acc[a][b2] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        q += (int64_t)W * in_sp;
        float4 vm[WC];
#pragma unroll
        for (int j = 0; j < WC; ++j) {
          const float4 cur = sr_ld4(q + (int64_t)j * in_sp);
          vm[j] = sr_max4(prev[j], cur);
          prev[j] = cur;
        }
        float4 mp[WC - 1];   // the max-pooled row: horizontal max of neighbouring columns
#pragma unroll
        for (int j = 0; j < WC - 1; ++j) mp[j] = sr_max4(vm[j], vm[j + 1]);
#pragma unroll
        for (int b2 = 0; b2 < BW; ++b2) {
          float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int j = 0; j < 4; ++j) h = sr_axpy4(f[j], mp[2 * b2 + j], h);
          if (i < 4) acc[0][b2] = sr_axpy4(f[i], h, acc[0][b2]);
          if (i >= 2) acc[1][b2] = sr_axpy4(f[i - 2], h, acc[1][b2]);
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < BW; ++b2)
          *reinterpret_cast<float4*>(ob + ((int64_t)(oy0 + a) * Wo + ox0 + b2) * out_sp + 4 * c4) = acc[a][b2];
    } else {
      for (int a = 0; a < 2; ++a)
        for (int b2 = 0; b2 < BW; ++b2) {
          const int oy = oy0 + a, ox = ox0 + b2;
          if (oy < Ho && ox < Wo)
            *reinterpret_cast<float4*>(ob + ((int64_t)oy * Wo + ox) * out_sp + 4 * c4) =
                sr_maxblur_generic(ib, in_sp, W, Hm, Wm, oy, ox, c4);
        }
    }
  }
}

// The streaming form (r05).  The block kernel above reads 12.25 float4 per output (its 7 x 7 windows overlap by 3 rows and 3
// columns) and leaves the vertical overlap to the L2; ATen's plain streams reach 6 TB/s on this map (scripts/micro/
// hbm_ceiling.py), the block kernel 3.4.  Here a thread owns 2 output columns x 4 channels and WALKS DOWN a band of output
// rows: per output row it loads two new input rows of 7 pixels (7 float4 per output), the previous input row and the two
// partial output rows live in registers, the next iteration's 14 loads are issued before this iteration's arithmetic.  A
// workgroup = 16 column blocks x 16 channel quads: every row step reads 67 consecutive pixels (17 KB), every output step
// writes 32 (8 KB).  Same operations in the same order per output as the block kernel (vertical max, horizontal max, horizontal
// blur j = 0..3, vertical blur i = 0..3): bit-identical.  Only outputs whose window needs no reflection take this path (rows
// 1 .. (H-4)/2, columns 1 .. (W-4)/2); the frame around them goes through sr_maxblur_generic in the trailing workgroups of the
// same launch.  Measured (64 images, [64,64,240,320] -> [64,64,120,160], 1.57 GB): 315-337 us = 4.7-5.0 TB/s (block kernel 460 us =
// 3.4 TB/s); non-temporal loads: 392 us (the 3-column overlap of neighbouring threads is served by the L1 / L2 they bypass).
struct SrPoolStreamParams {
  const float* in; int64_t in_sb; int in_sp;
  float* out; int64_t out_sb; int out_sp;
  int H, W, Ho, Wo, C4;
  int oy_lo, oy_hi, ox_lo, ox_hi;     // the streamed outputs: [oy_lo, oy_hi] x [ox_lo, ox_hi]
  int bands, band_rows, col_blocks, col_groups;   // col_blocks = ceil(columns / 2), col_groups = ceil(col_blocks * C4 / 256)
  int stream_wgs;                     // workgroups 0 .. stream_wgs - 1 stream, the rest take the frame
  int frame_rows_top, frame_rows_bottom, frame_cols_left, frame_cols_right;
};

// (explicit fused multiply-adds: left to -ffp-contract the compiler fuses `f0 h0 + f1 h1` as fma(f1, h1, f0 h0) in the loop
// body and as fma(f0, h0, f1 h1) in the peeled first iteration -- 1 ulp apart; this is the form the block kernel compiles to)
__device__ __forceinline__ float4 sr_fma4(float s, float4 a, float4 acc) {
  return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float4 sr_scale4(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

__global__ __launch_bounds__(256) void sr_maxblurpool_stream_kernel(SrPoolStreamParams p) {
  const float f[4] = {0.125f, 0.375f, 0.375f, 0.125f};
  const float* __restrict__ ib = p.in + (int64_t)blockIdx.y * p.in_sb;
  float* __restrict__ ob = p.out + (int64_t)blockIdx.y * p.out_sb;
  if ((int)blockIdx.x >= p.stream_wgs) {
    // ---- the frame: rows [0, oy_lo) and (oy_hi, Ho) at full width, columns [0, ox_lo) and (ox_hi, Wo) of the streamed rows
    const int Hm = p.H - 1, Wm = p.W - 1;
    const int full_rows = p.frame_rows_top + p.frame_rows_bottom, side = p.frame_cols_left + p.frame_cols_right;
    const int n_full = full_rows * p.Wo, n_side = (p.oy_hi - p.oy_lo + 1) * side;
    const int64_t total = (int64_t)(n_full + n_side) * p.C4;
    for (int64_t idx = (int64_t)((int)blockIdx.x - p.stream_wgs) * 256 + threadIdx.x; idx < total;
         idx += (int64_t)((int)gridDim.x - p.stream_wgs) * 256) {
      const int c4 = (int)(idx % p.C4);
      const int px = (int)(idx / p.C4);
      int oy, ox;
      if (px < n_full) {
        const int r = px / p.Wo;
        ox = px - r * p.Wo;
        oy = r < p.frame_rows_top ? r : p.oy_hi + 1 + (r - p.frame_rows_top);
      } else {
        const int q = px - n_full, r = q / side, c = q - r * side;
        oy = p.oy_lo + r;
        ox = c < p.frame_cols_left ? c : p.ox_hi + 1 + (c - p.frame_cols_left);
      }
      *reinterpret_cast<float4*>(ob + ((int64_t)oy * p.Wo + ox) * p.out_sp + 4 * c4) =
          sr_maxblur_generic(ib, p.in_sp, p.W, Hm, Wm, oy, ox, c4);
    }
    return;
  }
  const int band = (int)blockIdx.x / p.col_groups, cg = (int)blockIdx.x - band * p.col_groups;
  const int t = cg * 256 + (int)threadIdx.x;
  const int c4 = t % p.C4, cb = t / p.C4;
  if (cb >= p.col_blocks) return;
  const int ox0 = p.ox_lo + 2 * cb;
  const bool two = ox0 + 1 <= p.ox_hi;                       // the last block of an odd column count holds one output
  const int oy_first = p.oy_lo + band * p.band_rows;
  const int oy_last = min(oy_first + p.band_rows - 1, p.oy_hi);
  if (oy_first > oy_last) return;
  const int x0 = 2 * ox0 - 1;                                // first max-pooled (= input) column of the window
  int xo[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) xo[j] = min(x0 + j, p.W - 1) * p.in_sp + 4 * c4;   // (clamped: only a one-output block gets there)
  // input rows: max-pooled row m = max(input rows m, m + 1); output oy blurs max-pooled rows 2 oy - 1 .. 2 oy + 2
  const float* q = ib + (int64_t)(2 * oy_first - 1) * p.W * p.in_sp;
  const int64_t rs = (int64_t)p.W * p.in_sp;
  float4 prev[7], cur[2][7];
#pragma unroll
  for (int j = 0; j < 7; ++j) prev[j] = sr_ld4(q + xo[j]);
#pragma unroll
  for (int j = 0; j < 7; ++j) { cur[0][j] = sr_ld4(q + rs + xo[j]); cur[1][j] = sr_ld4(q + 2 * rs + xo[j]); }
  q += 3 * rs;
  float4 acc_old[2], acc_new[2];
  acc_old[0] = acc_old[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n = oy_last - oy_first + 1;
  for (int it = 0; it <= n; ++it) {           // iteration it: max-pooled rows 2 oy - 1, 2 oy of oy = oy_first + it
    float4 nxt[2][7];
    if (it < n) {                             // (the rows of iteration it + 1; iteration n needs rows 2 oy_last + 2, + 3: in range)
#pragma unroll
      for (int j = 0; j < 7; ++j) { nxt[0][j] = sr_ld4(q + xo[j]); nxt[1][j] = sr_ld4(q + rs + xo[j]); }
      q += 2 * rs;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float4 vm[7], mp[6];
#pragma unroll
      for (int j = 0; j < 7; ++j) { vm[j] = sr_max4(prev[j], cur[r][j]); prev[j] = cur[r][j]; }
#pragma unroll
      for (int j = 0; j < 6; ++j) mp[j] = sr_max4(vm[j], vm[j + 1]);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        float4 h = sr_scale4(f[0], mp[2 * b2]);
#pragma unroll
        for (int j = 1; j < 4; ++j) h = sr_fma4(f[j], mp[2 * b2 + j], h);
        if (r == 0) {
          acc_new[b2] = sr_scale4(f[0], h);
          acc_old[b2] = sr_fma4(f[2], h, acc_old[b2]);
        } else {
          acc_new[b2] = sr_fma4(f[1], h, acc_new[b2]);
          acc_old[b2] = sr_fma4(f[3], h, acc_old[b2]);
        }
      }
    }
    if (it >= 1) {
      float* o = ob + ((int64_t)(oy_first + it - 1) * p.Wo + ox0) * p.out_sp + 4 * c4;
      *reinterpret_cast<float4*>(o) = acc_old[0];
      if (two) *reinterpret_cast<float4*>(o + p.out_sp) = acc_old[1];
    }
    acc_old[0] = acc_new[0];
    acc_old[1] = acc_new[1];
    if (it < n) {
#pragma unroll
      for (int j = 0; j < 7; ++j) { cur[0][j] = nxt[0][j]; cur[1][j] = nxt[1][j]; }
    }
  }
}

// ------------------------------------------------------------------ InstanceNorm2d (+ LeakyReLU) ----------------
//
// Per (image, channel) statistics over H*W, biased variance, no affine (nn.InstanceNorm2d defaults; reference
// networks.py:188, 198).  Deterministic three-step reduction (no atomics):
//   1. sr_inorm_partial_kernel: a workgroup owns (image, chunk of pixels): chunk mean, then chunk M2 = sum (x-mean)^2
//      (the second read of the chunk hits L2), written to partial[b][chunk][{mean, M2}][C];
//   2. sr_inorm_finalize_kernel: Chan merge of the chunks in index order -> stats[b][{mean, rstd}][C];
//   3. sr_inorm_apply_kernel: y = (x - mean) * rstd, LeakyReLU, float4 streams (in place allowed).
struct SrInormParams {
  const float* in; int64_t in_sb; int in_sp;
  float* out; int64_t out_sb; int out_sp;
  float* partial; float* stats;
  int HW, C, C4, chunks, chunk_pix;
  float eps, slope;
};

__global__ __launch_bounds__(256) void sr_inorm_partial_kernel(SrInormParams p) {
  __shared__ float4 red[256];
  __shared__ float4 mean_s[64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lanes = p.C4;                 // float4 lanes across channels (<= 64)
  const int rows = 256 / lanes;           // pixel rows handled concurrently
  const int cl = threadIdx.x % lanes, pr = threadIdx.x / lanes;
  const int p0 = chunk * p.chunk_pix, p1 = min(p.HW, p0 + p.chunk_pix);
  const float* __restrict__ ib = p.in + (int64_t)b * p.in_sb + 4 * cl;
  const bool active = pr < rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active)
    for (int px = p0 + pr; px < p1; px += rows) {
      const float4 v = *reinterpret_cast<const float4*>(ib + (int64_t)px * p.in_sp);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < lanes) {
    float4 t = red[threadIdx.x];
    for (int r = 1; r < rows; ++r) {
      const float4 u = red[r * lanes + threadIdx.x];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    const float inv = 1.f / (float)(p1 - p0);
    mean_s[threadIdx.x] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
  }
  __syncthreads();
  const float4 m = mean_s[cl];
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active)
    for (int px = p0 + pr; px < p1; px += rows) {
      const float4 v = *reinterpret_cast<const float4*>(ib + (int64_t)px * p.in_sp);
      const float dx = v.x - m.x, dy = v.y - m.y, dz = v.z - m.z, dw = v.w - m.w;
      q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    }
  __syncthreads();
  red[threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < lanes) {
    float4 t = red[threadIdx.x];
    for (int r = 1; r < rows; ++r) {
      const float4 u = red[r * lanes + threadIdx.x];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    float* o = p.partial + ((int64_t)(b * p.chunks + chunk) * 2) * p.C + 4 * threadIdx.x;
    *reinterpret_cast<float4*>(o) = m;
    *reinterpret_cast<float4*>(o + p.C) = t;
  }
}

__global__ void sr_inorm_finalize_kernel(SrInormParams p, int B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * p.C) return;
  const int b = idx / p.C, c = idx - b * p.C;
  const float* pb = p.partial + (int64_t)b * p.chunks * 2 * p.C + c;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int k = 0; k < p.chunks; ++k) {  // Chan et al. pairwise update, fixed order
    const int p0 = k * p.chunk_pix;
    const float nk = (float)(min(p.HW, p0 + p.chunk_pix) - p0);
    const float mk = pb[(int64_t)k * 2 * p.C], qk = pb[(int64_t)k * 2 * p.C + p.C];
    const float nn = n + nk, d = mk - mean;
    mean += d * (nk / nn);
    m2 += qk + d * d * (n * nk / nn);
    n = nn;
  }
  p.stats[(int64_t)b * 2 * p.C + c] = mean;
  p.stats[(int64_t)b * 2 * p.C + p.C + c] = 1.f / sqrtf(m2 / n + p.eps);
}

// The same merge for many small chunks (the 64-pixel tiles of sr_conv1x1_stats_kernel): a workgroup owns (image, 16
// channels); 16 segments of consecutive chunks are merged concurrently, then the 16 segment records in order.
__global__ __launch_bounds__(256) void sr_inorm_finalize_seg_kernel(SrInormParams p) {
  __shared__ float sn[256], sm[256], sq[256];
  const int b = blockIdx.y, c = blockIdx.x * 16 + (threadIdx.x & 15), seg = threadIdx.x >> 4;
  const int per = (p.chunks + 15) / 16, k0 = seg * per, k1 = min(p.chunks, k0 + per);
  const float* pb = p.partial + (int64_t)b * p.chunks * 2 * p.C + (c < p.C ? c : 0);
  float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll 4
  for (int k = k0; k < k1; ++k) {
    const int p0 = k * p.chunk_pix;
    const float nk = (float)(min(p.HW, p0 + p.chunk_pix) - p0);
    const float mk = pb[(int64_t)k * 2 * p.C], qk = pb[(int64_t)k * 2 * p.C + p.C];
    const float nn = n + nk, d = mk - mean;
    mean += d * (nk / nn);
    m2 += qk + d * d * (n * nk / nn);
    n = nn;
  }
  sn[threadIdx.x] = n; sm[threadIdx.x] = mean; sq[threadIdx.x] = m2;
  __syncthreads();
  if (seg == 0 && c < p.C) {
    for (int s = 1; s < 16; ++s) {
      const float nk = sn[16 * s + threadIdx.x];
      if (nk == 0.f) continue;
      const float mk = sm[16 * s + threadIdx.x], qk = sq[16 * s + threadIdx.x];
      const float nn = n + nk, d = mk - mean;
      mean += d * (nk / nn);
      m2 += qk + d * d * (n * nk / nn);
      n = nn;
    }
    p.stats[(int64_t)b * 2 * p.C + c] = mean;
    p.stats[(int64_t)b * 2 * p.C + p.C + c] = 1.f / sqrtf(m2 / n + p.eps);
  }
}

__global__ __launch_bounds__(256) void sr_inorm_apply_kernel(SrInormParams p) {
  const int b = blockIdx.y;
  const float* __restrict__ ib = p.in + (int64_t)b * p.in_sb;
  float* __restrict__ ob = p.out + (int64_t)b * p.out_sb;
  const float* st = p.stats + (int64_t)b * 2 * p.C;
  const int64_t total = (int64_t)p.HW * p.C4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % p.C4);
    const int64_t px = idx / p.C4;
    const float4 m = *reinterpret_cast<const float4*>(st + 4 * c4);
    const float4 r = *reinterpret_cast<const float4*>(st + p.C + 4 * c4);
    float4 v = *reinterpret_cast<const float4*>(ib + px * p.in_sp + 4 * c4);
    v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
    if (p.slope >= 0.f) {
      v.x = v.x >= 0.f ? v.x : v.x * p.slope; v.y = v.y >= 0.f ? v.y : v.y * p.slope;
      v.z = v.z >= 0.f ? v.z : v.z * p.slope; v.w = v.w >= 0.f ? v.w : v.w * p.slope;
    }
    *reinterpret_cast<float4*>(ob + px * p.out_sp + 4 * c4) = v;
  }
}

// ------------------------------------------------------------------ conv 1x1 64 -> 128 + InstanceNorm statistics ----
//
// The first two operators of the matching encoder's tail (reference networks.py:187-188: Conv2d(64, 128, 1) ->
// InstanceNorm2d) as one pass over the activation: the 1x1 convolution runs on the fp32 matrix cores and, while a
// 64-pixel tile of its output is still in LDS on the way to coalesced stores, the tile's per-channel (mean, M2)
// record is written in sr_inorm_partial_kernel's format -- sr_inorm_finalize_seg_kernel (chunk_pix = 64) then yields the
// statistics without the separate read of the 128-channel map.  HBM-bound: 256 B in + 512 B out per pixel.
//
// Workgroup = 4 waves, persistent over (image, 64-pixel tile).  Wave w owns output channels 32w..32w+31 (M = co, its
// 32 A fragments stay in registers for the whole kernel), N = pixels (two 32-pixel MFMA tiles per step), K = 64 input
// channels as 32 k-steps of v_mfma_f32_32x32x2_f32 (k-step t multiplies channels t and 32+t).
#define SR_C1S_PIX 64
#define SR_C1S_XROW 68    // floats per staged input pixel (64 + 4: 16-lane float4 reads hit distinct banks)
#define SR_C1S_YROW 132   // floats per staged output pixel (128 + 4)
struct SrC1sParams {
  const float* in; int64_t in_sb; int in_sp;
  const float* w; const float* bias;       // [128][64], [128] or NULL
  float* out; int64_t out_sb; int out_sp;
  float* partial;                          // [B][tiles][{mean, M2}][128]
  int HW, tiles, total;
};

__global__ __launch_bounds__(256) void sr_conv1x1_stats_kernel(SrC1sParams p) {
  __shared__ __attribute__((aligned(16))) float xs[SR_C1S_PIX * SR_C1S_XROW];
  __shared__ __attribute__((aligned(16))) float ys[SR_C1S_PIX * SR_C1S_YROW];
  __shared__ float red[256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 31, kk = lane >> 5;
  float a[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) a[t] = p.w[(32 * wave + ln) * 64 + 32 * kk + t];
  float4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    bq[q] = p.bias ? *reinterpret_cast<const float4*>(p.bias + 32 * wave + 8 * q + 4 * kk) : make_float4(0.f, 0.f, 0.f, 0.f);

  const int spx = tid >> 4, sc4 = tid & 15;   // staging: 16 pixels x 16 float4 per pass
  float4 st[4];
  auto stage_load = [&](int item) {
    const int b = item / p.tiles, p0 = (item - b * p.tiles) * SR_C1S_PIX;
    const float* ib = p.in + (int64_t)b * p.in_sb + 4 * sc4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int px = p0 + 16 * q + spx;
      st[q] = px < p.HW ? *reinterpret_cast<const float4*>(ib + (int64_t)px * p.in_sp) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int item = blockIdx.x;
  if (item < p.total) stage_load(item);
  for (; item < p.total; item += gridDim.x) {
    const int b = item / p.tiles, tile = item - b * p.tiles, p0 = tile * SR_C1S_PIX;
    const int npx = min(SR_C1S_PIX, p.HW - p0);
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(&xs[(16 * q + spx) * SR_C1S_XROW + 4 * sc4]) = st[q];
    __syncthreads();   // xs ready; everyone is done reading ys of the previous tile
    if (item + (int)gridDim.x < p.total) stage_load(item + gridDim.x);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 xq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        xq[q] = *reinterpret_cast<const float4*>(&xs[(32 * j + ln) * SR_C1S_XROW + 32 * kk + 4 * q]);
      v16f acc = {0.f};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc = SR_MFMA(a[4 * q + 0], xq[q].x, acc);
        acc = SR_MFMA(a[4 * q + 1], xq[q].y, acc);
        acc = SR_MFMA(a[4 * q + 2], xq[q].z, acc);
        acc = SR_MFMA(a[4 * q + 3], xq[q].w, acc);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(&ys[(32 * j + ln) * SR_C1S_YROW + 32 * wave + 8 * q + 4 * kk]) =
            make_float4(acc[4 * q] + bq[q].x, acc[4 * q + 1] + bq[q].y, acc[4 * q + 2] + bq[q].z, acc[4 * q + 3] + bq[q].w);
    }
    __syncthreads();   // ys ready
    {
      float* ob = p.out + (int64_t)b * p.out_sb + 4 * (tid & 31);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int px = 8 * q + (tid >> 5);
        if (px < npx)
          *reinterpret_cast<float4*>(ob + (int64_t)(p0 + px) * p.out_sp) =
              *reinterpret_cast<const float4*>(&ys[px * SR_C1S_YROW + 4 * (tid & 31)]);
      }
    }
    // tile statistics: thread (half, co) sums 32 pixels of channel co
    const int co = tid & 127, half = tid >> 7;
    const int q0 = 32 * half, q1 = min(npx, q0 + 32);
    float s = 0.f;
    for (int px = q0; px < q1; ++px) s += ys[px * SR_C1S_YROW + co];
    red[tid] = s;
    __syncthreads();
    const float mean = (red[co] + red[128 + co]) / (float)npx;
    float m2 = 0.f;
    for (int px = q0; px < q1; ++px) {
      const float d = ys[px * SR_C1S_YROW + co] - mean;
      m2 += d * d;
    }
    __syncthreads();
    red[tid] = m2;
    __syncthreads();
    if (half == 0) {
      float* o = p.partial + (int64_t)item * 2 * 128 + co;
      o[0] = mean;
      o[128] = red[co] + red[128 + co];
    }
  }
}

// ------------------------------------------------------------------ conv 3x3, few output channels ----------------
//
// nn.Conv2d(Cin, Cout <= 16, 3, padding=1 [, padding_mode="replicate"]) with an optional InstanceNorm + LeakyReLU
// applied to its INPUT on the fly -- the 128 -> 16 conv that ends the matching encoder (networks.py:189-197), whose
// normalised 128-channel input would otherwise make an extra HBM round trip (1.26 GB per 64 images).  The 32-wide
// N-tiles of sr_conv_kernel waste half the matrix work on 16 channels, so this kernel uses v_mfma_f32_16x16x4_f32:
// M = 16 pixels of one output row, N = 16 channels, K = 4 input channels.  A workgroup (4 waves) owns 8 x 16 output
// pixels (wave w: rows 2w, 2w+1); per 32-channel slab the 10 x 18 halo is staged global -> registers (normalise,
// LeakyReLU, padding) -> LDS (36-float pixel rows: conflict-free ds_read_b128), double-buffered, one barrier per slab;
// a lane's float4 holds channels 16g + 4(l>>4) + s, s = 0..3 = its K-slot in 4 consecutive MFMAs (the packed weight
// uses the same order); weights stream from L2, 2 steps ahead.
#define SR_T16_ROW 36
#define SR_T16_HW 18
#define SR_T16_HH 10
#define SR_T16_TILE (SR_T16_HH * SR_T16_HW * SR_T16_ROW)
#define SR_T16_STAGE 6  // float4 per thread and slab (10*18*8 = 1440 <= 6*256)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SrT16Params {
  const float* in; int64_t in_sb; int in_sp;
  const float* stats;                 // [B][2][Cin] (mean, rstd) or null
  float in_slope;                     // LeakyReLU on the normalised input, < 0: none
  const float* wp;                    // packed [Cin/32][9 taps][2 g][64 lanes][4]
  const float* bias;
  float* out; int64_t out_sb; int out_sp;
  int H, W, Cin, Cout, replicate;
  int tiles_x, tiles_y, total;
  float out_slope;
  int xcd_order;                      // 1: tiles of a round are dealt to the XCDs in contiguous eighths (SR_T16_XCD, default 1)
};

__global__ void sr_t16_pack_kernel(const float* __restrict__ w /*[Cout,Cin,3,3]*/, float* __restrict__ packed, int Cout,
                                   int Cin) {
  const int total = (Cin / 32) * 9 * 2 * 64 * 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int s = e & 3, lane = (e >> 2) & 63, g = (e >> 8) & 1, tap = (e >> 9) % 9, slab = (e >> 9) / 9;
    const int co = lane & 15, ci = 32 * slab + 16 * g + 4 * (lane >> 4) + s;
    packed[e] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
  }
}

template <bool NORM, bool IN_ACT>
__global__ __launch_bounds__(256, 3) void sr_t16_kernel(SrT16Params p) {
  __shared__ __attribute__((aligned(16))) float tiles[2][SR_T16_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int slabs = p.Cin >> 5;
  const int cq = tid & 7;  // this thread stages channels 4*cq .. 4*cq+3 of every slab
  const float4* wl = reinterpret_cast<const float4*>(p.wp) + lane;

  for (int work = blockIdx.x; work < p.total; work += gridDim.x) {
    // XCD-aware work order (r04; as in sr_wino_kernel): in a full round the grid's tiles are dealt to the XCDs in contiguous
    // eighths, so that the halo two neighbouring tiles share is fetched into one L2
    int tile = work;
    if (p.xcd_order && (gridDim.x & 7) == 0) {
      const int G = (int)gridDim.x, r0 = work / G * G;
      if (r0 + G <= p.total) { const int bb = work - r0; tile = r0 + (bb & 7) * (G >> 3) + (bb >> 3); }
    }
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int oy0 = 8 * ty, ox0 = 16 * tx;
    const float* __restrict__ in_b = p.in + (int64_t)b * p.in_sb;
    const float* st = NORM ? p.stats + (int64_t)b * 2 * p.Cin : nullptr;

    int offs[SR_T16_STAGE];
#pragma unroll
    for (int it = 0; it < SR_T16_STAGE; ++it) {
      const int e = tid + it * 256, px = e >> 3;
      const int hy = px / SR_T16_HW, hx = px - hy * SR_T16_HW;
      int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      if (p.replicate) { iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1); }
      const bool ok = (px < SR_T16_HH * SR_T16_HW) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
      offs[it] = ok ? (iy * p.W + ix) * p.in_sp + 4 * cq : -1;
    }
    float4 stg[SR_T16_STAGE];
    auto stage_load = [&](int slab) {
#pragma unroll
      for (int it = 0; it < SR_T16_STAGE; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(in_b + (offs[it] >= 0 ? offs[it] + 32 * slab : 0));
        stg[it] = v;
      }
    };
    auto stage_store = [&](int slab, float* tl) {
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(1.f, 1.f, 1.f, 1.f);
      if (NORM) {
        m = *reinterpret_cast<const float4*>(st + 32 * slab + 4 * cq);
        r = *reinterpret_cast<const float4*>(st + p.Cin + 32 * slab + 4 * cq);
      }
#pragma unroll
      for (int it = 0; it < SR_T16_STAGE; ++it) {
        const int e = tid + it * 256, px = e >> 3;
        if (px < SR_T16_HH * SR_T16_HW) {
          float4 v = stg[it];
          if (NORM) { v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w; }
          if (IN_ACT) {
            v.x = v.x >= 0.f ? v.x : v.x * p.in_slope; v.y = v.y >= 0.f ? v.y : v.y * p.in_slope;
            v.z = v.z >= 0.f ? v.z : v.z * p.in_slope; v.w = v.w >= 0.f ? v.w : v.w * p.in_slope;
          }
          if (offs[it] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding pads the NORMALISED tensor
          *reinterpret_cast<float4*>(&tl[px * SR_T16_ROW + 4 * cq]) = v;
        }
      }
    };

    f32x4 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;

    __syncthreads();  // the previous tile's reads of buffer 0 are done
    stage_load(0);
    stage_store(0, tiles[0]);
    __syncthreads();
    for (int slab = 0; slab < slabs; ++slab) {
      const float* tl = tiles[slab & 1];
      const bool more = slab + 1 < slabs;
      if (more) stage_load(slab + 1);
      const float4* ws = wl + (size_t)slab * 9 * 2 * 64;
      float4 wA = ws[0], wB = ws[64];
#pragma unroll
      for (int step = 0; step < 18; ++step) {  // (tap, g)
        const int tap = step >> 1, g = step & 1, ky = tap / 3, kx = tap % 3;
        const float4 w = wA;
        wA = wB;
        if (step + 2 < 18) wB = ws[(step + 2) * 64];
        float4 a[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
          a[m] = *reinterpret_cast<const float4*>(&tl[((2 * wave + m + ky) * SR_T16_HW + li + kx) * SR_T16_ROW + 16 * g + 4 * lk]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, w.x, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, w.y, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, w.z, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, w.w, acc[m], 0, 0, 0);
        }
      }
      if (more) stage_store(slab + 1, tiles[(slab + 1) & 1]);
      __syncthreads();
    }

    // D[i = 4*(l>>4) + r][j = l&15]: pixel i of the row, channel j
    float* __restrict__ ob = p.out + (int64_t)b * p.out_sb;
    const float bv = (p.bias && li < p.Cout) ? p.bias[li] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int oy = oy0 + 2 * wave + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ox = ox0 + 4 * lk + r;
        if (oy < p.H && ox < p.W && li < p.Cout) {
          float v = acc[m][r] + bv;
          if (p.out_slope >= 0.f) v = v >= 0.f ? v : v * p.out_slope;
          ob[((int64_t)oy * p.W + ox) * p.out_sp + li] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ C ABI -------------

static int sr_cus() { return sr_device_cus(); }

extern "C" size_t sr_conv3x3_c16_packed_weight_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cout > 16 || Cin <= 0 || (Cin % 32)) return 0;
  return (size_t)(Cin / 32) * 9 * 2 * 64 * 4;
}

extern "C" int sr_conv3x3_c16_pack_weights(const float* weight, int Cout, int Cin, float* packed, void* stream_) {
  if (!weight || !packed) return SR_ERR_INVALID_ARGUMENT;
  if (Cout <= 0 || Cout > 16 || Cin <= 0 || (Cin % 32)) return SR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sr_t16_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream_, weight, packed, Cout, Cin);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_conv3x3_c16_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                       const float* in_stats, float in_leaky_slope, const float* packed_weight,
                                       const float* bias, float* out, int64_t out_batch_stride, int out_pix_stride,
                                       int B, int H, int W, int Cin, int Cout, int replicate, float leaky_slope,
                                       void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !packed_weight || !out) return SR_ERR_INVALID_ARGUMENT;
  if (Cout > 16 || (Cin % 32) || (in_pix_stride % 4) || (in_batch_stride % 4) || ((uintptr_t)in & 15) ||
      (in_stats && ((uintptr_t)in_stats & 15)))
    return SR_ERR_UNSUPPORTED;
  SrT16Params p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.stats = in_stats; p.in_slope = in_leaky_slope;
  p.wp = packed_weight; p.bias = bias;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.replicate = replicate;
  p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 7) / 8;
  p.total = p.tiles_x * p.tiles_y * B;
  p.out_slope = leaky_slope;
  int blocks = 3 * sr_cus();
  if (blocks > p.total) blocks = p.total;
  p.xcd_order = sr_opt(SR_OPT_T16_XCD);   // (ablation; results are identical)
  const bool norm = in_stats != nullptr, act = in_leaky_slope >= 0.f;
  if (norm && act) hipLaunchKernelGGL((sr_t16_kernel<true, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream_, p);
  else if (norm) hipLaunchKernelGGL((sr_t16_kernel<true, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream_, p);
  else if (act) hipLaunchKernelGGL((sr_t16_kernel<false, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream_, p);
  else hipLaunchKernelGGL((sr_t16_kernel<false, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream_, p);
  return sr_hip_rc(hipGetLastError());
}

extern "C" size_t sr_stem_packed_weight_floats(int Cout) { return Cout == 64 ? (size_t)SR_STEM_WFLOATS : 0; }

extern "C" int sr_stem_pack_weights(const float* weight, int Cout, float* packed, void* stream_) {
  if (!weight || !packed) return SR_ERR_INVALID_ARGUMENT;
  if (Cout != 64) return SR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sr_stem_pack_kernel, dim3(42), dim3(256), 0, (hipStream_t)stream_, weight, packed);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_stem7x7_fwd(const float* in, int64_t in_batch_stride, int64_t in_chan_stride, int64_t in_row_stride,
                              int64_t in_col_stride, const float* packed_weight, const float* scale,
                              const float* shift, float leaky_slope, float* out, int64_t out_batch_stride,
                              int out_pix_stride, int B, int H, int W, int Cout, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (Cout != 64) return SR_ERR_UNSUPPORTED;
  if (B == 0) return SR_OK;
  if (!in || !packed_weight || !out) return SR_ERR_INVALID_ARGUMENT;
  SrStemParams p;
  p.in = in; p.sb = in_batch_stride; p.sc = in_chan_stride; p.sy = in_row_stride; p.sx = in_col_stride;
  p.wp = packed_weight; p.scale = scale; p.shift = shift; p.slope = leaky_slope;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.H = H; p.W = W;
  p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
  p.tiles_x = (p.Wo + SR_STEM_T - 1) / SR_STEM_T; p.tiles_y = (p.Ho + SR_STEM_T - 1) / SR_STEM_T;
  p.total = p.tiles_x * p.tiles_y * B;
  int blocks = 2 * sr_cus();
  if (blocks > p.total) blocks = p.total;
  hipLaunchKernelGGL(sr_stem_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, p);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_maxblurpool_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out,
                                       int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C,
                                       void* stream_) {
  if (B < 0 || H < 4 || W < 4 || C <= 0) return SR_ERR_INVALID_ARGUMENT;  // reflection needs (H-1), (W-1) >= 3
  if (B == 0) return SR_OK;
  if (!in || !out) return SR_ERR_INVALID_ARGUMENT;
  if ((C % 4) || (in_pix_stride % 4) || (out_pix_stride % 4) || (in_batch_stride % 4) || (out_batch_stride % 4) ||
      ((uintptr_t)in & 15) || ((uintptr_t)out & 15))
    return SR_ERR_UNSUPPORTED;
  const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;
  if (sr_opt(SR_OPT_POOL_STREAM) && H >= 16 && W >= 16 && B <= 65535) {
    SrPoolStreamParams p;
    p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
    p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
    p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.C4 = C / 4;
    p.oy_lo = 1; p.oy_hi = (H - 4) / 2; p.ox_lo = 1; p.ox_hi = (W - 4) / 2;   // 2 o - 1 >= 0 and 2 o + 3 <= size - 1
    const int rows = p.oy_hi - p.oy_lo + 1, cols = p.ox_hi - p.ox_lo + 1;
    p.col_blocks = (cols + 1) / 2;
    p.col_groups = (int)(((int64_t)p.col_blocks * p.C4 + 255) / 256);
    // bands: two workgroups are resident per CU (176 registers); pick the band count whose workgroups fill whole rounds of those
    // slots (a band of r output rows reads 2 r + 3 input rows), at most ~10 rounds, at least 8 rows per band
    const int slots = 2 * sr_cus(), per_band = B * p.col_groups;
    int best_bands = 1;
    double best = -1.0;
    for (int nb = 1; nb <= (rows >= 8 ? rows / 8 : 1); ++nb) {
      const int br = (rows + nb - 1) / nb, real = (rows + br - 1) / br;
      const int64_t wgs = (int64_t)real * per_band;
      const int64_t rounds = (wgs + slots - 1) / slots;
      if (rounds > 10 && best >= 0.0) break;
      const double eff = (double)wgs / (double)(rounds * slots) * (2.0 * br) / (2.0 * br + 3.0);
      if (eff > best + 1e-9) { best = eff; best_bands = real; }
    }
    p.band_rows = (rows + best_bands - 1) / best_bands;
    p.bands = (rows + p.band_rows - 1) / p.band_rows;
    p.stream_wgs = p.bands * p.col_groups;
    p.frame_rows_top = p.oy_lo; p.frame_rows_bottom = Ho - 1 - p.oy_hi;
    p.frame_cols_left = p.ox_lo; p.frame_cols_right = Wo - 1 - p.ox_hi;
    const int64_t frame = ((int64_t)(p.frame_rows_top + p.frame_rows_bottom) * Wo +
                           (int64_t)rows * (p.frame_cols_left + p.frame_cols_right)) * p.C4;
    int frame_wgs = (int)((frame + 255) / 256 < 1024 ? (frame + 255) / 256 : 1024);
    hipLaunchKernelGGL(sr_maxblurpool_stream_kernel, dim3(p.stream_wgs + frame_wgs, B), dim3(256), 0, (hipStream_t)stream_, p);
    return sr_hip_rc(hipGetLastError());
  }
  int bw = 2;   // output columns per thread.  4 (SR_POOL_BW=4, Wo % 4 == 0) issues 9.6 instead of 12.25 loads per output and is
  if (sr_opt(SR_OPT_POOL_BW) == 4 && Wo % 4 == 0) bw = 4;   // SLOWER: 0.70 vs 0.45 ms per 64 images
                                                                                            // (half the threads, twice the registers)
  const int64_t total = (int64_t)((Ho + 1) / 2) * ((Wo + bw - 1) / bw) * (C / 4);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  int xcd = ((int64_t)blocks * 256 >= total) && (((int64_t)blocks * B) % 8 == 0);   // no striding, whole eighths
  if (sr_opt(SR_OPT_POOL_XCD) == 0) xcd = 0;          // (ablation; results are identical)
  if (bw == 4)
    hipLaunchKernelGGL((sr_maxblurpool_kernel<4>), dim3(blocks, B), dim3(256), 0, (hipStream_t)stream_, in, in_batch_stride,
                       in_pix_stride, out, out_batch_stride, out_pix_stride, H, W, Ho, Wo, C / 4, xcd);
  else
    hipLaunchKernelGGL((sr_maxblurpool_kernel<2>), dim3(blocks, B), dim3(256), 0, (hipStream_t)stream_, in, in_batch_stride,
                       in_pix_stride, out, out_batch_stride, out_pix_stride, H, W, Ho, Wo, C / 4, xcd);
  return sr_hip_rc(hipGetLastError());
}

static void sr_inorm_chunks(int B, int HW, int* chunks, int* chunk_pix) {
  int want = (8 * sr_cus() + B - 1) / B;          // ~8 workgroups per CU over the batch
  int most = (HW + 255) / 256;                    // at least 256 pixels per chunk
  int n = want < most ? want : most;
  if (n < 1) n = 1;
  *chunk_pix = (HW + n - 1) / n;
  *chunks = (HW + *chunk_pix - 1) / *chunk_pix;
}

extern "C" size_t sr_instance_norm_workspace_bytes(int B, int H, int W, int C) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  int chunks, chunk_pix;
  sr_inorm_chunks(B, H * W, &chunks, &chunk_pix);
  return ((size_t)B * chunks * 2 * C + (size_t)B * 2 * C) * sizeof(float);
}

extern "C" int sr_instance_norm_stats_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, int B, int H,
                                           int W, int C, float eps, float* stats, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !stats || !workspace) return SR_ERR_INVALID_ARGUMENT;
  if ((C % 4) || C > 256 || (in_pix_stride % 4) || (in_batch_stride % 4) || ((uintptr_t)in & 15) ||
      ((uintptr_t)workspace & 15))
    return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_instance_norm_workspace_bytes(B, H, W, C)) return SR_ERR_WORKSPACE_TOO_SMALL;
  SrInormParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.out = nullptr; p.out_sb = 0; p.out_sp = 0;
  p.HW = H * W; p.C = C; p.C4 = C / 4;
  sr_inorm_chunks(B, p.HW, &p.chunks, &p.chunk_pix);
  p.partial = (float*)workspace;
  p.stats = stats;
  p.eps = eps; p.slope = -1.0f;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(sr_inorm_partial_kernel, dim3(p.chunks, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(sr_inorm_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, p, B);
  return sr_hip_rc(hipGetLastError());
}

extern "C" size_t sr_conv1x1_stats_workspace_bytes(int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout != 128) return 0;
  const size_t tiles = ((size_t)H * W + SR_C1S_PIX - 1) / SR_C1S_PIX;
  return (size_t)B * tiles * 2 * Cout * sizeof(float);
}

extern "C" int sr_conv1x1_stats_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                         const float* weight, const float* bias, float* out, int64_t out_batch_stride,
                                         int out_pix_stride, int B, int H, int W, int Cin, int Cout, float eps,
                                         float* stats, void* workspace, size_t workspace_bytes, void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !weight || !out || !stats || !workspace) return SR_ERR_INVALID_ARGUMENT;
  if (Cin != 64 || Cout != 128 || (in_pix_stride % 4) || (in_batch_stride % 4) || (out_pix_stride % 4) ||
      (out_batch_stride % 4) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || (bias && ((uintptr_t)bias & 15)))
    return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_conv1x1_stats_workspace_bytes(B, H, W, Cout)) return SR_ERR_WORKSPACE_TOO_SMALL;
  SrC1sParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.w = weight; p.bias = bias;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.partial = (float*)workspace;
  p.HW = H * W; p.tiles = (p.HW + SR_C1S_PIX - 1) / SR_C1S_PIX; p.total = p.tiles * B;
  int blocks = 3 * sr_cus();
  if (blocks > p.total) blocks = p.total;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(sr_conv1x1_stats_kernel, dim3(blocks), dim3(256), 0, stream, p);
  SrInormParams q;
  q.in = nullptr; q.in_sb = 0; q.in_sp = 0; q.out = nullptr; q.out_sb = 0; q.out_sp = 0;
  q.HW = p.HW; q.C = Cout; q.C4 = Cout / 4; q.chunks = p.tiles; q.chunk_pix = SR_C1S_PIX;
  q.partial = p.partial; q.stats = stats; q.eps = eps; q.slope = -1.0f;
  hipLaunchKernelGGL(sr_inorm_finalize_seg_kernel, dim3(Cout / 16, B), dim3(256), 0, stream, q);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_instance_norm_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out,
                                         int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C,
                                         float eps, float leaky_slope, void* workspace, size_t workspace_bytes,
                                         void* stream_) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !out || !workspace) return SR_ERR_INVALID_ARGUMENT;
  if ((C % 4) || C > 256 || (in_pix_stride % 4) || (out_pix_stride % 4) || (in_batch_stride % 4) ||
      (out_batch_stride % 4) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || ((uintptr_t)workspace & 15))
    return SR_ERR_UNSUPPORTED;
  if (workspace_bytes < sr_instance_norm_workspace_bytes(B, H, W, C)) return SR_ERR_WORKSPACE_TOO_SMALL;
  SrInormParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.HW = H * W; p.C = C; p.C4 = C / 4;
  sr_inorm_chunks(B, p.HW, &p.chunks, &p.chunk_pix);
  p.partial = (float*)workspace;
  p.stats = p.partial + (size_t)B * p.chunks * 2 * C;
  p.eps = eps; p.slope = leaky_slope;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(sr_inorm_partial_kernel, dim3(p.chunks, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(sr_inorm_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, p, B);
  const int64_t total = (int64_t)p.HW * p.C4;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sr_inorm_apply_kernel, dim3(blocks, B), dim3(256), 0, stream, p);
  return sr_hip_rc(hipGetLastError());
}
