// sr_mlp_volume_bwd.hip -- backward of the metadata-MLP plane sweep for gfx950 (SURVEY.md §8f "next" #3).
//
// What autograd computes through FeatureVolumeManager.build_cost_volume + MLP (reference modules/cost_volume.py:451-736,
// modules/networks.py:129-147) for g = dL/d cost_volume:
//   MLP       dz2 = g W3 lrelu'(z2);  dz1 = (W2^T dz2) lrelu'(z1);  df = W1^T dz1
//             dW3 += g h2,  db3 += g,  dW2 += dz2 h1^T,  db2 += dz2,  dW1 += dz1 f^T,  db1 += dz1
//   features  warped[k,c] = f[kC+c],  cur[c] = f[KC+c],  dot_k = m_k sum_c warped[k,c] cur[c]      (cost_volume.py:691-723)
//             d_warped[k,c] = df[kC+c] + df[dot_k] m_k cur[c];  d_cur[c] += df[KC+c] + sum_k df[dot_k] m_k warped[k,c]
//   sampling  d_src[b,k,c,tap_t] += w_t d_warped[k,c]  (grid_sample backward, cost_volume.py:201-212)
// every other MLP input (mask, z', plane depth, rays, angles, pose measures) depends on the geometry only: no gradient,
// exactly as in the reference (poses / intrinsics / planes are data).  Checked against oracle.mlp_volume_backward,
// which is pinned to the reference's autograd (tests/golden/grad_hero.npz).
//
// Two kernels with the same data flow: sr_mlp_volume_bwd_kernel (r01: plain fp32 VALU fmaf chains; kept as the
// SR_MLP_BWD_VALU=1 ablation, Cin <= 256) and sr_mlp_volume_bwd_mfma_kernel (r02: the six GEMM-shaped phases 2, 3, 5, 6,
// 7, 8 on v_mfma_f32_32x32x2_f32, any view count up to Cin = 416; see its comment).  A
// persistent workgroup (256 threads, one per CU) walks (image, 32-pixel tile) work items and, inside, the D planes:
//   1 assemble the 32 x Cin feature rows in LDS exactly as the forward sweep does (sr_project_sample: bit-identical
//     taps), remembering tap weights / texels;           2-3 the two hidden layers, keeping pre-activations in LDS;
//   4 dz2 (+ dW3, db2, db3);  5 dW2 += dz2 h1^T;  6 dz1 = W2^T dz2 (+ db1);  7 dW1 += dz1 f^T;
//   8 df for the channels that carry gradient;            9 scatter: d_cur in LDS, d_src with hardware fp32 atomics.
// The weight gradients live in registers for the whole launch (thread t owns column t of dW1 and a 64 x 1 strip of
// dW2) and are flushed once with atomics: 128 + 64 accumulators per thread, hence one workgroup per CU.
#include <stdlib.h>

#include "sr_common.h"

namespace {

constexpr int P = 32;      // pixels per tile
constexpr int HID = 128;   // hidden width (the reference's [202, 128, 128, 1])
constexpr int ZS = HID + 1;

struct SrMlpBwdParams {
  const float* grad_cv; int64_t g_sb, g_sd, g_sp;
  const float* cur;        // [B,C,h,w]
  const float* src_nhwc;   // [B*K, h*w, C]
  const float* invK;       // [B,16]
  const float* geom;       // [B*K, SR_GEOM_STRIDE]
  SrPlanes planes;
  const float* W1; const float* b1; const float* W2; const float* b2; const float* W3;   // nn.Linear layouts [out][in]
  const float* W1T; const float* W2T;                                                    // [in][out] copies
  float* d_cur;            // [B,C,h,w]
  float* d_src_nhwc;       // [B*K, h*w, C], zero-initialised
  float* dW1; float* db1; float* dW2; float* db2; float* dW3; float* db3;                 // zero-initialised
  int B, K, h, w, D, Cin, tiles_per_image, total_items;
  float inv_w, inv_h, slope;
};

__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.0f ? v : v * s; }
__device__ __forceinline__ float lrelu_grad(float v, float s) { return v > 0.0f ? 1.0f : s; }

__host__ __device__ inline size_t sr_mlp_bwd_lds_floats(int K, int C, int Cin) {
  const int FS = Cin | 1, NEED = K * C + C + K, DFS = NEED | 1;
  return (size_t)P * FS + 2 * (size_t)P * ZS + 2 * (size_t)P * K * 4 + (size_t)P * DFS + 2 * (size_t)P * C + P * 3 + P;
}

template <int C>
__global__ __launch_bounds__(256, 1) void sr_mlp_volume_bwd_kernel(SrMlpBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = p.K, Cin = p.Cin, N = p.h * p.w;
  const int FS = Cin | 1, NEED = K * C + C + K, DFS = NEED | 1;
  float* F = lds;                                   // [P][FS]   MLP inputs
  float* Z1 = F + P * FS;                           // [P][ZS]   z1, later dz1
  float* Z2 = Z1 + P * ZS;                          // [P][ZS]   z2, later dz2
  float* TW = Z2 + P * ZS;                          // [P][K][4] tap weights (0 outside the image)
  int* TI = reinterpret_cast<int*>(TW + P * K * 4); // [P][K][4] tap texels
  float* DF = reinterpret_cast<float*>(TI + P * K * 4);   // [P][DFS] df of [warped | cur | dot]
  float* CUR = DF + P * DFS;                        // [P][C]
  float* DCUR = CUR + P * C;                        // [P][C]
  float* R = DCUR + P * C;                          // [P][3]   invK (x+.5, y+.5, 1)
  float* G = R + P * 3;                             // [P]      dL/d cost of the current plane
  const int o_cur = K * C, o_mask = o_cur + C, o_z = o_mask + K, o_d = o_z + K, o_dot = o_d + 1;
  const int o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3, o_pd = o_sray + 3 * K;
  const int o_rm = o_pd + K, o_tm = o_rm + K;
  const int t = threadIdx.x;
  const int u = t & (HID - 1), pg = t >> 7;         // (hidden unit, half of the tile) in the layer phases
  const int pp = t & (P - 1), kq = t >> 5;          // (pixel, view residue mod 8) in the assembly / scatter phases
  const float slope = p.slope;

  float acc1[HID];   // dW1[o][t] for t < Cin
  float acc2[64];    // dW2[pg * 64 + oo][u]
  float a_b1 = 0.f, a_b2 = 0.f, a_w3 = 0.f, a_b3 = 0.f;
#pragma unroll
  for (int o = 0; o < HID; ++o) acc1[o] = 0.f;
#pragma unroll
  for (int o = 0; o < 64; ++o) acc2[o] = 0.f;

  for (int item = blockIdx.x; item < p.total_items; item += gridDim.x) {
    const int b = item / p.tiles_per_image, tile = item - b * p.tiles_per_image;
    const int pix0 = tile * P;
    // ---- 0: per-tile constants -------------------------------------------------------------------------
    for (int e = t; e < P * C; e += 256) {
      const int q = e / C, c = e - q * C;
      const int pix = min(pix0 + q, N - 1);
      CUR[e] = p.cur[((size_t)b * C + c) * N + pix];
      DCUR[e] = 0.f;
    }
    if (t < P) {
#pragma clang fp contract(off)
      const int pix = min(pix0 + t, N - 1);
      const int y = pix / p.w, x = pix - y * p.w;
      const float* iK = p.invK + 16 * (size_t)b;
      const float px = (float)x + 0.5f, py = (float)y + 0.5f;  // geometry_utils.py:34-44
      R[t * 3 + 0] = iK[0] * px + iK[1] * py + iK[2];
      R[t * 3 + 1] = iK[4] * px + iK[5] * py + iK[6];
      R[t * 3 + 2] = iK[8] * px + iK[9] * py + iK[10];
    }
    __syncthreads();
    const int mypix = min(pix0 + pp, N - 1);
    const bool active = pix0 + pp < N;
    const int my_y = mypix / p.w, my_x = mypix - my_y * p.w;
    const float* geom_b = p.geom + (size_t)b * K * SR_GEOM_STRIDE;
    const float* src_b = p.src_nhwc + (size_t)b * K * N * C;
    float* dsrc_b = p.d_src_nhwc + (size_t)b * K * N * C;
    const float* planes = p.planes.ptr + b * p.planes.sb + my_y * p.planes.sy + my_x * p.planes.sx;
    const float* gcv = p.grad_cv + b * p.g_sb + (int64_t)mypix * p.g_sp;

    for (int j = 0; j < p.D; ++j) {
      // ---- 1: features of plane j (cost_volume.py:641-723), one (pixel, view) pair per thread and pass ----
      {
        const float d = planes[j * p.planes.sd];
        float X0, X1, X2;
        {
#pragma clang fp contract(off)
          X0 = d * R[pp * 3 + 0]; X1 = d * R[pp * 3 + 1]; X2 = d * R[pp * 3 + 2];  // geometry_utils.py:56-57
        }
        float* f = F + pp * FS;
        if (kq == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) f[o_cur + c] = CUR[pp * C + c];
          f[o_d] = d;
          G[pp] = active ? gcv[j * p.g_sd] : 0.0f;
        }
        // F.normalize(X), eps 1e-12 (cost_volume.py:641-651)
        const float cden = fmaxf(sqrtf((X0 * X0 + X1 * X1) + X2 * X2), 1e-12f);
        const float cr0 = X0 / cden, cr1 = X1 / cden, cr2 = X2 / cden;
        const float n1 = fmaxf(sqrtf((cr0 * cr0 + cr1 * cr1) + cr2 * cr2), 1e-5f);
        for (int k = kq; k < K; k += 8) {
          const float* g = geom_b + k * SR_GEOM_STRIDE;
          SrSample s;
          sr_project_sample(g, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, s);
          const float* img = src_b + (size_t)k * N * C;
          const float wt[4] = {s.w_nw, s.w_ne, s.w_sw, s.w_se};
          const int ot[4] = {s.o_nw, s.o_ne, s.o_sw, s.o_se};
          float warped[C];
#pragma unroll
          for (int c = 0; c < C; ++c) warped[c] = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            TW[(pp * K + k) * 4 + q] = wt[q];
            TI[(pp * K + k) * 4 + q] = ot[q];
            const float4* tp = reinterpret_cast<const float4*>(img + (size_t)ot[q] * C);
#pragma unroll
            for (int c4 = 0; c4 < C / 4; ++c4) {
              const float4 v = tp[c4];
              warped[4 * c4 + 0] = fmaf(wt[q], v.x, warped[4 * c4 + 0]);
              warped[4 * c4 + 1] = fmaf(wt[q], v.y, warped[4 * c4 + 1]);
              warped[4 * c4 + 2] = fmaf(wt[q], v.z, warped[4 * c4 + 2]);
              warped[4 * c4 + 3] = fmaf(wt[q], v.w, warped[4 * c4 + 3]);
            }
          }
          float dot = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            f[k * C + c] = warped[c];
            dot = fmaf(warped[c], CUR[pp * C + c], dot);
          }
          const float m = s.zp > 0.0f ? 1.0f : 0.0f;      // cost_volume.py:231-232
          f[o_mask + k] = m;
          f[o_z + k] = s.zp;
          f[o_dot + k] = dot * m;                          // cost_volume.py:691-695
          // source ray normalize(X - t_k) and its angle to the current ray (cost_volume.py:654-688)
          const float v0 = X0 - g[12], v1 = X1 - g[13], v2 = X2 - g[14];
          const float sden = fmaxf(sqrtf((v0 * v0 + v1 * v1) + v2 * v2), 1e-12f);
          const float sr0 = v0 / sden, sr1 = v1 / sden, sr2 = v2 / sden;
          const float n2 = fmaxf(sqrtf((sr0 * sr0 + sr1 * sr1) + sr2 * sr2), 1e-5f);
          f[o_ang + k] = ((cr0 / n1) * (sr0 / n2) + (cr1 / n1) * (sr1 / n2)) + (cr2 / n1) * (sr2 / n2);
          f[o_sray + 3 * k + 0] = sr0; f[o_sray + 3 * k + 1] = sr1; f[o_sray + 3 * k + 2] = sr2;
          if (k == 0) { f[o_cray + 0] = cr0; f[o_cray + 1] = cr1; f[o_cray + 2] = cr2; }
          f[o_pd + k] = g[15]; f[o_rm + k] = g[16]; f[o_tm + k] = g[17];
        }
      }
      __syncthreads();
      // ---- 2: z1 = W1 f + b1 -------------------------------------------------------------------------------
      {
        float a[16];
        const float bb = p.b1[u];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = bb;
        const float* fp = F + (pg * 16) * FS;
        for (int i = 0; i < Cin; ++i) {
          const float wv = p.W1T[(size_t)i * HID + u];
#pragma unroll
          for (int q = 0; q < 16; ++q) a[q] = fmaf(wv, fp[q * FS + i], a[q]);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) Z1[(pg * 16 + q) * ZS + u] = a[q];
      }
      __syncthreads();
      // ---- 3: z2 = W2 lrelu(z1) + b2 -----------------------------------------------------------------------
      {
        float a[16];
        const float bb = p.b2[u];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = bb;
        const float* zp = Z1 + (pg * 16) * ZS;
        for (int i = 0; i < HID; ++i) {
          const float wv = p.W2T[(size_t)i * HID + u];
#pragma unroll
          for (int q = 0; q < 16; ++q) a[q] = fmaf(wv, lrelu(zp[q * ZS + i], slope), a[q]);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) Z2[(pg * 16 + q) * ZS + u] = a[q];
      }
      __syncthreads();
      // ---- 4: dz2 = g W3 lrelu'(z2) in place; dW3, db2, db3 ---------------------------------------------
      {
        const float w3 = p.W3[u];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int r = pg * 16 + q;
          const float z = Z2[r * ZS + u];
          const float g = G[r];
          a_w3 = fmaf(g, lrelu(z, slope), a_w3);
          const float dz = g * w3 * lrelu_grad(z, slope);
          Z2[r * ZS + u] = dz;
          a_b2 += dz;
        }
        if (t == 0)
          for (int r = 0; r < P; ++r) a_b3 += G[r];
      }
      __syncthreads();
      // ---- 5: dW2[o][i] += dz2[o] h1[i]   (thread: i = u, o = pg * 64 + oo) ---------------------------
      for (int r = 0; r < P; ++r) {
        const float hv = lrelu(Z1[r * ZS + u], slope);
        const float* zp = Z2 + r * ZS + pg * 64;
#pragma unroll
        for (int oo = 0; oo < 64; ++oo) acc2[oo] = fmaf(zp[oo], hv, acc2[oo]);
      }
      // ---- 6: dz1 = (W2^T dz2) lrelu'(z1) in place; db1 -----------------------------------------------
      {
        float a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = 0.f;
        const float* zp = Z2 + (pg * 16) * ZS;
        for (int o = 0; o < HID; ++o) {
          const float wv = p.W2[(size_t)o * HID + u];
#pragma unroll
          for (int q = 0; q < 16; ++q) a[q] = fmaf(wv, zp[q * ZS + o], a[q]);
        }
        __syncthreads();   // phase 5 of every thread has read its z1 values
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int r = pg * 16 + q;
          const float dz = a[q] * lrelu_grad(Z1[r * ZS + u], slope);
          Z1[r * ZS + u] = dz;
          a_b1 += dz;
        }
      }
      __syncthreads();
      // ---- 7: dW1[o][i] += dz1[o] f[i]   (thread: i = t) ------------------------------------------------
      if (t < Cin) {
        for (int r = 0; r < P; ++r) {
          const float fv = F[r * FS + t];
          const float* zp = Z1 + r * ZS;
#pragma unroll
          for (int o = 0; o < HID; ++o) acc1[o] = fmaf(zp[o], fv, acc1[o]);
        }
      }
      // ---- 8: df = W1^T dz1 for [warped | cur | dot] -----------------------------------------------------
      for (int e = t; e < P * NEED; e += 256) {
        const int r = e / NEED, q = e - r * NEED;
        const int i = q < o_mask ? q : o_dot + (q - o_mask);
        const float* zp = Z1 + r * ZS;
        float s = 0.f;
        for (int o = 0; o < HID; ++o) s = fmaf(p.W1[(size_t)o * Cin + i], zp[o], s);
        DF[r * DFS + q] = s;
      }
      __syncthreads();
      // ---- 9: back through dot product and bilinear sampling ----------------------------------------------
      {
        const float* f = F + pp * FS;
        const float* df = DF + pp * DFS;
        if (kq == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) atomicAdd(&DCUR[pp * C + c], df[o_cur + c]);
        }
        for (int k = kq; k < K; k += 8) {
          const float ddot = df[o_mask + k] * f[o_mask + k];     // DF column o_mask + k holds d f[o_dot + k]
          float* dimg = dsrc_b + (size_t)k * N * C;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float dwarp = fmaf(ddot, CUR[pp * C + c], df[k * C + c]);
            atomicAdd(&DCUR[pp * C + c], ddot * f[k * C + c]);
            if (active) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float wq = TW[(pp * K + k) * 4 + q];
                if (wq != 0.0f) unsafeAtomicAdd(dimg + (size_t)TI[(pp * K + k) * 4 + q] * C + c, wq * dwarp);
              }
            }
          }
        }
      }
      __syncthreads();
    }
    for (int e = t; e < P * C; e += 256) {
      const int q = e / C, c = e - q * C;
      if (pix0 + q < N) p.d_cur[((size_t)b * C + c) * N + pix0 + q] = DCUR[e];
    }
    __syncthreads();
  }
  // ---- flush the weight gradients --------------------------------------------------------------------------
  if (t < Cin) {
#pragma unroll
    for (int o = 0; o < HID; ++o) unsafeAtomicAdd(p.dW1 + (size_t)o * Cin + t, acc1[o]);
  }
#pragma unroll
  for (int oo = 0; oo < 64; ++oo) unsafeAtomicAdd(p.dW2 + (size_t)(pg * 64 + oo) * HID + u, acc2[oo]);
  unsafeAtomicAdd(p.db1 + u, a_b1);
  unsafeAtomicAdd(p.db2 + u, a_b2);
  unsafeAtomicAdd(p.dW3 + u, a_w3);
  if (t == 0) unsafeAtomicAdd(p.db3, a_b3);
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core version.  Same phases, same LDS buffers; the GEMM-shaped ones run on v_mfma_f32_32x32x2_f32 (fp32 in,
// fp32 accumulate) with the 32 pixels of the tile as one MFMA dimension:
//   2  Z1[p][u]  = F[p][:] . W1^T + b1         M = pixel, N = hidden (wave w: units 32w..), K = Cin     A: LDS, B: W1T (L2)
//   3  Z2[p][u]  = lrelu(Z1)[p][:] . W2^T + b2  M = pixel, N = hidden, K = 128
//   5  dW2[o][i] += dz2[:, o]^T . h1[:, i]       M = o (wave w: rows 32w..), N = i (4 tiles), K = pixel   A, B: LDS
//   6  dz1[p][i] = dz2[p][:] . W2                M = pixel, N = hidden, K = 128                            B: W2 (L2)
//   7  dW1[o][i] += dz1[:, o]^T . F[:, i]        M = o, N = i (NT1 tiles of 32 >= Cin), K = pixel          A, B: LDS
//   8  DF[p][q]  = dz1[p][:] . W1[:, idx(q)]     M = pixel, N = needed inputs (warped | cur | dot), K = 128
// A fragment = lane (m = lane % 32, k = lane / 32), B fragment = lane (k = lane / 32, n = lane % 32), accumulator register
// r <-> row (r & 3) + 8 (r >> 2) + 4 (lane / 32), column lane % 32.  LDS rows have odd strides (FS, ZS): column-of-rows
// A reads are conflict-free.  dW1 / dW2 stay in accumulator registers for the whole launch (wave w owns output rows
// 32w..32w+31: (NT1 + 4) x 16 registers, one wave per SIMD) and are flushed once with fp32 atomics, like the VALU version.
// Work per 32-pixel plane tile: ~2000 MFMAs (= 125 k MFMA cycles per workgroup) against ~0.3 MB of weights from L2.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// acc += A[32 x 2*ksteps] . B[2*ksteps x 32] on v_mfma_f32_32x32x2_f32: A from LDS (ap[2*ks]), B streamed from global / L2
// (bp[2*ks*bstride]) through a double-buffered batch of BK fragments, so that ~BK * 64 matrix-pipe cycles cover the L2
// latency of the next batch.  ACT: A = lrelu(A).
template <bool ACT, int BK = 16>
__device__ __forceinline__ f32x16 sr_gemm_lds_l2(f32x16 acc, const float* __restrict__ ap, const float* __restrict__ bp,
                                                 size_t bstride, int ksteps, float slope) {
  const int nb = ksteps / BK;
  float cur[BK], nxt[BK];
  if (nb > 0) {
#pragma unroll
    for (int u = 0; u < BK; ++u) cur[u] = bp[(size_t)2 * u * bstride];
  }
#pragma unroll 1
  for (int kb = 0; kb < nb; ++kb) {
    const bool more = kb + 1 < nb;
    const float* bn = bp + (size_t)2 * (more ? (kb + 1) * BK : 0) * bstride;   // (a valid address when nothing follows)
#pragma unroll
    for (int u = 0; u < BK; ++u) nxt[u] = bn[(size_t)2 * u * bstride];
    const float* aq = ap + 2 * kb * BK;
#pragma unroll
    for (int u = 0; u < BK; ++u) {
      float a = aq[2 * u];
      if (ACT) a = a > 0.0f ? a : a * slope;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cur[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < BK; ++u) cur[u] = nxt[u];
  }
#pragma unroll 1
  for (int ks = nb * BK; ks < ksteps; ++ks) {   // tail (Cin / 2 is not a multiple of the batch)
    float a = ap[2 * ks];
    if (ACT) a = a > 0.0f ? a : a * slope;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[(size_t)2 * ks * bstride], acc, 0, 0, 0);
  }
  return acc;
}

template <int C, int NT1>
__global__ __launch_bounds__(256, 1) void sr_mlp_volume_bwd_mfma_kernel(SrMlpBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = p.K, Cin = p.Cin, N = p.h * p.w;
  const int FS = Cin | 1, NEED = K * C + C + K, DFS = NEED | 1;
  float* F = lds;
  float* Z1 = F + P * FS;
  float* Z2 = Z1 + P * ZS;
  float* TW = Z2 + P * ZS;
  int* TI = reinterpret_cast<int*>(TW + P * K * 4);
  float* DF = reinterpret_cast<float*>(TI + P * K * 4);
  float* CUR = DF + P * DFS;
  float* DCUR = CUR + P * C;
  float* R = DCUR + P * C;
  float* G = R + P * 3;
  const int o_cur = K * C, o_mask = o_cur + C, o_z = o_mask + K, o_d = o_z + K, o_dot = o_d + 1;
  const int o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3, o_pd = o_sray + 3 * K;
  const int o_rm = o_pd + K, o_tm = o_rm + K;
  const int t = threadIdx.x;
  const int u = t & (HID - 1), pg = t >> 7;         // (hidden unit, half of the tile) in the elementwise phase 4
  const int pp = t & (P - 1), kq = t >> 5;          // (pixel, view residue mod 8) in the assembly / scatter phases
  const int wave = t >> 6, lane = t & 63, i = lane & 31, kk = lane >> 5;   // MFMA roles
  const float slope = p.slope;

  f32x16 accW1[NT1], accW2[4];
#pragma unroll
  for (int nb = 0; nb < NT1; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW1[nb][r] = 0.f;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW2[nb][r] = 0.f;
  float a_b1 = 0.f, a_b2 = 0.f, a_w3 = 0.f, a_b3 = 0.f;   // a_b1: column 32*wave + i (rows of this lane half)

  for (int item = blockIdx.x; item < p.total_items; item += gridDim.x) {
    const int b = item / p.tiles_per_image, tile = item - b * p.tiles_per_image;
    const int pix0 = tile * P;
    for (int e = t; e < P * C; e += 256) {
      const int q = e / C, c = e - q * C;
      const int pix = min(pix0 + q, N - 1);
      CUR[e] = p.cur[((size_t)b * C + c) * N + pix];
      DCUR[e] = 0.f;
    }
    if (t < P) {
#pragma clang fp contract(off)
      const int pix = min(pix0 + t, N - 1);
      const int y = pix / p.w, x = pix - y * p.w;
      const float* iK = p.invK + 16 * (size_t)b;
      const float px = (float)x + 0.5f, py = (float)y + 0.5f;  // geometry_utils.py:34-44
      R[t * 3 + 0] = iK[0] * px + iK[1] * py + iK[2];
      R[t * 3 + 1] = iK[4] * px + iK[5] * py + iK[6];
      R[t * 3 + 2] = iK[8] * px + iK[9] * py + iK[10];
    }
    __syncthreads();
    const int mypix = min(pix0 + pp, N - 1);
    const bool active = pix0 + pp < N;
    const int my_y = mypix / p.w, my_x = mypix - my_y * p.w;
    const float* geom_b = p.geom + (size_t)b * K * SR_GEOM_STRIDE;
    const float* src_b = p.src_nhwc + (size_t)b * K * N * C;
    float* dsrc_b = p.d_src_nhwc + (size_t)b * K * N * C;
    const float* planes = p.planes.ptr + b * p.planes.sb + my_y * p.planes.sy + my_x * p.planes.sx;
    const float* gcv = p.grad_cv + b * p.g_sb + (int64_t)mypix * p.g_sp;

    for (int j = 0; j < p.D; ++j) {
      // ---- 1: features of plane j (cost_volume.py:641-723), one (pixel, view) pair per thread and pass ----
      {
        const float d = planes[j * p.planes.sd];
        float X0, X1, X2;
        {
#pragma clang fp contract(off)
          X0 = d * R[pp * 3 + 0]; X1 = d * R[pp * 3 + 1]; X2 = d * R[pp * 3 + 2];  // geometry_utils.py:56-57
        }
        float* f = F + pp * FS;
        if (kq == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) f[o_cur + c] = CUR[pp * C + c];
          f[o_d] = d;
          G[pp] = active ? gcv[j * p.g_sd] : 0.0f;
        }
        const float cden = fmaxf(sqrtf((X0 * X0 + X1 * X1) + X2 * X2), 1e-12f);
        const float cr0 = X0 / cden, cr1 = X1 / cden, cr2 = X2 / cden;
        const float n1 = fmaxf(sqrtf((cr0 * cr0 + cr1 * cr1) + cr2 * cr2), 1e-5f);
        for (int k = kq; k < K; k += 8) {
          const float* g = geom_b + k * SR_GEOM_STRIDE;
          SrSample s;
          sr_project_sample(g, X0, X1, X2, p.h, p.w, p.inv_w, p.inv_h, s);
          const float* img = src_b + (size_t)k * N * C;
          const float wt[4] = {s.w_nw, s.w_ne, s.w_sw, s.w_se};
          const int ot[4] = {s.o_nw, s.o_ne, s.o_sw, s.o_se};
          float warped[C];
#pragma unroll
          for (int c = 0; c < C; ++c) warped[c] = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            TW[(pp * K + k) * 4 + q] = wt[q];
            TI[(pp * K + k) * 4 + q] = ot[q];
            const float4* tp = reinterpret_cast<const float4*>(img + (size_t)ot[q] * C);
#pragma unroll
            for (int c4 = 0; c4 < C / 4; ++c4) {
              const float4 v = tp[c4];
              warped[4 * c4 + 0] = fmaf(wt[q], v.x, warped[4 * c4 + 0]);
              warped[4 * c4 + 1] = fmaf(wt[q], v.y, warped[4 * c4 + 1]);
              warped[4 * c4 + 2] = fmaf(wt[q], v.z, warped[4 * c4 + 2]);
              warped[4 * c4 + 3] = fmaf(wt[q], v.w, warped[4 * c4 + 3]);
            }
          }
          float dot = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            f[k * C + c] = warped[c];
            dot = fmaf(warped[c], CUR[pp * C + c], dot);
          }
          const float m = s.zp > 0.0f ? 1.0f : 0.0f;      // cost_volume.py:231-232
          f[o_mask + k] = m;
          f[o_z + k] = s.zp;
          f[o_dot + k] = dot * m;                          // cost_volume.py:691-695
          const float v0 = X0 - g[12], v1 = X1 - g[13], v2 = X2 - g[14];
          const float sden = fmaxf(sqrtf((v0 * v0 + v1 * v1) + v2 * v2), 1e-12f);
          const float sr0 = v0 / sden, sr1 = v1 / sden, sr2 = v2 / sden;
          const float n2 = fmaxf(sqrtf((sr0 * sr0 + sr1 * sr1) + sr2 * sr2), 1e-5f);
          f[o_ang + k] = ((cr0 / n1) * (sr0 / n2) + (cr1 / n1) * (sr1 / n2)) + (cr2 / n1) * (sr2 / n2);
          f[o_sray + 3 * k + 0] = sr0; f[o_sray + 3 * k + 1] = sr1; f[o_sray + 3 * k + 2] = sr2;
          if (k == 0) { f[o_cray + 0] = cr0; f[o_cray + 1] = cr1; f[o_cray + 2] = cr2; }
          f[o_pd + k] = g[15]; f[o_rm + k] = g[16]; f[o_tm + k] = g[17];
        }
      }
      __syncthreads();
      // ---- 2: z1 = W1 f + b1 (MFMA: M = pixel, N = hidden units 32*wave.., K = Cin) ------------------------
      {
        f32x16 acc;
        const float bb = p.b1[32 * wave + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bb;
        const float* ap = F + i * FS + kk;
        const float* bp = p.W1T + (size_t)kk * HID + 32 * wave + i;
        acc = sr_gemm_lds_l2<false>(acc, ap, bp, HID, Cin / 2, slope);
#pragma unroll
        for (int r = 0; r < 16; ++r) Z1[((r & 3) + 8 * (r >> 2) + 4 * kk) * ZS + 32 * wave + i] = acc[r];
      }
      __syncthreads();
      // ---- 3: z2 = W2 lrelu(z1) + b2 -----------------------------------------------------------------------
      {
        f32x16 acc;
        const float bb = p.b2[32 * wave + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bb;
        const float* ap = Z1 + i * ZS + kk;
        const float* bp = p.W2T + (size_t)kk * HID + 32 * wave + i;
        acc = sr_gemm_lds_l2<true>(acc, ap, bp, HID, HID / 2, slope);
#pragma unroll
        for (int r = 0; r < 16; ++r) Z2[((r & 3) + 8 * (r >> 2) + 4 * kk) * ZS + 32 * wave + i] = acc[r];
      }
      __syncthreads();
      // ---- 4: dz2 = g W3 lrelu'(z2) in place; dW3, db2, db3 (elementwise) ---------------------------------
      {
        const float w3 = p.W3[u];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int r = pg * 16 + q;
          const float z = Z2[r * ZS + u];
          const float g = G[r];
          a_w3 = fmaf(g, lrelu(z, slope), a_w3);
          const float dz = g * w3 * lrelu_grad(z, slope);
          Z2[r * ZS + u] = dz;
          a_b2 += dz;
        }
        if (t == 0)
          for (int r = 0; r < P; ++r) a_b3 += G[r];
      }
      __syncthreads();
      // ---- 5: dW2[o][i] += dz2[:, o]^T h1[:, i]   (M = o in 32*wave.., N = i tile nb, K = pixel) -----------
#pragma unroll 4
      for (int ks = 0; ks < P / 2; ++ks) {
        const int prow = 2 * ks + kk;
        const float a = Z2[prow * ZS + 32 * wave + i];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          accW2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, lrelu(Z1[prow * ZS + 32 * nb + i], slope), accW2[nb], 0, 0, 0);
      }
      // ---- 6: dz1 = (dz2 W2) lrelu'(z1) in place; db1 -------------------------------------------------------
      {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ap = Z2 + i * ZS + kk;
        const float* bp = p.W2 + (size_t)kk * HID + 32 * wave + i;
        acc = sr_gemm_lds_l2<false>(acc, ap, bp, HID, HID / 2, slope);
        __syncthreads();   // phase 5 of every wave has read its z1 values
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* zp = Z1 + ((r & 3) + 8 * (r >> 2) + 4 * kk) * ZS + 32 * wave + i;
          const float dz = acc[r] * lrelu_grad(*zp, slope);
          *zp = dz;
          a_b1 += dz;
        }
      }
      __syncthreads();
      // ---- 7: dW1[o][i] += dz1[:, o]^T f[:, i]   (M = o in 32*wave.., N = i tile nb < NT1, K = pixel) ------
#pragma unroll 2
      for (int ks = 0; ks < P / 2; ++ks) {
        const int prow = 2 * ks + kk;
        const float a = Z1[prow * ZS + 32 * wave + i];
#pragma unroll
        for (int nb = 0; nb < NT1; ++nb)   // (columns >= Cin read past the row: finite or not, they are never flushed)
          accW1[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, F[prow * FS + 32 * nb + i], accW1[nb], 0, 0, 0);
      }
      // ---- 8: df = dz1 W1 for [warped | cur | dot]  (M = pixel, N = needed inputs, K = hidden) -------------
      for (int nb = wave; nb * 32 < NEED; nb += 4) {
        const int q = 32 * nb + i;
        const int col = q < NEED ? (q < o_mask ? q : o_dot + (q - o_mask)) : 0;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ap = Z1 + i * ZS + kk;
        const float* bp = p.W1 + (size_t)kk * Cin + col;
        acc = sr_gemm_lds_l2<false>(acc, ap, bp, (size_t)Cin, HID / 2, slope);
        if (q < NEED) {
#pragma unroll
          for (int r = 0; r < 16; ++r) DF[((r & 3) + 8 * (r >> 2) + 4 * kk) * DFS + q] = acc[r];
        }
      }
      __syncthreads();
      // ---- 9: back through dot product and bilinear sampling ----------------------------------------------
      // (a) d_cur: a thread owns (pixel, channel) pairs -- no atomics
      for (int e = t; e < P * C; e += 256) {
        const int q = e / C, c = e - q * C;
        const float* f = F + q * FS;
        const float* df = DF + q * DFS;
        float s = df[o_cur + c];
        for (int k = 0; k < K; ++k) s = fmaf(df[o_mask + k] * f[o_mask + k], f[k * C + c], s);   // ddot_k * warped[k][c]
        DCUR[e] += s;
      }
      // (b) d_src scatter with lane = channel: the 16 lanes of a (pixel, view, tap) unit add to the 64 contiguous bytes
      // of one texel, a wave instruction touches 4 texels (the r01 mapping -- lane = pixel -- touched 32 cache lines
      // per atomic instruction; the scatter's 4.4 G atomics per batch-8 step are what bounds this kernel)
      {
        const int c = t & (C - 1);
        for (int un = t / C; un < P * K * 4; un += 256 / C) {
          const int q = un & 3, pk = un >> 2;          // tap, (pixel, view)
          const int px = pk / K, k = pk - px * K;
          const float wq = TW[pk * 4 + q];
          if (wq != 0.0f && pix0 + px < N) {
            const float* f = F + px * FS;
            const float* df = DF + px * DFS;
            const float ddot = df[o_mask + k] * f[o_mask + k];     // DF column o_mask + k holds d f[o_dot + k]
            const float dwarp = fmaf(ddot, CUR[px * C + c], df[k * C + c]);
            unsafeAtomicAdd(dsrc_b + ((size_t)k * N + TI[pk * 4 + q]) * C + c, wq * dwarp);
          }
        }
      }
      __syncthreads();
    }
    for (int e = t; e < P * C; e += 256) {
      const int q = e / C, c = e - q * C;
      if (pix0 + q < N) p.d_cur[((size_t)b * C + c) * N + pix0 + q] = DCUR[e];
    }
    __syncthreads();
  }
  // ---- flush the weight gradients --------------------------------------------------------------------------
#pragma unroll
  for (int nb = 0; nb < NT1; ++nb) {
    const int col = 32 * nb + i;
    if (col < Cin) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        unsafeAtomicAdd(p.dW1 + (size_t)(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kk) * Cin + col, accW1[nb][r]);
    }
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      unsafeAtomicAdd(p.dW2 + (size_t)(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kk) * HID + 32 * nb + i, accW2[nb][r]);
  unsafeAtomicAdd(p.db1 + 32 * wave + i, a_b1);
  unsafeAtomicAdd(p.db2 + u, a_b2);
  unsafeAtomicAdd(p.dW3 + u, a_w3);
  if (t == 0) unsafeAtomicAdd(p.db3, a_b3);
}

// dst[c][r] = src[r][c]
__global__ void sr_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  const int r = e / cols, c = e - r * cols;
  dst[(size_t)c * rows + r] = src[e];
}

}  // namespace

extern "C" size_t sr_mlp_volume_bwd_scratch_bytes(int B, int K, int C, int h, int w, int hidden) {
  if (B < 0 || K < 0 || C < 0 || h < 0 || w < 0 || hidden < 0) return 0;
  const size_t cin = (size_t)C * (K + 1) + 10 * (size_t)K + 4;
  return sr_align_up((size_t)B * K * h * w * C * sizeof(float), 256) + sr_align_up(cin * hidden * sizeof(float), 256) +
         sr_align_up((size_t)hidden * hidden * sizeof(float), 256);
}

extern "C" int sr_mlp_volume_bwd(const float* grad_cv, int64_t g_sb, int64_t g_sd, int64_t g_sp, const float* cur,
                                 const float* invK_cur, const float* planes, int64_t ps_b, int64_t ps_d, int64_t ps_y,
                                 int64_t ps_x, const float* W1, const float* b1, const float* W2, const float* b2,
                                 const float* W3, float leaky_slope, int B, int K, int C, int h, int w, int D,
                                 int hidden, float* d_cur, float* d_src, float* dW1, float* db1, float* dW2,
                                 float* db2, float* dW3, float* db3, void* workspace, size_t workspace_bytes,
                                 void* scratch, size_t scratch_bytes, void* stream_) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (!dW1 || !db1 || !dW2 || !db2 || !dW3 || !db3) return SR_ERR_INVALID_ARGUMENT;
  const int Cin = C * (K + 1) + 10 * K + 4;
  const int use_valu = sr_opt(SR_OPT_MLP_BWD_VALU);   // 1: the r01 VALU kernel (ablation; up to 9 views)
  if (hidden != HID || C != 16 || Cin > (use_valu ? 256 : 416)) return SR_ERR_UNSUPPORTED;   // 16-channel features, <= 15 views
  hipStream_t stream = (hipStream_t)stream_;
  hipError_t e;
  if ((e = hipMemsetAsync(dW1, 0, (size_t)HID * Cin * sizeof(float), stream)) != hipSuccess) return sr_hip_rc(e);
  if ((e = hipMemsetAsync(db1, 0, HID * sizeof(float), stream)) != hipSuccess) return sr_hip_rc(e);
  if ((e = hipMemsetAsync(dW2, 0, (size_t)HID * HID * sizeof(float), stream)) != hipSuccess) return sr_hip_rc(e);
  if ((e = hipMemsetAsync(db2, 0, HID * sizeof(float), stream)) != hipSuccess) return sr_hip_rc(e);
  if ((e = hipMemsetAsync(dW3, 0, HID * sizeof(float), stream)) != hipSuccess) return sr_hip_rc(e);
  if ((e = hipMemsetAsync(db3, 0, sizeof(float), stream)) != hipSuccess) return sr_hip_rc(e);
  if (B == 0) return SR_OK;
  if (!grad_cv || !cur || !invK_cur || !planes || !W1 || !b1 || !W2 || !b2 || !W3 || !d_cur || !d_src || !workspace ||
      !scratch)
    return SR_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < sr_volume_workspace_bytes(B, K, C, h, w)) return SR_ERR_WORKSPACE_TOO_SMALL;
  if (scratch_bytes < sr_mlp_volume_bwd_scratch_bytes(B, K, C, h, w, hidden)) return SR_ERR_WORKSPACE_TOO_SMALL;
  if (((uintptr_t)scratch) & 255) return SR_ERR_INVALID_ARGUMENT;
  const int N = h * w;
  const size_t dsrc_bytes = (size_t)B * K * N * C * sizeof(float);
  float* d_src_nhwc = (float*)scratch;
  float* W1T = (float*)((char*)scratch + sr_align_up(dsrc_bytes, 256));
  float* W2T = (float*)((char*)W1T + sr_align_up((size_t)Cin * HID * sizeof(float), 256));
  if ((e = hipMemsetAsync(d_src_nhwc, 0, dsrc_bytes, stream)) != hipSuccess) return sr_hip_rc(e);
  hipLaunchKernelGGL(sr_transpose_kernel, dim3((HID * Cin + 255) / 256), dim3(256), 0, stream, W1, W1T, HID, Cin);
  hipLaunchKernelGGL(sr_transpose_kernel, dim3((HID * HID + 255) / 256), dim3(256), 0, stream, W2, W2T, HID, HID);

  SrMlpBwdParams p;
  p.grad_cv = grad_cv; p.g_sb = g_sb; p.g_sd = g_sd; p.g_sp = g_sp;
  p.cur = cur; p.src_nhwc = sr_ws_src_nhwc(workspace, B, K); p.invK = invK_cur; p.geom = sr_ws_geom(workspace);
  p.planes = {planes, ps_b, ps_d, ps_y, ps_x};
  p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.W3 = W3; p.W1T = W1T; p.W2T = W2T;
  p.d_cur = d_cur; p.d_src_nhwc = d_src_nhwc;
  p.dW1 = dW1; p.db1 = db1; p.dW2 = dW2; p.db2 = db2; p.dW3 = dW3; p.db3 = db3;
  p.B = B; p.K = K; p.h = h; p.w = w; p.D = D; p.Cin = Cin;
  p.tiles_per_image = (N + P - 1) / P;
  p.total_items = p.tiles_per_image * B;
  p.inv_w = 1.0f / (float)w; p.inv_h = 1.0f / (float)h; p.slope = leaky_slope;
  const size_t lds = sr_mlp_bwd_lds_floats(K, C, Cin) * sizeof(float);
  if (lds > 160 * 1024) return SR_ERR_UNSUPPORTED;
  const int cus = sr_device_cus();
  const int blocks = p.total_items < cus ? p.total_items : cus;
#define SR_BWD_LAUNCH(KERNEL)                                                                                          \
  {                                                                                                                    \
    e = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
    if (e != hipSuccess) return sr_hip_rc(e);                                                                          \
    hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(256), lds, stream, p);                                               \
  }
  const int nt1 = (Cin + 31) / 32;   // dW1 column tiles
  if (use_valu) SR_BWD_LAUNCH(sr_mlp_volume_bwd_kernel<16>)
  else if (nt1 <= 4) SR_BWD_LAUNCH((sr_mlp_volume_bwd_mfma_kernel<16, 4>))
  else if (nt1 <= 7) SR_BWD_LAUNCH((sr_mlp_volume_bwd_mfma_kernel<16, 7>))
  else if (nt1 <= 10) SR_BWD_LAUNCH((sr_mlp_volume_bwd_mfma_kernel<16, 10>))
  else SR_BWD_LAUNCH((sr_mlp_volume_bwd_mfma_kernel<16, 13>))
#undef SR_BWD_LAUNCH
  int rc = sr_hip_rc(hipGetLastError());
  if (rc != SR_OK) return rc;
  return sr_launch_unpack_nhwc(d_src_nhwc, d_src, B * K, C, N, stream);
}
