// sr_pw_tiled.hip -- the LDS-tiled form of the pointwise (1x1) convolution GEMM of sr_pw.hip, for gfx950.
//
// Same operator:  out[m, co] = act( sum_ci gate[img(m), ci] * in[m, ci] * W[co, ci] + bias[co] + residual[m, co] ),
// m = b * HW + p over BATCH-DENSE channels-last views (batch stride = HW * pixel stride: the M = B * HW rows are one
// strided matrix, pixel tiles may straddle images).  sr_pw_kernel lets every wave fetch its own A and B fragments from
// L1 / L2 (0.4 - 0.75 vector loads per MFMA): fine for the HBM-bound skip convolutions, but it tops out at 50 - 75 TFLOP/s on
// the MBConv expand / project GEMMs of the image-prior encoder (reference experiment_modules/depth_model.py:110-116; M =
// 2 400 ... 9 600 pixels, K, N = 160 ... 1 536), where the operand traffic through the CU's one vector-memory path is the
// limit.  Here a workgroup (4 waves as WM x WN) owns a (32 WM) x (32 WN NTW) output tile; per 32-channel K slab it stages
//   A   (32 WM) x 32 floats, pixel rows padded to 36 floats -> conflict-free ds_read_b128 A fragments; the squeeze-excite
//       gate of the row's image is multiplied in on the way (global -> registers -> LDS),
//   B   4 groups x 2 halves x (32 WN NTW) float4 of the weight packed in B-fragment order (sr_conv_pack_weights, ksize 1):
//       a straight copy, and a wave's fragment read is 1 KB contiguous,
// double-buffered (the next slab's global loads are in flight under the current slab's MFMAs, one barrier per slab), so
// each operand byte crosses the vector-memory path once per workgroup instead of once per wave: 0.19 KB per MFMA instead
// of 0.5 - 0.75.  Small-M problems split K across workgroups (grid.z): raw partial tiles go to a workspace and
// sr_launch_splitk_reduce adds them in index order, then bias + residual + activation -- deterministic, no atomics.
#include <stdlib.h>

#include "sr_common.h"

namespace {

typedef float pt_f16 __attribute__((ext_vector_type(16)));
typedef float pt_f4 __attribute__((ext_vector_type(4)));

constexpr int PT_KS = 32;             // channels per K slab
constexpr int PT_AROW = PT_KS + 4;    // floats per staged pixel row (padded: conflict-free b128 fragment reads)

struct SrPtParams {
  const float* in; int in_sp;          // row m at in + m * in_sp
  const float* wp;                     // packed [G][2][Co_pad][4]
  const float* bias; const float* gate;   // gate [B][Cin] or null
  const float* res; int res_sp;
  float* out; int out_sp;              // (split-K: the partial workspace, out_sp = Cout, + kpart * M * Cout)
  int64_t part_stride;
  int M, HW, Cin, Cout, Co_pad, G;     // G = packed groups
  int slabs, slabs_per_part, ksplit, tiles_n;
  float slope;
};

template <int WM, int WN, int NTW, bool GATE>
__global__ __launch_bounds__(256, 2) void sr_pw_tiled_kernel(SrPtParams p) {
  constexpr int BM = 32 * WM, BN = 32 * WN * NTW;
  constexpr int A_FLOATS = BM * PT_AROW, B_FLOATS = (PT_KS / 8) * 2 * BN * 4;
  constexpr int A_PER_THREAD = BM * (PT_KS / 4) / 256;      // float4 per thread and slab
  constexpr int B_PER_THREAD = (PT_KS / 8) * 2 * BN / 256;
  static_assert(WM * WN == 4 && A_PER_THREAD >= 1 && B_PER_THREAD >= 1, "4 waves; whole float4s per thread");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kk = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int kpart = blockIdx.y;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int s_begin = kpart * p.slabs_per_part, s_end = min(p.slabs, s_begin + p.slabs_per_part);

  // ---- staging roles ----
  // A: float4 index e = tid + 256 u -> row e / 8, channel quad e % 8
  int a_row[A_PER_THREAD];
  const float* a_ptr[A_PER_THREAD];
  const float* g_ptr[A_PER_THREAD];
  const int a_q = tid & 7;
#pragma unroll
  for (int u = 0; u < A_PER_THREAD; ++u) {
    a_row[u] = (tid >> 3) + 32 * u;
    const int m = m0 + a_row[u];
    const bool ok = m < p.M;
    a_ptr[u] = ok ? p.in + (int64_t)m * p.in_sp + 4 * a_q : nullptr;
    g_ptr[u] = (GATE && ok) ? p.gate + (int64_t)(m / p.HW) * p.Cin + 4 * a_q : nullptr;
  }
  // B: float4 index e = tid + 256 u -> (group, half) = e / BN, column e % BN
  auto stage_load = [&](int s, pt_f4 (&ra)[A_PER_THREAD], pt_f4 (&rb)[B_PER_THREAD]) {
    const int k0 = s * PT_KS;
#pragma unroll
    for (int u = 0; u < A_PER_THREAD; ++u) {
      pt_f4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ptr[u] && k0 + 4 * a_q < p.Cin) {
        v = *reinterpret_cast<const pt_f4*>(a_ptr[u] + k0);
        if (GATE) v = v * *reinterpret_cast<const pt_f4*>(g_ptr[u] + k0);
      }
      ra[u] = v;
    }
#pragma unroll
    for (int u = 0; u < B_PER_THREAD; ++u) {
      const int e = tid + 256 * u;
      const int gh = e / BN, col = e - gh * BN;      // gh = 2 * local group + half
      const int g = s * (PT_KS / 8) + (gh >> 1);
      pt_f4 v = {0.f, 0.f, 0.f, 0.f};
      if (g < p.G && n0 + col < p.Co_pad)
        v = *reinterpret_cast<const pt_f4*>(p.wp + ((int64_t)(g * 2 + (gh & 1)) * p.Co_pad + n0 + col) * 4);
      rb[u] = v;
    }
  };
  auto stage_store = [&](int buf, const pt_f4 (&ra)[A_PER_THREAD], const pt_f4 (&rb)[B_PER_THREAD]) {
    float* A = lds + buf * (A_FLOATS + B_FLOATS);
    float* Bt = A + A_FLOATS;
#pragma unroll
    for (int u = 0; u < A_PER_THREAD; ++u) *reinterpret_cast<pt_f4*>(&A[a_row[u] * PT_AROW + 4 * a_q]) = ra[u];
#pragma unroll
    for (int u = 0; u < B_PER_THREAD; ++u) *reinterpret_cast<pt_f4*>(&Bt[(tid + 256 * u) * 4]) = rb[u];
  };

  pt_f16 acc[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

  pt_f4 ra[A_PER_THREAD], rb[B_PER_THREAD];
  if (s_begin < s_end) {
    stage_load(s_begin, ra, rb);
    stage_store(0, ra, rb);
  }
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    const bool more = s + 1 < s_end;
    if (more) stage_load(s + 1, ra, rb);                 // in flight under this slab's MFMAs
    const float* A = lds + buf * (A_FLOATS + B_FLOATS);
    const float* Bt = A + A_FLOATS;
    const float* a_base = A + (32 * wm + i) * PT_AROW + 4 * kk;
    const float* b_base = Bt + (kk * BN + 32 * wn * NTW + i) * 4;
#pragma unroll
    for (int gl = 0; gl < PT_KS / 8; ++gl) {
      const pt_f4 a = *reinterpret_cast<const pt_f4*>(a_base + 8 * gl);
      pt_f4 b[NTW];
#pragma unroll
      for (int n = 0; n < NTW; ++n) b[n] = *reinterpret_cast<const pt_f4*>(b_base + (gl * 2 * BN + 32 * n) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[n][e], acc[n], 0, 0, 0);
    }
    if (more) stage_store(buf ^ 1, ra, rb);   // (buffer buf ^ 1 was last read one iteration ago, a barrier back)
    __syncthreads();
  }

  // ---- epilogue.  Lane (i, kk) holds column n0 + 32 (wn NTW + n) + i of rows m0 + 32 wm + (r & 3) + 8 (r >> 2) + 4 kk.
  const bool partial = p.ksplit > 1;
  const float slope = sr_uniform(partial ? -1.0f : p.slope);
  float* outp = p.out + (partial ? (int64_t)kpart * p.part_stride : (int64_t)0);
  const int rbase = m0 + 32 * wm + 4 * kk;
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int col = n0 + 32 * (wn * NTW + n) + i;
    const bool okc = col < p.Cout;
    const float bv = (!partial && p.bias && okc) ? p.bias[col] : 0.0f;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[n][r] + bv;
    if (!partial && p.res) {
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rbase + (r & 3) + 8 * (r >> 2);
        rv[r] = (okc && m < p.M) ? p.res[(int64_t)m * p.res_sp + col] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += rv[r];
    }
    sr_activate_group(v, slope);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = rbase + (r & 3) + 8 * (r >> 2);
      if (okc && m < p.M) outp[(int64_t)m * p.out_sp + col] = v[r];
    }
  }
}

int pt_num_cus() { return sr_device_cus(); }

struct PtPlan { int cfg, ks; };   // cfg: 0 = 64 x 128 (2x2 waves, 2 tiles each), 1 = 128 x 160 (4x1, 5), 2 = 128 x 64 (4x1, 2), 3 = 64 x 64 (2x2, 1)
constexpr int PT_BM[4] = {64, 128, 128, 64}, PT_BN[4] = {128, 160, 64, 64};

// Tile shape by padding waste over the channel blocks, K split so that the grid makes ~1.5 workgroups per CU while a part
// keeps >= 4 slabs.  SR_PT_CFG / SR_PT_KS force a plan (tests, sweeps).
PtPlan pt_plan(int M, int Cin, int Cout, bool can_split) {
  const int f_cfg = sr_opt(SR_OPT_PT_CFG), f_ks = sr_opt(SR_OPT_PT_KS);   // forced plan (tests, sweeps)
  const int slabs = (Cin + PT_KS - 1) / PT_KS;
  const long want = (long)pt_num_cus() * 3 / 2;
  PtPlan best = {0, 1};
  double best_cost = -1.0;
  for (int cfg = 0; cfg < 4; ++cfg) {
    if (f_cfg >= 0 && cfg != f_cfg) continue;
    const long tm = (M + PT_BM[cfg] - 1) / PT_BM[cfg], tn = (Cout + PT_BN[cfg] - 1) / PT_BN[cfg];
    const double pad = (double)(tm * PT_BM[cfg]) * (double)(tn * PT_BN[cfg]) / ((double)M * Cout);
    for (int ks = 1; ks <= 8; ks *= 2) {
      if (f_ks > 0 && ks != f_ks) continue;
      if (ks > 1 && (!can_split || slabs / ks < 4)) continue;
      const long wgs = tm * tn * ks;
      const double fill = wgs >= want ? 1.0 : (double)want / (double)wgs;
      // operand bytes per MFMA fall with the tile area; a split costs the partial round trip + one more launch
      const double reuse = 1.0 + 24.0 / (PT_BM[cfg] * PT_BN[cfg] / (double)(PT_BM[cfg] + PT_BN[cfg]));
      const double cost = pad * fill * reuse * (1.0 + (ks > 1 ? 0.10 + 2.0 * ks / slabs : 0.0));
      if (best_cost < 0 || cost < best_cost) { best = {cfg, ks}; best_cost = cost; }
    }
  }
  return best;
}

template <int WM, int WN, int NTW>
int pt_launch(const SrPtParams& p, int tiles_m, hipStream_t stream) {
  constexpr int BM = 32 * WM, BN = 32 * WN * NTW;
  const size_t lds = (size_t)2 * (BM * PT_AROW + (PT_KS / 8) * 2 * BN * 4) * sizeof(float);
  const dim3 grid((unsigned)(tiles_m * p.tiles_n), (unsigned)p.ksplit);
  if (p.gate) {
    hipError_t e = hipFuncSetAttribute((const void*)sr_pw_tiled_kernel<WM, WN, NTW, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return sr_hip_rc(e);
    hipLaunchKernelGGL((sr_pw_tiled_kernel<WM, WN, NTW, true>), grid, dim3(256), lds, stream, p);
  } else {
    hipError_t e = hipFuncSetAttribute((const void*)sr_pw_tiled_kernel<WM, WN, NTW, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return sr_hip_rc(e);
    hipLaunchKernelGGL((sr_pw_tiled_kernel<WM, WN, NTW, false>), grid, dim3(256), lds, stream, p);
  }
  return sr_hip_rc(hipGetLastError());
}

}  // namespace

// Workspace for the K-split plan of a shape (0 when the plan does not split).
extern "C" size_t sr_pw_conv_tiled_workspace_bytes(int M, int Cin, int Cout) {
  if (M <= 0 || Cin <= 0 || Cout <= 0 || Cout % 4 != 0) return 0;
  const PtPlan pl = pt_plan(M, Cin, Cout, true);
  return pl.ks > 1 ? (size_t)pl.ks * M * Cout * sizeof(float) : 0;
}

extern "C" int sr_pw_conv_tiled_plan(int M, int Cin, int Cout, int can_split, int* cfg, int* ks) {
  if (M <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  const PtPlan pl = pt_plan(M, Cin, Cout, can_split != 0 && Cout % 4 == 0);
  if (cfg) *cfg = pl.cfg;
  if (ks) *ks = pl.ks;
  return SR_OK;
}

extern "C" int sr_pw_conv_tiled_nhwc_fwd(const float* in, int in_pix_stride, const float* packed_w, const float* bias,
                                         const float* gate, const float* residual, int res_pix_stride, float* out,
                                         int out_pix_stride, int M, int HW, int Cin, int Cout, float act_code,
                                         void* workspace, size_t workspace_bytes, void* stream_) {
  if (M < 0 || HW <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (M == 0) return SR_OK;
  if (!in || !packed_w || !out) return SR_ERR_INVALID_ARGUMENT;
  if (Cin % 4 != 0 || in_pix_stride % 4 != 0 || (((uintptr_t)in) & 15) != 0 || (gate && (((uintptr_t)gate) & 15) != 0))
    return SR_ERR_UNSUPPORTED;   // 16-byte staging loads
  const bool can_split = workspace && (((uintptr_t)workspace) & 15) == 0 && Cout % 4 == 0 &&
                         (!residual || ((((uintptr_t)residual) & 15) == 0 && res_pix_stride % 4 == 0)) &&
                         (((uintptr_t)out) & 15) == 0 && out_pix_stride % 4 == 0 && (!bias || (((uintptr_t)bias) & 15) == 0);
  PtPlan pl = pt_plan(M, Cin, Cout, can_split);
  if (pl.ks > 1 && workspace_bytes < (size_t)pl.ks * M * Cout * sizeof(float)) pl = pt_plan(M, Cin, Cout, false);
  SrPtParams p;
  p.in = in; p.in_sp = in_pix_stride; p.wp = packed_w; p.bias = bias; p.gate = gate;
  p.res = residual; p.res_sp = res_pix_stride;
  p.M = M; p.HW = HW; p.Cin = Cin; p.Cout = Cout;
  p.Co_pad = ((Cout + 31) / 32) * 32;
  p.G = ((Cin + 63) / 64) * 8;
  p.slabs = (Cin + PT_KS - 1) / PT_KS;
  p.ksplit = pl.ks;
  p.slabs_per_part = (p.slabs + pl.ks - 1) / pl.ks;
  p.slope = act_code;
  p.part_stride = (int64_t)M * Cout;
  if (pl.ks > 1) { p.out = (float*)workspace; p.out_sp = Cout; }
  else { p.out = out; p.out_sp = out_pix_stride; }
  const int tiles_m = (M + PT_BM[pl.cfg] - 1) / PT_BM[pl.cfg];
  p.tiles_n = (Cout + PT_BN[pl.cfg] - 1) / PT_BN[pl.cfg];
  hipStream_t stream = (hipStream_t)stream_;
  int rc;
  switch (pl.cfg) {
    case 0: rc = pt_launch<2, 2, 2>(p, tiles_m, stream); break;
    case 1: rc = pt_launch<4, 1, 5>(p, tiles_m, stream); break;
    case 2: rc = pt_launch<4, 1, 2>(p, tiles_m, stream); break;
    default: rc = pt_launch<2, 2, 1>(p, tiles_m, stream); break;
  }
  if (rc == SR_OK && pl.ks > 1)   // out = act(sum_k partial[k] + bias + residual), partials added in index order
    rc = sr_launch_splitk_reduce((const float*)workspace, pl.ks, p.part_stride, bias, residual, (int64_t)0, res_pix_stride,
                                 out, (int64_t)0, out_pix_stride, 1, M, Cout, act_code, stream);
  return rc;
}
