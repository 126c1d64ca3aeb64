// sr_pw.hip -- pointwise (1x1) convolutions over dense channels-last maps as a hand-written fp32-MFMA GEMM for gfx950.
//
//   out[b, p, co] = act( sum_ci  gate[b, ci] * in[b, p, ci] * W[co, ci]  + bias[co] + residual[b, p, co] )
//
// The operator behind BasicBlock's 1x1 skip convolution (reference modules/layers.py:20-22, 57-62) and behind the MBConv
// expand / project convolutions of the image-prior encoder (reference experiment_modules/depth_model.py:110-116: timm
// tf_efficientnetv2_s), where `gate` is the squeeze-excite gate of the block (sigmoid(W2 silu(W1 mean)), one value per
// image and INPUT channel of the projection) -- applied while the A operand is loaded, so the gated activation tensor is
// never written.  r02 / r03 handed these GEMMs to hipBLASLt (sr_gemm1x1.hip), which picked its algorithm by timing, per
// process: fast, but two processes were not bit-identical.  This kernel is deterministic (fixed reduction order), serves the
// short-K / short-N / small-M shapes of the encoder with a launch plan of its own, and keeps the library behind a switch.
//
// GEMM view: M = pixels (a 32-row MFMA tile never straddles two images: ceil(HW / 32) tiles per image), N = output
// channels, K = input channels, v_mfma_f32_32x32x2_f32 (fp32 products, fp32 accumulation).  A wave owns a 32 x (32 NT)
// output tile.  Per group of 8 input channels it loads
//   A   one float4 per lane (row i, channels 8g + 4kk ..) straight from global memory -- the 32 rows of a tile are 32
//       cache lines that the next three groups hit again in L1; no LDS staging, no barrier in the K loop,
//   B   NT float4 per lane from the weight packed in B-fragment order (sr_conv_pack_weights, ksize 1: a wave's load is
//       two contiguous 512-byte pieces, shared by every wave that works on the same channel block),
// and issues 4 NT MFMAs; loads run PD groups ahead of the MFMAs through U rotating register sets.  Launch plan: with many
// tiles (decoder skips: M = 614 400) the 4 waves of a workgroup take 4 consecutive pixel tiles; with few (encoder stage 5:
// M = 2 400) they split the K range 2 or 4 ways and add their accumulators through LDS in a FIXED order (no atomics).
#include <stdlib.h>

#include "sr_common.h"

namespace {

typedef float pw_f16 __attribute__((ext_vector_type(16)));
typedef float pw_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int pw_u4 __attribute__((ext_vector_type(4)));
#define PW_RSRC_FLAGS 0x00020000
#define PW_OOB 0x7fffffffu

__device__ __forceinline__ pw_f4 pw_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(pw_f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// 16-bit activation I/O (IO = 1: fp16, 2: bf16; 0 = fp32): input / residual / output elements are 2 bytes in HBM, widened on
// load and rounded to nearest even on store; gate, weights, bias and the accumulation stay fp32 (training under autocast).
typedef unsigned int pw_u2 __attribute__((ext_vector_type(2)));
template <int IO>
__device__ __forceinline__ float pw_widen(unsigned short h) {
  if (IO == 1) return (float)__builtin_bit_cast(_Float16, h);
  return __builtin_bit_cast(float, (unsigned)h << 16);
}
template <int IO>
__device__ __forceinline__ unsigned short pw_narrow(float f) {
  if (IO == 1) return __builtin_bit_cast(unsigned short, (_Float16)f);
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
template <int IO>
__device__ __forceinline__ pw_f4 pw_load_a(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if (IO == 0) return __builtin_bit_cast(pw_f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
  const pw_u2 v = __builtin_bit_cast(pw_u2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
  return pw_f4{pw_widen<IO>((unsigned short)(v.x & 0xffffu)), pw_widen<IO>((unsigned short)(v.x >> 16)),
               pw_widen<IO>((unsigned short)(v.y & 0xffffu)), pw_widen<IO>((unsigned short)(v.y >> 16))};
}
template <int IO>
__device__ __forceinline__ float pw_load_elem(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if (IO == 0) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
  return pw_widen<IO>(__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0));
}
template <int IO>
__device__ __forceinline__ void pw_store_elem(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if (IO == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
  else __builtin_amdgcn_raw_buffer_store_b16(pw_narrow<IO>(v), r, (int)voff, (int)soff, 0);
}
#define PW_ES(io) ((io) == 0 ? 4u : 2u)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pw_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, PW_RSRC_FLAGS);
}

struct SrPwParams {
  const float* in; int64_t in_sb; int in_sp;
  const float* wp;                   // packed [G][2][Co_pad][4] (sr_conv_pack_weights, ksize 1)
  const float* bias;                 // [Cout] or null
  const float* gate;                 // [B][Cin] or null
  const float* res; int64_t res_sb; int res_sp;
  float* out; int64_t out_sb; int out_sp;
  int B, HW, Cin, Cout, Co_pad, G, G8;   // G = packed groups (Cin padded to 64), G8 = ceil(Cin / 8)
  int m_tiles, n_blocks, total_mt;      // pixel tiles per image, channel blocks, B * m_tiles
  float slope;
};

#ifndef SR_PW_U
#define SR_PW_U 4
#define SR_PW_PD 3
#endif
constexpr int PW_U = SR_PW_U;    // rotating operand register sets (the K loop is unrolled by U)
constexpr int PW_PD = SR_PW_PD;  // loads run PD groups ahead of their MFMAs

template <int NT, int KS, bool GATE, int IO = 0>
__global__ __launch_bounds__(256, 2) void sr_pw_kernel(SrPwParams p) {
  constexpr unsigned ES = PW_ES(IO);   // bytes per activation element
  extern __shared__ __attribute__((aligned(16))) float red[];   // KS > 1: [m local][KS - 1][NT][16][64]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kk = lane >> 5;
  constexpr int MT_PER_WG = 4 / KS;
  const int ml = wave / KS, kp = wave % KS;          // local pixel tile, K part
  const int item = blockIdx.x;
  const int nb = item % p.n_blocks;
  const int mt_g = (item / p.n_blocks) * MT_PER_WG + ml;
  const bool live = mt_g < p.total_mt;               // (wave-uniform) a ragged last workgroup has idle waves
  const int img = live ? mt_g / p.m_tiles : 0;
  const int row0 = live ? (mt_g - img * p.m_tiles) * 32 : 0;
  const int n0 = nb * (32 * NT);

  // K range of this wave: groups [g0, g1) of 8 input channels, a multiple of U long (steps past g1 load zeros)
  const int per = ((p.G8 + KS - 1) / KS + PW_U - 1) / PW_U * PW_U;
  const int g0 = kp * per;
  const int g1 = min(p.G8, g0 + per);
  const int steps = live && g1 > g0 ? (g1 - g0 + PW_U - 1) / PW_U * PW_U : 0;

  const __amdgpu_buffer_rsrc_t rs_in = pw_rsrc(reinterpret_cast<const char*>(p.in) + (int64_t)img * p.in_sb * ES,
                                               ((int64_t)(p.HW - 1) * p.in_sp + p.Cin) * ES);
  const __amdgpu_buffer_rsrc_t rs_w = pw_rsrc(p.wp, (int64_t)p.G * 2 * p.Co_pad * 16);
  const __amdgpu_buffer_rsrc_t rs_g = pw_rsrc(GATE ? p.gate + (int64_t)img * p.Cin : p.wp, GATE ? (int64_t)p.Cin * 4 : 0);
  const int row = row0 + i;
  // lane offsets (bytes): the group index rides in the scalar offset.  A lane whose row is past the image, or whose
  // channel quad is past Cin (Cin % 8 == 4, last group), reads through an out-of-range offset: 0.
  const unsigned a_voff = (row < p.HW) ? ((unsigned)row * (unsigned)p.in_sp + 4u * kk) * ES : PW_OOB;
  const bool k_tail = (p.Cin & 7) != 0;
  const unsigned w_voff = (unsigned)(kk * p.Co_pad + n0 + i) * 16u;
  const unsigned g_voff = 16u * kk;
  const int g_last_w = p.G - 1;   // weight groups past the packed array are read from its last group (their A operand is 0)

  pw_f16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

  pw_f4 a_f[PW_U], b_f[PW_U][NT], s_f[PW_U];
  auto issue = [&](int g, int slot) {   // loads of group g (wave-uniform): A, gate, NT weight fragments
    const bool in_range = g < g1;
    unsigned av = in_range ? a_voff : PW_OOB;
    if (k_tail && 8 * g + 4 * kk + 4 > p.Cin) av = PW_OOB;
    a_f[slot] = pw_load_a<IO>(rs_in, av, (unsigned)g * 8u * ES);
    if (GATE) s_f[slot] = pw_load(rs_g, g_voff + (unsigned)g * 32u, 0u);   // (lane offset: past Cin -> out of range -> 0)
    const unsigned gw = (unsigned)min(g, g_last_w);
#pragma unroll
    for (int n = 0; n < NT; ++n) b_f[slot][n] = pw_load(rs_w, w_voff + 512u * n, gw * 2u * (unsigned)p.Co_pad * 16u);
  };
  if (steps > 0) {
#pragma unroll
    for (int s = 0; s < PW_PD; ++s) issue(g0 + s, s);
  }
  for (int gb = 0; gb < steps; gb += PW_U) {
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      issue(g0 + gb + u + PW_PD, (u + PW_PD) % PW_U);
      pw_f4 a = a_f[u];
      if (GATE) a = a * s_f[u];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b_f[u][n][e], acc[n], 0, 0, 0);
    }
  }

  if (KS > 1) {   // K parts 1 .. KS-1 hand their accumulators to part 0, which adds them in index order
    if (kp > 0 && live) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((ml * (KS - 1) + kp - 1) * NT + n) * 16 + r) * 64 + lane] = acc[n][r];
    }
    __syncthreads();
    if (kp == 0 && live) {
#pragma unroll
      for (int q = 0; q < KS - 1; ++q)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[n][r] += red[(((ml * (KS - 1) + q) * NT + n) * 16 + r) * 64 + lane];
    }
  }
  if (kp != 0 || !live) return;

  // ---- epilogue: + bias + residual, activation, store.  Lane (i, kk) holds column n0 + 32 n + i of rows
  // row0 + (r & 3) + 8 (r >> 2) + 4 kk: a wave's store covers 2 rows x 128 contiguous bytes.
  const float slope = sr_uniform(p.slope);
  const char* resb = p.res ? reinterpret_cast<const char*>(p.res) + (int64_t)img * p.res_sb * ES : nullptr;
  const __amdgpu_buffer_rsrc_t rs_out = pw_rsrc(reinterpret_cast<char*>(p.out) + (int64_t)img * p.out_sb * ES,
                                                ((int64_t)(p.HW - 1) * p.out_sp + p.Cout) * ES);
  const __amdgpu_buffer_rsrc_t rs_res = pw_rsrc(resb ? (const void*)resb : (const void*)p.wp,
                                                resb ? ((int64_t)(p.HW - 1) * p.res_sp + p.Cout) * ES : (int64_t)0);
  const bool full = (row0 + 32 <= p.HW) & (n0 + 32 * NT <= p.Cout);   // uniform
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + 32 * n + i;
    const bool okc = col < p.Cout;
    const float bv = (p.bias && okc) ? p.bias[col] : 0.0f;
    const unsigned o_base = ((unsigned)(row0 + 4 * kk) * (unsigned)p.out_sp + (unsigned)col) * ES;
    const unsigned r_base = ((unsigned)(row0 + 4 * kk) * (unsigned)p.res_sp + (unsigned)col) * ES;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[n][r] + bv;
    if (resb) {
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        const bool ok = full || (okc && row0 + 4 * kk + dr < p.HW);
        rv[r] = pw_load_elem<IO>(rs_res, ok ? r_base : PW_OOB, (unsigned)dr * (unsigned)p.res_sp * ES);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += rv[r];
    }
    sr_activate_group(v, slope);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      const bool ok = full || (okc && row0 + 4 * kk + dr < p.HW);
      pw_store_elem<IO>(v[r], rs_out, ok ? o_base : PW_OOB, (unsigned)dr * (unsigned)p.out_sp * ES);
    }
  }
}

int pw_num_cus() { return sr_device_cus(); }

struct PwPlan { int nt, ks; };

// Launch plan: the widest channel block (fewest A re-reads) that still makes >= ~1.5 waves per SIMD; below that, split K
// across the waves of a workgroup.  SR_PW_NT / SR_PW_KS force a plan (tests, sweeps).
PwPlan pw_plan(int B, int HW, int Cin, int Cout) {
  const int f_nt = sr_opt(SR_OPT_PW_NT), f_ks = sr_opt(SR_OPT_PW_KS);   // forced plan (tests, sweeps); 0 = the cost model
  const long mt = (long)B * ((HW + 31) / 32);
  const int n32 = (Cout + 31) / 32;
  const int g8 = (Cin + 7) / 8;
  const long want = (long)pw_num_cus() * 4 * 3 / 2;
  PwPlan best = {1, 1};
  double best_cost = -1.0;
  const int nts[4] = {4, 2, 1, 5};
  for (int a = 0; a < 4; ++a) {
    const int nt = nts[a];
    if (nt == 5 && n32 % 5 != 0) continue;            // (160 / 960 output channels: five 32-wide tiles, no padding)
    if (f_nt > 0 && nt != f_nt) continue;
    for (int ks = 1; ks <= 4; ks *= 2) {
      if (f_ks > 0 && ks != f_ks) continue;
      if (ks > 1 && g8 / ks < 8) continue;            // a K part shorter than 8 groups is all pipeline fill
      if (ks > 1 && nt > 2) continue;                  // (LDS reduce buffer: keep it <= 24 KB)
      if (nt == 5 && ks > 1) continue;
      const int nblocks = (n32 + nt - 1) / nt;
      const long waves = mt * nblocks * ks;
      const double pad = (double)(nblocks * nt) / n32;                         // padded channel tiles
      const double fill = waves >= want ? 1.0 : (double)want / (double)waves;  // idle SIMDs
      // A traffic per MFMA falls with nt; the reduce costs about one extra K group per part
      const double cost = pad * fill * (1.0 + 0.25 / nt) * (1.0 + (ks > 1 ? 1.5 * ks / g8 : 0.0));
      if (best_cost < 0 || cost < best_cost) { best = {nt, ks}; best_cost = cost; }
    }
  }
  return best;
}

template <int NT, int KS>
int pw_launch(const SrPwParams& p, hipStream_t stream, int io) {
  const int mt_per_wg = 4 / KS;
  const long wgs = ((long)p.total_mt + mt_per_wg - 1) / mt_per_wg * p.n_blocks;
  const size_t lds = KS > 1 ? (size_t)mt_per_wg * (KS - 1) * NT * 16 * 64 * sizeof(float) : 0;
  if (io == 1) hipLaunchKernelGGL((sr_pw_kernel<NT, KS, false, 1>), dim3((unsigned)wgs), dim3(256), lds, stream, p);
  else if (io == 2) hipLaunchKernelGGL((sr_pw_kernel<NT, KS, false, 2>), dim3((unsigned)wgs), dim3(256), lds, stream, p);
  else if (p.gate) hipLaunchKernelGGL((sr_pw_kernel<NT, KS, true>), dim3((unsigned)wgs), dim3(256), lds, stream, p);
  else hipLaunchKernelGGL((sr_pw_kernel<NT, KS, false>), dim3((unsigned)wgs), dim3(256), lds, stream, p);
  return sr_hip_rc(hipGetLastError());
}

}  // namespace

// 1 when sr_pw_conv_nhwc_fwd serves the shape (channel counts in whole float4 quads); otherwise callers use
// sr_conv2d_nhwc_fwd.
extern "C" int sr_pw_conv_supported(int Cin, int Cout) { return (Cin > 0 && Cout > 0 && Cin % 4 == 0) ? 1 : 0; }

extern "C" int sr_pw_conv_plan(int B, int HW, int Cin, int Cout, int* nt, int* ks) {
  if (B <= 0 || HW <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  const PwPlan pl = pw_plan(B, HW, Cin, Cout);
  if (nt) *nt = pl.nt;
  if (ks) *ks = pl.ks;
  return SR_OK;
}

static int pw_run(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_w,
                  const float* bias, const float* gate, const float* residual, int64_t res_batch_stride,
                  int res_pix_stride, float* out, int64_t out_batch_stride, int out_pix_stride, int B,
                  int HW, int Cin, int Cout, float act_code, void* stream_, int io) {
  if (B < 0 || HW <= 0 || Cin <= 0 || Cout <= 0 || io < 0 || io > 2) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !packed_w || !out) return SR_ERR_INVALID_ARGUMENT;
  if (io && gate) return SR_ERR_UNSUPPORTED;   // (the gated projection is an inference path: fp32)
  if (Cin % 4 != 0 || in_pix_stride % 4 != 0 || in_batch_stride % 4 != 0 || (((uintptr_t)in) & (io ? 7 : 15)) != 0 ||
      (gate && (((uintptr_t)gate) & 15) != 0))
    return SR_ERR_UNSUPPORTED;   // 4-channel A fragments (16 bytes of fp32, 8 of fp16 / bf16)
  const int64_t lim = (int64_t)1 << 31;
  if (((int64_t)(HW - 1) * in_pix_stride + Cin) * 4 >= lim || ((int64_t)(HW - 1) * out_pix_stride + Cout) * 4 >= lim ||
      (residual && ((int64_t)(HW - 1) * res_pix_stride + Cout) * 4 >= lim))
    return SR_ERR_UNSUPPORTED;   // per-image byte offsets are 32-bit
  SrPwParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.wp = packed_w; p.bias = bias; p.gate = gate;
  p.res = residual; p.res_sb = res_batch_stride; p.res_sp = res_pix_stride;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.B = B; p.HW = HW; p.Cin = Cin; p.Cout = Cout;
  p.Co_pad = ((Cout + 31) / 32) * 32;
  p.G = ((Cin + 63) / 64) * 8;
  p.G8 = (Cin + 7) / 8;
  p.m_tiles = (HW + 31) / 32;
  p.total_mt = B * p.m_tiles;
  p.slope = act_code;
  const PwPlan pl = pw_plan(B, HW, Cin, Cout);
  p.n_blocks = (p.Co_pad / 32 + pl.nt - 1) / pl.nt;
  hipStream_t stream = (hipStream_t)stream_;
  switch (pl.nt * 10 + pl.ks) {
    case 11: return pw_launch<1, 1>(p, stream, io);
    case 12: return pw_launch<1, 2>(p, stream, io);
    case 14: return pw_launch<1, 4>(p, stream, io);
    case 21: return pw_launch<2, 1>(p, stream, io);
    case 22: return pw_launch<2, 2>(p, stream, io);
    case 24: return pw_launch<2, 4>(p, stream, io);
    case 41: return pw_launch<4, 1>(p, stream, io);
    case 51: return pw_launch<5, 1>(p, stream, io);
    default: return SR_ERR_UNSUPPORTED;
  }
}

extern "C" int sr_pw_conv_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_w,
                                   const float* bias, const float* gate, const float* residual, int64_t res_batch_stride,
                                   int res_pix_stride, float* out, int64_t out_batch_stride, int out_pix_stride, int B,
                                   int HW, int Cin, int Cout, float act_code, void* stream_) {
  return pw_run(in, in_batch_stride, in_pix_stride, packed_w, bias, gate, residual, res_batch_stride, res_pix_stride, out,
                out_batch_stride, out_pix_stride, B, HW, Cin, Cout, act_code, stream_, 0);
}

// The same operator on fp16 (io_dtype = 1) / bf16 (2) activation tensors (input, residual, output; strides in ELEMENTS);
// weights, bias and the accumulation stay fp32.  No gate (the gated projection is an inference path).
extern "C" int sr_pw_conv_io_nhwc_fwd(const void* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_w,
                                      const float* bias, const void* residual, int64_t res_batch_stride, int res_pix_stride,
                                      void* out, int64_t out_batch_stride, int out_pix_stride, int B, int HW, int Cin, int Cout,
                                      float act_code, int io_dtype, void* stream_) {
  return pw_run((const float*)in, in_batch_stride, in_pix_stride, packed_w, bias, nullptr, (const float*)residual,
                res_batch_stride, res_pix_stride, (float*)out, out_batch_stride, out_pix_stride, B, HW, Cin, Cout, act_code,
                stream_, io_dtype);
}
