// sr_wino_split.hip -- split-precision variant of the Winograd F(2x2, 3x3) convolution (gfx950).  A FENCED EXPERIMENT:
// selected only by SR_WINO_SPLIT=bf16|f16 (read per call), never the default, never what bench.py's headline measures
// (DESIGN.md 3.3e).  Same operator, tensors, launch plan, LDS staging, in-register input transform and epilogue as
// sr_wino_kernel<NT, true, true> (sr_wino.hip; reference modules/layers.py:24-85); what changes is the multiply:
//   * every fp32 operand -- the transformed input V = B^T d B in the kernel, the transformed weight U = G g G^T at pack
//     time -- is written as two 16-bit pieces, x = x_hi + x_lo (round-to-nearest each, the subtraction is exact);
//   * a product is three products on the 16-bit matrix pipe, x_hi w_hi + x_hi w_lo + x_lo w_hi, exact in the fp32
//     accumulator of v_mfma_f32_32x32x16_{bf16,f16} (16 x the fp32 MFMA rate -> 3/16 of the fp32 MFMA time);
//   * a 16-channel slab is ONE k-step: lane (tile i, kk) holds channels 4 kk + e of both 8-channel groups -- the eight
//     A values it already has in registers -- so per slab a wave issues 4 frequencies x NT x 3 MFMAs of 32 cycles
//     instead of 8 x 4 x NT of 64.
// The packed weight keeps the fp32 layout's size and addressing: record (xi, 2 ch) holds the hi pieces of slab ch,
// record (xi, 2 ch + 1) the lo pieces, 16 bytes per lane each.
#include <string.h>
#include <type_traits>

#include "sr_wino.h"

typedef unsigned ws_u4 __attribute__((ext_vector_type(4)));

// WS_USCALE: the packed weight pieces hold U * 2^k (exact), the epilogue multiplies by 2^-k where it adds the bias.  fp16
// pieces of |x| < 2^-2 have a denormal low piece (absolute resolution 2^-25 instead of relative 2^-24): typical weights
// (|U| ~ 0.05) would lose 3-4 bits; scaled by 2^8 every |U| >= 2^-10 keeps full precision and |U| < 255 stays in range.
template <int FMT> struct WsFmt;
template <> struct WsFmt<1> {
  static constexpr float USCALE = 1.0f;
  typedef __bf16 e2 __attribute__((ext_vector_type(2)));
  typedef __bf16 e8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void split(float a, float b, unsigned& hi, unsigned& lo) {
    const wn_f2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, e2));
    const wn_f2 v0 = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - v0, e2));
  }
  static __device__ __forceinline__ f32x16 mfma(ws_u4 a, ws_u4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(e8, a), __builtin_bit_cast(e8, b), c, 0, 0, 0);
  }
};
template <> struct WsFmt<2> {   // fp16 pieces: 22-24 bits inside 2^-14 <= |x| < 65504; beyond that range the result is inf / NaN
  static constexpr float USCALE = 256.0f;
  typedef _Float16 e2 __attribute__((ext_vector_type(2)));
  typedef _Float16 e8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void split(float a, float b, unsigned& hi, unsigned& lo) {
    const wn_f2 v = {a, b};
    const e2 h = __builtin_convertvector(v, e2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v - __builtin_convertvector(h, wn_f2), e2));
  }
  static __device__ __forceinline__ f32x16 mfma(ws_u4 a, ws_u4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(e8, a), __builtin_bit_cast(e8, b), c, 0, 0, 0);
  }
};

// U = G g G^T per (co, ci) as 16-bit pieces in B-fragment order: word (xi, g8 = 2 ch + piece, kk, co, dw) holds the pieces of
// U_xi[co][16 ch + 8 (j >> 2) + 4 kk + (j & 3)], j = 2 dw, 2 dw + 1 -- the k-slot order of the kernel's A operand.
template <int FMT>
__global__ void sr_wino_pack_split_kernel(const float* __restrict__ w, unsigned* __restrict__ wu, int Co, int Ci, int G,
                                          int Co_pad) {
  const int64_t total = (int64_t)16 * G * 2 * Co_pad * 4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int dw = (int)(e & 3);
    int64_t r = e >> 2;
    const int co = (int)(r % Co_pad); r /= Co_pad;
    const int kk = (int)(r & 1); r >>= 1;
    const int g8 = (int)(r % G);
    const int xi = (int)(r / G);
    const int ch = g8 >> 1, piece = g8 & 1;
    float u[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = 2 * dw + t;
      const int ci = 16 * ch + 8 * (j >> 2) + 4 * kk + (j & 3);
      float v = 0.0f;
      if (co < Co && ci < Ci) {
        const float* g = w + ((int64_t)co * Ci + ci) * 9;
        const int ur = xi >> 2, uc = xi & 3;
        float rowv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
          rowv[c] = ur == 0 ? g0 : (ur == 1 ? (g0 + g1 + g2) * 0.5f : (ur == 2 ? (g0 - g1 + g2) * 0.5f : g2));
        }
        v = uc == 0 ? rowv[0]
                    : (uc == 1 ? (rowv[0] + rowv[1] + rowv[2]) * 0.5f
                               : (uc == 2 ? (rowv[0] - rowv[1] + rowv[2]) * 0.5f : rowv[2]));
      }
      u[t] = v * WsFmt<FMT>::USCALE;
    }
    unsigned hi, lo;
    WsFmt<FMT>::split(u[0], u[1], hi, lo);
    wu[e] = piece ? lo : hi;
  }
}

#define WS_PIN(...) asm volatile("" : __VA_ARGS__)
#ifndef WS_ABL
#define WS_ABL 0   // phase ablation (tuning builds only; results are wrong): 1 no staging loads, 2 no residual loads / output
#endif             // stores, 4 no MFMAs, 8 no weight loads, 16 no input transform

#ifndef SR_WS_NT1_WGS
#define SR_WS_NT1_WGS 3   // workgroups per CU of the 32-channel instantiation: 3 fit its 46 KB of LDS and 168 registers (3 spills); the
#endif                    // latency-bound split kernel gains 9 - 18 % on the 60x80 / 30x40 layers from the third (2: 45.9 / 55.1 us, 3: 42.0 / 45.1)
template <int NT, int FMT>
__global__ __launch_bounds__(256, NT == 1 ? SR_WS_NT1_WGS : 2) void sr_wino_split_kernel(SrWinoParams p) {
  typedef WsFmt<FMT> S;
  constexpr unsigned ES = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* O = lds;
  float* rawA = lds + WN_V_FLOATS(NT);
  float* rawB = lds + WN_O_FLOATS(NT);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kk = lane >> 5;
  const int chunks = (p.G >> 1) / p.ksplit;
  const int64_t rec = (int64_t)2 * p.Co_pad;

  // transform role of wave w: frequency row ur = w of V = B^T d B (see sr_wino_kernel)
  const int t_ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1), t_rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float t_sign = wave == 1 ? 1.0f : -1.0f;
  const int rv_base = ((2 * (i >> 3)) * WN_PW + 2 * (i & 7)) * WN_ROW + 4 * kk;
  int rv_a = rv_base + t_ra * WN_PW * WN_ROW, rv_b = rv_base + t_rb * WN_PW * WN_ROW;

  struct Region { int b, oy0, ox0, co0, ks; };
  const int xcd_g = (int)gridDim.x;
  auto decode = [&](int wk) {
    if (p.xcd_order && (xcd_g & 7) == 0) {
      const int r0 = wk / xcd_g * xcd_g;
      if (r0 + xcd_g <= p.total) { const int bb = wk - r0; wk = r0 + (bb & 7) * (xcd_g >> 3) + (bb >> 3); }
    }
    Region r;
    r.ks = wk % p.ksplit; wk /= p.ksplit;
    const int cb = wk % p.co_blocks; wk /= p.co_blocks;
    const int rx = wk % p.regions_x; wk /= p.regions_x;
    const int ry = wk % p.regions_y;
    r.b = wk / p.regions_y;
    r.oy0 = ry * (2 * WN_TR); r.ox0 = rx * (2 * WN_TC); r.co0 = cb * (32 * NT);
    return r;
  };
  // ---- staging: global -> registers -> LDS (the vector path of sr_wino_kernel) ----
  int offs[WN_STAGE_PER_THREAD];
  unsigned s_org = 0;
  __amdgpu_buffer_rsrc_t rs_in = wn_rsrc(p.in, 0);
  const int64_t in_img_bytes = ((int64_t)(p.H * p.W - 1) * p.in_sp + p.Cin) * ES;
  const int c_quad = 4 * (tid & 3);
  auto aim = [&](const Region& r) {
    const char* in_b = reinterpret_cast<const char*>(p.in) + (int64_t)r.b * p.in_sb * ES;
    rs_in = wn_rsrc(in_b, in_img_bytes);
    const bool interior = (r.oy0 >= 1) & (r.oy0 + 2 * WN_TR + 1 <= p.H) & (r.ox0 >= 1) & (r.ox0 + 2 * WN_TC + 1 <= p.W);
    if (interior) {
      s_org = (unsigned)(((r.oy0 - 1) * p.W + (r.ox0 - 1)) * p.in_sp) * ES;
#pragma unroll
      for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) {
        const int e = tid + it * 256;
        const int px = e >> 2;
        const int py = px / WN_PW, pxx = px - py * WN_PW;
        offs[it] = (it < WN_STAGE_PER_THREAD - 1 || e < WN_STAGE_ELEMS) ? (int)(((py * p.W + pxx) * p.in_sp + c_quad) * ES)
                                                                        : (int)WN_OOB;
      }
    } else {
      s_org = 0;
#pragma unroll
      for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) {
        const int e = tid + it * 256;
        const int px = e >> 2;
        const int py = px / WN_PW, pxx = px - py * WN_PW;
        const int iy = r.oy0 - 1 + py, ix = r.ox0 - 1 + pxx;
        const bool ok = (e < WN_STAGE_ELEMS) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        offs[it] = ok ? (int)(((iy * p.W + ix) * p.in_sp + c_quad) * ES) : (int)WN_OOB;
      }
    }
  };
  const bool c_tail = (p.Cin & 15) != 0;
  auto stage_load = [&](int c0, float4 (&stg)[WN_STAGE_PER_THREAD]) {
    const unsigned so = s_org + (unsigned)c0 * ES;
    if (c_tail && c0 + c_quad >= p.Cin) {
#pragma unroll
      for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) stg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int it = 0; it < WN_STAGE_PER_THREAD; ++it)
        stg[it] = (WS_ABL & 1) ? make_float4(1.f, 2.f, 3.f, 4.f) : wn_buf_load(rs_in, (unsigned)offs[it], so);
    }
  };
  static_assert(WN_STAGE_PER_THREAD == 3 && 2 * 256 < WN_STAGE_ELEMS, "only the third slot of a thread can be a spare");
  int st_lds0 = (tid >> 2) * WN_ROW + 4 * (tid & 3);
  int st_lds2 = tid + 512 < WN_STAGE_ELEMS ? st_lds0 + 128 * WN_ROW : (tid + 512 - WN_STAGE_ELEMS) * WN_ROW + 16;
  auto stage_store = [&](const float4 (&stg)[WN_STAGE_PER_THREAD], float* raw) {
    *reinterpret_cast<float4*>(&raw[st_lds0]) = stg[0];
    *reinterpret_cast<float4*>(&raw[st_lds0 + 64 * WN_ROW]) = stg[1];
    *reinterpret_cast<float4*>(&raw[st_lds2]) = stg[2];
  };
  // weight fragments: step s = (uc = s >> 1, piece = s & 1) -> record (xi = 4 wave + uc, 2 ch + piece)
  unsigned wu_lane = 0;
  auto load_b = [&](int ch, int s, ws_u4 (&dst)[NT]) {
    const int xi = 4 * wave + (s >> 1), g = s & 1;
    const char* wrec = reinterpret_cast<const char*>(reinterpret_cast<const float4*>(p.wu) + (int64_t)(xi * p.G + 2 * ch + g) * rec);
#pragma unroll
    for (int n = 0; n < NT; ++n)
      dst[n] = (WS_ABL & 8) ? ws_u4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}
                            : *reinterpret_cast<const ws_u4*>(wrec + (wu_lane + 512u * n));
  };
  auto rv_fma = [&](const float4 da, const float4 db) {
    const wn_f2 sg = {t_sign, t_sign};
    const wn_f2 lo = __builtin_elementwise_fma(sg, wn_f2{db.x, db.y}, wn_f2{da.x, da.y});
    const wn_f2 hi = __builtin_elementwise_fma(sg, wn_f2{db.z, db.w}, wn_f2{da.z, da.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  auto rv_col = [&](const float* raw, int g, int c) {
    const float4 da = *reinterpret_cast<const float4*>(&raw[rv_a + 8 * g + c * WN_ROW]);
    const float4 db = *reinterpret_cast<const float4*>(&raw[rv_b + 8 * g + c * WN_ROW]);
    return rv_fma(da, db);
  };
  auto rv_row = [&](const float4 (&w)[4], float4 (&a)[4]) {
    a[0] = f4sub(w[0], w[2]); a[1] = f4add(w[1], w[2]); a[2] = f4sub(w[2], w[1]); a[3] = f4sub(w[1], w[3]);
  };

  float4 stg[WN_STAGE_PER_THREAD];
  const bool chain = !(chunks & 1);
  bool staged = false;
  ws_u4 b_f[4][NT];
  float4 av[2][4];
  float4 wq[4];
  Region reg = decode(blockIdx.x < (unsigned)p.total ? (int)blockIdx.x : 0), nxt = reg;
  for (int work = blockIdx.x; work < p.total; work += gridDim.x) {
    const int b = reg.b, oy0 = reg.oy0, ox0 = reg.ox0, co0 = reg.co0, sl0 = reg.ks * chunks;
    wu_lane = (unsigned)(kk * p.Co_pad + co0 + i) * 16u;
    const bool has_next = chain && (work + (int)gridDim.x < p.total);
    if (!staged) {
      WS_PIN("+v"(st_lds0), "+v"(st_lds2));
      aim(reg);
      stage_load(sl0 * 16, stg);
      stage_store(stg, rawB);
      __syncthreads();
    }
    staged = has_next;

    f32x16 acc[4][NT];
    auto slab = [&](auto first_tag, const int ch) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const bool more = ch + 1 < chunks;
      if (more) stage_load((sl0 + ch + 1) * 16, stg);
      else if (has_next) {
        nxt = decode(work + gridDim.x);
        aim(nxt);
        stage_load(nxt.ks * chunks * 16, stg);
      } else {
#pragma unroll
        for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) stg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float* raw = (ch & 1) ? rawA : rawB;
      float* raw_next = (ch & 1) ? rawB : rawA;
      WS_PIN("+v"(rv_a), "+v"(rv_b), "+v"(st_lds0), "+v"(st_lds2), "+v"(wu_lane));
      // the pieces of frequencies 0 and 1 fly while the slab is transformed
#pragma unroll
      for (int s = 0; s < 4; ++s) load_b(sl0 + ch, s, b_f[s]);
      if (!(WS_ABL & 16) || FIRST) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
          for (int c = 0; c < 4; ++c) wq[c] = rv_col(raw, g, c);
          rv_row(wq, av[g]);
        }
      }
#pragma unroll
      for (int uc = 0; uc < 4; ++uc) {
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        S::split(av[0][uc].x, av[0][uc].y, h0, l0);
        S::split(av[0][uc].z, av[0][uc].w, h1, l1);
        S::split(av[1][uc].x, av[1][uc].y, h2, l2);
        S::split(av[1][uc].z, av[1][uc].w, h3, l3);
        const ws_u4 ah = {h0, h1, h2, h3}, al = {l0, l1, l2, l3};
        const int sh = (2 * uc) & 3, sl = (2 * uc + 1) & 3;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if (FIRST) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[uc][n] = S::mfma(ah, b_f[sh][n], zero);
          } else if (!(WS_ABL & 4)) {
            acc[uc][n] = S::mfma(ah, b_f[sh][n], acc[uc][n]);
          }
        }
        if (!(WS_ABL & 4)) {
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[uc][n] = S::mfma(ah, b_f[sl][n], acc[uc][n]);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[uc][n] = S::mfma(al, b_f[sh][n], acc[uc][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (uc < 2) {   // this frequency's register sets are free once its MFMAs are issued: frequency uc + 2 takes them
          load_b(sl0 + ch, 2 * uc + 4, b_f[sh]);
          load_b(sl0 + ch, 2 * uc + 5, b_f[sl]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more || has_next) stage_store(stg, raw_next);
      __syncthreads();
    };
    slab(std::true_type{}, 0);
    for (int ch = 1; ch < chunks; ++ch) slab(std::false_type{}, ch);

    // ---- epilogue: Y = A^T M A, + bias + residual, LeakyReLU, store (the vector epilogue of sr_wino_kernel) ----
    const bool partial = p.ksplit > 1;
    const float* __restrict__ resp = (p.res && !partial)
        ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.res) + (int64_t)b * p.res_sb * ES) : nullptr;
    float* __restrict__ outp = partial ? p.part + reg.ks * p.part_stride + (int64_t)b * p.H * p.W * p.Cout
                                       : reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + (int64_t)b * p.out_sb * ES);
    const unsigned out_sp = partial ? (unsigned)p.Cout : (unsigned)p.out_sp;
    const float* bias_p = partial ? nullptr : p.bias;
    const float slope = sr_uniform(partial ? -1.0f : p.slope);
    {
      constexpr int CO = 32 * NT;
      constexpr int CG = CO / 4;
      constexpr int UNITS = 32 * CG / 256;
      constexpr int TILE_ROWS_PER_UNIT = (256 / CG) / 8;
      const int cg = tid % CG;
      const int tile0 = tid / CG;
      const int tr0 = tile0 >> 3, tc0 = tile0 & 7;
      const __amdgpu_buffer_rsrc_t rs_out = wn_rsrc(outp, ((int64_t)(p.H * p.W - 1) * out_sp + p.Cout) * ES);
      const __amdgpu_buffer_rsrc_t rs_res =
          wn_rsrc(resp ? (const void*)resp : (const void*)p.wu,
                  resp ? ((int64_t)(p.H * p.W - 1) * p.res_sp + p.Cout) * ES : (int64_t)0);
      const __amdgpu_buffer_rsrc_t rs_bias = wn_rsrc(bias_p ? (const void*)bias_p : (const void*)p.wu,
                                                     bias_p ? (int64_t)p.Cout * 4 : (int64_t)0);
      const bool okc = co0 + 4 * cg < p.Cout;
      const bool full = (oy0 + 2 * WN_TR <= p.H) & (ox0 + 2 * WN_TC <= p.W) & (co0 + CO <= p.Cout);
      const unsigned pix0 = (unsigned)((2 * tr0) * p.W + 2 * tc0);
      const unsigned v_out = (pix0 * out_sp + 4u * cg) * ES, v_res = (pix0 * (unsigned)p.res_sp + 4u * cg) * ES;
      const unsigned s_out0 = ((unsigned)(oy0 * p.W + ox0) * out_sp + (unsigned)co0) * ES;
      const unsigned s_res0 = ((unsigned)(oy0 * p.W + ox0) * (unsigned)p.res_sp + (unsigned)co0) * ES;
      auto d_pix = [&](int it, int q) {
        return (unsigned)((2 * TILE_ROWS_PER_UNIT * it + (q >> 1)) * p.W + (q & 1));
      };
      int o_wr = (wave * 2 * 32 + 4 * kk) * CO + i, o_rd = tile0 * CO + 4 * cg;
      asm volatile("" : "+v"(o_wr), "+v"(o_rd));
      float neg1s = -1.0f;
      asm volatile("" : "+s"(neg1s));
      const wn_f2 neg1 = {neg1s, neg1s};
      unsigned rsp4 = (unsigned)p.res_sp * ES, osp4 = out_sp * ES;
      WS_PIN("+s"(rsp4), "+s"(osp4));
      const bool fast_leaky = slope >= 0.0f && slope <= 1.0f;
      const wn_f2 slope2 = {slope, slope};
      auto epilogue = [&](auto full_tag, auto res_tag) {
        constexpr bool FULL = decltype(full_tag)::value, RES = decltype(res_tag)::value;
        unsigned okm = 0;
        if (!FULL) {
#pragma unroll
          for (int it = 0; it < UNITS; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int oy = oy0 + 2 * (tr0 + TILE_ROWS_PER_UNIT * it) + (q >> 1), ox = ox0 + 2 * tc0 + (q & 1);
              okm |= (unsigned)(okc & (oy < p.H) & (ox < p.W)) << (4 * it + q);
            }
        }
        auto lane_off = [&](unsigned v, int it, int q) { return (FULL || ((okm >> (4 * it + q)) & 1u)) ? v : WN_OOB; };
        float4 rv[UNITS][4];
        if (RES) {
#pragma unroll
          for (int it = 0; it < UNITS; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              rv[it][q] = (WS_ABL & 2) ? make_float4(0.f, 0.f, 0.f, 0.f)
                                       : wn_buf_load(rs_res, lane_off(v_res, it, q), s_res0 + d_pix(it, q) * rsp4);
        }
        const float4 bv = wn_buf_load(rs_bias, (FULL || okc) ? 16u * cg : WN_OOB, (unsigned)co0 * 4u);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const wn_f2 m0 = {acc[0][n][r], acc[0][n][r + 1]}, m1 = {acc[1][n][r], acc[1][n][r + 1]};
            const wn_f2 m2 = {acc[2][n][r], acc[2][n][r + 1]}, m3 = {acc[3][n][r], acc[3][n][r + 1]};
            const wn_f2 c0 = (m0 + m1) + m2;
            const wn_f2 c1 = __builtin_elementwise_fma(neg1, m3, __builtin_elementwise_fma(neg1, m2, m1));
            float* o0 = &O[o_wr + ((r & 3) + 8 * (r >> 2)) * CO + 32 * n];
            float* o1 = o0 + 32 * CO;
            o0[0] = c0.x; o0[CO] = c0.y;
            o1[0] = c1.x; o1[CO] = c1.y;
          }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < UNITS; ++it) {
          float4 t[4][2];
#pragma unroll
          for (int ur = 0; ur < 4; ++ur)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
              t[ur][bb] = *reinterpret_cast<const float4*>(&O[o_rd + ((ur * 2 + bb) * 32 + (256 / CG) * it) * CO]);
          const float4 y[4] = {f4add(f4add(t[0][0], t[1][0]), t[2][0]), f4add(f4add(t[0][1], t[1][1]), t[2][1]),
                               f4sub(f4sub(t[1][0], t[2][0]), t[3][0]), f4sub(f4sub(t[1][1], t[2][1]), t[3][1])};
          float o16[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v;
            if (S::USCALE != 1.0f) {   // undo the weight scaling where the bias is added: a packed fma instead of a packed add
              const wn_f2 us = {1.0f / S::USCALE, 1.0f / S::USCALE};
              const wn_f2 lo = __builtin_elementwise_fma(wn_f2{y[q].x, y[q].y}, us, wn_f2{bv.x, bv.y});
              const wn_f2 hi = __builtin_elementwise_fma(wn_f2{y[q].z, y[q].w}, us, wn_f2{bv.z, bv.w});
              v = make_float4(lo.x, lo.y, hi.x, hi.y);
            } else {
              v = f4add(y[q], bv);
            }
            if (RES) v = f4add(v, rv[it][q]);
            if (fast_leaky) {
              const wn_f2 lo = wn_f2{v.x, v.y} * slope2, hi = wn_f2{v.z, v.w} * slope2;
              v = make_float4(sr_vmax(v.x, lo.x), sr_vmax(v.y, lo.y), sr_vmax(v.z, hi.x), sr_vmax(v.w, hi.y));
            }
            o16[4 * q + 0] = v.x; o16[4 * q + 1] = v.y; o16[4 * q + 2] = v.z; o16[4 * q + 3] = v.w;
          }
          if (!fast_leaky) sr_activate_group(o16, slope);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (!(WS_ABL & 2) || o16[4 * q] == 1.2345e33f)
              wn_buf_store(make_float4(o16[4 * q], o16[4 * q + 1], o16[4 * q + 2], o16[4 * q + 3]), rs_out,
                           lane_off(v_out, it, q), s_out0 + d_pix(it, q) * osp4);
        }
      };
      if (full) { if (resp) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::true_type{}, std::false_type{}); }
      else { if (resp) epilogue(std::false_type{}, std::true_type{}); else epilogue(std::false_type{}, std::false_type{}); }
      __syncthreads();
    }
    reg = has_next ? nxt : decode(work + (int)gridDim.x < p.total ? work + (int)gridDim.x : work);
  }
}

// ------------------------------------------------------------------ host side (called from sr_wino.hip) -------------

int sr_wino_split_mode() { return sr_opt(SR_OPT_WINO_SPLIT); }

int sr_wino_split_pack(const float* weight, int Cout, int Cin, float* packed, int mode, hipStream_t stream) {
  const int G = ((Cin + 15) / 16) * 2, Co_pad = ((Cout + 31) / 32) * 32;
  unsigned* wu = reinterpret_cast<unsigned*>(packed);
  if (mode == 1) hipLaunchKernelGGL((sr_wino_pack_split_kernel<1>), dim3(256), dim3(256), 0, stream, weight, wu, Cout, Cin, G, Co_pad);
  else if (mode == 2) hipLaunchKernelGGL((sr_wino_pack_split_kernel<2>), dim3(256), dim3(256), 0, stream, weight, wu, Cout, Cin, G, Co_pad);
  else return SR_ERR_INVALID_ARGUMENT;
  return sr_hip_rc(hipGetLastError());
}

int sr_wino_split_launch(const SrWinoParams& p, int nt, int blocks, int mode, hipStream_t stream) {
  const size_t lds = (size_t)WN_LDS_FLOATS(nt) * sizeof(float);
  if (nt == 1 && SR_WS_NT1_WGS != 2) {   // (tuning builds) the persistent grid follows the occupancy the kernel was compiled for
    blocks = blocks / 2 * SR_WS_NT1_WGS;
    if (blocks > p.total) blocks = p.total;
  }
#define WS_LAUNCH(NTV, FMTV)                                                                                      \
  {                                                                                                               \
    hipError_t e = hipFuncSetAttribute((const void*)sr_wino_split_kernel<NTV, FMTV>,                              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
    if (e != hipSuccess) return sr_hip_rc(e);                                                                     \
    hipLaunchKernelGGL((sr_wino_split_kernel<NTV, FMTV>), dim3(blocks), dim3(256), lds, stream, p);               \
  }
  if (mode == 1 && nt == 2) WS_LAUNCH(2, 1)
  else if (mode == 1) WS_LAUNCH(1, 1)
  else if (mode == 2 && nt == 2) WS_LAUNCH(2, 2)
  else if (mode == 2) WS_LAUNCH(1, 2)
  else return SR_ERR_INVALID_ARGUMENT;
#undef WS_LAUNCH
  return sr_hip_rc(hipGetLastError());
}
