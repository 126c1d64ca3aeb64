// sr_wino.hip -- 3x3 / stride-1 convolutions through Winograd F(2x2, 3x3) on the fp32 matrix cores (gfx950).
//
// Same operator as sr_conv2d_nhwc_fwd (conv + bias + residual + LeakyReLU of the reference's BasicBlock,
// modules/layers.py:24-85), same fp32 arithmetic class: Y = A^T [ (G g G^T) . (B^T d B) ] A  (Lavin & Gray) needs
// 16 multiplies per 2x2 output tile and (input, output) channel pair instead of 36, i.e. 2.25x fewer MFMA FLOPs;
// the transforms use only +, - and exact scalings by 1/2, and the fp32 error vs an fp64 reference is the
// same as the direct algorithm's (3.1e-7 vs 2.7e-7 on a 64-channel layer, tests/).
//
// A workgroup (4 waves, 2 workgroups per CU) owns a region of 4 x 8 Winograd tiles (= 8 x 16 output pixels = the 32
// rows of one MFMA M-tile) x 32*NT output channels.  Per 16-channel slab of the input:
//   S  the 10 x 18 pixel input patch is staged global -> registers -> LDS, double-buffered (raw A / raw B); with an even
//      slab count the NEXT region's first slab follows the same way (it lands behind the O area, which the epilogue
//      leaves alone),
//   T  wave w applies B^T d B for frequency row ur = w -- exactly the rows it multiplies.  Lane (i, kk) transforms tile
//      i for the channel quads 2g + kk, g = 0, 1: precisely the A operands of its own MFMAs, so V lives in REGISTERS
//      (r03; it used to make a ds_write / ds_read round trip through a 40 KB V array).  The transform of group g = 1 is
//      issued under the MFMAs of group 0, and (SR_WINO_PIPE) the transform of the NEXT slab's group 0 under the MFMAs of
//      group 1: the next slab is stored to LDS at step 3, the slab barrier sits after step 5,
//   M  wave w multiplies the 4 "frequencies" xi = 4w..4w+3:  M_xi[tile, co] += V_xi[tile, ci] . U_xi[ci, co]
//      (U streams from L2 in B-fragment order, prefetched 3 steps ahead through 4 rotating register sets).
// Epilogue: the output transform is separable -- wave w holds a whole frequency row, so the column half (M A) happens
// in registers and 2 of 4 values per (tile, channel) cross LDS (one 64-KB pass); a thread then owns (tile, 4 channels)
// units: float4 residual loads, bias, LeakyReLU, float4 stores straight into the consumer's concat slice.
// The live set (128 accumulator + 32 weight + 32 operand + 12 staging registers ...) fits 256 VGPRs without scratch spills.
#include <type_traits>

#include "sr_wino.h"

// U = G g G^T per (co, ci), stored in MFMA B-fragment order: element (xi, g8, kk, co, e) = U_xi[co][8*g8 + 4*kk + e]
__global__ void sr_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ wu, int Co, int Ci, int G,
                                    int Co_pad) {
  const int64_t total = (int64_t)16 * G * 2 * Co_pad * 4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int el = (int)(e & 3);
    int64_t r = e >> 2;
    const int co = (int)(r % Co_pad); r /= Co_pad;
    const int kk = (int)(r & 1); r >>= 1;
    const int g8 = (int)(r % G);
    const int xi = (int)(r / G);
    const int ci = 8 * g8 + 4 * kk + el;
    float v = 0.0f;
    if (co < Co && ci < Ci) {
      const float* g = w + ((int64_t)co * Ci + ci) * 9;
      const int ur = xi >> 2, uc = xi & 3;
      // row transform (G g): rows [g0; (g0+g1+g2)/2; (g0-g1+g2)/2; g2], then the same along columns
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
    wu[e] = v;
  }
}

// ---- 16-bit activation I/O (r04; training under torch.autocast, reference options.py:100-101 / train.py:132) ----
// IO = 0: fp32 tensors (inference, the measured path).  IO = 1 / 2: the input, the residual and the output are fp16 / bf16
// tensors in HBM -- four channels are ONE 8-byte load or store --, widened to fp32 on the way into LDS and rounded (to
// nearest even) on the way out; weights, bias, the transforms and the MFMA accumulation stay fp32.  Offsets are in bytes,
// so only the element size changes: WN_ES(IO).
#define WN_ES(io) ((io) == 0 ? 4u : 2u)
typedef unsigned int wn_u2 __attribute__((ext_vector_type(2)));
template <int IO>
__device__ __forceinline__ float wn_widen(unsigned short h) {
  if (IO == 1) return (float)__builtin_bit_cast(_Float16, h);
  return __builtin_bit_cast(float, (unsigned)h << 16);
}
template <int IO>
__device__ __forceinline__ unsigned short wn_narrow(float f) {
  if (IO == 1) return __builtin_bit_cast(unsigned short, (_Float16)f);
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
template <int IO>
__device__ __forceinline__ float4 wn_io_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if (IO == 0) return wn_buf_load(r, voff, soff);
  const wn_u2 v = __builtin_bit_cast(wn_u2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
  return make_float4(wn_widen<IO>((unsigned short)(v.x & 0xffffu)), wn_widen<IO>((unsigned short)(v.x >> 16)),
                     wn_widen<IO>((unsigned short)(v.y & 0xffffu)), wn_widen<IO>((unsigned short)(v.y >> 16)));
}
template <int IO>
__device__ __forceinline__ void wn_io_store(const float4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if (IO == 0) { wn_buf_store(v, r, voff, soff); return; }
  wn_u2 t;
  t.x = (unsigned)wn_narrow<IO>(v.x) | ((unsigned)wn_narrow<IO>(v.y) << 16);
  t.y = (unsigned)wn_narrow<IO>(v.z) | ((unsigned)wn_narrow<IO>(v.w) << 16);
  __builtin_amdgcn_raw_buffer_store_b64(t, r, (int)(voff + soff), 0, 0);
}

template <int NT, bool VEC4, bool VOUT, int IO = 0>
__global__ __launch_bounds__(256, NT == 1 ? SR_WINO_NT1_WAVES : SR_WINO_WAVES) void sr_wino_kernel(SrWinoParams p) {
  static_assert(IO == 0 || (VEC4 && VOUT), "16-bit I/O exists for the vector staging / epilogue instantiation only");
  constexpr unsigned ES = WN_ES(IO);   // bytes per activation element
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* O = lds;                           // [4 ur][2][32 tiles][32*NT co]   (epilogue only; aliases raw A)
  float* rawA = lds + WN_V_FLOATS(NT);      // [10*18][20]  odd slabs  (inside the O area: dead by the epilogue)
  float* rawB = lds + WN_O_FLOATS(NT);      // [10*18][20]  even slabs (behind it: survives the epilogue)
  const int tid = threadIdx.x;
  // (readfirstlane: the wave index is uniform, so everything derived from it -- the weight-record offsets of the MFMA
  // loop above all -- is scalar arithmetic; as a plain `tid >> 6` it cost 7 VALU instructions per MFMA step)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kk = lane >> 5;
  const int chunks = (p.G >> 1) / p.ksplit;  // input slabs per work item
  const int64_t rec = (int64_t)2 * p.Co_pad;
  constexpr int STEPS = 8;            // (8-channel group, frequency) steps per slab and wave
  static_assert(8 % SR_WINO_NB == 0 && SR_WINO_PD < SR_WINO_NB, "the register rotation must line up across slabs");
  constexpr int NB = SR_WINO_NB, PD = SR_WINO_PD;  // weight prefetch: PD steps ahead through NB rotating register sets

  // Transform role: wave w produces the frequency ROW ur = w of V = B^T d B (the rows it alone multiplies).  Row ur of
  // B^T d needs two patch rows: (0,2) d0-d2, (1,2) d1+d2, (2,1) d2-d1, (1,3) d1-d3.  Lane (i, kk) does it for tile i (the
  // MFMA row it feeds) and the channel quads 2g + kk -- the A operands of its own MFMAs.
  const int t_ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1), t_rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float t_sign = wave == 1 ? 1.0f : -1.0f;
  const int rv_base = ((2 * (i >> 3)) * WN_PW + 2 * (i & 7)) * WN_ROW + 4 * kk;   // patch offset of tile i, quad kk
  // the two patch rows this wave's transform reads, as ONE register each: every other term of an rv_ld address is an
  // immediate (left to itself the compiler hoists the 16 distinct addresses out of the work loop and spills them)
  int rv_a = rv_base + t_ra * WN_PW * WN_ROW, rv_b = rv_base + t_rb * WN_PW * WN_ROW;
  // (`SR_WN_PIN`: an empty asm that makes the value opaque at this point, so that addresses derived from it are formed
  // HERE as register + immediate instead of being hoisted out of the loops, one register -- then one spill -- each)
#define SR_WN_PIN(...) asm volatile("" : __VA_ARGS__)

#ifdef SR_WINO_TRACE
  int tr_region = -1;
  if (tid == 0) {  // event 15 of region 0: HW_ID (CU / SE) and XCC_ID, to group co-resident workgroups
    p.trace[((size_t)blockIdx.x * SR_TR_REGIONS) * SR_TR_EVENTS + 15] =
        ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) |
        (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  }
#endif
  // Region coordinates of a work item and the per-thread staging offsets of its 10x18 input patch.
  struct Region { int b, oy0, ox0, co0, ks; };
  // XCD-aware work order.  Workgroup b runs on XCD b % 8 (round-robin dispatch) and each XCD has its own L2.  In round r
  // the grid works on items [r G, (r + 1) G): give XCD x the CONTIGUOUS eighth [r G + x G/8, r G + (x + 1) G/8) of them
  // (consecutive items are horizontally adjacent regions, G/8 of them about three region rows), so that the halo
  // pixels two neighbouring regions share are fetched into ONE L2 instead of two (r02 PMC: 1.40x the algorithmic
  // bytes).  The last, partial round keeps the plain order.
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
  // ---- staging: global -> registers -> LDS ----
  // Vector path (VEC4): lane offsets in BYTES relative to the image base, through the buffer descriptor `rs_in`.  For an
  // interior region (the whole 10x18 patch inside the image) they are the per-thread constants `rel` plus the scalar
  // patch origin `s_org`; a border region computes them with the image test (outside -> WN_OOB -> 0).  Scalar path:
  // element offsets from the image pointer, -1 = outside (r03 code, unaligned inputs only).
  int offs[WN_STAGE_PER_THREAD];
  unsigned s_org = 0;
  const float* in_b = p.in;
  __amdgpu_buffer_rsrc_t rs_in = wn_rsrc(p.in, 0);
  const int64_t in_img_bytes = ((int64_t)(p.H * p.W - 1) * p.in_sp + p.Cin) * ES;
  const int c_quad = 4 * (tid & 3);      // (tid + 256 it) & 3 == tid & 3: a thread stages the same channel quad of a slab
  auto aim = [&](const Region& r) {  // point the staging loads at region r
    in_b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in) + (int64_t)r.b * p.in_sb * ES);
    if (VEC4) {
      rs_in = wn_rsrc(in_b, in_img_bytes);
      const bool interior = (r.oy0 >= 1) & (r.oy0 + 2 * WN_TR + 1 <= p.H) & (r.ox0 >= 1) & (r.ox0 + 2 * WN_TC + 1 <= p.W);
      if (interior) {   // no image test: offsets relative to the patch origin, which rides in the scalar operand
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
    } else {
#pragma unroll
      for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) {
        const int e = tid + it * 256;
        const int px = e >> 2, q = e & 3;
        const int py = px / WN_PW, pxx = px - py * WN_PW;
        const int iy = r.oy0 - 1 + py, ix = r.ox0 - 1 + pxx;
        const bool ok = (e < WN_STAGE_ELEMS) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        offs[it] = ok ? (iy * p.W + ix) * p.in_sp + 4 * q : -1;
      }
    }
  };
  const bool c_tail = (p.Cin & 15) != 0;   // the last slab is cut by Cin: channel quads at or beyond it read 0
  auto stage_load = [&](int c0, float4 (&stg)[WN_STAGE_PER_THREAD]) {
    if (VEC4) {
      const unsigned so = s_org + (unsigned)c0 * ES;
      if (c_tail && c0 + c_quad >= p.Cin) {   // (lane-divergent only in the last slab of a ragged channel count)
#pragma unroll
        for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) stg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int it = 0; it < WN_STAGE_PER_THREAD; ++it)
          stg[it] = SR_WN_DBG(4) ? make_float4(0.f, 0.f, 0.f, 0.f) : wn_io_load<IO>(rs_in, (unsigned)offs[it], so);
      }
    } else {
#pragma unroll
      for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) {
        const int c = c0 + 4 * ((tid + it * 256) & 3);
        const bool ok = (offs[it] >= 0) & (c < p.Cin) & !SR_WN_DBG(4);
        const float* src = in_b + (ok ? offs[it] + c0 : 0);
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
  };
  // LDS side of the staging: element e -> pixel e >> 2, quad e & 3.  The 48 thread slots beyond the 720 elements of a
  // patch write their (zero) registers into the pad quad of a pixel row instead of being branched around.
  static_assert(WN_STAGE_PER_THREAD == 3 && 2 * 256 < WN_STAGE_ELEMS, "only the third slot of a thread can be a spare");
  int st_lds0 = (tid >> 2) * WN_ROW + 4 * (tid & 3);   // slots 0 / 1: + it * 64 pixels
  int st_lds2 = tid + 512 < WN_STAGE_ELEMS ? st_lds0 + 128 * WN_ROW : (tid + 512 - WN_STAGE_ELEMS) * WN_ROW + 16;
  auto stage_store = [&](const float4 (&stg)[WN_STAGE_PER_THREAD], float* raw) {
    *reinterpret_cast<float4*>(&raw[st_lds0]) = stg[0];
    *reinterpret_cast<float4*>(&raw[st_lds0 + 64 * WN_ROW]) = stg[1];
#ifdef SR_WINO_DBG_NOPAD
    if (tid + 512 < WN_STAGE_ELEMS)
#endif
    *reinterpret_cast<float4*>(&raw[st_lds2]) = stg[2];
  };
  (void)rv_base;
  // weight fragments: scalar record base (SGPR pair) + a 32-bit lane index -- the saddr form of global_load, no 64-bit
  // vector address arithmetic in the MFMA stream
  unsigned wu_lane = 0;   // BYTE offset of this lane's fragment inside a record (a 32-bit offset: what the saddr form takes)
  auto load_b = [&](int ch, int s, float4 (&dst)[NT]) {   // step s = (g, uc): all four frequencies of group 0, then group 1
    const int xi = 4 * wave + (s & 3), g = s >> 2;
    const char* wrec = reinterpret_cast<const char*>(reinterpret_cast<const float4*>(p.wu) +
                                                     (SR_WN_DBG(32) ? (int64_t)0 : (int64_t)(xi * p.G + 2 * ch + g) * rec));
#pragma unroll
    for (int n = 0; n < NT; ++n) dst[n] = *reinterpret_cast<const float4*>(wrec + (wu_lane + 512u * n));
  };
  // rv_col(raw, g, c): row `ur = wave` of B^T d at patch column c, channel quad 2g + kk; rv_row: the four frequencies
  // uc = 0..3 of that row from its four columns = the A operands of steps (g, uc).
  auto rv_fma = [&](const float4 da, const float4 db) {
    const wn_f2 sg = {t_sign, t_sign};
    const wn_f2 lo = __builtin_elementwise_fma(sg, wn_f2{db.x, db.y}, wn_f2{da.x, da.y});
    const wn_f2 hi = __builtin_elementwise_fma(sg, wn_f2{db.z, db.w}, wn_f2{da.z, da.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  auto rv_ld = [&](const float* raw, int g, int c, float4& da, float4& db) {
    da = *reinterpret_cast<const float4*>(&raw[rv_a + 8 * g + c * WN_ROW]);
    db = *reinterpret_cast<const float4*>(&raw[rv_b + 8 * g + c * WN_ROW]);
  };
  auto rv_col = [&](const float* raw, int g, int c) {
    float4 da, db;
    rv_ld(raw, g, c, da, db);
    return rv_fma(da, db);
  };
  auto rv_row = [&](const float4 (&w)[4], float4 (&a)[4]) {
    a[0] = f4sub(w[0], w[2]); a[1] = f4add(w[1], w[2]); a[2] = f4sub(w[2], w[1]); a[3] = f4sub(w[1], w[3]);
  };

  // Software pipeline: slab c of a region sits in raw buffer (c odd ? A : B); slab c+1 is fetched into registers at
  // the top of slab c.  SR_WINO_PIPE: it is stored to the other buffer after step 3 (the last read of this wave's own
  // transform of slab c is in step 3, the other buffer's last readers finished a slab ago), the workgroup barrier sits
  // after step 5, and steps 6 / 7 carry the transform of slab c+1's first channel group -- so a slab boundary costs no
  // serial LDS latency chain.  Otherwise: stored at the end of slab c, barrier, transform at the top of slab c+1.
  // With an even slab count the NEXT region's slab 0 follows the same way during the last slab -- it lands in B,
  // which the epilogue leaves alone, and is issued in front of the epilogue's stores (VMEM returns in order).  r04: the
  // rest of the next region's prologue -- its first weight fragments and the transform of its first channel group --
  // is then issued in the SECOND half of the epilogue (behind the O exchange, where the accumulators are dead), under
  // the output stores, instead of in front of the next region's first MFMA.
  float4 stg[WN_STAGE_PER_THREAD];
  const bool chain = !(chunks & 1);
  bool staged = false;      // the next region's slab 0 sits in raw B
  bool primed_w = false, primed_t = false;   // ... and its first weight fragments / first A operands are already in registers
  float4 b_f[NB][NT];
  float4 av[2][4];   // A operands of the current slab: [g][uc]
  float4 wq[4];
  Region reg = decode(blockIdx.x < (unsigned)p.total ? (int)blockIdx.x : 0), nxt = reg;
  // SR_WINO_STAGGER (experiment, default 0): two persistent workgroups share a CU and run items of equal length; if they start
  // together they stay in phase -- both in their MFMA streams, then both in their epilogues.  The switch delays the second half
  // of the grid once, at launch (host side: what it does and does not buy).
  if (p.stagger > 0 && blockIdx.x >= (gridDim.x >> 1))
    for (int s = 0; s < p.stagger; ++s) __builtin_amdgcn_s_sleep(127);
  for (int work = blockIdx.x; work < p.total; work += gridDim.x) {
    const int b = reg.b, oy0 = reg.oy0, ox0 = reg.ox0, co0 = reg.co0, sl0 = reg.ks * chunks;
    wu_lane = (unsigned)(kk * p.Co_pad + co0 + i) * 16u;
    const bool has_next = chain && (work + (int)gridDim.x < p.total);
#ifdef SR_WINO_TRACE
    ++tr_region;
#endif
    SR_TR(0);
    if (!staged) {
      SR_WN_PIN("+v"(st_lds0), "+v"(st_lds2));
      aim(reg);
      stage_load(sl0 * 16, stg);
      stage_store(stg, rawB);
      __syncthreads();
    }
    staged = has_next;

    // (not zeroed: the first MFMA into each accumulator takes the constant 0 as its C operand -- zeroing 128 registers
    // cost 128 VALU instructions per region, each ~13 clocks while the co-resident workgroup keeps the matrix pipe busy)
    f32x16 acc[4][NT];

    if (!primed_w) {
#pragma unroll
      for (int s = 0; s < PD; ++s) load_b(sl0, s, b_f[s]);
    }
    SR_TR(1);

#if SR_WINO_PIPE
    float4 pa[2], pb[2];   // patch rows on their way from LDS to the next slab's transform
    if (!primed_t) {  // first slab of the region: its group 0 has no MFMAs to hide under
      SR_WN_PIN("+v"(rv_a), "+v"(rv_b));
#pragma unroll
      for (int c = 0; c < 4; ++c) wq[c] = rv_col(rawB, 0, c);
      rv_row(wq, av[0]);
    }
#endif
    primed_w = primed_t = false;
    auto slab = [&](auto first_tag, const int ch) {
      constexpr bool FIRST = decltype(first_tag)::value;   // first slab of the region: accumulators start from 0
      const bool more = ch + 1 < chunks;
      if (more) stage_load((sl0 + ch + 1) * 16, stg);
      else if (has_next) {
        nxt = decode(work + gridDim.x);
        aim(nxt);
        stage_load(nxt.ks * chunks * 16, stg);
      } else {   // nothing follows: the unconditional store below then writes zeros (keeps `stg` from living across regions)
#pragma unroll
        for (int it = 0; it < WN_STAGE_PER_THREAD; ++it) stg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float* raw = (ch & 1) ? rawA : rawB;
      float* raw_next = (ch & 1) ? rawB : rawA;
      SR_WN_PIN("+v"(rv_a), "+v"(rv_b), "+v"(st_lds0), "+v"(st_lds2), "+v"(wu_lane));
#if !SR_WINO_PIPE
      if (!SR_WN_DBG(2)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wq[c] = rv_col(raw, 0, c);
        rv_row(wq, av[0]);
      }
#endif
      if (ch < 5) SR_TR(2 + 2 * ch);

      // ---- M: this wave's 4 frequencies x 2 channel groups (same products in the same order as ever: bit-identical) ----
      if (!SR_WN_DBG(8))
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int cbuf = s % NB, g = s >> 2, uc = s & 3;
        const float4 a = av[g][uc];
        // k-step outer, N-tile inner: consecutive MFMAs hit different accumulators.  The prefetches of later steps and
        // the transform pieces sit BETWEEN the MFMA pairs, where the wave has idle issue cycles; sched_barrier pins them.
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if (FIRST && s < 4) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[uc][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b_f[cbuf][n].x, zero, 0, 0, 0);
          } else {
            acc[uc][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b_f[cbuf][n].x, acc[uc][n], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + PD < STEPS) load_b(sl0 + ch, s + PD, b_f[(s + PD) % NB]);
        else if (more) load_b(sl0 + ch + 1, s + PD - STEPS, b_f[(s + PD) % NB]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[uc][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b_f[cbuf][n].y, acc[uc][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // steps 0-3: one patch column of this slab's channel group 1 each (LDS latency << a step's 8 MFMAs)
        if (s < 4) wq[s] = rv_col(raw, 1, s);
#if SR_WINO_PIPE
        // steps 6, 7: the next slab's channel group 0 (its patch was stored at step 3, barrier after step 5); the LDS
        // reads are issued one slot before their use.  Unconditional (no branch in the MFMA stream): after the last slab
        // of a region the values are simply not used.
        if (s == 6) { wq[0] = rv_fma(pa[0], pb[0]); wq[1] = rv_fma(pa[1], pb[1]);
                      rv_ld(raw_next, 0, 2, pa[0], pb[0]); rv_ld(raw_next, 0, 3, pa[1], pb[1]); }
        if (s == 7) { wq[2] = rv_fma(pa[0], pb[0]); wq[3] = rv_fma(pa[1], pb[1]); }
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[uc][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b_f[cbuf][n].z, acc[uc][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[uc][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b_f[cbuf][n].w, acc[uc][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s == 3) {
          rv_row(wq, av[1]);
#if SR_WINO_PIPE
          stage_store(stg, raw_next);   // unconditional: after the last slab of the last region it stores stale registers
#endif                                  // into a buffer nobody reads before it is restaged
          __builtin_amdgcn_sched_barrier(0);
        }
#if SR_WINO_PIPE
        if (s == 5) {   // every wave has stored its share of the next slab and finished reading this one
          __syncthreads();
          rv_ld(raw_next, 0, 0, pa[0], pb[0]); rv_ld(raw_next, 0, 1, pa[1], pb[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
      }
      if (ch < 5) SR_TR(3 + 2 * ch);   // this wave's MFMAs issued
#if SR_WINO_PIPE
      if (more) rv_row(wq, av[0]);     // (after the last slab of a region: the next region's, done under the epilogue)
#else
      if (more || has_next) stage_store(stg, raw_next);
      __syncthreads();
#endif
    };
    slab(std::true_type{}, 0);
    for (int ch = 1; ch < chunks; ++ch) slab(std::false_type{}, ch);
    SR_TR(12);

    // ---- epilogue: Y = A^T M A, + bias + residual, LeakyReLU, store ----
    const bool partial = p.ksplit > 1;
    // (image bases in BYTES of the activation element: with 16-bit I/O p.res / p.out point at fp16 / bf16 data; the
    // split-K partial workspace is always fp32 and 16-bit I/O launches never split K)
    const float* __restrict__ resp = (p.res && !partial)
        ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.res) + (int64_t)b * p.res_sb * ES) : nullptr;
    float* __restrict__ outp = partial ? p.part + reg.ks * p.part_stride + (int64_t)b * p.H * p.W * p.Cout
                                       : reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + (int64_t)b * p.out_sb * ES);
    const unsigned out_sp = partial ? (unsigned)p.Cout : (unsigned)p.out_sp;
    const float* bias_p = partial ? nullptr : p.bias;
    const float slope = sr_uniform(partial ? -1.0f : p.slope);   // scalar: the activation code is tested once per group
    // Y = A^T M A is separable: wave w holds the whole frequency ROW ur = w (its 4 accumulators are the columns
    // uc = 0..3), so the column half (M A) is done in registers and only 2 of 4 values per (tile, channel) go
    // through LDS: O[ur][b][tile][co] (64 KB for both N-tiles -> one pass, two barriers).
    // The next region's prologue pieces that ride in the epilogue's second half (PIPE builds with a staged next region):
    auto prime_next = [&]() {
#if SR_WINO_PIPE && SR_WINO_PRIME
      if (has_next) {
        SR_WN_PIN("+v"(rv_a), "+v"(rv_b));
        if (SR_WINO_PRIME & 1) {
          wu_lane = (unsigned)(kk * p.Co_pad + nxt.co0 + i) * 16u;
          const int nsl0 = nxt.ks * chunks;
#pragma unroll
          for (int s = 0; s < PD; ++s) load_b(nsl0, s, b_f[s]);
          primed_w = true;
        }
        if (SR_WINO_PRIME & 2) {
#pragma unroll
          for (int c = 0; c < 4; ++c) wq[c] = rv_col(rawB, 0, c);
          rv_row(wq, av[0]);
          primed_t = true;
        }
      }
#endif
    };
    if (!SR_WN_DBG(16)) {
      constexpr int CO = 32 * NT;           // channels per workgroup
      if (VOUT) {
        // Vector epilogue (Cout % 4 == 0, 16-byte aligned output / residual rows): a thread owns (tile, 4 consecutive
        // channels) units -- float4 residual loads, ds_read_b128 of the exchanged slab, float4 stores.  All of them
        // through buffer descriptors: ONE lane offset per tensor (the unit's first pixel, channel quad), the other
        // seven (unit, pixel) positions are scalar deltas; border regions / channel tails switch lanes off by offset.
        constexpr int CG = CO / 4;            // channel groups per workgroup
        constexpr int UNITS = 32 * CG / 256;  // = NT
        constexpr int TILE_ROWS_PER_UNIT = (256 / CG) / 8;   // tile rows between a thread's consecutive units
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
#ifdef SR_WINO_FORCE_BORDER   // (test builds: every region through the per-position epilogue)
        const bool full = false;
#else
        const bool full = (oy0 + 2 * WN_TR <= p.H) & (ox0 + 2 * WN_TC <= p.W) & (co0 + CO <= p.Cout);   // uniform
#endif
        const unsigned pix0 = (unsigned)((2 * tr0) * p.W + 2 * tc0);
        const unsigned v_out = (pix0 * out_sp + 4u * cg) * ES, v_res = (pix0 * (unsigned)p.res_sp + 4u * cg) * ES;
        const unsigned s_out0 = ((unsigned)(oy0 * p.W + ox0) * out_sp + (unsigned)co0) * ES;
        const unsigned s_res0 = ((unsigned)(oy0 * p.W + ox0) * (unsigned)p.res_sp + (unsigned)co0) * ES;
        auto d_pix = [&](int it, int q) {   // scalar: pixel delta of (unit, pixel) from the unit-0 / pixel-0 position
          return (unsigned)((2 * TILE_ROWS_PER_UNIT * it + (q >> 1)) * p.W + (q & 1));
        };
        // FULL = the region lies inside the image and the channel block inside Cout (every region of the 240x320 and
        // 120x160 levels): one lane offset per tensor serves all eight (unit, pixel) positions.  Otherwise a per-position
        // offset, WN_OOB outside the image / past Cout.
        // LDS offsets of the O exchange as ONE register each + immediates (laundered per region: hoisted out of the work
        // loop, the compiler kept 16 precomputed addresses in scratch and reloaded them in front of every ds_write)
        int o_wr = (wave * 2 * 32 + 4 * kk) * CO + i, o_rd = tile0 * CO + 4 * cg;
        asm volatile("" : "+v"(o_wr), "+v"(o_rd));
        // a - b as fma(-1, b, a) with a -1 the compiler cannot see: v_pk_fma_f32 (the same single rounding as the
        // subtraction).  Written as a - b, the compiler scalarises the 32 packed subtractions of the column transform
        // into 64 v_sub_f32.
        float neg1s = -1.0f;
        asm volatile("" : "+s"(neg1s));
        const wn_f2 neg1 = {neg1s, neg1s};
        unsigned rsp4 = (unsigned)p.res_sp * ES, osp4 = out_sp * ES;   // (pinned: the eight scalar deltas are recomputed per
        SR_WN_PIN("+s"(rsp4), "+s"(osp4));                             //  region on the SALU, not parked in VGPR lanes)
        const bool fast_leaky = slope >= 0.0f && slope <= 1.0f;        // LeakyReLU as max(v, slope v): v_pk_mul + 2 v_max per pair
        const wn_f2 slope2 = {slope, slope};
        auto epilogue = [&](auto full_tag, auto res_tag) {
          constexpr bool FULL = decltype(full_tag)::value, RES = decltype(res_tag)::value;
          unsigned okm = 0;   // bit 4 it + q: position inside the image and channel quad below Cout
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
                rv[it][q] = wn_io_load<IO>(rs_res, lane_off(v_res, it, q), s_res0 + d_pix(it, q) * rsp4);
          }
          const float4 bv = wn_buf_load(rs_bias, (FULL || okc) ? 16u * cg : WN_OOB, (unsigned)co0 * 4u);
          SR_TR(10);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            // whole-vector arithmetic: packed fp32 adds, half the VALU instructions of the element-wise form (every VALU
            // instruction here costs matrix-pipe time of the co-resident workgroup)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {   // register PAIRS spelled out: v_pk_add_f32 (the f32x16 form compiled to scalar adds)
              const wn_f2 m0 = {acc[0][n][r], acc[0][n][r + 1]}, m1 = {acc[1][n][r], acc[1][n][r + 1]};
              const wn_f2 m2 = {acc[2][n][r], acc[2][n][r + 1]}, m3 = {acc[3][n][r], acc[3][n][r + 1]};
              const wn_f2 c0 = (m0 + m1) + m2;
              const wn_f2 c1 = __builtin_elementwise_fma(neg1, m3, __builtin_elementwise_fma(neg1, m2, m1));   // (m1 - m2) - m3
              float* o0 = &O[o_wr + ((r & 3) + 8 * (r >> 2)) * CO + 32 * n];
              float* o1 = o0 + 32 * CO;
              o0[0] = c0.x; o0[CO] = c0.y;
              o1[0] = c1.x; o1[CO] = c1.y;
            }
          }
          SR_TR(11);
          __syncthreads();
          SR_TR(13);
          __builtin_amdgcn_sched_barrier(0);   // (keeps the prologue pieces out of the column transform: registers)
          prime_next();   // the accumulators are dead: next region's first weights + first transform, under the stores
          __builtin_amdgcn_sched_barrier(0);
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
              float4 v = f4add(y[q], bv);
              if (RES) v = f4add(v, rv[it][q]);
              if (fast_leaky) {   // (same values as sr_activate_group's max(v, slope v), with the products packed)
                const wn_f2 lo = wn_f2{v.x, v.y} * slope2, hi = wn_f2{v.z, v.w} * slope2;
                v = make_float4(sr_vmax(v.x, lo.x), sr_vmax(v.y, lo.y), sr_vmax(v.z, hi.x), sr_vmax(v.w, hi.y));
              }
              o16[4 * q + 0] = v.x; o16[4 * q + 1] = v.y; o16[4 * q + 2] = v.z; o16[4 * q + 3] = v.w;
            }
            if (!fast_leaky) sr_activate_group(o16, slope);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (!SR_WN_DBG(1) || o16[4 * q] == 1.2345e33f)
                wn_io_store<IO>(make_float4(o16[4 * q], o16[4 * q + 1], o16[4 * q + 2], o16[4 * q + 3]), rs_out,
                             lane_off(v_out, it, q), s_out0 + d_pix(it, q) * osp4);
            }
          }
        };
        if (full) { if (resp) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::true_type{}, std::false_type{}); }
        else { if (resp) epilogue(std::false_type{}, std::true_type{}); else epilogue(std::false_type{}, std::false_type{}); }
#ifdef SR_WINO_TRACE
        if (tr_region > 0) SR_TR(15);
#endif
        __syncthreads();
        SR_TR(14);
      } else {
        constexpr int UNITS = 32 * CO / 256;  // (tile, channel) units per thread
        // (1) residual values first: their latency hides under the LDS exchange (and never sits between stores)
        float rv[UNITS][4];
        bool ok[UNITS][4];
        unsigned opix[UNITS][4];
        const int co = tid & (CO - 1);
        const int cog = co0 + co;
        const bool okc = cog < p.Cout;
#pragma unroll
        for (int it = 0; it < UNITS; ++it) {
          const int tile = tid / CO + (256 / CO) * it;
          const int tr = tile >> 3, tc = tile & 7;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int oy = oy0 + 2 * tr + (q >> 1), ox = ox0 + 2 * tc + (q & 1);
            ok[it][q] = okc & (oy < p.H) & (ox < p.W);
            opix[it][q] = (unsigned)(oy * p.W + ox);
            const bool ld = ok[it][q] & (resp != nullptr);
            const float v = (resp ? resp : p.in)[ld ? opix[it][q] * (unsigned)p.res_sp + cog : 0u];
            rv[it][q] = ld ? v : 0.0f;
          }
        }
        // (2) column transform in registers, then LDS
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tile = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[3][n][r];
            O[((wave * 2 + 0) * 32 + tile) * CO + 32 * n + i] = (m0 + m1) + m2;
            O[((wave * 2 + 1) * 32 + tile) * CO + 32 * n + i] = (m1 - m2) - m3;
          }
        __syncthreads();
        prime_next();
        // (3) row transform, + bias + residual, LeakyReLU, store
        const float bv = (bias_p && okc) ? bias_p[cog] : 0.0f;
#pragma unroll
        for (int it = 0; it < UNITS; ++it) {
          const int tile = tid / CO + (256 / CO) * it;
          float t[4][2];
#pragma unroll
          for (int ur = 0; ur < 4; ++ur)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) t[ur][bb] = O[((ur * 2 + bb) * 32 + tile) * CO + co];
          const float y[4] = {(t[0][0] + t[1][0]) + t[2][0], (t[0][1] + t[1][1]) + t[2][1],
                              (t[1][0] - t[2][0]) - t[3][0], (t[1][1] - t[2][1]) - t[3][1]};
          float o4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) o4[q] = y[q] + bv + rv[it][q];
          sr_activate_group(o4, slope);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (ok[it][q] && (!SR_WN_DBG(1) || o4[q] == 1.2345e33f)) outp[opix[it][q] * out_sp + cog] = o4[q];
        }
        __syncthreads();
      }
    }
    reg = has_next ? nxt : decode(work + (int)gridDim.x < p.total ? work + (int)gridDim.x : work);
  }
}

// split-K finish: out = act(sum_ks partial[ks] + bias + residual), partials added in index order (deterministic)
__global__ __launch_bounds__(256) void sr_wino_reduce_kernel(const float* __restrict__ part, int ksplit,
                                                             int64_t part_stride, const float* __restrict__ bias,
                                                             const float* __restrict__ res, int64_t res_sb, int res_sp,
                                                             float* __restrict__ out, int64_t out_sb, int out_sp,
                                                             int HW, int C4, float slope) {
  const int b = blockIdx.y;
  const int64_t total = (int64_t)HW * C4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    const int64_t px = idx / C4;
    const float* q = part + ((int64_t)b * HW + px) * (4 * C4) + 4 * c4;
    float4 v = *reinterpret_cast<const float4*>(q);
    for (int k = 1; k < ksplit; ++k) v = f4add(v, *reinterpret_cast<const float4*>(q + k * part_stride));
    if (bias) v = f4add(v, *reinterpret_cast<const float4*>(bias + 4 * c4));
    if (res) v = f4add(v, *reinterpret_cast<const float4*>(res + (int64_t)b * res_sb + px * res_sp + 4 * c4));
    float o4[4] = {v.x, v.y, v.z, v.w};
    sr_activate_group(o4, sr_uniform(slope));
    *reinterpret_cast<float4*>(out + (int64_t)b * out_sb + px * out_sp + 4 * c4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
  }
}

int sr_launch_splitk_reduce(const float* part, int ksplit, int64_t part_stride, const float* bias, const float* res,
                            int64_t res_sb, int res_sp, float* out, int64_t out_sb, int out_sp, int B, int HW, int Cout,
                            float slope, hipStream_t stream) {
  const int64_t total = (int64_t)HW * (Cout / 4);
  const int rblocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(sr_wino_reduce_kernel, dim3(rblocks, B), dim3(256), 0, stream, part, ksplit, part_stride, bias, res,
                     res_sb, res_sp, out, out_sb, out_sp, HW, Cout / 4, slope);
  return sr_hip_rc(hipGetLastError());
}

// ------------------------------------------------------------------ C ABI -------------

static int sr_wino_num_cus() { return sr_device_cus(); }

extern "C" size_t sr_wino_packed_weight_floats(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  const size_t G = (size_t)((Cin + 15) / 16) * 2, Co_pad = (size_t)((Cout + 31) / 32) * 32;
  return 16 * G * 2 * Co_pad * 4;
}

extern "C" int sr_wino_pack_weights(const float* weight, int Cout, int Cin, float* packed, void* stream_) {
  if (!weight || !packed || Cout <= 0 || Cin <= 0) return SR_ERR_INVALID_ARGUMENT;
  const int G = ((Cin + 15) / 16) * 2, Co_pad = ((Cout + 31) / 32) * 32;
  const int split = sr_wino_split_mode();   // fenced experiment (sr_wino_split.hip): the same buffer, 16-bit pieces inside
  if (split < 0) return SR_ERR_INVALID_ARGUMENT;
  if (split) return sr_wino_split_pack(weight, Cout, Cin, packed, split, (hipStream_t)stream_);
  hipLaunchKernelGGL(sr_wino_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream_, weight, packed, Cout, Cin, G,
                     Co_pad);
  return sr_hip_rc(hipGetLastError());
}

// 1 if the Winograd kernel is the better choice for this 3x3 / stride-1 conv: enough 8x16-pixel regions to fill the
// machine and little padding waste.  SR_CONV_WINO=0 disables, =2 forces it wherever it is applicable.
extern "C" int sr_conv_prefers_wino(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (ksize != 3 || stride != 1 || B <= 0) return 0;
  const int mode = sr_opt(SR_OPT_CONV_WINO);
  if (mode == 0) return 0;
  if (mode == 2) return 1;
  const long regions = (long)((H + 7) / 8) * ((W + 15) / 16);
  const double util = (double)H * W / (double)(regions * 128);
  const int co_pad = ((Cout + 31) / 32) * 32;
  const int nt = (co_pad % 64 == 0) ? 2 : 1;
  const long tiles = regions * B * (co_pad / (32 * nt));
  (void)tiles;
  return (util >= 0.4 && Cin >= 16) ? 1 : 0;
}

// Launch plan.  Output channels per workgroup: 64 (NT = 2) shares one transformed input slab between two N-tiles; 32
// (NT = 1) makes twice as many, roughly 0.58x as long work items.  Split-K (ks > 1) cuts a work item's chain of input
// slabs into ks independent items whose raw partial outputs a second kernel adds up -- for the deep low-resolution
// layers (e.g. 384 channels at 15x20: 24 slabs in a row on a handful of workgroups).  Launch time = the schedule of
// the busiest CU (+ reduce): its two persistent workgroups own n1 >= n2 items; n2 pairs run concurrently (each item
// then takes 2 x t_thr), the remaining n1 - n2 run alone (t_lat each).  Item length ~ slabs + 2 (prologue / epilogue);
// constants in units of a full NT = 2 item, fitted on the r03 kernel (scripts/wino_plan_sweep.py,
// profiles/r03_wino_plan_sweep.txt: every 3x3 shape of the hero conv stack at batch 8 and 1 under each forced plan).
struct SrWinoPlan { int nt, ks; };
static SrWinoPlan sr_wino_plan(int B, int H, int W, int Cin, int Cout, bool allow_split) {
  const int co_pad = ((Cout + 31) / 32) * 32;
  const int slabs = (Cin + 15) / 16;
  const int forced_nt = sr_opt(SR_OPT_WINO_NT), forced_ks = sr_opt(SR_OPT_WINO_KSPLIT);
  const long regions = (long)((H + 2 * WN_TR - 1) / (2 * WN_TR)) * ((W + 2 * WN_TC - 1) / (2 * WN_TC)) * B;
  const long cus = sr_wino_num_cus(), slots = 2 * cus;
  SrWinoPlan best = {co_pad % 64 == 0 ? 2 : 1, 1};
  double best_cost = -1.0;
  for (int nt = 2; nt >= 1; --nt) {
    if (nt == 2 && co_pad % 64 != 0) continue;
    if ((forced_nt == 1 || forced_nt == 2) && nt != forced_nt && !(forced_nt == 2 && co_pad % 64 != 0)) continue;
    for (int ks = 1; ks <= 8; ks *= 2) {
      if (ks > 1 && (!allow_split || slabs % ks != 0 || slabs / ks < 4 || Cout % 4 != 0)) continue;
      if (forced_ks > 0 && ks != forced_ks && allow_split && slabs % forced_ks == 0 && slabs / forced_ks >= 1) continue;
      const long items = regions * (co_pad / (32 * nt)) * ks;
      const double f = ((double)slabs / ks + 2.0) / ((double)slabs + 2.0);
      const double t_lat = nt == 2 ? 1.0 : 0.6, t_thr = nt == 2 ? 0.82 : 0.46;
      const long n1 = (items + slots - 1) / slots, n2 = items / slots + (items % slots > cus ? 1 : 0);
      double cost = ((double)n2 * 2.0 * t_thr + (double)(n1 - n2) * t_lat) * f;
      if (ks > 1) cost += 2.7 / ((double)slabs + 2.0);            // the reduce launch
      if (nt == 1 || ks > 1) cost *= 1.03;                         // deviate from the default only for a gain
      if (best_cost < 0 || cost < best_cost) { best = {nt, ks}; best_cost = cost; }
    }
  }
  return best;
}

extern "C" int sr_wino_splitk_factor(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 1;
  return sr_wino_plan(B, H, W, Cin, Cout, true).ks;
}

extern "C" size_t sr_wino_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const int ks = sr_wino_splitk_factor(B, H, W, Cin, Cout);
  return ks > 1 ? (size_t)ks * B * H * W * Cout * sizeof(float) : 0;
}

// Symbol of the kernel instantiation sr_conv3x3_wino_nhwc_fwd launches (for profilers / bench): `aligned_in` = input
// rows 16-byte aligned and Cin % 4 == 0, `aligned_out` = output / residual / bias rows 16-byte aligned and Cout % 4 == 0.
extern "C" const char* sr_wino_kernel_name(int B, int H, int W, int Cin, int Cout, int aligned_in, int aligned_out) {
  static thread_local char buf[64];
  const int vin = aligned_in != 0, vout = vin && aligned_out;
  snprintf(buf, sizeof(buf), "sr_wino_kernel<%d, %s, %s>", sr_wino_plan(B, H, W, Cin, Cout, vout != 0).nt,
           vin ? "true" : "false", vout ? "true" : "false");
  return buf;
}

static int sr_wino_run(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_u,
                       const float* bias, const float* residual, int64_t res_batch_stride, int res_pix_stride,
                       float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin, int Cout,
                       float leaky_slope, void* workspace, size_t workspace_bytes, void* stream_, int io = 0) {
  if (io < 0 || io > 2) return SR_ERR_INVALID_ARGUMENT;
  const uintptr_t amask = io ? 7 : 15;   // 4 channels = 16 bytes of fp32 or 8 bytes of fp16 / bf16
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!in || !packed_u || !out) return SR_ERR_INVALID_ARGUMENT;
  SrWinoParams p;
  p.in = in; p.in_sb = in_batch_stride; p.in_sp = in_pix_stride;
  p.wu = packed_u; p.bias = bias;
  p.res = residual; p.res_sb = res_batch_stride; p.res_sp = res_pix_stride;
  p.out = out; p.out_sb = out_batch_stride; p.out_sp = out_pix_stride;
  p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.Co_pad = ((Cout + 31) / 32) * 32;
  p.G = ((Cin + 15) / 16) * 2;
  p.regions_x = (W + 2 * WN_TC - 1) / (2 * WN_TC);
  p.regions_y = (H + 2 * WN_TR - 1) / (2 * WN_TR);
  p.slope = leaky_slope;
  p.vec4 = (((uintptr_t)in & amask) == 0) && (in_pix_stride % 4 == 0) && (in_batch_stride % 4 == 0) && (Cin % 4 == 0);
  // vector epilogue: whole 4-channel groups, aligned output / residual / bias rows
  const bool vout = p.vec4 && (Cout % 4 == 0) && (((uintptr_t)out & amask) == 0) && (out_pix_stride % 4 == 0) &&
                    (out_batch_stride % 4 == 0) && (!bias || ((uintptr_t)bias & 15) == 0) &&
                    (!residual || ((((uintptr_t)residual & amask) == 0) && (res_pix_stride % 4 == 0) &&
                                   (res_batch_stride % 4 == 0)));
  if (io && !vout) return SR_ERR_UNSUPPORTED;   // 16-bit I/O: the vector instantiation only
  const int split = sr_wino_split_mode();
  if (split < 0) return SR_ERR_INVALID_ARGUMENT;
  if (split && (io || !vout)) return SR_ERR_UNSUPPORTED;   // split precision: fp32 tensors, the vector instantiation only
  const int64_t lim = (int64_t)1 << 31;          // per-image byte offsets are 32-bit (buffer addressing)
  // (the input-side limit applies to the vector STAGING too -- sr_wino_kernel<NT, true, false> --, not only to the vector
  // epilogue: its buffer descriptors carry 32-bit byte offsets just the same.  Callers fall back to sr_conv2d_nhwc_fwd.)
  if (p.vec4 && ((int64_t)((int64_t)H * W - 1) * in_pix_stride + Cin) * 4 >= lim) return SR_ERR_UNSUPPORTED;
  if (vout && (((int64_t)((int64_t)H * W - 1) * out_pix_stride + Cout) * 4 >= lim ||
               (residual && ((int64_t)((int64_t)H * W - 1) * res_pix_stride + Cout) * 4 >= lim)))
    return SR_ERR_UNSUPPORTED;
  const bool can_split = !io && vout && workspace && (((uintptr_t)workspace & 15) == 0);   // (partials are fp32: fp32 I/O only)
  SrWinoPlan plan = sr_wino_plan(B, H, W, Cin, Cout, can_split);
  if (plan.ks > 1 && workspace_bytes < (size_t)plan.ks * B * H * W * Cout * sizeof(float))
    plan = sr_wino_plan(B, H, W, Cin, Cout, false);
  const int nt = plan.nt;
  p.ksplit = plan.ks;
  p.part = (float*)workspace;
  p.part_stride = (int64_t)B * H * W * Cout;
  p.co_blocks = p.Co_pad / (32 * nt);
  p.total = p.regions_x * p.regions_y * p.co_blocks * B * p.ksplit;
#ifdef SR_WINO_ABLATION   // (ablation builds: read per call, so that one process can sweep the switches)
  { const char* e = getenv("SR_WINO_DEBUG"); p.debug = e ? atoi(e) : 0; }
#else
  p.debug = 0;
#endif
  p.xcd_order = sr_opt(SR_OPT_WINO_XCD);
  p.stagger = sr_opt(SR_OPT_WINO_STAGGER);   // experiment switch, see below
  hipStream_t stream = (hipStream_t)stream_;
  int blocks = sr_wino_num_cus() * (nt == 1 ? SR_WINO_NT1_WAVES : SR_WINO_WAVES);
  if (sr_opt(SR_OPT_WINO_WG_PER_CU) == 1) blocks = sr_wino_num_cus();  // ablation
  if (blocks > p.total) blocks = p.total;
  if (p.stagger < 0) p.stagger = 0;
  // (Measured, r04: an offset of 3 x s_sleep(127) is -14 % / -15 % on 64 -> 64 / 192 -> 64 @ 8x240x320 in an isolated loop over
  // one layer -- 250 -> 216 us, 614 -> 519 us -- and NOTHING in the step: the same layers run 142 us on average inside the model
  // with or without it, 28.44 vs 28.64 ms per step.  Launched back to back from a queue that runs ahead of the GPU the
  // workgroups of a launch do not start in phase in the first place; the loop's launch gaps made them.  Off by default.)
  const size_t lds = (size_t)WN_LDS_FLOATS(nt) * sizeof(float);
#ifdef SR_WINO_TRACE
  static unsigned long long* trace_buf = nullptr;
  static int launches = 0;
  const size_t trace_n = (size_t)blocks * SR_TR_REGIONS * SR_TR_EVENTS;
  if (!trace_buf) (void)hipMalloc((void**)&trace_buf, (size_t)1024 * SR_TR_REGIONS * SR_TR_EVENTS * 8);
  (void)hipMemsetAsync(trace_buf, 0, trace_n * 8, stream);
  p.trace = trace_buf;
#endif
#define SR_WINO_LAUNCH_IO(NTV, IOV)                                                                              \
  {                                                                                                               \
    hipError_t e = hipFuncSetAttribute((const void*)sr_wino_kernel<NTV, true, true, IOV>,                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
    if (e != hipSuccess) return sr_hip_rc(e);                                                                     \
    hipLaunchKernelGGL((sr_wino_kernel<NTV, true, true, IOV>), dim3(blocks), dim3(256), lds, stream, p);          \
  }
#define SR_WINO_LAUNCH(NTV, V4, VO)                                                                               \
  {                                                                                                               \
    hipError_t e = hipFuncSetAttribute((const void*)sr_wino_kernel<NTV, V4, VO>,                                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
    if (e != hipSuccess) return sr_hip_rc(e);                                                                     \
    hipLaunchKernelGGL((sr_wino_kernel<NTV, V4, VO>), dim3(blocks), dim3(256), lds, stream, p);                   \
  }
  if (split) { const int rc = sr_wino_split_launch(p, nt, blocks, split, stream); if (rc) return rc; }
  else if (io == 1 && nt == 2) SR_WINO_LAUNCH_IO(2, 1)
  else if (io == 1) SR_WINO_LAUNCH_IO(1, 1)
  else if (io == 2 && nt == 2) SR_WINO_LAUNCH_IO(2, 2)
  else if (io == 2) SR_WINO_LAUNCH_IO(1, 2)
  else if (nt == 2 && vout) SR_WINO_LAUNCH(2, true, true)
  else if (nt == 2 && p.vec4) SR_WINO_LAUNCH(2, true, false)
  else if (nt == 2) SR_WINO_LAUNCH(2, false, false)
  else if (vout) SR_WINO_LAUNCH(1, true, true)
  else if (p.vec4) SR_WINO_LAUNCH(1, true, false)
  else SR_WINO_LAUNCH(1, false, false)
#undef SR_WINO_LAUNCH
#undef SR_WINO_LAUNCH_IO
#ifdef SR_WINO_TRACE
  {
    const char* path = getenv("SR_WINO_TRACE_FILE");
    const char* at = getenv("SR_WINO_TRACE_LAUNCH");
    if (path && ++launches == (at ? atoi(at) : 10)) {
      (void)hipStreamSynchronize(stream);
      unsigned long long* host = (unsigned long long*)malloc(trace_n * 8);
      (void)hipMemcpy(host, trace_buf, trace_n * 8, hipMemcpyDeviceToHost);
      FILE* f = fopen(path, "wb");
      if (f) { int hdr[4] = {blocks, SR_TR_REGIONS, SR_TR_EVENTS, p.G / 2}; fwrite(hdr, 4, 4, f); fwrite(host, 8, trace_n, f); fclose(f); }
      free(host);
    }
  }
#endif
  int rc = sr_hip_rc(hipGetLastError());
  if (rc == SR_OK && p.ksplit > 1) {
    rc = sr_launch_splitk_reduce(p.part, p.ksplit, p.part_stride, bias, residual, res_batch_stride, res_pix_stride, out,
                                 out_batch_stride, out_pix_stride, B, H * W, Cout, leaky_slope, stream);
  }
  return rc;
}

extern "C" int sr_conv3x3_wino_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                        const float* packed_u, const float* bias, const float* residual,
                                        int64_t res_batch_stride, int res_pix_stride, float* out,
                                        int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                        int Cout, float leaky_slope, void* stream_) {
  return sr_wino_run(in, in_batch_stride, in_pix_stride, packed_u, bias, residual, res_batch_stride, res_pix_stride, out,
                     out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, leaky_slope, nullptr, 0, stream_);
}

// The same operator on fp16 (io_dtype = 1) / bf16 (2) activation tensors: input, residual and output in HBM are 16-bit
// (strides in ELEMENTS), weights / bias / transforms / accumulation fp32; io_dtype = 0 is sr_conv3x3_wino_nhwc_fwd.
extern "C" int sr_conv3x3_wino_io_nhwc_fwd(const void* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_u,
                                           const float* bias, const void* residual, int64_t res_batch_stride,
                                           int res_pix_stride, void* out, int64_t out_batch_stride, int out_pix_stride, int B,
                                           int H, int W, int Cin, int Cout, float leaky_slope, int io_dtype, void* stream_) {
  return sr_wino_run((const float*)in, in_batch_stride, in_pix_stride, packed_u, bias, (const float*)residual,
                     res_batch_stride, res_pix_stride, (float*)out, out_batch_stride, out_pix_stride, B, H, W, Cin, Cout,
                     leaky_slope, nullptr, 0, stream_, io_dtype);
}

extern "C" int sr_conv3x3_wino_splitk_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                               const float* packed_u, const float* bias, const float* residual,
                                               int64_t res_batch_stride, int res_pix_stride, float* out,
                                               int64_t out_batch_stride, int out_pix_stride, int B, int H, int W,
                                               int Cin, int Cout, float leaky_slope, void* workspace,
                                               size_t workspace_bytes, void* stream_) {
  return sr_wino_run(in, in_batch_stride, in_pix_stride, packed_u, bias, residual, res_batch_stride, res_pix_stride, out,
                     out_batch_stride, out_pix_stride, B, H, W, Cin, Cout, leaky_slope, workspace, workspace_bytes,
                     stream_);
}
