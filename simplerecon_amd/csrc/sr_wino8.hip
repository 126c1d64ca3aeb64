// sr_wino8.hip -- Winograd F(2x2, 3x3) with ONE 8-wave workgroup per CU and a continuous MFMA stream (gfx950).
//
// Same operator, same packed weights, same arithmetic and the same order of every floating-point operation as
// sr_wino_kernel<2, true, true> (sr_wino.hip) -- results are bit-identical -- but a different schedule.  The 4-wave kernel
// keeps 128 accumulator registers per lane, which leaves nothing to software-pipeline with: its input transform and its
// epilogue run as serial phases, a workgroup issues MFMAs only ~38 % of its time on a 4-slab layer and two co-resident
// workgroups cannot close the gap (r02 ablations and s_memtime trace).  Here a region's 64 output
// channels are split over two waves (wave = frequency row u x channel half h: 64 accumulators per lane), which frees the
// registers to
//   * park a finished region's column-transformed values (M A: 32 registers) while the accumulators start over: the LDS
//     exchange, row transform, residual add and stores of region r are issued in four pieces between the MFMAs of the
//     first four slabs of region r + 1;
//   * run the input transform of slab g + 1 (raw -> V, double-buffered V) and the global -> LDS staging of slab g + 2
//     between the MFMAs of slab g; the slab stream is continuous across regions (weights are prefetched across the
//     region boundary too), so after the prologue a wave never stops issuing MFMAs except at the one barrier per slab.
// LDS: V 2 x 40 KB + raw patches 2 x 14.1 KB + exchange slab 32 KB = 140 KB, one workgroup per CU, two waves per SIMD.
//
// Wave (u, h): transform row u of B^T d B for tiles 16 h .. 16 h + 15, multiplies the 4 frequencies 4 u .. 4 u + 3 of
// all 32 tiles with the weights of channels 32 h .. 32 h + 31.
#include <type_traits>

#include "sr_wino.h"

#define W8_V_FLOATS WN_V_FLOATS            // [16 freq][32 tiles][20]
#define W8_RAW_FLOATS WN_RAW_FLOATS        // [10 * 18 px][20]
#define W8_O_FLOATS (4 * 32 * 64)          // [4 ur][32 tiles][64 co]: one column (bb) of the exchange at a time
#define W8_DUMMY_FLOATS (27 * 64 + 64)    // reach of the column-piece writes from a lane's base (dummy target)
#define W8_LDS_FLOATS (2 * W8_V_FLOATS + 2 * W8_RAW_FLOATS + W8_O_FLOATS + W8_DUMMY_FLOATS)
#define W8_NB 4
#define W8_PD 3

struct W8Region { int b, oy0, ox0, co0, ks; };

// compile-time loop: the body sees its index as a constant in the front end already (register arrays indexed by a
// `#pragma unroll` loop variable are only constant after unrolling, too late for some of them to leave memory)
template <int S, int N, typename F>
__device__ __forceinline__ void w8_static_for(F&& f) {
  if constexpr (S < N) {
    f(std::integral_constant<int, S>{});
    w8_static_for<S + 1, N>(f);
  }
}

typedef unsigned int w8_u4 __attribute__((ext_vector_type(4)));
#define W8_RSRC_FLAGS 0x00020000      // raw buffer descriptor word 3 on gfx9-family parts
#define W8_OOB 0x7fffffffu            // voffset beyond any num_records: the load returns 0, the store is dropped

// Buffer loads / stores with a uniform descriptor and a 32-bit lane offset: out-of-range lanes (image border, channel
// tail, "no epilogue piece due in this slab") are switched off by their OFFSET, not by a branch -- the slab body below
// stays straight-line code and the compiler's s_waitcnt counts stay exact (a conditional memory op between a prefetch
// and its use makes it wait for everything).
__device__ __forceinline__ float4 w8_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const w8_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void w8_buf_store(const float4& v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const w8_u4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, 0, 0);
}

__global__ __launch_bounds__(512, 2) void sr_wino8_kernel(SrWinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Vbuf = lds;
  float* const Rbuf = lds + 2 * W8_V_FLOATS;
  float* const O = lds + 2 * W8_V_FLOATS + 2 * W8_RAW_FLOATS;
  float* const Dummy = O + W8_O_FLOATS;   // where the column-piece writes go when no column piece is due
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = wave & 3, h = wave >> 2;
  const int i = lane & 31, kk = lane >> 5;
  const int chunks = (p.G >> 1) / p.ksplit;   // input slabs per region
  const bool partial = p.ksplit > 1;
  constexpr int NB = W8_NB, PD = W8_PD;

  // transform role (see sr_wino.hip): row u of B^T d needs patch rows (t_ra, t_rb): d_ra + t_sign * d_rb
  const int tq = lane & 3, tt = 16 * h + (lane >> 2);
  const int t_ra = u == 0 ? 0 : (u == 2 ? 2 : 1), t_rb = u == 0 ? 2 : (u == 1 ? 2 : (u == 2 ? 1 : 3));
  const float t_sign = u == 1 ? 1.0f : -1.0f;
  const int t_base = ((2 * (tt >> 3)) * WN_PW + 2 * (tt & 7)) * WN_ROW + 4 * tq;
  const int t_oa = t_base + t_ra * WN_PW * WN_ROW, t_ob = t_base + t_rb * WN_PW * WN_ROW;
  const int t_vo = ((4 * u) * 32 + tt) * WN_ROW + 4 * tq;
  // MFMA role: A fragment of (frequency xi, 8-channel group g) at V[(xi * 32 + i) * 20 + 8 g + 4 kk]
  const int a_base = ((4 * u) * 32 + i) * WN_ROW + 4 * kk;
  // epilogue role: thread = (tile, 4 consecutive channels); column-piece role: accumulator element r of this lane is
  // tile (r & 3) + 8 (r >> 2) + 4 kk, channel 32 h + i
  const int e_tile = tid >> 4, e_cg = tid & 15;
  const int o_lane = (u * 32 + 4 * kk) * 64 + 32 * h + i;

  auto decode = [&](int wk_) __attribute__((always_inline)) {   // unsigned: no sign fix-ups around the divisions
    W8Region r;
    unsigned wk = (unsigned)wk_;
    const unsigned ksn = (unsigned)p.ksplit, cbn = (unsigned)p.co_blocks, rxn = (unsigned)p.regions_x, ryn = (unsigned)p.regions_y;
    r.ks = (int)(wk % ksn); wk /= ksn;
    const unsigned cb = wk % cbn; wk /= cbn;
    const unsigned rx = wk % rxn; wk /= rxn;
    const unsigned ry = wk % ryn;
    r.b = (int)(wk / ryn);
    r.oy0 = (int)ry * (2 * WN_TR); r.ox0 = (int)rx * (2 * WN_TC); r.co0 = (int)cb * 64;
    return r;
  };

  // ---- staging cursor: the slab stream (region, slab) in execution order, three slabs ahead of the MFMAs ----
  int st_work = blockIdx.x, st_ch = 0;
  bool st_valid = st_work < p.total;
  unsigned offs0 = W8_OOB, offs1 = W8_OOB;   // byte offset of this thread's two patch elements inside the image, or OOB
  const unsigned in_bytes = (unsigned)(((int64_t)(p.H * p.W - 1) * p.in_sp + p.Cin) * 4);
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, W8_RSRC_FLAGS);
  int st_c0 = 0;
  auto aim = [&](int wk) __attribute__((always_inline)) {
    const W8Region r = decode(wk);
    rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (int64_t)r.b * p.in_sb), 0, (int)in_bytes, W8_RSRC_FLAGS);
    st_c0 = r.ks * chunks * 16;
    auto one = [&](int e) __attribute__((always_inline)) {
      const int px = e >> 2, q = e & 3;
      const int py = px / WN_PW, pxx = px - py * WN_PW;
      const int iy = r.oy0 - 1 + py, ix = r.ox0 - 1 + pxx;
      const bool ok = (e < WN_STAGE_ELEMS) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
      return ok ? (unsigned)((iy * p.W + ix) * p.in_sp + 4 * q) * 4u : W8_OOB;
    };
    offs0 = one(tid);
    offs1 = one(tid + 512);
  };
  if (st_valid) aim(st_work);
  auto stage_load = [&](float4 (&stg)[2]) __attribute__((always_inline)) {  // loads the cursor's slab (zeros past the end of the stream) and advances
    const int c0 = st_c0 + st_ch * 16;
    const bool okc = st_valid & (c0 + 4 * (tid & 3) < p.Cin);
    stg[0] = w8_buf_load(rs_in, (okc & (offs0 != W8_OOB)) ? offs0 + (unsigned)c0 * 4u : W8_OOB, 0);
    stg[1] = w8_buf_load(rs_in, (okc & (offs1 != W8_OOB)) ? offs1 + (unsigned)c0 * 4u : W8_OOB, 0);
  };
  auto stage_advance = [&]() __attribute__((always_inline)) {
    if (st_valid) {
      if (++st_ch == chunks) {
        st_ch = 0;
        st_work += gridDim.x;
        st_valid = st_work < p.total;
        if (st_valid) aim(st_work);
      }
    }
  };
  auto stage_store = [&](const float4 (&stg)[2], float* raw) __attribute__((always_inline)) {
    *reinterpret_cast<float4*>(&raw[(tid >> 2) * WN_ROW + 4 * (tid & 3)]) = stg[0];
    if (tid + 512 < WN_STAGE_ELEMS)
      *reinterpret_cast<float4*>(&raw[((tid + 512) >> 2) * WN_ROW + 4 * (tid & 3)]) = stg[1];
  };

  auto t_load = [&](const float* raw, int c, float4& da, float4& db) __attribute__((always_inline)) {
    da = *reinterpret_cast<const float4*>(&raw[t_oa + c * WN_ROW]);
    db = *reinterpret_cast<const float4*>(&raw[t_ob + c * WN_ROW]);
  };
  auto t_pair = [&](const float4& da, const float4& db) __attribute__((always_inline)) {  // d_ra[c] + t_sign * d_rb[c]
    const wn_f2 sg = {t_sign, t_sign};
    const wn_f2 lo = __builtin_elementwise_fma(sg, wn_f2{db.x, db.y}, wn_f2{da.x, da.y});
    const wn_f2 hi = __builtin_elementwise_fma(sg, wn_f2{db.z, db.w}, wn_f2{da.z, da.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  };
  // the whole row transform of one slab in one go (prologue only)
  auto transform_all = [&](const float* raw, float* V) __attribute__((always_inline)) {
    float4 wv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 da, db;
      t_load(raw, c, da, db);
      wv[c] = t_pair(da, db);
    }
    float* vrow = V + t_vo;
    *reinterpret_cast<float4*>(vrow + 0 * 32 * WN_ROW) = f4sub(wv[0], wv[2]);
    *reinterpret_cast<float4*>(vrow + 1 * 32 * WN_ROW) = f4add(wv[1], wv[2]);
    *reinterpret_cast<float4*>(vrow + 2 * 32 * WN_ROW) = f4sub(wv[2], wv[1]);
    *reinterpret_cast<float4*>(vrow + 3 * 32 * WN_ROW) = f4sub(wv[1], wv[3]);
  };

  // ---- epilogue of a finished region: its coordinates and the descriptors of its output / residual image ----
  // The column half of Y = A^T M A happens in registers (the wave holds a whole frequency row): col0 = (m0 + m1) + m2,
  // col1 = (m1 - m2) - m3 -- 32 registers that wait for their turn in the exchange slab while the accumulators are
  // already collecting the next region.
  W8Region er = {0, 0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0, W8_RSRC_FLAGS);
  __amdgpu_buffer_rsrc_t rs_res = rs_out;
  const __amdgpu_buffer_rsrc_t rs_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.wu), 0, (p.bias && !partial) ? p.Cout * 4 : 0, W8_RSRC_FLAGS);
  const unsigned out_sp = partial ? (unsigned)p.Cout : (unsigned)p.out_sp;
  const float e_slope = partial ? -1.0f : p.slope;
  // per region and thread: pixel index of the tile's (row 0, column 0) output and which of its 2 x 2 pixels exist
  int e_pix = 0;
  unsigned e_okbits = 0;   // bit 2 * row + bb
  auto aim_epilogue = [&](const W8Region& r) __attribute__((always_inline)) {
    er = r;
    float* outp = partial ? p.part + r.ks * p.part_stride + (int64_t)r.b * p.H * p.W * p.Cout : p.out + (int64_t)r.b * p.out_sb;
    rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)(((int64_t)(p.H * p.W - 1) * out_sp + p.Cout) * 4),
                                               W8_RSRC_FLAGS);
    const bool has_res = p.res && !partial;
    rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? p.res + (int64_t)r.b * p.res_sb : p.wu), 0,
                                               has_res ? (int)(((int64_t)(p.H * p.W - 1) * p.res_sp + p.Cout) * 4) : 0,
                                               W8_RSRC_FLAGS);
    const int oy = r.oy0 + 2 * (e_tile >> 3), ox = r.ox0 + 2 * (e_tile & 7);
    const bool okc = r.co0 + 4 * e_cg < p.Cout;
    e_pix = oy * p.W + ox;
    e_okbits = ((okc & (oy < p.H) & (ox < p.W)) ? 1u : 0u) | ((okc & (oy < p.H) & (ox + 1 < p.W)) ? 2u : 0u) |
               ((okc & (oy + 1 < p.H) & (ox < p.W)) ? 4u : 0u) | ((okc & (oy + 1 < p.H) & (ox + 1 < p.W)) ? 8u : 0u);
  };
  // pixel (row, column bb) of this thread's tile: byte offset in an image with pixel stride sp, or OOB
  auto e_off = [&](int bb, int row, unsigned sp, bool active) __attribute__((always_inline)) {
    const bool ok = active & (((e_okbits >> (2 * row + bb)) & 1u) != 0u);
    return ok ? ((unsigned)(e_pix + row * p.W + bb) * sp + (unsigned)(er.co0 + 4 * e_cg)) * 4u : W8_OOB;
  };
  auto res_loads = [&](int bb, bool active, float4 (&rv)[2], float4& bv) __attribute__((always_inline)) {
    bv = w8_buf_load(rs_bias, (unsigned)(er.co0 + 4 * e_cg) * 4u, 0);   // Cout is a multiple of 4: whole groups or OOB
    rv[0] = w8_buf_load(rs_res, e_off(bb, 0, (unsigned)p.res_sp, active), 0);
    rv[1] = w8_buf_load(rs_res, e_off(bb, 1, (unsigned)p.res_sp, active), 0);
  };
  auto col_write = [&](const f32x16& c0v, const f32x16& c1v, int piece, auto r0_) __attribute__((always_inline)) {  // 4 elements
    const bool due = (piece == 0) | (piece == 2);
    float* base = due ? O + o_lane : Dummy + lane;
    w8_static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
      constexpr int r = decltype(r0_)::value + decltype(k_)::value;
      base[((r & 3) + 8 * (r >> 2)) * 64] = piece == 0 ? c0v[r] : c1v[r];
    });
  };
  auto out_reads = [&](float4 (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int ur = 0; ur < 4; ++ur) t[ur] = *reinterpret_cast<const float4*>(&O[(ur * 32 + e_tile) * 64 + 4 * e_cg]);
  };
  auto out_sums = [&](const float4 (&t)[4], float4 (&ey)[2]) __attribute__((always_inline)) {
    ey[0] = f4add(f4add(t[0], t[1]), t[2]);
    ey[1] = f4sub(f4sub(t[1], t[2]), t[3]);
  };
  auto out_row = [&](int bb, int row, bool active, const float4 (&ey)[2], const float4 (&rv)[2], const float4& bv) __attribute__((always_inline)) {
    float4 v = f4add(f4add(ey[row], bv), rv[row]);
    if (e_slope >= 0.0f) {          // LeakyReLU / ReLU: the same expression as sr_activate, one uniform branch per row
      v.x = fmaxf(v.x, 0.0f) + e_slope * fminf(v.x, 0.0f);
      v.y = fmaxf(v.y, 0.0f) + e_slope * fminf(v.y, 0.0f);
      v.z = fmaxf(v.z, 0.0f) + e_slope * fminf(v.z, 0.0f);
      v.w = fmaxf(v.w, 0.0f) + e_slope * fminf(v.w, 0.0f);
    } else if (e_slope < -1.5f) {   // SiLU
      v.x = sr_activate(v.x, e_slope);
      v.y = sr_activate(v.y, e_slope);
      v.z = sr_activate(v.z, e_slope);
      v.w = sr_activate(v.w, e_slope);
    }
    w8_buf_store(v, rs_out, e_off(bb, row, out_sp, active));
  };
  auto flush_serial = [&](const f32x16& c0v, const f32x16& c1v) __attribute__((always_inline)) {  // the four pieces back to back
    w8_static_for<0, 2>([&](auto bb_) __attribute__((always_inline)) {
      constexpr int bb = decltype(bb_)::value;
      float4 frv[2], fey[2], fet[4], fbv;
      res_loads(bb, true, frv, fbv);
      w8_static_for<0, 4>([&](auto q_) __attribute__((always_inline)) {
        col_write(c0v, c1v, 2 * bb, std::integral_constant<int, 4 * decltype(q_)::value>{});
      });
      __syncthreads();
      out_reads(fet);
      out_sums(fet, fey);
      out_row(bb, 0, true, fey, frv, fbv);
      out_row(bb, 1, true, fey, frv, fbv);
      __syncthreads();
    });
  };

  // ---- weights: B fragments stream from L2, PD steps ahead through NB rotating register sets ----
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wu, 0, (int)((int64_t)16 * p.G * 2 * p.Co_pad * 16), W8_RSRC_FLAGS);
  const unsigned w_lane = (unsigned)(kk * p.Co_pad + 32 * h + i) * 16u;
  const unsigned w_rec = (unsigned)(2 * p.Co_pad) * 16u;   // bytes per (frequency, 8-channel group) record
  auto w_base = [&](const W8Region& r, int ch) __attribute__((always_inline)) {  // byte offset of (frequency 4 u, group 0) of slab ch of region r
    return (unsigned)r.co0 * 16u + (unsigned)(2 * (r.ks * chunks + ch) + 4 * u * p.G) * w_rec;
  };
  unsigned w_off[8];   // lane offset of step s: frequency 4 u + s / 2, channel group s % 2 (loop-invariant registers)
#pragma unroll
  for (int s = 0; s < 8; ++s) w_off[s] = w_lane + (unsigned)((s >> 1) * p.G + (s & 1)) * w_rec;
  auto w_step = [&](unsigned base, auto s_) __attribute__((always_inline)) {
    return w8_buf_load(rs_w, w_off[decltype(s_)::value], base);
  };

  // ---- loop-carried state ----
  float4 b_f[NB], a_f[4];
  float4 stg[2];            // the slab that T reads two slabs from now, on its way from global memory
  float4 rv[2], bv;         // residual / bias values of the next output piece
  f32x16 acc[4], col0 = {}, col1 = {};
  // control of the CURRENT slab (uniform) and, computed inside it, of the next one
  int work = blockIdx.x;
  if (work >= p.total) return;
  W8Region reg = decode(work);
  bool has_next = work + (int)gridDim.x < p.total;
  W8Region nreg = has_next ? decode(work + (int)gridDim.x) : reg;
  const bool ovl = chunks >= 4 && !(p.debug & 64);
  bool pend = false;        // a finished region's column values wait in col0 / col1 for their epilogue
  int ch = 0, pz = 0;
  unsigned wcur = w_base(reg, 0), wnxt = chunks == 1 ? w_base(nreg, 0) : w_base(reg, 1);

  // ---- prologue: slabs 0 and 1 of the stream into the raw buffers, slab 2 into registers, slab 0 transformed ----
  stage_load(stg); stage_advance(); stage_store(stg, Rbuf);
  stage_load(stg); stage_advance(); stage_store(stg, Rbuf + W8_RAW_FLOATS);
  stage_load(stg); stage_advance();
  __syncthreads();
  transform_all(Rbuf, Vbuf);
  w8_static_for<0, PD>([&](auto s_) __attribute__((always_inline)) { b_f[decltype(s_)::value] = w_step(wcur, s_); });
  __syncthreads();
  a_f[0] = *reinterpret_cast<const float4*>(&Vbuf[a_base]);
  a_f[1] = *reinterpret_cast<const float4*>(&Vbuf[a_base + 8]);

#ifdef SR_WINO_TRACE
  int tr_k = 0;
#define W8_TR()                                                                                                   \
  do {                                                                                                            \
    if (tid == 0 && tr_k < SR_TR_REGIONS * SR_TR_EVENTS)                                                           \
      p.trace[(size_t)blockIdx.x * SR_TR_REGIONS * SR_TR_EVENTS + tr_k++] = __builtin_amdgcn_s_memtime();          \
  } while (0)
#else
#define W8_TR() do {} while (0)
#endif

  // One slab: 32 MFMAs of this wave (4 frequencies x 2 channel groups x 4 k-steps) with everything else in the issue
  // slots between them -- straight-line code, one barrier (after step 5):
  //   slot A (after .x): weight fragment of step s + PD
  //   slot B (after .y): steps 0-3 the transform of the NEXT slab (patch reads one step before their use); steps 2-5
  //                      one epilogue piece of the PREVIOUS region (piece 0 / 2: column values -> exchange slab,
  //                      piece 1 / 3: exchange slab -> row transform, bias, residual, activation, stores; when no piece
  //                      is due the LDS writes go to a dummy line and the stores are out of range); step 5: hand-over
  //                      of the staged slab; step 6 (behind the barrier): global loads of the slab after that, residual
  //                      loads of the next slab's output piece, control of the next slab; steps 4 / 6 of a region's last
  //                      slab: column half of ITS output transform (acc[0..2] are final by then)
  //   slot C (after .z): A fragment of step s + 2 (steps 6, 7: from the next slab's V -- complete behind the barrier)
  // Everything the barrier orders lies on one side of it: T writes / staged patch (before) vs. their readers in the next
  // slab; this slab's V reads (all issued by step 5) vs. the next slab's T writes; an epilogue piece (steps 2-5) vs. the
  // next piece one slab later.
  while (true) {
    const bool first = ch == 0, last = ch + 1 == chunks;
    const int piece = (pend && ch < 4) ? ch : -1;
    const bool outp = (piece == 1) | (piece == 3);
    const int ebb = piece >> 1;   // output column of an output piece
    const float* Vc = Vbuf + pz * W8_V_FLOATS;
    float* Vn = Vbuf + (pz ^ 1) * W8_V_FLOATS;
    const float* rawT = Rbuf + (pz ^ 1) * W8_RAW_FLOATS;
    float* rawS = Rbuf + pz * W8_RAW_FLOATS;
    float* vrow = Vn + t_vo;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 wv0, wv1, wv2, wv3, da0, db0, da1, db1, ey[2], et[4];
    // control of the next slab (filled in at step 6)
    int n_ch = 0;
    bool n_has_next = has_next, n_pend = pend;
    int nn_b = 0, nn_oy0 = 0, nn_ox0 = 0, nn_co0 = 0, nn_ks = 0;   // the region after the next one (decoded in a slot)
    unsigned n_wnxt = wnxt;
    W8_TR();
    w8_static_for<0, 8>([&](auto s_) __attribute__((always_inline)) {
      constexpr int s = decltype(s_)::value;
      constexpr int cb = s % NB, ca = s & 3;
      if (!(s & 1) && first) acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].x, b_f[cb].x, zero16, 0, 0, 0);
      else acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].x, b_f[cb].x, acc[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (s + PD < 8) b_f[(s + PD) % NB] = w_step(wcur, std::integral_constant<int, (s + PD) % 8>{});
      else b_f[(s + PD) % NB] = w_step(wnxt, std::integral_constant<int, (s + PD) % 8>{});
      __builtin_amdgcn_sched_barrier(0);
      acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].y, b_f[cb].y, acc[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- slot B ----
      if (s == 0) { t_load(rawT, 0, da0, db0); t_load(rawT, 2, da1, db1); }
      if (s == 1) {
        wv0 = t_pair(da0, db0); wv2 = t_pair(da1, db1);
        *reinterpret_cast<float4*>(vrow + 0 * 32 * WN_ROW) = f4sub(wv0, wv2);
        t_load(rawT, 1, da0, db0); t_load(rawT, 3, da1, db1);
      }
      if (s == 2) {
        wv1 = t_pair(da0, db0); wv3 = t_pair(da1, db1);
        *reinterpret_cast<float4*>(vrow + 1 * 32 * WN_ROW) = f4add(wv1, wv2);
        *reinterpret_cast<float4*>(vrow + 2 * 32 * WN_ROW) = f4sub(wv2, wv1);
      }
      if (s == 3) {
        *reinterpret_cast<float4*>(vrow + 3 * 32 * WN_ROW) = f4sub(wv1, wv3);
        out_reads(et);
      }
      if (s >= 2 && s <= 5) col_write(col0, col1, piece, std::integral_constant<int, (s >= 2 && s <= 5) ? 4 * (s - 2) : 0>{});
      if (s == 4) {
        out_sums(et, ey);
        out_row(ebb, 0, outp, ey, rv, bv);
        if (last) {   // acc[0], acc[1] are final (steps 0-3); the previous col0 left in slab 0
          asm volatile("" ::: "memory");   // keep this a branch: if-converted it would run (and select) in every slab
          col0 = acc[0] + acc[1];
        }
      }
      if (s == 5) {
        out_row(ebb, 1, outp, ey, rv, bv);
        stage_store(stg, rawS);
      }
      if (s == 6) {   // behind the barrier: short -- both waves of a SIMD are here at the same time
        stage_load(stg);
        // residual / bias values of the next slab's output piece (pieces 1 and 3 of the region in `er`)
        const int n_piece = (pend && !last && ch + 1 < 4) ? ch + 1 : -1;
        if (!last) res_loads(n_piece >> 1, (n_piece == 1) | (n_piece == 3), rv, bv);
      }
      if (s == 7) {
        stage_advance();
        if (last) {   // acc[2] is final (steps 4-5)
          asm volatile("" ::: "memory");
          col0 = col0 + acc[2];
          col1 = acc[1] - acc[2];
        }
        // control of the next slab
        n_ch = ch + 1;
        nn_b = nreg.b; nn_oy0 = nreg.oy0; nn_ox0 = nreg.ox0; nn_co0 = nreg.co0; nn_ks = nreg.ks;
        if (last) {
          n_ch = 0;
          n_has_next = work + 2 * (int)gridDim.x < p.total;
          if (n_has_next) {
            const W8Region d = decode(work + 2 * (int)gridDim.x);
            nn_b = d.b; nn_oy0 = d.oy0; nn_ox0 = d.ox0; nn_co0 = d.co0; nn_ks = d.ks;
          }
          n_pend = ovl && has_next;
        }
        const bool n_last = n_ch + 1 == chunks;
        // weights of the slab after the next one: same region, or slab 0 of the region that follows it
        const int f_co0 = last ? (n_last ? nn_co0 : nreg.co0) : (n_last ? nreg.co0 : reg.co0);
        const int f_ks = last ? (n_last ? nn_ks : nreg.ks) : (n_last ? nreg.ks : reg.ks);
        const int f_ch = n_last ? 0 : n_ch + 1;
        n_wnxt = (unsigned)f_co0 * 16u + (unsigned)(2 * (f_ks * chunks + f_ch) + 4 * u * p.G) * w_rec;
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].z, b_f[cb].z, acc[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- slot C ----
      if (s + 2 < 8)
        a_f[(s + 2) & 3] = *reinterpret_cast<const float4*>(&Vc[a_base + ((s + 2) >> 1) * 32 * WN_ROW + 8 * ((s + 2) & 1)]);
      else
        a_f[(s + 2) & 3] = *reinterpret_cast<const float4*>(&Vn[a_base + ((s - 6) >> 1) * 32 * WN_ROW + 8 * ((s - 6) & 1)]);
      __builtin_amdgcn_sched_barrier(0);
      acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].w, b_f[cb].w, acc[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s == 5) {
        W8_TR();
        __syncthreads();
      }
    });
    if (last) {
      col1 = col1 - acc[3];
      aim_epilogue(reg);
      if (!(ovl && has_next)) flush_serial(col0, col1);
      if (!has_next) break;
    }
    if (last) {
      work += (int)gridDim.x;
      reg = nreg;
      nreg.b = nn_b; nreg.oy0 = nn_oy0; nreg.ox0 = nn_ox0; nreg.co0 = nn_co0; nreg.ks = nn_ks;
    }
    has_next = n_has_next; pend = n_pend;
    ch = n_ch; pz ^= 1;
    wcur = wnxt; wnxt = n_wnxt;
  }
}

int sr_wino8_supported(const SrWinoParams& p, bool vout, int nt) {
  // whole float4 channel groups on both sides, 64-channel blocks, at least two slabs per region; every image (input,
  // output / partial output, residual) and the packed weights are addressed through buffer descriptors with 32-bit
  // byte offsets
  if (!vout || nt != 2 || p.ksplit < 1 || ((p.G >> 1) / p.ksplit) < 2 || p.co_blocks * 64 != p.Co_pad) return 0;
  const int64_t lim = (int64_t)1 << 31, px = (int64_t)p.H * p.W;
  const int64_t osp = p.ksplit > 1 ? p.Cout : p.out_sp;
  if (px * p.in_sp * 4 >= lim || px * osp * 4 >= lim || (p.res && px * p.res_sp * 4 >= lim)) return 0;
  if ((int64_t)16 * p.G * 2 * p.Co_pad * 16 >= lim) return 0;
  return 1;
}

int sr_wino8_launch(const SrWinoParams& p, int blocks, hipStream_t stream) {
  const size_t lds = (size_t)W8_LDS_FLOATS * sizeof(float);
  const hipError_t e = hipFuncSetAttribute((const void*)sr_wino8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
  if (e != hipSuccess) return sr_hip_rc(e);
  if (blocks > p.total) blocks = p.total;
  hipLaunchKernelGGL(sr_wino8_kernel, dim3(blocks), dim3(512), lds, stream, p);
  return sr_hip_rc(hipGetLastError());
}
