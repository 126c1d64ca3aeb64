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
#define W8_LDS_FLOATS (2 * W8_V_FLOATS + 2 * W8_RAW_FLOATS + W8_O_FLOATS)
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

__global__ __launch_bounds__(512, 2) void sr_wino8_kernel(SrWinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Vbuf = lds;
  float* const Rbuf = lds + 2 * W8_V_FLOATS;
  float* const O = lds + 2 * W8_V_FLOATS + 2 * W8_RAW_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = wave & 3, h = wave >> 2;
  const int i = lane & 31, kk = lane >> 5;
  const int chunks = (p.G >> 1) / p.ksplit;   // input slabs per region
  const int64_t rec = (int64_t)2 * p.Co_pad;  // float4 units per (frequency, 8-channel group) weight record
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
  // epilogue role: thread = (tile, 4 consecutive channels)
  const int e_tile = tid >> 4, e_cg = tid & 15;

  auto decode = [&](int wk) __attribute__((always_inline)) {
    W8Region r;
    r.ks = wk % p.ksplit; wk /= p.ksplit;
    const int cb = wk % p.co_blocks; wk /= p.co_blocks;
    const int rx = wk % p.regions_x; wk /= p.regions_x;
    const int ry = wk % p.regions_y;
    r.b = wk / p.regions_y;
    r.oy0 = ry * (2 * WN_TR); r.ox0 = rx * (2 * WN_TC); r.co0 = cb * 64;
    return r;
  };

  // ---- staging cursor: the slab stream (region, slab) in execution order, two slabs ahead of the MFMAs ----
  int st_work = blockIdx.x, st_ch = 0;
  bool st_valid = st_work < p.total;
  int offs[2];
  const float* in_b = p.in;
  int st_c0 = 0;
  auto aim = [&](int wk) __attribute__((always_inline)) {
    const W8Region r = decode(wk);
    in_b = p.in + (int64_t)r.b * p.in_sb;
    st_c0 = r.ks * chunks * 16;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + it * 512;
      const int px = e >> 2, q = e & 3;
      const int py = px / WN_PW, pxx = px - py * WN_PW;
      const int iy = r.oy0 - 1 + py, ix = r.ox0 - 1 + pxx;
      const bool ok = (e < WN_STAGE_ELEMS) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
      offs[it] = ok ? (iy * p.W + ix) * p.in_sp + 4 * q : -1;
    }
  };
  if (st_valid) aim(st_work);
  auto stage_load = [&](float4 (&stg)[2]) __attribute__((always_inline)) -> bool {  // loads the cursor's slab and advances the cursor
    const bool valid = st_valid;
    const int c0 = st_c0 + st_ch * 16;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = c0 + 4 * ((tid + it * 512) & 3);
      const bool ok = valid & (offs[it] >= 0) & (c < p.Cin);
      const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(in_b + c0) +
                                                         (unsigned)(ok ? offs[it] : 0) * 4u);
      stg[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (valid) {
      if (++st_ch == chunks) {
        st_ch = 0;
        st_work += gridDim.x;
        st_valid = st_work < p.total;
        if (st_valid) aim(st_work);
      }
    }
    return valid;
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

  // ---- the whole row transform of one slab in one go (prologue only) ----
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

  // ---- epilogue pieces of a finished region (coordinates in `er`, accumulators in accE) ----
  W8Region er = {0, 0, 0, 0, 0};
  // The column half of Y = A^T M A happens in registers when a region's last MFMA has retired (col_transform): the wave
  // holds a whole frequency row, so (M A) costs 4 adds per accumulator element and leaves 2 values -- 32 registers that
  // wait for their turn in the exchange slab while the accumulators are already collecting the next region.
  auto col_transform = [&](const f32x16 (&acc)[4], f32x16& col0, f32x16& col1) __attribute__((always_inline)) {
    col0 = (acc[0] + acc[1]) + acc[2];   // whole-vector arithmetic: 16 independent lanes of registers
    col1 = (acc[1] - acc[2]) - acc[3];
  };
  auto col_piece = [&](const f32x16& colx, auto r0_) __attribute__((always_inline)) {  // 4 elements -> exchange slab
    w8_static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
      constexpr int r = decltype(r0_)::value + decltype(k_)::value;
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * kk;
      O[(u * 32 + tile) * 64 + 32 * h + i] = colx[r];
    });
  };
  auto res_loads = [&](int bb, float4 (&rv)[2], float4& bv) __attribute__((always_inline)) {
    const bool partial = p.ksplit > 1;
    bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && !partial && er.co0 + 4 * e_cg < p.Cout) bv = *reinterpret_cast<const float4*>(p.bias + er.co0 + 4 * e_cg);
    const float* resp = (p.res && !partial) ? p.res + (int64_t)er.b * p.res_sb : nullptr;
    const int cog = er.co0 + 4 * e_cg;
    const int ox = er.ox0 + 2 * (e_tile & 7) + bb;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      const int oy = er.oy0 + 2 * (e_tile >> 3) + row;
      const bool ld = (resp != nullptr) & (cog < p.Cout) & (oy < p.H) & (ox < p.W);
      const float4 v = *reinterpret_cast<const float4*>(
          reinterpret_cast<const char*>(resp ? resp : p.in) +
          (ld ? (unsigned)(oy * p.W + ox) * (unsigned)p.res_sp + (unsigned)cog : 0u) * 4u);
      rv[row] = ld ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // row half + bias + residual + LeakyReLU + store of the two pixels (rows 0 / 1 of the tile, column bb), in three
  // steps so that the exchanged values, the sums and the residuals are never all live at once
  auto out_reads = [&](float4 (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int ur = 0; ur < 4; ++ur) t[ur] = *reinterpret_cast<const float4*>(&O[(ur * 32 + e_tile) * 64 + 4 * e_cg]);
  };
  auto out_sums = [&](const float4 (&t)[4], float4 (&ey)[2]) __attribute__((always_inline)) {
    ey[0] = f4add(f4add(t[0], t[1]), t[2]);
    ey[1] = f4sub(f4sub(t[1], t[2]), t[3]);
  };
  auto out_row = [&](int bb, int row, const float4 (&ey)[2], const float4 (&rv)[2], const float4& bv) __attribute__((always_inline)) {
    const bool partial = p.ksplit > 1;
    float* outp = partial ? p.part + er.ks * p.part_stride + (int64_t)er.b * p.H * p.W * p.Cout
                          : p.out + (int64_t)er.b * p.out_sb;
    const unsigned out_sp = partial ? (unsigned)p.Cout : (unsigned)p.out_sp;
    const float slope = partial ? -1.0f : p.slope;
    const int cog = er.co0 + 4 * e_cg;
    const bool okc = cog < p.Cout;
    const int ox = er.ox0 + 2 * (e_tile & 7) + bb;
    const int oy = er.oy0 + 2 * (e_tile >> 3) + row;
    float4 v = f4add(f4add(ey[row], bv), rv[row]);
    v.x = sr_activate(v.x, slope);
    v.y = sr_activate(v.y, slope);
    v.z = sr_activate(v.z, slope);
    v.w = sr_activate(v.w, slope);
    if (okc & (oy < p.H) & (ox < p.W))
      *reinterpret_cast<float4*>(reinterpret_cast<char*>(outp) + ((unsigned)(oy * p.W + ox) * out_sp + (unsigned)cog) * 4u) = v;
  };
  auto flush_serial = [&](const f32x16& col0, const f32x16& col1) __attribute__((always_inline)) {  // the four pieces back to back (no MFMAs to hide under)
    w8_static_for<0, 2>([&](auto bb_) __attribute__((always_inline)) {
      constexpr int bb = decltype(bb_)::value;
      float4 rv[2], ey[2], et[4], bv;
      res_loads(bb, rv, bv);
      w8_static_for<0, 4>([&](auto q_) __attribute__((always_inline)) {
        if constexpr (bb == 0) col_piece(col0, std::integral_constant<int, 4 * decltype(q_)::value>{});
        else col_piece(col1, std::integral_constant<int, 4 * decltype(q_)::value>{});
      });
      __syncthreads();
      out_reads(et);
      out_sums(et, ey);
      out_row(bb, 0, ey, rv, bv);
      out_row(bb, 1, ey, rv, bv);
      __syncthreads();
    });
  };

  // ---- weights: B fragments stream from L2, PD steps ahead through NB rotating register sets ----
  // uniform base pointer + 32-bit unsigned lane offset in bytes (the scalar-base addressing mode: no 64-bit lane math)
  const char* const wu_c = reinterpret_cast<const char*>(p.wu);
  const unsigned w_lane = (unsigned)(kk * p.Co_pad + 32 * h + i) * 16u;
  auto w_base = [&](const W8Region& r, int ch) __attribute__((always_inline)) {  // (frequency 4 u, group 0) of slab ch of region r
    return wu_c + ((int64_t)r.co0 + (int64_t)(2 * (r.ks * chunks + ch) + 4 * u * p.G) * rec) * 16;
  };
  auto w_step = [&](const char* base, int s) __attribute__((always_inline)) {  // step s: frequency 4 u + s / 2, channel group s % 2
    return *reinterpret_cast<const float4*>(base + (int64_t)((s >> 1) * p.G + (s & 1)) * rec * 16 + w_lane);
  };
  float4 b_f[NB], a_f[2];

  // One slab of MFMAs (this wave: 4 frequencies x 2 channel groups x 4 k-steps = 32) with, in the issue slots between
  // them: weight / A-fragment prefetch, the transform of the next slab (steps 0-3), one epilogue piece of the previous
  // region (steps 4-7) and, at the end, the hand-over of the slab staged at the top.
  //   piece: -1 none, 0 column half bb = 0, 1 outputs bb = 0 (residual loads at the top), 2 column half bb = 1, 3 outputs bb = 1
  //   first: first slab of a region -- the accumulators start from the inline constant 0
  // One body for all cases (uniform run-time branches in the slots): separate instantiations per piece made the
  // register allocator keep the accumulators in different registers per copy and shuffle them at the joins.
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
#define W8_TR_BAR() W8_TR()
  auto chunk = [&](int piece, bool first, f32x16 (&accM)[4], const f32x16& col0, const f32x16& col1, int pz, const char* wcur,
                   const char* wnxt, bool do_t) __attribute__((always_inline)) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* Vc = Vbuf + pz * W8_V_FLOATS;
    float* Vn = Vbuf + (pz ^ 1) * W8_V_FLOATS;
    const float* rawT = Rbuf + (pz ^ 1) * W8_RAW_FLOATS;
    float* rawS = Rbuf + pz * W8_RAW_FLOATS;
    float4 stg[2];
    const bool staged = stage_load(stg);
    a_f[0] = *reinterpret_cast<const float4*>(&Vc[a_base]);
    float4 wv0, wv1, wv2, wv3, da0, db0, da1, db1, rv[2], ey[2], et[4], bv;
    if (piece == 1 || piece == 3) res_loads(piece >> 1, rv, bv);   // consumed five steps further down
    w8_static_for<0, 8>([&](auto s_) __attribute__((always_inline)) {
      constexpr int s = decltype(s_)::value;
      constexpr int cb = s % NB, ca = s & 1;
      if (!(s & 1) && first) accM[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].x, b_f[cb].x, zero16, 0, 0, 0);
      else accM[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].x, b_f[cb].x, accM[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      b_f[(s + PD) % NB] = (s + PD < 8) ? w_step(wcur, s + PD) : w_step(wnxt, s + PD - 8);
      __builtin_amdgcn_sched_barrier(0);
      accM[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].y, b_f[cb].y, accM[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < 8)
        a_f[ca ^ 1] = *reinterpret_cast<const float4*>(&Vc[a_base + ((s + 1) >> 1) * 32 * WN_ROW + 8 * ((s + 1) & 1)]);
      __builtin_amdgcn_sched_barrier(0);
      accM[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].z, b_f[cb].z, accM[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // side work of this step
      if (s < 4) {
        if (do_t) {
          float* vrow = Vn + t_vo;
          // patch reads are issued one step before their use (no LDS latency inside a slot)
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
          if (s == 3) *reinterpret_cast<float4*>(vrow + 3 * 32 * WN_ROW) = f4sub(wv1, wv3);
        }
      } else {
        if (piece == 0) col_piece(col0, std::integral_constant<int, (s >= 4 ? 4 * (s - 4) : 0)>{});
        else if (piece == 2) col_piece(col1, std::integral_constant<int, (s >= 4 ? 4 * (s - 4) : 0)>{});
        else if (piece == 1 || piece == 3) {
          if (s == 4) out_reads(et);
          if (s == 5) { out_sums(et, ey); out_row(piece >> 1, 0, ey, rv, bv); }
          if (s == 6) out_row(piece >> 1, 1, ey, rv, bv);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      accM[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].w, b_f[cb].w, accM[s >> 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (staged) stage_store(stg, rawS);
    W8_TR_BAR();
    __syncthreads();
  };

  // ---- prologue: slabs 0 and 1 of the stream into the raw buffers, slab 0 transformed, first weights on their way ----
  int work = blockIdx.x;
  if (work >= p.total) return;
  W8Region reg = decode(work);
  {
    float4 stg[2];
    if (stage_load(stg)) stage_store(stg, Rbuf);
    if (stage_load(stg)) stage_store(stg, Rbuf + W8_RAW_FLOATS);
    __syncthreads();
    transform_all(Rbuf, Vbuf);
    const char* w0 = w_base(reg, 0);
#pragma unroll
    for (int s = 0; s < PD; ++s) b_f[s] = w_step(w0, s);
    __syncthreads();
  }

  f32x16 acc[4], col0 = {}, col1 = {};
  const bool ovl = chunks >= 4 && !(p.debug & 64);
  bool pend = false;   // a finished region's column values wait in colv for their epilogue
  int pz = 0;          // parity of the global slab index
  while (work < p.total) {
    const int next_work = work + (int)gridDim.x;
    const bool has_next = next_work < p.total;
    const W8Region nreg = has_next ? decode(next_work) : reg;
    for (int ch = 0; ch < chunks; ++ch) {
      const bool last = ch + 1 == chunks;
      const char* wcur = w_base(reg, ch);
      const char* wnxt = last ? w_base(nreg, 0) : w_base(reg, ch + 1);
      const bool do_t = !last || has_next;
      const int piece = (pend && ch < 4) ? ch : -1;
      W8_TR();
      chunk(piece, ch == 0, acc, col0, col1, pz, wcur, wnxt, do_t);
      pz ^= 1;
    }
    col_transform(acc, col0, col1);
    er = reg;
    if (ovl && has_next) pend = true;
    else { flush_serial(col0, col1); pend = false; }
    work = next_work;
    reg = nreg;
  }
}

int sr_wino8_supported(const SrWinoParams& p, bool vout, int nt) {
  // whole float4 channel groups on both sides, 64-channel blocks, at least two slabs per region (the staging cursor
  // runs two slabs ahead and crosses at most one region boundary)
  return vout && nt == 2 && p.ksplit >= 1 && ((p.G >> 1) / p.ksplit) >= 2 && p.co_blocks * 64 == p.Co_pad;
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
