// sr_wino8.hip -- Winograd F(2x2, 3x3) with ONE 8-wave workgroup per CU and a continuous MFMA stream (gfx950).
//
// Same operator, same packed weights, same arithmetic and the same order of every floating-point operation as
// sr_wino_kernel<2, true, true> (sr_wino.hip) -- results are bit-identical -- but a different schedule, built on two
// measurements (scripts/micro/mfma_overlap.hip, profiles/r02_mfma_valu_overlap.txt):
//   * a VALU instruction never overlaps with an MFMA of its OWN wave (+5 clk each), and under another wave's MFMA the VALU
//     issues at ~1 per 13 clk: with two waves per SIMD the matrix pipe delivers 64 + ~3.5 k clk per MFMA when every MFMA
//     is accompanied by k VALU instructions (k = 2: 86 %, 4: 80 %, 8: 69 %, 16: 54 %).  SALU, s_nop and s_waitcnt are free.
//   * the 4-wave kernel's transform and epilogue phases are exactly such VALU (and latency) stretches: a workgroup issues
//     MFMAs ~38 % of its time on a 4-slab layer and a co-resident workgroup cannot fill the rest.
// So: no phases.  A region's 64 output channels are split over two waves (wave = frequency row u x channel half h: 64
// accumulators per lane), which frees the registers to
//   * park a finished region's column-transformed values (M A: 32 registers) while the accumulators start over: the LDS
//     exchange, row transform, residual add and stores of region r are issued in four pieces between the MFMAs of the
//     first four slabs of region r + 1;
//   * run the input transform of slab g + 1 (raw -> V, double-buffered V) and the global -> LDS staging of slab g + 2
//     between the MFMAs of slab g; the slab stream is continuous across regions (weights are prefetched across the region
//     boundary too),
// and every non-MFMA instruction in the slab body is counted: packed fp32 arithmetic, addresses folded into instruction
// offsets, buffer descriptors instead of branches and selects, region stepping by mixed-radix increments instead of
// divisions.
// LDS: V 2 x 40 KB + raw patches 2 x 14.1 KB + exchange slab 32 KB + 7 KB = 147 KB, one workgroup per CU, two waves per SIMD.
//
// Wave (u, h): transform row u of B^T d B for tiles 16 h .. 16 h + 15, multiplies the 4 frequencies 4 u .. 4 u + 3 of
// all 32 tiles with the weights of channels 32 h .. 32 h + 31.
#include <type_traits>

#include "sr_wino.h"

#define W8_V_FLOATS WN_V_FLOATS            // [16 freq][32 tiles][20]
#define W8_RAW_FLOATS WN_RAW_FLOATS        // [10 * 18 px][20]
#define W8_O_FLOATS (4 * 32 * 64)          // [4 ur][32 tiles][64 co]: one column (bb) of the exchange at a time
#define W8_DUMMY_FLOATS (27 * 64 + 64)     // reach of the column-piece writes from a lane's base (dummy target)
#define W8_LDS_FLOATS (2 * W8_V_FLOATS + 2 * W8_RAW_FLOATS + W8_O_FLOATS + W8_DUMMY_FLOATS)
#ifndef W8_ABL
#define W8_ABL 0   // timing ablations (wrong results): 1 no transform, 2 no staging, 4 no barrier, 8 no epilogue pieces, 16 no weight loads
#endif
#ifndef W8_NB
#define W8_NB 4
#define W8_PD 3
#endif

// a region in mixed-radix digits of its work index: (ks, cb, rx, ry, b)
struct W8Region { int ks, cb, rx, ry, b; };

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
typedef float w8_f4 __attribute__((ext_vector_type(4)));
typedef float w8_f2 __attribute__((ext_vector_type(2)));
#define W8_RSRC_FLAGS 0x00020000      // raw buffer descriptor word 3 on gfx9-family parts
#define W8_OOB 0x7fffffffu            // voffset beyond any num_records: the load returns 0, the store is dropped

// Buffer loads / stores with a uniform descriptor and a 32-bit lane offset: out-of-range lanes (image border, channel
// tail, "no epilogue piece due in this slab") are switched off by their OFFSET, not by a branch -- the slab body below
// stays straight-line code and the compiler's s_waitcnt counts stay exact (a conditional memory op between a prefetch
// and its use makes it wait for everything).
__device__ __forceinline__ w8_f4 w8_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const w8_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return __builtin_bit_cast(w8_f4, v);
}
__device__ __forceinline__ void w8_buf_store(const w8_f4& v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w8_u4, v), r, (int)voff, 0, 0);
}

// Packed fp32 arithmetic, spelled out: left to itself the compiler splits most float2 operations of this kernel into two
// scalar ones, and every VALU instruction here costs matrix-pipe time.
__device__ __forceinline__ w8_f2 w8_pk_add(w8_f2 a, w8_f2 b) {
  w8_f2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ w8_f2 w8_pk_sub(w8_f2 a, w8_f2 b) {
  w8_f2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ w8_f2 w8_pk_fma(w8_f2 a, w8_f2 b, w8_f2 c) {
  w8_f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ w8_f2 w8_pk_mul(w8_f2 a, w8_f2 b) {
  w8_f2 r;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ w8_f4 w8_add4(w8_f4 a, w8_f4 b) {
  const w8_f2 lo = w8_pk_add(a.lo, b.lo), hi = w8_pk_add(a.hi, b.hi);
  return w8_f4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ w8_f4 w8_sub4(w8_f4 a, w8_f4 b) {
  const w8_f2 lo = w8_pk_sub(a.lo, b.lo), hi = w8_pk_sub(a.hi, b.hi);
  return w8_f4{lo.x, lo.y, hi.x, hi.y};
}

// One step of a slab: 4 MFMAs of one frequency / channel group with the three issue slots between them (see the comment
// in front of the slab loop).  A macro, not a lambda: the body is plain code of the loop, so its scalar control stays in
// SGPRs (a lambda captures by reference, and conditional expressions over captured variables defeated the promotion of
// the whole kernel's state out of scratch memory).
#define W8_STEP(S)                                                                                                   \
  do {                                                                                                               \
    constexpr int s = S;                                                                                             \
    constexpr int cb = s % NB, ca = s & 3;                                                                           \
    if (!(s & 1) && first) acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].x, b_f[cb].x, zero16, 0, 0, 0); \
    else acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].x, b_f[cb].x, acc[s >> 1], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0); \
    if constexpr (W8_ABL & 16) {} \
    else if constexpr (s + PD < 8) b_f[(s + PD) % NB] = w_step(wcur, std::integral_constant<int, (s + PD) % 8>{}); \
    else b_f[(s + PD) % NB] = w_step(wnxt, std::integral_constant<int, (s + PD) % 8>{}); \
    __builtin_amdgcn_sched_barrier(0); \
    acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].y, b_f[cb].y, acc[s >> 1], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0); \
    if (s == 0 && !(W8_ABL & 1)) { t_load(rawT, 0, da0, db0); t_load(rawT, 2, da1, db1); } \
    if (s == 1 && !(W8_ABL & 1)) { \
      wv0 = t_pair(da0, db0); wv2 = t_pair(da1, db1); \
      *reinterpret_cast<w8_f4*>(vrow + 0 * 32 * WN_ROW) = w8_sub4(wv0, wv2); \
      t_load(rawT, 1, da0, db0); t_load(rawT, 3, da1, db1); \
    } \
    if (s == 2 && !(W8_ABL & 1)) { \
      wv1 = t_pair(da0, db0); wv3 = t_pair(da1, db1); \
      *reinterpret_cast<w8_f4*>(vrow + 1 * 32 * WN_ROW) = w8_add4(wv1, wv2); \
      *reinterpret_cast<w8_f4*>(vrow + 2 * 32 * WN_ROW) = w8_sub4(wv2, wv1); \
    } \
    if (s == 3) { \
      if (!(W8_ABL & 1)) *reinterpret_cast<w8_f4*>(vrow + 3 * 32 * WN_ROW) = w8_sub4(wv1, wv3); \
      if (!(W8_ABL & (8 | 64))) out_reads(et); \
    } \
    if (s >= 2 && s <= 5 && !(W8_ABL & (8 | 32))) col_write(col0, cbase, std::integral_constant<int, (s >= 2 && s <= 5) ? 4 * (s - 2) : 0>{}); \
    if (s == 4) { \
      if (!(W8_ABL & 8)) { out_sums(et, ey); out_row(eo0, ey[0], rv[0], bv); } \
      if (last) { \
        asm volatile("" ::: "memory"); \
        col0 = acc[0] + acc[1]; \
      } \
    } \
    if (s == 5) { \
      if (!(W8_ABL & 8)) out_row(eo1, ey[1], rv[1], bv); \
      if (!(W8_ABL & 2)) stage_store(stg, rawS); \
    } \
    if (s == 6) { \
      if (!(W8_ABL & 2)) stage_load(stg); \
      if (!last && !(W8_ABL & (8 | 128))) { \
        if (ch == 0) res_loads(std::integral_constant<int, 0>{}, pend, rv, bv); \
        else res_loads(std::integral_constant<int, 1>{}, pend & (ch == 2), rv, bv); \
      } \
      if (piece == 1) { \
        asm volatile("" ::: "memory"); \
        col0 = col1; \
      } \
    } \
    if (s == 7) { \
      stage_advance(); \
      n_ch = ch + 1; \
      if (last) { \
        n_ch = 0; \
        n_has_next = work + 2 * (int)gridDim.x < p.total; \
        advance(nn_ks, nn_cb, nn_rx, nn_ry, nn_b); \
        n_pend = ovl && has_next; \
      } \
      { \
        const bool n_last = n_ch + 1 == chunks; \
        int f_cb = reg_cb, f_ks = reg_ks; \
        if (n_last | last) { f_cb = nreg_cb; f_ks = nreg_ks; } \
        if (n_last & last) { f_cb = nn_cb; f_ks = nn_ks; } \
        const int f_ch = n_last ? 0 : n_ch + 1; \
        n_wnxt = (unsigned)f_cb * 1024u + (unsigned)(2 * (f_ks * chunks + f_ch)) * w_rec; \
      } \
      if (last) { \
        asm volatile("" ::: "memory"); \
        col0 = col0 + acc[2]; \
        col1 = acc[1] - acc[2]; \
      } \
    } \
    __builtin_amdgcn_sched_barrier(0); \
    acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].z, b_f[cb].z, acc[s >> 1], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0); \
    if (s + 2 < 8) \
      a_f[(s + 2) & 3] = *reinterpret_cast<const w8_f4*>(&Vc[((s + 2) >> 1) * 32 * WN_ROW + 8 * ((s + 2) & 1)]); \
    else \
      a_f[(s + 2) & 3] = *reinterpret_cast<const w8_f4*>(&Vn[((s - 6) >> 1) * 32 * WN_ROW + 8 * ((s - 6) & 1)]); \
    __builtin_amdgcn_sched_barrier(0); \
    acc[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_f[ca].w, b_f[cb].w, acc[s >> 1], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0); \
    if (s == 5) { \
      W8_TR(); \
      if (!(W8_ABL & 4)) __syncthreads(); \
      W8_TR(); \
    } \
  } while (0)

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
  // (tiles j, j + 4, j + 8, j + 12 share a 16-lane LDS access group: conflict-free patch reads and V writes, see sr_wino.hip)
#ifdef SR_WINO_TLINEAR
  const int tq = lane & 3, tt = 16 * h + (lane >> 2);
#else
  const int tq = lane & 3, tt = 16 * h + ((((lane >> 2) & 3) << 2) | (lane >> 4));
#endif
  const int t_ra = u == 0 ? 0 : (u == 2 ? 2 : 1), t_rb = u == 0 ? 2 : (u == 1 ? 2 : (u == 2 ? 1 : 3));
  const float t_s1 = u == 1 ? 1.0f : -1.0f;
  const w8_f2 t_sign = {t_s1, t_s1};
  const int t_base = ((2 * (tt >> 3)) * WN_PW + 2 * (tt & 7)) * WN_ROW + 4 * tq;
  const int t_oa = t_base + t_ra * WN_PW * WN_ROW, t_ob = t_base + t_rb * WN_PW * WN_ROW;
  const int t_vo = ((4 * u) * 32 + tt) * WN_ROW + 4 * tq;
  // MFMA role: A fragment of (frequency xi, 8-channel group g) at V[(xi * 32 + i) * 20 + 8 g + 4 kk]
  const int a_base = ((4 * u) * 32 + i) * WN_ROW + 4 * kk;
  // epilogue role: thread = (tile, 4 consecutive channels); column-piece role: accumulator element r of this lane is
  // tile (r & 3) + 8 (r >> 2) + 4 kk, channel 32 h + i
  const int e_tile = tid >> 4, e_cg = tid & 15;
  const int o_lane = (u * 32 + 4 * kk) * 64 + 32 * h + i;

  // ---- regions: mixed-radix digits of the work index, stepped by gridDim.x per visit (no divisions in the loop) ----
  const int KS = p.ksplit, CB = p.co_blocks, RX = p.regions_x, RY = p.regions_y;
  auto decode = [&](unsigned wk) __attribute__((always_inline)) {
    W8Region r;
    r.ks = (int)(wk % (unsigned)KS); wk /= (unsigned)KS;
    r.cb = (int)(wk % (unsigned)CB); wk /= (unsigned)CB;
    r.rx = (int)(wk % (unsigned)RX); wk /= (unsigned)RX;
    r.ry = (int)(wk % (unsigned)RY);
    r.b = (int)(wk / (unsigned)RY);
    return r;
  };
  const W8Region dstep = decode(gridDim.x);
  const int d_ks = dstep.ks, d_cb = dstep.cb, d_rx = dstep.rx, d_ry = dstep.ry, d_b = dstep.b;
  // the region gridDim.x work items further on, in place (plain ints: a struct assigned under a condition inside the
  // slab body ends up in scratch memory, and a descriptor built from it in a waterfall loop)
  auto advance = [&](int& ks, int& cb, int& rx, int& ry, int& b) __attribute__((always_inline)) {
    int c;
    ks += d_ks; c = ks >= KS; ks -= c ? KS : 0;
    cb += d_cb + c; c = cb >= CB; cb -= c ? CB : 0;
    rx += d_rx + c; c = rx >= RX; rx -= c ? RX : 0;
    ry += d_ry + c; c = ry >= RY; ry -= c ? RY : 0;
    b += d_b + c;
  };

  // ---- staging cursor: the slab stream (region, slab) in execution order, three slabs ahead of the MFMAs ----
  int st_work = blockIdx.x, st_ch = 0;
  bool st_valid = st_work < p.total;
  const W8Region st_first = decode((unsigned)st_work);
  int st_ks = st_first.ks, st_rx = st_first.rx, st_ry = st_first.ry, st_b = st_first.b, st_cb = st_first.cb;
  unsigned offs0 = W8_OOB, offs1 = W8_OOB;   // byte offset of this thread's two patch elements inside the image, or OOB
  const unsigned in_bytes = (unsigned)(((int64_t)(p.H * p.W - 1) * p.in_sp + p.Cin) * 4);
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, W8_RSRC_FLAGS);
  int st_c0 = 0;
  // this thread's two patch elements: pixel (py, px) of the 10 x 18 patch, channel quad q (constant per thread)
  const int s_py0 = (tid >> 2) / WN_PW, s_px0 = (tid >> 2) - s_py0 * WN_PW;
  const int s_py1 = ((tid + 512) >> 2) / WN_PW, s_px1 = ((tid + 512) >> 2) - s_py1 * WN_PW;
  const bool s_has1 = tid + 512 < WN_STAGE_ELEMS;
  auto aim = [&]() __attribute__((always_inline)) {   // point the staging loads at region st_reg
    rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (int64_t)st_b * p.in_sb), 0, (int)in_bytes, W8_RSRC_FLAGS);
    st_c0 = st_ks * chunks * 16;
    const int oy0 = st_ry * (2 * WN_TR) - 1, ox0 = st_rx * (2 * WN_TC) - 1;
    const int iy0 = oy0 + s_py0, ix0 = ox0 + s_px0, iy1 = oy0 + s_py1, ix1 = ox0 + s_px1;
    const bool ok0 = st_valid & (iy0 >= 0) & (iy0 < p.H) & (ix0 >= 0) & (ix0 < p.W);
    const bool ok1 = st_valid & s_has1 & (iy1 >= 0) & (iy1 < p.H) & (ix1 >= 0) & (ix1 < p.W);
    offs0 = ok0 ? (unsigned)((iy0 * p.W + ix0) * p.in_sp + 4 * (tid & 3)) * 4u : W8_OOB;
    offs1 = ok1 ? (unsigned)((iy1 * p.W + ix1) * p.in_sp + 4 * (tid & 3)) * 4u : W8_OOB;
  };
  aim();
  auto stage_load = [&](w8_f4 (&stg)[2]) __attribute__((always_inline)) {  // the cursor's slab (zeros past the end of the stream)
    // the channel offset goes through the scalar offset of the instruction; a quad beyond Cin is switched off by its lane offset
    const unsigned c0b = (unsigned)(st_c0 + st_ch * 16) * 4u;
    const bool okc = st_c0 + st_ch * 16 + 4 * (tid & 3) < p.Cin;
    stg[0] = w8_buf_load(rs_in, (W8_ABL & 512) ? (unsigned)(tid & 63) * 16u : (okc ? offs0 : W8_OOB), (W8_ABL & 512) ? 0u : c0b);
    stg[1] = w8_buf_load(rs_in, (W8_ABL & 512) ? (unsigned)(tid & 63) * 16u : (okc ? offs1 : W8_OOB), (W8_ABL & 512) ? 0u : c0b);
  };
  auto stage_advance = [&]() __attribute__((always_inline)) {
    if (st_valid) {
      if (++st_ch == chunks) {
        st_ch = 0;
        st_work += gridDim.x;
        st_valid = st_work < p.total;
        advance(st_ks, st_cb, st_rx, st_ry, st_b);
        aim();
      }
    }
  };
  auto stage_store = [&](const w8_f4 (&stg)[2], float* raw) __attribute__((always_inline)) {
    *reinterpret_cast<w8_f4*>(&raw[(tid >> 2) * WN_ROW + 4 * (tid & 3)]) = stg[0];
    if (s_has1) *reinterpret_cast<w8_f4*>(&raw[((tid + 512) >> 2) * WN_ROW + 4 * (tid & 3)]) = stg[1];
  };

  auto t_load = [&](const float* raw, int c, w8_f4& da, w8_f4& db) __attribute__((always_inline)) {
    da = *reinterpret_cast<const w8_f4*>(&raw[t_oa + c * WN_ROW]);
    db = *reinterpret_cast<const w8_f4*>(&raw[t_ob + c * WN_ROW]);
  };
  auto t_pair = [&](const w8_f4& da, const w8_f4& db) __attribute__((always_inline)) {  // d_ra[c] + t_sign * d_rb[c]
    const w8_f2 lo = w8_pk_fma(t_sign, db.lo, da.lo), hi = w8_pk_fma(t_sign, db.hi, da.hi);
    return w8_f4{lo.x, lo.y, hi.x, hi.y};
  };
  // the whole row transform of one slab in one go (prologue only)
  auto transform_all = [&](const float* raw, float* V) __attribute__((always_inline)) {
    w8_f4 wv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      w8_f4 da, db;
      t_load(raw, c, da, db);
      wv[c] = t_pair(da, db);
    }
    float* vrow = V + t_vo;
    *reinterpret_cast<w8_f4*>(vrow + 0 * 32 * WN_ROW) = w8_sub4(wv[0], wv[2]);
    *reinterpret_cast<w8_f4*>(vrow + 1 * 32 * WN_ROW) = w8_add4(wv[1], wv[2]);
    *reinterpret_cast<w8_f4*>(vrow + 2 * 32 * WN_ROW) = w8_sub4(wv[2], wv[1]);
    *reinterpret_cast<w8_f4*>(vrow + 3 * 32 * WN_ROW) = w8_sub4(wv[1], wv[3]);
  };

  // ---- epilogue of a finished region ----
  // The column half of Y = A^T M A happens in registers (the wave holds a whole frequency row): col0 = (m0 + m1) + m2,
  // col1 = (m1 - m2) - m3 -- 32 registers that wait for their turn in the exchange slab while the accumulators are
  // already collecting the next region.
  __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0, W8_RSRC_FLAGS);
  __amdgpu_buffer_rsrc_t rs_res = rs_out;
  const __amdgpu_buffer_rsrc_t rs_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.wu), 0, (p.bias && !partial) ? p.Cout * 4 : 0, W8_RSRC_FLAGS);
  const unsigned out_sp = partial ? (unsigned)p.Cout : (unsigned)p.out_sp;
  const float e_slope = partial ? -1.0f : p.slope;
  const bool e_fast_leaky = e_slope >= 0.0f && e_slope <= 1.0f;   // LeakyReLU as max(v, slope * v)
  const w8_f2 e_slope2 = {e_slope, e_slope};
  // per region and thread: byte offsets of the tile's four output pixels (row, column bb) in the output and in the
  // residual image (OOB outside the image / past Cout) and the bias quad
  unsigned e_oo[4] = {W8_OOB, W8_OOB, W8_OOB, W8_OOB}, e_ro[4] = {W8_OOB, W8_OOB, W8_OOB, W8_OOB};
  unsigned e_bo = W8_OOB;
  auto aim_epilogue = [&](int r_ks, int r_cb, int r_rx, int r_ry, int r_b) __attribute__((always_inline)) {
    float* outp = partial ? p.part + r_ks * p.part_stride + (int64_t)r_b * p.H * p.W * p.Cout : p.out + (int64_t)r_b * p.out_sb;
    rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)(((int64_t)(p.H * p.W - 1) * out_sp + p.Cout) * 4),
                                               W8_RSRC_FLAGS);
    const bool has_res = p.res && !partial;
    rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? p.res + (int64_t)r_b * p.res_sb : p.wu), 0,
                                               has_res ? (int)(((int64_t)(p.H * p.W - 1) * p.res_sp + p.Cout) * 4) : 0,
                                               W8_RSRC_FLAGS);
    const int oy = r_ry * (2 * WN_TR) + 2 * (e_tile >> 3), ox = r_rx * (2 * WN_TC) + 2 * (e_tile & 7);
    const int cog = r_cb * 64 + 4 * e_cg;
    const bool okc = cog < p.Cout;
    e_bo = (unsigned)cog * 4u;
    w8_static_for<0, 4>([&](auto q_) __attribute__((always_inline)) {
      constexpr int q = decltype(q_)::value;   // q = 2 * row + bb
      const bool ok = okc & (oy + (q >> 1) < p.H) & (ox + (q & 1) < p.W);
      const unsigned pix = (unsigned)((oy + (q >> 1)) * p.W + ox + (q & 1));
      e_oo[q] = ok ? (pix * out_sp + (unsigned)cog) * 4u : W8_OOB;
      e_ro[q] = ok ? (pix * (unsigned)p.res_sp + (unsigned)cog) * 4u : W8_OOB;
    });
  };
  auto res_loads = [&](auto bb_, bool active, w8_f4 (&rv)[2], w8_f4& bv) __attribute__((always_inline)) {
    constexpr int bb = decltype(bb_)::value;
    bv = w8_buf_load(rs_bias, e_bo, 0);   // Cout is a multiple of 4: whole groups or out of range
    rv[0] = w8_buf_load(rs_res, (W8_ABL & 1024) ? (unsigned)(tid & 63) * 16u : (active ? e_ro[bb] : W8_OOB), 0);
    rv[1] = w8_buf_load(rs_res, (W8_ABL & 1024) ? (unsigned)(tid & 63) * 16u : (active ? e_ro[2 + bb] : W8_OOB), 0);
  };
  auto col_write = [&](const f32x16& cv, float* base, auto r0_) __attribute__((always_inline)) {  // 4 elements -> exchange slab
    w8_static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
      constexpr int r = decltype(r0_)::value + decltype(k_)::value;
      base[((r & 3) + 8 * (r >> 2)) * 64] = cv[r];
    });
  };
  auto out_reads = [&](w8_f4 (&t)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int ur = 0; ur < 4; ++ur) t[ur] = *reinterpret_cast<const w8_f4*>(&O[(ur * 32 + e_tile) * 64 + 4 * e_cg]);
  };
  auto out_sums = [&](const w8_f4 (&t)[4], w8_f4 (&ey)[2]) __attribute__((always_inline)) {
    ey[0] = w8_add4(w8_add4(t[0], t[1]), t[2]);
    ey[1] = w8_sub4(w8_sub4(t[1], t[2]), t[3]);
  };
  auto out_row = [&](unsigned off, const w8_f4& y, const w8_f4& r, const w8_f4& bv) __attribute__((always_inline)) {
    w8_f4 v = w8_add4(w8_add4(y, bv), r);
    if (e_fast_leaky) {            // max(v, slope v) = max(v, 0) + slope min(v, 0) for 0 <= slope <= 1 (up to the sign of zero)
      const w8_f2 lo = w8_pk_mul(v.lo, e_slope2), hi = w8_pk_mul(v.hi, e_slope2);
      v = w8_f4{fmaxf(v.x, lo.x), fmaxf(v.y, lo.y), fmaxf(v.z, hi.x), fmaxf(v.w, hi.y)};
    } else if (e_slope >= 0.0f || e_slope < -1.5f) {
      v.x = sr_activate(v.x, e_slope);
      v.y = sr_activate(v.y, e_slope);
      v.z = sr_activate(v.z, e_slope);
      v.w = sr_activate(v.w, e_slope);
    }
    if (W8_ABL & 2048) w8_buf_store(v, rs_out, off == W8_OOB ? W8_OOB : (unsigned)tid * 16u);
    else if (!(W8_ABL & 256)) w8_buf_store(v, rs_out, off);
    else asm volatile("" : : "v"(v));
  };
  auto flush_serial = [&](const f32x16& c0v, const f32x16& c1v) __attribute__((always_inline)) {  // the four pieces back to back
    w8_static_for<0, 2>([&](auto bb_) __attribute__((always_inline)) {
      constexpr int bb = decltype(bb_)::value;
      w8_f4 frv[2], fey[2], fet[4], fbv;
      res_loads(bb_, true, frv, fbv);
      w8_static_for<0, 4>([&](auto q_) __attribute__((always_inline)) {
        if constexpr (bb == 0) col_write(c0v, O + o_lane, std::integral_constant<int, 4 * decltype(q_)::value>{});
        else col_write(c1v, O + o_lane, std::integral_constant<int, 4 * decltype(q_)::value>{});
      });
      __syncthreads();
      out_reads(fet);
      out_sums(fet, fey);
      out_row(e_oo[bb], fey[0], frv[0], fbv);
      out_row(e_oo[2 + bb], fey[1], frv[1], fbv);
      __syncthreads();
    });
  };

  // ---- weights: B fragments stream from L2, PD steps ahead through NB rotating register sets ----
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wu, 0, (int)((int64_t)16 * p.G * 2 * p.Co_pad * 16), W8_RSRC_FLAGS);
  const unsigned w_rec = (unsigned)(2 * p.Co_pad) * 16u;   // bytes per (frequency, 8-channel group) record
  unsigned w_off[8];   // lane offset of step s: frequency 4 u + s / 2, channel group s % 2 (loop-invariant registers)
  {
    const unsigned w_lane = (unsigned)(kk * p.Co_pad + 32 * h + i) * 16u + (unsigned)(4 * u * p.G) * w_rec;
    w8_static_for<0, 8>([&](auto s_) __attribute__((always_inline)) {
      constexpr int s = decltype(s_)::value;
      w_off[s] = w_lane + (unsigned)((s >> 1) * p.G + (s & 1)) * w_rec;
    });
  }
  auto w_base = [&](int r_cb, int r_ks, int ch) __attribute__((always_inline)) {  // byte offset of slab ch of a region (scalar)
    return (unsigned)r_cb * 1024u + (unsigned)(2 * (r_ks * chunks + ch)) * w_rec;
  };
  auto w_step = [&](unsigned base, auto s_) __attribute__((always_inline)) {
    return w8_buf_load(rs_w, w_off[decltype(s_)::value], base);
  };

  // ---- loop-carried state ----
  w8_f4 b_f[NB], a_f[4];
  w8_f4 stg[2];             // the slab that T reads two slabs from now, on its way from global memory
  w8_f4 rv[2], bv;          // residual / bias values of the next output piece
  f32x16 acc[4], col0 = {}, col1 = {};
  int work = blockIdx.x;
  if (work >= p.total) return;
  int reg_ks = st_ks, reg_cb = st_cb, reg_rx = st_rx, reg_ry = st_ry, reg_b = st_b;   // the region being multiplied
  bool has_next = work + (int)gridDim.x < p.total;
  int nreg_ks = reg_ks, nreg_cb = reg_cb, nreg_rx = reg_rx, nreg_ry = reg_ry, nreg_b = reg_b;   // the one after it
  if (has_next) advance(nreg_ks, nreg_cb, nreg_rx, nreg_ry, nreg_b);
  const bool ovl = chunks >= 4 && !(p.debug & 64);
  bool pend = false;        // a finished region's column values wait in col0 / col1 for their epilogue
  int ch = 0, pz = 0;
  unsigned wcur = w_base(reg_cb, reg_ks, 0), wnxt = chunks == 1 ? w_base(nreg_cb, nreg_ks, 0) : w_base(reg_cb, reg_ks, 1);

  if (p.stagger_cu > 0) {   // ablation: start the workgroups at different phases of a region (de-phases their memory bursts)
    if (tid == 0) {
      const long long wait = (long long)((blockIdx.x * 2654435761u) >> 24) * p.stagger_cu / 256;
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while ((long long)(__builtin_amdgcn_s_memtime() - t0) < wait) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
  }
  // ---- prologue: slabs 0 and 1 of the stream into the raw buffers, slab 2 into registers, slab 0 transformed ----
  stage_load(stg); stage_advance(); stage_store(stg, Rbuf);
  stage_load(stg); stage_advance(); stage_store(stg, Rbuf + W8_RAW_FLOATS);
  stage_load(stg); stage_advance();
  __syncthreads();
  transform_all(Rbuf, Vbuf);
  w8_static_for<0, PD>([&](auto s_) __attribute__((always_inline)) { b_f[decltype(s_)::value] = w_step(wcur, s_); });
  __syncthreads();
  a_f[0] = *reinterpret_cast<const w8_f4*>(&Vbuf[a_base]);
  a_f[1] = *reinterpret_cast<const w8_f4*>(&Vbuf[a_base + 8]);

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
  //                      of the staged slab; step 6 (behind the barrier): global loads of the slab after that and the
  //                      residual loads of the next slab's output piece; step 7: control of the next slab; steps 4 / 7
  //                      of a region's last slab: column half of ITS output transform (acc[0..2] are final by then)
  //   slot C (after .z): A fragment of step s + 2 (steps 6, 7: from the next slab's V -- complete behind the barrier)
  // Everything the barrier orders lies on one side of it: T writes / staged patch (before) vs. their readers in the next
  // slab; this slab's V reads (all issued by step 5) vs. the next slab's T writes; an epilogue piece (steps 2-5) vs. the
  // next piece one slab later.
  while (true) {
    const bool first = ch == 0, last = ch + 1 == chunks;
    const int piece = (pend && ch < 4) ? ch : -1;
    const bool outp = (piece == 1) | (piece == 3);
    // output piece of this slab: byte offsets of its two pixels (column bb = piece / 2), out of range when none is due
    const unsigned eo0 = outp ? (piece == 1 ? e_oo[0] + 0u : e_oo[1] + 0u) : W8_OOB, eo1 = outp ? (piece == 1 ? e_oo[2] + 0u : e_oo[3] + 0u) : W8_OOB;
    float* const cbase = ((piece == 0) | (piece == 2)) ? O + o_lane : Dummy + lane;
    const float* Vc = Vbuf + pz * W8_V_FLOATS + a_base;
    const float* Vn = Vbuf + (pz ^ 1) * W8_V_FLOATS + a_base;
    const float* rawT = Rbuf + (pz ^ 1) * W8_RAW_FLOATS;
    float* rawS = Rbuf + pz * W8_RAW_FLOATS;
    float* vrow = Vbuf + (pz ^ 1) * W8_V_FLOATS + t_vo;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    w8_f4 wv0, wv1, wv2, wv3, da0, db0, da1, db1, ey[2], et[4];
    // control of the next slab (filled in at step 7)
    int n_ch = 0;
    bool n_has_next = has_next, n_pend = pend;
    int nn_ks = nreg_ks, nn_cb = nreg_cb, nn_rx = nreg_rx, nn_ry = nreg_ry, nn_b = nreg_b;   // the region after the next one
    unsigned n_wnxt = wnxt;
    W8_TR();
    W8_STEP(0); W8_STEP(1); W8_STEP(2); W8_STEP(3); W8_STEP(4); W8_STEP(5); W8_STEP(6); W8_STEP(7);
    W8_TR();
    if (last) {
      col1 = col1 - acc[3];
      aim_epilogue(reg_ks, reg_cb, reg_rx, reg_ry, reg_b);
      if (!(ovl && has_next)) flush_serial(col0, col1);
      if (!has_next) break;
      work += (int)gridDim.x;
      reg_ks = nreg_ks; reg_cb = nreg_cb; reg_rx = nreg_rx; reg_ry = nreg_ry; reg_b = nreg_b;
      nreg_ks = nn_ks; nreg_cb = nn_cb; nreg_rx = nn_rx; nreg_ry = nn_ry; nreg_b = nn_b;
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
