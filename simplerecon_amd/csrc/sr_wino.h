// sr_wino.h -- definitions of the Winograd F(2x2, 3x3) kernel (sr_wino.hip: 4 waves, two workgroups per CU).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "sr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SR_WINO_WAVES
#define SR_WINO_WAVES 2  // waves per SIMD the register allocation must allow (2 = two workgroups per CU)
#endif

#ifndef SR_WINO_NT1_WAVES
#define SR_WINO_NT1_WAVES 2  // the same for the 32-output-channel instantiation (64 accumulator registers, 46 KB of LDS)
#endif

#ifndef SR_WINO_NB
#define SR_WINO_NB 4   // rotating weight-fragment register sets; must divide the 8 steps of a slab
#define SR_WINO_PD 3   // prefetch distance in steps (< NB)
#endif

#ifndef SR_WINO_PIPE
#define SR_WINO_PIPE 1   // 1: next slab stored mid-slab, barrier after step 5, its first transform half under steps 6-7
#endif

#ifndef SR_WINO_PRIME
#define SR_WINO_PRIME 0   // 1: the next region's first weight fragments + first transform are issued in the epilogue's second half
#endif

#define WN_TR 4
#define WN_TC 8
#define WN_PH (2 * WN_TR + 2)  // 10 patch rows
#define WN_PW (2 * WN_TC + 2)  // 18 patch cols
#define WN_ROW 20              // floats per staged pixel / per V row (16 channels + 4 pad)
#define WN_RAW_FLOATS (WN_PH * WN_PW * WN_ROW)
#define WN_O_FLOATS(nt) (8 * 32 * 32 * (nt))
// Raw buffer A shares the first 64 KB with the epilogue slab O (A is dead by the epilogue); raw buffer B lives behind
// them so that the NEXT region's first slab can be staged while the current region finishes (its last MFMA phase and
// its epilogue): 78 KB, 2 workgroups per CU.
#define WN_V_FLOATS(nt) (WN_O_FLOATS(nt) - WN_RAW_FLOATS)   // offset of raw A: the tail of the O area
#define WN_LDS_FLOATS(nt) (WN_O_FLOATS(nt) + WN_RAW_FLOATS)
#define WN_STAGE_ELEMS (WN_PH * WN_PW * 4)  // float4 elements per slab (720)
#define WN_STAGE_PER_THREAD 3

// Phase-ablation switches (env SR_WINO_DEBUG) exist only in -DSR_WINO_ABLATION builds; in production they are compile-time 0,
// which keeps dead branches out of the hot loops (they cost registers: the MLP sweep spilled because of them).
#ifdef SR_WINO_ABLATION
#define SR_WN_DBG(bit) (p.debug & (bit))
#else
#define SR_WN_DBG(bit) 0
#endif

struct SrWinoParams {
  const float* in; int64_t in_sb; int in_sp;
  const float* wu;                                  // packed U: [16][G][2][Co_pad][4]
  const float* bias;
  const float* res; int64_t res_sb; int res_sp;
  float* out; int64_t out_sb; int out_sp;
  int H, W, Cin, Cout, Co_pad, G;                   // stride 1, pad 1: output is H x W
  int regions_x, regions_y, co_blocks, total;
  float slope;
  int vec4;
  int debug;  // ablation bits (env SR_WINO_DEBUG), 0 in production
  // split-K: a work item covers 1/ksplit of the input slabs and stores its raw partial output (no bias / residual /
  // activation) to part + ks * part_stride (dense channels-last [B, H*W, Cout]); sr_wino_reduce_kernel finishes.
  int ksplit; float* part; int64_t part_stride;
  int xcd_order;   // 1: items of a round are dealt to the XCDs in contiguous eighths (SR_WINO_XCD, default 1)
  int stagger;     // the second half of the persistent grid (the second workgroup of every CU) starts this many s_sleep(127) late
#ifdef SR_WINO_TRACE
  unsigned long long* trace;  // [blocks][SR_TR_REGIONS][SR_TR_EVENTS] shader-clock stamps (debug builds only)
#endif
};

#ifdef SR_WINO_TRACE
#define SR_TR_REGIONS 12
#define SR_TR_EVENTS 16
#define SR_TR(ev)                                                                                          \
  do {                                                                                                     \
    if (tid == 0 && tr_region < SR_TR_REGIONS)                                                              \
      p.trace[((size_t)blockIdx.x * SR_TR_REGIONS + tr_region) * SR_TR_EVENTS + (ev)] = clock64();          \
  } while (0)
#else
#define SR_TR(ev) do {} while (0)
#endif

// float4 add / sub as two packed fp32 pairs (v_pk_add_f32): the transforms are pure add / sub work
typedef float wn_f2 __attribute__((ext_vector_type(2)));
#ifdef SR_WINO_NOPK   // ablation: plain fp32 adds instead of v_pk_add_f32 (packed fp32 VALU beside MFMAs)
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
#else
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) {
  const wn_f2 lo = wn_f2{a.x, a.y} - wn_f2{b.x, b.y}, hi = wn_f2{a.z, a.w} - wn_f2{b.z, b.w};
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
  const wn_f2 lo = wn_f2{a.x, a.y} + wn_f2{b.x, b.y}, hi = wn_f2{a.z, a.w} + wn_f2{b.z, b.w};
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
#endif

// ---- buffer addressing (r04) ----------------------------------------------------------------------------------------
// Every VALU instruction the kernel issues outside its MFMA stream costs ~13 clocks while the co-resident workgroup
// keeps the matrix pipe busy, and r03's epilogue + prologue issued ~680 of them per region (64-bit
// address arithmetic, eight predicated residual-load / store branches, a 3 x division-by-18 re-aim of the staging
// loads).  The vector instantiations now address global memory through buffer descriptors: a wave-uniform descriptor
// (image base, byte range) + a 32-bit lane offset + a scalar offset operand.  A lane is switched off by its OFFSET
// (WN_OOB: the load returns 0, the store is dropped) instead of a branch, the per-(tile, pixel) deltas ride in the
// scalar offset operand (SALU work), and an interior region -- every one at 240x320 / 120x160 -- needs no per-lane
// predicate at all.
typedef unsigned int wn_u4 __attribute__((ext_vector_type(4)));
typedef float wn_f4 __attribute__((ext_vector_type(4)));
#define WN_RSRC_FLAGS 0x00020000      // raw buffer descriptor word 3 on gfx9-family parts
#define WN_OOB 0x7fffffffu            // a lane offset beyond any num_records
__device__ __forceinline__ float4 wn_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const wn_f4 v = __builtin_bit_cast(wn_f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
// Stores take their scalar delta through the LANE offset (one v_add), not through the scalar-offset operand: a 16-byte
// buffer store reads its data registers some cycles after it issues, and a VALU write to them in that window corrupts the
// store ("VMEM store of more than 8 bytes" hazard, 2 wait states on gfx940+).  The compiler inserts those wait states only
// when the scalar-offset operand is NOT a register -- with an SGPR offset it assumes there is no hazard, and on gfx950
// there is: the border-region epilogue, which recomputes a lane offset between two stores, wrote the OFFSET into the
// first channel of the previous store (found by tests/test_gpu_image_encoder.py at 480x640, r04).
__device__ __forceinline__ void wn_buf_store(const float4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const wn_f4 t = {v.x, v.y, v.z, v.w};
  // (WN_OOB + soff stays below 2^32 and above every num_records: a switched-off lane stays switched off)
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wn_u4, t), r, (int)(voff + soff), 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wn_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, WN_RSRC_FLAGS);
}

// (`SR_WN_PIN`: an empty asm that makes a value opaque at that point, so that addresses derived from it are formed there
// as register + immediate instead of being hoisted out of the loops, one register -- then one spill -- each)

// ---- split-precision variant (sr_wino_split.hip; fenced experiment, SR_WINO_SPLIT=bf16|f16) ----
int sr_wino_split_mode();   // 0 off, 1 bf16, 2 f16, -1 unknown value (read per call)
int sr_wino_split_pack(const float* weight, int Cout, int Cin, float* packed, int mode, hipStream_t stream);
int sr_wino_split_launch(const SrWinoParams& p, int nt, int blocks, int mode, hipStream_t stream);
