// sr_tsdf.hip -- TSDF fusion of depth maps (SURVEY.md §8f "next" #2; reference tools/tsdf.py:238-320
// TSDFFuser.integrate_depth, :218-236 project_to_camera, :99-111 generate_voxel_coords).  gfx950 only.
//
// The reference materialises, per batch of depth maps, [B,3,N] projected voxel coordinates, a [B,1,N] grid_sample
// result and five more [B,N] temporaries (N = voxels), then runs a boolean-mask gather / scatter per frame; all of
// it in fp16 (OurFuser.fuse_frames feeds .half() tensors, fusers_helper.py:62-68).  Here one kernel streams the
// volume ONCE per batch: a thread owns 8 consecutive voxels along z (one 16-byte load of values and of weights),
// applies the frames of the batch in order in registers and writes the voxels back only if something changed.
// HBM-bound by construction: algorithmic bytes = 4 B per voxel read (+ 4 B per touched voxel written) + the depth maps.
//
// Arithmetic: every torch op on fp16 tensors computes in fp32 and rounds its result to fp16, so each step below is
// an fp32 operation followed by one rounding (H(.)); products of two halves are exact in fp32, so the two matmuls are
// sequential fp32 sums rounded once.  Scalar operands follow torch's rules (probed on the CPU path): tensor +- scalar
// and comparisons round the scalar to fp16, tensor / scalar divides by the scalar as fp32.  Results are bit-identical
// to the reference executed on CPU, including its handling of non-finite sampling coordinates (texel 0).
#include <hip/hip_fp16.h>

#include "sr_common.h"

struct SrTsdfParams {
  __half* values; __half* weights;       // [X,Y,Z], z fastest
  const __half* coords;                  // [3,X,Y,Z] explicit voxel coordinates, or null: origin + index * voxel_size
  int X, Y, Z;
  float ox, oy, oz, voxel_size;
  const __half* depth;                   // [B,H,W]
  const uint8_t* mask;                   // [B,H,W] bool or null
  const __half* K; const __half* T;      // [B,16] row-major 4x4
  int B, H, W;
  float min_depth, max_depth, depth_range, trunc, maxW;  // python scalars as fp32
  int64_t groups;                        // X*Y*Z / 8
};

// fp32 operations must stay separate operations (torch rounds after each one): no FMA contraction in this file
#pragma clang fp contract(off)

__device__ __forceinline__ float H(float v) { return __half2float(__float2half_rn(v)); }
// torch.clamp / np.clip semantics (NaN propagates; fminf / fmaxf would drop it)
__device__ __forceinline__ float sr_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void sr_tsdf_integrate_kernel(SrTsdfParams p) {
  extern __shared__ float Ps[];  // [B][12]: rows 0..2 of H(K @ T), as fp32
  for (int e = threadIdx.x; e < p.B * 12; e += blockDim.x) {
    const int b = e / 12, r = (e % 12) / 4, c = e % 4;
    const __half* Kb = p.K + 16 * b;
    const __half* Tb = p.T + 16 * b;
    float acc = __half2float(Kb[4 * r + 0]) * __half2float(Tb[0 + c]);
    acc = acc + __half2float(Kb[4 * r + 1]) * __half2float(Tb[4 + c]);
    acc = acc + __half2float(Kb[4 * r + 2]) * __half2float(Tb[8 + c]);
    acc = acc + __half2float(Kb[4 * r + 3]) * __half2float(Tb[12 + c]);
    Ps[e] = H(acc);
  }
  __syncthreads();

  const float Wh = H((float)p.W), Hh = H((float)p.H);  // img_size is an fp16 tensor (tsdf.py:255)
  const float min_h = H(p.min_depth), max_h = H(p.max_depth), ntrunc_h = H(-p.trunc);
  const int zg = p.Z >> 3;  // 8-voxel groups along z

  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < p.groups; g += (int64_t)gridDim.x * blockDim.x) {
    const int gz = (int)(g % zg);
    const int64_t col = g / zg;
    const int iy = (int)(col % p.Y), ix = (int)(col / p.Y);
    const int64_t base = g * 8;
    union U { uint4 q; __half h[8]; };
    U uv, uw;
    uv.q = *reinterpret_cast<const uint4*>(p.values + base);
    uw.q = *reinterpret_cast<const uint4*>(p.weights + base);
    float vx, vy;
    if (!p.coords) {
      vx = H(p.ox + (float)ix * p.voxel_size);  // generate_voxel_coords (:108) in fp32, then .half() (:89)
      vy = H(p.oy + (float)iy * p.voxel_size);
    }
    bool changed = false;
#pragma unroll 1
    for (int b = 0; b < p.B; ++b) {
      const float* P = Ps + 12 * b;
      const __half* dimg = p.depth + (int64_t)b * p.H * p.W;
      const uint8_t* mimg = p.mask ? p.mask + (int64_t)b * p.H * p.W : nullptr;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        float x, y, z;
        if (p.coords) {
          const int64_t n = (int64_t)p.X * p.Y * p.Z;
          x = __half2float(p.coords[base + v]);
          y = __half2float(p.coords[n + base + v]);
          z = __half2float(p.coords[2 * n + base + v]);
        } else {
          x = vx; y = vy;
          z = H(p.oz + (float)(8 * gz + v) * p.voxel_size);
        }
        // cam_points = H(P @ [x y z 1]) (:232)
        const float vz = H(((P[8] * x + P[9] * y) + P[10] * z) + P[11]);
        if (!(vz > 0.0f) || !(vz < max_h)) continue;  // valid_points needs 0 < vox_depth < max_depth (:296-298)
        const float c0 = H(((P[0] * x + P[1] * y) + P[2] * z) + P[3]);
        const float c1 = H(((P[4] * x + P[5] * y) + P[6] * z) + P[7]);
        const float px = H(c0 / vz), py = H(c1 / vz);  // (:233)
        // 2 * pix / img_size - 1 (:268), then grid_sample's unnormalise ((g + 1) * size - 1) / 2 in Half (:275-279)
        const float gx = H(H(H(2.0f * px) / Wh) - 1.0f), gy = H(H(H(2.0f * py) / Hh) - 1.0f);
        const float fx = H(H(H(H(gx + 1.0f) * Wh) - 1.0f) / 2.0f), fy = H(H(H(H(gy + 1.0f) * Hh) - 1.0f) / 2.0f);
        // nearest: nearbyint; a non-finite coordinate addresses texel 0 on the reference's (CPU) path
        const float xn = isfinite(fx) ? rintf(fx) : 0.0f, yn = isfinite(fy) ? rintf(fy) : 0.0f;
        float sd = 0.0f;  // padding_mode="zeros"
        if (xn >= 0.0f && xn < (float)p.W && yn >= 0.0f && yn < (float)p.H) {
          const int o = (int)yn * p.W + (int)xn;
          sd = (mimg && !mimg[o]) ? -1.0f : __half2float(dimg[o]);  // depth[~mask] = -1 (:270-272)
        }
        if (!(sd > 0.0f)) continue;
        // confidence (:282-284)
        const float cf0 = sr_clampf(H(1.0f - H(H(sd - min_h) / p.depth_range)), 0.0f, 1.0f);
        const float conf = H(cf0 * cf0);
        const float dist = H(sd - vz);                                  // (:287)
        const float tv = sr_clampf(H(dist / p.trunc), -1.0f, 1.0f);     // (:288)
        if (!(dist > ntrunc_h) || !(conf > 0.0f)) continue;             // (:291-293)
        const float ov = __half2float(uv.h[v]), ow = __half2float(uw.h[v]);
        const float rate = conf < ow ? 2.0f : 5.0f;                     // (:312)
        const float nw = H(H(conf * rate) / p.maxW);                    // (:315)
        const float tw = H(ow + nw);                                    // (:316)
        uv.h[v] = __float2half_rn(H(H(ov * ow) + H(tv * nw)) / tw);     // (:319)
        uw.h[v] = __float2half_rn(tw > 1.0f ? 1.0f : tw);               // (:320)
        changed = true;
      }
    }
    if (changed) {
      *reinterpret_cast<uint4*>(p.values + base) = uv.q;
      *reinterpret_cast<uint4*>(p.weights + base) = uw.q;
    }
  }
}

extern "C" int sr_tsdf_integrate_fwd(void* tsdf_values, void* tsdf_weights, const void* voxel_coords, int X, int Y, int Z,
                                     float origin_x, float origin_y, float origin_z, float voxel_size,
                                     const void* depth, const uint8_t* depth_mask, const void* K, const void* T,
                                     int B, int H, int W, float min_depth, float max_depth, float depth_range,
                                     float truncation, float maxW, void* stream_) {
  if (X <= 0 || Y <= 0 || Z <= 0 || B < 0 || H <= 0 || W <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!tsdf_values || !tsdf_weights || !depth || !K || !T) return SR_ERR_INVALID_ARGUMENT;
  if ((Z % 8) || ((uintptr_t)tsdf_values & 15) || ((uintptr_t)tsdf_weights & 15)) return SR_ERR_UNSUPPORTED;
  if (B > 1024) return SR_ERR_UNSUPPORTED;
  SrTsdfParams p;
  p.values = (__half*)tsdf_values; p.weights = (__half*)tsdf_weights; p.coords = (const __half*)voxel_coords;
  p.X = X; p.Y = Y; p.Z = Z;
  p.ox = origin_x; p.oy = origin_y; p.oz = origin_z; p.voxel_size = voxel_size;
  p.depth = (const __half*)depth; p.mask = depth_mask; p.K = (const __half*)K; p.T = (const __half*)T;
  p.B = B; p.H = H; p.W = W;
  p.min_depth = min_depth; p.max_depth = max_depth;
  p.depth_range = depth_range;
  p.trunc = truncation; p.maxW = maxW;
  p.groups = (int64_t)X * Y * Z / 8;
  int64_t blocks = (p.groups + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(sr_tsdf_integrate_kernel, dim3((unsigned)blocks), dim3(256), (size_t)B * 12 * sizeof(float),
                     (hipStream_t)stream_, p);
  return sr_hip_rc(hipGetLastError());
}
