// sr_tsdf.hip -- TSDF fusion of depth maps (SURVEY.md §8f "next" #2; reference tools/tsdf.py:238-320
// TSDFFuser.integrate_depth, :218-236 project_to_camera, :99-111 generate_voxel_coords).  gfx950 only.
//
// The reference materialises, per batch of depth maps, [B,3,N] projected voxel coordinates, a [B,1,N] grid_sample
// result and five more [B,N] temporaries (N = voxels), then runs a boolean-mask gather / scatter per frame; all of
// it in fp16 (OurFuser.fuse_frames feeds .half() tensors, fusers_helper.py:62-68).  Here one kernel streams the
// volume ONCE per batch: a thread owns 8 consecutive voxels along z (one 16-byte load of values and of weights),
// applies the frames of the batch in order in registers and writes the voxels back only if something changed.
// Bricks that no frame of the batch can touch are rejected on their corners without reading them (see the kernel).
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
  int tile_x, tile_y, tile_z;            // workgroup tile: multiples of the 8 x 8 x 32 sub-brick
};

// fp32 operations must stay separate operations (torch rounds after each one): no FMA contraction in this file
#pragma clang fp contract(off)

__device__ __forceinline__ float H(float v) { return __half2float(__float2half_rn(v)); }
// torch.clamp / np.clip semantics (NaN propagates; fminf / fmaxf would drop it)
__device__ __forceinline__ float sr_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// A workgroup owns a tile of 16 x 16 x 64 voxels.  Before touching memory it tests, per frame, whether the tile can
// contain a voxel the frame updates at all -- conservatively, on the tile's 8 corner voxels: the voxel depth is affine
// in the coordinates and, for corners in front of the camera, the projected tile lies inside the convex hull of the
// projected corners.  Margins cover the fp16 roundings of the exact path (relative 2^-11 per rounding: 0.2 % + 0.01 on
// depth, 1 % + 2 px on pixel coordinates); tiles near the camera plane (where fp16 pixel coordinates may overflow -- the
// reference then samples texel 0) are never culled by pixel position.  Inside a surviving tile the 8 sub-bricks of
// 8 x 8 x 32 voxels are walked with wave = one 8-voxel z-group of 64 neighbouring columns, so the same test on a
// thread's own segment (its two end voxels) is nearly wave-uniform; only then the exact per-voxel path runs.  The
// volume is read (and written) only by threads that reach an update, so a batch costs about one pass over the
// voxels inside the view frusta plus a cheap rejection pass over the rest.
// (16 x 16 x 64 for large volumes, a single sub-brick when there are too few tiles to fill the chip).

__global__ __launch_bounds__(256) void sr_tsdf_integrate_kernel(SrTsdfParams p) {
  extern __shared__ float Ps[];  // [B][12]: rows 0..2 of H(K @ T), as fp32; then B frame flags (bytes) + a counter
  unsigned char* flags = reinterpret_cast<unsigned char*>(Ps + 12 * p.B);
  __shared__ int kept_frames;
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

  const float Wh = H((float)p.W), Hh = H((float)p.H);  // img_size is an fp16 tensor (tsdf.py:255)
  const float min_h = H(p.min_depth), max_h = H(p.max_depth), ntrunc_h = H(-p.trunc);
  const int zg = p.Z >> 3;  // 8-voxel groups along z
  const int ntx = (p.X + p.tile_x - 1) / p.tile_x, nty = (p.Y + p.tile_y - 1) / p.tile_y;
  const int ntz = (p.Z + p.tile_z - 1) / p.tile_z;
  const int sbx = p.tile_x >> 3, sby = p.tile_y >> 3, sbz = p.tile_z >> 5, nsb = sbx * sby * sbz;
  const int ntiles = ntx * nty * ntz;
  const int64_t nvox = (int64_t)p.X * p.Y * p.Z;
  const int tcx = (threadIdx.x >> 3) & 7, tcy = threadIdx.x & 7, tzg = threadIdx.x >> 6;  // column in the sub-brick, z-group

  // world coordinate of a voxel along an axis (fp16 value as fp32), explicit or generated
  auto coord = [&](int axis, int ix, int iy, int iz) -> float {
    if (p.coords) return __half2float(p.coords[axis * nvox + ((int64_t)ix * p.Y + iy) * p.Z + iz]);
    const float o = axis == 0 ? p.ox : (axis == 1 ? p.oy : p.oz);
    const int i = axis == 0 ? ix : (axis == 1 ? iy : iz);
    return H(o + (float)i * p.voxel_size);  // generate_voxel_coords (:108) in fp32, then .half() (:89)
  };
  // conservative "cannot be updated by this frame" test for the box / segment spanned by unrounded corner projections
  auto cullable = [&](float zmin, float zmax, float pxmin, float pxmax, float pymin, float pymax) -> bool {
    const float mz = 0.01f + 2e-3f * fmaxf(fabsf(zmin), fabsf(zmax));
    if (zmax < -mz || zmin > max_h + mz) return true;  // valid_points needs 0 < vox_depth < max_depth
    if (zmin > 0.25f) {  // everything well in front of the camera: the pixel hull is meaningful
      const float ax = fmaxf(fabsf(pxmin), fabsf(pxmax)), ay = fmaxf(fabsf(pymin), fabsf(pymax));
      if (ax < 30000.0f && ay < 30000.0f) {  // no fp16 overflow anywhere inside
        const float mx = 2.0f + 0.01f * ax, my = 2.0f + 0.01f * ay;
        if (pxmax < -mx || pxmin > (float)p.W + mx || pymax < -my || pymin > (float)p.H + my) return true;
      }
    }
    return false;
  };

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tzi = tile % ntz, tyi = (tile / ntz) % nty, txi = tile / (ntz * nty);
    const int x0 = txi * p.tile_x, y0 = tyi * p.tile_y, z0 = tzi * p.tile_z;
    const int x1 = min(x0 + p.tile_x, p.X) - 1, y1 = min(y0 + p.tile_y, p.Y) - 1, z1 = min(z0 + p.tile_z, p.Z) - 1;
    __syncthreads();  // P ready (first tile) / previous tile done with the flags
    if (threadIdx.x == 0) kept_frames = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
      bool keep = true;
      if (!p.coords) {  // explicit coordinates need not be monotonic in the index: never culled
        const float* P = Ps + 12 * b;
        float zmin = INFINITY, zmax = -INFINITY, pxmin = INFINITY, pxmax = -INFINITY, pymin = INFINITY, pymax = -INFINITY;
        for (int c = 0; c < 8; ++c) {
          const float x = coord(0, (c & 1) ? x1 : x0, 0, 0), y = coord(1, 0, (c & 2) ? y1 : y0, 0);
          const float z = coord(2, 0, 0, (c & 4) ? z1 : z0);
          const float vz = ((P[8] * x + P[9] * y) + P[10] * z) + P[11];
          const float rz = __builtin_amdgcn_rcpf(vz);  // 1-ulp reciprocal: this test only needs ~1 % accuracy
          const float px = (((P[0] * x + P[1] * y) + P[2] * z) + P[3]) * rz;
          const float py = (((P[4] * x + P[5] * y) + P[6] * z) + P[7]) * rz;
          zmin = fminf(zmin, vz); zmax = fmaxf(zmax, vz);
          pxmin = fminf(pxmin, px); pxmax = fmaxf(pxmax, px); pymin = fminf(pymin, py); pymax = fmaxf(pymax, py);
        }
        keep = !cullable(zmin, zmax, pxmin, pxmax, pymin, pymax);
      }
      flags[b] = keep ? 1 : 0;
      if (keep) atomicAdd(&kept_frames, 1);
    }
    __syncthreads();
    if (kept_frames == 0) continue;

#pragma unroll 1
    for (int sb = 0; sb < nsb; ++sb) {  // sub-bricks of 8 x 8 x 32 voxels
      const int sz = sb % sbz, sy = (sb / sbz) % sby, sx = sb / (sbz * sby);
      const int ix = x0 + 8 * sx + tcx, iy = y0 + 8 * sy + tcy, gz = (z0 >> 3) + 4 * sz + tzg;
      if (ix >= p.X || iy >= p.Y || gz >= zg) continue;
      const int64_t base = (((int64_t)ix * p.Y + iy) * p.Z) + 8 * gz;
      union U { uint4 q; __half h[8]; };
      U uv, uw;
      bool loaded = false, changed = false;
      const float vx = coord(0, ix, iy, 8 * gz), vy = coord(1, ix, iy, 8 * gz);
#pragma unroll 1
      for (int b = 0; b < p.B; ++b) {
        if (!flags[b]) continue;
        const float* P = Ps + 12 * b;
        if (!p.coords) {  // the same conservative test on this thread's 8-voxel segment (its two end voxels)
          const float za = coord(2, ix, iy, 8 * gz), zb = coord(2, ix, iy, 8 * gz + 7);
          const float xy2 = P[8] * vx + P[9] * vy, xy0 = P[0] * vx + P[1] * vy, xy1 = P[4] * vx + P[5] * vy;
          const float va = (xy2 + P[10] * za) + P[11], vb = (xy2 + P[10] * zb) + P[11];
          const float ra = __builtin_amdgcn_rcpf(va), rb = __builtin_amdgcn_rcpf(vb);
          const float pxa = ((xy0 + P[2] * za) + P[3]) * ra, pxb = ((xy0 + P[2] * zb) + P[3]) * rb;
          const float pya = ((xy1 + P[6] * za) + P[7]) * ra, pyb = ((xy1 + P[6] * zb) + P[7]) * rb;
          if (cullable(fminf(va, vb), fmaxf(va, vb), fminf(pxa, pxb), fmaxf(pxa, pxb), fminf(pya, pyb), fmaxf(pya, pyb)))
            continue;
        }
        const __half* dimg = p.depth + (int64_t)b * p.H * p.W;
        const uint8_t* mimg = p.mask ? p.mask + (int64_t)b * p.H * p.W : nullptr;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          float x = vx, y = vy;
          if (p.coords) { x = coord(0, ix, iy, 8 * gz + v); y = coord(1, ix, iy, 8 * gz + v); }
          const float z = coord(2, ix, iy, 8 * gz + v);
          // cam_points = H(P @ [x y z 1]) (:232)
          const float vz = H(((P[8] * x + P[9] * y) + P[10] * z) + P[11]);
          if (!(vz > 0.0f) || !(vz < max_h)) continue;  // valid_points needs 0 < vox_depth < max_depth (:296-298)
          const float c0 = H(((P[0] * x + P[1] * y) + P[2] * z) + P[3]);
          const float c1 = H(((P[4] * x + P[5] * y) + P[6] * z) + P[7]);
          const float px = H(c0 / vz), py = H(c1 / vz);  // (:233)
          // a finite pixel coordinate well outside the image samples the zero padding (the chain below moves it by
          // < 1 % + 1 px); non-finite / huge ones must take the exact path (they address texel 0 on the reference's path)
          if ((fabsf(px) < 30000.0f && (px < -2.0f - 0.01f * fabsf(px) || px > (float)p.W + 2.0f + 0.01f * fabsf(px))) ||
              (fabsf(py) < 30000.0f && (py < -2.0f - 0.01f * fabsf(py) || py > (float)p.H + 2.0f + 0.01f * fabsf(py))))
            if (fabsf(px) < 30000.0f && fabsf(py) < 30000.0f) continue;
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
          if (!loaded) {
            uv.q = *reinterpret_cast<const uint4*>(p.values + base);
            uw.q = *reinterpret_cast<const uint4*>(p.weights + base);
            loaded = true;
          }
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
  // tile = 16 x 16 x 64 voxels when that still gives every CU several tiles, else one 8 x 8 x 32 sub-brick
  p.tile_x = 16; p.tile_y = 16; p.tile_z = 64;
  auto ntiles = [&]() {
    return (int64_t)((X + p.tile_x - 1) / p.tile_x) * ((Y + p.tile_y - 1) / p.tile_y) * ((Z + p.tile_z - 1) / p.tile_z);
  };
  if (ntiles() < 8192) { p.tile_x = 8; p.tile_y = 8; p.tile_z = 32; }
  int64_t blocks = ntiles();
  if (blocks > 0x7fffffff) return SR_ERR_UNSUPPORTED;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sr_tsdf_integrate_kernel, dim3((unsigned)blocks), dim3(256),
                     (size_t)B * 12 * sizeof(float) + (size_t)((B + 15) / 16 * 16), (hipStream_t)stream_, p);
  return sr_hip_rc(hipGetLastError());
}
