// sr_geometry.hip -- standalone forms of the reference's small geometry helpers (utils/geometry_utils.py) for callers
// outside the fused sweeps.  Inside the sweeps the same arithmetic is fused (sr_common.h: sr_project_sample_xy; the
// operation order below is the one used there, FP contraction off, so results are bit-identical to what the sweeps
// compute internally and to the fp32 CPU oracle).
#include "sr_common.h"

// BackprojectDepth.forward (geometry_utils.py:51-59): X = (depth * invK[:3,:3] (x+.5, y+.5, 1), 1)
__global__ __launch_bounds__(256) void sr_backproject_kernel(const float* __restrict__ depth, const float* __restrict__ invK,
                                                            float* __restrict__ out, int h, int w) {
#pragma clang fp contract(off)
  const int N = h * w, pix = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (pix >= N) return;
  const int y = pix / w, x = pix - y * w;
  const float* iK = invK + 16 * (size_t)b;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;  // geometry_utils.py:34-44
  const float r0 = iK[0] * px + iK[1] * py + iK[2], r1 = iK[4] * px + iK[5] * py + iK[6], r2 = iK[8] * px + iK[9] * py + iK[10];
  const float d = depth[(size_t)b * N + pix];
  float* o = out + (size_t)b * 4 * N + pix;
  o[0] = d * r0; o[N] = d * r1; o[2 * (size_t)N] = d * r2; o[3 * (size_t)N] = 1.0f;
}

// Project3D.forward (geometry_utils.py:72-89): q = (K T)[:3] X;  z' = q_z + eps;  (q_x s, q_y s, z'), s = |q_z| > eps ? 1/z' : 1
__global__ __launch_bounds__(256) void sr_project3d_kernel(const float* __restrict__ pts, const float* __restrict__ Km,
                                                          const float* __restrict__ T, float* __restrict__ out, int N,
                                                          float eps) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= N) return;
  const float* Kb = Km + 16 * (size_t)b;
  const float* Tb = T + 16 * (size_t)b;
  const float* X = pts + (size_t)b * 4 * N + i;
  const float X0 = X[0], X1 = X[N], X2 = X[2 * (size_t)N], X3 = X[3 * (size_t)N];
  float q[3];
  for (int r = 0; r < 3; ++r) {
    float P[4];
    for (int c = 0; c < 4; ++c) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += Kb[r * 4 + k] * Tb[k * 4 + c];  // P = K @ T, as sr_geom_kernel
      P[c] = s;
    }
    q[r] = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[3] * X3;
  }
  const float zp = q[2] + eps;
  const float sc = (fabsf(q[2]) > eps) ? 1.0f / zp : 1.0f;
  float* o = out + (size_t)b * 3 * N + i;
  o[0] = q[0] * sc; o[N] = q[1] * sc; o[2 * (size_t)N] = zp;
}

// Adjoints of the two modules with respect to what the reference's losses differentiate: the depth map (BackprojectDepth)
// and the points (Project3D); intrinsics and poses are data.
//   d_depth[n] = sum_{c<3} g[c,n] * (invK[:3,:3] pix_n)[c]
__global__ __launch_bounds__(256) void sr_backproject_bwd_kernel(const float* __restrict__ g, const float* __restrict__ invK,
                                                                float* __restrict__ d_depth, int h, int w) {
  const int N = h * w, pix = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (pix >= N) return;
  const int y = pix / w, x = pix - y * w;
  const float* iK = invK + 16 * (size_t)b;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  const float r0 = iK[0] * px + iK[1] * py + iK[2], r1 = iK[4] * px + iK[5] * py + iK[6], r2 = iK[8] * px + iK[9] * py + iK[10];
  const float* gb = g + (size_t)b * 4 * N + pix;
  d_depth[(size_t)b * N + pix] = gb[0] * r0 + gb[N] * r1 + gb[2 * (size_t)N] * r2;
}

//   out = (q_x s, q_y s, q_z + eps), s = 1 / (q_z + eps) where |q_z| > eps, else 1 (a constant):
//   d_q = (g_x s, g_y s, g_z - [|q_z| > eps] (g_x q_x + g_y q_y) s^2);  d_X[c] = sum_r P[r][c] d_q[r]
__global__ __launch_bounds__(256) void sr_project3d_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pts,
                                                              const float* __restrict__ Km, const float* __restrict__ T,
                                                              float* __restrict__ d_pts, int N, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= N) return;
  const float* Kb = Km + 16 * (size_t)b;
  const float* Tb = T + 16 * (size_t)b;
  const float* X = pts + (size_t)b * 4 * N + i;
  const float X0 = X[0], X1 = X[N], X2 = X[2 * (size_t)N], X3 = X[3 * (size_t)N];
  float P[3][4], q[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 4; ++c) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += Kb[r * 4 + k] * Tb[k * 4 + c];
      P[r][c] = s;
    }
    q[r] = P[r][0] * X0 + P[r][1] * X1 + P[r][2] * X2 + P[r][3] * X3;
  }
  const bool far = fabsf(q[2]) > eps;
  const float sc = far ? 1.0f / (q[2] + eps) : 1.0f;
  const float* gb = g + (size_t)b * 3 * N + i;
  const float g0 = gb[0], g1 = gb[N], g2 = gb[2 * (size_t)N];
  const float dq0 = g0 * sc, dq1 = g1 * sc, dq2 = g2 - (far ? (g0 * q[0] + g1 * q[1]) * sc * sc : 0.0f);
  float* o = d_pts + (size_t)b * 4 * N + i;
  for (int c = 0; c < 4; ++c) o[(size_t)c * N] = P[0][c] * dq0 + P[1][c] * dq1 + P[2][c] * dq2;
}

// pose_distance (geometry_utils.py:178-191): (sqrt(t_m^2 + R_m^2), R_m, t_m) -- the values sr_geom_kernel feeds the MLP
__global__ void sr_pose_distance_kernel(const float* __restrict__ T, float* __restrict__ out, int n) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* Tc = T + 16 * (size_t)i;
  const float t0 = Tc[3], t1 = Tc[7], t2 = Tc[11];
  const float tr = (Tc[0] + Tc[5]) + Tc[10];
  const float rm = sqrtf(2.0f * (1.0f - fminf(3.0f, tr) / 3.0f));
  const float tm = sqrtf((t0 * t0 + t1 * t1) + t2 * t2);
  out[3 * (size_t)i + 0] = sqrtf(tm * tm + rm * rm);
  out[3 * (size_t)i + 1] = rm;
  out[3 * (size_t)i + 2] = tm;
}

// get_camera_rays (geometry_utils.py:143-175): normalize(points - centre) (world frame) or normalize(T[:3,:4] (points, 1))
// (camera frame); F.normalize: x / max(|x|, 1e-12)
__global__ __launch_bounds__(256) void sr_camera_rays_kernel(const float* __restrict__ pts, const float* __restrict__ T,
                                                            float* __restrict__ out, int N, int in_camera_frame) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= N) return;
  const float* Tb = T + 16 * (size_t)b;
  const float* X = pts + (size_t)b * 3 * N + i;
  const float X0 = X[0], X1 = X[N], X2 = X[2 * (size_t)N];
  float r0, r1, r2;
  if (in_camera_frame) {
    r0 = ((Tb[0] * X0 + Tb[1] * X1) + Tb[2] * X2) + Tb[3];
    r1 = ((Tb[4] * X0 + Tb[5] * X1) + Tb[6] * X2) + Tb[7];
    r2 = ((Tb[8] * X0 + Tb[9] * X1) + Tb[10] * X2) + Tb[11];
  } else {
    r0 = X0 - Tb[3]; r1 = X1 - Tb[7]; r2 = X2 - Tb[11];
  }
  const float nrm = fmaxf(sqrtf((r0 * r0 + r1 * r1) + r2 * r2), 1e-12f);
  float* o = out + (size_t)b * 3 * N + i;
  o[0] = r0 / nrm; o[N] = r1 / nrm; o[2 * (size_t)N] = r2 / nrm;
}

extern "C" int sr_backproject_fwd(const float* depth, const float* invK, float* out_points, int B, int h, int w,
                                  void* stream) {
  if (B < 0 || h <= 0 || w <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!depth || !invK || !out_points) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_backproject_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, depth, invK,
                     out_points, h, w);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_project3d_fwd(const float* points, const float* K, const float* T, float* out, int B, int N, float eps,
                                void* stream) {
  if (B < 0 || N <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!points || !K || !T || !out) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_project3d_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, points, K, T, out, N,
                     eps);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_backproject_bwd(const float* grad_points, const float* invK, float* grad_depth, int B, int h, int w,
                                  void* stream) {
  if (B < 0 || h <= 0 || w <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!grad_points || !invK || !grad_depth) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_backproject_bwd_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, grad_points,
                     invK, grad_depth, h, w);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_project3d_bwd(const float* grad_out, const float* points, const float* K, const float* T, float* grad_points,
                                int B, int N, float eps, void* stream) {
  if (B < 0 || N <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!grad_out || !points || !K || !T || !grad_points) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_project3d_bwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, grad_out, points, K,
                     T, grad_points, N, eps);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_pose_distance_fwd(const float* T, float* out, int n, void* stream) {
  if (n < 0) return SR_ERR_INVALID_ARGUMENT;
  if (n == 0) return SR_OK;
  if (!T || !out) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_pose_distance_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, T, out, n);
  return sr_hip_rc(hipGetLastError());
}

extern "C" int sr_camera_rays_fwd(const float* points, const float* T, float* out, int B, int N, int in_camera_frame,
                                  void* stream) {
  if (B < 0 || N <= 0) return SR_ERR_INVALID_ARGUMENT;
  if (B == 0) return SR_OK;
  if (!points || !T || !out) return SR_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(sr_camera_rays_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, points, T, out, N,
                     in_camera_frame);
  return sr_hip_rc(hipGetLastError());
}
