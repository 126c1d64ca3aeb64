/*
 * simplerecon_hip.h -- C ABI of libsimplerecon_hip.so (MI355X / gfx950).
 *
 * The upstream reference (nianticlabs/simplerecon) has NO native code and NO FFI:
 * its "plugin point" for this path is a Python class contract plus an attribute
 * swap (`model.cost_volume = model.cost_volume.to_fast()`, reference test.py:196-198).
 * This header is therefore the boundary a maintainer binds from Python (ctypes stub
 * in INTEGRATION.md); each entry point cites the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless stated; tensors are dense in the
 *    stated layout; the caller owns and allocates every buffer (no allocation, no
 *    ownership transfer, no global mutable state, no host synchronisation inside);
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *  - return value: 0 = ok, SR_ERR_* for argument errors, 1000 + hipError_t for a
 *    failed launch.  Nothing throws across the ABI.  Re-entrant; safe from several
 *    host threads on different streams as long as workspaces are not shared.
 *  - 4x4 matrices are row-major, 16 floats.
 */
#ifndef SIMPLERECON_HIP_H_
#define SIMPLERECON_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_OK 0
#define SR_ERR_INVALID_ARGUMENT 1
#define SR_ERR_UNSUPPORTED 2
#define SR_ERR_WORKSPACE_TOO_SMALL 3
#define SR_ERR_HIP_BASE 1000

/* ---- run-time switches (r05) ----------------------------------------------------------------------------------------
 * Every tuning / ablation switch of the library, and the two fenced split-precision modes (which change the numerics of the
 * calls that honour them), is an entry of ONE option table: read atomically by the entry points, set explicitly with
 * sr_option_set().  An entry is initialised once, at the library's first option access, from the environment variable of the
 * same name (tuning runs: `SR_WINO_XCD=0 python ...`); after that the environment is never read again -- no getenv() on any
 * launch path, no setenv() by any host.  The table is PROCESS-WIDE: a value set while another thread is launching applies to
 * that thread's next launch, and a HIP graph captured under a mode keeps the kernels of that mode.  Integer options take the
 * integer; the split modes take 0 = off (fp32 MFMA: the product path), 1 = bf16 pieces, 2 = f16 pieces (-1 = the unknown
 * string an environment variable held: the entry points that honour the mode return SR_ERR_INVALID_ARGUMENT). */
enum {
  SR_OPT_MLP_SPLIT = 0, SR_OPT_WINO_SPLIT, SR_OPT_WINO_XCD, SR_OPT_WINO_STAGGER, SR_OPT_WINO_WG_PER_CU, SR_OPT_WINO_NT,
  SR_OPT_WINO_KSPLIT, SR_OPT_CONV_WINO, SR_OPT_CONV_TILE, SR_OPT_CONV_KSPLIT, SR_OPT_MLP_VEC_STORE, SR_OPT_MLP_XCD,
  SR_OPT_MLP_BWD_VALU, SR_OPT_T16_XCD, SR_OPT_POOL_BW, SR_OPT_POOL_XCD, SR_OPT_PW_NT, SR_OPT_PW_KS, SR_OPT_PT_CFG, SR_OPT_PT_KS,
  SR_OPT_DOT_LDS, SR_OPT_DOT_QUAD, SR_OPT_DOT_LDS_G, SR_OPT_DOT_LDS_CULL, SR_OPT_DOT_LDS_CAP, SR_OPT_GEMM_AUTOTUNE,
  SR_OPT_UPSAMPLE_QUAD, SR_OPT_POOL_STREAM, SR_OPT_MLP_RESERVE_CUS,
  SR_OPT_COUNT
};
int sr_option_count(void);
const char* sr_option_name(int id);               /* = the environment variable that seeds it, e.g. "SR_WINO_XCD" */
int sr_option_id(const char* name);               /* -1: no such option */
int sr_option_get(int id, int* value);
int sr_option_set(int id, int value, int* previous /* may be NULL */);
int sr_option_default(int id, int* value);

/* ABI version; bumped on any signature change. */
int sr_abi_version(void);

/* Name of the GPU architecture this library was compiled for ("gfx950"). */
const char* sr_target_arch(void);

/* ------------------------------------------------------------------ cost volume --
 *
 * Depth planes are addressed  planes[b*ps_b + j*ps_d + y*ps_y + x*ps_x]  (strides in
 * elements): (D,1,0,0) for the [B,D] values of generate_depth_planes (reference
 * modules/cost_volume.py:100-136), (D*h*w, h*w, w, 1) for a caller-supplied
 * depth_planes_bdhw (cost_volume.py:247, 297-299).
 *
 * The volume is written at  out_cv[b*cv_sb + j*cv_sd + (y*w + x)*cv_sp]:
 * (D*h*w, h*w, 1) = the reference's b,d,h,w layout, (D*h*w, 1, D) = channels-last.
 */

/* Scratch bytes needed by sr_dot_volume_fwd / sr_mlp_volume_fwd for these sizes. */
size_t sr_volume_workspace_bytes(int B, int K, int C, int h, int w);

/* Stage 1 of either sweep: per-(b,k) geometry records (P = K_src @ T_src_cur, reference
 * utils/geometry_utils.py:78; source camera centres and the DVMVS pose measures of
 * geometry_utils.py:178-191 when T_cur_src != NULL) and the channels-last repack of `src`
 * ([B,K,C,h,w] -> [B*K, h*w, C]) into `workspace`.  T_cur_src = cur_cam_T_src_cam ("src_poses"),
 * [B,K,16], may be NULL for the dot model. */
int sr_volume_prepare(const float* src, const float* K_src, const float* T_src_cur,
                      const float* T_cur_src, int B, int K, int C, int h, int w, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Stage 2 of the dot model: the sweep kernel alone, on a workspace filled by
 * sr_volume_prepare for the same sizes (lets a caller time / profile / re-run it alone). */
int sr_dot_volume_sweep(const float* cur, const float* invK_cur, const float* planes, int64_t ps_b,
                        int64_t ps_d, int64_t ps_y, int64_t ps_x, int B, int K, int C, int h, int w,
                        int D, float* out_cv, int64_t cv_sb, int64_t cv_sd, int64_t cv_sp,
                        float* out_lowest, uint8_t* out_mask, void* workspace,
                        size_t workspace_bytes, void* stream);

/* 1x1 convolution over a dense channels-last map as a plain library GEMM (hipBLASLt, fp32 in / out / accumulate):
 * out[m][co] = act( sum_ci in[m][ci] * weight[co][ci] + bias[co] [+ residual[m][co]] ) for the M = B*H*W pixels of a map
 * whose pixels are `*_pix_stride` floats apart (channel slices of wider buffers are fine).  `weight` is the UNPACKED
 * [Cout][Cin] nn.Conv2d weight (eval-mode BatchNorm folded by the caller); act 0 none, 1 SiLU, 2 ReLU (a residual is added
 * BEFORE the activation).  Replaces nn.Conv2d(k=1) + BatchNorm + SiLU of the image-prior encoder's MBConv blocks
 * (reference depth_model.py:110-116) and the 1x1 skip convs of BasicBlock (layers.py:58-65) where this is faster than
 * sr_conv2d_nhwc_fwd.  `workspace`: sr_gemm1x1_workspace_bytes(), 256-byte aligned.  SR_ERR_UNSUPPORTED when hipBLASLt
 * offers no algorithm for the shape (use sr_conv2d_nhwc_fwd then). */
size_t sr_gemm1x1_workspace_bytes(void);
int sr_gemm1x1_nhwc_fwd(const float* in, int in_pix_stride, const float* weight, const float* bias, const float* residual,
                        int res_pix_stride, float* out, int out_pix_stride, int M, int Cin, int Cout, int act,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Pointwise (1x1, stride 1) convolution as a hand-written fp32-MFMA GEMM (csrc/sr_pw.hip), deterministic:
 *   out[b,p,co] = act( sum_ci gate[b,ci] * in[b,p,ci] * W[co,ci] + bias[co] + residual[b,p,co] ),  p = 0 .. HW-1
 * on channels-last views (batch stride, pixel stride in floats; channel slices of wider buffers are fine).  `packed_w` comes
 * from sr_conv_pack_weights(ksize = 1) (eval-mode BatchNorm folded by the caller); `gate` ([B][Cin], may be null) is the
 * squeeze-excite gate of an MBConv block, applied to the A operand while it is loaded; `residual` may be null; `act_code`
 * as `leaky_slope` of sr_conv2d_nhwc_fwd.  Replaces BasicBlock's 1x1 skip convolution (reference modules/layers.py:20-22,
 * 57-62) and nn.Conv2d(k=1) + BatchNorm (+ SiLU) (+ squeeze-excite scaling of the input) of the image-prior encoder's
 * MBConv blocks (reference experiment_modules/depth_model.py:110-116).  Needs Cin % 4 == 0 and 16-byte aligned input rows
 * (sr_pw_conv_supported; SR_ERR_UNSUPPORTED otherwise -> sr_conv2d_nhwc_fwd).  sr_pw_conv_plan reports the launch plan
 * (32-channel tiles per wave, K split across the waves of a workgroup) chosen for a shape. */
/* 16-bit activation I/O (training under torch.autocast; reference options.py:100-101 `precision: 16`, train.py:132): the
 * same two operators on fp16 (io_dtype = 1) / bf16 (io_dtype = 2) input / residual / output tensors (strides in ELEMENTS;
 * four channels = one 8-byte access, 8-byte aligned rows), widened to fp32 on load and rounded to nearest even on store;
 * weights (packed as for the fp32 entry points), bias, transforms and the MFMA accumulation stay fp32.  The result equals the
 * fp32 entry point's result on the widened inputs, rounded once.  io_dtype = 0 forwards to the fp32 entry point.  The
 * Winograd form never splits K; the pointwise form takes no gate. */
int sr_conv3x3_wino_io_nhwc_fwd(const void* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_u,
                                const float* bias, const void* residual, int64_t res_batch_stride, int res_pix_stride,
                                void* out, int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                int Cout, float leaky_slope, int io_dtype, void* stream);
int sr_pw_conv_io_nhwc_fwd(const void* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_w,
                           const float* bias, const void* residual, int64_t res_batch_stride, int res_pix_stride, void* out,
                           int64_t out_batch_stride, int out_pix_stride, int B, int HW, int Cin, int Cout, float act_code,
                           int io_dtype, void* stream);

/* LDS-tiled form of the same operator for BATCH-DENSE views (batch stride = HW * pixel stride, so the M = B * HW pixel rows
 * are one strided matrix): a workgroup stages (64 | 128) x 32 input and 32 x (64 | 128 | 160) weight tiles through LDS,
 * double-buffered -- each operand byte crosses the CU's vector-memory path once per workgroup instead of once per wave,
 * which is what the MBConv expand / project GEMMs of the image-prior encoder (M = 2 400 ... 9 600) need.  Small problems
 * split K across workgroups: raw partial tiles go to `workspace` (sr_pw_conv_tiled_workspace_bytes; may be null: no
 * split) and are added in index order (deterministic).  Same arguments otherwise; sr_pw_conv_tiled_plan reports the tile
 * configuration (0: 64x128, 1: 128x160, 2: 128x64, 3: 64x64) and the K split chosen for a shape. */
size_t sr_pw_conv_tiled_workspace_bytes(int M, int Cin, int Cout);
int sr_pw_conv_tiled_plan(int M, int Cin, int Cout, int can_split, int* cfg, int* ks);
int sr_pw_conv_tiled_nhwc_fwd(const float* in, int in_pix_stride, const float* packed_w, const float* bias,
                              const float* gate, const float* residual, int res_pix_stride, float* out,
                              int out_pix_stride, int M, int HW, int Cin, int Cout, float act_code, void* workspace,
                              size_t workspace_bytes, void* stream);
int sr_pw_conv_supported(int Cin, int Cout);
int sr_pw_conv_plan(int B, int HW, int Cin, int Cout, int* nt, int* ks);
int sr_pw_conv_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_w,
                        const float* bias, const float* gate, const float* residual, int64_t res_batch_stride,
                        int res_pix_stride, float* out, int64_t out_batch_stride, int out_pix_stride, int B, int HW,
                        int Cin, int Cout, float act_code, void* stream);

/* Standalone forms of the reference's small geometry helpers (utils/geometry_utils.py) for callers outside the fused
 * sweeps; same operation order as the sweeps' internal arithmetic (FP contraction off).
 *  sr_backproject_fwd   BackprojectDepth.forward (:51-59): depth [B,h*w], invK [B,16] -> points [B,4,h*w]
 *  sr_project3d_fwd     Project3D.forward (:72-89): points [B,4,N], K [B,16], T = cam_T_world [B,16] -> [B,3,N]
 *                       (pixel x, pixel y, depth + eps)
 *  sr_pose_distance_fwd pose_distance (:178-191): T [n,16] -> [n,3] = (combined, R_measure, t_measure)
 *  sr_camera_rays_fwd   get_camera_rays (:143-175): points [B,3,N], T [B,16] (world_T_cam, or cam_T_world when
 *                       in_camera_frame) -> unit rays [B,3,N] */
int sr_backproject_fwd(const float* depth, const float* invK, float* out_points, int B, int h, int w, void* stream);
int sr_project3d_fwd(const float* points, const float* K, const float* T, float* out, int B, int N, float eps,
                     void* stream);
/* Adjoints for the training losses that call the two modules (reference losses.py via geometry_utils.py:51-59, 72-89):
 * d_depth [B,h*w] from d_points [B,4,h*w]; d_points [B,4,N] from d_out [B,3,N].  Intrinsics / poses are data. */
int sr_backproject_bwd(const float* grad_points, const float* invK, float* grad_depth, int B, int h, int w, void* stream);
int sr_project3d_bwd(const float* grad_out, const float* points, const float* K, const float* T, float* grad_points, int B,
                     int N, float eps, void* stream);
int sr_pose_distance_fwd(const float* T, float* out, int n, void* stream);
int sr_camera_rays_fwd(const float* points, const float* T, float* out, int B, int N, int in_camera_frame, void* stream);

/* Self-test of the LDS-staged sweep's packed reciprocal (csrc/sr_dot_volume_lds.hip): out_fast[i] = that reciprocal of
 * x[i], out_div[i] = the IEEE division 1.0f / x[i] every other kernel uses; they must agree bit for bit for
 * 2^-60 <= |x| <= 2^60 (outside that range the sweep itself takes the division).  No reference counterpart: the
 * reference divides in ATen (utils/geometry_utils.py:85). */
int sr_selftest_rcp(const float* x, float* out_fast, float* out_div, int n, void* stream);

/* Fused plane sweep of the dot-product model (= sr_volume_prepare + sr_dot_volume_sweep): replaces
 *   CostVolumeManager.build_cost_volume + forward      (cost_volume.py:237-380)
 *   = BackprojectDepth (utils/geometry_utils.py:51-59) + Project3D (:72-89)
 *   + F.grid_sample(bilinear, zeros, align_corners=False) (cost_volume.py:201-212)
 *   + sum_c(warped * cur) * (z' > 0), summed over views  (cost_volume.py:322-329)
 *   + argmax / gather of the depth planes               (cost_volume.py:338-342, 374-378)
 * in one launch per batch, never materialising the warped features.
 *
 *  cur        [B,C,h,w]      reference-frame matching features
 *  src        [B,K,C,h,w]    source-frame matching features
 *  K_src      [B,K,16]       source intrinsics at matching resolution (K_s1_b44)
 *  T_src_cur  [B,K,16]       src_cam_T_cur_cam  ("src_extrinsics")
 *  invK_cur   [B,16]         inverse reference intrinsics ("cur_invK")
 *  out_cv     see above;  out_lowest [B,h,w] (may be NULL);
 *  out_mask   [B,h,w] uint8 (may be NULL): any_k(z'>0) & any_k(2<u<w-2 & 2<v<h-2) at the
 *             LAST plane -- the rule of cost_volume.py:625-637 (the dot model itself
 *             returns None, cost_volume.py:286, 335).
 *  C must be a multiple of 4 and <= 32.
 */
int sr_dot_volume_fwd(const float* cur, const float* src, const float* K_src,
                      const float* T_src_cur, const float* invK_cur, const float* planes,
                      int64_t ps_b, int64_t ps_d, int64_t ps_y, int64_t ps_x, int B, int K, int C,
                      int h, int w, int D, float* out_cv, int64_t cv_sb, int64_t cv_sd,
                      int64_t cv_sp, float* out_lowest, uint8_t* out_mask, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Materialising front end of the sweeps, for the reference's public helper
 * CostVolumeManager.warp_features (cost_volume.py:139-234) / FastFeatureVolumeManager.warp_features
 * (:812-964): for Dp depth planes writes the back-projected points out_world [B,Dp,4,N] (may be NULL),
 * the source-camera depths z' out_depths [B,K,Dp,N], the bilinearly warped source features
 * out_warped [B,K,Dp,C,N], out_mask = (z' > 0) as float [B,K,Dp,N] and the sampling coordinates
 * out_pix [B,K,Dp,2,N] (may be NULL).  N = h*w.  B*K*Dp <= 65535. */
int sr_warp_features_fwd(const float* src, const float* K_src, const float* T_src_cur,
                         const float* invK_cur, const float* planes, int64_t ps_b, int64_t ps_d,
                         int64_t ps_y, int64_t ps_x, int B, int K, int C, int h, int w, int Dp,
                         float* out_world, float* out_depths, float* out_warped, float* out_mask,
                         float* out_pix, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------ metadata-MLP volume --
 *
 * Fused plane sweep of the hero model: replaces FeatureVolumeManager.build_cost_volume + forward
 * and FastFeatureVolumeManager (cost_volume.py:451-736, 967-1164, 345-380): per (b, d, y, x) the
 * C(1+K)+10K+4-channel vector of cost_volume.py:709-723 (warped source features, reference
 * features, masks, source depths, plane depth, dot products, ray angles, rays, pose measures) is
 * built in registers and pushed through MLP = Linear, LeakyReLU, Linear, LeakyReLU, Linear
 * (modules/networks.py:129-147, built at cost_volume.py:438) on fp32 MFMA.
 *
 *  T_cur_src  [B,K,16]  cur_cam_T_src_cam ("src_poses")
 *  W1 [hidden, Cin], b1 [hidden], W2 [hidden, hidden], b2 [hidden], W3 [1, hidden], b3 [1]
 *             = mlp.net.{0,2,4}.{weight,bias};  hidden must be 128, C must be 16.
 *  leaky_slope: 0.01 (nn.LeakyReLU default, networks.py:139); must lie in (0,1).
 *  out_mask as for the dot model (this model does return it, cost_volume.py:625-637).
 */
size_t sr_mlp_volume_workspace_bytes(int B, int K, int C, int h, int w, int hidden);

/* Packs the MLP parameters into the k-step order of the sweep kernel (into `workspace`).
 * Experiment switch (off by default, read per call by the pack AND the sweep, which must agree): environment
 * SR_MLP_SPLIT=bf16|f16 packs layers 1-2 as two 16-bit pieces per weight and selects the split-precision sweep kernel
 * (three exact 16-bit MFMA products per fp32 product, fp32 accumulate; K <= 7 source views, SR_ERR_UNSUPPORTED beyond;
 * unknown values: SR_ERR_INVALID_ARGUMENT).  Tensors handed in and results are fp32 either way. */
int sr_mlp_pack_weights(const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* W3, const float* b3, int hidden, int B, int K, int C, int h, int w,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The sweep kernel alone, on a workspace filled by sr_volume_prepare (with T_cur_src) and
 * sr_mlp_pack_weights for the same sizes. */
int sr_mlp_volume_sweep(const float* cur, const float* invK_cur, const float* planes, int64_t ps_b,
                        int64_t ps_d, int64_t ps_y, int64_t ps_x, float leaky_slope, int B, int K,
                        int C, int h, int w, int D, float* out_cv, int64_t cv_sb, int64_t cv_sd,
                        int64_t cv_sp, float* out_lowest, uint8_t* out_mask, void* workspace,
                        size_t workspace_bytes, void* stream);

/* = sr_volume_prepare + sr_mlp_pack_weights + sr_mlp_volume_sweep. */
int sr_mlp_volume_fwd(const float* cur, const float* src, const float* K_src, const float* T_src_cur,
                      const float* T_cur_src, const float* invK_cur, const float* planes, int64_t ps_b,
                      int64_t ps_d, int64_t ps_y, int64_t ps_x, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* W3, const float* b3, int hidden,
                      float leaky_slope, int B, int K, int C, int h, int w, int D, float* out_cv,
                      int64_t cv_sb, int64_t cv_sd, int64_t cv_sp, float* out_lowest, uint8_t* out_mask,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------- 2-D conv stack -------
 *
 * Channels-last activations: element (b, y, x, c) of a tensor lives at
 *   ptr[b*batch_stride + (y*W + x)*pix_stride + c]      (strides in elements),
 * so a channel slice of a wider concat buffer is addressed by offsetting `ptr` and keeping
 * the buffer's pix_stride -- producers write straight into the consumer's torch.cat layout
 * (reference modules/networks.py:83-89, 124).
 */

/* Floats needed for the packed form of a [Cout, Cin, k, k] Conv2d weight (k = 1 or 3). */
size_t sr_conv_packed_weight_floats(int Cout, int Cin, int ksize);

/* Packs an nn.Conv2d weight ([Cout,Cin,k,k], contiguous) into MFMA B-fragment order. */
int sr_conv_pack_weights(const float* weight, int Cout, int Cin, int ksize, float* packed, void* stream);

/* out = act( conv2d(in, W, stride, padding = k/2) + bias [+ residual] ), act = LeakyReLU(slope)
 * when leaky_slope >= 0, identity otherwise.  One launch replaces nn.Conv2d + bias + the
 * residual add + nn.LeakyReLU of BasicBlock.forward (reference modules/layers.py:68-85) and
 * conv3x3 / conv1x1 (layers.py:7-22).  ksize in {1,3}, stride in {1,2}; fp32 MFMA
 * (exact fp32 products and accumulation).  `residual` has the output's geometry. */
int sr_conv2d_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                       const float* packed_weight, const float* bias, const float* residual,
                       int64_t res_batch_stride, int res_pix_stride, float* out,
                       int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                       int Cout, int ksize, int stride, float leaky_slope, void* stream);

/* Same operator with padding_mode="replicate" (border texels repeated instead of zeros): the 128 -> 16 conv of
 * the matching encoder's tail (reference modules/networks.py:191-197). */
int sr_conv2d_replicate_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                 const float* packed_weight, const float* bias, const float* residual,
                                 int64_t res_batch_stride, int res_pix_stride, float* out,
                                 int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                 int Cout, int ksize, int stride, float leaky_slope, void* stream);

/* Same operator with explicit zero padding on each side (output size floor((H + top + bottom - ksize) / stride) + 1):
 * TensorFlow-"SAME" convolutions of the image-prior encoder pad 0 above / left and 1 below / right when a
 * stride-2 3x3 conv meets an even-sized map.  `leaky_slope` here -- and in every convolution entry point of this
 * header -- also carries the activation code: >= 0 LeakyReLU slope (0 = ReLU), SR_ACT_NONE = identity,
 * SR_ACT_SILU = x * sigmoid(x) (the Winograd entry points implement LeakyReLU / identity only). */
#define SR_ACT_NONE (-1.0f)
#define SR_ACT_SILU (-2.0f)
int sr_conv2d_padded_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                              const float* packed_weight, const float* bias, const float* residual,
                              int64_t res_batch_stride, int res_pix_stride, float* out,
                              int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                              int Cout, int ksize, int stride, int pad_top, int pad_left, int pad_bottom,
                              int pad_right, float leaky_slope, void* stream);

/* sr_conv2d_nhwc_fwd with a split-K launch plan for 1x1 / stride-1 convolutions whose long input-channel chain would
 * otherwise run on a handful of workgroups (e.g. 1536 -> 256 channels on 8 x 15x20 pixels): the channel slabs of a
 * tile are spread over up to 8 work items that store raw partial sums into `workspace`, and a second launch adds them
 * in index order (deterministic) and applies bias / residual / activation.  sr_conv_splitk_workspace_bytes() returns
 * the workspace size for a shape, 0 when the plan would not split (then this entry point equals sr_conv2d_nhwc_fwd
 * and `workspace` may be NULL). */
size_t sr_conv_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int sr_conv2d_splitk_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                              const float* packed_weight, const float* bias, const float* residual,
                              int64_t res_batch_stride, int res_pix_stride, float* out,
                              int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                              int Cout, int ksize, int stride, float leaky_slope, void* workspace,
                              size_t workspace_bytes, void* stream);

/* 3x3 / stride-1 / pad-1 convolution through Winograd F(2x2, 3x3) on the fp32 matrix cores: same operator and
 * epilogue as sr_conv2d_nhwc_fwd (fp32 products and accumulation; 2.25x fewer multiplies).  `packed_u` comes from
 * sr_wino_pack_weights (U = G g G^T in MFMA B-fragment order).  sr_conv_prefers_wino() tells whether this kernel
 * is expected to beat the direct one for a shape (enough 8x16-pixel regions, little padding). */
size_t sr_wino_packed_weight_floats(int Cout, int Cin);
/* Experiment switch (off by default; option SR_OPT_WINO_SPLIT, read by the pack AND the convolution, which must agree):
 * 1 = bf16 / 2 = f16 packs U as two 16-bit pieces (same buffer size) and selects the split-precision kernel (vector
 * instantiation and fp32 tensors only: SR_ERR_UNSUPPORTED otherwise; the value -1: SR_ERR_INVALID_ARGUMENT). */
int sr_wino_pack_weights(const float* weight, int Cout, int Cin, float* packed, void* stream);
int sr_conv_prefers_wino(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int sr_conv3x3_wino_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                             const float* packed_u, const float* bias, const float* residual,
                             int64_t res_batch_stride, int res_pix_stride, float* out,
                             int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                             int Cout, float leaky_slope, void* stream);

/* The same operator through Winograd F(4x4, 3x3) (csrc/sr_wino4.hip, r05): 2.25 multiplies per output and channel pair
 * instead of F(2x2)'s 4 -- 1.78x fewer MFMAs on the full-resolution 64-channel layers of the UNet++ decoder (reference
 * modules/networks.py:20-96, BasicBlock modules/layers.py:24-85).  Interpolation points (0, +-1/2, +-2, inf): exact fp32
 * transforms, fp32 products and accumulation; fp32 error ~1.3e-6 of the output range on a 64-channel layer (F(2x2): 3e-7).
 * `packed_u` comes from sr_wino4_pack_weights (U = G g G^T, computed in double, in MFMA A-fragment order).  Needs channel
 * counts in whole quads and 16-byte aligned rows (SR_ERR_UNSUPPORTED otherwise: use sr_conv3x3_wino_nhwc_fwd).
 * sr_conv_prefers_wino4(): which kernel FORM is expected to beat F(2x2) for the shape -- 0 none, 1 two 4-wave workgroups per
 * CU, 3 one wave-specialised 8-wave workgroup per CU (pass it as `variant` of sr_conv3x3_wino4_variant_nhwc_fwd); `mode`
 * 0 = never, 1 = the rule fitted on profiles/r05_wino4_shape_sweep.txt, 2 = form 1 wherever the kernel applies.  The mode
 * is an ARGUMENT (the Python host reads SR_CONV_WINO4 once at import): no environment reads in here. */
size_t sr_wino4_packed_weight_floats(int Cout, int Cin);
int sr_wino4_pack_weights(const float* weight, int Cout, int Cin, float* packed, void* stream);
int sr_conv_prefers_wino4(int B, int H, int W, int Cin, int Cout, int mode);
int sr_conv3x3_wino4_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_u,
                              const float* bias, const float* residual, int64_t res_batch_stride, int res_pix_stride,
                              float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                              int Cout, float leaky_slope, void* stream);

/* The same with the kernel form chosen by the caller: `variant` 0 = the default (what sr_conv3x3_wino4_nhwc_fwd launches),
 * 1 = two independent 4-wave workgroups per CU, 2 = one 8-wave workgroup per CU whose halves alternate transform and MFMA
 * phases (kept for A/B measurements).  Bit-identical results. */
int sr_conv3x3_wino4_variant_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* packed_u,
                                      const float* bias, const float* residual, int64_t res_batch_stride,
                                      int res_pix_stride, float* out, int64_t out_batch_stride, int out_pix_stride, int B,
                                      int H, int W, int Cin, int Cout, float leaky_slope, int variant, void* stream);

/* Split-K variant for layers with few output regions and a long chain of input slabs (deep low-resolution levels,
 * batch 1): work items cover Cin / ks input channels each and store raw partial outputs to `workspace`
 * ([ks][B, H*W, Cout] floats, 16-byte aligned); a second kernel adds them in index order (deterministic) and applies
 * bias + residual + LeakyReLU.  sr_wino_splitk_factor() = ks the launch plan picks for a shape (1: no split, no
 * workspace needed); with a NULL / too small workspace the call runs unsplit. */
int sr_wino_splitk_factor(int B, int H, int W, int Cin, int Cout);
size_t sr_wino_splitk_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int sr_conv3x3_wino_splitk_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride,
                                    const float* packed_u, const float* bias, const float* residual,
                                    int64_t res_batch_stride, int res_pix_stride, float* out,
                                    int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin,
                                    int Cout, float leaky_slope, void* workspace, size_t workspace_bytes,
                                    void* stream);

/* Name of the kernel instantiation sr_conv3x3_wino_nhwc_fwd launches for these arguments (the output-channel block
 * is chosen per launch); aligned_in / aligned_out = input / output+residual+bias rows are 16-byte aligned with
 * channel counts that are multiples of 4.  For profilers. */
const char* sr_wino_kernel_name(int B, int H, int W, int Cin, int Cout, int aligned_in, int aligned_out);

/* Name of the kernel instantiation sr_conv2d_nhwc_fwd launches for these arguments (tile shape is
 * chosen per launch); `aligned16` = input pointer / strides are 16-byte aligned.  For profilers. */
const char* sr_conv_kernel_name(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int aligned16);

/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) -- `upsample` of the
 * reference (utils/generic_utils.py:96-105) -- on channels-last data, [B,H,W,C] -> [B,2H,2W,C]. */
int sr_upsample2x_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out,
                           int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C,
                           void* stream);

/* out[i] = exp(in[i]) over n contiguous floats: depth_pred = exp(log_depth_pred) (reference depth_model.py:392-400). */
int sr_exp_fwd(const float* in, float* out, int64_t n, void* stream);

/* ------------------------------------------------------ matching-feature encoder -------
 *
 * ResnetMatchingEncoder (reference modules/networks.py:149-205): antialiased ResNet-18 stem + layer1, then
 * conv1x1 -> InstanceNorm -> LeakyReLU(0.2) -> conv3x3 (replicate padding) -> InstanceNorm.  layer1 and the tail
 * convolutions go through sr_conv2d_nhwc_fwd / sr_conv3x3_wino_nhwc_fwd / sr_conv2d_replicate_nhwc_fwd with the
 * eval-mode BatchNorm folded into weight and bias by the caller; the entry points below cover the rest.
 */

/* Packed form of the stem weight nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False).weight ([64,3,7,7]);
 * only Cout = 64 is implemented (returns 0 / SR_ERR_UNSUPPORTED otherwise). */
size_t sr_stem_packed_weight_floats(int Cout);
int sr_stem_pack_weights(const float* weight, int Cout, float* packed, void* stream);

/* out[b, y, x, co] = act( conv7x7_s2_p3(in)[b, co, y, x] * scale[co] + shift[co] ): encoder.conv1 + bn1 (eval mode:
 * scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale; NULL = identity) + ReLU
 * (leaky_slope = 0; < 0: no activation) -- reference networks.py:176-179.  `in` is the image with arbitrary element
 * strides (NCHW or channels-last), `out` is channels-last [B, H/2, W/2, 64]. */
int sr_stem7x7_fwd(const float* in, int64_t in_batch_stride, int64_t in_chan_stride, int64_t in_row_stride,
                   int64_t in_col_stride, const float* packed_weight, const float* scale, const float* shift,
                   float leaky_slope, float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H, int W,
                   int Cout, void* stream);

/* encoder.maxpool of the antialiased backbone: nn.MaxPool2d(kernel_size=2, stride=1) followed by
 * BlurPool(filt_size=4, stride=2, reflect padding (1,2,1,2), taps outer([1,3,3,1])/64), fused.
 * [B,H,W,C] -> [B,(H-2)/2+1,(W-2)/2+1,C] channels-last, C % 4 == 0, H, W >= 4. */
int sr_maxblurpool_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out,
                            int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C, void* stream);

/* nn.InstanceNorm2d(C) (affine=False, biased variance over H*W per image and channel) optionally followed by
 * LeakyReLU(leaky_slope) (< 0: none) -- reference networks.py:188-189, 198.  Deterministic (no atomics); `out` may
 * alias `in`.  C % 4 == 0, C <= 256.  Workspace: sr_instance_norm_workspace_bytes, 16-byte aligned. */
size_t sr_instance_norm_workspace_bytes(int B, int H, int W, int C);
int sr_instance_norm_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out,
                              int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C, float eps,
                              float leaky_slope, void* workspace, size_t workspace_bytes, void* stream);

/* Statistics only: stats[b][0][c] = mean, stats[b][1][c] = 1 / sqrt(var + eps) ([B,2,C] floats, 16-byte aligned) -- for
 * consumers that normalise on the fly (sr_conv3x3_c16_nhwc_fwd).  Same workspace as sr_instance_norm_nhwc_fwd. */
int sr_instance_norm_stats_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, int B, int H, int W, int C,
                                float eps, float* stats, void* workspace, size_t workspace_bytes, void* stream);

/* Conv2d(64, 128, 1) + bias AND the InstanceNorm2d statistics of its output in one pass (reference
 * modules/networks.py:187-188): `out` is the raw convolution, `stats` [B,2,128] = (mean, 1 / sqrt(var + eps)) as
 * sr_instance_norm_stats_nhwc would compute them from `out`.  weight [128][64] (the module's own layout, no packing).
 * Other channel counts: SR_ERR_UNSUPPORTED (callers run sr_conv1x1 + sr_instance_norm_stats_nhwc instead). */
size_t sr_conv1x1_stats_workspace_bytes(int B, int H, int W, int Cout);
int sr_conv1x1_stats_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* weight,
                              const float* bias, float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H,
                              int W, int Cin, int Cout, float eps, float* stats, void* workspace, size_t workspace_bytes,
                              void* stream);

/* nn.Conv2d(Cin, Cout <= 16, 3, padding=1, padding_mode = replicate ? "replicate" : "zeros") + bias [+ LeakyReLU] on
 * channels-last data, with an optional InstanceNorm (+ LeakyReLU(in_leaky_slope)) applied to the INPUT on the fly from
 * `in_stats` (sr_instance_norm_stats_nhwc; NULL = input used as is): the InstanceNorm -> LeakyReLU -> Conv2d(128, 16,
 * replicate) end of the matching encoder (reference modules/networks.py:188-197) without materialising the
 * normalised tensor.  Cin % 32 == 0; fp32 MFMA 16x16x4 (no padded output channels). */
size_t sr_conv3x3_c16_packed_weight_floats(int Cout, int Cin);
int sr_conv3x3_c16_pack_weights(const float* weight, int Cout, int Cin, float* packed, void* stream);
int sr_conv3x3_c16_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* in_stats,
                            float in_leaky_slope, const float* packed_weight, const float* bias, float* out,
                            int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int Cin, int Cout,
                            int replicate, float leaky_slope, void* stream);

/* ------------------------------------------------------------- TSDF fusion -----------
 *
 * TSDFFuser.integrate_depth (reference tools/tsdf.py:238-320; project_to_camera :218-236; voxel coordinates of
 * TSDF.from_bounds / generate_voxel_coords :69-111): fuses a batch of B depth maps, in order, into the fp16 TSDF volume.
 * All tensors are fp16 as the reference feeds them (OurFuser.fuse_frames, tools/fusers_helper.py:62-68); results are
 * bit-identical to the reference executed on CPU.
 *
 *  tsdf_values, tsdf_weights : [X,Y,Z] fp16, contiguous (z fastest), updated in place; Z % 8 == 0 (the reference rounds
 *                              volume dimensions to multiples of 8, tsdf.py:16,80-85), 16-byte aligned
 *  voxel_coords              : [3,X,Y,Z] fp16 explicit world coordinates, or NULL: origin + index * voxel_size
 *                              (fp32, rounded to fp16) as TSDF.from_bounds generates them -- saves 6 of 10 bytes/voxel
 *  depth [B,H,W] fp16, depth_mask [B,H,W] bool or NULL, K / T [B,4,4] fp16 row-major (intrinsics, cam_T_world)
 *  min_depth, max_depth, truncation (= truncation_size * voxel_size), maxW: the fuser's python scalars as fp32;
 *  depth_range = max_depth - min_depth evaluated in double, as fp32.
 */
int sr_tsdf_integrate_fwd(void* tsdf_values, void* tsdf_weights, const void* voxel_coords, int X, int Y, int Z,
                          float origin_x, float origin_y, float origin_z, float voxel_size, const void* depth,
                          const uint8_t* depth_mask, const void* K, const void* T, int B, int H, int W,
                          float min_depth, float max_depth, float depth_range, float truncation, float maxW,
                          void* stream);

/* ------------------------------------------------------ backward (training) -------------
 *
 * Backward of sr_dot_volume_sweep (reference: autograd through CostVolumeManager.build_cost_volume,
 * modules/cost_volume.py:237-335 -- grid_sample backward + the broadcasting mul / sum): given grad_cv = dL/d
 * cost_volume (strides g_sb, g_sd, g_sp like the forward's volume strides), writes d_cur [B,C,h,w] and / or
 * d_src [B,K,C,h,w] (either may be NULL).  Poses, intrinsics and depth planes are data (no gradient, as in the
 * reference).  `workspace` must have been filled by sr_volume_prepare for the same src / K_src / T_src_cur (what
 * sr_dot_volume_fwd leaves behind); `scratch` (sr_dot_volume_bwd_scratch_bytes, 16-byte aligned) holds the
 * channels-last d_src accumulation image.  d_src is accumulated with hardware fp32 atomics: its summation order, like
 * torch's grid_sample backward, is not fixed. */
size_t sr_dot_volume_bwd_scratch_bytes(int B, int K, int C, int h, int w);
int sr_dot_volume_bwd(const float* grad_cv, int64_t g_sb, int64_t g_sd, int64_t g_sp, const float* cur,
                      const float* invK_cur, const float* planes, int64_t ps_b, int64_t ps_d, int64_t ps_y,
                      int64_t ps_x, int B, int K, int C, int h, int w, int D, float* d_cur, float* d_src,
                      void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* ---- backward of the conv stack (reference: autograd through BasicBlock / CVEncoder / DepthDecoderPP, modules/layers.py:
 * 24-85, modules/networks.py:20-127; train.py:126-145).  Data gradients reuse the forward kernels on
 * sr_conv_flip_transpose_weights(W) (stride 2: on the sr_zero_stuff2x_nhwc'ed output gradient).
 *  sr_conv_wgrad_nhwc            d_weight [Cout,Cin,k,k] (written, not accumulated) = sum over pixels of grad_out x input
 *                                patches; fp32 MFMA, per-workgroup partial slabs in `workspace`
 *                                (sr_conv_wgrad_workspace_bytes) added in index order by a second kernel (deterministic)
 *  sr_bias_grad_nhwc             d_bias [C] = sum over pixels of grad_out
 *  sr_act_bwd                    grad * LeakyReLU'(saved output), dense arrays of n floats
 *  sr_act_bwd_bias_nhwc          both of the above in one float4 pass over a dense [pixels, C] gradient (C % 4 == 0):
 *                                grad_pre = grad * LeakyReLU'(out_saved) (out_saved null: no activation, nothing written),
 *                                d_bias = sum over pixels of grad_pre (null: skipped); per-block partial sums in
 *                                `workspace` (sr_act_bwd_bias_workspace_bytes) are added in block order (deterministic)
 *  sr_zero_stuff2x_nhwc          out [B,Hs,Ws,C] dense: out[:, 2y, 2x] = in[:, y, x], zero elsewhere
 *  sr_upsample2x_bwd_nhwc        adjoint of sr_upsample2x_nhwc_fwd: grad_out [B,2H,2W,C] -> grad_in [B,H,W,C]
 *  sr_conv_flip_transpose_weights  out [Cin,Cout,k,k] = W[co,ci,k-1-ky,k-1-kx]
 *  sr_mul_fwd                    out = a * b (backward of depth = exp(log_depth)) */
size_t sr_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int sr_conv_wgrad_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* grad_out,
                       int64_t g_batch_stride, int g_pix_stride, float* d_weight, int B, int H, int W, int Cin, int Cout,
                       int ksize, int stride, void* workspace, size_t workspace_bytes, void* stream);
int sr_bias_grad_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, float* d_bias, int B, int H, int W,
                      int C, void* stream);
int sr_act_bwd(const float* grad, const float* out_saved, float* grad_pre, int64_t n, float leaky_slope, void* stream);
size_t sr_act_bwd_bias_workspace_bytes(int64_t pixels, int C);
int sr_act_bwd_bias_nhwc(const float* grad, const float* out_saved, float* grad_pre, float* d_bias, int64_t pixels, int C,
                         float leaky_slope, void* workspace, size_t workspace_bytes, void* stream);
int sr_zero_stuff2x_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, float* out, int B, int H, int W,
                         int Hs, int Ws, int C, void* stream);
int sr_upsample2x_bwd_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, float* grad_in,
                           int64_t in_batch_stride, int in_pix_stride, int B, int H, int W, int C, void* stream);
int sr_conv_flip_transpose_weights(const float* weight, int Cout, int Cin, int ksize, float* out, void* stream);
int sr_mul_fwd(const float* a, const float* b, float* out, int64_t n, void* stream);

/* Backward of sr_mlp_volume_sweep (reference: autograd through FeatureVolumeManager.build_cost_volume + MLP,
 * modules/cost_volume.py:451-736, modules/networks.py:129-147): d_cur [B,C,h,w], d_src [B,K,C,h,w] and the gradients
 * of the six MLP tensors in their nn.Linear layouts (dW1 [hidden][Cin], db1, dW2 [hidden][hidden], db2, dW3 [1][hidden],
 * db3 [1]; all written, not accumulated).  W1..W3 are the UNPACKED nn.Linear weights.  `workspace` must have been filled
 * by sr_volume_prepare WITH T_cur_src (pose features) for the same sources; `scratch`
 * (sr_mlp_volume_bwd_scratch_bytes, 256-byte aligned) holds the channels-last d_src image and weight transposes.
 * hidden = 128, C = 16, up to 15 views (MLP width <= 416); SR_ERR_UNSUPPORTED otherwise.  The six GEMM-shaped phases run on
 * the fp32 matrix cores (v_mfma_f32_32x32x2_f32); SR_MLP_BWD_VALU=1 selects the round-1 VALU kernel (<= 9 views).
 * Feature-map and weight gradients are accumulated with hardware fp32 atomics (summation order not fixed). */
size_t sr_mlp_volume_bwd_scratch_bytes(int B, int K, int C, int h, int w, int hidden);
int sr_mlp_volume_bwd(const float* grad_cv, int64_t g_sb, int64_t g_sd, int64_t g_sp, const float* cur,
                      const float* invK_cur, const float* planes, int64_t ps_b, int64_t ps_d, int64_t ps_y,
                      int64_t ps_x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                      float leaky_slope, int B, int K, int C, int h, int w, int D, int hidden, float* d_cur,
                      float* d_src, float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3,
                      void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------ image-prior encoder ------------
 *
 * timm `tf_efficientnetv2_s` feature pyramid (reference modules/depth_model.py:110-116: the `encoder` of
 * DepthModel; third-party architecture).  Dense convolutions (stem, ConvBnAct, FusedMBConv, every 1x1) use
 * sr_conv2d_padded_nhwc_fwd / sr_conv2d_nhwc_fwd with eval-mode BatchNorm folded into weight and bias and
 * SR_ACT_SILU; the entry points below cover the depthwise / squeeze-excite part of the MBConv blocks.
 * All tensors channels-last fp32, C % 4 == 0, 16-byte aligned. */

/* Depthwise 3x3 convolution + bias + activation (`leaky_slope` as above).  weight9c: [9][C] tap-major
 * (= nn.Conv2d(C, C, 3, groups=C).weight[c, 0, ky, kx] at [(ky * 3 + kx) * C + c], BatchNorm scale folded in).
 * pool_partial (optional): [B][sr_dwconv3x3_pool_bands(Ho)][C] sums of the activated output over bands of
 * output rows -- the squeeze-excite average pool, finished by sr_se_gate_fwd. */
int sr_dwconv3x3_pool_bands(int Ho);
int sr_dwconv3x3_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* weight9c,
                          const float* bias, float* out, int64_t out_batch_stride, int out_pix_stride,
                          float* pool_partial, int B, int H, int W, int C, int stride, int pad_top, int pad_left,
                          int pad_bottom, int pad_right, float leaky_slope, void* stream);

/* Squeeze-excite gate: gate[b, c] = sigmoid(w_expand[c, :] . silu(w_reduce . mean[b, :] + b_reduce) + b_expand[c])
 * with mean[b, c] = (sum over bands of pool_partial[b, band, c]) / pixels.  w_reduce: [rd][C]; w_expand: [C][rd]. */
int sr_se_gate_fwd(const float* pool_partial, int bands, int pixels, const float* w_reduce, const float* b_reduce,
                   const float* w_expand, const float* b_expand, float* gate, int B, int C, int rd, void* stream);

/* EfficientNetV2's RGB stem (conv_stem of timm's tf_efficientnetv2_s, reference depth_model.py:110-116): act(conv3x3 /
 * stride 2 (3 -> 24) + bias) with explicit top / left zero padding (TF-"SAME": 0 / 0 on even images, the odd pixel goes
 * below / right), image read through its strides (NCHW or channels-last), output channels-last.  `weight27c` =
 * [ky][kx][ci][24] with the eval-mode BatchNorm scale folded in.  Byte work (29 MB in, 59 MB out per 8 images): a VALU
 * kernel, not the padded-K implicit GEMM.  Cout = 24 only (SR_ERR_UNSUPPORTED otherwise -> sr_conv2d_padded_nhwc_fwd). */
int sr_rgb_stem3x3s2_fwd(const float* image, int64_t sb, int64_t sc, int64_t sy, int64_t sx, const float* weight27c,
                         const float* bias, float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H, int W,
                         int Cout, int pad_top, int pad_left, int Ho, int Wo, float act_code, void* stream);

/* The front half of a stride-1 MBConv block in ONE launch (csrc/sr_mbconv_fused.hip, r05): 1x1 expansion (BatchNorm folded,
 * SiLU) -> depthwise 3x3 / pad 1 (BatchNorm folded, SiLU) -> squeeze-excite average pool -> squeeze-excite gates
 * sigmoid(W2 silu(W1 mean + b1) + b2) -- timm's InvertedResidual up to the projection (reference
 * experiment_modules/depth_model.py:110-116 builds the encoder from timm), which stays sr_pw_conv_nhwc_fwd with `gate`.
 * `w_expand` [mid][Cin], `w_dw9c` [9][mid] tap-major, `w_reduce` [rd][mid], `w_excite` [mid][rd]; `out` [B][H*W][mid]
 * channels-last view, `pool` [B][mid] (channel sums, by-product), `gate` [B][mid]; `counters`: B zeroed 32-bit words that the
 * call leaves zeroed (one arrival counter per image: the last workgroup of an image computes its gates).  `counters` must not be
 * shared by launches that may be in flight at the same time (different streams): one buffer per stream.  Deterministic.
 * sr_mbconv_fused_supported(): Cin in {128, 160, 256}, mid % 16 == 0, rd <= 64, H*W*64 bytes within the LDS budget. */
int sr_mbconv_fused_supported(int H, int W, int Cin, int mid, int rd);
int sr_mbconv_expand_dw_se_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* w_expand,
                               const float* b_expand, const float* w_dw9c, const float* b_dw, const float* w_reduce,
                               const float* b_reduce, const float* w_excite, const float* b_excite, float* out,
                               int64_t out_batch_stride, int out_pix_stride, float* pool, float* gate, unsigned* counters,
                               int B, int H, int W, int Cin, int mid, int rd, void* stream);

/* The squeeze-excite gates alone, gate[b][c] = sigmoid(W2 silu(W1 mean_b + b1) + b2)[c], from sr_dwconv3x3_nhwc_fwd's partial
 * sums (timm SqueezeExcite of the MBConv blocks, reference depth_model.py:110-116) for a consumer that applies them itself:
 * sr_pw_conv_nhwc_fwd(gate = ...) scales the projection's input while loading it.  `hidden`: scratch [B][rd]. */
int sr_se_gate2_fwd(const float* pool_partial, int bands, int pixels, const float* w_reduce, const float* b_reduce,
                    const float* w_expand, const float* b_expand, float* hidden, float* gate, int B, int C, int rd,
                    void* stream);

/* The whole squeeze-excite step of an MBConv block in two short launches: hidden[b, j] = silu(w_reduce[j] . mean[b]
 * + b_reduce[j]) (workspace `hidden`: B * rd floats), then out[b, y, x, c] = in[b, y, x, c] * gate[b, c] with the gate
 * of sr_se_gate_fwd (also written to `gate` [B][C] when non-null).  rd <= 256; in-place allowed. */
int sr_se_scale_nhwc_fwd(const float* pool_partial, int bands, const float* w_reduce, const float* b_reduce,
                         const float* w_expand, const float* b_expand, float* hidden, const float* in,
                         int64_t in_batch_stride, int in_pix_stride, float* out, int64_t out_batch_stride,
                         int out_pix_stride, float* gate, int B, int H, int W, int C, int rd, void* stream);

/* out = a + b on channels-last views (in-place allowed): the identity skip of the stage-0 ConvBnAct blocks. */
int sr_add_nhwc_fwd(const float* a, int64_t a_batch_stride, int a_pix_stride, const float* b, int64_t b_batch_stride,
                    int b_pix_stride, float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H, int W,
                    int C, void* stream);

/* out[b, y, x, c] = in[b, y, x, c] * gate[b, c] (in-place allowed). */
int sr_scale_channels_nhwc_fwd(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* gate,
                               float* out, int64_t out_batch_stride, int out_pix_stride, int B, int H, int W, int C,
                               void* stream);

/* ------------------------------------------------------ encoder training path -----------
 *
 * Backward (and training-mode forward) pieces of the two encoders, csrc/sr_train.hip: the reference trains
 * ResnetMatchingEncoder (modules/networks.py:149-205) and the timm EfficientNetV2-S pyramid (depth_model.py:110-116) end
 * to end with BatchNorm in training mode (train.py:126-145).  Channels-last fp32 views (batch stride, pixel stride);
 * `act_code` as `leaky_slope` above (>= 0 LeakyReLU slope, SR_ACT_NONE, SR_ACT_SILU; SR_ACT_SIGMOID for the small dense
 * layers).  Column reductions are two-stage and deterministic (no atomics).
 *
 *  sr_norm_stats_nhwc        mean / biased variance per (group, channel) over pixels; per_image = 0: one group = the whole
 *                            batch (BatchNorm2d training statistics), 1: one group per image (InstanceNorm2d)
 *  sr_norm_act_fwd_nhwc      y = act(gamma * (x - mean) / sqrt(var + eps) + beta); gamma / beta [C] or null
 *  sr_norm_act_bwd_nhwc      dx (+ d_gamma, d_beta [C] when non-null); train_stats = 1: the statistics are functions of x
 *  sr_rowsum_nhwc            out[b, c] = scale * sum over pixels of x (g null) or of x * g
 *  sr_maxblurpool_bwd_nhwc   adjoint of sr_maxblurpool_nhwc_fwd (first maximum of a window receives its gradient)
 *  sr_replicate_pad_nhwc_fwd / _bwd   y [B,H+2p,W+2p,C] dense <- x clamped at the borders; adjoint (dense in / out)
 *  sr_im2col7x7s2_nhwc       col [B,Ho,Wo,Kp] of the 7x7 / stride-2 / pad-3 stem over a strided 3-channel image: with
 *                            sr_conv_wgrad_nhwc (1x1) it yields the stem's weight gradient
 *  sr_dwconv3x3_bwd_nhwc     depthwise 3x3 (weight [C][3][3], explicit top / left pad): d_in dense and / or d_weight
 *  sr_scale_bwd_nhwc         d_in = grad_out * gate[b, c] (+ pool_scale * d_pool[b, c]): backward of the squeeze-excite scaling
 *  sr_small_linear_fwd / _bwd   y = act(x W^T + b) on [B, K] -> [B, N] (the two dense layers of squeeze-excite)
 *  sr_act_in_bwd             grad * act'(saved input)
 *  sr_conv_wgrad_padded_nhwc sr_conv_wgrad_nhwc with explicit top / left zero padding and gradient size Ho x Wo */
#define SR_ACT_SIGMOID (-3.0f)
size_t sr_norm_workspace_bytes(int B, int HW, int C, int per_image);
int sr_norm_stats_nhwc(const float* x, int64_t x_batch_stride, int x_pix_stride, int B, int HW, int C, int per_image,
                       float* mean, float* var, void* workspace, size_t workspace_bytes, void* stream);
int sr_norm_act_fwd_nhwc(const float* x, int64_t x_batch_stride, int x_pix_stride, const float* mean, const float* var,
                         float eps, const float* gamma, const float* beta, float act_code, int per_image, float* y,
                         int64_t y_batch_stride, int y_pix_stride, int B, int HW, int C, void* stream);
int sr_norm_act_bwd_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, const float* x,
                         int64_t x_batch_stride, int x_pix_stride, const float* mean, const float* var, float eps,
                         const float* gamma, const float* beta, float act_code, int per_image, int train_stats,
                         float* d_in, int64_t d_batch_stride, int d_pix_stride, float* d_gamma, float* d_beta, int B,
                         int HW, int C, void* workspace, size_t workspace_bytes, void* stream);
int sr_rowsum_nhwc(const float* x, int64_t x_batch_stride, int x_pix_stride, const float* g, int64_t g_batch_stride,
                   int g_pix_stride, int B, int HW, int C, float scale, float* out, void* workspace,
                   size_t workspace_bytes, void* stream);
size_t sr_maxblurpool_bwd_workspace_bytes(int B, int H, int W, int C);
int sr_maxblurpool_bwd_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, const float* x,
                            int64_t x_batch_stride, int x_pix_stride, float* grad_in, int64_t d_batch_stride,
                            int d_pix_stride, int B, int H, int W, int C, void* workspace, size_t workspace_bytes,
                            void* stream);
int sr_replicate_pad_nhwc_fwd(const float* x, int64_t x_batch_stride, int x_pix_stride, float* y, int B, int H, int W,
                              int C, int pad, void* stream);
int sr_replicate_pad_nhwc_bwd(const float* grad_padded, float* grad_in, int B, int H, int W, int C, int pad, void* stream);
int sr_im2col7x7s2_nhwc(const float* image, int64_t batch_stride, int64_t chan_stride, int64_t row_stride,
                        int64_t col_stride, float* col, int B, int H, int W, int Kp, void* stream);
size_t sr_dwconv3x3_bwd_workspace_bytes(int B, int Ho, int Wo, int C);
int sr_dwconv3x3_bwd_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, const float* x,
                          int64_t x_batch_stride, int x_pix_stride, const float* weight, float* d_in, float* d_weight,
                          int B, int H, int W, int C, int stride, int pad_top, int pad_left, int Ho, int Wo,
                          void* workspace, size_t workspace_bytes, void* stream);
int sr_scale_bwd_nhwc(const float* grad_out, int64_t g_batch_stride, int g_pix_stride, const float* gate,
                      const float* d_pool, float pool_scale, float* d_in, int B, int HW, int C, void* stream);
int sr_small_linear_fwd(const float* x, const float* W, const float* bias, float* pre, float* y, int B, int K, int N,
                        float act_code, void* stream);
int sr_small_linear_bwd(const float* dy, const float* pre, const float* x, const float* W, float* dx, float* dW, float* db,
                        int B, int K, int N, float act_code, void* stream);
int sr_act_in_bwd(const float* grad, const float* pre, float* grad_pre, int64_t n, float act_code, void* stream);
/* out = act(a + b) on dense arrays (b null: act(a)); pre (optional) = a + b */
int sr_add_act_fwd(const float* a, const float* b, float* pre, float* out, int64_t n, float act_code, void* stream);
size_t sr_conv_wgrad_padded_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ksize);
int sr_conv_wgrad_padded_nhwc(const float* in, int64_t in_batch_stride, int in_pix_stride, const float* grad_out,
                              int64_t g_batch_stride, int g_pix_stride, float* d_weight, int B, int H, int W, int Cin,
                              int Cout, int ksize, int stride, int pad_top, int pad_left, int Ho, int Wo, void* workspace,
                              size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLERECON_HIP_H_ */
