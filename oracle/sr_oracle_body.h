/*
 * sr_oracle_body.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the SimpleRecon plane-sweep cost-volume / matching-MLP /
 * BasicBlock-conv hot path, written from the reference's algorithm (not copied):
 * every function cites the reference file:line it follows (paths relative to the
 * upstream repo root, nianticlabs/simplerecon).  Included twice by sr_oracle.c
 * with REAL = float (suffix _f32: mirrors the reference's fp32 arithmetic) and
 * REAL = double (suffix _f64: arbitration "truth" for tolerance budgeting).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this.  The product (simplerecon_amd/) never links or imports it.
 *
 * Parity status: PINNED against outputs of the reference's own modules executed
 * in the build container (tests/golden/ npz files, generator tests/golden/make_golden.py).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* ---- geometry ----------------------------------------------------------- */

/* 4x4 row-major matmul, P = K @ T          (utils/geometry_utils.py:78) */
static void FN(mat44_mul)(const float* A, const float* Bm, REAL* out) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      REAL s = 0;
      for (int k = 0; k < 4; ++k) s += (REAL)A[i * 4 + k] * (REAL)Bm[k * 4 + j];
      out[i * 4 + j] = s;
    }
}

typedef struct {
  REAL pix_x, pix_y; /* q_x*s, q_y*s                     (geometry_utils.py:87) */
  REAL zp;           /* z' = q_z + eps                   (geometry_utils.py:84) */
  REAL X[3];         /* reference-camera 3-D point       (geometry_utils.py:56-57) */
} FN(proj_t);

/* BackprojectDepth.forward (geometry_utils.py:51-59) followed by
 * Project3D.forward (geometry_utils.py:72-89) for ONE pixel / plane / view. */
static inline void FN(backproject_project)(const float* invK /*4x4*/, const REAL* P /*4x4*/,
                                           int x, int y, REAL d, FN(proj_t)* o) {
  const REAL eps = (REAL)1e-8f; /* Project3D eps buffer is float32(1e-8) (geometry_utils.py:66-70) */
  const REAL px = (REAL)x + (REAL)0.5, py = (REAL)y + (REAL)0.5; /* geometry_utils.py:34-44 */
  REAL r[3];
  for (int i = 0; i < 3; ++i)
    r[i] = (REAL)invK[i * 4 + 0] * px + (REAL)invK[i * 4 + 1] * py + (REAL)invK[i * 4 + 2];
  for (int i = 0; i < 3; ++i) o->X[i] = d * r[i];
  REAL q[3];
  for (int i = 0; i < 3; ++i)
    q[i] = P[i * 4 + 0] * o->X[0] + P[i * 4 + 1] * o->X[1] + P[i * 4 + 2] * o->X[2] + P[i * 4 + 3];
  const REAL aq = q[2] < 0 ? -q[2] : q[2];
  o->zp = q[2] + eps;
  const REAL s = (aq > eps) ? (REAL)1 / o->zp : (REAL)1;
  o->pix_x = q[0] * s;
  o->pix_y = q[1] * s;
}

/* F.grid_sample(bilinear, zeros, align_corners=False) of a C-channel NCHW map at
 * the point given in reference pixel coordinates (cost_volume.py:199-212):
 *   uv = 2*pix*(1/w, 1/h) - 1 ;  ix = ((uv+1)*w - 1)/2  (ATen unnormalize)      */
static inline void FN(bilinear_sample)(const float* img /*[C,h,w]*/, int C, int h, int w,
                                       REAL pix_x, REAL pix_y, REAL* out /*[C]*/) {
  const REAL sx = (REAL)(float)(1.0 / (double)w), sy = (REAL)(float)(1.0 / (double)h);
  const REAL u = (REAL)2 * pix_x * sx - (REAL)1, v = (REAL)2 * pix_y * sy - (REAL)1;
  const REAL ix = ((u + (REAL)1) * (REAL)w - (REAL)1) / (REAL)2;
  const REAL iy = ((v + (REAL)1) * (REAL)h - (REAL)1) / (REAL)2;
  const REAL fx0 = FLOOR(ix), fy0 = FLOOR(iy);
  const REAL fx1 = fx0 + 1, fy1 = fy0 + 1;
  const REAL w_nw = (fx1 - ix) * (fy1 - iy), w_ne = (ix - fx0) * (fy1 - iy);
  const REAL w_sw = (fx1 - ix) * (iy - fy0), w_se = (ix - fx0) * (iy - fy0);
  const int vx0 = (fx0 >= 0 && fx0 <= (REAL)(w - 1)), vx1 = (fx1 >= 0 && fx1 <= (REAL)(w - 1));
  const int vy0 = (fy0 >= 0 && fy0 <= (REAL)(h - 1)), vy1 = (fy1 >= 0 && fy1 <= (REAL)(h - 1));
  const long x0 = vx0 ? (long)fx0 : 0, x1 = vx1 ? (long)fx1 : 0;
  const long y0 = vy0 ? (long)fy0 : 0, y1 = vy1 ? (long)fy1 : 0;
  const long plane = (long)h * w;
  for (int c = 0; c < C; ++c) {
    const float* p = img + c * plane;
    REAL acc = 0;
    if (vx0 && vy0) acc += (REAL)p[y0 * w + x0] * w_nw;
    if (vx1 && vy0) acc += (REAL)p[y0 * w + x1] * w_ne;
    if (vx0 && vy1) acc += (REAL)p[y1 * w + x0] * w_sw;
    if (vx1 && vy1) acc += (REAL)p[y1 * w + x1] * w_se;
    out[c] = acc;
  }
}

/* ---- dot-product cost volume -------------------------------------------- */

/* CostVolumeManager.build_cost_volume + forward (cost_volume.py:237-380).
 * planes are addressed planes[b*ps_b + j*ps_d + y*ps_y + x*ps_x] so both the
 * expanded [B,D] view of generate_depth_planes (cost_volume.py:100-136) and a
 * caller-supplied depth_planes_bdhw (cost_volume.py:297-299) are covered.
 * out_mask may be NULL; when given it is the overall_mask rule of
 * FeatureVolumeManager (cost_volume.py:625-637) -- the dot model itself returns
 * None (cost_volume.py:286,335). */
int FN(sr_oracle_dot_volume)(const float* cur /*[B,C,h,w]*/, const float* src /*[B,K,C,h,w]*/,
                             const float* K_src /*[B,K,16]*/, const float* T_src_cur /*[B,K,16]*/,
                             const float* invK_cur /*[B,16]*/, const float* planes, long ps_b,
                             long ps_d, long ps_y, long ps_x, int B, int K, int C, int h, int w,
                             int D, OUT_T* out_cv /*[B,D,h,w]*/, OUT_T* out_lowest /*[B,h,w]*/,
                             unsigned char* out_mask /*[B,h,w] or NULL*/) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return 1;
  const long N = (long)h * w;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < h; ++y) {
      REAL* P = (REAL*)malloc(sizeof(REAL) * 16 * K);
      REAL* warped = (REAL*)malloc(sizeof(REAL) * C);
      for (int k = 0; k < K; ++k)
        FN(mat44_mul)(K_src + ((long)b * K + k) * 16, T_src_cur + ((long)b * K + k) * 16, P + 16 * k);
      for (int x = 0; x < w; ++x) {
        REAL best = 0, best_d = 0;
        for (int j = 0; j < D; ++j) {
          const REAL d = (REAL)planes[b * ps_b + j * ps_d + y * ps_y + x * ps_x];
          REAL cost = 0;
          int any_depth = 0, any_bounds = 0;
          for (int k = 0; k < K; ++k) {
            FN(proj_t) pr;
            FN(backproject_project)(invK_cur + (long)b * 16, P + 16 * k, x, y, d, &pr);
            FN(bilinear_sample)(src + ((long)b * K + k) * C * N, C, h, w, pr.pix_x, pr.pix_y, warped);
            REAL dot = 0; /* cost_volume.py:322-326 */
            for (int c = 0; c < C; ++c) dot += warped[c] * (REAL)cur[((long)b * C + c) * N + (long)y * w + x];
            const REAL m = pr.zp > 0 ? (REAL)1 : (REAL)0; /* cost_volume.py:231-232 */
            cost += dot * m;                               /* cost_volume.py:329 */
            any_depth |= (pr.zp > 0);
            any_bounds |= (pr.pix_x > 2 && pr.pix_x < (REAL)(w - 2) && pr.pix_y > 2 &&
                           pr.pix_y < (REAL)(h - 2)); /* cost_volume.py:77-97 */
          }
          out_cv[((long)b * D + j) * N + (long)y * w + x] = (OUT_T)cost;
          if (j == 0 || cost > best) { best = cost; best_d = d; } /* argmax, first max wins (cost_volume.py:374-378) */
          if (j == D - 1 && out_mask) out_mask[(long)b * N + (long)y * w + x] = (unsigned char)(any_depth && any_bounds);
        }
        if (out_lowest) out_lowest[(long)b * N + (long)y * w + x] = (OUT_T)best_d;
      }
      free(P);
      free(warped);
    }
  return 0;
}

/* Backward of the dot-product volume w.r.t. cur / src features (SURVEY.md §8f "next" #3): what autograd computes
 * through CostVolumeManager.build_cost_volume -- F.grid_sample backward (cost_volume.py:201-212: each bilinear tap
 * receives weight x upstream gradient, out-of-image taps nothing) and the backward of the masked dot product
 * (cost_volume.py:322-329).  Geometry is data.  grad_cv is dense [B,D,h,w].  Serial over pixels inside an image
 * (the scatter into d_src must not race): parallel over the batch only. */
int FN(sr_oracle_dot_volume_bwd)(const float* grad_cv, const float* cur, const float* src, const float* K_src,
                                 const float* T_src_cur, const float* invK_cur, const float* planes, long ps_b,
                                 long ps_d, long ps_y, long ps_x, int B, int K, int C, int h, int w, int D,
                                 OUT_T* d_cur /*[B,C,h,w]*/, OUT_T* d_src /*[B,K,C,h,w]*/) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0) return 1;
  const long N = (long)h * w;
  for (long i = 0; i < (long)B * C * N; ++i) d_cur[i] = 0;
  for (long i = 0; i < (long)B * K * C * N; ++i) d_src[i] = 0;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    REAL* P = (REAL*)malloc(sizeof(REAL) * 16 * K);
    for (int k = 0; k < K; ++k)
      FN(mat44_mul)(K_src + ((long)b * K + k) * 16, T_src_cur + ((long)b * K + k) * 16, P + 16 * k);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const long pix = (long)y * w + x;
        for (int j = 0; j < D; ++j) {
          const REAL d = (REAL)planes[b * ps_b + j * ps_d + y * ps_y + x * ps_x];
          const REAL g = (REAL)grad_cv[((long)b * D + j) * N + pix];
          for (int k = 0; k < K; ++k) {
            FN(proj_t) pr;
            FN(backproject_project)(invK_cur + (long)b * 16, P + 16 * k, x, y, d, &pr);
            if (!(pr.zp > 0)) continue; /* mask_k = 0: no gradient through this view */
            /* the taps of bilinear_sample above */
            const REAL sx = (REAL)(float)(1.0 / (double)w), sy = (REAL)(float)(1.0 / (double)h);
            const REAL u = (REAL)2 * pr.pix_x * sx - (REAL)1, v = (REAL)2 * pr.pix_y * sy - (REAL)1;
            const REAL ix = ((u + (REAL)1) * (REAL)w - (REAL)1) / (REAL)2;
            const REAL iy = ((v + (REAL)1) * (REAL)h - (REAL)1) / (REAL)2;
            const REAL fx0 = FLOOR(ix), fy0 = FLOOR(iy);
            const REAL fx1 = fx0 + 1, fy1 = fy0 + 1;
            const REAL wt[4] = {(fx1 - ix) * (fy1 - iy), (ix - fx0) * (fy1 - iy), (fx1 - ix) * (iy - fy0),
                                (ix - fx0) * (iy - fy0)};
            const REAL tx[4] = {fx0, fx1, fx0, fx1}, ty[4] = {fy0, fy0, fy1, fy1};
            for (int t = 0; t < 4; ++t) {
              if (!(tx[t] >= 0 && tx[t] <= (REAL)(w - 1) && ty[t] >= 0 && ty[t] <= (REAL)(h - 1))) continue;
              const long tap = (long)ty[t] * w + (long)tx[t];
              const REAL gw = g * wt[t];
              for (int c = 0; c < C; ++c) {
                const long sidx = (((long)b * K + k) * C + c) * N + tap;
                const long cidx = ((long)b * C + c) * N + pix;
                d_cur[cidx] += (OUT_T)(gw * (REAL)src[sidx]);
                d_src[sidx] += (OUT_T)(gw * (REAL)cur[cidx]);
              }
            }
          }
        }
      }
    free(P);
  }
  return 0;
}

/* ---- metadata-MLP feature volume ---------------------------------------- */

static inline REAL FN(leaky)(REAL v, REAL slope) { return v > 0 ? v : v * slope; }

/* FeatureVolumeManager.build_cost_volume + forward (cost_volume.py:451-736, 345-380).
 * MLP = Linear, LeakyReLU(0.01), Linear, LeakyReLU(0.01), Linear (networks.py:129-147).
 * pose_feats[b,k,{0,1,2}] = (pose_dist, R_measure, t_measure) from pose_distance
 * (geometry_utils.py:178-191), computed by the caller like cost_volume.py:516-542.
 * Feature order = cost_volume.py:709-723:
 *  [0,KC) warped k*C+c | [KC,KC+C) cur | mask_k | z'_k | d | dot_k | ray_angle_k |
 *  cur_ray(3) | src_ray k*3+i | pose_dist_k | R_measure_k | t_measure_k            */
int FN(sr_oracle_mlp_volume)(const float* cur, const float* src, const float* K_src,
                             const float* T_src_cur, const float* T_cur_src /*[B,K,16]*/,
                             const float* invK_cur, const float* pose_feats /*[B,K,3]*/,
                             const float* planes, long ps_b, long ps_d, long ps_y, long ps_x,
                             const float* W1 /*[H,Cin]*/, const float* b1, const float* W2 /*[H,H]*/,
                             const float* b2, const float* W3 /*[1,H]*/, const float* b3, int B,
                             int K, int C, int h, int w, int D, int Hd, OUT_T* out_cv,
                             OUT_T* out_lowest, unsigned char* out_mask) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0 || Hd <= 0) return 1;
  const long N = (long)h * w;
  const int Cin = C * (K + 1) + 10 * K + 4; /* cost_volume.py:420-435 */
  const REAL slope = (REAL)0.01f;           /* nn.LeakyReLU default (networks.py:139) */
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < h; ++y) {
      REAL* P = (REAL*)malloc(sizeof(REAL) * 16 * K);
      REAL* f = (REAL*)malloc(sizeof(REAL) * Cin);
      REAL* h1 = (REAL*)malloc(sizeof(REAL) * Hd);
      REAL* h2 = (REAL*)malloc(sizeof(REAL) * Hd);
      for (int k = 0; k < K; ++k)
        FN(mat44_mul)(K_src + ((long)b * K + k) * 16, T_src_cur + ((long)b * K + k) * 16, P + 16 * k);
      const int o_cur = K * C, o_mask = o_cur + C, o_z = o_mask + K, o_d = o_z + K, o_dot = o_d + 1;
      const int o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3, o_pd = o_sray + 3 * K;
      const int o_rm = o_pd + K, o_tm = o_rm + K;
      for (int x = 0; x < w; ++x) {
        REAL best = 0, best_d = 0;
        for (int c = 0; c < C; ++c) f[o_cur + c] = (REAL)cur[((long)b * C + c) * N + (long)y * w + x];
        for (int k = 0; k < K; ++k) {
          f[o_pd + k] = (REAL)pose_feats[((long)b * K + k) * 3 + 0];
          f[o_rm + k] = (REAL)pose_feats[((long)b * K + k) * 3 + 1];
          f[o_tm + k] = (REAL)pose_feats[((long)b * K + k) * 3 + 2];
        }
        for (int j = 0; j < D; ++j) {
          const REAL d = (REAL)planes[b * ps_b + j * ps_d + y * ps_y + x * ps_x];
          int any_depth = 0, any_bounds = 0;
          f[o_d] = d;
          for (int k = 0; k < K; ++k) {
            FN(proj_t) pr;
            FN(backproject_project)(invK_cur + (long)b * 16, P + 16 * k, x, y, d, &pr);
            FN(bilinear_sample)(src + ((long)b * K + k) * C * N, C, h, w, pr.pix_x, pr.pix_y, f + k * C);
            REAL dot = 0; /* cost_volume.py:691-695 */
            for (int c = 0; c < C; ++c) dot += f[k * C + c] * f[o_cur + c];
            const REAL m = pr.zp > 0 ? (REAL)1 : (REAL)0;
            f[o_mask + k] = m;
            f[o_z + k] = pr.zp;
            f[o_dot + k] = dot * m;
            /* rays (cost_volume.py:641-669, geometry_utils.py:169-173); F.normalize eps 1e-12 */
            REAL cn = SQRT(pr.X[0] * pr.X[0] + pr.X[1] * pr.X[1] + pr.X[2] * pr.X[2]);
            REAL cden = cn > (REAL)1e-12 ? cn : (REAL)1e-12;
            REAL cr[3], sr[3], sv[3];
            const float* Tcs = T_cur_src + ((long)b * K + k) * 16;
            for (int i = 0; i < 3; ++i) {
              cr[i] = pr.X[i] / cden;
              sv[i] = pr.X[i] - (REAL)Tcs[i * 4 + 3];
            }
            REAL sn = SQRT(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
            REAL sden = sn > (REAL)1e-12 ? sn : (REAL)1e-12;
            for (int i = 0; i < 3; ++i) sr[i] = sv[i] / sden;
            if (k == 0)
              for (int i = 0; i < 3; ++i) f[o_cray + i] = cr[i]; /* cost_volume.py:672-681: view 0's copy */
            for (int i = 0; i < 3; ++i) f[o_sray + 3 * k + i] = sr[i];
            /* F.cosine_similarity(eps=1e-5) (cost_volume.py:683-688), torch>=1.12 form */
            REAL n1 = SQRT(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
            REAL n2 = SQRT(sr[0] * sr[0] + sr[1] * sr[1] + sr[2] * sr[2]);
            n1 = n1 > (REAL)1e-5f ? n1 : (REAL)1e-5f;
            n2 = n2 > (REAL)1e-5f ? n2 : (REAL)1e-5f;
            f[o_ang + k] = (cr[0] / n1) * (sr[0] / n2) + (cr[1] / n1) * (sr[1] / n2) + (cr[2] / n1) * (sr[2] / n2);
            any_depth |= (pr.zp > 0);
            any_bounds |= (pr.pix_x > 2 && pr.pix_x < (REAL)(w - 2) && pr.pix_y > 2 && pr.pix_y < (REAL)(h - 2));
          }
          for (int o = 0; o < Hd; ++o) { /* networks.py:134-139 */
            REAL s = (REAL)b1[o];
            const float* wr = W1 + (long)o * Cin;
            for (int i = 0; i < Cin; ++i) s += (REAL)wr[i] * f[i];
            h1[o] = FN(leaky)(s, slope);
          }
          for (int o = 0; o < Hd; ++o) {
            REAL s = (REAL)b2[o];
            const float* wr = W2 + (long)o * Hd;
            for (int i = 0; i < Hd; ++i) s += (REAL)wr[i] * h1[i];
            h2[o] = FN(leaky)(s, slope);
          }
          REAL cost = (REAL)b3[0]; /* disable_final_activation=True (cost_volume.py:438) */
          for (int i = 0; i < Hd; ++i) cost += (REAL)W3[i] * h2[i];
          out_cv[((long)b * D + j) * N + (long)y * w + x] = (OUT_T)cost;
          if (j == 0 || cost > best) { best = cost; best_d = d; }
          if (j == D - 1 && out_mask) out_mask[(long)b * N + (long)y * w + x] = (unsigned char)(any_depth && any_bounds);
        }
        if (out_lowest) out_lowest[(long)b * N + (long)y * w + x] = (OUT_T)best_d;
      }
      free(P); free(f); free(h1); free(h2);
    }
  return 0;
}

/* Backward of the metadata-MLP volume (SURVEY.md §8f "next" #3) -- what autograd computes through
 * FeatureVolumeManager.build_cost_volume + MLP for grad_cv = dL/d cost_volume (dense [B,D,h,w]):
 *   MLP:  d3 = g;  dh2 = W3^T d3;  dz2 = dh2 * lrelu'(z2);  dh1 = W2^T dz2;  dz1 = dh1 * lrelu'(z1);  df = W1^T dz1
 *         dW3 += g h2^T, db3 += g, dW2 += dz2 h1^T, db2 += dz2, dW1 += dz1 f^T, db1 += dz1      (networks.py:129-147)
 *   features that depend on the matching features (cost_volume.py:691-723):
 *         warped[k,c] = f[kC+c] (unmasked), cur[c] = f[KC+c], dot_k = m_k sum_c warped[k,c] cur[c]
 *     => d_warped[k,c] = df[kC+c] + df[o_dot+k] m_k cur[c];   d_cur[c] += df[KC+c] + sum_k df[o_dot+k] m_k warped[k,c]
 *     => grid_sample backward: d_src[b,k,c,tap_t] += w_t d_warped[k,c] for the in-image taps (cost_volume.py:201-212)
 *   every other channel (mask, z', d, rays, angles, pose measures) is a function of the geometry only: no gradient.
 * Weight gradients are accumulated in REAL per thread and merged at the end; d_src entries with omp atomic. */
int FN(sr_oracle_mlp_volume_bwd)(const float* grad_cv, const float* cur, const float* src, const float* K_src,
                                 const float* T_src_cur, const float* T_cur_src, const float* invK_cur,
                                 const float* pose_feats, const float* planes, long ps_b, long ps_d, long ps_y,
                                 long ps_x, const float* W1, const float* b1, const float* W2, const float* b2,
                                 const float* W3, const float* b3, int B, int K, int C, int h, int w, int D, int Hd,
                                 OUT_T* d_cur, OUT_T* d_src, OUT_T* dW1, OUT_T* db1, OUT_T* dW2, OUT_T* db2,
                                 OUT_T* dW3, OUT_T* db3) {
  if (B < 0 || K <= 0 || C <= 0 || h <= 0 || w <= 0 || D <= 0 || Hd <= 0) return 1;
  (void)b3;
  const long N = (long)h * w;
  const int Cin = C * (K + 1) + 10 * K + 4;
  const REAL slope = (REAL)0.01f;
  const long nW = (long)Hd * Cin + Hd + (long)Hd * Hd + Hd + Hd + 1;   /* dW1 | db1 | dW2 | db2 | dW3 | db3 */
  REAL* total = (REAL*)calloc((size_t)nW, sizeof(REAL));
  for (long i = 0; i < (long)B * C * N; ++i) d_cur[i] = 0;
  for (long i = 0; i < (long)B * K * C * N; ++i) d_src[i] = 0;
#pragma omp parallel
  {
    REAL* acc = (REAL*)calloc((size_t)nW, sizeof(REAL));
    REAL *aW1 = acc, *ab1 = aW1 + (long)Hd * Cin, *aW2 = ab1 + Hd, *ab2 = aW2 + (long)Hd * Hd, *aW3 = ab2 + Hd,
         *ab3 = aW3 + Hd;
    REAL* P = (REAL*)malloc(sizeof(REAL) * 16 * K);
    REAL* f = (REAL*)malloc(sizeof(REAL) * Cin);
    REAL* df = (REAL*)malloc(sizeof(REAL) * Cin);
    REAL* z1 = (REAL*)malloc(sizeof(REAL) * Hd);
    REAL* h1 = (REAL*)malloc(sizeof(REAL) * Hd);
    REAL* z2 = (REAL*)malloc(sizeof(REAL) * Hd);
    REAL* h2 = (REAL*)malloc(sizeof(REAL) * Hd);
    REAL* dz1 = (REAL*)malloc(sizeof(REAL) * Hd);
    REAL* dz2 = (REAL*)malloc(sizeof(REAL) * Hd);
    REAL* tw = (REAL*)malloc(sizeof(REAL) * 4 * K);   /* tap weights, 0 for out-of-image taps */
    long* ti = (long*)malloc(sizeof(long) * 4 * K);   /* tap texel index */
    REAL* mk = (REAL*)malloc(sizeof(REAL) * K);
    REAL* dcur = (REAL*)malloc(sizeof(REAL) * C);
    const int o_cur = K * C, o_mask = o_cur + C, o_z = o_mask + K, o_d = o_z + K, o_dot = o_d + 1;
    const int o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3, o_pd = o_sray + 3 * K;
    const int o_rm = o_pd + K, o_tm = o_rm + K;
#pragma omp for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < h; ++y) {
        for (int k = 0; k < K; ++k)
          FN(mat44_mul)(K_src + ((long)b * K + k) * 16, T_src_cur + ((long)b * K + k) * 16, P + 16 * k);
        for (int x = 0; x < w; ++x) {
          const long pix = (long)y * w + x;
          for (int c = 0; c < C; ++c) { f[o_cur + c] = (REAL)cur[((long)b * C + c) * N + pix]; dcur[c] = 0; }
          for (int k = 0; k < K; ++k) {
            f[o_pd + k] = (REAL)pose_feats[((long)b * K + k) * 3 + 0];
            f[o_rm + k] = (REAL)pose_feats[((long)b * K + k) * 3 + 1];
            f[o_tm + k] = (REAL)pose_feats[((long)b * K + k) * 3 + 2];
          }
          for (int j = 0; j < D; ++j) {
            const REAL d = (REAL)planes[b * ps_b + j * ps_d + y * ps_y + x * ps_x];
            const REAL g = (REAL)grad_cv[((long)b * D + j) * N + pix];
            f[o_d] = d;
            /* ---- forward features, as in sr_oracle_mlp_volume, remembering the taps ---- */
            for (int k = 0; k < K; ++k) {
              FN(proj_t) pr;
              FN(backproject_project)(invK_cur + (long)b * 16, P + 16 * k, x, y, d, &pr);
              const REAL sx = (REAL)(float)(1.0 / (double)w), sy = (REAL)(float)(1.0 / (double)h);
              const REAL u = (REAL)2 * pr.pix_x * sx - (REAL)1, v = (REAL)2 * pr.pix_y * sy - (REAL)1;
              const REAL ix = ((u + (REAL)1) * (REAL)w - (REAL)1) / (REAL)2;
              const REAL iy = ((v + (REAL)1) * (REAL)h - (REAL)1) / (REAL)2;
              const REAL fx0 = FLOOR(ix), fy0 = FLOOR(iy);
              const REAL fx1 = fx0 + 1, fy1 = fy0 + 1;
              const REAL wt[4] = {(fx1 - ix) * (fy1 - iy), (ix - fx0) * (fy1 - iy), (fx1 - ix) * (iy - fy0),
                                  (ix - fx0) * (iy - fy0)};
              const REAL tx[4] = {fx0, fx1, fx0, fx1}, ty[4] = {fy0, fy0, fy1, fy1};
              for (int c = 0; c < C; ++c) f[k * C + c] = 0;
              for (int t = 0; t < 4; ++t) {
                const int ok = (tx[t] >= 0 && tx[t] <= (REAL)(w - 1) && ty[t] >= 0 && ty[t] <= (REAL)(h - 1));
                tw[4 * k + t] = ok ? wt[t] : (REAL)0;
                ti[4 * k + t] = ok ? (long)ty[t] * w + (long)tx[t] : 0;
                if (ok)
                  for (int c = 0; c < C; ++c)
                    f[k * C + c] += (REAL)src[(((long)b * K + k) * C + c) * N + ti[4 * k + t]] * wt[t];
              }
              REAL dot = 0;
              for (int c = 0; c < C; ++c) dot += f[k * C + c] * f[o_cur + c];
              const REAL m = pr.zp > 0 ? (REAL)1 : (REAL)0;
              mk[k] = m;
              f[o_mask + k] = m;
              f[o_z + k] = pr.zp;
              f[o_dot + k] = dot * m;
              REAL cn = SQRT(pr.X[0] * pr.X[0] + pr.X[1] * pr.X[1] + pr.X[2] * pr.X[2]);
              REAL cden = cn > (REAL)1e-12 ? cn : (REAL)1e-12;
              REAL cr[3], sr[3], sv[3];
              const float* Tcs = T_cur_src + ((long)b * K + k) * 16;
              for (int i = 0; i < 3; ++i) { cr[i] = pr.X[i] / cden; sv[i] = pr.X[i] - (REAL)Tcs[i * 4 + 3]; }
              REAL sn = SQRT(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
              REAL sden = sn > (REAL)1e-12 ? sn : (REAL)1e-12;
              for (int i = 0; i < 3; ++i) sr[i] = sv[i] / sden;
              if (k == 0) for (int i = 0; i < 3; ++i) f[o_cray + i] = cr[i];
              for (int i = 0; i < 3; ++i) f[o_sray + 3 * k + i] = sr[i];
              REAL n1 = SQRT(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
              REAL n2 = SQRT(sr[0] * sr[0] + sr[1] * sr[1] + sr[2] * sr[2]);
              n1 = n1 > (REAL)1e-5f ? n1 : (REAL)1e-5f;
              n2 = n2 > (REAL)1e-5f ? n2 : (REAL)1e-5f;
              f[o_ang + k] = (cr[0] / n1) * (sr[0] / n2) + (cr[1] / n1) * (sr[1] / n2) + (cr[2] / n1) * (sr[2] / n2);
            }
            /* ---- MLP forward keeping pre-activations ---- */
            for (int o = 0; o < Hd; ++o) {
              REAL s0 = (REAL)b1[o];
              const float* wr = W1 + (long)o * Cin;
              for (int i = 0; i < Cin; ++i) s0 += (REAL)wr[i] * f[i];
              z1[o] = s0; h1[o] = FN(leaky)(s0, slope);
            }
            for (int o = 0; o < Hd; ++o) {
              REAL s0 = (REAL)b2[o];
              const float* wr = W2 + (long)o * Hd;
              for (int i = 0; i < Hd; ++i) s0 += (REAL)wr[i] * h1[i];
              z2[o] = s0; h2[o] = FN(leaky)(s0, slope);
            }
            /* ---- MLP backward ---- */
            ab3[0] += g;
            for (int i = 0; i < Hd; ++i) {
              aW3[i] += g * h2[i];
              dz2[i] = g * (REAL)W3[i] * (z2[i] > 0 ? (REAL)1 : slope);
              ab2[i] += dz2[i];
            }
            for (int i = 0; i < Hd; ++i) dz1[i] = 0;
            for (int o = 0; o < Hd; ++o) {
              const float* wr = W2 + (long)o * Hd;
              REAL* ar = aW2 + (long)o * Hd;
              for (int i = 0; i < Hd; ++i) { ar[i] += dz2[o] * h1[i]; dz1[i] += (REAL)wr[i] * dz2[o]; }
            }
            for (int i = 0; i < Hd; ++i) { dz1[i] *= (z1[i] > 0 ? (REAL)1 : slope); ab1[i] += dz1[i]; }
            for (int i = 0; i < Cin; ++i) df[i] = 0;
            for (int o = 0; o < Hd; ++o) {
              const float* wr = W1 + (long)o * Cin;
              REAL* ar = aW1 + (long)o * Cin;
              for (int i = 0; i < Cin; ++i) { ar[i] += dz1[o] * f[i]; df[i] += (REAL)wr[i] * dz1[o]; }
            }
            /* ---- features -> matching features ---- */
            for (int c = 0; c < C; ++c) dcur[c] += df[o_cur + c];
            for (int k = 0; k < K; ++k) {
              const REAL ddot = df[o_dot + k] * mk[k];
              for (int c = 0; c < C; ++c) {
                const REAL dwarp = df[k * C + c] + ddot * f[o_cur + c];
                dcur[c] += ddot * f[k * C + c];
                for (int t = 0; t < 4; ++t) {
                  if (tw[4 * k + t] == 0) continue;
                  const OUT_T add = (OUT_T)(tw[4 * k + t] * dwarp);
#pragma omp atomic
                  d_src[(((long)b * K + k) * C + c) * N + ti[4 * k + t]] += add;
                }
              }
            }
          }
          for (int c = 0; c < C; ++c) d_cur[((long)b * C + c) * N + pix] = (OUT_T)dcur[c];
        }
      }
#pragma omp critical
    for (long i = 0; i < nW; ++i) total[i] += acc[i];
    free(acc); free(P); free(f); free(df); free(z1); free(h1); free(z2); free(h2); free(dz1); free(dz2);
    free(tw); free(ti); free(mk); free(dcur);
  }
  {
    const REAL* t = total;
    for (long i = 0; i < (long)Hd * Cin; ++i) dW1[i] = (OUT_T)t[i];
    t += (long)Hd * Cin;
    for (int i = 0; i < Hd; ++i) db1[i] = (OUT_T)t[i];
    t += Hd;
    for (long i = 0; i < (long)Hd * Hd; ++i) dW2[i] = (OUT_T)t[i];
    t += (long)Hd * Hd;
    for (int i = 0; i < Hd; ++i) db2[i] = (OUT_T)t[i];
    t += Hd;
    for (int i = 0; i < Hd; ++i) dW3[i] = (OUT_T)t[i];
    t += Hd;
    db3[0] = (OUT_T)t[0];
  }
  free(total);
  return 0;
}

/* ---- dumps of the MLP input vector (debug/parity aid) -------------------- */

/* Writes the Cin-vector the MLP sees at (b, j, y, x); same code path as above,
 * used by tests to pin the channel ORDER (cost_volume.py:709-723) against the
 * reference's mlp_input tensor. */
int FN(sr_oracle_mlp_input)(const float* cur, const float* src, const float* K_src,
                            const float* T_src_cur, const float* T_cur_src, const float* invK_cur,
                            const float* pose_feats, REAL d, int b, int y, int x, int K, int C,
                            int h, int w, OUT_T* out /*[Cin]*/) {
  const long N = (long)h * w;
  const int o_cur = K * C, o_mask = o_cur + C, o_z = o_mask + K, o_d = o_z + K, o_dot = o_d + 1;
  const int o_ang = o_dot + K, o_cray = o_ang + K, o_sray = o_cray + 3, o_pd = o_sray + 3 * K;
  const int o_rm = o_pd + K, o_tm = o_rm + K;
  REAL P[16], warped[64], curv[64];
  if (C > 64) return 1;
  for (int c = 0; c < C; ++c) {
    curv[c] = (REAL)cur[((long)b * C + c) * N + (long)y * w + x];
    out[o_cur + c] = (OUT_T)curv[c];
  }
  out[o_d] = (OUT_T)d;
  for (int k = 0; k < K; ++k) {
    FN(mat44_mul)(K_src + ((long)b * K + k) * 16, T_src_cur + ((long)b * K + k) * 16, P);
    FN(proj_t) pr;
    FN(backproject_project)(invK_cur + (long)b * 16, P, x, y, d, &pr);
    FN(bilinear_sample)(src + ((long)b * K + k) * C * N, C, h, w, pr.pix_x, pr.pix_y, warped);
    REAL dot = 0;
    for (int c = 0; c < C; ++c) { dot += warped[c] * curv[c]; out[k * C + c] = (OUT_T)warped[c]; }
    const REAL m = pr.zp > 0 ? (REAL)1 : (REAL)0;
    out[o_mask + k] = (OUT_T)m; out[o_z + k] = (OUT_T)pr.zp; out[o_dot + k] = (OUT_T)(dot * m);
    REAL cn = SQRT(pr.X[0] * pr.X[0] + pr.X[1] * pr.X[1] + pr.X[2] * pr.X[2]);
    REAL cden = cn > (REAL)1e-12 ? cn : (REAL)1e-12;
    const float* Tcs = T_cur_src + ((long)b * K + k) * 16;
    REAL cr[3], sv[3], sr[3];
    for (int i = 0; i < 3; ++i) { cr[i] = pr.X[i] / cden; sv[i] = pr.X[i] - (REAL)Tcs[i * 4 + 3]; }
    REAL sn = SQRT(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
    REAL sden = sn > (REAL)1e-12 ? sn : (REAL)1e-12;
    for (int i = 0; i < 3; ++i) sr[i] = sv[i] / sden;
    if (k == 0) for (int i = 0; i < 3; ++i) out[o_cray + i] = (OUT_T)cr[i];
    for (int i = 0; i < 3; ++i) out[o_sray + 3 * k + i] = (OUT_T)sr[i];
    REAL n1 = SQRT(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    REAL n2 = SQRT(sr[0] * sr[0] + sr[1] * sr[1] + sr[2] * sr[2]);
    n1 = n1 > (REAL)1e-5f ? n1 : (REAL)1e-5f; n2 = n2 > (REAL)1e-5f ? n2 : (REAL)1e-5f;
    out[o_ang + k] = (OUT_T)((cr[0] / n1) * (sr[0] / n2) + (cr[1] / n1) * (sr[1] / n2) + (cr[2] / n1) * (sr[2] / n2));
    out[o_pd + k] = (OUT_T)pose_feats[((long)b * K + k) * 3 + 0];
    out[o_rm + k] = (OUT_T)pose_feats[((long)b * K + k) * 3 + 1];
    out[o_tm + k] = (OUT_T)pose_feats[((long)b * K + k) * 3 + 2];
  }
  return 0;
}

/* ---- 2-D conv building blocks ------------------------------------------- */

/* nn.Conv2d(k x k, stride, padding=pad, bias) on NCHW  (modules/layers.py:7-22).
 * Optional fused epilogue used to restate BasicBlock.forward (layers.py:68-85):
 *   out = conv(x) + bias [+ residual] ; if (leaky_slope >= 0) out = LeakyReLU(out)   */
int FN(sr_oracle_conv2d)(const OUT_T* in /*[B,Ci,H,W]*/, const float* wgt /*[Co,Ci,k,k]*/,
                         const float* bias /*[Co] or NULL*/, const OUT_T* residual /*[B,Co,Ho,Wo] or NULL*/,
                         int B, int Ci, int H, int W, int Co, int ksz, int stride, int pad,
                         REAL leaky_slope, OUT_T* out /*[B,Co,Ho,Wo]*/) {
  const int Ho = (H + 2 * pad - ksz) / stride + 1, Wo = (W + 2 * pad - ksz) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Co; ++co) {
      REAL* acc = (REAL*)malloc(sizeof(REAL) * (long)Ho * Wo);
      for (long i = 0; i < (long)Ho * Wo; ++i) acc[i] = bias ? (REAL)bias[co] : (REAL)0;
      for (int ci = 0; ci < Ci; ++ci) {
        const OUT_T* ip = in + ((long)b * Ci + ci) * H * W;
        for (int ky = 0; ky < ksz; ++ky)
          for (int kx = 0; kx < ksz; ++kx) {
            const REAL wv = (REAL)wgt[(((long)co * Ci + ci) * ksz + ky) * ksz + kx];
            for (int oy = 0; oy < Ho; ++oy) {
              const int iy = oy * stride - pad + ky;
              if (iy < 0 || iy >= H) continue;
              int ox0 = 0, ox1 = Wo;
              while (ox0 < Wo && ox0 * stride - pad + kx < 0) ++ox0;
              while (ox1 > ox0 && (ox1 - 1) * stride - pad + kx >= W) --ox1;
              const OUT_T* row = ip + (long)iy * W - pad + kx;
              REAL* arow = acc + (long)oy * Wo;
              for (int ox = ox0; ox < ox1; ++ox) arow[ox] += wv * (REAL)row[ox * stride];
            }
          }
      }
      OUT_T* op = out + ((long)b * Co + co) * Ho * Wo;
      const OUT_T* rp = residual ? residual + ((long)b * Co + co) * Ho * Wo : 0;
      for (long i = 0; i < (long)Ho * Wo; ++i) {
        REAL v = acc[i];
        if (rp) v += (REAL)rp[i];
        if (leaky_slope >= 0) v = FN(leaky)(v, leaky_slope);
        op[i] = (OUT_T)v;
      }
      free(acc);
    }
  return 0;
}

/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False)
 * (utils/generic_utils.py:96-105): src = (dst+0.5)/2 - 0.5 clamped at 0,
 * neighbours clamped to the border (ATen upsample_bilinear2d). */
int FN(sr_oracle_upsample2x)(const OUT_T* in /*[B,C,H,W]*/, int B, int C, int H, int W,
                             OUT_T* out /*[B,C,2H,2W]*/) {
  const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for schedule(static)
  for (long bc = 0; bc < (long)B * C; ++bc) {
    const OUT_T* ip = in + bc * H * W;
    OUT_T* op = out + bc * Ho * Wo;
    for (int oy = 0; oy < Ho; ++oy) {
      REAL sy = ((REAL)oy + (REAL)0.5) * (REAL)0.5 - (REAL)0.5;
      if (sy < 0) sy = 0;
      const int y0 = (int)sy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
      const REAL ly = sy - (REAL)y0, hy = (REAL)1 - ly;
      for (int ox = 0; ox < Wo; ++ox) {
        REAL sx = ((REAL)ox + (REAL)0.5) * (REAL)0.5 - (REAL)0.5;
        if (sx < 0) sx = 0;
        const int x0 = (int)sx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const REAL lx = sx - (REAL)x0, hx = (REAL)1 - lx;
        op[(long)oy * Wo + ox] =
            (OUT_T)(hy * (hx * (REAL)ip[(long)y0 * W + x0] + lx * (REAL)ip[(long)y0 * W + x1]) +
                    ly * (hx * (REAL)ip[(long)y1 * W + x0] + lx * (REAL)ip[(long)y1 * W + x1]));
      }
    }
  }
  return 0;
}

#undef FN
#undef CAT
#undef CAT_
