/*
 * ivx_oracle.c -- CPU restatement (plain C) of the ImVoxelNet forward hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (imvoxelnet_amd/)
 * links, imports or calls this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference repo SamsungLabs/imvoxelnet).  Compile with -ffp-contract=off:
 * where the reference arithmetic is fused (torch.bmm -> FMA chain, measured)
 * the fusion is written explicitly with fmaf().
 *
 * Pinned against: tests/golden/ (fixtures generated from the imported reference by
 * oracle/gen_golden.py) and the reference's own known-answer tests
 * (tests/test_nms.py, tests/test_box3d.py::test_boxes3d_overlaps).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* get_points: mmdet3d/models/detectors/imvoxelnet.py:132-141
 *   new_origin = origin - n_voxels / 2. * voxel_size      (fp32)
 *   points     = idx * voxel_size + new_origin            (fp32 mul, fp32 add)
 * out is [3][X][Y][Z] (z fastest), exactly the reference tensor.            */
void ivxo_get_points(const int64_t *n_voxels, const float *voxel_size,
                     const float *origin, float *out) {
  const int64_t X = n_voxels[0], Y = n_voxels[1], Z = n_voxels[2];
  const int64_t N = X * Y * Z;
  float no[3];
  for (int a = 0; a < 3; ++a) {
    float half = (float)n_voxels[a] / 2.0f;
    float t = half * voxel_size[a];
    no[a] = origin[a] - t;
  }
  for (int64_t i = 0; i < X; ++i)
    for (int64_t j = 0; j < Y; ++j)
      for (int64_t k = 0; k < Z; ++k) {
        int64_t n = (i * Y + j) * Z + k;
        float px = (float)i * voxel_size[0];
        float py = (float)j * voxel_size[1];
        float pz = (float)k * voxel_size[2];
        out[0 * N + n] = px + no[0];
        out[1 * N + n] = py + no[1];
        out[2 * N + n] = pz + no[2];
      }
}

/* ------------------------------------------------------------------------ */
/* _compute_projection: mmdet3d/models/detectors/imvoxelnet.py:114-129
 *   intrinsic = K[:3,:3]; intrinsic[:2] /= ratio; P_v = intrinsic @ E_v[:3]
 * K4 is the 4x4 intrinsic, E4 is V x 4x4, P is V x 3 x 4.  `ratio` is the
 * python float ori_shape[0] / (img_shape[0] / stride); torch divides the
 * fp32 tensor by the scalar cast to fp32.  The 3x3 @ 3x4 product is an FMA
 * chain over k (same as bmm, measured against the reference).              */
void ivxo_compute_projection(const float *K4, const float *E4, int V,
                             double ratio, float *P) {
  float K[9];
  const float r = (float)ratio;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float v = K4[i * 4 + j];
      if (i < 2) v = v / r;
      K[i * 3 + j] = v;
    }
  for (int v = 0; v < V; ++v) {
    const float *E = E4 + (size_t)v * 16;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        float acc = K[i * 3 + 0] * E[0 * 4 + j];
        acc = fmaf(K[i * 3 + 1], E[1 * 4 + j], acc);
        acc = fmaf(K[i * 3 + 2], E[2 * 4 + j], acc);
        P[(size_t)v * 12 + i * 4 + j] = acc;
      }
  }
}

/* ------------------------------------------------------------------------ */
/* Voxel -> pixel index for one view: imvoxelnet.py:149-154
 *   (u,v,w) = P @ (X,Y,Z,1)        torch.bmm == fma chain in k order
 *   x = round_half_even(u / w), y = round_half_even(v / w)  -> int64
 *   valid = x>=0 & y>=0 & x<width & y<height & w>0
 * The float->int64 conversion of inf/NaN/huge values is INT64_MIN on x86
 * (cvttss2si); such voxels fail x>=0 and are invalid.                       */
static inline int64_t f2i64_x86(float f) {
  if (!(f > -9.2233720368547758e18f && f < 9.2233720368547758e18f))
    return INT64_MIN;
  return (int64_t)f;
}

static inline int project_one(const float *P, float X, float Y, float Z,
                              int width, int height, int64_t *xo, int64_t *yo) {
  float u = P[0] * X;
  u = fmaf(P[1], Y, u);
  u = fmaf(P[2], Z, u);
  u = fmaf(P[3], 1.0f, u);
  float v = P[4] * X;
  v = fmaf(P[5], Y, v);
  v = fmaf(P[6], Z, v);
  v = fmaf(P[7], 1.0f, v);
  float w = P[8] * X;
  w = fmaf(P[9], Y, w);
  w = fmaf(P[10], Z, w);
  w = fmaf(P[11], 1.0f, w);
  int64_t x = f2i64_x86(nearbyintf(u / w));
  int64_t y = f2i64_x86(nearbyintf(v / w));
  *xo = x;
  *yo = y;
  return (x >= 0) && (y >= 0) && (x < width) && (y < height) && (w > 0.0f);
}

/* backproject: imvoxelnet.py:145-160.  features [V][C][FH][FW] (NCHW, full
 * map; the reference crops [:, :, :height, :width] before the call, here the
 * crop is expressed by height/width < FH/FW).  points [3][N].
 * Outputs: volume [V][C][N] (zeros where invalid), valid [V][N] (0/1),
 * optional xi/yi [V][N] int64 (may be NULL).                                */
void ivxo_backproject(const float *features, int V, int C, int FH, int FW,
                      int height, int width, const float *points,
                      const float *P, int64_t N, float *volume, uint8_t *valid,
                      int64_t *xi, int64_t *yi) {
  for (int v = 0; v < V; ++v) {
    const float *Pv = P + (size_t)v * 12;
    const float *Fv = features + (size_t)v * C * FH * FW;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      int64_t x, y;
      int ok = project_one(Pv, points[n], points[N + n], points[2 * N + n],
                           width, height, &x, &y);
      valid[(size_t)v * N + n] = (uint8_t)ok;
      if (xi) xi[(size_t)v * N + n] = x;
      if (yi) yi[(size_t)v * N + n] = y;
      for (int c = 0; c < C; ++c)
        volume[((size_t)v * C + c) * N + n] =
            ok ? Fv[((size_t)c * FH + y) * FW + x] : 0.0f;
    }
  }
}

/* backproject + view mean: imvoxelnet.py:69-74
 *   volume = volume.sum(0) ; cnt = valid.sum(0) ; volume = volume / cnt
 *   volume[:, cnt == 0] = 0 ; valid = cnt > 0
 * The view sum is sequential in v (fp32), which is what torch's outer
 * reduction does for V <= a few hundred rows (verified on the fixtures).
 * out [C][N], valid_out [N].                                                */
void ivxo_backproject_mean(const float *features, int V, int C, int FH, int FW,
                           int height, int width, const float *points,
                           const float *P, int64_t N, float *out,
                           uint8_t *valid_out) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    int cnt = 0;
    for (int c = 0; c < C; ++c) out[(size_t)c * N + n] = 0.0f;
    for (int v = 0; v < V; ++v) {
      int64_t x, y;
      int ok = project_one(P + (size_t)v * 12, points[n], points[N + n],
                           points[2 * N + n], width, height, &x, &y);
      if (!ok) continue;
      ++cnt;
      const float *Fv = features + (size_t)v * C * FH * FW;
      for (int c = 0; c < C; ++c)
        out[(size_t)c * N + n] += Fv[((size_t)c * FH + y) * FW + x];
    }
    valid_out[n] = cnt > 0;
    if (cnt > 0) {
      const float d = (float)cnt;
      for (int c = 0; c < C; ++c) out[(size_t)c * N + n] /= d;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Direct convolution (3-D; 2-D is D=1,kd=1).  Restates torch.nn.Conv3d as the
 * reference necks use it (mmdet3d/models/necks/imvoxelnet.py:46-60,99-113,
 * 181-188): cross-correlation, zero padding, NCDHW activations,
 * weights [Co][Ci][kd][kh][kw], optional bias.  Followed by the optional
 * eval-mode BatchNorm affine y*scale+shift, residual add and ReLU so one call
 * restates conv->bn->(+identity)->relu of BasicBlock3d (:209-230).
 * Accumulation order: ci, kd, kh, kw sequential fp32 (no FMA).              */
void ivxo_conv3d(const float *in, int B, int Ci, int D, int H, int W,
                 const float *wgt, int Co, int kd, int kh, int kw, int sd,
                 int sh, int sw, int pd, int ph, int pw, const float *bias,
                 const float *scale, const float *shift, const float *residual,
                 int relu, float *out) {
  const int Do = (D + 2 * pd - kd) / sd + 1;
  const int Ho = (H + 2 * ph - kh) / sh + 1;
  const int Wo = (W + 2 * pw - kw) / sw + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Co; ++co)
      for (int d = 0; d < Do; ++d)
        for (int h = 0; h < Ho; ++h)
          for (int w = 0; w < Wo; ++w) {
            float acc = 0.0f;
            for (int ci = 0; ci < Ci; ++ci)
              for (int a = 0; a < kd; ++a) {
                int id = d * sd - pd + a;
                if (id < 0 || id >= D) continue;
                for (int e = 0; e < kh; ++e) {
                  int ih = h * sh - ph + e;
                  if (ih < 0 || ih >= H) continue;
                  for (int f = 0; f < kw; ++f) {
                    int iw = w * sw - pw + f;
                    if (iw < 0 || iw >= W) continue;
                    float x = in[((((size_t)b * Ci + ci) * D + id) * H + ih) * W + iw];
                    float g = wgt[((((size_t)co * Ci + ci) * kd + a) * kh + e) * kw + f];
                    acc += x * g;
                  }
                }
              }
            if (bias) acc += bias[co];
            if (scale) acc = acc * scale[co] + shift[co];
            size_t o = ((((size_t)b * Co + co) * Do + d) * Ho + h) * Wo + w;
            if (residual) acc += residual[o];
            if (relu && acc < 0.0f) acc = 0.0f;
            out[o] = acc;
          }
}

/* ------------------------------------------------------------------------ */
/* Rotated BEV overlap / IoU: mmdet3d/ops/iou3d/src/iou3d_kernel.cu:16-251.
 * Boxes are (x1, y1, x2, y2, angle) fp32.                                    */
#define IVXO_EPS 1e-8f

typedef struct { float x, y; } pt_t;

static inline float cross2(pt_t a, pt_t b) { return a.x * b.y - a.y * b.x; } /* .cu:36-38 */
static inline float cross3(pt_t p1, pt_t p2, pt_t p0) {                      /* .cu:40-43 */
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static inline pt_t psub(pt_t a, pt_t b) { pt_t r = {a.x - b.x, a.y - b.y}; return r; }

static int check_rect_cross(pt_t p1, pt_t p2, pt_t q1, pt_t q2) {          /* .cu:45-52 */
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) &&
         fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) &&
         fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

static int check_in_box2d(const float *box, pt_t p) {                       /* .cu:54-77 */
  const float MARGIN = 1e-5f;
  float center_x = (box[0] + box[2]) / 2;
  float center_y = (box[1] + box[3]) / 2;
  float angle_cos = cosf(-box[4]), angle_sin = sinf(-box[4]);
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * angle_sin + center_x;
  float rot_y = -(p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos + center_y;
  return (rot_x > box[0] - MARGIN && rot_x < box[2] + MARGIN &&
          rot_y > box[1] - MARGIN && rot_y < box[3] + MARGIN);
}

static int seg_intersection(pt_t p1, pt_t p0, pt_t q1, pt_t q0, pt_t *ans) { /* .cu:79-109 */
  if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > IVXO_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void rotate_around_center(pt_t c, float ac, float as, pt_t *p) {     /* .cu:111-119 */
  float nx = (p->x - c.x) * ac + (p->y - c.y) * as + c.x;
  float ny = -(p->x - c.x) * as + (p->y - c.y) * ac + c.y;
  p->x = nx;
  p->y = ny;
}

static int point_cmp(pt_t a, pt_t b, pt_t c) {                              /* .cu:121-125 */
  return atan2f(a.y - c.y, a.x - c.x) > atan2f(b.y - c.y, b.x - c.x);
}

float ivxo_box_overlap(const float *box_a, const float *box_b) {            /* .cu:127-242 */
  float a_x1 = box_a[0], a_y1 = box_a[1], a_x2 = box_a[2], a_y2 = box_a[3], a_angle = box_a[4];
  float b_x1 = box_b[0], b_y1 = box_b[1], b_x2 = box_b[2], b_y2 = box_b[3], b_angle = box_b[4];
  pt_t center_a = {(a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2};
  pt_t center_b = {(b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2};
  pt_t A[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  pt_t Bc[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_cos = cosf(a_angle), a_sin = sinf(a_angle);
  float b_cos = cosf(b_angle), b_sin = sinf(b_angle);
  for (int k = 0; k < 4; ++k) {
    rotate_around_center(center_a, a_cos, a_sin, &A[k]);
    rotate_around_center(center_b, b_cos, b_sin, &Bc[k]);
  }
  A[4] = A[0];
  Bc[4] = Bc[0];
  pt_t cp[16];
  pt_t pc = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      if (seg_intersection(A[i + 1], A[i], Bc[j + 1], Bc[j], &cp[cnt])) {
        pc.x = pc.x + cp[cnt].x;
        pc.y = pc.y + cp[cnt].y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (check_in_box2d(box_a, Bc[k])) {
      pc.x = pc.x + Bc[k].x;
      pc.y = pc.y + Bc[k].y;
      cp[cnt++] = Bc[k];
    }
    if (check_in_box2d(box_b, A[k])) {
      pc.x = pc.x + A[k].x;
      pc.y = pc.y + A[k].y;
      cp[cnt++] = A[k];
    }
  }
  pc.x /= cnt;
  pc.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (point_cmp(cp[i], cp[i + 1], pc)) {
        pt_t t = cp[i];
        cp[i] = cp[i + 1];
        cp[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; ++k)
    area += cross2(psub(cp[k], cp[0]), psub(cp[k + 1], cp[0]));
  return (float)(fabsf(area) / 2.0);
}

float ivxo_iou_bev(const float *a, const float *b) {                        /* .cu:244-251 */
  float sa = (a[2] - a[0]) * (a[3] - a[1]);
  float sb = (b[2] - b[0]) * (b[3] - b[1]);
  float s_overlap = ivxo_box_overlap(a, b);
  return s_overlap / fmaxf(sa + sb - s_overlap, IVXO_EPS);
}

static float iou_normal(const float *a, const float *b) {                   /* .cu:335-343 */
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0]) * (a[3] - a[1]);
  float Sb = (b[2] - b[0]) * (b[3] - b[1]);
  return interS / fmaxf(Sa + Sb - interS, IVXO_EPS);
}

/* boxes_overlap_bev_gpu: iou3d_kernel.cu:253-266, iou3d.cpp:38-64          */
void ivxo_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = ivxo_box_overlap(a + 5 * i, b + 5 * j);
}

void ivxo_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = ivxo_iou_bev(a + 5 * i, b + 5 * j);
}

/* nms_gpu / nms_normal_gpu on boxes ALREADY sorted by score (descending), as the
 * python wrapper hands them over (iou3d_utils.py:39-47):
 *   mask word (row i, col block c) bit j = iou(i, 64c+j) > thr, only j > i inside
 *   the diagonal block (iou3d_kernel.cu:284-333), then the serial greedy bit
 *   scan (iou3d.cpp:127-143).  Returns the number kept; keep[] gets indices
 *   into the sorted order.                                                   */
static int nms_sorted(const float *boxes, int n, float thr, int64_t *keep, int rotated) {
  if (n <= 0) return 0;
  const int cb = (n + 63) / 64;
  uint64_t *mask = (uint64_t *)calloc((size_t)n * cb, sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < cb; ++c) {
      uint64_t t = 0;
      int col_size = n - c * 64 < 64 ? n - c * 64 : 64;
      int start = (i / 64 == c) ? (i % 64) + 1 : 0;
      for (int j = start; j < col_size; ++j) {
        const float *bj = boxes + 5 * (size_t)(c * 64 + j);
        float v = rotated ? ivxo_iou_bev(boxes + 5 * (size_t)i, bj) : iou_normal(boxes + 5 * (size_t)i, bj);
        if (v > thr) t |= 1ULL << j;
      }
      mask[(size_t)i * cb + c] = t;
    }
  uint64_t *remv = (uint64_t *)calloc(cb, sizeof(uint64_t));
  int num = 0;
  for (int i = 0; i < n; ++i) {
    int nb = i / 64, ib = i % 64;
    if (!(remv[nb] & (1ULL << ib))) {
      keep[num++] = i;
      for (int j = nb; j < cb; ++j) remv[j] |= mask[(size_t)i * cb + j];
    }
  }
  free(remv);
  free(mask);
  return num;
}

int ivxo_nms_rotated_sorted(const float *boxes, int n, float thr, int64_t *keep) {
  return nms_sorted(boxes, n, thr, keep, 1);
}
int ivxo_nms_normal_sorted(const float *boxes, int n, float thr, int64_t *keep) {
  return nms_sorted(boxes, n, thr, keep, 0);
}

/* ------------------------------------------------------------------------ */
/* aligned_3d_nms: mmdet3d/core/post_processing/box3d_nms.py:91-138.
 * `order` is argsort(scores) ascending (the caller passes it so tie order is
 * whatever torch.argsort produced).  pick[] receives box indices in
 * descending-score order; returns count.                                    */
int ivxo_aligned_3d_nms(const float *boxes, const float *scores, const int64_t *classes,
                        const int64_t *order, int n, float thresh, int64_t *pick) {
  (void)scores;
  int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (n > 0 ? n : 1));
  float *area = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
  memcpy(cur, order, sizeof(int64_t) * n);
  for (int i = 0; i < n; ++i) {
    const float *b = boxes + 6 * (size_t)i;
    area[i] = (b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]);
  }
  int len = n, np = 0;
  while (len != 0) {
    int64_t i = cur[len - 1];
    pick[np++] = i;
    const float *bi = boxes + 6 * (size_t)i;
    int m = 0;
    for (int t = 0; t < len - 1; ++t) {
      int64_t j = cur[t];
      const float *bj = boxes + 6 * (size_t)j;
      float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]), zz1 = fmaxf(bi[2], bj[2]);
      float xx2 = fminf(bi[3], bj[3]), yy2 = fminf(bi[4], bj[4]), zz2 = fminf(bi[5], bj[5]);
      float il = fmaxf(0.0f, xx2 - xx1), iw = fmaxf(0.0f, yy2 - yy1), ih = fmaxf(0.0f, zz2 - zz1);
      float inter = il * iw * ih;
      float iou = inter / (area[i] + area[j] - inter);
      iou = iou * (classes[i] == classes[j] ? 1.0f : 0.0f);
      if (iou <= thresh) cur[m++] = j;
    }
    len = m;
  }
  free(cur);
  free(area);
  return np;
}

int ivxo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
