// Host side of the KITTI AP evaluation: the per-image matching loops that the reference runs under numba.jit
// (mmdet3d/core/evaluation/kitti_utils/eval.py:83-112 image_box_overlap, :161-279 compute_statistics_jit,
// :291-338 fused_compute_statistics).  Plain C++ on host pointers -- no device work here; the rotated BEV / 3-D
// overlaps that feed these loops come from the device kernel (ivx_boxes_overlap_bev).  All matrices are row-major
// double; `ld` is the row pitch in elements.
#include <cmath>
#include <cstdint>
#include <vector>

#include "ivx_common.h"

namespace {

// eval.py:83-112.  criterion -1: IoU, 0: / area(box), 1: / area(query), else: intersection area.
void image_overlap(const double *boxes, int n, const double *q, int k, int criterion, double *out) {
  for (int i = 0; i < n; ++i) {
    const double *b = boxes + 4 * (size_t)i;
    const double barea = (b[2] - b[0]) * (b[3] - b[1]);
    for (int j = 0; j < k; ++j) {
      const double *c = q + 4 * (size_t)j;
      double v = 0.0;
      const double iw = std::fmin(b[2], c[2]) - std::fmax(b[0], c[0]);
      if (iw > 0) {
        const double ih = std::fmin(b[3], c[3]) - std::fmax(b[1], c[1]);
        if (ih > 0) {
          const double qarea = (c[2] - c[0]) * (c[3] - c[1]);
          double ua = 1.0;
          if (criterion == -1) ua = barea + qarea - iw * ih;
          else if (criterion == 0) ua = barea;
          else if (criterion == 1) ua = qarea;
          v = iw * ih / ua;
        }
      }
      out[(size_t)i * k + j] = v;
    }
  }
}

struct Stats {
  long long tp, fp, fn;
  double similarity;
};

// eval.py:161-279.  overlaps[j*ld + i]: detection j vs ground truth i.  gt rows: x1,y1,x2,y2,alpha; dt rows: + score.
Stats match_image(const double *overlaps, int ld, const double *gt, int n_gt, const double *dt, int n_dt,
                  const int64_t *ign_gt, const int64_t *ign_dt, const double *dc, int n_dc, int metric, double min_overlap,
                  double thresh, bool compute_fp, bool compute_aos, double *thr_out, int *n_thr_out) {
  const double NO_DETECTION = -10000000.0;
  std::vector<char> assigned(n_dt, 0), below(n_dt, 0);
  if (compute_fp)
    for (int j = 0; j < n_dt; ++j) below[j] = dt[6 * (size_t)j + 5] < thresh;
  Stats s = {0, 0, 0, 0.0};
  std::vector<double> delta;
  int n_thr = 0;
  for (int i = 0; i < n_gt; ++i) {
    if (ign_gt[i] == -1) continue;
    int det = -1;
    double valid = NO_DETECTION, max_overlap = 0.0;
    bool assigned_ignored = false;
    for (int j = 0; j < n_dt; ++j) {
      if (ign_dt[j] == -1 || assigned[j] || below[j]) continue;
      const double ov = overlaps[(size_t)j * ld + i];
      const double score = dt[6 * (size_t)j + 5];
      if (!compute_fp && ov > min_overlap && score > valid) {
        det = j;
        valid = score;
      } else if (compute_fp && ov > min_overlap && (ov > max_overlap || assigned_ignored) && ign_dt[j] == 0) {
        max_overlap = ov;
        det = j;
        valid = 1;
        assigned_ignored = false;
      } else if (compute_fp && ov > min_overlap && valid == NO_DETECTION && ign_dt[j] == 1) {
        det = j;
        valid = 1;
        assigned_ignored = true;
      }
    }
    if (valid == NO_DETECTION && ign_gt[i] == 0) {
      ++s.fn;
    } else if (valid != NO_DETECTION && (ign_gt[i] == 1 || ign_dt[det] == 1)) {
      assigned[det] = 1;
    } else if (valid != NO_DETECTION) {
      ++s.tp;
      if (thr_out) thr_out[n_thr] = dt[6 * (size_t)det + 5];
      ++n_thr;
      if (compute_aos) delta.push_back(gt[5 * (size_t)i + 4] - dt[6 * (size_t)det + 4]);
      assigned[det] = 1;
    }
  }
  if (compute_fp) {
    for (int j = 0; j < n_dt; ++j)
      if (!(assigned[j] || ign_dt[j] == -1 || ign_dt[j] == 1 || below[j])) ++s.fp;
    long long nstuff = 0;
    if (metric == 0 && n_dc > 0 && n_dt > 0) {
      // detections falling on DontCare regions are not false positives (2-D metric only): overlap / area(detection)
      std::vector<double> dtb(4 * (size_t)n_dt), odc((size_t)n_dt * n_dc);
      for (int j = 0; j < n_dt; ++j)
        for (int q = 0; q < 4; ++q) dtb[4 * (size_t)j + q] = dt[6 * (size_t)j + q];
      image_overlap(dtb.data(), n_dt, dc, n_dc, 0, odc.data());
      for (int i = 0; i < n_dc; ++i)
        for (int j = 0; j < n_dt; ++j) {
          if (assigned[j] || ign_dt[j] == -1 || ign_dt[j] == 1 || below[j]) continue;
          if (odc[(size_t)j * n_dc + i] > min_overlap) {
            assigned[j] = 1;
            ++nstuff;
          }
        }
    }
    s.fp -= nstuff;
    if (compute_aos) {
      double sum = 0.0;   // the reference sums `fp` zeros followed by the tp terms
      for (double d : delta) sum += (1.0 + std::cos(d)) / 2.0;
      s.similarity = (s.tp > 0 || s.fp > 0) ? sum : -1.0;
    }
  }
  if (n_thr_out) *n_thr_out = n_thr;
  return s;
}

}  // namespace

extern "C" int ivx_kitti_image_box_overlap(const double *boxes, int32_t n, const double *query, int32_t k, int32_t criterion,
                                           double *out) {
  IVX_REQUIRE(n >= 0 && k >= 0, "ivx_kitti_image_box_overlap: negative size");
  if (n == 0 || k == 0) return IVX_OK;
  IVX_REQUIRE(boxes && query && out, "ivx_kitti_image_box_overlap: null argument");
  image_overlap(boxes, n, query, k, criterion, out);
  return IVX_OK;
}

extern "C" int ivx_kitti_compute_statistics(const double *overlaps, int32_t ld, const double *gt_datas, int32_t n_gt,
                                            const double *dt_datas, int32_t n_dt, const int64_t *ignored_gt,
                                            const int64_t *ignored_det, const double *dc_bboxes, int32_t n_dc, int32_t metric,
                                            double min_overlap, double thresh, int32_t compute_fp, int32_t compute_aos,
                                            double *stats4, double *thresholds, int32_t *n_thresholds) {
  IVX_REQUIRE(n_gt >= 0 && n_dt >= 0 && n_dc >= 0 && ld >= n_gt, "ivx_kitti_compute_statistics: bad sizes");
  IVX_REQUIRE(stats4, "ivx_kitti_compute_statistics: null stats");
  IVX_REQUIRE((n_gt == 0 || (gt_datas && ignored_gt)) && (n_dt == 0 || (dt_datas && ignored_det)) &&
                  (n_gt == 0 || n_dt == 0 || overlaps) && (n_dc == 0 || dc_bboxes),
              "ivx_kitti_compute_statistics: null argument");
  int nt = 0;
  const Stats s = match_image(overlaps, ld, gt_datas, n_gt, dt_datas, n_dt, ignored_gt, ignored_det, dc_bboxes, n_dc, metric,
                              min_overlap, thresh, compute_fp != 0, compute_aos != 0, thresholds, &nt);
  stats4[0] = (double)s.tp; stats4[1] = (double)s.fp; stats4[2] = (double)s.fn; stats4[3] = s.similarity;
  if (n_thresholds) *n_thresholds = nt;
  return IVX_OK;
}

// eval.py:291-338 over a list of images whose rows are concatenated: image i owns gt rows [sum gt_nums[:i], +gt_nums[i]),
// dt rows likewise, and its own overlap matrix ov_ptrs[i] ([dt_nums[i], gt_nums[i]], pitch gt_nums[i]).
// pr[t*4 + {0,1,2,3}] += tp, fp, fn, similarity for every score threshold t.
extern "C" int ivx_kitti_fused_statistics(const double *const *ov_ptrs, int32_t n_img, const int32_t *gt_nums,
                                          const int32_t *dt_nums, const int32_t *dc_nums, const double *gt_datas,
                                          const double *dt_datas, const double *dontcares, const int64_t *ignored_gts,
                                          const int64_t *ignored_dets, int32_t metric, double min_overlap,
                                          const double *thresholds, int32_t n_thr, int32_t compute_aos, double *pr) {
  IVX_REQUIRE(n_img >= 0 && n_thr >= 0, "ivx_kitti_fused_statistics: negative size");
  if (n_img == 0 || n_thr == 0) return IVX_OK;
  IVX_REQUIRE(ov_ptrs && gt_nums && dt_nums && dc_nums && thresholds && pr, "ivx_kitti_fused_statistics: null argument");
  size_t g = 0, d = 0, c = 0;
  for (int i = 0; i < n_img; ++i) {
    IVX_REQUIRE(gt_nums[i] >= 0 && dt_nums[i] >= 0 && dc_nums[i] >= 0, "ivx_kitti_fused_statistics: negative count");
    for (int t = 0; t < n_thr; ++t) {
      const Stats s = match_image(ov_ptrs[i], gt_nums[i], gt_datas + 5 * g, gt_nums[i], dt_datas + 6 * d, dt_nums[i],
                                  ignored_gts + g, ignored_dets + d, dontcares + 4 * c, dc_nums[i], metric, min_overlap,
                                  thresholds[t], true, compute_aos != 0, nullptr, nullptr);
      pr[4 * (size_t)t + 0] += (double)s.tp;
      pr[4 * (size_t)t + 1] += (double)s.fp;
      pr[4 * (size_t)t + 2] += (double)s.fn;
      if (s.similarity != -1.0) pr[4 * (size_t)t + 3] += s.similarity;
    }
    g += gt_nums[i]; d += dt_nums[i]; c += dc_nums[i];
  }
  return IVX_OK;
}

// First pass of eval_class (eval.py:500-514): per image, match with compute_fp = false / thresh = 0 and append the
// scores of the matched detections.  `scores_out` needs sum(gt_nums) entries; *n_out receives the count.
extern "C" int ivx_kitti_collect_scores(const double *const *ov_ptrs, int32_t n_img, const int32_t *gt_nums,
                                        const int32_t *dt_nums, const double *gt_datas, const double *dt_datas,
                                        const int64_t *ignored_gts, const int64_t *ignored_dets, int32_t metric,
                                        double min_overlap, double *scores_out, int64_t *n_out) {
  IVX_REQUIRE(n_img >= 0 && n_out, "ivx_kitti_collect_scores: bad argument");
  *n_out = 0;
  if (n_img == 0) return IVX_OK;
  IVX_REQUIRE(ov_ptrs && gt_nums && dt_nums && scores_out, "ivx_kitti_collect_scores: null argument");
  size_t g = 0, d = 0;
  int64_t total = 0;
  for (int i = 0; i < n_img; ++i) {
    IVX_REQUIRE(gt_nums[i] >= 0 && dt_nums[i] >= 0, "ivx_kitti_collect_scores: negative count");
    int nt = 0;
    match_image(ov_ptrs[i], gt_nums[i], gt_datas + 5 * g, gt_nums[i], dt_datas + 6 * d, dt_nums[i], ignored_gts + g,
                ignored_dets + d, nullptr, 0, metric, min_overlap, 0.0, false, false, scores_out + total, &nt);
    total += nt;
    g += gt_nums[i]; d += dt_nums[i];
  }
  *n_out = total;
  return IVX_OK;
}
