// Error reporting + version for libimvoxel_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/imvoxel.h"

static thread_local char g_err[512] = "";

void ivx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ivx_version(void) { return 410; /* 0.4.1: ivx_bottleneck_fwd_pio, ivx_stem_pool_fwd_pair, the SURVEY 8(b) export names, include/imvoxel_lab.h; 0.4.0: ivx_pair_io / ivx_conv_fwd_pio (chained fp16-pair activations), ivx_model_cfg.trunk_operands; 0.3.1: ivx_conv_desc.wino_operands, IVX_BF16_PAIR / IVX_F16_PAIR, ivx_model_cfg.wino_operands (0.3.0: head / DCNv2 / LayoutHead fields, ivx_model_detect) */ }
extern "C" const char *ivx_last_error(void) { return g_err; }

// SURVEY.md section 8(b) names (include/imvoxel.h, last section): the same entry points under the survey's spelling
extern "C" int ivx_anchor_head_decode(const ivx_anchor_head_desc *d, const float *head_out, const float *anchors, void *workspace, int64_t workspace_bytes,
                                      float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count, int64_t *cand_idx, float *cand_boxes,
                                      float *cand_scores, ivx_stream_t stream) {
  return ivx_anchor_head_get_bboxes(d, head_out, anchors, workspace, workspace_bytes, out_boxes, out_scores, out_labels, out_count, cand_idx, cand_boxes,
                                    cand_scores, stream);
}
extern "C" int ivx_fcos3d_head_decode(const float *head_out, const uint8_t *valid0, const float *level_vs, const float *level_new_origin, float scale,
                                      int32_t B, int32_t nx, int32_t ny, int32_t nz, int32_t CH, int32_t n_classes, int32_t n_reg, int32_t level, int32_t X,
                                      int32_t Y, int32_t Z, int32_t nms_pre, void *workspace, int64_t workspace_bytes, float *cand_boxes,
                                      float *cand_scores, int32_t *cand_count, ivx_stream_t stream) {
  return ivx_fcos_head_level_candidates(head_out, valid0, level_vs, level_new_origin, scale, B, nx, ny, nz, CH, n_classes, n_reg, level, X, Y, Z, nms_pre,
                                        workspace, workspace_bytes, cand_boxes, cand_scores, cand_count, stream);
}
extern "C" int ivx_nms_rotated_bev(const float *boxes_sorted, int32_t n, float thresh, void *workspace, int64_t workspace_bytes, int64_t *keep,
                                   int32_t *num_out, ivx_stream_t stream) {
  return ivx_nms_bev(boxes_sorted, n, thresh, 1, workspace, workspace_bytes, keep, num_out, stream);
}
extern "C" int ivx_nms_aligned3d(const float *boxes, const float *scores, const int64_t *classes, int32_t n, float thresh, int64_t *pick,
                                 int32_t *num_out, ivx_stream_t stream) {
  return ivx_aligned_3d_nms(boxes, scores, classes, n, thresh, pick, num_out, stream);
}
