/* A host WITHOUT Python for the INDOOR families: runs the reference's end-to-end golden cases (tests/golden/e2e_indoor.bin, re-encoded
 * from e2e_indoor.npz, which oracle/gen_golden.py generated from the imported reference ImVoxelNet.simple_test) through
 * ivx_model_detect of the model-level C-ABI:
 *     FPN level-0 maps -> host camera set-up (inside the library) -> multi-view unprojection -> FastIndoorImVoxelNeck ->
 *     ScanNetImVoxelHeadV2 / SunRgbdImVoxelHeadV2 (fused head conv, per-level candidates, cross-level aligned / rotated multi-class NMS)
 * and compares the detections and the valid mask with the reference's outputs.  Plain C11 + the HIP runtime C API; the same source
 * is built against oracle/_cpuabi/libimvoxel_cpu.so (the CPU restatement of the ABI) for the CPU test suite.
 *
 *   run:   tests/c/e2e_indoor tests/golden/e2e_indoor.bin [bf16]
 * With `bf16` the handle is created with ivx_model_cfg.storage = IVX_BF16 (the optional reduced-precision mode BASELINE config 5 names,
 * inside the C-ABI): the FPN maps are handed over as bf16, every activation and weight of the neck / head convolutions is bf16 (fp32
 * accumulate), the head outputs and the tails fp32.  The comparison with the fp32 reference is then a loose one: identical valid
 * mask, and at least 80 % of the reference's detections found (same label, box within 0.05 m / rad, score within 0.05).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/imvoxel.h"

typedef struct {
  char name[160];
  int dtype, ndim;
  int64_t shape[6], numel;
  void *data;
} entry_t;

static entry_t *g_ent;
static int g_n;
static int g_bf16;      /* argv[2] == "bf16" */

static uint16_t to_bf16(float f) {      /* round to nearest even */
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
  x += 0x7fffu + ((x >> 16) & 1u);
  return (uint16_t)(x >> 16);
}

static int load_fixture(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); return -1; }
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "IVXF0001", 8)) { fprintf(stderr, "bad magic\n"); return -1; }
  int32_t n;
  if (fread(&n, 4, 1, f) != 1) return -1;
  g_ent = (entry_t *)calloc((size_t)n, sizeof(entry_t));
  g_n = n;
  for (int i = 0; i < n; ++i) {
    entry_t *e = &g_ent[i];
    int32_t len, hdr[2];
    if (fread(&len, 4, 1, f) != 1 || len <= 0 || len >= (int)sizeof(e->name)) return -1;
    if (fread(e->name, 1, (size_t)len, f) != (size_t)len || fread(hdr, 4, 2, f) != 2) return -1;
    e->dtype = hdr[0]; e->ndim = hdr[1];
    e->numel = 1;
    for (int d = 0; d < e->ndim; ++d) {
      if (fread(&e->shape[d], 8, 1, f) != 1) return -1;
      e->numel *= e->shape[d];
    }
    const size_t esz = e->dtype == 0 ? 4 : e->dtype == 1 ? 8 : 1;
    e->data = malloc((size_t)e->numel * esz + 8);
    if (fread(e->data, esz, (size_t)e->numel, f) != (size_t)e->numel) return -1;
  }
  fclose(f);
  return 0;
}

static const entry_t *get2(const char *pre, const char *name) {
  char key[200];
  snprintf(key, sizeof(key), "%s%s", pre, name);
  for (int i = 0; i < g_n; ++i)
    if (!strcmp(g_ent[i].name, key)) return &g_ent[i];
  fprintf(stderr, "fixture entry %s missing\n", key);
  exit(2);
}
static float getf(const char *pre, const char *name) { return *(const float *)get2(pre, name)->data; }

#define CK(call)                                                                                   \
  do {                                                                                             \
    int rc_ = (call);                                                                              \
    if (rc_ != IVX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ivx_last_error()); return 1; } \
  } while (0)
#define HK(call)                                                                             \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
  } while (0)

static int run_case(const char *pre, int head_type) {
  const int B = 2;
  const entry_t *fpn0 = get2(pre, "fpn0"), *nv = get2(pre, "n_voxels"), *vs = get2(pre, "voxel_size");
  const int V = (int)(fpn0->shape[0] / B), Cf = (int)fpn0->shape[1], FH = (int)fpn0->shape[2], FW = (int)fpn0->shape[3], H = FH * 4, W = FW * 4;

  ivx_model_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.neck_type = IVX_NECK_FAST;
  cfg.with_trunk = 0;                       /* the fixture starts at the FPN level-0 maps */
  cfg.fpn_channels = Cf;
  cfg.neck_out_channels = (int)get2(pre, "sd::bbox_head.cls_conv.weight")->shape[1];
  cfg.fast_n_blocks[0] = cfg.fast_n_blocks[1] = cfg.fast_n_blocks[2] = 1;
  for (int a = 0; a < 3; ++a) { cfg.n_voxels[a] = (int)((int64_t *)nv->data)[a]; cfg.voxel_size[a] = ((float *)vs->data)[a]; }
  cfg.head_type = head_type;
  cfg.head_classes = (int)getf(pre, "head_kw::n_classes");
  cfg.head_nms_pre = (int)getf(pre, "test_cfg::nms_pre");
  cfg.head_score_thr = getf(pre, "test_cfg::score_thr");
  if (head_type == IVX_HEAD_SCANNET) {
    cfg.head_nms_thr = getf(pre, "test_cfg::iou_thr");
  } else {
    cfg.head_nms_thr = getf(pre, "test_cfg::nms_thr");
    cfg.head_use_rotate_nms = (int)getf(pre, "test_cfg::use_rotate_nms");
  }
  cfg.winograd = 1; cfg.winograd_tile = 0;
  cfg.storage = g_bf16 ? IVX_BF16 : IVX_F32;

  ivx_model *m = NULL;
  CK(ivx_create(&cfg, &m));
  int loaded = 0;
  char sdp[64];
  snprintf(sdp, sizeof(sdp), "%ssd::", pre);
  const size_t lp = strlen(sdp);
  for (int i = 0; i < g_n; ++i) {
    const entry_t *e = &g_ent[i];
    if (strncmp(e->name, sdp, lp) || !strncmp(e->name + lp, "backbone.", 9) || e->dtype != 0) continue;
    CK(ivx_weights_load(m, e->name + lp, (const float *)e->data, e->shape, e->ndim));
    ++loaded;
  }
  CK(ivx_weights_finalize(m, NULL));

  /* FPN maps: the fixture holds [B*V,C,h,w]; the library takes channels-last [B*V,1,h,w,C] */
  const int BV = B * V;
  const size_t n_map = (size_t)BV * FH * FW * Cf;
  float *maps = (float *)malloc(n_map * 4);
  for (int b = 0; b < BV; ++b)
    for (int c = 0; c < Cf; ++c)
      for (int y = 0; y < FH; ++y)
        for (int x = 0; x < FW; ++x)
          maps[(((size_t)b * FH + y) * FW + x) * Cf + c] = ((float *)fpn0->data)[(((size_t)b * Cf + c) * FH + y) * FW + x];

  /* the fields of img_meta the path reads; the library computes projections, origins and crops from them */
  ivx_sample_meta metas[2];
  memset(metas, 0, sizeof(metas));
  for (int b = 0; b < B; ++b) {
    char k[64];
    snprintf(k, sizeof(k), "meta%d::img_shape", b);
    const int64_t *ishape = (const int64_t *)get2(pre, k)->data;
    snprintf(k, sizeof(k), "meta%d::ori_shape", b);
    const int64_t *oshape = (const int64_t *)get2(pre, k)->data;
    snprintf(k, sizeof(k), "meta%d::intrinsic", b);
    memcpy(metas[b].intrinsic, get2(pre, k)->data, 16 * sizeof(float));
    snprintf(k, sizeof(k), "meta%d::extrinsic", b);
    metas[b].extrinsics = (const float *)get2(pre, k)->data;
    snprintf(k, sizeof(k), "meta%d::origin", b);
    memcpy(metas[b].origin, get2(pre, k)->data, 3 * sizeof(float));
    metas[b].img_h = (int32_t)ishape[0]; metas[b].img_w = (int32_t)ishape[1]; metas[b].ori_h = (int32_t)oshape[0];
  }

  const int64_t ws_bytes = ivx_model_detect_workspace_bytes(m, B, V, H, W);
  const int M = ivx_model_max_detections(m, B, V, H, W);
  if (ws_bytes < 0 || M <= 0) { fprintf(stderr, "planning failed: %s\n", ivx_last_error()); return 1; }
  const int NV = cfg.n_voxels[0] * cfg.n_voxels[1] * cfg.n_voxels[2];
  float *d_maps, *d_boxes, *d_scores;
  int32_t *d_count;
  int64_t *d_labels;
  uint8_t *d_valid;
  void *d_ws;
  HK(hipMalloc((void **)&d_maps, n_map * 4));
  HK(hipMalloc((void **)&d_boxes, (size_t)B * M * 7 * 4));
  HK(hipMalloc((void **)&d_scores, (size_t)B * M * 4));
  HK(hipMalloc((void **)&d_labels, (size_t)B * M * 8));
  HK(hipMalloc((void **)&d_count, (size_t)B * 4));
  HK(hipMalloc((void **)&d_valid, (size_t)B * NV));
  HK(hipMalloc(&d_ws, (size_t)ws_bytes));
  if (g_bf16) {      /* the sub-path tensors of a bf16 handle are bf16: convert the maps on the host */
    uint16_t *mb = (uint16_t *)malloc(n_map * 2);
    for (size_t i = 0; i < n_map; ++i) mb[i] = to_bf16(maps[i]);
    HK(hipMemcpy(d_maps, mb, n_map * 2, hipMemcpyHostToDevice));
    free(mb);
  } else {
    HK(hipMemcpy(d_maps, maps, n_map * 4, hipMemcpyHostToDevice));
  }

  CK(ivx_model_detect(m, d_maps, B, V, H, W, metas, d_ws, ws_bytes, d_boxes, d_scores, d_labels, d_count, d_valid, NULL, NULL, NULL));
  HK(hipDeviceSynchronize());

  float *boxes = (float *)malloc((size_t)B * M * 7 * 4), *scores = (float *)malloc((size_t)B * M * 4);
  int64_t *labels = (int64_t *)malloc((size_t)B * M * 8);
  int32_t count[2];
  uint8_t *valid = (uint8_t *)malloc((size_t)B * NV);
  HK(hipMemcpy(boxes, d_boxes, (size_t)B * M * 7 * 4, hipMemcpyDeviceToHost));
  HK(hipMemcpy(scores, d_scores, (size_t)B * M * 4, hipMemcpyDeviceToHost));
  HK(hipMemcpy(labels, d_labels, (size_t)B * M * 8, hipMemcpyDeviceToHost));
  HK(hipMemcpy(count, d_count, sizeof(count), hipMemcpyDeviceToHost));
  HK(hipMemcpy(valid, d_valid, (size_t)B * NV, hipMemcpyDeviceToHost));

  int bad = 0;
  const entry_t *rv = get2(pre, "valids");
  if (memcmp(valid, rv->data, (size_t)B * NV)) { fprintf(stderr, "%s valid mask differs from the reference\n", pre); ++bad; }
  for (int b = 0; b < B; ++b) {
    char k[64];
    snprintf(k, sizeof(k), "res%d::scores", b);
    const entry_t *rs = get2(pre, k);
    snprintf(k, sizeof(k), "res%d::boxes", b);
    const entry_t *rb = get2(pre, k);
    snprintf(k, sizeof(k), "res%d::labels", b);
    const entry_t *rl = get2(pre, k);
    if (g_bf16) {      /* reduced precision: how many of the reference's detections are found */
      int found = 0;
      for (int j = 0; j < (int)rs->numel; ++j) {
        int hit = 0;
        for (int i = 0; i < count[b] && !hit; ++i) {
          if (labels[b * M + i] != ((int64_t *)rl->data)[j]) continue;
          float d = 0.f;
          for (int c = 0; c < 7; ++c) d = fmaxf(d, fabsf(boxes[((size_t)b * M + i) * 7 + c] - ((float *)rb->data)[j * 7 + c]));
          hit = d <= 0.05f && fabsf(scores[b * M + i] - ((float *)rs->data)[j]) <= 0.05f;
        }
        found += hit;
      }
      printf("%s sample %d (bf16 storage): %d detections, reference %lld, %d of them found\n", pre, b, count[b], (long long)rs->numel, found);
      if (found * 5 < (int)rs->numel * 4) { fprintf(stderr, "%s sample %d: fewer than 80 %% of the reference's detections found\n", pre, b); ++bad; }
      continue;
    }
    if (count[b] != (int)rs->numel) { fprintf(stderr, "%s sample %d: %d detections, reference %lld\n", pre, b, count[b], (long long)rs->numel); ++bad; continue; }
    /* rows are paired by (label, box): two fp32 summation orders may order scores that agree to ~1e-6 differently */
    float ds = 0.f, db = 0.f;
    for (int i = 0; i < count[b]; ++i) {
      int best = -1;
      float bd = 1e30f;
      for (int j = 0; j < count[b]; ++j) {
        if (labels[b * M + i] != ((int64_t *)rl->data)[j]) continue;
        float d = 0.f;
        for (int c = 0; c < 7; ++c) d = fmaxf(d, fabsf(boxes[((size_t)b * M + i) * 7 + c] - ((float *)rb->data)[j * 7 + c]));
        if (d < bd) { bd = d; best = j; }
      }
      if (best < 0) { fprintf(stderr, "%s sample %d det %d: no reference detection with its label\n", pre, b, i); ++bad; continue; }
      db = fmaxf(db, bd);
      ds = fmaxf(ds, fabsf(scores[b * M + i] - ((float *)rs->data)[best]));
    }
    printf("%s sample %d: %d detections, max |dscore| %.2e, max |dbox| %.2e\n", pre, b, count[b], ds, db);
    if (ds > 1e-5f || db > 1e-3f) { fprintf(stderr, "%s sample %d: out of tolerance (1e-5 scores / 1e-3 boxes)\n", pre, b); ++bad; }
  }
  CK(ivx_destroy(m));
  (void)hipFree(d_maps); (void)hipFree(d_boxes); (void)hipFree(d_scores); (void)hipFree(d_labels); (void)hipFree(d_count); (void)hipFree(d_valid); (void)hipFree(d_ws);
  printf("%s %d weight tensors loaded, %d views, workspace %lld bytes, up to %d detections per sample\n", pre, loaded, V, (long long)ws_bytes, M);
  return bad;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s e2e_indoor.bin\n", argv[0]); return 2; }
  if (load_fixture(argv[1])) return 2;
  g_bf16 = argc > 2 && !strcmp(argv[2], "bf16");
  int bad = run_case("scannet::", IVX_HEAD_SCANNET);
  bad += run_case("sunrgbd::", IVX_HEAD_SUNRGBD);
  if (bad) { printf("C e2e_indoor FAILED (%d problems)\n", bad); return 1; }
  printf("C e2e_indoor OK%s: ScanNet + SUN RGB-D families end to end, no Python involved\n", g_bf16 ? " (bf16 storage)" : "");
  return 0;
}
