/* A host WITHOUT Python: runs the reference's end-to-end golden case (tests/golden/e2e_small.bin, re-encoded from
 * e2e_small.npz, which oracle/gen_golden.py generated from the imported reference ImVoxelNet.simple_test) through the
 * model-level C-ABI of libimvoxel_hip.so:
 *     FPN level-0 maps -> unprojection -> KittiImVoxelNeck -> Anchor3DHead -> decode + rotated NMS
 * and compares the detections and the valid mask with the reference's outputs.  Plain C11 + the HIP runtime C API.
 *
 *   build: gcc -std=c11 -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/c/e2e_small.c -o tests/c/e2e_small \
 *              -Limvoxelnet_amd/csrc -limvoxel_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,...   (tests/c/build.py)
 *   run:   tests/c/e2e_small tests/golden/e2e_small.bin
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/imvoxel.h"

typedef struct {
  char name[128];
  int dtype, ndim;
  int64_t shape[6], numel;
  void *data;
} entry_t;

static entry_t *g_ent;
static int g_n;

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

static const entry_t *get(const char *name) {
  for (int i = 0; i < g_n; ++i)
    if (!strcmp(g_ent[i].name, name)) return &g_ent[i];
  fprintf(stderr, "fixture entry %s missing\n", name);
  exit(2);
}

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

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s e2e_small.bin\n", argv[0]); return 2; }
  if (load_fixture(argv[1])) return 2;
  const int B = 2, V = 1;
  const entry_t *fpn0 = get("fpn0"), *nv = get("n_voxels"), *vs = get("voxel_size"), *rg = get("ranges");
  const int Cf = (int)fpn0->shape[1], FH = (int)fpn0->shape[2], FW = (int)fpn0->shape[3], H = FH * 4, W = FW * 4;

  ivx_model_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.neck_type = IVX_NECK_KITTI;
  cfg.with_trunk = 0;                       /* the fixture starts at the FPN level-0 maps */
  cfg.fpn_channels = Cf;
  cfg.neck_out_channels = (int)get("sd::bbox_head.conv_cls.weight")->shape[1];
  for (int a = 0; a < 3; ++a) { cfg.n_voxels[a] = (int)((int64_t *)nv->data)[a]; cfg.voxel_size[a] = ((float *)vs->data)[a]; }
  cfg.num_classes = 1; cfg.n_sizes = 1; cfg.n_rotations = 2;
  for (int a = 0; a < 6; ++a) cfg.anchor_range[a] = ((float *)rg->data)[a];
  cfg.anchor_sizes[0] = 1.6f; cfg.anchor_sizes[1] = 3.9f; cfg.anchor_sizes[2] = 1.56f;
  cfg.anchor_rotations[0] = 0.f; cfg.anchor_rotations[1] = 1.57f;
  cfg.nms_pre = (int)*(float *)get("test_cfg::nms_pre")->data;
  cfg.max_num = (int)*(float *)get("test_cfg::max_num")->data;
  cfg.use_rotate_nms = (int)*(float *)get("test_cfg::use_rotate_nms")->data;
  cfg.score_thr = *(float *)get("test_cfg::score_thr")->data;
  cfg.nms_thr = *(float *)get("test_cfg::nms_thr")->data;
  cfg.dir_offset = 0.f; cfg.dir_limit_offset = 1.f;
  cfg.winograd = 1; cfg.winograd_tile = 0;

  ivx_model *m = NULL;
  CK(ivx_create(&cfg, &m));
  int loaded = 0;
  for (int i = 0; i < g_n; ++i) {
    const entry_t *e = &g_ent[i];
    if (strncmp(e->name, "sd::", 4) || !strncmp(e->name, "sd::backbone.", 13) || e->dtype != 0) continue;
    CK(ivx_weights_load(m, e->name + 4, (const float *)e->data, e->shape, e->ndim));
    ++loaded;
  }
  CK(ivx_weights_finalize(m, NULL));

  /* FPN maps: the fixture holds [B,C,h,w]; the library takes channels-last [B*V,1,h,w,C] */
  const size_t n_map = (size_t)B * FH * FW * Cf;
  float *maps = (float *)malloc(n_map * 4);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < Cf; ++c)
      for (int y = 0; y < FH; ++y)
        for (int x = 0; x < FW; ++x)
          maps[(((size_t)b * FH + y) * FW + x) * Cf + c] = ((float *)fpn0->data)[(((size_t)b * Cf + c) * FH + y) * FW + x];

  /* per-sample camera set-up on the host (detectors/imvoxelnet.py:114-129, :139) */
  float proj[2 * 12], new_origin[2 * 3];
  int32_t crop[2 * 2];
  for (int b = 0; b < B; ++b) {
    char key[64];
    snprintf(key, sizeof(key), "meta%d::img_shape", b);
    const int64_t *ishape = (const int64_t *)get(key)->data;
    snprintf(key, sizeof(key), "meta%d::ori_shape", b);
    const int64_t *oshape = (const int64_t *)get(key)->data;
    const double ratio = (double)oshape[0] / ((double)ishape[0] / 4.0);
    snprintf(key, sizeof(key), "meta%d::intrinsic", b);
    const float *K = (const float *)get(key)->data;
    snprintf(key, sizeof(key), "meta%d::extrinsic", b);
    const float *E = (const float *)get(key)->data;
    snprintf(key, sizeof(key), "meta%d::origin", b);
    const float *origin = (const float *)get(key)->data;
    CK(ivx_compute_projection(K, E, V, ratio, proj + b * 12));
    CK(ivx_voxel_new_origin(origin, cfg.n_voxels, cfg.voxel_size, new_origin + b * 3));
    crop[b * 2 + 0] = (int32_t)(ishape[0] / 4);
    crop[b * 2 + 1] = (int32_t)(ishape[1] / 4);
  }

  const int64_t ws_bytes = ivx_model_workspace_bytes(m, B, V, H, W);
  if (ws_bytes < 0) { fprintf(stderr, "ivx_model_workspace_bytes: %s\n", ivx_last_error()); return 1; }
  const int M = cfg.max_num, NV = cfg.n_voxels[0] * cfg.n_voxels[1] * cfg.n_voxels[2];
  float *d_maps, *d_proj, *d_no, *d_boxes, *d_scores;
  int32_t *d_crop, *d_count;
  int64_t *d_labels;
  uint8_t *d_valid;
  void *d_ws;
  HK(hipMalloc((void **)&d_maps, n_map * 4));
  HK(hipMalloc((void **)&d_proj, sizeof(proj)));
  HK(hipMalloc((void **)&d_no, sizeof(new_origin)));
  HK(hipMalloc((void **)&d_crop, sizeof(crop)));
  HK(hipMalloc((void **)&d_boxes, (size_t)B * M * 7 * 4));
  HK(hipMalloc((void **)&d_scores, (size_t)B * M * 4));
  HK(hipMalloc((void **)&d_labels, (size_t)B * M * 8));
  HK(hipMalloc((void **)&d_count, (size_t)B * 4));
  HK(hipMalloc((void **)&d_valid, (size_t)B * NV));
  HK(hipMalloc(&d_ws, (size_t)ws_bytes));
  HK(hipMemcpy(d_maps, maps, n_map * 4, hipMemcpyHostToDevice));
  HK(hipMemcpy(d_proj, proj, sizeof(proj), hipMemcpyHostToDevice));
  HK(hipMemcpy(d_no, new_origin, sizeof(new_origin), hipMemcpyHostToDevice));
  HK(hipMemcpy(d_crop, crop, sizeof(crop), hipMemcpyHostToDevice));

  CK(ivx_model_forward(m, d_maps, B, V, H, W, d_proj, d_no, d_crop, d_ws, ws_bytes, d_boxes, d_scores, d_labels, d_count, d_valid, NULL));
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
  const entry_t *rv = get("valids");
  if (memcmp(valid, rv->data, (size_t)B * NV)) { fprintf(stderr, "valid mask differs from the reference\n"); ++bad; }
  for (int b = 0; b < B; ++b) {
    char key[64];
    snprintf(key, sizeof(key), "res%d::scores", b);
    const entry_t *rs = get(key);
    snprintf(key, sizeof(key), "res%d::boxes", b);
    const entry_t *rb = get(key);
    snprintf(key, sizeof(key), "res%d::labels", b);
    const entry_t *rl = get(key);
    if (count[b] != (int)rs->numel) { fprintf(stderr, "sample %d: %d detections, reference %lld\n", b, count[b], (long long)rs->numel); ++bad; continue; }
    float ds = 0.f, db = 0.f;
    for (int i = 0; i < count[b]; ++i) {
      ds = fmaxf(ds, fabsf(scores[b * M + i] - ((float *)rs->data)[i]));
      if (labels[b * M + i] != ((int64_t *)rl->data)[i]) { fprintf(stderr, "sample %d det %d: label differs\n", b, i); ++bad; }
      for (int c = 0; c < 7; ++c) db = fmaxf(db, fabsf(boxes[((size_t)b * M + i) * 7 + c] - ((float *)rb->data)[i * 7 + c]));
    }
    printf("sample %d: %d detections, max |dscore| %.2e, max |dbox| %.2e\n", b, count[b], ds, db);
    if (ds > 1e-5f || db > 1e-4f) { fprintf(stderr, "sample %d: out of tolerance (1e-5 scores / 1e-4 boxes)\n", b); ++bad; }
  }
  CK(ivx_destroy(m));
  if (bad) { printf("C e2e_small FAILED (%d problems)\n", bad); return 1; }
  printf("C e2e_small OK: %d weight tensors loaded, workspace %lld bytes, no Python involved\n", loaded, (long long)ws_bytes);
  return 0;
}
