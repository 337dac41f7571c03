// Detection tail on the device: Anchor3DHead.get_bboxes_single (sigmoid + top-k + decode + BEV NMS +
// yaw fix-up), the stand-alone BEV NMS (iou3d_cuda.nms_gpu / nms_normal_gpu without the host round
// trip), pairwise rotated overlap, and aligned_3d_nms.  See include/imvoxel.h for the reference
// lines each entry point replaces.  Latency-bound code: a handful of small launches per batch.
// Compiled with -ffp-contract=off so the box arithmetic is the reference's operation order.
#include "ivx_common.h"

#include <math.h>

#define IVX_PI_F 3.14159274101257324f /* float(np.pi) */
#define IVX_NMS_EPS 1e-8f

// ------------------------------------------------------------------------------------------------
// Rotated-rectangle geometry: mmdet3d/ops/iou3d/src/iou3d_kernel.cu:16-251 restated for wave64.
struct Pt { float x, y; };
__device__ inline float cross2(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }
__device__ inline float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

__device__ inline int rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ inline int in_box2d(const float *box, Pt p) {
  const float MARGIN = 1e-5f;
  float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  float ac = cosf(-box[4]), as = sinf(-box[4]);
  float rx = (p.x - cx) * ac + (p.y - cy) * as + cx;
  float ry = -(p.x - cx) * as + (p.y - cy) * ac + cy;
  return (rx > box[0] - MARGIN && rx < box[2] + MARGIN && ry > box[1] - MARGIN && ry < box[3] + MARGIN);
}

__device__ inline int seg_isect(Pt p1, Pt p0, Pt q1, Pt q0, Pt *ans) {
  if (rect_cross(p0, p1, q0, q1) == 0) return 0;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > IVX_NMS_EPS) {
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

__device__ inline void rot_center(Pt c, float ac, float as, Pt *p) {
  float nx = (p->x - c.x) * ac + (p->y - c.y) * as + c.x;
  float ny = -(p->x - c.x) * as + (p->y - c.y) * ac + c.y;
  p->x = nx;
  p->y = ny;
}

__device__ float box_overlap_dev(const float *ba, const float *bb) {
  float a_x1 = ba[0], a_y1 = ba[1], a_x2 = ba[2], a_y2 = ba[3], a_ang = ba[4];
  float b_x1 = bb[0], b_y1 = bb[1], b_x2 = bb[2], b_y2 = bb[3], b_ang = bb[4];
  Pt ca = {(a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2};
  Pt cb = {(b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2};
  Pt A[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  Pt Bc[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_cos = cosf(a_ang), a_sin = sinf(a_ang), b_cos = cosf(b_ang), b_sin = sinf(b_ang);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rot_center(ca, a_cos, a_sin, &A[k]);
    rot_center(cb, b_cos, b_sin, &Bc[k]);
  }
  A[4] = A[0];
  Bc[4] = Bc[0];
  Pt cp[16];
  Pt pc = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      Pt t;
      if (seg_isect(A[i + 1], A[i], Bc[j + 1], Bc[j], &t)) {
        cp[cnt] = t;
        pc.x = pc.x + t.x;
        pc.y = pc.y + t.y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box2d(ba, Bc[k])) {
      pc.x = pc.x + Bc[k].x;
      pc.y = pc.y + Bc[k].y;
      cp[cnt++] = Bc[k];
    }
    if (in_box2d(bb, A[k])) {
      pc.x = pc.x + A[k].x;
      pc.y = pc.y + A[k].y;
      cp[cnt++] = A[k];
    }
  }
  pc.x /= cnt;
  pc.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i) {
      const bool gt = atan2f(cp[i].y - pc.y, cp[i].x - pc.x) > atan2f(cp[i + 1].y - pc.y, cp[i + 1].x - pc.x);
      if (gt) {
        Pt t = cp[i];
        cp[i] = cp[i + 1];
        cp[i + 1] = t;
      }
    }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    Pt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y};
    Pt v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += cross2(u, v);
  }
  return fabsf(area) / 2.0f;
}

__device__ inline float iou_bev_dev(const float *a, const float *b) {
  float sa = (a[2] - a[0]) * (a[3] - a[1]);
  float sb = (b[2] - b[0]) * (b[3] - b[1]);
  float so = box_overlap_dev(a, b);
  return so / fmaxf(sa + sb - so, IVX_NMS_EPS);
}

__device__ inline float iou_normal_dev(const float *a, const float *b) {
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0]) * (a[3] - a[1]);
  float Sb = (b[2] - b[0]) * (b[3] - b[1]);
  return interS / fmaxf(Sa + Sb - interS, IVX_NMS_EPS);
}

// ------------------------------------------------------------------------------------------------
// Suppression mask (iou3d_kernel.cu:284-333 / :345-396 restated for wave64): one wave per (row i, column block c);
// lane j computes IoU(i, 64c + j) and the wave's 64-bit ballot IS the mask word -- one rotated-IoU evaluation per
// lane instead of the reference's 64 sequential ones per thread.  Only column blocks >= the row's block are built
// (the greedy scan, iou3d.cpp:127-143, never reads the others); inside the diagonal block only columns > row.
// boxes [nb][box_stride][5]; n comes from n_arr[bz] (per batch item) or n_fixed.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *boxes, const int *n_arr, int n_fixed, int box_stride,
                                                      int cb_stride, float thr, int rotated, unsigned long long *mask) {
  const int bz = blockIdx.z;
  const int n = n_arr ? n_arr[bz] : n_fixed;
  const int col_start = blockIdx.x;
  const float *bx = boxes + (size_t)bz * box_stride * 5;
  const int col = col_start * 64 + threadIdx.x;
  // rows beyond gridDim.y wrap around (the grid's y extent stops at 65535; n may be 65536)
  for (int row = blockIdx.y; row < n; row += gridDim.y) {
    if (col_start * 64 >= n || col_start < (row >> 6)) continue;
    bool hit = false;
    if (col < n && col > row) {
      float a[5], b[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        a[q] = bx[(size_t)row * 5 + q];
        b[q] = bx[(size_t)col * 5 + q];
      }
      const float v = rotated ? iou_bev_dev(a, b) : iou_normal_dev(a, b);
      hit = v > thr;
    }
    const unsigned long long w = __ballot(hit);
    if (threadIdx.x == 0) mask[((size_t)bz * box_stride + row) * cb_stride + col_start] = w;
  }
}

// Greedy scan by one wave: lane w owns removal word w (n <= 4096).  Returns the number kept (all lanes)
// and writes kept indices (ascending == descending score) to keep_s (LDS or global), at most max_keep.
__device__ int greedy_scan_wave(const unsigned long long *mask, int n, int cb_stride, int max_keep, int *keep_out) {
  const int lane = threadIdx.x & 63;
  unsigned long long remv = 0;
  int nk = 0;
  for (int i = 0; i < n && nk < max_keep; ++i) {
    const int word = i >> 6, bit = i & 63;
    const unsigned long long rw = __shfl(remv, word, 64);
    if (!((rw >> bit) & 1ULL)) {
      if (lane == 0) keep_out[nk] = i;
      ++nk;
      if (lane >= word && lane * 64 < n) remv |= mask[(size_t)i * cb_stride + lane];
    }
  }
  return nk;
}

// ---- General N (> 4096 candidates; the reference op takes any N, iou3d.cpp:95-147 with col_blocks = DIVUP(N, 64)).
// The removal set no longer fits one word per lane, so it lives in the scanning wave's LDS (IVX_NMS_MAX_N / 64 words); the
// visiting order is either the row order (triangular mask over score-sorted boxes: only the words >= the row's own are
// defined) or an explicit order over a FULL hit matrix indexed by original box (order != NULL; the next 64 entries are
// prefetched into the lanes).  One wave, one dependent step per visited box, as in the 64-word form above -- same kept
// sequence by construction.  Kept entries go straight to global memory: out[nk] = remap ? remap[i] : i.
#define IVX_NMS_MAX_N 65536
template <typename OutT>
__device__ int greedy_scan_big(const unsigned long long *mask, const int *order, int n, int cb, int max_keep, const int *remap,
                               unsigned long long *remv /* LDS, cb words */, OutT *out) {
  const int lane = threadIdx.x & 63;
  for (int w = lane; w < cb; w += 64) remv[w] = 0ULL;
  __syncthreads();                                  // (a one-wave workgroup: orders the LDS accesses)
  int nk = 0, chunk = 0;
  for (int i = 0; i < n && nk < max_keep; ++i) {
    if (order && (i & 63) == 0) chunk = (i + lane < n) ? order[i + lane] : 0;
    const int a = order ? __shfl(chunk, i & 63, 64) : i;
    const unsigned long long rw = remv[a >> 6];
    if (!((rw >> (a & 63)) & 1ULL)) {
      if (lane == 0) out[nk] = (OutT)(remap ? remap[i] : i);
      ++nk;
      const unsigned long long *row = mask + (size_t)a * cb;
      for (int w = (order ? 0 : (a >> 6)) + lane; w < cb; w += 64) remv[w] |= row[w];
      __syncthreads();
    }
  }
  return nk;
}

__global__ __launch_bounds__(64) void nms_scan_big_kernel(const unsigned long long *mask, const int *order, int n, int cb, const int *remap,
                                                          long long *keep, int *num_out) {
  __shared__ unsigned long long remv[IVX_NMS_MAX_N / 64];
  const int nk = greedy_scan_big(mask, order, n, cb, n, remap, remv, keep);
  if (threadIdx.x == 0) *num_out = nk;
}

// Rank sort of n 64-bit keys, descending, for lists beyond one workgroup's LDS sort: rank[i] = #{j : key[j] > key[i]} (the keys
// are unique -- their low word is ~index).  O(n^2) compares, n^2 / 256 per workgroup from LDS tiles: 0.4 G compares at n = 20 000.
// grid (ceil(n / 256), lists); keys [lists][stride].
__global__ __launch_bounds__(256) void rank_sort_kernel(const unsigned long long *keys, int n, int stride, int *rank) {
  __shared__ unsigned long long tile[1024];
  const unsigned long long *k = keys + (size_t)blockIdx.y * stride;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long mine = i < n ? k[i] : 0ULL;
  int r = 0;
  for (int base = 0; base < n; base += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) tile[t] = base + t < n ? k[base + t] : 0ULL;
    __syncthreads();
    const int lim = n - base < 1024 ? n - base : 1024;
    for (int t = 0; t < lim; ++t) r += tile[t] > mine ? 1 : 0;
  }
  if (i < n) rank[(size_t)blockIdx.y * stride + i] = r;
}

// ------------------------------------------------------------------------------------------------
// Anchor head: scores
struct HeadP {
  const float *head_out;  // [B, HW, CH]
  const float *anchors;   // [HW*A, 7]
  int B, HW, CH, A, ncls, cls_off, reg_off, dir_off;
  int Hh, Ww, transposed;
  int n;                  // HW * A
  int nms_pre, max_num, kpad, cb;
  float score_thr, nms_thr, dir_offset, dir_limit_offset;
  int rotated;
};

// memory position of logical location hw = y*W + x
__device__ inline int hw_mem(const HeadP &p, int hw) {
  if (!p.transposed) return hw;
  const int y = hw / p.Ww, x = hw - y * p.Ww;
  return x * p.Hh + y;
}

__device__ inline float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void anchor_scores_kernel(const HeadP p, float *keys) {
  const size_t total = (size_t)p.B * p.n;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / p.n);
    const int i = (int)(idx % p.n);
    const int hw = i / p.A, a = i % p.A;
    const float *src = p.head_out + ((size_t)b * p.HW + hw_mem(p, hw)) * p.CH + p.cls_off + a * p.ncls;
    float m = sigmoid_ref(src[0]);
    for (int c = 1; c < p.ncls; ++c) {
      const float s = sigmoid_ref(src[c]);
      m = s > m ? s : m;
    }
    keys[idx] = m;
  }
}

__device__ inline unsigned int f2key(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key2f(unsigned int k) {
  unsigned int u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Bitonic sort of `len` (power of two) 64-bit keys in LDS, descending.
__device__ void bitonic_sort_desc(unsigned long long *s, int len) {
  for (int k = 2; k <= len; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = s[i], b = s[ixj];
          const bool up = ((i & k) == 0);  // descending blocks first
          if (up ? (a < b) : (a > b)) {
            s[i] = b;
            s[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

// top-k (torch.topk semantics: k largest, sorted descending; ties -> lower index first) of keys[b][0..n)
// by an 8-bit radix select on the composite key (score bits << 32 | ~index), then an LDS bitonic sort.
// One workgroup of 1024 threads per batch item.  kk = min(nms_pre, n) candidates are produced (when
// n <= nms_pre the reference keeps all of them unsorted; sorting them changes nothing downstream because
// nms_gpu sorts by score itself).  Writes topk_idx[b][0..kpad) (-1 padded), cnt[b] = kk and
// n1[b] = #(score > score_thr) (a prefix of the sorted candidates).
// First wave only (lane = threadIdx.x < 64): a[0..256) are counts; finds the one index c with
//   sum(a[j], j > c) < rem <= sum(a[j], j >= c)      (requires sum(a) >= rem >= 1)
// The lane that owns c (4 consecutive entries per lane, suffix sums by wave shuffles) returns true with *idx = c and
// *above = sum(a[j], j > c); every other lane returns false.
__device__ __forceinline__ bool suffix_find_256(const unsigned int *a, unsigned int rem, int lane, int *idx, unsigned int *above) {
  const unsigned int v0 = a[4 * lane], v1 = a[4 * lane + 1], v2 = a[4 * lane + 2], v3 = a[4 * lane + 3];
  const unsigned int mine = v0 + v1 + v2 + v3;
  unsigned int suf = mine;                       // -> sum over lanes >= lane
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int t = __shfl_down(suf, off);
    if (lane + off < 64) suf += t;
  }
  unsigned int cum = suf - mine;                 // entries of the lanes above
  if (!(cum < rem && suf >= rem)) return false;
  const unsigned int v[4] = {v0, v1, v2, v3};
  int j = 3;
  for (; j > 0; --j) {
    if (cum + v[j] >= rem) break;
    cum += v[j];
  }
  *idx = 4 * lane + j;
  *above = cum;
  return true;
}

__device__ void topk_select_body(const HeadP &p, const float *keys, int *topk_idx, int *cnt_out, int *n1_out,
                                 unsigned long long *sk /* LDS, kpad entries */) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long s_prefix, s_mask;
  __shared__ int s_remaining, s_done, s_cnt, s_n1;
  const int b = blockIdx.x;
  const float *kb = keys + (size_t)b * p.n;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int k = (p.nms_pre > 0 && p.nms_pre < p.n) ? p.nms_pre : p.n;
  int *out_idx = topk_idx + (size_t)b * p.kpad;

  if (tid == 0) {
    s_prefix = 0;
    s_mask = 0;
    s_remaining = k;
    s_done = (k >= p.n) ? 1 : 0;  // everything selected: threshold key 0
    s_cnt = 0;
    s_n1 = 0;
  }
  __syncthreads();
  if (!s_done) {
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      for (int i = tid; i < 256; i += nt) hist[i] = 0;
      __syncthreads();
      const unsigned long long prefix = s_prefix, msk = s_mask;
      for (int i = tid; i < p.n; i += nt) {
        const unsigned long long key = ((unsigned long long)f2key(kb[i]) << 32) | (unsigned int)(~(unsigned int)i);
        if ((key & msk) == prefix) atomicAdd(&hist[(unsigned int)(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid < 64) {        // the bin holding the rem-th largest key of the prefix: suffix sums across the first wave
        int chosen = 0;
        unsigned int cum = 0;
        if (suffix_find_256(hist, (unsigned int)s_remaining, tid, &chosen, &cum)) {
          const int rem = s_remaining - (int)cum;
          s_prefix = prefix | ((unsigned long long)chosen << shift);
          s_mask = msk | (0xffULL << shift);
          s_remaining = rem;
          if ((int)hist[chosen] == rem) s_done = 1;  // the whole bucket is taken: lower bits are irrelevant
        }
      }
      __syncthreads();
      if (s_done) break;
    }
  }
  const unsigned long long thr_key = s_prefix;  // every key >= thr_key is selected: exactly k of them
  for (int i = tid; i < p.kpad; i += nt) sk[i] = 0ULL;
  __syncthreads();
  for (int i = tid; i < p.n; i += nt) {
    const unsigned long long key = ((unsigned long long)f2key(kb[i]) << 32) | (unsigned int)(~(unsigned int)i);
    if (key >= thr_key) {
      const int pos = atomicAdd(&s_cnt, 1);
      if (pos < p.kpad) sk[pos] = key;
    }
  }
  __syncthreads();
  bitonic_sort_desc(sk, p.kpad);
  int local = 0;
  for (int i = tid; i < p.kpad; i += nt) {
    if (i < k) {
      const unsigned long long key = sk[i];
      out_idx[i] = (int)(~(unsigned int)(key & 0xffffffffULL));
      if (key2f((unsigned int)(key >> 32)) > p.score_thr) ++local;
    } else {
      out_idx[i] = -1;
    }
  }
  atomicAdd(&s_n1, local);
  __syncthreads();
  if (tid == 0) {
    cnt_out[b] = k;
    n1_out[b] = s_n1;
  }
}

__global__ __launch_bounds__(1024) void topk_select_kernel(const HeadP p, const float *keys, int *topk_idx, int *cnt_out,
                                                           int *n1_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];  // kpad entries
  topk_select_body(p, keys, topk_idx, cnt_out, n1_out, sk);
}

// ---- the same top-k for long score lists (the 80 x 80 x 32 level of the indoor heads: 204 800 scores at batch 1, the KITTI
// anchor grid: 105 288 per sample).  One workgroup re-reading the list up to nine times is latency-bound (604 us at 204 800);
// here the list is read four times by the whole chip:
//   topk_hist_kernel x 3:  radix histograms of the score key, 11 + 11 + 10 bits from the top, each restricted to the prefix
//                          found so far; per-workgroup LDS histogram with wave-aggregated atomics (scores of a head cluster in
//                          a few bins: one atomic per distinct bin of a wave), flushed to the sample's global histogram
//   topk_thresh_kernel x 3: the bin holding the k-th largest key at that level (scan from the top) -> after three levels the
//                          exact 32-bit key of the k-th largest score
//   topk_compact_kernel:   every key >= that threshold -> candidate list (composite key: score bits << 32 | ~index)
//   topk_final_kernel:     LDS bitonic sort of the candidates, the first k leave -- the same k keys in the same order as the
//                          radix select above (composite keys are unique, ties resolve to the lower index); more than
//                          TOPK_CAP candidates (thousands of bit-equal scores at the threshold) fall back to that select.
#define TOPK_CAP 8192
#define TOPK_BINS 2048
struct TopkState {          // per sample
  unsigned int prefix;      // key bits decided so far (right-aligned)
  unsigned int rem;         // how many keys are still to take inside the prefix
  unsigned int thr;         // final: the k-th largest 32-bit key
  unsigned int cand_cnt;
};

__device__ __forceinline__ bool topk_level_bin(unsigned int key, int level, unsigned int prefix, unsigned int *bin) {
  if (level == 0) { *bin = key >> 21; return true; }
  if (level == 1) { *bin = (key >> 10) & 2047u; return (key >> 21) == prefix; }
  *bin = key & 1023u;
  return (key >> 10) == prefix;
}

__global__ __launch_bounds__(256) void topk_hist_kernel(const float *keys, int n, int level, const TopkState *state, unsigned int *hist) {
  __shared__ unsigned int lh[TOPK_BINS];
  const int b = blockIdx.y;
  const float *kb = keys + (size_t)b * n;
  unsigned int *h = hist + ((size_t)b * 3 + level) * TOPK_BINS;
  const unsigned int prefix = level ? state[b].prefix : 0u;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < TOPK_BINS; i += 256) lh[i] = 0;
  __syncthreads();
  for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    unsigned int bin = 0;
    const bool act = i < n && topk_level_bin(f2key(kb[i]), level, prefix, &bin);
    // the scores of a head cluster: at the first level most of a wave lands in ONE bin (64 same-address LDS atomics
    // serialise), deeper levels spread out.  One aggregation round for the first active lane's bin when at least 8 lanes
    // share it, plain LDS atomics for everything else.
    const unsigned long long actm = __ballot(act);
    if (!actm) continue;
    const int leader = __ffsll((long long)actm) - 1;
    const unsigned int lb = __shfl(bin, leader);
    const unsigned long long same = __ballot(act && bin == lb);
    const bool agg = __popcll(same) >= 8;
    if (agg && lane == leader) atomicAdd(&lh[lb], (unsigned int)__popcll(same));
    if (act && !(agg && bin == lb)) atomicAdd(&lh[bin], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TOPK_BINS; i += 256)
    if (lh[i]) atomicAdd(&h[i], lh[i]);
}

__global__ __launch_bounds__(256) void topk_thresh_kernel(const unsigned int *hist, int n, int nms_pre, int level, TopkState *state) {
  __shared__ unsigned int part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned int *h = hist + ((size_t)b * 3 + level) * TOPK_BINS;
  const int bins = level == 2 ? 1024 : 2048, per = bins / 256;
  unsigned int sum = 0;
  for (int j = 0; j < per; ++j) sum += h[tid * per + j];
  part[tid] = sum;
  __syncthreads();
  const int k = (nms_pre > 0 && nms_pre < n) ? nms_pre : n;
  const unsigned int rem = level ? state[b].rem : (unsigned int)k;
  const unsigned int prefix = level ? state[b].prefix : 0u;
  int c = 0;
  unsigned int cum = 0;
  if (tid < 64 && suffix_find_256(part, rem, tid, &c, &cum)) {     // one lane: the 8- (4-) bin chunk, then the bin inside it
    int T = c * per;
    for (int j = per - 1; j >= 0; --j) {
      const unsigned int v = h[c * per + j];
      if (cum + v >= rem || j == 0) { T = c * per + j; break; }
      cum += v;
    }
    const unsigned int np = level == 2 ? ((prefix << 10) | (unsigned int)T) : ((prefix << 11) | (unsigned int)T);
    state[b].prefix = np;
    state[b].rem = rem - cum;
    if (level == 2) {
      state[b].thr = k >= n ? 0u : np;      // everything is taken when k == n
      state[b].cand_cnt = 0;
    }
  }
}

__global__ __launch_bounds__(256) void topk_compact_kernel(const float *keys, int n, TopkState *state, unsigned long long *cand) {
  const int b = blockIdx.y;
  const float *kb = keys + (size_t)b * n;
  unsigned long long *cb = cand + (size_t)b * TOPK_CAP;
  const unsigned int thr = state[b].thr;
  const int lane = threadIdx.x & 63;
  for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    unsigned int fk = 0;
    bool sel = false;
    if (i < n) {
      fk = f2key(kb[i]);
      sel = fk >= thr;
    }
    const unsigned long long m = __ballot(sel);
    if (!m) continue;
    unsigned int pos = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) pos = atomicAdd(&state[b].cand_cnt, (unsigned int)__popcll(m));
    pos = __shfl(pos, leader);
    if (sel) {
      const unsigned int at = pos + (unsigned int)__popcll(m & ((1ULL << lane) - 1ULL));
      if (at < TOPK_CAP) cb[at] = ((unsigned long long)fk << 32) | (unsigned int)(~(unsigned int)i);
    }
  }
}

__global__ __launch_bounds__(1024) void topk_final_kernel(const HeadP p, const float *keys, const unsigned long long *cand,
                                                          const TopkState *state, int *topk_idx, int *cnt_out, int *n1_out) {
  __shared__ __attribute__((aligned(16))) unsigned long long sk[TOPK_CAP];   // 64 KB, static (a workgroup may hold up to 160 KB)
  __shared__ int s_n1f;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const unsigned int c = state[b].cand_cnt;
  if (c > TOPK_CAP) {                       // too many bit-equal scores at the threshold: exact select over the whole list
    topk_select_body(p, keys, topk_idx, cnt_out, n1_out, sk);
    return;
  }
  const int k = (p.nms_pre > 0 && p.nms_pre < p.n) ? p.nms_pre : p.n;
  int len = p.kpad;
  while (len < (int)c) len <<= 1;           // power of two >= max(c, kpad), <= TOPK_CAP
  const unsigned long long *cb = cand + (size_t)b * TOPK_CAP;
  for (int i = tid; i < len; i += nt) sk[i] = i < (int)c ? cb[i] : 0ULL;
  if (tid == 0) s_n1f = 0;
  __syncthreads();
  bitonic_sort_desc(sk, len);
  int *out_idx = topk_idx + (size_t)b * p.kpad;
  int local = 0;
  for (int i = tid; i < p.kpad; i += nt) {
    if (i < k) {
      const unsigned long long key = sk[i];
      out_idx[i] = (int)(~(unsigned int)(key & 0xffffffffULL));
      if (key2f((unsigned int)(key >> 32)) > p.score_thr) ++local;
    } else {
      out_idx[i] = -1;
    }
  }
  atomicAdd(&s_n1f, local);
  __syncthreads();
  if (tid == 0) {
    cnt_out[b] = k;
    n1_out[b] = s_n1f;
  }
}

static int64_t topk_scratch_bytes(int B) { return ivx_align_up((int64_t)B * (3 * TOPK_BINS * 4 + TOPK_CAP * 8 + (int64_t)sizeof(TopkState)), 256); }
static thread_local int g_topk_mode = 0;     // A/B and test knob: 1 = always the one-workgroup select
extern "C" int ivx_topk_set_mode(int32_t single_workgroup) {
  g_topk_mode = single_workgroup ? 1 : 0;
  return IVX_OK;
}

// top-k of keys[b][0..n) for every b; scratch: topk_scratch_bytes(B) bytes (256-aligned)
static void launch_topk(const HeadP &hp, const float *keys, int *topk, int *cnt, int *n1, void *scratch, hipStream_t st) {
  if (hp.n < 16384 || hp.kpad > TOPK_CAP || g_topk_mode == 1) {
    hipLaunchKernelGGL(topk_select_kernel, dim3(hp.B), dim3(1024), (size_t)hp.kpad * 8, st, hp, keys, topk, cnt, n1);
    return;
  }
  unsigned int *hist = (unsigned int *)scratch;
  unsigned long long *cand = (unsigned long long *)((char *)scratch + (size_t)hp.B * 3 * TOPK_BINS * 4);
  TopkState *state = (TopkState *)((char *)cand + (size_t)hp.B * TOPK_CAP * 8);
  (void)hipMemsetAsync(hist, 0, (size_t)hp.B * 3 * TOPK_BINS * 4, st);
  int g = (hp.n + 2047) / 2048;
  const int gmax = hp.B >= 4 ? 64 : 128;
  if (g > gmax) g = gmax;
  for (int level = 0; level < 3; ++level) {
    hipLaunchKernelGGL(topk_hist_kernel, dim3(g, hp.B), dim3(256), 0, st, keys, hp.n, level, state, hist);
    hipLaunchKernelGGL(topk_thresh_kernel, dim3(hp.B), dim3(256), 0, st, hist, hp.n, hp.nms_pre, level, state);
  }
  hipLaunchKernelGGL(topk_compact_kernel, dim3(g, hp.B), dim3(256), 0, st, keys, hp.n, state, cand);
  hipLaunchKernelGGL(topk_final_kernel, dim3(hp.B), dim3(1024), 0, st, hp, keys, cand, state, topk, cnt, n1);
}

// ---- top-k for more than 4096 candidates per list (the reference's topk takes any nms_pre, dense_heads/anchor3d_head.py:468-490).  The
// three radix levels above give the exact 32-bit key of the k-th largest score; every key >= it is compacted (any count: thousands of
// bit-equal scores at the threshold -- e.g. a level with fewer valid voxels than nms_pre -- all become candidates), ranked among the
// candidates by composite key (score bits << 32 | ~index: unique, ties resolve to the lower index; O(c^2) compares from LDS tiles,
// 0.1 G at c = 10 000) and the first k ranks are scattered to their places.  Same indices in the same order as the forms above.
__global__ __launch_bounds__(256) void topk_compact_big_kernel(const float *keys, int n, TopkState *state, unsigned long long *cand) {
  const int b = blockIdx.y;
  const float *kb = keys + (size_t)b * n;
  unsigned long long *cb = cand + (size_t)b * n;
  const unsigned int thr = state[b].thr;
  const int lane = threadIdx.x & 63;
  for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    unsigned int fk = 0;
    bool sel = false;
    if (i < n) {
      fk = f2key(kb[i]);
      sel = fk >= thr;
    }
    const unsigned long long m = __ballot(sel);
    if (!m) continue;
    unsigned int pos = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) pos = atomicAdd(&state[b].cand_cnt, (unsigned int)__popcll(m));
    pos = __shfl(pos, leader);
    if (sel) cb[pos + (unsigned int)__popcll(m & ((1ULL << lane) - 1ULL))] = ((unsigned long long)fk << 32) | (unsigned int)(~(unsigned int)i);
  }
}

__global__ __launch_bounds__(256) void topk_rank_big_kernel(const HeadP p, const unsigned long long *cand, const TopkState *state, int *topk_idx,
                                                            int *n1_out) {
  __shared__ unsigned long long tile[1024];
  const int b = blockIdx.y;
  const int c = (int)state[b].cand_cnt;
  if ((int)(blockIdx.x * 256) >= c) return;                  // (whole workgroups: the barriers below stay uniform)
  const unsigned long long *k = cand + (size_t)b * p.n;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long mine = i < c ? k[i] : 0ULL;
  int r = 0;
  for (int base = 0; base < c; base += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) tile[t] = base + t < c ? k[base + t] : 0ULL;
    __syncthreads();
    const int lim = c - base < 1024 ? c - base : 1024;
    for (int t = 0; t < lim; ++t) r += tile[t] > mine ? 1 : 0;
  }
  const int kk = (p.nms_pre > 0 && p.nms_pre < p.n) ? p.nms_pre : p.n;
  if (i < c && r < kk) {
    topk_idx[(size_t)b * p.kpad + r] = (int)(~(unsigned int)(mine & 0xffffffffULL));
    if (key2f((unsigned int)(mine >> 32)) > p.score_thr) atomicAdd(&n1_out[b], 1);
  }
}

__global__ void topk_count_big_kernel(const HeadP p, int *cnt_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < p.B) cnt_out[b] = (p.nms_pre > 0 && p.nms_pre < p.n) ? p.nms_pre : p.n;
}

static int64_t topk_scratch_big_bytes(int B, int n) {
  return ivx_align_up((int64_t)B * 3 * TOPK_BINS * 4, 256) + ivx_align_up((int64_t)B * (int64_t)sizeof(TopkState), 256) + ivx_align_up((int64_t)B * n * 8, 256);
}

static void launch_topk_big(const HeadP &hp, const float *keys, int *topk, int *cnt, int *n1, void *scratch, hipStream_t st) {
  unsigned int *hist = (unsigned int *)scratch;
  TopkState *state = (TopkState *)((char *)scratch + ivx_align_up((int64_t)hp.B * 3 * TOPK_BINS * 4, 256));
  unsigned long long *cand = (unsigned long long *)((char *)state + ivx_align_up((int64_t)hp.B * (int64_t)sizeof(TopkState), 256));
  (void)hipMemsetAsync(hist, 0, (size_t)hp.B * 3 * TOPK_BINS * 4, st);
  (void)hipMemsetAsync(topk, 0xff, (size_t)hp.B * hp.kpad * 4, st);        // -1: the slots past k
  (void)hipMemsetAsync(n1, 0, (size_t)hp.B * 4, st);
  int g = (hp.n + 2047) / 2048;
  const int gmax = hp.B >= 4 ? 64 : 128;
  if (g > gmax) g = gmax;
  for (int level = 0; level < 3; ++level) {
    hipLaunchKernelGGL(topk_hist_kernel, dim3(g, hp.B), dim3(256), 0, st, keys, hp.n, level, state, hist);
    hipLaunchKernelGGL(topk_thresh_kernel, dim3(hp.B), dim3(256), 0, st, hist, hp.n, hp.nms_pre, level, state);
  }
  hipLaunchKernelGGL(topk_compact_big_kernel, dim3(g, hp.B), dim3(256), 0, st, keys, hp.n, state, cand);
  hipLaunchKernelGGL(topk_rank_big_kernel, dim3((hp.n + 255) / 256, hp.B), dim3(256), 0, st, hp, cand, state, topk, n1);
  hipLaunchKernelGGL(topk_count_big_kernel, dim3((hp.B + 63) / 64), dim3(64), 0, st, hp, cnt);
}

// Decode the selected candidates (DeltaXYZWLHRBBoxCoder.decode, coders/delta_xyzwhlr_bbox_coder.py:56-90),
// direction argmax (anchor3d_head.py:465-466), BEV xyxyr boxes (lidar_box3d.py:86-90, utils.py:64-82).
__global__ __launch_bounds__(64) void decode_kernel(const HeadP p, const float *keys, const int *topk_idx,
                                                    float *cand_boxes, float *cand_scores, int *cand_dir, float *cand_bev) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= p.kpad) return;
  const int idx = topk_idx[(size_t)b * p.kpad + j];
  float *ob = cand_boxes + ((size_t)b * p.kpad + j) * 7;
  float *bev = cand_bev + ((size_t)b * p.kpad + j) * 5;
  if (idx < 0) {
#pragma unroll
    for (int q = 0; q < 7; ++q) ob[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 5; ++q) bev[q] = 0.f;
    cand_scores[(size_t)b * p.kpad + j] = 0.f;
    cand_dir[(size_t)b * p.kpad + j] = 0;
    return;
  }
  const int hw = idx / p.A, a = idx % p.A;
  const float *row = p.head_out + ((size_t)b * p.HW + hw_mem(p, hw)) * p.CH;
  const float *an = p.anchors + (size_t)idx * 7;
  const float *dl = row + p.reg_off + a * 7;
  const float xa = an[0], ya = an[1], wa = an[3], la = an[4], ha = an[5], ra = an[6];
  float za = an[2];
  const float xt = dl[0], yt = dl[1], zt = dl[2], wt = dl[3], lt = dl[4], ht = dl[5], rt = dl[6];
  za = za + ha / 2;
  const float diag = sqrtf(la * la + wa * wa);
  const float xg = xt * diag + xa;
  const float yg = yt * diag + ya;
  float zg = zt * ha + za;
  const float lg = expf(lt) * la;
  const float wg = expf(wt) * wa;
  const float hg = expf(ht) * ha;
  const float rg = rt + ra;
  zg = zg - hg / 2;
  ob[0] = xg; ob[1] = yg; ob[2] = zg; ob[3] = wg; ob[4] = lg; ob[5] = hg; ob[6] = rg;
  const float hwid = wg / 2, hlen = lg / 2;
  bev[0] = xg - hwid; bev[1] = yg - hlen; bev[2] = xg + hwid; bev[3] = yg + hlen; bev[4] = rg;
  const float d0 = row[p.dir_off + a * 2], d1 = row[p.dir_off + a * 2 + 1];
  cand_dir[(size_t)b * p.kpad + j] = d1 > d0 ? 1 : 0;
  cand_scores[(size_t)b * p.kpad + j] = keys[(size_t)b * p.n + idx];
}

// Greedy scan + gather of the kept boxes (first max_num in descending score) + yaw fix-up
// (anchor3d_head.py:510-515 with limit_period, structures/utils.py:5-18).  One workgroup per batch item.
// lds_words > 0 (round 6): the workgroup first copies the sample's hit matrix (n1 rows x cb words, lds_words * 8 bytes of dynamic LDS) into LDS, so the
// scanning wave's one dependent load per KEPT box is an LDS read instead of an L2 round trip (nuScenes, nms_pre 1000 / 500 kept: 235 -> ~60 us); the kept
// sequence is the same by construction.
__global__ __launch_bounds__(256) void nms_finalize_kernel(const HeadP p, const int *n1_arr, const unsigned long long *mask,
                                                           const float *cand_boxes, const float *cand_scores,
                                                           const int *cand_dir, float *out_boxes, float *out_scores,
                                                           long long *out_labels, int *out_count, const int lds_words) {
  __shared__ int keep_s[4096];
  __shared__ int s_nk;
  extern __shared__ __attribute__((aligned(16))) unsigned long long mask_s[];
  const int b = blockIdx.x;
  const int n1 = n1_arr[b];
  const unsigned long long *mrow = mask + (size_t)b * p.kpad * p.cb;
  const bool staged = lds_words > 0 && n1 * p.cb <= lds_words;
  if (staged) {
    const int nw = n1 * p.cb;
    for (int i = threadIdx.x; i < nw; i += 256) mask_s[i] = mrow[i];
    __syncthreads();
  }
  if (threadIdx.x < 64) {
    const int nk = greedy_scan_wave(staged ? mask_s : mrow, n1, p.cb, p.max_num, keep_s);
    if (threadIdx.x == 0) s_nk = nk;
  }
  __syncthreads();
  const int nk = s_nk;
  for (int j = threadIdx.x; j < p.max_num; j += blockDim.x) {
    float *ob = out_boxes + ((size_t)b * p.max_num + j) * 7;
    if (j < nk) {
      const int i = keep_s[j];
      const float *cb_ = cand_boxes + ((size_t)b * p.kpad + i) * 7;
#pragma unroll
      for (int q = 0; q < 6; ++q) ob[q] = cb_[q];
      const float val = cb_[6] - p.dir_offset;
      const float t = floorf(val / IVX_PI_F + p.dir_limit_offset);
      const float dir_rot = val - t * IVX_PI_F;
      ob[6] = (dir_rot + p.dir_offset) + IVX_PI_F * (float)cand_dir[(size_t)b * p.kpad + i];
      out_scores[(size_t)b * p.max_num + j] = cand_scores[(size_t)b * p.kpad + i];
    } else {
#pragma unroll
      for (int q = 0; q < 7; ++q) ob[q] = 0.f;
      out_scores[(size_t)b * p.max_num + j] = 0.f;
    }
    out_labels[(size_t)b * p.max_num + j] = 0;
  }
  if (threadIdx.x == 0) out_count[b] = nk;
}

// The same for more than 4096 candidates (any nms_pre / max_num): the removal words live in the scanning wave's LDS, the kept list in
// global memory (keep_g [B][max_num]); one wave per batch item scans, then gathers.
__global__ __launch_bounds__(64) void nms_finalize_big_kernel(const HeadP p, const int *n1_arr, const unsigned long long *mask,
                                                              const float *cand_boxes, const float *cand_scores, const int *cand_dir,
                                                              int *keep_g, float *out_boxes, float *out_scores, long long *out_labels,
                                                              int *out_count) {
  __shared__ unsigned long long remv[IVX_NMS_MAX_N / 64];
  const int b = blockIdx.x;
  int *keep = keep_g + (size_t)b * p.max_num;
  const int nk = greedy_scan_big<int>(mask + (size_t)b * p.kpad * p.cb, nullptr, n1_arr[b], p.cb, p.max_num, nullptr, remv, keep);
  __threadfence_block();
  __syncthreads();
  for (int j = threadIdx.x; j < p.max_num; j += 64) {
    float *ob = out_boxes + ((size_t)b * p.max_num + j) * 7;
    if (j < nk) {
      const int i = __builtin_nontemporal_load(keep + j);
      const float *cb_ = cand_boxes + ((size_t)b * p.kpad + i) * 7;
#pragma unroll
      for (int q = 0; q < 6; ++q) ob[q] = cb_[q];
      const float val = cb_[6] - p.dir_offset;
      const float t = floorf(val / IVX_PI_F + p.dir_limit_offset);
      const float dir_rot = val - t * IVX_PI_F;
      ob[6] = (dir_rot + p.dir_offset) + IVX_PI_F * (float)cand_dir[(size_t)b * p.kpad + i];
      out_scores[(size_t)b * p.max_num + j] = cand_scores[(size_t)b * p.kpad + i];
    } else {
#pragma unroll
      for (int q = 0; q < 7; ++q) ob[q] = 0.f;
      out_scores[(size_t)b * p.max_num + j] = 0.f;
    }
    out_labels[(size_t)b * p.max_num + j] = 0;
  }
  if (threadIdx.x == 0) out_count[b] = nk;
}

__global__ void export_cands_kernel(const HeadP p, const int *topk_idx, const float *cand_boxes, const float *cand_scores,
                                    long long *o_idx, float *o_boxes, float *o_scores) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.nms_pre) return;
  const bool in = j < p.kpad;
  const int idx = in ? topk_idx[(size_t)b * p.kpad + j] : -1;
  if (o_idx) o_idx[(size_t)b * p.nms_pre + j] = idx;
  if (o_boxes)
    for (int q = 0; q < 7; ++q) o_boxes[((size_t)b * p.nms_pre + j) * 7 + q] = (in && idx >= 0) ? cand_boxes[((size_t)b * p.kpad + j) * 7 + q] : 0.f;
  if (o_scores) o_scores[(size_t)b * p.nms_pre + j] = (in && idx >= 0) ? cand_scores[(size_t)b * p.kpad + j] : 0.f;
}

static int next_pow2(int v) {
  int r = 1;
  while (r < v) r <<= 1;
  return r;
}

struct HeadWs {
  int64_t keys, topk, cnt, n1, boxes, scores, dir, bev, mask, topk_scratch, keep, total;
  bool big;
};

static int head_layout(const ivx_anchor_head_desc *d, HeadP *p, HeadWs *w) {
  IVX_REQUIRE(d, "ivx_anchor_head: null desc");
  IVX_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->CH > 0 && d->num_anchors > 0, "ivx_anchor_head: non-positive dims");
  IVX_REQUIRE(d->num_classes == 1, "ivx_anchor_head: only num_classes == 1 (the KITTI / nuScenes ImVoxelNet configs) is built; got %d",
              d->num_classes);
  IVX_REQUIRE(d->cls_off >= 0 && d->reg_off >= 0 && d->dir_off >= 0 &&
                  d->cls_off + d->num_anchors * d->num_classes <= d->CH && d->reg_off + d->num_anchors * 7 <= d->CH &&
                  d->dir_off + d->num_anchors * 2 <= d->CH,
              "ivx_anchor_head: channel blocks exceed CH");
  const int64_t n64 = (int64_t)d->H * d->W * d->num_anchors;
  IVX_REQUIRE(n64 < (1LL << 31) && (int64_t)d->B * n64 < (1LL << 40), "ivx_anchor_head: too many anchors");
  const int n = (int)n64;
  const int k = (d->nms_pre > 0 && d->nms_pre < n) ? d->nms_pre : n;
  // up to 4096 candidates: the tuned forms (LDS sorts, removal words in registers); beyond: rank sort + LDS removal words, any nms_pre the
  // NMS mask can index (IVX_NMS_MAX_N), as the reference's topk + nms_gpu (dense_heads/anchor3d_head.py:468-490, ops/iou3d/src/iou3d.cpp:95-147)
  const bool big = k > 4096 || d->max_num > 4096;
  IVX_REQUIRE(k <= IVX_NMS_MAX_N, "ivx_anchor_head: at most %d NMS candidates per sample (nms_pre=%d, anchors=%d)", IVX_NMS_MAX_N, d->nms_pre, n);
  IVX_REQUIRE(d->max_num > 0 && d->max_num <= IVX_NMS_MAX_N, "ivx_anchor_head: max_num must be in 1..%d", IVX_NMS_MAX_N);
  IVX_REQUIRE(d->nms_pre > 0, "ivx_anchor_head: nms_pre must be positive (size of the candidate outputs)");
  p->Hh = d->H; p->Ww = d->W; p->transposed = d->hw_transposed ? 1 : 0;
  p->B = d->B; p->HW = d->H * d->W; p->CH = d->CH; p->A = d->num_anchors; p->ncls = d->num_classes;
  p->cls_off = d->cls_off; p->reg_off = d->reg_off; p->dir_off = d->dir_off; p->n = n;
  p->nms_pre = d->nms_pre; p->max_num = d->max_num; p->kpad = big ? (k + 63) / 64 * 64 : next_pow2(k < 64 ? 64 : k); p->cb = p->kpad / 64;
  w->big = big;
  p->score_thr = d->score_thr; p->nms_thr = d->nms_thr; p->dir_offset = d->dir_offset; p->dir_limit_offset = d->dir_limit_offset;
  p->rotated = d->use_rotate_nms;
  int64_t o = 0;
  w->keys = o; o = ivx_align_up(o + (int64_t)d->B * n * 4, 256);
  w->topk = o; o = ivx_align_up(o + (int64_t)d->B * p->kpad * 4, 256);
  w->cnt = o; o = ivx_align_up(o + (int64_t)d->B * 4, 256);
  w->n1 = o; o = ivx_align_up(o + (int64_t)d->B * 4, 256);
  w->boxes = o; o = ivx_align_up(o + (int64_t)d->B * p->kpad * 7 * 4, 256);
  w->scores = o; o = ivx_align_up(o + (int64_t)d->B * p->kpad * 4, 256);
  w->dir = o; o = ivx_align_up(o + (int64_t)d->B * p->kpad * 4, 256);
  w->bev = o; o = ivx_align_up(o + (int64_t)d->B * p->kpad * 5 * 4, 256);
  w->mask = o; o = ivx_align_up(o + (int64_t)d->B * p->kpad * p->cb * 8, 256);
  w->topk_scratch = o; o += big ? topk_scratch_big_bytes(d->B, n) : topk_scratch_bytes(d->B);
  w->keep = o; o = ivx_align_up(o + (big ? (int64_t)d->B * d->max_num * 4 : 0), 256);
  w->total = o;
  return IVX_OK;
}

extern "C" int64_t ivx_anchor_head_workspace_bytes(const ivx_anchor_head_desc *d) {
  HeadP p;
  HeadWs w;
  if (head_layout(d, &p, &w) != IVX_OK) return -1;
  return w.total;
}

extern "C" int ivx_anchor_head_get_bboxes(const ivx_anchor_head_desc *d, const float *head_out, const float *anchors,
                                          void *workspace, int64_t workspace_bytes, float *out_boxes, float *out_scores,
                                          int64_t *out_labels, int32_t *out_count, int64_t *cand_idx, float *cand_boxes,
                                          float *cand_scores, ivx_stream_t stream) {
  HeadP p;
  HeadWs w;
  int rc = head_layout(d, &p, &w);
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(head_out && anchors && workspace && out_boxes && out_scores && out_labels && out_count, "ivx_anchor_head_get_bboxes: null argument");
  if (workspace_bytes < w.total) {
    ivx_set_error("ivx_anchor_head_get_bboxes: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)w.total);
    return IVX_ERR_WORKSPACE;
  }
  IVX_REQUIRE(((uintptr_t)workspace & 255) == 0, "ivx_anchor_head_get_bboxes: workspace must be 256-byte aligned");
  p.head_out = head_out;
  p.anchors = anchors;
  hipStream_t st = (hipStream_t)stream;
  char *ws = (char *)workspace;
  float *keys = (float *)(ws + w.keys);
  int *topk = (int *)(ws + w.topk);
  int *cnt = (int *)(ws + w.cnt);
  int *n1 = (int *)(ws + w.n1);
  float *cboxes = (float *)(ws + w.boxes);
  float *cscores = (float *)(ws + w.scores);
  int *cdir = (int *)(ws + w.dir);
  float *cbev = (float *)(ws + w.bev);
  unsigned long long *mask = (unsigned long long *)(ws + w.mask);

  const size_t total = (size_t)p.B * p.n;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(anchor_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, keys);
  if (w.big) launch_topk_big(p, keys, topk, cnt, n1, (char *)workspace + w.topk_scratch, st);
  else launch_topk(p, keys, topk, cnt, n1, (char *)workspace + w.topk_scratch, st);
  hipLaunchKernelGGL(decode_kernel, dim3(p.kpad / 64, p.B), dim3(64), 0, st, p, keys, topk, cboxes, cscores, cdir, cbev);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(p.cb, p.kpad > 65535 ? 65535 : p.kpad, p.B), dim3(64), 0, st, cbev, n1, 0, p.kpad, p.cb, p.nms_thr, p.rotated, mask);
  if (w.big)
    hipLaunchKernelGGL(nms_finalize_big_kernel, dim3(p.B), dim3(64), 0, st, p, n1, mask, cboxes, cscores, cdir, (int *)(ws + w.keep), out_boxes,
                       out_scores, (long long *)out_labels, out_count);
  else
  {
    // hit matrix of one sample in LDS when it fits beside the kept list (kpad x cb words: 128 KB at nms_pre 1000); rows the scan never reaches
    // (beyond n1) are not copied
    const long long words = (long long)p.kpad * p.cb;
    int lds_words = words * 8 <= 136 * 1024 ? (int)words : 0;
    if (lds_words * 8 > 48 * 1024) {       // above the default dynamic-LDS limit: raise it once per process; if the runtime refuses, scan from L2 as before
      static const bool raised = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024) == hipSuccess;
      if (!raised) { (void)hipGetLastError(); lds_words = 0; }
    }
    hipLaunchKernelGGL(nms_finalize_kernel, dim3(p.B), dim3(256), (size_t)lds_words * 8, st, p, n1, mask, cboxes, cscores, cdir, out_boxes, out_scores,
                       (long long *)out_labels, out_count, lds_words);
  }
  if (cand_idx || cand_boxes || cand_scores)
    hipLaunchKernelGGL(export_cands_kernel, dim3((p.nms_pre + 63) / 64, p.B), dim3(64), 0, st, p, topk, cboxes, cscores,
                       (long long *)cand_idx, cand_boxes, cand_scores);
  IVX_CHECK_LAUNCH("ivx_anchor_head_get_bboxes");
  return IVX_OK;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone BEV NMS
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long *mask, int n, int cb, long long *keep, int *num_out) {
  __shared__ int keep_s[4096];
  const int nk = greedy_scan_wave(mask, n, cb, n, keep_s);
  __syncthreads();
  for (int j = threadIdx.x; j < nk; j += 64) keep[j] = keep_s[j];
  if (threadIdx.x == 0) *num_out = nk;
}

extern "C" int64_t ivx_nms_workspace_bytes(int32_t n) {
  if (n < 0 || n > IVX_NMS_MAX_N) return -1;
  const int64_t cb = (n + 63) / 64;
  return ivx_align_up((int64_t)(n > 0 ? n : 1) * (cb > 0 ? cb : 1) * 8, 256);
}

extern "C" int ivx_nms_bev(const float *boxes_sorted, int32_t n, float thresh, int32_t rotated, void *workspace,
                           int64_t workspace_bytes, int64_t *keep, int32_t *num_out, ivx_stream_t stream) {
  IVX_REQUIRE(n >= 0 && n <= IVX_NMS_MAX_N, "ivx_nms_bev: n must be in 0..%d (got %d)", IVX_NMS_MAX_N, n);
  IVX_REQUIRE(keep && num_out, "ivx_nms_bev: null output");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_out, 0, sizeof(int32_t), st);
    if (e != hipSuccess) {
      ivx_set_error("ivx_nms_bev: memset failed: %s", hipGetErrorString(e));
      return IVX_ERR_HIP;
    }
    return IVX_OK;
  }
  IVX_REQUIRE(boxes_sorted && workspace, "ivx_nms_bev: null argument");
  if (workspace_bytes < ivx_nms_workspace_bytes(n)) {
    ivx_set_error("ivx_nms_bev: workspace too small");
    return IVX_ERR_WORKSPACE;
  }
  const int cb = (n + 63) / 64;
  unsigned long long *mask = (unsigned long long *)workspace;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, n > 65535 ? 65535 : n, 1), dim3(64), 0, st, boxes_sorted, (const int *)nullptr, n, n, cb, thresh, rotated, mask);
  if (n <= 4096)
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, st, mask, n, cb, (long long *)keep, num_out);
  else       // removal words in LDS instead of one per lane: same kept sequence
    hipLaunchKernelGGL(nms_scan_big_kernel, dim3(1), dim3(64), 0, st, mask, (const int *)nullptr, n, cb, (const int *)nullptr, (long long *)keep,
                       num_out);
  IVX_CHECK_LAUNCH("ivx_nms_bev");
  return IVX_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused multi-class BEV NMS (box3d_multiclass_nms, core/post_processing/box3d_nms.py:8-88): the reference loops over
// the classes on the host (mask filter, sort, nms_gpu with a blocking D2H, gathers, concat, final top-max_num); here
// every class is one slice of four launches and nothing returns to the host.
//   mc_select_sort: per class, candidates with score > score_thr sorted by score (descending; ties -> lower index),
//                   their boxes gathered contiguously for the mask kernel
//   nms_mask_kernel (shared with the single-class path, blockIdx.z = class)
//   mc_scan:        one wave per class runs the greedy scan
//   mc_finalize:    class-major concatenation when it fits max_num, else a k-way merge of the (already sorted)
//                   per-class lists = the reference's final scores.sort(descending)[:max_num]
struct McP {
  const float *boxes;    // [n][5]
  const float *scores;   // [n][score_stride]
  int n, ns, npad, score_stride, num_classes, max_num, rotated;
  float score_thr, nms_thr;
  int *sidx;             // [C][ns]  candidate -> original index, sorted
  float *sscore;         // [C][ns]
  float *cboxes;         // [C][ns][5]
  int *n_arr;            // [C]
  int *kept;             // [C][ns]  positions (in the sorted list) that survive
  float *kscore;         // [C][ns]  their scores (descending), compact
  int *nk;               // [C]
  unsigned long long *mask;
};

__global__ __launch_bounds__(1024) void mc_select_sort_kernel(const McP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long mk[];   // npad keys
  __shared__ int s_cnt;
  const int c = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int local = 0;
  for (int i = tid; i < p.npad; i += blockDim.x) {
    unsigned long long k = 0ULL;
    if (i < p.n) {
      const float sc = p.scores[(size_t)i * p.score_stride + c];
      if (sc > p.score_thr) {
        k = ((unsigned long long)f2key(sc) << 32) | (unsigned int)(~(unsigned int)i);
        ++local;
      }
    }
    mk[i] = k;
  }
  atomicAdd(&s_cnt, local);
  __syncthreads();
  bitonic_sort_desc(mk, p.npad);
  const int cnt = s_cnt;
  if (tid == 0) p.n_arr[c] = cnt;
  for (int j = tid; j < cnt; j += blockDim.x) {
    const unsigned long long k = mk[j];
    const int idx = (int)(~(unsigned int)(k & 0xffffffffULL));
    p.sidx[(size_t)c * p.ns + j] = idx;
    p.sscore[(size_t)c * p.ns + j] = key2f((unsigned int)(k >> 32));
#pragma unroll
    for (int q = 0; q < 5; ++q) p.cboxes[((size_t)c * p.ns + j) * 5 + q] = p.boxes[(size_t)idx * 5 + q];
  }
}

// Shared hit matrix (three or more classes): the boxes, hence the pairwise IoUs, are the same for every class -- only the
// visiting order differs -- so hit[a][b] = IoU(box a, box b) > thr is built ONCE over the original indices (full rows:
// the order is not known here) instead of once per class over the sorted candidates.
__global__ __launch_bounds__(64) void nms_hit_full_kernel(const float *boxes, int n, int cb, float thr, int rotated,
                                                          unsigned long long *mask) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  for (int row = blockIdx.y; row < n; row += gridDim.y) {      // (grid.y stops at 65535; n may be 65536)
    bool hit = false;
    if (col < n && col != row) {
      float a[5], b[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        a[q] = boxes[(size_t)row * 5 + q];
        b[q] = boxes[(size_t)col * 5 + q];
      }
      const float v = rotated ? iou_bev_dev(a, b) : iou_normal_dev(a, b);
      hit = v > thr;
    }
    const unsigned long long w = __ballot(hit);
    if (threadIdx.x == 0) mask[(size_t)row * cb + blockIdx.x] = w;
  }
}

// Greedy scan of class c over the shared hit matrix: candidates are visited in score order (sidx), the removal set is a
// bit per ORIGINAL box index (lane w owns word w).  Row a is only consulted once box a has been kept, and then only its
// bits for boxes visited LATER matter, so every decision uses iou(kept earlier box, later box) -- the argument order of
// the reference's nms_gpu -- and the result is identical to the per-class form (which does half the IoUs per class and
// is therefore kept for one or two classes).
__device__ int greedy_scan_shared(const unsigned long long *mask, const int *order_s, int n, int cb, int *keep_out) {
  const int lane = threadIdx.x & 63;
  unsigned long long remv = 0;
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    const int a = order_s[i];
    const unsigned long long rw = __shfl(remv, a >> 6, 64);
    if (!((rw >> (a & 63)) & 1ULL)) {
      if (lane == 0) keep_out[nk] = i;
      ++nk;
      if (lane < cb) remv |= mask[(size_t)a * cb + lane];
    }
  }
  return nk;
}

__global__ __launch_bounds__(64) void mc_scan_kernel(const McP p, const int shared) {
  __shared__ int keep_s[4096];
  __shared__ int order_s[4096];
  const int c = blockIdx.x;
  const int n = p.n_arr[c];
  const int cb = p.ns >> 6;
  int nk;
  if (shared) {
    for (int i = threadIdx.x; i < n; i += 64) order_s[i] = p.sidx[(size_t)c * p.ns + i];
    __syncthreads();
    nk = greedy_scan_shared(p.mask, order_s, n, cb, keep_s);
  } else {
    nk = greedy_scan_wave(p.mask + (size_t)c * p.ns * cb, n, cb, n, keep_s);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nk; j += 64) {
    p.kept[(size_t)c * p.ns + j] = keep_s[j];
    p.kscore[(size_t)c * p.ns + j] = p.sscore[(size_t)c * p.ns + keep_s[j]];
  }
  if (threadIdx.x == 0) p.nk[c] = nk;
}

// Final selection.  One thread per surviving box (class c, position j < min(nk[c], max_num) of its score-ordered list).
// When everything fits max_num the output is the class-major concatenation.  Otherwise the output is the max_num best
// by score (descending; ties -> lower class, then earlier position): the rank of an element is the number of elements
// that precede it, found with one binary search per class over the compact, descending per-class score lists -- fully
// parallel and deterministic (the reference does scores.sort(descending=True)[:max_num] on the concatenation).
__global__ __launch_bounds__(256) void mc_finalize_kernel(const McP p, long long *out_idx, long long *out_label, int *out_count) {
  __shared__ int s_nk[64], s_before[65];
  const int C = p.num_classes;
  if (threadIdx.x < 64) s_nk[threadIdx.x] = threadIdx.x < C ? p.nk[threadIdx.x] : 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int c = 0; c < 64; ++c) {
      s_before[c] = acc;
      acc += s_nk[c];
    }
    s_before[64] = acc;
  }
  __syncthreads();
  const int all = s_before[64];
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = all <= p.max_num ? all : p.max_num;
  const int cap = p.ns < p.max_num ? p.ns : p.max_num;      // elements per class that can reach the output
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)C * cap) return;
  const int c = (int)(e / cap), j = (int)(e - (long long)c * cap);
  if (j >= s_nk[c]) return;
  const long long src = p.sidx[(size_t)c * p.ns + p.kept[(size_t)c * p.ns + j]];
  if (all <= p.max_num) {
    out_idx[s_before[c] + j] = src;
    out_label[s_before[c] + j] = c;
    return;
  }
  const float sc = p.kscore[(size_t)c * p.ns + j];
  int rank = j;
  for (int o = 0; o < C; ++o) {
    if (o == c) continue;
    // number of leading elements of class o that come before (sc, c): score > sc, or == sc when o < c
    const float *ks = p.kscore + (size_t)o * p.ns;
    int lo = 0, hi = s_nk[o];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const float v = ks[mid];
      const bool before = (v > sc) || (v == sc && o < c);
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo;
    if (rank >= p.max_num) return;
  }
  out_idx[rank] = src;
  out_label[rank] = c;
}

// ---- n > 4096: the per-class LDS sort becomes keys + rank sort + scatter, the hit matrix is always the shared one (full rows over
// the original indices) and the per-class scan keeps its removal bits in LDS (greedy_scan_big with an explicit order).
__global__ __launch_bounds__(256) void mc_keys_big_kernel(const McP p, unsigned long long *keys) {
  const int c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const float sc = p.scores[(size_t)i * p.score_stride + c];
  // below the threshold: high word 0 (below every kept key: f2key of a score > thr >= -inf is > 0 ... f2key(x) >= 1 for any
  // non-NaN x except -max; a filtered entry still gets a unique key, so the ranks form a permutation)
  const bool ok = sc > p.score_thr;
  keys[(size_t)c * p.ns + i] = ((unsigned long long)(ok ? f2key(sc) : 0u) << 32) | (unsigned int)(~(unsigned int)i);
  if (ok) atomicAdd(&p.n_arr[c], 1);
}

__global__ __launch_bounds__(256) void mc_scatter_big_kernel(const McP p, const unsigned long long *keys, const int *rank) {
  const int c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const unsigned long long k = keys[(size_t)c * p.ns + i];
  if ((k >> 32) == 0ULL) return;                      // filtered
  const int j = rank[(size_t)c * p.ns + i];
  p.sidx[(size_t)c * p.ns + j] = i;
  p.sscore[(size_t)c * p.ns + j] = key2f((unsigned int)(k >> 32));
}

__global__ __launch_bounds__(64) void mc_scan_big_kernel(const McP p) {
  __shared__ unsigned long long remv[IVX_NMS_MAX_N / 64];
  const int c = blockIdx.x;
  const int n = p.n_arr[c];
  int *kept = p.kept + (size_t)c * p.ns;
  const int nk = greedy_scan_big(p.mask, p.sidx + (size_t)c * p.ns, n, p.ns >> 6, n, (const int *)nullptr, remv, kept);
  __syncthreads();
  for (int j = threadIdx.x; j < nk; j += 64) p.kscore[(size_t)c * p.ns + j] = p.sscore[(size_t)c * p.ns + kept[j]];
  if (threadIdx.x == 0) p.nk[c] = nk;
}

static int mc_layout(int32_t n, int32_t num_classes, McP *p, int64_t *total, int64_t *o_keys_out = nullptr, int64_t *o_rank_out = nullptr) {
  IVX_REQUIRE(n >= 0 && n <= IVX_NMS_MAX_N, "ivx_multiclass_nms_bev: n must be in 0..%d (got %d)", IVX_NMS_MAX_N, n);
  IVX_REQUIRE(num_classes >= 1 && num_classes <= 64, "ivx_multiclass_nms_bev: num_classes must be in 1..64 (got %d)", num_classes);
  const int ns = n > 0 ? (n + 63) / 64 * 64 : 64;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { const int64_t o = off; off += ivx_align_up(bytes, 256); return o; };
  const int64_t o_sidx = take((int64_t)num_classes * ns * 4), o_ss = take((int64_t)num_classes * ns * 4);
  const int64_t o_cb = take((int64_t)num_classes * ns * 5 * 4), o_na = take((int64_t)num_classes * 4);
  const int64_t o_kept = take((int64_t)num_classes * ns * 4), o_nk = take((int64_t)num_classes * 4);
  const int64_t o_ks = take((int64_t)num_classes * ns * 4);
  const bool big = n > 4096;
  const int64_t o_mask = take((int64_t)(big ? 1 : num_classes) * ns * (ns / 64) * 8);     // big: one shared hit matrix
  const int64_t o_keys = big ? take((int64_t)num_classes * ns * 8) : 0, o_rank = big ? take((int64_t)num_classes * ns * 4) : 0;
  if (o_keys_out) *o_keys_out = o_keys;
  if (o_rank_out) *o_rank_out = o_rank;
  if (p) {
    p->ns = ns;
    p->npad = next_pow2(n > 1 ? n : 2);
    p->sidx = (int *)o_sidx; p->sscore = (float *)o_ss; p->cboxes = (float *)o_cb; p->n_arr = (int *)o_na;
    p->kept = (int *)o_kept; p->nk = (int *)o_nk; p->mask = (unsigned long long *)o_mask; p->kscore = (float *)o_ks;
  }
  *total = off;
  return IVX_OK;
}

extern "C" int64_t ivx_multiclass_nms_workspace_bytes(int32_t n, int32_t num_classes) {
  int64_t t = 0;
  if (mc_layout(n, num_classes, nullptr, &t) != IVX_OK) return -1;
  return t;
}

extern "C" int ivx_multiclass_nms_bev(const float *boxes, const float *scores, int32_t n, int32_t score_stride, int32_t num_classes,
                                      float score_thr, float nms_thr, int32_t rotated, int32_t max_num, void *workspace,
                                      int64_t workspace_bytes, int64_t *out_idx, int64_t *out_label, int32_t *out_count,
                                      ivx_stream_t stream) {
  McP p;
  int64_t need = 0, o_keys = 0, o_rank = 0;
  if (mc_layout(n, num_classes, &p, &need, &o_keys, &o_rank) != IVX_OK) return IVX_ERR_INVALID_ARG;
  IVX_REQUIRE(out_idx && out_label && out_count, "ivx_multiclass_nms_bev: null output");
  IVX_REQUIRE(score_stride >= num_classes && max_num > 0, "ivx_multiclass_nms_bev: bad score_stride / max_num");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(out_count, 0, sizeof(int32_t), st);
    if (e != hipSuccess) {
      ivx_set_error("ivx_multiclass_nms_bev: memset failed: %s", hipGetErrorString(e));
      return IVX_ERR_HIP;
    }
    return IVX_OK;
  }
  IVX_REQUIRE(boxes && scores && workspace, "ivx_multiclass_nms_bev: null argument");
  if (workspace_bytes < need) {
    ivx_set_error("ivx_multiclass_nms_bev: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
    return IVX_ERR_WORKSPACE;
  }
  char *w = (char *)workspace;
  p.sidx = (int *)(w + (int64_t)p.sidx); p.sscore = (float *)(w + (int64_t)p.sscore); p.cboxes = (float *)(w + (int64_t)p.cboxes);
  p.n_arr = (int *)(w + (int64_t)p.n_arr); p.kept = (int *)(w + (int64_t)p.kept); p.nk = (int *)(w + (int64_t)p.nk);
  p.kscore = (float *)(w + (int64_t)p.kscore);
  p.mask = (unsigned long long *)(w + (int64_t)p.mask);
  p.boxes = boxes; p.scores = scores; p.n = n; p.score_stride = score_stride; p.num_classes = num_classes; p.max_num = max_num;
  p.rotated = rotated; p.score_thr = score_thr; p.nms_thr = nms_thr;
  const int cb = p.ns / 64;
  if (n > 4096) {
    unsigned long long *keys = (unsigned long long *)(w + o_keys);
    int *rank = (int *)(w + o_rank);
    if (hipMemsetAsync(p.n_arr, 0, (size_t)num_classes * 4, st) != hipSuccess) {
      ivx_set_error("ivx_multiclass_nms_bev: memset failed");
      return IVX_ERR_HIP;
    }
    const dim3 g((n + 255) / 256, num_classes);
    hipLaunchKernelGGL(mc_keys_big_kernel, g, dim3(256), 0, st, p, keys);
    hipLaunchKernelGGL(rank_sort_kernel, g, dim3(256), 0, st, keys, n, p.ns, rank);
    hipLaunchKernelGGL(mc_scatter_big_kernel, g, dim3(256), 0, st, p, keys, rank);
    hipLaunchKernelGGL(nms_hit_full_kernel, dim3(cb, n > 65535 ? 65535 : n), dim3(64), 0, st, boxes, n, cb, nms_thr, rotated, p.mask);
    hipLaunchKernelGGL(mc_scan_big_kernel, dim3(num_classes), dim3(64), 0, st, p);
    const int cap = p.ns < max_num ? p.ns : max_num;
    const unsigned blocks = (unsigned)(((long long)num_classes * cap + 255) / 256);
    hipLaunchKernelGGL(mc_finalize_kernel, dim3(blocks), dim3(256), 0, st, p, (long long *)out_idx, (long long *)out_label, out_count);
    IVX_CHECK_LAUNCH("ivx_multiclass_nms_bev");
    return IVX_OK;
  }
  hipLaunchKernelGGL(mc_select_sort_kernel, dim3(num_classes), dim3(1024), (size_t)p.npad * 8, st, p);
  const int shared = num_classes >= 3;
  if (shared)
    hipLaunchKernelGGL(nms_hit_full_kernel, dim3(cb, n > 65535 ? 65535 : n), dim3(64), 0, st, boxes, n, cb, nms_thr, rotated, p.mask);
  else
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, n > 65535 ? 65535 : n, num_classes), dim3(64), 0, st, p.cboxes, (const int *)p.n_arr, 0, p.ns, cb, nms_thr,
                       rotated, p.mask);
  hipLaunchKernelGGL(mc_scan_kernel, dim3(num_classes), dim3(64), 0, st, p, shared);
  {
    const int cap = p.ns < max_num ? p.ns : max_num;
    const unsigned blocks = (unsigned)(((long long)num_classes * cap + 255) / 256);
    hipLaunchKernelGGL(mc_finalize_kernel, dim3(blocks), dim3(256), 0, st, p, (long long *)out_idx, (long long *)out_label, out_count);
  }
  IVX_CHECK_LAUNCH("ivx_multiclass_nms_bev");
  return IVX_OK;
}

__global__ void overlap_pairs_kernel(const float *a, int na, const float *b, int nb, int iou, float *out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= nb || i >= na) return;
  float ba[5], bb[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    ba[q] = a[(size_t)i * 5 + q];
    bb[q] = b[(size_t)j * 5 + q];
  }
  out[(size_t)i * nb + j] = iou ? iou_bev_dev(ba, bb) : box_overlap_dev(ba, bb);
}

extern "C" int ivx_boxes_overlap_bev(const float *a, int32_t na, const float *b, int32_t nb, int32_t iou, float *out,
                                     ivx_stream_t stream) {
  IVX_REQUIRE(na >= 0 && nb >= 0 && na <= 65535, "ivx_boxes_overlap_bev: bad sizes");
  if (na == 0 || nb == 0) return IVX_OK;
  IVX_REQUIRE(a && b && out, "ivx_boxes_overlap_bev: null argument");
  hipLaunchKernelGGL(overlap_pairs_kernel, dim3((nb + 63) / 64, na), dim3(64), 0, (hipStream_t)stream, a, na, b, nb, iou, out);
  IVX_CHECK_LAUNCH("ivx_boxes_overlap_bev");
  return IVX_OK;
}

// ------------------------------------------------------------------------------------------------
// aligned_3d_nms (box3d_nms.py:91-138): boxes in descending score order; a box is picked iff no earlier
// picked box of the same class has IoU > thresh with it (NaN IoU suppresses, as `iou <= thresh` is false).
__device__ void aligned_nms_body(const float *boxes, const float *scores, const long long *classes, int n, int npad, float thresh,
                                 long long *pick, int *num_out, unsigned long long *sk /* LDS: npad keys, then npad flag bytes */) {
  unsigned char *supp = reinterpret_cast<unsigned char *>(sk + npad);
  __shared__ int s_np;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < npad; i += nt) {
    sk[i] = i < n ? (((unsigned long long)f2key(scores[i]) << 32) | (unsigned int)(~(unsigned int)i)) : 0ULL;
    supp[i] = 0;
  }
  if (tid == 0) s_np = 0;
  __syncthreads();
  bitonic_sort_desc(sk, npad);
  for (int i = 0; i < n; ++i) {
    if (supp[i]) continue;  // uniform: LDS value, synchronised by the barrier below
    const int bi = (int)(~(unsigned int)(sk[i] & 0xffffffffULL));
    if (tid == 0) pick[s_np++] = bi;
    const float *B1 = boxes + (size_t)bi * 6;
    const float x1 = B1[0], y1 = B1[1], z1 = B1[2], x2 = B1[3], y2 = B1[4], z2 = B1[5];
    const float ai = (x2 - x1) * (y2 - y1) * (z2 - z1);
    const long long ci = classes[bi];
    for (int j = i + 1 + tid; j < n; j += nt) {
      if (supp[j]) continue;
      const int bj = (int)(~(unsigned int)(sk[j] & 0xffffffffULL));
      const float *B2 = boxes + (size_t)bj * 6;
      const float xx1 = fmaxf(x1, B2[0]), yy1 = fmaxf(y1, B2[1]), zz1 = fmaxf(z1, B2[2]);
      const float xx2 = fminf(x2, B2[3]), yy2 = fminf(y2, B2[4]), zz2 = fminf(z2, B2[5]);
      const float il = fmaxf(0.f, xx2 - xx1), iw = fmaxf(0.f, yy2 - yy1), ih = fmaxf(0.f, zz2 - zz1);
      const float inter = il * iw * ih;
      const float aj = (B2[3] - B2[0]) * (B2[4] - B2[1]) * (B2[5] - B2[2]);
      float iou = inter / (ai + aj - inter);
      iou = iou * (ci == classes[bj] ? 1.0f : 0.0f);
      if (!(iou <= thresh)) supp[j] = 1;
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) *num_out = s_np;
}

__global__ __launch_bounds__(1024) void aligned_nms_kernel(const float *boxes, const float *scores, const long long *classes,
                                                           int n, int npad, float thresh, long long *pick, int *num_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  aligned_nms_body(boxes, scores, classes, n, npad, thresh, pick, num_out, sk);
}

extern "C" int ivx_aligned_3d_nms(const float *boxes, const float *scores, const int64_t *classes, int32_t n, float thresh,
                                  int64_t *pick, int32_t *num_out, ivx_stream_t stream) {
  IVX_REQUIRE(n >= 0 && n <= 4096, "ivx_aligned_3d_nms: n must be in 0..4096 (got %d)", n);
  IVX_REQUIRE(pick && num_out, "ivx_aligned_3d_nms: null output");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_out, 0, sizeof(int32_t), st);
    if (e != hipSuccess) {
      ivx_set_error("ivx_aligned_3d_nms: memset failed: %s", hipGetErrorString(e));
      return IVX_ERR_HIP;
    }
    return IVX_OK;
  }
  IVX_REQUIRE(boxes && scores && classes, "ivx_aligned_3d_nms: null argument");
  const int npad = next_pow2(n < 64 ? 64 : n);
  hipLaunchKernelGGL(aligned_nms_kernel, dim3(1), dim3(1024), (size_t)npad * 9, st, boxes, scores, (const long long *)classes, n, npad,
                     thresh, (long long *)pick, num_out);
  IVX_CHECK_LAUNCH("ivx_aligned_3d_nms");
  return IVX_OK;
}

// ---- the same NMS, class-parallel (ivx_aligned_3d_nms_ws).  Boxes of different classes never suppress each other (their IoU
// is multiplied by 0), so the greedy chain -- one dependent step per kept box, ~1 us each in the kernel above, 1.0 ms for the
// 3 000 candidates of a ScanNet scene -- splits into independent chains per class:
//   aligned_sort_kernel:      sort by score (same composite key), boxes / classes / source indices laid out in sorted order,
//                             and a regularity check (finite corners, every extent in (0, 1e6))
//   aligned_class_nms_kernel: workgroup w takes the sorted candidates with class % 64 == w into LDS and runs the greedy chain
//                             among them (the class test stays inside, so any class values are handled)
//   aligned_collect_kernel:   kept flags -> pick list in sorted (descending score) order
// The split relies on iou * 0 == 0, i.e. on a finite IoU: a degenerate pair (0 / 0, inf / inf) gives NaN, which the reference's
// `iou <= thresh` treats as suppression ACROSS classes.  With every extent in (0, 1e6) the union is positive and finite, so
// when the check fails the collect kernel runs the one-workgroup form above instead -- same picks in every case.
#define ANMS_WG 64
#define ANMS_STAGE 1536
struct AnmsScratch {      // offsets into the caller's workspace
  float *sbox;            // [n][6] sorted
  long long *scls;        // [n]
  int *sidx;              // [n]
  unsigned char *keep;    // [npad]
  int *irregular;         // [1]
};

__global__ __launch_bounds__(1024) void aligned_sort_kernel(const float *boxes, const float *scores, const long long *classes, int n,
                                                            int npad, AnmsScratch w) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];  // npad keys
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < npad; i += nt)
    sk[i] = i < n ? (((unsigned long long)f2key(scores[i]) << 32) | (unsigned int)(~(unsigned int)i)) : 0ULL;
  if (tid == 0) *w.irregular = 0;
  __syncthreads();
  bitonic_sort_desc(sk, npad);
  int bad = 0;
  for (int i = tid; i < npad; i += nt) {
    w.keep[i] = 0;
    if (i >= n) continue;
    const int bi = (int)(~(unsigned int)(sk[i] & 0xffffffffULL));
    const float *B = boxes + (size_t)bi * 6;
    float *o = w.sbox + (size_t)i * 6;
    const float x1 = B[0], y1 = B[1], z1 = B[2], x2 = B[3], y2 = B[4], z2 = B[5];
    o[0] = x1; o[1] = y1; o[2] = z1; o[3] = x2; o[4] = y2; o[5] = z2;
    w.scls[i] = classes[bi];
    w.sidx[i] = bi;
    const float ex = x2 - x1, ey = y2 - y1, ez = z2 - z1;
    const bool fin = fabsf(x1) < 1e30f && fabsf(y1) < 1e30f && fabsf(z1) < 1e30f && fabsf(x2) < 1e30f && fabsf(y2) < 1e30f && fabsf(z2) < 1e30f;
    if (!(fin && ex > 0.f && ey > 0.f && ez > 0.f && ex < 1e6f && ey < 1e6f && ez < 1e6f)) bad = 1;   // (NaN fails every test)
  }
  if (bad) atomicOr(w.irregular, 1);
}

__global__ __launch_bounds__(256) void aligned_class_nms_kernel(int n, float thresh, AnmsScratch w) {
  __shared__ int pos[4096];                     // sorted positions of this workgroup's candidates, ascending
  __shared__ float lb[ANMS_STAGE * 6];
  __shared__ long long lc[ANMS_STAGE];
  __shared__ unsigned char supp[4096];
  __shared__ int s_cnt, wave_cnt[4];
  if (*w.irregular) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {     // order-preserving compaction of {i : class % 64 == this workgroup}
    const int i = base + tid;
    const bool mine = i < n && (int)((unsigned long long)w.scls[i] % ANMS_WG) == (int)blockIdx.x;
    const unsigned long long m = __ballot(mine);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = s_cnt;
    for (int q = 0; q < wv; ++q) off += wave_cnt[q];
    if (mine) pos[off + __popcll(m & ((1ULL << lane) - 1ULL))] = i;
    __syncthreads();
    if (tid == 0) s_cnt += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  const int nc = s_cnt;
  if (nc == 0) return;
  const bool staged = nc <= ANMS_STAGE;
  for (int j = tid; j < nc; j += 256) {
    supp[j] = 0;
    if (staged) {
      const float *B = w.sbox + (size_t)pos[j] * 6;
#pragma unroll
      for (int q = 0; q < 6; ++q) lb[j * 6 + q] = B[q];
      lc[j] = w.scls[pos[j]];
    }
  }
  __syncthreads();
  for (int i = 0; i < nc; ++i) {
    if (supp[i]) continue;                       // uniform: LDS value, synchronised by the barrier below
    if (tid == 0) w.keep[pos[i]] = 1;
    const float *B1 = staged ? lb + i * 6 : w.sbox + (size_t)pos[i] * 6;
    const float x1 = B1[0], y1 = B1[1], z1 = B1[2], x2 = B1[3], y2 = B1[4], z2 = B1[5];
    const float ai = (x2 - x1) * (y2 - y1) * (z2 - z1);
    const long long ci = staged ? lc[i] : w.scls[pos[i]];
    for (int j = i + 1 + tid; j < nc; j += 256) {
      if (supp[j]) continue;
      const float *B2 = staged ? lb + j * 6 : w.sbox + (size_t)pos[j] * 6;
      const float xx1 = fmaxf(x1, B2[0]), yy1 = fmaxf(y1, B2[1]), zz1 = fmaxf(z1, B2[2]);
      const float xx2 = fminf(x2, B2[3]), yy2 = fminf(y2, B2[4]), zz2 = fminf(z2, B2[5]);
      const float il = fmaxf(0.f, xx2 - xx1), iw = fmaxf(0.f, yy2 - yy1), ih = fmaxf(0.f, zz2 - zz1);
      const float inter = il * iw * ih;
      const float aj = (B2[3] - B2[0]) * (B2[4] - B2[1]) * (B2[5] - B2[2]);
      float iou = inter / (ai + aj - inter);
      iou = iou * (ci == (staged ? lc[j] : w.scls[pos[j]]) ? 1.0f : 0.0f);
      if (!(iou <= thresh)) supp[j] = 1;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void aligned_collect_kernel(const float *boxes, const float *scores, const long long *classes, int n,
                                                               int npad, float thresh, AnmsScratch w, long long *pick, int *num_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];  // fallback: npad keys + npad flags; else 1024 ints
  if (*w.irregular) {                     // degenerate boxes: the exact one-workgroup form
    aligned_nms_body(boxes, scores, classes, n, npad, thresh, pick, num_out, sk);
    return;
  }
  int *cnt = reinterpret_cast<int *>(sk);   // per-thread kept counts -> exclusive offsets
  const int tid = threadIdx.x;
  const int per = npad / 1024 > 0 ? npad / 1024 : 1;          // npad is a power of two >= 64
  const int i0 = tid * per;
  int c = 0;
  for (int q = 0; q < per; ++q) c += (i0 + q < n && w.keep[i0 + q]) ? 1 : 0;
  cnt[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {                  // inclusive scan (Hillis-Steele)
    const int v = tid >= off ? cnt[tid - off] : 0;
    __syncthreads();
    cnt[tid] += v;
    __syncthreads();
  }
  int o = cnt[tid] - c;
  for (int q = 0; q < per; ++q)
    if (i0 + q < n && w.keep[i0 + q]) pick[o++] = w.sidx[i0 + q];
  if (tid == 1023) *num_out = cnt[1023];
}

// ---- n > 4096: score order by keys + rank sort, suppression mask over the sorted candidates (bit = NOT (iou * same_class <= thresh),
// the reference's own predicate, box3d_nms.py:131-137: NaN IoUs suppress across classes without any special case), greedy scan with
// the removal bits in LDS; pick[j] = original index of the j-th kept candidate.
__global__ __launch_bounds__(256) void aligned_keys_big_kernel(const float *scores, int n, unsigned long long *keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = ((unsigned long long)f2key(scores[i]) << 32) | (unsigned int)(~(unsigned int)i);
}

__global__ __launch_bounds__(256) void aligned_scatter_big_kernel(const float *boxes, const long long *classes, int n, const int *rank,
                                                                  AnmsScratch w) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = rank[i];
#pragma unroll
  for (int q = 0; q < 6; ++q) w.sbox[(size_t)j * 6 + q] = boxes[(size_t)i * 6 + q];
  w.scls[j] = classes[i];
  w.sidx[j] = i;
}

__global__ __launch_bounds__(64) void aligned_mask_big_kernel(int n, int cb, float thresh, AnmsScratch w, unsigned long long *mask) {
  const int col_start = blockIdx.x;
  const int col = col_start * 64 + threadIdx.x;
  for (int row = blockIdx.y; row < n; row += gridDim.y) {      // (grid.y stops at 65535; n may be 65536)
  if (col_start < (row >> 6)) continue;
  bool hit = false;
  if (col < n && col > row) {
    const float *B1 = w.sbox + (size_t)row * 6, *B2 = w.sbox + (size_t)col * 6;
    const float x1 = B1[0], y1 = B1[1], z1 = B1[2], x2 = B1[3], y2 = B1[4], z2 = B1[5];
    const float ai = (x2 - x1) * (y2 - y1) * (z2 - z1);
    const float xx1 = fmaxf(x1, B2[0]), yy1 = fmaxf(y1, B2[1]), zz1 = fmaxf(z1, B2[2]);
    const float xx2 = fminf(x2, B2[3]), yy2 = fminf(y2, B2[4]), zz2 = fminf(z2, B2[5]);
    const float il = fmaxf(0.f, xx2 - xx1), iw = fmaxf(0.f, yy2 - yy1), ih = fmaxf(0.f, zz2 - zz1);
    const float inter = il * iw * ih;
    const float aj = (B2[3] - B2[0]) * (B2[4] - B2[1]) * (B2[5] - B2[2]);
    float iou = inter / (ai + aj - inter);
    iou = iou * (w.scls[row] == w.scls[col] ? 1.0f : 0.0f);
    hit = !(iou <= thresh);
  }
  const unsigned long long m = __ballot(hit);
  if (threadIdx.x == 0) mask[(size_t)row * cb + col_start] = m;
  }
}

static int64_t anms_big_bytes(int64_t n, int64_t *o_keys, int64_t *o_rank, int64_t *o_mask) {
  const int64_t cb = (n + 63) / 64;
  int64_t off = ivx_align_up(n * 24, 256) + ivx_align_up(n * 8, 256) + ivx_align_up(n * 4, 256);
  if (o_keys) *o_keys = off;
  off += ivx_align_up(n * 8, 256);
  if (o_rank) *o_rank = off;
  off += ivx_align_up(n * 4, 256);
  if (o_mask) *o_mask = off;
  off += ivx_align_up(n * cb * 8, 256);
  return off;
}

extern "C" int64_t ivx_aligned_3d_nms_workspace_bytes(int32_t n) {
  if (n < 0 || n > IVX_NMS_MAX_N) return -1;
  if (n > 4096) return anms_big_bytes(n, nullptr, nullptr, nullptr);
  const int64_t npad = next_pow2(n < 64 ? 64 : n);
  return ivx_align_up(npad * 24, 256) + ivx_align_up(npad * 8, 256) + ivx_align_up(npad * 4, 256) + ivx_align_up(npad, 256) + 256;
}

extern "C" int ivx_aligned_3d_nms_ws(const float *boxes, const float *scores, const int64_t *classes, int32_t n, float thresh,
                                     void *workspace, int64_t workspace_bytes, int64_t *pick, int32_t *num_out, ivx_stream_t stream) {
  IVX_REQUIRE(n >= 0 && n <= IVX_NMS_MAX_N, "ivx_aligned_3d_nms_ws: n must be in 0..%d (got %d)", IVX_NMS_MAX_N, n);
  IVX_REQUIRE(pick && num_out, "ivx_aligned_3d_nms_ws: null output");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_out, 0, sizeof(int32_t), st);
    if (e != hipSuccess) {
      ivx_set_error("ivx_aligned_3d_nms_ws: memset failed: %s", hipGetErrorString(e));
      return IVX_ERR_HIP;
    }
    return IVX_OK;
  }
  IVX_REQUIRE(boxes && scores && classes && workspace, "ivx_aligned_3d_nms_ws: null argument");
  IVX_REQUIRE(((uintptr_t)workspace & 255) == 0, "ivx_aligned_3d_nms_ws: workspace must be 256-byte aligned");
  if (workspace_bytes < ivx_aligned_3d_nms_workspace_bytes(n)) {
    ivx_set_error("ivx_aligned_3d_nms_ws: workspace too small");
    return IVX_ERR_WORKSPACE;
  }
  if (n > 4096) {
    int64_t o_keys, o_rank, o_mask;
    anms_big_bytes(n, &o_keys, &o_rank, &o_mask);
    char *b = (char *)workspace;
    AnmsScratch w;
    w.sbox = (float *)b;
    w.scls = (long long *)(b + ivx_align_up((int64_t)n * 24, 256));
    w.sidx = (int *)(b + ivx_align_up((int64_t)n * 24, 256) + ivx_align_up((int64_t)n * 8, 256));
    w.keep = nullptr; w.irregular = nullptr;
    unsigned long long *keys = (unsigned long long *)(b + o_keys), *mask = (unsigned long long *)(b + o_mask);
    int *rank = (int *)(b + o_rank);
    const int cb = (n + 63) / 64;
    const dim3 g((n + 255) / 256);
    hipLaunchKernelGGL(aligned_keys_big_kernel, g, dim3(256), 0, st, scores, n, keys);
    hipLaunchKernelGGL(rank_sort_kernel, dim3(g.x, 1), dim3(256), 0, st, keys, n, n, rank);
    hipLaunchKernelGGL(aligned_scatter_big_kernel, g, dim3(256), 0, st, boxes, (const long long *)classes, n, rank, w);
    hipLaunchKernelGGL(aligned_mask_big_kernel, dim3(cb, n > 65535 ? 65535 : n), dim3(64), 0, st, n, cb, thresh, w, mask);
    hipLaunchKernelGGL(nms_scan_big_kernel, dim3(1), dim3(64), 0, st, mask, (const int *)nullptr, n, cb, (const int *)w.sidx, (long long *)pick,
                       num_out);
    IVX_CHECK_LAUNCH("ivx_aligned_3d_nms_ws");
    return IVX_OK;
  }
  const int npad = next_pow2(n < 64 ? 64 : n);
  AnmsScratch w;
  char *o = (char *)workspace;
  w.sbox = (float *)o; o += ivx_align_up((int64_t)npad * 24, 256);
  w.scls = (long long *)o; o += ivx_align_up((int64_t)npad * 8, 256);
  w.sidx = (int *)o; o += ivx_align_up((int64_t)npad * 4, 256);
  w.keep = (unsigned char *)o; o += ivx_align_up((int64_t)npad, 256);
  w.irregular = (int *)o;
  hipLaunchKernelGGL(aligned_sort_kernel, dim3(1), dim3(1024), (size_t)npad * 8, st, boxes, scores, (const long long *)classes, n, npad, w);
  hipLaunchKernelGGL(aligned_class_nms_kernel, dim3(ANMS_WG), dim3(256), 0, st, n, thresh, w);
  const size_t lds = (size_t)npad * 9 > 4096 ? (size_t)npad * 9 : 4096;
  hipLaunchKernelGGL(aligned_collect_kernel, dim3(1), dim3(1024), lds, st, boxes, scores, (const long long *)classes, n, npad, thresh, w,
                     (long long *)pick, num_out);
  IVX_CHECK_LAUNCH("ivx_aligned_3d_nms_ws");
  return IVX_OK;
}

// ------------------------------------------------------------------------------------------------
// Indoor (FCOS-style) head tail: ImVoxelHeadV2._get_bboxes_single per level
// (mmdet3d/models/dense_heads/imvoxel_head_v2.py:216-285, 305-313 / 444-449, 419-438 / 547-555):
// trilinear-resized valid mask (.round() => >= 5 of the 8 contributing level-0 voxels), sigmoid scores
// cls * centerness * valid, top-k(nms_pre) on the class maximum, level points, distance decoding.
struct FcosP {
  const float *head_out;     // [B, n, CH]  fused conv output: [centerness | reg (R) | cls (ncls)]
  const uint8_t *valid0;     // [B, X, Y, Z] level-0 valid mask
  const float *vs;           // [B, 3] level voxel size  (voxel_size * 2^level, fp32, built on the host)
  const float *new_origin;   // [B, 3] origin - n_level/2 * vs_level
  float scale;               // mmcv Scale parameter of this level
  int B, nx, ny, nz, n, CH, ncls, R, level, X, Y, Z, kpad, k;
};

__device__ inline float fcos_valid(const FcosP &p, int b, int i) {
  const int iz = i % p.nz;
  const int t = i / p.nz;
  const int iy = t % p.ny, ix = t / p.ny;
  const uint8_t *v = p.valid0 + (size_t)b * p.X * p.Y * p.Z;
  if (p.level == 0) return v[((size_t)ix * p.Y + iy) * p.Z + iz] ? 1.0f : 0.0f;
  const int h = (1 << (p.level - 1)) - 1;
  const int x0 = (ix << p.level) + h, y0 = (iy << p.level) + h, z0 = (iz << p.level) + h;
  int cnt = 0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int f = 0; f < 2; ++f) cnt += v[((size_t)(x0 + a) * p.Y + (y0 + e)) * p.Z + (z0 + f)] ? 1 : 0;
  return cnt >= 5 ? 1.0f : 0.0f;   // mean of 8 samples, torch.round is half-to-even: 4/8 -> 0
}

__global__ __launch_bounds__(256) void fcos_scores_kernel(const FcosP p, float *keys) {
  const size_t total = (size_t)p.B * p.n;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / p.n), i = (int)(idx % p.n);
    const float *row = p.head_out + idx * p.CH;
    const float ctr = sigmoid_ref(row[0]);
    const float vf = fcos_valid(p, b, i);
    float m = -1.0f;
    for (int c = 0; c < p.ncls; ++c) {
      const float s = (sigmoid_ref(row[1 + p.R + c]) * ctr) * vf;
      m = s > m ? s : m;
    }
    keys[idx] = m;
  }
}

__global__ __launch_bounds__(64) void fcos_decode_kernel(const FcosP p, const int *topk_idx, float *cand_boxes, float *cand_scores) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= p.k) return;
  const int i = topk_idx[(size_t)b * p.kpad + j];
  float *ob = cand_boxes + ((size_t)b * p.k + j) * p.R;
  float *os = cand_scores + ((size_t)b * p.k + j) * p.ncls;
  if (i < 0) {
    for (int q = 0; q < p.R; ++q) ob[q] = 0.f;
    for (int c = 0; c < p.ncls; ++c) os[c] = 0.f;
    return;
  }
  const int iz = i % p.nz;
  const int t = i / p.nz;
  const int iy = t % p.ny, ix = t / p.ny;
  const float *vs = p.vs + b * 3, *no = p.new_origin + b * 3;
  const float px = (float)ix * vs[0] + no[0], py = (float)iy * vs[1] + no[1], pz = (float)iz * vs[2] + no[2];
  const float *row = p.head_out + ((size_t)b * p.n + i) * p.CH;
  float d[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) d[q] = expf(row[1 + q] * p.scale);
  if (p.R == 6) {   // ScanNet: axis-aligned corners (imvoxel_head_v2.py:547-555)
    ob[0] = px - d[0]; ob[1] = py - d[2]; ob[2] = pz - d[4];
    ob[3] = px + d[1]; ob[4] = py + d[3]; ob[5] = pz + d[5];
  } else {          // SUN RGB-D: rotated box (imvoxel_head_v2.py:419-438, rotation_3d_in_axis(axis=2))
    const float alpha = row[1 + 6];
    const float sx = (d[1] - d[0]) / 2, sy = (d[3] - d[2]) / 2, sz = (d[5] - d[4]) / 2;
    const float c = cosf(alpha), s = sinf(alpha);
    ob[0] = px + (sx * c + sy * s);
    ob[1] = py + (-sx * s + sy * c);
    ob[2] = pz + sz;
    ob[3] = d[0] + d[1]; ob[4] = d[2] + d[3]; ob[5] = d[4] + d[5];
    ob[6] = alpha;
  }
  const float ctr = sigmoid_ref(row[0]);
  const float vf = fcos_valid(p, b, i);
  for (int c = 0; c < p.ncls; ++c) os[c] = (sigmoid_ref(row[1 + p.R + c]) * ctr) * vf;
}

extern "C" int64_t ivx_fcos_head_workspace_bytes(int32_t B, int32_t n, int32_t nms_pre) {
  if (B <= 0 || n <= 0) return -1;
  const int k = (nms_pre > 0 && nms_pre < n) ? nms_pre : n;
  if (k > IVX_NMS_MAX_N) return -1;
  const int kpad = k > 4096 ? (k + 63) / 64 * 64 : next_pow2(k < 64 ? 64 : k);
  return ivx_align_up((int64_t)B * n * 4, 256) + ivx_align_up((int64_t)B * kpad * 4, 256) + 2 * ivx_align_up((int64_t)B * 4, 256) +
         (k > 4096 ? topk_scratch_big_bytes(B, n) : topk_scratch_bytes(B));
}

extern "C" int ivx_fcos_head_level_candidates(const float *head_out, const uint8_t *valid0, const float *level_vs,
                                              const float *level_new_origin, float scale, int32_t B, int32_t nx, int32_t ny,
                                              int32_t nz, int32_t CH, int32_t n_classes, int32_t n_reg, int32_t level, int32_t X,
                                              int32_t Y, int32_t Z, int32_t nms_pre, void *workspace, int64_t workspace_bytes,
                                              float *cand_boxes, float *cand_scores, int32_t *cand_count, ivx_stream_t stream) {
  IVX_REQUIRE(head_out && valid0 && level_vs && level_new_origin && workspace && cand_boxes && cand_scores && cand_count,
              "ivx_fcos_head_level_candidates: null argument");
  IVX_REQUIRE(B > 0 && nx > 0 && ny > 0 && nz > 0 && n_classes > 0 && (n_reg == 6 || n_reg == 7) && CH >= 1 + n_reg + n_classes,
              "ivx_fcos_head_level_candidates: bad dims");
  IVX_REQUIRE(level >= 0 && level < 8 && (nx << level) == X && (ny << level) == Y && (nz << level) == Z,
              "ivx_fcos_head_level_candidates: level grid must be the level-0 grid / 2^level");
  const int64_t n64 = (int64_t)nx * ny * nz;
  IVX_REQUIRE(n64 < (1LL << 31), "ivx_fcos_head_level_candidates: grid too large");
  const int n = (int)n64;
  const int k = (nms_pre > 0 && nms_pre < n) ? nms_pre : n;
  IVX_REQUIRE(k <= IVX_NMS_MAX_N, "ivx_fcos_head_level_candidates: at most %d candidates per level (got %d)", IVX_NMS_MAX_N, k);
  const int64_t need = ivx_fcos_head_workspace_bytes(B, n, nms_pre);
  if (workspace_bytes < need) {
    ivx_set_error("ivx_fcos_head_level_candidates: workspace too small");
    return IVX_ERR_WORKSPACE;
  }
  IVX_REQUIRE(((uintptr_t)workspace & 255) == 0, "ivx_fcos_head_level_candidates: workspace must be 256-byte aligned");
  FcosP p;
  p.head_out = head_out; p.valid0 = valid0; p.vs = level_vs; p.new_origin = level_new_origin; p.scale = scale;
  p.B = B; p.nx = nx; p.ny = ny; p.nz = nz; p.n = n; p.CH = CH; p.ncls = n_classes; p.R = n_reg; p.level = level;
  p.X = X; p.Y = Y; p.Z = Z; p.k = k; p.kpad = k > 4096 ? (k + 63) / 64 * 64 : next_pow2(k < 64 ? 64 : k);
  char *ws = (char *)workspace;
  float *keys = (float *)ws;
  int *topk = (int *)(ws + ivx_align_up((int64_t)B * n * 4, 256));
  int *n1 = (int *)(ws + ivx_align_up((int64_t)B * n * 4, 256) + ivx_align_up((int64_t)B * p.kpad * 4, 256));
  const int64_t n1_bytes = 2 * ivx_align_up((int64_t)B * 4, 256);
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)B * n;
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fcos_scores_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, keys);
  HeadP hp = {};
  hp.n = n; hp.nms_pre = nms_pre; hp.kpad = p.kpad; hp.score_thr = 0.f; hp.B = B;
  if (k > 4096) launch_topk_big(hp, keys, topk, cand_count, n1, (char *)n1 + n1_bytes, st);
  else launch_topk(hp, keys, topk, cand_count, n1, (char *)n1 + n1_bytes, st);
  hipLaunchKernelGGL(fcos_decode_kernel, dim3((k + 63) / 64, B), dim3(64), 0, st, p, topk, cand_boxes, cand_scores);
  IVX_CHECK_LAUNCH("ivx_fcos_head_level_candidates");
  return IVX_OK;
}

// ------------------------------------------------------------------------------------------------
// Indoor head tail after the per-level candidates: the cross-level NMS and the result rows of
// ImVoxelHeadV2._get_bboxes_single / _nms (mmdet3d/models/dense_heads/imvoxel_head_v2.py:258-277 torch.cat over levels,
// ScanNet _nms :528-545, SUN RGB-D _nms :397-417) on the device for a batch, nothing returned to the host in between.
//   ScanNet (n_reg 6):  class maximum + label per candidate, `score > score_thr` filter, class-aware aligned 3-D NMS, corners ->
//                       (centre, size), fake yaw 0;  candidates under the threshold enter the NMS with score -inf -- they sort
//                       behind every real candidate, so they can neither suppress one nor be picked before one, and the
//                       picks that are real form a prefix: same result as filtering first (box3d_nms.py:91-138 is greedy in
//                       descending score), with a host-known problem size
//   SUN RGB-D (n_reg 7): BEV boxes (x -+ dx/2, y -+ dy/2, alpha), fused multi-class NMS (ivx_multiclass_nms_bev), gather
// Output rows are the box object's tensor: (x, y, z_bottom, dx, dy, dz, yaw) -- the gravity-centre z moved to the bottom face
// exactly as BaseInstance3DBoxes.__init__(origin=(.5,.5,.5)) does (base_box3d.py:63-66: z += dz * (0 - 0.5)).
struct IndoorP {
  const float *cb[4];      // per level [B, k_l, R]
  const float *cs[4];      // per level [B, k_l, ncls]
  int k[4], koff[5];
  int B, L, K, R, ncls, max_num;
  float score_thr;
};

// ScanNet: concatenated corner boxes [B,K,6], class maximum (first maximum, as torch.max(dim=1)) as score (-inf when <= thr) + label
__global__ __launch_bounds__(256) void indoor_prep_scannet_kernel(const IndoorP p, float *boxes, float *scores, float *raw_scores,
                                                                  long long *labels) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.K) return;
  int l = 0;
  while (l + 1 < p.L && j >= p.koff[l + 1]) ++l;
  const int r = j - p.koff[l];
  const float *sb = p.cb[l] + ((size_t)b * p.k[l] + r) * 6;
  const float *ss = p.cs[l] + ((size_t)b * p.k[l] + r) * p.ncls;
  float best = ss[0];
  int lab = 0;
  for (int c = 1; c < p.ncls; ++c)
    if (ss[c] > best) { best = ss[c]; lab = c; }
  float *ob = boxes + ((size_t)b * p.K + j) * 6;
#pragma unroll
  for (int q = 0; q < 6; ++q) ob[q] = sb[q];
  raw_scores[(size_t)b * p.K + j] = best;
  scores[(size_t)b * p.K + j] = best > p.score_thr ? best : -__builtin_inff();
  labels[(size_t)b * p.K + j] = lab;
}

__global__ __launch_bounds__(256) void indoor_finish_scannet_kernel(const IndoorP p, const float *boxes, const float *raw_scores,
                                                                    const long long *labels, const long long *pick, const int *npick,
                                                                    float *out_boxes, float *out_scores, long long *out_labels, int *out_count) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.max_num) return;
  const int np = npick[b];
  float *ob = out_boxes + ((size_t)b * p.max_num + j) * 7;
  bool real = false;
  int i = 0;
  if (j < np) {
    i = (int)pick[(size_t)b * p.K + j];
    real = raw_scores[(size_t)b * p.K + i] > p.score_thr;
  }
  if (real) {
    const float *c = boxes + ((size_t)b * p.K + i) * 6;
    const float dz = c[5] - c[2];
    ob[0] = (c[0] + c[3]) / 2.f; ob[1] = (c[1] + c[4]) / 2.f;
    ob[2] = (c[2] + c[5]) / 2.f + dz * -0.5f;
    ob[3] = c[3] - c[0]; ob[4] = c[4] - c[1]; ob[5] = dz; ob[6] = 0.f;
    out_scores[(size_t)b * p.max_num + j] = raw_scores[(size_t)b * p.K + i];
    out_labels[(size_t)b * p.max_num + j] = labels[(size_t)b * p.K + i];
  } else {
#pragma unroll
    for (int q = 0; q < 7; ++q) ob[q] = 0.f;
    out_scores[(size_t)b * p.max_num + j] = 0.f;
    out_labels[(size_t)b * p.max_num + j] = 0;
  }
  // the real picks are a prefix of the pick list (descending score): count them once per sample
  if (j == 0) {
    int lo = 0, hi = np;            // first pick that is not real
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (raw_scores[(size_t)b * p.K + (int)pick[(size_t)b * p.K + mid]] > p.score_thr) lo = mid + 1; else hi = mid;
    }
    out_count[b] = lo < p.max_num ? lo : p.max_num;
  }
}

// SUN RGB-D: concatenated boxes [B,K,7], BEV boxes [B,K,5], scores [B,K,ncls]
__global__ __launch_bounds__(256) void indoor_prep_sunrgbd_kernel(const IndoorP p, float *boxes, float *bev, float *scores) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.K) return;
  int l = 0;
  while (l + 1 < p.L && j >= p.koff[l + 1]) ++l;
  const int r = j - p.koff[l];
  const float *sb = p.cb[l] + ((size_t)b * p.k[l] + r) * 7;
  const float *ss = p.cs[l] + ((size_t)b * p.k[l] + r) * p.ncls;
  float *ob = boxes + ((size_t)b * p.K + j) * 7;
#pragma unroll
  for (int q = 0; q < 7; ++q) ob[q] = sb[q];
  float *v = bev + ((size_t)b * p.K + j) * 5;
  v[0] = sb[0] - sb[3] / 2.f; v[1] = sb[1] - sb[4] / 2.f; v[2] = sb[0] + sb[3] / 2.f; v[3] = sb[1] + sb[4] / 2.f; v[4] = sb[6];
  for (int c = 0; c < p.ncls; ++c) scores[((size_t)b * p.K + j) * p.ncls + c] = ss[c];
}

__global__ __launch_bounds__(256) void indoor_finish_sunrgbd_kernel(const IndoorP p, const float *boxes, const float *scores, const long long *idx,
                                                                    const long long *lab, const int *cnt, float *out_boxes, float *out_scores,
                                                                    long long *out_labels, int *out_count) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.max_num) return;
  const int n = cnt[b];
  float *ob = out_boxes + ((size_t)b * p.max_num + j) * 7;
  if (j < n) {
    const int i = (int)idx[(size_t)b * p.max_num + j], c = (int)lab[(size_t)b * p.max_num + j];
    const float *s = boxes + ((size_t)b * p.K + i) * 7;
    ob[0] = s[0]; ob[1] = s[1]; ob[2] = s[2] + s[5] * -0.5f; ob[3] = s[3]; ob[4] = s[4]; ob[5] = s[5]; ob[6] = s[6];
    out_scores[(size_t)b * p.max_num + j] = scores[((size_t)b * p.K + i) * p.ncls + c];
    out_labels[(size_t)b * p.max_num + j] = c;
  } else {
#pragma unroll
    for (int q = 0; q < 7; ++q) ob[q] = 0.f;
    out_scores[(size_t)b * p.max_num + j] = 0.f;
    out_labels[(size_t)b * p.max_num + j] = 0;
  }
  if (j == 0) out_count[b] = n;
}

struct IndoorWs { int64_t boxes, bev, scores, raw, labels, pick, npick, nms, nms_bytes, total; };

static int indoor_layout(const ivx_indoor_tail_desc *d, IndoorP *p, IndoorWs *w) {
  IVX_REQUIRE(d, "ivx_indoor_tail: null descriptor");
  IVX_REQUIRE(d->B > 0 && d->n_levels >= 1 && d->n_levels <= 4 && d->n_classes >= 1 && (d->n_reg == 6 || d->n_reg == 7), "ivx_indoor_tail: bad dims");
  IVX_REQUIRE(d->n_reg == 6 || d->n_classes <= 64, "ivx_indoor_tail: the multi-class NMS takes at most 64 classes");
  p->B = d->B; p->L = d->n_levels; p->R = d->n_reg; p->ncls = d->n_classes; p->score_thr = d->score_thr;
  int K = 0;
  for (int l = 0; l < d->n_levels; ++l) {
    IVX_REQUIRE(d->k[l] > 0, "ivx_indoor_tail: level %d has no candidates", l);
    p->k[l] = d->k[l]; p->koff[l] = K; K += d->k[l];
  }
  p->koff[d->n_levels] = K;
  IVX_REQUIRE(K <= 65536, "ivx_indoor_tail: at most 65536 candidates per sample (got %d)", K);
  p->K = K;
  p->max_num = d->max_num;
  IVX_REQUIRE(d->max_num > 0, "ivx_indoor_tail: max_num must be positive (ScanNet: the total number of candidates; SUN RGB-D: test_cfg.nms_pre)");
  int64_t o = 0;
  auto take = [&](int64_t bytes) { const int64_t at = o; o += ivx_align_up(bytes, 256); return at; };
  w->boxes = take((int64_t)d->B * K * d->n_reg * 4);
  w->bev = d->n_reg == 7 ? take((int64_t)d->B * K * 5 * 4) : 0;
  w->scores = take((int64_t)d->B * K * (d->n_reg == 7 ? d->n_classes : 1) * 4);
  w->raw = d->n_reg == 6 ? take((int64_t)d->B * K * 4) : 0;
  w->labels = take((int64_t)d->B * (d->n_reg == 6 ? K : d->max_num) * 8);
  w->pick = take((int64_t)d->B * (d->n_reg == 6 ? K : d->max_num) * 8);
  w->npick = take((int64_t)d->B * 4);
  w->nms_bytes = d->n_reg == 6 ? ivx_aligned_3d_nms_workspace_bytes(K) : ivx_multiclass_nms_workspace_bytes(K, d->n_classes);
  IVX_REQUIRE(w->nms_bytes >= 0, "ivx_indoor_tail: %s", ivx_last_error());
  w->nms = take(w->nms_bytes);
  w->total = o;
  return IVX_OK;
}

extern "C" int64_t ivx_indoor_tail_workspace_bytes(const ivx_indoor_tail_desc *d) {
  IndoorP p;
  IndoorWs w;
  if (indoor_layout(d, &p, &w) != IVX_OK) return -1;
  return w.total;
}

extern "C" int ivx_indoor_tail_get_bboxes(const ivx_indoor_tail_desc *d, const float *const *cand_boxes, const float *const *cand_scores,
                                          void *workspace, int64_t workspace_bytes, float *out_boxes, float *out_scores, int64_t *out_labels,
                                          int32_t *out_count, ivx_stream_t stream) {
  IndoorP p;
  IndoorWs w;
  int rc = indoor_layout(d, &p, &w);
  if (rc != IVX_OK) return rc;
  IVX_REQUIRE(cand_boxes && cand_scores && workspace && out_boxes && out_scores && out_labels && out_count, "ivx_indoor_tail_get_bboxes: null argument");
  for (int l = 0; l < p.L; ++l) {
    IVX_REQUIRE(cand_boxes[l] && cand_scores[l], "ivx_indoor_tail_get_bboxes: null candidates of level %d", l);
    p.cb[l] = cand_boxes[l]; p.cs[l] = cand_scores[l];
  }
  IVX_REQUIRE(((uintptr_t)workspace & 255) == 0, "ivx_indoor_tail_get_bboxes: workspace must be 256-byte aligned");
  if (workspace_bytes < w.total) {
    ivx_set_error("ivx_indoor_tail_get_bboxes: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)w.total);
    return IVX_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char *ws = (char *)workspace;
  float *boxes = (float *)(ws + w.boxes), *scores = (float *)(ws + w.scores);
  long long *labels = (long long *)(ws + w.labels), *pick = (long long *)(ws + w.pick);
  int *npick = (int *)(ws + w.npick);
  const dim3 gk((p.K + 255) / 256, p.B), gm((p.max_num + 255) / 256, p.B);
  if (p.R == 6) {
    float *raw = (float *)(ws + w.raw);
    hipLaunchKernelGGL(indoor_prep_scannet_kernel, gk, dim3(256), 0, st, p, boxes, scores, raw, labels);
    for (int b = 0; b < p.B; ++b) {       // indoor batches are small (the reference tests them at batch 1)
      rc = ivx_aligned_3d_nms_ws(boxes + (size_t)b * p.K * 6, scores + (size_t)b * p.K, (const int64_t *)(labels + (size_t)b * p.K), p.K, d->nms_thr,
                                 ws + w.nms, w.nms_bytes, (int64_t *)(pick + (size_t)b * p.K), npick + b, stream);
      if (rc != IVX_OK) return rc;
    }
    hipLaunchKernelGGL(indoor_finish_scannet_kernel, gm, dim3(256), 0, st, p, boxes, raw, labels, pick, npick, out_boxes, out_scores,
                       (long long *)out_labels, out_count);
  } else {
    float *bev = (float *)(ws + w.bev);
    hipLaunchKernelGGL(indoor_prep_sunrgbd_kernel, gk, dim3(256), 0, st, p, boxes, bev, scores);
    for (int b = 0; b < p.B; ++b) {
      rc = ivx_multiclass_nms_bev(bev + (size_t)b * p.K * 5, scores + (size_t)b * p.K * p.ncls, p.K, p.ncls, p.ncls, d->score_thr, d->nms_thr,
                                  d->use_rotate_nms, p.max_num, ws + w.nms, w.nms_bytes, (int64_t *)(pick + (size_t)b * p.max_num),
                                  (int64_t *)(labels + (size_t)b * p.max_num), npick + b, stream);
      if (rc != IVX_OK) return rc;
    }
    hipLaunchKernelGGL(indoor_finish_sunrgbd_kernel, gm, dim3(256), 0, st, p, boxes, scores, pick, labels, npick, out_boxes, out_scores,
                       (long long *)out_labels, out_count);
  }
  IVX_CHECK_LAUNCH("ivx_indoor_tail_get_bboxes");
  return IVX_OK;
}
