// mn_kernels_prepare.h -- producer hand-off and mask post-processing on the device
// ("next" rows 1-2 of SURVEY.md section 8f).
//
// Reference work replaced (the Cityscapes caller, egs/cityscape/local/segment.py):
//   F.sigmoid + .npy save        utils/inference_utils.py:44,96,122-126
//   np.load + cv2.resize(..., seg_size) of [H,W,K] maps (bilinear, INTER_LINEAR)   segment.py:110-123
//   the binding's clip            utils/csegment/c_segment.pyx:53-55
//     -> mn_prepare_maps: logits or probabilities [K][Hin][Win] -> clipped probabilities
//        [K][Hout][Wout], one pass, no host round trip
//   cv2.resize(mask, original size, INTER_NEAREST)                                  segment.py:146-149
//     -> mn_upsample_mask
// Interpolation follows cv2's INTER_LINEAR for float images: source coordinate
// (dst + 0.5) * scale - 0.5, clamped at both borders, horizontal then vertical blend in float32.
// torch.nn.functional.interpolate(mode="bilinear", align_corners=False) uses the same mapping and
// serves as the float reference in the tests (cv2 itself is not installed in this image).
#pragma once

#include "mn_device.h"

__device__ __forceinline__ void mn_lin_coord(int d, float scale, int ssize, int* s0, int* s1, float* f) {
  float fx = ((float)d + 0.5f) * scale - 0.5f;
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.0f; sx = 0; }
  if (sx >= ssize - 1) { fx = 0.0f; sx = ssize - 1; }
  *s0 = sx;
  *s1 = min(sx + 1, ssize - 1);
  *f = fx;
}

__global__ __launch_bounds__(256) void mn_prepare_maps(const float* __restrict__ in, int K, int Hin,
                                                       int Win, float* __restrict__ out, int Hout,
                                                       int Wout, int apply_sigmoid, int clip) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int k = blockIdx.z;
  if (x >= Wout) return;
  const float sx_scale = (float)Win / (float)Wout, sy_scale = (float)Hin / (float)Hout;
  int x0, x1, y0, y1;
  float fx, fy;
  mn_lin_coord(x, sx_scale, Win, &x0, &x1, &fx);
  mn_lin_coord(y, sy_scale, Hin, &y0, &y1, &fy);
  const float* p = in + (size_t)k * Hin * Win;
  float a = p[(size_t)y0 * Win + x0], b = p[(size_t)y0 * Win + x1];
  float c = p[(size_t)y1 * Win + x0], d = p[(size_t)y1 * Win + x1];
  if (apply_sigmoid) {
    a = 1.0f / (1.0f + expf(-a)); b = 1.0f / (1.0f + expf(-b));
    c = 1.0f / (1.0f + expf(-c)); d = 1.0f / (1.0f + expf(-d));
  }
  const float t0 = a * (1.0f - fx) + b * fx;
  const float t1 = c * (1.0f - fx) + d * fx;
  float v = t0 * (1.0f - fy) + t1 * fy;
  if (clip) v = mn_clip(v);
  out[((size_t)k * Hout + y) * Wout + x] = v;
}

// cv2 INTER_NEAREST: source index = min(floor(dst * src/dst_size), src - 1)
__global__ __launch_bounds__(256) void mn_upsample_mask(const int* __restrict__ in, int Hin, int Win,
                                                        int* __restrict__ out, int Hout, int Wout) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= Wout) return;
  const int sx = min((int)floorf((float)x * ((float)Win / (float)Wout)), Win - 1);
  const int sy = min((int)floorf((float)y * ((float)Hin / (float)Hout)), Hin - 1);
  out[(size_t)y * Wout + x] = in[(size_t)sy * Win + sx];
}

// ---- run boundaries of the instance mask in column-major order (COCO RLE) -----------------------
// The reference encodes every instance with pycocotools' RLE over the Fortran-ordered binary mask
// (egs/cityscape/local/segment.py:165-186: maskUtils.encode(np.asfortranarray(mask == i)), one
// full-image pass per instance).  All instances' runs are delimited by the positions where the
// label changes along the column-major scan, so ONE pass finds them: element j of the scan is
// pixel (row j % H, column j / H); a change point is (j, label before, label at j).  They are
// written in scan order (block counts -> mn_rank_scan -> scatter), the host groups them by
// label and forms the counts / the compressed string.
#define MN_RLE_ITEMS 1024

__device__ __forceinline__ int mn_cm_label(const int* __restrict__ mask, int H, int W, int j) {
  const int x = j / H, y = j - x * H;
  return mask[(size_t)y * W + x];
}

__global__ __launch_bounds__(256) void mn_rle_count(const int* __restrict__ mask, int H, int W,
                                                    int* __restrict__ block_count) {
  __shared__ int sh[4];
  const int N = H * W;
  int c = 0;
  for (int k = threadIdx.x; k < MN_RLE_ITEMS; k += 256) {
    const int j = blockIdx.x * MN_RLE_ITEMS + k;
    if (j < N) {
      const int cur = mn_cm_label(mask, H, W, j);
      const int prev = j > 0 ? mn_cm_label(mask, H, W, j - 1) : 0;
      c += (cur != prev) ? 1 : 0;
    }
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void mn_rle_scatter(const int* __restrict__ mask, int H, int W,
                                                      const int* __restrict__ block_offset,
                                                      int* __restrict__ out_pos,
                                                      int* __restrict__ out_prev,
                                                      int* __restrict__ out_cur) {
  __shared__ int sh_w[4];
  const int N = H * W;
  int running = block_offset[blockIdx.x];
  for (int k0 = 0; k0 < MN_RLE_ITEMS; k0 += 256) {
    const int j = blockIdx.x * MN_RLE_ITEMS + k0 + threadIdx.x;
    int cur = 0, prev = 0;
    if (j < N) {
      cur = mn_cm_label(mask, H, W, j);
      prev = j > 0 ? mn_cm_label(mask, H, W, j - 1) : 0;
    }
    const bool f = j < N && cur != prev;
    const u64 m = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh_w[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += sh_w[w];
    const int tot = sh_w[0] + sh_w[1] + sh_w[2] + sh_w[3];
    if (f) {
      const int idx = running + woff + __popcll(m & ((1ull << lane) - 1ull));
      out_pos[idx] = j;
      out_prev[idx] = prev;
      out_cur[idx] = cur;
    }
    running += tot;
    __syncthreads();
  }
}

// ---- sameness targets from an instance mask (training-side twin of the synthetic generator) ----
// target[k][r][c] = (mask[r + di_k][c + dj_k] == mask[r][c]), 1 where the neighbour is outside the
// image (utils/dataset.py:259-277: np.roll compare, border rows/columns forced to 1).
__global__ __launch_bounds__(256) void mn_sameness_targets(const int* __restrict__ mask, int H, int W,
                                                           ImgParams P, float* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (p >= H * W) return;
  const int r = p / W, c = p - r * W;
  const int rr = r + P.di[k], cc = c + P.dj[k];
  float v = 1.0f;
  if (rr >= 0 && rr < H && cc >= 0 && cc < W) v = (mask[(size_t)rr * W + cc] == mask[p]) ? 1.0f : 0.0f;
  out[(size_t)k * H * W + p] = v;
}

// ---- per-instance confidence: log-prob margin of the instance class over background ------------
// score[label - 1] = lp[cls] - lp[0] of the object (Object::GetNoBackLogprob, segment.h:109; the
// reference's COCO results carry a constant score 1, egs/cityscape/local/segment.py:181).
__global__ __launch_bounds__(256) void mn_instance_scores(ImgParams P, ObjState S,
                                                          const int* __restrict__ label,
                                                          float* __restrict__ scores) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N || S.parent[p] != p) return;
  const int l = label[p];
  if (l <= 0) return;
  const bool valid = S.lpvalid[p] != 0;
  scores[l - 1] = mn_obj_lp(P, S, valid, p, S.ocls[p]) - mn_obj_lp(P, S, valid, p, 0);
}

// ---- run-length wire format of the multi-GPU mask exchange ---------------------------------------
// The final masks are piecewise constant: instead of 2 bytes per pixel the exchange ships the
// row-major label CHANGE POINTS (a few thousand per image).  Wire, int32 words:
//   [0] number of change points, -1 if they do not fit `cap`   [1] K (instances)
//   [2..3] float64 total log-likelihood                         [4 .. 4+cap) positions, ascending
//   then cap int16 labels (label from that position on), then max_instances int8 classes.
// A mask is label 0 before the first change point.  24x smaller than int32 masks, 10.6x smaller
// than the int16 map at cap = n_pixels / 32 (397 KB instead of 4.2 MB per 1024x2048 image).
__device__ __forceinline__ size_t mn_runs_words(int cap, int max_instances) {
  return 4 + (size_t)cap + (size_t)(cap + 1) / 2 + (size_t)(max_instances + 3) / 4;
}

__global__ __launch_bounds__(256) void mn_runs_count(const int* __restrict__ mask, int N,
                                                     int* __restrict__ block_count) {
  __shared__ int sh[4];
  int c = 0;
  for (int k = threadIdx.x; k < MN_RLE_ITEMS; k += 256) {
    const int j = blockIdx.x * MN_RLE_ITEMS + k;
    if (j < N) c += (mask[j] != (j > 0 ? mask[j - 1] : 0)) ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void mn_runs_scatter(const int* __restrict__ mask, int N,
                                                       const int* __restrict__ block_offset,
                                                       const int* __restrict__ total, int cap,
                                                       int max_instances, int num_instances,
                                                       double total_logprob,
                                                       const int* __restrict__ table,
                                                       int* __restrict__ wire) {
  __shared__ int sh_w[4];
  short* labels = reinterpret_cast<short*>(wire + 4 + cap);
  signed char* classes = reinterpret_cast<signed char*>(wire + 4 + cap + (cap + 1) / 2);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      wire[0] = *total <= cap ? *total : -1;
      wire[1] = num_instances;
      const unsigned long long bits = (unsigned long long)__double_as_longlong(total_logprob);
      wire[2] = (int)(bits & 0xFFFFFFFFull);
      wire[3] = (int)(bits >> 32);
    }
    for (int i = threadIdx.x; i < max_instances; i += 256)
      classes[i] = (signed char)(i < num_instances ? table[i] : -1);
  }
  int running = block_offset[blockIdx.x];
  for (int k0 = 0; k0 < MN_RLE_ITEMS; k0 += 256) {
    const int j = blockIdx.x * MN_RLE_ITEMS + k0 + threadIdx.x;
    int cur = 0, prev = 0;
    if (j < N) { cur = mask[j]; prev = j > 0 ? mask[j - 1] : 0; }
    const bool f = j < N && cur != prev;
    const u64 m = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh_w[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += sh_w[w];
    const int tot = sh_w[0] + sh_w[1] + sh_w[2] + sh_w[3];
    if (f) {
      const int idx = running + woff + __popcll(m & ((1ull << lane) - 1ull));
      if (idx < cap) { wire[4 + idx] = j; labels[idx] = (short)cur; }
    }
    running += tot;
    __syncthreads();
  }
}

// wire -> dense int32 mask and int32 class table (-1 padded): the label of pixel p is the label of
// the last change point at or before p (binary search), 0 before the first
__global__ __launch_bounds__(256) void mn_runs_unpack(const int* __restrict__ wire, int N, int cap,
                                                      int max_instances, int* __restrict__ mask,
                                                      int* __restrict__ table, long long wire_stride) {
  // blockIdx.y = which wire of a batch (the gathered wires of all ranks in ONE launch); a wire whose
  // header is damaged (count beyond its capacity: a peer built with another capacity) must not send the
  // search past its own sections
  wire += (size_t)blockIdx.y * (size_t)wire_stride;
  mask += (size_t)blockIdx.y * (size_t)N;
  if (table) table += (size_t)blockIdx.y * (size_t)max_instances;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(wire[0], cap);
  const short* labels = reinterpret_cast<const short*>(wire + 4 + cap);
  const signed char* classes = reinterpret_cast<const signed char*>(wire + 4 + cap + (cap + 1) / 2);
  if (table && p < max_instances) table[p] = classes[p];
  if (p >= N) return;
  int lo = 0, hi = n < 0 ? 0 : n;             // first change point with position > p
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (wire[4 + mid] <= p) lo = mid + 1; else hi = mid;
  }
  mask[p] = lo > 0 ? (int)labels[lo - 1] : 0;
}
