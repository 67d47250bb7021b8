// mn_kernels_output.h -- labels, mask, class table, prune, certificate and log-likelihood.
//
// Reference work replaced:
//   OutputMask                     utils/csegment/segment.cc:491-517   -> roots/rank/mask kernels
//   prune + output_mask (Python)   utils/segmenter.py:351-389          -> mn_prune_* kernels
//   ComputeTotalLogprobFromScratch utils/csegment/segment.cc:314-350   -> mn_verify_edges
// Labels are handed out in ascending surviving object id (the reference uses hash-map iteration
// order; results are compared up to a permutation of 1..K).
#pragma once

#include "mn_device.h"
#include "mn_kernels_merge.h"

#define MN_SCAN_ITEMS 1024   /* pixels per block in the label ranking */

// ---- Python-variant prune (segmenter.py:351-375) ----------------------------------------------
// background = class-0 object with the most pixels (first such in ascending id);
// every other object with lp[cls] - lp[0] < threshold is dumped into it (label 0).
__global__ __launch_bounds__(256) void mn_prune_find_background(ImgParams P, ObjState S,
                                                                u64* __restrict__ bg_key) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N || S.parent[p] != p || S.ocls[p] != 0) return;
  atomicMax(bg_key, ((u64)(unsigned)S.osize[p] << 32) | (u64)(0xFFFFFFFFu - (unsigned)p));
}

__global__ __launch_bounds__(256) void mn_prune_mark(ImgParams P, ObjState S, float threshold,
                                                     const u64* __restrict__ bg_key,
                                                     unsigned char* __restrict__ pruned,
                                                     Counters* __restrict__ cnt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  pruned[p] = 0;
  if (S.parent[p] != p) return;
  const bool valid = S.lpvalid[p] != 0;
  const float score = mn_obj_lp(P, S, valid, p, S.ocls[p]) - mn_obj_lp(P, S, valid, p, 0);
  if (!(score < threshold)) return;
  const u64 k = *bg_key;
  if (k == 0) { cnt->error = MN_ERR_NO_BACKGROUND; return; }   // reference: NameError
  const int bg = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
  if (p != bg) pruned[p] = 1;
}

// ---- label ranking: instances are the live objects whose class is not 0 -----------------------
__device__ __forceinline__ int mn_is_instance(const ObjState& S, const unsigned char* pruned, int p) {
  return (S.parent[p] == p && S.ocls[p] != 0 && !(pruned && pruned[p])) ? 1 : 0;
}

__global__ __launch_bounds__(256) void mn_rank_count(int N, ObjState S,
                                                     const unsigned char* __restrict__ pruned,
                                                     int* __restrict__ block_count,
                                                     int* __restrict__ n_objects) {
  __shared__ int sh[4];
  const int base = blockIdx.x * MN_SCAN_ITEMS;
  int c = 0, live = 0;
  for (int k = threadIdx.x; k < MN_SCAN_ITEMS; k += 256) {
    const int p = base + k;
    if (p < N) { c += mn_is_instance(S, pruned, p); live += (S.parent[p] == p) ? 1 : 0; }
  }
  for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off); live += __shfl_xor(live, off); }
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = c; if (live) atomicAdd(n_objects, live); }
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// exclusive scan of the per-block counts by one workgroup; writes the total too
__global__ __launch_bounds__(1024) void mn_rank_scan(int nblocks, int* __restrict__ block_count,
                                                     int* __restrict__ total) {
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? block_count[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_count[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// label[root] = rank + 1 in ascending id; object_class[rank] = class (segment.cc:509)
__global__ __launch_bounds__(256) void mn_rank_assign(int N, ObjState S,
                                                      const unsigned char* __restrict__ pruned,
                                                      const int* __restrict__ block_offset,
                                                      int* __restrict__ label,
                                                      int* __restrict__ object_class) {
  __shared__ int sh_w[4];
  const int base = blockIdx.x * MN_SCAN_ITEMS;
  int running = block_offset[blockIdx.x];
  for (int k0 = 0; k0 < MN_SCAN_ITEMS; k0 += 256) {
    const int p = base + k0 + threadIdx.x;
    const int f = (p < N) ? mn_is_instance(S, pruned, p) : 0;
    const u64 m = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sh_w[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += sh_w[w];
    const int tot = sh_w[0] + sh_w[1] + sh_w[2] + sh_w[3];
    if (p < N) {
      if (f) {
        const int rank = running + woff + before;
        label[p] = rank + 1;
        object_class[rank] = S.ocls[p];
      } else {
        label[p] = 0;
      }
    }
    running += tot;
    __syncthreads();
  }
}

// Last pass of the output: union forest -> root (kept for the certificate), mask[p] = label of the
// root, optional partition, and the -1 padding of the class table from index K on
// (segment.cc:497-509) -- K is read from the device, entries below K were written by mn_rank_assign.
__global__ __launch_bounds__(256) void mn_write_mask(int N, const int* __restrict__ parent,
                                                     const int* __restrict__ label,
                                                     const int* __restrict__ num_instances,
                                                     int* __restrict__ root, int* __restrict__ mask,
                                                     int* __restrict__ partition,
                                                     int* __restrict__ object_class) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  int r = p;
  while (parent[r] != r) r = parent[r];
  if (root) root[p] = r;
  mask[p] = label[r];
  if (partition) partition[p] = r;
  if (p >= *num_instances) object_class[p] = -1;
}

// The same, four pixels per lane (N % 4 == 0, buffers 16-byte aligned): 16-byte loads and stores for what is
// streamed, four gathers for the labels.  `root` may be null (only the per-pixel certificate reads it).
__global__ __launch_bounds__(256) void mn_write_mask4(int N, const int* __restrict__ parent,
                                                      const int* __restrict__ label,
                                                      const int* __restrict__ num_instances,
                                                      int* __restrict__ root, int* __restrict__ mask,
                                                      int* __restrict__ partition,
                                                      int* __restrict__ object_class) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * g >= N) return;
  int4 r = *reinterpret_cast<const int4*>(parent + 4 * (size_t)g);
  const int p0 = 4 * g;
  // (parent is flat on the fast path; anywhere else a few steps)
  while (parent[r.x] != r.x) r.x = parent[r.x];
  while (parent[r.y] != r.y) r.y = parent[r.y];
  while (parent[r.z] != r.z) r.z = parent[r.z];
  while (parent[r.w] != r.w) r.w = parent[r.w];
  if (root) *reinterpret_cast<int4*>(root + p0) = r;
  *reinterpret_cast<int4*>(mask + p0) = make_int4(label[r.x], label[r.y], label[r.z], label[r.w]);
  if (partition) *reinterpret_cast<int4*>(partition + p0) = r;
  const int k = *num_instances;
  if (p0 >= k) *reinterpret_cast<int4*>(object_class + p0) = make_int4(-1, -1, -1, -1);
  else if (p0 + 3 >= k) {
    if (p0 >= k) object_class[p0] = -1;
    if (p0 + 1 >= k) object_class[p0 + 1] = -1;
    if (p0 + 2 >= k) object_class[p0 + 2] = -1;
    if (p0 + 3 >= k) object_class[p0 + 3] = -1;
  }
}

// ---- certificate + log-likelihood over the pixel graph ----------------------------------------
// One lane per pixel.  Sums (float64, reduced per block, combined in block order by
// mn_verify_reduce so the result does not depend on scheduling):
//   class term  log class[cls(obj(p))][p], sameness term log p over edges inside one object,
//   differentness term log(1-p) over edges between objects          (segment.cc:314-350)
// Violations of the sign-separability certificate: an edge inside an object whose
// log-odds are not > 0, an edge between objects whose log-odds are not < 0, or a pixel whose own
// arg-max class differs from its object's class.
__global__ __launch_bounds__(256) void mn_verify_edges(ImgParams P, ObjState S,
                                                       const unsigned char* __restrict__ cls0,
                                                       const int* __restrict__ root,
                                                       double* __restrict__ partial,
                                                       int* __restrict__ violations) {
  __shared__ double sh[3][4];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double t_cls = 0.0, t_same = 0.0, t_diff = 0.0;
  int bad = 0, bad_cls = 0;
  if (p < P.N) {
    const int r = p / P.W, c = p - r * P.W;
    const int ro = root[p];
    const int oc = S.ocls[ro];
    t_cls = (double)logf(mn_ld_class(P, oc, p));
    if (cls0[p] != oc) bad_cls++;
    // for omf >= 1e-20 the sign of the float gain (log v - log(1-v)) * omf is the sign of v - 0.5
    // (the two logs of a float next to 0.5 are still 4 ulp apart and the product cannot underflow),
    // so only the one log that enters the sums is evaluated
    const bool by_value = P.omf >= 1e-20f;
    constexpr int G = 10;                     // offsets whose loads are in flight together
    for (int k0 = 0; k0 < P.O; k0 += G) {
      float vv[G];
      int rq[G];
      bool in[G];
#pragma unroll
      for (int j = 0; j < G; j++) {
        const int k = k0 + j;
        in[j] = false;
        vv[j] = 0.5f;
        rq[j] = ro;
        if (k < P.O) {
          const int rr = r + P.di[k], cc = c + P.dj[k];
          if (rr >= 0 && rr < P.H && cc >= 0 && cc < P.W) {
            in[j] = true;
            vv[j] = P.same[(size_t)k * P.N + p];
            rq[j] = root[rr * P.W + cc];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < G; j++) {
        if (!in[j]) continue;
        const float v = mn_same_value(P, vv[j]);
        const bool inside = rq[j] == ro;
        const float lg = inside ? logf(v) : mn_log1m(v);
        bool pos, neg;
        if (by_value) { pos = v > 0.5f; neg = v < 0.5f; }
        else {
          const float g = (logf(v) - mn_log1m(v)) * P.omf;
          pos = g > 0.0f; neg = g < 0.0f;
        }
        if (inside) { t_same += (double)lg; if (!pos) bad++; }
        else        { t_diff += (double)lg; if (!neg) bad++; }
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    t_cls += __shfl_xor(t_cls, off);
    t_same += __shfl_xor(t_same, off);
    t_diff += __shfl_xor(t_diff, off);
    bad += __shfl_xor(bad, off);
    bad_cls += __shfl_xor(bad_cls, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[0][wave] = t_cls; sh[1][wave] = t_same; sh[2][wave] = t_diff;
    if (bad) atomicAdd(violations, bad);
    if (bad_cls) atomicAdd(violations + 3, bad_cls);
  }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[(size_t)blockIdx.x * 3 + threadIdx.x] =
        ((sh[threadIdx.x][0] + sh[threadIdx.x][1]) + sh[threadIdx.x][2]) + sh[threadIdx.x][3];
}

// The same sums with 4 consecutive pixels of one row per lane (W % 4 == 0): 16-byte loads of the
// sameness values and of the neighbours' roots (unaligned), a quarter of the load instructions.
struct __attribute__((packed, aligned(4))) mn_int4_unaligned { int x, y, z, w; };

#ifndef MN_VERIFY4_THREADS
#define MN_VERIFY4_THREADS 256
#endif
__global__ __launch_bounds__(MN_VERIFY4_THREADS) void mn_verify_edges4(ImgParams P, ObjState S,
                                                        const unsigned char* __restrict__ cls0,
                                                        const int* __restrict__ root,
                                                        double* __restrict__ partial,
                                                        int* __restrict__ violations) {
  __shared__ double sh[3][MN_VERIFY4_THREADS / 64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double t_cls = 0.0, t_same = 0.0, t_diff = 0.0;
  int bad = 0, bad_cls = 0;
  if (i < (P.N >> 2)) {
    const int p0 = 4 * i;
    const int r = p0 / P.W, c = p0 - r * P.W;
    const int4 ro = *reinterpret_cast<const int4*>(root + p0);
    const uchar4 own = *reinterpret_cast<const uchar4*>(cls0 + p0);
    const int oc0 = S.ocls[ro.x], oc1 = S.ocls[ro.y], oc2 = S.ocls[ro.z], oc3 = S.ocls[ro.w];
    t_cls = ((double)logf(mn_ld_class(P, oc0, p0)) + (double)logf(mn_ld_class(P, oc1, p0 + 1))) +
            ((double)logf(mn_ld_class(P, oc2, p0 + 2)) + (double)logf(mn_ld_class(P, oc3, p0 + 3)));
    bad_cls = (own.x != oc0) + (own.y != oc1) + (own.z != oc2) + (own.w != oc3);
    const bool by_value = P.omf >= 1e-20f;       // see mn_verify_edges
    auto edge = [&](int col, float raw, int q, int mine) {
      if (col < 0 || col >= P.W) return;
      const float v = mn_same_value(P, raw);
      const bool inside = q == mine;
      const float lg = inside ? logf(v) : mn_log1m(v);
      bool pos, neg;
      if (by_value) { pos = v > 0.5f; neg = v < 0.5f; }
      else {
        const float g = (logf(v) - mn_log1m(v)) * P.omf;
        pos = g > 0.0f; neg = g < 0.0f;
      }
      if (inside) { t_same += (double)lg; if (!pos) bad++; }
      else        { t_diff += (double)lg; if (!neg) bad++; }
    };
    constexpr int G = 5;                         // offsets whose loads are in flight together
    for (int k0 = 0; k0 < P.O; k0 += G) {
      float4 vv[G];
      int4 rq[G];
      int first[G];
#pragma unroll
      for (int j = 0; j < G; j++) {
        const int k = k0 + j;
        first[j] = INT_MIN;
        if (k < P.O) {
          const int rr = r + P.di[k];
          if (rr >= 0 && rr < P.H) {
            first[j] = c + P.dj[k];
            vv[j] = *reinterpret_cast<const float4*>(P.same + (size_t)k * P.N + p0);
            const long long q0 = (long long)rr * P.W + first[j];
            if (q0 >= 0 && q0 + 3 < P.N) {
              const mn_int4_unaligned t = *reinterpret_cast<const mn_int4_unaligned*>(root + q0);
              rq[j] = make_int4(t.x, t.y, t.z, t.w);
            } else {
              rq[j].x = (q0 >= 0 && q0 < P.N) ? root[q0] : 0;
              rq[j].y = (q0 + 1 >= 0 && q0 + 1 < P.N) ? root[q0 + 1] : 0;
              rq[j].z = (q0 + 2 >= 0 && q0 + 2 < P.N) ? root[q0 + 2] : 0;
              rq[j].w = (q0 + 3 >= 0 && q0 + 3 < P.N) ? root[q0 + 3] : 0;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < G; j++) {
        if (first[j] == INT_MIN) continue;
        edge(first[j], vv[j].x, rq[j].x, ro.x);
        edge(first[j] + 1, vv[j].y, rq[j].y, ro.y);
        edge(first[j] + 2, vv[j].z, rq[j].z, ro.z);
        edge(first[j] + 3, vv[j].w, rq[j].w, ro.w);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    t_cls += __shfl_xor(t_cls, off);
    t_same += __shfl_xor(t_same, off);
    t_diff += __shfl_xor(t_diff, off);
    bad += __shfl_xor(bad, off);
    bad_cls += __shfl_xor(bad_cls, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[0][wave] = t_cls; sh[1][wave] = t_same; sh[2][wave] = t_diff;
    if (bad) atomicAdd(violations, bad);
    if (bad_cls) atomicAdd(violations + 3, bad_cls);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < MN_VERIFY4_THREADS / 64; w++) t += sh[threadIdx.x][w];
    partial[(size_t)blockIdx.x * 3 + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void mn_verify_reduce(int nblocks, const double* __restrict__ partial,
                                                        float omf, double* __restrict__ out) {
  __shared__ double sh[3][256];
  double a[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
    for (int j = 0; j < 3; j++) a[j] += partial[(size_t)b * 3 + j];
  for (int j = 0; j < 3; j++) sh[j][threadIdx.x] = a[j];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off)
      for (int j = 0; j < 3; j++) sh[j][threadIdx.x] += sh[j][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sh[0][0] + (sh[2][0] + sh[1][0]) * (double)omf;
    out[1] = sh[0][0]; out[2] = sh[1][0]; out[3] = sh[2][0];
  }
}

// Quotient condition of the certificate: no record between two final objects may still be
// mergeable (priority must be negative with a margin that covers float32 accumulation).
__global__ __launch_bounds__(256) void mn_verify_records(ImgParams P, ObjState S, RecList L, int R,
                                                         int* __restrict__ violations,
                                                         const int* __restrict__ spec_records) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (spec_records) R = min(R, *spec_records);     // components mode: count known on the device only
  if (i >= R) return;
  const u64 key = L.key[i];
  if (key == MN_EMPTY) return;
  int mc;
  bool pos;
  const float oml = mn_fixed_to_float(L.S[i]);
  const float f = mn_score(P, S, mn_key_u(key), mn_key_v(key), oml, &mc, &pos);
  const float margin = 1e-6f + 1e-5f * fabsf(P.bias);
  if (!(f < -margin)) atomicAdd(violations + 4, 1);
}

// ---- wire format of the multi-GPU exchange ------------------------------------------------------
// One int16 buffer per image: [n_pixels labels][count][max_instances classes, -1 padded][4 words =
// the float64 total log-likelihood].  Labels are 0..K with K <= max_instances (4096), classes
// < 128: half the bytes of the int32 mask on the xGMI links, and mask, class table and
// log-likelihood travel in ONE all-gather.
__global__ __launch_bounds__(256) void mn_pack_wire(int n_pixels, int max_instances, int num_instances,
                                                    double total_logprob,
                                                    const int* __restrict__ mask,
                                                    const int* __restrict__ table,
                                                    short* __restrict__ wire) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = n_pixels >> 2;
  if (i < n4) {
    const int4 m = *reinterpret_cast<const int4*>(mask + 4 * (size_t)i);
    short4 o;
    o.x = (short)m.x; o.y = (short)m.y; o.z = (short)m.z; o.w = (short)m.w;
    *reinterpret_cast<short4*>(wire + 4 * (size_t)i) = o;
  }
  if (i < n_pixels - (n4 << 2)) wire[(n4 << 2) + i] = (short)mask[(n4 << 2) + i];
  if (i <= max_instances) {
    short v;
    if (i == 0) v = (short)num_instances;
    else v = (i <= num_instances) ? (short)table[i - 1] : (short)-1;
    wire[(size_t)n_pixels + i] = v;
  }
  if (i < 4) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(total_logprob);
    wire[(size_t)n_pixels + 1 + max_instances + i] = (short)((bits >> (16 * i)) & 0xFFFFull);
  }
}

