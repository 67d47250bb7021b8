// mn_kernels_cc.h -- component contraction: phase 1 of the merge in one union-find sweep, for
// inputs where phase 1 is provably order-independent.
//
// Claim (csegment variant, object_merge_factor > 0, merge_logprob_bias >= 0).  Let the pixel
// graph be SIGN-SEPARABLE: the connected components of the edges with positive log-odds are such
// that (a) every edge inside a component has gain >= tau, (b) every edge between components has
// gain <= -tau, (c) all pixels of a component share one arg-max class; tau > 0 is the float32
// rounding margin of fill_params (2 N ulp(bias): the quotient below must not vanish in the sum).  Then at every moment of
// the reference's run (utils/csegment/segment.cc:539-727) every record between two sub-objects of
// one component scores  sum(log-odds)*omf / (n1+n2) + bias  >  bias  (class delta 0, segment.cc:
// 107-150) and every record between sub-objects of different components scores < bias (negative
// log-odds, class delta <= 0).  The queue pops in descending priority, so no cross record is
// popped while any intra record is alive: the reference first merges every component completely
// -- in whatever order -- and only then turns to the records between components.  The state at
// that moment (objects = components, one fully summed record per adjacent pair) is what this
// file builds directly:
//   mn_cc_tiles    16 x 64-pixel tiles labelled in LDS over the two unit offsets;
//   mn_cc_hook     lock-free union-find over the implicit pixel graph (positive edges), root =
//                  lowest pixel id of the component; unit offsets first, then the rest;
//   mn_cc_flatten  parent[p] = root (and the roots' accumulators cleared);
//   mn_cc_class_sums  the class pass: arg-max class of every pixel, component sizes and class
//                  log-prob sums (block table in LDS, 64-bit fixed-point atomics);
//   mn_cc_edges    conditions (a), (b), (c); records between components summed into
//                  the hash table (per lane while the key repeats, then per block in LDS);
//   mn_cc_finish   fixed-point sums -> float object state.
// The second phase (records between components, where the bias lets a 1.6 M-pixel background
// swallow small instances) is then run by the sequential finisher in the reference's order,
// starting from freshly scored records.  If (a)-(c) fail the caller falls back to the general
// rounds: a spurious positive link across a boundary joins two components, which then contain
// negative edges, so the check -- not luck -- keeps this shortcut safe.
#pragma once

#include "mn_device.h"
#include "mn_kernels_merge.h"

#define MN_LP_FIX 4294967296.0   /* 2^32: fixed-point scale of class log-prob sums */

__device__ __forceinline__ int mn_cc_find(int* __restrict__ parent, int x) {
  int p = parent[x];
  while (p != x) {
    const int g = parent[p];
    if (g != p) parent[x] = g;      // path halving (benign race: only ever points further up)
    x = p;
    p = g;
  }
  return x;
}

// Tile stage: a block owns a tile of 16 rows x 64 columns and labels it in LDS.
//  1. rows, no atomics: a wave covers the 64 pixels of one tile row; lanes joined by positive edges
//     of the horizontal unit offset (index kh, direction +1 column) form runs, and every pixel
//     points at the first pixel of its run (one ballot and bit arithmetic);
//  2. columns: positive edges of the vertical unit offset (index kv, direction dv = +-1 row) that
//     stay inside the tile are united by a union-find on the LDS labels (one lane per distinct
//     pair of roots in a wave);
//  3. the flattened labels go to `parent` as pixel ids.  Local order (row, column) is the global
//     pixel order, so "larger root under smaller" keeps holding across stages.
// The global sweep that follows then only does real work on tile borders: 1/16 of the vertical and
// 1/64 of the horizontal unit edges, against trees that are already flat.
#define MN_CC_TILE_ROWS 16
__global__ __launch_bounds__(1024) void mn_cc_tiles(ImgParams P, int* __restrict__ parent, int kh,
                                                    int kv, int dv) {
  __shared__ int lab[MN_CC_TILE_ROWS * 64];
  const int t = threadIdx.x, lane = t & 63, i = t >> 6;
  const int r = (int)blockIdx.y * MN_CC_TILE_ROWS + i, c = (int)blockIdx.x * 64 + lane;
  const bool in = r < P.H && c < P.W;
  const int p = in ? r * P.W + c : 0;
  bool link = false;                         // positive edge between this pixel and the next one
  if (in && kh >= 0 && c + 1 < P.W) link = mn_same_value(P, P.same[(size_t)kh * P.N + p]) > 0.5f;
  bool vlink = false;
  const int ni = i + dv;
  if (in && kv >= 0 && ni >= 0 && ni < MN_CC_TILE_ROWS && r + dv >= 0 && r + dv < P.H)
    vlink = mn_same_value(P, P.same[(size_t)kv * P.N + p]) > 0.5f;
  const u64 m = __ballot(link);
  // run start = one past the highest lane below `lane` that has NO link to its successor
  const u64 below = lane ? (~m & ((1ull << lane) - 1ull)) : 0ull;
  const int start = below ? (64 - __clzll((long long)below)) : 0;
  lab[t] = i * 64 + start;
  __syncthreads();
  int a = 0, b = 0;
  bool want = false;
  if (vlink) {
    a = mn_cc_find(lab, t);
    b = mn_cc_find(lab, ni * 64 + lane);
    want = a != b;
  }
  u64 todo = __ballot(want);
  const u64 key = mn_key(a, b);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const u64 k0v = ((u64)(unsigned)__shfl((int)(key >> 32), first) << 32) |
                    (u64)(unsigned)__shfl((int)(key & 0xFFFFFFFFull), first);
    const bool mine = want && key == k0v;
    if (lane == first) {
      while (a != b) {                                  // hook the larger root under the smaller
        if (a < b) { const int x = a; a = b; b = x; }
        const int old = atomicMin(&lab[a], b);
        if (old == a) break;
        a = mn_cc_find(lab, old);
        b = mn_cc_find(lab, b);
      }
    }
    todo &= ~__ballot(mine);
  }
  __syncthreads();
  if (!in) return;
  int x = t;
  while (lab[x] != x) x = lab[x];
  parent[p] = ((int)blockIdx.y * MN_CC_TILE_ROWS + (x >> 6)) * P.W + (int)blockIdx.x * 64 + (x & 63);
}

// Border stage: after mn_cc_tiles the only unit-offset edges still open are those that cross a
// tile border, and the 16 (or 64) edges of one border segment almost always ask for the same
// union.  One block per tile: wave 0 takes the 16 edges across the tile's right border, wave 1 the
// 64 edges across its lower (dv = +1) or upper (dv = -1) border; one lane per distinct pair of
// roots does the union.  ~4 K unions for a 1024x2048 image instead of a sweep over every pixel.
__global__ __launch_bounds__(128) void mn_cc_borders(ImgParams P, int* __restrict__ parent, int kh,
                                                     int kv, int dv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (int)blockIdx.y * MN_CC_TILE_ROWS, c0 = (int)blockIdx.x * 64;
  int p = -1, q = -1;
  if (wave == 0) {                                   // right border: (r0 + lane, c0 + 63) -> next column
    const int r = r0 + lane, c = c0 + 63;
    if (lane < MN_CC_TILE_ROWS && r < P.H && c + 1 < P.W) {
      p = r * P.W + c;
      if (mn_same_value(P, P.same[(size_t)kh * P.N + p]) > 0.5f) q = p + 1;
    }
  } else {                                           // the border the vertical offset crosses
    const int r = dv > 0 ? r0 + MN_CC_TILE_ROWS - 1 : r0, c = c0 + lane;
    if (r < P.H && r + dv >= 0 && r + dv < P.H && c < P.W) {
      p = r * P.W + c;
      if (mn_same_value(P, P.same[(size_t)kv * P.N + p]) > 0.5f) q = p + dv * P.W;
    }
  }
  int a = 0, b = 0;
  bool want = false;
  if (q >= 0) {
    a = mn_cc_find(parent, p);
    b = mn_cc_find(parent, q);
    want = a != b;
  }
  u64 todo = __ballot(want);
  const u64 key = mn_key(a, b);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const u64 k0v = ((u64)(unsigned)__shfl((int)(key >> 32), first) << 32) |
                    (u64)(unsigned)__shfl((int)(key & 0xFFFFFFFFull), first);
    const bool mine = want && key == k0v;
    if (lane == first) {
      while (a != b) {                                  // hook the larger root under the smaller
        if (a < b) { const int x = a; a = b; b = x; }
        const int old = atomicMin(&parent[a], b);
        if (old == a) break;
        a = mn_cc_find(parent, old);
        b = mn_cc_find(parent, b);
      }
    }
    todo &= ~__ballot(mine);
  }
}

// Offsets [k0, k1) only: the sweep runs first over the two unit offsets, which already connect
// almost every component, is flattened, and then runs over the rest, whose edges then find equal
// roots at once (no atomic).
__global__ __launch_bounds__(256) void mn_cc_hook(ImgParams P, int* __restrict__ parent, int k0,
                                                  int k1) {
  const int tile = mn_xcd_tile((P.N + 255) >> 8, P.banded);
  if (tile < 0) return;
  const int p = tile * 256 + threadIdx.x;
  const bool live = p < P.N;
  const int lane = threadIdx.x & 63;
  const int r = live ? p / P.W : 0, c = live ? p - r * P.W : 0;
  constexpr int G = 4;                        // offsets whose loads are in flight together
  for (int kb = k0; kb < k1; kb += G) {
    float vv[G];
    int pq[G], qq[G];
    const int pp = live ? parent[p] : 0;
#pragma unroll
    for (int j = 0; j < G; j++) {
      const int k = kb + j;
      vv[j] = 0.0f;
      pq[j] = pp;
      qq[j] = -1;
      if (live && k < k1) {
        const int rr = r + P.di[k], cc = c + P.dj[k];
        if (rr >= 0 && rr < P.H && cc >= 0 && cc < P.W) {
          qq[j] = rr * P.W + cc;
          vv[j] = P.same[(size_t)k * P.N + p];
          pq[j] = parent[qq[j]];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < G; j++) {
    if (kb + j >= k1) break;                  // uniform
    bool want = false;
    int a = 0, b = 0;
    if (qq[j] >= 0 && pq[j] != pp) {          // equal parents: already one set, nothing to do
      const float v = mn_same_value(P, vv[j]);
      if (v > 0.5f) {                         // log-odds > 0 (see mn_cc_edges)
        a = mn_cc_find(parent, p);
        b = mn_cc_find(parent, qq[j]);
        want = a != b;
      }
    }
    // the 64 pixels of a wave mostly ask for the same few unions (runs of a row against the runs
    // of another row): one lane per distinct (root, root) pair does it
    u64 todo = __ballot(want);
    const u64 key = mn_key(a, b);
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const u64 k0v = ((u64)(unsigned)__shfl((int)(key >> 32), first) << 32) |
                      (u64)(unsigned)__shfl((int)(key & 0xFFFFFFFFull), first);
      const bool mine = want && key == k0v;
      if (lane == first) {
        while (a != b) {                                // hook the larger root under the smaller
          if (a < b) { const int t = a; a = b; b = t; }
          const int old = atomicMin(&parent[a], b);
          if (old == a) break;
          a = mn_cc_find(parent, old);
          b = mn_cc_find(parent, b);
        }
      }
      todo &= ~__ballot(mine);
    }
    }
  }
}

// `lp_acc` set (last flatten): component sizes and the class sums of the roots start from zero --
// only the roots' slots of the C planes are ever used, so they are cleared here instead of by a
// memset of all C * N words.
__global__ __launch_bounds__(256) void mn_cc_flatten(int N, int C, int* __restrict__ parent,
                                                     int* __restrict__ osize,
                                                     i64* __restrict__ lp_acc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  int x = p;
  while (parent[x] != x) x = parent[x];
  parent[p] = x;
  if (lp_acc) {
    osize[p] = 0;
    if (x == p)
      for (int c = 0; c < C; c++) lp_acc[(size_t)c * N + p] = 0;
  }
}

// Class pass of components mode: one sweep over the C class planes gives every pixel its arg-max
// class (same rule as mn_class_pass: first maximum of logf) AND adds its class log-probs and its
// count to the sums of its component.  4 consecutive pixels per lane (16-byte loads); a lane whose
// four pixels share a root -- nearly all do -- issues one LDS atomic per class into the block's
// table (root -> C+1 fixed-point sums); the block then issues ONE global atomic per root and class:
// a 1.6 M-pixel background is a hot word for every wave of the image, and one word takes only ~88
// atomics/us.  64-bit fixed-point sums (2^-32) are order-independent.
#define MN_CC_SUM_THREADS 1024
#define MN_CC_SUM_SLOTS 64
__device__ __forceinline__ int mn_lds_root_slot(int* s_root, int root) {
  unsigned h = ((unsigned)root * 2654435761u) >> 26;             // 6 bits
#pragma unroll 1
  for (int t = 0; t < MN_CC_SUM_SLOTS; t++) {
    int cur = s_root[h];                       // plain read first: the slot is usually there already
    if (cur == root) return (int)h;
    if (cur == -1) cur = atomicCAS(&s_root[h], -1, root);
    if (cur == -1 || cur == root) return (int)h;
    h = (h + 1) & (MN_CC_SUM_SLOTS - 1);
  }
  return -1;
}

__device__ __forceinline__ void mn_cc_add(const ImgParams& P, const ObjState& S, int* s_root,
                                          u64* s_val, i64* __restrict__ lp_acc, int root, int c,
                                          int slot, i64 x) {
  if (slot >= 0) atomicAdd(&s_val[slot * (P.C + 1) + c], (u64)x);
  else if (c == P.C) atomicAdd(&S.osize[root], (int)x);
  else atomicAdd(reinterpret_cast<u64*>(&lp_acc[(size_t)c * P.N + root]), (u64)x);
}

__global__ __launch_bounds__(MN_CC_SUM_THREADS) void mn_cc_class_sums(
    ImgParams P, ObjState S, unsigned char* __restrict__ cls0, i64* __restrict__ lp_acc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  u64* s_val = reinterpret_cast<u64*>(cc_smem);                   // [SLOTS][C+1], index C = count
  __shared__ int s_root[MN_CC_SUM_SLOTS];
  const int nval = MN_CC_SUM_SLOTS * (P.C + 1);
  for (int i = threadIdx.x; i < nval; i += MN_CC_SUM_THREADS) s_val[i] = 0;
  if (threadIdx.x < MN_CC_SUM_SLOTS) s_root[threadIdx.x] = -1;
  __syncthreads();
  const int n4 = P.N >> 2;
  const int i = blockIdx.x * MN_CC_SUM_THREADS + threadIdx.x;
  if (i < n4) {
    const int4 r = *reinterpret_cast<const int4*>(S.parent + 4 * (size_t)i);
    const bool same = r.x == r.y && r.x == r.z && r.x == r.w;
    const int s0 = mn_lds_root_slot(s_root, r.x);
    const int s1 = same ? s0 : mn_lds_root_slot(s_root, r.y);
    const int s2 = same ? s0 : mn_lds_root_slot(s_root, r.z);
    const int s3 = same ? s0 : mn_lds_root_slot(s_root, r.w);
    float4 best;
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    // the next plane's load is issued before this plane's values are used (one extra float4 of
    // registers: the kernel stays at two 1024-thread blocks per CU, which matters more here than
    // deeper staging -- three planes in flight at 85 VGPRs measured slower)
    float4 nxt = *reinterpret_cast<const float4*>(P.cls + 4 * (size_t)i);
    for (int c = 0; c < P.C; c++) {
      {
        float4 v = nxt;
        if (c + 1 < P.C)
          nxt = *reinterpret_cast<const float4*>(P.cls + (size_t)(c + 1) * P.N + 4 * (size_t)i);
        if (P.clip) { v.x = mn_clip(v.x); v.y = mn_clip(v.y); v.z = mn_clip(v.z); v.w = mn_clip(v.w); }
        float4 l;
        l.x = logf(v.x); l.y = logf(v.y); l.z = logf(v.z); l.w = logf(v.w);
        if (c == 0) {
          best = l;
        } else {
          if (l.x > best.x) { best.x = l.x; b0 = c; }
          if (l.y > best.y) { best.y = l.y; b1 = c; }
          if (l.z > best.z) { best.z = l.z; b2 = c; }
          if (l.w > best.w) { best.w = l.w; b3 = c; }
        }
        // float * 2^32 is exact, so each term is the double-precision product rounded to nearest
        const i64 f0 = __float2ll_rn(l.x * 4294967296.0f), f1 = __float2ll_rn(l.y * 4294967296.0f);
        const i64 f2 = __float2ll_rn(l.z * 4294967296.0f), f3 = __float2ll_rn(l.w * 4294967296.0f);
        if (same) {
          mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c, s0, (f0 + f1) + (f2 + f3));
        } else {
          mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, c, s0, f0);
          mn_cc_add(P, S, s_root, s_val, lp_acc, r.y, c, s1, f1);
          mn_cc_add(P, S, s_root, s_val, lp_acc, r.z, c, s2, f2);
          mn_cc_add(P, S, s_root, s_val, lp_acc, r.w, c, s3, f3);
        }
      }
    }
    if (same) {
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 4);
    } else {
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.x, P.C, s0, 1);
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.y, P.C, s1, 1);
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.z, P.C, s2, 1);
      mn_cc_add(P, S, s_root, s_val, lp_acc, r.w, P.C, s3, 1);
    }
    uchar4 o;
    o.x = (unsigned char)b0; o.y = (unsigned char)b1; o.z = (unsigned char)b2; o.w = (unsigned char)b3;
    *reinterpret_cast<uchar4*>(S.ocls + 4 * (size_t)i) = o;
    *reinterpret_cast<uchar4*>(cls0 + 4 * (size_t)i) = o;
  }
  // tail pixels when N is not a multiple of 4: straight to the global sums
  const int tail0 = n4 << 2;
  if (i < P.N - tail0) {
    const int p = tail0 + i;
    const int root = S.parent[p];
    float best = 0.0f;
    int b = 0;
    for (int c = 0; c < P.C; c++) {
      const float l = logf(mn_ld_class(P, c, p));
      if (c == 0 || l > best) { best = l; b = c; }
      mn_cc_add(P, S, s_root, s_val, lp_acc, root, c, -1, __float2ll_rn(l * 4294967296.0f));
    }
    mn_cc_add(P, S, s_root, s_val, lp_acc, root, P.C, -1, 1);
    S.ocls[p] = (unsigned char)b;
    cls0[p] = (unsigned char)b;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nval; j += MN_CC_SUM_THREADS) {
    const u64 v = s_val[j];
    if (v == 0) continue;
    const int slot = j / (P.C + 1), c = j - slot * (P.C + 1);
    mn_cc_add(P, S, s_root, s_val, lp_acc, s_root[slot], c, -1, (i64)v);
  }
}

// insert with a bounded probe sequence: the table is sized for "few records between components";
// an input that is not separable may produce millions, so a full table must end the sweep (the
// caller falls back) instead of spinning
__device__ __forceinline__ bool mn_tab_insert_bounded(const HashTab& T, u64 key, i64 s,
                                                      int* __restrict__ tcount = nullptr,
                                                      int count = 0) {
  unsigned slot = mn_hash(key) & T.mask;
#pragma unroll 1
  for (int t = 0; t < 256; t++) {
    const u64 prev = atomicCAS(&T.key[slot], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) {
      atomicAdd(reinterpret_cast<u64*>(&T.S[slot]), (u64)s);
      if (tcount) atomicAdd(&tcount[slot], count);      // pixel edges folded into this record
      T.touched[slot] = 1;
      return true;
    }
    slot = (slot + 1) & T.mask;
  }
  return false;
}

// Conditions (a), (b) on every edge; records between components summed into the table:
// wave-aggregated by key, then collected in a per-block LDS table so that a record shared by the
// whole boundary of a large instance costs one global insert per block.
#define MN_CC_EDGE_THREADS 256   /* measured at 1024x2048: 1024 -> 52.1 us, 512 -> 49.7, 256 -> 47.3 */
#define MN_CC_EDGE_G 5          /* offsets staged together: 3 -> 52.2 us, 5 -> 52.1, 10 -> 57.1 */
#define MN_CC_EDGE_SLOTS 256
__device__ __forceinline__ bool mn_cc_lds_add(u64* s_key, u64* s_sum, const HashTab& T, u64 key,
                                              i64 s) {
  unsigned h = (mn_hash(key) >> 7) & (MN_CC_EDGE_SLOTS - 1);
#pragma unroll 1
  for (int t = 0; t < 32; t++) {
    u64 prev = s_key[h];                       // plain read first: the slot is usually there already
    if (prev == MN_EMPTY) prev = atomicCAS(&s_key[h], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) { atomicAdd(&s_sum[h], (u64)s); return true; }
    h = (h + 1) & (MN_CC_EDGE_SLOTS - 1);
  }
  return mn_tab_insert_bounded(T, key, s);    // block table crowded: straight to the global one
}

// with the number of pixel edges per record alongside the sum (edge sweep of the fast certificate)
__device__ __forceinline__ bool mn_cc_lds_add_cnt(u64* s_key, u64* s_sum, int* s_cnt, const HashTab& T,
                                                  int* __restrict__ tcount, u64 key, i64 s, int n) {
  unsigned h = (mn_hash(key) >> 7) & (MN_CC_EDGE_SLOTS - 1);
#pragma unroll 1
  for (int t = 0; t < 32; t++) {
    u64 prev = s_key[h];
    if (prev == MN_EMPTY) prev = atomicCAS(&s_key[h], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) {
      atomicAdd(&s_sum[h], (u64)s);
      atomicAdd(&s_cnt[h], n);
      return true;
    }
    h = (h + 1) & (MN_CC_EDGE_SLOTS - 1);
  }
  return mn_tab_insert_bounded(T, key, s, tcount, n);
}

__global__ __launch_bounds__(MN_CC_EDGE_THREADS) void mn_cc_edges(ImgParams P, ObjState S, HashTab T,
                                                                  const unsigned char* __restrict__ cls0,
                                                                  int* __restrict__ violations) {
  __shared__ u64 s_key[MN_CC_EDGE_SLOTS];
  __shared__ u64 s_sum[MN_CC_EDGE_SLOTS];
  if (threadIdx.x < MN_CC_EDGE_SLOTS) { s_key[threadIdx.x] = MN_EMPTY; s_sum[threadIdx.x] = 0; }
  __syncthreads();
  const int tile = mn_xcd_tile((P.N + MN_CC_EDGE_THREADS - 1) / MN_CC_EDGE_THREADS, P.banded);
  const int p = tile < 0 ? P.N : tile * MN_CC_EDGE_THREADS + threadIdx.x;
  const bool live = p < P.N;
  int bad = 0, over = 0;
  const int root = live ? S.parent[p] : 0;
  if (live && cls0[p] != cls0[root]) bad++;                          // (c) one class per component
  const int r = live ? p / P.W : 0, c0 = live ? p - r * P.W : 0;
  // a pixel next to a boundary sees the same neighbouring component through most of its offsets:
  // the lane sums its cross edges while the key stays the same and adds to the block table (LDS
  // atomics) only when it changes
  u64 ckey = MN_EMPTY;
  i64 csum = 0;
  constexpr int G = 10;                                 // offsets whose loads are in flight together
  for (int k0 = 0; k0 < P.O; k0 += G) {
    float v[G];
    int rq[G];
    bool in[G];
#pragma unroll
    for (int j = 0; j < G; j++) {
      const int k = k0 + j;
      in[j] = false;
      v[j] = 0.5f;
      rq[j] = root;
      if (live && k < P.O) {
        const int rr = r + P.di[k], cc = c0 + P.dj[k];
        if (rr >= 0 && rr < P.H && cc >= 0 && cc < P.W) {
          in[j] = true;
          v[j] = P.same[(size_t)k * P.N + p];
          rq[j] = S.parent[rr * P.W + cc];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < G; j++) {
      if (!in[j]) continue;
      // the gain omf * log-odds has the sign of v - 0.5 (the fixed-point log-odds of a float
      // next to 0.5 are still 2^7 units); sep_hi / sep_lo widen 0.5 by the margin that keeps the
      // float32 priority of a record strictly on its side of the bias (fill_params)
      const float x = mn_same_value(P, v[j]);
      if (rq[j] == root) { if (!(x >= P.sep_hi)) bad++; continue; }  // (a)
      if (!(x <= P.sep_lo)) bad++;                                   // (b)
      const u64 key = mn_key(root, rq[j]);
      const i64 s = mn_edge_fixed(x);
      if (key == ckey) { csum += s; continue; }
      if (ckey != MN_EMPTY && !mn_cc_lds_add(s_key, s_sum, T, ckey, csum)) over++;
      ckey = key;
      csum = s;
    }
  }
  if (ckey != MN_EMPTY && !mn_cc_lds_add(s_key, s_sum, T, ckey, csum)) over++;
  __syncthreads();
  if (threadIdx.x < MN_CC_EDGE_SLOTS && s_key[threadIdx.x] != MN_EMPTY)
    if (!mn_tab_insert_bounded(T, s_key[threadIdx.x], (i64)s_sum[threadIdx.x])) over++;
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(violations, bad);
  if (over) atomicAdd(violations + 1, over);   // table full: not a verdict on the maps
}

// The same sweep with 4 consecutive pixels of one row per lane (W % 4 == 0): the sameness values
// come as one 16-byte load per offset, the four neighbour roots as one (unaligned) 16-byte load,
// and a lane next to a boundary folds up to 40 cross edges into its running sum before it touches
// the block table.
struct __attribute__((packed, aligned(4))) mn_int4u { int x, y, z, w; };
__device__ __forceinline__ int4 mn_ld_int4_unaligned(const int* __restrict__ p) {
  const mn_int4u t = *reinterpret_cast<const mn_int4u*>(p);
  return make_int4(t.x, t.y, t.z, t.w);
}

// This sweep also lays the ground for the certificate and the log-likelihood of the final
// partition, so that no further sweep over the sameness planes is needed after the merge
// (mn_cc_certificate): per block the float64 sums  sum log v  over edges inside components and
// sum log(1-v)  over edges between components (partial[2b], partial[2b+1]), and per record the
// number of pixel edges folded into it (tcount, parallel to the table).
__global__ __launch_bounds__(MN_CC_EDGE_THREADS) void mn_cc_edges4(ImgParams P, ObjState S, HashTab T,
                                                                   const unsigned char* __restrict__ cls0,
                                                                   int* __restrict__ violations,
                                                                   int* __restrict__ tcount,
                                                                   double* __restrict__ partial) {
  __shared__ u64 s_key[MN_CC_EDGE_SLOTS];
  __shared__ u64 s_sum[MN_CC_EDGE_SLOTS];
  __shared__ int s_cnt[MN_CC_EDGE_SLOTS];
  __shared__ double s_part[2][MN_CC_EDGE_THREADS / 64];
  if (threadIdx.x < MN_CC_EDGE_SLOTS) {
    s_key[threadIdx.x] = MN_EMPTY; s_sum[threadIdx.x] = 0; s_cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  double t_same = 0.0, t_diff = 0.0;
  int ccnt = 0;
  const int n4 = P.N >> 2;
  const int i = blockIdx.x * MN_CC_EDGE_THREADS + threadIdx.x;
  const bool live = i < n4;
  const int p0 = live ? 4 * i : 0;
  int bad = 0, over = 0;
  const int r = p0 / P.W, c0 = p0 - r * P.W;
  int root0 = 0, root1 = 0, root2 = 0, root3 = 0;
  if (live) {
    const int4 rv = *reinterpret_cast<const int4*>(S.parent + p0);
    root0 = rv.x; root1 = rv.y; root2 = rv.z; root3 = rv.w;
    const uchar4 own = *reinterpret_cast<const uchar4*>(cls0 + p0);
    bad += (own.x != cls0[root0]) + (own.y != cls0[root1]) +         // (c) one class per component
           (own.z != cls0[root2]) + (own.w != cls0[root3]);
  }
  u64 ckey = MN_EMPTY;
  i64 csum = 0;
  constexpr int G = MN_CC_EDGE_G;                                  // offsets whose loads are in flight together
  for (int k0 = 0; k0 < P.O; k0 += G) {
    float4 v[G];
    int4 rq[G];
    int first[G];                                       // column of the first neighbour, or INT_MIN
#pragma unroll
    for (int j = 0; j < G; j++) {
      const int k = k0 + j;
      first[j] = INT_MIN;
      if (live && k < P.O) {
        const int rr = r + P.di[k];
        if (rr >= 0 && rr < P.H) {
          first[j] = c0 + P.dj[k];
          v[j] = *reinterpret_cast<const float4*>(P.same + (size_t)k * P.N + p0);
          const long long q0 = (long long)rr * P.W + first[j];
          if (q0 >= 0 && q0 + 3 < P.N) {
            rq[j] = mn_ld_int4_unaligned(S.parent + q0);
          } else {                                      // first / last pixels of the image
            rq[j].x = (q0 >= 0 && q0 < P.N) ? S.parent[q0] : 0;
            rq[j].y = (q0 + 1 >= 0 && q0 + 1 < P.N) ? S.parent[q0 + 1] : 0;
            rq[j].z = (q0 + 2 >= 0 && q0 + 2 < P.N) ? S.parent[q0 + 2] : 0;
            rq[j].w = (q0 + 3 >= 0 && q0 + 3 < P.N) ? S.parent[q0 + 3] : 0;
          }
        }
      }
    }
    auto edge = [&](int col, float raw, int q, int own) {
      if (col < 0 || col >= P.W) return;
      const float x = mn_same_value(P, raw);                         // margins: see mn_cc_edges
      if (q == own) {                                                // (a)
        if (!(x >= P.sep_hi)) bad++;
        t_same += (double)logf(x);
        return;
      }
      if (!(x <= P.sep_lo)) bad++;                                   // (b)
      const u64 key = mn_key(own, q);
      const float ld = mn_log1m(x);
      t_diff += (double)ld;
      const i64 sx = __float2ll_rn((logf(x) - ld) * (float)MN_FIX_ONE);   // = mn_edge_fixed(x)
      if (key == ckey) { csum += sx; ccnt++; return; }
      if (ckey != MN_EMPTY && !mn_cc_lds_add_cnt(s_key, s_sum, s_cnt, T, tcount, ckey, csum, ccnt)) over++;
      ckey = key;
      csum = sx;
      ccnt = 1;
    };
#pragma unroll
    for (int j = 0; j < G; j++) {
      if (first[j] == INT_MIN) continue;
      edge(first[j], v[j].x, rq[j].x, root0);
      edge(first[j] + 1, v[j].y, rq[j].y, root1);
      edge(first[j] + 2, v[j].z, rq[j].z, root2);
      edge(first[j] + 3, v[j].w, rq[j].w, root3);
    }
  }
  if (ckey != MN_EMPTY && !mn_cc_lds_add_cnt(s_key, s_sum, s_cnt, T, tcount, ckey, csum, ccnt)) over++;
  for (int off = 32; off > 0; off >>= 1) {
    bad += __shfl_xor(bad, off);
    t_same += __shfl_xor(t_same, off);
    t_diff += __shfl_xor(t_diff, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (bad) atomicAdd(violations, bad);
    s_part[0][threadIdx.x >> 6] = t_same;
    s_part[1][threadIdx.x >> 6] = t_diff;
  }
  __syncthreads();
  int late = 0;
  if (threadIdx.x < MN_CC_EDGE_SLOTS && s_key[threadIdx.x] != MN_EMPTY)
    if (!mn_tab_insert_bounded(T, s_key[threadIdx.x], (i64)s_sum[threadIdx.x], tcount, s_cnt[threadIdx.x]))
      late = 1;
  if (late || over) atomicAdd(violations + 1, late + over);   // table full: not a verdict on the maps
  if (threadIdx.x < 2) {                                // block order: the sum is reproducible
    double t = 0.0;
    for (int w = 0; w < MN_CC_EDGE_THREADS / 64; w++) t += s_part[threadIdx.x][w];
    partial[(size_t)blockIdx.x * 2 + threadIdx.x] = t;
  }
}

// After the merge: what the sweeps above could not know.  One lane per pixel, work only at the
// component roots (compsize > 0): the class term of the log-likelihood  lp[cls]  of every final
// object, and the pixels whose own arg-max class differs from their final object's class (a
// component has one class, so that is the component's size or nothing).
#define MN_CC_CERT_THREADS 1024
__global__ __launch_bounds__(MN_CC_CERT_THREADS) void mn_cc_certificate(ImgParams P, ObjState S,
                                                         const unsigned char* __restrict__ cls0,
                                                         const int* __restrict__ compsize,
                                                         double* __restrict__ partial_cls,
                                                         int* __restrict__ violations) {
  __shared__ double sh[MN_CC_CERT_THREADS / 64];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double t_cls = 0.0;
  int bad_cls = 0;
  if (p < P.N) {
    const int cs = compsize[p];
    if (cs > 0) {
      int f = p;
      while (S.parent[f] != f) f = S.parent[f];
      const int oc = S.ocls[f];
      if (cls0[p] != oc) bad_cls = cs;
      if (f == p) t_cls = (double)mn_obj_lp(P, S, S.lpvalid[p] != 0, p, oc);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    t_cls += __shfl_xor(t_cls, off);
    bad_cls += __shfl_xor(bad_cls, off);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[threadIdx.x >> 6] = t_cls;
    if (bad_cls) atomicAdd(violations + 3, bad_cls);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < MN_CC_CERT_THREADS / 64; w++) t += sh[w];
    partial_cls[blockIdx.x] = t;
  }
}

// total = class term + omf * (sum over edges: log v inside final objects, log(1-v) between them).
// The edge sweep summed it for the components; a record merged afterwards moves its edges from
// "between" to "inside", i.e. adds its log-odds sum, and makes each of them an edge whose sign
// contradicts the partition (the finisher accumulated both).
__global__ __launch_bounds__(256) void mn_cc_cert_reduce(int nb_edges, const double* __restrict__ partial_edges,
                                                         int nb_cls, const double* __restrict__ partial_cls,
                                                         const Counters* __restrict__ cnt, float omf,
                                                         double* __restrict__ out,
                                                         int* __restrict__ violations) {
  __shared__ double sh[3][256];
  double a[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nb_cls; b += 256) a[0] += partial_cls[b];
  for (int b = threadIdx.x; b < nb_edges; b += 256) {
    a[1] += partial_edges[(size_t)b * 2];
    a[2] += partial_edges[(size_t)b * 2 + 1];
  }
  for (int j = 0; j < 3; j++) sh[j][threadIdx.x] = a[j];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off)
      for (int j = 0; j < 3; j++) sh[j][threadIdx.x] += sh[j][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double moved = (double)cnt->merged_S * (1.0 / MN_FIX_ONE);
    out[0] = sh[0][0] + ((sh[2][0] + sh[1][0]) + moved) * (double)omf;
    out[1] = sh[0][0]; out[2] = sh[1][0]; out[3] = sh[2][0];
    if (cnt->merged_E) atomicAdd(violations, cnt->merged_E);
  }
}

__global__ __launch_bounds__(256) void mn_cc_finish(ImgParams P, ObjState S,
                                                    const i64* __restrict__ lp_acc,
                                                    int* __restrict__ compsize) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  const bool is_root = S.parent[p] == p;
  compsize[p] = is_root ? S.osize[p] : 0;  // kept for mn_cc_certificate: osize grows in the merge
  if (!is_root) return;
  if (S.osize[p] <= 1) return;            // a lone pixel keeps reading its class planes
  for (int c = 0; c < P.C; c++)
    S.lpsum[(size_t)c * P.N + p] = (float)((double)lp_acc[(size_t)c * P.N + p] * (1.0 / MN_LP_FIX));
  S.lpvalid[p] = 1;
}
