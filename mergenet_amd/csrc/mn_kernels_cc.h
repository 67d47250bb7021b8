// mn_kernels_cc.h -- component contraction: phase 1 of the merge in one union-find sweep, for
// inputs where phase 1 is provably order-independent.
//
// Claim (csegment variant, object_merge_factor > 0, merge_logprob_bias >= 0).  Let the pixel
// graph be SIGN-SEPARABLE: the connected components of the edges with positive log-odds are such
// that (a) no edge inside a component has log-odds <= 0, (b) no edge between components has
// log-odds >= 0, (c) all pixels of a component share one arg-max class.  Then at every moment of
// the reference's run (utils/csegment/segment.cc:539-727) every record between two sub-objects of
// one component scores  sum(log-odds)*omf / (n1+n2) + bias  >  bias  (class delta 0, segment.cc:
// 107-150) and every record between sub-objects of different components scores < bias (negative
// log-odds, class delta <= 0).  The queue pops in descending priority, so no cross record is
// popped while any intra record is alive: the reference first merges every component completely
// -- in whatever order -- and only then turns to the records between components.  The state at
// that moment (objects = components, one fully summed record per adjacent pair) is what this
// file builds directly:
//   mn_cc_hook     lock-free union-find over the implicit pixel graph (positive edges), root =
//                  lowest pixel id of the component;
//   mn_cc_flatten  parent[p] = root;
//   mn_cc_sums     condition (c), component sizes and class log-prob sums (running sums per
//                  wave chunk, 64-bit fixed-point atomics: order-independent);
//   mn_cc_edges    conditions (a), (b) on every edge, records between components summed into
//                  the hash table (wave-aggregated by key);
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

// Row stage, no atomics: a wave covers 64 consecutive pixels; lanes joined by positive edges of the
// horizontal unit offset (index kh, direction +1 column) form runs, and every pixel points at the
// first pixel of its run (found with one ballot and bit arithmetic).  The union-find sweep then
// starts from flat trees of up to 64 pixels instead of single pixels.
__global__ __launch_bounds__(256) void mn_cc_rows(ImgParams P, int* __restrict__ parent, int kh) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool link = false;                         // positive edge between this pixel and the next one
  if (p < P.N) {
    const int c = p % P.W;
    if (c + 1 < P.W && p + 1 < P.N) {
      const float v = mn_same_value(P, P.same[(size_t)kh * P.N + p]);
      link = v > 0.5f && mn_fixed_to_float(mn_edge_fixed(v)) * P.omf > 0.0f;
    }
  }
  const u64 m = __ballot(link);
  if (p >= P.N) return;
  // run start = one past the highest lane below `lane` that has NO link to its successor
  const u64 below = lane ? (~m & ((1ull << lane) - 1ull)) : 0ull;
  const int start = below ? (64 - __clzll((long long)below)) : 0;
  parent[p] = p - lane + start;
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
  for (int k = k0; k < k1; k++) {
    bool want = false;
    int a = 0, b = 0;
    if (live) {
      const int rr = r + P.di[k], cc = c + P.dj[k];
      if (rr >= 0 && rr < P.H && cc >= 0 && cc < P.W) {
        const float v = mn_same_value(P, P.same[(size_t)k * P.N + p]);
        if (v > 0.5f && mn_fixed_to_float(mn_edge_fixed(v)) * P.omf > 0.0f) {   // log-odds > 0
          a = mn_cc_find(parent, p);
          b = mn_cc_find(parent, rr * P.W + cc);
          want = a != b;
        }
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

__global__ __launch_bounds__(256) void mn_cc_flatten(int N, int* __restrict__ parent) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  int x = p;
  while (parent[x] != x) x = parent[x];
  parent[p] = x;
}

// wave-level "sum by key": lanes holding the same key (and active) add their values; the first lane
// of each group returns true and the group's sum
__device__ __forceinline__ bool mn_wave_sum_by_key(bool active, u64 key, i64* value) {
  bool leader = false;
  u64 todo = __ballot(active);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const u64 k0 = ((u64)(unsigned)__shfl((int)(key >> 32), first) << 32) |
                   (u64)(unsigned)__shfl((int)(key & 0xFFFFFFFFull), first);
    const bool mine = active && key == k0;
    const u64 grp = __ballot(mine);
    i64 s = mine ? *value : 0;
    for (int off = 32; off > 0; off >>= 1) {
      const long long hi = __shfl_xor((int)(s >> 32), off);
      const unsigned lo = (unsigned)__shfl_xor((int)(s & 0xFFFFFFFFll), off);
      s += (i64)(((u64)(unsigned)hi << 32) | lo);
    }
    if (lane == first) { leader = true; *value = s; }
    todo &= ~grp;
  }
  return leader;
}

// Component sizes and class log-prob sums.  A wave walks MN_CC_CHUNK consecutive pixels; every lane
// keeps a running sum for "its" root and the wave only flushes (wave-aggregated by root, one atomic
// per distinct root) when some lane's root changes or the chunk ends -- a 1.6 M-pixel background
// would otherwise serialise ten thousand atomics on one word per class.
#define MN_CC_CHUNK 512
#define MN_CC_ITERS (MN_CC_CHUNK / 64)
__global__ __launch_bounds__(256) void mn_cc_sums(ImgParams P, ObjState S,
                                                  const unsigned char* __restrict__ cls0,
                                                  i64* __restrict__ lp_acc,
                                                  int* __restrict__ violations) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int begin = wave_global * MN_CC_CHUNK;
  if (begin >= P.N) return;
  // roots of this lane's pixels and the iterations before which the wave must flush (some lane
  // changes root there) are the same for every plane: computed once
  int root[MN_CC_ITERS];
  bool flush_before[MN_CC_ITERS];
  int bad = 0;
#pragma unroll
  for (int i = 0; i < MN_CC_ITERS; i++) {
    const int p = begin + i * 64 + lane;
    root[i] = p < P.N ? S.parent[p] : -1;
    if (p < P.N && cls0[p] != cls0[root[i]]) bad++;               // (c) one class per component
  }
#pragma unroll
  for (int i = 0; i < MN_CC_ITERS; i++) {
    const bool chg = i > 0 && root[i] >= 0 && root[i - 1] >= 0 && root[i] != root[i - 1];
    flush_before[i] = __ballot(chg) != 0;
    if (i > 0 && root[i] < 0) root[i] = root[i - 1];              // tail lanes keep their last root
  }
  // plane -1 = pixel counts, planes 0..C-1 = class log-probs
  // Big components are hot words for these atomics (every wave of the image adds to the same
  // few roots), and one word takes ~88 atomics/us: waves walk the planes in staggered order so the
  // adds of one moment spread over C+1 words per root.
  const int nplanes = P.C + 1;
  const int shift = wave_global % nplanes;
  for (int ci = 0; ci < nplanes; ci++) {
    int c = ci + shift;
    if (c >= nplanes) c -= nplanes;
    c -= 1;
    float val[MN_CC_ITERS];
#pragma unroll
    for (int i = 0; i < MN_CC_ITERS; i++) {
      const int p = begin + i * 64 + lane;
      val[i] = (c >= 0 && p < P.N) ? mn_ld_class(P, c, p) : 1.0f;
    }
    i64 acc = 0;
    int cur = -1;
#pragma unroll
    for (int i = 0; i < MN_CC_ITERS; i++) {
      const int p = begin + i * 64 + lane;
      if (flush_before[i]) {
        if (mn_wave_sum_by_key(cur >= 0, (u64)(unsigned)cur, &acc)) {
          if (c < 0) atomicAdd(&S.osize[cur], (int)acc);
          else atomicAdd(reinterpret_cast<u64*>(&lp_acc[(size_t)c * P.N + cur]), (u64)acc);
        }
        acc = 0;
        cur = -1;
      }
      if (p < P.N) {
        cur = root[i];
        acc += (c < 0) ? (i64)1 : __double2ll_rn((double)logf(val[i]) * MN_LP_FIX);
      }
    }
    if (mn_wave_sum_by_key(cur >= 0, (u64)(unsigned)cur, &acc)) {
      if (c < 0) atomicAdd(&S.osize[cur], (int)acc);
      else atomicAdd(reinterpret_cast<u64*>(&lp_acc[(size_t)c * P.N + cur]), (u64)acc);
    }
  }
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
  if (lane == 0 && bad) atomicAdd(violations, bad);
}

// insert with a bounded probe sequence: the table is sized for "few records between components";
// an input that is not separable may produce millions, so a full table must end the sweep (the
// caller falls back) instead of spinning
__device__ __forceinline__ bool mn_tab_insert_bounded(const HashTab& T, u64 key, i64 s) {
  unsigned slot = mn_hash(key) & T.mask;
  for (int t = 0; t < 256; t++) {
    const u64 prev = atomicCAS(&T.key[slot], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) {
      atomicAdd(reinterpret_cast<u64*>(&T.S[slot]), (u64)s);
      T.touched[slot] = 1;
      return true;
    }
    slot = (slot + 1) & T.mask;
  }
  return false;
}

// Conditions (a), (b) on every edge; records between components summed into the table
// (wave-aggregated by key before the insert).
__global__ __launch_bounds__(256) void mn_cc_edges(ImgParams P, ObjState S, HashTab T,
                                                   int* __restrict__ violations) {
  const int tile = mn_xcd_tile((P.N + 255) >> 8, P.banded);
  if (tile < 0) return;
  const int p = tile * 256 + threadIdx.x;
  const bool live = p < P.N;
  int bad = 0;
  const int root = live ? S.parent[p] : 0;
  const int r = live ? p / P.W : 0, c0 = live ? p - r * P.W : 0;
  constexpr int G = 5;                                  // offsets whose loads are in flight together
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
      bool cross = false;
      u64 key = 0;
      i64 s = 0;
      if (in[j]) {
        s = mn_edge_fixed(mn_same_value(P, v[j]));
        const float g = mn_fixed_to_float(s) * P.omf;
        if (rq[j] == root) { if (!(g > 0.0f)) bad++; }               // (a)
        else { cross = true; key = mn_key(root, rq[j]); if (!(g < 0.0f)) bad++; }   // (b)
      }
      if (__ballot(cross) == 0) continue;
      if (mn_wave_sum_by_key(cross, key, &s)) {
        if (!mn_tab_insert_bounded(T, key, s)) bad++;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(violations, bad);
}

__global__ __launch_bounds__(256) void mn_cc_finish(ImgParams P, ObjState S,
                                                    const i64* __restrict__ lp_acc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N || S.parent[p] != p) return;
  if (S.osize[p] <= 1) return;            // a lone pixel keeps reading its class planes
  for (int c = 0; c < P.C; c++)
    S.lpsum[(size_t)c * P.N + p] = (float)((double)lp_acc[(size_t)c * P.N + p] * (1.0 / MN_LP_FIX));
  S.lpvalid[p] = 1;
}
