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
//   mn_cc_tiles    16 x 64-pixel tiles labelled in LDS over the two unit offsets;
//   mn_cc_hook     lock-free union-find over the implicit pixel graph (positive edges), root =
//                  lowest pixel id of the component; unit offsets first, then the rest;
//   mn_cc_flatten  parent[p] = root (and the roots' accumulators cleared);
//   mn_cc_sums     condition (c), component sizes and class log-prob sums (running sums per
//                  lane, block table in LDS, 64-bit fixed-point atomics: order-independent);
//   mn_cc_edges    conditions (a), (b) on every edge, records between components summed into
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
// only the roots' slots of the C planes (and of the best-record array) are ever used, so they are
// cleared here instead of by a memset of all (C + 1) * N words.
__global__ __launch_bounds__(256) void mn_cc_flatten(int N, int C, int* __restrict__ parent,
                                                     int* __restrict__ osize,
                                                     i64* __restrict__ lp_acc,
                                                     u64* __restrict__ best) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  int x = p;
  while (parent[x] != x) x = parent[x];
  parent[p] = x;
  if (lp_acc) {
    osize[p] = 0;
    if (x == p) {
      best[p] = 0;                            // best-record slot, filled by mn_compact
      for (int c = 0; c < C; c++) lp_acc[(size_t)c * N + p] = 0;
    }
  }
}

// Component sizes and class log-prob sums.  A wave walks MN_CC_CHUNK consecutive pixels of one
// plane (blockIdx.y); every lane keeps a running sum for "its" root and adds it to a per-block
// table in LDS when some lane's root changes or the chunk ends.  The block then issues ONE global
// atomic per root: a 1.6 M-pixel background is a hot word for every wave of the image, and one
// word takes only ~88 atomics/us.
#define MN_CC_CHUNK 512
#define MN_CC_ITERS (MN_CC_CHUNK / 64)
#define MN_CC_SUM_THREADS 1024
#define MN_CC_SUM_SLOTS 128
#define MN_CC_SUM_PLANES 1
__device__ __forceinline__ int mn_lds_root_slot(int* s_root, int root) {
  unsigned h = ((unsigned)root * 2654435761u) >> 25;             // 7 bits
  for (int t = 0; t < MN_CC_SUM_SLOTS; t++) {
    int cur = s_root[h];                       // plain read first: the slot is usually there already
    if (cur == root) return (int)h;
    if (cur == -1) cur = atomicCAS(&s_root[h], -1, root);
    if (cur == -1 || cur == root) return (int)h;
    h = (h + 1) & (MN_CC_SUM_SLOTS - 1);
  }
  return -1;
}

__global__ __launch_bounds__(MN_CC_SUM_THREADS) void mn_cc_sums(ImgParams P, ObjState S,
                                                                const unsigned char* __restrict__ cls0,
                                                                i64* __restrict__ lp_acc,
                                                                int* __restrict__ violations) {
  __shared__ int s_root[MN_CC_SUM_SLOTS];
  __shared__ u64 s_val[MN_CC_SUM_PLANES][MN_CC_SUM_SLOTS];
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long begin_ll = (long long)wave_global * MN_CC_CHUNK;
  const int begin = begin_ll < P.N ? (int)begin_ll : P.N;        // idle waves still reach the barriers
  if (threadIdx.x < MN_CC_SUM_SLOTS) {
    s_root[threadIdx.x] = -1;
    for (int j = 0; j < MN_CC_SUM_PLANES; j++) s_val[j][threadIdx.x] = 0;
  }
  // roots of this lane's pixels and the iterations before which the wave must flush (some lane
  // changes root there) are the same for every plane: computed once
  int root[MN_CC_ITERS];
  bool flush_before[MN_CC_ITERS];
  int bad = 0;
#pragma unroll
  for (int i = 0; i < MN_CC_ITERS; i++) {
    const int p = begin + i * 64 + lane;
    root[i] = p < P.N ? S.parent[p] : -1;
    if (blockIdx.y == 0 && p < P.N && cls0[p] != cls0[root[i]]) bad++;   // (c) one class per component
  }
#pragma unroll
  for (int i = 0; i < MN_CC_ITERS; i++) {
    const bool chg = i > 0 && root[i] >= 0 && root[i - 1] >= 0 && root[i] != root[i - 1];
    flush_before[i] = __ballot(chg) != 0;
    if (i > 0 && root[i] < 0) root[i] = root[i - 1];              // tail lanes keep their last root
  }
  __syncthreads();
  // blockIdx.y selects MN_CC_SUM_PLANES consecutive planes (plane -1 = pixel counts, planes
  // 0..C-1 = class log-probs); their loads are all issued before the first one is used
  const int c_first = (int)blockIdx.y * MN_CC_SUM_PLANES - 1;
  float val[MN_CC_SUM_PLANES][MN_CC_ITERS];
#pragma unroll
  for (int j = 0; j < MN_CC_SUM_PLANES; j++) {
    const int c = c_first + j;
#pragma unroll
    for (int i = 0; i < MN_CC_ITERS; i++) {
      const int p = begin + i * 64 + lane;
      val[j][i] = (c >= 0 && c < P.C && p < P.N) ? mn_ld_class(P, c, p) : 1.0f;
    }
  }
#pragma unroll
  for (int j = 0; j < MN_CC_SUM_PLANES; j++) {
    const int c = c_first + j;
    if (c >= P.C) break;                      // uniform
    i64 acc = 0;
    int cur = -1;
#pragma unroll
    for (int i = 0; i <= MN_CC_ITERS; i++) {
      if (i == MN_CC_ITERS || flush_before[i]) {
        // every lane adds its own partial sum to the block table: 64 LDS atomics on one address
        // cost less than the dozen cross-lane permutes of a 64-bit wave reduction
        if (cur >= 0) {
          const int slot = mn_lds_root_slot(s_root, cur);
          if (slot >= 0) atomicAdd(&s_val[j][slot], (u64)acc);
          else if (c < 0) atomicAdd(&S.osize[cur], (int)acc);
          else atomicAdd(reinterpret_cast<u64*>(&lp_acc[(size_t)c * P.N + cur]), (u64)acc);
        }
        acc = 0;
        cur = -1;
      }
      if (i < MN_CC_ITERS && begin + i * 64 + lane < P.N) {
        cur = root[i];
        // float * 2^32 is exact, so this is the double-precision product rounded to nearest
        acc += (c < 0) ? (i64)1 : __float2ll_rn(logf(val[j][i]) * 4294967296.0f);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < MN_CC_SUM_SLOTS) {
    const int r = s_root[threadIdx.x];
    if (r >= 0) {
#pragma unroll
      for (int j = 0; j < MN_CC_SUM_PLANES; j++) {
        const int c = c_first + j;
        const u64 v = s_val[j][threadIdx.x];
        if (c >= P.C || v == 0) continue;
        if (c < 0) atomicAdd(&S.osize[r], (int)(i64)v);
        else atomicAdd(reinterpret_cast<u64*>(&lp_acc[(size_t)c * P.N + r]), v);
      }
    }
  }
  if (blockIdx.y != 0) return;                // condition (c) is counted once, by the first block row
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

// Conditions (a), (b) on every edge; records between components summed into the table:
// wave-aggregated by key, then collected in a per-block LDS table so that a record shared by the
// whole boundary of a large instance costs one global insert per block.
#define MN_CC_EDGE_THREADS 1024
#define MN_CC_EDGE_SLOTS 256
__device__ __forceinline__ bool mn_cc_lds_add(u64* s_key, u64* s_sum, const HashTab& T, u64 key,
                                              i64 s) {
  unsigned h = (mn_hash(key) >> 7) & (MN_CC_EDGE_SLOTS - 1);
  for (int t = 0; t < 32; t++) {
    u64 prev = s_key[h];                       // plain read first: the slot is usually there already
    if (prev == MN_EMPTY) prev = atomicCAS(&s_key[h], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) { atomicAdd(&s_sum[h], (u64)s); return true; }
    h = (h + 1) & (MN_CC_EDGE_SLOTS - 1);
  }
  return mn_tab_insert_bounded(T, key, s);    // block table crowded: straight to the global one
}

__global__ __launch_bounds__(MN_CC_EDGE_THREADS) void mn_cc_edges(ImgParams P, ObjState S, HashTab T,
                                                                  int* __restrict__ violations) {
  __shared__ u64 s_key[MN_CC_EDGE_SLOTS];
  __shared__ u64 s_sum[MN_CC_EDGE_SLOTS];
  if (threadIdx.x < MN_CC_EDGE_SLOTS) { s_key[threadIdx.x] = MN_EMPTY; s_sum[threadIdx.x] = 0; }
  __syncthreads();
  const int tile = mn_xcd_tile((P.N + MN_CC_EDGE_THREADS - 1) / MN_CC_EDGE_THREADS, P.banded);
  const int p = tile < 0 ? P.N : tile * MN_CC_EDGE_THREADS + threadIdx.x;
  const bool live = p < P.N;
  int bad = 0;
  const int root = live ? S.parent[p] : 0;
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
      // sign of the gain omf * log-odds: the fixed-point log-odds of a float v has the sign of
      // v - 0.5 (|log-odds| >= 2^-23 next to 0.5, i.e. 2^7 fixed-point units) and the host
      // only takes this path for omf >= 1e-20, so the product cannot underflow to zero
      const float x = mn_same_value(P, v[j]);
      if (rq[j] == root) { if (!(x > 0.5f)) bad++; continue; }       // (a)
      if (!(x < 0.5f)) bad++;                                        // (b)
      const u64 key = mn_key(root, rq[j]);
      const i64 s = mn_edge_fixed(x);
      if (key == ckey) { csum += s; continue; }
      if (ckey != MN_EMPTY && !mn_cc_lds_add(s_key, s_sum, T, ckey, csum)) bad++;
      ckey = key;
      csum = s;
    }
  }
  if (ckey != MN_EMPTY && !mn_cc_lds_add(s_key, s_sum, T, ckey, csum)) bad++;
  __syncthreads();
  if (threadIdx.x < MN_CC_EDGE_SLOTS && s_key[threadIdx.x] != MN_EMPTY)
    if (!mn_tab_insert_bounded(T, s_key[threadIdx.x], (i64)s_sum[threadIdx.x])) bad++;
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
