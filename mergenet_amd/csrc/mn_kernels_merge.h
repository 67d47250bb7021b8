// mn_kernels_merge.h -- phase B, parallel rounds of the lazy-greedy merge.
//
// Reference work replaced: RunSegmentation + Merge (utils/csegment/segment.cc:539-727).  The
// reference pops one record at a time from a priority queue; here every round
//   1. re-scores all live records (fresh) next to their remembered (stored) priority,
//   2. lets every object pick its best visible record (64-bit atomicMax of (priority, partner)),
//   3. merges the records that are the best of BOTH endpoints -- a matching, so merges of one
//      round never share an object -- and repeats 2-3 among still unmatched objects for records
//      whose likelihood gain is positive,
//   4. folds the records of absorbed objects into the survivors' records by re-inserting every
//      record under its relabelled (min,max) key into an open-addressing table, log-odds summed
//      in 2^-30 fixed point so that the sums do not depend on arrival order.
// Laziness is kept: a record remembers the priority it was last scored at (AdjacencyRecord::
// merge_priority); only records incident to an absorbed object are re-scored by a merge
// (segment.cc:650-707), a survivor's other records stay stale until selected ("popped",
// segment.cc:554-565): stale-high ones are refreshed eagerly (order-neutral), stale-low ones keep
// competing with their stored value and are refreshed instead of merged when selected.
#pragma once

#include "mn_device.h"

struct RecList {
  u64* key;      // (u << 32) | v, u < v ; MN_EMPTY = dead
  i64* S;        // summed log-odds, 2^-30 fixed point   (AdjacencyRecord::obj_merge_logprob)
  float* st;     // stored priority                      (AdjacencyRecord::merge_priority)
};

struct HashTab {
  u64* key;
  i64* S;
  float* st;
  unsigned char* touched;
  unsigned mask;   // capacity - 1 (capacity is a power of two)
};

struct Counters {
  int n_records;     // appended by the compaction kernels
  int n_visible;     // records with stored priority >= 0
  int n_merged;      // objects absorbed this round
  int n_selected;    // records selected (merged or refreshed)
  int finisher_steps;
  int finisher_merges;
  int error;
  int pad;
};

// ---- round 0 on the implicit pixel graph ------------------------------------------------------

__global__ __launch_bounds__(256) void mn_init_objects(int N, int* __restrict__ osize,
                                                       int* __restrict__ parent,
                                                       int* __restrict__ mate) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  osize[p] = 1;
  parent[p] = p;
  mate[p] = -1;
}

// p and q merge when each is the other's best (mutually best record).
__global__ __launch_bounds__(256) void mn_pix_match(int N, const u64* __restrict__ best,
                                                    unsigned char* __restrict__ matched,
                                                    int* __restrict__ mate) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const u64 b = best[p];
  if (b == 0) return;
  const int q = mn_pack_partner(b);
  const u64 bq = best[q];
  if (bq == 0 || mn_pack_partner(bq) != p) return;
  matched[p] = 1;
  mate[p] = q;
}

// Merge of two single pixels: the lower id survives (equal sizes keep obj1, segment.cc:612-616).
__global__ __launch_bounds__(256) void mn_pix_apply(ImgParams P, ObjState S,
                                                    const int* __restrict__ mate,
                                                    Counters* __restrict__ cnt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  const int q = mate[p];
  if (q < 0 || q < p) return;       // the lower pixel of a pair does the work
  if (S.parent[p] != p || S.osize[p] != 1) return;   // already merged in an earlier sub-round
  const int cu = S.ocls[p], cv = S.ocls[q];
  int mc = cu;
  float best = 0.0f;
  for (int c = 0; c < P.C; c++) {
    const float a = logf(mn_ld_class(P, c, p));
    const float b = logf(mn_ld_class(P, c, q));
    const float j = a + b;
    S.lpsum[(size_t)c * P.N + p] = j;
    if (cu != cv && (c == 0 || j > best)) { best = j; mc = c; }
  }
  S.lpvalid[p] = 1;
  S.ocls[p] = (unsigned char)mc;
  S.osize[p] = 2;
  S.parent[q] = p;
  atomicAdd(&cnt->n_merged, 1);
}

// ---- open-addressing table of records ---------------------------------------------------------

__device__ __forceinline__ unsigned mn_tab_insert(const HashTab& T, u64 key, i64 s) {
  unsigned slot = mn_hash(key) & T.mask;
  for (;;) {
    const u64 prev = atomicCAS(&T.key[slot], MN_EMPTY, key);
    if (prev == MN_EMPTY || prev == key) break;
    slot = (slot + 1) & T.mask;
  }
  atomicAdd(reinterpret_cast<u64*>(&T.S[slot]), (u64)s);
  return slot;
}

// Records of the pixel graph under the current labelling: every in-bounds (pixel, offset) pair
// whose endpoints lie in different objects, summed per object pair.
__global__ __launch_bounds__(256) void mn_build_from_pixels(ImgParams P, ObjState S, HashTab T) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  const int r = p / P.W, c = p - r * P.W;
  const int u = S.parent[p];
  for (int k = 0; k < P.O; k++) {
    const int rr = r + P.di[k], cc = c + P.dj[k];
    if (rr < 0 || rr >= P.H || cc < 0 || cc >= P.W) continue;
    const int q = rr * P.W + cc;
    const int v = S.parent[q];
    if (u == v) continue;
    const float x = mn_same_value(P, P.same[(size_t)k * P.N + p]);
    const unsigned slot = mn_tab_insert(T, mn_key(u, v), mn_edge_fixed(x));
    T.touched[slot] = 1;
  }
}

// Table -> compact list; touched records get a fresh priority (they were re-keyed or folded).
__global__ __launch_bounds__(256) void mn_compact(ImgParams P, ObjState S, HashTab T, RecList L,
                                                  Counters* __restrict__ cnt) {
  const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot > T.mask) return;
  const u64 key = T.key[slot];
  if (key == MN_EMPTY) return;
  const i64 s = T.S[slot];
  float st;
  if (T.touched[slot]) {
    int mc;
    bool pos;
    st = mn_score(P, S, mn_key_u(key), mn_key_v(key), mn_fixed_to_float(s), &mc, &pos);
  } else {
    st = T.st[slot];
  }
  const int idx = atomicAdd(&cnt->n_records, 1);
  L.key[idx] = key;
  L.S[idx] = s;
  L.st[idx] = st;
}

// ---- rounds on the explicit record list -------------------------------------------------------

// fresh priority of every record; eager refresh of stale-high records; best visible record per
// object (ball = "best of all").
__global__ __launch_bounds__(256) void mn_rec_score(ImgParams P, ObjState S, RecList L, int R,
                                                    float* __restrict__ fresh,
                                                    unsigned char* __restrict__ aux,
                                                    u64* __restrict__ ball,
                                                    Counters* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const u64 key = L.key[i];
  const int u = mn_key_u(key), v = mn_key_v(key);
  int mc;
  bool pos;
  const float f = mn_score(P, S, u, v, mn_fixed_to_float(L.S[i]), &mc, &pos);
  float st = L.st[i];
  if (st >= 0.0f && f < st) { st = f; L.st[i] = f; }
  fresh[i] = f;
  aux[i] = (unsigned char)((mc & 0x7F) | (pos ? 0x80 : 0));   // aux: merged class | gain>0 flag
  if (st >= 0.0f) {
    atomicMax(&ball[u], mn_pack(st, v));
    atomicMax(&ball[v], mn_pack(st, u));
    atomicAdd(&cnt->n_visible, 1);
  }
}

// later sub-rounds: records between two unmatched objects with positive gain propose again
__global__ __launch_bounds__(256) void mn_rec_propose(RecList L, int R,
                                                      const unsigned char* __restrict__ aux,
                                                      const unsigned char* __restrict__ matched,
                                                      u64* __restrict__ bsub) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float st = L.st[i];
  if (!(st >= 0.0f) || !(aux[i] & 0x80)) return;
  const u64 key = L.key[i];
  const int u = mn_key_u(key), v = mn_key_v(key);
  if (matched[u] || matched[v]) return;
  atomicMax(&bsub[u], mn_pack(st, v));
  atomicMax(&bsub[v], mn_pack(st, u));
}

// a record is selected when it is the best of both its endpoints
__global__ __launch_bounds__(256) void mn_rec_match(RecList L, int R, const u64* __restrict__ bcur,
                                                    const unsigned char* __restrict__ aux,
                                                    int later_subround,
                                                    unsigned char* __restrict__ matched,
                                                    unsigned char* __restrict__ sel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float st = L.st[i];
  if (!(st >= 0.0f)) return;
  if (later_subround && !(aux[i] & 0x80)) return;
  const u64 key = L.key[i];
  const int u = mn_key_u(key), v = mn_key_v(key);
  if (bcur[u] != mn_pack(st, v) || bcur[v] != mn_pack(st, u)) return;
  sel[i] = 1;
  matched[u] = 1;
  matched[v] = 1;
}

// Selected records: refresh (stale-low) or merge (segment.cc:560-565, 602-642).
__global__ __launch_bounds__(256) void mn_rec_apply(ImgParams P, ObjState S, RecList L, int R,
                                                    const float* __restrict__ fresh,
                                                    const unsigned char* __restrict__ aux,
                                                    const unsigned char* __restrict__ sel,
                                                    Counters* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R || !sel[i]) return;
  atomicAdd(&cnt->n_selected, 1);
  const float f = fresh[i], st = L.st[i];
  // csegment merges when the re-scored priority equals the popped one (segment.cc:561); a
  // record whose priority rose since it was stored is re-queued with the new value instead.
  // pysegmenter merges on >= (segmenter.py:470).
  if (P.variant == MN_VARIANT_CSEGMENT && f != st) { L.st[i] = f; return; }
  const u64 key = L.key[i];
  int a = mn_key_u(key), b = mn_key_v(key);
  const int na = S.osize[a], nb = S.osize[b];
  if (na < nb) { const int t = a; a = b; b = t; }     // tie keeps the lower id (segment.cc:612)
  const bool va = S.lpvalid[a] != 0, vb = S.lpvalid[b] != 0;
  for (int c = 0; c < P.C; c++)
    S.lpsum[(size_t)c * P.N + a] = mn_obj_lp(P, S, va, a, c) + mn_obj_lp(P, S, vb, b, c);
  S.lpvalid[a] = 1;
  S.ocls[a] = (unsigned char)(aux[i] & 0x7F);
  S.osize[a] = na + nb;
  S.parent[b] = a;
  atomicAdd(&cnt->n_merged, 1);
}

// Re-insert every record under its relabelled key; records inside one object disappear.
__global__ __launch_bounds__(256) void mn_rebuild(ObjState S, RecList L, int R, HashTab T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const u64 key = L.key[i];
  if (key == MN_EMPTY) return;
  const int u = mn_key_u(key), v = mn_key_v(key);
  const int nu = S.parent[u], nv = S.parent[v];
  if (nu == nv) return;
  const unsigned slot = mn_tab_insert(T, mn_key(nu, nv), L.S[i]);
  if (nu != u || nv != v) T.touched[slot] = 1;   // incident to an absorbed object: re-score
  else T.st[slot] = L.st[i];                     // untouched: keeps its stored priority
}
