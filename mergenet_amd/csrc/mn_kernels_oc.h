// mn_kernels_oc.h -- contraction of order-free clusters of OBJECTS (general rounds).
//
// Reference work replaced: the pops of RunSegmentation (utils/csegment/segment.cc:539-573) that
// merge objects whose order cannot matter.  The argument of components mode (mn_kernels_cc.h) does
// not need pixels: take the current objects and the current records, join objects along records
// with summed log-odds >= +tau between objects of one class, and call a cluster VALID when
//   (a) every record inside it has log-odds >= +tau (and joins two objects of the cluster's class),
//   (b) every record from it to any other object has log-odds <= -tau.
// Any record between two unions of objects of a valid cluster is a sum of positive records with
// class delta 0 (equal classes: ComputeClassDeltaLogprob, segment.cc:107-122), so it scores > bias
// whatever has merged before; any record from a union of its objects to the outside is a sum of
// negative records with class delta <= 0 and scores < bias, whatever the OUTSIDE objects have done
// meanwhile.  The queue pops in descending priority: the reference merges a valid cluster completely
// before it pops any of the cluster's outward records, and the objects outside never see the order
// in which that happened.  So a valid cluster is contracted in one step, whether or not the other
// clusters are valid -- exact with respect to the state the records are in (all fresh: the rounds
// call this right after a refresh, as the hand-over to the sequential finisher always did).
// tau = 2 N ulp(bias) / omf keeps the float32 quotient away from the bias (see fill_params).
//
// Sign-separable maps contract completely here (the rounds become components mode); on other maps
// the clusters become valid as the rounds resolve the ambiguous objects along the boundaries, and what
// used to be thousands of sequential finisher steps over fragments is one contraction.
#pragma once
#include "mn_device.h"
#include "mn_kernels_cc.h"

__device__ __forceinline__ int mn_oc_find(const int* __restrict__ up, int x) {
  int p = up[x];
  while (p != x) { x = p; p = up[x]; }
  return x;
}

// Live objects are the self-parented ids; the per-object passes run over all N ids (a few
// microseconds of trivial work), the per-record passes over the list.

// every live object starts as its own cluster
__global__ __launch_bounds__(256) void mn_oc_init(int N, const int* __restrict__ parent,
                                                  int* __restrict__ up, unsigned char* __restrict__ bad) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= N || parent[x] != x) return;
  up[x] = x;
  bad[x] = 0;
}

// Union along the positive records between objects of one class (larger root under smaller).
// `stride` > 1 takes every stride-th record only: a sample is enough to build the few huge clusters
// a map consists of with little contention; after a flatten the full pass finds equal roots at
// nearly every record and costs two reads.
__global__ __launch_bounds__(256) void mn_oc_link(ObjState S, RecList L, int R, i64 tau,
                                                  int* __restrict__ up, int stride) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * stride;
  if (i >= R) return;
  const u64 key = L.key[i];
  if (key == MN_EMPTY) return;
  const int u = mn_key_u(key), v = mn_key_v(key);
  int a = up[u], b = up[v];
  if (a == b) return;
  if (L.S[i] < tau || S.ocls[u] != S.ocls[v]) return;
  a = mn_oc_find(up, a); b = mn_oc_find(up, b);
  while (a != b) {
    if (a < b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&up[a], b);
    if (old == a) break;
    a = mn_oc_find(up, old);
    b = mn_oc_find(up, b);
  }
}

__global__ __launch_bounds__(256) void mn_oc_flatten(int N, const int* __restrict__ parent,
                                                     int* __restrict__ up) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= N || parent[x] != x) return;
  const int r = mn_oc_find(up, x);
  if (up[x] != r) up[x] = r;        // (a racing chase through x still ends at r)
}

// conditions (a) and (b) per record; a violation condemns the cluster(s) it touches
__global__ __launch_bounds__(256) void mn_oc_check(ObjState S, RecList L, int R, i64 tau,
                                                   const int* __restrict__ up,
                                                   unsigned char* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const u64 key = L.key[i];
  if (key == MN_EMPTY) return;
  const int u = mn_key_u(key), v = mn_key_v(key);
  const int ru = up[u], rv = up[v];
  const i64 s = L.S[i];
  if (ru == rv) {
    if (s < tau || S.ocls[u] != S.ocls[v]) bad[ru] = 1;
  } else if (s > -tau) {
    bad[ru] = 1; bad[rv] = 1;
  }
}

// `acc`: the [C][N] i64 planes of components mode (free once the cores stand)
__global__ __launch_bounds__(256) void mn_oc_clear(ImgParams P, const int* __restrict__ parent,
                                                   const int* __restrict__ up,
                                                   const unsigned char* __restrict__ bad,
                                                   i64* __restrict__ acc) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= P.N || parent[x] != x || up[x] != x || bad[x]) return;
  for (int c = 0; c < P.C; c++) acc[(size_t)c * P.N + x] = 0;
}

// Members of clusters that stand hand their class vectors (2^-24 fixed point: the sums do not
// depend on arrival order) and sizes to the root and point at it.  matched[] marks every object of
// a contracted cluster, which is what mn_rebuild re-keys and re-scores.  A root like the background
// takes a hundred thousand members, and one word takes only ~88 atomics/us: the block sums per
// root in LDS first (the table of mn_cc_sums) and flushes once; ids are pixel ids, so the members
// of a block mostly share a root.
#define MN_OC_THREADS 1024
__global__ __launch_bounds__(MN_OC_THREADS) void mn_oc_gather(ImgParams P, ObjState S,
                                                              const int* __restrict__ up,
                                                              const unsigned char* __restrict__ bad,
                                                              i64* __restrict__ acc,
                                                              unsigned char* __restrict__ matched,
                                                              Counters* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  u64* s_val = reinterpret_cast<u64*>(cc_smem);                   // [SLOTS][C+1], index C = size
  __shared__ int s_root[MN_CC_SUM_SLOTS];
  const int nval = MN_CC_SUM_SLOTS * (P.C + 1);
  for (int j = threadIdx.x; j < nval; j += MN_OC_THREADS) s_val[j] = 0;
  if (threadIdx.x < MN_CC_SUM_SLOTS) s_root[threadIdx.x] = -1;
  __syncthreads();
  const int x = blockIdx.x * MN_OC_THREADS + threadIdx.x;
  if (x < P.N && S.parent[x] == x) {
    const int r = up[x];
    if (r != x && !bad[r]) {
      const bool vx = S.lpvalid[x] != 0;
      const int slot = mn_lds_root_slot(s_root, r);
      for (int c = 0; c < P.C; c++)
        mn_cc_add(P, S, s_root, s_val, acc, r, c, slot,
                  __double2ll_rn((double)mn_obj_lp(P, S, vx, x, c) * MN_LP_FIX));
      mn_cc_add(P, S, s_root, s_val, acc, r, P.C, slot, (i64)S.osize[x]);
      S.parent[x] = r;
      matched[x] = 1;
      matched[r] = 1;
      cnt->any_selected = 1;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nval; j += MN_OC_THREADS) {
    const u64 v = s_val[j];
    if (v == 0) continue;
    const int slot = j / (P.C + 1), c = j - slot * (P.C + 1);
    mn_cc_add(P, S, s_root, s_val, acc, s_root[slot], c, -1, (i64)v);
  }
}

// roots that received members: own vector + accumulator -> summed class log-probs
__global__ __launch_bounds__(256) void mn_oc_finish(ImgParams P, ObjState S,
                                                    const i64* __restrict__ acc,
                                                    const unsigned char* __restrict__ matched) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= P.N || !matched[x] || S.parent[x] != x) return;
  const bool vx = S.lpvalid[x] != 0;
  for (int c = 0; c < P.C; c++) {
    const i64 own = __double2ll_rn((double)mn_obj_lp(P, S, vx, x, c) * MN_LP_FIX);
    S.lpsum[(size_t)c * P.N + x] = (float)((double)(acc[(size_t)c * P.N + x] + own) * (1.0 / MN_LP_FIX));
  }
  S.lpvalid[x] = 1;
}
