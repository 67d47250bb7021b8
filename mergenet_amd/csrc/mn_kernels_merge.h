// mn_kernels_merge.h -- phase B, parallel rounds of the lazy-greedy merge.
//
// Reference work replaced: RunSegmentation + Merge (utils/csegment/segment.cc:539-727).  The
// reference pops one record at a time from a priority queue; here every round
//   1. has every live record scored (fresh) next to its remembered (stored) priority and the best
//      visible record of every object found by a 64-bit atomicMax of (priority, gain>0, ~partner)
//      -- fused into the compaction kernel of the previous round (mn_compact),
//   2. computes the band threshold of the round (mn_band_threshold): only records whose gain is
//      within a factor of the round's best gain may merge, which keeps the global descending
//      order of the reference's queue to within that factor,
//   3. pairs objects on the best-record forest (mn_obj_match_mutual / propose / accept): a
//      matching, so the merges of one round never share an object,
//   4. merges the pairs (mn_rec_apply) and folds the records of absorbed objects into the
//      survivors' records by re-inserting every record under its relabelled (min,max) key into an
//      open-addressing table (mn_rebuild), log-odds summed in 2^-30 fixed point so that the sums
//      do not depend on arrival order.
// Laziness is kept: a record remembers the priority it was last scored at (AdjacencyRecord::
// merge_priority); only records incident to an absorbed object are re-scored by a merge
// (segment.cc:650-707), a survivor's other records stay stale until selected ("popped",
// segment.cc:554-565): stale-high ones are lowered eagerly, stale-low ones keep competing with
// their stored value and are refreshed instead of merged when selected.
#pragma once

#include "mn_device.h"

struct RecList {
  u64* key;      // (u << 32) | v, u < v ; MN_EMPTY = dead
  i64* S;        // summed log-odds, 2^-30 fixed point   (AdjacencyRecord::obj_merge_logprob)
  float* st;     // stored priority                      (AdjacencyRecord::merge_priority)
  float* fr;     // priority under the objects' current state (scored when the list is built)
  unsigned char* aux;   // merged class (7 bits) | likelihood gain > 0 (bit 7)
};

struct HashTab {
  u64* key;
  i64* S;
  float* st;
  unsigned char* touched;
  unsigned mask;   // capacity - 1 (capacity is a power of two)
};

struct Counters {
  int n_records;     // appended by the compaction kernel (one atomic per 1024 slots)
  int any_selected;  // set when a round selected at least one record (plain store, no atomic:
                     // a single hot counter word caps out near 88 atomics/us on this chip)
  int n_merged;      // merges done by the sequential finisher
  int pad0;
  int finisher_steps;
  int finisher_merges;
  int error;
  int merged_E;          // pixel edges of the records the LDS finisher merged (fast certificate)
  long long merged_S;    // and their summed log-odds, fixed point
};

// ---- several memsets in one launch --------------------------------------------------------------
// A dozen hipMemsetAsync calls per image cost ~6 us each plus the gaps between them, which is a
// tenth of an image in components mode: the regions are filled by one kernel instead.
#define MN_FILL_JOBS 16
struct FillJobs {
  void* ptr[MN_FILL_JOBS];
  unsigned long long bytes[MN_FILL_JOBS];
  unsigned pattern[MN_FILL_JOBS];            // the byte value replicated four times
  int count;
};

__global__ __launch_bounds__(256) void mn_fill_many(FillJobs J) {
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t gsize = (size_t)gridDim.x * blockDim.x;
  for (int j = 0; j < J.count; j++) {
    unsigned char* base = static_cast<unsigned char*>(J.ptr[j]);
    const size_t n = J.bytes[j];
    size_t head = (16 - (reinterpret_cast<size_t>(base) & 15)) & 15;
    if (head > n) head = n;
    const size_t body = (n - head) >> 4, tail = n - head - (body << 4);
    const unsigned v = J.pattern[j];
    uint4* dst = reinterpret_cast<uint4*>(base + head);
    for (size_t i = gtid; i < body; i += gsize) dst[i] = make_uint4(v, v, v, v);
    if (gtid < head) base[gtid] = (unsigned char)v;
    if (gtid < tail) base[head + (body << 4) + gtid] = (unsigned char)v;
  }
}

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
// progress[s] is raised when sub-round s pairs anything; a sub-round that finds progress[s-1] == 0
// returns at once (nothing changed, so it would reproduce the previous, empty, outcome).
__global__ __launch_bounds__(256) void mn_pix_match(int N, const u64* __restrict__ best,
                                                    unsigned char* __restrict__ matched,
                                                    int* __restrict__ mate,
                                                    int* __restrict__ progress, int s,
                                                    Counters* __restrict__ cnt) {
  if (s > 0 && !progress[s - 1]) return;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const u64 b = best[p];
  if (b == 0) return;
  const int q = mn_pack_partner(b);
  if ((unsigned)q >= (unsigned)N) { cnt->error = MN_ERR_INTERNAL; return; }   // stale slot: counted, not followed
  const u64 bq = best[q];
  if (bq == 0 || mn_pack_partner(bq) != p) return;
  matched[p] = 1;
  mate[p] = q;
  progress[s] = 1;
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

// Table -> compact list, fused with the scoring of the NEXT round: every record gets its fresh
// priority (touched records take it as their stored priority, the others keep theirs unless it is
// stale-high), the packed key goes into both endpoints' best slot, the block maximum feeds the
// band threshold.  A block owns 1024 consecutive slots and reserves its output range with ONE
// atomic.
// PER = slots per lane: 4 for the big tables; a table of a few thousand slots is scanned one slot per
// lane (a lane's slots are scored one after the other, each a chain of dependent loads: with 4 per
// lane a small table took 45 us of pure latency).
#define MN_COMPACT_SLOTS 1024
template <int PER>
__global__ __launch_bounds__(256) void mn_compact(ImgParams P, ObjState S, HashTab T, RecList L,
                                                  u64* __restrict__ ball,
                                                  unsigned* __restrict__ gmax,
                                                  Counters* __restrict__ cnt,
                                                  const int* __restrict__ tcount = nullptr,
                                                  int* __restrict__ lcount = nullptr) {
  __shared__ int sh_w[PER][4];
  __shared__ int sh_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned base = blockIdx.x * (256 * PER);
  u64 key[PER];
  int before[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const unsigned slot = base + j * 256 + threadIdx.x;
    key[j] = slot <= T.mask ? T.key[slot] : MN_EMPTY;
    const u64 m = __ballot(key[j] != MN_EMPTY);
    before[j] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sh_w[j][wave] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int j = 0; j < PER; j++)
      for (int w = 0; w < 4; w++) { const int t = sh_w[j][w]; sh_w[j][w] = tot; tot += t; }
    sh_base = tot ? atomicAdd(&cnt->n_records, tot) : 0;
  }
  __syncthreads();
  unsigned mybits = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    if (key[j] == MN_EMPTY) continue;
    const unsigned slot = base + j * 256 + threadIdx.x;
    const i64 s = T.S[slot];
    const int u = mn_key_u(key[j]), v = mn_key_v(key[j]);
    int mc;
    bool pos;
    const float f = mn_score(P, S, u, v, mn_fixed_to_float(s), &mc, &pos);
    float st = T.touched[slot] ? f : T.st[slot];
    if (st >= 0.0f && f < st) st = f;                 // stale-high: lowered right away
    const int idx = sh_base + sh_w[j][wave] + before[j];
    L.key[idx] = key[j];
    L.S[idx] = s;
    L.st[idx] = st;
    L.fr[idx] = f;
    L.aux[idx] = (unsigned char)((mc & 0x7F) | (pos ? 0x80 : 0));
    if (lcount) lcount[idx] = tcount[slot];           // pixel edges per record (components mode)
    if (ball && st >= 0.0f) {                         // (no best-record slots: the finisher takes the list)
      // a plain look first: most records lose against what is already there (an object has tens
      // of records, a running maximum changes ~ln(n) times), and a lost race only costs the atomic
      const u64 ku = mn_pack(st, v, pos), kv = mn_pack(st, u, pos);
      if (ku > ball[u]) atomicMax(&ball[u], ku);
      if (kv > ball[v]) atomicMax(&ball[v], kv);
      mybits = max(mybits, (st == 0.0f) ? 0u : __float_as_uint(st));
    }
  }
  // highest visible priority of the round (for the band threshold), spread over 64 words
  if (!ball) return;
  for (int off = 32; off > 0; off >>= 1) mybits = max(mybits, (unsigned)__shfl_xor((int)mybits, off));
  if (lane == 0 && mybits) atomicMax(&gmax[(blockIdx.x * 4 + wave) & 63], mybits);
}

// ---- rounds on the explicit record list -------------------------------------------------------

// Band threshold of the round.  The reference pops records in globally descending priority; a
// round may only merge records whose likelihood gain is within a factor `gamma` of the round's
// best gain, so weak (bias-driven) merges of big objects wait until the strong ones are done
// everywhere, as they do in the queue.  csegment: gain/den = priority - bias; pysegmenter carries
// the bias inside the fraction, so the band is taken on the priority itself.
__global__ void mn_band_threshold(const unsigned* __restrict__ gmax, float bias, int variant,
                                  float gamma, float* __restrict__ theta) {
  unsigned m = gmax[threadIdx.x & 63];
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
  if (threadIdx.x == 0) {
    const float pmax = __uint_as_float(m);
    float t;
    if (variant == MN_VARIANT_CSEGMENT) t = pmax > bias ? bias + gamma * (pmax - bias) : 0.0f;
    else t = gamma * pmax;
    *theta = t;
  }
}

// ---- matching on the best-record forest (object-level passes) ----------------------------------
// ball[u] names u's best visible record.  Sub-round 0 pairs the objects that name each other.
// Later sub-rounds work on the forest "u -> partner(ball[u])" only: an object whose own choice
// has just been taken by somebody else ("jilted") accepts the best still-free object that chose
// it, provided that record has positive likelihood gain.  Every merged record is therefore the
// best record of at least one of its endpoints, and no object is in two pairs.
__device__ __forceinline__ bool mn_in_band(u64 packed, float theta) {
  return __uint_as_float((unsigned)(packed >> 32)) >= theta;
}

__global__ __launch_bounds__(256) void mn_obj_match_mutual(int N, const u64* __restrict__ ball,
                                                           const float* __restrict__ theta,
                                                           unsigned char* __restrict__ matched,
                                                           int* __restrict__ mate,
                                                           int* __restrict__ progress,
                                                           Counters* __restrict__ cnt) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N) return;
  const u64 b = ball[u];
  if (b == 0 || !mn_in_band(b, *theta)) return;
  const int t = mn_pack_partner(b);
  // a partner id decoded from a slot nobody wrote this round would be a wild read (the round-1
  // fault): counted as an internal error instead of followed
  if ((unsigned)t >= (unsigned)N) { cnt->error = MN_ERR_INTERNAL; return; }
  const u64 bt = ball[t];
  if (bt == 0 || mn_pack_partner(bt) != u) return;
  matched[u] = 1;
  mate[u] = t;
  progress[0] = 1;
}

__global__ __launch_bounds__(256) void mn_obj_propose(int N, const u64* __restrict__ ball,
                                                      const float* __restrict__ theta,
                                                      const unsigned char* __restrict__ matched,
                                                      u64* __restrict__ inbest,
                                                      const int* __restrict__ progress, int s,
                                                      Counters* __restrict__ cnt) {
  if (!progress[s - 1]) return;        // the previous sub-round paired nothing: nothing can change
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= N || matched[u]) return;
  const u64 b = ball[u];
  if (b == 0 || !mn_pack_gain_pos(b) || !mn_in_band(b, *theta)) return;
  const int t = mn_pack_partner(b);
  if ((unsigned)t >= (unsigned)N) { cnt->error = MN_ERR_INTERNAL; return; }
  if (matched[t]) return;
  const u64 bt = ball[t];
  if (bt == 0) return;
  const int tt = mn_pack_partner(bt);
  if ((unsigned)tt >= (unsigned)N) { cnt->error = MN_ERR_INTERNAL; return; }
  if (!matched[tt]) return;            // t still waits for its own choice
  atomicMax(&inbest[t], (b & 0xFFFFFFFF00000000ull) | (u64)(0x7FFFFFFFu - (unsigned)u));
}

__global__ __launch_bounds__(256) void mn_obj_accept(int N, u64* __restrict__ inbest,
                                                     unsigned char* __restrict__ matched,
                                                     int* __restrict__ mate,
                                                     int* __restrict__ progress, int s,
                                                     Counters* __restrict__ cnt) {
  if (!progress[s - 1]) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const u64 k = inbest[t];
  if (k == 0) return;
  inbest[t] = 0;                       // self-cleaning: no memset between sub-rounds
  if (matched[t]) return;
  const int u = mn_pack_partner(k);
  if ((unsigned)u >= (unsigned)N) { cnt->error = MN_ERR_INTERNAL; return; }
  matched[t] = 1; mate[t] = u;
  matched[u] = 1; mate[u] = t;
  progress[s] = 1;
}

// Hand-over to the sequential finisher: every record gets its current priority.  The parallel
// rounds grow objects in a different order than the reference's queue, so WHICH records are
// left stale (a survivor's untouched records, segment.cc:650-707) is an artefact of the rounds;
// the sequential phase starts from the state "all records fresh", as the exact mode does.
__global__ __launch_bounds__(256) void mn_rec_refresh(ImgParams P, ObjState S, RecList L, int R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const u64 key = L.key[i];
  if (key == MN_EMPTY) return;
  int mc;
  bool pos;
  L.st[i] = mn_score(P, S, mn_key_u(key), mn_key_v(key), mn_fixed_to_float(L.S[i]), &mc, &pos);
}

// Selected records: refresh (stale-low) or merge (segment.cc:560-565, 602-642).
// `touch` (64 words): how many records have a matched object at either end -- the only ones the
// rebuild re-inserts, i.e. what the next table has to hold (one atomic per block, spread over the words)
__global__ __launch_bounds__(256) void mn_rec_apply(ImgParams P, ObjState S, RecList L, int R,
                                                    const int* __restrict__ mate,
                                                    Counters* __restrict__ cnt,
                                                    unsigned* __restrict__ touch) {
  __shared__ int sh_n;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) sh_n = 0;
  __syncthreads();
  const u64 key0 = i < R ? L.key[i] : MN_EMPTY;
  int mu = -1, mv = -1;
  if (key0 != MN_EMPTY) { mu = mate[mn_key_u(key0)]; mv = mate[mn_key_v(key0)]; }
  const u64 nz = __ballot(mu >= 0 || mv >= 0);
  if ((threadIdx.x & 63) == 0 && nz) atomicAdd(&sh_n, __popcll(nz));
  __syncthreads();
  if (threadIdx.x == 0 && sh_n) atomicAdd(&touch[blockIdx.x & 63], (unsigned)sh_n);
  if (key0 == MN_EMPTY || mu != mn_key_v(key0)) return;   // not the record of a matched pair
  cnt->any_selected = 1;
  const float f = L.fr[i], st = L.st[i];
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
  S.ocls[a] = (unsigned char)(L.aux[i] & 0x7F);
  S.osize[a] = na + nb;
  S.parent[b] = a;
}

// Records of the finished round -> table / next list.  A record whose endpoints both sat the round
// out ("idle": neither matched) cannot change -- same key, same log-odds, same object state, hence
// the same fresh priority, and nothing can fold into it, because a fold needs a survivor at one
// end -- so it goes straight to the next list (block-aggregated append) together with its entry
// for the next round's best-record slots; only the others are re-inserted under their relabelled
// key (records inside one object disappear there).
#define MN_REBUILD_ITEMS 1024
__global__ __launch_bounds__(256) void mn_rebuild(ObjState S, RecList L, int R,
                                                  const unsigned char* __restrict__ matched,
                                                  HashTab T, RecList Out, u64* __restrict__ ball,
                                                  unsigned* __restrict__ gmax,
                                                  Counters* __restrict__ cnt, int fresh_all) {
  __shared__ int sh_w[4][4];
  __shared__ int sh_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = blockIdx.x * MN_REBUILD_ITEMS;
  u64 key[4];
  int before[4];
  bool idle[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = base + j * 256 + threadIdx.x;
    key[j] = i < R ? L.key[i] : MN_EMPTY;
    idle[j] = false;
    if (key[j] != MN_EMPTY) {
      const int u = mn_key_u(key[j]), v = mn_key_v(key[j]);
      idle[j] = !matched[u] && !matched[v];
      if (!idle[j]) {
        const int nu = S.parent[u], nv = S.parent[v];
        if (nu != nv) {
          const unsigned slot = mn_tab_insert(T, mn_key(nu, nv), L.S[i]);
          // (fresh_all: after a cluster contraction every record of a contracted object is scored
          //  anew, as components mode scores the records between its components)
          if (nu != u || nv != v || fresh_all) T.touched[slot] = 1;   // incident to an absorbed object: re-score
          else T.st[slot] = L.st[i];                     // keeps its stored priority
        }
      }
    }
    const u64 m = __ballot(idle[j]);
    before[j] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sh_w[j][wave] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int j = 0; j < 4; j++)
      for (int w = 0; w < 4; w++) { const int t = sh_w[j][w]; sh_w[j][w] = tot; tot += t; }
    sh_base = tot ? atomicAdd(&cnt->n_records, tot) : 0;
  }
  __syncthreads();
  unsigned mybits = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (!idle[j]) continue;
    const int i = base + j * 256 + threadIdx.x;
    const int idx = sh_base + sh_w[j][wave] + before[j];
    const float st = L.st[i];
    const unsigned char ax = L.aux[i];
    Out.key[idx] = key[j];
    Out.S[idx] = L.S[i];
    Out.st[idx] = st;
    Out.fr[idx] = L.fr[i];
    Out.aux[idx] = ax;
    if (st >= 0.0f) {
      const int u = mn_key_u(key[j]), v = mn_key_v(key[j]);
      const bool pos = (ax & 0x80) != 0;
      const u64 ku = mn_pack(st, v, pos), kv = mn_pack(st, u, pos);
      if (ku > ball[u]) atomicMax(&ball[u], ku);
      if (kv > ball[v]) atomicMax(&ball[v], kv);
      mybits = max(mybits, (st == 0.0f) ? 0u : __float_as_uint(st));
    }
  }
  for (int off = 32; off > 0; off >>= 1) mybits = max(mybits, (unsigned)__shfl_xor((int)mybits, off));
  if (lane == 0 && mybits) atomicMax(&gmax[(blockIdx.x * 4 + wave) & 63], mybits);
}
