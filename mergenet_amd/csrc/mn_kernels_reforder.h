// mn_kernels_reforder.h -- device side of mn_reforder.h (mn_options.tie_order = MN_TIES_REFERENCE): the
// reference's merge loop with its own order among bit-equal priorities, on the state the exact engine's set-up
// kernels leave (class vectors, arg-max classes, per-record log-odds: bit-identical to the reference's).
#pragma once
#include "mn_reforder.h"

// per pixel: the object as the reference constructs it (segment.cc:5-21, 197-206), an empty map
__global__ __launch_bounds__(256) void mn_ro_prepare_objects(ImgParams P, XState X, RoState S) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  S.osize[p] = 1;
  S.ocls[p] = X.obj[p].cls;
  S.bcount[p] = 1; S.nelem[p] = 0; S.head[p] = MN_RO_NULL; S.single[p] = MN_RO_NULL; S.boff[p] = 0;
  if (p == 0) {
    for (int i = 0; i < 16; i++) S.ctl[i] = 0;
  }
}

// per record slot (pixel * O + k): ends, log-odds, initial priority (segment.cc:24-46, 107-150)
__global__ __launch_bounds__(256) void mn_ro_prepare_records(ImgParams P, XState X, RoState S) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (size_t)S.NL) return;
  const XRec R = X.rec[r];
  if (R.key == MN_EMPTY) { S.r1[r] = -1; S.r2[r] = -1; S.oml[r] = 0.0f; S.prio[r] = -1.0f; return; }
  S.r1[r] = mn_key_u(R.key); S.r2[r] = mn_key_v(R.key);
  S.oml[r] = R.S;
  int mc;
  S.prio[r] = mn_ro_score(S, (int)r, &mc);
}

// The maps of the constructor's loop (segment.cc:209-231), one lane per OBJECT: a map depends only on the order
// of ITS OWN inserts, which is the creation order of the records that touch the pixel -- as source pixel
// (slot pixel * O + k, node 2r) and as target of the pixel at -offset_k (node 2r + 1).  Bucket arrays come from
// the shared arena by an atomic bump (where they land does not matter).  The pushes stay sequential (mn_ro_loop).
__global__ __launch_bounds__(64) void mn_ro_build_maps(ImgParams P, RoState S, int* __restrict__ failed) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= P.N) return;
  const int row = x / P.W, col = x - row * P.W;
  long long ids[2 * MN_MAX_OFFSETS];                     // node ids (2r + side) in creation order of the records
  int n = 0;
  for (int k = 0; k < P.O; k++) {
    // as target: the record of source pixel (row - di, col - dj) and offset k
    const int sr = row - P.di[k], sc = col - P.dj[k];
    if (sr >= 0 && sr < P.H && sc >= 0 && sc < P.W) ids[n++] = 2 * ((long long)(sr * P.W + sc) * P.O + k) + 1;
    // as source
    const int tr = row + P.di[k], tc = col + P.dj[k];
    if (tr >= 0 && tr < P.H && tc >= 0 && tc < P.W) ids[n++] = 2 * ((long long)x * P.O + k);
  }
  for (int i = 1; i < n; i++) {                          // ascending record id
    const long long v = ids[i];
    int j = i - 1;
    while (j >= 0 && ids[j] > v) { ids[j + 1] = ids[j]; j--; }
    ids[j + 1] = v;
  }
  for (int i = 0; i < n; i++) {
    const long long r = ids[i] >> 1;
    if (S.r1[r] < 0) continue;
    if (!mn_ro_insert(S, x, (int)ids[i], mn_ro_key(S.r1[r], S.r2[r]))) { atomicAdd(failed, 1); return; }
  }
}

// ---- the queue, worked by the whole wave -------------------------------------------------------------------
// The maps are a chain of dependent accesses and stay with lane 0; the binary heap is not: the ancestors of a
// slot are known in advance (one round trip fetches all of them), and going down, the 62 descendants of the
// hole over five levels are fetched together and the path through them is settled by cross-lane reads.  The
// MOVES are those of std::push_heap / std::pop_heap (mn_ro_push / mn_ro_pop are the scalar text the host
// checks against the library; the GPU tests check this form against the same vectors).
// value of lane `l` for a WAVE-UNIFORM l: a lane read on the scalar unit (v_readlane), not a trip through the LDS
// crossbar (ds_bpermute, ~100 cycles) -- the descent of a pop is a chain of 2 x 26 of them
__device__ __forceinline__ int mn_ro_lane(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ float mn_ro_lane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(l)));
}

__device__ __forceinline__ void mn_ro_wave_pushup(const RoState& S, long long hole, float pr, int rec, int lane) {
  // lane l: the ancestor at distance l + 1 of the hole (1-based index (hole + 1) >> (l + 1))
  const long long j = lane < 48 ? ((hole + 1) >> (lane + 1)) : 0;
  const bool valid = j >= 1;
  unsigned long long ae = 0ull;
  if (valid) ae = S.heap[j - 1];
  const unsigned long long movers = __ballot(valid && mn_ro_entry_prio(ae) < pr);      // __push_heap: while (parent < value)
  const int k = (~movers == 0ull) ? 64 : (__ffsll((long long)~movers) - 1);   // the first ancestor that stays
  if (lane < k) {
    const long long dest = ((hole + 1) >> lane) - 1;
    S.heap[dest] = ae;
  }
  if (lane == 0) {
    const long long dest = ((hole + 1) >> k) - 1;
    S.heap[dest] = mn_ro_entry(pr, rec);
  }
}

// pop: returns the top entry and, fetched while the hole travels down, the popped record's stored priority and
// second end (what the caller's stale test reads)
__device__ __forceinline__ void mn_ro_wave_pop(const RoState& S, long long& n, float* pr_out, int* rec_out,
                                               float* rec_prio, int* rec_r2, int lane) {
  // ONE round trip: the top (lane 62), the last entry (lane 63) and the 62 descendants of the root over five levels
  const long long len = n - 1;
  const int d = 31 - __clz(lane + 2);                    // lane l < 62: level d = floor(log2(l + 2)) (1..5) ...
  const int jl = (lane + 2) - (1 << d);                  // ... position jl in it
  float p = 0.0f; int rr = 0;
  {
    const long long idx0 = lane < 62 ? (long long)lane + 1 : (lane == 62 ? 0 : len);
    if (idx0 < n) { const unsigned long long e = S.heap[idx0]; p = mn_ro_entry_prio(e); rr = mn_ro_entry_rec(e); }
  }
  *pr_out = mn_ro_lane(p, 62); *rec_out = mn_ro_lane(rr, 62);
  const float vp = mn_ro_lane(p, 63);
  const int vr = mn_ro_lane(rr, 63);
  *rec_prio = S.prio[*rec_out];                          // (uniform loads, in flight while the hole goes down)
  *rec_r2 = S.r2[*rec_out];
  n = len;
  if (len == 0) return;
  const long long half = (len - 1) / 2;
  long long hole = 0;
  bool fetched = true;
  while (hole < half) {                                  // __adjust_heap: to the bottom along the larger children
    if (!fetched) {
      const long long idx = (((hole + 1) << d) + jl) - 1;
      const bool valid = lane < 62 && idx < len;
      p = 0.0f; rr = 0;
      if (valid) { const unsigned long long e = S.heap[idx]; p = mn_ro_entry_prio(e); rr = mn_ro_entry_rec(e); }
    }
    fetched = false;
    long long cur = hole;
    int curj = 0;
#pragma unroll
    for (int lev = 1; lev <= 5; lev++) {
      if (cur < half) {                                  // (uniform: cur is the same in every lane)
        const int lr = (1 << lev) - 2 + 2 * curj + 1, ll = lr - 1;
        const float pR = mn_ro_lane(p, lr), pL = mn_ro_lane(p, ll);
        const int chosen = __builtin_amdgcn_readfirstlane((pR < pL) ? ll : lr);          // the right child among equals
        if (lane == chosen) S.heap[cur] = mn_ro_entry(p, rr);
        curj = chosen - ((1 << lev) - 2);
        cur = (((hole + 1) << lev) + curj) - 1;
      }
    }
    hole = cur;
  }
  if ((len & 1) == 0 && hole == (len - 2) / 2) {         // a last node with a left child only
    const long long child = 2 * hole + 1;
    if (lane == 0) S.heap[hole] = S.heap[child];
    hole = child;
  }
  mn_ro_wave_pushup(S, hole, vp, vr, lane);              // ... and back up with the last entry's value
}

// ONE wave per image (block b runs image b of a batch: images are independent): lane 0 runs the maps and the
// records (mn_reforder.h), the wave the queue.  Comes back when `budget` pops (4 x budget records of the
// constructor's loop) are used up: the state is in memory, the next launch goes on; a block whose image has
// finished -- or waits for a larger workspace -- returns at once.
__global__ __launch_bounds__(64) void mn_ro_loop(const RoState* __restrict__ Ss, int O, long long budget) {
  __shared__ int s_nodes[66];                            // the records of a walk (64), their count, the next node
  RoState S = Ss[blockIdx.x];
  const int lane = threadIdx.x;
  const long long st0 = S.ctl[0];
  if (st0 != MN_RO_RUNNING && st0 != MN_RO_BUDGET) return;
  long long n = S.ctl[1], biggest = S.ctl[6], pops = 0;
  int status = MN_RO_RUNNING;
#ifdef MN_RO_STAMPS
#define MN_RO_STAMP(acc) { const long long t_ = wall_clock64(); acc += t_ - t_mark; t_mark = t_; }
#else
#define MN_RO_STAMP(acc)
#endif
  long long t_init = 0, t_pop = 0, t_merge = 0, t_mark = 0;    // (-DMN_RO_STAMPS: 100 MHz ticks per phase, MN_TRACE_EXACT prints them)
#ifdef MN_RO_STAMPS
  t_mark = wall_clock64();
#endif
  (void)t_mark;
  // ---- the constructor's loop (segment.cc:209-231): the maps are built (mn_ro_build_maps); the pushes, in
  //      creation order, 64 records per look ----
  long long r = S.ctl[5];
  if (r < S.NL) {
    long long left = budget * 4;
    while (r < S.NL && status == MN_RO_RUNNING) {
      if (left <= 0) { status = MN_RO_BUDGET; break; }
      left -= 64;
      const long long mine = r + lane;
      const float pr = mine < S.NL ? S.prio[mine] : -1.0f;
      unsigned long long want = __ballot(mine < S.NL && S.r1[mine < S.NL ? mine : 0] >= 0 && pr >= 0.0f);
      while (want) {
        const int l = __ffsll((long long)want) - 1;
        want &= want - 1ull;
        if (n >= S.hcap) { status = MN_RO_HEAP_FULL; break; }
        mn_ro_wave_pushup(S, n, mn_ro_lane(pr, l), (int)(r + l), lane);
        n++;
        biggest = n > biggest ? n : biggest;
      }
      if (status != MN_RO_RUNNING) break;                 // (the cursor stays: the run is repeated from scratch anyway)
      r += 64;
    }
    if (r > S.NL) r = S.NL;
    if (lane == 0) S.ctl[5] = r;
    MN_RO_STAMP(t_init)
  }
  // ---- RunSegmentation (segment.cc:539-573) ----
  if (status == MN_RO_RUNNING && r >= S.NL) {
    status = MN_RO_DONE;
    while (n > 0) {
      if (pops >= budget) { status = MN_RO_BUDGET; break; }
      float q, stored; int rec, second;
      mn_ro_wave_pop(S, n, &q, &rec, &stored, &second, lane);
      pops++;
      MN_RO_STAMP(t_pop)
      if (q != stored) continue;                          // a stale entry (segment.cc:554)
      if (second < 0) continue;                           // a merged record (segment.cc:557)
      float f = 0.0f; int mc = 0;
      if (lane == 0) { f = mn_ro_score(S, rec, &mc); S.prio[rec] = f; }
      f = mn_ro_lane(f, 0);
      if (f == q) {
        int a = 0, b = 0, it = MN_RO_NULL, rc = MN_RO_RUNNING;
        if (lane == 0) rc = mn_ro_merge_begin(S, rec, mc, &a, &b, &it);
        rc = mn_ro_lane(rc, 0); it = mn_ro_lane(it, 0); a = mn_ro_lane(a, 0); b = mn_ro_lane(b, 0);
        // The walk over the absorbed object's records, up to 64 at a time: lane 0 follows the list (the
        // iteration order of the reference's unordered_map), then every lane takes ONE record -- out of the
        // third object's map, look-up in the survivor's, fold or re-key, into the third object's map, fresh
        // priority: no two records of a walk share a third object -- then, IN LIST ORDER, lane 0 puts the
        // adopted ones into the survivor's map and the wave pushes the fresh priorities.
        while (rc == MN_RO_RUNNING && it != MN_RO_NULL) {
          if (lane == 0) {
            int x = it, cnt = 0;
            while (x != MN_RO_NULL && cnt < 64) { s_nodes[cnt++] = x; x = S.nnext[x]; }
            s_nodes[64] = cnt; s_nodes[65] = x;
          }
          __syncthreads();
          const int cnt = s_nodes[64];
          it = s_nodes[65];
          const int node = lane < cnt ? s_nodes[lane] : MN_RO_NULL;
          __syncthreads();
          int nx, prec = -1, myrc = MN_RO_RUNNING;
          float pp = 0.0f;
          unsigned long long akey = 0ull;
          if (node != MN_RO_NULL) myrc = mn_ro_merge_node(S, a, b, node, &nx, &pp, &prec, 1, &akey);
          const unsigned long long failed = __ballot(myrc != MN_RO_RUNNING);
          if (failed) { rc = mn_ro_lane(myrc, __ffsll((long long)failed) - 1); break; }
          for (int i = 0; i < cnt && rc == MN_RO_RUNNING; i++) {          // (uniform)
            const unsigned klo = (unsigned)mn_ro_lane((int)(akey & 0xFFFFFFFFull), i), khi = (unsigned)mn_ro_lane((int)(akey >> 32), i);
            const int nd = mn_ro_lane(node, i);
            const unsigned long long k = ((unsigned long long)khi << 32) | klo;
            if (k != 0ull) {
              int ok = 1;
              if (lane == 0) ok = mn_ro_insert(S, a, nd, k) ? 1 : 0;
              if (!mn_ro_lane(ok, 0)) rc = MN_RO_ARENA_FULL;
            }
          }
          for (int i = 0; i < cnt && rc == MN_RO_RUNNING; i++) {
            const int pr_rec = mn_ro_lane(prec, i);
            const float pr_val = mn_ro_lane(pp, i);
            if (pr_rec >= 0) {
              if (n >= S.hcap) { rc = MN_RO_HEAP_FULL; break; }
              mn_ro_wave_pushup(S, n, pr_val, pr_rec, lane);
              n++;
              biggest = n > biggest ? n : biggest;
            }
          }
        }
        if (rc != MN_RO_RUNNING) { status = rc; break; }
        if (lane == 0) mn_ro_merge_end(S, rec, a, b);
        MN_RO_STAMP(t_merge)
      } else if (f >= 0.0f) {
        if (n >= S.hcap) { status = MN_RO_HEAP_FULL; break; }
        mn_ro_wave_pushup(S, n, f, rec, lane);
        n++;
        biggest = n > biggest ? n : biggest;
      }
    }
  }
  if (lane == 0) {
    S.ctl[0] = status; S.ctl[1] = n; S.ctl[3] += pops; S.ctl[6] = biggest;
    S.ctl[8] += t_init; S.ctl[9] += t_pop; S.ctl[10] += t_merge;
  }
}

// hand-over to the exact engine's export and checks: object state, live records
__global__ __launch_bounds__(256) void mn_ro_finish(ImgParams P, XState X, RoState S) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < (size_t)P.N) {
    XObj o = X.obj[gid];
    o.size = S.osize[gid]; o.cls = S.ocls[gid];
    X.obj[gid] = o;
  }
  if (gid < (size_t)S.NL) {
    XRec R = X.rec[gid];
    const bool live = S.r1[gid] >= 0 && S.r2[gid] >= 0 && S.prio[gid] != 1.17549435e-38f;
    R.key = live ? mn_key(S.r1[gid], S.r2[gid]) : MN_EMPTY;
    R.S = S.oml[gid];
    X.rec[gid] = R;
  }
}
