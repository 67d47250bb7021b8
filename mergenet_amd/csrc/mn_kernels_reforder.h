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

// ---- the queue, worked by the whole wave -------------------------------------------------------------------
// The maps are a chain of dependent accesses and stay with lane 0; the binary heap is not: the ancestors of a
// slot are known in advance (one round trip fetches all of them), and going down, the 62 descendants of the
// hole over five levels are fetched together and the path through them is settled by cross-lane reads.  The
// MOVES are those of std::push_heap / std::pop_heap (mn_ro_push / mn_ro_pop are the scalar text the host
// checks against the library; the GPU tests check this form against the same vectors).
__device__ __forceinline__ void mn_ro_wave_pushup(const RoState& S, long long hole, float pr, int rec, int lane) {
  // lane l: the ancestor at distance l + 1 of the hole (1-based index (hole + 1) >> (l + 1))
  const long long j = lane < 48 ? ((hole + 1) >> (lane + 1)) : 0;
  const bool valid = j >= 1;
  float ap = 0.0f;
  int ar = 0;
  if (valid) { ap = S.hprio[j - 1]; ar = S.hrec[j - 1]; }
  const unsigned long long movers = __ballot(valid && ap < pr);      // __push_heap: while (parent < value)
  const int k = (~movers == 0ull) ? 64 : (__ffsll((long long)~movers) - 1);   // the first ancestor that stays
  if (lane < k) {
    const long long dest = ((hole + 1) >> lane) - 1;
    S.hprio[dest] = ap; S.hrec[dest] = ar;
  }
  if (lane == 0) {
    const long long dest = ((hole + 1) >> k) - 1;
    S.hprio[dest] = pr; S.hrec[dest] = rec;
  }
}

__device__ __forceinline__ void mn_ro_wave_pop(const RoState& S, long long& n, float* pr_out, int* rec_out, int lane) {
  // top and last entry in one round trip
  float tp = 0.0f; int tr = 0;
  if (lane == 0) { tp = S.hprio[0]; tr = S.hrec[0]; }
  if (lane == 1) { tp = S.hprio[n - 1]; tr = S.hrec[n - 1]; }
  *pr_out = __shfl(tp, 0); *rec_out = __shfl(tr, 0);
  const float vp = __shfl(tp, 1);
  const int vr = __shfl(tr, 1);
  const long long len = n - 1;
  n = len;
  if (len == 0) return;
  const long long half = (len - 1) / 2;
  long long hole = 0;
  // lane l < 62: descendant at level d = floor(log2(l + 2)) (1..5), position j in its level
  const int d = 31 - __clz(lane + 2);
  const int jl = (lane + 2) - (1 << d);
  while (hole < half) {                                  // __adjust_heap: to the bottom along the larger children
    const long long idx = (((hole + 1) << d) + jl) - 1;
    const bool valid = lane < 62 && idx < len;
    float p = 0.0f; int rr = 0;
    if (valid) { p = S.hprio[idx]; rr = S.hrec[idx]; }
    long long cur = hole;
    int curj = 0;
#pragma unroll
    for (int lev = 1; lev <= 5; lev++) {
      if (cur < half) {                                  // (uniform: cur is the same in every lane)
        const int lr = (1 << lev) - 2 + 2 * curj + 1, ll = lr - 1;
        const float pR = __shfl(p, lr), pL = __shfl(p, ll);
        const int chosen = (pR < pL) ? ll : lr;          // the right child among equals
        if (lane == chosen) { S.hprio[cur] = p; S.hrec[cur] = rr; }
        curj = chosen - ((1 << lev) - 2);
        cur = (((hole + 1) << lev) + curj) - 1;
      }
    }
    hole = cur;
  }
  if ((len & 1) == 0 && hole == (len - 2) / 2) {         // a last node with a left child only
    const long long child = 2 * hole + 1;
    if (lane == 0) { S.hprio[hole] = S.hprio[child]; S.hrec[hole] = S.hrec[child]; }
    hole = child;
  }
  mn_ro_wave_pushup(S, hole, vp, vr, lane);              // ... and back up with the last entry's value
}

// ONE wave: lane 0 runs the maps and the records (mn_reforder.h), the wave the queue.  Comes back when `budget`
// pops (4 x budget records of the constructor's loop) are used up: the state is in memory, the next launch goes on.
__global__ __launch_bounds__(64) void mn_ro_loop(RoState S, int O, long long budget) {
  const int lane = threadIdx.x;
  const long long st0 = S.ctl[0];
  if (st0 != MN_RO_RUNNING && st0 != MN_RO_BUDGET) return;
  long long n = S.ctl[1], biggest = S.ctl[6], pops = 0;
  int status = MN_RO_RUNNING;
  long long t_init = 0, t_pop = 0, t_merge = 0, t_mark = wall_clock64();   // (100 MHz; MN_TRACE_EXACT prints them)
  // ---- the constructor's loop (segment.cc:209-231) ----
  long long r = S.ctl[5];
  if (r < S.NL) {
    long long left = budget * 4;
    for (; r < S.NL && status == MN_RO_RUNNING; r++) {
      if (left-- <= 0) { status = MN_RO_BUDGET; break; }
      if (S.r1[r] < 0) continue;                          // (uniform)
      int rc = MN_RO_RUNNING;
      if (lane == 0) rc = mn_ro_init_record(S, O, r);
      rc = __shfl(rc, 0);
      if (rc != MN_RO_RUNNING) { status = rc; break; }
      const float pr = S.prio[r];
      if (pr >= 0.0f) {
        if (n >= S.hcap) { status = MN_RO_HEAP_FULL; break; }
        mn_ro_wave_pushup(S, n, pr, (int)r, lane);
        n++;
        biggest = n > biggest ? n : biggest;
      }
    }
    if (lane == 0) S.ctl[5] = r;
    { const long long t = wall_clock64(); t_init += t - t_mark; t_mark = t; }
  }
  // ---- RunSegmentation (segment.cc:539-573) ----
  if (status == MN_RO_RUNNING && r >= S.NL) {
    status = MN_RO_DONE;
    while (n > 0) {
      if (pops >= budget) { status = MN_RO_BUDGET; break; }
      float q; int rec;
      mn_ro_wave_pop(S, n, &q, &rec, lane);
      pops++;
      { const long long t = wall_clock64(); t_pop += t - t_mark; t_mark = t; }
      if (q != S.prio[rec]) continue;                     // a stale entry (uniform: every lane reads the same word)
      if (S.r2[rec] < 0) continue;
      float f = 0.0f; int mc = 0;
      if (lane == 0) { f = mn_ro_score(S, rec, &mc); S.prio[rec] = f; }
      f = __shfl(f, 0);
      if (f == q) {
        int a = 0, b = 0, it = MN_RO_NULL, rc = MN_RO_RUNNING;
        if (lane == 0) rc = mn_ro_merge_begin(S, rec, mc, &a, &b, &it);
        rc = __shfl(rc, 0); it = __shfl(it, 0);
        while (rc == MN_RO_RUNNING && it != MN_RO_NULL) {
          int nx = MN_RO_NULL, prec = -1; float pp = 0.0f;
          if (lane == 0) rc = mn_ro_merge_node(S, a, b, it, &nx, &pp, &prec);
          rc = __shfl(rc, 0); nx = __shfl(nx, 0); prec = __shfl(prec, 0); pp = __shfl(pp, 0);
          if (rc == MN_RO_RUNNING && prec >= 0) {
            if (n >= S.hcap) { rc = MN_RO_HEAP_FULL; break; }
            mn_ro_wave_pushup(S, n, pp, prec, lane);
            n++;
            biggest = n > biggest ? n : biggest;
          }
          it = nx;
        }
        if (rc != MN_RO_RUNNING) { status = rc; break; }
        if (lane == 0) mn_ro_merge_end(S, rec, a, b);
        { const long long t = wall_clock64(); t_merge += t - t_mark; t_mark = t; }
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
