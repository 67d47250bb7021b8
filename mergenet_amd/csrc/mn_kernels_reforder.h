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
    for (int i = 0; i < 8; i++) S.ctl[i] = 0;
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

// ONE lane: the constructor's loop over the records, then the merge loop; comes back when `budget` records /
// pops are used up (the state is in memory: the next launch goes on)
__global__ __launch_bounds__(64) void mn_ro_loop(RoState S, int O, long long budget) {
  if (threadIdx.x != 0) return;
  long long st = S.ctl[0];
  if (st != MN_RO_RUNNING && st != MN_RO_BUDGET) return;
  int rc = MN_RO_DONE;
  if (S.ctl[5] < S.NL) {
    rc = mn_ro_init(S, O, budget * 4);
    if (rc == MN_RO_BUDGET) { S.ctl[0] = MN_RO_BUDGET; return; }
  }
  if (rc == MN_RO_DONE) rc = mn_ro_run(S, budget);
  S.ctl[0] = rc;
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
