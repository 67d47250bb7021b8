// mn_kernels_tail.h -- the end of an image in components mode, in ONE single-workgroup kernel.
//
// After the contraction an image is a few dozen records between a few dozen components, and what
// is left to do -- table -> record list with fresh priorities, the sequential second phase
// (RunSegmentation / Merge, utils/csegment/segment.cc:539-727, in mn_fin_lds_run), the check that
// no record between final objects is still mergeable, certificate and log-likelihood
// (ComputeTotalLogprobFromScratch, segment.cc:314-350), labels 1..K and the class table
// (OutputMask, segment.cc:491-517) -- was seven launches of almost no work each (compact,
// finisher, verify_records, certificate, rank_count / scan / assign: ~70 us of launch latency and
// dependent round trips per image).  Here the stages run back to back in one workgroup of
// MN_FIN2_THREADS lanes; only the pixel-wide mask write stays a kernel of its own.  Speculative
// path only (the host has seen neither the record count nor the verdict): anything that does not
// fit -- violations, more records than `spec_limit`, more component roots than MN_TAIL_MAXROOTS --
// leaves the counters so that the host redoes the image on the ordinary path.
#pragma once

#include "mn_device.h"
#include "mn_kernels_merge.h"
#include "mn_kernels_finish.h"
#include "mn_kernels_cc.h"

#define MN_TAIL_MAXROOTS 2048
#define MN_TAIL_TABLE_CAP 4096   /* slots of the speculative attempt's record table (<= 1024 records) */

__global__ __launch_bounds__(MN_FIN2_THREADS) void mn_cc_tail(
    ImgParams P, ObjState S, HashTab T, const int* __restrict__ tcount, RecList L,
    int* __restrict__ lcount, int* maprec, int* __restrict__ lists,
    Counters* __restrict__ cnt, long long max_steps, int* __restrict__ scalars, int spec_limit,
    const unsigned char* __restrict__ cls0, const int* __restrict__ compsize,
    const int* __restrict__ rootlist, int nb_edges, const double* __restrict__ partial_edges,
    double* __restrict__ lp_out, int want_cert, int* label,
    int* __restrict__ object_class) {
  __shared__ int s_w[MN_FIN2_WAVES];
  __shared__ int s_total, s_k;
  __shared__ int s_ids[MN_TAIL_MAXROOTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nroots = scalars[8];
  if (scalars[6] != 0 || scalars[7] != 0) return;                       // not separable / list full (uniform)
  if (nroots > MN_TAIL_MAXROOTS) { if (tid == 0) scalars[7] = 1; return; }   // redo on the ordinary path

  // ---- 1. table -> record list (any fixed order will do: ties are broken by the keys) ----------
  // the slots of a lane are loaded together (independent loads: one round trip, not one per slot)
  const unsigned cap = T.mask + 1u;
  constexpr int PER = MN_TAIL_TABLE_CAP / MN_FIN2_THREADS;
  u64 keys[PER];
  int mine = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const unsigned slot = (unsigned)j * MN_FIN2_THREADS + tid;
    keys[j] = slot < cap ? T.key[slot] : MN_EMPTY;
  }
#pragma unroll
  for (int j = 0; j < PER; j++) mine += keys[j] != MN_EMPTY;
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < MN_FIN2_WAVES; w++) { if (w < wave) woff += s_w[w]; total += s_w[w]; }
  if (tid == 0) cnt->n_records = total;
  if (total > spec_limit || total > MN_FIN2_MAXR) return;               // uniform: host redoes the image
  int pos = woff + incl - mine;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const u64 key = keys[j];
    if (key == MN_EMPTY) continue;
    const unsigned slot = (unsigned)j * MN_FIN2_THREADS + tid;
    const i64 s = T.S[slot];
    int mc;
    bool gp;
    const float f = mn_score(P, S, mn_key_u(key), mn_key_v(key), mn_fixed_to_float(s), &mc, &gp);
    L.key[pos] = key;
    L.S[pos] = s;
    L.st[pos] = f;
    lcount[pos] = tcount[slot];
    pos++;
  }
  __syncthreads();

  // ---- 2. the sequential second phase --------------------------------------------------------
  mn_fin_lds_run(P, S, L, total, maprec, lists, cnt, max_steps, lcount);
  __syncthreads();

  // ---- 3. no record between final objects may still be mergeable (mn_verify_records) ----------
  if (want_cert) {
    const float margin = 1e-6f + 1e-5f * fabsf(P.bias);
    int still = 0;
    for (int i = tid; i < total; i += MN_FIN2_THREADS) {
      const u64 key = L.key[i];
      if (key == MN_EMPTY) continue;
      int mc;
      bool gp;
      const float f = mn_score(P, S, mn_key_u(key), mn_key_v(key), mn_fixed_to_float(L.S[i]), &mc, &gp);
      if (!(f < -margin)) still++;
    }
    if (still) atomicAdd(scalars + 4, still);
    // ---- 4. certificate and log-likelihood ------------------------------------------------------
    mn_cc_certificate_run(P, S, cls0, compsize, rootlist, nroots, nb_edges, partial_edges, cnt, lp_out, scalars);
  }

  // ---- 5. labels 1..K in ascending surviving id, class table (mn_rank_*) -------------------------
  // every final object is one of the component roots; an instance is a live root of class != 0
  for (int j = tid; j < nroots; j += MN_FIN2_THREADS) {
    const int r = rootlist[j];
    const bool alive = S.parent[r] == r;
    s_ids[j] = (alive && S.ocls[r] != 0) ? r : (alive ? -2 - r : -1);   // instance | live, class 0 | absorbed
  }
  if (tid == 0) { s_total = 0; s_k = 0; }
  __syncthreads();
  int n_alive = 0, n_inst = 0;
  for (int j = tid; j < nroots; j += MN_FIN2_THREADS) {
    const int v = s_ids[j];
    if (v == -1) continue;
    n_alive++;
    if (v < 0) { label[-2 - v] = 0; continue; }
    int rank = 0;
    for (int t = 0; t < nroots; t++) { const int o = s_ids[t]; rank += (o >= 0 && o < v) ? 1 : 0; }
    label[v] = rank + 1;
    object_class[rank] = S.ocls[v];
    n_inst++;
  }
  if (n_alive) atomicAdd(&s_total, n_alive);
  if (n_inst) atomicAdd(&s_k, n_inst);
  __syncthreads();
  if (tid == 0) { scalars[1] = s_k; scalars[2] = s_total; }
}
