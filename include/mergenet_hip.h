/* mergenet_hip.h -- C ABI of libmergenet_hip.so (MI355X / gfx950 pixel merger).
 *
 * Drop-in boundary for the MergeNet post-processor (the greedy merger that folds pixels into
 * instances by log-likelihood gain).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   c_run_segmentation      utils/csegment/segment.cc:742-765 (declared to Cython at
 *                           utils/csegment/c_segment.pyx:16-25)  -- same symbol, same signature
 *   ObjectSegmenterOption   utils/csegment/segment.h:245-268     -- mn_options (3 floats + mode)
 *   ObjectSegmenter ctor    utils/csegment/segment.cc:153-232    -- phase A (affinity scoring)
 *   RunSegmentation/Merge   utils/csegment/segment.cc:539-727    -- phase B (merge)
 *   OutputMask              utils/csegment/segment.cc:491-517    -- label / class-table output
 *   ObjectSegmenter (py)    utils/segmenter.py:225-483           -- MN_VARIANT_PYSEGMENTER
 *
 * Error behaviour: the reference returns void and calls exit(1) on internal inconsistencies
 * (segment.cc:39-43,96-100,665-673).  Here every entry point except the ABI-compatible
 * c_run_segmentation returns an int status (0 = MN_OK) and never exits; c_run_segmentation
 * keeps the void signature, prints the failure to stderr and leaves mn_last_status() set.
 *
 * Threading: an mn_context owns one GPU workspace and is used by one thread at a time; separate
 * contexts are independent (one per GPU / per process in the multi-GPU driver).
 */
#ifndef MERGENET_HIP_H_
#define MERGENET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MN_MAX_OFFSETS 32
#define MN_MAX_CLASSES 127

enum mn_status {
  MN_OK = 0,
  MN_ERR_ARGUMENT = -1,     /* null pointer, non-positive size, class_dim < num_classes, ...  */
  MN_ERR_OFFSETS = -2,      /* (0,0), a duplicate, or an offset together with its negation    */
  MN_ERR_NO_DEVICE = -3,    /* no usable HIP device / HIP call failed (message on stderr)     */
  MN_ERR_CAPACITY = -4,     /* image larger than the context was created for                  */
  MN_ERR_NO_BACKGROUND = -10, /* pysegmenter prune: no class-0 object (reference: NameError)  */
  MN_ERR_INTERNAL = -20,
  MN_ERR_UNPROVEN = -30     /* a proven result was asked for, the fast path could not certify its own and
                               the exact engine found no room for its workspace                   */
};

enum mn_variant {
  MN_VARIANT_CSEGMENT = 0,    /* utils/csegment semantics: den = n1+n2, bias outside, merge on == */
  MN_VARIANT_PYSEGMENTER = 1  /* utils/segmenter.py: den = n1*n2, bias inside, merge on >=, prune */
};

enum mn_mode {
  MN_MODE_AUTO = 0,      /* the reference's result, by the cheapest route that PROVES it: EXACT when the
                            image has <= exact_limit initial records; else COMPONENTS, kept when the
                            certificate holds (any order of the lazy greedy ends in this partition);
                            else EXACT (require_proof = -1: keep the fast path's unproven answer)   */
  MN_MODE_EXACT = 1,     /* the reference's sequential lazy-greedy order itself, any image size: the
                            exact engine (mn_kernels_exact.h) -- float32 state and operation order of
                            segment.cc, glibc's logf / expf restated bit for bit, pop = block-max queue in
                            LDS, per-object adjacency, pair table (seconds per image: DESIGN.md section 6) */
  MN_MODE_ROUNDS = 2,    /* parallel rounds + sequential finisher + certificate                  */
  MN_MODE_COMPONENTS = 3 /* sign-separable inputs: phase 1 of the merge (provably order-
                            independent there) by one union-find sweep over the positive edges,
                            then rounds/finisher on the records between components; falls back to
                            ROUNDS when the input is not sign-separable (mode_used tells)         */
};

enum mn_tie_order { MN_TIES_DEFAULT = 0, MN_TIES_REFERENCE = 1, MN_TIES_LOWEST_ID = 2 };
/* mn_stats.proof */
enum mn_proof {
  MN_PROOF_NONE = 0,             /* measured only: an approximation of the sequential order on order-dependent inputs */
  MN_PROOF_CERTIFICATE = 1,      /* ANY order of the lazy greedy ends in this partition          */
  MN_PROOF_SEQUENTIAL = 2,       /* the reference's sequential order was run and no choice among bit-equal priorities
                                    was left to the engine: every pop was forced (tied_steps == 0) or the reference's
                                    own heap / hash-map order among equals was reproduced (tie_order_used ==
                                    MN_TIES_REFERENCE)                                                               */
  MN_PROOF_SEQUENTIAL_TIES = 3   /* the sequential order was run, but some pops chose among bit-equal priorities by
                                    the engine's rule (lowest record id) where the reference's std::priority_queue
                                    chooses by heap position: equal to the reference on most vectors held at the
                                    benchmark's sizes, differing on the radius-4 blurred ones and on one blurred 1024x2048; require_proof = 1 redoes such an
                                    image in the reference's order                                                   */
};
#define MN_TIE_LIMIT_RECORDS 400000   /* MN_TIES_DEFAULT: largest image (initial records) redone in the reference's order */
#define MN_TIE_LIMIT_BATCH_RECORDS 1400000   /* ... inside mn_segment_exact_batch, where the images are redone together (256x512 at O = 10) */

typedef struct mn_options {
  float same_different_bias;   /* segment.h:246 */
  float object_merge_factor;   /* segment.h:247 */
  float merge_logprob_bias;    /* segment.h:248 */
  int variant;                 /* enum mn_variant */
  int mode;                    /* enum mn_mode */
  int clip_inputs;             /* 1: clip to [2^-23, 1-2^-23] on load (c_segment.pyx:53-55 fused) */
  int exact_limit;             /* AUTO: images with at most this many initial records go to EXACT right
                                  away (0 = default 32768)                                        */
  int finish_limit;            /* ROUNDS: hand over to the sequential finisher at <= this many
                                  live records (0 = default: 2048 in the rounds, 4096 records
                                  between components in components mode)                         */
  int subrounds;               /* ROUNDS: matching sub-rounds per round (0 = default 32)         */
  float prune_threshold;       /* pysegmenter prune threshold (segmenter.py:351; default 200)    */
  int compute_logprob;         /* 1: also evaluate the total log-likelihood (segment.cc:314-350) and
                                  the certificate; 0: skip both (total_logprob NaN, certified 0)  */
  int no_handover_refresh;     /* ROUNDS: 1 = keep stored priorities when the finisher takes over  */
  int band_permille;           /* ROUNDS: a round merges only records whose gain is >= this many
                                  thousandths of the round's best gain (0 = default 50, <0 = off;
                                  round 1, rounds from single pixels: parity with the reference lost at
                                  10, once in 65 runs at 25, never from 50 up; round 2, rounds from the
                                  cores: 100, 50 and 20 give the same verdict and pixel agreement on
                                  every reference vector, 50 is a fifth faster than 100)            */
  int debug_flags;             /* bit 0: use the generic edge pass where the fast form would run (tests
                                  compare the two); bit 1: no per-kernel timestamps in components mode
                                  (ms_cc_* stay 0; each is an event on the caller's stream); bit 2:
                                  general rounds from single pixels instead of from the cores; bit 4: only
                                  the sweep is timed (ms_cc_edges; the other ms_* stay 0) -- an event
                                  costs the host ~3.5 us to record and ~8 us to read; bit 5 (with
                                  bit 4): replay -- when mn_segment_launch is called again with the
                                  same buffers, shape, options and stream, the launches after the
                                  sweep are recorded into two hipGraphs (second call) and replayed
                                  (from the third): a loop over images through fixed buffers; bit 7:
                                  time the sweep with an event packet before and behind it instead
                                  of start/stop events on its own dispatch (round 2's first form:
                                  measures dispatch gap + kernel).  (Round 4 removed the opt-in engines
                                  that were measured slower or known to deviate: bits 3, 8-13.)          */
  int require_proof;           /* what happens to a result that is not PROVEN equal to the reference's
                                  sequential order (stats.proof == 0): 1 = it is redone in MN_MODE_EXACT,
                                  whatever mode was asked for, and -- if that run chose among bit-equal
                                  priorities by its own rule (proof 3) -- once more in the reference's order
                                  among equals (tie_order MN_TIES_REFERENCE; MN_ERR_UNPROVEN for the Python
                                  variant, whose heapq order is not restated); -1 = it is handed back as it is (the
                                  speculative fast path: an approximation on order-dependent inputs,
                                  stats.proof tells); 0 (default) = by mode: AUTO redoes it, an explicit
                                  MN_MODE_ROUNDS / MN_MODE_COMPONENTS request is taken as a request for
                                  that engine's own answer                                          */
  int core_radius;             /* general rounds: an offset counts as SHORT when both its components
                                  are at most this many pixels; a pixel is clean -- and may join a core
                                  ahead of the rounds -- when all its short edges are positive and
                                  same-class (0 = default; < 0 = every offset
                                  is short: the widest fringe, the closest to the reference's order)   */
  int tie_order;               /* MN_MODE_EXACT (and what AUTO redoes by it): who goes first among records with
                                  bit-equal priorities.  MN_TIES_LOWEST_ID: the record created first -- the
                                  exact engine's own rule.  MN_TIES_REFERENCE: what the reference's
                                  std::priority_queue and std::unordered_map (libstdc++, GCC 11: the build the
                                  golden vectors come from) make of it -- binary-heap position and hash-map
                                  iteration order, restated on flat arrays (mn_reforder.h); the maps are worked
                                  by ONE lane, the heap by its wave: the reference's very partition also on
                                  maps with plateaus of equal values, at 20-40x the exact engine's time
                                  (csegment variant only).  MN_TIES_DEFAULT (0): the exact engine first; if it
                                  met tied pops whose choices conflict (stats.tied_conflicts > 0: only then can
                                  the two rules end in different states) and the image has at most
                                  MN_TIE_LIMIT_RECORDS initial records, it is redone in the reference's order;
                                  larger images keep the exact engine's answer and say so (stats.proof == 3,
                                  stats.tie_order_used, stats.tied_steps, stats.tied_conflicts)              */
} mn_options;

typedef struct mn_stats {
  int status;
  int mode_used;               /* MN_MODE_EXACT, MN_MODE_ROUNDS or MN_MODE_COMPONENTS */
  int certified;               /* 1: result proven equal to the sequential reference partition
                                  (sign-separable input, see DESIGN.md "certificate")           */
  int num_instances;           /* labels 1..K written to the mask */
  int num_objects;             /* surviving objects including class-0 ones */
  int rounds;                  /* parallel rounds executed */
  int finisher_steps;          /* sequential steps (pops) executed by the finisher */
  int cert_edge_violations;    /* pixel edges whose log-odds sign contradicts the partition   */
  int cert_class_violations;   /* pixels whose own arg-max class differs from their object's  */
  int cert_record_violations;  /* records between final objects that are still mergeable      */
  long long initial_records;   /* in-bounds (pixel, offset) pairs */
  long long merges;            /* objects absorbed */
  double total_logprob;        /* A.4: sum lp[cls] + omf*(sum log p | log(1-p)); NaN if not asked */
  float ms_score;              /* phase A kernels (class pass + edge pass), HIP events */
  float ms_class_pass;
  float ms_edge_pass;
  float ms_merge;              /* phase B */
  float ms_output;             /* labels, mask, class table, certificate, log-likelihood */
  float ms_total;
  /* components mode only (0 otherwise), HIP events on the launch stream: */
  float ms_cc_label;           /* mn_cc_tiles + borders + flatten + hook: union-find on the 4 B/pixel masks */
  float ms_cc_sums;            /* mn_cc_class_sums: reads the C class planes                            */
  float ms_cc_edges;           /* mn_cc_sign: THE read of the O sameness planes (masks, negative edges)  */
  float ms_cc_cross;           /* mn_cc_cross: negative-edge list -> records between components         */
  int proof;                   /* enum mn_proof: why (and whether) the partition equals the reference's     */
  int cores_condemned;         /* general rounds: 1 if a core held an edge that was not positive and
                                  fell apart again (mn_core_check); 0 otherwise                    */
  int tied_steps;              /* MN_MODE_EXACT: pops at which a second live record held the bit-equal stored
                                  priority (saturates at INT_MAX).  0 = every pop was forced: the result is
                                  the reference's whatever its heap does among equals.  > 0: the reference's
                                  std::priority_queue picks by heap position, the engine by lowest record id;
                                  the two orders usually commute, DESIGN.md section 5 has the inputs where
                                  they do not (radius-4 blurred, clipped maps)                             */
  int tied_merges;             /* ... of which were merges */
  int tie_order_used;          /* MN_MODE_EXACT: MN_TIES_LOWEST_ID or MN_TIES_REFERENCE (0 on the other paths) */
  int tied_conflicts;          /* MN_MODE_EXACT: 0 = no tied pop's choice WROTE an object's state that another tied
                                  choice read or wrote (a merge writes its two ends and reads their neighbours):
                                  the tied events commute and every order among equals ends in this state (then
                                  proof == 2 even with tied_steps > 0); > 0 = at least one did (the engine stops
                                  looking at the first: a yes / no, not a count).  mn_kernels_exact.h "ties"      */
} mn_stats;

typedef struct mn_context mn_context;

/* Fill `o` with the Cityscapes caller's options (egs/cityscape/local/segment.py:134-136):
 * same_different_bias 0, object_merge_factor 1, merge_logprob_bias 0.03, csegment, AUTO. */
void mn_default_options(mn_options* o);

/* Create a context on HIP device `device` able to hold images up to max_height x max_width with
 * up to max_classes class planes and max_offsets offsets.  All device memory is allocated here;
 * mn_segment_device allocates nothing.  Returns NULL on failure (see mn_last_status). */
mn_context* mn_create(int device, int max_height, int max_width, int max_classes, int max_offsets);
void mn_destroy(mn_context* ctx);
size_t mn_workspace_bytes(const mn_context* ctx);

/* Segment one image whose probability maps are ALREADY ON THE DEVICE.
 *   d_class_pred [class_dim][H][W] float32, d_adj_pred [offset_dim][H][W] float32 (C order)
 *   offset_list  HOST pointer, [offset_dim][2] = (d_row, d_col)   (segment.cc:166-169)
 *   d_mask       [H][W] int32 out: 0 = every class-0 object, 1..K instances (segment.cc:491-517)
 *   d_object_class [H*W] int32 out: class of label k at k-1, -1 from index K on
 *   d_partition  optional [H*W] int32 out: surviving object id per pixel before the class-0
 *                collapse (may be NULL)
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  The call enqueues work and
 * synchronises the stream before returning when `stats` is non-NULL or the mode needs host
 * decisions (ROUNDS does, per round).  Inputs are never modified (the reference rewrites
 * adj_pred in place when same_different_bias != 0; here the bias is applied on load). */
int mn_segment_device(mn_context* ctx, const float* d_class_pred, int class_dim,
                      const float* d_adj_pred, int offset_dim, int img_width, int img_height,
                      int num_classes, const int* offset_list, int* d_mask, int* d_object_class,
                      int* d_partition, const mn_options* opts, void* stream, mn_stats* stats);

/* The same in two halves, for callers that keep the GPU busy across images: mn_segment_launch
 * queues one image and returns -- in components mode (the default for large images) without
 * having waited for anything; mn_segment_finish waits, reads the verdict (redoing the image on the
 * ordinary path if the speculative attempt does not hold) and fills `stats`.  Between the two the
 * context is busy and inputs, outputs and stream must stay alive; with two contexts on one stream
 * the launch of image i+1 can precede the finish of image i, which hides the host round trip
 * without letting kernels of different images overlap.  A negative return of mn_segment_launch
 * leaves nothing pending. */
int mn_segment_launch(mn_context* ctx, const float* d_class_pred, int class_dim,
                      const float* d_adj_pred, int offset_dim, int img_width, int img_height,
                      int num_classes, const int* offset_list, int* d_mask, int* d_object_class,
                      int* d_partition, const mn_options* opts, void* stream);
int mn_segment_finish(mn_context* ctx, mn_stats* stats);

/* Tuning aid: the sweep alone, `reps` launches back to back over `n_inputs` input sets in rotation (more than
 * 256 MB of inputs in all keeps the Infinity Cache from serving them); average microseconds per launch. */
int mn_sweep_time_device(mn_context* ctx, const float* const* d_class_pred, const float* const* d_adj_pred,
                         int n_inputs, int class_dim, int offset_dim, int img_width, int img_height,
                         int num_classes, const int* offset_list, const mn_options* opts, void* stream, int reps,
                         float* us_per_launch);

/* `count` images of ONE shape through the exact engine (MN_MODE_EXACT) together: the engine's loop is one
 * wavefront per image, so a batch is ONE launch with a workgroup per image -- images in flight are how the
 * sequential order gets throughput (the reference scales the same way, by processes: --num-jobs).  One
 * context per image (each holds its image's workspace, ~1.5 GB at 512x1024, ~6 GB at 1024x2048); arrays of
 * `count` device pointers (host arrays); d_partition may be NULL; stats: `count` entries or NULL.  The call
 * returns when all images are done.  Results are those of `count` separate MN_MODE_EXACT calls; the images the
 * tie policy sends to the reference-order loop (opts->tie_order; stats.tied_conflicts > 0) are redone TOGETHER,
 * one workgroup per image in one launch of that loop, and up to MN_TIE_LIMIT_BATCH_RECORDS initial records
 * instead of MN_TIE_LIMIT_RECORDS (the loop is sequential: images in flight are its throughput). */
int mn_segment_exact_batch(mn_context** ctxs, int count, const float* const* d_class_pred, int class_dim,
                           const float* const* d_adj_pred, int offset_dim, int img_width, int img_height,
                           int num_classes, const int* offset_list, int* const* d_mask,
                           int* const* d_object_class, int* const* d_partition, const mn_options* opts,
                           void* stream, mn_stats* stats);

/* Phase A alone (per-pixel class log-probs + argmax, per-edge log-odds and initial priorities,
 * best initial record per pixel).  Used by bench.py / profiles to time the affinity-scoring pass
 * against the HBM roofline, and by tests to compare phase-A arrays with the oracle.
 *   d_cls_out   [H*W] uint8  argmax class per pixel                     (segment.cc:18-20)
 *   d_best_out  [H*W] uint64 (priority bits << 32 | ~partner pixel id), 0 = no record >= 0 */
int mn_score_device(mn_context* ctx, const float* d_class_pred, int class_dim,
                    const float* d_adj_pred, int offset_dim, int img_width, int img_height,
                    int num_classes, const int* offset_list, const mn_options* opts, void* stream,
                    unsigned char* d_cls_out, unsigned long long* d_best_out, float* ms_class_pass,
                    float* ms_edge_pass);

/* The affinity-scoring sweep of the default path alone (mn_cc_sign: ONE read of the C class planes and the
 * O sameness planes) and what it leaves behind, for the parity test of the sweep itself against the
 * oracle's phase A (segment.cc:5-46):
 *   d_bits_out [H*W] uint32   bit k = out-edge of offset k is in bounds and positive (value >= sep_hi)
 *   d_neg_out  [O][H*W] f32   log-odds of every NEGATIVE in-bounds edge the sweep listed, NaN elsewhere
 *   d_cls_out  [H*W] uint8    per-pixel arg-max class            (only written in the fused-class form)
 *   d_gsum_out [C][H*W/4] i32 per lane and class the 2^-24 fixed-point log of the product of its four
 *                             pixels' class values             (only written in the fused-class form)
 *   logsum_out (host)         the certificate's sum over in-bounds edges of log max(v, 1 - v)
 *   info_out   (host) int[3]  {pixels per lane (4 | 1), fused-class form (1 | 0), edges inside the
 *                             float32 rounding margin of 0.5 (they fail the separability check)}
 * d_cls_out / d_gsum_out may be NULL.  Synchronises. */
int mn_sweep_device(mn_context* ctx, const float* d_class_pred, int class_dim, const float* d_adj_pred,
                    int offset_dim, int img_width, int img_height, int num_classes, const int* offset_list,
                    const mn_options* opts, void* stream, unsigned* d_bits_out, float* d_neg_out,
                    unsigned char* d_cls_out, int* d_gsum_out, double* logsum_out, int* info_out);

/* Phase A of the exact engine alone (tests pin it against the oracle, bit for bit): per record the
 * float32 log-odds obj_merge_logprob (segment.cc:33-36: logf and a double log, as glibc computes them)
 * and the initial merge priority (segment.cc:107-150), laid out [offset][source pixel] with NaN where
 * the edge leaves the image; d_cls_out [H*W] uint8 arg-max class (may be NULL).  Synchronises. */
int mn_exact_phase_a_device(mn_context* ctx, const float* d_class_pred, int class_dim,
                            const float* d_adj_pred, int offset_dim, int img_width, int img_height,
                            int num_classes, const int* offset_list, const mn_options* opts, void* stream,
                            unsigned char* d_cls_out, float* d_oml_out, float* d_prio_out);

/* Host-pointer convenience: copies in, runs mn_segment_device, copies out. */
int mn_segment_host(mn_context* ctx, const float* class_pred, int class_dim, const float* adj_pred,
                    int offset_dim, int img_width, int img_height, int num_classes,
                    const int* offset_list, int* mask, int* object_class, int* partition,
                    const mn_options* opts, mn_stats* stats);

/* ABI-compatible replacement of the reference entry point (utils/csegment/segment.cc:742-754).
 * Width precedes height.  Host pointers; inputs must already be clipped as the reference binding
 * does (c_segment.pyx:53-55).  Unlike the reference it does not rewrite adj_pred. */
void c_run_segmentation(float* class_pred, int class_dim, float* adj_pred, int offset_dim,
                        int img_width, int img_height, int num_classes, int* offset_list,
                        int* output, int* object_class, float same_different_bias,
                        float object_merge_factor, float merge_logprob_bias);

/* Producer hand-off on the device ("next" row 1 of the scope table): logits or probabilities
 * [channels][in_h][in_w] -> probabilities [channels][out_h][out_w] in ONE pass: optional sigmoid
 * (utils/inference_utils.py:44,96), bilinear resize with cv2.resize/INTER_LINEAR coordinates
 * (egs/cityscape/local/segment.py:115-123) and the binding's clip (c_segment.pyx:53-55).  Replaces
 * the .npy round trip between the network and the merger. */
int mn_prepare_device(mn_context* ctx, const float* d_in, int channels, int in_height, int in_width,
                      float* d_out, int out_height, int out_width, int apply_sigmoid, int clip,
                      void* stream);

/* Nearest-neighbour resize of the instance mask back to the image size, cv2 INTER_NEAREST
 * coordinates (egs/cityscape/local/segment.py:146-149). */
int mn_upsample_mask_device(mn_context* ctx, const int* d_mask, int in_height, int in_width,
                            int* d_out, int out_height, int out_width, void* stream);

/* Run boundaries of ALL instances in one pass, for COCO RLE (egs/cityscape/local/segment.py:
 * 165-186 encodes each instance with pycocotools over the Fortran-ordered binary mask).  The
 * label scan runs column-major; d_points gets [positions | label before | label at] with stride
 * `capacity`, *count the number of change points.  The host groups them per label. */
int mn_rle_points_device(mn_context* ctx, const int* d_mask, int height, int width, int* d_points,
                         int capacity, int* count, void* stream);

/* HOST helper of the RLE path: groups the change points of mn_rle_points_device per instance and
 * writes pycocotools' compressed counts strings (egs/cityscape/local/segment.py:165-186: one
 * maskUtils.encode per instance), in native code (the Python loop it replaces took 2.8 ms for 21
 * instances).  points = [positions | label before | label at] with stride `capacity` (host copy of
 * d_points), n of them.  Strings are concatenated into `out` (string k-1 = out[offsets[k-1] ..
 * offsets[k]) ), areas[k-1] = pixels of instance k (0 => the caller may drop it, as
 * egs/cityscape/local/evaluate.py:52-54 does).  Returns the bytes needed; nothing is written past
 * out_capacity.  Needs no GPU. */
long long mn_rle_encode_host(const int* points, int capacity, int n, int height, int width,
                             int num_instances, unsigned char* out, long long out_capacity,
                             long long* offsets, int* areas);

/* Sameness targets of an instance mask: out[k][r][c] = (mask[r+di][c+dj] == mask[r][c]), 1 outside
 * the image (utils/dataset.py:259-277).  d_out is float32 [offset_dim][H][W]. */
int mn_sameness_targets_device(mn_context* ctx, const int* d_mask, int height, int width,
                               const int* offset_list, int offset_dim, float* d_out, void* stream);

/* Confidence of the instances of the last mn_segment_device call: d_scores[k-1] = lp[cls] - lp[0]
 * of label k (segment.h:109); the reference's COCO results carry a constant score 1. */
int mn_instance_scores_device(mn_context* ctx, float* d_scores, void* stream);

/* Wire format of the multi-GPU mask exchange (the all-gather of final instance masks the north
 * star asks for; the reference has no exchange, its jobs write files: segment.py:59-61).  d_wire is
 * int16 [n_pixels + 1 + max_instances + 4]: the labels (0..K), K, the classes of labels 1..K
 * padded with -1, then the float64 total log-likelihood as four 16-bit words, low word first
 * (the scalars ride the same gather).  Needs no context; max_instances <= 32767. */
int mn_pack_wire_device(const int* d_mask, const int* d_object_class, int num_instances,
                        double total_logprob, int n_pixels, int max_instances, short* d_wire,
                        void* stream);

/* Run-length wire format of the same exchange: the row-major label CHANGE POINTS instead of the
 * label of every pixel (the masks are piecewise constant).  d_wire is int32[mn_runs_wire_words(
 * capacity, max_instances)]: [0] change points (-1: more than `capacity`, the image does not fit)
 * [1] K [2..3] float64 log-likelihood, `capacity` ascending positions, `capacity` int16 labels,
 * max_instances int8 classes.  At capacity = n_pixels / 32 the wire is 10.6x smaller than the int16
 * map (397 KB per 1024x2048 image).  No host synchronisation.  mn_unpack_runs_device restores the
 * dense mask (and, if d_table is given, max_instances int32 classes, -1 padded). */
size_t mn_runs_wire_words(int capacity, int max_instances);
int mn_pack_runs_device(mn_context* ctx, const int* d_mask, const int* d_object_class, int num_instances,
                        double total_logprob, int n_pixels, int capacity, int max_instances,
                        int* d_wire, void* stream);
int mn_unpack_runs_device(const int* d_wire, int n_pixels, int capacity, int max_instances,
                          int* d_mask, int* d_table, void* stream);
/* The same for `count` wires in ONE launch (the gathered wires of all ranks): wire i starts at
 * d_wires + i * wire_stride_words, mask i at d_masks + i * n_pixels, table i at d_tables + i * max_instances. */
int mn_unpack_runs_batch_device(const int* d_wires, long long wire_stride_words, int count, int n_pixels,
                                int capacity, int max_instances, int* d_masks, int* d_tables, void* stream);

int mn_last_status(void);
const char* mn_status_string(int status);
const char* mn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MERGENET_HIP_H_ */
