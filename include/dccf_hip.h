/*
 * dccf_hip.h — C ABI of the MI355X-native DCCF training / predict hot path (libdccf_hip.so).
 *
 * The reference (rutgerswiselab/DCCF) has no FFI: its plug-in boundary is a Python class picked by
 * `--model_name` (src/main.py:43,120-148) whose arithmetic is a sequence of stock PyTorch ops.  Each entry point
 * below replaces one such group of call sites; the host side (dccf_amd/, Python like the reference) keeps the
 * reference's class / runner interface and calls these through ctypes.  INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless a comment says "host"; PyTorch-ROCm owns all buffers;
 *   - ids are int64 as in the reference (`feed_dict['X']`, src/data_processor/DataProcessor.py:149-158);
 *   - all floating point is fp32, row-major, no padding; `stream` is a hipStream_t (NULL = default stream);
 *   - every function is stream-ordered and returns 0 on success, <0 for an argument error, >0 = hipError_t;
 *     `dccf_last_error()` returns a thread-local message.  No C++ exception crosses this boundary;
 *   - the library allocates nothing except the workspace owned by a `dccf_ctx` (grow-only; call
 *     `dccf_ctx_reserve` up front to keep the step functions allocation-free, e.g. for hipGraph capture).
 */
#ifndef DCCF_HIP_H
#define DCCF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dccf_ctx dccf_ctx;

/* ---- context ------------------------------------------------------------------------------------------- */
int dccf_ctx_create(dccf_ctx** out, int device);
int dccf_ctx_destroy(dccf_ctx* ctx);
/* Pre-size the workspace for batches of up to `max_rows` rows of X (train: 2*batch_size; eval: eval_batch_size). */
int dccf_ctx_reserve(dccf_ctx* ctx, int64_t max_rows, int32_t D, int32_t F, int32_t S, int32_t A);
const char* dccf_last_error(void);
int dccf_abi_version(void);
/* Optional per-kernel timing: HIP events on the launch stream around every kernel of dccf_predict / dccf_train_fwdbwd.
 * dccf_profile_read adds elapsed ms / launch counts since the last read into HOST arrays of 8 slots:
 * 0 prep, 1 mlp_fwd (extra layers), 2 noise_fwd, 3 pair_epilogue, 4 mlp_bwd (extra layers + user gradient), 5 bwd,
 * 6 opt_launch (the optimizer launch of dccf_train_step). */
/* Deterministic gradient scatter (tests; off unless the environment variable DCCF_DETERMINISTIC=1 is set when the context is
 * created).  The float atomics of the backward add a row's contributions in arrival order, so two runs of one step differ in
 * the last bits of a gradient row.  With on != 0 dccf_train_fwdbwd / dccf_train_step use no float atomic: every (batch row,
 * candidate) slot's gradient row is stored and the rows of one destination are added in slot order, gb and dW from partial sums in
 * index order, the loss by one workgroup — two runs, and every form of the step (split calls, prepared, hosted, lazy), then
 * agree bit for bit.  Slower (a test mode); not available in the replicated path's slot mode (ignored there). */
int dccf_ctx_set_deterministic(dccf_ctx* ctx, int on);
int dccf_profile(dccf_ctx* ctx, int enable);
int dccf_profile_read(dccf_ctx* ctx, double* ms, int64_t* counts);

/* ---- DCCF model view: replaces the attributes set up by DCCF._init_weights (src/models/DCCF.py:47-64) ----- */
typedef struct {
  int64_t user_num, item_num;
  int32_t D;            /* --u_vector_size == --i_vector_size (src/models/RecModel.py:17-27); any 1 <= D <= 256   */
                        /* (above 128: n_extra must be 0, dccf_predict_projected and the deterministic mode do not apply) */
  int32_t F;            /* feature width, taken from the .npy (src/models/DCCF.py:59); any F >= 1               */
  int32_t S;            /* --sample-num   (src/models/DCCF.py:19)                                              */
  int32_t A;            /* --attribute-num (src/models/DCCF.py:20)                                             */
  float   std;          /* --std          (src/models/DCCF.py:21)                                              */
  float   reserved;
  const float* U;       /* uid_embeddings.weight [user_num, D]                                                 */
  const float* V;       /* iid_embeddings.weight [item_num, D]                                                 */
  const float* W;       /* mlp.0.weight [D, D+F]                                                               */
  const float* b;       /* mlp.0.bias   [D]                                                                    */
  const float* feat;    /* feature_embedding [item_num, F]      (frozen)                                       */
  const float* expo;    /* expo_prob [user_num, item_num] (frozen) or NULL -> computed from the factors below  */
  /* on-the-fly exposure = IPSBiasedMF.predict(u, i) (src/models/IPSBiasedMF.py:37-57): exactly the number the   */
  /* dense file would hold (README.md:28-30), for tables too large for a dense U x I matrix                     */
  const float* ipsP;    /* [user_num, ipsD] */
  const float* ipsQ;    /* [item_num, ipsD] */
  const float* ipsBu;   /* [user_num]       */
  const float* ipsBi;   /* [item_num]       */
  const float* ipsProp; /* [item_num]       */
  float   ipsB0, ipsM;
  int32_t ipsD;
  int32_t n_extra;      /* --n_layers - 1 (src/models/DMF.py:14): D -> D layers after mlp.0 (src/models/DCCF.py:61-62,   */
                        /* 91-94), each Linear + relu + dropout; 0 <= n_extra <= DCCF_MAX_EXTRA                          */
  const float* Wl[7];   /* mlp.k.weight [D, D], k = 1 .. n_extra  (Wl[k-1])                                             */
  const float* bl[7];   /* mlp.k.bias   [D]                                                                             */
  const float* expo_gathered; /* optional [N, S+1]: Expo[u(n), cand[n][s]] of THIS call's batch, gathered by the caller (the
                         * row-sharded trainer: the owner of a user row holds that user's row of the exposure matrix and ships
                         * the 2 (S+1) values a pair needs with the embedding row).  Takes precedence over expo / the factors;
                         * needs injected candidates (rnd.mode 0 or 2): the caller gathered at exactly those.                  */
} dccf_model_t;
#define DCCF_MAX_EXTRA 7

/* ---- the random draws of DCCF.predict (src/models/DCCF.py:72,87,94) ---------------------------------------- */
typedef struct {
  int32_t mode;                /* 0 = injected ("golden"): use the three arrays; 1 = fused: Philox4x32-10(seed, step); */
                               /* 2 = candidates injected (sample_item), noise + dropout fused (row-sharded path)   */
  int32_t reserved;
  const int64_t* sample_item;  /* [N, S]      candidates as torch.randint would return them                      */
  const float*   noise;        /* [N*(S+1)*A, F]  N(0, std^2) draws (already scaled by std)                       */
  const uint8_t* keep;         /* [1 + n_extra][N*(S+1)*A, D]  dropout keep masks, layer-major (1 = kept) or NULL = keep all */
  uint64_t seed;               /* fused mode: stream key                                                        */
  uint64_t step;               /* fused mode: call counter (one value per forward)                              */
  /* Optional device-side step counter, so that one captured hipGraph can be replayed for every step of a run: with
   * k = *k_dev the call uses X + (k % x_steps) * x_stride (int64 elements) and the stream counter step + k. */
  const int64_t* k_dev;
  int64_t x_stride;
  int64_t x_steps;
} dccf_rand_t;

typedef struct {             /* dense-shaped gradients of the LOSS term, accumulated (+=) — zero them first     */
  float* gU;                 /* [user_num, D]   */
  float* gV;                 /* [item_num, D]   */
  float* gW;                 /* [D, D+F]        */
  float* gb;                 /* [D]             */
  uint8_t* touchedU;         /* optional [user_num]: set to 1 for every row of gU this call adds to (NULL = not kept)   */
  uint8_t* touchedV;         /* optional [item_num]: same for gV; consumed by dccf_dense_opt_step_rows                  */
  float* gWl[7];             /* [D, D] gradients of mlp.k.weight, k = 1 .. n_extra (unused entries NULL)                  */
  float* gbl[7];             /* [D]                                                                                     */
} dccf_grads_t;

/* DCCF.predict (src/models/DCCF.py:66-107): X int64 [N,2] -> prediction fp32 [N].  `dropout` is feed_dict['dropout']. */
int dccf_predict(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X, int64_t N,
                 float dropout, float* prediction, void* stream);

/* Evaluation-only predict with PROJECTED noise (an option; dccf_predict is the op-for-op path).  No gradient is taken at
 * evaluation, and z_noise = W_f eps with eps ~ N(0, std^2 I_F) iid per row is exactly N(0, std^2 W_f W_f^T): drawing D
 * normals xi and forming L xi (L L^T = std^2 W_f W_f^T) samples the same distribution with a K = D product instead of
 * K = F, and W_f feat[i] becomes a row of a table.  dccf_eval_prepare fills Pf [item_num, D] = feat W_f^T and
 * Lt [D, D] (Lt[k][d] = L[d][k], pivot-clamped fp64 Cholesky) — call it once after the parameters changed;
 * dccf_predict_projected then has the DISTRIBUTION of DCCF.predict (src/models/DCCF.py:66-107), not the same draws
 * (rnd.mode must be 1: candidates, xi and dropout from the Philox streams). */
int dccf_eval_prepare(dccf_ctx* ctx, const dccf_model_t* model, float* Pf, float* Lt, void* stream);
int dccf_predict_projected(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X, int64_t N,
                           float dropout, const float* Pf, const float* Lt, float* prediction, void* stream);

/* DCCF.forward + loss.backward() (src/models/DCCF.py:109-127, src/runners/BaseRunner.py:180-183) for the loss term:
 * rank==1: rows [0,N/2) positives, [N/2,N) their negatives, loss = -sum log sigmoid(pos-neg); rank==0: MSE vs Y.
 * Writes prediction [N], loss [1] (device scalar) and accumulates the four gradients. */
int dccf_train_fwdbwd(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X,
                      const float* Y, int64_t N, int32_t rank, float dropout, const dccf_grads_t* grads,
                      float* prediction, float* loss, void* stream);

/* ---- dense regularised optimizer step: replaces `+ model.l2()*l2` in the loss, clip_grad_value_(50) and
 * torch.optim.{SGD,Adagrad,Adam}(lr, weight_decay=l2).step()  (src/runners/BaseRunner.py:92-100,181-187).
 *   g_total = clip(g + l2 * 2p, +-clip);  then the optimizer with coupled weight decay `wd` (torch semantics);
 *   `step` is the 1-based step count (bias correction); s1/s2 are Adam's exp_avg/exp_avg_sq or Adagrad's sum (s2 unused);
 *   zero_grad != 0 writes zeros back into g (the next step's `optimizer.zero_grad()`, src/runners/BaseRunner.py:178). */
#define DCCF_OPT_GD 0
#define DCCF_OPT_ADAGRAD 1
#define DCCF_OPT_ADAM 2
int dccf_dense_opt_step(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                        float l2, float clip, int64_t step, int32_t zero_grad, void* stream);
/* Same step for a flat buffer whose first part is row-structured: segment q covers elements [seg_begin[q], seg_begin[q] +
 * seg_rows[q]*seg_width[q]) as rows of seg_width[q] floats (a multiple of 4 up to 256; seg_begin a multiple of 256) with one
 * "touched" byte per row (dccf_grads_t.touchedU/V).  Every such width takes the row-aware streaming pass described next and its
 * two-phase form (widths other than 16, 32, 64, 128: the row of a slot by a division instead of a shift, the bytes cleared by a
 * memset after the launch; segments below 2^32 elements); only the form HOSTED in the backward launch (dccf_train_step with
 * overlap == 2) needs one of the four widths.  A row whose byte is 0 has an all-zero gradient by construction, so
 * g is neither read nor re-zeroed for it (24 instead of 32 B/param of traffic); touched rows are read, zeroed and their
 * byte cleared.  Elements outside the segments are treated densely.  seg_* are HOST arrays, nseg <= 4. */
int dccf_dense_opt_step_rows(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                             float l2, float clip, int64_t step, int32_t nseg, const int64_t* seg_begin,
                             const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags, void* stream);
/* The same step in two launches, for callers that know BEFORE the backward which rows a step will touch (a de-duplicated
 * `list` of (segment << 40) | row with its device-side count, flags set for exactly those rows):
 *   phase 1: every row whose byte is 0 (gradient = l2 term only) — independent of the batch, may run on another stream
 *            while forward / backward / gradient exchange are in flight;
 *   phase 2: the listed rows (gradient read, zeroed, byte cleared) and everything outside the segments (W, b).
 * Together they equal dccf_dense_opt_step_rows bit for bit. */
int dccf_dense_opt_phase(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd, float l2,
                         float clip, int64_t step, int32_t nseg, const int64_t* seg_begin, const int64_t* seg_rows,
                         const int32_t* seg_width, uint8_t* const* seg_flags, int32_t phase, const int64_t* list,
                         const int32_t* cnt, int64_t max_rows, void* stream);

/* ---- ONE training step = the body of BaseRunner.fit's batch loop (src/runners/BaseRunner.py:172-188): forward, loss,
 * backward, + l2, clip, optimizer step, zero_grad.  Same results, bit for bit, as dccf_train_fwdbwd followed by
 * dccf_dense_opt_step_rows.  With overlap != 0 the optimizer pass over the rows this batch does NOT touch (all but a few
 * thousand of the U/V rows: their gradient is the l2 term alone, independent of the batch) runs on the context's
 * low-priority side stream WHILE forward/backward run on `stream` (overlap == 1) or as extra workgroups of the backward
 * launch (overlap == 2: a backward workgroup is one wave per SIMD, a second workgroup fits beside it on every CU); the
 * touched rows + W, b follow after the backward.  Both are measured options, not the default.
 * Needs grads->touchedU / touchedV == the flags of the segments that hold model->U / model->V, 4-byte aligned and padded
 * to a multiple of 4 bytes.
 * X_next (optional, overlap == 0): the NEXT call's batch — same N, fused draws (rnd.mode 1) with Philox step `step_next`.
 * The optimizer launch then also prepares that step (candidates, gathered exposures, zeroed accumulators, and the
 * transposed copy of W written while W is updated), so the next call starts with its forward kernel instead of k_prep.
 * The context remembers (pointer, N, step, seed, tables); a call that does not match runs k_prep as usual, and any other
 * call on the context (predict, a different batch) discards what was prepared. */
typedef struct {
  int32_t kind;              /* DCCF_OPT_*                                                          */
  int32_t overlap;
  float* p;                  /* flat parameters; model->U/V/W/b point into it                         */
  float* g;                  /* flat gradients;  grads->gU/... point into it                          */
  float* s1;
  float* s2;
  int64_t n;
  float lr, wd, l2, clip;
  int64_t step;              /* 1-based                                                              */
  int32_t nseg;
  int32_t reserved;
  const int64_t* seg_begin;  /* HOST arrays, as in dccf_dense_opt_step_rows                          */
  const int64_t* seg_rows;
  const int32_t* seg_width;
  uint8_t* const* seg_flags;
  /* ---- windowed lazy regularisation (dccf_train_step only; lazy_K == 0: every launch is the dense pass) ----------------
   * A row the batch does not touch sees the l2 term alone: its update at step t is a function of its own (p, m, v) and of
   * t.  With lazy_K = K > 0 the optimizer launch of step t therefore updates only (a) the rows the step touched, (b) W, b
   * and (c) ONE K-th of the untouched rows, which it advances by all the steps they are behind (<= K, replayed in registers
   * in the dense pass's exact operation order: bit-identical results) — 24 B/param/step of HBM traffic become 24/K, and the
   * launch is bound by the arithmetic of the replay instead.  A row about to be used is first brought up to step t - 1
   * (k_lazy_catchup, before the forward).  dccf_lazy_flush brings every row up to date (before evaluation, checkpoints, l2,
   * or any dense call).  All arrays are the caller's, in HBM:
   *   lazy_last  int32 [rows of all segments, segment after segment]: steps applied to the row so far
   *   lazy_claim int32 [2 x the same], 0-initialised: the last step of each parity that used the row (two arrays: the
   *              optimizer launch of step t claims the rows of step t + 1 while its other roles read the claims of step t)
   *   lazy_list  int32 [2 x lazy_list_cap], lazy_list_cap >= N (S + 2): the rows of a step, one entry per (row of X, candidate)
   *              slot and per user slot, -1 where another slot owns the row; by step parity like the claims
 *   lazy_cnt   int32 [16], zero-initialised: the window whose marks are pending (a step's window is marked by the NEXT
 *              optimizer launch instead of a launch of its own), two records by step parity
   *   lazy_scal  float [4 * lazy_nscal], 16-byte aligned: (-(lr / (1 - 0.9^s)), c = sqrt(1 - 0.999^s), RN(1 / c), 0) for
   *              s = lazy_t0 .. lazy_t0 + lazy_nscal - 1, as dccf_lazy_scalars writes them (the same double arithmetic as the
   *              dense launch); must cover [step - lazy_K, step] */
  int32_t lazy_K;
  int32_t lazy_nscal;
  int32_t* lazy_last;
  int32_t* lazy_claim;
  int32_t* lazy_list;
  int32_t* lazy_cnt;
  const float* lazy_scal;
  int64_t lazy_t0;
  int64_t lazy_list_cap;
  int64_t lazy_id;       /* names THIS set of lazy_* arrays: a new value whenever they are re-created or cleared (the context
                          * remembers for which arrays the previous optimizer launch claimed the next step's rows) */
  int64_t* lazy_host;    /* HOST int64 [2] owned by the caller, or NULL (no check): [0] = the last step whose lazy optimizer launch
                          * went out (the library writes it; the caller sets it when it brings every row to a step by other means).
                          * The lazy arrays are only valid if steps arrive one by one — no row more than lazy_K steps behind — so a
                          * step's launches need opt->step == [0] + 1 and dccf_lazy_flush opt->step == [0]: anything else is an
                          * argument error.  Should a row be further behind all the same, the kernels replay it from step - lazy_K
                          * only (never an index in front of lazy_scal) and set lazy_cnt[15] = 1 for the caller to check. */
} dccf_opt_t;
/* HOST: out[4 i .. 4 i + 3] = the Adam step scalars of step t0 + i (see lazy_scal), i < n. */
int dccf_lazy_scalars(float lr, int64_t t0, int32_t n, float* out_host);
/* The two launches of a lazy step for a caller that drives the step itself (the row-sharded trainer, dccf_amd/sharded.py;
 * dccf_train_step does the same internally).  dccf_lazy_catchup_rows, BEFORE anything reads the parameters: the rows the step
 * will touch — rows_a[0..n_a) of segment seg_a, rows_b[0..n_b) of segment seg_b, duplicates allowed — are claimed for
 * opt->step, listed (n_a + n_b <= lazy_list_cap) and brought up to step - 1.  dccf_lazy_opt_step, once their gradient rows are
 * complete in opt->g: the listed rows get step opt->step with their gradient (which is zeroed), everything outside the row
 * segments (W, b) the dense step, and this step's window of the other rows is advanced; nslots = n_a + n_b of the catch-up. */
int dccf_lazy_catchup_rows(const dccf_opt_t* opt, const int32_t* rows_a, int64_t n_a, int32_t seg_a, const int32_t* rows_b,
                           int64_t n_b, int32_t seg_b, void* stream);
int dccf_lazy_opt_step(const dccf_opt_t* opt, int64_t nslots, void* stream);
/* Brings every row of the segments up to opt->step (rows already there are untouched).  Needs lazy_K > 0. */
int dccf_lazy_flush(const dccf_opt_t* opt, void* stream);
/* The context's side stream (hipStream_t): least priority, or confined to the first n CUs when the environment variable
 * DCCF_SIDE_CUS=n is set at its creation (the mask interleaves over the 8 XCDs).  For callers that run
 * dccf_dense_opt_phase(1) beside other work themselves (dccf_amd/replicated.py). */
int dccf_ctx_side_stream(dccf_ctx* ctx, void** out);
/* Diagnostics: *out = how many dccf_train_step calls on this context started from a step prepared by the previous call
 * (X_next matched) instead of launching k_prep. */
int dccf_ctx_prepared_steps(const dccf_ctx* ctx, int64_t* out);
/* Diagnostics: *out = item rows whose untouched-row optimizer pass rode in the backward launch of the last training call on
 * this context (dccf_train_step at 2B <= 2048; 0 otherwise) — what bench.py needs to price the two launches by their bytes. */
int dccf_ctx_hosted_rows(const dccf_ctx* ctx, int64_t* out);
int dccf_train_step(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X, const float* Y,
                    int64_t N, int32_t rank, float dropout, const dccf_grads_t* grads, const dccf_opt_t* opt,
                    float* prediction, float* loss, const int64_t* X_next, uint64_t step_next, void* stream);

/* Graph-replayable form of the two calls above: the 1-based step is step + *k_dev (bias corrections computed on the
 * device); dccf_advance adds 1 to *k_dev (last node of a captured step). */
int dccf_dense_opt_step_dev(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                            float l2, float clip, int64_t step, const int64_t* k_dev, int32_t nseg,
                            const int64_t* seg_begin, const int64_t* seg_rows, const int32_t* seg_width,
                            uint8_t* const* seg_flags, void* stream);
int dccf_advance(int64_t* k_dev, void* stream);
/* BaseModel.l2 (src/models/BaseModel.py:179-187): out[0] += sum p^2  (out must be zeroed by the caller). */
int dccf_sumsq(const float* p, int64_t n, float* out, void* stream);

/* ---- MF family: RecModel / BiasedMF / IPSBiasedMF predict (src/models/RecModel.py:38-48, BiasedMF.py:17-33,
 * IPSBiasedMF.py:37-57) ------------------------------------------------------------------------------------ */
typedef struct {
  int64_t user_num, item_num;
  int32_t D;
  int32_t kind;           /* 0 RecModel, 1 BiasedMF, 2 IPSBiasedMF */
  const float* P;         /* uid_embeddings.weight [user_num, D] */
  const float* Q;         /* iid_embeddings.weight [item_num, D] */
  const float* bu;        /* user_bias.weight [user_num] (kind>=1) */
  const float* bi;        /* item_bias.weight [item_num] (kind>=1) */
  const float* b0;        /* global_bias [1] device scalar (kind>=1) */
  const float* prop;      /* propensity [item_num] (kind==2) */
  float M;                /* --M (src/models/IPSBiasedMF.py:14) */
  float reserved;
} mf_model_t;

typedef struct { float* gP; float* gQ; float* gbu; float* gbi; float* gb0;
                 uint8_t* touchedP; uint8_t* touchedQ;   /* optional, as in dccf_grads_t: one byte per gP / gQ row */
} mf_grads_t;

int mf_predict(const mf_model_t* model, const int64_t* X, int64_t N, float* prediction, void* stream);
int mf_train_fwdbwd(dccf_ctx* ctx, const mf_model_t* model, const int64_t* X, const float* Y, int64_t N, int32_t rank,
                    const mf_grads_t* grads, float* prediction, float* loss, void* stream);
/* The whole body of the batch loop for the MF family (src/runners/BaseRunner.py:172-188) in one call, under the windowed lazy
 * regularisation (opt->lazy_K > 0, opt's two row segments = P and Q of `model`): the batch's rows are claimed and caught up,
 * forward + loss + backward run, and one optimizer launch applies step opt->step to the batch's rows (with their gradient), to
 * everything outside the row segments (bias vectors, global bias: dense) and to this step's window of the other rows.
 * ids: unused (may be NULL; kept for ABI 5).  Results equal mf_train_fwdbwd + dccf_dense_opt_step_rows (untouched rows bit for bit). */
int mf_train_step(dccf_ctx* ctx, const mf_model_t* model, const int64_t* X, const float* Y, int64_t N, int32_t rank,
                  const mf_grads_t* grads, const dccf_opt_t* opt, int32_t* ids, float* prediction, float* loss, void* stream);
/* "save the full predicted user-item matrix as the exposure probability" (README.md:28-30): out [user_num, item_num], row-major,
 * 4-byte aligned; any embedding size D <= 256 (the contraction is zero-padded to 16 / 32 / 64 / 128; above 128: two launches). */
int mf_predict_full(const mf_model_t* model, float* out, void* stream);

/* ---- fused on-device training negatives: replaces DataProcessor._sample_neg_from_uid_list for train=True, neg_n=1
 * (src/data_processor/DataProcessor.py:446-524).  rows_indptr/rows: the train rows (sample ids) of each user (CSR);
 * hist_indptr/hist_items: each user's train positives, sorted ascending (CSR).  neg_out[sample_id] = negative item. */
int dccf_sample_train_negatives(const int64_t* rows_indptr, const int64_t* rows, const int64_t* hist_indptr,
                                const int64_t* hist_items, int64_t user_num, int64_t item_num, uint64_t seed,
                                uint64_t epoch, int64_t* neg_out, void* stream);
/* The epoch's feed dicts (src/data_processor/DataProcessor.py:160-207,227-250) as one tensor: with the epoch's permutation
 * perm [n] (shuffle_in_unison_scary, src/utils/utils.py:82-92) batch k of full [n / B, 2B, 2] is X = [pos ; neg] with
 * pos row j = (uid, iid)[perm[kB + j]] and neg row j = (uid, neg)[perm[kB + j]]; the n % B rows left over are the shorter last
 * batch tail [2 (n % B), 2].  A negative of -1 is stored as 0 and *bad (device int32, zeroed by the caller) set to 1.
 * perm == NULL: the permutation is a keyed bijection of (seed, epoch) computed inside the kernel (no array, no sort). */
int dccf_build_epoch_batches(const int64_t* uid, const int64_t* iid, const int64_t* neg, const int64_t* perm, int64_t n,
                             int64_t batch_size, int64_t* full, int64_t* tail, int32_t* bad, uint64_t seed, uint64_t epoch,
                             void* stream);
/* Eval negatives (src/data_processor/DataProcessor.py:408-444,446-524 with train=False): neg_n items per DISTINCT user of a
 * split (users [n_users], first-occurrence order), uniform over the items, outside the user's train + validation/test
 * history (hist CSR by user id, items sorted) and distinct.  Draw j of user u = word j%4 of Philox(c0=u, c1=j/4, c2=tag)
 * on stream 6; draws are consumed in order (accepted iff admissible and not accepted before), the first neg_n accepted
 * ones, in draw order, go to out [n_users, neg_n] (-1 where the reference would assert: fewer than neg_n items left).
 * tag separates the splits (1 validation, 2 test).  neg_n <= 2048. */
int dccf_sample_eval_negatives(const int64_t* users, int64_t n_users, const int64_t* hist_indptr, const int64_t* hist_items,
                               int64_t item_num, int32_t neg_n, uint64_t seed, uint64_t tag, int64_t* out, void* stream);

/* ---- ranking metrics on the device: replaces the sort / groupby('uid') / per-group loops of BaseModel.evaluate_method
 * (src/models/BaseModel.py:83-126) with dcg/ndcg(method=1)/precision/recall/hit of src/utils/rank_metrics.py:61-87,130-201.
 * indptr/rows: CSR of each user's rows of the eval split (indices into pred/label).  ks_host (HOST) / ks_dev (device): the
 * cut-offs, 1 <= k <= 1024 (16 ranks per walk of a user's rows), nk <= 4.  out [n_users, nk+1, 4] fp32: per user and cut-off (ndcg, hit, precision, recall); slot
 * nk holds the user's number of positives.  The caller averages over users (np.average in the reference). */
int rank_eval_topk(const float* pred, const float* label, const int64_t* indptr, const int64_t* rows, int64_t n_users,
                   const int32_t* ks_host, const int32_t* ks_dev, int32_t nk, float* out, void* stream);

/* ---- gradient exchange of the replicated data-parallel path (dccf_amd/replicated.py; new capability, the reference is
 * single-GPU: src/main.py:106,153-155).  Every rank holds the whole model; per step the ranks all-gather ONE buffer each:
 *   32-bit words [count | loss | 0 0 | ids int64[cap] | rows fp32[cap][D] | dense fp32[n - dense_begin]]
 * dp_export_touched fills this rank's buffer from its local gradient: every row whose "touched" byte is set (segments as
 * in dccf_dense_opt_step_rows, all of width D, flags 4-byte aligned and zero-padded to whole words) is moved out of g
 * (the row is zeroed, the byte cleared), id = (segment << 40) | row; the dense tail g[dense_begin:n] ([dW | db]) is moved
 * too, *loss copied.  dp_import_touched takes the G gathered buffers and leaves in g, for every listed row, the sum of the
 * ranks' rows IN RANK ORDER (bit-identical on every replica, no float atomics), sets the bytes again and sums the dense
 * tails and losses in rank order — g is then exactly what one GPU would hold after a backward over the G batches, and
 * dccf_dense_opt_step_rows finishes the step.  Scratch (R = total rows of the segments): mask uint32 [R] zero-initialised
 * (the import leaves it zero), where int32 [G*R].
 * The export appends behind the counter in buf[0]: reset != 0 zeroes it first (one more launch); a training loop passes its
 * local buffer as reset_buf to the import instead, which zeroes the counter for the next step's export. */
int64_t dp_buffer_words(int64_t cap, int32_t D, int64_t nd);
/* dp_import_touched + phase 2 of the two-phase optimizer step in one pass: the rank-ordered sums are not stored to g but fed
 * straight into the optimizer (kind, lr, ... as in dccf_dense_opt_step) for the summed rows and the dense tail; seg_flags
 * are the bytes that marked those rows for phase 1 (dccf_dense_opt_phase(1)) and are cleared.  Every row some rank touched
 * has an entry, so together with phase 1 every parameter is updated exactly once. */
int dp_import_apply(const float* bufs, int32_t G, int32_t kind, float* p, float* s1, float* s2, int64_t n, float lr, float wd,
                    float l2, float clip, int64_t step, int32_t nseg, const int64_t* seg_begin, const int64_t* seg_rows,
                    const int32_t* seg_width, uint8_t* const* seg_flags, int64_t dense_begin, float* loss_sum, int64_t cap,
                    int32_t D, uint32_t* mask, int32_t* where, float* reset_buf, void* stream);
/* Marks (bytes + de-duplicated list, as dccf_dense_opt_phase wants them) every row ANY of the G ranks will touch in the step
 * whose rank-0 Philox step word is step0: X_all int64 [G][N][2] is the replicated schedule, rank r's S candidates per row
 * are the STREAM_CAND draws of (seed, step0 + r) — what dccf_train_fwdbwd draws on rank r in fused mode.  segU / segV: the
 * segment indices of the user / item tables; *cnt must be 0 on entry, *cnt_next is zeroed (double-buffered counters). */
int dp_mark_global(const int64_t* X_all, int32_t G, int64_t N, int32_t S, int64_t item_num, uint64_t seed, uint64_t step0,
                   uint8_t* flagsU, uint8_t* flagsV, int32_t segU, int32_t segV, int64_t* list, int32_t* cnt,
                   int32_t* cnt_next, void* stream);
int dp_export_touched(float* g, int64_t n, int32_t nseg, const int64_t* seg_begin, const int64_t* seg_rows,
                      const int32_t* seg_width, uint8_t* const* seg_flags, int64_t dense_begin, const float* loss, float* buf,
                      int64_t cap, int32_t D, int32_t reset, void* stream);
int dp_import_touched(const float* bufs, int32_t G, float* g, int64_t n, int32_t nseg, const int64_t* seg_begin,
                      const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags, int64_t dense_begin,
                      float* loss_sum, int64_t cap, int32_t D, uint32_t* mask, int32_t* where, float* reset_buf, void* stream);

/* The three calls of one replicated data-parallel step, for a host loop that has to stay below ~100 us per step
 * (dccf_amd/replicated.py): state of the exchange in one struct, the optimizer in a dccf_opt_t whose segments carry the
 * LOCAL touched bytes (grads->touchedU/V).
 *   dccf_dp_local    dccf_train_fwdbwd (loss -> dp->loss) + dp_export_touched into dp->buf; with X_all != NULL (int64
 *                    [G][N][2], Philox step of rank 0 = step0) the global marking of dp_mark_global rides in the export
 *                    launch                                                                            [then: all-gather]
 *   dccf_dp_overlap  dccf_dense_opt_phase(1) on the global marks (of this parity) — enqueue it right after the all-gather
 *                    was launched
 *   dccf_dp_finish   overlap != 0: dp_import_apply;  else dp_import_touched + dccf_dense_opt_step_rows   [after the wait]
 * parity alternates 0, 1, 0, ... over the steps (the same value in the three calls of a step). */
typedef struct {
  int32_t G, rank, D, S;
  int64_t cap, dense_begin, item_num;
  uint64_t seed;
  float* buf;                /* this rank's export buffer, dp_buffer_words(cap, D, n - dense_begin) words  */
  float* bufs;               /* the G gathered buffers                                                      */
  float* loss;               /* [1] this rank's loss                                                        */
  float* loss_sum;           /* [1] sum over the ranks                                                      */
  uint8_t* gflagsU;          /* "touched by ANY rank" bytes of the user / item segment (overlap mode)       */
  uint8_t* gflagsV;
  uint8_t* gflagsU2;         /* the second set (steps of parity 1) — needed only for dccf_dp_next_t, else NULL */
  uint8_t* gflagsV2;
  uint8_t* lflagsU;          /* de-duplication marks of THIS rank's rows (zero between steps), padded to words; */
  uint8_t* lflagsV;          /* with llist [cap + 64] and lcnt [2] (by parity): dccf_dp_next_t only, else NULL   */
  int64_t* llist;
  int32_t* lcnt;
  dccf_ctx* ctx;             /* the context dccf_dp_local runs on (needed when pmask / pwhere are given)              */
  uint32_t* pmask;           /* optional [2][R] zeros and [2][G * R] 0x7fffffff (R = rows of all segments): the import */
  int32_t* pwhere;           /* tables of a prepared step are then built one step ahead too (no k_dp_scatter_ids)    */
  int32_t segU, segV;        /* indices of those segments in opt->seg_*                                      */
  int64_t* glist;            /* unused by dccf_dp_* (the marks are bytes only); may be NULL                   */
  int32_t* gcnt;
  uint32_t* mask;            /* scratch of the import, see dp_import_touched                                 */
  int32_t* where;
} dccf_dp_t;
int dccf_dp_local(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X, const float* Y, int64_t N,
                  float dropout, const dccf_grads_t* grads, const dccf_opt_t* opt, const dccf_dp_t* dp, const int64_t* X_all,
                  uint64_t step0, int32_t parity, float* prediction, void* stream);
/* What the host knows about the NEXT step (same N, overlap mode), handed to dccf_dp_overlap AND dccf_dp_finish of this step:
 * the optimizer launches then also prepare it — this rank's candidates / exposures / W^T (no k_prep), this rank's rows as a
 * list (the export becomes a gather of ~3 k listed rows instead of a scan of every flag), and every rank's rows marked in
 * the flag set of the other parity (no marking role).  dccf_dp_local recognises the prepared step by (X, X_all, N, Philox
 * step, parity); anything else discards the preparation.  Results are identical with or without it. */
typedef struct {
  dccf_ctx* ctx;
  const dccf_model_t* model;
  const int64_t* X_next;       /* this rank's batch of the next step [N][2]: the pointer dccf_dp_local will get as X */
  const int64_t* X_all_next;   /* [G][N][2]: the pointer dccf_dp_local will get as X_all                            */
  int64_t N;
  uint64_t step0_next;         /* Philox step word of rank 0 in the next step                                        */
} dccf_dp_next_t;
int dccf_dp_overlap(const dccf_opt_t* opt, const dccf_dp_t* dp, int32_t parity, const dccf_dp_next_t* next, void* stream);
int dccf_dp_finish(const dccf_opt_t* opt, const dccf_dp_t* dp, int32_t overlap, int32_t parity, const dccf_dp_next_t* next,
                   void* stream);

/* ---- row movers of the row-sharded multi-GPU path (dccf_amd/sharded.py; no reference counterpart — the reference is
 * single-GPU, src/main.py:106,153-155).  `tables` / `widths` are HOST arrays of up to 4 device pointers / row widths. */
/* out[dst[j] (j when dst is NULL), 0:sum(widths)] = [T0[idx[j], :] | T1[idx[j], :] | ...]; payload rows are ld floats
 * apart — the rows a peer needs, as one all-to-all payload */
int shard_pack_rows(const int32_t* idx, const int32_t* dst, int64_t n, const float* const* tables, const int32_t* widths,
                    int32_t ntables, float* out, int32_t ld, void* stream);
/* Tq[dst[j] (j when dst is NULL), :] = in[j, off_q : off_q + w_q]  — a received payload into the compact tables */
int shard_unpack_rows(const float* in, int32_t ld, int64_t n, const int32_t* dst, float* const* tables,
                      const int32_t* widths, int32_t ntables, void* stream);
/* g[idx[j], :] += rows[j, :]  — received gradient rows into the owner's gradient shard (float atomics) */
int shard_scatter_add(const int32_t* idx, int64_t n, const float* rows, int32_t width, float* g, uint8_t* flags,
                      void* stream);   /* flags: optional "touched" byte per row of g (dccf_dense_opt_step_rows) */

/* Several pack / unpack jobs in ONE launch (a step of the sharded path sends user rows, item rows and feature rows and
 * receives two payload kinds; at batch 128 every launch counts).  HOST array of up to 4 jobs, each as the arguments of
 * shard_pack_rows / shard_unpack_rows: `buf` is the payload (written by pack, read by unpack), rows `ld` floats apart.
 * shard_unpack_multi also zeroes zero[0 : zero_n] (the compact per-step gradient table) in the same launch. */
typedef struct {
  const int32_t* idx;        /* pack: source row of payload row j; unpack: unused                 */
  const int32_t* dst;        /* payload row (pack) / table row (unpack) of entry j; NULL = j      */
  int64_t n;
  float* tables[4];
  int32_t widths[4];
  int32_t ntables;
  int32_t ld;
  float* buf;
} shard_job_t;
int shard_pack_multi(const shard_job_t* jobs, int32_t njobs, void* stream);
int shard_unpack_multi(const shard_job_t* jobs, int32_t njobs, float* zero, int64_t zero_n, void* stream);

/* ---- the row-sharded trainer's collectives straight on RCCL (dccf_amd/sharded.py; no reference counterpart: the reference is
 * single-GPU, src/main.py:106,153-155).  torch.distributed issues the same ncclSend / ncclRecv / ncclAllReduce calls, at 25-40 us
 * of host time per call; these are one C call each on the caller's stream.  RCCL is resolved at run time from the copy the
 * process has mapped (PyTorch-ROCm's) — no link-time dependency.  dccf_comm_unique_id: on rank 0, 128 bytes to hand to every
 * rank (e.g. by a torch.distributed broadcast); dccf_comm_create: collective over the `world` ranks, the calling thread's
 * current HIP device.  dccf_comm_all_to_all_rows: peer q receives send_rows[q] rows of `width` floats (contiguous in `send`,
 * peers in rank order) and delivers recv_rows[q] rows (contiguous in `recv`, in rank order), one RCCL group;
 * dccf_comm_all_reduce_sum: in place.  Error codes >= 1000 are 1000 + ncclResult_t. */
int dccf_comm_unique_id(uint8_t* out128);
int dccf_comm_create(void** comm, const uint8_t* id128, int32_t world, int32_t rank);
int dccf_comm_destroy(void* comm);
int dccf_comm_all_to_all_rows(void* comm, const float* send, const int64_t* send_rows, float* recv, const int64_t* recv_rows,
                              int64_t width, void* stream);
/* two payloads of different row widths in ONE group (the sharded forward exchange: embedding-side rows + feature rows) */
int dccf_comm_all_to_all_rows2(void* comm, const float* send_a, const int64_t* send_rows_a, float* recv_a, const int64_t* recv_rows_a,
                               int64_t width_a, const float* send_b, const int64_t* send_rows_b, float* recv_b,
                               const int64_t* recv_rows_b, int64_t width_b, void* stream);
int dccf_comm_all_reduce_sum(void* comm, float* buf, int64_t n, void* stream);

/* ---- the fused-mode random streams written out (for parity tests: fused == injected on the same draws) ---------- */
int dccf_debug_candidates(int64_t N, int32_t S, int64_t item_num, uint64_t seed, uint64_t step, int64_t* out, void* stream);
int dccf_debug_noise(int64_t L, int32_t F, float std, uint64_t seed, uint64_t step, float* out, void* stream);
/* One optimizer step of `kind` (step number `step`) on n elements, element by element: ieee = 0 runs the element function
 * every optimizer kernel of the library uses (Adam: divisions and square root without the range scaling of the compiler's
 * IEEE expansions, see opt_device.hpp), ieee = 1 the same step on __fdiv_rn / __fsqrt_rn.  ieee = 2 / 3 (Adam): nothing is
 * updated; g[i] = sqrt(s2[i]) / sqrt(1 - 0.999^step) + 1e-8 by the library's functions (2) or the IEEE ones (3).  Tests assert
 * that 0 == 1 and 2 == 3 bit for bit.  ieee = 4: the step in the four-elements-at-a-time form the float4 kernels use
 * (16-byte aligned arrays); 4 == 0 bit for bit. */
int dccf_debug_opt_elem(int32_t kind, int32_t ieee, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                        float l2, float clip, int64_t step, void* stream);
int dccf_debug_keep(int64_t L, int32_t D, float dropout, uint64_t seed, uint64_t step, uint8_t* out, void* stream);
/* the keep mask of mlp layer `layer` (0 = mlp.0; the extra layers of --n_layers > 1 draw with the layer index in the counter) */
int dccf_debug_keep_layer(int64_t L, int32_t D, float dropout, uint64_t seed, uint64_t step, int32_t layer, uint8_t* out,
                          void* stream);

/* Copies one workspace array of the last call with these shapes to dst (device; NULL = only fill info[4] =
 * {DP, FP, element count, element size}).  which: 0 cand(int32) 1 WT 3 h 4 m 5 dmns.  Tests only. */
int dccf_debug_workspace(dccf_ctx* ctx, int64_t N, int32_t D, int32_t F, int32_t S, int32_t A, int32_t which, void* dst,
                         int64_t* info, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCCF_HIP_H */
