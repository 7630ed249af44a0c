// common.hpp — shared device helpers for libdccf_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/dccf_hip.h"

#define DCCF_ABI_VERSION 6
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------- errors
extern thread_local char g_dccf_err[512];
static inline int dccf_fail(int code, const char* msg) {
  snprintf(g_dccf_err, sizeof(g_dccf_err), "%s", msg);
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      snprintf(g_dccf_err, sizeof(g_dccf_err), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
               __FILE__, __LINE__);                                                            \
      return (int)_e;                                                                          \
    }                                                                                          \
  } while (0)
#define ARG_CHECK(cond, msg) \
  do {                       \
    if (!(cond)) return dccf_fail(-1, "argument error: " msg); \
  } while (0)

// ---------------------------------------------------------------------------------------------- context
#define DCCF_PROF_SLOTS 8     // 0 prep 1 mlp_fwd 2 noise_fwd 3 pair_epilogue 4 mlp_bwd 5 bwd 6 opt_launch (dccf_train_step)
#define DCCF_PROF_EVENTS 16384
struct dccf_ctx {
  int device;
  char* ws;         // one grow-only slab
  size_t ws_bytes;
  // optional per-kernel timing with HIP events on the launch stream (dccf_profile / dccf_profile_read)
  int prof_on;
  hipEvent_t* ev;
  int* ev_slot;
  int ev_used;
  // overlapped training step (dccf_train_step): low-priority side stream for the untouched-row optimizer pass, fork/join
  // events, and the de-duplicated list of the rows this step touches (built by k_prep, consumed by k_opt_touched)
  hipStream_t side;
  hipEvent_t ev_fork, ev_join;
  // what the last dccf_train_step prepared for the next one (candidates, exposures, W^T, zeroed accumulators)
  int prep_valid;
  float* gw_part;                // scratch of GwPart (grown on demand)
  size_t gw_part_bytes;
  // deterministic mode (dccf_ctx_set_deterministic): per-slot gradient rows of the backward + "first slot of a row" table
  int det;
  float* det_buf;
  size_t det_bytes;
  int* det_owner;
  int64_t det_owner_n;
  int64_t lazy_prep_step;        // optimizer step whose rows the last lazy optimizer launch claimed and caught up (-1: none)
  const void* lazy_prep_claim;   // ... in this claim array
  int64_t lazy_prep_id;          // ... of the arrays the caller named so (dccf_opt_t.lazy_id)
  int64_t prep_hits;
  const void* prep_X;
  const void* prep_U;
  const void* prep_W;
  // replicated path: the dp part of the prepared step (global marks in set prep_parity, local list) and its schedule
  int prep_dp, prep_pending, prep_parity;
  int prep_tables, cur_tables;     // the import tables of the prepared / of the running step were built ahead
  // slot mode of the backward (dccf_dp_local with tables): embedding gradient rows go straight into the slots of the
  // all-gather buffer (row -> slot via slot_where), there is no export pass; and what the import needs to recompute the
  // row of a slot (the replicated schedule of the running step)
  const int* slot_where;
  float* slot_rows;
  int64_t slot_offU, slot_offV;
  int slot_cap;
  const void* cur_Xall;
  uint64_t cur_step0;
  int64_t cur_N;
  // dccf_train_step, item table hosted: the rows of V a step touches, as bytes, known BEFORE its backward (two sets by
  // parity: the optimizer launch of step t marks step t + 1 while it consumes the marks of step t)
  uint8_t* hv_flags[2];
  int64_t hv_items;
  int hv_parity, hv_prepared;
  int64_t last_hosted_rows;  // item rows whose untouched-row pass the last training call hosted in its backward launch
  const void* prep_Xall;
  int64_t prep_N;
  uint64_t prep_step, prep_seed;
  int64_t* tl_list;
  int64_t tl_cap;
  int* tl_cnt;      // [2]: counters of the current / next step (double-buffered so no launch is spent on the reset)
  int tl_parity;
};
static inline void prof_begin(dccf_ctx* c, hipStream_t st) {
  if (c->prof_on && c->ev_used + 2 <= DCCF_PROF_EVENTS) (void)hipEventRecord(c->ev[c->ev_used], st);
}
static inline void prof_end(dccf_ctx* c, int slot, hipStream_t st) {
  if (c->prof_on && c->ev_used + 2 <= DCCF_PROF_EVENTS) {
    (void)hipEventRecord(c->ev[c->ev_used + 1], st);
    c->ev_slot[c->ev_used >> 1] = slot;
    c->ev_used += 2;
  }
}
int dccf_ws_ensure(dccf_ctx* ctx, size_t bytes);
int dccf_step_ensure(dccf_ctx* ctx, int64_t max_rows);

// What k_prep needs to mark the rows a training step touches before the forward starts (overlapped step).
// flag arrays are the uint8 "touched" bytes viewed as 32-bit words (atomicOr de-duplicates); list entries = (seg << 40) | row.
struct MarkPlan {
  uint32_t* flagU;
  uint32_t* flagV;
  int64_t tagU, tagV;
  int64_t* list;
  int* cnt;
  int* cnt_next;
};

struct dccf_opt_args;      // == dccf_opt_t of include/dccf_hip.h
#define OPT_PHASE_ALL 0
#define OPT_PHASE_UNTOUCHED 1
#define OPT_PHASE_TOUCHED 2
// opt_kernels.hip: one phase of the dense regularised optimizer step described by `o` (dccf_opt_t)
int dccf_opt_phase(const void* o, int phase, const int64_t* list, const int* cnt, int64_t max_rows, hipStream_t st);
struct PrepNext;
// dW of a training step as per-row-split partial sums instead of float atomics into gW (k_bwd writes split r's copy with plain
// stores, the optimizer launch that follows adds the copies in split order while it reads the gradient of W anyway)
struct GwPart {
  const float* part;        // [nsplit][stride] floats; NULL = gW holds the gradient
  int nsplit;
  int64_t stride;           // D (D + F)
  int64_t w_begin, w_end;   // elements of W inside the flat parameter / gradient buffer
};
int dccf_opt_all_prep(const void* o, const PrepNext* pn, hipStream_t st);
// the same with the flags of one segment replaced and only its MARKED rows belonging to the pass (the unmarked ones were
// updated by the pass hosted in the backward launch) — for its first rows_hosted rows; the rest of the segment is ordinary
int dccf_opt_all_prep_to(const void* o, int seg, uint8_t* flags, int64_t rows_hosted, const PrepNext* pn, hipStream_t st);
struct OptJob;
int dccf_opt_job(const void* o, OptJob* out);     // the validated job of a dccf_opt_t (for kernels that host a pass)
// phase 1 (rows whose byte in `flags` is 0) + the next step's slots / marks in the same launch (dccf_dp_overlap)
int dccf_opt_untouched_prep(const void* o, uint8_t* const* flags, const PrepNext* pn, hipStream_t st);
// windowed lazy regularisation (dccf_opt_t.lazy_K > 0): the rows of the running step (X, cand, first global row of the user /
// item segment) are claimed, listed and brought up to step - 1; then the step's optimizer launch
int dccf_lazy_catchup(const void* o, const int64_t* X, const int* cand, int64_t N, int S1, int segU, int segV, hipStream_t st);
// nslots = N (S + 2) of the catch-up; win_from >= 0: the window role starts at that global row (the rows of the step's window
// before it were advanced by workgroups hosted in the forward launch, LazyHost)
int dccf_lazy_step(const void* o, const PrepNext* pn, int64_t nslots, hipStream_t st, const GwPart* gp = nullptr, int64_t win_from = -1);
// claims somebody made for `step` (pn.cu_blocks of the previous launch) for a batch that did not come: forgotten
int dccf_lazy_reset_claims(const void* o, hipStream_t st);
// replicated multi-GPU path (dp_kernels.hip): rows flagged in (flags0, flags1) of segments (seg0, seg1) claimed + caught up;
// phase 1 = window + marks + next-step preparation; everything brought to step - 1
int dccf_lazy_catchup_flags(const void* o, const uint8_t* flags0, const uint8_t* flags1, int seg0, int seg1, hipStream_t st);
// the rows of a batch of (user, item) pairs X [N][2]: slot j < N = row X[2 j] of segment seg_u, slot N + j = row X[2 j + 1] of
// segment seg_v; *zero (may be NULL) is set to 0 by the same launch (mf_train_step: the loss accumulator)
int dccf_lazy_catchup_pairs(const void* o, const int64_t* X, int64_t N, int seg_u, int seg_v, float* zero, hipStream_t st);
int dccf_lazy_phase1(const void* o, const PrepNext* pn, hipStream_t st);
int dccf_lazy_flush_to_prev(const void* o, hipStream_t st);
// dccf_kernels.hip: workspace pointers / key of the step (X_next, N, step_next) into pn (w_begin / w_end / blocks and the
// dp fields are the caller's); and the record that makes the next matching call skip k_prep
int dccf_prep_next_fill(dccf_ctx* ctx, const dccf_model_t* M, int64_t N, const int64_t* X_next, uint64_t seed, uint64_t step_next,
                        PrepNext* pn);
void dccf_prep_next_commit(dccf_ctx* ctx, const dccf_model_t* M, int64_t N, const void* X_next, uint64_t seed, uint64_t step_next);
bool dccf_prep_matches(const dccf_ctx* ctx, const dccf_model_t* M, const dccf_rand_t* rnd, const void* X, int64_t N);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------- Philox4x32-10
// Counter-based RNG of the fused random draws (oracle/philox.py restates it bit for bit).
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u
#define STREAM_CAND 1u
#define STREAM_NOISE 2u
#define STREAM_DROP 3u
#define STREAM_NEG 4u
#define STREAM_EVALNEG 6u
#define STREAM_XI 7u

struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  return u32x4{c0, c1, c2, c3};
}

struct rng_key {
  uint32_t k0, k1, s0, s1;  // key words, step words
};
static inline rng_key make_key(uint64_t seed, uint32_t stream, uint64_t step) {
  rng_key k;
  k.k0 = (uint32_t)seed;
  k.k1 = (uint32_t)(seed >> 32) ^ stream;
  k.s0 = (uint32_t)step;
  k.s1 = (uint32_t)(step >> 32);
  return k;
}

// step words + k (the device-side step counter of a replayed graph)
__device__ __forceinline__ rng_key key_plus(rng_key k, int64_t add) {
  const uint64_t st = (((uint64_t)k.s1 << 32) | k.s0) + (uint64_t)add;
  k.s0 = (uint32_t)st;
  k.s1 = (uint32_t)(st >> 32);
  return k;
}
struct StepRef {          // how a kernel finds "its" batch and stream counter
  const int64_t* k_dev;
  int64_t x_stride, x_steps;
};
__device__ __forceinline__ int64_t step_k(const StepRef& r) { return r.k_dev ? *r.k_dev : 0; }
__device__ __forceinline__ const int64_t* step_X(const StepRef& r, const int64_t* X, int64_t k) {
  return r.k_dev ? X + (k % r.x_steps) * r.x_stride : X;
}

// 23-bit uniform strictly inside (0,1), exact in fp32.
// ((x >> 9) + 0.5) * 2^-23: the integer, its half and the product are all exact in fp32, so the single fma below gives the
// same bits as the add and the multiply it replaces (one instruction less per uniform in kernels bound by instruction issue)
__device__ __forceinline__ float u01(uint32_t x) { return fmaf((float)(x >> 9), 0x1p-23f, 0x1p-24f); }

// Box-Muller on the hardware transcendentals: v_log_f32 is log2, v_sin/v_cos take revolutions (sin(2*pi*x)).
// nscale = -2 * ln(2) * std^2, so r = std * sqrt(-2 ln u1).
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float nscale, float& z0, float& z1) {
  const float u1 = u01(xa), u2 = u01(xb);
  const float r = __builtin_amdgcn_sqrtf(nscale * __builtin_amdgcn_logf(u1));
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// the 4 normals of noise words (l, c1): f = 128*(c1/32) + (c1%32) + 32*o, o = 0..3
// Same operations as two box_muller() calls, written on 2-vectors so that the uniform conversions, the scalings of the
// logarithms and the final products issue as packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two IEEE operations
// per issue slot — the MFMA kernels are bound by instruction issue, profiles/r01_mfma_pmc.md).  Bit-identical results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void noise4(uint32_t l, uint32_t c1, const rng_key& k, float nscale, float out[4]) {
  const u32x4 r = philox4x32_10(l, c1, k.s0, k.s1, k.k0, k.k1);
  const f32x2 ia = {(float)(r.x >> 9), (float)(r.z >> 9)}, ib = {(float)(r.y >> 9), (float)(r.w >> 9)};
  const f32x2 sc = {0x1p-23f, 0x1p-23f}, hf = {0x1p-24f, 0x1p-24f};
  const f32x2 u1 = __builtin_elementwise_fma(ia, sc, hf), u2 = __builtin_elementwise_fma(ib, sc, hf);
  f32x2 lg = {__builtin_amdgcn_logf(u1.x), __builtin_amdgcn_logf(u1.y)};
  lg = lg * nscale;
  const f32x2 rr = {__builtin_amdgcn_sqrtf(lg.x), __builtin_amdgcn_sqrtf(lg.y)};
  const f32x2 cs = {__builtin_amdgcn_cosf(u2.x), __builtin_amdgcn_cosf(u2.y)};
  const f32x2 sn = {__builtin_amdgcn_sinf(u2.x), __builtin_amdgcn_sinf(u2.y)};
  const f32x2 z0 = rr * cs, z1 = rr * sn;
  out[0] = z0.x; out[1] = z1.x; out[2] = z0.y; out[3] = z1.y;
}

__device__ __forceinline__ uint32_t pick4(const u32x4& r, int i) {
  return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}

static inline uint32_t drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------- touched-row marking
// Sets the row's "touched" byte; the first setter appends the row to the step's list.  Divergence-safe: the ballot is over
// the lanes that reached this call.
__device__ __forceinline__ void mark_row(uint32_t* flags, int64_t row, int64_t tag, const MarkPlan& mp) {
  const uint32_t mask = 1u << (8 * (int)(row & 3));
  const uint32_t old = atomicOr(flags + (row >> 2), mask);
  const bool first = (old & mask) == 0;
  const uint64_t b = __ballot(first);
  if (b == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((unsigned long long)b) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(mp.cnt, __popcll(b));
  base = __shfl(base, leader, 64);
  if (first) mp.list[base + __popcll(b & ((1ull << lane) - 1))] = tag | row;
}

// ---------------------------------------------------------------------------------------------- per-step preparation
// Expo[u, i] (models/DCCF.py:98), dense or recomputed from the IPSBiasedMF factors that produced it (IPSBiasedMF.py:32-40)
__device__ __forceinline__ float expo_at(const dccf_model_t& M, int64_t u, int64_t i) {
  if (M.expo) return M.expo[u * M.item_num + i];
  float acc = 0.f;
  const float* __restrict__ pu = M.ipsP + u * M.ipsD;
  const float* __restrict__ qi = M.ipsQ + i * M.ipsD;
  if (((M.ipsD & 3) | (int)((uintptr_t)pu & 15) | (int)((uintptr_t)qi & 15)) == 0) {
    // 16-byte rows: a quarter of the loads, the same fma chain (one thread walks a row pair alone — the loads are its time)
    for (int k = 0; k < M.ipsD; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(pu + k), b = *reinterpret_cast<const float4*>(qi + k);
      acc = fmaf(a.x, b.x, acc);
      acc = fmaf(a.y, b.y, acc);
      acc = fmaf(a.z, b.z, acc);
      acc = fmaf(a.w, b.w, acc);
    }
  } else {
    for (int k = 0; k < M.ipsD; ++k) acc = fmaf(pu[k], qi[k], acc);
  }
  acc = acc + M.ipsBu[u] + M.ipsBi[i] + M.ipsB0;
  return acc / fmaxf(M.ipsProp[i], M.ipsM);
}


// Slot j = (n, s) of a batch: cand[n][0] = the true item, cand[n][s] = injected or Philox candidate (models/DCCF.py:72-74),
// eg[n][s] = Expo[u(n), cand[n][s]].  Returns the item.
__device__ __forceinline__ int64_t prep_cand(const dccf_model_t& M, const int64_t* X, const int64_t* sample_item, int* cand,
                                             float* eg, int64_t j, int S, int64_t item_num, int fused, const rng_key& key) {
  const int64_t n = j / (S + 1);
  const int s = (int)(j % (S + 1));
  int64_t it;
  if (s == 0) {
    it = X[2 * n + 1];
  } else if (!fused) {
    it = sample_item[n * S + (s - 1)];
  } else {
    const u32x4 r = philox4x32_10((uint32_t)n, (uint32_t)((s - 1) >> 2), key.s0, key.s1, key.k0, key.k1);
    it = (int64_t)(((uint64_t)pick4(r, (s - 1) & 3) * (uint64_t)item_num) >> 32);
  }
  cand[j] = (int)it;
  eg[j] = M.expo_gathered ? M.expo_gathered[j] : expo_at(M, X[2 * n], it);
  return it;
}

// What the optimizer launch of step t prepares for step t + 1 (dccf_train_step with X_next): the candidate / exposure
// slots, the zeroed accumulators, and — while it writes the new W anyway — the transposed copy W^T the forward reads.
// The next step then starts with its forward kernel; k_prep runs only when nothing (or something else) was prepared.
struct PrepNext {
  dccf_model_t M;
  float* WT;
  int* cand;
  float* eg;
  float* m;
  const int64_t* X;
  int64_t N, Lm, w_begin, w_end;     // [w_begin, w_end): elements of W inside the flat parameter buffer
  int S, DP, blocks;
  rng_key key;
  // replicated data-parallel path (dp_kernels.hip) only:
  MarkPlan lm;               // list != NULL: this rank's rows of the next step as a de-duplicated list (-> k_dp_export_list)
  const int64_t* X_all;      // != NULL: the replicated schedule [G][N][2] of the next step; every rank's rows get a byte
  uint8_t* gU;               //          in gU / gV ("touched by ANY rank": the rows phase 1 of the next step skips)
  uint8_t* gV;
  int G;
  rng_key gkey0;             // STREAM_CAND key of rank 0's next step; rank r draws with key + r
  uint8_t* markV;            // != NULL: byte per item row the next step touches (dccf_train_step, hosted item table)
  // != NULL: the import tables of the next step, built here instead of by k_dp_scatter_ids after the all-gather:
  // nmask[gid] bit r = rank r touches the row; nwhere[r * R + gid] = the slot of rank r's buffer that will hold it = the
  // row's FIRST position in rank r's canonical (n, s) order (atomicMin: the same on every rank)
  uint32_t* nmask;
  int* nwhere;
  int64_t R, offU, offV;     // rows of all segments; first global row index of the user / item segment
  // windowed lazy regularisation (k_lazy_opt): cu_blocks > 0 = that many workgroups claim, list and catch up the rows of the
  // NEXT step (its users X[.][0], true items and Philox candidates, recomputed from `key`) inside this step's optimizer launch
  int cu_blocks, cu_segU, cu_segV;
};

__device__ __forceinline__ void prep_next_slots(const PrepNext& pn, int64_t tid, int64_t nthreads) {
  const int S1 = pn.S + 1;
  const int64_t NS = pn.N * S1;
  const int64_t total = NS + pn.Lm;
  for (int64_t i = tid; i < total; i += nthreads) {
    if (i < NS) {
      const int64_t it = prep_cand(pn.M, pn.X, nullptr, pn.cand, pn.eg, i, pn.S, pn.M.item_num, 1, pn.key);
      if (pn.markV) pn.markV[it] = 1;
      if (pn.lm.list) {
        mark_row(pn.lm.flagV, it, pn.lm.tagV, pn.lm);
        if (i % S1 == 0) mark_row(pn.lm.flagU, pn.X[2 * (i / S1)], pn.lm.tagU, pn.lm);
      }
    } else {
      pn.m[i - NS] = 0.f;
    }
  }
  if (pn.X_all) {            // plain byte stores: nobody needs these rows as a list
    const int64_t all = (int64_t)pn.G * NS;
    for (int64_t i = tid; i < all; i += nthreads) {
      const int r = (int)(i / NS);
      const int64_t j = i % NS, n = j / S1;
      const int s = (int)(j % S1);
      const int64_t* X = pn.X_all + (int64_t)r * pn.N * 2;
      int64_t it;
      if (s == 0) {
        it = X[2 * n + 1];
        const int64_t u = X[2 * n];
        pn.gU[u] = 1;
        if (pn.nmask) {
          atomicMin(&pn.nwhere[(int64_t)r * pn.R + pn.offU + u], (int)(NS + n));
          atomicOr(&pn.nmask[pn.offU + u], 1u << r);
        }
      } else {
        const rng_key key = key_plus(pn.gkey0, r);
        const u32x4 rr = philox4x32_10((uint32_t)n, (uint32_t)((s - 1) >> 2), key.s0, key.s1, key.k0, key.k1);
        it = (int64_t)(((uint64_t)pick4(rr, (s - 1) & 3) * (uint64_t)pn.M.item_num) >> 32);
      }
      pn.gV[it] = 1;
      if (pn.nmask) {
        atomicMin(&pn.nwhere[(int64_t)r * pn.R + pn.offV + it], (int)j);
        atomicOr(&pn.nmask[pn.offV + it], 1u << r);
      }
    }
  }
}

// element e of the flat parameter buffer lies in W = [D][D+F]: mirror its new value into W^T [k][DP]
__device__ __forceinline__ void prep_next_wt(const PrepNext& pn, int64_t e, float v) {
  const int64_t idx = e - pn.w_begin;
  const int KF = pn.M.D + pn.M.F;
  const int d = (int)(idx / KF), k = (int)(idx % KF);
  pn.WT[(int64_t)k * pn.DP + d] = v;
}
