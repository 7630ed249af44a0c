// opt_device.hpp — element arithmetic and argument blocks of the dense regularised optimizer step (opt_kernels.hip):
// src/runners/BaseRunner.py:92-100,181-187 (+ l2, clip_grad_value_(50), torch.optim step).
#pragma once
#include "common.hpp"

// Op order mirrors torch 2.10's single-tensor CPU paths so that results agree with the oracle to rounding
// (contraction is disabled for this file's arithmetic via explicit __fmul_rn/__fadd_rn where it matters).
struct OptArgs {
  float lr, wd, l2, clip;
  float step_size_neg;   // Adam: -(lr / (1 - beta1^t))
  float bc2_sqrt;        // Adam: sqrt(1 - beta2^t)
  float bc2_rsqrt;       // Adam: RN(1 / bc2_sqrt) — what div_by_uniform needs (see there)
  int zero_grad;
  const int64_t* k_dev;  // graph-replayable form: step = step0 + *k_dev, bias corrections computed here
  int64_t step0;
};

__device__ __forceinline__ void opt_resolve(OptArgs& a) {
  if (a.k_dev) {
    const double t = (double)(a.step0 + *a.k_dev);
    a.step_size_neg = (float)(-((double)a.lr / (1.0 - pow(0.9, t))));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(0.999, t));
    a.bc2_rsqrt = (float)(1.0 / (double)a.bc2_sqrt);
  }
}

// ---- Adam's two divisions and its square root, without the range scaling of the compiler's IEEE expansions -----------
// `__fdiv_rn` is v_div_scale x2 + v_rcp + 7 fma-class ops + v_div_fixup (11 instructions, none of them packed), `__fsqrt_rn` is
// v_sqrt_f32 between two conditional v_ldexp (6 instructions): 28 of the 47 vector instructions of one Adam element-step —
// the windowed lazy replay (16.4 M element-steps per launch at the bench shape) is bound by exactly these.  The operands of
// Adam's divisions cannot reach the ranges the scaling is for, so the same Newton steps WITHOUT it give the same bits
// (tests/test_hip_parity.py::test_adam_element_function_equals_ieee: bit-equal to opt_elem_ieee) and pack two lanes of
// work into v_pk_fma_f32.  Every optimizer kernel (dense, row-aware, hosted, import, lazy) calls the ONE opt_elem below, so
// dense == lazy bit for bit holds by construction.
//
// x / c, c wave-uniform with rc = RN(1 / c) known (host, in double): q0 = x rc is within 2 ulp; one residual correction makes
// it faithful, the second rounds it correctly (Markstein: q faithful and y = RN(1/c) => RN(q + y (x - c q)) = RN(x / c)) as
// long as the residuals are exact, i.e. nothing under- or overflows: x = sqrt(v) is 0 or in [2^-75, 2^64], c in [0.03, 1].
__device__ __forceinline__ float div_by_uniform(float x, float c, float rc) {
  float q = __fmul_rn(x, rc);
  float r = fmaf(-c, q, x);
  q = fmaf(r, rc, q);
  r = fmaf(-c, q, x);
  return fmaf(r, rc, q);
}
// n / d for d in [1e-8, 2^70] (Adam's denominator: something non-negative + 1e-8) — the compiler's own sequence (reciprocal,
// one Newton step on it, quotient, two residual corrections) minus v_div_scale / v_div_fmas' scaling / v_div_fixup.  Equal to
// __fdiv_rn whenever |n| is 0 or in [2^-100, 2^100]; a quotient of -0 comes out as +0 (added to a parameter that is never -0).
__device__ __forceinline__ float div_normal(float n, float d) {
  float y = __builtin_amdgcn_rcpf(d);
  const float e = fmaf(-d, y, 1.0f);
  y = fmaf(e, y, y);
  float q = __fmul_rn(n, y);
  float r = fmaf(-d, q, n);
  q = fmaf(r, y, q);
  r = fmaf(-d, q, n);
  return fmaf(r, y, q);
}
// sqrt(v) as the bare v_sqrt_f32 (what __fsqrt_rn lowers to, between scalings by 2^32 / 2^-16 for v < 2^-96): for those v the
// instruction sees the same significand and exponent parity, and below 2^-126 whatever it returns (<= 2^-63) vanishes in
// "+ 1e-8" (ulp 2^-50 after the division by c >= 0.03).
__device__ __forceinline__ float sqrt_bare(float v) { return __builtin_amdgcn_sqrtf(v); }

// everything after the clip: the optimizer's coupled weight decay and its step
template <int KIND>
__device__ __forceinline__ void opt_elem_rest(float& p, float gt, float& s1, float& s2, const OptArgs& a) {
  gt = __fadd_rn(gt, __fmul_rn(a.wd, p));
  if (KIND == DCCF_OPT_GD) {
    p = __fadd_rn(p, __fmul_rn(-a.lr, gt));
  } else if (KIND == DCCF_OPT_ADAGRAD) {
    s1 = __fadd_rn(s1, __fmul_rn(gt, gt));
    const float sd = __fadd_rn(__fsqrt_rn(s1), 1e-10f);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-a.lr, gt), sd));
  } else {
    s1 = __fadd_rn(s1, __fmul_rn(0.1f, __fsub_rn(gt, s1)));                       // lerp_(g, 1-beta1), weight < 0.5
    s2 = __fadd_rn(__fmul_rn(s2, 0.999f), __fmul_rn(__fmul_rn(0.001f, gt), gt));  // mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = __fadd_rn(div_by_uniform(sqrt_bare(s2), a.bc2_sqrt, a.bc2_rsqrt), 1e-8f);
    p = __fadd_rn(p, div_normal(__fmul_rn(a.step_size_neg, s1), denom));
  }
}

template <int KIND>
__device__ __forceinline__ void opt_elem(float& p, float& g, float& s1, float& s2, const OptArgs& a) {
  // explicit l2 term of the loss, then the clip, then the optimizer's coupled weight decay
  float gt = __fadd_rn(g, __fmul_rn(a.l2, __fmul_rn(2.0f, p)));
  gt = __builtin_amdgcn_fmed3f(gt, -a.clip, a.clip);      // clamp(gt, -clip, clip), clip >= 0 (checked on the host): one instruction
  opt_elem_rest<KIND>(p, gt, s1, s2, a);
  if (a.zero_grad) g = 0.f;
}

// Four elements at once: the same operations in the same order per element (hence the same bits as four opt_elem calls); the
// head g + l2 (2 p) is written on 2-vectors because the compiler packs everything AFTER the clip into v_pk_*_f32 on its own but
// leaves these three operations scalar (12 of the 83 vector instructions of four Adam element-steps).
typedef float opt_f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__device__ __forceinline__ void opt_elem4(float4& p, float4& g, float4& s1, float4& s2, const OptArgs& a) {
  const opt_f2 two = {2.0f, 2.0f}, l2v = {a.l2, a.l2};
  const opt_f2 plo = {p.x, p.y}, phi = {p.z, p.w}, glo = {g.x, g.y}, ghi = {g.z, g.w};
  const opt_f2 tlo = glo + l2v * (two * plo), thi = ghi + l2v * (two * phi);
  const float gx = __builtin_amdgcn_fmed3f(tlo.x, -a.clip, a.clip), gy = __builtin_amdgcn_fmed3f(tlo.y, -a.clip, a.clip);
  const float gz = __builtin_amdgcn_fmed3f(thi.x, -a.clip, a.clip), gw = __builtin_amdgcn_fmed3f(thi.y, -a.clip, a.clip);
  opt_elem_rest<KIND>(p.x, gx, s1.x, s2.x, a);
  opt_elem_rest<KIND>(p.y, gy, s1.y, s2.y, a);
  opt_elem_rest<KIND>(p.z, gz, s1.z, s2.z, a);
  opt_elem_rest<KIND>(p.w, gw, s1.w, s2.w, a);
  if (a.zero_grad) g = make_float4(0.f, 0.f, 0.f, 0.f);
}

// The same element step on the compiler's IEEE division / square root: what opt_elem is checked against (dccf_debug_opt_elem).
template <int KIND>
__device__ __forceinline__ void opt_elem_ieee(float& p, float& g, float& s1, float& s2, const OptArgs& a) {
  // explicit l2 term of the loss, then the clip, then the optimizer's coupled weight decay
  float gt = __fadd_rn(g, __fmul_rn(a.l2, __fmul_rn(2.0f, p)));
  gt = fminf(fmaxf(gt, -a.clip), a.clip);
  gt = __fadd_rn(gt, __fmul_rn(a.wd, p));
  if (KIND == DCCF_OPT_GD) {
    p = __fadd_rn(p, __fmul_rn(-a.lr, gt));
  } else if (KIND == DCCF_OPT_ADAGRAD) {
    s1 = __fadd_rn(s1, __fmul_rn(gt, gt));
    const float sd = __fadd_rn(__fsqrt_rn(s1), 1e-10f);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-a.lr, gt), sd));
  } else {
    s1 = __fadd_rn(s1, __fmul_rn(0.1f, __fsub_rn(gt, s1)));                       // lerp_(g, 1-beta1), weight < 0.5
    s2 = __fadd_rn(__fmul_rn(s2, 0.999f), __fmul_rn(__fmul_rn(0.001f, gt), gt));  // mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(s2), a.bc2_sqrt), 1e-8f);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(a.step_size_neg, s1), denom));
  }
  if (a.zero_grad) g = 0.f;
}


// Row segments of the flat buffer: one "touched" byte per row (see dccf_dense_opt_step_rows).
struct RowSegs {
  int64_t begin[4], end[4];
  int width[4];
  uint8_t* flags[4];
  int n;
  int to_mask;      // bit q: only the MARKED rows of segment q belong to this pass (the others were updated elsewhere)
};

// Everything a kernel needs to run (part of) one optimizer step.
struct OptJob {
  float *p, *g, *s1, *s2;
  int64_t n;
  int kind;
  OptArgs a;
  RowSegs sg;
};

// The untouched-row pass (phase 1 of the two-phase step) as a device function, for kernels that host it as extra
// workgroups (k_bwd): float4 slots of the row segments whose row byte is 0 get the gradient-free update; everything else
// (marked rows, the dense tail) is left to the touched-row pass.  UN slots in flight per thread: the hosting workgroups
// are few (4 waves per CU), so the memory-level parallelism has to come from the lane.
template <int KIND, int UN>
__device__ __forceinline__ void opt_untouched_pass(const OptJob& j, int64_t bid, int64_t nblk, int nthreads) {
  float* __restrict__ p = j.p;
  float* __restrict__ s1 = j.s1;
  float* __restrict__ s2 = j.s2;
  const RowSegs& sg = j.sg;
  const int64_t n4 = j.n / 4;
  const int64_t stride = nblk * nthreads;
  int wsh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wsh[q] = q < sg.n ? 31 - __clz(sg.width[q]) : 0;
  // software-pipelined: the loads of the next UN slots are in flight while the current ones are computed and stored (the
  // hosting kernel gives this pass one wave per SIMD, so nothing else overlaps its memory latency)
  float4 pv[UN], av[UN], bv[UN], pn[UN], an[UN], bn[UN];
  uint8_t fv[UN], fn[UN];
#define OPT_UP_LOAD(I0, PV, AV, BV, FV)                                                                    \
  _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                                         \
    const int64_t i_ = (I0) + u * stride;                                                                  \
    const int64_t ic_ = min(i_, n4 - 1);                                                                   \
    const int64_t e_ = ic_ * 4;                                                                            \
    const uint8_t* fl_ = nullptr;                                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
      if (q < sg.n && e_ >= sg.begin[q] && e_ < sg.end[q]) fl_ = sg.flags[q] + ((e_ - sg.begin[q]) >> wsh[q]); \
    PV[u] = reinterpret_cast<const float4*>(p)[ic_];                                                       \
    AV[u] = BV[u] = make_float4(0, 0, 0, 0);                                                               \
    if (KIND != DCCF_OPT_GD) AV[u] = reinterpret_cast<const float4*>(s1)[ic_];                             \
    if (KIND == DCCF_OPT_ADAM) BV[u] = reinterpret_cast<const float4*>(s2)[ic_];                           \
    FV[u] = (i_ < n4 && fl_ != nullptr) ? *fl_ : (uint8_t)1;      /* 1 = not this pass's slot */           \
  }
  int64_t i0 = bid * nthreads + threadIdx.x;
  if (i0 >= n4) return;
  OPT_UP_LOAD(i0, pv, av, bv, fv)
  for (; i0 < n4; i0 += stride * UN) {
    OPT_UP_LOAD(i0 + stride * UN, pn, an, bn, fn)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (fv[u] == 0) {
        const int64_t i = i0 + u * stride;
        float4 gv = make_float4(0, 0, 0, 0);
        opt_elem4<KIND>(pv[u], gv, av[u], bv[u], j.a);
        reinterpret_cast<float4*>(p)[i] = pv[u];
        if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[i] = av[u];
        if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[i] = bv[u];
      }
      pv[u] = pn[u]; av[u] = an[u]; bv[u] = bn[u]; fv[u] = fn[u];
    }
  }
#undef OPT_UP_LOAD
}

// ---- windowed lazy regularisation: see dccf_opt_t.lazy_* (include/dccf_hip.h).  Rows are numbered globally: segment q's row r is grow = row_off[q] + r.
struct LazyArgs {
  int K, nscal;
  int64_t t0, step;          // scal[4 (s - t0) ..+2] = {step_size_neg, bc2_sqrt, bc2_rsqrt} of step s
  int* last;
  int* claim;                // two arrays of R entries by step parity (lazy_claim_of): launch t reads the claims of step t while its
  int* list;                 // catch-up role writes those of step t + 1; likewise two lists of list_cap slots (lazy_list_of)
  int64_t R, list_cap;       // rows of all segments; slots of ONE list
  int* cnt;
  int pend_slot;             // which record of cnt this launch READS: (step - 1) & 1 for a step, step & 1 for a flush
  const float* scal;
  int64_t row_off[4];        // first global row of segment q
  int64_t rows[4];
};

__device__ __forceinline__ int* lazy_claim_of(const LazyArgs& z, int t) { return z.claim + (int64_t)(t & 1) * z.R; }
__device__ __forceinline__ int* lazy_list_of(const LazyArgs& z, int t) { return z.list + (int64_t)(t & 1) * z.list_cap; }

// The window of the PREVIOUS lazy step is marked one launch late (its rows' `last` entries are written by the next optimizer
// launch instead of a launch of their own): until then a row of that window that was behind counts as being at that step.
// z.cnt holds two records {step, w0 lo, w0 hi, w1 lo, w1 hi} by step parity: launch t reads (t - 1) & 1 and writes t & 1.
struct LazyPend {
  int t;
  int64_t w0, w1;
};
__device__ __forceinline__ LazyPend lazy_pend_read(const LazyArgs& z) {
  const int* r = z.cnt + 5 * z.pend_slot;
  LazyPend q;
  q.t = r[0];
  q.w0 = (int64_t)(uint32_t)r[1] | ((int64_t)r[2] << 32);
  q.w1 = (int64_t)(uint32_t)r[3] | ((int64_t)r[4] << 32);
  return q;
}
__device__ __forceinline__ int lazy_eff_last(const LazyArgs& z, const LazyPend& q, int64_t grow) {
  const int f = z.last[grow];
  return (grow >= q.w0 && grow < q.w1 && f < q.t) ? q.t : f;
}

// lazy_eff_last with the invariant enforced: no row is more than K steps behind the step being launched.  A row that is
// (a caller skipped a step number, or cleared `last` behind the library's back) is replayed from t - K only — never an index in
// front of the step-scalar table — and the error flag z.cnt[15] is raised for the next flush / check to report.
__device__ __forceinline__ int lazy_from(const LazyArgs& z, const LazyPend& q, int64_t grow, int t) {
  const int f = lazy_eff_last(z, q, grow);
  if (f < t - z.K) {
    z.cnt[15] = 1;
    return t - z.K;
  }
  return f;
}

// `steps` optimizer steps with a zero loss gradient on one element, starting after step `from`: exactly what the dense pass
// would have done to it launch by launch
template <int KIND>
__device__ __forceinline__ void lazy_replay(float& p, float& s1, float& s2, OptArgs a, const LazyArgs& z, int from, int to) {
  for (int s = from + 1; s <= to; ++s) {
    if (KIND == DCCF_OPT_ADAM) {
      const float4 sc = reinterpret_cast<const float4*>(z.scal)[s - z.t0];
      a.step_size_neg = sc.x; a.bc2_sqrt = sc.y; a.bc2_rsqrt = sc.z;
    }
    float g0 = 0.f;
    opt_elem<KIND>(p, g0, s1, s2, a);
  }
}
template <int KIND>
__device__ __forceinline__ void lazy_replay4(float4& p, float4& s1, float4& s2, OptArgs a, const LazyArgs& z, int from, int to) {
  for (int s = from + 1; s <= to; ++s) {
    if (KIND == DCCF_OPT_ADAM) {
      const float4 sc = reinterpret_cast<const float4*>(z.scal)[s - z.t0];
      a.step_size_neg = sc.x; a.bc2_sqrt = sc.y; a.bc2_rsqrt = sc.z;
    }
    float4 g0 = make_float4(0, 0, 0, 0);
    opt_elem4<KIND>(p, g0, s1, s2, a);
  }
}


// lazy_replay4 with a WAVE-UNIFORM loop counter: `lo` <= every lane's `from` (lanes with nothing to do pass from = to).  The
// step scalars are then read by scalar loads the compiler can issue ahead, not by a vector load per lane and step inside the
// dependent chain; a lane joins at its own first step.
__device__ __forceinline__ int wave_min_int(int v) {          // (every lane of the wave must call it)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return __builtin_amdgcn_readfirstlane(v);
}
template <int KIND>
__device__ __forceinline__ void lazy_replay4_u(float4& p, float4& s1, float4& s2, OptArgs a, const float4* sct, int sct_base,
                                               int from, int to, int lo) {
  for (int s = lo + 1; s <= to; ++s) {
    if (KIND == DCCF_OPT_ADAM) {
      const float4 sc = sct[s - sct_base];          // (sct: the launch's copy of the step scalars in LDS, or the table itself)
      a.step_size_neg = sc.x; a.bc2_sqrt = sc.y; a.bc2_rsqrt = sc.z;
    }
    if (s > from) {
      float4 g0 = make_float4(0, 0, 0, 0);
      opt_elem4<KIND>(p, g0, s1, s2, a);
    }
  }
}

// The window of step t: the float4 slots of the rows [win0, win1) of the global row space are advanced to step t (rows this
// step uses — claim == t — excepted: the step's list updates them with their gradient).  All lanes busy: rows are contiguous.
// Runs as workgroups bid of nblk — of the optimizer launch, or hosted in the backward launch, whose role waves leave most of
// the vector ALU's issue slots empty at small batches.
template <int KIND>
__device__ __forceinline__ void lazy_window_pass(float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2,
                                                 const OptArgs& a, const RowSegs& sg, const LazyArgs& z, int64_t win0, int64_t win1,
                                                 int flush, int64_t bid, int64_t nblk, int nthreads, const float4* sct, int sct_base) {
  const int t = (int)z.step;
  const LazyPend pend = lazy_pend_read(z);
  const int* __restrict__ claim = lazy_claim_of(z, t);
  for (int q = 0; q < sg.n; ++q) {
    const int64_t r0 = max(win0, z.row_off[q]) - z.row_off[q], r1 = min(win1, z.row_off[q] + z.rows[q]) - z.row_off[q];
    if (r1 <= r0) continue;
    const int w4 = sg.width[q] >> 2;
    const int64_t nslot = (r1 - r0) * w4, stride = nblk * nthreads;
    // Software-pipelined: a slot's row state (`last`, claim) and its p / s1 / s2 are fetched one iteration AHEAD, all at once and
    // whether or not the slot turns out to need the replay (nearly all do: only the step's own rows do not), so the replay of slot k
    // (K steps x ~45 vector instructions) runs while slot k + 1's loads are in flight.  Hosted in the forward launch a workgroup has 2
    // waves per SIMD and a handful of slots per thread: fetched one after the other (state, then data, then the replay) the loop was
    // two round trips per slot and latency-bound.
    struct Slot {
      int64_t grow, i;
      int last, claim;
      bool live;
      float4 pv, av, bv;
    };
    // slot -> (row, float4 of the row): a shift for the usual widths, a 32-bit division otherwise (a window has far fewer than 2^32 slots; the
    // compiler's 64-bit division is ~60 vector instructions per slot on the pipe this loop is bound by)
    const bool small = nslot < (int64_t)4294967295LL;
    const int w4sh = (w4 & (w4 - 1)) == 0 ? 31 - __clz(w4) : -1;
    auto fetch = [&](int64_t x0, Slot& f) {
      const int64_t x = x0 + threadIdx.x;
      f.live = x < nslot;
      const int64_t xc = f.live ? x : nslot - 1;
      int64_t rw;
      if (small) {
        const uint32_t x32 = (uint32_t)xc;
        rw = w4sh >= 0 ? (int64_t)(x32 >> w4sh) : (int64_t)(x32 / (uint32_t)w4);
      } else {
        rw = xc / w4;
      }
      const int64_t row = r0 + rw;
      f.grow = z.row_off[q] + row;
      f.i = (sg.begin[q] >> 2) + r0 * w4 + xc;
      f.last = z.last[f.grow];
      f.claim = claim[f.grow];
      f.pv = reinterpret_cast<const float4*>(p)[f.i];
      f.av = make_float4(0, 0, 0, 0);
      f.bv = make_float4(0, 0, 0, 0);
      if (KIND != DCCF_OPT_GD) f.av = reinterpret_cast<const float4*>(s1)[f.i];
      if (KIND == DCCF_OPT_ADAM) f.bv = reinterpret_cast<const float4*>(s2)[f.i];
    };
    int64_t x0 = bid * nthreads;                    // (uniform per workgroup: every lane reaches wave_min_int)
    if (x0 >= nslot) continue;
    Slot cur, nxt;
    fetch(x0, cur);
    for (; x0 < nslot; x0 += stride) {
      const bool more = x0 + stride < nslot;
      if (more) fetch(x0 + stride, nxt);
      // lazy_from on the fetched state: the effective `last` (the previous step's window is marked one launch late), bounded by K
      int from = (cur.grow >= pend.w0 && cur.grow < pend.w1 && cur.last < pend.t) ? pend.t : cur.last;
      if (from < t - z.K) {
        z.cnt[15] = 1;
        from = t - z.K;
      }
      // a row this step touched is the list's business (its last is t - 1 until that wave has updated it: never replay it here)
      const bool need = cur.live && from < t && (flush || cur.claim != t);
      const int fr = need ? from : t;
      const int lo = max(wave_min_int(fr), sct_base);      // (no row is more than K steps behind: the bound is for memory safety)
      if (lo < t && need) {
        lazy_replay4_u<KIND>(cur.pv, cur.av, cur.bv, a, sct, sct_base, fr, t, lo);
        reinterpret_cast<float4*>(p)[cur.i] = cur.pv;
        if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[cur.i] = cur.av;
        if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[cur.i] = cur.bv;
      }
      if (more) cur = nxt;
    }
  }
}

// Part of a lazy step's window hosted as extra workgroups of ANOTHER launch of the same step (the forward: at batch 128 its 176
// tiles leave 80 of the 256 CUs idle): rows [win0, win1) of the step's window are advanced to the step there, and the optimizer
// launch that follows starts its own window role at win1.  The marks stay with the NEXT optimizer launch, as for the whole window.
struct LazyHost {
  float *p, *s1, *s2;
  OptArgs a;
  RowSegs sg;
  LazyArgs z;
  int64_t win0, win1;        // the hosted sub-window (global rows)
  int kind, blocks;          // blocks == 0: nothing hosted
};
// fills everything but blocks / the sub-window's end (win1 = the end of the step's whole window on return)
int dccf_lazy_host_args(const void* o, LazyHost* out);
template <int KIND>
__device__ __forceinline__ void lazy_hosted_window(const LazyHost& h, int bid, int nthreads, const float4* sct, int sct_base) {
  lazy_window_pass<KIND>(h.p, h.s1, h.s2, h.a, h.sg, h.z, h.win0, h.win1, 0, bid, h.blocks, nthreads, sct, sct_base);
}

// opt_kernels.hip: validation + bias corrections (host side) of one optimizer step
int opt_make_job(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd, float l2, float clip,
                 int64_t step, const int64_t* k_dev, int32_t nseg, const int64_t* seg_begin, const int64_t* seg_rows,
                 const int32_t* seg_width, uint8_t* const* seg_flags, OptJob* out);

