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
  int zero_grad;
  const int64_t* k_dev;  // graph-replayable form: step = step0 + *k_dev, bias corrections computed here
  int64_t step0;
};

__device__ __forceinline__ void opt_resolve(OptArgs& a) {
  if (a.k_dev) {
    const double t = (double)(a.step0 + *a.k_dev);
    a.step_size_neg = (float)(-((double)a.lr / (1.0 - pow(0.9, t))));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(0.999, t));
  }
}

template <int KIND>
__device__ __forceinline__ void opt_elem(float& p, float& g, float& s1, float& s2, const OptArgs& a) {
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
        opt_elem<KIND>(pv[u].x, gv.x, av[u].x, bv[u].x, j.a);
        opt_elem<KIND>(pv[u].y, gv.y, av[u].y, bv[u].y, j.a);
        opt_elem<KIND>(pv[u].z, gv.z, av[u].z, bv[u].z, j.a);
        opt_elem<KIND>(pv[u].w, gv.w, av[u].w, bv[u].w, j.a);
        reinterpret_cast<float4*>(p)[i] = pv[u];
        if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[i] = av[u];
        if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[i] = bv[u];
      }
      pv[u] = pn[u]; av[u] = an[u]; bv[u] = bn[u]; fv[u] = fn[u];
    }
  }
#undef OPT_UP_LOAD
}

// opt_kernels.hip: validation + bias corrections (host side) of one optimizer step
int opt_make_job(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd, float l2, float clip,
                 int64_t step, const int64_t* k_dev, int32_t nseg, const int64_t* seg_begin, const int64_t* seg_rows,
                 const int32_t* seg_width, uint8_t* const* seg_flags, OptJob* out);
