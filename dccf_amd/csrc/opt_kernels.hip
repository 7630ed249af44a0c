// opt_kernels.hip — context/workspace, the dense regularised optimizer step and BaseModel.l2.
//
// Replaces, per training step of the reference (src/runners/BaseRunner.py:178-187):
//     loss = out['loss'] + model.l2() * l2          -> grad += l2 * (2 p)           (autograd of BaseModel.l2)
//     clip_grad_value_(model.parameters(), 50)      -> clamp
//     torch.optim.{SGD,Adagrad,Adam}(weight_decay=l2).step()
// Because of the explicit l2 term EVERY parameter element moves EVERY step: the step is one
// streaming pass over p, g and the optimizer state — HBM bound: Adam reads p,g,m,v and writes p,m,v,g(=0),
// 32 B per parameter (28 B without the fused zero_grad).
#include "common.hpp"

// No implicit FMA contraction in this file: a*b+c written as two operations stays two roundings (explicit fmaf() calls
// are still FMAs).  It keeps "fused draws == injected draws" bit for bit and the optimizer in torch's op order.
#pragma clang fp contract(off)

thread_local char g_dccf_err[512] = "";

extern "C" const char* dccf_last_error(void) { return g_dccf_err; }
extern "C" int dccf_abi_version(void) { return DCCF_ABI_VERSION; }

extern "C" int dccf_ctx_create(dccf_ctx** out, int device) {
  ARG_CHECK(out != nullptr, "out is NULL");
  HIP_TRY(hipSetDevice(device));
  dccf_ctx* c = new dccf_ctx();
  c->device = device;
  c->ws = nullptr;
  c->ws_bytes = 0;
  c->prof_on = 0;
  c->ev = nullptr;
  c->ev_slot = nullptr;
  c->ev_used = 0;
  *out = c;
  return 0;
}

// Per-kernel timing: HIP events recorded on the launch stream around every kernel of dccf_predict / dccf_train_fwdbwd.
extern "C" int dccf_profile(dccf_ctx* ctx, int enable) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (enable && !ctx->ev) {
    ctx->ev = new hipEvent_t[DCCF_PROF_EVENTS];
    ctx->ev_slot = new int[DCCF_PROF_EVENTS / 2];
    for (int i = 0; i < DCCF_PROF_EVENTS; ++i) HIP_TRY(hipEventCreate(&ctx->ev[i]));
  }
  ctx->prof_on = enable ? 1 : 0;
  ctx->ev_used = 0;
  return 0;
}

// Adds the elapsed ms and launch counts since the last read into ms[DCCF_PROF_SLOTS] / counts[...] (host arrays).
extern "C" int dccf_profile_read(dccf_ctx* ctx, double* ms, int64_t* counts) {
  ARG_CHECK(ctx && ms && counts, "NULL argument");
  for (int i = 0; i + 1 < ctx->ev_used; i += 2) {
    HIP_TRY(hipEventSynchronize(ctx->ev[i + 1]));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, ctx->ev[i], ctx->ev[i + 1]));
    const int s = ctx->ev_slot[i >> 1];
    ms[s] += (double)t;
    counts[s] += 1;
  }
  ctx->ev_used = 0;
  return 0;
}

extern "C" int dccf_ctx_destroy(dccf_ctx* ctx) {
  if (!ctx) return 0;
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->ev) {
    for (int i = 0; i < DCCF_PROF_EVENTS; ++i) (void)hipEventDestroy(ctx->ev[i]);
    delete[] ctx->ev;
    delete[] ctx->ev_slot;
  }
  delete ctx;
  return 0;
}

int dccf_ws_ensure(dccf_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return 0;
  // grow-only; growing synchronises the device (earlier launches may still use the old slab)
  HIP_TRY(hipDeviceSynchronize());
  if (ctx->ws) HIP_TRY(hipFree(ctx->ws));
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  const size_t want = align_up(bytes + bytes / 8, 1 << 20);
  HIP_TRY(hipMalloc((void**)&ctx->ws, want));
  ctx->ws_bytes = want;
  return 0;
}

// ---------------------------------------------------------------------------------------------- optimizer
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

template <int KIND>
__global__ __launch_bounds__(256) void k_dense_opt(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                   float* __restrict__ s2, int64_t n, OptArgs a) {
  opt_resolve(a);
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 gv = reinterpret_cast<float4*>(g)[i];
    float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
    if (KIND != DCCF_OPT_GD) av = reinterpret_cast<float4*>(s1)[i];
    if (KIND == DCCF_OPT_ADAM) bv = reinterpret_cast<float4*>(s2)[i];
    opt_elem<KIND>(pv.x, gv.x, av.x, bv.x, a);
    opt_elem<KIND>(pv.y, gv.y, av.y, bv.y, a);
    opt_elem<KIND>(pv.z, gv.z, av.z, bv.z, a);
    opt_elem<KIND>(pv.w, gv.w, av.w, bv.w, a);
    reinterpret_cast<float4*>(p)[i] = pv;
    if (a.zero_grad) reinterpret_cast<float4*>(g)[i] = gv;
    if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[i] = av;
    if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[i] = bv;
  }
  // tail
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    float pv = p[i], gv = g[i], av = 0.f, bv = 0.f;
    if (KIND != DCCF_OPT_GD) av = s1[i];
    if (KIND == DCCF_OPT_ADAM) bv = s2[i];
    opt_elem<KIND>(pv, gv, av, bv, a);
    p[i] = pv;
    if (a.zero_grad) g[i] = gv;
    if (KIND != DCCF_OPT_GD) s1[i] = av;
    if (KIND == DCCF_OPT_ADAM) s2[i] = bv;
  }
}

// Row-aware variant: g of a row whose "touched" byte is 0 is all zeros by construction -> not read, not re-zeroed.
// A wave handles 64 consecutive float4 = 256 consecutive floats = whole rows (row widths 16..128 divide 256 and segments
// start on a 256-float boundary), so every lane of a row sees the byte before the row's first lane clears it.
struct RowSegs {
  int64_t begin[4], end[4];
  int width[4];
  uint8_t* flags[4];
  int n;
};

template <int KIND>
__global__ __launch_bounds__(256) void k_dense_opt_rows(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                        float* __restrict__ s2, int64_t n, OptArgs a, RowSegs sg) {
  opt_resolve(a);
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t e = i * 4;
    bool touched = true;
    uint8_t* fl = nullptr;
    bool first = false;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < sg.n && e >= sg.begin[q] && e < sg.end[q]) {
        const int64_t off = e - sg.begin[q];
        fl = sg.flags[q] + off / sg.width[q];
        first = off % sg.width[q] == 0;
      }
    if (fl) touched = *fl != 0;
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 gv = make_float4(0, 0, 0, 0);
    if (touched) gv = reinterpret_cast<float4*>(g)[i];
    float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
    if (KIND != DCCF_OPT_GD) av = reinterpret_cast<float4*>(s1)[i];
    if (KIND == DCCF_OPT_ADAM) bv = reinterpret_cast<float4*>(s2)[i];
    opt_elem<KIND>(pv.x, gv.x, av.x, bv.x, a);
    opt_elem<KIND>(pv.y, gv.y, av.y, bv.y, a);
    opt_elem<KIND>(pv.z, gv.z, av.z, bv.z, a);
    opt_elem<KIND>(pv.w, gv.w, av.w, bv.w, a);
    reinterpret_cast<float4*>(p)[i] = pv;
    if (touched) reinterpret_cast<float4*>(g)[i] = make_float4(0, 0, 0, 0);
    if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[i] = av;
    if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[i] = bv;
    if (fl && touched && first) *fl = 0;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {   // dense tail
    float pv = p[i], gv = g[i], av = 0.f, bv = 0.f;
    if (KIND != DCCF_OPT_GD) av = s1[i];
    if (KIND == DCCF_OPT_ADAM) bv = s2[i];
    opt_elem<KIND>(pv, gv, av, bv, a);
    p[i] = pv;
    g[i] = 0.f;
    if (KIND != DCCF_OPT_GD) s1[i] = av;
    if (KIND == DCCF_OPT_ADAM) s2[i] = bv;
  }
}

static int opt_rows_impl(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd, float l2,
                         float clip, int64_t step, const int64_t* k_dev, int32_t nseg, const int64_t* seg_begin,
                         const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags, void* stream) {
  ARG_CHECK(p && g && n >= 0 && step >= 1, "NULL p/g, n < 0 or step < 1");
  ARG_CHECK(kind == DCCF_OPT_GD || kind == DCCF_OPT_ADAGRAD || kind == DCCF_OPT_ADAM, "unknown optimizer kind");
  ARG_CHECK(kind == DCCF_OPT_GD || s1, "optimizer state s1 is NULL");
  ARG_CHECK(kind != DCCF_OPT_ADAM || s2, "optimizer state s2 is NULL");
  ARG_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && (!s1 || (uintptr_t)s1 % 16 == 0) &&
                (!s2 || (uintptr_t)s2 % 16 == 0),
            "buffers must be 16-byte aligned");
  ARG_CHECK(nseg >= 0 && nseg <= 4 && (nseg == 0 || (seg_begin && seg_rows && seg_width && seg_flags)), "bad segments");
  RowSegs sg;
  sg.n = nseg;
  for (int q = 0; q < nseg; ++q) {
    const int w = seg_width[q];
    ARG_CHECK(w == 16 || w == 32 || w == 64 || w == 128, "segment row width must be 16, 32, 64 or 128");
    ARG_CHECK(seg_begin[q] % 256 == 0 && seg_rows[q] >= 0 && seg_begin[q] + seg_rows[q] * w <= n && seg_flags[q],
              "segment must start on a 256-float boundary, lie inside the buffer and have flags");
    sg.begin[q] = seg_begin[q];
    sg.end[q] = seg_begin[q] + seg_rows[q] * w;
    sg.width[q] = w;
    sg.flags[q] = seg_flags[q];
  }
  if (n == 0) return 0;
  OptArgs a;
  a.lr = lr; a.wd = wd; a.l2 = l2; a.clip = clip; a.zero_grad = 1;
  a.k_dev = k_dev; a.step0 = step;
  const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
  a.step_size_neg = (float)(-((double)lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  const int64_t work = (n + 3) / 4;
  const int grid = (int)min((int64_t)(256 * 16), (work + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (kind == DCCF_OPT_GD) hipLaunchKernelGGL(k_dense_opt_rows<DCCF_OPT_GD>, dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a, sg);
  else if (kind == DCCF_OPT_ADAGRAD) hipLaunchKernelGGL(k_dense_opt_rows<DCCF_OPT_ADAGRAD>, dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a, sg);
  else hipLaunchKernelGGL(k_dense_opt_rows<DCCF_OPT_ADAM>, dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a, sg);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dccf_dense_opt_step_rows(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr,
                                        float wd, float l2, float clip, int64_t step, int32_t nseg, const int64_t* seg_begin,
                                        const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags,
                                        void* stream) {
  return opt_rows_impl(kind, p, g, s1, s2, n, lr, wd, l2, clip, step, nullptr, nseg, seg_begin, seg_rows, seg_width, seg_flags,
                       stream);
}

extern "C" int dccf_dense_opt_step_dev(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                                       float l2, float clip, int64_t step, const int64_t* k_dev, int32_t nseg,
                                       const int64_t* seg_begin, const int64_t* seg_rows, const int32_t* seg_width,
                                       uint8_t* const* seg_flags, void* stream) {
  ARG_CHECK(k_dev != nullptr, "k_dev is NULL");
  return opt_rows_impl(kind, p, g, s1, s2, n, lr, wd, l2, clip, step, k_dev, nseg, seg_begin, seg_rows, seg_width, seg_flags,
                       stream);
}

__global__ void k_advance(int64_t* k) { *k += 1; }
extern "C" int dccf_advance(int64_t* k_dev, void* stream) {
  ARG_CHECK(k_dev != nullptr, "k_dev is NULL");
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, k_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dccf_dense_opt_step(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                                   float l2, float clip, int64_t step, int32_t zero_grad, void* stream) {
  ARG_CHECK(p && g && n >= 0 && step >= 1, "NULL p/g, n < 0 or step < 1");
  ARG_CHECK(kind == DCCF_OPT_GD || kind == DCCF_OPT_ADAGRAD || kind == DCCF_OPT_ADAM, "unknown optimizer kind");
  ARG_CHECK(kind == DCCF_OPT_GD || s1, "optimizer state s1 is NULL");
  ARG_CHECK(kind != DCCF_OPT_ADAM || s2, "optimizer state s2 is NULL");
  ARG_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && (!s1 || (uintptr_t)s1 % 16 == 0) &&
                (!s2 || (uintptr_t)s2 % 16 == 0),
            "buffers must be 16-byte aligned");
  if (n == 0) return 0;
  OptArgs a;
  a.lr = lr; a.wd = wd; a.l2 = l2; a.clip = clip; a.zero_grad = zero_grad;
  a.k_dev = nullptr; a.step0 = step;
  // bias corrections in double like torch's Python scalars (torch/optim/adam.py::_single_tensor_adam)
  const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
  a.step_size_neg = (float)(-((double)lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  const int64_t work = (n + 3) / 4;
  const int grid = (int)min((int64_t)(256 * 16), (work + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (kind == DCCF_OPT_GD) hipLaunchKernelGGL(k_dense_opt<DCCF_OPT_GD>, dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a);
  else if (kind == DCCF_OPT_ADAGRAD) hipLaunchKernelGGL(k_dense_opt<DCCF_OPT_ADAGRAD>, dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a);
  else hipLaunchKernelGGL(k_dense_opt<DCCF_OPT_ADAM>, dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------- sum of squares
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ p, int64_t n, float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc = fmaf(p[i], p[i], acc);
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int dccf_sumsq(const float* p, int64_t n, float* out, void* stream) {
  ARG_CHECK(p && out && n >= 0, "bad arguments");
  if (n == 0) return 0;
  const int grid = (int)min((int64_t)1024, (n + 255) / 256);
  hipLaunchKernelGGL(k_sumsq, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, n, out);
  HIP_TRY(hipGetLastError());
  return 0;
}
