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
#include "opt_device.hpp"

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
  c->side = nullptr;
  c->ev_fork = c->ev_join = nullptr;
  c->tl_list = nullptr;
  c->tl_cap = 0;
  c->tl_cnt = nullptr;
  c->tl_parity = 0;
  c->prep_valid = 0;
  c->lazy_prep_step = -1;
  c->gw_part = nullptr;
  c->gw_part_bytes = 0;
  c->lazy_prep_claim = nullptr;
  c->lazy_prep_id = 0;
  c->prep_hits = 0;
  c->prep_dp = c->prep_pending = c->prep_parity = 0;
  c->prep_tables = c->cur_tables = 0;
  c->slot_where = nullptr;
  c->slot_rows = nullptr;
  c->hv_flags[0] = c->hv_flags[1] = nullptr;
  c->hv_items = 0;
  c->hv_parity = c->hv_prepared = 0;
  c->cur_Xall = nullptr;
  c->prep_Xall = nullptr;
  c->det = (getenv("DCCF_DETERMINISTIC") && atoi(getenv("DCCF_DETERMINISTIC")) != 0) ? 1 : 0;
  c->det_buf = nullptr;
  c->det_bytes = 0;
  c->det_owner = nullptr;
  c->det_owner_n = 0;
  *out = c;
  return 0;
}

extern "C" int dccf_ctx_set_deterministic(dccf_ctx* ctx, int on) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  ctx->det = on ? 1 : 0;
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
  if (ctx->gw_part) (void)hipFree(ctx->gw_part);
  if (ctx->det_buf) (void)hipFree(ctx->det_buf);
  if (ctx->det_owner) (void)hipFree(ctx->det_owner);
  for (int q = 0; q < 2; ++q)
    if (ctx->hv_flags[q]) (void)hipFree(ctx->hv_flags[q]);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->tl_list) (void)hipFree(ctx->tl_list);
  if (ctx->tl_cnt) (void)hipFree(ctx->tl_cnt);
  delete ctx;
  return 0;
}

// Side stream, events and touched-row list of the overlapped step; the list grows like the workspace does.
int dccf_step_ensure(dccf_ctx* ctx, int64_t max_rows) {
  if (!ctx->side) {
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));      // lo = least priority: forward/backward win the dispatcher
    // DCCF_SIDE_CUS=n: confine the side stream to n of the 256 CUs (mask bits interleave over the 8 XCDs, so every XCD
    // gives n/8) — the optimizer pass saturates HBM from a subset of the CUs and leaves the rest to forward/backward
    int ncu = 0;
    if (const char* e = getenv("DCCF_SIDE_CUS")) ncu = atoi(e);
    if (ncu > 0 && ncu < 256) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < ncu; ++i) mask[i >> 5] |= 1u << (i & 31);
      HIP_TRY(hipExtStreamCreateWithCUMask(&ctx->side, 8, mask));
    } else {
      HIP_TRY(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, lo));
    }
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    HIP_TRY(hipMalloc((void**)&ctx->tl_cnt, 2 * sizeof(int)));
    HIP_TRY(hipMemset(ctx->tl_cnt, 0, 2 * sizeof(int)));
  }
  if (max_rows > ctx->tl_cap) {
    HIP_TRY(hipDeviceSynchronize());
    if (ctx->tl_list) HIP_TRY(hipFree(ctx->tl_list));
    ctx->tl_list = nullptr;
    ctx->tl_cap = 0;
    const int64_t want = max_rows + max_rows / 8 + 1024;
    HIP_TRY(hipMalloc((void**)&ctx->tl_list, (size_t)want * sizeof(int64_t)));
    ctx->tl_cap = want;
  }
  return 0;
}

extern "C" int dccf_ctx_prepared_steps(const dccf_ctx* ctx, int64_t* out) {
  ARG_CHECK(ctx && out, "NULL argument");
  *out = ctx->prep_hits;
  return 0;
}

extern "C" int dccf_ctx_side_stream(dccf_ctx* ctx, void** out) {
  ARG_CHECK(ctx && out, "NULL argument");
  if (int e = dccf_step_ensure(ctx, 0)) return e;
  *out = (void*)ctx->side;
  return 0;
}

int dccf_ws_ensure(dccf_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return 0;
  // grow-only; growing synchronises the device (earlier launches may still use the old slab)
  HIP_TRY(hipDeviceSynchronize());
  ctx->prep_valid = 0;                       // the prepared slots lived in the old slab
  if (ctx->ws) HIP_TRY(hipFree(ctx->ws));
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  const size_t want = align_up(bytes + bytes / 8, 1 << 20);
  HIP_TRY(hipMalloc((void**)&ctx->ws, want));
  ctx->ws_bytes = want;
  return 0;
}

// ---------------------------------------------------------------------------------------------- optimizer
// (element arithmetic, OptArgs, RowSegs and the fused-slice role: opt_device.hpp)

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
    opt_elem4<KIND>(pv, gv, av, bv, a);
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

// Element intervals outside the row segments ("dense": always read their gradient).
struct DenseSegs {
  int64_t begin[5], end[5];
  int n;
};

// The rows on the step's touched list (half a wave per row).
template <int KIND>
__device__ __forceinline__ void touched_rows(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                             float* __restrict__ s2, const OptArgs& a, const RowSegs& sg,
                                             const int64_t* __restrict__ list, int n, int block, int row_blocks) {
  const int half = threadIdx.x >> 5, l = threadIdx.x & 31;
  for (int e = block * 8 + half; e < n; e += row_blocks * 8) {
    const int64_t tag = list[e];
    const int q = (int)(tag >> 40);
    const int64_t row = tag & ((1LL << 40) - 1);
    const int w4 = sg.width[q] >> 2;
    const int64_t i0 = (sg.begin[q] + row * sg.width[q]) >> 2;
    if (l < w4) {
      const int64_t i = i0 + l;
      float4 pv = reinterpret_cast<float4*>(p)[i];
      float4 gv = reinterpret_cast<float4*>(g)[i];
      float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
      if (KIND != DCCF_OPT_GD) av = reinterpret_cast<float4*>(s1)[i];
      if (KIND == DCCF_OPT_ADAM) bv = reinterpret_cast<float4*>(s2)[i];
      opt_elem4<KIND>(pv, gv, av, bv, a);
      reinterpret_cast<float4*>(p)[i] = pv;
      reinterpret_cast<float4*>(g)[i] = make_float4(0, 0, 0, 0);
      if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[i] = av;
      if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[i] = bv;
    }
    if (l == 0) sg.flags[q][row] = 0;
  }
}

// Row-aware variant: g of a row whose "touched" byte is 0 is all zeros by construction -> not read, not re-zeroed.
// A wave handles 64 consecutive float4 = 256 consecutive floats = whole rows (row widths 16..128 divide 256 and segments
// start on a 256-float boundary), so every lane of a row sees the byte before the row's first lane clears it.
// NT: non-temporal loads / stores of p, m, v — for buffers far beyond the 256 MiB Infinity Cache, where every byte is read once and
// written once per launch and keeping lines in L2 / the LLC buys nothing (in-cache sizes are slower with it: DESIGN.md section 4)
typedef float opt_f4v __attribute__((ext_vector_type(4)));      // (the nontemporal builtins take clang vectors, not HIP's float4 struct)
template <bool NT>
__device__ __forceinline__ float4 opt_ld4(const float4* q) {
  if (NT) {
    const opt_f4v v = __builtin_nontemporal_load(reinterpret_cast<const opt_f4v*>(q));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *q;
}
template <bool NT>
__device__ __forceinline__ void opt_st4(float4* q, const float4& v) {
  if (NT) {
    const opt_f4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<opt_f4v*>(q));
  } else {
    *q = v;
  }
}
// GW: row widths that are not a power of two (any multiple of 4 up to 256: src/models/RecModel.py:17-27 accepts any embedding size) —
// the row of a slot by a 32-bit division instead of a shift (segments stay below 2^32 elements: checked at launch)
template <int KIND, int UN, bool TO, bool NT = false, bool GW = false>
__global__ __launch_bounds__(256) void k_dense_opt_rows(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                        float* __restrict__ s2, int64_t n, OptArgs a, RowSegs sg,
                                                        int phase, PrepNext pn) {
  opt_resolve(a);
  // the next step's candidate / exposure slots and marks (independent of this pass) take the FIRST workgroups: their
  // dependent chains (Philox -> gather -> atomics) then run under the streaming pass instead of after it
  if ((int)blockIdx.x < pn.blocks) {
    prep_next_slots(pn, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)pn.blocks * blockDim.x);
    return;
  }
  const int nblk = (int)gridDim.x - pn.blocks;
  const int bid = (int)blockIdx.x - pn.blocks;
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)nblk * blockDim.x;
  // row widths are powers of two (16..128): shifts instead of 64-bit divisions
  int wsh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wsh[q] = q < sg.n ? 31 - __clz(sg.width[q]) : 0;
  // UN float4 slots in flight per thread: 2 pays once the tables exceed the Infinity Cache (5.0 -> 5.5 TB/s at 1.4 G
  // parameters), 1 is better at Electronics size (fewer registers, 8 waves per SIMD)
  for (int64_t i0 = bid * (int64_t)blockDim.x + threadIdx.x; i0 < n4; i0 += stride * UN) {
    float4 pv[UN], av[UN], bv[UN], gv[UN];
    uint8_t* fl[UN];
    bool live[UN], first[UN], touched[UN];
    // the state loads do not wait for the flag byte: they are needed whatever it says (phase 1 wastes them on the few
    // touched rows)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * stride;
      live[u] = i < n4;
      const int64_t ic = live[u] ? i : n4 - 1;
      const int64_t e = ic * 4;
      fl[u] = nullptr;
      first[u] = false;
      bool tonly = false;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < sg.n && e >= sg.begin[q] && e < sg.end[q]) {
          const int64_t off = e - sg.begin[q];
          if (GW) {
            const uint32_t row = (uint32_t)off / (uint32_t)sg.width[q];
            fl[u] = sg.flags[q] + row;
            first[u] = (uint32_t)off == row * (uint32_t)sg.width[q];
          } else {
            fl[u] = sg.flags[q] + (off >> wsh[q]);
            first[u] = (off & (sg.width[q] - 1)) == 0;
          }
          if (TO) tonly = (sg.to_mask >> q) & 1;
        }
      av[u] = bv[u] = gv[u] = make_float4(0, 0, 0, 0);
      // a wave's 64 slots lie almost always in ONE segment (segments start on 256-float boundaries; only a segment's last
      // wave may reach into what follows it), so this is a scalar branch and the common path keeps its unconditional loads
      if (TO && __ballot(tonly) != 0) {
        // (some lanes in) a segment whose unmarked rows were updated by the pass hosted in the backward launch: the byte
        // decides before anything is loaded — most of these slots have nothing to do.  Lanes past the segment's end take
        // the ordinary rule.
        const bool marked = fl[u] ? *fl[u] != 0 : true;
        touched[u] = marked;
        const bool skip = tonly && !marked;
        if (skip) live[u] = false;
        pv[u] = make_float4(0, 0, 0, 0);
        if (!skip) {
          pv[u] = reinterpret_cast<const float4*>(p)[ic];
          if (KIND != DCCF_OPT_GD) av[u] = reinterpret_cast<const float4*>(s1)[ic];
          if (KIND == DCCF_OPT_ADAM) bv[u] = reinterpret_cast<const float4*>(s2)[ic];
        }
      } else {
        pv[u] = opt_ld4<NT>(reinterpret_cast<const float4*>(p) + ic);
        if (KIND != DCCF_OPT_GD) av[u] = opt_ld4<NT>(reinterpret_cast<const float4*>(s1) + ic);
        if (KIND == DCCF_OPT_ADAM) bv[u] = opt_ld4<NT>(reinterpret_cast<const float4*>(s2) + ic);
        touched[u] = fl[u] ? *fl[u] != 0 : true;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * stride;
      if (!live[u]) continue;
      if (phase == OPT_PHASE_UNTOUCHED && (!fl[u] || touched[u])) continue;     // those wait for the backward (k_opt_touched)
      if (touched[u]) gv[u] = reinterpret_cast<float4*>(g)[i];
      opt_elem4<KIND>(pv[u], gv[u], av[u], bv[u], a);
      opt_st4<NT>(reinterpret_cast<float4*>(p) + i, pv[u]);
      if (pn.blocks && i * 4 >= pn.w_begin && i * 4 < pn.w_end) {      // the forward's transposed copy of W, for the next step
        prep_next_wt(pn, i * 4, pv[u].x);
        prep_next_wt(pn, i * 4 + 1, pv[u].y);
        prep_next_wt(pn, i * 4 + 2, pv[u].z);
        prep_next_wt(pn, i * 4 + 3, pv[u].w);
      }
      if (touched[u]) reinterpret_cast<float4*>(g)[i] = make_float4(0, 0, 0, 0);
      if (KIND != DCCF_OPT_GD) opt_st4<NT>(reinterpret_cast<float4*>(s1) + i, av[u]);
      if (KIND == DCCF_OPT_ADAM) opt_st4<NT>(reinterpret_cast<float4*>(s2) + i, bv[u]);
      // (GW: a row of such a width may straddle two waves — the other wave must still see the byte: the host clears the bytes after the launch)
      if (!GW && fl[u] && touched[u] && first[u]) *fl[u] = 0;
    }
  }
  if (phase == OPT_PHASE_UNTOUCHED) return;
  for (int64_t i = n4 * 4 + bid * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {   // dense tail
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

// Second half of the overlapped step: the rows on the step's touched list (half a wave per row) and the elements outside
// the row segments (W, b: dense), after the backward has produced their gradients.
template <int KIND>
__global__ __launch_bounds__(256) void k_opt_touched(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                     float* __restrict__ s2, OptArgs a, RowSegs sg,
                                                     const int64_t* __restrict__ list, const int* __restrict__ cnt,
                                                     int row_blocks, DenseSegs ds) {
  opt_resolve(a);
  if ((int)blockIdx.x < row_blocks) {
    touched_rows<KIND>(p, g, s1, s2, a, sg, list, *cnt, blockIdx.x, row_blocks);
    return;
  }
  const int64_t tid = (int64_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)(gridDim.x - row_blocks) * blockDim.x;
  for (int d = 0; d < ds.n; ++d)
    for (int64_t i = ds.begin[d] + tid; i < ds.end[d]; i += stride) {
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

int opt_make_job(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd, float l2,
                    float clip, int64_t step, const int64_t* k_dev, int32_t nseg, const int64_t* seg_begin,
                    const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags, OptJob* out) {
  ARG_CHECK(p && g && n >= 0 && step >= 1, "NULL p/g, n < 0 or step < 1");
  ARG_CHECK(kind == DCCF_OPT_GD || kind == DCCF_OPT_ADAGRAD || kind == DCCF_OPT_ADAM, "unknown optimizer kind");
  ARG_CHECK(kind == DCCF_OPT_GD || s1, "optimizer state s1 is NULL");
  ARG_CHECK(kind != DCCF_OPT_ADAM || s2, "optimizer state s2 is NULL");
  ARG_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && (!s1 || (uintptr_t)s1 % 16 == 0) &&
                (!s2 || (uintptr_t)s2 % 16 == 0),
            "buffers must be 16-byte aligned");
  ARG_CHECK(nseg >= 0 && nseg <= 4 && (nseg == 0 || (seg_begin && seg_rows && seg_width && seg_flags)), "bad segments");
  RowSegs& sg = out->sg;
  memset(&sg, 0, sizeof(sg));
  sg.n = nseg;
  for (int q = 0; q < nseg; ++q) {
    const int w = seg_width[q];
    // 16 / 32 / 64 / 128: every form of the pass.  Any other multiple of 4 up to 128 (src/models/RecModel.py:17-27 accepts any
    // embedding size): the row roles of the lazy optimizer (float4 slots of a row, integer arithmetic on the row number); the
    // streaming row-aware pass — whole rows per wave, shifts — then runs as the plain dense pass (launch_job)
    ARG_CHECK(w >= 4 && w <= 256 && w % 4 == 0, "segment row width must be a multiple of 4 in [4, 256]");
    ARG_CHECK(seg_begin[q] % 256 == 0 && seg_rows[q] >= 0 && seg_begin[q] + seg_rows[q] * w <= n && seg_flags[q],
              "segment must start on a 256-float boundary, lie inside the buffer and have flags");
    sg.begin[q] = seg_begin[q];
    sg.end[q] = seg_begin[q] + seg_rows[q] * w;
    sg.width[q] = w;
    sg.flags[q] = seg_flags[q];
  }
  ARG_CHECK(clip >= 0.f, "clip must be >= 0");
  OptArgs& a = out->a;
  a.lr = lr; a.wd = wd; a.l2 = l2; a.clip = clip; a.zero_grad = 1;
  a.k_dev = k_dev; a.step0 = step;
  // bias corrections in double like torch's Python scalars (torch/optim/adam.py::_single_tensor_adam)
  const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
  a.step_size_neg = (float)(-((double)lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.bc2_rsqrt = (float)(1.0 / (double)a.bc2_sqrt);
  out->p = p; out->g = g; out->s1 = s1; out->s2 = s2; out->n = n; out->kind = kind;
  return 0;
}

// the complement of the row segments, in ascending order
static int dense_complement(const RowSegs& sg, int64_t n, DenseSegs* ds, int64_t* total) {
  ds->n = 0;
  int order[4] = {0, 1, 2, 3};
  for (int i = 0; i < sg.n; ++i)
    for (int j = i + 1; j < sg.n; ++j)
      if (sg.begin[order[j]] < sg.begin[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  int64_t at = 0;
  *total = 0;
  for (int i = 0; i <= sg.n; ++i) {
    const int64_t b = i < sg.n ? sg.begin[order[i]] : n;
    ARG_CHECK(b >= at, "row segments overlap");
    if (b > at) { ds->begin[ds->n] = at; ds->end[ds->n] = b; ++ds->n; *total += b - at; }
    if (i < sg.n) at = sg.end[order[i]];
  }
  return 0;
}

#define BY_KIND(kind, K, ...)                                                               \
  if (kind == DCCF_OPT_GD) hipLaunchKernelGGL(K<DCCF_OPT_GD>, __VA_ARGS__);                 \
  else if (kind == DCCF_OPT_ADAGRAD) hipLaunchKernelGGL(K<DCCF_OPT_ADAGRAD>, __VA_ARGS__);  \
  else hipLaunchKernelGGL(K<DCCF_OPT_ADAM>, __VA_ARGS__)

static bool tile_width(int w) { return w == 16 || w == 32 || w == 64 || w == 128; }

static int launch_job(const OptJob& j0, int phase, const int64_t* list, const int* cnt, int64_t max_rows, hipStream_t st,
                      const PrepNext* pnp = nullptr) {
  if (j0.n == 0) return 0;
  OptJob j = j0;
  bool general = false;
  for (int q = 0; q < j.sg.n; ++q) general = general || !tile_width(j.sg.width[q]);
  if (general) {
    // row widths that are not a power of two: the GW instances (a division per slot instead of a shift)
    for (int q = 0; q < j.sg.n; ++q) ARG_CHECK(j.sg.end[q] - j.sg.begin[q] < 4294967296LL, "row segment too large for a width that is not 16, 32, 64 or 128");
  }
  if (phase == OPT_PHASE_TOUCHED) {
    ARG_CHECK(list && cnt, "touched phase needs the row list");
    DenseSegs ds;
    int64_t dense_total = 0;
    if (int e = dense_complement(j.sg, j.n, &ds, &dense_total)) return e;
    const int row_blocks = (int)min((int64_t)2048, (max_rows + 7) / 8);
    const int dense_blocks = (int)min((int64_t)1024, (dense_total + 255) / 256);
    const int grid = row_blocks + dense_blocks;
    if (grid == 0) return 0;
    BY_KIND(j.kind, k_opt_touched, dim3(grid), dim3(256), 0, st, j.p, j.g, j.s1, j.s2, j.a, j.sg, list, cnt, row_blocks, ds);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const int64_t work = (j.n + 3) / 4;
  // one float4 slot per thread at Electronics size (16384 workgroups): 61.5 -> 59.3 us against 4096 grid-striding workgroups
  // (DCCF_OPT_TUNE=1: the knobs below are read at every launch — scripts/dense_opt_bench.py sweeps them in one process)
  static const bool tune = getenv("DCCF_OPT_TUNE") != nullptr;
  // Buffers beyond the 256 MiB Infinity Cache (p + m + v of 2.5e7 parameters = 300 MB): ONE float4 slot per thread — no grid
  // stride —, non-temporal loads and stores: 6.28 TB/s = 78.5 % of the 8 TB/s peak on 268 M parameters = the box's copy rate
  // (profiles/r03_dense_opt_bench.json; 2 or 4 slots in flight per lane, default-policy accesses or a capped grid: 5.1-5.7 TB/s).
  // In-cache sizes keep the capped grid and the default policy (non-temporal is slower there: DESIGN.md section 4).
  static const int64_t gmax_env = getenv("DCCF_OPT_GRID") ? atoll(getenv("DCCF_OPT_GRID")) : 0;
  static const int un0 = getenv("DCCF_OPT_UN") ? atoi(getenv("DCCF_OPT_UN")) : 1;
  static const int nt0 = getenv("DCCF_OPT_NT") ? atoi(getenv("DCCF_OPT_NT")) : 1;
  // (33 M parameters = 400 MB of p + m + v: 5.83 against 5.52 TB/s; 67 M: 6.46 / 5.98; 134 M: 6.23 / 5.75 — non-temporal and one slot per
  // thread win as soon as the state no longer fits the 256 MiB cache; 16.4 M = 197 MB, the Electronics model, stays below)
  static const int64_t big0 = getenv("DCCF_OPT_BIG_N") ? atoll(getenv("DCCF_OPT_BIG_N")) : 25000000LL;
  const int64_t big_n = tune && getenv("DCCF_OPT_BIG_N") ? atoll(getenv("DCCF_OPT_BIG_N")) : big0;
  const bool big = j.n >= big_n && !j.sg.to_mask;
  const int64_t gmax_dflt = big ? ((int64_t)1 << 24) : 16384;
  const int64_t gmax = tune && getenv("DCCF_OPT_GRID") ? atoll(getenv("DCCF_OPT_GRID")) : (gmax_env ? gmax_env : gmax_dflt);
  const int un_big = tune && getenv("DCCF_OPT_UN") ? atoi(getenv("DCCF_OPT_UN")) : un0;
  const int nt_big = tune && getenv("DCCF_OPT_NT") ? atoi(getenv("DCCF_OPT_NT")) : nt0;
  PrepNext pn;
  memset(&pn, 0, sizeof(pn));
  if (pnp) pn = *pnp;
  const int grid = (int)min(gmax, (work + 255) / 256) + pn.blocks;
#define OPT_ROWS_LAUNCH(KIND_, UN_, TO_, NT_) \
  hipLaunchKernelGGL((k_dense_opt_rows<KIND_, UN_, TO_, NT_>), dim3(grid), dim3(256), 0, st, j.p, j.g, j.s1, j.s2, j.n, j.a, j.sg, phase, pn)
#define OPT_ROWS_KIND4(UN_, TO_, NT_)                                               \
  if (j.kind == DCCF_OPT_GD) OPT_ROWS_LAUNCH(DCCF_OPT_GD, UN_, TO_, NT_);           \
  else if (j.kind == DCCF_OPT_ADAGRAD) OPT_ROWS_LAUNCH(DCCF_OPT_ADAGRAD, UN_, TO_, NT_); \
  else OPT_ROWS_LAUNCH(DCCF_OPT_ADAM, UN_, TO_, NT_)
#define OPT_ROWS_KIND(UN_, TO_) OPT_ROWS_KIND4(UN_, TO_, false)
#define OPT_ROWS_LAUNCH_GW(KIND_, TO_) \
  hipLaunchKernelGGL((k_dense_opt_rows<KIND_, 1, TO_, false, true>), dim3(grid), dim3(256), 0, st, j.p, j.g, j.s1, j.s2, j.n, j.a, j.sg, phase, pn)
#define OPT_ROWS_KIND_GW(TO_)                                                   \
  if (j.kind == DCCF_OPT_GD) OPT_ROWS_LAUNCH_GW(DCCF_OPT_GD, TO_);              \
  else if (j.kind == DCCF_OPT_ADAGRAD) OPT_ROWS_LAUNCH_GW(DCCF_OPT_ADAGRAD, TO_); \
  else OPT_ROWS_LAUNCH_GW(DCCF_OPT_ADAM, TO_)
  // TO: a segment in "only the marked rows" mode (its other rows were updated by the pass hosted in the backward launch)
  if (general) {
    ARG_CHECK(j.sg.to_mask == 0, "the pass hosted in the backward launch needs row widths of 16, 32, 64 or 128");
    OPT_ROWS_KIND_GW(false);
    if (phase == OPT_PHASE_ALL)
      for (int q = 0; q < j.sg.n; ++q)
        HIP_TRY(hipMemsetAsync(j.sg.flags[q], 0, (size_t)((j.sg.end[q] - j.sg.begin[q]) / j.sg.width[q]), st));
  } else if (j.n >= big_n) {
    if (j.sg.to_mask) { OPT_ROWS_KIND(2, true); }
    else if (un_big >= 4) { if (nt_big) { OPT_ROWS_KIND4(4, false, true); } else { OPT_ROWS_KIND4(4, false, false); } }
    else if (un_big <= 1) { if (nt_big) { OPT_ROWS_KIND4(1, false, true); } else { OPT_ROWS_KIND4(1, false, false); } }
    else { if (nt_big) { OPT_ROWS_KIND4(2, false, true); } else { OPT_ROWS_KIND4(2, false, false); } }
  } else {
    if (j.sg.to_mask) { OPT_ROWS_KIND(1, true); } else { OPT_ROWS_KIND(1, false); }
  }
#undef OPT_ROWS_KIND_GW
#undef OPT_ROWS_LAUNCH_GW
#undef OPT_ROWS_KIND
#undef OPT_ROWS_KIND4
#undef OPT_ROWS_LAUNCH
  HIP_TRY(hipGetLastError());
  return 0;
}

static int opt_rows_impl(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd, float l2,
                         float clip, int64_t step, const int64_t* k_dev, int32_t nseg, const int64_t* seg_begin,
                         const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags, void* stream) {
  OptJob j;
  if (int e = opt_make_job(kind, p, g, s1, s2, n, lr, wd, l2, clip, step, k_dev, nseg, seg_begin, seg_rows, seg_width, seg_flags, &j))
    return e;
  return launch_job(j, OPT_PHASE_ALL, nullptr, nullptr, 0, (hipStream_t)stream);
}

static int opt_job(const void* ov, OptJob* out) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  ARG_CHECK(o != nullptr, "opt is NULL");
  return opt_make_job(o->kind, o->p, o->g, o->s1, o->s2, o->n, o->lr, o->wd, o->l2, o->clip, o->step, nullptr, o->nseg,
                  o->seg_begin, o->seg_rows, o->seg_width, o->seg_flags, out);
}

int dccf_opt_job(const void* ov, OptJob* out) { return opt_job(ov, out); }

int dccf_opt_phase(const void* ov, int phase, const int64_t* list, const int* cnt, int64_t max_rows, hipStream_t st) {
  OptJob j;
  if (int e = opt_job(ov, &j)) return e;
  return launch_job(j, phase, list, cnt, max_rows, st);
}

int dccf_opt_all_prep_to(const void* ov, int seg, uint8_t* flags, int64_t rows_hosted, const PrepNext* pn, hipStream_t st) {
  OptJob j;
  if (int e = opt_job(ov, &j)) return e;
  ARG_CHECK(seg >= 0 && seg < j.sg.n && flags, "bad segment / flags");
  const int w = j.sg.width[seg];
  const int64_t rows = (j.sg.end[seg] - j.sg.begin[seg]) / w;
  ARG_CHECK(rows_hosted >= 0 && rows_hosted <= rows && (rows_hosted * w) % 256 == 0, "hosted rows must end on a 256-float boundary");
  j.sg.flags[seg] = flags;
  if (rows_hosted == rows) {
    j.sg.to_mask = 1 << seg;
  } else if (rows_hosted > 0) {
    ARG_CHECK(j.sg.n < 4, "no free segment slot for the split");
    const int q2 = j.sg.n++;                       // the ordinary rest of the segment
    j.sg.begin[q2] = j.sg.begin[seg] + rows_hosted * w;
    j.sg.end[q2] = j.sg.end[seg];
    j.sg.width[q2] = w;
    j.sg.flags[q2] = flags + rows_hosted;
    j.sg.end[seg] = j.sg.begin[q2];
    j.sg.to_mask = 1 << seg;
  }
  if (pn) ARG_CHECK(pn->w_begin % 4 == 0 && pn->w_end % 4 == 0 && pn->w_end <= j.n, "W must be 16-byte aligned inside the flat buffer");
  return launch_job(j, OPT_PHASE_ALL, nullptr, nullptr, 0, st, pn);
}

// The whole pass + the next step's preparation in the same launch (dccf_train_step with X_next).
int dccf_opt_all_prep(const void* ov, const PrepNext* pn, hipStream_t st) {
  OptJob j;
  if (int e = opt_job(ov, &j)) return e;
  ARG_CHECK(pn->w_begin % 4 == 0 && pn->w_end % 4 == 0 && pn->w_end <= j.n, "W must be 16-byte aligned inside the flat buffer");
  return launch_job(j, OPT_PHASE_ALL, nullptr, nullptr, 0, st, pn);
}

int dccf_opt_untouched_prep(const void* ov, uint8_t* const* flags, const PrepNext* pn, hipStream_t st) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  ARG_CHECK(o != nullptr && flags != nullptr, "opt / flags is NULL");
  OptJob j;
  if (int e = opt_make_job(o->kind, o->p, o->g, o->s1, o->s2, o->n, o->lr, o->wd, o->l2, o->clip, o->step, nullptr, o->nseg,
                           o->seg_begin, o->seg_rows, o->seg_width, flags, &j))
    return e;
  return launch_job(j, OPT_PHASE_UNTOUCHED, nullptr, nullptr, 0, st, pn);
}

extern "C" int dccf_dense_opt_step_rows(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr,
                                        float wd, float l2, float clip, int64_t step, int32_t nseg, const int64_t* seg_begin,
                                        const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags,
                                        void* stream) {
  return opt_rows_impl(kind, p, g, s1, s2, n, lr, wd, l2, clip, step, nullptr, nseg, seg_begin, seg_rows, seg_width, seg_flags,
                       stream);
}

extern "C" int dccf_dense_opt_phase(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                                    float l2, float clip, int64_t step, int32_t nseg, const int64_t* seg_begin,
                                    const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags,
                                    int32_t phase, const int64_t* list, const int32_t* cnt, int64_t max_rows, void* stream) {
  ARG_CHECK(phase == OPT_PHASE_UNTOUCHED || phase == OPT_PHASE_TOUCHED, "phase must be 1 (untouched rows) or 2 (listed rows + dense)");
  OptJob j;
  if (int e = opt_make_job(kind, p, g, s1, s2, n, lr, wd, l2, clip, step, nullptr, nseg, seg_begin, seg_rows, seg_width, seg_flags, &j))
    return e;
  return launch_job(j, phase, list, cnt, max_rows, (hipStream_t)stream);
}

extern "C" int dccf_dense_opt_step_dev(int32_t kind, float* p, float* g, float* s1, float* s2, int64_t n, float lr, float wd,
                                       float l2, float clip, int64_t step, const int64_t* k_dev, int32_t nseg,
                                       const int64_t* seg_begin, const int64_t* seg_rows, const int32_t* seg_width,
                                       uint8_t* const* seg_flags, void* stream) {
  ARG_CHECK(k_dev != nullptr, "k_dev is NULL");
  return opt_rows_impl(kind, p, g, s1, s2, n, lr, wd, l2, clip, step, k_dev, nseg, seg_begin, seg_rows, seg_width, seg_flags,
                       stream);
}

// ---------------------------------------------------------------------------------------------- windowed lazy regularisation
#define LAZY_KMAX 64      // largest lazy_K (the launch keeps K + 1 step-scalar entries in LDS)
// (LazyArgs, lazy_replay and the window pass: opt_device.hpp — the backward launch can host the window)
// Before the forward of step t: every distinct row the step reads (the batch's users, its candidates) is claimed by ONE wave
// (atomicMax on claim), appended to the step's list and brought up to step t - 1.
// one slot of a step's row list, served by a 16-lane group (a float4 per lane, two for 128-wide rows): the group claims the row
// (the winner of a row's slots lists it) and brings it up to step t - 1.  `live` = the group has a slot at all.
template <int KIND>
__device__ __forceinline__ void lazy_catchup_slot(float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2, const OptArgs& a,
                                                  const RowSegs& sg, const LazyArgs& z, int* claim, int* list, int64_t j, int q,
                                                  int64_t row, int lane, bool live) {
  const int t = (int)z.step;
  const int sub = lane & 15;
  const int64_t grow = z.row_off[q] + row;
  int won = 0;
  if (live && sub == 0) {
    won = atomicMax(&claim[grow], t) < t ? 1 : 0;
    list[j] = won ? (int)grow : -1;        // the step's rows, one entry per slot (-1: another slot owns the row): no shared
  }                                        // counter — 3,000 appends to one address took 40 us
  won = __shfl(won, lane & 48, 64);
  if (!won) return;
  const LazyPend pend = lazy_pend_read(z);
  const int from = lazy_from(z, pend, grow, t);
  if (from >= t - 1) return;
  const int w4 = sg.width[q] >> 2;
  const int64_t b4 = (sg.begin[q] + row * sg.width[q]) >> 2;
  for (int c = sub; c < w4; c += 16) {
    float4 pv = reinterpret_cast<float4*>(p)[b4 + c];
    float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
    if (KIND != DCCF_OPT_GD) av = reinterpret_cast<float4*>(s1)[b4 + c];
    if (KIND == DCCF_OPT_ADAM) bv = reinterpret_cast<float4*>(s2)[b4 + c];
    lazy_replay4<KIND>(pv, av, bv, a, z, from, t - 1);
    reinterpret_cast<float4*>(p)[b4 + c] = pv;
    if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[b4 + c] = av;
    if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[b4 + c] = bv;
  }
  if (sub == 0) z.last[grow] = t - 1;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_lazy_catchup(float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2,
                                                      OptArgs a, RowSegs sg, LazyArgs z, const int64_t* __restrict__ X,
                                                      const int* __restrict__ cand, int64_t N, int S1, int segU, int segV) {
  const int lane = threadIdx.x & 63, grp = lane >> 4;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t slots = N * S1 + N;
  int* claim = lazy_claim_of(z, (int)z.step);
  int* list = lazy_list_of(z, (int)z.step);
  for (int64_t j0 = wave * 4; j0 < slots; j0 += nw * 4) {
    const int64_t j = j0 + grp;
    const bool live = j < slots;
    const int64_t jc = live ? j : slots - 1;
    if (jc < N * S1) lazy_catchup_slot<KIND>(p, s1, s2, a, sg, z, claim, list, jc, segV, cand[jc], lane, live);
    else lazy_catchup_slot<KIND>(p, s1, s2, a, sg, z, claim, list, jc, segU, X[2 * (jc - N * S1)], lane, live);
  }
}

// the same for a caller that names the rows itself (the row-sharded trainer: the rows it is about to send to its peers):
// slot j < n_a is row rows_a[j] of segment seg_a, slot n_a + j row rows_b[j] of segment seg_b
template <int KIND>
__global__ __launch_bounds__(256) void k_lazy_catchup_rows(float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2,
                                                           OptArgs a, RowSegs sg, LazyArgs z, const int* __restrict__ rows_a,
                                                           int64_t n_a, int seg_a, const int* __restrict__ rows_b, int64_t n_b,
                                                           int seg_b) {
  const int lane = threadIdx.x & 63, grp = lane >> 4;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int* claim = lazy_claim_of(z, (int)z.step);
  int* list = lazy_list_of(z, (int)z.step);
  const int64_t slots = n_a + n_b;
  for (int64_t j0 = wave * 4; j0 < slots; j0 += nw * 4) {
    const int64_t j = j0 + grp;
    const bool live = j < slots;
    const int64_t jc = live ? j : slots - 1;
    if (jc < n_a) lazy_catchup_slot<KIND>(p, s1, s2, a, sg, z, claim, list, jc, seg_a, rows_a[jc], lane, live);
    else lazy_catchup_slot<KIND>(p, s1, s2, a, sg, z, claim, list, jc, seg_b, rows_b[jc - n_a], lane, live);
  }
}

// the same for a batch of (user, item) pairs (the MF family's train step): the ids are read from X itself
template <int KIND>
__global__ __launch_bounds__(256) void k_lazy_catchup_pairs(float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2,
                                                            OptArgs a, RowSegs sg, LazyArgs z, const int64_t* __restrict__ X,
                                                            int64_t N, int seg_u, int seg_v, float* __restrict__ zero) {
  if (zero && blockIdx.x == 0 && threadIdx.x == 0) *zero = 0.f;
  const int lane = threadIdx.x & 63, grp = lane >> 4;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int* claim = lazy_claim_of(z, (int)z.step);
  int* list = lazy_list_of(z, (int)z.step);
  const int64_t slots = 2 * N;
  for (int64_t j0 = wave * 4; j0 < slots; j0 += nw * 4) {
    const int64_t j = j0 + grp;
    const bool live = j < slots;
    const int64_t jc = live ? j : slots - 1;
    if (jc < N) lazy_catchup_slot<KIND>(p, s1, s2, a, sg, z, claim, list, jc, seg_u, (int)X[2 * jc], lane, live);
    else lazy_catchup_slot<KIND>(p, s1, s2, a, sg, z, claim, list, jc, seg_v, (int)X[2 * (jc - N) + 1], lane, live);
  }
}

// The optimizer launch of step t.  Workgroups by role: [0, pn.blocks) the next step's preparation; then `lb` workgroups walk the
// step's list (one wave per row: gradient read, step t applied, gradient zeroed, byte cleared); `db` workgroups take the dense
// tail (W, b, ...: everything outside the row segments); the rest advance this step's window — the rows
// [R w / K, R (w + 1) / K) of the global row space, w = t mod K — to step t (rows the step touched excepted).
template <int KIND>
__global__ __launch_bounds__(256) void k_lazy_opt(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                                                  float* __restrict__ s2, OptArgs a, RowSegs sg, DenseSegs ds, LazyArgs z,
                                                  int lb, int db, int mb, int64_t win0, int64_t win1, int flush, int nslots,
                                                  PrepNext pn, GwPart gp, int64_t winS) {
  // the step scalars of the steps this launch can replay (t - K + 1 .. t), copied to LDS once: the replay loops read them with
  // a broadcast ds_read instead of a global load per lane and step inside their dependent chains
  __shared__ float4 s_sct[LAZY_KMAX + 1];
  const int sct_base = max((int)z.step - z.K, 0);
  if (KIND == DCCF_OPT_ADAM) {
    const int s = sct_base + (int)threadIdx.x;
    if ((int)threadIdx.x <= z.K && s >= (int)z.t0 && s <= (int)z.step) s_sct[threadIdx.x] = reinterpret_cast<const float4*>(z.scal)[s - z.t0];
    __syncthreads();
  }
  if ((int)blockIdx.x < pn.blocks) {
    prep_next_slots(pn, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)pn.blocks * blockDim.x);
    return;
  }
  const int t = (int)z.step;
  int bid = (int)blockIdx.x - pn.blocks;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (bid < pn.cu_blocks) {
    // ---- the rows of the NEXT step (known: X_next, and its candidates are a counter-based stream): claimed (the winner of a
    // row's slots lists it) and brought up to THIS step — unless another role of this launch does that: a row on this step's
    // list (claim of step t) gets step t there, a row of this step's window is advanced there.  Claims and lists are kept
    // per step parity, so what this role writes is not what the other roles read.  The next step then starts with its forward.
    const int S1 = pn.S + 1;
    // (replicated multi-GPU step: the rows of EVERY rank's next batch, from the replicated schedule; no list — the import
    // works from its own tables)
    const int G = pn.X_all ? pn.G : 1;
    const int64_t NS = pn.N * S1, per_rank = NS + pn.N, slots = (int64_t)G * per_rank;
    const int* __restrict__ claim_t = lazy_claim_of(z, t);
    int* claim_n = lazy_claim_of(z, t + 1);
    int* list_n = pn.X_all ? nullptr : lazy_list_of(z, t + 1);
    const LazyPend pend = lazy_pend_read(z);
    // a slot per 16-lane group, a float4 per lane (see the list role)
    const int grp = lane >> 4, sub = lane & 15;
    for (int64_t jg0 = ((int64_t)bid * 4 + wv) * 4; jg0 < slots; jg0 += (int64_t)pn.cu_blocks * 16) {
      const int64_t jg = jg0 + grp;
      const bool live = jg < slots;
      const int64_t jc = live ? jg : slots - 1;
      const int r = (int)(jc / per_rank);
      const int64_t j = jc - (int64_t)r * per_rank;
      const int64_t* __restrict__ Xr = pn.X_all ? pn.X_all + (int64_t)r * pn.N * 2 : pn.X;
      int q;
      int64_t row;
      if (j < NS) {
        const int64_t n = j / S1;
        const int sl = (int)(j % S1);
        q = pn.cu_segV;
        if (sl == 0) {
          row = Xr[2 * n + 1];
        } else {
          const rng_key key = pn.X_all ? key_plus(pn.gkey0, r) : pn.key;
          const u32x4 rr = philox4x32_10((uint32_t)n, (uint32_t)((sl - 1) >> 2), key.s0, key.s1, key.k0, key.k1);
          row = (int64_t)(((uint64_t)pick4(rr, (sl - 1) & 3) * (uint64_t)pn.M.item_num) >> 32);
        }
      } else {
        q = pn.cu_segU;
        row = Xr[2 * (j - NS)];
      }
      const int64_t grow = z.row_off[q] + row;
      int won = 0;
      if (live && sub == 0) {
        won = atomicMax(&claim_n[grow], t + 1) < t + 1 ? 1 : 0;
        if (list_n) list_n[j] = won ? (int)grow : -1;
      }
      won = __shfl(won, lane & 48, 64);
      // not mine: lost the row to another slot; on this step's list (that role brings it to t); in this step's window (ditto)
      if (!won || claim_t[grow] == t || (grow >= win0 && grow < win1)) continue;
      const int from = lazy_from(z, pend, grow, t);
      if (from >= t) continue;
      const int w4 = sg.width[q] >> 2;
      const int64_t b4 = (sg.begin[q] + row * sg.width[q]) >> 2;
      for (int c = sub; c < w4; c += 16) {
        float4 pv = reinterpret_cast<float4*>(p)[b4 + c];
        float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
        if (KIND != DCCF_OPT_GD) av = reinterpret_cast<float4*>(s1)[b4 + c];
        if (KIND == DCCF_OPT_ADAM) bv = reinterpret_cast<float4*>(s2)[b4 + c];
        lazy_replay4<KIND>(pv, av, bv, a, z, from, t);
        reinterpret_cast<float4*>(p)[b4 + c] = pv;
        if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[b4 + c] = av;
        if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[b4 + c] = bv;
      }
      // (the marks role of this launch may be writing the pending window's step into the same entry: both are maxima)
      if (sub == 0) atomicMax(&z.last[grow], t);
    }
    return;
  }
  bid -= pn.cu_blocks;
  if (bid < lb) {                              // ---- the rows this step touched
    // a row per 16-lane group, a float4 per lane (two for 128-wide rows): 4 rows per store instruction — a CU retires a
    // dword-per-lane store only every ~40 ns, and four rows of a wave progress together instead of one after the other
    const int* __restrict__ list = lazy_list_of(z, t);
    const int grp = lane >> 4, sub = lane & 15;
    for (int e0 = (bid * 4 + wv) * 4; e0 < nslots; e0 += lb * 16) {
      const int e = e0 + grp;
      const int64_t grow = e < nslots ? list[e] : -1;
      if (grow < 0) continue;
      int q = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (k < sg.n && grow >= z.row_off[k]) q = k;
      const int64_t row = grow - z.row_off[q];
      const int w4 = sg.width[q] >> 2;
      const int64_t b4 = (sg.begin[q] + row * sg.width[q]) >> 2;
      for (int c = sub; c < w4; c += 16) {
        float4 pv = reinterpret_cast<float4*>(p)[b4 + c], gv = reinterpret_cast<float4*>(g)[b4 + c];
        float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
        if (KIND != DCCF_OPT_GD) av = reinterpret_cast<float4*>(s1)[b4 + c];
        if (KIND == DCCF_OPT_ADAM) bv = reinterpret_cast<float4*>(s2)[b4 + c];
        opt_elem4<KIND>(pv, gv, av, bv, a);
        reinterpret_cast<float4*>(p)[b4 + c] = pv;
        reinterpret_cast<float4*>(g)[b4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND != DCCF_OPT_GD) reinterpret_cast<float4*>(s1)[b4 + c] = av;
        if (KIND == DCCF_OPT_ADAM) reinterpret_cast<float4*>(s2)[b4 + c] = bv;
      }
      if (sub == 0) {
        z.last[grow] = t;
        sg.flags[q][row] = 0;
      }
    }
    return;
  }
  bid -= lb;
  if (bid < db) {                              // ---- everything outside the row segments: dense, with its gradient
    const int64_t tid = (int64_t)bid * blockDim.x + threadIdx.x, stride = (int64_t)db * blockDim.x;
    for (int d = 0; d < ds.n; ++d)
      for (int64_t i = ds.begin[d] + tid; i < ds.end[d]; i += stride) {
        float pv = p[i], gv = g[i], av = 0.f, bv = 0.f;
        if (KIND != DCCF_OPT_GD) av = s1[i];
        if (KIND == DCCF_OPT_ADAM) bv = s2[i];
        if (gp.part && i >= gp.w_begin && i < gp.w_end) {
          // dW arrived as one partial sum per row split of the backward (plain stores there instead of 32 KB of float atomics
          // per CU): added here in split order — 8 loads in flight (32 were slower), unconditional from clamped indices
          const float* q = gp.part + (i - gp.w_begin);
          for (int r0 = 0; r0 < gp.nsplit; r0 += 8) {
            float tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tv[u] = q[(int64_t)min(r0 + u, gp.nsplit - 1) * gp.stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) gv = __fadd_rn(gv, r0 + u < gp.nsplit ? tv[u] : 0.f);
          }
        }
        opt_elem<KIND>(pv, gv, av, bv, a);
        p[i] = pv;
        g[i] = 0.f;
        if (KIND != DCCF_OPT_GD) s1[i] = av;
        if (KIND == DCCF_OPT_ADAM) s2[i] = bv;
        if (pn.blocks && i >= pn.w_begin && i < pn.w_end) prep_next_wt(pn, i, pv);      // W^T for the next step's forward
      }
    return;
  }
  bid -= db;
  const int wb = (int)gridDim.x - pn.blocks - pn.cu_blocks - lb - db - mb;
  if (bid < wb) {                              // ---- the window (unless the backward launch hosted it)
    // (winS > win0: the head of the window was advanced by workgroups hosted in this step's forward launch)
    lazy_window_pass<KIND>(p, s1, s2, a, sg, z, winS, win1, flush, bid, wb, blockDim.x, s_sct, sct_base);
    return;
  }
  bid -= wb;
  // ---- the PREVIOUS lazy step's window gets its marks now (that launch is complete), and this launch's window is recorded
  const LazyPend pend = lazy_pend_read(z);
  const int* __restrict__ claim = lazy_claim_of(z, t);
  for (int64_t r = pend.w0 + (int64_t)bid * blockDim.x + threadIdx.x; r < pend.w1; r += (int64_t)mb * blockDim.x)
    if (z.last[r] < pend.t && claim[r] != t) atomicMax(&z.last[r], pend.t);      // (a maximum: the catch-up role may be writing t)
  if (bid == 0 && threadIdx.x == 0) {
    int* w = z.cnt + 5 * (t & 1);
    w[0] = t;
    w[1] = (int)(uint32_t)win0; w[2] = (int)(win0 >> 32);
    w[3] = (int)(uint32_t)win1; w[4] = (int)(win1 >> 32);
  }
}

// last[row] = t for the rows of the window, after k_lazy_opt (every lane of a row must have read the old value first)
__global__ __launch_bounds__(256) void k_lazy_mark(LazyArgs z, int64_t win0, int64_t win1, int flush) {
  const int t = (int)z.step;
  for (int64_t r = win0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < win1; r += (int64_t)gridDim.x * blockDim.x)
    if (z.last[r] < t && (flush || lazy_claim_of(z, t)[r] != t)) z.last[r] = t;
  if (blockIdx.x == 0 && threadIdx.x < 10) z.cnt[threadIdx.x] = threadIdx.x % 5 == 0 ? -1 : 0;      // nothing is pending after a flush
}

// The lazy arrays are only consistent if the steps arrive one by one: no row may be more than lazy_K steps behind (the replay
// tables and the window schedule assume it).  lazy_host[0] (HOST, the caller's; optional) holds the last step whose optimizer
// launch went out: a launch that belongs to step s (LZ_PRE: before it, LZ_STEP: its optimizer launch) needs s == last + 1, a
// flush s == last; anything else is an argument error instead of a silent skip / an out-of-table read.
enum { LZ_ANY = 0, LZ_PRE = 1, LZ_STEP = 2, LZ_FLUSH = 3 };
static int lazy_args(const dccf_opt_t* o, const OptJob& j, LazyArgs* z, int mode = LZ_ANY) {
  if (o->lazy_host && mode != LZ_ANY) {
    const int64_t last = o->lazy_host[0];
    if (mode == LZ_FLUSH) ARG_CHECK(o->step == last, "lazy optimizer: a flush must name the last step that was launched (opt->step == lazy_host[0])");
    else ARG_CHECK(o->step == last + 1, "lazy optimizer: steps must arrive one by one (opt->step == lazy_host[0] + 1)");
  }
  ARG_CHECK(o->lazy_K >= 2 && o->lazy_K <= LAZY_KMAX && o->lazy_last && o->lazy_claim && o->lazy_list && o->lazy_cnt && o->lazy_list_cap > 0,
            "lazy optimizer: lazy_K >= 2 and all arrays");
  ARG_CHECK(j.sg.n >= 1, "lazy optimizer needs row segments");
  ARG_CHECK(o->kind != DCCF_OPT_ADAM || (o->lazy_scal && o->lazy_t0 <= max((int64_t)1, o->step - o->lazy_K + 1) &&
                                         o->step < o->lazy_t0 + o->lazy_nscal && (uintptr_t)o->lazy_scal % 16 == 0),
            "lazy optimizer: the step-scalar table does not cover [step - K + 1, step]");
  z->K = o->lazy_K; z->nscal = o->lazy_nscal; z->t0 = o->lazy_t0; z->step = o->step;
  z->pend_slot = (int)((o->step - 1) & 1);
  z->last = o->lazy_last; z->claim = o->lazy_claim; z->list = o->lazy_list; z->cnt = o->lazy_cnt; z->scal = o->lazy_scal;
  int64_t off = 0;
  for (int q = 0; q < 4; ++q) {
    z->row_off[q] = off;
    z->rows[q] = q < j.sg.n ? (j.sg.end[q] - j.sg.begin[q]) / j.sg.width[q] : 0;
    off += z->rows[q];
  }
  z->R = off;
  z->list_cap = o->lazy_list_cap;
  return 0;
}

int dccf_lazy_reset_claims(const void* ov, hipStream_t st) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  OptJob j;
  if (int e = opt_job(ov, &j)) return e;
  LazyArgs z;
  if (int e = lazy_args(o, j, &z)) return e;
  HIP_TRY(hipMemsetAsync(z.claim + (o->step & 1) * z.R, 0, (size_t)z.R * sizeof(int), st));
  return 0;
}

extern "C" int dccf_lazy_scalars(float lr, int64_t t0, int32_t n, float* out_host) {
  ARG_CHECK(out_host && n >= 0 && t0 >= 0, "bad arguments");
  for (int i = 0; i < n; ++i) {
    const double s = (double)(t0 + i);
    // (step 0 is never applied; its slot keeps the table aligned)
    const double bc1 = 1.0 - pow(0.9, s), bc2 = 1.0 - pow(0.999, s);
    const float c = s >= 1.0 ? (float)sqrt(bc2) : 1.f;
    out_host[4 * i] = s >= 1.0 ? (float)(-((double)lr / bc1)) : 0.f;
    out_host[4 * i + 1] = c;
    out_host[4 * i + 2] = (float)(1.0 / (double)c);
    out_host[4 * i + 3] = 0.f;
  }
  return 0;
}

int dccf_lazy_catchup(const void* ov, const int64_t* X, const int* cand, int64_t N, int S1, int segU, int segV, hipStream_t st) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  OptJob j;
  if (int e = opt_job(ov, &j)) return e;
  LazyArgs z;
  if (int e = lazy_args(o, j, &z, LZ_PRE)) return e;
  ARG_CHECK(segU >= 0 && segU < j.sg.n && segV >= 0 && segV < j.sg.n, "bad segment index");
  const int64_t slots = N * S1 + N;
  if (slots == 0) return 0;
  ARG_CHECK(slots <= z.list_cap, "lazy optimizer: the step's row list is too short (lazy_list_cap)");
  const int grid = (int)min((int64_t)2048, (slots + 3) / 4);
  BY_KIND(j.kind, k_lazy_catchup, dim3(grid), dim3(256), 0, st, j.p, j.s1, j.s2, j.a, j.sg, z, X, cand, N, S1, segU, segV);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dccf_lazy_catchup_rows(const dccf_opt_t* o, const int32_t* rows_a, int64_t n_a, int32_t seg_a, const int32_t* rows_b,
                                      int64_t n_b, int32_t seg_b, void* stream) {
  ARG_CHECK(o != nullptr && n_a >= 0 && n_b >= 0 && (n_a == 0 || rows_a) && (n_b == 0 || rows_b), "NULL opt / rows");
  OptJob j;
  if (int e = opt_job(o, &j)) return e;
  LazyArgs z;
  if (int e = lazy_args(o, j, &z, LZ_PRE)) return e;
  ARG_CHECK((n_a == 0 || (seg_a >= 0 && seg_a < j.sg.n)) && (n_b == 0 || (seg_b >= 0 && seg_b < j.sg.n)), "bad segment index");
  const int64_t slots = n_a + n_b;
  if (slots == 0) return 0;
  ARG_CHECK(slots <= z.list_cap, "lazy optimizer: the step's row list is too short (lazy_list_cap)");
  const int grid = (int)min((int64_t)2048, (slots + 3) / 4);
  BY_KIND(j.kind, k_lazy_catchup_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, j.p, j.s1, j.s2, j.a, j.sg, z, rows_a, n_a,
          (int)seg_a, rows_b, n_b, (int)seg_b);
  HIP_TRY(hipGetLastError());
  return 0;
}

int dccf_lazy_catchup_pairs(const void* ov, const int64_t* X, int64_t N, int seg_u, int seg_v, float* zero, hipStream_t st) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  ARG_CHECK(o != nullptr && X != nullptr && N >= 1, "NULL opt / X");
  OptJob j;
  if (int e = opt_job(o, &j)) return e;
  LazyArgs z;
  if (int e = lazy_args(o, j, &z, LZ_PRE)) return e;
  ARG_CHECK(seg_u >= 0 && seg_u < j.sg.n && seg_v >= 0 && seg_v < j.sg.n, "bad segment index");
  ARG_CHECK(2 * N <= z.list_cap, "lazy optimizer: the step's row list is too short (lazy_list_cap)");
  const int grid = (int)min((int64_t)2048, (2 * N + 3) / 4);
  BY_KIND(j.kind, k_lazy_catchup_pairs, dim3(grid), dim3(256), 0, st, j.p, j.s1, j.s2, j.a, j.sg, z, X, N, seg_u, seg_v, zero);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int lazy_launch(const dccf_opt_t* o, int flush, const PrepNext* pnp, int64_t nslots, hipStream_t st, const GwPart* gpp = nullptr,
                       int64_t win_from = -1) {
  OptJob j;
  if (int e = opt_job(o, &j)) return e;
  LazyArgs z;
  if (int e = lazy_args(o, j, &z, flush ? LZ_FLUSH : LZ_STEP)) return e;
  DenseSegs ds;
  int64_t dense_total = 0;
  if (int e = dense_complement(j.sg, j.n, &ds, &dense_total)) return e;
  if (flush) z.pend_slot = (int)(o->step & 1);       // the window of the step just done
  const int64_t R = z.row_off[3] + z.rows[3];
  const int64_t w = o->step % o->lazy_K;
  const int64_t win0 = flush ? 0 : R * w / o->lazy_K, win1 = flush ? R : R * (w + 1) / o->lazy_K;
  PrepNext pn;
  memset(&pn, 0, sizeof(pn));
  if (pnp) pn = *pnp;
  int maxw4 = 4;
  for (int q = 0; q < j.sg.n; ++q) maxw4 = max(maxw4, j.sg.width[q] >> 2);
  const int lb = (flush || nslots == 0) ? 0 : (int)min((int64_t)256, (nslots + 3) / 4);
  const int db = flush ? 0 : (int)min((int64_t)256, (dense_total + 255) / 256);
  static const int wb_cap = getenv("DCCF_LAZY_WB") ? max(1, atoi(getenv("DCCF_LAZY_WB"))) : 8192;
  const int64_t winS = (!flush && win_from > win0) ? min(win_from, win1) : win0;
  const int wb = (int)max((int64_t)1, min((int64_t)(flush ? 8192 : wb_cap), ((win1 - winS) * maxw4 + 255) / 256));
  const int mb = flush ? 0 : 32;       // (a flush marks with a launch of its own: every row, and nothing stays pending)
  ARG_CHECK(nslots <= z.list_cap && (pn.cu_blocks == 0 || pn.X_all || pn.N * (pn.S + 2) <= z.list_cap), "lazy optimizer: lazy_list_cap too small");
  if (flush) pn.cu_blocks = 0;
  const int grid = pn.blocks + pn.cu_blocks + lb + db + wb + mb;
  GwPart gp;
  memset(&gp, 0, sizeof(gp));
  if (gpp && !flush) gp = *gpp;
  BY_KIND(j.kind, k_lazy_opt, dim3(grid), dim3(256), 0, st, j.p, j.g, j.s1, j.s2, j.a, j.sg, ds, z, lb, db, mb, win0, win1, flush,
          (int)nslots, pn, gp, winS);
  if (flush)
    hipLaunchKernelGGL(k_lazy_mark, dim3((unsigned)max((int64_t)1, min((int64_t)1024, (win1 - win0 + 255) / 256))), dim3(256), 0, st, z,
                       win0, win1, flush);
  HIP_TRY(hipGetLastError());
  if (!flush && o->lazy_host) o->lazy_host[0] = o->step;
  return 0;
}

int dccf_lazy_step(const void* ov, const PrepNext* pn, int64_t nslots, hipStream_t st, const GwPart* gp, int64_t win_from) {
  return lazy_launch((const dccf_opt_t*)ov, 0, pn, nslots, st, gp, win_from);
}

int dccf_lazy_host_args(const void* ov, LazyHost* out) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  OptJob j;
  if (int e = opt_job(o, &j)) return e;
  memset(out, 0, sizeof(*out));
  if (int e = lazy_args(o, j, &out->z, LZ_PRE)) return e;
  out->p = j.p; out->s1 = j.s1; out->s2 = j.s2; out->a = j.a; out->sg = j.sg; out->kind = j.kind;
  const int64_t R = out->z.row_off[3] + out->z.rows[3];
  const int64_t w = o->step % o->lazy_K;
  out->win0 = R * w / o->lazy_K;
  out->win1 = R * (w + 1) / o->lazy_K;
  return 0;
}
extern "C" int dccf_lazy_opt_step(const dccf_opt_t* opt, int64_t nslots, void* stream) {
  ARG_CHECK(opt != nullptr && opt->lazy_K > 0 && nslots >= 0, "dccf_lazy_opt_step needs a lazy optimizer (lazy_K > 0)");
  return lazy_launch(opt, 0, nullptr, nslots, (hipStream_t)stream);
}

// Replicated multi-GPU path: the rows ANY rank touches at step t are known before the step as bytes (the replicated
// schedule): each flagged row is claimed for step t and brought up to step t - 1.  A wave scans 64 rows' bytes, then all its
// lanes serve the flagged rows one after the other.
template <int KIND>
__global__ __launch_bounds__(256) void k_lazy_catchup_flags(float* __restrict__ p, float* __restrict__ s1, float* __restrict__ s2,
                                                            OptArgs a, RowSegs sg, LazyArgs z, const uint8_t* __restrict__ f0,
                                                            const uint8_t* __restrict__ f1, int seg0, int seg1) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int t = (int)z.step;
  const LazyPend pend = lazy_pend_read(z);
  for (int k = 0; k < 2; ++k) {
    const int q = k ? seg1 : seg0;
    const uint8_t* fl = k ? f1 : f0;
    const int64_t rows = z.rows[q];
    const int w = sg.width[q];
    for (int64_t r0 = wave * 64; r0 < rows; r0 += nw * 64) {
      const int64_t r = r0 + lane;
      uint64_t bits = __ballot(r < rows && fl[r < rows ? r : 0] != 0);
      while (bits) {
        const int b = __ffsll((unsigned long long)bits) - 1;
        bits &= bits - 1;
        const int64_t row = r0 + b, grow = z.row_off[q] + row;
        const int from = __builtin_amdgcn_readfirstlane(lazy_from(z, pend, grow, t));
        if (lane == 0) lazy_claim_of(z, t)[grow] = t;
        if (from >= t - 1) continue;
        float* pr = p + sg.begin[q] + row * w;
        float* ar = s1 ? s1 + sg.begin[q] + row * w : nullptr;
        float* br = s2 ? s2 + sg.begin[q] + row * w : nullptr;
        for (int c = lane; c < w; c += 64) {
          float pv = pr[c], av = KIND != DCCF_OPT_GD ? ar[c] : 0.f, bv = KIND == DCCF_OPT_ADAM ? br[c] : 0.f;
          lazy_replay<KIND>(pv, av, bv, a, z, from, t - 1);
          pr[c] = pv;
          if (KIND != DCCF_OPT_GD) ar[c] = av;
          if (KIND == DCCF_OPT_ADAM) br[c] = bv;
        }
        if (lane == 0) z.last[grow] = t - 1;
      }
    }
  }
}

int dccf_lazy_catchup_flags(const void* ov, const uint8_t* flags0, const uint8_t* flags1, int seg0, int seg1, hipStream_t st) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  OptJob j;
  if (int e = opt_job(ov, &j)) return e;
  LazyArgs z;
  if (int e = lazy_args(o, j, &z, LZ_PRE)) return e;
  ARG_CHECK(flags0 && flags1 && seg0 >= 0 && seg0 < j.sg.n && seg1 >= 0 && seg1 < j.sg.n, "bad flags / segments");
  const int64_t rows = max(z.rows[seg0], z.rows[seg1]);
  const int grid = (int)max((int64_t)1, min((int64_t)2048, (rows + 255) / 256));
  BY_KIND(j.kind, k_lazy_catchup_flags, dim3(grid), dim3(256), 0, st, j.p, j.s1, j.s2, j.a, j.sg, z, flags0, flags1, seg0, seg1);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Phase 1 of a replicated step in lazy form: this step's window (claimed rows excepted), the previous window's marks and the
// next step's preparation — no list, no dense tail (the import applies those with the rank-ordered sums)
int dccf_lazy_phase1(const void* ov, const PrepNext* pnp, hipStream_t st) {
  const dccf_opt_t* o = (const dccf_opt_t*)ov;
  OptJob j;
  if (int e = opt_job(o, &j)) return e;
  LazyArgs z;
  if (int e = lazy_args(o, j, &z, LZ_STEP)) return e;
  DenseSegs ds;
  memset(&ds, 0, sizeof(ds));
  const int64_t R = z.row_off[3] + z.rows[3];
  const int64_t w = o->step % o->lazy_K;
  const int64_t win0 = R * w / o->lazy_K, win1 = R * (w + 1) / o->lazy_K;
  PrepNext pn;
  memset(&pn, 0, sizeof(pn));
  if (pnp) pn = *pnp;
  int maxw4 = 4;
  for (int q = 0; q < j.sg.n; ++q) maxw4 = max(maxw4, j.sg.width[q] >> 2);
  const int wb = (int)max((int64_t)1, min((int64_t)8192, ((win1 - win0) * maxw4 + 255) / 256));
  const int mb = 32;
  GwPart gp;
  memset(&gp, 0, sizeof(gp));
  BY_KIND(j.kind, k_lazy_opt, dim3(pn.blocks + pn.cu_blocks + wb + mb), dim3(256), 0, st, j.p, j.g, j.s1, j.s2, j.a, j.sg, ds, z, 0, 0, mb, win0, win1,
          0, 0, pn, gp, win0);
  HIP_TRY(hipGetLastError());
  if (o->lazy_host) o->lazy_host[0] = o->step;
  return 0;
}

// every row up to step - 1 (a step whose rows were not known in advance starts from a table that is current)
int dccf_lazy_flush_to_prev(const void* ov, hipStream_t st) {
  dccf_opt_t o = *(const dccf_opt_t*)ov;
  if (o.step <= 1) return 0;
  o.step -= 1;
  return lazy_launch(&o, 1, nullptr, 0, st);
}

extern "C" int dccf_lazy_flush(const dccf_opt_t* opt, void* stream) {
  ARG_CHECK(opt != nullptr && opt->lazy_K > 0, "dccf_lazy_flush needs a lazy optimizer (lazy_K > 0)");
  ARG_CHECK(opt->kind != DCCF_OPT_ADAM || opt->step < opt->lazy_t0 + opt->lazy_nscal, "step-scalar table too short");
  return lazy_launch(opt, 1, nullptr, 0, (hipStream_t)stream);
}

__global__ void k_advance(int64_t* k) { *k += 1; }
extern "C" int dccf_advance(int64_t* k_dev, void* stream) {
  ARG_CHECK(k_dev != nullptr, "k_dev is NULL");
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, k_dev);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int KIND, int IEEE>
__global__ void k_debug_opt_elem(float* p, float* g, float* s1, float* s2, int64_t n, OptArgs a, int denom_only) {
  if (denom_only == 2) {         // the four-at-a-time form (opt_elem4) on whole groups of four; the tail like mode 0
    for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n / 4; i4 += (int64_t)gridDim.x * blockDim.x) {
      float4 pv = reinterpret_cast<float4*>(p)[i4], gv = reinterpret_cast<float4*>(g)[i4];
      float4 av = s1 ? reinterpret_cast<float4*>(s1)[i4] : make_float4(0, 0, 0, 0);
      float4 bv = s2 ? reinterpret_cast<float4*>(s2)[i4] : make_float4(0, 0, 0, 0);
      opt_elem4<KIND>(pv, gv, av, bv, a);
      reinterpret_cast<float4*>(p)[i4] = pv; reinterpret_cast<float4*>(g)[i4] = gv;
      if (s1) reinterpret_cast<float4*>(s1)[i4] = av;
      if (s2) reinterpret_cast<float4*>(s2)[i4] = bv;
    }
    for (int64_t i = n / 4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      float pv = p[i], gv = g[i], av = s1 ? s1[i] : 0.f, bv = s2 ? s2[i] : 0.f;
      opt_elem<KIND>(pv, gv, av, bv, a);
      p[i] = pv; g[i] = gv;
      if (s1) s1[i] = av;
      if (s2) s2[i] = bv;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pv = p[i], gv = g[i], av = s1 ? s1[i] : 0.f, bv = s2 ? s2[i] : 0.f;
    if (denom_only) {      // Adam's denominator of the element's second moment AS GIVEN -> g (exhaustive checks of the first division)
      g[i] = IEEE ? __fadd_rn(__fdiv_rn(__fsqrt_rn(bv), a.bc2_sqrt), 1e-8f)
                  : __fadd_rn(div_by_uniform(sqrt_bare(bv), a.bc2_sqrt, a.bc2_rsqrt), 1e-8f);
      continue;
    }
    if (IEEE) opt_elem_ieee<KIND>(pv, gv, av, bv, a);
    else opt_elem<KIND>(pv, gv, av, bv, a);
    p[i] = pv; g[i] = gv;
    if (s1) s1[i] = av;
    if (s2) s2[i] = bv;
  }
}
extern "C" int dccf_debug_opt_elem(int32_t kind, int32_t ieee, float* p, float* g, float* s1, float* s2, int64_t n, float lr,
                                   float wd, float l2, float clip, int64_t step, void* stream) {
  ARG_CHECK(p && g && n >= 0 && step >= 1, "NULL p/g, n < 0 or step < 1");
  ARG_CHECK(kind == DCCF_OPT_GD || kind == DCCF_OPT_ADAGRAD || kind == DCCF_OPT_ADAM, "unknown optimizer kind");
  ARG_CHECK((kind == DCCF_OPT_GD || s1) && (kind != DCCF_OPT_ADAM || s2), "optimizer state is NULL");
  ARG_CHECK(ieee >= 0 && ieee <= 4 && (ieee < 2 || ieee == 4 || kind == DCCF_OPT_ADAM),
            "ieee is 0 / 1 (a step), 2 / 3 (Adam's denominator only) or 4 (a step, four elements at a time)");
  ARG_CHECK(ieee != 4 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)s1 | (uintptr_t)s2) & 15) == 0, "mode 4 needs 16-byte aligned arrays");
  const int denom_only = ieee == 4 ? 2 : (ieee >> 1);
  ieee = ieee == 4 ? 0 : (ieee & 1);
  ARG_CHECK(clip >= 0.f, "clip must be >= 0");
  if (n == 0) return 0;
  OptArgs a;
  a.lr = lr; a.wd = wd; a.l2 = l2; a.clip = clip; a.zero_grad = 0;
  a.k_dev = nullptr; a.step0 = step;
  const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
  a.step_size_neg = (float)(-((double)lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.bc2_rsqrt = (float)(1.0 / (double)a.bc2_sqrt);
  const int grid = (int)min((int64_t)4096, (n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
#define DBG_ELEM(K) { if (ieee) hipLaunchKernelGGL((k_debug_opt_elem<K, 1>), dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a, denom_only); \
                      else hipLaunchKernelGGL((k_debug_opt_elem<K, 0>), dim3(grid), dim3(256), 0, st, p, g, s1, s2, n, a, denom_only); }
  if (kind == DCCF_OPT_GD) DBG_ELEM(DCCF_OPT_GD) else if (kind == DCCF_OPT_ADAGRAD) DBG_ELEM(DCCF_OPT_ADAGRAD) else DBG_ELEM(DCCF_OPT_ADAM)
#undef DBG_ELEM
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
  ARG_CHECK(clip >= 0.f, "clip must be >= 0");
  if (n == 0) return 0;
  OptArgs a;
  a.lr = lr; a.wd = wd; a.l2 = l2; a.clip = clip; a.zero_grad = zero_grad;
  a.k_dev = nullptr; a.step0 = step;
  // bias corrections in double like torch's Python scalars (torch/optim/adam.py::_single_tensor_adam)
  const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
  a.step_size_neg = (float)(-((double)lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.bc2_rsqrt = (float)(1.0 / (double)a.bc2_sqrt);
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
