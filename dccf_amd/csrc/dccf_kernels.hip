// dccf_kernels.hip — DCCF.predict / DCCF.forward+backward as hand-written gfx950 kernels.
//
// Reference semantics: src/models/DCCF.py:66-127.  With N rows of X, S1 = S+1 candidates per row and
// A noise draws per candidate, the reference materialises L = N*S1*A rows x[l] = [V[cand(l)] ; feat[i0(l)] + eps[l]] of
// width D+F and pushes them through Linear(D+F -> D): z = x W^T + b.  Here the rows are never materialised:
//
//   forward  (k_noise_fwd)  Z = X W^T on the fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32).  A workgroup owns 32 rows l;
//            K (the NC feature chunks of 128 + the item part) is split evenly over its 8 waves, each keeps its slices of
//            W^T in registers; the A operand feat + eps is produced in registers (Philox4x32-10 + Box-Muller, in the lane
//            layout the MFMA wants).  Partials meet in LDS; the epilogue adds b, applies relu + dropout, stores h [L,D]
//            and the row dot m[l] = <U[u], h[l]>.
//   epilogue (k_pair_epilogue)  softmax over the candidates of Expo[u, cand] (one lane per candidate), prediction,
//            BPR / MSE loss and d loss / d m.  Its own kernel for predict and for large batches; at training batch sizes
//            every backward wave recomputes it for the batch row it walks (wave_dm) and the kernel disappears.
//   backward (k_bwd)  one barrier-free kernel of role waves that walk the batch rows: dW_f += dz^T eps with eps
//            REGENERATED from the same counters directly as the MFMA B operand (+ one k-step per batch row for the
//            feature term), dW_i += dz^T V[cand], dV[cand] += dz W_i, dU, db; 32x32 accumulators stay in registers over
//            the workgroup's whole range and leave through one shaped float-atomic pass.  Software-pipelined loads (a
//            role workgroup is one wave per SIMD).  Extra grid rows host a slice of the dense optimizer pass
//            (dccf_train_step), and in the replicated multi-GPU path the gradient rows go straight into the all-gather
//            buffer ("slot mode").
//   step     (run_dccf / dccf_train_step)  forward -> backward -> optimizer launch, which also prepares the next step
//            (candidates, exposures, W^T, item-row marks) when the caller names the next batch: no k_prep in steady state.
//
// HBM layout (fp32 row-major): U [user_num,D], V [item_num,D], W [D,D+F], b [D], feat [item_num,F], expo [user_num,
// item_num].  Workspace per call (ctx slab): cand int32 [N,S1]; WT [(D+FP),DP] = W transposed, zero padded (DP = D
// rounded to 32/64, FP = F rounded up to 2, 6 or 7 chunks of 128); h [L,DP]; m [L]; dmns [N*S1] (holds the gathered
// exposures Expo[u, cand] between k_prep and k_pair_epilogue, d loss / d mean_a m afterwards).
#include "common.hpp"
#include "opt_device.hpp"

// Tuning knobs (environment, read once): the defaults are the measured optima of DESIGN.md section 4.
struct Knobs {
  double hostv_frac;      // DCCF_HOSTV_FRAC   share of the item table whose untouched-row pass rides in the backward launch
  int64_t fold_max_n;     // DCCF_FOLD_MAX_N   largest 2B for which the pair epilogue is folded into the backward
  int64_t hosted_wgs;     // DCCF_HOSTED_WGS   workgroups of a hosted optimizer pass
  bool hostv;             // DCCF_NO_HOSTV=1   turns the hosted item-table pass off
  int64_t bwd_wgs;        // DCCF_BWD_WGS      role workgroups of the backward at 2B < 2048 (row splits = this / roles)
  bool lazy_cu;           // DCCF_LAZY_NO_CU=1 the lazy optimizer launch does not catch up the next step's rows (a launch of its own does)
  bool gw_part;           // DCCF_NO_GW_PART=1 dW of a lazy step as float atomics into gW (not per-split partial sums)
  int64_t lazy_cu_blocks; // DCCF_LAZY_CU_BLOCKS workgroups of the next-step catch-up role (256: 128 / 512 / 768 measured, no better)
  double lazy_host_frac;  // DCCF_LAZY_HOST_FRAC share of a lazy step's window advanced by idle workgroup slots of the forward launch (0: off)
  int64_t lazy_host_blocks; // DCCF_LAZY_HOST_BLOCKS cap on those workgroups (default: every CU the tiles leave idle)
};
static const Knobs& knobs() {
  static const Knobs k = [] {
    Knobs q;
    q.hostv_frac = getenv("DCCF_HOSTV_FRAC") ? atof(getenv("DCCF_HOSTV_FRAC")) : 0.75;
    q.fold_max_n = getenv("DCCF_FOLD_MAX_N") ? atoll(getenv("DCCF_FOLD_MAX_N")) : 1024;
    q.hosted_wgs = getenv("DCCF_HOSTED_WGS") ? atoll(getenv("DCCF_HOSTED_WGS")) : 256;
    q.hostv = getenv("DCCF_NO_HOSTV") == nullptr;
    q.bwd_wgs = getenv("DCCF_BWD_WGS") ? atoll(getenv("DCCF_BWD_WGS")) : 256;
    q.lazy_cu = getenv("DCCF_LAZY_NO_CU") == nullptr;
    q.gw_part = getenv("DCCF_NO_GW_PART") == nullptr;
    q.lazy_cu_blocks = getenv("DCCF_LAZY_CU_BLOCKS") ? max(1, atoi(getenv("DCCF_LAZY_CU_BLOCKS"))) : 256;
    q.lazy_host_frac = getenv("DCCF_LAZY_HOST_FRAC") ? atof(getenv("DCCF_LAZY_HOST_FRAC")) : 0.25;
    q.lazy_host_blocks = getenv("DCCF_LAZY_HOST_BLOCKS") ? max(1, atoi(getenv("DCCF_LAZY_HOST_BLOCKS"))) : 256;
    return q;
  }();
  return k;
}

// No implicit FMA contraction in this file: a*b+c written as two operations stays two roundings (explicit fmaf() calls
// are still FMAs).  It keeps "fused draws == injected draws" bit for bit and the optimizer in torch's op order.
#pragma clang fp contract(off)

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// Pins a loaded value as "used here": hipcc otherwise sinks a load into the only branch that consumes it
// (`cond ? f(load) : 0` becomes s_cbranch_execz + load + s_waitcnt vmcnt(0)), serialising a k-loop on memory latency.
#define KEEP(x) asm volatile("" ::"v"(x))

struct Lay {
  int DP, FP, NC, S1, ND, GY, DT;      // DT: the kernels' column tile (template D_) for a model of width D
  int64_t N, L, NS;
  size_t cand, WT, h, m, dmns, total;
  size_t hx, dh;             // n_layers > 1: outputs of the extra layers [NX][L, DP]; two d loss / d h buffers [2][L, DP]
};

static Lay make_layout(int64_t N, int D, int F, int S, int A, int NX = 0) {
  Lay y;
  y.S1 = S + 1;
  y.N = N;
  y.NS = N * y.S1;
  y.L = y.NS * A;
  y.DT = D <= 16 ? 16 : (D <= 32 ? 32 : (D <= 64 ? 64 : (D <= 128 ? 128 : 256)));
  y.DP = y.DT <= 32 ? 32 : y.DT;                 // the row stride of h / W^T = the kernels' compile-time DP of that tile
  y.NC = (int)align_up(F, 128) / 128;
  // W^T is zero-padded to the forward's compile-time chunk count: 2, 6 or 7 chunks in one pass, passes of 6 beyond (F > 896)
  y.FP = (y.NC <= 2 ? 2 : (y.NC <= 6 ? 6 : (y.NC == 7 ? 7 : (y.NC + 5) / 6 * 6))) * 128;
  if (D != y.DT || D > 128) y.FP = (y.NC + 5) / 6 * 6 * 128;   // (fewer kernel instances; the 256 tile is run-time width only)
  y.ND = y.DP == 32 ? 1 : 2;
  y.GY = y.DP == 32 ? 1 : (D + 63) / 64;         // 64-column groups that hold real columns (a 192-wide model in the 256 tile: 3)
  size_t o = 0;
  y.cand = o;  o += align_up((size_t)y.NS * 4, 256);
  y.WT = o;    o += align_up((size_t)(D + y.FP) * y.DP * 4, 256);
  y.h = o;     o += align_up((size_t)y.L * y.DP * 4, 256);
  y.m = o;     o += align_up((size_t)y.L * 4, 256);
  y.dmns = o;  o += align_up((size_t)y.NS * 4, 256);
  y.hx = o;    o += (size_t)NX * align_up((size_t)y.L * y.DP * 4, 256);
  y.dh = o;    o += (size_t)(NX ? 2 : 0) * align_up((size_t)y.L * y.DP * 4, 256);
  y.total = o;
  return y;
}

// ================================================================================================ K0: prep
// WT[k][d] = W[d][k] (zero padded), cand[n][0] = true item, cand[n][s] = injected or Philox candidate
// (models/DCCF.py:72-74), eg[n][s] = Expo[u(n), cand[n][s]] (gathered HERE, before the optimizer side stream starts to
// saturate HBM: the epilogue's dependent gathers would otherwise queue behind it), m = 0 (only when two column halves
// add into it), loss = 0.
__global__ void k_prep(dccf_model_t M, float* __restrict__ WT, int D, int F, int DP, int FP,
                       const int64_t* X, const int64_t* __restrict__ sample_item, int* __restrict__ cand,
                       float* __restrict__ eg, int64_t N, int S, int64_t item_num, int fused, rng_key key,
                       float* __restrict__ m, int64_t Lm, float* __restrict__ loss, StepRef sr, MarkPlan mp,
                       uint8_t* __restrict__ markV) {
  const float* __restrict__ W = M.W;
  {
    const int64_t k = step_k(sr);
    X = step_X(sr, X, k);
    key = key_plus(key, k);
  }
  if (mp.list && blockIdx.x == 0 && threadIdx.x == 0) *mp.cnt_next = 0;    // the NEXT step's counter (double-buffered)
  const int64_t nWT = (int64_t)(D + FP) * DP;
  const int64_t NS = N * (S + 1);
  const int64_t total = nWT + NS + Lm + 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nWT) {
      const int k = (int)(i / DP), d = (int)(i % DP);
      WT[i] = (d < D && k < D + F) ? W[(int64_t)d * (D + F) + k] : 0.f;
    } else if (i < nWT + NS) {
      const int64_t j = i - nWT;
      const int64_t n = j / (S + 1);
      const int s = (int)(j % (S + 1));
      const int64_t it = prep_cand(M, X, sample_item, cand, eg, j, S, item_num, fused, key);
      if (markV) markV[it] = 1;         // hosted item table: the rows this step touches, known before its backward
      if (mp.list) {        // overlapped step: every row this batch reads/updates, once (wave-aggregated append)
        mark_row(mp.flagV, it, mp.tagV, mp);
        if (s == 0) mark_row(mp.flagU, X[2 * n], mp.tagU, mp);
      }
    } else if (i < nWT + NS + Lm) {
      m[i - nWT - NS] = 0.f;
    } else if (loss) {
      loss[0] = 0.f;
    }
  }
}

#ifdef DCCF_TRACE
// Development aid (never built by default): wall-clock stamps (100 MHz) of workgroup 0's waves at phase boundaries.
__device__ long long dccf_trace[8 * 16];
#define TRACE(slot) if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) dccf_trace[(threadIdx.x >> 6) * 16 + (slot)] = wall_clock64()
__device__ long long dccf_trace_b[8 * 8 * 8];     // backward: [role][wave][slot]
#define TRACEB(role, slot) if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (role) < 8) dccf_trace_b[(((role) * 8) + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64()
extern "C" int dccf_debug_trace_read_b(long long* out) {
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(dccf_trace_b), sizeof(long long) * 8 * 8 * 8));
  return 0;
}
extern "C" int dccf_debug_trace_read(long long* out) {
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(dccf_trace), sizeof(long long) * 8 * 16));
  return 0;
}
#else
#define TRACE(slot)
#define TRACEB(role, slot)
#endif

// Reductions over groups of GS consecutive lanes (GS a power of two, groups aligned).  Inside a 16-lane row they are DPP
// operand modifiers of the add / max itself — quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: after
// each step every lane of a 2 / 4 / 8 / 16-lane block holds the block's value — instead of ds_bpermute round trips through the
// LDS crossbar (what __shfl_xor compiles to: ~100 cycles each in a dependent chain; the folded pair epilogue of k_bwd runs 16
// of them per batch row with nothing to overlap them at one wave per SIMD).
__device__ __forceinline__ float dpp_quad_1032(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_quad_2301(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_half_mirror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_mirror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
}
template <int GS>
__device__ __forceinline__ float group_sum(float v) {
  if (GS >= 2) v += dpp_quad_1032(v);
  if (GS >= 4) v += dpp_quad_2301(v);
  if (GS >= 8) v += dpp_half_mirror(v);
  if (GS >= 16) v += dpp_mirror(v);
#pragma unroll
  for (int o = 16; o < GS; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int GS>
__device__ __forceinline__ float group_max(float v) {
  if (GS >= 2) v = fmaxf(v, dpp_quad_1032(v));
  if (GS >= 4) v = fmaxf(v, dpp_quad_2301(v));
  if (GS >= 8) v = fmaxf(v, dpp_half_mirror(v));
  if (GS >= 16) v = fmaxf(v, dpp_mirror(v));
#pragma unroll
  for (int o = 16; o < GS; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ================================================================================================ K1: forward
// Workgroup = 8 waves, one 32-row tile per iteration, K split EVENLY over the waves (two waves share a SIMD and the
// fp32 MFMA shares the VALU pipe with the generator, so an uneven split leaves SIMDs idle: measured 15 us for the two
// SIMDs that held two 128-wide feature chunks against 9 us for the one that held one).
//   A "group" is 4 k-steps.  Feature groups: q = (tq, c2), tq < NC the 128-wide chunk, c2 < 16; lane (row = lane&31,
//   h = lane>>5) handles f = 128*tq + (2*c2+h) + 32*o, o = 0..3 — ONE Philox call yields the 4 normals eps(l, f).  Wave w
//   owns feature groups [w*2*NC, (w+1)*2*NC) (q = 16*tq + c2) and, while they last, item group w (k = 8*w + 2*o + h of
//   V[cand]) — its slices of W^T stay in registers.  Partials meet in LDS; the epilogue adds b, applies relu + dropout,
//   stores h and m[l] = <U[u], h[l]>; its operands (user row, dropout draw, bias) are fetched before the k-loop.
// NCM >= NC is the compile-time number of chunks (W^T is zero-padded to it: a group past F multiplies zeros).
// D_ is the column TILE (16, 32, 64 or 128); the model's real width Dr <= D_ (src/models/RecModel.py:17-27 accepts any) is a
// run-time value: it is the stride of U / V / keep rows and of the feature part inside W^T, columns d >= Dr and item
// k-steps k >= Dr multiply zeros.  MP (multi-pass, F > 896): the feature chunks are walked in passes of NCM, the wave's W^T
// slice is re-read per pass (the register-resident slice of the single-pass form does not fit).
template <int D_, int MODE, int NCM, bool FAL, bool MP, bool GEN>   // MODE 0: fused Philox draws, 1: injected noise / keep mask;
                                                 // GEN: Dr < D_ (run-time row width); else Dr == D_ at compile time
                                                 // FAL: F == 128 * NCM exactly (no column clamp in the k-loop)
__global__ __launch_bounds__(512) void k_noise_fwd(const float* __restrict__ WT, const float* __restrict__ bias,
                                                   const float* __restrict__ U, const float* __restrict__ V,
                                                   const float* __restrict__ feat, const int64_t* X,
                                                   const int* __restrict__ cand, const float* __restrict__ noise,
                                                   const uint8_t* __restrict__ keep, float* __restrict__ hbuf,
                                                   float* __restrict__ m, int64_t L, int S1,
                                                   int A, int F, rng_key nkey, rng_key dkey, float nscale,
                                                   uint32_t drop_thr, float kscale, StepRef sr, int store_h,
                                                   float* __restrict__ zero1, int Dr_, int npass, int tile_blocks, LazyHost lh) {
  // workgroups past the tiles (lazy training step at small batches: 176 tiles leave 80 CUs idle): the head of this step's lazy
  // window, advanced here instead of in the optimizer launch that follows (opt_device.hpp: LazyHost)
  extern __shared__ float zpart[];   // [8][32][DW]
  if (MODE == 0 && !GEN && !MP && lh.blocks && (int)blockIdx.x >= tile_blocks) {
    const int bid = (int)blockIdx.x - tile_blocks;
    // the step scalars of the steps the replay can reach, copied to LDS first (a broadcast ds_read per step instead of a global
    // load inside every lane's dependent chain — what k_lazy_opt does too)
    float4* sct = reinterpret_cast<float4*>(zpart);
    const int sct_base = max((int)lh.z.step - lh.z.K, 0);
    if (lh.kind == DCCF_OPT_ADAM) {
      const int s = sct_base + (int)threadIdx.x;
      if ((int)threadIdx.x <= lh.z.K && s >= (int)lh.z.t0 && s <= (int)lh.z.step) sct[threadIdx.x] = reinterpret_cast<const float4*>(lh.z.scal)[s - lh.z.t0];
      __syncthreads();
    }
    if (lh.kind == DCCF_OPT_GD) lazy_hosted_window<DCCF_OPT_GD>(lh, bid, 512, sct, sct_base);
    else if (lh.kind == DCCF_OPT_ADAGRAD) lazy_hosted_window<DCCF_OPT_ADAGRAD>(lh, bid, 512, sct, sct_base);
    else lazy_hosted_window<DCCF_OPT_ADAM>(lh, bid, 512, sct, sct_base);
    return;
  }
  const int Dr = GEN ? Dr_ : D_;
  TRACE(0);
  // prepared step (no k_prep ran): the loss accumulator k_pair_epilogue adds into starts at 0
  if (zero1 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero1 = 0.f;
  {
    const int64_t k = step_k(sr);
    X = step_X(sr, X, k);
    nkey = key_plus(nkey, k);
    dkey = key_plus(dkey, k);
  }
  constexpr int D = D_;
  constexpr int DP = D <= 32 ? 32 : (D + 63) / 64 * 64;
  constexpr int ND = D <= 32 ? 1 : 2;
  constexpr int DW = ND * 32;
  constexpr int NW = 8;
  constexpr int NGF = 2 * NCM;       // feature groups per wave
  constexpr int GI = D / 8;          // item groups in total (KI = D/2 k-steps of 2 columns)
  constexpr int NGI = (GI + NW - 1) / NW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int dbase = blockIdx.y * DW;
  // register-resident B operands
  float bf[NGF * 4 * ND], bi[NGI * 4 * ND];
  // FAL (F a whole number of chunks, feature rows 16-byte aligned): the wave's feature k-steps come in SUPER-GROUPS of 16 — lane
  // (row, h) of super-group (tq, sg) takes the columns f = 128 tq + 32 o + m, m = 8 sg + 4 h + j, o, j < 4: four runs of four
  // CONSECUTIVE columns = four dwordx4 loads instead of sixteen dword loads that each touched a different line of the row (the
  // plain form's group takes f = 128 tq + (2 c2 + h) + 32 o: every 128-byte line of a feature row was requested by 16 different
  // instructions; scripts/bf16x6_bench.hip: 20.4 -> 17.6 us for the B = 128 tile loop, 542 -> 456 us at evaluation size).  The
  // noise keeps its definition — eps(l, 128 tq + m + 32 o) = word o of Philox(l, 32 tq + m) — so the four calls m = 8 sg + 4 h + j
  // supply exactly the 16 values; only the ORDER of the k-steps (fp32 summation order) differs from the plain form.
  constexpr bool SG = FAL && (NCM % 2 == 0);
#define FWD_LOAD_BF(PASS)                                                                                             \
  if (SG) {                                                                                                           \
    _Pragma("unroll") for (int sl = 0; sl < NGF / 4; ++sl) {                                                          \
      const int sq = wave * (NGF / 4) + sl, tq = (PASS) * NCM + (sq >> 2), sg = sq & 3;                               \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
        _Pragma("unroll") for (int o = 0; o < 4; ++o)                                                                 \
          _Pragma("unroll") for (int nt = 0; nt < ND; ++nt)                                                           \
            bf[((sl * 4 + j) * 4 + o) * ND + nt] =                                                                    \
                WT[(int64_t)(Dr + tq * 128 + 32 * o + 8 * sg + 4 * h + j) * DP + dbase + nt * 32 + c31];              \
    }                                                                                                                 \
  } else {                                                                                                            \
  _Pragma("unroll") for (int g = 0; g < NGF; ++g) {                                                                   \
    const int q = wave * NGF + g, tq = (PASS) * NCM + (q >> 4), c2 = q & 15;                                         \
    _Pragma("unroll") for (int o = 0; o < 4; ++o)                                                                     \
      _Pragma("unroll") for (int nt = 0; nt < ND; ++nt)                                                               \
        bf[(g * 4 + o) * ND + nt] = WT[(int64_t)(Dr + tq * 128 + 2 * c2 + h + 32 * o) * DP + dbase + nt * 32 + c31]; \
  }                                                                                                                   \
  }
  if (!MP) { FWD_LOAD_BF(0) }
#pragma unroll
  for (int gi = 0; gi < NGI; ++gi) {
    const int jg = wave * NGI + gi;
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int nt = 0; nt < ND; ++nt) {
        const int kk = 2 * (jg * 4 + o) + h;         // item k-step: rows >= Dr of W^T belong to the feature part
        const float wv = WT[(int64_t)min(kk, Dr - 1) * DP + dbase + nt * 32 + c31];
        bi[(gi * 4 + o) * ND + nt] = (jg < GI && kk < Dr) ? wv : 0.f;
      }
  }
  const uint32_t rows_per_n = (uint32_t)(S1 * A);
  const int64_t ntiles = (L + 31) / 32;
  const int dcol = dbase + lane;
  const bool dv = lane < DW && dcol < Dr;
  const float bias_d = bias[dv ? dcol : 0];
  TRACE(1);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += tile_blocks) {
    // epilogue operands of this wave's 4 rows: independent of the k-loop, fetched first.
    // !GEN: the wave takes the CONSECUTIVE rows 4 wave .. 4 wave + 3 — ONE Philox call per lane (= column) then yields the
    // dropout draws of all four (a call covers 4 consecutive rows of a column), kept as four wave-uniform column masks — and in
    // the epilogue a lane owns 4 consecutive columns of one of them (RPL lanes per row), so that h leaves as ONE dwordx4 store
    // instruction per wave instead of four dword ones (a CU retires a dword-per-lane store only every ~40 ns).
    // GEN (run-time row width, rows not 16-byte aligned): rows wave, wave + 8, ..., a lane per column.
    constexpr int RPL = DW / 4;
    const int erow = min(lane / RPL, 3);
    const bool eact = lane / RPL < 4 && dbase + 4 * (lane % RPL) < Dr;      // (D = 16: half of the tile's 32 columns are padding)
    const int ecol = eact ? 4 * (lane % RPL) : 0;
    float uval[4];
    uint32_t kbits = 0;
    if (!GEN) {
      const int64_t base = tile * 32 + 4 * wave;
      const int64_t lre = base + erow, lrec = lre < L ? lre : (L - 1);
      const int64_t ue = X[2 * (int64_t)((uint32_t)lrec / rows_per_n)];
      const float4 u4 = *reinterpret_cast<const float4*>(&U[ue * Dr + dbase + ecol]);
      uval[0] = u4.x; uval[1] = u4.y; uval[2] = u4.z; uval[3] = u4.w;
      uint64_t km[4];
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t lrc = base + i < L ? base + i : (L - 1);
          km[i] = __ballot(dv && (keep ? keep[lrc * Dr + (dv ? dcol : 0)] != 0 : true));
        }
      } else if (drop_thr) {
        const u32x4 r4 = philox4x32_10((uint32_t)(base >> 2), (uint32_t)(dv ? dcol : 0), dkey.s0, dkey.s1, dkey.k0, dkey.k1);
        km[0] = __ballot(dv && r4.x >= drop_thr);
        km[1] = __ballot(dv && r4.y >= drop_thr);
        km[2] = __ballot(dv && r4.z >= drop_thr);
        km[3] = __ballot(dv && r4.w >= drop_thr);
      } else {
        km[0] = km[1] = km[2] = km[3] = ~0ull;
      }
      const uint64_t mine = erow == 0 ? km[0] : (erow == 1 ? km[1] : (erow == 2 ? km[2] : km[3]));
      kbits = (uint32_t)(mine >> ecol) & 15u;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t lr = tile * 32 + wave + 8 * i;
        const int64_t lrc = lr < L ? lr : (L - 1);
        const int64_t u = X[2 * (int64_t)((uint32_t)lrc / rows_per_n)];
        uval[i] = U[u * Dr + (dv ? dcol : 0)];
        bool kept = true;
        if (MODE == 1) {
          if (keep) kept = keep[lrc * Dr + (dv ? dcol : 0)] != 0;
        } else if (drop_thr) {
          const u32x4 r4 = philox4x32_10((uint32_t)(lrc >> 2), (uint32_t)(dv ? dcol : 0), dkey.s0, dkey.s1, dkey.k0, dkey.k1);
          kept = pick4(r4, (int)(lrc & 3)) >= drop_thr;
        }
        kbits |= (kept ? 1u : 0u) << i;
      }
    }
    const int64_t l = tile * 32 + c31;
    const bool lv = l < L;
    f32x16 acc[ND];
#pragma unroll
    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    // NOTE: no load sits inside a conditional expression: hipcc branches around such a load and waits vmcnt(0) for it,
    // which serialises the k-steps.  Rows past L read row L-1 (their results are never stored); columns past F read
    // column F-1 (their W rows are zero).
    const int64_t lc = lv ? l : L - 1;
    {
      const int64_t it0 = X[2 * (int64_t)((uint32_t)lc / rows_per_n) + 1];
      const float* frow = feat + it0 * F;
      const float* nrow = MODE == 1 ? noise + lc * F : nullptr;
      int fbase = h;
      asm volatile("" : "+v"(fbase));               // opaque per tile: the clamped offsets below must not be hoisted
      for (int pass = 0; pass < (MP ? npass : 1); ++pass) {
      if (MP) { FWD_LOAD_BF(pass) }
      if (SG) {
        int fb4 = 4 * h;
        asm volatile("" : "+v"(fb4));               // (opaque per tile, as fbase)
#pragma unroll
        for (int sl = 0; sl < NGF / 4; ++sl) {
          const int sq = wave * (NGF / 4) + sl, tq = pass * NCM + (sq >> 2), sg = sq & 3;
          float4 fv4[4], nv4[4];
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            const int f0 = fb4 + tq * 128 + 32 * o + 8 * sg;
            fv4[o] = *reinterpret_cast<const float4*>(frow + f0);
            if (MODE == 1) nv4[o] = *reinterpret_cast<const float4*>(nrow + f0);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float a[4];
            if (MODE == 0) noise4((uint32_t)l, (uint32_t)(tq * 32 + 8 * sg + 4 * h + j), nkey, nscale, a);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
              const float fe = j == 0 ? fv4[o].x : (j == 1 ? fv4[o].y : (j == 2 ? fv4[o].z : fv4[o].w));
              if (MODE == 1) a[o] = j == 0 ? nv4[o].x : (j == 1 ? nv4[o].y : (j == 2 ? nv4[o].z : nv4[o].w));
              const float av = fe + a[o];           // sample_feature_embeddings = feature + noise (DCCF.py:87)
#pragma unroll
              for (int nt = 0; nt < ND; ++nt) acc[nt] = MFMA32(av, bf[((sl * 4 + j) * 4 + o) * ND + nt], acc[nt]);
            }
          }
        }
      } else
#pragma unroll                                      // out of the tile loop and kept live (they would spill the W slice)
      for (int g = 0; g < NGF; ++g) {
        const int q = wave * NGF + g, tq = pass * NCM + (q >> 4), c2 = q & 15;
        float a[4], fv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          // (F a whole number of chunks: no clamp, and the constant part of the column folds into the load's immediate
          // offset — the pipe these kernels are bound by issues every one of those instructions)
          const int f = FAL ? fbase + tq * 128 + 2 * c2 + 32 * o : min(fbase + tq * 128 + 2 * c2 + 32 * o, F - 1);
          fv[o] = frow[f];
          if (MODE == 1) a[o] = nrow[f];
        }
        if (MODE == 0) noise4((uint32_t)l, (uint32_t)(tq * 32 + 2 * c2 + h), nkey, nscale, a);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          a[o] = fv[o] + a[o];                      // sample_feature_embeddings = feature + noise (DCCF.py:87)
#pragma unroll
          for (int nt = 0; nt < ND; ++nt) acc[nt] = MFMA32(a[o], bf[(g * 4 + o) * ND + nt], acc[nt]);
        }
      }
      }
    }
    {
      const float* vrow = V + (int64_t)cand[(uint32_t)lc / (uint32_t)A] * Dr;
#pragma unroll
      for (int gi = 0; gi < NGI; ++gi) {
        const int jg = min(wave * NGI + gi, GI - 1);      // waves past the item part multiply zeros (bi = 0)
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const float a = vrow[min(2 * (jg * 4 + o) + h, Dr - 1)];    // (k >= Dr: a real element times a zero of W^T)
#pragma unroll
          for (int nt = 0; nt < ND; ++nt) acc[nt] = MFMA32(a, bi[(gi * 4 + o) * ND + nt], acc[nt]);
        }
      }
    }
    KEEP(acc[0][0]);
    TRACE(2);
#pragma unroll
    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        zpart[(wave * 32 + row) * DW + nt * 32 + c31] = acc[nt][r];
      }
    __syncthreads();
    TRACE(3);
    if (!GEN) {
      const int row = 4 * wave + erow;
      const int64_t lr = tile * 32 + row;
      const bool on = eact && lr < L;
      float part = 0.f;
      if (on) {
        float4 z = *reinterpret_cast<const float4*>(&zpart[row * DW + ecol]);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const float4 t = *reinterpret_cast<const float4*>(&zpart[(w * 32 + row) * DW + ecol]);
          z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w;
        }
        const float4 b4 = *reinterpret_cast<const float4*>(&bias[dbase + ecol]);
        z.x += b4.x; z.y += b4.y; z.z += b4.z; z.w += b4.w;
        float4 hv;
        hv.x = (z.x > 0.f && (kbits & 1u)) ? z.x * kscale : 0.f;
        hv.y = (z.y > 0.f && (kbits & 2u)) ? z.y * kscale : 0.f;
        hv.z = (z.z > 0.f && (kbits & 4u)) ? z.z * kscale : 0.f;
        hv.w = (z.w > 0.f && (kbits & 8u)) ? z.w * kscale : 0.f;
        if (store_h) *reinterpret_cast<float4*>(&hbuf[lr * DP + dbase + ecol]) = hv;          // the backward's input
        part = uval[0] * hv.x + uval[1] * hv.y + uval[2] * hv.z + uval[3] * hv.w;
      }
      part = group_sum<RPL>(part);          // (DPP inside the 16-lane row: no LDS round trip)
      if (on && (lane % RPL) == 0) {
        if (gridDim.y == 1) m[lr] = part;
        else atomicAdd(&m[lr], part);
      }
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 8 * i;
      const int64_t lr = tile * 32 + row;
      if (lr < L) {
        float part = 0.f;
        if (dv) {
          float z = zpart[row * DW + lane];
#pragma unroll
          for (int w = 1; w < NW; ++w) z += zpart[(w * 32 + row) * DW + lane];
          z += bias_d;
          const float hv = (z > 0.f && ((kbits >> i) & 1u)) ? z * kscale : 0.f;
          if (store_h) hbuf[lr * DP + dcol] = hv;          // the backward's input; evaluation does not need it
          part = uval[i] * hv;
        }
        part = wave_sum(part);
        if (lane == 0) {
          if (gridDim.y == 1) m[lr] = part;
          else atomicAdd(&m[lr], part);
        }
      }
    }
    }
    TRACE(4);
    __syncthreads();
  }
  TRACE(5);
#undef FWD_LOAD_BF
}

// ================================================================================================ K2: pair epilogue
// One lane per candidate: softmax_s(Expo[u, cand[n,s]]) (DCCF.py:98), prediction = mean_a sum_s w m (DCCF.py:100),
// loss and d loss / d m (DCCF.py:116-125).  A group of GS lanes serves one pair (rank 1) or one row.

template <int GS>
__device__ __forceinline__ float row_predict(const float* eg, const float* m, int64_t n, int s, int S1, int A,
                                             float& w_over_A) {
  const bool valid = s < S1;
  const float e = valid ? eg[n * S1 + s] : -INFINITY;
  const float mx = group_max<GS>(e);
  const float ex = valid ? expf(e - mx) : 0.f;
  const float w = ex / group_sum<GS>(ex);
  float tot = 0.f;
  for (int a = 0; a < A; ++a) tot += group_sum<GS>(valid ? w * m[(n * S1 + s) * A + a] : 0.f);
  w_over_A = w / (float)A;
  return tot / (float)A;
}

template <int GS>
__global__ __launch_bounds__(256) void k_pair_epilogue(int S1, int A, const float* __restrict__ Y,
                                                       const float* __restrict__ m, float* dmns,
                                                       float* __restrict__ pred, float* __restrict__ loss, int64_t N,
                                                       int rank, int train) {
  // dmns[n][s] arrives holding Expo[u, cand] (k_prep) and leaves holding d loss / d (mean_a m) — same lane, same slot
  const int s = threadIdx.x % GS;
  const int64_t gid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / GS;
  const int64_t ngr = (int64_t)gridDim.x * blockDim.x / GS;
  const bool pairs = train && rank == 1;
  const int64_t units = pairs ? N / 2 : N;
  const int64_t rounds = (units + ngr - 1) / ngr;      // every group runs the same number of rounds (shuffles are
  float lsum = 0.f;                                    // wave-wide: no lane may leave early)
  for (int64_t it = 0; it < rounds; ++it) {
    const int64_t k0 = it * ngr + gid;
    const bool live = k0 < units;
    const int64_t k = live ? k0 : 0;
    if (pairs) {
      const int64_t B = N / 2;
      float wp, wn;
      const float pp = row_predict<GS>(dmns, m, k, s, S1, A, wp);
      const float pn = row_predict<GS>(dmns, m, B + k, s, S1, A, wn);
      const float d = pp - pn;
      const float sg = 1.f / (1.f + expf(-d));
      const float gp = -(1.f - sg);                 // d loss / d pos = -sigmoid(neg - pos)
      if (live && s < S1) {
        dmns[k * S1 + s] = wp * gp;
        dmns[(B + k) * S1 + s] = wn * (-gp);
      }
      if (live && s == 0) {
        pred[k] = pp;
        pred[B + k] = pn;
        lsum += -logf(sg);
      }
    } else {
      float w;
      const float p = row_predict<GS>(dmns, m, k, s, S1, A, w);
      if (live && s == 0) pred[k] = p;
      if (train && live) {
        const float diff = p - Y[k];
        if (s < S1) dmns[k * S1 + s] = w * (2.f * diff / (float)N);
        if (s == 0) lsum += diff * diff / (float)N;
      }
    }
  }
  if (train) {
    __shared__ float red[4];
    lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
  }
}

// Waves per backward workgroup.  8 (two per SIMD, 225 VGPRs) was measured too: the dx role drops 22 -> 14 us and the chunk
// roles 19 -> 18 us at B=128 (kernel 28.7 -> 27.7 us), but B=4096 loses 4 % (0.411 -> 0.428 ms) — the k-loop is bound by the
// shared VALU / fp32-MFMA pipe (Philox + Box-Muller ~600 cycles + 8 MFMAs = 512 cycles per k-step), not by latency.
#define BWD_NW 4
#ifdef DCCF_BWD_NO_WPE
#define BWD_WPE
#else
#define BWD_WPE __attribute__((amdgpu_waves_per_eu(2)))
#endif
// ================================================================================================ K3: backward
// dz[l][d] = dm[l] * U[u(l)][d] * [h[l][d] > 0] * kscale is never stored: every role rebuilds the operand it needs from
// h, dm and the user row.  grid = (row splits, roles x column halves); a workgroup has ONE role and 4 waves that split
// the batch rows n of its row split (rpn = S1*A consecutive rows l per n):
//   role tq < NC    gW[:, D+128tq ..] += dz^T eps   MFMA, M = d, N = f (4 tiles), K = the rows of n, two rows per k-step; eps
//                   is REGENERATED by one Philox call per k-step directly as the B operand (its 4 normals = the 4 N
//                   tiles).  The feature term dz^T feat costs ONE more k-step per n: A = sum of the n's dz rows (kept in
//                   a register while walking them), B = feat[i0(n)].  Role 0 also emits gU[u] += sum dm h (one atomic
//                   row add per batch row) and gb += sum dz — by-products of building A.
//   role NC         gW[:, 0:D] += dz^T V[cand]      same A operand, B = candidate rows.
//   role NC+1       gV[cand] += dz W_i              MFMA, M = rows, N = d', K = d; W_i in registers; the A noise copies
//                   of a candidate are summed in registers and rows leave as 128-B float-atomic segments.
// Float atomics issue at about one 256-B wave instruction per 50 ns PER CU (MI355X_MICROARCH.md), so what a workgroup
// may dump is small: the 4 waves' 32x32 accumulators are first summed through LDS and each CU then emits 32 KB, shaped
// as two 128-B row segments per wave instruction.
struct BwdArgs {
  const float *W, *U, *V, *feat;
  const int64_t* X;
  const int* cand;
  const float *dmns, *hbuf, *noise;
  const float* dh;           // DH (n_layers > 1): d loss / d h0 [L, DP] from the layers above; dz0 = dh * [h0 > 0] * kscale
  float *gU, *gV, *gW, *gb;
  float* gw_part;            // != NULL: dW leaves as this row split's partial sums (plain stores into copy blockIdx.x), not as
  int64_t gw_stride;         // float atomics into gW — the optimizer launch that follows adds the copies (GwPart)
  uint8_t *touchedU, *touchedV;
  int64_t N;
  int S1, A, F, NC;
  int Dr;                    // the model's row width (<= the template's tile D): stride of U / V / gU / gV rows, W is [Dr][Dr+F]
  float kscale, nscale;
  rng_key nkey;
  StepRef sr;
  // FOLD (S1 <= 16): the pair epilogue runs inside this kernel — dmns holds Expo[u, cand] and stays read-only
  const float *m, *Y;
  float *pred, *loss;
  int rank;
  // slot mode (replicated path): row x of table U / V accumulates in slot_rows[slot_where[off + x]] instead of gU / gV
  const int* slot_where;
  float* slot_rows;
  int64_t slot_offU, slot_offV;
  int slot_cap;              // slots in the buffer: a lookup is clamped into it (a table that does not match the batch —
                             // a caller that rewrote an announced batch in place — must not become a wild store)
  // hosted optimizer pass (dccf_train_step overlap == 2): the grid rows past the roles run the untouched-row pass of
  // `oj` while the roles compute — a backward workgroup is one wave per SIMD at <= 256 VGPRs, so a second workgroup fits
  // beside it on every CU, and the pass needs HBM, not the VALU / MFMA pipe
  OptJob oj;
  int opt_rows_y;            // grid rows that host it (0: none)
  // deterministic mode (dccf_ctx_set_deterministic; run-time-width instances only): no float atomics — the user gradient of
  // batch row n goes to det_u[n], the item gradient of candidate slot (n, s) (A == 2; of row l otherwise) to det_v[slot], the
  // by-product gb of wave w of row split x to det_gb[x * BWD_NW + w]; k_det_sum adds them per destination in slot order
  float *det_u, *det_v, *det_gb;
  int det_ld;                // floats per det_gb entry
};

// FOLD: what k_pair_epilogue would have stored for batch row n, recomputed by the wave that walks n (every role needs it;
// 66 loads and a few shuffles per n instead of a kernel between forward and backward).  16-lane groups: even groups take
// row n, odd groups its BPR partner (n +- N/2), so one xor-16 shuffle gives every lane both predictions.  Returns
// d loss / d (mean_a m[n][s]) for s = lane & 15, valid in lanes 0 .. S1-1; `emit` (one role) writes prediction and loss.
// Split in two so that a wave can fetch the operands of its NEXT batch row while it works on the current one.
struct DmIn {
  float e, m[4], y;          // Expo[u, cand[row][s]], m[(row, s, a)] for a < A <= 4, Y[row] (rank 0)
};
__device__ __forceinline__ void wave_dm_load(const BwdArgs& p, int64_t n, DmIn& in) {
  const int lane = threadIdx.x & 63, g = lane >> 4, s = min(lane & 15, p.S1 - 1);      // clamped: unconditional loads
  const int64_t B = p.N / 2;
  const int64_t other = p.rank == 1 ? (n < B ? n + B : n - B) : n;
  const int64_t row = (g & 1) ? other : n;
  in.e = p.dmns[row * p.S1 + s];
#pragma unroll
  for (int a = 0; a < 4; ++a) in.m[a] = p.m[(row * p.S1 + s) * p.A + min(a, p.A - 1)];
  in.y = 0.f;
  if (p.rank != 1) in.y = p.Y[row];
}
__device__ __forceinline__ float wave_dm_calc(const BwdArgs& p, int64_t n, const DmIn& in, float& lsum, bool emit) {
  const int lane = threadIdx.x & 63, g = lane >> 4, s = lane & 15;
  const int64_t B = p.N / 2;
  const int64_t other = p.rank == 1 ? (n < B ? n + B : n - B) : n;
  const int64_t row = (g & 1) ? other : n;
  // (row_predict<16>, operand for operand)
  const bool valid = s < p.S1;
  const float e = valid ? in.e : -INFINITY;
  const float mx = group_max<16>(e);
  const float ex = valid ? expf(e - mx) : 0.f;
  const float w0 = ex / group_sum<16>(ex);
  float tot = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
    if (a < p.A) tot += group_sum<16>(valid ? w0 * in.m[a] : 0.f);
  const float w = w0 / (float)p.A;
  const float pm = tot / (float)p.A;
  float dm;
  if (p.rank == 1) {
    const float po = __shfl_xor(pm, 16, 64);
    const bool pos = row < B;
    const float pp = pos ? pm : po, pn = pos ? po : pm;
    const float d = pp - pn;
    const float sg = 1.f / (1.f + expf(-d));
    const float gp = -(1.f - sg);                 // d loss / d pos = -sigmoid(neg - pos)
    dm = pos ? w * gp : w * (-gp);
    if (emit && lane == 0) {
      p.pred[n] = pm;
      if (n < B) lsum += -logf(sg);
    }
  } else {
    const float diff = pm - in.y;
    dm = w * (2.f * diff / (float)p.N);
    if (emit && lane == 0) {
      p.pred[n] = pm;
      lsum += diff * diff / (float)p.N;
    }
  }
  return dm;
}

// roles "feature chunk" (CHUNK) and "item": A = dz^T for the rows of n, B = eps (regenerated / injected) or V[cand]
// (the item role covers at most 128 columns of dW_i = 4 N tiles: a 256-wide tile has two item roles, `ig` = which one)
template <int D, int MODE, bool CHUNK, bool FOLD, bool DH, bool GEN>
__device__ __forceinline__ void bwd_col_role(const BwdArgs& p, float* red, int role, int dbase, int ig = 0) {
  constexpr int DP = D <= 32 ? 32 : (D + 63) / 64 * 64;
  constexpr int ND = D <= 32 ? 1 : 2;
  constexpr int NT = (D + 31) / 32;
  constexpr int NB = CHUNK ? 4 : (NT < 4 ? NT : 4);      // N tiles of this role
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int S1 = p.S1, A = p.A, F = p.F, Dr = GEN ? p.Dr : D;
  const int rpn = S1 * A;
  const int KS = (rpn + 1) / 2;
  const int64_t n0 = (int64_t)blockIdx.x * BWD_NW + wave, nstride = (int64_t)gridDim.x * BWD_NW;
  TRACEB(role, 0);
  f32x16 acc[ND][NB];
#pragma unroll
  for (int mt = 0; mt < ND; ++mt)
#pragma unroll
    for (int o = 0; o < NB; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][o][r] = 0.f;
  float gb_acc[ND];
  bool dok[ND];
#pragma unroll
  for (int mt = 0; mt < ND; ++mt) {
    gb_acc[mt] = 0.f;
    dok[mt] = dbase + mt * 32 + c31 < Dr;
  }
  float lsum = 0.f;
  // One wave per SIMD walks its batch rows alone, so nothing hides a load round trip but the code itself: the ids of the
  // NEXT batch row are fetched while this one is worked on, the row's own operands (user row, feature row, first batch of
  // k-steps, the folded epilogue) go out together, and every batch of four k-steps is fetched while the previous one is in
  // the MFMA pipe (5 exposed round trips per batch row -> 1; in-kernel timestamps: main loop 17.5 -> see DESIGN.md).
  // All loads are unconditional from clamped addresses (a load inside `cond ? load : 0` makes hipcc branch around it and
  // wait vmcnt(0)); rows past the batch row read its last row and are zeroed through dm.
#define BWD_LOAD_BATCH(J0, HV, DMV, BQ, DHV)                                                                         \
  _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                             \
    const int r_ = 2 * ((J0) + jj) + h;                                                                          \
    const int rc_ = min(r_, rpn - 1);                                                                            \
    const int64_t lr_ = n * rpn + rc_;                                                                           \
    const int64_t ns_ = n * S1 + rc_ / A;                                                                        \
    if (!FOLD && !DH) DMV[jj] = p.dmns[ns_];                                                                     \
    _Pragma("unroll") for (int mt = 0; mt < ND; ++mt) HV[jj][mt] = p.hbuf[lr_ * DP + dbase + mt * 32 + c31];     \
    if (DH) { _Pragma("unroll") for (int mt = 0; mt < ND; ++mt) DHV[jj][mt] = p.dh[lr_ * DP + dbase + mt * 32 + c31]; } \
    if (!CHUNK) {                                                                                                \
      const float* vrow_ = p.V + (int64_t)p.cand[ns_] * Dr;                                                      \
      _Pragma("unroll") for (int nt = 0; nt < NB; ++nt) BQ[jj][nt] = vrow_[min((ig * 4 + nt) * 32 + c31, Dr - 1)]; \
    } else if (MODE == 1) {                                                                                      \
      _Pragma("unroll") for (int o = 0; o < NB; ++o)                                                             \
        BQ[jj][o] = p.noise[lr_ * F + min(role * 128 + 32 * o + c31, F - 1)];                                    \
    }                                                                                                            \
  }
  int64_t u_nx = 0, i_nx = 0;
  DmIn din;
  if (n0 < p.N) {
    u_nx = p.X[2 * n0];
    i_nx = p.X[2 * n0 + 1];
    if (FOLD) wave_dm_load(p, n0, din);
  }
  for (int64_t n = n0; n < p.N; n += nstride) {
    const int64_t u = u_nx;
    const float* frow = p.feat + i_nx * F;
    DmIn dnx;
    {                                  // the next batch row's ids and epilogue operands travel while this one is computed
      const int64_t nn = min(n + nstride, p.N - 1);
      u_nx = p.X[2 * nn];
      i_nx = p.X[2 * nn + 1];
      if (FOLD) wave_dm_load(p, nn, dnx);
    }
    float dmn = 0.f;
    int64_t urow_g = u;            // where gU[u] accumulates: the row itself, or its slot of the all-gather buffer
    if (CHUNK && role == 0 && p.slot_where) urow_g = min((unsigned)p.slot_where[p.slot_offU + u], (unsigned)(p.slot_cap - 1));
    float uv[ND], asum[ND], due[ND], fb[NB];
#pragma unroll
    for (int mt = 0; mt < ND; ++mt) {
      uv[mt] = p.U[u * Dr + min(dbase + mt * 32 + c31, Dr - 1)];
      asum[mt] = 0.f;
      due[mt] = 0.f;
    }
    if (CHUNK) {
#pragma unroll
      for (int o = 0; o < NB; ++o) fb[o] = frow[min(role * 128 + 32 * o + c31, F - 1)];   // columns >= F are never stored
    }
    float hv[4][ND], dmv[4], bq[4][NB], dhv[4][ND];
    BWD_LOAD_BATCH(0, hv, dmv, bq, dhv)
    if (FOLD) {
      dmn = wave_dm_calc(p, n, din, lsum, CHUNK && role == 0 && dbase == 0);
      din = dnx;
    }
#pragma unroll
    for (int mt = 0; mt < ND; ++mt) {
      KEEP(uv[mt]);
      uv[mt] = dok[mt] ? uv[mt] : 0.f;
    }
    if (n == n0) TRACEB(role, 3);
    for (int j0 = 0; j0 < KS; j0 += 4) {
      float hv2[4][ND], dmv2[4], bq2[4][NB], dhv2[4][ND];
      BWD_LOAD_BATCH(j0 + 4, hv2, dmv2, bq2, dhv2)    // the next batch (past the end: clamped rows, never used)
      if (FOLD) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dmv[jj] = __shfl(dmn, min(2 * (j0 + jj) + h, rpn - 1) / A, 64);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {              // pin the current batch (its loads only), then mask
        if (!DH) KEEP(dmv[jj]);
#pragma unroll
        for (int mt = 0; mt < ND; ++mt) KEEP(hv[jj][mt]);
        if (DH) {
#pragma unroll
          for (int mt = 0; mt < ND; ++mt) KEEP(dhv[jj][mt]);
        }
        if (!CHUNK || MODE == 1) {
#pragma unroll
          for (int o = 0; o < NB; ++o) KEEP(bq[jj][o]);
        }
        dmv[jj] = 2 * (j0 + jj) + h < rpn ? dmv[jj] : 0.f;
#pragma unroll
        for (int mt = 0; mt < ND; ++mt) hv[jj][mt] = (dok[mt] && (!DH || 2 * (j0 + jj) + h < rpn)) ? hv[jj][mt] : 0.f;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if (CHUNK && MODE == 0)
          noise4((uint32_t)(n * rpn + min(2 * (j0 + jj) + h, rpn - 1)), (uint32_t)(role * 32 + c31), p.nkey, p.nscale, bq[jj]);
        float a[ND];
#pragma unroll
        for (int mt = 0; mt < ND; ++mt) {
          if (DH) a[mt] = hv[jj][mt] > 0.f ? dhv[jj][mt] * p.kscale : 0.f;    // (rows past the batch row: h masked to 0 above)
          else a[mt] = hv[jj][mt] > 0.f ? (dmv[jj] * uv[mt]) * p.kscale : 0.f;     // 0 for the k-steps past KS (dm = 0)
          asum[mt] += a[mt];
          if (!DH) due[mt] = fmaf(dmv[jj], hv[jj][mt], due[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < ND; ++mt)
#pragma unroll
          for (int o = 0; o < NB; ++o) acc[mt][o] = MFMA32(a[mt], bq[jj][o], acc[mt][o]);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {              // next batch becomes current
        dmv[jj] = dmv2[jj];
#pragma unroll
        for (int mt = 0; mt < ND; ++mt) hv[jj][mt] = hv2[jj][mt];
        if (DH) {
#pragma unroll
          for (int mt = 0; mt < ND; ++mt) dhv[jj][mt] = dhv2[jj][mt];
        }
        if (!CHUNK || MODE == 1) {
#pragma unroll
          for (int o = 0; o < NB; ++o) bq[jj][o] = bq2[jj][o];
        }
      }
    }
    if (n == n0) TRACEB(role, 4);
    float tot[ND];
#pragma unroll
    for (int mt = 0; mt < ND; ++mt) tot[mt] = asum[mt] + __shfl_xor(asum[mt], 32, 64);
    if (CHUNK) {       // feature term: one k-step, A = sum of the n's dz rows (k = 0 half only), B = feat[i0(n)]
#pragma unroll
      for (int mt = 0; mt < ND; ++mt) {
        const float a2 = h == 0 ? tot[mt] : 0.f;
#pragma unroll
        for (int o = 0; o < NB; ++o) acc[mt][o] = MFMA32(a2, fb[o], acc[mt][o]);
      }
    }
    if (CHUNK && role == 0) {
#pragma unroll
      for (int mt = 0; mt < ND; ++mt) {
        const float dt = due[mt] + __shfl_xor(due[mt], 32, 64);
        if (h == 0 && dok[mt]) {
          if (!DH) {           // (n_layers > 1: the user gradient comes from the LAST layer's output, k_gu_last)
            if (GEN && p.det_u) {
              p.det_u[n * Dr + dbase + mt * 32 + c31] = dt;
            } else {
              float* gu = (p.slot_where ? p.slot_rows : p.gU) + urow_g * Dr;
              atomicAdd(&gu[dbase + mt * 32 + c31], dt);
            }
          }
          gb_acc[mt] += tot[mt];
        }
      }
      if (!DH && p.touchedU && lane == 0 && !(GEN && p.det_u)) p.touchedU[u] = 1;
    }
  }
  KEEP(acc[0][0][0]);
  TRACEB(role, 1);
  // sum the 4 waves' accumulators through LDS in rounds of 32 registers (32 KB of LDS, so several workgroups fit on
  // a CU), each wave emits a quarter of every round
  constexpr int NQ = ND * NB * 16;        // accumulator registers per lane
  constexpr int RQ = 128 / BWD_NW;        // registers per round: BWD_NW * RQ * 256 B = 32 KB of LDS
#pragma unroll
  for (int q0 = 0; q0 < NQ; q0 += RQ) {
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < RQ; ++qq) {
      const int q = q0 + qq;
      red[(wave * RQ + qq) * 64 + lane] = acc[q / (NB * 16)][(q / 16) % NB][q % 16];
    }
    __syncthreads();
    if (q0 == 0) TRACEB(role, 5);
    // every wave emits RQ / BWD_NW registers of the round: ALL their LDS reads first, then the sums, then the stores back to
    // back (as a rolled loop this was one LDS round trip + one atomic per iteration: 1.15 us per round, 4.6 of the 22 us of a
    // chunk role at B = 128)
    constexpr int PER = RQ / BWD_NW;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    if (p.gw_part && ((Dr | F) & 3) == 0) {
      // partial-sum copy, rows 16-byte aligned: a lane takes 4 consecutive columns of one accumulator register (4 lanes' values:
      // one ds_read_b128 per wave copy) and stores them as ONE dwordx4 — 8 store instructions of 1 KB per round instead of 32
      // of 256 B (a CU retires a dword-per-lane store or atomic instruction only every ~40 ns: 4.8 us per chunk role)
      float* __restrict__ dstw = p.gw_part + (int64_t)blockIdx.x * p.gw_stride;
      constexpr int PERW = RQ / 4 / BWD_NW;
#pragma unroll
      for (int k = 0; k < PERW; ++k) {
        const int qq = 4 * (wv + k * BWD_NW) + (lane >> 4), j = lane & 15;
        float4 t = *reinterpret_cast<const float4*>(&red[qq * 64 + 4 * j]);
#pragma unroll
        for (int w = 1; w < BWD_NW; ++w) {
          const float4 u = *reinterpret_cast<const float4*>(&red[(w * RQ + qq) * 64 + 4 * j]);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        const int q = q0 + qq;
        const int mt = q / (NB * 16), o = (q / 16) % NB, r = q % 16;
        const int hh = j >> 3, c = 4 * (j & 7);
        const int d = dbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int col = CHUNK ? Dr + role * 128 + 32 * o + c : (ig * 4 + o) * 32 + c;
        const bool ok = d < Dr && (CHUNK ? (role * 128 + 32 * o + c < F) : (col < Dr));
        if (ok) *reinterpret_cast<float4*>(&dstw[(int64_t)d * (Dr + F) + col]) = t;
      }
      continue;
    }
    float part[PER][BWD_NW];
#pragma unroll
    for (int k = 0; k < PER; ++k)
#pragma unroll
      for (int w = 0; w < BWD_NW; ++w) part[k][w] = red[(w * RQ + wv + k * BWD_NW) * 64 + lane];
    float* __restrict__ dst = p.gw_part ? p.gw_part + (int64_t)blockIdx.x * p.gw_stride : p.gW;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int q = q0 + wv + k * BWD_NW;
      const int mt = q / (NB * 16), o = (q / 16) % NB, r = q % 16;
      float v = part[k][0];
#pragma unroll
      for (int w = 1; w < BWD_NW; ++w) v += part[k][w];
      const int d = dbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int col = CHUNK ? Dr + role * 128 + 32 * o + c31 : (ig * 4 + o) * 32 + c31;
      const bool ok = d < Dr && (CHUNK ? (role * 128 + 32 * o + c31 < F) : (col < Dr));
      if (ok) {
        if (p.gw_part) dst[(int64_t)d * (Dr + F) + col] = v;
        else atomicAdd(&dst[(int64_t)d * (Dr + F) + col], v);
      }
    }
  }
  TRACEB(role, 6);
  if (CHUNK && role == 0 && h == 0) {
#pragma unroll
    for (int mt = 0; mt < ND; ++mt)
      if (dok[mt]) {
        if (GEN && p.det_gb) p.det_gb[((int64_t)blockIdx.x * BWD_NW + wave) * p.det_ld + dbase + mt * 32 + c31] = gb_acc[mt];
        else atomicAdd(&p.gb[dbase + mt * 32 + c31], gb_acc[mt]);
      }
  }
  if (FOLD && CHUNK && role == 0 && dbase == 0 && lane == 0 && lsum != 0.f) atomicAdd(p.loss, lsum);
  TRACEB(role, 2);
}
#undef BWD_LOAD_BATCH

// role "dx": gV[cand] += dz W_i   (MFMA: M = rows of n, N = d', K = d)
// (a wave keeps at most 128 rows of W_i in registers: a 256-wide tile has two dx roles, `kh` = which half of the contraction — their
// partial rows meet in the float atomics)
template <int D, bool FOLD, bool DH, bool GEN>
__device__ __forceinline__ void bwd_dx_role(const BwdArgs& p, int dbase, int kh = 0) {
  constexpr int DP = D <= 32 ? 32 : (D + 63) / 64 * 64;
  constexpr int ND = D <= 32 ? 1 : 2;
  constexpr int KD = (D < 128 ? D : 128) / 2;
  const int k0 = kh * 128;                       // first contraction index of this role
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int S1 = p.S1, A = p.A, F = p.F, Dr = GEN ? p.Dr : D;
  const int rpn = S1 * A;
  const int64_t n0 = (int64_t)blockIdx.x * BWD_NW + wave, nstride = (int64_t)gridDim.x * BWD_NW;
  TRACEB(7, 0);
  float wd[KD][ND];                         // W_i rows 2j+h, this block's column half
#pragma unroll
  for (int j = 0; j < KD; ++j)
#pragma unroll
    for (int nt = 0; nt < ND; ++nt) {
      const int dd = dbase + nt * 32 + c31;
      const float wx = p.W[(int64_t)min(k0 + 2 * j + h, Dr - 1) * (Dr + F) + min(dd, Dr - 1)];
      wd[j][nt] = (dd < Dr && k0 + 2 * j + h < Dr) ? wx : 0.f;
    }
  const int RT = (rpn + 31) / 32;
  for (int64_t n = n0; n < p.N; n += nstride) {
    const float* urow = p.U + p.X[2 * n] * Dr;
    float dmn = 0.f, lnone = 0.f;
    if (FOLD) {
      DmIn din;
      wave_dm_load(p, n, din);
      dmn = wave_dm_calc(p, n, din, lnone, false);
    }
    for (int t = 0; t < RT; ++t) {
      const int rr = t * 32 + c31;
      const bool lv = rr < rpn;
      const int rc = lv ? rr : rpn - 1;             // clamped: loads are unconditional, dm zeroes the row
      const int64_t l = n * rpn + rc;
      float dm0;
      if (FOLD) dm0 = __shfl(dmn, rc / A, 64);
      else dm0 = p.dmns[n * S1 + rc / A];
      KEEP(dm0);
      const float dmv = lv ? dm0 : 0.f;
      const float* hrow = p.hbuf + l * DP;
      const float* dhrow = DH ? p.dh + l * DP : hrow;
      // where the 8 candidate rows this lane will emit go (A == 2): candidate id, or in slot mode its slot of the all-gather
      // buffer — fetched now, so the dependent lookups are over when the accumulators are ready
      int sl8[8];
      if (A == 2) {
#pragma unroll
        for (int r2 = 0; r2 < 8; ++r2) {
          const int r = 2 * r2;
          const int ro = min(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, rpn - 1);
          const int ci = p.cand[n * S1 + (ro >> 1)];
          sl8[r2] = p.slot_where ? (int)min((unsigned)p.slot_where[p.slot_offV + ci], (unsigned)(p.slot_cap - 1)) : ci;
          if (GEN && p.det_v) sl8[r2] = (int)(n * S1 + (ro >> 1));
        }
      }
      f32x16 acc[ND];
#pragma unroll
      for (int nt = 0; nt < ND; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
      constexpr int JB = KD < 8 ? KD : 8;           // k-steps per load batch
#pragma unroll
      for (int j0 = 0; j0 < KD; j0 += JB) {
        float hvv[JB], uxx[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {           // issue the batch's loads ...
          hvv[jj] = hrow[k0 + 2 * (j0 + jj) + h];                  // (columns >= Dr of h are never written: masked below)
          uxx[jj] = DH ? dhrow[k0 + 2 * (j0 + jj) + h] : urow[min(k0 + 2 * (j0 + jj) + h, Dr - 1)];
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {           // ... then pin them (one wait for the batch)
          KEEP(hvv[jj]);
          KEEP(uxx[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const bool on = hvv[jj] > 0.f && (D == Dr || k0 + 2 * (j0 + jj) + h < Dr);
          const float a = DH ? ((on && lv) ? uxx[jj] * p.kscale : 0.f) : (on ? (dmv * uxx[jj]) * p.kscale : 0.f);
#pragma unroll
          for (int nt = 0; nt < ND; ++nt) acc[nt] = MFMA32(a, wd[j0 + jj][nt], acc[nt]);
        }
      }
      // rows -> gV[cand]; with A == 2 rows (2q, 2q+1) are one candidate and sit in adjacent registers
#pragma unroll
      for (int nt = 0; nt < ND; ++nt) {
        const int dd = dbase + nt * 32 + c31;
        if (A == 2) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int ro = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ro < rpn && dd < Dr) {
              const int64_t ci = sl8[r >> 1];
              if (GEN && p.det_v) {
                p.det_v[ci * Dr + dd] = acc[nt][r] + acc[nt][r + 1];
              } else {
                float* gv = (p.slot_where ? p.slot_rows : p.gV) + ci * Dr;
                atomicAdd(&gv[dd], acc[nt][r] + acc[nt][r + 1]);
                if (p.touchedV && c31 == 0 && nt == 0) p.touchedV[ci] = 1;
              }
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ro < rpn && dd < Dr) {
              const int64_t ci = p.cand[n * S1 + ro / A];
              if (GEN && p.det_v) {
                p.det_v[(n * rpn + ro) * Dr + dd] = acc[nt][r];
              } else {
                float* gv = p.slot_where ? p.slot_rows + (int64_t)min((unsigned)p.slot_where[p.slot_offV + ci], (unsigned)(p.slot_cap - 1)) * Dr : p.gV + ci * Dr;
                atomicAdd(&gv[dd], acc[nt][r]);
                if (p.touchedV && c31 == 0 && nt == 0) p.touchedV[ci] = 1;
              }
            }
          }
        }
      }
    }
  }
}

template <int D_, int MODE, bool FOLD, bool DH, bool GEN>
__global__ __launch_bounds__(64 * BWD_NW) BWD_WPE void k_bwd(BwdArgs p) {
  extern __shared__ float red[];          // [BWD_NW waves][128 / BWD_NW regs][64 lanes]
  {
    const int64_t k = step_k(p.sr);
    p.X = step_X(p.sr, p.X, k);
    p.nkey = key_plus(p.nkey, k);
  }
  constexpr int DW = (D_ <= 32 ? 1 : 2) * 32;
  constexpr int GY = D_ <= 64 ? 1 : D_ / 64;
  constexpr int NI = D_ <= 128 ? 1 : D_ / 128;         // item roles (128 columns of dW_i each), dx roles (128 rows of W_i each)
  const int role = blockIdx.y / GY;
  const int dbase = (blockIdx.y % GY) * DW;
  if ((int)blockIdx.y >= (p.NC + 2 * NI) * GY) {       // hosted untouched-row optimizer pass
    opt_resolve(p.oj.a);
    const int64_t bid = (int64_t)(blockIdx.y - (p.NC + 2 * NI) * GY) * gridDim.x + blockIdx.x;
    const int64_t nblk = (int64_t)p.opt_rows_y * gridDim.x;
    if (p.oj.kind == DCCF_OPT_GD) opt_untouched_pass<DCCF_OPT_GD, 4>(p.oj, bid, nblk, 64 * BWD_NW);
    else if (p.oj.kind == DCCF_OPT_ADAGRAD) opt_untouched_pass<DCCF_OPT_ADAGRAD, 4>(p.oj, bid, nblk, 64 * BWD_NW);
    else opt_untouched_pass<DCCF_OPT_ADAM, 4>(p.oj, bid, nblk, 64 * BWD_NW);
    return;
  }
  if (GEN && dbase >= p.Dr) return;                    // a column group of the tile that holds no real column (192 in the 256 tile)
  if (role < p.NC) bwd_col_role<D_, MODE, true, FOLD, DH, GEN>(p, red, role, dbase);
  else if (role < p.NC + NI) bwd_col_role<D_, MODE, false, FOLD, DH, GEN>(p, red, role, dbase, role - p.NC);
  else bwd_dx_role<D_, FOLD, DH, GEN>(p, dbase, role - p.NC - NI);
  if (role >= p.NC + NI) TRACEB(7, 2);
}

// ================================================================================================ K4: extra mlp layers
// --n_layers > 1 (src/models/DMF.py:14): after mlp.0 the reference runs n_layers - 1 more Linear(D -> D) + relu + dropout
// (src/models/DCCF.py:61-62,91-94) before the dot with the user row.  Each extra layer is one forward launch over the rows l
// (h_k = drop(relu(h_{k-1} W_k^T + b_k)), the last one also writes m[l] = <U[u], h_k[l]>) and one backward launch
// (dz_k = dh_k * [h_k > 0] * kscale;  gW_k += dz_k^T h_{k-1};  gb_k += sum dz_k;  dh_{k-1} = dz_k W_k).  The first layer's
// backward (k_bwd, DH) then starts from dh_0 instead of dm * U[u].  fp32 MFMA like the rest; W_k (<= 64 KB) sits in LDS.
// These launches exist only for n_layers > 1: the default path is untouched.
struct MlpArgs {
  const float* hin;          // [L, DP] h_{k-1}
  float* hout;               // forward: [L, DP] h_k (written);  backward: read as h_k
  const float* W;            // mlp.k.weight [Dr, Dr]
  const float* b;            // mlp.k.bias [Dr]
  const float* U;
  const int64_t* X;
  const uint8_t* keep;       // injected keep mask of this layer [L, Dr] (NULL: keep all / fused draws)
  float* m;                  // forward, last layer: m[l] = <U[u(l)], h_k[l]>
  const float* dmns;         // backward, last layer: d loss / d (mean_a m) per (n, s)
  const float* dhin;         // backward, other layers: dh_k [L, DP]
  float* dhout;              // backward: dh_{k-1} [L, DP]
  float* gW;                 // backward: [Dr, Dr] (+=)
  float* gb;                 // backward: [Dr] (+=)
  int64_t L;
  int rpn, A, Dr, DP, layer, last, fused;
  int det;                   // backward, deterministic mode: ONE workgroup, its waves add gW_k / gb_k one after the other
  uint32_t drop_thr;
  float kscale;
  rng_key dkey;
  StepRef sr;
};

template <int D_>
__global__ __launch_bounds__(256) void k_mlp_fwd(MlpArgs p) {
  extern __shared__ float wl[];            // [DPF][DPF]: wl[k][n] = W[n][k] (zero beyond Dr)
  constexpr int DPF = D_ <= 32 ? 32 : (D_ <= 64 ? 64 : 128);
  constexpr int NT = DPF / 32, KH = DPF / 2;
  {
    const int64_t k = step_k(p.sr);
    p.X = step_X(p.sr, p.X, k);
    p.dkey = key_plus(p.dkey, k);
  }
  const int Dr = p.Dr, DP = p.DP;
  for (int idx = threadIdx.x; idx < DPF * DPF; idx += blockDim.x) {
    const int k = idx / DPF, n = idx % DPF;
    wl[idx] = (k < Dr && n < Dr) ? p.W[(int64_t)n * Dr + k] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int64_t ntiles = (p.L + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t l = tile * 32 + c31;
    const float* hrow = p.hin + (l < p.L ? l : p.L - 1) * DP;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    for (int j0 = 0; j0 < KH; j0 += 8) {
      float a[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) a[jj] = hrow[h * KH + j0 + jj];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int k = h * KH + j0 + jj;
        const float av = (k < Dr && l < p.L) ? a[jj] : 0.f;      // columns >= Dr of the first layer's h are never written
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = MFMA32(av, wl[k * DPF + nt * 32 + c31], acc[nt]);
      }
    }
    // epilogue: + b, relu, dropout; h_k stored over the whole padded width (zeros beyond Dr)
    float part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = nt * 32 + c31;
      const bool cv = col < Dr;
      const float bd = p.b[cv ? col : 0];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int64_t l4 = tile * 32 + 8 * rq + 4 * h;          // 4 consecutive rows, l4 % 4 == 0: one Philox call
        u32x4 r4{0, 0, 0, 0};
        if (p.fused && p.drop_thr)
          r4 = philox4x32_10((uint32_t)(l4 >> 2), (uint32_t)(cv ? col : 0) | ((uint32_t)p.layer << 16), p.dkey.s0, p.dkey.s1,
                             p.dkey.k0, p.dkey.k1);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int64_t lr = l4 + w;
          const int64_t lrc = lr < p.L ? lr : p.L - 1;
          bool kept = true;
          if (!p.fused) {
            if (p.keep) kept = p.keep[lrc * Dr + (cv ? col : 0)] != 0;
          } else if (p.drop_thr) {
            kept = pick4(r4, w) >= p.drop_thr;
          }
          const float z = acc[nt][rq * 4 + w] + bd;
          const float hv = (cv && z > 0.f && kept) ? z * p.kscale : 0.f;
          if (lr < p.L) p.hout[lr * DP + col] = hv;
          if (p.last) {
            const int64_t u = p.X[2 * (int64_t)((uint32_t)lrc / (uint32_t)p.rpn)];
            part[rq * 4 + w] += p.U[u * Dr + (cv ? col : 0)] * hv;
          }
        }
      }
    }
    if (p.last) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = part[r];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        const int64_t lr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (c31 == 0 && lr < p.L) p.m[lr] = v;
      }
    }
  }
}

// dz of layer k at (row l, column d) — shared by the two operand layouts of k_mlp_bwd
__device__ __forceinline__ float mlp_dz(const MlpArgs& p, int64_t l, int d) {
  if (l >= p.L || d >= p.Dr) return 0.f;
  const float hk = p.hout[l * p.DP + d];
  float dh;
  if (p.last) {
    const int64_t u = p.X[2 * (int64_t)((uint32_t)l / (uint32_t)p.rpn)];
    dh = p.dmns[(uint32_t)l / (uint32_t)p.A] * p.U[u * p.Dr + d];
  } else {
    dh = p.dhin[l * p.DP + d];
  }
  return hk > 0.f ? dh * p.kscale : 0.f;
}

template <int D_>
__global__ __launch_bounds__(256) void k_mlp_bwd(MlpArgs p) {
  extern __shared__ float wl[];            // [DPF][DPF]: wl[d][d'] = W[d][d'] (zero beyond Dr)
  constexpr int DPF = D_ <= 32 ? 32 : (D_ <= 64 ? 64 : 128);
  constexpr int NT = DPF / 32, KH = DPF / 2, MT = NT, TS = 4 / MT;
  {
    const int64_t k = step_k(p.sr);
    p.X = step_X(p.sr, p.X, k);
  }
  const int Dr = p.Dr, DP = p.DP;
  for (int idx = threadIdx.x; idx < DPF * DPF; idx += blockDim.x) {
    const int d = idx / DPF, n = idx % DPF;
    wl[idx] = (d < Dr && n < Dr) ? p.W[(int64_t)d * Dr + n] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int cm = wave % MT, ts = wave / MT;          // this wave's 32-column block and tile sub-stream
  const int64_t ntiles = (p.L + 31) / 32;
  f32x16 accW[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[nt][r] = 0.f;
  float gbacc = 0.f;
  for (int64_t tile = (int64_t)blockIdx.x * TS + ts; tile < ntiles; tile += (int64_t)gridDim.x * TS) {
    // (1) dh_{k-1}[rows of the tile][cm*32 ..] = dz W_k : M = rows, N = this wave's columns, K = d (k-step j: d = h*KH + j)
    {
      const int64_t l = tile * 32 + c31;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int j = 0; j < KH; ++j) {
        const int d = h * KH + j;
        acc = MFMA32(mlp_dz(p, l, d), wl[d * DPF + cm * 32 + c31], acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t lr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (lr < p.L) p.dhout[lr * DP + cm * 32 + c31] = acc[r];
      }
    }
    // (2) gW_k[cm*32 + m][:] += dz^T h_{k-1} : M = this wave's d block, N = d', K = rows (k-step j: row = 2j + h)
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int64_t lr = tile * 32 + 2 * j + h;
      const float a = mlp_dz(p, lr, cm * 32 + c31);
      gbacc += a;
      const float* hp = p.hin + (lr < p.L ? lr : p.L - 1) * DP;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = hp[nt * 32 + c31];
        accW[nt] = MFMA32(a, (nt * 32 + c31 < Dr && lr < p.L) ? bv : 0.f, accW[nt]);
      }
    }
  }
  gbacc += __shfl_xor(gbacc, 32, 64);
  if (p.det) {               // deterministic mode (ONE workgroup): the waves add one after the other, plain read-modify-write
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int mrow = cm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = nt * 32 + c31;
            if (mrow < Dr && n < Dr) p.gW[(int64_t)mrow * Dr + n] = __fadd_rn(p.gW[(int64_t)mrow * Dr + n], accW[nt][r]);
          }
        if (h == 0 && cm * 32 + c31 < Dr) p.gb[cm * 32 + c31] = __fadd_rn(p.gb[cm * 32 + c31], gbacc);
      }
      __threadfence_block();
      __syncthreads();
    }
    return;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mrow = cm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = nt * 32 + c31;
      if (mrow < Dr && n < Dr && accW[nt][r] != 0.f) atomicAdd(&p.gW[(int64_t)mrow * Dr + n], accW[nt][r]);
    }
  if (h == 0 && cm * 32 + c31 < Dr && gbacc != 0.f) atomicAdd(&p.gb[cm * 32 + c31], gbacc);
}

// gU[u(n)] += sum_l dm[l] h_last[l]  (n_layers > 1: the user row meets the LAST layer's output, src/models/DCCF.py:96);
// one wave per batch row, one atomic row add (into gU or, in the replicated path's slot mode, the row's buffer slot)
__global__ __launch_bounds__(256) void k_gu_last(const float* __restrict__ hlast, const float* __restrict__ dmns, const int64_t* X,
                                                 float* gU, uint8_t* touchedU, int64_t N, int S1, int A, int Dr, int DP,
                                                 const int* slot_where, float* slot_rows, int64_t slot_offU, int slot_cap,
                                                 StepRef sr, float* det_u) {
  X = step_X(sr, X, step_k(sr));
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int rpn = S1 * A;
  for (int64_t n = w0; n < N; n += nw) {
    float a0 = 0.f, a1 = 0.f;
    for (int r = 0; r < rpn; ++r) {
      const float dm = dmns[n * S1 + r / A];
      const float* hr = hlast + (n * rpn + r) * DP;
      if (lane < Dr) a0 = fmaf(dm, hr[lane], a0);
      if (lane + 64 < Dr) a1 = fmaf(dm, hr[lane + 64], a1);
    }
    const int64_t u = X[2 * n];
    if (det_u) {               // deterministic mode: the row of batch row n, summed per user by k_det_sum
      if (lane < Dr) det_u[n * Dr + lane] = a0;
      if (lane + 64 < Dr) det_u[n * Dr + lane + 64] = a1;
      continue;
    }
    float* g = slot_where ? slot_rows + (int64_t)min((unsigned)slot_where[slot_offU + u], (unsigned)(slot_cap - 1)) * Dr : gU + u * Dr;
    if (lane < Dr) atomicAdd(&g[lane], a0);
    if (lane + 64 < Dr) atomicAdd(&g[lane + 64], a1);
    if (touchedU && lane == 0) touchedU[u] = 1;
  }
}

// ================================================================================================ deterministic scatter
// dccf_ctx_set_deterministic (tests; SURVEY.md section 7 "offer a sorted-segment deterministic mode"): float atomics add a row's
// contributions in arrival order, so two runs of the same step differ in the last bits of a gradient row — which Adam turns into
// fractions of lr.  In this mode the backward stores one row per SLOT (user slot n; item slot (n, s), or row l when A != 2) and
// two small launches add the slots of a destination row in ascending slot order: k_det_owner finds the first slot of every
// destination (atomicMin: an integer minimum does not depend on arrival order) and sets the "touched" bytes, k_det_sum lets the
// wave of that first slot walk the later slots.  Its last workgroups add the backward's per-wave gb partial sums and per-split dW
// partial sums in index order.  O(slots^2 / 64) id compares: a test mode, not a fast one.
struct DetArgs {
  const int64_t* X;
  const int* cand;
  int64_t N, NV, user_num;   // user slots, item slots; destination id = u, or user_num + item
  int S1, A, Dr;
  const float *det_u, *det_v;
  float *gU, *gV;
  uint8_t *touchedU, *touchedV;
  int* owner;                // [user_num + item_num], INT_MAX between calls
  // by-products
  const float* det_gb; int n_gb, det_ld; float* gb;        // gb[c] += sum over the n_gb entries, c < Dr
  const float* gw_part; int nsplit; int64_t gw_stride; float* gW;      // gW[i] += sum over the splits, i < gw_stride
  StepRef sr;
};
__device__ __forceinline__ int64_t det_dest(const DetArgs& p, int64_t j) {
  if (j < p.N) return p.X[2 * j];
  const int64_t jv = j - p.N;
  return p.user_num + p.cand[p.A == 2 ? jv : jv / p.A];
}
__global__ __launch_bounds__(256) void k_det_owner(DetArgs p) {
  p.X = step_X(p.sr, p.X, step_k(p.sr));
  const int64_t slots = p.N + p.NV;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < slots; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t d = det_dest(p, j);
    atomicMin(&p.owner[d], (int)j);
    if (j < p.N) { if (p.touchedU) p.touchedU[d] = 1; }
    else if (p.touchedV) p.touchedV[d - p.user_num] = 1;
  }
}
__global__ __launch_bounds__(256) void k_det_sum(DetArgs p, int row_blocks) {
  p.X = step_X(p.sr, p.X, step_k(p.sr));
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x >= row_blocks) {          // ---- by-products, one thread per element, sources in index order
    const int64_t tid = (int64_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x, nt = (int64_t)(gridDim.x - row_blocks) * blockDim.x;
    if (p.det_gb)
      for (int64_t c = tid; c < p.Dr; c += nt) {
        float v = p.gb[c];
        for (int e = 0; e < p.n_gb; ++e) v = __fadd_rn(v, p.det_gb[(int64_t)e * p.det_ld + c]);
        p.gb[c] = v;
      }
    if (p.gw_part)
      for (int64_t i = tid; i < p.gw_stride; i += nt) {
        float v = p.gW[i];
        for (int r = 0; r < p.nsplit; ++r) v = __fadd_rn(v, p.gw_part[(int64_t)r * p.gw_stride + i]);
        p.gW[i] = v;
      }
    return;
  }
  const int64_t slots = p.N + p.NV;
  const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)row_blocks * blockDim.x) >> 6;
  for (int64_t j = w0; j < slots; j += nw) {
    const int64_t d = det_dest(p, j);
    if (p.owner[d] != (int)j) continue;         // (wave-uniform)
    const bool user = j < p.N;
    const int64_t end = user ? p.N : slots;
    float* g = user ? p.gU + d * p.Dr : p.gV + (d - p.user_num) * p.Dr;
    const float* src = user ? p.det_u : p.det_v;
    const int64_t sbase = user ? 0 : p.N;                           // row of slot j' = src + (j' - sbase) * Dr
    float a0 = lane < p.Dr ? g[lane] : 0.f, a1 = lane + 64 < p.Dr ? g[lane + 64] : 0.f;
    for (int64_t j0 = j; j0 < end; j0 += 64) {
      const int64_t jj = j0 + lane;
      uint64_t bits = __ballot(jj < end && det_dest(p, jj < end ? jj : j) == d);
      while (bits) {
        const int b = __ffsll((unsigned long long)bits) - 1;
        bits &= bits - 1;
        const float* r = src + (j0 + b - sbase) * p.Dr;
        if (lane < p.Dr) a0 = __fadd_rn(a0, r[lane]);
        if (lane + 64 < p.Dr) a1 = __fadd_rn(a1, r[lane + 64]);
      }
    }
    if (lane < p.Dr) g[lane] = a0;
    if (lane + 64 < p.Dr) g[lane + 64] = a1;
    if (lane == 0) p.owner[d] = 0x7fffffff;     // (non-owners compare against their own slot index: either value differs from it)
  }
}

// ================================================================================================ host side
static int check_model(const dccf_model_t* M) {
  ARG_CHECK(M != nullptr, "model is NULL");
  ARG_CHECK(M->D >= 1 && M->D <= 256, "D must be in [1, 256]");
  ARG_CHECK(M->D <= 128 || M->n_extra == 0, "embedding sizes above 128 are built for --n_layers 1 (the extra layers keep W_k in LDS)");
  ARG_CHECK(M->F >= 1 && M->F <= 65536, "F must be in [1, 65536]");
  ARG_CHECK(M->S >= 0 && M->S <= 63 && M->A >= 1 && M->A <= 64, "S in [0,63], A in [1,64]");
  ARG_CHECK(M->user_num > 0 && M->item_num > 0 && M->item_num < 2147483647LL, "bad user_num / item_num");
  ARG_CHECK(M->U && M->V && M->W && M->b && M->feat, "NULL parameter / feature pointer");
  ARG_CHECK(M->expo || M->expo_gathered || (M->ipsP && M->ipsQ && M->ipsBu && M->ipsBi && M->ipsProp && M->ipsD > 0),
            "need expo, expo_gathered or the IPS factors");
  ARG_CHECK(M->n_extra >= 0 && M->n_extra <= DCCF_MAX_EXTRA, "n_extra (--n_layers - 1) must be in [0, 7]");
  for (int k = 0; k < M->n_extra; ++k) ARG_CHECK(M->Wl[k] && M->bl[k], "NULL weight / bias of an extra mlp layer");
  return 0;
}

// the column tile of a model of width D (any D in [1, 128]: src/models/RecModel.py:17-27)
#define BY_D(D, CALL)                   \
  if ((D) <= 16) { CALL(16); }          \
  else if ((D) <= 32) { CALL(32); }     \
  else if ((D) <= 64) { CALL(64); }     \
  else { CALL(128); }
// forward / backward also have the 256-wide tile, as run-time-width instances only (embedding sizes 129 .. 256)
#define BY_D_WIDE(D, CALL, CALL_WIDE)   \
  if ((D) <= 16) { CALL(16); }          \
  else if ((D) <= 32) { CALL(32); }     \
  else if ((D) <= 64) { CALL(64); }     \
  else if ((D) <= 128) { CALL(128); }   \
  else { CALL_WIDE(256); }

// The optimizer half of dccf_train_step, threaded through run_dccf: `overlap` forks the untouched-row pass onto the
// context's side stream right after k_prep has marked the rows of this batch.
// ---------------------------------------------------------------------------------------------- prepared next step
// Did an earlier call prepare exactly this step (same batch pointer, size, Philox step, seed, tables, device-drawn candidates)?
bool dccf_prep_matches(const dccf_ctx* ctx, const dccf_model_t* M, const dccf_rand_t* rnd, const void* X, int64_t N) {
  return ctx->prep_valid && rnd->mode == 1 && rnd->k_dev == nullptr && ctx->prep_X == X && ctx->prep_N == N &&
         ctx->prep_step == rnd->step && ctx->prep_seed == rnd->seed && ctx->prep_U == (const void*)M->U &&
         ctx->prep_W == (const void*)M->W;
}

int dccf_prep_next_fill(dccf_ctx* ctx, const dccf_model_t* M, int64_t N, const int64_t* X_next, uint64_t seed, uint64_t step_next,
                        PrepNext* pn) {
  const Lay y = make_layout(N, M->D, M->F, M->S, M->A, M->n_extra);
  if (int e = dccf_ws_ensure(ctx, y.total)) return e;
  memset(pn, 0, sizeof(*pn));
  char* ws = ctx->ws;
  pn->M = *M;
  pn->WT = (float*)(ws + y.WT);
  pn->cand = (int*)(ws + y.cand);
  pn->eg = (float*)(ws + y.dmns);
  pn->m = (float*)(ws + y.m);
  pn->X = X_next;
  pn->N = N;
  pn->Lm = y.GY > 1 ? y.L : 0;
  pn->S = M->S;
  pn->DP = y.DP;
  pn->key = make_key(seed, STREAM_CAND, step_next);
  return 0;
}

void dccf_prep_next_commit(dccf_ctx* ctx, const dccf_model_t* M, int64_t N, const void* X_next, uint64_t seed, uint64_t step_next) {
  ctx->prep_valid = 1;
  ctx->prep_X = X_next;
  ctx->prep_N = N;
  ctx->prep_step = step_next;
  ctx->prep_seed = seed;
  ctx->prep_U = M->U;
  ctx->prep_W = M->W;
}

struct StepPlan {
  const dccf_opt_t* opt;
  bool overlap;
  bool hosted;               // overlap == 2: the untouched-row pass rides in the backward launch (no side stream)
  int hostv_seg;             // >= 0: segment of the item table; its untouched rows are updated by extra workgroups of the
                             // backward launch (marks from the previous step's optimizer launch or from k_prep)
  MarkPlan mark;
  int64_t max_rows;
  const int64_t* X_next;     // same N, Philox step `step_next`: prepared inside the optimizer launch (NULL = not known)
  uint64_t step_next;
  int lazy_segU, lazy_segV;  // >= 0: windowed lazy regularisation (dccf_opt_t.lazy_K > 0); the segments of U and V
};

static int run_dccf(dccf_ctx* ctx, const dccf_model_t* M, const dccf_rand_t* rnd, const int64_t* X, const float* Y,
                    int64_t N, int rank, float dropout, const dccf_grads_t* G, float* pred, float* loss, bool train,
                    hipStream_t st, const StepPlan* plan = nullptr) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (int e = check_model(M)) return e;
  ARG_CHECK(N >= 0 && N * (int64_t)(M->S + 1) * M->A < 4294967296LL, "N*(S+1)*A must be < 2^32");
  ARG_CHECK(rnd != nullptr && (N == 0 || (X != nullptr && pred != nullptr)), "NULL rnd / X / prediction");
  ARG_CHECK(dropout >= 0.f && dropout < 1.f, "dropout must be in [0,1)");
  const bool fused = rnd->mode != 0;            // noise + dropout drawn on device
  const bool fused_cand = rnd->mode == 1;       // candidates drawn on device
  ARG_CHECK(rnd->mode >= 0 && rnd->mode <= 2, "rnd.mode must be 0, 1 or 2");
  if (!fused && N > 0) ARG_CHECK(rnd->noise != nullptr, "injected mode needs noise");
  if (!fused_cand && N > 0) ARG_CHECK(M->S == 0 || rnd->sample_item, "injected candidates need sample_item");
  ARG_CHECK(!M->expo_gathered || !fused_cand, "expo_gathered needs injected candidates (rnd.mode 0 or 2)");
  if (train) {
    ARG_CHECK(G && G->gU && G->gV && G->gW && G->gb && loss, "NULL gradient / loss pointer");
    ARG_CHECK(rank == 0 || rank == 1, "rank must be 0 or 1");
    if (rank == 1) ARG_CHECK(N % 2 == 0, "rank==1 needs [positives ; negatives] (even N)");
    if (rank == 0) ARG_CHECK(Y != nullptr || N == 0, "rank==0 needs Y");
  }
  if (N == 0) {
    if (train) HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), st));
    if (plan && plan->lazy_segU >= 0) return dccf_lazy_step(plan->opt, nullptr, 0, st);
    if (plan) return dccf_opt_phase(plan->opt, OPT_PHASE_ALL, nullptr, nullptr, 0, st);
    return 0;
  }
  const int D = M->D, F = M->F, S1 = M->S + 1, A = M->A, NX = M->n_extra;
  if (train && NX > 0) {
    for (int k = 0; k < NX; ++k) ARG_CHECK(G->gWl[k] && G->gbl[k], "NULL gradient pointer of an extra mlp layer");
  }
  const Lay y = make_layout(N, D, F, M->S, A, NX);
  if (int e = dccf_ws_ensure(ctx, y.total)) return e;
  char* ws = ctx->ws;
  int* cand = (int*)(ws + y.cand);
  float* WT = (float*)(ws + y.WT);
  float* hbuf = (float*)(ws + y.h);
  float* m = (float*)(ws + y.m);
  float* dmns = (float*)(ws + y.dmns);
  const size_t hstride = align_up((size_t)y.L * y.DP * 4, 256);     // bytes between the h / dh buffers of the extra layers

  const rng_key ckey = make_key(rnd->seed, STREAM_CAND, rnd->step);
  const rng_key nkey = make_key(rnd->seed, STREAM_NOISE, rnd->step);
  const rng_key dkey = make_key(rnd->seed, STREAM_DROP, rnd->step);
  const float nscale = -2.0f * 0.69314718055994530942f * M->std * M->std;
  const float kscale = dropout > 0.f ? 1.0f / (float)(1.0 - (double)dropout) : 1.0f;
  const uint32_t thr = dropout > 0.f ? drop_threshold(dropout) : 0u;
  const int64_t ntiles = (y.L + 31) / 32;
  StepRef sr;
  sr.k_dev = rnd->k_dev;
  sr.x_stride = rnd->x_stride;
  sr.x_steps = rnd->x_steps > 0 ? rnd->x_steps : 1;

  // Did the previous dccf_train_step prepare exactly this step (same batch pointer, size, Philox step, seed, tables)?
  const bool prepared = train && !(plan && plan->overlap) && dccf_prep_matches(ctx, M, rnd, X, N);
  // hosted item table (dccf_train_step, no other overlap mode): two sets of "item row touched" bytes owned by the context
  const bool lazy = plan && train && plan->lazy_segU >= 0;
  const bool det_mode = ctx->det && train && !ctx->slot_where;
  int64_t lazy_win_from = -1;     // >= 0: the head of this step's lazy window was advanced by workgroups of the forward launch
  int gw_splits = 0;         // > 0: this step's backward left dW as that many partial sums in ctx->gw_part (GwPart)
  const bool hostv = plan && !plan->overlap && plan->hostv_seg >= 0 && train && !lazy;
  if (hostv) {
    const int64_t nb = (M->item_num + 3) / 4 * 4;
    if (ctx->hv_items != nb) {
      for (int q = 0; q < 2; ++q) {
        if (ctx->hv_flags[q]) HIP_TRY(hipFree(ctx->hv_flags[q]));
        HIP_TRY(hipMalloc((void**)&ctx->hv_flags[q], (size_t)nb));
        HIP_TRY(hipMemsetAsync(ctx->hv_flags[q], 0, (size_t)nb, st));
      }
      ctx->hv_items = nb;
      ctx->hv_prepared = 0;
    }
    // marks somebody prepared for a step that is not this one must not survive
    if (ctx->hv_prepared && !prepared) HIP_TRY(hipMemsetAsync(ctx->hv_flags[ctx->hv_parity], 0, (size_t)nb, st));
  } else if (ctx->hv_prepared && ctx->hv_items) {
    HIP_TRY(hipMemsetAsync(ctx->hv_flags[ctx->hv_parity], 0, (size_t)ctx->hv_items, st));
  }
  ctx->hv_prepared = 0;
  // how much of the item table rides in the backward launch: as much as streams while the roles compute (the hosted waves
  // are few and share the issue pipe; the whole table would stretch the backward by 8 us at B=128)
  int64_t hv_rows = 0;
  if (hostv) {
    const double frac = knobs().hostv_frac;
    const int64_t unit = 256 / M->D;
    hv_rows = min(M->item_num, (int64_t)(frac * (double)M->item_num) / unit * unit);
    if (frac >= 1.0) hv_rows = M->item_num * M->D % 256 == 0 ? M->item_num : M->item_num / unit * unit;
  }
  ctx->last_hosted_rows = hv_rows;
  ctx->prep_valid = 0;       // consumed — or overwritten by the k_prep below
  if (prepared) ++ctx->prep_hits;
  if (!prepared) {
    const int64_t total = (int64_t)(D + y.FP) * y.DP + y.NS + (y.GY > 1 ? y.L : 0) + 1;
    const int grid = (int)min((int64_t)2048, (total + 255) / 256);
    MarkPlan mark;
    memset(&mark, 0, sizeof(mark));
    if (plan && plan->overlap) mark = plan->mark;
    prof_begin(ctx, st);
    hipLaunchKernelGGL(k_prep, dim3(grid), dim3(256), 0, st, *M, WT, D, F, y.DP, y.FP, X, rnd->sample_item, cand, dmns, N,
                       M->S, M->item_num, fused_cand ? 1 : 0, ckey, m, y.GY > 1 ? y.L : (int64_t)0, train ? loss : nullptr, sr,
                       mark, hostv ? ctx->hv_flags[ctx->hv_parity] : (uint8_t*)nullptr);
    prof_end(ctx, 0, st);
  }
  if (lazy) {
    // the rows this step reads (its users and candidates) must be at the previous step before anything reads them.  The
    // optimizer launch of the previous step did that when it knew this batch (pn.cu_blocks); otherwise a launch of its own
    const bool mine = ctx->lazy_prep_step == (int64_t)plan->opt->step && ctx->lazy_prep_claim == (const void*)plan->opt->lazy_claim &&
                      ctx->lazy_prep_id == plan->opt->lazy_id;
    if (!(mine && prepared)) {
      // (claims made for this step number on behalf of a batch that did not come must not shadow the real rows)
      if (mine)
        if (int e = dccf_lazy_reset_claims(plan->opt, st)) return e;
      prof_begin(ctx, st);
      if (int e = dccf_lazy_catchup(plan->opt, X, cand, N, S1, plan->lazy_segU, plan->lazy_segV, st)) return e;
      prof_end(ctx, 7, st);
    }
    ctx->lazy_prep_step = -1;
  }
  if (plan && plan->overlap && !plan->hosted) {
    // fork: rows NOT on this batch's list see only the l2 term -> their optimizer pass needs nothing from this step
    HIP_TRY(hipEventRecord(ctx->ev_fork, st));
    HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    if (int e = dccf_opt_phase(plan->opt, OPT_PHASE_UNTOUCHED, nullptr, nullptr, 0, ctx->side)) return e;
    HIP_TRY(hipEventRecord(ctx->ev_join, ctx->side));
  }
  {
    // one forward kernel for every size: the evenly split K loop is as fast as a rows-per-wave variant at eval size
    // (0.333 vs 0.330 ms at B = 4096, 24.4 vs 24.2 M eval rows/s) and has no tile-count quantisation in between
    const int tile_blocks = (int)min((int64_t)1024, ntiles);
    // lazy training step, small batch: the tiles leave CUs idle (176 tiles at B = 128) — their workgroup slots advance the head
    // of this step's lazy window (the claims of the step are complete: the previous optimizer launch or k_lazy_catchup made them)
    LazyHost lh;
    memset(&lh, 0, sizeof(lh));
    lazy_win_from = -1;
    if (lazy && fused && y.GY == 1 && D == y.DT && y.FP <= 896 && tile_blocks < 256 && knobs().lazy_host_frac > 0.0 && !det_mode) {
      if (int e = dccf_lazy_host_args(plan->opt, &lh)) return e;
      const int64_t span = (int64_t)((double)(lh.win1 - lh.win0) * knobs().lazy_host_frac);
      lh.win1 = lh.win0 + span;
      lh.blocks = span > 0 ? (int)min((int64_t)(256 - tile_blocks), (int64_t)knobs().lazy_host_blocks) : 0;
      if (lh.blocks > 0) lazy_win_from = lh.win1;
    }
    const dim3 grid((unsigned)(tile_blocks + lh.blocks), y.GY), block(512);
    const size_t smem = (size_t)8 * 32 * y.ND * 32 * 4;
    // the whole-chunk instances read the feature (and injected noise) rows as dwordx4
    const bool fal_ok = (uintptr_t)M->feat % 16 == 0 && (fused || (uintptr_t)rnd->noise % 16 == 0);
    prof_begin(ctx, st);
#define LAUNCH_FWD3(D_, MODE_, NCM_, FAL_, MP_, GEN_)                                                                 \
  {                                                                                                                  \
    static bool once = false;                                                                                        \
    if (!once) {                                                                                                     \
      HIP_TRY(hipFuncSetAttribute((const void*)k_noise_fwd<D_, MODE_, NCM_, FAL_, MP_, GEN_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((k_noise_fwd<D_, MODE_, NCM_, FAL_, MP_, GEN_>), grid, block, smem, st, WT, M->b, M->U, M->V, M->feat, X, cand, \
                       rnd->noise, rnd->keep, hbuf, m, y.L, S1, A, F, nkey, dkey, nscale, thr, kscale, sr,          \
                       (train || NX > 0) ? 1 : 0, prepared ? loss : (float*)nullptr, D, y.FP / 768, tile_blocks, lh); \
  }
#define LAUNCH_FWD2(D_, MODE_)                                                                          \
  if (D != D_) { if (y.FP == 768) LAUNCH_FWD3(D_, MODE_, 6, false, false, true) else LAUNCH_FWD3(D_, MODE_, 6, false, true, true) } \
  else if (y.FP == 256) { if (F == 256 && fal_ok) LAUNCH_FWD3(D_, MODE_, 2, true, false, false) else LAUNCH_FWD3(D_, MODE_, 2, false, false, false) }      \
  else if (y.FP == 768) { if (F == 768 && fal_ok) LAUNCH_FWD3(D_, MODE_, 6, true, false, false) else LAUNCH_FWD3(D_, MODE_, 6, false, false, false) } \
  else if (y.FP == 896) LAUNCH_FWD3(D_, MODE_, 7, false, false, false)                                  \
  else LAUNCH_FWD3(D_, MODE_, 6, false, true, false)
#define LAUNCH_FWD(D_) if (fused) LAUNCH_FWD2(D_, 0) else LAUNCH_FWD2(D_, 1)
#define LAUNCH_FWD2W(D_, MODE_) if (y.FP == 768) LAUNCH_FWD3(D_, MODE_, 6, false, false, true) else LAUNCH_FWD3(D_, MODE_, 6, false, true, true)
#define LAUNCH_FWDW(D_) if (fused) LAUNCH_FWD2W(D_, 0) else LAUNCH_FWD2W(D_, 1)
    BY_D_WIDE(D, LAUNCH_FWD, LAUNCH_FWDW)
#undef LAUNCH_FWDW
#undef LAUNCH_FWD2W
#undef LAUNCH_FWD
#undef LAUNCH_FWD2
#undef LAUNCH_FWD3
    prof_end(ctx, 2, st);
  }
  // --n_layers > 1: the extra D -> D layers, one launch each (the last writes m)
  MlpArgs ma;
  memset(&ma, 0, sizeof(ma));
  const size_t mlp_smem = (size_t)y.DP * y.DP * 4;
  const int mlp_grid = (int)max((int64_t)1, min((int64_t)1024, (ntiles + 3) / 4));
#define MLP_ATTR(KERNEL)                                                                                              \
  {                                                                                                                  \
    static bool once = false;                                                                                        \
    if (!once) {                                                                                                     \
      HIP_TRY(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));      \
      once = true;                                                                                                   \
    }                                                                                                                \
  }
  if (NX > 0) {
    prof_begin(ctx, st);
    ma.U = M->U; ma.X = X; ma.m = m; ma.dmns = dmns; ma.L = y.L; ma.rpn = S1 * A; ma.A = A; ma.Dr = D; ma.DP = y.DP;
    ma.fused = fused ? 1 : 0; ma.drop_thr = thr; ma.kscale = kscale; ma.dkey = dkey; ma.sr = sr;
    for (int k = 1; k <= NX; ++k) {
      ma.hin = k == 1 ? hbuf : (const float*)(ws + y.hx + (size_t)(k - 2) * hstride);
      ma.hout = (float*)(ws + y.hx + (size_t)(k - 1) * hstride);
      ma.W = M->Wl[k - 1]; ma.b = M->bl[k - 1];
      ma.keep = (!fused && rnd->keep) ? rnd->keep + (size_t)k * (size_t)y.L * D : nullptr;
      ma.layer = k; ma.last = k == NX ? 1 : 0;
#define LAUNCH_MF(D_) MLP_ATTR(k_mlp_fwd<D_>) hipLaunchKernelGGL(k_mlp_fwd<D_>, dim3(mlp_grid), dim3(256), mlp_smem, st, ma)
      BY_D(D, LAUNCH_MF)
#undef LAUNCH_MF
    }
    prof_end(ctx, 1, st);
  }
  // training with at most 16 candidates per row: the pair epilogue is folded into the backward (wave_dm)
  // deterministic mode (dccf_ctx_set_deterministic): no float atomics anywhere in the backward — run-time-width instances, the
  // pair epilogue as its own launch (one workgroup: its loss sum then has one order), per-slot gradient rows + k_det_sum
  const bool det = ctx->det && train && !ctx->slot_where;
  ARG_CHECK(!det || N * (int64_t)(S1 * A + 1) < 2139062143LL, "deterministic mode: too many slots");
  ARG_CHECK(!det || D <= 128, "deterministic mode covers embedding sizes up to 128 (above, two roles add into one gradient row)");
  const bool fold = train && NX == 0 && D == y.DT && D <= 128 && S1 <= 16 && A <= 4 && N <= knobs().fold_max_n && !det;      // (the 256 tile has run-time-width instances only)
  if (!fold) {
    const int64_t units = (train && rank == 1) ? N / 2 : N;
    const int GS = S1 <= 16 ? 16 : (S1 <= 32 ? 32 : 64);
    const int grid = det ? 1 : (int)min((int64_t)2048, (units * GS + 255) / 256);
    prof_begin(ctx, st);
#define LAUNCH_PE(GS_)                                                                                              \
  hipLaunchKernelGGL((k_pair_epilogue<GS_>), dim3(grid), dim3(256), 0, st, S1, A, Y, m, dmns, pred, loss, N, rank, \
                     train ? 1 : 0)
    if (GS == 16) LAUNCH_PE(16); else if (GS == 32) LAUNCH_PE(32); else LAUNCH_PE(64);
#undef LAUNCH_PE
    prof_end(ctx, 3, st);
  }
  if (train) {
    // roles x column halves on grid.y, row splits on grid.x: about one workgroup per CU in total, 4 batch rows (one per
    // wave) per workgroup at least
    const int bwd_gy = y.DT <= 64 ? 1 : y.DT / 64;                 // the kernel's column groups (compile time, of the tile)
    const int bwd_ni = y.DT <= 128 ? 1 : y.DT / 128;               // item roles = dx roles (128 columns / rows of W_i each)
    const int roles = (y.NC + 2 * bwd_ni) * bwd_gy;
    // small batches: ~1 workgroup per CU; large ones: ~4 per CU (4 waves per SIMD fill the shared VALU / fp32-MFMA pipe)
    const int64_t gx = max((int64_t)1, min((N + BWD_NW - 1) / BWD_NW, (int64_t)max(1, (N >= 2048 ? 1024 : (int)knobs().bwd_wgs) / roles)));
    // hosted optimizer pass: as many extra workgroups as CUs (one beside each role workgroup)
    // (hosting the lazy window here as well was measured and lost: its replay is vector-ALU work with long dependent chains, and
    // the hosted waves — one per SIMD beside the 256-VGPR role waves — cannot issue it fast enough: backward 25 -> 68 us)
    const int opt_rows_y = ((plan && plan->overlap && plan->hosted) || hostv) ? (int)((knobs().hosted_wgs + gx - 1) / gx) : 0;
    const dim3 grid((unsigned)gx, (unsigned)(roles + opt_rows_y));
    const size_t smem = (size_t)4 * 32 * 64 * 4;
    prof_begin(ctx, st);
#define LAUNCH_BWD3(D_, MODE_, FOLD_, DH_, GEN_)                                                                     \
  {                                                                                                                  \
    static bool once = false;                                                                                        \
    if (!once) {                                                                                                     \
      HIP_TRY(hipFuncSetAttribute((const void*)k_bwd<D_, MODE_, FOLD_, DH_, GEN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((k_bwd<D_, MODE_, FOLD_, DH_, GEN_>), grid, dim3(64 * BWD_NW), smem, st, ba);                 \
  }
#define LAUNCH_BWD2(D_, MODE_)                                                                                       \
  if (D != D_ || det) { if (NX > 0) LAUNCH_BWD3(D_, MODE_, false, true, true) else LAUNCH_BWD3(D_, MODE_, false, false, true) } \
  else if (NX > 0) LAUNCH_BWD3(D_, MODE_, false, true, false)                                                       \
  else if (fold) LAUNCH_BWD3(D_, MODE_, true, false, false)                                                         \
  else LAUNCH_BWD3(D_, MODE_, false, false, false)
#define LAUNCH_BWD(D_) if (fused) LAUNCH_BWD2(D_, 0) else LAUNCH_BWD2(D_, 1)
    BwdArgs ba;
    ba.dh = nullptr;
    ba.det_u = ba.det_v = ba.det_gb = nullptr;
    ba.det_ld = y.DP * y.GY > 64 ? 128 : 64;
    const int64_t det_nv = A == 2 ? y.NS : y.L;            // item slots of the deterministic mode
    const int64_t gx_det = max((int64_t)1, min((N + BWD_NW - 1) / BWD_NW, (int64_t)max(1, (N >= 2048 ? 1024 : (int)knobs().bwd_wgs) / ((y.NC + 2 * (y.DT <= 128 ? 1 : y.DT / 128)) * (y.DT <= 64 ? 1 : y.DT / 64)))));
    if (det) {
      const size_t need = ((size_t)(N + det_nv) * D + (size_t)gx_det * BWD_NW * ba.det_ld) * sizeof(float);
      if (need > ctx->det_bytes) {
        if (ctx->det_buf) HIP_TRY(hipFree(ctx->det_buf));
        ctx->det_buf = nullptr; ctx->det_bytes = 0;
        HIP_TRY(hipMalloc((void**)&ctx->det_buf, need));
        ctx->det_bytes = need;
      }
      const int64_t rows = M->user_num + M->item_num;
      if (rows > ctx->det_owner_n) {
        if (ctx->det_owner) HIP_TRY(hipFree(ctx->det_owner));
        ctx->det_owner = nullptr; ctx->det_owner_n = 0;
        HIP_TRY(hipMalloc((void**)&ctx->det_owner, (size_t)rows * sizeof(int)));
        HIP_TRY(hipMemsetAsync(ctx->det_owner, 0x7f, (size_t)rows * sizeof(int), st));
        ctx->det_owner_n = rows;
      }
      ba.det_u = ctx->det_buf;
      ba.det_v = ba.det_u + (size_t)N * D;
      ba.det_gb = ba.det_v + (size_t)det_nv * D;
    }
    if (NX > 0) {
      prof_begin(ctx, st);
      // the extra layers, last to first: gW_k, gb_k and dh_{k-1}; then the user gradient from the last layer's output
      float* dhA = (float*)(ws + y.dh);
      float* dhB = (float*)(ws + y.dh + hstride);
      const int bgrid = det ? 1 : (int)max((int64_t)1, min((int64_t)512, (ntiles + 3) / 4));
      ma.det = det ? 1 : 0;
      for (int k = NX; k >= 1; --k) {
        ma.hin = k == 1 ? hbuf : (const float*)(ws + y.hx + (size_t)(k - 2) * hstride);
        ma.hout = (float*)(ws + y.hx + (size_t)(k - 1) * hstride);
        ma.W = M->Wl[k - 1]; ma.b = M->bl[k - 1];
        ma.gW = G->gWl[k - 1]; ma.gb = G->gbl[k - 1];
        ma.layer = k; ma.last = k == NX ? 1 : 0;
        ma.dhin = ((NX - k) & 1) ? dhA : dhB;          // (unused by the last layer)
        ma.dhout = ((NX - k) & 1) ? dhB : dhA;
#define LAUNCH_MB(D_) MLP_ATTR(k_mlp_bwd<D_>) hipLaunchKernelGGL(k_mlp_bwd<D_>, dim3(bgrid), dim3(256), mlp_smem, st, ma)
        BY_D(D, LAUNCH_MB)
#undef LAUNCH_MB
      }
      ba.dh = ma.dhout;
      hipLaunchKernelGGL(k_gu_last, dim3((unsigned)min((int64_t)1024, (N + 3) / 4)), dim3(256), 0, st,
                         (const float*)(ws + y.hx + (size_t)(NX - 1) * hstride), (const float*)dmns, X, G->gU, G->touchedU, N, S1, A,
                         D, y.DP, ctx->slot_where, ctx->slot_rows, ctx->slot_offU, ctx->slot_cap, sr, ba.det_u);
      prof_end(ctx, 4, st);
    }
    ba.W = M->W; ba.U = M->U; ba.V = M->V; ba.feat = M->feat; ba.X = X; ba.cand = cand; ba.dmns = dmns; ba.hbuf = hbuf;
    ba.noise = rnd->noise; ba.gU = G->gU; ba.gV = G->gV; ba.gW = G->gW; ba.gb = G->gb;
    ba.touchedU = G->touchedU; ba.touchedV = G->touchedV; ba.N = N; ba.S1 = S1; ba.A = A;
    ba.F = F; ba.NC = y.NC; ba.Dr = D; ba.kscale = kscale; ba.nscale = nscale; ba.nkey = nkey; ba.sr = sr;
    ba.m = m; ba.Y = Y; ba.pred = pred; ba.loss = loss; ba.rank = rank;
    ba.slot_where = ctx->slot_where; ba.slot_rows = ctx->slot_rows; ba.slot_offU = ctx->slot_offU; ba.slot_offV = ctx->slot_offV;
    ba.slot_cap = ctx->slot_cap;
    // small batches under the lazy optimizer: dW as per-split partial sums that the optimizer launch adds (32 KB of float
    // atomics per CU were the last 6 us of every chunk-role workgroup: ~1 wave instruction per 50 ns per CU)
    ba.gw_part = nullptr;
    ba.gw_stride = (int64_t)D * (D + F);
    gw_splits = 0;
    if (det || (lazy && knobs().gw_part && N < 2048 && !ctx->slot_where && plan->opt->p <= M->W &&
        M->W + (int64_t)D * (D + F) <= plan->opt->p + plan->opt->n && G->gW == plan->opt->g + (M->W - plan->opt->p))) {
      const size_t need = (size_t)gx * (size_t)ba.gw_stride * sizeof(float);
      if (need > ctx->gw_part_bytes) {
        if (ctx->gw_part) HIP_TRY(hipFree(ctx->gw_part));
        ctx->gw_part = nullptr;
        ctx->gw_part_bytes = 0;
        HIP_TRY(hipMalloc((void**)&ctx->gw_part, need));
        ctx->gw_part_bytes = need;
      }
      ba.gw_part = ctx->gw_part;
      gw_splits = (int)gx;
    }
    memset(&ba.oj, 0, sizeof(ba.oj));
    ba.opt_rows_y = opt_rows_y;
    if (opt_rows_y) {
      if (int e = dccf_opt_job(plan->opt, &ba.oj)) return e;
      if (hostv) {          // the item segment alone, as a job of its own: marks = this step's bytes
        const int q = plan->hostv_seg;
        OptJob& j = ba.oj;
        const int w = j.sg.width[q];
        const int64_t b0 = j.sg.begin[q], e0 = b0 + hv_rows * w;
        j.p += b0; j.g += b0;
        if (j.s1) j.s1 += b0;
        if (j.s2) j.s2 += b0;
        j.n = e0 - b0;
        memset(&j.sg, 0, sizeof(j.sg));
        j.sg.n = 1;
        j.sg.begin[0] = 0; j.sg.end[0] = e0 - b0; j.sg.width[0] = w;
        j.sg.flags[0] = ctx->hv_flags[ctx->hv_parity];
        ba.touchedV = nullptr;       // (the bytes are there already; the optimizer launch consumes them)
      }
    }
#define LAUNCH_BWDW(D_) if (fused) LAUNCH_BWD3(D_, 0, false, false, true) else LAUNCH_BWD3(D_, 1, false, false, true)
    BY_D_WIDE(D, LAUNCH_BWD, LAUNCH_BWDW)
#undef LAUNCH_BWDW
#undef LAUNCH_BWD
#undef LAUNCH_BWD2
#undef LAUNCH_BWD3
    if (det) {               // per-slot rows -> gU / gV in slot order; gb and dW partial sums in index order
      ARG_CHECK(gx == gx_det, "deterministic mode: row splits");
      DetArgs da;
      memset(&da, 0, sizeof(da));
      da.X = X; da.cand = cand; da.N = N; da.NV = det_nv; da.user_num = M->user_num; da.S1 = S1; da.A = A; da.Dr = D;
      da.det_u = ba.det_u; da.det_v = ba.det_v; da.gU = G->gU; da.gV = G->gV;
      da.touchedU = G->touchedU; da.touchedV = G->touchedV; da.owner = ctx->det_owner;
      da.det_gb = ba.det_gb; da.n_gb = (int)gx * BWD_NW; da.det_ld = ba.det_ld; da.gb = G->gb;
      da.gw_part = ctx->gw_part; da.nsplit = gw_splits; da.gw_stride = ba.gw_stride; da.gW = G->gW;
      da.sr = sr;
      const int64_t slots = N + det_nv;
      hipLaunchKernelGGL(k_det_owner, dim3((unsigned)min((int64_t)1024, (slots + 255) / 256)), dim3(256), 0, st, da);
      const int row_blocks = (int)min((int64_t)2048, (slots + 3) / 4);
      hipLaunchKernelGGL(k_det_sum, dim3((unsigned)(row_blocks + 64)), dim3(256), 0, st, da, row_blocks);
      gw_splits = 0;         // (dW is in gW already: the optimizer launch must not add the partial sums again)
    }
    prof_end(ctx, 5, st);
  }
  HIP_TRY(hipGetLastError());
  if (plan) {
    prof_begin(ctx, st);
    if (lazy) {
      GwPart gp;
      memset(&gp, 0, sizeof(gp));
      if (gw_splits > 0) {
        gp.part = ctx->gw_part;
        gp.nsplit = gw_splits;
        gp.stride = (int64_t)D * (D + F);
        gp.w_begin = M->W - plan->opt->p;
        gp.w_end = gp.w_begin + gp.stride;
      }
      // touched rows + W, b + this step's window of the untouched rows (+ the next step's preparation)
      const bool prep_ok = plan->X_next && fused_cand && rnd->k_dev == nullptr && plan->opt->p <= M->W &&
                           M->W + (int64_t)D * (D + F) <= plan->opt->p + plan->opt->n;
      if (prep_ok) {
        PrepNext pn;
        if (int e = dccf_prep_next_fill(ctx, M, N, plan->X_next, rnd->seed, plan->step_next, &pn)) return e;
        pn.w_begin = M->W - plan->opt->p;
        pn.w_end = pn.w_begin + (int64_t)D * (D + F);
        pn.blocks = (int)min((int64_t)64, (y.NS + pn.Lm + 255) / 256);
        if (knobs().lazy_cu) {
          pn.cu_blocks = (int)min((int64_t)knobs().lazy_cu_blocks, (N * (int64_t)(S1 + 1) + 3) / 4);
          pn.cu_segU = plan->lazy_segU;
          pn.cu_segV = plan->lazy_segV;
        }
        if (int e = dccf_lazy_step(plan->opt, &pn, N * (int64_t)(S1 + 1), st, &gp, lazy_win_from)) return e;
        dccf_prep_next_commit(ctx, M, N, plan->X_next, rnd->seed, plan->step_next);
        if (pn.cu_blocks) {
          ctx->lazy_prep_step = (int64_t)plan->opt->step + 1;
          ctx->lazy_prep_claim = plan->opt->lazy_claim;
          ctx->lazy_prep_id = plan->opt->lazy_id;
        }
      } else {
        if (int e = dccf_lazy_step(plan->opt, nullptr, N * (int64_t)(S1 + 1), st, &gp, lazy_win_from)) return e;
      }
    } else if (plan->overlap) {
      if (!plan->hosted) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
      if (int e = dccf_opt_phase(plan->opt, OPT_PHASE_TOUCHED, plan->mark.list, plan->mark.cnt, plan->max_rows, st)) return e;
    } else if (plan->X_next && fused_cand && rnd->k_dev == nullptr && plan->opt->p <= M->W &&
               M->W + (int64_t)D * (D + F) <= plan->opt->p + plan->opt->n && (M->W - plan->opt->p) % 4 == 0) {
      // the optimizer launch also prepares the next step: candidates + exposures of X_next, zeroed accumulators, and W^T
      // written while W is updated — the next call starts with its forward kernel
      PrepNext pn;
      if (int e = dccf_prep_next_fill(ctx, M, N, plan->X_next, rnd->seed, plan->step_next, &pn)) return e;
      pn.w_begin = M->W - plan->opt->p;
      pn.w_end = pn.w_begin + (int64_t)D * (D + F);
      pn.blocks = (int)min((int64_t)64, (y.NS + pn.Lm + 255) / 256);
      if (hostv) {
        pn.markV = ctx->hv_flags[1 - ctx->hv_parity];
        if (int e = dccf_opt_all_prep_to(plan->opt, plan->hostv_seg, ctx->hv_flags[ctx->hv_parity], hv_rows, &pn, st)) return e;
        ctx->hv_prepared = 1;
      } else {
        if (int e = dccf_opt_all_prep(plan->opt, &pn, st)) return e;
      }
      dccf_prep_next_commit(ctx, M, N, plan->X_next, rnd->seed, plan->step_next);
    } else if (hostv) {
      if (int e = dccf_opt_all_prep_to(plan->opt, plan->hostv_seg, ctx->hv_flags[ctx->hv_parity], hv_rows, nullptr, st)) return e;
    } else {
      if (int e = dccf_opt_phase(plan->opt, OPT_PHASE_ALL, nullptr, nullptr, 0, st)) return e;
    }
    if (hostv) ctx->hv_parity ^= 1;
    prof_end(ctx, 6, st);
  }
  return 0;
}

extern "C" int dccf_train_step(dccf_ctx* ctx, const dccf_model_t* M, const dccf_rand_t* rnd, const int64_t* X, const float* Y,
                               int64_t N, int32_t rank, float dropout, const dccf_grads_t* G, const dccf_opt_t* opt,
                               float* prediction, float* loss, const int64_t* X_next, uint64_t step_next, void* stream) {
  ARG_CHECK(ctx && M && G && opt, "NULL argument");
  ARG_CHECK(rnd && rnd->k_dev == nullptr, "dccf_train_step takes a host-side step (no k_dev)");
  StepPlan plan;
  memset(&plan, 0, sizeof(plan));
  plan.opt = opt;
  plan.overlap = opt->overlap != 0 && N > 0;
  plan.hosted = opt->overlap == 2;
  plan.hostv_seg = -1;
  // (at 2B > 2048 the backward is long and the hosted pass buys nothing: measured +0.6 % at 2B = 8192)
  if (!plan.overlap && N > 0 && N <= 2048 && rank == 1 && rnd->mode == 1 && opt->nseg >= 1 && opt->seg_begin && opt->seg_rows && opt->seg_width &&
      G->touchedV && knobs().hostv) {
    for (int q = 0; q < opt->nseg; ++q)
      if (opt->p + opt->seg_begin[q] == M->V && opt->seg_rows[q] == M->item_num && opt->seg_width[q] == M->D &&
          opt->seg_flags[q] == G->touchedV && (M->D == 16 || M->D == 32 || M->D == 64 || M->D == 128))
        plan.hostv_seg = q;
  }
  plan.X_next = (N > 0 && rank == 1) ? X_next : nullptr;
  plan.step_next = step_next;
  plan.lazy_segU = plan.lazy_segV = -1;
  if (opt->lazy_K > 0) {
    ARG_CHECK(!plan.overlap, "the lazy optimizer (lazy_K > 0) excludes the overlap modes");
    ARG_CHECK(opt->nseg >= 2 && opt->seg_begin && opt->seg_rows && opt->seg_width && opt->seg_flags, "lazy optimizer needs the U and V row segments");
    for (int q = 0; q < opt->nseg; ++q) {
      if (opt->p + opt->seg_begin[q] == M->U && opt->seg_rows[q] == M->user_num && opt->seg_width[q] == M->D) plan.lazy_segU = q;
      if (opt->p + opt->seg_begin[q] == M->V && opt->seg_rows[q] == M->item_num && opt->seg_width[q] == M->D) plan.lazy_segV = q;
    }
    ARG_CHECK(plan.lazy_segU >= 0 && plan.lazy_segV >= 0 && G->touchedU == opt->seg_flags[plan.lazy_segU] &&
                  G->touchedV == opt->seg_flags[plan.lazy_segV],
              "lazy optimizer: model->U / model->V must be row segments of opt->p whose flags are grads->touchedU / touchedV");
    plan.hostv_seg = -1;
  }
  if (plan.overlap) {
    // the segments that hold U and V, by address
    int qU = -1, qV = -1;
    ARG_CHECK(opt->nseg >= 2 && opt->nseg <= 4 && opt->seg_begin && opt->seg_rows && opt->seg_width && opt->seg_flags && opt->p,
              "overlap needs the U and V row segments");
    for (int q = 0; q < opt->nseg; ++q) {
      if (opt->p + opt->seg_begin[q] == M->U) qU = q;
      if (opt->p + opt->seg_begin[q] == M->V) qV = q;
    }
    ARG_CHECK(qU >= 0 && qV >= 0 && qU != qV, "model->U / model->V are not the start of a row segment of opt->p");
    ARG_CHECK(opt->seg_rows[qU] == M->user_num && opt->seg_rows[qV] == M->item_num && opt->seg_width[qU] == M->D &&
                  opt->seg_width[qV] == M->D,
              "U / V segments do not match the model");
    ARG_CHECK(G->touchedU == opt->seg_flags[qU] && G->touchedV == opt->seg_flags[qV] && G->touchedU && G->touchedV,
              "grads->touchedU/V must be the flags of the U / V segments");
    ARG_CHECK((uintptr_t)G->touchedU % 4 == 0 && (uintptr_t)G->touchedV % 4 == 0, "touched flags must be 4-byte aligned");
    plan.max_rows = N * (int64_t)(M->S + 2);
    if (int e = dccf_step_ensure(ctx, plan.max_rows)) return e;
    plan.mark.flagU = (uint32_t*)G->touchedU;
    plan.mark.flagV = (uint32_t*)G->touchedV;
    plan.mark.tagU = (int64_t)qU << 40;
    plan.mark.tagV = (int64_t)qV << 40;
    plan.mark.list = ctx->tl_list;
    plan.mark.cnt = ctx->tl_cnt + ctx->tl_parity;
    plan.mark.cnt_next = ctx->tl_cnt + (ctx->tl_parity ^ 1);
    ctx->tl_parity ^= 1;
  }
  return run_dccf(ctx, M, rnd, X, Y, N, rank, dropout, G, prediction, loss, true, (hipStream_t)stream, &plan);
}

// ================================================================================================ projected-noise predict
// Evaluation only (no gradient is taken): z_noise = W_f eps with eps ~ N(0, std^2 I_F) iid per row IS a D-dimensional
// Gaussian N(0, std^2 W_f W_f^T).  Drawing xi ~ N(0, I_D) and forming Lt^T xi with L L^T = std^2 W_f W_f^T samples exactly
// that distribution with D instead of F normals and a K = D instead of K = F product per row; the deterministic part
// W_f feat[i] is one row of a table computed once per evaluation.  The output has the distribution of DCCF.predict
// (src/models/DCCF.py:84-97) — not its random stream, which no implementation can reproduce anyway.
//   k_feat_proj   Pf[i][d] = sum_f feat[i][f] W[d][D+f]                      (fp32 MFMA, one wave per 32 items)
//   k_gram_chol   G = std^2 W_f W_f^T in fp64, pivot-clamped Cholesky, Lt[k][d] = L[d][k]   (one workgroup)
//   k_fwd_proj    z = V[cand] W_i^T + xi Lt + Pf[i0] + b, relu, dropout, m = <U[u], h>   (rows-per-wave, K = 2D)
template <int D_>
__global__ __launch_bounds__(256) void k_feat_proj(const float* __restrict__ WT, const float* __restrict__ feat, int64_t item_num,
                                                   int F, float* __restrict__ Pf, int Dr) {
  constexpr int D = D_;
  constexpr int DP = D <= 32 ? 32 : (D + 63) / 64 * 64;
  constexpr int NT = DP / 32;
  const int lane = threadIdx.x & 63, h = lane >> 5, c31 = lane & 31;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ntiles = (item_num + 31) / 32;
  for (int64_t t = wave; t < ntiles; t += nw) {
    const int64_t i = min(t * 32 + c31, item_num - 1);
    const float* frow = feat + i * F;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    for (int k = 0; k < F; k += 8) {                  // 4 k-steps per batch of loads
      float a[4], b[4][NT];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int f = min(k + 2 * o + h, F - 1);
        a[o] = frow[f];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[o][nt] = WT[(int64_t)(Dr + f) * DP + nt * 32 + c31];
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float av = (k + 2 * o + h < F) ? a[o] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = MFMA32(av, b[o][nt], acc[nt]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int d = nt * 32 + c31;
        if (row < item_num && d < Dr) Pf[row * Dr + d] = acc[nt][r];
      }
  }
}

__global__ __launch_bounds__(256) void k_gram_chol(const float* __restrict__ W, int D, int F, float std, float* __restrict__ Lt) {
  extern __shared__ double gd[];                      // [D][D + 1]
  const int LD = D + 1;
  for (int idx = threadIdx.x; idx < D * D; idx += blockDim.x) {
    const int a = idx / D, b = idx % D;
    if (b > a) continue;                              // lower triangle
    const float* wa = W + (int64_t)a * (D + F) + D;
    const float* wb = W + (int64_t)b * (D + F) + D;
    double s = 0.0;
    for (int f = 0; f < F; ++f) s += (double)wa[f] * (double)wb[f];
    gd[a * LD + b] = s * (double)std * (double)std;
  }
  __syncthreads();
  for (int k = 0; k < D; ++k) {                       // right-looking Cholesky; a non-positive pivot (rank-deficient W_f)
    if (threadIdx.x == 0) {                           // gives a zero column: no variance in that direction, as it should be
      const double p = gd[k * LD + k];
      gd[k * LD + k] = p > 0.0 ? sqrt(p) : 0.0;
    }
    __syncthreads();
    const double piv = gd[k * LD + k];
    for (int a = k + 1 + threadIdx.x; a < D; a += blockDim.x) gd[a * LD + k] = piv > 0.0 ? gd[a * LD + k] / piv : 0.0;
    __syncthreads();
    const int n = D - k - 1;
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
      const int a = k + 1 + idx / n, b = k + 1 + idx % n;
      if (b <= a) gd[a * LD + b] -= gd[a * LD + k] * gd[b * LD + k];
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < D * D; idx += blockDim.x) {
    const int k = idx / D, d = idx % D;
    Lt[idx] = d >= k ? (float)gd[d * LD + k] : 0.f;   // Lt[k][d] = L[d][k]
  }
}

template <int D_>
__global__ __launch_bounds__(512) void k_fwd_proj(const float* __restrict__ WT, const float* __restrict__ Lt,
                                                  const float* __restrict__ Pf, const float* __restrict__ bias,
                                                  const float* __restrict__ U, const float* __restrict__ V, const int64_t* X,
                                                  const int* __restrict__ cand, float* __restrict__ m, int64_t L, int S1, int A,
                                                  rng_key xkey, rng_key dkey, uint32_t drop_thr, float kscale, int Dr) {
  extern __shared__ float wl[];                       // [2D][DW]: rows 0..D-1 = W_i^T, rows D..2D-1 = Lt (zero beyond Dr)
  constexpr int D = D_;
  constexpr int DP = D <= 32 ? 32 : (D + 63) / 64 * 64;
  constexpr int ND = D <= 32 ? 1 : 2;
  constexpr int DW = ND * 32;
  constexpr int NWV = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int dbase = blockIdx.y * DW;
  const uint32_t rows_per_n = (uint32_t)(S1 * A);
  for (int idx = threadIdx.x; idx < 2 * D * DW; idx += blockDim.x) {
    const int k = idx / DW, c = idx % DW;
    const int d = dbase + c;
    wl[idx] = k < D ? (k < Dr ? WT[(int64_t)k * DP + d] : 0.f) : ((d < Dr && k - D < Dr) ? Lt[(k - D) * Dr + d] : 0.f);
  }
  __syncthreads();
  const int64_t ntiles = (L + 31) / 32;
  const float one = -2.0f * 0.69314718055994530942f;  // noise4's scale for unit variance
  for (int64_t tile = (int64_t)blockIdx.x * NWV + wave; tile < ntiles; tile += (int64_t)gridDim.x * NWV) {
    const int64_t l = tile * 32 + c31;
    const int64_t lc = l < L ? l : L - 1;
    const float* vrow = V + (int64_t)cand[(uint32_t)lc / (uint32_t)A] * Dr;
    f32x16 acc[ND];
#pragma unroll
    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll 2
    for (int g = 0; g < D / 8; ++g) {                 // 4 k-steps: k = 8g + 2o + h
      float av[4], xi[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) av[o] = vrow[min(8 * g + 2 * o + h, Dr - 1)];
      noise4((uint32_t)l, (uint32_t)(2 * g + h), xkey, one, xi);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int k = 8 * g + 2 * o + h;
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
          acc[nt] = MFMA32(av[o], wl[k * DW + nt * 32 + c31], acc[nt]);
          acc[nt] = MFMA32(xi[o], wl[(D + k) * DW + nt * 32 + c31], acc[nt]);
        }
      }
    }
    const int64_t base = tile * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
      int64_t urow[4], prow[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int64_t lr = base + w + 8 * g4 + 4 * h;
        const int64_t n = (int64_t)((uint32_t)(lr < L ? lr : L - 1) / rows_per_n);
        urow[w] = X[2 * n] * Dr;
        prow[w] = X[2 * n + 1] * Dr;
      }
#pragma unroll
      for (int nt = 0; nt < ND; ++nt) {
        const int d = dbase + nt * 32 + c31;
        const bool dv = d < Dr;
        const int dc = dv ? d : 0;
        const float bd = bias[dc];
        u32x4 r4{0, 0, 0, 0};
        if (drop_thr) r4 = philox4x32_10((uint32_t)((base >> 2) + 2 * g4 + h), (uint32_t)d, dkey.s0, dkey.s1, dkey.k0, dkey.k1);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const bool kept = drop_thr ? pick4(r4, w) >= drop_thr : true;
          const float z = acc[nt][g4 * 4 + w] + Pf[prow[w] + dc] + bd;
          const float hv = (dv && z > 0.f && kept) ? z * kscale : 0.f;
          part[w] = fmaf(U[urow[w] + dc], hv, part[w]);
        }
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float v = part[w];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        const int64_t lr = base + w + 8 * g4 + 4 * h;
        if (c31 == 0 && lr < L) {
          if (gridDim.y == 1) m[lr] = v;
          else atomicAdd(&m[lr], v);
        }
      }
    }
  }
}

extern "C" int dccf_eval_prepare(dccf_ctx* ctx, const dccf_model_t* M, float* Pf, float* Lt, void* stream) {
  ARG_CHECK(ctx && Pf && Lt, "NULL argument");
  if (int e = check_model(M)) return e;
  ARG_CHECK(M->n_extra == 0, "projected evaluation noise covers --n_layers 1 (use --eval_noise full)");
  ARG_CHECK(M->D <= 128, "projected evaluation noise covers embedding sizes up to 128 (use --eval_noise full)");
  hipStream_t st = (hipStream_t)stream;
  const int D = M->D, F = M->F;
  const Lay y = make_layout(0, D, F, M->S, M->A);
  if (int e = dccf_ws_ensure(ctx, y.total)) return e;
  float* WT = (float*)(ctx->ws + y.WT);
  {
    const int64_t total = (int64_t)(D + y.FP) * y.DP + 1;
    StepRef sr;
    memset(&sr, 0, sizeof(sr));
    sr.x_steps = 1;
    MarkPlan mark;
    memset(&mark, 0, sizeof(mark));
    rng_key k0 = make_key(0, STREAM_CAND, 0);
    hipLaunchKernelGGL(k_prep, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *M, WT, D, F, y.DP, y.FP, (const int64_t*)nullptr,
                       (const int64_t*)nullptr, (int*)nullptr, (float*)nullptr, (int64_t)0, M->S, M->item_num, 0, k0,
                       (float*)nullptr, (int64_t)0, (float*)nullptr, sr, mark, (uint8_t*)nullptr);
  }
  const int grid = (int)min((int64_t)2048, ((M->item_num + 31) / 32 + 3) / 4);
#define LAUNCH_FP(D_) hipLaunchKernelGGL(k_feat_proj<D_>, dim3(grid), dim3(256), 0, st, WT, M->feat, M->item_num, F, Pf, D)
  BY_D(D, LAUNCH_FP)
#undef LAUNCH_FP
  const size_t smem = (size_t)D * (D + 1) * sizeof(double);
  static bool once = false;
  if (!once) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_gram_chol, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    once = true;
  }
  hipLaunchKernelGGL(k_gram_chol, dim3(1), dim3(256), smem, st, M->W, D, F, M->std, Lt);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dccf_predict_projected(dccf_ctx* ctx, const dccf_model_t* M, const dccf_rand_t* rnd, const int64_t* X, int64_t N,
                                      float dropout, const float* Pf, const float* Lt, float* prediction, void* stream) {
  ARG_CHECK(ctx && rnd && Pf && Lt, "NULL argument");
  if (int e = check_model(M)) return e;
  ARG_CHECK(rnd->mode == 1 && rnd->k_dev == nullptr, "projected predict draws everything on the device (rnd.mode = 1)");
  ARG_CHECK(M->n_extra == 0, "projected evaluation noise covers --n_layers 1 (use --eval_noise full)");
  ARG_CHECK(M->D <= 128, "projected evaluation noise covers embedding sizes up to 128 (use --eval_noise full)");
  ARG_CHECK(N >= 0 && N * (int64_t)(M->S + 1) * M->A < 4294967296LL, "N*(S+1)*A must be < 2^32");
  ARG_CHECK(N == 0 || (X && prediction), "NULL X / prediction");
  ARG_CHECK(dropout >= 0.f && dropout < 1.f, "dropout must be in [0,1)");
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int D = M->D, F = M->F, S1 = M->S + 1, A = M->A;
  const Lay y = make_layout(N, D, F, M->S, A);
  if (int e = dccf_ws_ensure(ctx, y.total)) return e;
  char* ws = ctx->ws;
  int* cand = (int*)(ws + y.cand);
  float* WT = (float*)(ws + y.WT);
  float* m = (float*)(ws + y.m);
  float* dmns = (float*)(ws + y.dmns);
  StepRef sr;
  memset(&sr, 0, sizeof(sr));
  sr.x_steps = 1;
  MarkPlan mark;
  memset(&mark, 0, sizeof(mark));
  {
    const int64_t total = (int64_t)(D + y.FP) * y.DP + y.NS + (y.GY > 1 ? y.L : 0) + 1;
    hipLaunchKernelGGL(k_prep, dim3((unsigned)min((int64_t)2048, (total + 255) / 256)), dim3(256), 0, st, *M, WT, D, F, y.DP, y.FP, X,
                       rnd->sample_item, cand, dmns, N, M->S, M->item_num, 1, make_key(rnd->seed, STREAM_CAND, rnd->step), m,
                       y.GY > 1 ? y.L : (int64_t)0, (float*)nullptr, sr, mark, (uint8_t*)nullptr);
  }
  const float kscale = dropout > 0.f ? 1.0f / (float)(1.0 - (double)dropout) : 1.0f;
  const uint32_t thr = dropout > 0.f ? drop_threshold(dropout) : 0u;
  const int64_t ntiles = (y.L + 31) / 32;
  const dim3 grid((unsigned)min((int64_t)1024, (ntiles + 7) / 8), y.GY);
  const size_t smem = (size_t)2 * y.DT * (y.ND * 32) * 4;
#define LAUNCH_PJ(D_)                                                                                               \
  {                                                                                                                  \
    static bool once = false;                                                                                        \
    if (!once) {                                                                                                     \
      HIP_TRY(hipFuncSetAttribute((const void*)k_fwd_proj<D_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
      once = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL(k_fwd_proj<D_>, grid, dim3(512), smem, st, WT, Lt, Pf, M->b, M->U, M->V, X, cand, m, y.L, S1, A, \
                       make_key(rnd->seed, STREAM_XI, rnd->step), make_key(rnd->seed, STREAM_DROP, rnd->step), thr, kscale, D); \
  }
  BY_D(D, LAUNCH_PJ)
#undef LAUNCH_PJ
  {
    const int GS = S1 <= 16 ? 16 : (S1 <= 32 ? 32 : 64);
    const int grid2 = (int)min((int64_t)2048, (N * GS + 255) / 256);
#define LAUNCH_PE2(GS_) hipLaunchKernelGGL((k_pair_epilogue<GS_>), dim3(grid2), dim3(256), 0, st, S1, A, (const float*)nullptr, m, dmns, prediction, (float*)nullptr, N, 1, 0)
    if (GS == 16) LAUNCH_PE2(16); else if (GS == 32) LAUNCH_PE2(32); else LAUNCH_PE2(64);
#undef LAUNCH_PE2
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dccf_predict(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X,
                            int64_t N, float dropout, float* prediction, void* stream) {
  return run_dccf(ctx, model, rnd, X, nullptr, N, 1, dropout, nullptr, prediction, nullptr, false, (hipStream_t)stream);
}

extern "C" int dccf_train_fwdbwd(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X,
                                 const float* Y, int64_t N, int32_t rank, float dropout, const dccf_grads_t* grads,
                                 float* prediction, float* loss, void* stream) {
  return run_dccf(ctx, model, rnd, X, Y, N, rank, dropout, grads, prediction, loss, true, (hipStream_t)stream);
}

// Diagnostics (bench.py prices the launches by the bytes they move): item rows whose untouched-row optimizer pass rode in the
// backward launch of the LAST training call on this context (0: the optimizer launch did the whole pass).
extern "C" int dccf_ctx_hosted_rows(const dccf_ctx* ctx, int64_t* out) {
  ARG_CHECK(ctx && out, "NULL argument");
  *out = ctx->last_hosted_rows;
  return 0;
}

extern "C" int dccf_ctx_reserve(dccf_ctx* ctx, int64_t max_rows, int32_t D, int32_t F, int32_t S, int32_t A) {
  ARG_CHECK(ctx != nullptr && max_rows >= 0 && D > 0 && F > 0 && S >= 0 && A > 0, "bad reserve arguments");
  const Lay y = make_layout(max_rows, D, F, S, A);
  return dccf_ws_ensure(ctx, y.total);
}

// ================================================================================================ debug: workspace
// Copies one workspace array of the LAST call with these shapes to `dst` (device) — for the parity tests only.
// which: 0 cand(int32 [N*S1]) 1 WT 3 h [L,DP] 4 m [L] 5 dmns [N*S1]; info = DP, FP, count, 4
extern "C" int dccf_debug_workspace(dccf_ctx* ctx, int64_t N, int32_t D, int32_t F, int32_t S, int32_t A, int32_t which,
                                    void* dst, int64_t* info, void* stream) {
  ARG_CHECK(ctx && info, "NULL argument");
  const Lay y = make_layout(N, D, F, S, A);
  ARG_CHECK(y.total <= ctx->ws_bytes, "workspace smaller than this layout");
  size_t off = 0, cnt = 0;
  switch (which) {
    case 0: off = y.cand; cnt = (size_t)y.NS; break;
    case 1: off = y.WT; cnt = (size_t)(D + y.FP) * y.DP; break;
    case 3: off = y.h; cnt = (size_t)y.L * y.DP; break;
    case 4: off = y.m; cnt = (size_t)y.L; break;
    case 5: off = y.dmns; cnt = (size_t)y.NS; break;
    default: return dccf_fail(-1, "argument error: unknown workspace array");
  }
  info[0] = y.DP; info[1] = y.FP; info[2] = (int64_t)cnt; info[3] = 4;
  if (dst) HIP_TRY(hipMemcpyAsync(dst, ctx->ws + off, cnt * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// ================================================================================================ debug streams
__global__ void k_dbg_cand(int64_t N, int S, int64_t item_num, rng_key key, int64_t* out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N * S; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / S;
    const int s = (int)(i % S);
    const u32x4 r = philox4x32_10((uint32_t)n, (uint32_t)(s >> 2), key.s0, key.s1, key.k0, key.k1);
    out[i] = (int64_t)(((uint64_t)pick4(r, s & 3) * (uint64_t)item_num) >> 32);
  }
}
__global__ void k_dbg_noise(int64_t L, int F, rng_key key, float nscale, float* out) {
  const int nc1 = ((F + 127) / 128) * 32;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < L * nc1; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = i / nc1;
    const int c1 = (int)(i % nc1);
    float z[4];
    noise4((uint32_t)l, (uint32_t)c1, key, nscale, z);
    for (int o = 0; o < 4; ++o) {
      const int f = 128 * (c1 / 32) + (c1 % 32) + 32 * o;
      if (f < F) out[l * F + f] = z[o];
    }
  }
}
__global__ void k_dbg_keep(int64_t L, int D, rng_key key, uint32_t thr, uint8_t* out, int layer) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < L * D; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = i / D;
    const int d = (int)(i % D);
    const u32x4 r = philox4x32_10((uint32_t)(l >> 2), (uint32_t)d | ((uint32_t)layer << 16), key.s0, key.s1, key.k0, key.k1);
    out[i] = (thr == 0 || pick4(r, (int)(l & 3)) >= thr) ? 1 : 0;
  }
}

extern "C" int dccf_debug_candidates(int64_t N, int32_t S, int64_t item_num, uint64_t seed, uint64_t step, int64_t* out,
                                     void* stream) {
  ARG_CHECK(out && N >= 0 && S >= 0 && item_num > 0, "bad arguments");
  if (N * S == 0) return 0;
  hipLaunchKernelGGL(k_dbg_cand, dim3(512), dim3(256), 0, (hipStream_t)stream, N, S, item_num,
                     make_key(seed, STREAM_CAND, step), out);
  HIP_TRY(hipGetLastError());
  return 0;
}
extern "C" int dccf_debug_noise(int64_t L, int32_t F, float std, uint64_t seed, uint64_t step, float* out, void* stream) {
  ARG_CHECK(out && L >= 0 && F > 0, "bad arguments");
  if (L == 0) return 0;
  hipLaunchKernelGGL(k_dbg_noise, dim3(1024), dim3(256), 0, (hipStream_t)stream, L, F, make_key(seed, STREAM_NOISE, step),
                     -2.0f * 0.69314718055994530942f * std * std, out);
  HIP_TRY(hipGetLastError());
  return 0;
}
extern "C" int dccf_debug_keep_layer(int64_t L, int32_t D, float dropout, uint64_t seed, uint64_t step, int32_t layer,
                                     uint8_t* out, void* stream) {
  ARG_CHECK(out && L >= 0 && D > 0 && dropout >= 0.f && dropout < 1.f && layer >= 0 && layer <= DCCF_MAX_EXTRA, "bad arguments");
  if (L == 0) return 0;
  hipLaunchKernelGGL(k_dbg_keep, dim3(1024), dim3(256), 0, (hipStream_t)stream, L, D, make_key(seed, STREAM_DROP, step),
                     dropout > 0.f ? drop_threshold(dropout) : 0u, out, layer);
  HIP_TRY(hipGetLastError());
  return 0;
}
extern "C" int dccf_debug_keep(int64_t L, int32_t D, float dropout, uint64_t seed, uint64_t step, uint8_t* out,
                               void* stream) {
  return dccf_debug_keep_layer(L, D, dropout, seed, step, 0, out, stream);
}
