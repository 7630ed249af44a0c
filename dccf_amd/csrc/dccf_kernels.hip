// dccf_kernels.hip — DCCF.predict / DCCF.forward+backward as hand-written gfx950 kernels.
//
// Reference semantics: src/models/DCCF.py:66-127.  With N rows of X, S1 = S+1 candidates per row and
// A noise draws per candidate, the reference materialises L = N*S1*A rows x[l] = [V[cand(l)] ; feat[i0(l)] + eps[l]] of
// width D+F and pushes them through Linear(D+F -> D): z = x W^T + b.  Here the rows are never materialised:
//
//   forward  (k_noise_fwd)  Z = X W^T on the fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32).  A workgroup owns 32 rows l;
//            wave w < NC owns the 128-wide feature chunk w of K and keeps its slice of W_f in registers; its A operand
//            feat + eps is produced in registers (Philox4x32-10 + Box-Muller, in the lane layout the MFMA wants); one more
//            wave owns the item part of K (V[cand] rows, W_i in registers).  Partials meet in LDS; the epilogue adds b,
//            applies relu + dropout, stores h [L,D] and the row dot m[l] = <U[u], h[l]>.
//   epilogue (k_pair_epilogue)  softmax over the candidates of Expo[u, cand] (one lane per candidate), prediction,
//            BPR / MSE loss and d loss / d m.
//   backward (k_bwd_misc)   three role waves per 32 rows: dW_i += dz^T V[cand] (MFMA), dV[cand] += dz W_i (MFMA, rows
//            leave through 128-B float-atomic segments, the A noise copies summed in registers first), and a streaming
//            wave for dU[u] += dm h (row-run reduction before the atomic), db and the dz rows.
//            (k_noise_bwd)  dW_f += dz^T (feat + eps): eps is REGENERATED from the same counters directly as the MFMA
//            B operand; 32x32 accumulators stay in registers over the workgroup's whole row range.
//
// HBM layout (fp32 row-major): U [user_num,D], V [item_num,D], W [D,D+F], b [D], feat [item_num,F], expo [user_num,
// item_num].  Workspace per call (ctx slab): cand int32 [N,S1]; WT [(D+FP),DP] = W transposed, zero padded (DP = D
// rounded to 32/64, FP = F rounded to 128); h [L,DP]; dz [L,DP]; m [L]; dmns [N*S1]; it0 int32 [L] (true item per row).
#include "common.hpp"

// No implicit FMA contraction in this file: a*b+c written as two operations stays two roundings (explicit fmaf() calls
// are still FMAs).  It keeps "fused draws == injected draws" bit for bit and the optimizer in torch's op order.
#pragma clang fp contract(off)

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct Lay {
  int DP, FP, NC, S1, ND, GY;
  int64_t N, L, NS;
  size_t cand, WT, h, dz, m, dmns, it0, total;
};

static Lay make_layout(int64_t N, int D, int F, int S, int A) {
  Lay y;
  y.S1 = S + 1;
  y.N = N;
  y.NS = N * y.S1;
  y.L = y.NS * A;
  y.DP = D <= 32 ? 32 : (int)align_up(D, 64);
  y.FP = (int)align_up(F, 128);
  y.NC = y.FP / 128;
  y.ND = y.DP == 32 ? 1 : 2;
  y.GY = y.DP == 32 ? 1 : y.DP / 64;
  size_t o = 0;
  y.cand = o;  o += align_up((size_t)y.NS * 4, 256);
  y.WT = o;    o += align_up((size_t)(D + y.FP) * y.DP * 4, 256);
  y.h = o;     o += align_up((size_t)y.L * y.DP * 4, 256);
  y.dz = o;    o += align_up((size_t)y.L * y.DP * 4, 256);
  y.m = o;     o += align_up((size_t)y.L * 4, 256);
  y.dmns = o;  o += align_up((size_t)y.NS * 4, 256);
  y.it0 = o;   o += align_up((size_t)y.L * 4, 256);
  y.total = o;
  return y;
}

// ================================================================================================ K0: prep
// WT[k][d] = W[d][k] (zero padded), cand[n][0] = true item, cand[n][s] = injected or Philox candidate
// (models/DCCF.py:72-74), m = 0 (only when two column halves add into it), loss = 0.
__global__ void k_prep(const float* __restrict__ W, float* __restrict__ WT, int D, int F, int DP, int FP,
                       const int64_t* __restrict__ X, const int64_t* __restrict__ sample_item, int* __restrict__ cand,
                       int64_t N, int S, int64_t item_num, int fused, rng_key key, float* __restrict__ m, int64_t Lm,
                       float* __restrict__ loss) {
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
    } else if (i < nWT + NS + Lm) {
      m[i - nWT - NS] = 0.f;
    } else if (loss) {
      loss[0] = 0.f;
    }
  }
}

// ================================================================================================ K1: forward
// Workgroup = NC chunk waves + 1 item wave, one 32-row tile per iteration.
//   chunk wave tq:  lane (row = lane&31, h = lane>>5), k-step (c2, o):  f = 128*tq + (2*c2+h) + 32*o;  ONE Philox call per
//                   4 k-steps yields the 4 normals eps(l, f) of o = 0..3; A = feat[i0(l)][f] + eps;  B = wreg (registers).
//   item wave:      k-step j: A = V[cand(l)][2j+h], B = W_i^T rows 2j+h (registers).
template <int D_, int MODE>   // MODE 0: fused Philox draws, 1: injected noise / keep mask
__global__ __launch_bounds__(512) void k_noise_fwd(const float* __restrict__ WT, const float* __restrict__ bias,
                                                   const float* __restrict__ U, const float* __restrict__ V,
                                                   const float* __restrict__ feat, const int64_t* __restrict__ X,
                                                   const int* __restrict__ cand, const float* __restrict__ noise,
                                                   const uint8_t* __restrict__ keep, float* __restrict__ hbuf,
                                                   float* __restrict__ m, int* __restrict__ it0row, int64_t L, int S1,
                                                   int A, int F, rng_key nkey, rng_key dkey, float nscale,
                                                   uint32_t drop_thr, float kscale) {
  extern __shared__ float zpart[];   // [NW][32][DW]
  constexpr int D = D_;
  constexpr int DP = D <= 32 ? 32 : (D + 63) / 64 * 64;
  constexpr int ND = D <= 32 ? 1 : 2;
  constexpr int DW = ND * 32;
  constexpr int KI = D / 2;          // k-steps of the item part
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6, NC = NW - 1;
  const int h = lane >> 5, c31 = lane & 31;
  const int dbase = blockIdx.y * DW;
  const bool chunk = wave < NC;
  // register-resident B operand: ONE array serves both roles (chunk: [c2][o][nt], item: [j][nt]; KI*ND <= 64*ND)
  float breg[64 * ND];
  if (chunk) {
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2)
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int nt = 0; nt < ND; ++nt)
          breg[(c2 * 4 + o) * ND + nt] = WT[(int64_t)(D + wave * 128 + 2 * c2 + h + 32 * o) * DP + dbase + nt * 32 + c31];
  } else {
#pragma unroll
    for (int j = 0; j < KI; ++j)
#pragma unroll
      for (int nt = 0; nt < ND; ++nt) breg[j * ND + nt] = WT[(int64_t)(2 * j + h) * DP + dbase + nt * 32 + c31];
  }
  const uint32_t rows_per_n = (uint32_t)(S1 * A);
  const int64_t ntiles = (L + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t l = tile * 32 + c31;
    const bool lv = l < L;
    f32x16 acc[ND];
#pragma unroll
    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    if (chunk) {
      const int64_t it0 = lv ? X[2 * (int64_t)((uint32_t)l / rows_per_n) + 1] : 0;
      const float* frow = feat + it0 * F;
#pragma unroll
      for (int c2 = 0; c2 < 16; ++c2) {
        float a[4];
        const int c = 2 * c2 + h;
        if (MODE == 1) {
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            const int f = wave * 128 + c + 32 * o;
            a[o] = (lv && f < F) ? noise[l * F + f] : 0.f;
          }
        } else {
          noise4((uint32_t)l, (uint32_t)(wave * 32 + c), nkey, nscale, a);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int f = wave * 128 + c + 32 * o;
          const float fv = (lv && f < F) ? frow[f] : 0.f;
          a[o] = __fadd_rn(fv, a[o]);               // sample_feature_embeddings = feature + noise (DCCF.py:87); no
                                                    // contraction with Box-Muller's multiply: fused == injected bit for bit
#pragma unroll
          for (int nt = 0; nt < ND; ++nt) acc[nt] = MFMA32(a[o], breg[(c2 * 4 + o) * ND + nt], acc[nt]);
        }
      }
    } else {
      const float* vrow = V + (int64_t)(lv ? cand[(uint32_t)l / (uint32_t)A] : 0) * D;
#pragma unroll
      for (int j = 0; j < KI; ++j) {
        const float a = lv ? vrow[2 * j + h] : 0.f;
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) acc[nt] = MFMA32(a, breg[j * ND + nt], acc[nt]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        zpart[(wave * 32 + row) * DW + nt * 32 + c31] = acc[nt][r];
      }
    __syncthreads();
    for (int row = wave; row < 32; row += NW) {
      const int64_t lr = tile * 32 + row;
      if (lr >= L) break;
      const int64_t ns = lr / A;
      const int64_t n = ns / S1;
      const int64_t u = X[2 * n];
      float part = 0.f;
      for (int d0 = lane; d0 < DW; d0 += 64) {
        const int d = dbase + d0;
        if (d < D) {
          float z = zpart[(NC * 32 + row) * DW + d0];        // item part first, then the feature chunks in order
          for (int w = 0; w < NC; ++w) z += zpart[(w * 32 + row) * DW + d0];
          z += bias[d];
          bool kept = true;
          if (MODE == 1) {
            if (keep) kept = keep[lr * D + d] != 0;
          } else if (drop_thr) {
            const u32x4 r4 = philox4x32_10((uint32_t)lr, (uint32_t)(d >> 2), dkey.s0, dkey.s1, dkey.k0, dkey.k1);
            kept = pick4(r4, d & 3) >= drop_thr;
          }
          const float hv = (z > 0.f && kept) ? z * kscale : 0.f;
          hbuf[lr * DP + d] = hv;
          part = fmaf(U[u * D + d], hv, part);
        }
      }
      part = wave_sum(part);
      if (lane == 0) {
        if (gridDim.y == 1) m[lr] = part;
        else atomicAdd(&m[lr], part);
        if (blockIdx.y == 0) it0row[lr] = (int)X[2 * n + 1];
      }
    }
    __syncthreads();
  }
}

// ================================================================================================ K2: pair epilogue
// One lane per candidate: softmax_s(Expo[u, cand[n,s]]) (DCCF.py:98), prediction = mean_a sum_s w m (DCCF.py:100),
// loss and d loss / d m (DCCF.py:116-125).  A group of GS lanes serves one pair (rank 1) or one row.
__device__ __forceinline__ float expo_at(const dccf_model_t& M, int64_t u, int64_t i) {
  if (M.expo) return M.expo[u * M.item_num + i];
  float acc = 0.f;
  for (int k = 0; k < M.ipsD; ++k) acc = fmaf(M.ipsP[u * M.ipsD + k], M.ipsQ[i * M.ipsD + k], acc);
  acc = acc + M.ipsBu[u] + M.ipsBi[i] + M.ipsB0;
  return acc / fmaxf(M.ipsProp[i], M.ipsM);
}

template <int GS>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int GS>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

template <int GS>
__device__ __forceinline__ float row_predict(const dccf_model_t& M, const int64_t* X, const int* cand, const float* m,
                                             int64_t n, int s, int S1, int A, float& w_over_A) {
  const bool valid = s < S1;
  const int64_t u = X[2 * n];
  const float e = valid ? expo_at(M, u, cand[n * S1 + s]) : -INFINITY;
  const float mx = group_max<GS>(e);
  const float ex = valid ? expf(e - mx) : 0.f;
  const float w = ex / group_sum<GS>(ex);
  float tot = 0.f;
  for (int a = 0; a < A; ++a) tot += group_sum<GS>(valid ? w * m[(n * S1 + s) * A + a] : 0.f);
  w_over_A = w / (float)A;
  return tot / (float)A;
}

template <int GS>
__global__ __launch_bounds__(256) void k_pair_epilogue(dccf_model_t M, const int64_t* __restrict__ X,
                                                       const float* __restrict__ Y, const int* __restrict__ cand,
                                                       const float* __restrict__ m, float* __restrict__ dmns,
                                                       float* __restrict__ pred, float* __restrict__ loss, int64_t N,
                                                       int rank, int train) {
  const int S1 = M.S + 1, A = M.A;
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
      const float pp = row_predict<GS>(M, X, cand, m, k, s, S1, A, wp);
      const float pn = row_predict<GS>(M, X, cand, m, B + k, s, S1, A, wn);
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
      const float p = row_predict<GS>(M, X, cand, m, k, s, S1, A, w);
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

// ================================================================================================ K3: backward, small terms
// dz[l][d] = dm[l] * U[u(l)][d] * [h[l][d] > 0] * kscale.  Three role waves share one 32-row tile per iteration:
//   wave 0  gW[:, 0:D] += dz^T V[cand]   (MFMA: M = d, N = d', K = rows; accumulators live over the block's tiles)
//   wave 1  gV[cand] += dz W_i           (MFMA: M = rows, N = d', K = d; W_i in registers; the A noise copies of a
//                                          candidate are summed in registers, rows leave as 128-B atomic segments)
//   wave 2  gU[u] += sum dm h (run-length reduction over the rows of one batch row), gb += dz, dz rows -> workspace
template <int D_>
__global__ __launch_bounds__(192) void k_bwd_misc(const float* __restrict__ W, const float* __restrict__ U,
                                                  const float* __restrict__ V, const int64_t* __restrict__ X,
                                                  const int* __restrict__ cand, const float* __restrict__ dmns,
                                                  const float* __restrict__ hbuf, float* __restrict__ dzbuf,
                                                  float* __restrict__ gU, float* __restrict__ gV,
                                                  float* __restrict__ gW, float* __restrict__ gb, int64_t L, int S1, int A,
                                                  int F, float kscale) {
  constexpr int D = D_;
  constexpr int DP = D <= 32 ? 32 : (D + 63) / 64 * 64;
  constexpr int ND = D <= 32 ? 1 : 2;     // 32-wide tiles of this block's column half
  constexpr int DW = ND * 32;
  constexpr int NT = (D + 31) / 32;       // 32-wide tiles over all of D
  constexpr int KD = D / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int dbase = blockIdx.y * DW;
  const uint32_t rows_per_n = (uint32_t)(S1 * A);
  const int64_t ntiles = (L + 31) / 32;
  if (wave == 0) {
    f32x16 acc[ND][NT];
#pragma unroll
    for (int mt = 0; mt < ND; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll 2
      for (int j = 0; j < 16; ++j) {
        const int64_t l = tile * 32 + 2 * j + h;
        const bool lv = l < L;
        const uint32_t ns = (uint32_t)l / (uint32_t)A;
        const float dmv = lv ? dmns[ns] * kscale : 0.f;
        const int64_t u = lv ? X[2 * (int64_t)((uint32_t)l / rows_per_n)] : 0;
        const float* vrow = V + (int64_t)(lv ? cand[ns] : 0) * D;
        float a[ND], b[NT];
#pragma unroll
        for (int mt = 0; mt < ND; ++mt) {
          const int d = dbase + mt * 32 + c31;
          const bool ok = lv && d < D;
          const float hv = ok ? hbuf[l * DP + d] : 0.f;
          const float uv = ok ? U[u * D + d] : 0.f;
          a[mt] = hv > 0.f ? dmv * uv : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = (lv && nt * 32 + c31 < D) ? vrow[nt * 32 + c31] : 0.f;
#pragma unroll
        for (int mt = 0; mt < ND; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA32(a[mt], b[nt], acc[mt][nt]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < ND; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = dbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int k = nt * 32 + c31;
          if (d < D && k < D) atomicAdd(&gW[(int64_t)d * (D + F) + k], acc[mt][nt][r]);
        }
  } else if (wave == 1) {
    float wd[KD][ND];                         // W_i rows 2j+h, this block's column half
#pragma unroll
    for (int j = 0; j < KD; ++j)
#pragma unroll
      for (int nt = 0; nt < ND; ++nt) {
        const int dd = dbase + nt * 32 + c31;
        wd[j][nt] = dd < D ? W[(int64_t)(2 * j + h) * (D + F) + dd] : 0.f;
      }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t l = tile * 32 + c31;
      const bool lv = l < L;
      const float dmv = lv ? dmns[(uint32_t)l / (uint32_t)A] * kscale : 0.f;
      const float* urow = U + (lv ? X[2 * (int64_t)((uint32_t)l / rows_per_n)] : 0) * D;
      const float* hrow = hbuf + l * DP;
      f32x16 acc[ND];
#pragma unroll
      for (int nt = 0; nt < ND; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
      for (int j = 0; j < KD; ++j) {
        const int d = 2 * j + h;
        const float hv = lv ? hrow[d] : 0.f;
        const float uv = lv ? urow[d] : 0.f;
        const float a = hv > 0.f ? dmv * uv : 0.f;
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) acc[nt] = MFMA32(a, wd[j][nt], acc[nt]);
      }
      // rows of the tile -> gV[cand]; with A == 2 rows (2q, 2q+1) are one candidate and sit in adjacent registers
#pragma unroll
      for (int nt = 0; nt < ND; ++nt) {
        const int dd = dbase + nt * 32 + c31;
        if (A == 2) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int64_t lr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (lr < L && dd < D) atomicAdd(&gV[(int64_t)cand[lr >> 1] * D + dd], acc[nt][r] + acc[nt][r + 1]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t lr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (lr < L && dd < D) atomicAdd(&gV[(int64_t)cand[(uint32_t)lr / (uint32_t)A] * D + dd], acc[nt][r]);
          }
        }
      }
    }
  } else {
    float gb_acc = 0.f;
    const int d = dbase + lane;
    const bool dv = lane < DW && d < D;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int64_t cur_n = -1, u = 0;
      float ud = 0.f, due = 0.f;
      for (int row = 0; row < 32; ++row) {
        const int64_t l = tile * 32 + row;
        if (l >= L) break;
        const int64_t n = (uint32_t)l / rows_per_n;
        if (n != cur_n) {
          if (cur_n >= 0 && dv) atomicAdd(&gU[u * D + d], due);
          cur_n = n;
          u = X[2 * n];
          ud = dv ? U[u * D + d] : 0.f;
          due = 0.f;
        }
        const float dmv = dmns[(uint32_t)l / (uint32_t)A];
        const float hv = dv ? hbuf[l * DP + d] : 0.f;
        due = fmaf(dmv, hv, due);
        const float dzv = hv > 0.f ? dmv * kscale * ud : 0.f;
        if (lane < DW) dzbuf[l * DP + d] = dzv;
        gb_acc += dzv;
      }
      if (cur_n >= 0 && dv) atomicAdd(&gU[u * D + d], due);
    }
    if (dv) atomicAdd(&gb[d], gb_acc);
  }
}

// ================================================================================================ K4: backward, dW_f
// gW[:, D:] += dz^T (feat + eps) over the rows.  Wave task = 32 rows x one 128-wide f chunk x ND*32 rows d of gW; the
// B operand is feat[i0(l)][f] + eps(l, f) with eps regenerated by ONE Philox call per k-step (its 4 normals are the 4
// N-tiles).  Accumulators (ND*4 tiles of 32x32) stay in registers over the block's row range and leave through one
// float-atomic pass shaped as two 128-B row segments per wave instruction.
template <int ND, int MODE>
__global__ __launch_bounds__(512) void k_noise_bwd(const float* __restrict__ dzbuf, const float* __restrict__ noise,
                                                   const float* __restrict__ feat, const int* __restrict__ it0row,
                                                   float* __restrict__ gW, int64_t L, int D, int F, int DP, rng_key nkey,
                                                   float nscale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int dbase = blockIdx.y * ND * 32;
  f32x16 acc[ND][4];
#pragma unroll
  for (int mt = 0; mt < ND; ++mt)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][o][r] = 0.f;
  const int64_t ntiles = (L + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll 2
    for (int j = 0; j < 16; ++j) {
      const int64_t l = tile * 32 + 2 * j + h;
      const bool lv = l < L;
      float a[ND], bq[4];
#pragma unroll
      for (int mt = 0; mt < ND; ++mt) a[mt] = lv ? dzbuf[l * DP + dbase + mt * 32 + c31] : 0.f;
      const float* frow = feat + (int64_t)(lv ? it0row[l] : 0) * F;
      if (MODE == 0) {
        noise4((uint32_t)l, (uint32_t)(wave * 32 + c31), nkey, nscale, bq);
      } else {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int f = wave * 128 + 32 * o + c31;
          bq[o] = (lv && f < F) ? noise[l * F + f] : 0.f;
        }
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int f = wave * 128 + 32 * o + c31;
        bq[o] = __fadd_rn((lv && f < F) ? frow[f] : 0.f, bq[o]);
      }
#pragma unroll
      for (int mt = 0; mt < ND; ++mt)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[mt][o] = MFMA32(a[mt], bq[o], acc[mt][o]);
    }
  }
#pragma unroll
  for (int mt = 0; mt < ND; ++mt)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int f = wave * 128 + 32 * o + c31;
        if (d < D && f < F) atomicAdd(&gW[(int64_t)d * (D + F) + D + f], acc[mt][o][r]);
      }
}

// ================================================================================================ host side
static int check_model(const dccf_model_t* M) {
  ARG_CHECK(M != nullptr, "model is NULL");
  ARG_CHECK(M->D == 16 || M->D == 32 || M->D == 64 || M->D == 128, "D must be 16, 32, 64 or 128");
  ARG_CHECK(M->F >= 1 && M->F <= 896, "F must be in [1, 896] (7 feature-chunk waves + 1 item wave per workgroup)");
  ARG_CHECK(M->S >= 0 && M->S <= 63 && M->A >= 1 && M->A <= 64, "S in [0,63], A in [1,64]");
  ARG_CHECK(M->user_num > 0 && M->item_num > 0 && M->item_num < 2147483647LL, "bad user_num / item_num");
  ARG_CHECK(M->U && M->V && M->W && M->b && M->feat, "NULL parameter / feature pointer");
  ARG_CHECK(M->expo || (M->ipsP && M->ipsQ && M->ipsBu && M->ipsBi && M->ipsProp && M->ipsD > 0),
            "need expo or the IPS factors");
  return 0;
}

#define BY_D(D, CALL)              \
  switch (D) {                     \
    case 16: { CALL(16); } break;  \
    case 32: { CALL(32); } break;  \
    case 64: { CALL(64); } break;  \
    default: { CALL(128); } break; \
  }

static int run_dccf(dccf_ctx* ctx, const dccf_model_t* M, const dccf_rand_t* rnd, const int64_t* X, const float* Y,
                    int64_t N, int rank, float dropout, const dccf_grads_t* G, float* pred, float* loss, bool train,
                    hipStream_t st) {
  ARG_CHECK(ctx != nullptr, "ctx is NULL");
  if (int e = check_model(M)) return e;
  ARG_CHECK(N >= 0 && N * (int64_t)(M->S + 1) * M->A < 4294967296LL, "N*(S+1)*A must be < 2^32");
  ARG_CHECK(rnd != nullptr && (N == 0 || (X != nullptr && pred != nullptr)), "NULL rnd / X / prediction");
  ARG_CHECK(dropout >= 0.f && dropout < 1.f, "dropout must be in [0,1)");
  const bool fused = rnd->mode == 1;
  ARG_CHECK(rnd->mode == 0 || rnd->mode == 1, "rnd.mode must be 0 or 1");
  if (!fused && N > 0) ARG_CHECK((M->S == 0 || rnd->sample_item) && rnd->noise, "injected mode needs sample_item and noise");
  if (train) {
    ARG_CHECK(G && G->gU && G->gV && G->gW && G->gb && loss, "NULL gradient / loss pointer");
    ARG_CHECK(rank == 0 || rank == 1, "rank must be 0 or 1");
    if (rank == 1) ARG_CHECK(N % 2 == 0, "rank==1 needs [positives ; negatives] (even N)");
    if (rank == 0) ARG_CHECK(Y != nullptr || N == 0, "rank==0 needs Y");
  }
  if (N == 0) {
    if (train) HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), st));
    return 0;
  }
  const int D = M->D, F = M->F, S1 = M->S + 1, A = M->A;
  const Lay y = make_layout(N, D, F, M->S, A);
  if (int e = dccf_ws_ensure(ctx, y.total)) return e;
  char* ws = ctx->ws;
  int* cand = (int*)(ws + y.cand);
  float* WT = (float*)(ws + y.WT);
  float* hbuf = (float*)(ws + y.h);
  float* dzbuf = (float*)(ws + y.dz);
  float* m = (float*)(ws + y.m);
  float* dmns = (float*)(ws + y.dmns);
  int* it0row = (int*)(ws + y.it0);

  const rng_key ckey = make_key(rnd->seed, STREAM_CAND, rnd->step);
  const rng_key nkey = make_key(rnd->seed, STREAM_NOISE, rnd->step);
  const rng_key dkey = make_key(rnd->seed, STREAM_DROP, rnd->step);
  const float nscale = -2.0f * 0.69314718055994530942f * M->std * M->std;
  const float kscale = dropout > 0.f ? 1.0f / (float)(1.0 - (double)dropout) : 1.0f;
  const uint32_t thr = dropout > 0.f ? drop_threshold(dropout) : 0u;
  const int64_t ntiles = (y.L + 31) / 32;

  {
    const int64_t total = (int64_t)(D + y.FP) * y.DP + y.NS + (y.GY > 1 ? y.L : 0) + 1;
    const int grid = (int)min((int64_t)2048, (total + 255) / 256);
    prof_begin(ctx, st);
    hipLaunchKernelGGL(k_prep, dim3(grid), dim3(256), 0, st, M->W, WT, D, F, y.DP, y.FP, X, rnd->sample_item, cand, N,
                       M->S, M->item_num, fused ? 1 : 0, ckey, m, y.GY > 1 ? y.L : (int64_t)0, train ? loss : nullptr);
    prof_end(ctx, 0, st);
  }
  {
    const dim3 grid((unsigned)min((int64_t)1024, ntiles), y.GY), block(64 * (y.NC + 1));
    const size_t smem = (size_t)(y.NC + 1) * 32 * y.ND * 32 * 4;
    prof_begin(ctx, st);
#define LAUNCH_FWD(D_)                                                                                               \
  if (fused)                                                                                                         \
    hipLaunchKernelGGL((k_noise_fwd<D_, 0>), grid, block, smem, st, WT, M->b, M->U, M->V, M->feat, X, cand, rnd->noise, \
                       rnd->keep, hbuf, m, it0row, y.L, S1, A, F, nkey, dkey, nscale, thr, kscale);                  \
  else                                                                                                               \
    hipLaunchKernelGGL((k_noise_fwd<D_, 1>), grid, block, smem, st, WT, M->b, M->U, M->V, M->feat, X, cand, rnd->noise, \
                       rnd->keep, hbuf, m, it0row, y.L, S1, A, F, nkey, dkey, nscale, thr, kscale);
    BY_D(D, LAUNCH_FWD)
#undef LAUNCH_FWD
    prof_end(ctx, 2, st);
  }
  {
    const int64_t units = (train && rank == 1) ? N / 2 : N;
    const int GS = S1 <= 16 ? 16 : (S1 <= 32 ? 32 : 64);
    const int grid = (int)min((int64_t)2048, (units * GS + 255) / 256);
    prof_begin(ctx, st);
#define LAUNCH_PE(GS_)                                                                                              \
  hipLaunchKernelGGL((k_pair_epilogue<GS_>), dim3(grid), dim3(256), 0, st, *M, X, Y, cand, m, dmns, pred, loss, N, rank, \
                     train ? 1 : 0)
    if (GS == 16) LAUNCH_PE(16); else if (GS == 32) LAUNCH_PE(32); else LAUNCH_PE(64);
#undef LAUNCH_PE
    prof_end(ctx, 3, st);
  }
  if (train) {
    {
      const dim3 grid((unsigned)min((int64_t)512, ntiles), y.GY);
      prof_begin(ctx, st);
#define LAUNCH_MISC(D_)                                                                                             \
  hipLaunchKernelGGL((k_bwd_misc<D_>), grid, dim3(192), 0, st, M->W, M->U, M->V, X, cand, dmns, hbuf, dzbuf, G->gU, G->gV, \
                     G->gW, G->gb, y.L, S1, A, F, kscale);
      BY_D(D, LAUNCH_MISC)
#undef LAUNCH_MISC
      prof_end(ctx, 4, st);
    }
    {
      // row-splits of the reduction: every block ends with ND*4*16*64*4 B of float atomics per wave, so few blocks
      // with >= 2 tiles each when the batch is small, one block per CU when it is large
      const int64_t gx = min((int64_t)256, max(min(ntiles, (int64_t)64), ntiles / 2));
      const dim3 grid((unsigned)gx, y.GY), block(64 * y.NC);
      prof_begin(ctx, st);
#define LAUNCH_BWD(ND_, MODE_) \
  hipLaunchKernelGGL((k_noise_bwd<ND_, MODE_>), grid, block, 0, st, dzbuf, rnd->noise, M->feat, it0row, G->gW, y.L, D, F, y.DP, nkey, nscale)
      if (y.ND == 1) { if (fused) LAUNCH_BWD(1, 0); else LAUNCH_BWD(1, 1); }
      else           { if (fused) LAUNCH_BWD(2, 0); else LAUNCH_BWD(2, 1); }
#undef LAUNCH_BWD
      prof_end(ctx, 5, st);
    }
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

extern "C" int dccf_ctx_reserve(dccf_ctx* ctx, int64_t max_rows, int32_t D, int32_t F, int32_t S, int32_t A) {
  ARG_CHECK(ctx != nullptr && max_rows >= 0 && D > 0 && F > 0 && S >= 0 && A > 0, "bad reserve arguments");
  const Lay y = make_layout(max_rows, D, F, S, A);
  return dccf_ws_ensure(ctx, y.total);
}

// ================================================================================================ debug: workspace
// Copies one workspace array of the LAST call with these shapes to `dst` (device) — for the parity tests only.
// which: 0 cand(int32 [N*S1]) 1 WT 3 h [L,DP] 4 m [L] 5 dmns [N*S1] 6 dz [L,DP] 7 it0(int32 [L]); info = DP, FP, count, 4
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
    case 6: off = y.dz; cnt = (size_t)y.L * y.DP; break;
    case 7: off = y.it0; cnt = (size_t)y.L; break;
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
__global__ void k_dbg_keep(int64_t L, int D, rng_key key, uint32_t thr, uint8_t* out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < L * D; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = i / D;
    const int d = (int)(i % D);
    const u32x4 r = philox4x32_10((uint32_t)l, (uint32_t)(d >> 2), key.s0, key.s1, key.k0, key.k1);
    out[i] = (thr == 0 || pick4(r, d & 3) >= thr) ? 1 : 0;
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
extern "C" int dccf_debug_keep(int64_t L, int32_t D, float dropout, uint64_t seed, uint64_t step, uint8_t* out,
                               void* stream) {
  ARG_CHECK(out && L >= 0 && D > 0 && dropout >= 0.f && dropout < 1.f, "bad arguments");
  if (L == 0) return 0;
  hipLaunchKernelGGL(k_dbg_keep, dim3(1024), dim3(256), 0, (hipStream_t)stream, L, D, make_key(seed, STREAM_DROP, step),
                     dropout > 0.f ? drop_threshold(dropout) : 0u, out);
  HIP_TRY(hipGetLastError());
  return 0;
}
