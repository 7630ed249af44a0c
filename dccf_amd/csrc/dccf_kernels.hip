// dccf_kernels.hip — DCCF.predict / DCCF.forward+backward as hand-written gfx950 kernels.
//
// Reference semantics: src/models/DCCF.py:66-127.  With N rows of X, S1 = S+1 candidates per row and
// A noise draws per candidate, the reference materialises L = N*S1*A rows of width D+F and pushes them through
// Linear(D+F -> D).  Here the layer is split by linearity,
//     z[l] = W_i V[cand(l)] + (W_f feat[i0(l)] + b)  +  W_f eps[l]
//            \_________ "base": once per (n,s) ________/    \_ the only term that is different for every l _/
// and only the last term is a real GEMM ([L,F] x [F,D]).  It runs on the fp32 MFMA (v_mfma_f32_32x32x2_f32, exact
// fp32) with the A operand — the Gaussian noise — generated in registers by Philox4x32-10 + Box-Muller in exactly
// the lane layout the MFMA wants, so the [L,F] noise tensor never exists in memory.  The backward needs the same
// eps for dW_f = dz^T eps; it is regenerated from the same counters (again directly as an MFMA operand).
//
// HBM data layout (all fp32 row-major): U [user_num,D], V [item_num,D], W [D,D+F], b [D], feat [item_num,F],
// expo [user_num,item_num].  Workspace (per call, in the ctx slab): cand int32 [N,S1]; WT [(D+FP),DP] = W transposed
// and zero padded (DP = D rounded to 32, FP = F rounded to 128); base [N*S1,DP]; h [L,DP] (overwritten by dz in the
// backward); m [L]; dmns [N*S1]; dzn [N,DP].
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Lay {
  int DP, FP, NC, S1, ND, GY;
  int64_t N, L, NS;
  size_t cand, WT, base, h, m, dmns, dzn, total;
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
  y.base = o;  o += align_up((size_t)y.NS * y.DP * 4, 256);
  y.h = o;     o += align_up((size_t)y.L * y.DP * 4, 256);
  y.m = o;     o += align_up((size_t)y.L * 4, 256);
  y.dmns = o;  o += align_up((size_t)y.NS * 4, 256);
  y.dzn = o;   o += align_up((size_t)N * y.DP * 4, 256);
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

// ================================================================================================ K1: base
// base[(n,s)][d] = sum_k W_i[d][k] V[cand[n,s]][k] + ( b[d] + sum_f W_f[d][f] feat[i0(n)][f] )
// One block = RB batch rows; the K range is split over the 256/DP thread groups and reduced through LDS.
#define BASE_RB 4
__global__ __launch_bounds__(256) void k_base(const float* __restrict__ WT, const float* __restrict__ bias,
                                              const float* __restrict__ V, const float* __restrict__ feat,
                                              const int64_t* __restrict__ X, const int* __restrict__ cand,
                                              float* __restrict__ base, int64_t N, int S1, int D, int F, int DP) {
  extern __shared__ float sm[];
  float* xf = sm;                               // [RB][F]
  float* xv = xf + BASE_RB * F;                 // [RB*S1][D]
  float* gp = xv + BASE_RB * S1 * D;            // [G][RB][DP]  partial G sums
  const int G = 256 / DP;                       // thread groups
  const int grp = threadIdx.x / DP, d = threadIdx.x % DP;
  const int64_t ngroups = (N + BASE_RB - 1) / BASE_RB;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t n0 = g * BASE_RB;
    const int nr = (int)min((int64_t)BASE_RB, N - n0);
    for (int i = threadIdx.x; i < nr * F; i += 256) {
      const int r = i / F, f = i % F;
      xf[i] = feat[X[2 * (n0 + r) + 1] * F + f];
    }
    for (int i = threadIdx.x; i < nr * S1 * D; i += 256) {
      const int j = i / D, k = i % D;
      xv[i] = V[(int64_t)cand[n0 * S1 + j] * D + k];
    }
    __syncthreads();
    {   // G part: this group's slice of f
      float acc[BASE_RB];
#pragma unroll
      for (int r = 0; r < BASE_RB; ++r) acc[r] = 0.f;
      const int f0 = (int)((int64_t)F * grp / G), f1 = (int)((int64_t)F * (grp + 1) / G);
      for (int f = f0; f < f1; ++f) {
        const float w = WT[(int64_t)(D + f) * DP + d];
#pragma unroll
        for (int r = 0; r < BASE_RB; ++r) acc[r] = fmaf(w, xf[r * F + f], acc[r]);
      }
#pragma unroll
      for (int r = 0; r < BASE_RB; ++r) gp[(grp * BASE_RB + r) * DP + d] = acc[r];
    }
    __syncthreads();
    // item part + sum of the G partials (fixed order -> deterministic)
    for (int j = grp; j < nr * S1; j += G) {
      const int r = j / S1;
      float acc = bias[d < D ? d : 0];
      for (int q = 0; q < G; ++q) acc += gp[(q * BASE_RB + r) * DP + d];
      float e = 0.f;
      for (int k = 0; k < D; ++k) e = fmaf(WT[(int64_t)k * DP + d], xv[j * D + k], e);
      base[(n0 * S1 + j) * DP + d] = d < D ? acc + e : 0.f;
    }
    __syncthreads();
  }
}

// ================================================================================================ K2: noise forward
// Wave task = 32 rows l  x  one 128-wide f chunk (tq = wave)  x  ND*32 columns d.  The W_f chunk stays in registers for
// the life of the block (wreg, 64*ND VGPRs); per 4 MFMA k-steps a lane makes ONE Philox call whose 4 normals are the
// A operands of those steps:  lane (row = lane&31, h = lane>>5), step (c2, o):  f = 128*tq + (2*c2+h) + 32*o.
// The NC chunk partials meet in LDS; the epilogue adds base, applies relu + dropout, stores h and the row dot m[l].
template <int ND, int MODE>   // MODE 0: fused Philox, 1: injected noise
__global__ __launch_bounds__(512) void k_noise_fwd(const float* __restrict__ WT, const float* __restrict__ base,
                                                   const float* __restrict__ U, const int64_t* __restrict__ X,
                                                   const float* __restrict__ noise, const uint8_t* __restrict__ keep,
                                                   float* __restrict__ hbuf, float* __restrict__ m, int64_t L, int S1,
                                                   int A, int D, int F, int DP, rng_key nkey, rng_key dkey,
                                                   float nscale, uint32_t drop_thr, float kscale) {
  extern __shared__ float zpart[];   // [NC][32][DW]
  constexpr int DW = ND * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NC = blockDim.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int dbase = blockIdx.y * DW;
  float wreg[16][4][ND];
#pragma unroll
  for (int c2 = 0; c2 < 16; ++c2)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int nt = 0; nt < ND; ++nt)
        wreg[c2][o][nt] = WT[(int64_t)(D + wave * 128 + 2 * c2 + h + 32 * o) * DP + dbase + nt * 32 + c31];

  const int64_t ntiles = (L + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t l = tile * 32 + c31;
    f32x16 acc[ND];
#pragma unroll
    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2) {
      float a[4];
      const int c = 2 * c2 + h;
      if (MODE == 1) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int f = wave * 128 + c + 32 * o;
          a[o] = (l < L && f < F) ? noise[l * F + f] : 0.f;
        }
      } else {
        noise4((uint32_t)l, (uint32_t)(wave * 32 + c), nkey, nscale, a);
      }
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int nt = 0; nt < ND; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[o], wreg[c2][o][nt], acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < ND; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        zpart[(wave * 32 + row) * DW + nt * 32 + c31] = acc[nt][r];
      }
    __syncthreads();
    for (int row = wave; row < 32; row += NC) {
      const int64_t lr = tile * 32 + row;
      if (lr >= L) break;
      const int64_t ns = lr / A;
      const int64_t n = ns / S1;
      const int64_t u = X[2 * n];
      float part = 0.f;
      for (int d0 = lane; d0 < DW; d0 += 64) {
        const int d = dbase + d0;
        if (d < D) {
          float z = 0.f;
          for (int w = 0; w < NC; ++w) z += zpart[(w * 32 + row) * DW + d0];
          z += base[ns * DP + d];
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
      }
    }
    __syncthreads();
  }
}

// ================================================================================================ K3: pair epilogue
// softmax over the S1 candidates of Expo[u, cand] (models/DCCF.py:98), prediction (DCCF.py:100), BPR / MSE loss and
// d loss / d m  (DCCF.py:116-125).  One thread per pair (rank 1) or per row (rank 0 / predict).
__device__ __forceinline__ float expo_at(const dccf_model_t& M, int64_t u, int64_t i) {
  if (M.expo) return M.expo[u * M.item_num + i];
  float acc = 0.f;
  for (int k = 0; k < M.ipsD; ++k) acc = fmaf(M.ipsP[u * M.ipsD + k], M.ipsQ[i * M.ipsD + k], acc);
  acc = acc + M.ipsBu[u] + M.ipsBi[i] + M.ipsB0;
  return acc / fmaxf(M.ipsProp[i], M.ipsM);
}

__device__ float row_predict(const dccf_model_t& M, const int64_t* X, const int* cand, const float* m, float* dmns,
                             int64_t n, int S1, int A, bool train) {
  const int64_t u = X[2 * n];
  float mx = -INFINITY;
  for (int s = 0; s < S1; ++s) mx = fmaxf(mx, expo_at(M, u, cand[n * S1 + s]));
  float den = 0.f;
  for (int s = 0; s < S1; ++s) den += expf(expo_at(M, u, cand[n * S1 + s]) - mx);
  float tot = 0.f;
  for (int a = 0; a < A; ++a) {
    float pa = 0.f;
    for (int s = 0; s < S1; ++s) {
      const float w = expf(expo_at(M, u, cand[n * S1 + s]) - mx) / den;
      pa = fmaf(w, m[(n * S1 + s) * A + a], pa);
      if (train && a == 0) dmns[n * S1 + s] = w / (float)A;
    }
    tot += pa;
  }
  return tot / (float)A;
}

__global__ __launch_bounds__(256) void k_pair_epilogue(dccf_model_t M, const int64_t* __restrict__ X,
                                                       const float* __restrict__ Y, const int* __restrict__ cand,
                                                       const float* __restrict__ m, float* __restrict__ dmns,
                                                       float* __restrict__ pred, float* __restrict__ loss, int64_t N,
                                                       int rank, int train) {
  const int S1 = M.S + 1, A = M.A;
  const int64_t units = (train && rank == 1) ? N / 2 : N;
  float lsum = 0.f;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < units; k += (int64_t)gridDim.x * blockDim.x) {
    if (train && rank == 1) {
      const int64_t B = N / 2;
      const float pp = row_predict(M, X, cand, m, dmns, k, S1, A, true);
      const float pn = row_predict(M, X, cand, m, dmns, B + k, S1, A, true);
      pred[k] = pp;
      pred[B + k] = pn;
      const float d = pp - pn;
      const float sg = 1.f / (1.f + expf(-d));
      lsum += -logf(sg);
      const float gp = -(1.f - sg);
      for (int s = 0; s < S1; ++s) {
        dmns[k * S1 + s] *= gp;
        dmns[(B + k) * S1 + s] *= -gp;
      }
    } else {
      const float p = row_predict(M, X, cand, m, dmns, k, S1, A, train != 0);
      pred[k] = p;
      if (train) {
        const float diff = p - Y[k];
        lsum += diff * diff / (float)N;
        const float gp = 2.f * diff / (float)N;
        for (int s = 0; s < S1; ++s) dmns[k * S1 + s] *= gp;
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

// ================================================================================================ K4: small backward
// Per batch row n (thread = (row r, column d)):  dz = dm * U[u] * [h > 0] * kscale  (written over h for K5),
// dU[u] += sum_l dm h,  dzs[(n,s)] = sum_a dz,  dzn[n] = sum_s dzs.  Then per candidate row: dV[cand] += W_i^T dzs,
// staged in LDS where rows of the block that hit the same item are summed first (one atomic row-add per distinct
// item); gW_i += dzs (x) V[cand] and gb += dz accumulate in registers over the block's loop.
__global__ __launch_bounds__(256) void k_bwd_small(const float* __restrict__ W, const float* __restrict__ U,
                                                   const float* __restrict__ V, const int64_t* __restrict__ X,
                                                   const int* __restrict__ cand, const float* __restrict__ dmns,
                                                   float* __restrict__ hbuf, float* __restrict__ dzn,
                                                   float* __restrict__ gU, float* __restrict__ gV,
                                                   float* __restrict__ gW, float* __restrict__ gb, int64_t N, int S1,
                                                   int A, int D, int F, int DP, float kscale) {
  extern __shared__ float sm[];
  const int RB = 256 / DP;
  float* dzs = sm;                          // [RB*S1][DP]
  float* stage = dzs + RB * S1 * DP;        // [RB*S1][DP]  dV rows
  int* first = (int*)(stage + RB * S1 * DP);   // [RB*S1]
  int* items = first + RB * S1;             // [RB*S1]
  const int r = threadIdx.x / DP, d = threadIdx.x % DP;
  const int DR = D / RB > 0 ? D / RB : 1;   // rows of gW_i this thread owns: d in [r*DR, r*DR+DR)
  float gwi[64];                            // DR <= 64 (D=128, RB=2)
#pragma unroll
  for (int q = 0; q < 64; ++q) gwi[q] = 0.f;
  float gb_acc = 0.f;
  const int64_t ngroups = (N + RB - 1) / RB;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t n = g * RB + r;
    const bool valid = n < N && d < D;
    const int nrows = (int)min((int64_t)RB, N - g * RB) * S1;
    if (n < N) {
      const int64_t u = X[2 * n];
      const float ud = valid ? U[u * D + d] : 0.f;
      float due = 0.f, dzn_acc = 0.f;
      for (int s = 0; s < S1; ++s) {
        const float dmv = dmns[n * S1 + s];
        float dzs_acc = 0.f;
        for (int a = 0; a < A; ++a) {
          const int64_t l = (n * S1 + s) * A + a;
          const float hv = valid ? hbuf[l * DP + d] : 0.f;
          due = fmaf(dmv, hv, due);
          const float dzv = hv > 0.f ? dmv * ud * kscale : 0.f;
          hbuf[l * DP + d] = dzv;
          dzs_acc += dzv;
        }
        dzs[(r * S1 + s) * DP + d] = dzs_acc;
        dzn_acc += dzs_acc;
      }
      dzn[n * DP + d] = dzn_acc;
      gb_acc += dzn_acc;
      if (valid) atomicAdd(&gU[u * D + d], due);
    }
    if (threadIdx.x < nrows) items[threadIdx.x] = cand[g * RB * S1 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < nrows) {   // duplicate-index detection inside the block's window
      int f0 = threadIdx.x;
      const int me = items[threadIdx.x];
      for (int j = 0; j < (int)threadIdx.x; ++j)
        if (items[j] == me) { f0 = j; break; }
      first[threadIdx.x] = f0;
    }
    // dV rows into LDS + gW_i accumulation
    for (int j = r; j < nrows; j += RB) {
      float dv = 0.f;
      if (d < D)
        for (int k = 0; k < D; ++k) dv = fmaf(W[(int64_t)k * (D + F) + d], dzs[j * DP + k], dv);
      stage[j * DP + d] = dv;
    }
    if (d < D) {
      for (int j = 0; j < nrows; ++j) {
        const float vj = V[(int64_t)items[j] * D + d];
#pragma unroll
        for (int q = 0; q < 64; ++q)
          if (q < DR) gwi[q] = fmaf(dzs[j * DP + r * DR + q], vj, gwi[q]);
      }
    }
    __syncthreads();
    for (int j = r; j < nrows; j += RB) {
      if (first[j] != j || d >= D) continue;
      float v = stage[j * DP + d];
      for (int j2 = j + 1; j2 < nrows; ++j2)
        if (first[j2] == j) v += stage[j2 * DP + d];
      atomicAdd(&gV[(int64_t)items[j] * D + d], v);
    }
    __syncthreads();
  }
  if (d < D) {
#pragma unroll
    for (int q = 0; q < 64; ++q)
      if (q < DR && r * DR + q < D) atomicAdd(&gW[(int64_t)(r * DR + q) * (D + F) + d], gwi[q]);
  }
  // gb: reduce the RB row-threads of a column through LDS
  __syncthreads();
  sm[threadIdx.x] = gb_acc;
  __syncthreads();
  if (threadIdx.x < DP && threadIdx.x < D) {
    float s = 0.f;
    for (int q = 0; q < RB; ++q) s += sm[q * DP + threadIdx.x];
    atomicAdd(&gb[threadIdx.x], s);
  }
}

// ================================================================================================ K5: noise backward
// gW[:, D:] += A^T B over rows:  MODE 0/1: A = dz [L,DP], B = eps [L,F] (regenerated / injected);  MODE 2: A = dzn [N,DP],
// B = feat[X[n,1]] (the W_f feat term).  Wave task = 32 rows x one 128-wide f chunk x ND*32 rows d of gW; the
// 32x32 accumulators (ND*4 tiles) live in registers over the block's whole row range and leave through one
// float-atomic pass shaped as two 128-B row segments per wave instruction.
template <int ND, int MODE>
__global__ __launch_bounds__(512) void k_noise_bwd(const float* __restrict__ Asrc, const float* __restrict__ Bsrc,
                                                   const int64_t* __restrict__ X, float* __restrict__ gW, int64_t R,
                                                   int D, int F, int DP, rng_key nkey, float nscale) {
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
  const int64_t ntiles = (R + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll 2
    for (int j = 0; j < 16; ++j) {
      const int64_t l = tile * 32 + 2 * j + h;
      float a[ND], bq[4];
#pragma unroll
      for (int mt = 0; mt < ND; ++mt) a[mt] = l < R ? Asrc[l * DP + dbase + mt * 32 + c31] : 0.f;
      if (MODE == 0) {
        noise4((uint32_t)l, (uint32_t)(wave * 32 + c31), nkey, nscale, bq);
      } else {
        const int64_t row = (MODE == 2 && l < R) ? X[2 * l + 1] : l;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int f = wave * 128 + 32 * o + c31;
          bq[o] = (l < R && f < F) ? Bsrc[row * F + f] : 0.f;
        }
      }
#pragma unroll
      for (int mt = 0; mt < ND; ++mt)
#pragma unroll
        for (int o = 0; o < 4; ++o)
          acc[mt][o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bq[o], acc[mt][o], 0, 0, 0);
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
  ARG_CHECK(M->F >= 1 && M->F <= 1024, "F must be in [1, 1024]");
  ARG_CHECK(M->S >= 0 && M->S <= 255 && M->A >= 1 && M->A <= 64, "S in [0,255], A in [1,64]");
  ARG_CHECK(M->user_num > 0 && M->item_num > 0 && M->item_num < 2147483647LL, "bad user_num / item_num");
  ARG_CHECK(M->U && M->V && M->W && M->b && M->feat, "NULL parameter / feature pointer");
  ARG_CHECK(M->expo || (M->ipsP && M->ipsQ && M->ipsBu && M->ipsBi && M->ipsProp && M->ipsD > 0),
            "need expo or the IPS factors");
  return 0;
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
    if (rank == 0) ARG_CHECK(Y != nullptr, "rank==0 needs Y");
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
  float* base = (float*)(ws + y.base);
  float* hbuf = (float*)(ws + y.h);
  float* m = (float*)(ws + y.m);
  float* dmns = (float*)(ws + y.dmns);
  float* dzn = (float*)(ws + y.dzn);

  const rng_key ckey = make_key(rnd->seed, STREAM_CAND, rnd->step);
  const rng_key nkey = make_key(rnd->seed, STREAM_NOISE, rnd->step);
  const rng_key dkey = make_key(rnd->seed, STREAM_DROP, rnd->step);
  const float nscale = -2.0f * 0.69314718055994530942f * M->std * M->std;
  const float kscale = dropout > 0.f ? 1.0f / (float)(1.0 - (double)dropout) : 1.0f;
  const uint32_t thr = dropout > 0.f ? drop_threshold(dropout) : 0u;

  {
    const int64_t total = (int64_t)(D + y.FP) * y.DP + y.NS + (y.GY > 1 ? y.L : 0) + 1;
    const int grid = (int)min((int64_t)2048, (total + 255) / 256);
    prof_begin(ctx, st);
    hipLaunchKernelGGL(k_prep, dim3(grid), dim3(256), 0, st, M->W, WT, D, F, y.DP, y.FP, X, rnd->sample_item, cand, N,
                       M->S, M->item_num, fused ? 1 : 0, ckey, m, y.GY > 1 ? y.L : (int64_t)0, train ? loss : nullptr);
    prof_end(ctx, 0, st);
  }
  {
    const int64_t ng = (N + BASE_RB - 1) / BASE_RB;
    const int grid = (int)min((int64_t)2048, ng);
    const size_t smem = ((size_t)BASE_RB * F + (size_t)BASE_RB * S1 * D + (size_t)256 * BASE_RB) * 4;
    ARG_CHECK(smem <= 160 * 1024, "S too large for the base kernel's LDS tile");
    prof_begin(ctx, st);
    hipLaunchKernelGGL(k_base, dim3(grid), dim3(256), smem, st, WT, M->b, M->V, M->feat, X, cand, base, N, S1, D, F,
                       y.DP);
    prof_end(ctx, 1, st);
  }
  {
    const int64_t ntiles = (y.L + 31) / 32;
    const dim3 grid((unsigned)min((int64_t)1024, ntiles), y.GY), block(64 * y.NC);
    const size_t smem = (size_t)y.NC * 32 * y.ND * 32 * 4;
#define LAUNCH_FWD(ND_, MODE_)                                                                                      \
  hipLaunchKernelGGL((k_noise_fwd<ND_, MODE_>), grid, block, smem, st, WT, base, M->U, X, rnd->noise, rnd->keep, hbuf, \
                     m, y.L, S1, A, D, F, y.DP, nkey, dkey, nscale, thr, kscale)
    prof_begin(ctx, st);
    if (y.ND == 1) { if (fused) LAUNCH_FWD(1, 0); else LAUNCH_FWD(1, 1); }
    else           { if (fused) LAUNCH_FWD(2, 0); else LAUNCH_FWD(2, 1); }
    prof_end(ctx, 2, st);
#undef LAUNCH_FWD
  }
  {
    const int64_t units = (train && rank == 1) ? N / 2 : N;
    const int grid = (int)min((int64_t)2048, (units + 255) / 256);
    prof_begin(ctx, st);
    hipLaunchKernelGGL(k_pair_epilogue, dim3(grid), dim3(256), 0, st, *M, X, Y, cand, m, dmns, pred, loss, N, rank,
                       train ? 1 : 0);
    prof_end(ctx, 3, st);
  }
  if (train) {
    {
      const int RB = 256 / y.DP;
      const int64_t ng = (N + RB - 1) / RB;
      const int grid = (int)min((int64_t)512, ng);
      const size_t smem = (size_t)RB * S1 * y.DP * 4 * 2 + (size_t)RB * S1 * 4 * 2;
      ARG_CHECK(smem <= 160 * 1024, "S too large for the backward kernel's LDS tile");
      prof_begin(ctx, st);
      hipLaunchKernelGGL(k_bwd_small, dim3(grid), dim3(256), max(smem, (size_t)1024), st, M->W, M->U, M->V, X, cand,
                         dmns, hbuf, dzn, G->gU, G->gV, G->gW, G->gb, N, S1, A, D, F, y.DP, kscale);
      prof_end(ctx, 4, st);
    }
    {
      const int64_t ntiles = (y.L + 31) / 32;
      const dim3 grid((unsigned)min((int64_t)256, ntiles), y.GY), block(64 * y.NC);
#define LAUNCH_BWD(ND_, MODE_, A_, B_, R_, G_)                                                                    \
  hipLaunchKernelGGL((k_noise_bwd<ND_, MODE_>), G_, block, 0, st, A_, B_, X, G->gW, R_, D, F, y.DP, nkey, nscale)
      prof_begin(ctx, st);
      if (y.ND == 1) { if (fused) LAUNCH_BWD(1, 0, hbuf, nullptr, y.L, grid); else LAUNCH_BWD(1, 1, hbuf, rnd->noise, y.L, grid); }
      else           { if (fused) LAUNCH_BWD(2, 0, hbuf, nullptr, y.L, grid); else LAUNCH_BWD(2, 1, hbuf, rnd->noise, y.L, grid); }
      prof_end(ctx, 5, st);
      const int64_t ntn = (N + 31) / 32;
      const dim3 gridn((unsigned)min((int64_t)256, ntn), y.GY);
      prof_begin(ctx, st);
      if (y.ND == 1) LAUNCH_BWD(1, 2, dzn, M->feat, N, gridn);
      else           LAUNCH_BWD(2, 2, dzn, M->feat, N, gridn);
      prof_end(ctx, 6, st);
#undef LAUNCH_BWD
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
// which: 0 cand(int32 [N*S1]) 1 WT 2 base [N*S1,DP] 3 h/dz [L,DP] 4 m [L] 5 dmns [N*S1] 6 dzn [N,DP]; info[0..3] = DP, FP, count, elem size
extern "C" int dccf_debug_workspace(dccf_ctx* ctx, int64_t N, int32_t D, int32_t F, int32_t S, int32_t A, int32_t which,
                                    void* dst, int64_t* info, void* stream) {
  ARG_CHECK(ctx && info, "NULL argument");
  const Lay y = make_layout(N, D, F, S, A);
  ARG_CHECK(y.total <= ctx->ws_bytes, "workspace smaller than this layout");
  size_t off = 0, cnt = 0;
  switch (which) {
    case 0: off = y.cand; cnt = (size_t)y.NS; break;
    case 1: off = y.WT; cnt = (size_t)(D + y.FP) * y.DP; break;
    case 2: off = y.base; cnt = (size_t)y.NS * y.DP; break;
    case 3: off = y.h; cnt = (size_t)y.L * y.DP; break;
    case 4: off = y.m; cnt = (size_t)y.L; break;
    case 5: off = y.dmns; cnt = (size_t)y.NS; break;
    case 6: off = y.dzn; cnt = (size_t)N * y.DP; break;
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
