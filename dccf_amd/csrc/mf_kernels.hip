// mf_kernels.hip — RecModel / BiasedMF / IPSBiasedMF: pairwise predict, fused forward+BPR+backward with LDS-staged
// duplicate-row reduction, the full U x I exposure matrix (fp32 MFMA + bias/propensity epilogue), and the fused
// on-device training-negative sampler.
//
// Reference: src/models/RecModel.py:38-48, src/models/BiasedMF.py:17-33, src/models/IPSBiasedMF.py:37-57,
// src/models/BaseModel.py:203-219 (BPR / MSE), README.md:28-30 (full matrix), and
// src/data_processor/DataProcessor.py:446-524 (negatives).
//
// The pairwise kernels are gather / scatter-add kernels — HBM-bound byte movers: one embedding row is one
// wave-instruction (D=64 -> 256 B), rows are staged in LDS, duplicate (table,row) targets inside a block's window are
// chained and summed on chip so that each distinct row leaves through ONE 256-B float-atomic row add.
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float mf_score(const mf_model_t& M, int64_t u, int64_t i, float dot, float& inv_prop) {
  float p = dot;
  inv_prop = 1.f;
  if (M.kind >= 1) p = p + M.bu[u] + M.bi[i] + M.b0[0];
  if (M.kind == 2) {
    const float pr = fmaxf(M.prop[i], M.M);
    p = p / pr;
    inv_prop = pr;   // NOTE: holds the clipped propensity; gradients divide by it
  }
  return p;
}

// ------------------------------------------------------------------------------------------------ predict
__global__ __launch_bounds__(256) void k_mf_predict(mf_model_t M, const int64_t* __restrict__ X, int64_t N,
                                                    float* __restrict__ pred) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t n = wave; n < N; n += nw) {
    const int64_t u = X[2 * n], i = X[2 * n + 1];
    float acc = 0.f;
    for (int d = lane; d < M.D; d += 64) acc = fmaf(M.P[u * M.D + d], M.Q[i * M.D + d], acc);
    acc = wave_sum(acc);
    float ip;
    if (lane == 0) pred[n] = mf_score(M, u, i, acc, ip);
  }
}

extern "C" int mf_predict(const mf_model_t* M, const int64_t* X, int64_t N, float* prediction, void* stream) {
  ARG_CHECK(M && X && prediction && N >= 0, "NULL argument");
  ARG_CHECK(M->P && M->Q && M->D >= 1 && M->kind >= 0 && M->kind <= 2, "bad model");
  ARG_CHECK(M->kind == 0 || (M->bu && M->bi && M->b0), "bias pointers missing");
  ARG_CHECK(M->kind != 2 || M->prop, "propensity missing");
  if (N == 0) return 0;
  const int grid = (int)min((int64_t)2048, (N + 3) / 4);
  hipLaunchKernelGGL(k_mf_predict, dim3(grid), dim3(256), 0, (hipStream_t)stream, *M, X, N, prediction);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ train fwd+bwd
// A block walks chunks of CH = 64 rows of X.  rank 1: a chunk is 32 pairs = rows [k0,k0+32) and [B+k0, B+k0+32).
#define MF_CH 64
__global__ __launch_bounds__(256) void k_mf_train(mf_model_t M, mf_grads_t G, const int64_t* __restrict__ X,
                                                  const float* __restrict__ Y, int64_t N, int rank,
                                                  float* __restrict__ pred, float* __restrict__ loss) {
  extern __shared__ float sm[];
  const int D = M.D;
  float* Pr = sm;                         // [CH][D]  user rows
  float* Qr = Pr + MF_CH * D;             // [CH][D]  item rows
  float* gs = Qr + MF_CH * D;             // [CH]     d loss / d (raw score)  (after the propensity division)
  float* pl = gs + MF_CH;                 // [CH]     predictions
  int64_t* uid = (int64_t*)(pl + MF_CH);  // [CH]
  int64_t* iid = uid + MF_CH;             // [CH]
  int64_t* rown = iid + MF_CH;            // [CH]  global row of each chunk slot (-1 = empty)
  float* ipr = (float*)(rown + MF_CH);    // [CH]  clipped propensity
  int* nxtU = (int*)(ipr + MF_CH);        // [CH]  chain of later slots with the same user, -1 end; -2 = not a head
  int* nxtI = nxtU + MF_CH;               // [CH]
  int* headU = nxtI + MF_CH;              // [CH]  1 if first occurrence
  int* headI = headU + MF_CH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t B = N / 2;
  const int64_t units = rank == 1 ? B : N;
  const int per = rank == 1 ? MF_CH / 2 : MF_CH;
  const int64_t nchunks = (units + per - 1) / per;
  float lsum = 0.f;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t k0 = c * per;
    if (threadIdx.x < MF_CH) {
      const int s = threadIdx.x;
      int64_t n = -1;
      if (rank == 1) {
        const int64_t k = k0 + (s % per);
        if (k < B) n = s < per ? k : B + k;
      } else if (k0 + s < N) {
        n = k0 + s;
      }
      rown[s] = n;
      uid[s] = n >= 0 ? X[2 * n] : -1;
      iid[s] = n >= 0 ? X[2 * n + 1] : -1;
    }
    __syncthreads();
    // phase 0: gather rows, scores
    for (int s = wave; s < MF_CH; s += 4) {
      if (rown[s] < 0) continue;
      const int64_t u = uid[s], i = iid[s];
      float acc = 0.f;
      for (int d = lane; d < D; d += 64) {
        const float pv = M.P[u * D + d], qv = M.Q[i * D + d];
        Pr[s * D + d] = pv;
        Qr[s * D + d] = qv;
        acc = fmaf(pv, qv, acc);
      }
      acc = wave_sum(acc);
      if (lane == 0) {
        float ip;
        const float p = mf_score(M, u, i, acc, ip);
        pl[s] = p;
        ipr[s] = ip;
        pred[rown[s]] = p;
      }
    }
    __syncthreads();
    // phase 1: loss gradient per slot + duplicate chains (slot order; users vs users, items vs items)
    if (threadIdx.x < MF_CH) {
      const int s = threadIdx.x;
      float g = 0.f;
      if (rown[s] >= 0) {
        if (rank == 1) {
          const int j = s % per;
          const float d = pl[j] - pl[per + j];
          const float sg = 1.f / (1.f + expf(-d));
          const float gp = -(1.f - sg);
          g = s < per ? gp : -gp;
          if (s < per) lsum += -logf(sg);
        } else {
          const float diff = pl[s] - Y[rown[s]];
          g = 2.f * diff / (float)N;
          lsum += diff * diff / (float)N;
        }
        g = g / ipr[s];
      }
      gs[s] = g;
      int hu = 1, hi = 1, nu = -1, ni = -1;
      if (rown[s] < 0) { hu = 0; hi = 0; }
      else {
        for (int j = 0; j < s; ++j) {
          if (uid[j] == uid[s]) hu = 0;
          if (iid[j] == iid[s]) hi = 0;
        }
        for (int j = MF_CH - 1; j > s; --j) {
          if (uid[j] == uid[s]) nu = j;
          if (iid[j] == iid[s]) ni = j;
        }
      }
      headU[s] = hu; headI[s] = hi; nxtU[s] = nu; nxtI[s] = ni;
    }
    __syncthreads();
    // phase 2: one atomic row add per distinct user / item of the chunk
    for (int s = wave; s < MF_CH; s += 4) {
      if (rown[s] < 0) continue;
      if (headU[s]) {
        for (int d = lane; d < D; d += 64) {
          float v = 0.f;
          for (int j = s; j >= 0; j = nxtU[j]) v = fmaf(gs[j], Qr[j * D + d], v);
          atomicAdd(&G.gP[uid[s] * D + d], v);
        }
        if (G.touchedP && lane == 0) G.touchedP[uid[s]] = 1;
        if (M.kind >= 1 && lane == 0) {
          float v = 0.f;
          for (int j = s; j >= 0; j = nxtU[j]) v += gs[j];
          atomicAdd(&G.gbu[uid[s]], v);
        }
      }
      if (headI[s]) {
        for (int d = lane; d < D; d += 64) {
          float v = 0.f;
          for (int j = s; j >= 0; j = nxtI[j]) v = fmaf(gs[j], Pr[j * D + d], v);
          atomicAdd(&G.gQ[iid[s] * D + d], v);
        }
        if (G.touchedQ && lane == 0) G.touchedQ[iid[s]] = 1;
        if (M.kind >= 1 && lane == 0) {
          float v = 0.f;
          for (int j = s; j >= 0; j = nxtI[j]) v += gs[j];
          atomicAdd(&G.gbi[iid[s]], v);
        }
      }
    }
    if (M.kind >= 1 && threadIdx.x == 0) {
      float v = 0.f;
      for (int s = 0; s < MF_CH; ++s) v += gs[s];
      atomicAdd(G.gb0, v);
    }
    __syncthreads();
  }
  __shared__ float red[4];
  lsum = wave_sum(lsum);
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
}

// Small batches (the reference's default 128 pairs): the chunked kernel above runs 4 workgroups through three barriers
// and takes 37 us; here ONE WAVE owns one unit (a BPR pair, or one MSE row): rows gathered straight into registers
// (lane = column, up to 4 columns per lane), scores by a wave reduction, gradients leave as one atomic row add per
// embedding row (duplicates inside a batch are rare at this size and the atomics sum them anyway).
__global__ __launch_bounds__(256) void k_mf_train_small(mf_model_t M, mf_grads_t G, const int64_t* __restrict__ X,
                                                        const float* __restrict__ Y, int64_t N, int rank,
                                                        float* __restrict__ pred, float* __restrict__ loss) {
  const int D = M.D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t B = N / 2;
  const int64_t units = rank == 1 ? B : N;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float lsum = 0.f, b0sum = 0.f;
  for (int64_t k = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6); k < units; k += nw) {
    const int nrow = rank == 1 ? 2 : 1;
    int64_t u[2], it[2], row[2];
    float pv[2][4], qv[2][4], sc[2], ip[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      row[r] = r == 0 ? k : B + k;
      const bool on = r < nrow;
      u[r] = X[2 * (on ? row[r] : k)];
      it[r] = X[2 * (on ? row[r] : k) + 1];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d = lane + 64 * c;
        const int dc = d < D ? d : 0;
        pv[r][c] = M.P[u[r] * D + dc];
        qv[r][c] = M.Q[it[r] * D + dc];
        if (d < D) acc = fmaf(pv[r][c], qv[r][c], acc);
      }
      acc = wave_sum(acc);
      sc[r] = mf_score(M, u[r], it[r], acc, ip[r]);
    }
    float g[2] = {0.f, 0.f};
    if (rank == 1) {
      const float dd = sc[0] - sc[1];
      const float sg = 1.f / (1.f + expf(-dd));
      const float gp = -(1.f - sg);
      g[0] = gp / ip[0];
      g[1] = -gp / ip[1];
      if (lane == 0) {
        pred[k] = sc[0];
        pred[B + k] = sc[1];
        lsum += -logf(sg);
      }
    } else {
      const float diff = sc[0] - Y[k];
      g[0] = (2.f * diff / (float)N) / ip[0];
      if (lane == 0) {
        pred[k] = sc[0];
        lsum += diff * diff / (float)N;
      }
    }
    const bool same_user = rank == 1 && u[0] == u[1];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r >= nrow) break;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d = lane + 64 * c;
        if (d >= D) break;
        // the user row of a BPR pair is shared by its positive and negative row: one atomic row add for both
        if (same_user) { if (r == 0) atomicAdd(&G.gP[u[0] * D + d], fmaf(g[0], qv[0][c], g[1] * qv[1][c])); }
        else atomicAdd(&G.gP[u[r] * D + d], g[r] * qv[r][c]);
        atomicAdd(&G.gQ[it[r] * D + d], g[r] * pv[r][c]);
      }
      if (lane == 0) {
        if (M.kind >= 1) {
          atomicAdd(&G.gbu[u[r]], g[r]);
          atomicAdd(&G.gbi[it[r]], g[r]);
        }
        if (G.touchedP) G.touchedP[u[r]] = 1;
        if (G.touchedQ) G.touchedQ[it[r]] = 1;
      }
    }
    b0sum += g[0] + (nrow == 2 ? g[1] : 0.f);
  }
  __shared__ float red[8];             // one atomic per workgroup on the two single-address sums
  if (lane == 0) { red[wv] = lsum; red[4 + wv] = b0sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
    if (M.kind >= 1) atomicAdd(G.gb0, red[4] + red[5] + red[6] + red[7]);
  }
}

static int mf_train_launch(const mf_model_t* M, const int64_t* X, const float* Y, int64_t N, int32_t rank, const mf_grads_t* G,
                           float* prediction, float* loss, void* stream, bool zero_loss);
extern "C" int mf_train_fwdbwd(dccf_ctx* ctx, const mf_model_t* M, const int64_t* X, const float* Y, int64_t N,
                               int32_t rank, const mf_grads_t* G, float* prediction, float* loss, void* stream) {
  (void)ctx;
  return mf_train_launch(M, X, Y, N, rank, G, prediction, loss, stream, true);
}
// zero_loss = false: an earlier launch on the stream already set *loss = 0 (mf_train_step)
static int mf_train_launch(const mf_model_t* M, const int64_t* X, const float* Y, int64_t N, int32_t rank, const mf_grads_t* G,
                           float* prediction, float* loss, void* stream, bool zero_loss) {
  ARG_CHECK(M && X && G && prediction && loss && N >= 0, "NULL argument");
  ARG_CHECK(M->P && M->Q && M->D >= 1 && M->D <= 256 && M->kind >= 0 && M->kind <= 2, "bad model");
  ARG_CHECK(G->gP && G->gQ, "NULL gradient pointer");
  ARG_CHECK(M->kind == 0 || (M->bu && M->bi && M->b0 && G->gbu && G->gbi && G->gb0), "bias pointers missing");
  ARG_CHECK(M->kind != 2 || M->prop, "propensity missing");
  ARG_CHECK(rank == 0 || rank == 1, "rank must be 0 or 1");
  if (rank == 1) ARG_CHECK(N % 2 == 0, "rank==1 needs [positives ; negatives] (even N)");
  if (rank == 0) ARG_CHECK(Y != nullptr, "rank==0 needs Y");
  hipStream_t st = (hipStream_t)stream;
  if (zero_loss) HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), st));
  if (N == 0) return 0;
  const int per = rank == 1 ? MF_CH / 2 : MF_CH;
  const int64_t units = rank == 1 ? N / 2 : N;
  if (units <= 1024 && M->D <= 256) {        // one wave per pair / row (measured: 11.6 vs 36.8 us at 128 pairs)
    hipLaunchKernelGGL(k_mf_train_small, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, *M, *G, X, Y, N, rank, prediction,
                       loss);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const int64_t nchunks = (units + per - 1) / per;
  const size_t smem = (size_t)2 * MF_CH * M->D * 4 + MF_CH * (4 + 4 + 8 + 8 + 8 + 4 + 4 + 4 + 4 + 4);
  const int grid = (int)min((int64_t)2048, nchunks);
  hipLaunchKernelGGL(k_mf_train, dim3(grid), dim3(256), smem, st, *M, *G, X, Y, N, rank, prediction, loss);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ one call per MF train step
// The body of the reference's batch loop for the MF family (src/runners/BaseRunner.py:172-188 around BaseModel.forward,
// src/models/BaseModel.py:203-219) under the windowed lazy regularisation of DESIGN.md section 4b: the rows of the batch are
// claimed and brought up to step - 1 (k_lazy_catchup_pairs, which reads the ids from X), forward + loss + backward accumulate their gradient rows
// (mf_train_fwdbwd), and ONE optimizer launch updates those rows with their gradient, everything outside the two row segments
// (bias vectors, global bias) densely, and this step's window of the other rows (dccf_lazy_opt_step) — the dense pass over all
// (user_num + item_num) x D parameters, 80 % of an IPSBiasedMF step at batch 128, shrinks to one K-th.  Same results as the dense
// step: untouched rows bit-identical.
extern "C" int mf_train_step(dccf_ctx* ctx, const mf_model_t* M, const int64_t* X, const float* Y, int64_t N, int32_t rank,
                             const mf_grads_t* G, const dccf_opt_t* opt, int32_t* ids, float* prediction, float* loss,
                             void* stream) {
  (void)ctx;
  (void)ids;                                   // (scratch of the first form of this call; the catch-up reads X itself)
  ARG_CHECK(M && X && G && opt && prediction && loss && N >= 1, "NULL argument / empty batch");
  ARG_CHECK(opt->lazy_K > 0 && opt->nseg == 2 && opt->seg_rows[0] == M->user_num && opt->seg_rows[1] == M->item_num &&
                opt->seg_width[0] == M->D && opt->seg_width[1] == M->D,
            "mf_train_step needs the lazy optimizer over the two row segments (P, Q) of this model");
  ARG_CHECK(2 * N <= opt->lazy_list_cap, "batch too large for the lazy row list");
  ARG_CHECK(M->user_num < 2147483647LL && M->item_num < 2147483647LL, "row ids must fit 32 bits");
  // three launches: catch-up (+ loss = 0), forward / backward, optimizer
  if (int e = dccf_lazy_catchup_pairs(opt, X, N, 0, 1, loss, (hipStream_t)stream)) return e;
  mf_grads_t g = *G;
  g.touchedP = g.touchedQ = nullptr;          // the step's rows are on the lazy list: no bytes to keep
  if (int e = mf_train_launch(M, X, Y, N, rank, &g, prediction, loss, stream, false)) return e;
  return dccf_lazy_opt_step(opt, 2 * N, stream);
}

// ------------------------------------------------------------------------------------------------ full U x I matrix
// out[u][i] = (P[u].Q[i] + bu[u] + bi[i] + b0) / max(prop[i], M): 128 x 128 tile per block, 4 waves x (64 x 64),
// fp32 MFMA 32x32x2 over K = D, operands staged in LDS with a one-float row pad (conflict-free ds_read_b32).
// Output-write bound (U*I*4 bytes) for D <= 64, so the stores decide: the finished tile goes back through LDS and
// leaves as whole 512-B row segments (a wave instruction = 256 contiguous bytes), and consecutive tiles of a row band
// are given to the SAME XCD (workgroups are dealt to the 8 XCDs round-robin, each with its own L2), so the two cache
// lines a tile shares with its left/right neighbours (item_num is not a multiple of 32) are completed in one L2.
#define FT 128
__global__ __launch_bounds__(256) void k_mf_full(mf_model_t M, float* __restrict__ out, int64_t nblk) {
  extern __shared__ float sm[];
  const int D = M.D, LD = D + 1;
  float* Ps = sm;              // [FT][LD]
  float* Qs = Ps + FT * LD;    // [FT][LD]
  float* Cs = sm;              // [FT][FT + 1] after the k-loop (the launch sizes LDS for the larger of the two uses)
  const int64_t gx = (M.item_num + FT - 1) / FT;
  // XCD-aware order: physical block b runs on XCD b % 8; XCD x walks the contiguous tile range [x * per, (x + 1) * per)
  const int64_t per = (nblk + 7) / 8;
  const int64_t tile = ((int64_t)blockIdx.x % 8) * per + (int64_t)blockIdx.x / 8;
  if (tile >= nblk) return;
  const int64_t u0 = (tile / gx) * FT, i0 = (tile % gx) * FT;
  for (int idx = threadIdx.x; idx < FT * D; idx += 256) {
    const int r = idx / D, k = idx % D;
    Ps[r * LD + k] = (u0 + r < M.user_num) ? M.P[(u0 + r) * D + k] : 0.f;
    Qs[r * LD + k] = (i0 + r < M.item_num) ? M.Q[(i0 + r) * D + k] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int wu = (wave >> 1) * 64, wi = (wave & 1) * 64;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  for (int k = 0; k < D; k += 2) {
    float av[2], bv[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) av[a] = Ps[(wu + a * 32 + c31) * LD + k + h];
#pragma unroll
    for (int b = 0; b < 2; ++b) bv[b] = Qs[(wi + b * 32 + c31) * LD + k + h];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
  }
  __syncthreads();               // every wave is done reading Ps / Qs: the space becomes the output tile
  const float b0 = M.kind >= 1 ? M.b0[0] : 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int ci = wi + b * 32 + c31;
    const int64_t i = min(i0 + ci, M.item_num - 1);
    const float bi = M.kind >= 1 ? M.bi[i] : 0.f;
    const float pr = M.kind == 2 ? fmaxf(M.prop[i], M.M) : 1.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ru = wu + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[a][b][r];
        if (M.kind >= 1) v = v + M.bu[min(u0 + ru, M.user_num - 1)] + bi + b0;
        if (M.kind == 2) v = v / pr;
        Cs[ru * (FT + 1) + ci] = v;
      }
  }
  __syncthreads();
  // row segments out: wave w writes rows w, w + 4, ...; two 256-B instructions per row
  const int64_t ncol = min((int64_t)FT, M.item_num - i0);
  for (int ru = wave; ru < FT; ru += 4) {
    const int64_t u = u0 + ru;
    if (u >= M.user_num) break;
    float* orow = out + u * M.item_num + i0;
    if (lane < ncol) orow[lane] = Cs[ru * (FT + 1) + lane];
    if (lane + 64 < ncol) orow[lane + 64] = Cs[ru * (FT + 1) + lane + 64];
  }
}

// Persistent form for D in {16, 32, 64, 128}: a workgroup (8 waves) keeps ONE band of 128 users in LDS and walks a range
// of 64-item tiles of it.  Per tile: the next Q tile (and its bias / propensity columns) is prefetched into registers
// while the current one is multiplied (fp32 MFMA 32x32x2, wave w = user rows 32(w>>1).. x items 32(w&1)..), the finished
// tile goes through LDS and leaves as 256-B row segments — one wave store instruction per row.  The stores are issued
// AFTER the prefetched registers were consumed: on gfx9 a store counts in vmcnt like a load, so waiting for the prefetch
// behind them would wait for the stores to be acknowledged.  Neighbouring tiles of a band are written by the same CU a
// few us apart, so the cache lines they share (item_num is not a multiple of 32) are completed in one L2.
#define FI 64
template <int D, int FU>
__global__ __launch_bounds__(FU * 4) void k_mf_full_band(mf_model_t M, float* __restrict__ out, int splits) {
  extern __shared__ float sm[];
  constexpr int LD = D + 1;
  constexpr int NT = FU * 4;                // threads: one wave per 32 x 32 MFMA tile of the FU x 64 output tile
  constexpr int QR = (FI * D + NT - 1) / NT;  // floats of a Q tile per thread
  float* Ps = sm;                           // [FU][LD]
  float* Qs = Ps + FU * LD;                 // [FI][LD]
  float* Cs = Qs + FI * LD;                 // [FU][FI]   the output tile
  float* Bu = Cs + FU * FI;                 // [FU]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  const int wu = (wave >> 1) * 32, wi = (wave & 1) * 32;
  const int64_t gi = (M.item_num + FI - 1) / FI;
  const int64_t band = blockIdx.x / splits, sp = blockIdx.x % splits;
  const int64_t t0 = gi * sp / splits, t1 = gi * (sp + 1) / splits;
  const int64_t u0 = band * FU;
  if (t0 >= t1) return;
  for (int idx = threadIdx.x; idx < FU * D; idx += NT) {
    const int r = idx / D, k = idx % D;
    Ps[r * LD + k] = M.P[min(u0 + r, M.user_num - 1) * D + k];
  }
  if (threadIdx.x < FU) Bu[threadIdx.x] = M.kind >= 1 ? M.bu[min(u0 + threadIdx.x, M.user_num - 1)] : 0.f;
  const float b0 = M.kind >= 1 ? M.b0[0] : 0.f;
  float qreg[QR], bin, prn;
  auto prefetch = [&](int64_t t) {
    const int64_t i0 = t * FI;
#pragma unroll
    for (int j = 0; j < QR; ++j) {
      const int idx = min((int)threadIdx.x + j * NT, FI * D - 1);
      qreg[j] = M.Q[min(i0 + idx / D, M.item_num - 1) * D + idx % D];
    }
    const int64_t i = min(i0 + wi + c31, M.item_num - 1);
    bin = M.kind >= 1 ? M.bi[i] : 0.f;
    prn = M.kind == 2 ? fmaxf(M.prop[i], M.M) : 1.f;
  };
  auto commit = [&]() {                      // prefetched registers -> the Q tile in LDS
#pragma unroll
    for (int j = 0; j < QR; ++j) {
      const int idx = threadIdx.x + j * NT;
      if (idx < FI * D) Qs[(idx / D) * LD + idx % D] = qreg[j];
    }
  };
  prefetch(t0);
  commit();
  float bic = bin, prc = prn;
  __syncthreads();
  for (int64_t t = t0; t < t1; ++t) {
    const int64_t i0 = t * FI;
    if (t + 1 < t1) prefetch(t + 1);          // in flight during the k-loop
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int k = 0; k < D; k += 2) {
      const float av = Ps[(wu + c31) * LD + k + h];
      const float bv = Qs[(wi + c31) * LD + k + h];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ru = wu + (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = acc[r];
      if (M.kind >= 1) v = v + Bu[ru] + bic + b0;
      if (M.kind == 2) v = v / prc;
      Cs[ru * FI + wi + c31] = v;
    }
    __syncthreads();                          // the Q tile has been read by every wave, the output tile is complete
    if (t + 1 < t1) {
      commit();                               // waits for the prefetch (no store is outstanding behind it)
      bic = bin;
      prc = prn;
    }
    const int64_t ncol = min((int64_t)FI, M.item_num - i0);
    for (int ru = wave; ru < FU; ru += NT / 64) {
      const int64_t u = u0 + ru;
      if (u >= M.user_num) break;
      if (lane < ncol) out[u * M.item_num + i0 + lane] = Cs[ru * FI + lane];
    }
    __syncthreads();                          // next Q tile visible; the output tile may be overwritten
  }
}

template <int D, int FU>
static int launch_full_band(const mf_model_t* M, float* out, hipStream_t st) {
  constexpr int LD = D + 1;
  const size_t smem = (size_t)(FU * LD + FI * LD + FU * FI + FU) * 4;
  const int64_t bands = (M->user_num + FU - 1) / FU, gi = (M->item_num + FI - 1) / FI;
  int splits = (int)max((int64_t)1, min(gi, (4096 + bands - 1) / bands));
  if (getenv("DCCF_FULL_SPLITS")) splits = (int)max((int64_t)1, min(gi, (int64_t)atoi(getenv("DCCF_FULL_SPLITS"))));
  ARG_CHECK(bands * splits < 2147483647LL, "matrix too large for one launch");
  HIP_TRY(hipFuncSetAttribute((const void*)k_mf_full_band<D, FU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((k_mf_full_band<D, FU>), dim3((unsigned)(bands * splits)), dim3(FU * 4), smem, st, *M, out, splits);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Q [I][D] -> Q^T in TILE-MAJOR order [Ipad / TW][D][TW] (zero padded): the B operand of a tile of TW items is one contiguous
// D x TW block (a wave reads 128-B runs of it).  A plain [D][Ipad] layout puts the D rows of a tile 4 * Ipad bytes apart — a
// stride with its low address bits all zero, so the 2 x D/2 runs a wave fetches per tile fell on the same L2 channels
// (measured at D = 64: 7.4 ms for the loads alone, with neither MFMA nor stores in the loop).
__global__ __launch_bounds__(256) void k_transpose_q(const float* __restrict__ Q, int64_t I, int D, int64_t Ipad, int TW,
                                                     float* __restrict__ QT) {
  for (int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; x < Ipad * D; x += (int64_t)gridDim.x * blockDim.x) {
    const int64_t T = x / ((int64_t)D * TW);
    const int k = (int)((x / TW) % D), c = (int)(x % TW);
    const int64_t i = T * TW + c;
    QT[x] = i < I ? Q[i * D + k] : 0.f;
  }
}


// Row-band form (D in {16, 32, 64}): a workgroup (8 waves) owns 32 users, whose factors every wave keeps in registers as
// the MFMA A operand, and walks the items 256 at a time: wave w multiplies the 32 x 32 block w of the 32 x 256 tile (B
// operand straight from Q^T, two 128-B runs per k-step, next tile's in flight), the tile meets in LDS (double-buffered:
// one barrier per tile) and leaves as FULL 1-KB row segments — wave w stores rows w, w+8, w+16, w+24.  Why this shape: a
// pure-write microbenchmark of this matrix (75 k x 64 k, rows 4-byte aligned) reaches 4.7 TB/s with fill, 3.8-4.1 TB/s
// when a workgroup marches along 32 rows with >= 256-B segments per store, 2.5 TB/s with 128-row bands and 1.9 TB/s with
// the 128-B segments MFMA accumulators give directly.
template <int D, int RB, int NW>
__global__ __launch_bounds__(64 * NW) void k_mf_full_rows(mf_model_t M, float* __restrict__ out, int splits,
                                                         const float* __restrict__ QT, int64_t Ipad, int64_t nbands, int order) {
  constexpr int TW = NW * 32;                 // items per tile: one 32 x 32 block per wave
  __shared__ float Cs[2][32][TW + 4];
  __shared__ float Bu[RB * 32];
  constexpr int KS = D / 2, Q4 = D / 4;
  // (the wave index as a scalar: row pointers and tile bases then live in SGPRs and their arithmetic leaves the vector pipe,
  // which the fp32 MFMA shares)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c31 = lane & 31;
  const int64_t gt = (M.item_num + TW - 1) / TW;
  // workgroup -> (band, piece).  order 0: the piece index fastest — XCD x (= workgroup id mod 8) reads the pieces = x (mod 8),
  // but the ~64 workgroups resident on an XCD cover ALL its splits / 8 pieces at once.  order 1 (splits a multiple of 8): the XCD
  // keeps ONE piece while the bands stream past it — id = ((sp_hi * nbands) + band) * 8 + sp_lo — so its live slice of Q^T is
  // 1 / splits of it (1 MB at D = 128, 64 k items) instead of 1 / 8 (4.1 MB: more than the XCD's 4 MB L2)
  int64_t band, sp;
  if (order == 1) {
    const int64_t lo = blockIdx.x & 7, rest = blockIdx.x >> 3;
    band = rest % nbands;
    sp = (rest / nbands) * 8 + lo;
  } else {
    band = blockIdx.x / splits;
    sp = blockIdx.x % splits;
  }
  const int64_t T0 = gt * sp / splits, T1 = gt * (sp + 1) / splits;
  const int64_t u0 = band * (32 * RB);
  float pa[RB][KS];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const float4* prow = reinterpret_cast<const float4*>(M.P + min(u0 + b * 32 + c31, M.user_num - 1) * D);
#pragma unroll
    for (int j = 0; j < Q4; ++j) {
      const float4 v = prow[j];
      pa[b][2 * j] = h ? v.y : v.x;
      pa[b][2 * j + 1] = h ? v.w : v.z;
    }
  }
  if (threadIdx.x < RB * 32) Bu[threadIdx.x] = M.kind >= 1 ? M.bu[min(u0 + threadIdx.x, M.user_num - 1)] : 0.f;
  __syncthreads();
  const float b0 = M.kind >= 1 ? M.b0[0] : 0.f;
  // ONE register copy of the B operand: the next tile's columns are fetched right after the k-loop consumed the current
  // ones (the epilogue, the barrier and the other resident workgroups cover the round trip) — at D = 64 the second copy cost
  // the registers that decide between two and three waves per SIMD
  float qb[KS], bin = 0.f, prn = 1.f;
  auto prefetch = [&](int64_t T) {
    const int64_t ic = T * TW + wave * 32 + c31;                // < Ipad (Q^T is padded to whole tiles)
    const int64_t i = min(ic, M.item_num - 1);
    const float* qc = QT + T * (int64_t)(D * TW) + h * TW + wave * 32 + c31;      // tile-major Q^T: [T][k][TW]
#pragma unroll
    for (int k = 0; k < KS; ++k) qb[k] = qc[2 * k * TW];
    bin = M.kind >= 1 ? M.bi[i] : 0.f;
    prn = M.kind == 2 ? fmaxf(M.prop[i], M.M) : 1.f;
  };
  if (T0 < T1) prefetch(T0);
  int buf = 0;
  for (int64_t T = T0; T < T1; ++T) {
    const float bic = bin + b0, prc = prn;
    // v / prc for 16 values with one divisor: one division for r = 1 / prc, then q = v r refined by one residual step
    // (q + (v - q prc) r: the correction step of the division expansion itself) — 3 instructions per value instead of ~10 on
    // the pipe the fp32 MFMA shares with the vector ALU
    const float rinv = 1.0f / prc;
    const int64_t c0 = T * TW;
    // the RB sub-bands of 32 users share the B operand (Q traffic / RB) and are RB independent accumulator chains
    f32x16 accs[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accs[b][r] = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int b = 0; b < RB; ++b) accs[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[b][k], qb[k], accs[b], 0, 0, 0);
    if (T + 1 < T1) prefetch(T + 1);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const f32x16 acc = accs[b];
      float (*C)[TW + 4] = Cs[buf];
      buf ^= 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[r];
        if (M.kind >= 1) v = v + Bu[b * 32 + row] + bic;
        if (M.kind == 2) {
          const float q = v * rinv;
          v = fmaf(fmaf(-q, prc, v), rinv, q);
        }
        C[row][wave * 32 + c31] = v;
      }
      __syncthreads();     // the tile is complete (the other buffer is free again: every wave stored it before arriving)
      if (c0 + TW <= M.item_num && u0 + b * 32 + 32 <= M.user_num) {      // whole tile inside the matrix: no per-store tests
        float* orow = out + (u0 + b * 32 + wave) * M.item_num + c0 + lane;
#pragma unroll
        for (int rr = 0; rr < 32 / NW; ++rr) {
#pragma unroll
          for (int q = 0; q < TW / 64; ++q) orow[q * 64] = C[wave + NW * rr][q * 64 + lane];
          orow += (int64_t)NW * M.item_num;
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 32 / NW; ++rr) {
          const int row = wave + NW * rr;
          const int64_t u = u0 + b * 32 + row;
          if (u < M.user_num) {
            float* orow = out + u * M.item_num + c0;
#pragma unroll
            for (int q = 0; q < TW / 64; ++q) {
              const int c = q * 64 + lane;
              if (c0 + c < M.item_num) orow[c] = C[row][c];
            }
          }
        }
      }
    }
  }
}

// ---- line form (round 3): B through LDS by DMA, line-aligned stores --------------------------------------------------------------
// Q [I][D], bi, prop -> the operand stream of k_mf_full_lines: per tile of 32 items a [D / 8][2][32][4] image — the 16 bytes lane
// (c, h) feeds to four consecutive k-steps of v_mfma_f32_32x32x2_f32 (k = 2 (4 q + j) + h, j < 4) are one ds_read_b128 — and per tile
// 64 floats of epilogue operands: bi[i] (kind >= 1) and max(prop[i], M) (kind 2).  Padded with zero tiles to Tpad.
// (k < kvalid of the model's columns [koff, koff + kvalid) are real, the rest of the D-wide image is zero: embedding sizes that are no
// tile of the kernel are padded here, sizes above 128 run as two passes over the halves of the contraction)
__global__ __launch_bounds__(256) void k_full_lines_prep(mf_model_t M, int D, int64_t Tpad, float* __restrict__ QT4, float* __restrict__ AUX,
                                                         int koff, int kvalid, int kind) {
  const int64_t n = Tpad * D * 32;
  for (int64_t x = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
    const int64_t T = x / (D * 32);
    const int rem = (int)(x % (D * 32));
    const int j = rem & 3, col = (rem >> 2) & 31, qh = rem >> 7;
    const int k = 2 * (4 * (qh >> 1) + j) + (qh & 1);
    const int64_t it = T * 32 + col;
    QT4[x] = (it < M.item_num && k < kvalid) ? M.Q[it * M.D + koff + k] : 0.f;
    if (rem < 64) {
      const int64_t i2 = T * 32 + (rem & 31);
      float v = rem < 32 ? 0.f : 1.f;
      if (i2 < M.item_num) {
        if (rem < 32 && kind >= 1) v = M.bi[i2];
        if (rem >= 32 && kind == 2) v = fmaxf(M.prop[i2], M.M);
      }
      AUX[T * 64 + rem] = v;
    }
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Line form; D = the contraction's tile, 16 / 32 / 64 / 128 (the model's embedding size is padded to it: koff / kvalid).  A workgroup is 4 waves = 4 bands of 32 users (A operand register-resident) that multiply the SAME
// tiles of 32 items: the tiles arrive 16 KB at a time (128 / D of them) by LDS-DMA (global_load_lds_dwordx4: no registers, no address
// arithmetic per k-step, 4 wave-instructions per wave and 16 KB instead of D / 2 dword loads per wave and TILE), two buffers, one barrier
// per 16 KB; the L2 -> CU traffic of Q^T is a quarter of k_mf_full_rows's, where every wave fetched its own operand.  blockIdx & 7
// (the XCD) picks the item range.
// The stores are LINE-ALIGNED.  item_num is odd as a rule, so a row of `out` starts anywhere in a 128-byte line and a tile's 128-byte
// row piece straddles two lines; written that way the matrix leaves at 3.5 TB/s, as whole lines at 5.7 (scripts/store_shapes.hip,
// profiles/r03_store_shapes.txt).  Every wave keeps 64 floats per row in LDS in MEMORY-LINE coordinates: the 32 new values of row u go to
// positions (d_u + j) mod 64, d_u = (address of out[u][0] / 4) mod 32 being the row's phase — that completes one of the two lines, which
// leaves as dwordx4 stores (8 rows x 128 B per instruction), while the other line keeps the row's last d_u values for the next tile.
// Only the first line of a workgroup's item range and the flush after its last tile are masked dword stores.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mf_full_lines(mf_model_t M, float* __restrict__ out,
                                                                                                     const float* __restrict__ QT4,
                                                                                                     const float* __restrict__ AUX, int splits,
                                                                                                     int koff, int kvalid, int kind, int add, int splitbar) {
  // koff / kvalid: this launch contracts the model's columns [koff, koff + kvalid) (kvalid <= D: the rest of the tile is zero);
  // kind: the epilogue of THIS launch (0: none); add: the contraction continues the partial product already in `out`
  constexpr int S = 128 / D, TILE = D * 32, CHUNK = S * TILE, NQ = D / 8, CW = 64;           // CHUNK = 4096 floats = 16 KB
  static_assert(S * D == 128 && CHUNK == 4096, "D must be 16, 32, 64 or 128");
  extern __shared__ __attribute__((aligned(128))) float sm[];   // [2][CHUNK] operand | [2][S][64] epilogue operands | [4][32][CW] windows
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, c31 = lane & 31;
  int* const bar = reinterpret_cast<int*>(sm + 2 * CHUNK + 2 * S * 64 + 4 * 32 * CW);      // splitbar: arrivals counter
  if (splitbar && threadIdx.x == 0) *bar = 0;
  int arrivals = 0;
  float* const Ax = sm + 2 * CHUNK;
  float* const Cw = sm + 2 * CHUNK + 2 * S * 64 + wave * 32 * CW;
  const int64_t I = M.item_num, U = M.user_num;
  const int64_t gt = (I + 31) / 32;
  const int64_t grp = blockIdx.x / splits, sp = blockIdx.x % splits;
  const int64_t T0 = gt * sp / splits, T1 = gt * (sp + 1) / splits;
  if (T0 >= T1) return;                                         // (the whole workgroup: before any barrier)
  const int64_t u0 = (grp * 4 + wave) * 32;
  const bool rows_full = u0 + 32 <= U;
  const float b0 = kind >= 1 ? M.b0[0] : 0.f;
  float pa[D / 2];                                              // pa[i] = P[u][koff + 2 i + h]
  if (kvalid == D && M.D == D && (uintptr_t)M.P % 16 == 0) {
    const float4* prow = reinterpret_cast<const float4*>(M.P + min(u0 + c31, U - 1) * D);
#pragma unroll
    for (int j = 0; j < D / 4; ++j) {
      const float4 v = prow[j];
      pa[2 * j] = h ? v.y : v.x;
      pa[2 * j + 1] = h ? v.w : v.z;
    }
  } else {
    const float* prow = M.P + min(u0 + c31, U - 1) * M.D + koff;
#pragma unroll
    for (int i = 0; i < D / 2; ++i) {
      const float v = prow[min(2 * i + h, kvalid - 1)];        // (the load itself is unconditional: a clamped index)
      pa[i] = 2 * i + h < kvalid ? v : 0.f;
    }
  }
  const uint32_t obase = (uint32_t)((uintptr_t)out >> 2);
  const uint32_t cw_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)Cw;
  float bu[16];
  uint32_t wadr[2][16];                                        // LDS byte address of this lane's value of accumulator row r: even / odd tiles
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    bu[r] = kind >= 1 ? M.bu[min(u0 + row, U - 1)] : 0.f;
    const uint32_t d = (obase + (uint32_t)(((u0 + row) * I) & 31)) & 31u;
    wadr[0][r] = cw_lds + (uint32_t)(row * CW + (int)((d + c31) & 63u)) * 4u;
    wadr[1][r] = cw_lds + (uint32_t)(row * CW + (int)((d + c31 + 32u) & 63u)) * 4u;
  }
  uint32_t dr[4];                                               // phases of the rows this lane stores: row 8 i + (lane >> 3), piece lane & 7
#pragma unroll
  for (int i = 0; i < 4; ++i) dr[i] = (obase + (uint32_t)(((u0 + 8 * i + (lane >> 3)) * I) & 31)) & 31u;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)sm);
  // (inline asm, not __builtin_amdgcn_global_load_lds: hipcc drains a builtin DMA with vmcnt(0) before the next ds_read of the array)
  auto dma = [&](int64_t Tc, int buf) {
    const float4* src = reinterpret_cast<const float4*>(QT4) + Tc * (int64_t)(TILE / 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = j * 4 + wave;                              // 16 pieces of 1 KB: wave-uniform
      const uint32_t ldsb = lds0 + (uint32_t)(buf * CHUNK + pc * 256) * 4u;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src + pc * 64 + lane), "s"(ldsb) : "memory", "m0");
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
      if ((s & 3) == wave) {                                    // the tile's 64 epilogue operands: one 256-byte DMA
        const uint32_t ldsb = lds0 + (uint32_t)(2 * CHUNK + (buf * S + s) * 64) * 4u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(AUX + (Tc + s) * 64 + lane), "s"(ldsb) : "memory", "m0");
      }
  };
  const int64_t colmin = T0 * 32, colmax = min(T1 * 32, I);
  // the line at window floats [lineoff, lineoff + 32) holds columns [cT - d_u, cT - d_u + 32) of row u
  auto emit = [&](int lineoff, int64_t cT, bool masked) {
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(Cw + (8 * i + (lane >> 3)) * CW + lineoff + 4 * (lane & 7));
    if (!masked) {
      const float* base = out + u0 * I + cT;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* p = base + ((int64_t)(8 * i + (lane >> 3)) * I - (int64_t)dr[i] + 4 * (lane & 7));
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v[i]) : "memory");
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = u0 + 8 * i + (lane >> 3), col0 = cT - (int64_t)dr[i] + 4 * (lane & 7);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (row < U && col0 + e >= colmin && col0 + e < colmax) out[row * I + col0 + e] = v[i][e];
      }
    }
  };
  dma(T0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (T0 + S < T1) dma(T0 + S, 1);
  // one chunk: S tiles out of operand buffer bpar; wp0 = window line of its first tile
  auto chunk = [&](int64_t Tc, const int bpar, const int wp0) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t T = Tc + s;
      if (s == 0 || T < T1) {
        const int wp = (wp0 + s) & 1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float4* bq = reinterpret_cast<const float4*>(sm + bpar * CHUNK + s * TILE) + h * 32 + c31;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float4 b = bq[q * 64];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[4 * q], b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[4 * q + 1], b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[4 * q + 2], b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[4 * q + 3], b.w, acc, 0, 0, 0);
        }
        const float bic = Ax[(bpar * S + s) * 64 + c31] + b0, prc = Ax[(bpar * S + s) * 64 + 32 + c31];
        if (s == S - 1 || T + 1 >= T1) {
          // the chunk's operand is consumed: when every wave is here its buffer is free for the chunk after the next, and the next
          // chunk (in flight since the previous barrier) has landed.  The stores this wait also covers were issued a tile ago.
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (!splitbar) {
            __builtin_amdgcn_s_barrier();
            if (Tc + 2 * S < T1) dma(Tc + 2 * S, bpar);
          } else {
            // split barrier: ARRIVE here (this wave is done with the chunk's operand and its pieces of the next chunk are in LDS), WAIT
            // after the tile's epilogue — the waves of a workgroup no longer stand at the barrier while they have an epilogue to do
            if (lane == 0) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            arrivals += 4;
          }
        }
        if (add) {                                               // second pass of an embedding size above 128 (rare: plain loads)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = u0 + (r & 3) + 8 * (r >> 2) + 4 * h, col = T * 32 + c31;
            if (row < U && col < I) acc[r] += out[row * I + col];
          }
        }
        if (kind == 2) {
          // v / prc with one divisor per column: r = 1 / prc once, q = v r, one residual step q + (v - q prc) r (= the correctly rounded
          // quotient's own correction step): 3 packed instructions per 2 values on the pipe the fp32 MFMA shares with the vector ALU
          const float rinv = 1.0f / prc;
          const f32x2 bic2 = {bic, bic}, rinv2 = {rinv, rinv}, nprc2 = {-prc, -prc};
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 v = {acc[r], acc[r + 1]};
            const f32x2 b2 = {bu[r], bu[r + 1]};
            v = v + b2 + bic2;
            const f32x2 q = v * rinv2;
            v = __builtin_elementwise_fma(__builtin_elementwise_fma(q, nprc2, v), rinv2, q);
            acc[r] = v[0];
            acc[r + 1] = v[1];
          }
        } else if (kind == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = acc[r] + bu[r] + bic;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<__attribute__((address_space(3))) float*>(wadr[wp][r]) = acc[r];
        const int64_t cT = T * 32;
        emit(wp * 32, cT, !(rows_full && T > T0 && cT + 32 <= colmax));
        if (splitbar && (s == S - 1 || T + 1 >= T1)) {
          while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < arrivals) __builtin_amdgcn_s_sleep(1);
          if (Tc + 2 * S < T1) dma(Tc + 2 * S, bpar);
        }
      }
    }
  };
  int64_t Tc = T0;
  for (; Tc + S < T1; Tc += 2 * S) {
    chunk(Tc, 0, 0);
    chunk(Tc + S, 1, S & 1);
  }
  if (Tc < T1) chunk(Tc, 0, 0);
  emit((int)((T1 - T0) & 1) * 32, T1 * 32, true);              // the rows' leftovers: columns [T1 * 32 - d_u, T1 * 32) below colmax
}

template <int D>
static int launch_full_lines(const mf_model_t* M, float* out, hipStream_t st, int koff, int kvalid, int kind, int add) {
  constexpr int S = 128 / D;
  const int64_t groups = (M->user_num + 127) / 128, gt = (M->item_num + 31) / 32;
  const int splits = 8;                       // = the XCDs: workgroups are dealt round-robin, XCD x only ever reads item range x
  ARG_CHECK(groups * splits < 2147483647LL, "matrix too large for one launch");
  ARG_CHECK((uintptr_t)out % 4 == 0, "out must be 4-byte aligned");
  const int64_t Tpad = gt + 2 * S;            // a chunk's DMA reads whole chunks
  float* QT4 = nullptr;
  HIP_TRY(hipMallocAsync((void**)&QT4, (size_t)Tpad * (D * 32 + 64) * sizeof(float), st));
  float* AUX = QT4 + Tpad * D * 32;
  hipLaunchKernelGGL(k_full_lines_prep, dim3((unsigned)min((int64_t)4096, (Tpad * D * 32 + 255) / 256)), dim3(256), 0, st, *M, D, Tpad, QT4, AUX,
                     koff, kvalid, kind);
  const size_t smem = (size_t)(2 * 4096 + 2 * S * 64 + 4 * 32 * 64 + 4) * sizeof(float);
  // (measured, 75 k x 64 k: D = 32 5.07 -> 4.82 ms, 64 7.18 -> 7.03, 128 11.47 -> 11.32, 16 4.30 -> 4.38: on from 32)
  const int splitbar = getenv("DCCF_FULL_SPLITBAR") ? atoi(getenv("DCCF_FULL_SPLITBAR")) : (D >= 32 ? 1 : 0);
  HIP_TRY(hipFuncSetAttribute((const void*)k_mf_full_lines<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((k_mf_full_lines<D>), dim3((unsigned)(groups * splits)), dim3(256), smem, st, *M, out, QT4, AUX, splits, koff, kvalid, kind, add, splitbar);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipFreeAsync(QT4, st));
  return 0;
}

template <int D, int RB, int NW = 8>
static int launch_full_rows(const mf_model_t* M, float* out, hipStream_t st) {
  const int64_t bands = (M->user_num + 32 * RB - 1) / (32 * RB), gt = (M->item_num + NW * 32 - 1) / (NW * 32);
  // item range in a multiple of 8 pieces, workgroup = (band, piece) with the piece index fastest: workgroups go round-robin
  // over the 8 XCDs, so XCD x only ever reads the pieces = x (mod 8) of Q^T — 1/8 of it (2 MB at D = 64) stays in that
  // XCD's 4 MB L2 while every band streams past it (measured: 10.5 -> 9.8 ms at D = 64 against one piece per band)
  int splits = gt >= 64 ? 32 : (gt >= 16 ? 8 : 1);
  if (getenv("DCCF_FULL_SPLITS")) splits = (int)max((int64_t)1, min(gt, (int64_t)atoi(getenv("DCCF_FULL_SPLITS"))));
  ARG_CHECK(bands * splits < 2147483647LL, "matrix too large for one launch");
  ARG_CHECK((uintptr_t)M->P % 16 == 0, "P must be 16-byte aligned");
  const int64_t Ipad = gt * NW * 32;
  float* QT = nullptr;
  HIP_TRY(hipMallocAsync((void**)&QT, (size_t)Ipad * D * sizeof(float), st));
  hipLaunchKernelGGL(k_transpose_q, dim3((unsigned)min((int64_t)4096, (Ipad * D + 255) / 256)), dim3(256), 0, st, M->Q, M->item_num, D,
                     Ipad, NW * 32, QT);
  int order = (splits % 8 == 0 && D >= 128) ? 1 : 0;
  if (getenv("DCCF_FULL_ORDER")) order = (atoi(getenv("DCCF_FULL_ORDER")) == 1 && splits % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((k_mf_full_rows<D, RB, NW>), dim3((unsigned)(bands * splits)), dim3(64 * NW), 0, st, *M, out, splits, QT, Ipad, bands,
                     order);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipFreeAsync(QT, st));
  return 0;
}

extern "C" int mf_predict_full(const mf_model_t* M, float* out, void* stream) {
  ARG_CHECK(M && out, "NULL argument");
  ARG_CHECK(M->P && M->Q && M->D >= 1 && M->D <= 256 && M->kind >= 0 && M->kind <= 2, "bad model (embedding sizes up to 256)");
  ARG_CHECK(M->kind == 0 || (M->bu && M->bi && M->b0), "bias pointers missing");
  ARG_CHECK(M->kind != 2 || M->prop, "propensity missing");
  ARG_CHECK(M->user_num > 0 && M->item_num > 0, "empty matrix");
  const char* form = getenv("DCCF_FULL_FORM");
  const bool old_form = form && (!strcmp(form, "rows") || !strcmp(form, "band") || !strcmp(form, "tile"));
  if (!old_form || M->D > 128 || M->D % 2) {
    // the line form (round 3: profiles/r03_full_matrix_bench.json) for EVERY embedding size (src/models/RecModel.py:17-27 accepts
    // any): the contraction is padded with zeros to the kernel's next tile (16, 32, 64, 128); sizes above 128 run as two launches —
    // the first writes the plain product of the columns [0, 128), the second continues it with the rest and applies the epilogue
    auto pass = [&](int koff, int kvalid, int kind, int add) -> int {
      if (kvalid <= 16) return launch_full_lines<16>(M, out, (hipStream_t)stream, koff, kvalid, kind, add);
      if (kvalid <= 32) return launch_full_lines<32>(M, out, (hipStream_t)stream, koff, kvalid, kind, add);
      if (kvalid <= 64) return launch_full_lines<64>(M, out, (hipStream_t)stream, koff, kvalid, kind, add);
      return launch_full_lines<128>(M, out, (hipStream_t)stream, koff, kvalid, kind, add);
    };
    if (M->D <= 128) return pass(0, M->D, M->kind, 0);
    if (int e = pass(0, 128, 0, 0)) return e;
    return pass(128, M->D - 128, M->kind, 1);
  }
  switch (M->D) {
    // forms per D as measured (CDs-shaped 75k x 64k, profiles/r02_full_matrix_bench.json): the row-band form (32 or 64 users
    // per workgroup, A operand in registers, tile-major Q^T, 1-KB row segments) for D <= 64: 5.1 / 6.0 / 9.1 ms at D = 16 /
    // 32 / 64 (round 1: 6.1 / 8.7 / 12.8 with the LDS-band form at D >= 32; rocBLAS sgemm without the epilogue: 7.1 / 9.3 ms
    // at D = 32 / 64); D = 128: 4 waves per workgroup (204 VGPRs), 14.9 ms against 22.0 with the LDS-band form
    case 16: return launch_full_rows<16, 1>(M, out, (hipStream_t)stream);
    case 32: return launch_full_rows<32, 2>(M, out, (hipStream_t)stream);
    case 64: return launch_full_rows<64, 1>(M, out, (hipStream_t)stream);
    case 128: {
      const char* f = getenv("DCCF_FULL_FORM");
      if (f && !strcmp(f, "band")) return launch_full_band<128, 128>(M, out, (hipStream_t)stream);      // 22.0 ms
      return launch_full_rows<128, 1, 4>(M, out, (hipStream_t)stream);      // 14.9 ms (8 waves: 17.9; rocBLAS product only: 12.5)
    }
    default: break;                            // other even D: the one-tile-per-workgroup form below
  }
  const int64_t nblk = ((M->item_num + FT - 1) / FT) * ((M->user_num + FT - 1) / FT);
  ARG_CHECK(nblk < 2147483647LL, "matrix too large for one launch");
  const dim3 grid((unsigned)((nblk + 7) / 8 * 8));
  const size_t smem = max((size_t)2 * FT * (M->D + 1) * 4, (size_t)FT * (FT + 1) * 4);
  HIP_TRY(hipFuncSetAttribute((const void*)k_mf_full, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(k_mf_full, grid, dim3(256), smem, (hipStream_t)stream, *M, out, nblk);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ train negatives
// One thread per user: its deg(u) train rows get distinct uniform negatives outside its train history (sorted CSR ->
// binary search) — the reference's per-epoch tmp_history_dict rule.  Draw j of user u is word j%4 of
// Philox(c0=u, c1=j/4, c2=epoch); oracle/philox.py::train_negatives restates it bit for bit.
__global__ __launch_bounds__(256) void k_sample_neg(const int64_t* __restrict__ rows_indptr,
                                                    const int64_t* __restrict__ rows,
                                                    const int64_t* __restrict__ hist_indptr,
                                                    const int64_t* __restrict__ hist_items, int64_t user_num,
                                                    int64_t item_num, rng_key key, int64_t* __restrict__ neg_out) {
  for (int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; u < user_num; u += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r0 = rows_indptr[u], r1 = rows_indptr[u + 1];
    const int64_t h0 = hist_indptr[u], h1 = hist_indptr[u + 1];
    uint32_t j = 0;
    u32x4 cur{0, 0, 0, 0};
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t remain = item_num - (h1 - h0) - (r - r0);
      const bool low = 5 * remain < item_num;   // remain / item_num < 0.2
      int64_t it = -1;
      // what can still be drawn: in the low regime the reference samples from range(1, item_num) (DataProcessor.py:490-493),
      // so item 0 does not count unless it is already excluded (history is sorted: 0 can only sit first)
      int64_t admissible = remain;
      if (low) {
        bool zero_out = h1 > h0 && hist_items[h0] == 0;
        for (int64_t q = r0; q < r && !zero_out; ++q) zero_out = neg_out[rows[q]] == 0;
        if (!zero_out) admissible -= 1;
      }
      // nothing left: the reference asserts (DataProcessor.py:495) or raises from np.random.choice; the host checks for -1
      if (admissible < 1) { neg_out[rows[r]] = -1; continue; }
      bool found = false;
      for (uint32_t tries = 0; tries < (1u << 22); ++tries) {     // every thread has an exit (P(miss) < e^-800 when reached)
        if ((j & 3) == 0) cur = philox4x32_10((uint32_t)u, j >> 2, key.s0, key.s1, key.k0, key.k1);
        it = (int64_t)(((uint64_t)pick4(cur, j & 3) * (uint64_t)item_num) >> 32);
        ++j;
        if (low && it == 0) continue;
        int64_t lo = h0, hi = h1;   // binary search in the sorted history
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (hist_items[mid] < it) lo = mid + 1; else hi = mid;
        }
        if (lo < h1 && hist_items[lo] == it) continue;
        bool dup = false;
        for (int64_t q = r0; q < r; ++q)
          if (neg_out[rows[q]] == it) { dup = true; break; }
        if (!dup) { found = true; break; }
      }
      neg_out[rows[r]] = found ? it : -1;
    }
  }
}

extern "C" int dccf_sample_train_negatives(const int64_t* rows_indptr, const int64_t* rows, const int64_t* hist_indptr,
                                           const int64_t* hist_items, int64_t user_num, int64_t item_num, uint64_t seed,
                                           uint64_t epoch, int64_t* neg_out, void* stream) {
  ARG_CHECK(rows_indptr && rows && hist_indptr && hist_items && neg_out, "NULL argument");
  ARG_CHECK(user_num > 0 && item_num > 0 && item_num < 4294967296LL, "bad user_num / item_num");
  const int grid = (int)min((int64_t)4096, (user_num + 255) / 256);
  rng_key key = make_key(seed, STREAM_NEG, 0);
  key.s0 = (uint32_t)epoch;
  key.s1 = 0;
  hipLaunchKernelGGL(k_sample_neg, dim3(grid), dim3(256), 0, (hipStream_t)stream, rows_indptr, rows, hist_indptr,
                     hist_items, user_num, item_num, key, neg_out);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ epoch batches
// The feed dicts of an epoch (src/data_processor/DataProcessor.py:160-207,227-250) as ONE tensor: with the epoch's
// permutation `perm` (the in-unison shuffle, src/utils/utils.py:82-92) batch k is X = [pos rows ; neg rows],
// pos row j = (uid, iid)[perm[k*B + j]], neg row j = (uid, neg)[perm[k*B + j]] — rows j and B + j carry the same uid.  The
// n % B rows left over form the shorter last batch `tail` [2r, 2].  A negative of -1 (a user with nothing left to draw,
// dccf_sample_train_negatives) is replaced by 0 and reported through *bad: no id below 0 ever reaches a kernel as a row index.
// perm == NULL: the epoch's permutation is computed on the fly — a keyed bijection of [0, 2^b) (b = bits of n - 1, at least 2): a
// 12-round alternating Feistel network (halves of ceil(b/2) / floor(b/2) bits, round function = splitmix64's finalizer of the right
// half + the round key), walked until it lands below n (cycle walking: fewer than two passes on average).  It plays
// shuffle_in_unison_scary's role (src/utils/utils.py:82-92) without a sort: no permutation array, no extra launches
// (torch.randperm is a key sort of five launches, more than a 20-step epoch can hide).  oracle/philox.py::epoch_perm restates it
// bit for bit and holds it to a chi-square on small domains (round 2's multiply-add-xorshift rounds failed that test).
#define EPOCH_PERM_ROUNDS 12
struct PermKeys { uint64_t k[EPOCH_PERM_ROUNDS]; };
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ int64_t epoch_perm(int64_t i, int64_t n, int b, const PermKeys& ks) {
  const int lb = b >> 1, hb = b - lb;
  uint64_t x = (uint64_t)i;
  do {
    int wl = hb, wr = lb;
    uint64_t L = x >> wr, R = x & ((1ull << wr) - 1);
#pragma unroll
    for (int r = 0; r < EPOCH_PERM_ROUNDS; ++r) {
      const uint64_t F = mix64(R + ks.k[r]);
      const uint64_t nr = L ^ (F & ((1ull << wl) - 1));
      L = R;
      R = nr;
      const int t = wl; wl = wr; wr = t;
    }
    x = (L << wr) | R;
  } while (x >= (uint64_t)n);
  return (int64_t)x;
}

__global__ __launch_bounds__(256) void k_epoch_batches(const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
                                                       const int64_t* __restrict__ neg, const int64_t* __restrict__ perm, int64_t n,
                                                       int64_t B, int64_t* __restrict__ full, int64_t* __restrict__ tail,
                                                       int32_t* __restrict__ bad, int pbits, PermKeys pk) {
  const int64_t nb = n / B, r = n - nb * B;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = perm ? perm[i] : epoch_perm(i, n, pbits, pk);
    const int64_t u = uid[s], ip = iid[s];
    int64_t ng = neg[s];
    if (ng < 0) {
      *bad = 1;
      ng = 0;
    }
    const int64_t k = i / B, j = i - k * B;
    int64_t* pos_row = k < nb ? full + (k * 2 * B + j) * 2 : tail + j * 2;
    int64_t* neg_row = k < nb ? full + (k * 2 * B + B + j) * 2 : tail + (r + j) * 2;
    pos_row[0] = u; pos_row[1] = ip;
    neg_row[0] = u; neg_row[1] = ng;
  }
}

extern "C" int dccf_build_epoch_batches(const int64_t* uid, const int64_t* iid, const int64_t* neg, const int64_t* perm, int64_t n,
                                        int64_t batch_size, int64_t* full, int64_t* tail, int32_t* bad, uint64_t seed, uint64_t epoch,
                                        void* stream) {
  ARG_CHECK(n >= 0 && batch_size > 0 && bad, "bad arguments");
  if (n == 0) return 0;
  ARG_CHECK(uid && iid && neg && (n < batch_size || full) && (n % batch_size == 0 || tail), "NULL argument");
  const int grid = (int)min((int64_t)2048, (n + 255) / 256);
  int b = 2;
  while (b < 62 && (1ll << b) < n) ++b;
  // the round keys from (seed, epoch): splitmix64 steps
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + epoch * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
  PermKeys pk;
  for (int i = 0; i < EPOCH_PERM_ROUNDS; ++i) {
    z += 0x9E3779B97F4A7C15ull;
    uint64_t x = z;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    pk.k[i] = x ^ (x >> 31);
  }
  hipLaunchKernelGGL(k_epoch_batches, dim3(grid), dim3(256), 0, (hipStream_t)stream, uid, iid, neg, perm, n, batch_size, full, tail, bad,
                     b, pk);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------ eval negatives
// neg_n negatives per DISTINCT user of an eval split (src/data_processor/DataProcessor.py:408-444, first-occurrence rule
// :420-426): uniform over the items, outside the user's train + validation/test history (sorted CSR -> binary search) and
// distinct among themselves (:446-524, train=False).  Draw j of user u is word j%4 of Philox(c0=u, c1=j/4, c2=tag) on
// STREAM_EVALNEG; draws are consumed IN ORDER: draw j is accepted iff its item is admissible and no earlier accepted draw
// has it; the first neg_n accepted draws, in draw order, are the result (oracle/philox.py::eval_negatives restates it).
// One wave per user, 64 draws per round; a 4096-slot LDS hash table per wave holds the accepted items, `owner` the
// lowest draw index that proposed each item, so duplicates inside a round resolve to the earliest draw.
#define EN_SLOTS 4096
__global__ __launch_bounds__(256) void k_sample_eval_neg(const int64_t* __restrict__ users, int64_t n_users,
                                                         const int64_t* __restrict__ hist_indptr,
                                                         const int64_t* __restrict__ hist_items, int64_t item_num, int neg_n,
                                                         rng_key key, int64_t* __restrict__ out) {
  extern __shared__ int en_lds[];                 // [4 waves][keys EN_SLOTS | owner EN_SLOTS]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* keys = en_lds + wave * 2 * EN_SLOTS;
  int* owner = keys + EN_SLOTS;
  for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < n_users; w += (int64_t)gridDim.x * 4) {
    const int64_t u = users[w];
    const int64_t h0 = hist_indptr[u], h1 = hist_indptr[u + 1];
    for (int i = lane; i < EN_SLOTS; i += 64) { keys[i] = -1; owner[i] = 0x7fffffff; }
    __builtin_amdgcn_wave_barrier();
    int64_t remain = item_num - (h1 - h0);
    const bool low = 5 * remain < item_num;       // remain / item_num < 0.2: the reference then never draws item 0 (:490-493)
    if (low) {                                    // ... so item 0 does not count as available
      int64_t lo = h0, hi = h1;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (hist_items[mid] < 0) lo = mid + 1; else hi = mid; }
      if (!(lo < h1 && hist_items[lo] == 0)) remain -= 1;
    }
    if (remain < neg_n) {                         // reference asserts (:488)
      for (int i = lane; i < neg_n; i += 64) out[w * neg_n + i] = -1;
      continue;
    }
    int need = neg_n, outpos = 0;
    for (uint32_t j0 = 0; need > 0 && j0 < (1u << 24); j0 += 64) {      // the round cap is an exit every wave reaches
      const uint32_t j = j0 + lane;
      const u32x4 r = philox4x32_10((uint32_t)u, j >> 2, key.s0, key.s1, key.k0, key.k1);
      const int c = (int)(((uint64_t)pick4(r, j & 3) * (uint64_t)item_num) >> 32);
      bool ok = !(low && c == 0);
      if (ok) {
        int64_t lo = h0, hi = h1;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (hist_items[mid] < c) lo = mid + 1; else hi = mid; }
        ok = !(lo < h1 && hist_items[lo] == c);
      }
      int slot = 0;
      if (ok) {
        uint32_t h = ((uint32_t)c * 2654435761u) >> 20;          // 12 bits
        for (;;) {
          const int k = keys[h];
          if (k == c) break;
          if (k == -1) {
            const int old = atomicCAS(&keys[h], -1, c);
            if (old == -1 || old == c) break;
          }
          h = (h + 1) & (EN_SLOTS - 1);
        }
        slot = (int)h;
        atomicMin(&owner[slot], (int)j);
      }
      __builtin_amdgcn_wave_barrier();
      const bool acc = ok && owner[slot] == (int)j;               // items accepted in earlier rounds have owner < j0
      const uint64_t bal = __ballot(acc);
      const int rank = __popcll(bal & ((1ull << lane) - 1));
      if (acc && rank < need) out[w * neg_n + outpos + rank] = c;
      const int cnt = min((int)__popcll(bal), need);
      outpos += cnt;
      need -= cnt;
      __builtin_amdgcn_wave_barrier();
    }
    if (need > 0)
      for (int i = lane; i < need; i += 64) out[w * neg_n + outpos + i] = -1;
  }
}

extern "C" int dccf_sample_eval_negatives(const int64_t* users, int64_t n_users, const int64_t* hist_indptr,
                                          const int64_t* hist_items, int64_t item_num, int32_t neg_n, uint64_t seed,
                                          uint64_t tag, int64_t* out, void* stream) {
  ARG_CHECK(n_users >= 0 && (n_users == 0 || (users && hist_indptr && hist_items && out)), "NULL argument");
  ARG_CHECK(item_num > 0 && item_num < 2147483647LL, "bad item_num");
  ARG_CHECK(neg_n >= 1 && neg_n <= EN_SLOTS / 2, "neg_n must be in [1, 2048]");
  if (n_users == 0) return 0;
  const size_t smem = (size_t)4 * 2 * EN_SLOTS * sizeof(int);       // 128 KB
  static bool once = false;
  if (!once) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_sample_eval_neg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    once = true;
  }
  rng_key key = make_key(seed, STREAM_EVALNEG, 0);
  key.s0 = (uint32_t)tag;
  key.s1 = 0;
  const int grid = (int)min((int64_t)1024, (n_users + 3) / 4);
  hipLaunchKernelGGL(k_sample_eval_neg, dim3(grid), dim3(256), smem, (hipStream_t)stream, users, n_users, hist_indptr,
                     hist_items, item_num, neg_n, key, out);
  HIP_TRY(hipGetLastError());
  return 0;
}
