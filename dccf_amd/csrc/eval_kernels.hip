// eval_kernels.hip — ranking metrics on the device: replaces the sort_values / groupby('uid') / per-group Python loops of
// BaseModel.evaluate_method (src/models/BaseModel.py:83-126) and dcg_at_k / ndcg_at_k(method=1) / precision_at_k
// (src/utils/rank_metrics.py:61-87,130-201).
//
// One wave per user.  The user's candidate rows come from a CSR built once per eval split (the eval sets are fixed for
// a run, src/data_processor/DataProcessor.py:73-111).  Every lane keeps the best K scores of the rows it walked (K <= 16,
// insertion into a small sorted register array); K rounds of a wave-wide arg-max over the lane heads then pop the
// user's global top-K in order.  Ties are broken by the lower position in the user's row list (the reference's
// DataFrame.sort_values is an unstable quicksort, so its tie order is arbitrary).
#include "common.hpp"

#define EVAL_KMAX 16          // ranks per round (the lane-local sorted list)
#define EVAL_K_LIMIT 1024     // largest cut-off

// out[u][j][0..3] = ndcg@k_j, hit@k_j, precision@k_j, recall@k_j ; out[u][nk][0] = number of positives
__global__ __launch_bounds__(256) void k_rank_eval(const float* __restrict__ pred, const float* __restrict__ label,
                                                   const int64_t* __restrict__ indptr, const int64_t* __restrict__ rows,
                                                   int64_t n_users, const int* __restrict__ ks, int nk, int kmax,
                                                   float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t u = wave; u < n_users; u += nw) {
    const int64_t r0 = indptr[u], r1 = indptr[u + 1];
    float best[EVAL_KMAX];
    int bpos[EVAL_KMAX];
    float npos = 0.f;
    float dcg[4] = {0.f, 0.f, 0.f, 0.f}, hits[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t cnt = r1 - r0;
    // cut-offs beyond EVAL_KMAX: rounds of EVAL_KMAX pops; a round only admits rows that come strictly AFTER the last popped
    // (score, position) in the ranking order, so round j continues the list where round j - 1 stopped (one more walk of the
    // user's rows per 16 ranks — the common k <= 16 is a single round, as before)
    float fs = INFINITY;
    int fp = -1;
    bool done = false;
    for (int t0 = 0; t0 < kmax && t0 < cnt && !done; t0 += EVAL_KMAX) {
      const int kd = min(kmax - t0, EVAL_KMAX);
#pragma unroll
      for (int j = 0; j < EVAL_KMAX; ++j) { best[j] = -INFINITY; bpos[j] = 0x7fffffff; }
      for (int64_t r = r0 + lane; r < r1; r += 64) {
        const int64_t row = rows[r];
        float s = pred[row];
        int p = (int)(r - r0);
        if (t0 == 0 && label[row] > 0.f) npos += 1.f;
        if (t0 > 0 && !(s < fs || (s == fs && p > fp))) continue;
        // insertion into the lane's sorted top list (descending score, ascending position on ties)
#pragma unroll
        for (int j = 0; j < EVAL_KMAX; ++j) {
          if (j < kd) {
            const bool better = s > best[j] || (s == best[j] && p < bpos[j]);
            const float ts = better ? best[j] : s;
            const int tp = better ? bpos[j] : p;
            best[j] = better ? s : best[j];
            bpos[j] = better ? p : bpos[j];
            s = ts;
            p = tp;
          }
        }
      }
      // pop the next kd ranks: arg-max over the lane heads
      for (int t = t0; t < t0 + kd; ++t) {
        float hs = best[0];
        int hp = bpos[0];
        float ms = hs;
        int mp = hp;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float os = __shfl_xor(ms, o, 64);
          const int op = __shfl_xor(mp, o, 64);
          const bool take = os > ms || (os == ms && op < mp);
          ms = take ? os : ms;
          mp = take ? op : mp;
        }
        // fewer candidates than k: the reference's lists are just shorter (a NaN score is never listed)
        if (t >= cnt || mp == 0x7fffffff) { done = true; break; }
        if (hs == ms && hp == mp) {                // the winning lane pops its head
#pragma unroll
          for (int j = 0; j < EVAL_KMAX - 1; ++j) { best[j] = best[j + 1]; bpos[j] = bpos[j + 1]; }
          best[EVAL_KMAX - 1] = -INFINITY;
          bpos[EVAL_KMAX - 1] = 0x7fffffff;
        }
        fs = ms;
        fp = mp;
        const float l = label[rows[r0 + mp]];
        const float disc = 1.f / log2f((float)(t + 2));
        for (int j = 0; j < nk; ++j)
          if (t < ks[j]) {
            dcg[j] += l * disc;
            hits[j] += l;
          }
      }
    }
    npos = wave_sum(npos);
    if (lane == 0) {
      for (int j = 0; j < nk; ++j) {
        // ideal DCG of binary labels: the positives first (ndcg_at_k sorts the labels descending)
        float idcg = 0.f;
        const int ideal = (int)fminf(npos, (float)ks[j]);
        for (int t = 0; t < ideal; ++t) idcg += 1.f / log2f((float)(t + 2));
        float* o = out + (u * (nk + 1) + j) * 4;
        o[0] = idcg > 0.f ? dcg[j] / idcg : 0.f;
        o[1] = hits[j] > 0.f ? 1.f : 0.f;
        o[2] = hits[j] / (float)ks[j];
        o[3] = hits[j] / npos;        // 0/0 = NaN for a user without positives, as in the reference
      }
      out[(u * (nk + 1) + nk) * 4] = npos;
    }
  }
}

// ks_host: HOST array of the cut-offs (each in [1, EVAL_K_LIMIT]); ks_dev: the same values on the device
extern "C" int rank_eval_topk(const float* pred, const float* label, const int64_t* indptr, const int64_t* rows,
                                int64_t n_users, const int32_t* ks_host, const int32_t* ks_dev, int32_t nk, float* out,
                                void* stream) {
  ARG_CHECK(ks_host && ks_dev, "NULL argument");
  ARG_CHECK(n_users == 0 || (pred && label && indptr && rows && out), "NULL argument");
  ARG_CHECK(n_users >= 0 && nk >= 1 && nk <= 4, "1..4 cut-offs");
  int kmax = 0;
  for (int j = 0; j < nk; ++j) {
    ARG_CHECK(ks_host[j] >= 1 && ks_host[j] <= EVAL_K_LIMIT, "cut-offs must be in [1, 1024]");
    kmax = ks_host[j] > kmax ? ks_host[j] : kmax;
  }
  if (n_users == 0) return 0;
  const int grid = (int)min((int64_t)4096, (n_users + 3) / 4);
  hipLaunchKernelGGL(k_rank_eval, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, label, indptr, rows, n_users, ks_dev,
                     nk, kmax, out);
  HIP_TRY(hipGetLastError());
  return 0;
}
