// dp_kernels.hip — gradient exchange of the replicated data-parallel path (dccf_amd/replicated.py): every GPU keeps the
// whole model (16.4 M parameters + the 48.5 GB exposure matrix fit one MI355X many times over), trains its own pairs, and
// the ranks exchange only what a step produced: the few thousand embedding rows its batch touched and the dense [dW|db].
//
//   dp_export_touched   compacts (row id, gradient row) of every row whose "touched" byte is set into this rank's slot of
//                       the all-gather buffer, zeroes those rows of g and clears the bytes; copies (and zeroes) the dense
//                       tail [dW | db] and the loss.
//   dp_import_touched   after the all-gather: g[row] = sum over ranks IN RANK ORDER of the received rows (so that every
//                       replica computes bit-identical sums and the replicas never drift apart), sets the "touched" bytes,
//                       dense tail = sum over ranks in rank order.  No atomics on data: a per-row bit mask of the
//                       contributing ranks + a where-table make the rank-ordered walk possible.
//   dccf_dp_local / dccf_dp_overlap / dccf_dp_finish   one step in three calls around the collective.  With a
//                       dccf_dp_next_t every step prepares the next inside its optimizer launches (candidates, W^T, the
//                       rows any rank will touch, the mask / where tables of the next import): the backward then writes
//                       its gradient rows straight into the buffer slots the tables name ("slot mode": no export pass, no
//                       ids in the payload, no table pass after the wait) and a step is forward, backward | all-gather ||
//                       untouched-row optimizer pass | rank-ordered sums + optimizer.
//
// Buffer of one rank (32-bit words): [count | loss | pad pad | ids int64[cap] | rows fp32[cap][D] | dense fp32[nd]].
#include "common.hpp"
#include "opt_device.hpp"

#define DP_HDR 4
#define KEEPI(x) asm volatile("" ::"v"(x))
#define DP_GMAX 16

struct DpLay {
  int64_t cap, nd, ids_off, rows_off, dense_off, words;
  int D;
};
static DpLay dp_layout(int64_t cap, int D, int64_t nd) {
  DpLay y;
  y.cap = cap; y.D = D; y.nd = nd;
  y.ids_off = DP_HDR;
  y.rows_off = y.ids_off + 2 * cap;
  y.dense_off = y.rows_off + cap * D;
  y.words = (y.dense_off + nd + 3) / 4 * 4;
  return y;
}

extern "C" int64_t dp_buffer_words(int64_t cap, int32_t D, int64_t nd) { return dp_layout(cap, D, nd).words; }

// ---------------------------------------------------------------------------------------------- global marking
// Which rows will ANY rank's batch touch this step?  Known without communication: the schedule X_all [G][N][2] is
// replicated and rank r's candidates are the Philox stream of (seed, step0 + r) — the same draws k_prep makes on rank r
// (models/DCCF.py:72-74).  Rows outside this set see only the l2 term, so their optimizer pass does not have to wait for
// the exchange.  Runs as its own launch (dp_mark_global) or as extra workgroups of the export kernel (dccf_dp_local).
struct MarkGlobal {
  const int64_t* X_all;
  int G, S;
  int64_t N, item_num;
  rng_key key0;
  MarkPlan mp;
};

__device__ __forceinline__ void dp_mark_rows(const MarkGlobal& mg, int64_t tid, int64_t nthreads) {
  const int S1 = mg.S + 1;
  const int64_t per = mg.N * S1, total = (int64_t)mg.G * per;
  for (int64_t i = tid; i < total; i += nthreads) {
    const int r = (int)(i / per);
    const int64_t j = i % per, n = j / S1;
    const int s = (int)(j % S1);
    const int64_t* X = mg.X_all + (int64_t)r * mg.N * 2;
    int64_t it;
    if (s == 0) {
      it = X[2 * n + 1];
    } else {
      const rng_key key = key_plus(mg.key0, r);
      const u32x4 rr = philox4x32_10((uint32_t)n, (uint32_t)((s - 1) >> 2), key.s0, key.s1, key.k0, key.k1);
      it = (int64_t)(((uint64_t)pick4(rr, (s - 1) & 3) * (uint64_t)mg.item_num) >> 32);
    }
    if (mg.mp.list) {
      mark_row(mg.mp.flagV, it, mg.mp.tagV, mg.mp);
      if (s == 0) mark_row(mg.mp.flagU, X[2 * n], mg.mp.tagU, mg.mp);
    } else {             // nobody needs the rows as a list (dccf_dp_local): plain byte stores
      reinterpret_cast<uint8_t*>(mg.mp.flagV)[it] = 1;
      if (s == 0) reinterpret_cast<uint8_t*>(mg.mp.flagU)[X[2 * n]] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void k_dp_mark(MarkGlobal mg) {
  if (mg.mp.cnt_next && blockIdx.x == 0 && threadIdx.x == 0) *mg.mp.cnt_next = 0;      // (bytes only: no list, no counters)
  dp_mark_rows(mg, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

static int make_mark_global(const int64_t* X_all, int32_t G, int64_t N, int32_t S, int64_t item_num, uint64_t seed, uint64_t step0,
                            uint8_t* flagsU, uint8_t* flagsV, int32_t segU, int32_t segV, int64_t* list, int32_t* cnt,
                            int32_t* cnt_next, MarkGlobal* mg) {
  ARG_CHECK(X_all && flagsU && flagsV && ((list && cnt && cnt_next) || (!list && !cnt && !cnt_next)), "NULL argument");
  ARG_CHECK(G >= 1 && N >= 1 && S >= 0 && item_num > 0, "bad sizes");
  ARG_CHECK((uintptr_t)flagsU % 4 == 0 && (uintptr_t)flagsV % 4 == 0, "flags must be 4-byte aligned (padded to whole words)");
  mg->X_all = X_all; mg->G = G; mg->S = S; mg->N = N; mg->item_num = item_num;
  mg->key0 = make_key(seed, STREAM_CAND, step0);
  mg->mp.flagU = (uint32_t*)flagsU;
  mg->mp.flagV = (uint32_t*)flagsV;
  mg->mp.tagU = (int64_t)segU << 40;
  mg->mp.tagV = (int64_t)segV << 40;
  mg->mp.list = list;
  mg->mp.cnt = cnt;
  mg->mp.cnt_next = cnt_next;
  return 0;
}

// ---------------------------------------------------------------------------------------------- export
// A wave looks at 256 consecutive flag bytes of one segment (one 32-bit word per lane): a wave-wide prefix sum places its
// set bytes behind ONE atomicAdd on the buffer's counter, then 16-lane groups ship four rows at a time (float4 columns).
__global__ __launch_bounds__(256) void k_dp_export(float* __restrict__ g, RowSegs sg, int64_t dense_begin, const float* loss,
                                                   float* __restrict__ buf, DpLay y, int scan_blocks, int work_blocks,
                                                   MarkGlobal mg) {
  __shared__ int wl[4][256];                  // per wave: the rows found in the current chunk
  if ((int)blockIdx.x >= work_blocks) {       // global marking role (independent of the export: different flag arrays)
    const int mb = blockIdx.x - work_blocks;
    if (mb == 0 && threadIdx.x == 0 && mg.mp.cnt_next) *mg.mp.cnt_next = 0;
    dp_mark_rows(mg, mb * (int64_t)blockDim.x + threadIdx.x, (int64_t)(gridDim.x - work_blocks) * blockDim.x);
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* count = reinterpret_cast<int*>(buf);
  int64_t* ids = reinterpret_cast<int64_t*>(buf + y.ids_off);
  float* rows = buf + y.rows_off;
  if ((int)blockIdx.x >= scan_blocks) {       // dense tail + loss
    const int64_t tid = (int64_t)(blockIdx.x - scan_blocks) * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)(work_blocks - scan_blocks) * blockDim.x;
    for (int64_t i = tid; i < y.nd; i += stride) {
      buf[y.dense_off + i] = g[dense_begin + i];
      g[dense_begin + i] = 0.f;
    }
    if (tid == 0) buf[1] = loss ? loss[0] : 0.f;
    return;
  }
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)scan_blocks * blockDim.x) >> 6;
  const int d4 = y.D >> 2, grp = lane >> 4, sub = lane & 15;
  for (int q = 0; q < sg.n; ++q) {
    const int64_t nrows = (sg.end[q] - sg.begin[q]) / sg.width[q];
    const int64_t nwords = (nrows + 3) / 4;          // flag arrays are padded to whole words (zeros)
    uint32_t* fw = reinterpret_cast<uint32_t*>(sg.flags[q]);
    for (int64_t w0 = wave * 64; w0 < nwords; w0 += nw * 64) {
      const int64_t wi = w0 + lane;
      const uint32_t word = wi < nwords ? fw[wi] : 0u;
      if (__ballot(word != 0) == 0) continue;
      int mine = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) mine += ((word >> (8 * b)) & 0xffu) != 0;
      int incl = mine;                               // inclusive prefix sum over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int total = __shfl(incl, 63, 64);
      int base = 0;
      if (lane == 0) base = atomicAdd(count, total);
      base = __shfl(base, 0, 64);
      int k = incl - mine;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (((word >> (8 * b)) & 0xffu) != 0) {
          const int64_t row = wi * 4 + b;
          wl[wv][k] = (int)(row - w0 * 4);
          if (base + k < y.cap) ids[base + k] = ((int64_t)q << 40) | row;
          ++k;
        }
      if (word != 0) fw[wi] = 0u;                    // the bytes are consumed
      __builtin_amdgcn_wave_barrier();
      for (int t0 = 0; t0 < total; t0 += 4) {
        const int t = t0 + grp;
        if (t < total && base + t < y.cap) {
          const int64_t row = w0 * 4 + wl[wv][t];
          float4* grow = reinterpret_cast<float4*>(g + sg.begin[q] + row * sg.width[q]);
          float4* dst = reinterpret_cast<float4*>(rows + (int64_t)(base + t) * y.D);
          for (int c = sub; c < d4; c += 16) {
            dst[c] = grow[c];
            grow[c] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// The same export when the step's rows are already known as a de-duplicated list (a prepared step: the previous step's
// optimizer launch drew this step's candidates and listed its rows, dccf_dp_overlap): one 16-lane group per list entry,
// no flag scan, no atomics.  The order of the entries is the (arbitrary) order of the list — the import does not care.
__global__ __launch_bounds__(256) void k_dp_export_list(float* __restrict__ g, RowSegs sg, int64_t dense_begin, const float* loss,
                                                        float* __restrict__ buf, DpLay y, int row_blocks,
                                                        const int64_t* __restrict__ list, const int* __restrict__ cnt,
                                                        uint8_t* lfU, uint8_t* lfV, int segU, const int* __restrict__ where_r,
                                                        int64_t offU, int64_t offV) {
  if ((int)blockIdx.x >= row_blocks) {       // dense tail + loss + the entry count
    const int64_t tid = (int64_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)(gridDim.x - row_blocks) * blockDim.x;
    for (int64_t i = tid; i < y.nd; i += stride) {
      buf[y.dense_off + i] = g[dense_begin + i];
      g[dense_begin + i] = 0.f;
    }
    if (tid == 0) {
      buf[1] = loss ? loss[0] : 0.f;
      reinterpret_cast<int*>(buf)[0] = where_r ? (int)y.cap : min(*cnt, (int)y.cap);   // table slots are sparse
    }
    return;
  }
  const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
  const int t = blockIdx.x * 16 + grp;
  const int n = min(*cnt, (int)y.cap);
  if (t >= n) return;
  const int64_t id = list[t];
  const int q = (int)(id >> 40);
  const int64_t row = id & ((1LL << 40) - 1);
  // slot: the list position, or — when the import tables were built ahead — the slot every rank expects the row in
  const int slot = where_r ? where_r[(q == segU ? offU : offV) + row] : t;
  if (slot < 0 || slot >= (int)y.cap) return;      // (cannot happen: the table was built from the same draws as the list)
  float4* grow = reinterpret_cast<float4*>(g + sg.begin[q] + row * y.D);
  float4* dst = reinterpret_cast<float4*>(buf + y.rows_off + (int64_t)slot * y.D);
  const int d4 = y.D >> 2;
  for (int c = sub; c < d4; c += 16) {
    dst[c] = grow[c];
    grow[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (sub == 0) {
    reinterpret_cast<int64_t*>(buf + y.ids_off)[slot] = id;
    (q == segU ? lfU : lfV)[row] = 0;              // the de-duplication mark is consumed
    if (sg.flags[q]) sg.flags[q][row] = 0;         // and the byte the backward set
  }
}

static int dp_export_impl(float* g, int64_t n, int32_t nseg, const int64_t* seg_begin, const int64_t* seg_rows,
                          const int32_t* seg_width, uint8_t* const* seg_flags, int64_t dense_begin, const float* loss,
                          float* buf, int64_t cap, int32_t D, int32_t reset, const MarkGlobal* mgp, void* stream) {
  ARG_CHECK(g && buf && nseg >= 1 && nseg <= 4 && seg_begin && seg_rows && seg_width && seg_flags, "NULL / bad segments");
  ARG_CHECK(cap >= 1 && dense_begin >= 0 && dense_begin <= n, "bad cap / dense_begin");
  RowSegs sg;
  memset(&sg, 0, sizeof(sg));
  sg.n = nseg;
  int64_t max_rows = 0;
  for (int q = 0; q < nseg; ++q) {
    ARG_CHECK(seg_width[q] == D, "every row segment must have width D");
    ARG_CHECK(seg_flags[q] && (uintptr_t)seg_flags[q] % 4 == 0, "flags must be 4-byte aligned (and padded to whole words)");
    ARG_CHECK(seg_begin[q] + seg_rows[q] * D <= dense_begin, "row segments must lie below dense_begin");
    sg.begin[q] = seg_begin[q];
    sg.end[q] = seg_begin[q] + seg_rows[q] * D;
    sg.width[q] = D;
    sg.flags[q] = seg_flags[q];
    max_rows = max(max_rows, seg_rows[q]);
  }
  const DpLay y = dp_layout(cap, D, n - dense_begin);
  hipStream_t st = (hipStream_t)stream;
  ARG_CHECK(D % 4 == 0, "D must be a multiple of 4");
  if (reset) HIP_TRY(hipMemsetAsync(buf, 0, DP_HDR * sizeof(float), st));     // else: dp_import_touched(reset_buf) did it
  const int scan_blocks = (int)max((int64_t)1, min((int64_t)1024, (max_rows / 4 + 255) / 256));
  const int dense_blocks = (int)max((int64_t)1, min((int64_t)256, (y.nd + 255) / 256));
  MarkGlobal mg;
  memset(&mg, 0, sizeof(mg));
  int mark_blocks = 0;
  if (mgp) {
    mg = *mgp;
    mark_blocks = (int)min((int64_t)256, ((int64_t)mg.G * mg.N * (mg.S + 1) + 255) / 256);
  }
  hipLaunchKernelGGL(k_dp_export, dim3(scan_blocks + dense_blocks + mark_blocks), dim3(256), 0, st, g, sg, dense_begin, loss, buf,
                     y, scan_blocks, scan_blocks + dense_blocks, mg);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dp_export_touched(float* g, int64_t n, int32_t nseg, const int64_t* seg_begin, const int64_t* seg_rows,
                                 const int32_t* seg_width, uint8_t* const* seg_flags, int64_t dense_begin, const float* loss,
                                 float* buf, int64_t cap, int32_t D, int32_t reset, void* stream) {
  return dp_export_impl(g, n, nseg, seg_begin, seg_rows, seg_width, seg_flags, dense_begin, loss, buf, cap, D, reset, nullptr,
                        stream);
}

// ---------------------------------------------------------------------------------------------- import
// Two launches.  (A) one thread per received entry (r, e): records where rank r keeps destination row x
// (where[r][x] = e) and sets bit r of mask[x].  (B) a 16-lane group per entry; the entry of the lowest contributing rank
// walks the other set bits of mask[x] in ASCENDING rank order and sums the rows — the order of the floating-point
// additions is the same on every replica — stores the sum (the row was zero: exported rows are zeroed), sets the
// "touched" byte and clears mask[x] for the next step.  Extra workgroups of (B) sum the dense tails and losses likewise.
struct DpRows {
  int64_t begin[4];      // element offset of segment q in g
  int64_t rowoff[4];     // first global row index of segment q
  uint8_t* flags[4];
  int n;
};

__global__ __launch_bounds__(256) void k_dp_scatter_ids(const float* __restrict__ bufs, int G, DpLay y, DpRows sg, int64_t R,
                                                        uint32_t* __restrict__ mask, int* __restrict__ where) {
  __shared__ int scnt[DP_GMAX];
  if (threadIdx.x < G) scnt[threadIdx.x] = min((int64_t)reinterpret_cast<const int*>(bufs + (int64_t)threadIdx.x * y.words)[0], y.cap);
  __syncthreads();
  const int cap = (int)y.cap, total = G * cap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / cap;
    const int e = i - r * cap;
    const float* b = bufs + (int64_t)r * y.words;
    if (e < scnt[r]) {
      const int64_t id = reinterpret_cast<const int64_t*>(b + y.ids_off)[e];
      const int64_t gid = sg.rowoff[(int)(id >> 40)] + (id & ((1LL << 40) - 1));
      where[(int64_t)r * R + gid] = e;
      atomicOr(&mask[gid], 1u << r);
    }
  }
}

// One 16-lane group per received entry (r, e).  The entry of the LOWEST contributing rank owns the destination row: it adds
// the other ranks' rows in ascending rank order, stores the sum, sets the byte and clears the mask (a later reader of a
// cleared mask is a non-owner and skips, as it would have anyway).
#ifdef DCCF_TRACE
__device__ long long dp_trace[16];
#define DPT(k) if (blockIdx.x == 0 && threadIdx.x == 0) dp_trace[k] = wall_clock64()
extern "C" int dp_debug_trace_read(long long* out) {
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(dp_trace), sizeof(long long) * 16));
  return 0;
}
#else
#define DPT(k)
#endif
// APPLY < 0: the sums go to g (+ "touched" bytes set) and a dense optimizer pass follows.  APPLY = optimizer kind: the
// optimizer is applied right here to the summed rows and the dense tail (phase 2 of the two-phase step: every row a rank
// touched has an entry, so this covers exactly the rows the untouched-row phase skipped) — g stays zero, the bytes of
// `sg.flags` (the global marks) are cleared.
template <int APPLY>
__device__ __forceinline__ void dp_apply4(const OptJob& j, int64_t i4, float4 gsum) {
  constexpr int K = APPLY < 0 ? 0 : APPLY;
  float4 pv = reinterpret_cast<float4*>(j.p)[i4];
  float4 av = make_float4(0, 0, 0, 0), bv = make_float4(0, 0, 0, 0);
  if (K != DCCF_OPT_GD) av = reinterpret_cast<float4*>(j.s1)[i4];
  if (K == DCCF_OPT_ADAM) bv = reinterpret_cast<float4*>(j.s2)[i4];
  opt_elem<K>(pv.x, gsum.x, av.x, bv.x, j.a);
  opt_elem<K>(pv.y, gsum.y, av.y, bv.y, j.a);
  opt_elem<K>(pv.z, gsum.z, av.z, bv.z, j.a);
  opt_elem<K>(pv.w, gsum.w, av.w, bv.w, j.a);
  reinterpret_cast<float4*>(j.p)[i4] = pv;
  if (K != DCCF_OPT_GD) reinterpret_cast<float4*>(j.s1)[i4] = av;
  if (K == DCCF_OPT_ADAM) reinterpret_cast<float4*>(j.s2)[i4] = bv;
}

// where the applying form mirrors the updated W (inside the dense tail) for the next step's forward (prepared step)
struct WMirror {
  float* WT;                 // NULL: nothing to mirror
  int64_t w_begin, w_end;    // elements of the flat buffer
  int KF, DP;
  int* cnt_reset;            // the local-list counter this step consumed (zeroed for the step after the next)
  int reset_where;           // tables built ahead (atomicMin): the owner puts the consumed where entries back to INT_MAX
  // slot mode (the backward wrote straight into the buffer slots, no ids travelled): slot e of rank r holds the row at
  // position e of rank r's canonical order — recomputed from the replicated schedule and the candidate stream
  const int64_t* X_all;      // NULL: read the ids from the buffers
  rng_key key0;
  int64_t N, NS, item_num;
  int S1;
  int segU, segV;
  // the local send buffer: its rows / dense tail are zeroed once summed (the next backward adds into them)
  float* local_buf;
  int self;
  // lazy regularisation: the applied rows are at step lazy_t afterwards
  int* lazy_last;
  int lazy_t;
};

template <int APPLY>
__global__ __launch_bounds__(256) void k_dp_sum_rows(const float* __restrict__ bufs, int G, float* __restrict__ g, DpRows sg,
                                                     int64_t dense_begin, float* __restrict__ loss_sum, DpLay y, int64_t R,
                                                     uint32_t* __restrict__ mask, int* where,
                                                     int row_blocks, float* reset_buf, OptJob job, WMirror wm) {
  if ((int)blockIdx.x >= row_blocks) {           // dense tail: sum in rank order
    const int64_t tid = (int64_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)(gridDim.x - row_blocks) * blockDim.x;
    for (int64_t i = tid; i < y.nd; i += stride) {
      float s = bufs[y.dense_off + i];
      for (int r = 1; r < G; ++r) s += bufs[(int64_t)r * y.words + y.dense_off + i];
      if (wm.local_buf) wm.local_buf[y.dense_off + i] = 0.f;
      if (APPLY < 0) {
        g[dense_begin + i] = s;
      } else {
        constexpr int K = APPLY < 0 ? 0 : APPLY;
        float pv = job.p[dense_begin + i], av = 0.f, bv = 0.f;
        if (K != DCCF_OPT_GD) av = job.s1[dense_begin + i];
        if (K == DCCF_OPT_ADAM) bv = job.s2[dense_begin + i];
        opt_elem<K>(pv, s, av, bv, job.a);
        job.p[dense_begin + i] = pv;
        if (wm.WT && dense_begin + i >= wm.w_begin && dense_begin + i < wm.w_end) {   // W^T for the next step's forward
          const int64_t idx = dense_begin + i - wm.w_begin;
          wm.WT[(idx % wm.KF) * wm.DP + idx / wm.KF] = pv;
        }
        if (K != DCCF_OPT_GD) job.s1[dense_begin + i] = av;
        if (K == DCCF_OPT_ADAM) job.s2[dense_begin + i] = bv;
      }
    }
    if (tid == 0 && reset_buf) reinterpret_cast<int*>(reset_buf)[0] = 0;      // the local export buffer's counter
    if (tid == 0 && wm.cnt_reset) *wm.cnt_reset = 0;
    if (tid == 0 && loss_sum) {
      float s = bufs[1];
      for (int r = 1; r < G; ++r) s += bufs[(int64_t)r * y.words + 1];
      loss_sum[0] = s;
    }
    return;
  }
  DPT(0);
  const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;       // 16 lanes x float4 per row
  const int d4 = y.D >> 2;
  const int cap = (int)y.cap, total = G * cap;              // < 2^31 (checked on the host): 32-bit index arithmetic
  const int t = blockIdx.x * 16 + grp;                      // one group per entry, no loop
  if (t >= total) return;
  {
    const int r = t / cap;
    const int e = t - r * cap;
    const float* b = bufs + (int64_t)r * y.words;
    DPT(1);
    // No entry count is read here (a header word read by every group of the grid is a hot spot): an entry is live iff
    // this step's scatter pass registered exactly it — bit r of the row's mask set and where[r][row] == e.  Slots past a
    // rank's count hold ids of earlier steps (or zeros) and fail that test.
    if (wm.local_buf && r == wm.self && sub < d4) {         // (the sums below read the gathered copy, not this buffer)
      float4* z = reinterpret_cast<float4*>(wm.local_buf + y.rows_off + (int64_t)e * y.D);
      z[sub] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sub + 16 < d4) z[sub + 16] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int q;
    int64_t row;
    if (wm.X_all) {
      const int64_t* X = wm.X_all + (int64_t)r * wm.N * 2;
      if (e < wm.NS) {
        const int64_t n = e / wm.S1;
        const int s1 = (int)(e - n * wm.S1);
        q = wm.segV;
        if (s1 == 0) {
          row = X[2 * n + 1];
        } else {
          const rng_key key = key_plus(wm.key0, r);
          const u32x4 rr = philox4x32_10((uint32_t)n, (uint32_t)((s1 - 1) >> 2), key.s0, key.s1, key.k0, key.k1);
          row = (int64_t)(((uint64_t)pick4(rr, (s1 - 1) & 3) * (uint64_t)wm.item_num) >> 32);
        }
      } else if (e < wm.NS + wm.N) {
        q = wm.segU;
        row = X[2 * (e - wm.NS)];
      } else {
        return;
      }
    } else {
      const int64_t id = reinterpret_cast<const int64_t*>(b + y.ids_off)[e];
      q = (int)(id >> 40);
      row = id & ((1LL << 40) - 1);
    }
    DPT(2);
    if (q >= sg.n) return;
    const int64_t gid = sg.rowoff[q] + row;
    uint32_t m = mask[gid];
    const int we = where[(int64_t)r * R + gid];
    DPT(3);
    if (m == 0u || (__ffs(m) - 1) != r || we != e || sub >= d4) return;   // stale slot, or not the owner of this row
    // D <= 128: two float4 per lane (columns sub and sub + 16), in named registers — an indexed array here ends up in LDS
    const bool two = sub + 16 < d4;
    float4 a0, a1 = make_float4(0, 0, 0, 0);
    {
      const float4* src = reinterpret_cast<const float4*>(b + y.rows_off + (int64_t)e * y.D);
      a0 = src[sub];
      if (two) a1 = src[sub + 16];
    }
    if (wm.reset_where && sub == 0) where[(int64_t)r * R + gid] = 0x7fffffff;
    m &= m - 1;
    while (m) {
      const int r2 = __ffs(m) - 1;
      m &= m - 1;
      const int e2 = where[(int64_t)r2 * R + gid];
      if (wm.reset_where && sub == 0) where[(int64_t)r2 * R + gid] = 0x7fffffff;
      const float4* src = reinterpret_cast<const float4*>(bufs + (int64_t)r2 * y.words + y.rows_off + (int64_t)e2 * y.D);
      const float4 v0 = src[sub];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      if (two) {
        const float4 v1 = src[sub + 16];
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      }
    }
    DPT(4);
    const int64_t i4 = (sg.begin[q] + row * y.D) >> 2;
    if (APPLY < 0) {
      float4* dst = reinterpret_cast<float4*>(g) + i4;
      dst[sub] = a0;
      if (two) dst[sub + 16] = a1;
    } else {
      dp_apply4<APPLY>(job, i4 + sub, a0);
      if (two) dp_apply4<APPLY>(job, i4 + sub + 16, a1);
    }
    if (sub == 0) {
      sg.flags[q][row] = APPLY < 0 ? 1 : 0;
      mask[gid] = 0u;
      if (APPLY >= 0 && wm.lazy_last) wm.lazy_last[gid] = wm.lazy_t;
    }
    DPT(5);
  }
  DPT(6);
}

static int dp_import_impl(const float* bufs, int32_t G, float* g, int64_t n, int32_t nseg, const int64_t* seg_begin,
                          const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags, int64_t dense_begin,
                          float* loss_sum, int64_t cap, int32_t D, uint32_t* mask, int32_t* where, float* reset_buf,
                          const OptJob* job, hipStream_t st, const WMirror* wmp = nullptr, bool scatter = true) {
  ARG_CHECK(bufs && (g || job) && G >= 1 && G <= DP_GMAX, "1..16 ranks");
  ARG_CHECK(nseg >= 1 && nseg <= 4 && seg_begin && seg_rows && seg_width && seg_flags, "bad segments");
  ARG_CHECK(D >= 4 && D <= 128 && D % 4 == 0 && cap >= 1 && dense_begin >= 0 && dense_begin <= n, "bad D / cap / dense_begin");
  ARG_CHECK(mask && where, "NULL scratch");
  ARG_CHECK((int64_t)G * cap < 2147483647LL, "G * cap must be < 2^31");
  DpRows sg;
  memset(&sg, 0, sizeof(sg));
  sg.n = nseg;
  int64_t R = 0;
  for (int q = 0; q < nseg; ++q) {
    ARG_CHECK(seg_width[q] == D && seg_flags[q], "every row segment must have width D and flags");
    ARG_CHECK(seg_begin[q] % 4 == 0, "segments must start on a 16-byte boundary");
    sg.begin[q] = seg_begin[q];
    sg.rowoff[q] = R;
    sg.flags[q] = seg_flags[q];
    R += seg_rows[q];
  }
  const DpLay y = dp_layout(cap, D, n - dense_begin);
  const int64_t total = (int64_t)G * cap;
  if (scatter)       // (a prepared step's tables were built one step ahead, inside the optimizer launch)
    hipLaunchKernelGGL(k_dp_scatter_ids, dim3((unsigned)min((int64_t)1024, (total + 255) / 256)), dim3(256), 0, st, bufs, G, y,
                       sg, R, mask, where);
  const int row_blocks = (int)((total + 15) / 16);          // one 16-lane group per entry
  const int dense_blocks = (int)max((int64_t)1, min((int64_t)256, (y.nd + 255) / 256));
  const dim3 grid(row_blocks + dense_blocks);
  OptJob none;
  memset(&none, 0, sizeof(none));
  WMirror wm;
  memset(&wm, 0, sizeof(wm));
  if (wmp) wm = *wmp;
#define DP_SUM(APPLY_, JOB_)                                                                                          \
  hipLaunchKernelGGL(k_dp_sum_rows<APPLY_>, grid, dim3(256), 0, st, bufs, G, g, sg, dense_begin, loss_sum, y, R, mask, where, \
                     row_blocks, reset_buf, JOB_, wm)
  if (!job) DP_SUM(-1, none);
  else if (job->kind == DCCF_OPT_GD) DP_SUM(DCCF_OPT_GD, *job);
  else if (job->kind == DCCF_OPT_ADAGRAD) DP_SUM(DCCF_OPT_ADAGRAD, *job);
  else DP_SUM(DCCF_OPT_ADAM, *job);
#undef DP_SUM
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int dp_import_touched(const float* bufs, int32_t G, float* g, int64_t n, int32_t nseg, const int64_t* seg_begin,
                                 const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags,
                                 int64_t dense_begin, float* loss_sum, int64_t cap, int32_t D, uint32_t* mask, int32_t* where,
                                 float* reset_buf, void* stream) {
  ARG_CHECK(g != nullptr, "g is NULL");
  return dp_import_impl(bufs, G, g, n, nseg, seg_begin, seg_rows, seg_width, seg_flags, dense_begin, loss_sum, cap, D, mask,
                        where, reset_buf, nullptr, (hipStream_t)stream);
}

extern "C" int dp_import_apply(const float* bufs, int32_t G, int32_t kind, float* p, float* s1, float* s2, int64_t n, float lr,
                               float wd, float l2, float clip, int64_t step, int32_t nseg, const int64_t* seg_begin,
                               const int64_t* seg_rows, const int32_t* seg_width, uint8_t* const* seg_flags,
                               int64_t dense_begin, float* loss_sum, int64_t cap, int32_t D, uint32_t* mask, int32_t* where,
                               float* reset_buf, void* stream) {
  OptJob job;
  // (g is not used by the applying form: pass p as a stand-in for the validation of the job)
  if (int e = opt_make_job(kind, p, p, s1, s2, n, lr, wd, l2, clip, step, nullptr, nseg, seg_begin, seg_rows, seg_width,
                           seg_flags, &job))
    return e;
  return dp_import_impl(bufs, G, nullptr, n, nseg, seg_begin, seg_rows, seg_width, seg_flags, dense_begin, loss_sum, cap, D, mask,
                        where, reset_buf, &job, (hipStream_t)stream);
}

extern "C" int dp_mark_global(const int64_t* X_all, int32_t G, int64_t N, int32_t S, int64_t item_num, uint64_t seed,
                              uint64_t step0, uint8_t* flagsU, uint8_t* flagsV, int32_t segU, int32_t segV, int64_t* list,
                              int32_t* cnt, int32_t* cnt_next, void* stream) {
  MarkGlobal mg;
  if (int e = make_mark_global(X_all, G, N, S, item_num, seed, step0, flagsU, flagsV, segU, segV, list, cnt, cnt_next, &mg)) return e;
  const int64_t total = (int64_t)G * N * (S + 1);
  const int grid = (int)min((int64_t)1024, (total + 255) / 256);
  hipLaunchKernelGGL(k_dp_mark, dim3(grid), dim3(256), 0, (hipStream_t)stream, mg);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------- one step in three calls
static uint8_t* dp_gflags(const dccf_dp_t* dp, int parity, int which) {      // double-buffered "touched by ANY rank" bytes
  uint8_t* a = which == 0 ? dp->gflagsU : dp->gflagsV;
  uint8_t* b = which == 0 ? dp->gflagsU2 : dp->gflagsV2;
  return (parity && b) ? b : a;
}
static size_t pad4(int64_t n) { return (size_t)((n + 3) / 4 * 4); }

// A prepared step that is not going to run as announced: its marks and list must not leak into later steps.
static int dp_discard_prepared(dccf_ctx* ctx, const dccf_model_t* M, const dccf_dp_t* dp, hipStream_t st) {
  if (ctx->prep_dp || ctx->prep_pending) {
    const int par = ctx->prep_parity;
    HIP_TRY(hipMemsetAsync(dp_gflags(dp, par, 0), 0, pad4(M->user_num), st));
    HIP_TRY(hipMemsetAsync(dp_gflags(dp, par, 1), 0, pad4(M->item_num), st));
    if (dp->lflagsU) HIP_TRY(hipMemsetAsync(dp->lflagsU, 0, pad4(M->user_num), st));
    if (dp->lflagsV) HIP_TRY(hipMemsetAsync(dp->lflagsV, 0, pad4(M->item_num), st));
    if (dp->lcnt) HIP_TRY(hipMemsetAsync(dp->lcnt, 0, 2 * sizeof(int32_t), st));
    if (ctx->prep_tables && dp->pmask && dp->pwhere) {
      const int64_t R = M->user_num + M->item_num;
      HIP_TRY(hipMemsetAsync(dp->pmask + (int64_t)par * R, 0, (size_t)R * 4, st));
      HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(dp->pwhere + (int64_t)par * dp->G * R), 0x7fffffff, (size_t)dp->G * R, st));
    }
    ctx->prep_valid = 0;
  }
  ctx->prep_dp = ctx->prep_pending = ctx->prep_tables = 0;
  return 0;
}

// the import tables built ahead need the two-segment layout [users | items] that R = user_num + item_num assumes
static bool dp_tables_usable(const dccf_opt_t* opt, const dccf_dp_t* dp, const dccf_model_t* M) {
  return dp->ctx && dp->pmask && dp->pwhere && opt->nseg == 2 && opt->seg_rows[dp->segU] == M->user_num &&
         opt->seg_rows[dp->segV] == M->item_num;
}
static int64_t dp_rowoff(const dccf_opt_t* opt, int seg) {
  int64_t o = 0;
  for (int q = 0; q < seg; ++q) o += opt->seg_rows[q];
  return o;
}

extern "C" int dccf_dp_local(dccf_ctx* ctx, const dccf_model_t* model, const dccf_rand_t* rnd, const int64_t* X, const float* Y,
                             int64_t N, float dropout, const dccf_grads_t* grads, const dccf_opt_t* opt, const dccf_dp_t* dp,
                             const int64_t* X_all, uint64_t step0, int32_t parity, float* prediction, void* stream) {
  ARG_CHECK(ctx && model && rnd && opt && dp && dp->loss && dp->buf, "NULL ctx / model / rnd / opt / dp");
  ARG_CHECK(parity == 0 || parity == 1, "parity must be 0 or 1");
  hipStream_t st = (hipStream_t)stream;
  // prepared by the previous step's dccf_dp_overlap + dccf_dp_finish: slots and W^T (-> no k_prep), this rank's rows as a
  // list (-> list export), every rank's rows marked in the flag set of this parity (-> no marking role)
  const bool prepared = X_all && ctx->prep_dp && !ctx->prep_pending && ctx->prep_parity == parity &&
                        ctx->prep_Xall == (const void*)X_all && dccf_prep_matches(ctx, model, rnd, X, N);
  if (!prepared) {
    if (int e = dp_discard_prepared(ctx, model, dp, st)) return e;
  }
  const bool lazy = opt->lazy_K > 0;
  bool marked_early = false;
  if (lazy) {
    ARG_CHECK(X_all != nullptr, "the lazy optimizer needs the replicated schedule (X_all) in every step");
    // rows ANY rank touches this step: known as bytes when the previous step prepared this one -> claimed and brought up to
    // step - 1 before the forward reads them; otherwise the whole table is brought up to date first, and the rows are claimed
    // once the export launch below has marked them
    // (phase 1 of the previous step does the claiming and the catch-up itself when it knows this step: pn.cu_blocks)
    const bool mine = ctx->lazy_prep_step == (int64_t)opt->step && ctx->lazy_prep_claim == (const void*)opt->lazy_claim &&
                      ctx->lazy_prep_id == opt->lazy_id;
    if (mine && !prepared)
      if (int e = dccf_lazy_reset_claims(opt, st)) return e;
    ctx->lazy_prep_step = -1;
    if (prepared) {
      if (!mine)
        if (int e = dccf_lazy_catchup_flags(opt, dp_gflags(dp, parity, 0), dp_gflags(dp, parity, 1), dp->segU, dp->segV, st)) return e;
    } else {
      // a step nobody announced (the first of an epoch): the replicated schedule still names the rows ANY rank touches — they
      // are marked first, by a launch of their own, and only they are claimed and brought up to step - 1 (the catch-up reads the
      // pending window's marks itself); bringing the whole table up to date instead cost ~70 us (DCCF_DP_FLUSH_UNPREPARED=1)
      static const bool flush_all = getenv("DCCF_DP_FLUSH_UNPREPARED") != nullptr;
      if (flush_all) {
        if (int e = dccf_lazy_flush_to_prev(opt, st)) return e;
      } else {
        if (int e = dp_mark_global(X_all, dp->G, N, dp->S, dp->item_num, dp->seed, step0, dp_gflags(dp, parity, 0),
                                   dp_gflags(dp, parity, 1), dp->segU, dp->segV, nullptr, nullptr, nullptr, stream))
          return e;
        if (int e = dccf_lazy_catchup_flags(opt, dp_gflags(dp, parity, 0), dp_gflags(dp, parity, 1), dp->segU, dp->segV, st)) return e;
        marked_early = true;
      }
    }
  }
  ctx->prep_dp = 0;
  ctx->cur_tables = prepared && ctx->prep_tables && dp->ctx == ctx && dp_tables_usable(opt, dp, model);
  ctx->prep_tables = 0;
  ctx->cur_Xall = X_all;
  ctx->cur_step0 = step0;
  ctx->cur_N = N;
  if (ctx->cur_tables) {
    // slot mode: the backward adds the embedding gradient rows, [dW | db] and the loss straight into this rank's all-gather
    // buffer (slot of a row = what the tables built one step ahead say) — no export pass, g is not touched
    const DpLay y = dp_layout(dp->cap, dp->D, opt->n - dp->dense_begin);
    const int64_t R = model->user_num + model->item_num;
    const int64_t oW = grads->gW - opt->g, ob = grads->gb - opt->g;
    ARG_CHECK(oW >= dp->dense_begin && ob >= dp->dense_begin && oW < opt->n && ob < opt->n, "gW / gb must lie in the dense tail of g");
    dccf_grads_t gs = *grads;
    gs.gW = dp->buf + y.dense_off + (oW - dp->dense_begin);
    gs.gb = dp->buf + y.dense_off + (ob - dp->dense_begin);
    for (int k = 0; k < model->n_extra; ++k) {        // --n_layers > 1: the extra layers' gradients are part of the dense tail
      const int64_t oWk = grads->gWl[k] - opt->g, obk = grads->gbl[k] - opt->g;
      ARG_CHECK(grads->gWl[k] && grads->gbl[k] && oWk >= dp->dense_begin && obk >= dp->dense_begin && oWk < opt->n && obk < opt->n,
                "the gradients of the extra mlp layers must lie in the dense tail of g");
      gs.gWl[k] = dp->buf + y.dense_off + (oWk - dp->dense_begin);
      gs.gbl[k] = dp->buf + y.dense_off + (obk - dp->dense_begin);
    }
    gs.touchedU = gs.touchedV = nullptr;
    ctx->slot_where = dp->pwhere + ((int64_t)parity * dp->G + dp->rank) * R;
    ctx->slot_rows = dp->buf + y.rows_off;
    ctx->slot_cap = (int)dp->cap;
    ctx->slot_offU = dp_rowoff(opt, dp->segU);
    ctx->slot_offV = dp_rowoff(opt, dp->segV);
    const int e = dccf_train_fwdbwd(ctx, model, rnd, X, Y, N, 1, dropout, &gs, prediction, dp->buf + 1, stream);
    ctx->slot_where = nullptr;
    ctx->slot_rows = nullptr;
    return e;
  }
  if (int e = dccf_train_fwdbwd(ctx, model, rnd, X, Y, N, 1, dropout, grads, prediction, dp->loss, stream)) return e;
  if (prepared) {
    ARG_CHECK(opt->nseg >= 2 && opt->nseg <= 4, "bad segments");
    RowSegs sg;
    memset(&sg, 0, sizeof(sg));
    sg.n = opt->nseg;
    for (int q = 0; q < opt->nseg; ++q) {
      ARG_CHECK(opt->seg_width[q] == dp->D, "every row segment must have width D");
      sg.begin[q] = opt->seg_begin[q];
      sg.end[q] = opt->seg_begin[q] + opt->seg_rows[q] * dp->D;
      sg.width[q] = dp->D;
      sg.flags[q] = opt->seg_flags[q];
    }
    const DpLay y = dp_layout(dp->cap, dp->D, opt->n - dp->dense_begin);
    const int row_blocks = (int)((dp->cap + 15) / 16);
    const int dense_blocks = (int)max((int64_t)1, min((int64_t)256, (y.nd + 255) / 256));
    const int64_t R = model->user_num + model->item_num;
    const int* where_r = ctx->cur_tables ? dp->pwhere + ((int64_t)parity * dp->G + dp->rank) * R : nullptr;
    hipLaunchKernelGGL(k_dp_export_list, dim3(row_blocks + dense_blocks), dim3(256), 0, st, opt->g, sg, dp->dense_begin, dp->loss,
                       dp->buf, y, row_blocks, dp->llist, dp->lcnt + parity, dp->lflagsU, dp->lflagsV, dp->segU, where_r,
                       dp_rowoff(opt, dp->segU), dp_rowoff(opt, dp->segV));
    HIP_TRY(hipGetLastError());
    return 0;
  }
  MarkGlobal mg;
  if (X_all && !marked_early) {     // overlap mode: the global marking rides in the export launch
    // (bytes only: the merged import needs no list of the marked rows, so dp->glist / dp->gcnt stay untouched)
    if (int e = make_mark_global(X_all, dp->G, N, dp->S, dp->item_num, dp->seed, step0, dp_gflags(dp, parity, 0),
                                 dp_gflags(dp, parity, 1), dp->segU, dp->segV, nullptr, nullptr, nullptr, &mg))
      return e;
  }
  if (int e = dp_export_impl(opt->g, opt->n, opt->nseg, opt->seg_begin, opt->seg_rows, opt->seg_width, opt->seg_flags,
                             dp->dense_begin, dp->loss, dp->buf, dp->cap, dp->D, 0, (X_all && !marked_early) ? &mg : nullptr, stream))
    return e;
  if (lazy && !marked_early) return dccf_lazy_catchup_flags(opt, dp_gflags(dp, parity, 0), dp_gflags(dp, parity, 1), dp->segU, dp->segV, st);
  return 0;
}

static int dp_global_flags(const dccf_opt_t* opt, const dccf_dp_t* dp, int parity, uint8_t** out) {
  ARG_CHECK(opt->nseg >= 2 && opt->nseg <= 4 && dp->segU >= 0 && dp->segU < opt->nseg && dp->segV >= 0 && dp->segV < opt->nseg &&
                dp->segU != dp->segV && dp->gflagsU && dp->gflagsV,
            "bad global-flag segments");
  ARG_CHECK(parity == 0 || parity == 1, "parity must be 0 or 1");
  for (int q = 0; q < opt->nseg; ++q) out[q] = opt->seg_flags[q];
  out[dp->segU] = dp_gflags(dp, parity, 0);
  out[dp->segV] = dp_gflags(dp, parity, 1);
  return 0;
}

static bool dp_next_usable(const dccf_opt_t* opt, const dccf_dp_t* dp, const dccf_dp_next_t* nx) {
  if (!nx || !nx->ctx || !nx->model || !nx->X_next || !nx->X_all_next || nx->N <= 0) return false;
  if (!dp->gflagsU2 || !dp->gflagsV2 || !dp->lflagsU || !dp->lflagsV || !dp->llist || !dp->lcnt) return false;
  const dccf_model_t* M = nx->model;
  return opt->p <= M->W && M->W + (int64_t)M->D * (M->D + M->F) <= opt->p + opt->n && (M->W - opt->p) >= dp->dense_begin &&
         nx->N * (M->S + 2) <= dp->cap;
}

extern "C" int dccf_dp_overlap(const dccf_opt_t* opt, const dccf_dp_t* dp, int32_t parity, const dccf_dp_next_t* next,
                               void* stream) {
  ARG_CHECK(opt && dp, "NULL opt / dp");
  uint8_t* gf[4];
  if (int e = dp_global_flags(opt, dp, parity, gf)) return e;
  if (opt->lazy_K > 0 && !dp_next_usable(opt, dp, next)) return dccf_lazy_phase1(opt, nullptr, (hipStream_t)stream);
  if (!dp_next_usable(opt, dp, next))
    return dccf_dense_opt_phase(opt->kind, opt->p, opt->g, opt->s1, opt->s2, opt->n, opt->lr, opt->wd, opt->l2, opt->clip, opt->step,
                                opt->nseg, opt->seg_begin, opt->seg_rows, opt->seg_width, gf, 1, nullptr, nullptr, 0, stream);
  // the pass over the rows no rank touches + the next step's preparation, in one launch
  dccf_ctx* ctx = next->ctx;
  const dccf_model_t* M = next->model;
  PrepNext pn;
  if (int e = dccf_prep_next_fill(ctx, M, next->N, next->X_next, dp->seed, next->step0_next + (uint64_t)dp->rank, &pn)) return e;
  const int64_t NS = next->N * (M->S + 1);
  pn.blocks = (int)min((int64_t)256, ((int64_t)dp->G * NS + 255) / 256);
  pn.lm.flagU = (uint32_t*)dp->lflagsU;
  pn.lm.flagV = (uint32_t*)dp->lflagsV;
  pn.lm.tagU = (int64_t)dp->segU << 40;
  pn.lm.tagV = (int64_t)dp->segV << 40;
  pn.lm.list = dp->llist;
  pn.lm.cnt = dp->lcnt + (1 - parity);
  pn.lm.cnt_next = nullptr;
  pn.X_all = next->X_all_next;
  pn.gU = dp_gflags(dp, 1 - parity, 0);
  pn.gV = dp_gflags(dp, 1 - parity, 1);
  pn.G = dp->G;
  pn.gkey0 = make_key(dp->seed, STREAM_CAND, next->step0_next);
  ctx->prep_tables = 0;
  if (dp->ctx == ctx && dp_tables_usable(opt, dp, M) && next->X_next == next->X_all_next + (int64_t)dp->rank * next->N * 2) {
    pn.lm.list = nullptr;            // slot mode: no export pass, hence no list of this rank's rows
    pn.R = M->user_num + M->item_num;
    pn.nmask = dp->pmask + (int64_t)(1 - parity) * pn.R;
    pn.nwhere = dp->pwhere + (int64_t)(1 - parity) * dp->G * pn.R;
    pn.offU = dp_rowoff(opt, dp->segU);
    pn.offV = dp_rowoff(opt, dp->segV);
    ctx->prep_tables = 1;
  }
  ctx->prep_valid = 0;
  ctx->prep_dp = 0;
  ctx->prep_pending = 1;             // committed by dccf_dp_finish (W^T is written there)
  ctx->prep_parity = 1 - parity;
  if (opt->lazy_K > 0) {
    // the rows ANY rank touches at the next step are claimed and caught up here too, while the all-gather is in flight
    const int64_t slots = (int64_t)dp->G * next->N * (M->S + 2);
    static const int cu_cap = getenv("DCCF_DP_CU_BLOCKS") ? atoi(getenv("DCCF_DP_CU_BLOCKS")) : 1024;      // 0: the flag-scan catch-up instead
    pn.cu_blocks = cu_cap > 0 ? (int)max((int64_t)1, min((int64_t)cu_cap, (slots + 3) / 4)) : 0;
    pn.cu_segU = dp->segU;
    pn.cu_segV = dp->segV;
    if (int e = dccf_lazy_phase1(opt, &pn, (hipStream_t)stream)) return e;
    if (pn.cu_blocks == 0) return 0;
    ctx->lazy_prep_step = (int64_t)opt->step + 1;
    ctx->lazy_prep_claim = opt->lazy_claim;
    ctx->lazy_prep_id = opt->lazy_id;
    return 0;
  }
  return dccf_opt_untouched_prep(opt, gf, &pn, (hipStream_t)stream);
}

extern "C" int dccf_dp_finish(const dccf_opt_t* opt, const dccf_dp_t* dp, int32_t overlap, int32_t parity,
                              const dccf_dp_next_t* next, void* stream) {
  ARG_CHECK(opt && dp && dp->bufs && dp->mask && dp->where, "NULL opt / dp");
  if (overlap) {
    uint8_t* gf[4];
    if (int e = dp_global_flags(opt, dp, parity, gf)) return e;
    OptJob job;
    if (int e = opt_make_job(opt->kind, opt->p, opt->p, opt->s1, opt->s2, opt->n, opt->lr, opt->wd, opt->l2, opt->clip, opt->step,
                             nullptr, opt->nseg, opt->seg_begin, opt->seg_rows, opt->seg_width, gf, &job))
      return e;
    WMirror wm;
    memset(&wm, 0, sizeof(wm));
    wm.cnt_reset = dp->lcnt ? dp->lcnt + parity : nullptr;
    dccf_ctx* ctx = next ? next->ctx : nullptr;
    const bool commit = ctx && ctx->prep_pending && ctx->prep_parity == 1 - parity && dp_next_usable(opt, dp, next);
    if (commit) {
      const dccf_model_t* M = next->model;
      PrepNext pn;       // (for the workspace address of W^T; the slab cannot move: same N as in dccf_dp_overlap)
      if (int e = dccf_prep_next_fill(ctx, M, next->N, next->X_next, dp->seed, next->step0_next + (uint64_t)dp->rank, &pn)) return e;
      wm.WT = pn.WT;
      wm.w_begin = M->W - opt->p;
      wm.w_end = wm.w_begin + (int64_t)M->D * (M->D + M->F);
      wm.KF = M->D + M->F;
      wm.DP = pn.DP;
    } else if (ctx && ctx->prep_pending) {
      if (int e = dp_discard_prepared(ctx, next->model, dp, (hipStream_t)stream)) return e;
    }
    // this step's import tables: built one step ahead (no scatter pass) or by k_dp_scatter_ids from the received ids
    uint32_t* mask = dp->mask;
    int32_t* where = dp->where;
    bool scatter = true;
    wm.local_buf = dp->buf;          // rows and dense tail of the send buffer are zero again after every step
    wm.self = dp->rank;
    if (opt->lazy_K > 0) {
      wm.lazy_last = opt->lazy_last;
      wm.lazy_t = (int)opt->step;
    }
    if (dp->ctx && dp->ctx->cur_tables) {
      dccf_ctx* ctx = dp->ctx;
      const int64_t R = opt->seg_rows[0] + opt->seg_rows[1];
      mask = dp->pmask + (int64_t)parity * R;
      where = dp->pwhere + (int64_t)parity * dp->G * R;
      wm.reset_where = 1;
      scatter = false;
      wm.X_all = (const int64_t*)ctx->cur_Xall;
      wm.key0 = make_key(dp->seed, STREAM_CAND, ctx->cur_step0);
      wm.N = ctx->cur_N;
      wm.S1 = dp->S + 1;
      wm.NS = ctx->cur_N * (dp->S + 1);
      wm.item_num = dp->item_num;
      wm.segU = dp->segU;
      wm.segV = dp->segV;
      ctx->cur_tables = 0;
    }
    if (int e = dp_import_impl(dp->bufs, dp->G, nullptr, opt->n, opt->nseg, opt->seg_begin, opt->seg_rows, opt->seg_width, gf,
                               dp->dense_begin, dp->loss_sum, dp->cap, dp->D, mask, where, dp->buf, &job,
                               (hipStream_t)stream, &wm, scatter))
      return e;
    if (commit) {
      dccf_prep_next_commit(ctx, next->model, next->N, next->X_next, dp->seed, next->step0_next + (uint64_t)dp->rank);
      ctx->prep_dp = 1;
      ctx->prep_pending = 0;
      ctx->prep_Xall = next->X_all_next;
    }
    return 0;
  }
  ARG_CHECK(opt->lazy_K == 0, "the lazy optimizer needs the overlapped form of the step (overlap != 0)");
  if (int e = dp_import_touched(dp->bufs, dp->G, opt->g, opt->n, opt->nseg, opt->seg_begin, opt->seg_rows, opt->seg_width,
                                opt->seg_flags, dp->dense_begin, dp->loss_sum, dp->cap, dp->D, dp->mask, dp->where, dp->buf, stream))
    return e;
  return dccf_dense_opt_step_rows(opt->kind, opt->p, opt->g, opt->s1, opt->s2, opt->n, opt->lr, opt->wd, opt->l2, opt->clip,
                                  opt->step, opt->nseg, opt->seg_begin, opt->seg_rows, opt->seg_width, opt->seg_flags, stream);
}
