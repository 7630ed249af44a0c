// shard_kernels.hip — row movers of the row-sharded multi-GPU path (dccf_amd/sharded.py): pack the rows a peer needs
// into an all-to-all payload, unpack a received payload into the compact per-step tables, add received gradient rows
// into the local gradient shard.  Pure HBM byte movers: one payload row per wave, lanes walk the columns (coalesced
// 256-B segments); the scatter-add issues one shaped float-atomic wave instruction per 64 columns of a row.
#include "common.hpp"

#define SHARD_MAX_TABLES 4
struct TableSet {
  float* t[SHARD_MAX_TABLES];
  int w[SHARD_MAX_TABLES];      // row width (floats) of each table
  int n;
  int total;
};

// out[dst[j] (or j), 0:total] = [T0[idx[j], :] | T1[idx[j], :] | ...]; out rows are `ld` floats apart (ld >= total)
__global__ __launch_bounds__(256) void k_pack(const int* __restrict__ idx, const int* __restrict__ dst, int64_t n, TableSet ts,
                                              float* __restrict__ out, int ld) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t j = wave; j < n; j += nw) {
    const int64_t r = idx[j];
    const int64_t o = dst ? dst[j] : j;
    int off = 0;
    for (int q = 0; q < ts.n; ++q) {
      for (int c = lane; c < ts.w[q]; c += 64) out[o * ld + off + c] = ts.t[q][r * ts.w[q] + c];
      off += ts.w[q];
    }
  }
}

// Tq[dst[j] (or j), :] = in[j, off_q : off_q + w_q]; payload rows are `ld` floats apart
__global__ __launch_bounds__(256) void k_unpack(const float* __restrict__ in, int64_t n, const int* __restrict__ dst,
                                                TableSet ts, int ld) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t j = wave; j < n; j += nw) {
    const int64_t r = dst ? dst[j] : j;
    int off = 0;
    for (int q = 0; q < ts.n; ++q) {
      for (int c = lane; c < ts.w[q]; c += 64) ts.t[q][r * ts.w[q] + c] = in[j * ld + off + c];
      off += ts.w[q];
    }
  }
}

// g[idx[j], :] += rows[j, :]
__global__ __launch_bounds__(256) void k_scatter_add(const int* __restrict__ idx, int64_t n, const float* __restrict__ rows,
                                                     int width, float* __restrict__ g, uint8_t* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t j = wave; j < n; j += nw) {
    const int64_t r = idx[j];
    for (int c = lane; c < width; c += 64) atomicAdd(&g[r * width + c], rows[j * width + c]);
    if (flags && lane == 0) flags[r] = 1;          // "touched" byte of the row-aware optimizer step
  }
}

// Several pack (or unpack) jobs in ONE launch: blockIdx.y picks the job.  A training step of the sharded path moves three
// payload kinds out (user rows, item rows, feature rows) and two in; at batch 128 every launch saved is ~5 % of the step.
#define SHARD_MAX_JOBS 4
struct JobSet {
  const int* idx[SHARD_MAX_JOBS];
  const int* dst[SHARD_MAX_JOBS];
  int64_t n[SHARD_MAX_JOBS];
  TableSet ts[SHARD_MAX_JOBS];
  float* buf[SHARD_MAX_JOBS];
  int ld[SHARD_MAX_JOBS];
  float* zero;               // unpack only: zero[0 : zero_n] = 0 (the compact gradient table of the step), by the extra grid row
  int64_t zero_n;
};

__global__ __launch_bounds__(256) void k_pack_multi(JobSet js) {
  const int q = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const TableSet& ts = js.ts[q];
  for (int64_t j = wave; j < js.n[q]; j += nw) {
    const int64_t r = js.idx[q][j];
    const int64_t o = js.dst[q] ? js.dst[q][j] : j;
    int off = 0;
    for (int t = 0; t < ts.n; ++t) {
      for (int c = lane; c < ts.w[t]; c += 64) js.buf[q][o * js.ld[q] + off + c] = ts.t[t][r * ts.w[t] + c];
      off += ts.w[t];
    }
  }
}

__global__ __launch_bounds__(256) void k_unpack_multi(JobSet js, int njobs) {
  const int q = blockIdx.y;
  if (q == njobs) {          // the extra grid row zeroes the step's compact gradient table
    float4* z = reinterpret_cast<float4*>(js.zero);
    const int64_t n4 = js.zero_n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
      z[i] = float4{0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < js.zero_n; i += (int64_t)gridDim.x * blockDim.x)
      js.zero[i] = 0.f;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const TableSet& ts = js.ts[q];
  for (int64_t j = wave; j < js.n[q]; j += nw) {
    const int64_t r = js.dst[q] ? js.dst[q][j] : j;
    int off = 0;
    for (int t = 0; t < ts.n; ++t) {
      for (int c = lane; c < ts.w[t]; c += 64) ts.t[t][r * ts.w[t] + c] = js.buf[q][j * js.ld[q] + off + c];
      off += ts.w[t];
    }
  }
}

static int make_set(TableSet& ts, float* const* tables, const int32_t* widths, int32_t ntables) {
  ARG_CHECK(tables && widths && ntables >= 1 && ntables <= SHARD_MAX_TABLES, "1..4 tables");
  ts.n = ntables;
  ts.total = 0;
  for (int q = 0; q < ntables; ++q) {
    ARG_CHECK(tables[q] != nullptr && widths[q] > 0, "NULL table or non-positive width");
    ts.t[q] = tables[q];
    ts.w[q] = widths[q];
    ts.total += widths[q];
  }
  return 0;
}

extern "C" int shard_pack_rows(const int32_t* idx, const int32_t* dst, int64_t n, const float* const* tables,
                               const int32_t* widths, int32_t ntables, float* out, int32_t ld, void* stream) {
  TableSet ts;
  if (int e = make_set(ts, (float* const*)tables, widths, ntables)) return e;
  ARG_CHECK(n >= 0 && (n == 0 || (idx && out)) && ld >= ts.total, "NULL idx / out or ld < row width");
  if (n == 0) return 0;
  const int grid = (int)min((int64_t)2048, (n + 3) / 4);
  hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, (hipStream_t)stream, idx, dst, n, ts, out, (int)ld);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int shard_unpack_rows(const float* in, int32_t ld, int64_t n, const int32_t* dst, float* const* tables,
                                 const int32_t* widths, int32_t ntables, void* stream) {
  TableSet ts;
  if (int e = make_set(ts, tables, widths, ntables)) return e;
  ARG_CHECK(n >= 0 && (n == 0 || in) && ld >= ts.total, "NULL payload or ld < row width");
  if (n == 0) return 0;
  const int grid = (int)min((int64_t)2048, (n + 3) / 4);
  hipLaunchKernelGGL(k_unpack, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, n, dst, ts, (int)ld);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int shard_scatter_add(const int32_t* idx, int64_t n, const float* rows, int32_t width, float* g, uint8_t* flags,
                                 void* stream) {
  ARG_CHECK(n >= 0 && width > 0 && (n == 0 || (idx && rows && g)), "bad arguments");
  if (n == 0) return 0;
  const int grid = (int)min((int64_t)2048, (n + 3) / 4);
  hipLaunchKernelGGL(k_scatter_add, dim3(grid), dim3(256), 0, (hipStream_t)stream, idx, n, rows, width, g, flags);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int make_jobs(JobSet& js, const shard_job_t* jobs, int32_t njobs, int64_t* nmax, bool pack) {
  ARG_CHECK(jobs && njobs >= 1 && njobs <= SHARD_MAX_JOBS, "1..4 jobs");
  memset(&js, 0, sizeof(js));
  *nmax = 0;
  for (int q = 0; q < njobs; ++q) {
    const shard_job_t& j = jobs[q];
    if (int e = make_set(js.ts[q], (float* const*)j.tables, j.widths, j.ntables)) return e;
    ARG_CHECK(j.n >= 0 && (j.n == 0 || (j.buf && (j.idx || !pack))) && j.ld >= js.ts[q].total, "NULL idx / payload or ld < row width");
    js.idx[q] = j.idx; js.dst[q] = j.dst; js.n[q] = j.n; js.buf[q] = j.buf; js.ld[q] = j.ld;
    if (j.n > *nmax) *nmax = j.n;
  }
  return 0;
}

extern "C" int shard_pack_multi(const shard_job_t* jobs, int32_t njobs, void* stream) {
  JobSet js;
  int64_t nmax;
  if (int e = make_jobs(js, jobs, njobs, &nmax, true)) return e;
  if (nmax == 0) return 0;
  const dim3 grid((unsigned)min((int64_t)1024, (nmax + 3) / 4), (unsigned)njobs);
  hipLaunchKernelGGL(k_pack_multi, grid, dim3(256), 0, (hipStream_t)stream, js);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int shard_unpack_multi(const shard_job_t* jobs, int32_t njobs, float* zero, int64_t zero_n, void* stream) {
  JobSet js;
  int64_t nmax;
  if (int e = make_jobs(js, jobs, njobs, &nmax, false)) return e;
  ARG_CHECK(zero_n >= 0 && (zero_n == 0 || (zero && (uintptr_t)zero % 16 == 0)), "zero buffer must be 16-byte aligned");
  js.zero = zero;
  js.zero_n = zero_n;
  if (nmax == 0 && zero_n == 0) return 0;
  const int64_t work = max(nmax, (zero_n + 1023) / 1024);
  const dim3 grid((unsigned)max((int64_t)1, min((int64_t)1024, (work + 3) / 4)), (unsigned)(njobs + (zero_n ? 1 : 0)));
  hipLaunchKernelGGL(k_unpack_multi, grid, dim3(256), 0, (hipStream_t)stream, js, (int)njobs);
  HIP_TRY(hipGetLastError());
  return 0;
}
