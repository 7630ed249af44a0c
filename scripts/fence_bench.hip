// fence_bench.hip — what the cross-workgroup "last arriver sums in fixed order" pattern costs on gfx950 (the
// split-K forward that was costed and not built): 256 workgroups x 512 threads, each owning a quarter of K of 3 tiles; per tile a
// workgroup stores its 32 x 64 partial (8 KB, one dwordx4 per thread), fences, and bumps the tile's counter; the workgroup that
// arrives last reads the four partials in fixed order and stores the sum.  Variants: 0 = stores only (no fence, no counter),
// 1 = fence + counter + last-arriver sum after every tile, 2 = all three partials first, then one fence, the three counters and the
// sums at the end, 3 = as 2 without any fence: the partials leave as relaxed agent-scope atomic stores and are read back as relaxed
// agent-scope atomic loads (both bypass the XCD's non-coherent L2), ordered against the counter by s_waitcnt alone.
// Diagnostic only; build and run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 scripts/fence_bench.hip -o /tmp/fence_bench && /tmp/fence_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int TILES_PER_WG = 3, KQ = 4, TILE_FLOATS = 32 * 64;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int MODE>
__global__ __launch_bounds__(512) void k(float* __restrict__ part, int* __restrict__ cnt, float* __restrict__ out, int ntiles,
                                         int spin) {
  __shared__ int last[TILES_PER_WG];
  const int kq = blockIdx.x & 3, i = blockIdx.x >> 2, tid = threadIdx.x;
  float acc = 0.f;
  for (int t = 0; t < TILES_PER_WG; ++t) {
    const int tile = i + t * (gridDim.x >> 2);
    if (tile >= ntiles) break;
    for (int s = 0; s < spin; ++s) acc = fmaf(acc, 1.0001f, (float)tid);          // stands in for the k-loop
    float4 v = {acc, acc + 1.f, acc + 2.f, (float)kq};
    if (MODE == 3) {
      float* d = &part[((size_t)tile * KQ + kq) * TILE_FLOATS + 4 * tid];
      __hip_atomic_store(d, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(d + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(d + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(d + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
      *reinterpret_cast<float4*>(&part[((size_t)tile * KQ + kq) * TILE_FLOATS + 4 * tid]) = v;
    if (MODE == 1) {
      __threadfence();
      __syncthreads();
      if (tid == 0) last[0] = atomicAdd(&cnt[tile], 1) == KQ - 1;
      __syncthreads();
      if (last[0]) {
        __threadfence();
        float4 s = ld4(&part[((size_t)tile * KQ) * TILE_FLOATS + 4 * tid]);
        for (int q = 1; q < KQ; ++q) {
          const float4 u = ld4(&part[((size_t)tile * KQ + q) * TILE_FLOATS + 4 * tid]);
          s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
        }
        *reinterpret_cast<float4*>(&out[(size_t)tile * TILE_FLOATS + 4 * tid]) = s;
        if (tid == 0) cnt[tile] = 0;
      }
      __syncthreads();
    }
  }
  if (MODE == 3) {
    __builtin_amdgcn_s_waitcnt(0);                      // my own stores are acknowledged
    __syncthreads();
    if (tid < TILES_PER_WG) {
      const int tile = i + tid * (gridDim.x >> 2);
      last[tid] = tile < ntiles ? (__hip_atomic_fetch_add(&cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == KQ - 1) : 0;
    }
    __syncthreads();
    for (int t = 0; t < TILES_PER_WG; ++t) {
      if (!last[t]) continue;
      const int tile = i + t * (gridDim.x >> 2);
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < KQ; ++q)
        for (int e = 0; e < 4; ++e)
          s[e] += __hip_atomic_load(&part[((size_t)tile * KQ + q) * TILE_FLOATS + 4 * tid + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *reinterpret_cast<float4*>(&out[(size_t)tile * TILE_FLOATS + 4 * tid]) = float4{s[0], s[1], s[2], s[3]};
      if (tid == 0) __hip_atomic_store(&cnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (MODE == 2) {
    __threadfence();
    __syncthreads();
    if (tid < TILES_PER_WG) {
      const int tile = i + tid * (gridDim.x >> 2);
      last[tid] = tile < ntiles ? (atomicAdd(&cnt[tile], 1) == KQ - 1) : 0;
    }
    __syncthreads();
    bool any = false;
    for (int t = 0; t < TILES_PER_WG; ++t) any |= last[t] != 0;
    if (any) __threadfence();
    for (int t = 0; t < TILES_PER_WG; ++t) {
      if (!last[t]) continue;
      const int tile = i + t * (gridDim.x >> 2);
      float4 s = ld4(&part[((size_t)tile * KQ) * TILE_FLOATS + 4 * tid]);
      for (int q = 1; q < KQ; ++q) {
        const float4 u = ld4(&part[((size_t)tile * KQ + q) * TILE_FLOATS + 4 * tid]);
        s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
      }
      *reinterpret_cast<float4*>(&out[(size_t)tile * TILE_FLOATS + 4 * tid]) = s;
      if (tid == 0) cnt[tile] = 0;
    }
  }
}

int main() {
  const int ntiles = 176, grid = 256;
  float *part, *out;
  int* cnt;
  CHECK(hipMalloc(&part, (size_t)ntiles * KQ * TILE_FLOATS * 4));
  CHECK(hipMalloc(&out, (size_t)ntiles * TILE_FLOATS * 4));
  CHECK(hipMalloc(&cnt, ntiles * 4));
  CHECK(hipMemset(cnt, 0, ntiles * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 200;
  for (int spin : {0, 200}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f, sum = 0.f;
      for (int pass = 0; pass < 3; ++pass) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, part, cnt, out, ntiles, spin);
          else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, part, cnt, out, ntiles, spin);
          else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, part, cnt, out, ntiles, spin);
          else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, part, cnt, out, ntiles, spin);
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        sum += ms;
      }
      printf("spin %5d mode %d: %.2f us per launch (best of 3 x %d back-to-back launches)\n", spin, mode, best / reps * 1e3f, reps);
    }
  }
  // correctness of the pattern: every tile's sum carries kq = 0 + 1 + 2 + 3 in .w
  std::vector<float> h((size_t)ntiles * TILE_FLOATS);
  CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int t = 0; t < ntiles; ++t)
    for (int e = 3; e < TILE_FLOATS; e += 4) bad += h[(size_t)t * TILE_FLOATS + e] != 6.f;
  printf("last-arriver sums wrong: %d\n", bad);
  return 0;
}
