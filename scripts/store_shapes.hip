// store_shapes.hip — how fast can a [U x I] fp32 matrix with an ODD row length be written, depending on the shape of one wave's store
// instruction?  (The full U x I predict kernel is MFMA-bound only if its 19.4 GB of output leaves at >= 2.5 TB/s beside the MFMAs.)
// Every variant walks the matrix as the predict kernel does: a wave owns 32 rows and moves along them in tiles of 64 columns, 4-wave
// workgroups holding 64 KB of LDS (two per CU), blockIdx % 8 picks the column split.
//   0: 2 rows x 128 B per instruction (the MFMA accumulator layout: lane (c, h) -> row (r & 3) + 8 (r >> 2) + 4 h, column c)
//   1: 1 row x 256 B per instruction (dword per lane)
//   2: 4 rows x 256 B per instruction (dwordx4 per lane, 4-byte aligned)
//   3: 2 rows x 256 B per instruction (dwordx2 per lane)
//   4: as 2, but only the 16-byte-aligned body of each row piece (what alignment alone is worth; the matrix is not complete)
//   hipcc --offload-arch=gfx950 -O3 scripts/store_shapes.hip -o /tmp/store_shapes && /tmp/store_shapes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, int64_t U, int64_t I, int splits) {
  __shared__ float pad[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (U < 0) pad[threadIdx.x] = 1.f;
  const int64_t gt = I / 64;                                       // whole tiles only
  const int64_t grp = blockIdx.x / splits, sp = blockIdx.x % splits;
  const int64_t T0 = gt * sp / splits, T1 = gt * (sp + 1) / splits;
  const int64_t u0 = (grp * 4 + wave) * 32;
  if (u0 + 32 > U) return;
  const int h = lane >> 5, c31 = lane & 31;
  const float v = (float)lane;
  for (int64_t T = T0; T < T1; ++T) {
    float* o = out + u0 * I + T * 64;
    if (SHAPE == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* orow = o + (int64_t)((r & 3) + 8 * (r >> 2) + 4 * h) * I + c31;
        orow[0] = v;
        orow[32] = v;
      }
    } else if (SHAPE == 1) {
#pragma unroll
      for (int r = 0; r < 32; ++r) o[r * I + lane] = v;
    } else if (SHAPE == 2) {
#pragma unroll
      for (int g = 0; g < 8; ++g) *reinterpret_cast<f4u*>(o + (4 * g + (lane >> 4)) * I + 4 * (lane & 15)) = f4u{v, v, v, v};
    } else if (SHAPE == 3) {
#pragma unroll
      for (int g = 0; g < 16; ++g) *reinterpret_cast<f2u*>(o + (2 * g + (lane >> 5)) * I + 2 * (lane & 31)) = f2u{v, v};
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        float* p = o + (4 * g + (lane >> 4)) * I;
        const int s = (int)((4 - (((uintptr_t)p >> 2) & 3)) & 3);
        if ((lane & 15) < 15) *reinterpret_cast<f4a*>(p + s + 4 * (lane & 15)) = f4a{v, v, v, v};
      }
    }
  }
}

template <int SHAPE>
static int run(float* out, int64_t U, int64_t I, const char* name) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  const int64_t groups = U / 128;
  for (int it = 0; it < 5; ++it) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<SHAPE>, dim3((unsigned)(groups * 8)), dim3(256), 0, 0, out, U, I, 8);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 1 && ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  const double bytes = (double)(groups * 128) * (double)(I / 64 * 64) * 4.0 * (SHAPE == 4 ? 60.0 / 64.0 : 1.0);
  printf("%-56s I = %lld : %7.3f ms  %6.2f TB/s\n", name, (long long)I, best, bytes / best / 1e9);
  fflush(stdout);
  return 0;
}

int main() {
  const int64_t U = 75258;
  float* out;
  CHECK(hipMalloc((void**)&out, (size_t)U * 64448 * 4));
  int rc = 0;
  for (int64_t I : {(int64_t)64443, (int64_t)64448}) {
    rc |= run<0>(out, U, I, "2 rows x 128 B per instruction (accumulator layout)");
    rc |= run<1>(out, U, I, "1 row x 256 B per instruction (dword)");
    rc |= run<3>(out, U, I, "2 rows x 256 B per instruction (dwordx2)");
    rc |= run<2>(out, U, I, "4 rows x 256 B per instruction (dwordx4, 4-byte aligned)");
    rc |= run<4>(out, U, I, "4 rows x 240 B, 16-byte aligned body only");
  }
  return rc;
}
