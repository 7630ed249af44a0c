// bf16x6_bench.hip — would an exact three-way bf16 split of the fp32 operands (6 bf16 MFMAs per fp32 product, on the matrix pipe that
// CO-EXECUTES with the vector ALU) beat the fp32-input MFMA (which shares the vector ALU's issue pipe with the Philox / Box-Muller
// generator: SQ_VALU_MFMA_COEXEC_CYCLES = 0) in the forward's inner loop?  Both variants run the forward's
// work per tile: a lane generates its A operand (feature + fresh N(0, std^2) noise, one Philox4x32-10 call + two Box-Muller pairs per
// 4 values) and multiplies it with a register-resident slice of W^T, 8 waves per workgroup, K split over the waves as in k_noise_fwd.
//   variant 0: v_mfma_f32_32x32x2_f32, 2 column halves: 8 MFMAs (512 pipe cycles) per 4 generated values
//   variant 1: a = a1 + a2 + a3 exactly (three bf16 = 24 significand bits, by truncation), W likewise (split once, outside the loop);
//              products with i + j <= 4 (6 of 9: the dropped ones are below 2^-24 |a||w|): 12 v_mfma_f32_32x32x16_bf16 per 8 values
// Prints time per launch for 176 and 256 workgroups and the largest difference between the two results relative to sum |a||w|.
// Diagnostic only; on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. scripts/bf16x6_bench.hip -o /tmp/bf16x6 && /tmp/bf16x6
#include "../dccf_amd/csrc/common.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

thread_local char g_dccf_err[512];
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

constexpr int NCM = 6, NGF = 2 * NCM;        // 12 groups of 4 values per wave and tile, as k_noise_fwd<64, 0, 6, ...>

// bit pattern helpers: the top 16 bits of an fp32 ARE a bf16 (truncation)
__device__ __forceinline__ uint32_t fbits(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float hi16(float x) { return __uint_as_float(fbits(x) & 0xffff0000u); }
// two fp32 (already bf16-representable) -> one register of two bf16: element 0 in the low half
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (fbits(lo) >> 16) | (fbits(hi) & 0xffff0000u); }

template <int VAR>
__global__ __launch_bounds__(512) void k(const float* __restrict__ WT, const float* __restrict__ feat, float* __restrict__ out, int ntiles,
                                         rng_key key, float nscale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c31 = lane & 31;
  constexpr int ND = 2;
  f32x16 acc[ND];
#pragma unroll
  for (int nt = 0; nt < ND; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  if (VAR == 0) {
    float bf[NGF * 4 * ND];
#pragma unroll
    for (int g = 0; g < NGF; ++g)
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) bf[(g * 4 + o) * ND + nt] = WT[(int64_t)((wave * NGF + g) * 8 + 2 * o + h) * 64 + nt * 32 + c31];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int l = tile * 32 + c31;
      const float* frow = feat + (int64_t)(l & 1023) * 768;
#pragma unroll
      for (int g = 0; g < NGF; ++g) {
        float a[4], fv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) fv[o] = frow[(wave * NGF + g) * 8 + 2 * o + h];
        noise4((uint32_t)l, (uint32_t)(wave * NGF + g) * 2 + h, key, nscale, a);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          a[o] = fv[o] + a[o];
#pragma unroll
          for (int nt = 0; nt < ND; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[o], bf[(g * 4 + o) * ND + nt], acc[nt], 0, 0, 0);
        }
      }
    }
  } else if (VAR == 2) {
    float bf[NGF * 4 * ND];
#pragma unroll
    for (int g = 0; g < NGF; ++g)
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) bf[(g * 4 + o) * ND + nt] = WT[(int64_t)((wave * NGF + g) * 8 + 4 * h + o) * 64 + nt * 32 + c31];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int l = tile * 32 + c31;
      const float* frow = feat + (int64_t)(l & 1023) * 768;
#pragma unroll
      for (int g = 0; g < NGF; ++g) {
        float a[4];
        const float4 f4 = *reinterpret_cast<const float4*>(frow + (wave * NGF + g) * 8 + 4 * h);
        const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
        noise4((uint32_t)l, (uint32_t)(wave * NGF + g) * 2 + h, key, nscale, a);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          a[o] = fv[o] + a[o];
#pragma unroll
          for (int nt = 0; nt < ND; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[o], bf[(g * 4 + o) * ND + nt], acc[nt], 0, 0, 0);
        }
      }
    }
  } else {
    // W^T slice, split into three bf16 pieces per element, in the 32x32x16 operand layout: lane (col = c31, h) holds k = 8 h + j, j < 8
    // of a block of 16; a block = two groups g = 2 m, 2 m + 1: k-slot (h, j) <-> group 2 m + (j >> 2), value o = j & 3 (the SAME
    // element of W^T that variant 0 multiplies that generated value with)
    u32x4v wb[NGF / 2][ND][3];
#pragma unroll
    for (int m = 0; m < NGF / 2; ++m)
#pragma unroll
      for (int nt = 0; nt < ND; ++nt) {
        float w1[8], w2[8], w3[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int g = 2 * m + (j >> 2), o = j & 3;
          const float w = WT[(int64_t)((wave * NGF + g) * 8 + 2 * o + h) * 64 + nt * 32 + c31];
          w1[j] = hi16(w);
          const float r1 = w - w1[j];
          w2[j] = hi16(r1);
          w3[j] = hi16(r1 - w2[j]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wb[m][nt][0][q] = pack2(w1[2 * q], w1[2 * q + 1]);
          wb[m][nt][1][q] = pack2(w2[2 * q], w2[2 * q + 1]);
          wb[m][nt][2][q] = pack2(w3[2 * q], w3[2 * q + 1]);
        }
      }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int l = tile * 32 + c31;
      const float* frow = feat + (int64_t)(l & 1023) * 768;
#pragma unroll
      for (int m = 0; m < NGF / 2; ++m) {
        float a[8], fv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) fv[j] = frow[(wave * NGF + 2 * m + (j >> 2)) * 8 + 2 * (j & 3) + h];
        noise4((uint32_t)l, (uint32_t)(wave * NGF + 2 * m) * 2 + h, key, nscale, a);
        noise4((uint32_t)l, (uint32_t)(wave * NGF + 2 * m + 1) * 2 + h, key, nscale, a + 4);
        u32x4v a1, a2, a3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x0 = fv[2 * q] + a[2 * q], x1 = fv[2 * q + 1] + a[2 * q + 1];
          const float p0 = hi16(x0), p1 = hi16(x1);
          const float r0 = x0 - p0, r1 = x1 - p1;
          const float s0 = hi16(r0), s1 = hi16(r1);
          a1[q] = pack2(p0, p1);
          a2[q] = pack2(s0, s1);
          a3[q] = pack2(r0 - s0, r1 - s1);          // (8 significant bits left: exact in bf16)
        }
#define MF(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), C_, 0, 0, 0)
#pragma unroll
        for (int nt = 0; nt < ND; ++nt) {
          // smallest terms first
          MF(a3, wb[m][nt][0], acc[nt]);
          MF(a1, wb[m][nt][2], acc[nt]);
          MF(a2, wb[m][nt][1], acc[nt]);
          MF(a2, wb[m][nt][0], acc[nt]);
          MF(a1, wb[m][nt][1], acc[nt]);
          MF(a1, wb[m][nt][0], acc[nt]);
        }
#undef MF
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < ND; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(((int64_t)blockIdx.x * 8 + wave) * ND + nt) * 1024 + r * 64 + lane] = acc[nt][r];
}

int main() {
  const int K = 8 * NGF * 8;                  // 768 k per tile
  std::vector<float> hW((size_t)K * 64), hF(1024 * 768);
  uint32_t s = 7;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hW) v = rnd() * 0.2f;
  for (auto& v : hF) v = rnd() * 0.1f;
  float *W, *F, *o0, *o1;
  const int maxwg = 1024;
  const size_t on = (size_t)maxwg * 8 * 2 * 1024;
  CHECK(hipMalloc((void**)&W, hW.size() * 4)); CHECK(hipMalloc((void**)&F, hF.size() * 4));
  CHECK(hipMalloc((void**)&o0, on * 4)); CHECK(hipMalloc((void**)&o1, on * 4));
  CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(F, hF.data(), hF.size() * 4, hipMemcpyHostToDevice));
  const rng_key key = make_key(2019, 2, 5);
  const float nscale = -2.0f * 0.69314718055994530942f * 0.1f * 0.1f;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct Cfg { int wgs, tiles; const char* name; };
  const Cfg cfgs[] = {{176, 176, "176 tiles on 176 workgroups (B = 128)"}, {256, 256, "256 tiles on 256 workgroups"},
                      {1024, 11264, "11264 tiles on 1024 workgroups (evaluation batch)"}};
  for (const Cfg& c : cfgs) {
    float best[3] = {1e9f, 1e9f, 1e9f};
    for (int var = 0; var < 3; ++var)
      for (int it = 0; it < 8; ++it) {
        CHECK(hipEventRecord(e0, 0));
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(c.wgs), dim3(512), 0, 0, W, F, o0, c.tiles, key, nscale);
        else if (var == 1) hipLaunchKernelGGL(k<1>, dim3(c.wgs), dim3(512), 0, 0, W, F, o1, c.tiles, key, nscale);
        else hipLaunchKernelGGL(k<2>, dim3(c.wgs), dim3(512), 0, 0, W, F, o1, c.tiles, key, nscale);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2 && ms < best[var]) best[var] = ms;
      }
    CHECK(hipGetLastError());
    const size_t cmp = (size_t)c.wgs * 8 * 2 * 1024;
    std::vector<float> h0(cmp), h1(cmp);
    CHECK(hipMemcpy(h0.data(), o0, cmp * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h1.data(), o1, cmp * 4, hipMemcpyDeviceToHost));
    double maxd = 0.0, maxv = 0.0;
    for (size_t i = 0; i < cmp; ++i) {
      maxd = fmax(maxd, fabs((double)h0[i] - (double)h1[i]));
      maxv = fmax(maxv, fabs((double)h0[i]));
    }
    printf("%-52s fp32 MFMA %8.2f us   bf16 x 6 %8.2f us   ratio %.3f   max|diff| %.3e (max |value| %.3e)   fp32 MFMA with dwordx4 feature loads %8.2f us\n", c.name, best[0] * 1e3,
           best[1] * 1e3, best[1] / best[0], maxd, maxv, best[2] * 1e3);
    fflush(stdout);
  }
  return 0;
}
