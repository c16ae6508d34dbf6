// Sustained v_mfma_f32_32x32x2_f32 rate from registers only (no memory traffic): the ceiling
// that the GEMM main loops are measured against.  Usage: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x16_bf16 from registers only; SAME = 1: six instructions in a row on one
// accumulator (the order the split-operand GEMM issues them), 0: round-robin over four accumulators.
template <int SAME>
__global__ __launch_bounds__(256) void k_mfma_bf16(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  u32x4 au = {threadIdx.x, 1u, 2u, 3u}, bu = {blockIdx.x, 5u, 6u, 7u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  for (int i = 0; i < iters; ++i) {
    if (SAME) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

// Same instruction stream with operands that differ from one instruction to the next (hashed bit
// patterns in six A and six B registers sets): data-dependent switching power, hence the sustained
// clock, is that of a real GEMM rather than of a constant-operand loop.
__global__ __launch_bounds__(256) void k_mfma_bf16_toggle(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 a[6], b[6];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 6; ++j) {
    u32x4 au, bu;
    for (int q = 0; q < 4; ++q) {
      h = h * 1664525u + 1013904223u; au[q] = (h & 0x807f807fu) | 0x3f003f00u;   // bf16 pairs in [0.5, 1) with random sign/mantissa
      h = h * 1664525u + 1013904223u; bu[q] = (h & 0x807f807fu) | 0x3f003f00u;
    }
    a[j] = __builtin_bit_cast(bf16x8, au); b[j] = __builtin_bit_cast(bf16x8, bu);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + t) % 6], acc[t], 0, 0, 0);
    if ((i & 255) == 255)   // keep the sums bounded
      for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] *= 1e-3f;
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mfma_f32_toggle(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 8; ++j) {
    h = h * 1664525u + 1013904223u; a[j] = __uint_as_float((h & 0x807fffffu) | 0x3f000000u);
    h = h * 1664525u + 1013904223u; b[j] = __uint_as_float((h & 0x807fffffu) | 0x3f000000u);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + t) & 7], acc[t], 0, 0, 0);
    if ((i & 255) == 255)
      for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] *= 1e-3f;
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

static void run_f32_toggle(float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {2000, 20000}) {
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k_mfma_f32_toggle, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mfma_f32_toggle, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("f32 32x32x2 varying operands, waves/SIMD 4, %.1f ms run: %.1f TFLOP/s\n", ms, flop / (ms * 1e-3) / 1e12);
  }
}

static void run_bf16_toggle(float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {4000, 40000}) {
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k_mfma_bf16_toggle, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mfma_bf16_toggle, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 24 * 32768.0;
    printf("bf16 32x32x16 varying operands, waves/SIMD 4, %.1f ms run: %.1f TFLOP/s (= %.1f fp32-equivalent at six terms)\n",
           ms, flop / (ms * 1e-3) / 1e12, flop / 6 / (ms * 1e-3) / 1e12);
  }
}

template <int SAME>
static void run_bf16(float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int wpb = 1; wpb <= 4; wpb *= 2) {
    const int blocks = 256 * wpb;
    hipLaunchKernelGGL(k_mfma_bf16<SAME>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mfma_bf16<SAME>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 24 * 32768.0;
    printf("bf16 32x32x16 %s, waves/SIMD %d: %.3f ms  %.1f TFLOP/s (= %.1f fp32-equivalent at six terms)\n",
           SAME ? "six in a row per accumulator" : "round-robin", wpb, ms, flop / (ms * 1e-3) / 1e12,
           flop / 6 / (ms * 1e-3) / 1e12);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wpb = 1; wpb <= 4; wpb *= 2) {       // workgroups per CU (4 waves each)
    const int blocks = 256 * wpb;
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", wpb, ms, flop / (ms * 1e-3) / 1e12);
  }
  run_bf16<0>(d);
  run_bf16<1>(d);
  run_bf16_toggle(d);
  run_f32_toggle(d);
  return 0;
}
