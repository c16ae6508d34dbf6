// Does VALU work of one wave overlap fp32 MFMA work of another wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// waves [0, mw) of each SIMD-slot do MFMA chains with NACC accumulators; the others do VALU chains.
// block = 256 * WPS threads: wave w runs on SIMD w & 3; role by (w >> 2)
template <int NACC, int KIND>
__global__ void k_roles(float* out, int mfma_iters, int valu_iters, int mfma_waves_per_simd, int valu_waves_per_simd) {
  const int wave = threadIdx.x >> 6;
  const int slot = wave >> 2;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  float s = 0.f;
  if (slot < mfma_waves_per_simd) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
      for (int j = 0; j < 32 / NACC; ++j)
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  } else if (slot < mfma_waves_per_simd + valu_waves_per_simd) {
    float x[8] = {a, b, a + 1.f, b + 1.f, a + 2.f, b + 2.f, a + 3.f, b + 3.f};
    for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v]) : "v"(b));
          if (KIND == 1) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[v]) : "v"(b));
          if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&x[v & 6]) : "v"(*(double*)&x[(v & 6)]));
        }
      }
    }
    for (int v = 0; v < 8; ++v) s += x[v];
  }
  if (s == 123.456f) out[0] = s;
}

// in-wave: NACC independent accumulators round robin with NV VALU per MFMA
template <int NACC, int NV>
__global__ __launch_bounds__(256) void k_inwave(float* out, int iters) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  float x[4] = {a, b, a + 1.f, b + 1.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32 / NACC; ++j)
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 3]) : "v"(b));
      }
  }
  float s = x[0] + x[1] + x[2] + x[3];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

template <class F>
static float time_it(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}

int main() {
  float* out; CK(hipMalloc(&out, 1024));
  const int mi = 256;   // 8192 MFMAs per wave
  for (int vi : {0, 64, 128, 256}) {
    // vi*256 VALU per wave
    float t_m = time_it([&] { hipLaunchKernelGGL((k_roles<1, 0>), dim3(256), dim3(512), 0, 0, out, mi, 0, 1, 1); }, 5);
    float t_v = time_it([&] { hipLaunchKernelGGL((k_roles<1, 0>), dim3(256), dim3(512), 0, 0, out, 0, vi, 1, 1); }, 5);
    float t_b = time_it([&] { hipLaunchKernelGGL((k_roles<1, 0>), dim3(256), dim3(512), 0, 0, out, mi, vi, 1, 1); }, 5);
    printf("cross-wave v_fma_f32  (%6d VALU/wave): MFMA only %7.1f us, VALU only %7.1f us, both %7.1f us\n", vi * 256, t_m, t_v, t_b);
  }
  for (int vi : {128}) {
    float t_v = time_it([&] { hipLaunchKernelGGL((k_roles<1, 1>), dim3(256), dim3(512), 0, 0, out, 0, vi, 1, 1); }, 5);
    float t_b = time_it([&] { hipLaunchKernelGGL((k_roles<1, 1>), dim3(256), dim3(512), 0, 0, out, mi, vi, 1, 1); }, 5);
    printf("cross-wave v_or_b32   (%6d VALU/wave): VALU only %7.1f us, both %7.1f us\n", vi * 256, t_v, t_b);
    t_v = time_it([&] { hipLaunchKernelGGL((k_roles<1, 2>), dim3(256), dim3(512), 0, 0, out, 0, vi, 1, 1); }, 5);
    t_b = time_it([&] { hipLaunchKernelGGL((k_roles<1, 2>), dim3(256), dim3(512), 0, 0, out, mi, vi, 1, 1); }, 5);
    printf("cross-wave v_pk_fma   (%6d VALU/wave): VALU only %7.1f us, both %7.1f us\n", vi * 256, t_v, t_b);
  }
  {
    float t2 = time_it([&] { hipLaunchKernelGGL((k_roles<1, 0>), dim3(256), dim3(512), 0, 0, out, mi, 0, 2, 0); }, 5);
    float t4 = time_it([&] { hipLaunchKernelGGL((k_roles<1, 0>), dim3(256), dim3(1024), 0, 0, out, mi / 2, 0, 4, 0); }, 5);
    printf("two MFMA waves per SIMD (2 x 8192 MFMAs): %7.1f us;  four waves x 4096: %7.1f us\n", t2, t4);
  }
#define IW(NACC, NV) { float t = time_it([&] { hipLaunchKernelGGL((k_inwave<NACC, NV>), dim3(256), dim3(256), 0, 0, out, mi); }, 5); \
    printf("in-wave, %d accumulators, %2d v_fma per MFMA: %7.1f us\n", NACC, NV, t); }
  IW(1, 0) IW(1, 4) IW(2, 0) IW(2, 4) IW(4, 0) IW(4, 2) IW(4, 4) IW(4, 8)
  return 0;
}
