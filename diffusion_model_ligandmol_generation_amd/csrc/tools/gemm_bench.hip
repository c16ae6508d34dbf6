// Stand-alone timing harness for the row-slab GEMM kernels (developer tool, not part of the
// library).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_bench gemm_bench.hip
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>

#include "../epilogues.h"
#include "../gemm3.h"
#include "../gemm_dw.h"
#include "../gemm4.h"

struct EpiNull {  // keeps the accumulators alive with one store per lane
  static constexpr int kStores = 0;
  float* C;
  __device__ __forceinline__ void operator()(f32x16 (&acc)[8], int row0, int by, int lane) const {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[ct][r];
    if (s == 123.456f) C[row0 + lane] = s;
  }
  template <class MB>
  __device__ __forceinline__ void cs(f32x16 (&acc)[4], int row0, int by, int half, int lane, const MB&) const {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[ct][r];
    if (s == 123.456f) C[row0 + lane] = s;
  }
};


// Experimental copies of the column-split main loop with parts removed (results are garbage):
// MODE 0 = as shipped, 1 = no DMA inside the loop, 2 = no DMA and no barriers, 3 = MFMA only.
template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_cs_dbg(GemmCore g, APtrPlain ap, EpiNull epi) {
  __shared__ __attribute__((aligned(1024))) float smem3[G3_LDS_BYTES / 4];
  constexpr int A_FL = 128 * G2_ROWF;
  constexpr int ST = G3_STAGE_FLOATS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = wave & 3, half = wave >> 2;
  const int by = blockIdx.y;
  const int row_blk = blockIdx.x * 128;
  const float* __restrict__ W = g.W[by];
  f32x16 acc[4];
  for (int ct = 0; ct < 4; ++ct)
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const int pr = lane >> 2, ps = lane & 3;
  const int lslot = ps ^ ((pr >> 2) & 3);
  int a_row = row_blk + wave * 16 + pr;
  a_row = a_row < g.M ? a_row : g.M - 1;
  const float* w_src[2];
  for (int i = 0; i < 2; ++i) {
    int n = (wave * 2 + i) * 16 + pr;
    n = n < g.n_valid ? n : g.n_valid - 1;
    w_src[i] = W + (size_t)n * g.ldw + lslot * 4;
  }
  auto issue = [&](int kc, int buf) {
    float* sA = smem3 + buf * ST;
    float* sW = sA + A_FL;
    const int k0 = kc * G2_BK;
    lds_dma16(ap.ptr(a_row, k0 + lslot * 4), sA + wave * 256);
    for (int i = 0; i < 2; ++i) lds_dma16(w_src[i] + k0, sW + (wave * 2 + i) * 256);
  };
  const int nk = g.K / G2_BK;
  const int fi = lane & 31, hk = lane >> 5;
  const int swz = (fi >> 2) & 3;
  const int off0 = ((0 * 2 + hk) ^ swz) * 4, off1 = ((1 * 2 + hk) ^ swz) * 4;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  int cur = 0;
  f32x4 a0, a1, b[2][2];
  {
    const float* sA = smem3;
    const float* sW = sA + A_FL + (half * 128 + fi) * G2_ROWF;
    const float* arow = sA + (slab * 32 + fi) * G2_ROWF;
    a0 = *reinterpret_cast<const f32x4*>(arow + off0); a1 = *reinterpret_cast<const f32x4*>(arow + off1);
    b[0][0] = *reinterpret_cast<const f32x4*>(sW + off0); b[0][1] = *reinterpret_cast<const f32x4*>(sW + off1);
    b[1][0] = b[0][1]; b[1][1] = b[0][0];
  }
  for (int kc = 0; kc < nk; ++kc) {
    if (MODE <= 1) {
      if (MODE == 0) { if (kc + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F73); else __builtin_amdgcn_s_waitcnt(0x0F70); }
      __builtin_amdgcn_s_barrier();
    }
    if (MODE == 0 && kc + 2 < nk) issue(kc + 2, cur == 0 ? 2 : cur - 1);
    const float* sA = smem3 + cur * ST;
    const float* sW = sA + A_FL + (half * 128 + fi) * G2_ROWF;
    const float* arow = sA + (slab * 32 + fi) * G2_ROWF;
    if (MODE <= 2) {
      a0 = *reinterpret_cast<const f32x4*>(arow + off0);
      a1 = *reinterpret_cast<const f32x4*>(arow + off1);
      b[0][0] = *reinterpret_cast<const f32x4*>(sW + off0);
      b[0][1] = *reinterpret_cast<const f32x4*>(sW + off1);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[ct & 1][0][j], acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE <= 2 && ct + 1 < 4) {
        b[(ct + 1) & 1][0] = *reinterpret_cast<const f32x4*>(sW + (ct + 1) * 32 * G2_ROWF + off0);
        b[(ct + 1) & 1][1] = *reinterpret_cast<const f32x4*>(sW + (ct + 1) * 32 * G2_ROWF + off1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[ct & 1][1][j], acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  HalfMailbox mb{smem3 + 3 * ST, wave, lane};
  epi.cs(acc, row_blk + slab * 32, by, half, lane, mb);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class F>
static float time_it(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int M = 29696, E = 63488;
  std::vector<float> h((size_t)E * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  float *A, *W, *C, *C2, *bias, *mod;
  int* mol;
  CK(hipMalloc(&A, (size_t)E * 1024 * 4)); CK(hipMalloc(&W, 1024 * 1024 * 4)); CK(hipMalloc(&C, (size_t)E * 1024 * 4));
  CK(hipMalloc(&C2, (size_t)E * 1024 * 4)); CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&mod, 1536 * 4));
  CK(hipMalloc(&mol, E * 4));
  CK(hipMemcpy(A, h.data(), (size_t)E * 1024 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), 1024 * 1024 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(mod, h.data(), 1536 * 4, hipMemcpyHostToDevice));
  CK(hipMemset(mol, 0, E * 4));

  auto core = [&](int Mr, int K, int nb, int ldw) {
    GemmCore g; memset(&g, 0, sizeof g);
    g.M = Mr; g.K = K; g.n_valid = 256; g.ldw = ldw;
    for (int i = 0; i < nb; ++i) g.W[i] = W + (size_t)i * 256 * ldw;
    return g;
  };
  auto report = [&](const char* name, float ms, double flops) {
    printf("%-44s %8.1f us  %6.1f TFLOP/s\n", name, ms * 1e3, flops / (ms * 1e-3) / 1e12);
  };
  {  // FFN up: K=256, N=1024, GELU
    GemmCore g = core(M, 256, 4, 256);
    APtrPlain ap{A, 256};
    EpiBiasAct<8, ACT_GELU, false> e; memset(&e, 0, sizeof e);
    e.M = M; e.n_valid = 256; e.col_block = 256; e.C = C; e.ldc = 1024; e.alpha = 1.f;
    for (int i = 0; i < 4; ++i) e.bias[i] = bias + i * 256;
    float ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiBiasAct<8, ACT_GELU, false>>), dim3(232, 4), dim3(256), 0, 0, g, ap, e); }, 20);
    report("v2 ffn_up  M29696 K256 N1024 bias+GELU", ms, 2.0 * M * 256 * 1024);
    EpiBiasAct<8, ACT_NONE, false> e0; memcpy(&e0, &e, sizeof e);
    ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiBiasAct<8, ACT_NONE, false>>), dim3(232, 4), dim3(256), 0, 0, g, ap, e0); }, 20);
    report("v2         M29696 K256 N1024 bias only", ms, 2.0 * M * 256 * 1024);
    EpiNull en{C};
    ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiNull>), dim3(232, 4), dim3(256), 0, 0, g, ap, en); }, 20);
    report("v2         M29696 K256 N1024 no epilogue", ms, 2.0 * M * 256 * 1024);
    {  // 128-column blocks: twice the workgroups, half the accumulators
      GemmCore g4; memset(&g4, 0, sizeof g4);
      g4.M = M; g4.K = 256; g4.n_valid = 128; g4.ldw = 256;
      for (int i = 0; i < 8; ++i) g4.W[i] = W + (size_t)i * 128 * 256;
      EpiBiasAct<4, ACT_GELU, false> e4; memset(&e4, 0, sizeof e4);
      e4.M = M; e4.n_valid = 128; e4.col_block = 128; e4.C = C; e4.ldc = 1024; e4.alpha = 1.f;
      for (int i = 0; i < 8; ++i) e4.bias[i] = bias + i * 128;
      ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<4, APtrPlain, EpiBiasAct<4, ACT_GELU, false>>), dim3(232, 8), dim3(256), 0, 0, g4, ap, e4); }, 20);
      report("v2 NT=4    M29696 K256 N1024 bias+GELU", ms, 2.0 * M * 256 * 1024);
      EpiBiasAct<4, ACT_NONE, false> e40; memcpy(&e40, &e4, sizeof e4);
      ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<4, APtrPlain, EpiBiasAct<4, ACT_NONE, false>>), dim3(232, 8), dim3(256), 0, 0, g4, ap, e40); }, 20);
      report("v2 NT=4    M29696 K256 N1024 bias only", ms, 2.0 * M * 256 * 1024);
    }
    ALoadPlain al{A, 256, M, 256};
    ms = time_it([&] { hipLaunchKernelGGL((gemm_rows_kernel<8, W_ALIGNED, ALoadPlain, EpiBiasAct<8, ACT_GELU, false>>), dim3(232, 4), dim3(256), 0, 0, g, al, e); }, 20);
    report("v1 ffn_up  M29696 K256 N1024 bias+GELU", ms, 2.0 * M * 256 * 1024);
    ms = time_it([&] { hipLaunchKernelGGL((gemm_rows_kernel<8, W_ALIGNED, ALoadPlain, EpiNull>), dim3(232, 4), dim3(256), 0, 0, g, al, en); }, 20);
    report("v1         M29696 K256 N1024 no epilogue", ms, 2.0 * M * 256 * 1024);
  }
  {  // FFN down: K=1024, N=256, gated residual + modulated LN
    GemmCore g = core(M, 1024, 1, 1024);
    APtrPlain ap{A, 1024};
    EpiGate<true> e{M, bias, C2, mod, 0, 0, mod, 256, 512, mol, C, C + (size_t)M * 256, nullptr};
    float ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiGate<true>>), dim3(232, 1), dim3(256), 0, 0, g, ap, e); }, 20);
    report("v2 ffn_down M29696 K1024 N256 gate+LNmod", ms, 2.0 * M * 1024 * 256);
    EpiNull en{C};
    ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiNull>), dim3(232, 1), dim3(256), 0, 0, g, ap, en); }, 20);
    report("v2          M29696 K1024 N256 no epilogue", ms, 2.0 * M * 1024 * 256);
  }
  {  // edge MLP 2: M=E, K=256, N=256, LN
    GemmCore g = core(E, 256, 1, 256);
    APtrPlain ap{A, 256};
    EpiLN<ACT_NONE, ACT_NONE, false> e; memset(&e, 0, sizeof e);
    e.M = E; e.bias[0] = bias; e.gamma[0] = bias + 256; e.beta[0] = bias + 512; e.eps = 1e-5f; e.C = C; e.ldc = 256;
    float ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiLN<ACT_NONE, ACT_NONE, false>>), dim3(496, 1), dim3(256), 0, 0, g, ap, e); }, 20);
    report("v2 edge_mlp2 M63488 K256 N256 LN", ms, 2.0 * E * 256 * 256);
    EpiNull en{C};
    ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiNull>), dim3(496, 1), dim3(256), 0, 0, g, ap, en); }, 20);
    report("v2           M63488 K256 N256 no epilogue", ms, 2.0 * E * 256 * 256);
    GemmCore g2 = core(M, 256, 1, 256);
    e.M = M;
    ms = time_it([&] { hipLaunchKernelGGL((gemm_rows2_kernel<8, APtrPlain, EpiLN<ACT_NONE, ACT_NONE, false>>), dim3(232, 1), dim3(256), 0, 0, g2, ap, e); }, 20);
    report("v2 node_mlp2 M29696 K256 N256 LN", ms, 2.0 * M * 256 * 256);
  }
  printf("--- column-split 8-wave kernel\n");
  {
    EpiNull en{C};
    {
      GemmCore g = core(M, 256, 4, 256);
      APtrPlain ap{A, 256};
      EpiBiasAct<8, ACT_GELU, false> e; memset(&e, 0, sizeof e);
      e.M = M; e.n_valid = 256; e.col_block = 256; e.C = C; e.ldc = 1024; e.alpha = 1.f;
      for (int i = 0; i < 4; ++i) e.bias[i] = bias + i * 256;
      float ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiBiasAct<8, ACT_GELU, false>>), dim3(232, 4), dim3(512), 0, 0, g, ap, e); }, 20);
      report("cs ffn_up  M29696 K256 N1024 bias+GELU", ms, 2.0 * M * 256 * 1024);
      ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiNull>), dim3(232, 4), dim3(512), 0, 0, g, ap, en); }, 20);
      report("cs         M29696 K256 N1024 no epilogue", ms, 2.0 * M * 256 * 1024);
      GemmCore g3 = core(M, 256, 3, 256);
      ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiNull>), dim3(232, 3), dim3(512), 0, 0, g3, ap, en); }, 20);
      report("cs qkv     M29696 K256 N768 no epilogue", ms, 2.0 * M * 256 * 768);
    }
    {
      GemmCore g = core(M, 1024, 1, 1024);
      APtrPlain ap{A, 1024};
      EpiGate<true> e{M, bias, C2, mod, 0, 0, mod, 256, 512, mol, C, C + (size_t)M * 256, nullptr};
      float ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiGate<true>>), dim3(232, 1), dim3(512), 0, 0, g, ap, e); }, 20);
      report("cs ffn_down M29696 K1024 N256 gate+LNmod", ms, 2.0 * M * 1024 * 256);
      ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiNull>), dim3(232, 1), dim3(512), 0, 0, g, ap, en); }, 20);
      report("cs          M29696 K1024 N256 no epilogue", ms, 2.0 * M * 1024 * 256);
    }
    {
      GemmCore g = core(E, 256, 1, 256);
      APtrPlain ap{A, 256};
      EpiLN<ACT_NONE, ACT_NONE, false> e; memset(&e, 0, sizeof e);
      e.M = E; e.bias[0] = bias; e.gamma[0] = bias + 256; e.beta[0] = bias + 512; e.eps = 1e-5f; e.C = C; e.ldc = 256;
      float ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiLN<ACT_NONE, ACT_NONE, false>>), dim3(496, 1), dim3(512), 0, 0, g, ap, e); }, 20);
      report("cs edge_mlp2 M63488 K256 N256 LN", ms, 2.0 * E * 256 * 256);
      ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiNull>), dim3(496, 1), dim3(512), 0, 0, g, ap, en); }, 20);
      report("cs           M63488 K256 N256 no epilogue", ms, 2.0 * E * 256 * 256);
      GemmCore g2 = core(M, 256, 1, 256);
      e.M = M;
      ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiLN<ACT_NONE, ACT_NONE, false>>), dim3(232, 1), dim3(512), 0, 0, g2, ap, e); }, 20);
      report("cs node_mlp2 M29696 K256 N256 LN", ms, 2.0 * M * 256 * 256);
      ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiNull>), dim3(232, 1), dim3(512), 0, 0, g2, ap, en); }, 20);
      report("cs           M29696 K256 N256 no epilogue", ms, 2.0 * M * 256 * 256);
    }
  }
  printf("--- split-operand kernel (fp32 = 3 x bf16, six bf16 MFMA terms)\n");
  {
    uint16_t* Ws;
    CK(hipMalloc(&Ws, (size_t)1024 * 1024 * 6));
    auto split = [&](int K, int nb, int ldw, SplitW& sw) {
      for (int b = 0; b < nb; ++b) {
        uint16_t* dst = Ws + (size_t)b * (K / 16) * G4_CHUNK_ELEMS(4);
        hipLaunchKernelGGL(k_split_weight, dim3((256 * (K / 16) + 255) / 256), dim3(256), 0, 0, W + (size_t)b * 256 * ldw, ldw, 256, K, 256, dst);
        sw.blk[b] = dst;
      }
    };
    EpiNull en{C};
    {
      GemmCore g = core(M, 256, 4, 256);
      SplitW sw; memset(&sw, 0, sizeof sw);
      split(256, 4, 256, sw);
      APtrPlain ap{A, 256};
      EpiBiasAct<8, ACT_GELU, false> e; memset(&e, 0, sizeof e);
      e.M = M; e.n_valid = 256; e.col_block = 256; e.C = C; e.ldc = 1024; e.alpha = 1.f;
      for (int i = 0; i < 4; ++i) e.bias[i] = bias + i * 256;
      float ms;
      for (int rep = 0; rep < 4; ++rep) {
        ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiBiasAct<8, ACT_GELU, false>, 4>), dim3(232, 4), dim3(512), 0, 0, g, sw, ap, e); }, 50);
        report("split ffn_up M29696 K256 N1024 GELU", ms, 2.0 * M * 256 * 1024);
        ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiNull, 4, 3>), dim3(232, 4), dim3(512), 0, 0, g, sw, ap, en); }, 50);
        report("split A3     M29696 K256 N1024 no epi", ms, 2.0 * M * 256 * 1024);
        ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiNull, 4, 2>), dim3(232, 4), dim3(512), 0, 0, g, sw, ap, en); }, 50);
        report("split A2     M29696 K256 N1024 no epi", ms, 2.0 * M * 256 * 1024);
        ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiBiasAct<8, ACT_GELU, false>, 4, 2>), dim3(232, 4), dim3(512), 0, 0, g, sw, ap, e); }, 50);
        report("split A2 ffn_up M29696 K256 N1024 GELU", ms, 2.0 * M * 256 * 1024);
        ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiNull>), dim3(232, 4), dim3(512), 0, 0, g, ap, en); }, 50);
        report("fp32         M29696 K256 N1024 no epi", ms, 2.0 * M * 256 * 1024);
      }
      {  // 128-column blocks (two accumulators per wave): twice the workgroups
        uint16_t* Ws2;
        CK(hipMalloc(&Ws2, (size_t)1024 * 256 * 6));
        SplitW sw2; memset(&sw2, 0, sizeof sw2);
        for (int b = 0; b < 8; ++b) {
          uint16_t* dst = Ws2 + (size_t)b * 16 * G4_CHUNK_ELEMS(2);
          hipLaunchKernelGGL(k_split_weight, dim3((128 * 16 + 255) / 256), dim3(256), 0, 0, W + (size_t)b * 128 * 256, 256, 128, 256, 128, dst);
          sw2.blk[b] = dst;
        }
        GemmCore g4; memset(&g4, 0, sizeof g4);
        g4.M = M; g4.K = 256; g4.n_valid = 128; g4.ldw = 256;
        for (int i = 0; i < 8; ++i) g4.W[i] = W + (size_t)i * 128 * 256;
        EpiBiasAct<4, ACT_GELU, false> e4; memset(&e4, 0, sizeof e4);
        e4.M = M; e4.n_valid = 128; e4.col_block = 128; e4.C = C; e4.ldc = 1024; e4.alpha = 1.f;
        for (int i = 0; i < 8; ++i) e4.bias[i] = bias + i * 128;
        for (int rep = 0; rep < 3; ++rep) {
          ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiBiasAct<4, ACT_GELU, false>, 2>), dim3(232, 8), dim3(512), 0, 0, g4, sw2, ap, e4); }, 50);
          report("split 128-col ffn_up M29696 K256 N1024 GELU", ms, 2.0 * M * 256 * 1024);
          ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiBiasAct<8, ACT_GELU, false>, 4>), dim3(232, 4), dim3(512), 0, 0, g, sw, ap, e); }, 50);
          report("split 256-col ffn_up M29696 K256 N1024 GELU", ms, 2.0 * M * 256 * 1024);
        }
      }
      // accuracy: plain store epilogue, both kernels against a double-precision host product
      EpiBiasAct<8, ACT_NONE, false> e0; memcpy(&e0, &e, sizeof e);
      for (int i = 0; i < 4; ++i) e0.bias[i] = nullptr;
      std::vector<float> c32((size_t)256 * 1024), c16((size_t)256 * 1024);
      hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiBiasAct<8, ACT_NONE, false>>), dim3(232, 4), dim3(512), 0, 0, g, ap, e0);
      CK(hipMemcpy(c32.data(), C, c32.size() * 4, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiBiasAct<8, ACT_NONE, false>, 4>), dim3(232, 4), dim3(512), 0, 0, g, sw, ap, e0);
      CK(hipMemcpy(c16.data(), C, c16.size() * 4, hipMemcpyDeviceToHost));
      double e32 = 0, e16 = 0, nrm = 0;
      for (int m = 0; m < 256; ++m)
        for (int n = 0; n < 1024; n += 7) {
          double ref = 0;
          for (int k = 0; k < 256; ++k) ref += (double)h[(size_t)m * 256 + k] * (double)h[(size_t)n * 256 + k];
          e32 = fmax(e32, fabs(c32[(size_t)m * 1024 + n] - ref));
          e16 = fmax(e16, fabs(c16[(size_t)m * 1024 + n] - ref));
          nrm = fmax(nrm, fabs(ref));
        }
      printf("max |err| vs double: fp32 MFMA %.3e   split bf16 %.3e   (max |ref| %.3f)\n", e32, e16, nrm);
    }
    {
      GemmCore g = core(M, 1024, 1, 1024);
      SplitW sw; memset(&sw, 0, sizeof sw);
      split(1024, 1, 1024, sw);
      APtrPlain ap{A, 1024};
      EpiGate<true> e{M, bias, C2, mod, 0, 0, mod, 256, 512, mol, C, C + (size_t)M * 256, nullptr};
      float ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiGate<true>, 4>), dim3(232, 1), dim3(512), 0, 0, g, sw, ap, e); }, 50);
      report("split ffn_down M29696 K1024 N256 gate", ms, 2.0 * M * 1024 * 256);
      ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiNull, 4>), dim3(232, 1), dim3(512), 0, 0, g, sw, ap, en); }, 20);
      report("split          M29696 K1024 N256 no epi", ms, 2.0 * M * 1024 * 256);
    }
    {
      GemmCore g = core(E, 256, 1, 256);
      SplitW sw; memset(&sw, 0, sizeof sw);
      split(256, 1, 256, sw);
      APtrPlain ap{A, 256};
      EpiLN<ACT_NONE, ACT_NONE, false> e; memset(&e, 0, sizeof e);
      e.M = E; e.bias[0] = bias; e.gamma[0] = bias + 256; e.beta[0] = bias + 512; e.eps = 1e-5f; e.C = C; e.ldc = 256;
      float ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_split_kernel<APtrPlain, EpiLN<ACT_NONE, ACT_NONE, false>, 4>), dim3(496, 1), dim3(512), 0, 0, g, sw, ap, e); }, 50);
      report("split edge_mlp2 M63488 K256 N256 LN", ms, 2.0 * E * 256 * 256);
    }
  }
  printf("--- main-loop ablation (ffn_up shape, by=4; and ffn_down shape K=1024 by=1)\n");
  {
    EpiNull en{C};
    GemmCore g = core(M, 256, 4, 256);
    APtrPlain ap{A, 256};
    GemmCore gd = core(M, 1024, 1, 1024);
    APtrPlain apd{A, 1024};
    float ms;
#define RUN_DBG(MODE, name)                                                                                                   \
    ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_dbg<MODE>), dim3(232, 4), dim3(512), 0, 0, g, ap, en); }, 20);             \
    report(name " by=4 K=256", ms, 2.0 * M * 256 * 1024);                                                                     \
    ms = time_it([&] { hipLaunchKernelGGL((gemm_cs_dbg<MODE>), dim3(232, 1), dim3(512), 0, 0, gd, apd, en); }, 20);           \
    report(name " by=1 K=1024", ms, 2.0 * M * 1024 * 256);
    RUN_DBG(0, "dbg0 full")
    RUN_DBG(1, "dbg1 no DMA in loop")
    RUN_DBG(2, "dbg2 no DMA, no barrier")
    RUN_DBG(3, "dbg3 MFMA only")
  }
  printf("--- weight-gradient kernel: dW[nout,kin] += dY^T X over M rows\n");
  {
    auto run_dw = [&](int nout, int kin, int rows, int target, bool narrow = false) {
      const int bx = (nout + 127) / 128, by = narrow ? kin / 128 : kin / 256;
      int splits = (target + bx * by - 1) / (bx * by);
      int m_split = ((rows + splits - 1) / splits + 15) / 16 * 16;
      splits = (rows + m_split - 1) / m_split;
      GemmDwArgs a{A, nout, nout, A + (size_t)E * 512, kin, kin, rows, m_split, C, kin, nullptr};
      float ms = narrow ? time_it([&] { hipLaunchKernelGGL(gemm_dw_kernel<2>, dim3(bx, by, splits), dim3(512), 0, 0, a); }, 20)
                        : time_it([&] { hipLaunchKernelGGL(gemm_dw_kernel<4>, dim3(bx, by, splits), dim3(512), 0, 0, a); }, 20);
      char name[96];
      snprintf(name, sizeof name, "dW %4dx%4d rows %d blocks %d (m %d)%s", nout, kin, rows, bx * by * splits, m_split,
               narrow ? " narrow" : "");
      report(name, ms, 2.0 * rows * nout * kin);
    };
    for (int target : {128, 256, 384, 768}) run_dw(256, 256, M, target);
    for (int target : {128, 256, 384, 768}) run_dw(256, 256, M, target, true);
    run_dw(256, 256, E, 384, true);
    for (int target : {256, 384, 768}) run_dw(1024, 256, M, target);
    for (int target : {256, 384, 768}) run_dw(256, 1024, M, target);
    run_dw(768, 256, M, 384);
    run_dw(256, 256, E, 384);
  }
  printf("--- two independent GEMMs: back to back on one stream vs on two streams\n");
  {
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
    // node MLP-1 shape (M=N atoms, K=512, 256 columns, SiLU) and coordinate-gate shape (M=E edges, K=256, 128 columns)
    GemmCore ga = core(M, 512, 1, 512);
    APtrPlain apa{A, 512};
    EpiBiasAct<8, ACT_SILU, false> ea; memset(&ea, 0, sizeof ea);
    ea.M = M; ea.n_valid = 256; ea.col_block = 256; ea.C = C; ea.ldc = 256; ea.alpha = 1.f; ea.bias[0] = bias;
    GemmCore gb; memset(&gb, 0, sizeof gb);
    gb.M = E; gb.K = 256; gb.n_valid = 128; gb.ldw = 256; gb.W[0] = W;
    APtrPlain apb{A + (size_t)M * 512, 256};
    EpiBiasAct<4, ACT_SILU, false> eb; memset(&eb, 0, sizeof eb);
    eb.M = E; eb.n_valid = 128; eb.col_block = 128; eb.C = C2; eb.ldc = 128; eb.alpha = 1.f; eb.bias[0] = bias;
    auto la = [&](hipStream_t st) { hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiBiasAct<8, ACT_SILU, false>>), dim3(232, 1), dim3(512), 0, st, ga, apa, ea); };
    auto lb = [&](hipStream_t st) { hipLaunchKernelGGL((gemm_rows2_kernel<4, APtrPlain, EpiBiasAct<4, ACT_SILU, false>>), dim3(496, 1), dim3(256), 0, st, gb, apb, eb); };
    for (int mode = 0; mode < 2; ++mode) {
      for (int w = 0; w < 3; ++w) { la(s1); lb(mode ? s2 : s1); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      for (int r = 0; r < 20; ++r) {
        if (mode) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s2, ej, 0)); }
        la(s1); lb(mode ? s2 : s1);
        if (mode) { CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); }
      }
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-44s %8.1f us per pair\n", mode ? "two streams (fork/join per pair)" : "one stream", ms / 20 * 1e3);
    }
  }
  printf("--- cold operands: rotating over 6 buffer sets (A, resid, C each 30 MB; W 6 copies)\n");
  {
    float* big;
    CK(hipMalloc(&big, (size_t)6 * 3 * M * 256 * 4));
    CK(hipMemset(big, 0, (size_t)6 * 3 * M * 256 * 4));
    int it = 0;
    GemmCore g2 = core(M, 256, 1, 256);
    APtrPlain ap{A, 256};
    EpiLN<ACT_NONE, ACT_NONE, true> e; memset(&e, 0, sizeof e);
    e.M = M; e.bias[0] = bias; e.gamma[0] = bias + 256; e.beta[0] = bias + 512; e.eps = 1e-5f; e.ldc = 256; e.ldr = 256;
    auto run = [&](bool rotate) {
      const int k = rotate ? (it++ % 6) : 0;
      float* base = big + (size_t)k * 3 * M * 256;
      ap.A = base; e.resid = base + (size_t)M * 256; e.C = base + (size_t)2 * M * 256;
      g2.W[0] = W + (size_t)k * 65536;
      hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiLN<ACT_NONE, ACT_NONE, true>>), dim3(232, 1), dim3(512), 0, 0, g2, ap, e);
    };
    float ms = time_it([&] { run(false); }, 24);
    report("cs node_mlp2 +resid, same buffers", ms, 2.0 * M * 256 * 256);
    ms = time_it([&] { run(true); }, 24);
    report("cs node_mlp2 +resid, rotating buffers", ms, 2.0 * M * 256 * 256);
    auto run_inplace = [&](bool rotate) {
      const int k = rotate ? (it++ % 6) : 0;
      float* base = big + (size_t)k * 3 * M * 256;
      ap.A = base; e.resid = base + (size_t)M * 256; e.C = base + (size_t)M * 256;
      g2.W[0] = W + (size_t)k * 65536;
      hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiLN<ACT_NONE, ACT_NONE, true>>), dim3(232, 1), dim3(512), 0, 0, g2, ap, e);
    };
    ms = time_it([&] { run_inplace(true); }, 24);
    report("cs node_mlp2 +resid in place, rotating", ms, 2.0 * M * 256 * 256);
  }
  return 0;
}
