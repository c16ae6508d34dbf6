// Where the split-operand GEMM's time goes: copies of its main loop with parts removed (results are garbage).
// Developer tool, not part of the library.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o split_ablate split_ablate.hip
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../epilogues.h"
#include "../gemm3.h"
#include "../gemm4.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct EpiNull {
  static constexpr int kStores = 0;
  float* C;
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

// MODE bits: 1 = no operand split (VALU), 2 = no DMA inside the loop, 4 = no barrier, 8 = no LDS fragment reads
template <int MODE, class EPI>
__global__ __launch_bounds__(512, 2) void split_dbg(GemmCore g, SplitW sw, APtrPlain ap, EPI epi) {
  constexpr int NTW = 4, WAVES = 8, NSA = 2;
  constexpr int NCOL = 64 * NTW;
  constexpr int A_FL = 128 * G2_ROWF;
  constexpr int CHUNK = 3 * 2 * NCOL * 8;
  constexpr int B_FL = CHUNK / 2;
  constexpr int PIECES = B_FL / 256;
  constexpr int PW = (PIECES + WAVES - 1) / WAVES;
  constexpr int PLANE_B = 2 * NCOL * 16;
  __shared__ __attribute__((aligned(1024))) float smem4[NSA * A_FL + 2 * B_FL + 4 * 256];
  float* const sA_base = smem4;
  float* const sB_base = smem4 + NSA * A_FL;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = wave & 3, half = wave >> 2;
  const int by = blockIdx.y;
  const int row_blk = blockIdx.x * 128;
  f32x16 acc[NTW];
#pragma unroll
  for (int ct = 0; ct < NTW; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const int pr = lane >> 2, ps = lane & 3;
  const int lslot = ps ^ ((pr >> 2) & 3);
  int a_row = row_blk + wave * 16 + pr;
  a_row = a_row < g.M ? a_row : g.M - 1;
  const uint16_t* bsrc[PW];
  int bdst[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    int p = wave * PW + i;
    p = p < PIECES ? p : PIECES - 1;
    bsrc[i] = sw.blk[by] + p * 512 + lane * 8;
    bdst[i] = p * 256;
  }
  auto issue = [&](int kc, int buf) {
    lds_dma16(ap.ptr(a_row, kc * G2_BK + lslot * 4), sA_base + buf * A_FL + wave * 256);
#pragma unroll
    for (int i = 0; i < PW; ++i)
      lds_dma16(reinterpret_cast<const float*>(bsrc[i] + (size_t)kc * CHUNK), sB_base + buf * B_FL + bdst[i]);
  };
  const int nk = g.K / G2_BK;
  const int fi = lane & 31, hk = lane >> 5;
  const int swz = (fi >> 2) & 3;
  const int off0 = ((0 * 2 + hk) ^ swz) * 4, off1 = ((1 * 2 + hk) ^ swz) * 4;
  const int b_lane = (hk * NCOL + half * 32 * NTW + fi) * 16;
  issue(0, 0);
  u32x4 b[2][3];
  for (int kc = 0; kc < nk; ++kc) {
    if (!(MODE & 2) || kc == 0) wait_vm_count<0>();
    if (!(MODE & 4)) __builtin_amdgcn_s_barrier();
    if (!(MODE & 2) && kc + 1 < nk) {
      if (MODE & 16) lds_dma16(ap.ptr(a_row, (kc + 1) * G2_BK + lslot * 4), sA_base + ((kc + 1) & 1) * A_FL + wave * 256);
      else issue(kc + 1, (kc + 1) & 1);
    }
    const int buf = (MODE & 2) ? 0 : (kc & 1);
    const float* sA = sA_base + buf * A_FL;
    const char* sB = reinterpret_cast<const char*>(sB_base + buf * B_FL) + b_lane;
    const float* arow = sA + (slab * 32 + fi) * G2_ROWF;
    f32x4 a0, a1;
    if (!(MODE & 8) || kc == 0) {
      a0 = *reinterpret_cast<const f32x4*>(arow + off0);
      a1 = *reinterpret_cast<const f32x4*>(arow + off1);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[0][pl] = *reinterpret_cast<const u32x4*>(sB + pl * PLANE_B);
    }
    u32x4 as[3];
    if (MODE & 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        as[0][j] = __float_as_uint(j < 2 ? a0[2 * j] : a1[2 * j - 4]);
        as[1][j] = __float_as_uint(j < 2 ? a0[2 * j + 1] : a1[2 * j - 3]);
        as[2][j] = as[0][j] ^ as[1][j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint32_t p0, p1, p2;
        split_pair(a0[2 * j], a0[2 * j + 1], p0, p1, p2);
        as[0][j] = p0; as[1][j] = p1; as[2][j] = p2;
        split_pair(a1[2 * j], a1[2 * j + 1], p0, p1, p2);
        as[0][2 + j] = p0; as[1][2 + j] = p1; as[2][2 + j] = p2;
      }
    }
    const bf16x8 A0 = __builtin_bit_cast(bf16x8, as[0]);
    const bf16x8 A1 = __builtin_bit_cast(bf16x8, as[1]);
    const bf16x8 A2 = __builtin_bit_cast(bf16x8, as[2]);
    if (MODE & 32) {
      // accumulators of a pair of column tiles take turns: consecutive MFMAs are independent
      u32x4 bb[2][3];
#pragma unroll
      for (int cp = 0; cp < NTW; cp += 2) {
        if (!(MODE & 8)) {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              bb[q][pl] = *reinterpret_cast<const u32x4*>(sB + pl * PLANE_B + (cp + q) * 32 * 16);
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bb[q][pl] = b[0][pl];
        }
        const bf16x8 B00 = __builtin_bit_cast(bf16x8, bb[0][0]), B01 = __builtin_bit_cast(bf16x8, bb[0][1]), B02 = __builtin_bit_cast(bf16x8, bb[0][2]);
        const bf16x8 B10 = __builtin_bit_cast(bf16x8, bb[1][0]), B11 = __builtin_bit_cast(bf16x8, bb[1][1]), B12 = __builtin_bit_cast(bf16x8, bb[1][2]);
        acc[cp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B00, acc[cp], 0, 0, 0);
        acc[cp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B10, acc[cp + 1], 0, 0, 0);
        acc[cp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B02, acc[cp], 0, 0, 0);
        acc[cp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B12, acc[cp + 1], 0, 0, 0);
        acc[cp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B01, acc[cp], 0, 0, 0);
        acc[cp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B11, acc[cp + 1], 0, 0, 0);
        acc[cp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B00, acc[cp], 0, 0, 0);
        acc[cp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B10, acc[cp + 1], 0, 0, 0);
        acc[cp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B01, acc[cp], 0, 0, 0);
        acc[cp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B11, acc[cp + 1], 0, 0, 0);
        acc[cp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B00, acc[cp], 0, 0, 0);
        acc[cp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B10, acc[cp + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int ct = 0; ct < NTW; ++ct) {
      const int bi = (MODE & 8) ? 0 : (ct & 1);
      const bf16x8 B0 = __builtin_bit_cast(bf16x8, b[bi][0]);
      const bf16x8 B1 = __builtin_bit_cast(bf16x8, b[bi][1]);
      const bf16x8 B2 = __builtin_bit_cast(bf16x8, b[bi][2]);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B0, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B2, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(MODE & 8) && ct + 1 < NTW) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          b[(ct + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(sB + pl * PLANE_B + (ct + 1) * 32 * 16);
      }
      if ((MODE & 16) && !(MODE & 2) && ct < PW && kc + 1 < nk)
        lds_dma16(reinterpret_cast<const float*>(bsrc[ct] + (size_t)(kc + 1) * CHUNK), sB_base + ((kc + 1) & 1) * B_FL + bdst[ct]);
      __builtin_amdgcn_sched_barrier(0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[ct], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  }
  HalfMailbox mb{smem4 + NSA * A_FL + 2 * B_FL, wave, lane};
  epi.cs(acc, row_blk + slab * 32, by, half, lane, mb);
}

template <class F>
static float time_it(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int MMAX = 131072;
  float *A, *W, *C, *bias;
  CK(hipMalloc(&A, (size_t)MMAX * 1024 * 4)); CK(hipMalloc(&W, (size_t)1024 * 1024 * 4));
  CK(hipMalloc(&C, (size_t)MMAX * 1024 * 4)); CK(hipMalloc(&bias, 4096 * 4));
  std::vector<float> h((size_t)1024 * 1024);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  for (size_t off = 0; off < (size_t)MMAX * 1024; off += h.size()) CK(hipMemcpy(A + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  uint16_t* Ws;
  CK(hipMalloc(&Ws, (size_t)1024 * 1024 * 6));
  struct Shape { int M, K, nb; };
  const Shape shapes[] = {{29696, 256, 1}, {29696, 256, 4}, {29696, 1024, 1}, {131072, 256, 1}, {131072, 256, 4}};
  for (const Shape& sh : shapes) {
    GemmCore g; memset(&g, 0, sizeof g);
    g.M = sh.M; g.K = sh.K; g.n_valid = 256; g.ldw = sh.K;
    SplitW sw; memset(&sw, 0, sizeof sw);
    for (int b = 0; b < sh.nb; ++b) {
      g.W[b] = W + (size_t)b * 256 * sh.K;
      uint16_t* dst = Ws + (size_t)b * (sh.K / 16) * G4_CHUNK_ELEMS(4);
      hipLaunchKernelGGL(k_split_weight, dim3((256 * (sh.K / 16) + 255) / 256), dim3(256), 0, 0, g.W[b], sh.K, 256, sh.K, 256, dst);
      sw.blk[b] = dst;
    }
    APtrPlain ap{A, sh.K};
    EpiBiasAct<8, ACT_NONE, false> e; memset(&e, 0, sizeof e);
    e.M = sh.M; e.n_valid = 256; e.col_block = 256; e.C = C; e.ldc = 256 * sh.nb; e.alpha = 1.f;
    for (int i = 0; i < sh.nb; ++i) e.bias[i] = bias + i * 256;
    EpiNull en{C};
    const dim3 grid((sh.M + 127) / 128, sh.nb);
    const double mfma_us = 2.0 * sh.M * sh.K * 256.0 * sh.nb * 6.0 / 2.5e15 * 1e6;
    printf("M %d K %d N %d (MFMA-only floor at 2.5 PF/s: %.1f us)\n", sh.M, sh.K, 256 * sh.nb, mfma_us);
#define RUN(MODE, EPIV, name) { float ms = time_it([&] { hipLaunchKernelGGL((split_dbg<MODE, decltype(EPIV)>), grid, dim3(512), 0, 0, g, sw, ap, EPIV); }, 30); printf("  %-46s %8.1f us\n", name, ms * 1e3); }
    RUN(0, e, "as shipped, bias + store epilogue")
    RUN(0, en, "as shipped, no epilogue")
    RUN(1, en, "no operand split")
    RUN(2, en, "no DMA in the loop")
    RUN(3, en, "no DMA, no operand split")
    RUN(6, en, "no DMA, no barrier")
    RUN(7, en, "no DMA, no barrier, no split")
    RUN(15, en, "MFMA only")
    RUN(8, en, "no LDS fragment reads (DMA, barrier, split stay)")
    RUN(32 | 15, en, "MFMA only, two accumulators taking turns")
    RUN(32, en, "two accumulators taking turns, no epilogue")
    RUN(32, e, "two accumulators taking turns, bias + store")
    RUN(16, en, "W-piece DMAs behind the MFMA groups, no epilogue")
    RUN(16, e, "W-piece DMAs behind the MFMA groups, bias + store")
  }
  CK(hipDeviceSynchronize());
  printf("ok\n");
  return 0;
}
