// Stand-alone check and timing of the fused feed-forward kernel against the two-launch path
// (developer tool, not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o chain_bench chain_bench.hip
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>
#include <type_traits>
#include <algorithm>

#include "../epilogues.h"
#include "../gemm3.h"
#include "ffn_chain.h"
#include "../edge_chain.h"
#include "../dit_chain.h"
#include "../node_chain.h"
#include "../kernels.h"

// register-only MFMA streams, one wave per SIMD: NACC accumulators used round-robin
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_chain(float* out, int iters) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32 / NACC; ++j)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

// one dependent MFMA chain with NV independent vector instructions in every gap (KIND 0: v_fma_f32,
// 1: v_or_b32, 2: v_rcp_f32)
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k_mfma_valu(float* out, int iters) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  float x[4] = {a, b, a + 1.f, b + 1.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 3]) : "v"(b));
        if (KIND == 1) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[v & 3]) : "v"(b));
        if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[v & 3]));
      }
    }
  }
  float s = x[0] + x[1] + x[2] + x[3];
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) out[0] = s;
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

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * scale;
  }
  float* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

#define STAMPRUN(A, name)       { \
        for (int r = 0; r < 50; ++r) hipLaunchKernelGGL((ffn_chain_kernel<true, 16 + A>), dim3(nblk), dim3(256), 0, 0, fa); \
        CK(hipDeviceSynchronize()); \
        std::vector<unsigned long long> st((size_t)nblk * 8); \
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost)); \
        std::vector<double> pro, loop, epi, clk, s1, s2; \
        for (int b = 0; b < nblk; ++b) { \
          const unsigned long long* q = &st[(size_t)b * 8]; \
          pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2])); \
          clk.push_back((double)(q[3] - q[0]) / (double)(q[5] - q[4]) * 100.0); s1.push_back((double)q[6]); s2.push_back((double)q[7]); \
        } \
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }; \
        printf("  [%s] stamps (median over workgroups): prologue+tile0 %.0f cyc, 32 tile steps %.0f cyc (%.0f per step, ideal 16384), epilogue %.0f cyc, clock %.0f MHz\n", \
               name, med(pro), med(loop), med(loop) / 32, med(epi), med(clk)); \
        { \
          unsigned long long r0 = ~0ull, r1 = 0, rs = 0; double tmax = 0, tmin = 1e30; \
          for (int b = 0; b < nblk; ++b) { \
            const unsigned long long* q = &st[(size_t)b * 8]; \
            r0 = std::min(r0, q[4]); r1 = std::max(r1, q[5]); rs = std::max(rs, q[4]); \
            tmax = std::max(tmax, (double)(q[3] - q[0])); tmin = std::min(tmin, (double)(q[3] - q[0])); \
          } \
          printf("  wall (first start -> last end) %.2f us, last start - first start %.2f us, per-WG cycles min %.0f max %.0f\n", (r1 - r0) * 0.01, (rs - r0) * 0.01, tmin, tmax); \
        } \
        printf("  stage 1 per step %.0f cyc, stage 2 per step %.0f cyc (ideal 8192 each)\n", med(s1) / 31, med(s2) / 32); \
      }

static float max_diff(const float* a, const float* b, size_t n, double* maxabs) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double d = 0, m = 0;
  for (size_t i = 0; i < n; ++i) {
    const double e = fabs((double)ha[i] - hb[i]);
    if (!(e <= d)) d = e;   // NaN propagates
    m = fmax(m, fabs((double)ha[i]));
  }
  *maxabs = m;
  return (float)d;
}

// edge path of an E(3) layer: three launches (edge pre, second edge linear + LN, coordinate gate) against the fused kernel
static void edge_test(int N, int E) {
  float* P = dev_rand((size_t)N * 512, 1.0f, 11);
  float* rbf = dev_rand((size_t)E * 32, 1.0f, 12);
  float* rel = dev_rand((size_t)E * 4, 1.0f, 13);
  float* Wen0 = dev_rand((size_t)256 * 544, 0.05f, 14);
  float* b1 = dev_rand(256, 0.1f, 15);
  float* W2 = dev_rand((size_t)256 * 256, 0.0625f, 16);
  float* b2 = dev_rand(256, 0.1f, 17);
  float* gamma = dev_rand(256, 1.0f, 18);
  float* beta = dev_rand(256, 0.1f, 19);
  float* Wc1 = dev_rand((size_t)128 * 256, 0.0625f, 20);
  float* bc1 = dev_rand(128, 0.1f, 21);
  float* wc2 = dev_rand(128, 0.1f, 22);
  float* bc2 = dev_rand(1, 0.1f, 23);
  std::vector<int> hs(E), hd(E);
  unsigned s = 99;
  for (int i = 0; i < E; ++i) {
    s = s * 1664525u + 1013904223u; hs[i] = (int)((s >> 8) % (unsigned)N);
    s = s * 1664525u + 1013904223u; hd[i] = (int)((s >> 8) % (unsigned)N);
  }
  int *src, *dst;
  CK(hipMalloc(&src, E * 4)); CK(hipMalloc(&dst, E * 4));
  CK(hipMemcpy(src, hs.data(), E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dst, hd.data(), E * 4, hipMemcpyHostToDevice));
  // P with b1 folded into its second half, as the fused kernel expects it
  std::vector<float> hP((size_t)N * 512), hb1(256);
  CK(hipMemcpy(hP.data(), P, hP.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb1.data(), b1, 256 * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < 256; ++c) hP[(size_t)i * 512 + 256 + c] += hb1[c];
  float* Pb;
  CK(hipMalloc(&Pb, hP.size() * 4));
  CK(hipMemcpy(Pb, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
  float *A1, *msgA, *msgB, *wrA, *wrB;
  CK(hipMalloc(&A1, (size_t)E * 256 * 4)); CK(hipMalloc(&msgA, (size_t)E * 256 * 4)); CK(hipMalloc(&msgB, (size_t)E * 256 * 4));
  CK(hipMalloc(&wrA, (size_t)E * 4 * 4)); CK(hipMalloc(&wrB, (size_t)E * 4 * 4));
  CK(hipMemset(wrA, 0, (size_t)E * 16)); CK(hipMemset(wrB, 0, (size_t)E * 16));
  const int nblk = (E + 127) / 128;
  GemmCore g0; memset(&g0, 0, sizeof g0);
  g0.M = E; g0.K = 32; g0.n_valid = 256; g0.ldw = 544; g0.W[0] = Wen0 + 512;
  EpiEdgePre e0{E, b1, P, src, dst, A1, nullptr};
  GemmCore g1; memset(&g1, 0, sizeof g1);
  g1.M = E; g1.K = 256; g1.n_valid = 256; g1.ldw = 256; g1.W[0] = W2;
  EpiLN<ACT_NONE, ACT_NONE, false> e1; memset(&e1, 0, sizeof e1);
  e1.M = E; e1.bias[0] = b2; e1.gamma[0] = gamma; e1.beta[0] = beta; e1.eps = 1e-5f; e1.C = msgA; e1.ldc = 256;
  GemmCore g2; memset(&g2, 0, sizeof g2);
  g2.M = E; g2.K = 256; g2.n_valid = 128; g2.ldw = 256; g2.W[0] = Wc1;
  EpiCoord e2{E, bc1, wc2, bc2, rel, wrA, nullptr, nullptr};
  auto three = [&] {
    hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiEdgePre>), dim3(nblk, 1), dim3(512), 0, 0, g0, APtrPlain{rbf, 32}, e0);
    hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiLN<ACT_NONE, ACT_NONE, false>>), dim3(nblk, 1), dim3(512), 0, 0, g1, APtrPlain{A1, 256}, e1);
    hipLaunchKernelGGL((gemm_rows2_kernel<4, APtrPlain, EpiCoord>), dim3(nblk, 1), dim3(256), 0, 0, g2, APtrPlain{msgA, 256}, e2);
  };
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)nblk * 8 * 8));
  EdgeChainArgs ea{E, rbf, rel, src, dst, Pb, Wen0 + 512, 544, W2, b2, gamma, beta, Wc1, bc1, wc2, bc2, msgB, wrB, stamps};
  auto one = [&] { hipLaunchKernelGGL(edge_chain_kernel<false>, dim3(nblk), dim3(256), 0, 0, ea); };
  three(); one();
  CK(hipDeviceSynchronize());
  double m1, m2;
  const float d1 = max_diff(msgA, msgB, (size_t)E * 256, &m1);
  const float d2 = max_diff(wrA, wrB, (size_t)E * 4, &m2);
  printf("edge path N=%d E=%d: max |msg diff| %.3e (max |msg| %.3f)  max |wrel diff| %.3e (max %.3f)\n", N, E, d1, m1, d2, m2);
#define EDGE_STAMPS(A, name) {\
    hipLaunchKernelGGL((edge_chain_kernel<true, A>), dim3(nblk), dim3(256), 0, 0, ea); \
    CK(hipDeviceSynchronize()); \
    std::vector<unsigned long long> st((size_t)nblk * 8); \
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost)); \
    std::vector<double> ph[5]; \
    unsigned long long t0 = ~0ull, t1 = 0; \
    for (int b = 0; b < nblk; ++b) { \
      for (int k = 0; k < 5; ++k) ph[k].push_back((double)(st[b * 8 + k + 1] - st[b * 8 + k])); \
      t0 = std::min(t0, st[b * 8]); t1 = std::max(t1, st[b * 8 + 5]); \
    } \
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }; \
    printf("  [%-22s] stamps (median cycles of wave 0): prologue %.0f, 8 a1/W2 steps %.0f (ideal 74880), LN+store %.0f, 4 coord steps %.0f (ideal 33280), tail %.0f\n", \
           name, med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4])); \
  }
  EDGE_STAMPS(0, "full") EDGE_STAMPS(1, "no gathers") EDGE_STAMPS(2, "no DMA") EDGE_STAMPS(4, "no barrier") EDGE_STAMPS(8, "no SiLU") EDGE_STAMPS(15, "MFMA + reads only")
  const float t3 = time_it(three, 20), t1 = time_it(one, 20);
  const double fl = 2.0 * E * (32.0 * 256 + 256.0 * 256 + 256.0 * 128);
  printf("  three launches %8.1f us (%6.1f TFLOP/s)   fused %8.1f us (%6.1f TFLOP/s)\n", t3 * 1e3, fl / (t3 * 1e-3) / 1e12, t1 * 1e3,
         fl / (t1 * 1e-3) / 1e12);
}

// everything between two attention kernels: out-projection launch + fused feed-forward launch + in-projection launch
// against the one fused kernel
static void dit_test(int M, int stride) {
  const int B = 1024;
  float* att = dev_rand((size_t)M * 256, 1.0f, 31);
  float* x = dev_rand((size_t)M * 256, 1.0f, 32);
  float* Wo = dev_rand((size_t)256 * 256, 0.0625f, 33);
  float* bo = dev_rand(256, 0.1f, 34);
  float* W1 = dev_rand((size_t)1024 * 256, 0.0625f, 35);
  float* b1 = dev_rand(1024, 0.1f, 36);
  float* W2 = dev_rand((size_t)256 * 1024, 0.03125f, 37);
  float* b2 = dev_rand(256, 0.1f, 38);
  float* Wq = dev_rand((size_t)768 * 256, 0.0625f, 39);
  float* bq = dev_rand(768, 0.1f, 40);
  float* mod = dev_rand((size_t)B * 1536, 0.5f, 41);
  float* modn = dev_rand((size_t)B * 1536, 0.5f, 42);
  std::vector<int> hm(M);
  for (int i = 0; i < M; ++i) hm[i] = (int)((long long)i * B / M);
  int* mol;
  CK(hipMalloc(&mol, M * 4));
  CK(hipMemcpy(mol, hm.data(), M * 4, hipMemcpyHostToDevice));
  float *X1, *XN2, *X2a, *XN1, *QA, *X2b, *QB;
  CK(hipMalloc(&X1, (size_t)M * 256 * 4)); CK(hipMalloc(&XN2, (size_t)M * 256 * 4)); CK(hipMalloc(&X2a, (size_t)M * 256 * 4));
  CK(hipMalloc(&XN1, (size_t)M * 256 * 4)); CK(hipMalloc(&QA, (size_t)M * 768 * 4));
  CK(hipMalloc(&X2b, (size_t)M * 256 * 4)); CK(hipMalloc(&QB, (size_t)M * 768 * 4));
  const int nblk = (M + 127) / 128;
  GemmCore g0; memset(&g0, 0, sizeof g0);
  g0.M = M; g0.K = 256; g0.n_valid = 256; g0.ldw = 256; g0.W[0] = Wo;
  EpiGate<true> e0{M, bo, x, mod, stride, 512, mod, 768, 1024, mol, X1, XN2, nullptr};
  FfnChainArgs fa{M, XN2, W1, b1, W2, b2, X1, mod, stride, 1280, modn, 0, 256, mol, X2a, XN1, nullptr};
  GemmCore g2; memset(&g2, 0, sizeof g2);
  g2.M = M; g2.K = 256; g2.n_valid = 256; g2.ldw = 256;
  for (int i = 0; i < 3; ++i) g2.W[i] = Wq + (size_t)i * 256 * 256;
  EpiBiasAct<8, ACT_NONE, false> e2; memset(&e2, 0, sizeof e2);
  e2.M = M; e2.n_valid = 256; e2.col_block = 256; e2.C = QA; e2.ldc = 768; e2.alpha = 1.f;
  for (int i = 0; i < 3; ++i) e2.bias[i] = bq + i * 256;
  auto three = [&] {
    hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiGate<true>>), dim3(nblk, 1), dim3(512), 0, 0, g0, APtrPlain{att, 256}, e0);
    hipLaunchKernelGGL(ffn_chain_kernel<true>, dim3(nblk), dim3(256), 0, 0, fa);
    hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiBiasAct<8, ACT_NONE, false>>), dim3(nblk, 3), dim3(512), 0, 0, g2, APtrPlain{XN1, 256}, e2);
  };
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)nblk * 8 * 8));
  DitChainArgs da; memset(&da, 0, sizeof da);
  da.M = M; da.att = att; da.x = x; da.xout = X2b; da.Wo = Wo; da.bo = bo; da.W1 = W1; da.b1 = b1; da.W2 = W2; da.b2 = b2;
  da.mod = mod; da.mod_next = modn; da.Wqkv = Wq; da.bqkv = bq; da.qkv = QB; da.stamps = stamps;
  auto one = [&] {
    hipLaunchKernelGGL((dit_chain_kernel<DIT_TAIL_QKV>), dim3(nblk), dim3(256), 0, 0, da);
  };
  three(); one();
  CK(hipDeviceSynchronize());
  double m1, m2;
  const float d1 = max_diff(X2a, X2b, (size_t)M * 256, &m1);
  const float d2 = max_diff(QA, QB, (size_t)M * 768, &m2);
  printf("attention-layer tail M=%d mod_stride=%d: max |x2 diff| %.3e (max |x2| %.3f)  max |qkv diff| %.3e (max %.3f)\n", M, stride, d1, m1, d2, m2);
  if (stride == 0) {
    hipLaunchKernelGGL((dit_chain_kernel<DIT_TAIL_QKV, false, true>), dim3(nblk), dim3(256), 0, 0, da);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nblk * 8);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ph[6];
    for (int b = 0; b < nblk; ++b)
      for (int k = 0; k < 6; ++k) ph[k].push_back((double)(st[b * 8 + k + 1] - st[b * 8 + k]));
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("  stamps (median cycles of wave 0): prologue %.0f, out-proj 8 steps %.0f (ideal 66560), gate+LN %.0f, FFN 64 steps %.0f (ideal 532480), gate+LN %.0f, in-proj 24 steps %.0f (ideal 199680)\n",
           med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), med(ph[5]));
  }
  const float t3 = time_it(three, 10), t1 = time_it(one, 10);
  const double fl = 2.0 * M * (256.0 * 256 + 2.0 * 256 * 1024 + 256.0 * 768);
  printf("  three launches %8.1f us (%6.1f TFLOP/s)   fused %8.1f us (%6.1f TFLOP/s)\n", t3 * 1e3, fl / (t3 * 1e-3) / 1e12, t1 * 1e3,
         fl / (t1 * 1e-3) / 1e12);
}

// node update of an E(3) layer + what follows it (next layer's per-node edge projections, or the first
// attention layer's pre-norm and in-projection): separate launches against the fused kernel
static void node_test(int M, int tail, int stride) {
  const int B = 1024;
  float* hin = dev_rand((size_t)M * 256, 1.0f, 51);
  float* agg = dev_rand((size_t)M * 256, 1.0f, 52);
  float* Wn1 = dev_rand((size_t)256 * 512, 0.05f, 53);
  float* b1 = dev_rand(256, 0.1f, 54);
  float* Wn2 = dev_rand((size_t)256 * 256, 0.0625f, 55);
  float* b2 = dev_rand(256, 0.1f, 56);
  float* gamma = dev_rand(256, 1.0f, 57);
  float* beta = dev_rand(256, 0.1f, 58);
  float* Wen0 = dev_rand((size_t)256 * 544, 0.05f, 59);
  float* b1e = dev_rand(256, 0.1f, 60);
  float* Wq = dev_rand((size_t)768 * 256, 0.0625f, 61);
  float* bq = dev_rand(768, 0.1f, 62);
  float* mod = dev_rand((size_t)B * 1536, 0.5f, 63);
  std::vector<int> hm(M);
  for (int i = 0; i < M; ++i) hm[i] = (int)((long long)i * B / M);
  int* mol;
  CK(hipMalloc(&mol, M * 4));
  CK(hipMemcpy(mol, hm.data(), M * 4, hipMemcpyHostToDevice));
  const int ldt = tail == NODE_TAIL_PROJ ? 512 : 768;
  float *hid, *hA, *hB, *xn, *tA, *tB;
  CK(hipMalloc(&hid, (size_t)M * 256 * 4)); CK(hipMalloc(&hA, (size_t)M * 256 * 4)); CK(hipMalloc(&hB, (size_t)M * 256 * 4));
  CK(hipMalloc(&xn, (size_t)M * 256 * 4)); CK(hipMalloc(&tA, (size_t)M * ldt * 4)); CK(hipMalloc(&tB, (size_t)M * ldt * 4));
  const int nblk = (M + 127) / 128;
  GemmCore g0; memset(&g0, 0, sizeof g0);
  g0.M = M; g0.K = 512; g0.n_valid = 256; g0.ldw = 512; g0.W[0] = Wn1;
  APtrConcat ap0{hin, agg, 256, 256, 256};
  EpiBiasAct<8, ACT_SILU, false> e0; memset(&e0, 0, sizeof e0);
  e0.M = M; e0.n_valid = 256; e0.col_block = 256; e0.bias[0] = b1; e0.C = hid; e0.ldc = 256; e0.alpha = 1.f;
  GemmCore g1; memset(&g1, 0, sizeof g1);
  g1.M = M; g1.K = 256; g1.n_valid = 256; g1.ldw = 256; g1.W[0] = Wn2;
  EpiLN<ACT_NONE, ACT_NONE, true> e1; memset(&e1, 0, sizeof e1);
  e1.M = M; e1.bias[0] = b2; e1.gamma[0] = gamma; e1.beta[0] = beta; e1.eps = 1e-5f; e1.C = hA; e1.ldc = 256; e1.resid = hin; e1.ldr = 256;
  GemmCore g2; memset(&g2, 0, sizeof g2);
  EpiBiasAct<8, ACT_NONE, false> e2; memset(&e2, 0, sizeof e2);
  e2.M = M; e2.n_valid = 256; e2.col_block = 256; e2.C = tA; e2.ldc = ldt; e2.alpha = 1.f;
  g2.M = M; g2.K = 256; g2.n_valid = 256;
  if (tail == NODE_TAIL_PROJ) { g2.ldw = 544; g2.W[0] = Wen0; g2.W[1] = Wen0 + 256; e2.bias[1] = b1e; }
  else { g2.ldw = 256; for (int i = 0; i < 3; ++i) { g2.W[i] = Wq + (size_t)i * 256 * 256; e2.bias[i] = bq + i * 256; } }
  auto sep = [&] {
    hipLaunchKernelGGL((gemm_cs_kernel<APtrConcat, EpiBiasAct<8, ACT_SILU, false>>), dim3(nblk, 1), dim3(512), 0, 0, g0, ap0, e0);
    hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiLN<ACT_NONE, ACT_NONE, true>>), dim3(nblk, 1), dim3(512), 0, 0, g1, APtrPlain{hid, 256}, e1);
    if (tail == NODE_TAIL_PROJ) {
      hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiBiasAct<8, ACT_NONE, false>>), dim3(nblk, 2), dim3(512), 0, 0, g2, APtrPlain{hA, 256}, e2);
    } else {
      hipLaunchKernelGGL(k_ln_mod, dim3((M + 15) / 16), dim3(256), 0, 0, hA, mod, stride, 0, 256, mol, xn, M);
      hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiBiasAct<8, ACT_NONE, false>>), dim3(nblk, 3), dim3(512), 0, 0, g2, APtrPlain{xn, 256}, e2);
    }
  };
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)nblk * 8 * 8));
  NodeChainArgs na{M, hin, agg, hB, Wn1, b1, Wn2, b2, gamma, beta, tail == NODE_TAIL_PROJ ? Wen0 : Wq, tail == NODE_TAIL_PROJ ? b1e : bq, tB, mod, stamps};
  auto one = [&] {
    if (tail == NODE_TAIL_PROJ) hipLaunchKernelGGL((node_chain_kernel<NODE_TAIL_PROJ>), dim3(nblk), dim3(256), 0, 0, na);
    else hipLaunchKernelGGL((node_chain_kernel<NODE_TAIL_QKV>), dim3(nblk), dim3(256), 0, 0, na);
  };
  sep(); one();
  CK(hipDeviceSynchronize());
  double m1, m2;
  const float d1 = max_diff(hA, hB, (size_t)M * 256, &m1);
  const float d2 = max_diff(tA, tB, (size_t)M * ldt, &m2);
  printf("node update M=%d tail=%s mod_stride=%d: max |h diff| %.3e (max |h| %.3f)  max |tail diff| %.3e (max %.3f)\n", M,
         tail == NODE_TAIL_PROJ ? "edge projections" : "pre-norm + in-projection", stride, d1, m1, d2, m2);
  if (tail == NODE_TAIL_PROJ) {
    hipLaunchKernelGGL((node_chain_kernel<NODE_TAIL_PROJ, true>), dim3(nblk), dim3(256), 0, 0, na);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nblk * 8);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ph[6];
    for (int b = 0; b < nblk; ++b)
      for (int k = 0; k < 6; ++k) ph[k].push_back((double)(st[b * 8 + k + 1] - st[b * 8 + k]));
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("  stamps (median cycles of wave 0): prologue %.0f, agg half %.0f, h half + SiLU %.0f, second linear %.0f (ideal 66560 each), LN + residual %.0f, tail 16 steps %.0f (ideal 133120)\n",
           med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), med(ph[5]));
  }
  const float t3 = time_it(sep, 10), t1 = time_it(one, 10);
  const double fl = 2.0 * M * (512.0 * 256 + 256.0 * 256 + 256.0 * ldt);
  printf("  separate launches %8.1f us (%6.1f TFLOP/s)   fused %8.1f us (%6.1f TFLOP/s)\n", t3 * 1e3, fl / (t3 * 1e-3) / 1e12, t1 * 1e3,
         fl / (t1 * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "node") == 0) {
    node_test(29696, NODE_TAIL_PROJ, 0);
    node_test(29696, NODE_TAIL_QKV, 0);
    node_test(29696 - 50, NODE_TAIL_QKV, 0);
    return 0;
  }
  if (argc > 1 && strcmp(argv[1], "dit") == 0) {
    dit_test(29696, 0);
    dit_test(29696 - 50, 0);
    return 0;
  }
  if (argc > 1 && strcmp(argv[1], "edge") == 0) {
    edge_test(29696, 63488);
    edge_test(29696, 63488 - 77);
    edge_test(4883, 505730);
    return 0;
  }
  const int M = argc > 1 ? atoi(argv[1]) : 29696;
  const int B = 1024;   // molecules (modulation rows when not shared)
  float* XN = dev_rand((size_t)M * 256, 1.0f, 1);
  float* W1 = dev_rand((size_t)1024 * 256, 0.0625f, 2);
  float* b1 = dev_rand(1024, 0.1f, 3);
  float* W2 = dev_rand((size_t)256 * 1024, 0.03125f, 4);
  float* b2 = dev_rand(256, 0.1f, 5);
  float* resid = dev_rand((size_t)M * 256, 1.0f, 6);
  float* mod = dev_rand((size_t)B * 1536, 0.5f, 7);
  std::vector<int> hm(M);
  for (int i = 0; i < M; ++i) hm[i] = (int)((long long)i * B / M);
  int* mol;
  CK(hipMalloc(&mol, M * 4));
  CK(hipMemcpy(mol, hm.data(), M * 4, hipMemcpyHostToDevice));
  float *U, *Xa, *XNa, *Xb, *XNb;
  CK(hipMalloc(&U, (size_t)M * 1024 * 4));
  CK(hipMalloc(&Xa, (size_t)M * 256 * 4)); CK(hipMalloc(&XNa, (size_t)M * 256 * 4));
  CK(hipMalloc(&Xb, (size_t)M * 256 * 4)); CK(hipMalloc(&XNb, (size_t)M * 256 * 4));
  const int nblk = (M + 127) / 128;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)nblk * 8 * 8));

  {
    const int iters = 256;   // 8192 MFMAs per wave = the fused kernel's count
    const double fl = 256.0 * 4 * iters * 32 * 4096;
    float ms = time_it([&] { hipLaunchKernelGGL(k_mfma_chain<1>, dim3(256), dim3(256), 0, 0, U, iters); }, 10);
    printf("register-only MFMA, 1 accumulator : %8.1f us (%6.1f TFLOP/s, %.1f cycles/MFMA at 2.4 GHz)\n", ms * 1e3, fl / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 32));
    ms = time_it([&] { hipLaunchKernelGGL(k_mfma_chain<2>, dim3(256), dim3(256), 0, 0, U, iters); }, 10);
    printf("register-only MFMA, 2 accumulators: %8.1f us (%6.1f TFLOP/s)\n", ms * 1e3, fl / (ms * 1e-3) / 1e12);
    ms = time_it([&] { hipLaunchKernelGGL(k_mfma_chain<4>, dim3(256), dim3(256), 0, 0, U, iters); }, 10);
    printf("register-only MFMA, 4 accumulators: %8.1f us (%6.1f TFLOP/s)\n", ms * 1e3, fl / (ms * 1e-3) / 1e12);
  }
  {
    const int iters = 256;
#define MV(NV, KIND, name) { float ms = time_it([&] { hipLaunchKernelGGL((k_mfma_valu<NV, KIND>), dim3(256), dim3(256), 0, 0, U, iters); }, 10); \
      printf("MFMA chain + %2d %-9s per gap: %8.1f us\n", NV, name, ms * 1e3); }
    MV(0, 0, "-") MV(2, 0, "v_fma_f32") MV(4, 0, "v_fma_f32") MV(8, 0, "v_fma_f32") MV(12, 0, "v_fma_f32")
    MV(4, 1, "v_or_b32") MV(8, 1, "v_or_b32") MV(2, 2, "v_rcp_f32") MV(4, 2, "v_rcp_f32")
  }
  for (int stride : {0, 1536}) {
    // two-launch path
    GemmCore g1; memset(&g1, 0, sizeof g1);
    g1.M = M; g1.K = 256; g1.n_valid = 256; g1.ldw = 256;
    for (int i = 0; i < 4; ++i) g1.W[i] = W1 + (size_t)i * 256 * 256;
    APtrPlain ap1{XN, 256};
    EpiBiasAct<8, ACT_GELU, false> e1; memset(&e1, 0, sizeof e1);
    e1.M = M; e1.n_valid = 256; e1.col_block = 256; e1.C = U; e1.ldc = 1024; e1.alpha = 1.f;
    for (int i = 0; i < 4; ++i) e1.bias[i] = b1 + i * 256;
    GemmCore g2; memset(&g2, 0, sizeof g2);
    g2.M = M; g2.K = 1024; g2.n_valid = 256; g2.ldw = 1024; g2.W[0] = W2;
    APtrPlain ap2{U, 1024};
    EpiGate<true> e2{M, b2, resid, mod, stride, 512, mod, 768, 1024, mol, Xa, XNa, nullptr};
    auto two = [&] {
      hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiBiasAct<8, ACT_GELU, false>>), dim3(nblk, 4), dim3(512), 0, 0, g1, ap1, e1);
      hipLaunchKernelGGL((gemm_cs_kernel<APtrPlain, EpiGate<true>>), dim3(nblk, 1), dim3(512), 0, 0, g2, ap2, e2);
    };
    FfnChainArgs fa{M, XN, W1, b1, W2, b2, resid, mod, stride, 512, mod, 768, 1024, mol, Xb, XNb, stamps};
    auto one = [&] { hipLaunchKernelGGL(ffn_chain_kernel<true>, dim3(nblk), dim3(256), 0, 0, fa); };
    two(); one();
    CK(hipDeviceSynchronize());
    std::vector<float> xa((size_t)M * 256), xb((size_t)M * 256), na((size_t)M * 256), nb((size_t)M * 256);
    CK(hipMemcpy(xa.data(), Xa, xa.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(xb.data(), Xb, xb.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(na.data(), XNa, na.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nb.data(), XNb, nb.size() * 4, hipMemcpyDeviceToHost));
    double dx = 0, dn = 0, mx = 0;
    size_t bad = 0;
    for (size_t i = 0; i < xa.size(); ++i) {
      dx = fmax(dx, fabs((double)xa[i] - xb[i])); dn = fmax(dn, fabs((double)na[i] - nb[i]));
      mx = fmax(mx, fabs((double)xa[i]));
      if (!(fabs((double)xa[i] - xb[i]) < 1e-3)) ++bad;
    }
    printf("mod_stride %4d: max |x diff| %.3e  max |xn diff| %.3e  (max |x| %.3f, %zu elements off)\n", stride, dx, dn, mx, bad);
    const double flops = 2.0 * M * 256 * 1024 * 2;
    float ms2 = time_it(two, 20), ms1 = time_it(one, 20);
    if (stride == 0) {
#define ABLRUN(A, name) { float msa = time_it([&] { hipLaunchKernelGGL((ffn_chain_kernel<true, A>), dim3(nblk), dim3(256), 0, 0, fa); }, 20); printf("  ablation %-34s %8.1f us\n", name, msa * 1e3); }
      STAMPRUN(0, "full") STAMPRUN(1, "no GELU") STAMPRUN(2, "no reads") STAMPRUN(4, "no DMA") STAMPRUN(7, "MFMA only+barrier")
      ABLRUN(1, "no GELU")
      ABLRUN(2, "no LDS reads in loop")
      ABLRUN(4, "no DMA in loop")
      ABLRUN(12, "no DMA, no barrier")
      ABLRUN(15, "MFMA only in loop")
    }
    printf("  two launches %8.1f us (%6.1f TFLOP/s)   fused %8.1f us (%6.1f TFLOP/s)\n", ms2 * 1e3, flops / (ms2 * 1e-3) / 1e12,
           ms1 * 1e3, flops / (ms1 * 1e-3) / 1e12);
  }
  return 0;
}
