// Fused node update of an E(3) layer (inference form), model1.py:109-114, plus what follows it:
//   hid = SiLU([h | agg] Wn1^T + b1)                                   [M,256]  (registers only)
//   h'  = h + LN(hid Wn2^T + b2; eps 1e-5) gamma + beta                 -> HBM (residual stream)
//   TAIL_PROJ: P' = h' [W1a' | W1b']^T + (0 | b1e')                     -> HBM: per-node halves of the NEXT layer's first edge linear
//   TAIL_QKV:  qkv = (LN(h'; eps 1e-6) (1 + scale_msa) + shift_msa) Wqkv^T + bqkv   -> HBM: first attention layer's input
// One workgroup = 4 waves = 128 atoms (layout and tile streaming: chain.h).  Ring of three 32-KiB buffers,
// tile s in buffer s % 3, fetched two steps ahead:
//   s = t        WIDE tiles of Wn1 rows 32t.., k = 256..511 (agg half of the first linear, all eight hidden tiles)
//   s = 8 + t    WIDE tiles of Wn1 rows 32t.., k = 0..255   (h half), then SiLU: the hidden slab
//   s = 16 + o   WIDE tiles of Wn2 rows 32o..
//   s = 24 + q   WIDE tiles of the tail's weight (16 for TAIL_PROJ, 24 for TAIL_QKV)
// Only two 128-register arrays are live at any time (agg + hidden, h + hidden, hidden + out, out + h: h is
// read again for the residual); a third one does not fit the register file beside the addresses and
// fragments, and a spilled slab is reloaded in front of every MFMA group that uses it.
#pragma once
#include "chain.h"
#include "chain_args.h"



#define NC_VEC_B1 0
#define NC_VEC_B2 256
#define NC_VEC_G 512
#define NC_VEC_BE 768
#define NC_VEC_BT 1024     // tail bias: 512 (first half zero) or 768
#define NC_VEC_MOD 1792    // 512: shift_msa | scale_msa
#define NC_VEC_FL 2304

template <int TAIL, bool DBG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void node_chain_kernel(NodeChainArgs p) {
  __shared__ __attribute__((aligned(1024))) float smem[3 * CH_TILE_FL + NC_VEC_FL];
  float* const vec = smem + 3 * CH_TILE_FL;
  const int tid = threadIdx.x;
  ChainLane cl;
  cl.init(smem, tid);
  const int h = cl.h;
  const int row = blockIdx.x * 128 + cl.wave * 32 + cl.fi;
  const int rr = row < p.M ? row : p.M - 1;
  constexpr int NQ = TAIL == NODE_TAIL_PROJ ? 16 : TAIL == NODE_TAIL_QKV ? 24 : 0;
  auto stamp = [&](int k) {
    if (DBG && tid == 0) p.stamps[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // piece n of ring tile s < 24
  auto fetch = [&](int s, int n) {
    float* buf = smem + (s % 3) * CH_TILE_FL;
    if (s < 16) cl.dma_wide(p.Wn1, 512, 32 * (s & 7), 256, s < 8 ? 256 : 0, buf, n);
    else cl.dma_wide(p.Wn2, 256, 32 * (s - 16), 256, 0, buf, n);
  };
  auto fetch_tail = [&](int q, int n) {
    if (q < NQ) {
      float* buf = smem + (q % 3) * CH_TILE_FL;
      if (TAIL == NODE_TAIL_PROJ) cl.dma_wide(p.Wt, 544, 32 * (q & 7), 256, 256 * (q >> 3), buf, n);
      else cl.dma_wide(p.Wt, 256, 32 * q, 768, 0, buf, n);
    }
  };
#pragma unroll
  for (int n = 0; n < 8; ++n) { fetch(0, n); fetch(1, n); }

  // small vectors -> LDS: the loads go out here, the LDS writes follow the loads of the slab rows (stage_write below),
  // so that the prologue waits for memory once
  constexpr int NV = NC_VEC_FL / 4;   // float4 slots
  f32x4 sv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i;
    const float* src = nullptr;
    if (idx < 64) src = p.b1 + idx * 4;
    else if (idx < 128) src = p.b2 + (idx - 64) * 4;
    else if (idx < 192) src = p.gamma + (idx - 128) * 4;
    else if (idx < 256) src = p.beta + (idx - 192) * 4;
    else if (TAIL == NODE_TAIL_PROJ && idx >= 320 && idx < 384) src = p.bt != nullptr ? p.bt + (idx - 320) * 4 : nullptr;
    else if (TAIL == NODE_TAIL_QKV && idx < 448) src = p.bt + (idx - 256) * 4;
    else if (TAIL == NODE_TAIL_QKV && idx >= 448 && idx < 576) src = p.mod + (idx - 448) * 4;
    sv[i] = src != nullptr ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto stage_write = [&]() {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + 256 * i;
      if (idx < NV) *reinterpret_cast<f32x4*>(vec + idx * 4) = sv[i];
    }
  };
  auto vec4 = [&](int off, int m) -> f32x4 { return *reinterpret_cast<const f32x4*>(vec + off + 8 * m + 4 * h); };

  auto open_step = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  auto slab4 = [&](f32x16 (&xs)[8], auto J) {
    constexpr int m = decltype(J)::value;
    return f32x4{xs[m >> 2][4 * (m & 3)], xs[m >> 2][4 * (m & 3) + 1], xs[m >> 2][4 * (m & 3) + 2], xs[m >> 2][4 * (m & 3) + 3]};
  };
  // WIDE step over buffer BF; hook(J) behind every group
  auto wide_step = [&](auto BF, f32x16& acc, f32x16 (&xs)[8], auto&& hook) {
    constexpr int bf = decltype(BF)::value;
    chain_groups<32>(
        [&](auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (bf < 2) return lds_read128<(j >> 3) * 256 + bf * CH_TILE_FL * 4>(cl.a_wide[j & 7]);
          else return lds_read128<(j >> 3) * 256>(cl.a_wide[j & 7] + 2 * CH_TILE_FL * 4);
        },
        [&](auto J) { return slab4(xs, J); }, [&](auto) -> f32x16& { return acc; }, hook);
  };
  auto init_tile = [&](f32x16& a, int off, int tile) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = vec4(off, tile * 4 + g);
      a[4 * g] = b[0]; a[4 * g + 1] = b[1]; a[4 * g + 2] = b[2]; a[4 * g + 3] = b[3];
    }
  };
  auto load_slab = [&](f32x16 (&xs)[8], const float* base) {
    const float* r = base + (size_t)rr * 256 + 4 * h;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(r + 8 * m);
      xs[m >> 2][4 * (m & 3)] = a[0]; xs[m >> 2][4 * (m & 3) + 1] = a[1]; xs[m >> 2][4 * (m & 3) + 2] = a[2]; xs[m >> 2][4 * (m & 3) + 3] = a[3];
    }
  };

  auto fetch_any = [&](auto S, int n) {   // ring tile S (compile-time), the tail's tiles included
    constexpr int s2 = decltype(S)::value;
    if constexpr (s2 < 24) fetch(s2, n);
    else fetch_tail(s2 - 24, n);
  };
  f32x16 hs[8], as[8], out[8], pre[8];
  auto step = [&](auto S, f32x16& acc, f32x16 (&xs)[8]) {   // ring step S: acc += tile . xs, fetch of tile S + 2 behind it
    constexpr int s0 = decltype(S)::value;
    wide_step(std::integral_constant<int, s0 % 3>{}, acc, xs, [&](auto J) {
      constexpr int j = decltype(J)::value;
      if (j >= 8 && j < 16) fetch_any(std::integral_constant<int, s0 + 2>{}, j - 8);
    });
  };
  load_slab(as, p.agg);
  stage_write();
  open_step();
  stamp(1);
  // ---- first linear, agg half: pre[t] = b1 + Wn1[32t.., 256..511] . agg
  static_for<0, 8>([&](auto T) {
    constexpr int t = decltype(T)::value;
    if (t > 0) open_step();
    init_tile(pre[t], NC_VEC_B1, t);
    step(std::integral_constant<int, t>{}, pre[t], as);
  });
  stamp(2);
  // ---- h half and SiLU: pre becomes the hidden slab
  load_slab(hs, p.h);
  static_for<0, 8>([&](auto T) {
    constexpr int t = decltype(T)::value;
    open_step();
    step(std::integral_constant<int, 8 + t>{}, pre[t], hs);
    silu_tile(pre[t]);
  });
  stamp(3);
  // ---- second linear: out[o] = b2 + Wn2[32o.., :] . hidden
  static_for<0, 8>([&](auto O) {
    constexpr int o = decltype(O)::value;
    open_step();
    init_tile(out[o], NC_VEC_B2, o);
    step(std::integral_constant<int, 16 + o>{}, out[o], pre);
  });
  stamp(4);
  load_slab(hs, p.h);   // again, for the residual

  // ---- h' = h + LN(out) gamma + beta
  {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += out[o][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = out[o][r] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const f32x4 gm = vec4(NC_VEC_G, m), be = vec4(NC_VEC_BE, m);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        hs[m >> 2][4 * (m & 3) + k] += ((out[m >> 2][4 * (m & 3) + k] - mean) * rstd) * gm[k] + be[k];
    }
  }
  stamp(5);
  float* const hrow = p.hout + (size_t)rr * 256 + 4 * h;
  auto store_h = [&](auto J) {   // one 16-byte piece per group
    constexpr int j = decltype(J)::value;
    if (row < p.M) *reinterpret_cast<f32x4*>(hrow + 8 * j) = slab4(hs, J);
  };
  if constexpr (TAIL == NODE_TAIL_NONE) {
    static_for<0, 32>(store_h);
  } else {
  // ---- tail: NQ output tiles over the new slab, three per trip
  f32x16 (&xs)[8] = TAIL == NODE_TAIL_QKV ? as : hs;   // the tail's B slab
  if (TAIL == NODE_TAIL_QKV) {   // pre-norm of the first attention layer: LN(h'; eps 1e-6) (1 + scale) + shift
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += hs[o][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = hs[o][r] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-6f);
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const f32x4 sh = vec4(NC_VEC_MOD, m);
      const f32x4 sc = vec4(NC_VEC_MOD + 256, m);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        as[m >> 2][4 * (m & 3) + k] = ((hs[m >> 2][4 * (m & 3) + k] - mean) * rstd) * (1.0f + sc[k]) + sh[k];
    }
  }
  constexpr int LDT = TAIL == NODE_TAIL_PROJ ? 512 : 768;
  float* const trow = p.tout + (size_t)rr * LDT + 4 * h;
  f32x16 qa[3];
  auto store_q = [&](f32x16& a, int tile, auto J) {   // 4 pieces of output tile `tile`, behind groups 0..3
    constexpr int j = decltype(J)::value;
    if (j < 4 && row < p.M)
      *reinterpret_cast<f32x4*>(trow + 32 * tile + 8 * j) = f32x4{a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]};
  };
#pragma unroll 1
  for (int q0 = 0; q0 < NQ; q0 += 3) {
    static_for<0, 3>([&](auto K) {
      constexpr int k = decltype(K)::value;
      const int q = q0 + k;
      if (q < NQ) {
        open_step();
        init_tile(qa[k], NC_VEC_BT, q);
        wide_step(K, qa[k], xs, [&](auto J) {
          constexpr int j = decltype(J)::value;
          if (j >= 8 && j < 16) fetch_tail(q + 2, j - 8);
          if (q == 0) store_h(J);                       // the first step also stores h'
          else store_q(qa[(k + 2) % 3], q - 1, J);
        });
      }
    });
  }
  static_for<0, 4>([&](auto J) { store_q(qa[(NQ - 1) % 3], NQ - 1, J); });
}
  stamp(6);
}
