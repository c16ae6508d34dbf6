// (First of the chain kernels, superseded in the library by dit_chain.h; kept as the reference launch of chain_bench.)
// Fused feed-forward block of a DiT layer (inference form), model1.py:266-273 / :298-300:
//   u  = GELU(xn W1^T + b1)                  [M,1024]   (never leaves the chip)
//   x  = resid + gate[mol] * (u W2^T + b2)   [M,256]
//   xn' = LN(x; eps 1e-6, no affine) * (1 + scale[mol]) + shift[mol]   (LNMOD: the next pre-norm)
//
// Both products run TRANSPOSED on v_mfma_f32_32x32x2_f32 so that the first product's accumulator
// tile is the second product's B operand with no lane movement:
//   stage 1   U^T tile [32 hidden x 32 atoms] = W1 tile [32 x 256] . xn^T      (A = W1 rows, B = xn^T)
//   stage 2   O^T tile [32 out x 32 atoms]  += W2[out tile, hidden tile] . G   (A = W2 rows, B = G = GELU(U^T))
// The accumulator map (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) puts atom
// j = lane & 31 on the lane and hidden unit 8g + 4h + q (h = lane >> 5) in register 4g + q, and the
// B operand of an MFMA wants k = first / second element of a pair on lane half 0 / 1: register
// 4g + q therefore serves as B for the k pair {8g + q, 8g + 4 + q}, and the A operand (W2) is read
// with the same permutation (lane half h takes 4 consecutive hidden units at 8g + 4h).
//
// One workgroup = 4 waves = 128 atoms, one wave per SIMD; wave w owns atoms [32w, 32w + 32):
//   * its xn slab (32 x 256) stays in 128 VGPRs for the whole kernel (lane (j,h) holds
//     xn[j][8m + 4h + e] in xr[m][e]); the 32 x 256 output slab in 8 accumulators (128 registers);
//   * the 1024 hidden units go by in 32 tiles of 32.  Per tile a wave issues 128 + 128 MFMAs
//     and reads 32 KiB + 32 KiB of weights from LDS, shared by the four waves and fetched ONCE per
//     workgroup by LDS-DMA (global_load_lds_dwordx4) into double buffers, one barrier per tile;
//   * GELU of tile t runs on the VALU in the shadow of stage 1 of tile t + 1 (the matrix pipe has
//     no other wave to switch to);
//   * the residual rows are prefetched into the dead xn registers under the last stage 2.
// LDS images are XOR-swizzled on the source side of the DMA so that ds_read_b128 is conflict-free:
//   W1 tile: row i (1 KiB = 64 slots of 16 B): physical slot = logical ^ (i & 15)
//   W2 tile: row R (128 B = 8 slots):          physical slot = logical ^ ((R >> 1) & 7)
// In the transposed layout a LayerNorm row is a lane (plus its partner lane ^ 32), so the row
// statistics need one cross-half exchange instead of a half-wave reduction per row.
#pragma once
#include "../chain.h"

// GELU (erf form) with the constants of erf_fast2 folded: with t = |x| / sqrt 2 and
// p = 1 + a1 t + ... + a6 t^6, erf(t) = 1 - p^-16, so
//   GELU(x) = x Phi(x) = max(x, 0) - |x| r,   r = 1 / (2 p^16) = (2^(1/16) p)^-16.
// FC_Gk = a_k 2^(-k/2) 2^(1/16) is the coefficient of |x|^k: 6 FMAs, 4 squarings, one reciprocal,
// one multiply, one max, one subtract per element (the vector instructions of a wave do not overlap
// its fp32 MFMAs, so every one of them costs matrix time).  Absolute error <= 8e-7 like gelu_exact2.
#define FC_G0 1.0442737340927124f
#define FC_G1 0.052075162529945374f
#define FC_G2 0.02207699790596962f
#define FC_G3 0.003422739217057824f
#define FC_G4 3.9686136005911976e-05f
#define FC_G5 5.105520904180594e-05f
#define FC_G6 5.62129980608006e-06f
__device__ __forceinline__ float gelu_folded(float x) {
  const float t = fabsf(x);
  float p = t * FC_G6 + FC_G5;
  p = p * t + FC_G4; p = p * t + FC_G3; p = p * t + FC_G2; p = p * t + FC_G1; p = p * t + FC_G0;
  p = p * p; p = p * p; p = p * p; p = p * p;
  return fmaxf(x, 0.f) - fabsf(x * __builtin_amdgcn_rcpf(p));
}

#define FC_HID 1024
#define FC_D 256
#define FC_TILES (FC_HID / 32)
#define FC_TILE_FL 8192                       // floats in one 32-KiB weight tile
#define FC_LDS_BYTES (4 * FC_TILE_FL * 4)     // W1 x 2, W2 x 2

struct FfnChainArgs {
  int M;
  const float* XN;      // [M,256] modulated pre-norm input
  const float* W1;      // [1024,256]
  const float* b1;      // [1024]
  const float* W2;      // [256,1024]
  const float* b2;      // [256]
  const float* resid;   // [M,256]
  const float* mod; int mod_stride; int gate_off;     // adaLN rows of this layer ([B or 1, stride])
  const float* mod2; int shift_off; int scale_off;    // rows feeding the next pre-norm (LNMOD)
  const int32_t* atom_mol;
  float* X;             // [M,256] (may be resid)
  float* XNout;         // [M,256] (LNMOD; may be XN)
  unsigned long long* stamps;   // diagnostic builds (ABL & 16): [grid][8] shader-clock / real-time stamps
};

template <bool LNMOD, int ABL = 0>   // ABL: timing ablations (developer switch; results are wrong when non-zero)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ffn_chain_kernel(FfnChainArgs p) {
  __shared__ __attribute__((aligned(1024))) float smem[4 * FC_TILE_FL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31, h = lane >> 5;
  const int row = blockIdx.x * 128 + wave * 32 + fi;
  const int rr = row < p.M ? row : p.M - 1;

  // ---- weight tiles by LDS-DMA: 8 + 8 one-KiB pieces per wave and tile
  // W1 piece n = hidden row 8 wave + n of the tile (64 slots); lane l fills physical slot l
  // W2 piece n = out rows 8 (8 wave + n) .. + 7 (8 slots each); lane l fills row l >> 3, slot l & 7
  const int lx1 = lane ^ ((wave & 1) * 8);
  auto dma_w1 = [&](int t, int buf, int n) {
    const int i = wave * 8 + n;
    lds_dma16(p.W1 + (size_t)(t * 32 + i) * FC_D + (lx1 ^ n) * 4, smem + buf * FC_TILE_FL + i * 256);
  };
  auto dma_w2 = [&](int t, int buf, int n) {
    const int piece = wave * 8 + n;
    const int R = piece * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((R >> 1) & 7);
    lds_dma16(p.W2 + (size_t)R * FC_HID + t * 32 + ls * 4, smem + (2 + buf) * FC_TILE_FL + piece * 256);
  };
#pragma unroll
  for (int n = 0; n < 8; ++n) dma_w1(0, 0, n);

  // ---- this wave's xn slab
  f32x4 xr[32];
  {
    const float* xrow = p.XN + (size_t)rr * FC_D + 4 * h;
#pragma unroll
    for (int m = 0; m < 32; ++m) xr[m] = *reinterpret_cast<const f32x4*>(xrow + 8 * m);
  }
  // output accumulators start at b2 (register 4g + q of tile o is column 32 o + 8g + 4h + q)
  f32x16 out[8];
#pragma unroll
  for (int o = 0; o < 8; ++o)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + o * 32 + 8 * g + 4 * h);
      out[o][4 * g + 0] = b[0]; out[o][4 * g + 1] = b[1]; out[o][4 * g + 2] = b[2]; out[o][4 * g + 3] = b[3];
    }
  auto load_b1 = [&](int t, f32x16& u) {   // hidden-tile accumulators start at b1
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1 + t * 32 + 8 * g + 4 * h);
      u[4 * g + 0] = b[0]; u[4 * g + 1] = b[1]; u[4 * g + 2] = b[2]; u[4 * g + 3] = b[3];
    }
  };

  // ---- fragment reads.  The compiler's wait-count pass treats an LDS-DMA in flight as a flat
  // access that may return out of order and then drains the LDS queue (lgkmcnt(0)) before every
  // use of a fragment, which exposes a full LDS latency per group to the one wave on the SIMD.
  // The reads are therefore issued as asm (unknown to that pass) with explicit counted waits;
  // the barrier at the top of every tile (vmcnt(0) first) orders them behind the DMA writes.
  // LDS byte addresses: W1 fragment of group m = a1[m & 7] + 256 (m >> 3) + buffer offset,
  //                     W2 fragment of group 4 o + g = a2[g] + 4096 o + buffer offset.
  const unsigned lds0 = (unsigned)(size_t)(lds_float_ptr)smem;
  const int w1_x = (fi & 15) ^ h, w2_x = ((fi >> 1) & 7) ^ h;
  unsigned a1[8], a2[4];
#pragma unroll
  for (int k = 0; k < 8; ++k) a1[k] = lds0 + fi * 1024 + ((2 * k) ^ w1_x) * 16;
#pragma unroll
  for (int g = 0; g < 4; ++g) a2[g] = lds0 + 2 * FC_TILE_FL * 4 + fi * 128 + ((2 * g) ^ w2_x) * 16;

  f32x16 ucur, unext;
  f32x4 a[4];
  unsigned long long dbg_t0 = 0, dbg_s1 = 0, dbg_s2 = 0;
  float gx0 = 0.f, gx1 = 0.f, gt0 = 0.f, gt1 = 0.f, gp0 = 0.f, gp1 = 0.f;   // GELU state carried between groups
  // One tile step = 64 groups of 4 MFMAs: groups 0..31 stage 1 of tile t + 1 (WITH1), groups
  // 32..63 stage 2 of tile t.  Group i: issue the read of group i + 3, wait for the read of group
  // i, 4 MFMAs; between them a quarter of a GELU pair of tile t (stage-1 groups) or one DMA piece
  // (the first 16 stage-2 groups).
  auto tile_step = [&](auto with1, int t, unsigned off1, unsigned off2, auto gelu_on) {
    constexpr bool WITH1 = decltype(with1)::value;
    constexpr bool GELU = decltype(gelu_on)::value;
    constexpr int FIRST = WITH1 ? 0 : 32;
    const int t1 = t + 1, t2 = t + 2 < FC_TILES ? t + 2 : FC_TILES - 1;   // W1 tile t + 2 (clamped: a spare refill of the free buffer)
    auto issue_read = [&](auto J) {
      constexpr int j = decltype(J)::value;
      if (j < 32) a[j & 3] = lds_read128<(j >> 3) * 256>(a1[j & 7] + off1);
      else a[j & 3] = lds_read128<((j - 32) >> 2) * 4096>(a2[(j - 32) & 3] + off2);
    };
    static_for<FIRST, FIRST + 3>([&](auto J) { issue_read(J); });
    static_for<FIRST, 64>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if ((ABL & 16) && (i == FIRST || i == 32)) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (i == 32 && WITH1) dbg_s1 += now - dbg_t0;
        dbg_t0 = now;
      }
      if (i + 3 < 64 && !((ABL & 2))) issue_read(std::integral_constant<int, (i + 3 < 64 ? i + 3 : 63)>{});
      lds_wait_frag<(63 - i < 3 ? 63 - i : 3)>(a[i & 3]);
      __builtin_amdgcn_sched_barrier(0);
      if (i < 32) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          unext = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3][e], xr[i][e], unext, 0, 0, 0);
        // GELU of pair k = i >> 2 of tile t in four quarters, one per group (gelu_folded, step by
        // step; scalar instructions: packed fp32 issues slowly beside MFMAs)
        if (GELU && !(ABL & 1)) {
          const int k = i >> 2;
          if ((i & 3) == 0) {
            gx0 = ucur[2 * k]; gx1 = ucur[2 * k + 1];
            gt0 = fabsf(gx0); gt1 = fabsf(gx1);
            gp0 = gt0 * FC_G6 + FC_G5; gp1 = gt1 * FC_G6 + FC_G5;
            gp0 = gp0 * gt0 + FC_G4; gp1 = gp1 * gt1 + FC_G4;
            gp0 = gp0 * gt0 + FC_G3; gp1 = gp1 * gt1 + FC_G3;
          } else if ((i & 3) == 1) {
            gp0 = gp0 * gt0 + FC_G2; gp1 = gp1 * gt1 + FC_G2;
            gp0 = gp0 * gt0 + FC_G1; gp1 = gp1 * gt1 + FC_G1;
            gp0 = gp0 * gt0 + FC_G0; gp1 = gp1 * gt1 + FC_G0;
            gp0 = gp0 * gp0; gp1 = gp1 * gp1;
          } else if ((i & 3) == 2) {
            gp0 = gp0 * gp0; gp1 = gp1 * gp1; gp0 = gp0 * gp0; gp1 = gp1 * gp1;
            gp0 = gp0 * gp0; gp1 = gp1 * gp1;
            gp0 = __builtin_amdgcn_rcpf(gp0); gp1 = __builtin_amdgcn_rcpf(gp1);
          } else {
            ucur[2 * k] = fmaxf(gx0, 0.f) - fabsf(gx0 * gp0);
            ucur[2 * k + 1] = fmaxf(gx1, 0.f) - fabsf(gx1 * gp1);
          }
        }
        // every gap between two MFMAs takes at most three of the group's vector instructions
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
      } else {
        constexpr int o = (i - 32) >> 2, g = (i - 32) & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          out[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3][q], ucur[4 * g + q], out[o], 0, 0, 0);
        // the weight tiles of the coming steps: one DMA piece in each of the first 16 stage-2 groups
        // (stage 2 has nothing else in its MFMA gaps)
        if (WITH1 && !(ABL & 4) && i < 48) {
          if (i < 40) dma_w1(t2, t & 1, i - 32);
          else dma_w2(t1, t1 & 1, i - 40);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (ABL & 16) dbg_s2 += __builtin_amdgcn_s_memtime() - dbg_t0;
  };

  auto stamp = [&](int k) {
    if ((ABL & 16) && tid == 0) {
      p.stamps[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
      if (k == 0 || k == 3) p.stamps[blockIdx.x * 8 + 4 + (k ? 1 : 0)] = __builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);
  load_b1(0, unext);
  __syncthreads();                        // W1 tile 0 landed
  // stage 1 of tile 0; its MFMA shadow fetches W1 tile 1 and W2 tile 0
  {
    static_for<0, 3>([&](auto J) { constexpr int j = decltype(J)::value; a[j & 3] = lds_read128<(j >> 3) * 256>(a1[j & 7]); });
    static_for<0, 32>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (i + 3 < 32) a[(i + 3) & 3] = lds_read128<((i + 3) >> 3) * 256>(a1[(i + 3) & 7]);
      lds_wait_frag<(31 - i < 3 ? 31 - i : 3)>(a[i & 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        unext = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3][e], xr[i][e], unext, 0, 0, 0);
      if ((i & 1) == 0) {
        if (i < 16) dma_w1(1, 1, i >> 1);
        else dma_w2(0, 0, (i - 16) >> 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  ucur = unext;
  load_b1(1, unext);
  f32x16 bpre = unext;                     // b1 of tile t + 2, fetched a whole step before it is needed
  stamp(1);
#pragma unroll 1
  for (int t = 0; t < FC_TILES - 1; ++t) {
    // the DMAs issued during the previous step (W1 tile t + 1, W2 tile t) have landed in every
    // wave, and every wave is done with the buffers that this step refills
    if (!(ABL & 8)) __syncthreads();
    load_b1(t + 2 < FC_TILES ? t + 2 : FC_TILES - 1, bpre);
    tile_step(std::true_type{}, t, (unsigned)(((t + 1) & 1) * FC_TILE_FL * 4), (unsigned)((t & 1) * FC_TILE_FL * 4), std::true_type{});
    ucur = unext;
    unext = bpre;
  }
  {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) ucur[k] = gelu_folded(ucur[k]);
    // xn is dead: its registers take the residual rows while the last stage 2 runs
    const float* rrow = p.resid + (size_t)rr * FC_D + 4 * h;
#pragma unroll
    for (int m = 0; m < 32; ++m) xr[m] = *reinterpret_cast<const f32x4*>(rrow + 8 * m);
    tile_step(std::false_type{}, FC_TILES - 1, 0u, (unsigned)(((FC_TILES - 1) & 1) * FC_TILE_FL * 4), std::false_type{});
  }

  stamp(2);
  if ((ABL & 16) && tid == 0) { p.stamps[blockIdx.x * 8 + 6] = dbg_s1; p.stamps[blockIdx.x * 8 + 7] = dbg_s2; }
  // ---- epilogue: lane (j,h) holds columns 8m + 4h .. + 3 (m = 0..31) of atom row j
  float v[32][4];
  const size_t mo = (size_t)p.atom_mol[rr] * p.mod_stride;
  {
    const float* gp = p.mod + mo + p.gate_off + 4 * h;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const f32x4 gt = *reinterpret_cast<const f32x4*>(gp + 8 * m);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[m][q] = xr[m][q] + gt[q] * out[m >> 2][4 * (m & 3) + q];
    }
  }
  float mean = 0.f, rstd = 0.f;
  if (LNMOD) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 32; ++m) s += (v[m][0] + v[m][1]) + (v[m][2] + v[m][3]);
    s += __shfl_xor(s, 32);
    mean = s * (1.0f / 256.0f);
    float q2 = 0.f;
#pragma unroll
    for (int m = 0; m < 32; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float d = v[m][q] - mean; q2 += d * d; }
    q2 += __shfl_xor(q2, 32);
    rstd = 1.0f / sqrtf(q2 * (1.0f / 256.0f) + 1e-6f);
  }
  if (row < p.M) {
    float* xo = p.X + (size_t)row * FC_D + 4 * h;
    float* xno = p.XNout + (size_t)row * FC_D + 4 * h;
    const float* scp = p.mod2 + mo + p.scale_off + 4 * h;
    const float* shp = p.mod2 + mo + p.shift_off + 4 * h;
    // the scale / shift vectors are fetched eight column groups at a time, ahead of that batch's
    // stores: a load placed after a store through a possibly aliasing pointer stays behind it
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      f32x4 sc[8], sh[8];
      if (LNMOD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          sc[i] = *reinterpret_cast<const f32x4*>(scp + 8 * (mb * 8 + i));
          sh[i] = *reinterpret_cast<const f32x4*>(shp + 8 * (mb * 8 + i));
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = mb * 8 + i;
        *reinterpret_cast<f32x4*>(xo + 8 * m) = f32x4{v[m][0], v[m][1], v[m][2], v[m][3]};
        if (LNMOD) {
          f32x4 y;
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = ((v[m][q] - mean) * rstd) * (1.0f + sc[i][q]) + sh[i][q];
          *reinterpret_cast<f32x4*>(xno + 8 * m) = y;
        }
      }
    }
  }
  stamp(3);
}
