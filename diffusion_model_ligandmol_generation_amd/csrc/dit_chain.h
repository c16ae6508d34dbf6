// Fused attention-layer tail (inference form), model1.py:295-300 (+ :293 of the next layer):
//   x1  = x + gate_msa * (att Wo^T + bo)
//   xn2 = LN(x1; eps 1e-6, no affine) (1 + scale_mlp) + shift_mlp
//   x2  = x1 + gate_mlp * (GELU(xn2 W1^T + b1) W2^T + b2)                       -> HBM (residual stream)
//   DIT_TAIL_QKV:    qkv' = (LN(x2) (1 + scale_msa') + shift_msa') Wqkv'^T + bqkv'   -> HBM (the NEXT layer's attention input)
//   DIT_TAIL_HEADS:  y_k = LN(GELU(x2 W0_k^T + b0_k); eps 1e-5) g_k + be_k for the atom and the coordinate head (model1.py:636-655),
//                    atom_pred = y_a W3a^T + b3a [M,119],  pos_pred = y_p W3p^T + b3p + pos_upd [M,3]        -> HBM
// i.e. everything between two attention kernels as ONE launch; x1 makes one round trip through HBM
// (written behind the first feed-forward step, read back behind the last), the [M,1024] hidden
// activations, xn2 and the next layer's pre-norm never leave the registers (layout: chain.h).
//
// MOLWAVE form (molecules of at most 32 atoms): wave w of workgroup b owns molecule 4b + w instead of 32 consecutive
// rows, and the kernel STARTS with the layer's own pre-norm, in-projection and attention, head by head - q and k tiles in
// the transposed layout, the v tile with the operand roles swapped (keys in the registers, head dimension on the lane),
// so that S^T = K Q^T and O^T = V^T P^T take both operands straight from accumulator registers and the probability
// tile never moves: no attention launch, no qkv in memory (24 more ring steps in front, no DIT_TAIL_QKV).
//
// One workgroup = 4 waves = 128 atoms.  Weight tiles go through a ring of four 32-KiB LDS buffers, tile
// s in buffer s & 3, fetched two steps ahead behind the MFMAs of step s - 2:
//   s = 0..7          WIDE tiles of Wo          (out-projection, one output tile each)
//   s = 8 + 2t, 9 + 2t  WIDE tile t of W1, TALL tile t of W2   (t = 0..31)
//   s = 72 + q        WIDE tiles of Wqkv'       (q = 0..23, DIT_TAIL_QKV)
//   s = 72..79, 80..83, 84..91   WIDE tiles of W0_atom, W3_atom (rows clamped to 119), W0_pos   (DIT_TAIL_HEADS)
// The small vectors (the adaLN row - the kernels serve calls whose molecules share one timestep - and the biases) are
// staged in LDS once.
#pragma once
#include "chain.h"
#include "chain_args.h"


// GELU (erf form) of an accumulator tile, two elements per packed instruction: with t = |x| and
// p = sum_k c_k t^k (c_k = a_k 2^(-k/2) 2^(1/16), a_k of Abramowitz-Stegun 7.1.28),
//   GELU(x) = max(x, 0) - |x| / (2 (1 + a1 t/sqrt2 + ...)^16) = x / 2 + |x| (1/2 - p^-16).   |error| <= 8e-7
__device__ __forceinline__ void gelu_tile(f32x16& u) {
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const f32x2 x = {u[k], u[k + 1]};
    const f32x2 t = __builtin_elementwise_abs(x);
    f32x2 p = t * 5.62129980608006e-06f + 5.105520904180594e-05f;
    p = p * t + 3.9686136005911976e-05f;
    p = p * t + 0.003422739217057824f;
    p = p * t + 0.02207699790596962f;
    p = p * t + 0.052075162529945374f;
    p = p * t + 1.0442737340927124f;
    p = p * p; p = p * p; p = p * p; p = p * p;
    const f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    // max(x, 0) - t r = x / 2 + t (1/2 - r): two packed instructions instead of a multiply, two max and two subtracts
    const f32x2 y = x * 0.5f + t * (0.5f - r);
    u[k] = y[0]; u[k + 1] = y[1];
  }
}

#define DC_VEC_MOD 0        // 1536 floats: this layer's adaLN row
#define DC_VEC_MODN 1536    // 512: next layer's shift_msa | scale_msa
#define DC_VEC_BO 2048
#define DC_VEC_B2 2304
#define DC_VEC_B1 2560      // 1024
#define DC_VEC_BQKV 3584    // 768: in-projection bias (the next layer's for DIT_TAIL_QKV, this layer's for MOLWAVE)
#define DC_VEC_HB0 4352     // DIT_TAIL_HEADS: [2][256] hidden bias (atom, pos)
#define DC_VEC_HG 4864      //   [2][256] LayerNorm weight
#define DC_VEC_HBE 5376     //   [2][256] LayerNorm bias
#define DC_VEC_B3A 5888     //   128: atom head output bias (119 valid)
#define DC_VEC_W3P 6016     //   [3][256] coordinate head output weight
#define DC_VEC_FL 6784

template <int TAIL, bool MOLWAVE = false, bool DBG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dit_chain_kernel(DitChainArgs p) {
  constexpr bool TAIL_QKV = TAIL == DIT_TAIL_QKV;
  static_assert(!(MOLWAVE && TAIL_QKV), "the molecule-per-wave form computes its own in-projection");
  constexpr int SA = MOLWAVE ? 24 : 0;   // ring steps of the attention front
  __shared__ __attribute__((aligned(1024))) float smem[4 * CH_TILE_FL + DC_VEC_FL];
  float* const vec = smem + 4 * CH_TILE_FL;
  const int tid = threadIdx.x;
  ChainLane cl;
  cl.init(smem, tid);
  const int fi = cl.fi, h = cl.h;
  int rr, n_mol = 32;
  bool valid;
  if (MOLWAVE) {
    const int mol = blockIdx.x * 4 + cl.wave;
    const int mc = mol < p.B ? mol : p.B - 1;
    const int a0 = p.mol_ptr[mc];
    n_mol = p.mol_ptr[mc + 1] - a0;
    // (an empty molecule: every lane invalid, rows read from a valid address; its all-masked softmax stays in
    // this wave's registers and is never stored)
    rr = a0 + (fi < n_mol ? fi : (n_mol > 0 ? n_mol - 1 : 0));
    rr = rr < p.M ? rr : p.M - 1;
    valid = mol < p.B && fi < n_mol;
  } else {
    const int row = blockIdx.x * 128 + cl.wave * 32 + fi;
    rr = row < p.M ? row : p.M - 1;
    valid = row < p.M;
  }
  auto stamp = [&](int k) {
    if (DBG && tid == 0) p.stamps[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // piece n of ring tile s (uniform s)
  auto fetch = [&](int sa, int n) {
    float* buf = smem + (sa & 3) * CH_TILE_FL;
    const int s = sa - SA;
    if (MOLWAVE && s < 0) {   // attention front: head sa / 3, q / k / v rows of this layer's in-projection
      const int hd = sa / 3, kind = sa - 3 * hd;
      cl.dma_wide(p.Wqkv_self, 256, 256 * kind + 32 * hd, 768, 0, buf, n);
    } else if (s < 8) cl.dma_wide(p.Wo, 256, 32 * s, 256, 0, buf, n);
    else if (s < 72) {
      const int t = (s - 8) >> 1;
      if (s & 1) cl.dma_tall(p.W2, 1024, 32 * t, buf, n);
      else cl.dma_wide(p.W1, 256, 32 * t, 1024, 0, buf, n);
    } else if (TAIL_QKV) cl.dma_wide(p.Wqkv, 256, 32 * (s - 72), 768, 0, buf, n);
    else if (TAIL == DIT_TAIL_HEADS) {
      if (s < 80) cl.dma_wide(p.W0a, 256, 32 * (s - 72), 256, 0, buf, n);
      else if (s < 84) cl.dma_wide(p.W3a, 256, 32 * (s - 80), 119, 0, buf, n);
      else cl.dma_wide(p.W0p, 256, 32 * (s - 84), 256, 0, buf, n);
    }
  };
  constexpr int S_END = SA + (TAIL_QKV ? 96 : TAIL == DIT_TAIL_HEADS ? 92 : 72);   // ring steps s + SA below
#pragma unroll
  for (int n = 0; n < 8; ++n) { fetch(0, n); fetch(1, n); }

  // small vectors -> LDS: the loads go out here, the LDS writes follow the loads of the slab rows (stage_write below),
  // so that the prologue waits for memory once
  constexpr int NV = 1088;   // float4 slots
  f32x4 sv[5], hv[3];
  {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = tid + 256 * i;
      const float* src = nullptr;
      if (idx < 384) src = p.mod + idx * 4;
      else if (idx < 512) src = TAIL_QKV ? p.mod_next + (idx - 384) * 4 : nullptr;
      else if (idx < 576) src = p.bo + (idx - 512) * 4;
      else if (idx < 640) src = p.b2 + (idx - 576) * 4;
      else if (idx < 896) src = p.b1 + (idx - 640) * 4;
      else if (idx < 1088) src = TAIL_QKV ? p.bqkv + (idx - 896) * 4 : MOLWAVE ? p.bqkv_self + (idx - 896) * 4 : nullptr;
      sv[i] = src != nullptr ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  if (TAIL == DIT_TAIL_HEADS) {   // the heads' vectors (float4 slots 1088 ..): 128 x 6 hidden-layer vectors, 32 b3a, 192 W3p
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + 256 * i;
      const float* src = nullptr;
      if (idx < 64) src = p.b0a + idx * 4;
      else if (idx < 128) src = p.b0p + (idx - 64) * 4;
      else if (idx < 192) src = p.ga + (idx - 128) * 4;
      else if (idx < 256) src = p.gp + (idx - 192) * 4;
      else if (idx < 320) src = p.bea + (idx - 256) * 4;
      else if (idx < 384) src = p.bep + (idx - 320) * 4;
      else if (idx >= 416 && idx < 608) src = p.W3p + (idx - 416) * 4;
      hv[i] = src != nullptr ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx >= 384 && idx < 416) {   // 119 floats: not a multiple of four
#pragma unroll
        for (int k = 0; k < 4; ++k) hv[i][k] = (idx - 384) * 4 + k < 119 ? p.b3a[(idx - 384) * 4 + k] : 0.f;
      }
    }
  }
  auto stage_write = [&]() {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = tid + 256 * i;
      if (idx < NV) *reinterpret_cast<f32x4*>(vec + idx * 4) = sv[i];
    }
    if (TAIL == DIT_TAIL_HEADS) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i;
        if (idx < 608) *reinterpret_cast<f32x4*>(vec + DC_VEC_HB0 + idx * 4) = hv[i];
      }
    }
  };
  // adaLN vector `off` (0 shift_msa .. 1280 gate_mlp), features 8m + 4h .. + 3, for this lane's molecule
  auto modv = [&](int off, int m) -> f32x4 { return *reinterpret_cast<const f32x4*>(vec + DC_VEC_MOD + off + 8 * m + 4 * h); };
  auto modv_next = [&](int off, int m) -> f32x4 { return *reinterpret_cast<const f32x4*>(vec + DC_VEC_MODN + off + 8 * m + 4 * h); };
  auto vec4 = [&](int off, int m) -> f32x4 { return *reinterpret_cast<const f32x4*>(vec + off + 8 * m + 4 * h); };

  // fragment addresses: buffers 0 / 1 and 2 / 3 share a base each (the odd one sits 32 KiB further, inside
  // the 16-bit instruction offset); a register per address and buffer costs 48 registers and spills
  unsigned at2[4], aw2[8];
#pragma unroll
  for (int g = 0; g < 4; ++g) at2[g] = cl.a_tall[g] + 2 * CH_TILE_FL * 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) aw2[k] = cl.a_wide[k] + 2 * CH_TILE_FL * 4;
  // Every EVEN step opens with: all of this wave's memory operations done (the tiles fetched during the
  // previous two steps among them), then the workgroup barrier.  With four buffers and a fetch distance
  // of two that is enough for the odd step behind it as well: its tile was fetched two steps earlier, so
  // its pieces were waited for before this barrier, and the buffer that its own fetch refills held the
  // tile of the step before this barrier.
  auto open_step = [&](bool even = true) {
    if (even) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  // one WIDE step: acc += tile(buffer BF) . slab, slab group m = xs[m >> 2] registers 4 (m & 3) ..;  the
  // fetch of tile s + 2 rides behind groups 8..15, extra(J) behind every group
  auto wide_step = [&](auto BF, int s, f32x16& acc, f32x16 (&xs)[8], auto&& extra) {
    constexpr int bf = decltype(BF)::value;
    chain_groups<32>(
        [&](auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (bf < 2) return lds_read128<(j >> 3) * 256 + bf * CH_TILE_FL * 4>(cl.a_wide[j & 7]);
          else return lds_read128<(j >> 3) * 256 + (bf - 2) * CH_TILE_FL * 4>(aw2[j & 7]);
        },
        [&](auto J) {
          constexpr int m = decltype(J)::value;
          return f32x4{xs[m >> 2][4 * (m & 3)], xs[m >> 2][4 * (m & 3) + 1], xs[m >> 2][4 * (m & 3) + 2], xs[m >> 2][4 * (m & 3) + 3]};
        },
        [&](auto) -> f32x16& { return acc; },
        [&](auto J) {
          constexpr int j = decltype(J)::value;
          if (j >= 8 && j < 16 && s + SA + 2 < S_END) fetch(s + SA + 2, j - 8);
          extra(J);
        });
  };
  auto tall_step = [&](auto BF, int s, f32x16 (&out)[8], f32x16& u, auto&& extra) {
    constexpr int bf = decltype(BF)::value;
    chain_groups<32>(
        [&](auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (bf < 2) return lds_read128<(j >> 2) * 4096 + bf * CH_TILE_FL * 4>(cl.a_tall[j & 3]);
          else return lds_read128<(j >> 2) * 4096 + (bf - 2) * CH_TILE_FL * 4>(at2[j & 3]);
        },
        [&](auto J) { constexpr int g = decltype(J)::value & 3; return f32x4{u[4 * g], u[4 * g + 1], u[4 * g + 2], u[4 * g + 3]}; },
        [&](auto J) -> f32x16& { return out[decltype(J)::value >> 2]; },
        [&](auto J) {
          constexpr int j = decltype(J)::value;
          if (j >= 8 && j < 16 && s + SA + 2 < S_END) fetch(s + SA + 2, j - 8);
          extra(J);
        });
  };
  auto init_tile = [&](f32x16& a, int off, int tile) {   // accumulator tile <- bias vector (LDS)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = vec4(off, tile * 4 + g);
      a[4 * g] = b[0]; a[4 * g + 1] = b[1]; a[4 * g + 2] = b[2]; a[4 * g + 3] = b[3];
    }
  };
  // LN(v; eps 1e-6, no affine) (1 + scale) + shift over the 256 features of the lane's row
  auto ln_mod = [&](f32x16 (&v)[8], f32x16 (&y)[8], auto&& scale4, auto&& shift4) {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += v[o][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = v[o][r] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-6f);
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 sc = scale4(o * 4 + g), sh = shift4(o * 4 + g);
#pragma unroll
        for (int k = 0; k < 4; ++k) y[o][4 * g + k] = ((v[o][4 * g + k] - mean) * rstd) * (1.0f + sc[k]) + sh[k];
      }
  };

  f32x16 xs[8];   // the current B slab: (pre-norm,) att, then xn2, then the next layer's pre-norm
  f32x16 x1[8];
  auto load_rows = [&](f32x16 (&dst)[8], const float* base) {
    const float* r = base + (size_t)rr * 256 + 4 * h;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(r + 8 * m);
      dst[m >> 2][4 * (m & 3)] = a[0]; dst[m >> 2][4 * (m & 3) + 1] = a[1]; dst[m >> 2][4 * (m & 3) + 2] = a[2]; dst[m >> 2][4 * (m & 3) + 3] = a[3];
    }
  };
  if constexpr (!MOLWAVE) {
    load_rows(xs, p.att);
    stage_write();
    open_step();   // ring tiles 0, 1 and the small vectors are in LDS
  } else {
    // ---- attention front: pre-norm of the wave's molecule, then per head q / k / v tiles (ring steps 3 hd ..),
    // S^T = K Q^T, softmax over the molecule's keys, O^T = V^T P^T
    f32x16 xn[8];
    load_rows(x1, p.x);
    stage_write();
    open_step();
    ln_mod(x1, xn, [&](int m) { return modv(256, m); }, [&](int m) { return modv(0, m); });
    auto front_step = [&](auto S, f32x16& acc, auto SW) {   // ring step S over the pre-norm slab
      constexpr int s0 = decltype(S)::value, bf = s0 & 3;
      constexpr bool swap = decltype(SW)::value;
      chain_groups<32, swap>(
          [&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (bf < 2) return lds_read128<(j >> 3) * 256 + bf * CH_TILE_FL * 4>(cl.a_wide[j & 7]);
            else return lds_read128<(j >> 3) * 256 + (bf - 2) * CH_TILE_FL * 4>(aw2[j & 7]);
          },
          [&](auto J) {
            constexpr int m = decltype(J)::value;
            return f32x4{xn[m >> 2][4 * (m & 3)], xn[m >> 2][4 * (m & 3) + 1], xn[m >> 2][4 * (m & 3) + 2], xn[m >> 2][4 * (m & 3) + 3]};
          },
          [&](auto) -> f32x16& { return acc; },
          [&](auto J) {
            constexpr int j = decltype(J)::value;
            if (j >= 8 && j < 16) fetch(s0 + 2, j - 8);
          });
    };
    static_for<0, 8>([&](auto HD) {
      constexpr int hd = decltype(HD)::value;
      f32x16 qt, kt, vt;
      if (hd > 0) open_step(!((3 * hd) & 1));
      init_tile(qt, DC_VEC_BQKV, hd);
      front_step(std::integral_constant<int, 3 * hd>{}, qt, std::false_type{});
      open_step(!((3 * hd + 1) & 1));
      init_tile(kt, DC_VEC_BQKV, 8 + hd);
      front_step(std::integral_constant<int, 3 * hd + 1>{}, kt, std::false_type{});
      open_step(!((3 * hd + 2) & 1));
      {   // v tile with the roles swapped: keys in the registers, head dimension (and its bias) on the lane
        const float bv = vec[DC_VEC_BQKV + 512 + 32 * hd + fi];
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[r] = bv;
      }
      front_step(std::integral_constant<int, 3 * hd + 2>{}, vt, std::true_type{});
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[r], qt[r], st, 0, 0, 0);
      // lane = query; register 4g + q of lane half h = key 8g + 4h + q
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 8 * (r >> 2) + 4 * h + (r & 3);
        st[r] = key < n_mol ? st[r] * 0.17677669529663687f : -INFINITY;
        mx = fmaxf(mx, st[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] - mx); sum += st[r]; }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] *= inv;
      f32x16 ot;
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ot = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[r], st[r], ot, 0, 0, 0);
      xs[hd] = ot;
    });
    open_step(true);   // (SA = 24 is even: the first out-projection step opens with a barrier)
  }
  stamp(1);
  static_for<0, 8>([&](auto O) {
    constexpr int o = decltype(O)::value;
    if (o > 0) open_step(!(o & 1));
    init_tile(x1[o], DC_VEC_BO, o);
    wide_step(std::integral_constant<int, o & 3>{}, o, x1[o], xs, NoHook{});
  });
  stamp(2);
  float* const xrow_out = p.xout + (size_t)rr * 256 + 4 * h;
  {
    const float* xrow = p.x + (size_t)rr * 256 + 4 * h;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const f32x4 rs = *reinterpret_cast<const f32x4*>(xrow + 8 * m);
      const f32x4 gt = modv(512, m);
#pragma unroll
      for (int k = 0; k < 4; ++k) x1[m >> 2][4 * (m & 3) + k] = rs[k] + gt[k] * x1[m >> 2][4 * (m & 3) + k];
    }
    ln_mod(x1, xs, [&](int m) { return modv(1024, m); }, [&](int m) { return modv(768, m); });
  }

  stamp(3);
  // ---- feed-forward: ring steps 8 .. 71, two hidden tiles per trip (static buffer indices)
  f32x16 out[8];
  auto store_x1 = [&](auto J) {   // behind the first feed-forward step: x1 -> HBM, one 16-byte piece per group
    constexpr int j = decltype(J)::value;
    if (valid)
      *reinterpret_cast<f32x4*>(xrow_out + 8 * j) =
          f32x4{x1[j >> 2][4 * (j & 3)], x1[j >> 2][4 * (j & 3) + 1], x1[j >> 2][4 * (j & 3) + 2], x1[j >> 2][4 * (j & 3) + 3]};
  };
  auto load_x1 = [&](auto J) {    // behind the last one: back into the same registers
    constexpr int j = decltype(J)::value;
    const f32x4 v = *reinterpret_cast<const f32x4*>(xrow_out + 8 * j);
    x1[j >> 2][4 * (j & 3)] = v[0]; x1[j >> 2][4 * (j & 3) + 1] = v[1]; x1[j >> 2][4 * (j & 3) + 2] = v[2]; x1[j >> 2][4 * (j & 3) + 3] = v[3];
  };
  auto ffn_pair = [&](int t, auto FIRST, auto LAST) {   // hidden tiles t (buffers 0, 1) and t + 1 (buffers 2, 3)
    constexpr bool first = decltype(FIRST)::value, last = decltype(LAST)::value;
    f32x16 u;
    open_step();
    init_tile(u, DC_VEC_B1, t);
    if (first) wide_step(std::integral_constant<int, 0>{}, 8 + 2 * t, u, xs, store_x1);
    else wide_step(std::integral_constant<int, 0>{}, 8 + 2 * t, u, xs, NoHook{});
    gelu_tile(u);
    if (first) {   // (x1's registers are free from here on)
#pragma unroll
      for (int o = 0; o < 8; ++o) init_tile(out[o], DC_VEC_B2, o);
    }
    open_step(false);
    tall_step(std::integral_constant<int, 1>{}, 9 + 2 * t, out, u, NoHook{});
    open_step();
    init_tile(u, DC_VEC_B1, t + 1);
    wide_step(std::integral_constant<int, 2>{}, 10 + 2 * t, u, xs, NoHook{});
    gelu_tile(u);
    open_step(false);
    if (last) tall_step(std::integral_constant<int, 3>{}, 11 + 2 * t, out, u, load_x1);
    else tall_step(std::integral_constant<int, 3>{}, 11 + 2 * t, out, u, NoHook{});
  };
  ffn_pair(0, std::true_type{}, std::false_type{});
#pragma unroll 1
  for (int t = 2; t < 30; t += 2) ffn_pair(t, std::false_type{}, std::false_type{});
  ffn_pair(30, std::false_type{}, std::true_type{});

  stamp(4);
  // ---- x2 = x1 + gate_mlp * out
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const f32x4 gt = modv(1280, m);
#pragma unroll
    for (int k = 0; k < 4; ++k) x1[m >> 2][4 * (m & 3) + k] += gt[k] * out[m >> 2][4 * (m & 3) + k];
  }
  auto store_x2 = [&](auto J) {
    constexpr int j = decltype(J)::value;
    if (valid)
      *reinterpret_cast<f32x4*>(xrow_out + 8 * j) =
          f32x4{x1[j >> 2][4 * (j & 3)], x1[j >> 2][4 * (j & 3) + 1], x1[j >> 2][4 * (j & 3) + 2], x1[j >> 2][4 * (j & 3) + 3]};
  };
  if constexpr (TAIL == DIT_TAIL_NONE) {
    static_for<0, 32>(store_x2);
    return;
  }
  if constexpr (TAIL == DIT_TAIL_HEADS) {
    // hidden layer of head HD: yh = LN(GELU(x2 W0^T + b0)) g + be   (ring steps S0 .. S0 + 7)
    f32x16 yh[8];
    auto head_hidden = [&](auto HD, auto S0) {
      constexpr int hd = decltype(HD)::value, s0 = decltype(S0)::value;
      static_for<0, 8>([&](auto O) {
        constexpr int o = decltype(O)::value;
        open_step(!((s0 + o) & 1));
        init_tile(yh[o], DC_VEC_HB0 + hd * 256, o);
        if (s0 + o == 72) wide_step(std::integral_constant<int, (s0 + o) & 3>{}, s0 + o, yh[o], x1, store_x2);
        else wide_step(std::integral_constant<int, (s0 + o) & 3>{}, s0 + o, yh[o], x1, NoHook{});
        gelu_tile(yh[o]);
      });
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += yh[o][r];
      s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / 256.0f);
      float q = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = yh[o][r] - mean; q += d * d; }
      q += __shfl_xor(q, 32);
      const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        const f32x4 gm = vec4(DC_VEC_HG + hd * 256, m), be = vec4(DC_VEC_HBE + hd * 256, m);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          yh[m >> 2][4 * (m & 3) + k] = ((yh[m >> 2][4 * (m & 3) + k] - mean) * rstd) * gm[k] + be[k];
      }
    };
    head_hidden(std::integral_constant<int, 0>{}, std::integral_constant<int, 72>{});
    // atom head output: four tiles of 32 columns (119 valid), row stride 119 floats -> dword stores
    {
      float* const arow = p.atom_pred + (size_t)rr * 119 + 4 * h;
      static_for<0, 4>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        f32x16 acc;
        open_step(!(q & 1));
        init_tile(acc, DC_VEC_B3A, q);
        wide_step(std::integral_constant<int, (80 + q) & 3>{}, 80 + q, acc, yh, NoHook{});
        if (valid) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (32 * q + 8 * g + 4 * h + k < 119) arow[32 * q + 8 * g + k] = acc[4 * g + k];
        }
      });
    }
    head_hidden(std::integral_constant<int, 1>{}, std::integral_constant<int, 84>{});
    // coordinate head output: three dot products over the lane's 128 features, completed by the partner lane
    {
      float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 32; ++m)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const f32x4 w = vec4(DC_VEC_W3P + o * 256, m);
#pragma unroll
          for (int k = 0; k < 4; ++k) d[o] += yh[m >> 2][4 * (m & 3) + k] * w[k];
        }
#pragma unroll
      for (int o = 0; o < 3; ++o) d[o] += __shfl_xor(d[o], 32);
      if (h == 0 && valid) {
#pragma unroll
        for (int o = 0; o < 3; ++o) p.pos_pred[(size_t)rr * 3 + o] = d[o] + p.b3p[o] + p.pos_upd[(size_t)rr * 3 + o];
      }
    }
    return;
  }

  // ---- the next layer's pre-norm and in-projection: ring steps 72 .. 95, four output tiles per trip
  ln_mod(x1, xs, [&](int m) { return modv_next(256, m); }, [&](int m) { return modv_next(0, m); });
  stamp(5);
  float* const qrow = p.qkv + (size_t)rr * 768 + 4 * h;
  f32x16 qa, qb;
  auto store_q = [&](f32x16& a, int tile, auto J) {   // 4 pieces of output tile `tile`, behind groups 0..3
    constexpr int j = decltype(J)::value;
    if (j < 4 && valid)
      *reinterpret_cast<f32x4*>(qrow + 32 * tile + 8 * j) = f32x4{a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]};
  };
  auto qkv_quad = [&](int q, auto FIRST) {
    constexpr bool first = decltype(FIRST)::value;
    open_step();
    init_tile(qa, DC_VEC_BQKV, q);
    if (first) wide_step(std::integral_constant<int, 0>{}, 72 + q, qa, xs, store_x2);
    else wide_step(std::integral_constant<int, 0>{}, 72 + q, qa, xs, [&](auto J) { store_q(qb, q - 1, J); });
    open_step(false);
    init_tile(qb, DC_VEC_BQKV, q + 1);
    wide_step(std::integral_constant<int, 1>{}, 73 + q, qb, xs, [&](auto J) { store_q(qa, q, J); });
    open_step();
    init_tile(qa, DC_VEC_BQKV, q + 2);
    wide_step(std::integral_constant<int, 2>{}, 74 + q, qa, xs, [&](auto J) { store_q(qb, q + 1, J); });
    open_step(false);
    init_tile(qb, DC_VEC_BQKV, q + 3);
    wide_step(std::integral_constant<int, 3>{}, 75 + q, qb, xs, [&](auto J) { store_q(qa, q + 2, J); });
  };
  qkv_quad(0, std::true_type{});
#pragma unroll 1
  for (int q = 4; q < 24; q += 4) qkv_quad(q, std::false_type{});
  static_for<0, 4>([&](auto J) { store_q(qb, 23, J); });
  stamp(6);
}
