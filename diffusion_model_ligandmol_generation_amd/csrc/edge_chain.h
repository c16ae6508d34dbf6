// Fused edge path of an E(3) layer (inference form), model1.py:96-107:
//   a1   = SiLU(Pa[src] + Pb[dst] + Wc rbf)             [E,256]  (Pa | Pb = h [W1a | W1b]^T + (0 | b1), per NODE)
//   msg  = LN(a1 W2^T + b2; eps 1e-5) gamma + beta      [E,256]  -> HBM (the aggregation reads it)
//   cw   = tanh(wc2 . SiLU(msg Wc1^T + bc1) + bc2);  wrel = rel * cw      -> HBM
// a1 and the coordinate gate's hidden layer never leave the registers (see chain.h for the layout).
//
// One workgroup = 4 waves = 128 edges (CSR order), wave w owns edges [32w, 32w + 32).  Per wave:
//   8 x { gather P rows -> a1^T tile (16 MFMAs over the resident Wc tile) -> SiLU -> 128 MFMAs into msg^T }
//   LayerNorm in registers, msg stored, then 4 x { 128 MFMAs over a Wc1 tile -> SiLU -> dot with wc2 }.
// Weight tiles: Wc stays in LDS; the eight TALL tiles of W2 and the four WIDE tiles of Wc1 go through
// a ring of three 32-KiB buffers, fetched two steps ahead.
#pragma once
#include "chain.h"
#include "chain_args.h"


//
// REDUCE (graphs of high in-degree): msg never reaches memory.  The edges are in CSR order, so the rows of one
// destination atom are consecutive; after the coordinate gate every wave lays its 32 x 256 msg tile out in LDS (the
// weight ring is free by then, 32 KiB per wave), lane l sums features 4l .. 4l + 3 over the rows of each destination
// in edge order, and the sum of destination a inside slab s goes to row a + s of `part`.  (a, s) -> a + s is
// one-to-one because a does not decrease with s, and the rows of one atom are consecutive: the aggregation kernel
// adds rows a + s_first .. a + s_last.  A 127-edge destination leaves 4-5 partial rows instead of 127 message rows.
// In this form a wave's slab is 32 consecutive edges of ONE molecule counted from the molecule's first edge (table
// slab_e0 / slab_cnt, built by k_slab_scan / k_slab_fill; a molecule's last slab may be short), so how an atom's
// messages are grouped into partial sums - and with it every bit of the result - depends on the molecule alone, not
// on what precedes it in the batch.
//
// GEOM: the layer's edge geometry and radial MLP (model1.py:88-93: rel = pos[dst] - pos[src], d = ||rel + 1e-8||,
// rbf = W2 SiLU(w0 d + b0) + b2) run in the prologue instead of a launch of their own: a lane builds the 16 hidden
// units of its half of the edge's row in registers (B layout), the 32 x 32 weight sits in LDS as the first 32 rows
// of a TALL tile, and the accumulator tile of that product IS the rbf slab the first edge linear multiplies.
template <bool DBG = false, int ABL = 0, bool REDUCE = false, bool GEOM = false>   // ABL: timing ablations (developer switch; results are wrong when non-zero)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void edge_chain_kernel(EdgeChainArgs p) {
  __shared__ __attribute__((aligned(1024))) float smem[4 * CH_TILE_FL + 1024 + (GEOM ? 1024 : 0)];
  float* const ring = smem + CH_TILE_FL;      // smem[0 .. 8192): the Wc tile
  float* const vec = smem + 4 * CH_TILE_FL;   // b2 | gamma | beta | bc1 | wc2
  float* const geo = vec + 1024;              // GEOM: radial_basis.2 weight, rows 0..31 of a TALL tile (4 KiB)
  const int tid = threadIdx.x;
  ChainLane cl;
  cl.init(smem, tid);
  const int fi = cl.fi, h = cl.h;
  auto stamp = [&](int k) {
    if (DBG && tid == 0) p.stamps[blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

#pragma unroll
  for (int n = 0; n < 8; ++n) cl.dma_tall(p.Wc, p.ldwc, 0, smem, n);
#pragma unroll
  for (int n = 0; n < 8; ++n) cl.dma_tall(p.W2, 256, 0, ring, n);
#pragma unroll
  for (int n = 0; n < 8; ++n) cl.dma_tall(p.W2, 256, 32, ring + CH_TILE_FL, n);
  if constexpr (GEOM) {
    if (cl.wave == 0) {   // pieces 0..3 of a TALL tile = rows 0..31
#pragma unroll
      for (int n = 0; n < 4; ++n) cl.dma_tall(p.rW2, 32, 0, geo, n);
    }
  }
  // prologue loads: the edge's end points first (the P gathers depend on them), everything else behind them, the LDS
  // write of the small vectors last, so that the wave waits for memory twice (indices, then the rest) and not four times
  int e_raw, e;
  bool valid;
  if constexpr (REDUCE) {
    const int slab = blockIdx.x * 4 + cl.wave;
    const int e_lo = p.slab_e0[slab];
    e_raw = e_lo + fi;
    valid = fi < p.slab_cnt[slab];
    e = valid ? e_raw : e_lo;   // (an unused slab has e_lo = 0, count 0: its lanes recompute edge 0 and store nothing)
  } else {
    e_raw = blockIdx.x * 128 + cl.wave * 32 + fi;
    valid = e_raw < p.E;
    e = valid ? e_raw : p.E - 1;
  }
  const int i_src = p.src[e], i_dst = p.dst[e];
  const float* svp = tid < 64 ? p.b2 + tid * 4 : tid < 128 ? p.gamma + (tid - 64) * 4
                   : tid < 192 ? p.beta + (tid - 128) * 4 : tid < 224 ? p.bc1 + (tid - 192) * 4 : p.wc2 + (tid - 224) * 4;
  const f32x4 sv = *reinterpret_cast<const f32x4*>(svp);
  f32x4 rb[4];
  f32x4 relv;
  if constexpr (GEOM) {
    const float* ps = p.pos + (size_t)i_src * 3;
    const float* pd = p.pos + (size_t)i_dst * 3;
    const float rx = pd[0] - ps[0], ry = pd[1] - ps[1], rz = pd[2] - ps[2];
    const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
    const float dist = sqrtf(ex * ex + ey * ey + ez * ez);
    relv = f32x4{rx, ry, rz, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {   // hidden units k = 8m + 4h + (0..3) of this lane's half of the row
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.rw0 + 8 * m + 4 * h);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.rb0 + 8 * m + 4 * h);
      rb[m] = f32x4{silu(dist * w0[0] + b0[0]), silu(dist * w0[1] + b0[1]), silu(dist * w0[2] + b0[2]), silu(dist * w0[3] + b0[3])};
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m) rb[m] = *reinterpret_cast<const f32x4*>(p.rbf + (size_t)e * 32 + 8 * m + 4 * h);
    relv = *reinterpret_cast<const f32x4*>(p.rel + (size_t)e * 4);
  }
  const float bc2 = p.bc2[0];
  const float* pa = p.P + (size_t)i_src * 512 + 4 * h;
  const float* pb = p.P + (size_t)i_dst * 512 + 256 + 4 * h;
  f32x4 ga[4], gb[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    ga[g] = *reinterpret_cast<const f32x4*>(pa + 8 * g);
    gb[g] = *reinterpret_cast<const f32x4*>(pb + 8 * g);
  }
  f32x4 rbias[4];
  if constexpr (GEOM) {
#pragma unroll
    for (int g = 0; g < 4; ++g) rbias[g] = *reinterpret_cast<const f32x4*>(p.rb2 + 8 * g + 4 * h);
  }
  *reinterpret_cast<f32x4*>(vec + tid * 4) = sv;
  __syncthreads();
  if constexpr (GEOM) {   // rbf^T = W2 hidden^T + b2: four MFMA groups over the 32 x 32 weight
    f32x16 ur;
#pragma unroll
    for (int g = 0; g < 4; ++g) { ur[4 * g] = rbias[g][0]; ur[4 * g + 1] = rbias[g][1]; ur[4 * g + 2] = rbias[g][2]; ur[4 * g + 3] = rbias[g][3]; }
    unsigned ag[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ag[g] = cl.a_tall[g] + (4 * CH_TILE_FL + 1024) * 4;
    chain_groups<4>([&](auto J) { return lds_read128<0>(ag[decltype(J)::value]); },
                    [&](auto J) { return rb[decltype(J)::value]; }, [&](auto) -> f32x16& { return ur; });
#pragma unroll
    for (int m = 0; m < 4; ++m) rb[m] = f32x4{ur[4 * m], ur[4 * m + 1], ur[4 * m + 2], ur[4 * m + 3]};
  }
  stamp(1);
  f32x16 msg[8];
#pragma unroll
  for (int o = 0; o < 8; ++o)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(vec + o * 32 + 8 * g + 4 * h);
      msg[o][4 * g] = b[0]; msg[o][4 * g + 1] = b[1]; msg[o][4 * g + 2] = b[2]; msg[o][4 * g + 3] = b[3];
    }

  // fragment addresses per LDS buffer (0 = the Wc tile, 1..3 = the ring): every offset below is then an
  // instruction immediate
  unsigned at[4][4], aw[4][8];
#pragma unroll
  for (int bf = 0; bf < 4; ++bf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) at[bf][g] = cl.a_tall[g] + bf * CH_TILE_FL * 4;
#pragma unroll
    for (int k = 0; k < 8; ++k) aw[bf][k] = cl.a_wide[k] + bf * CH_TILE_FL * 4;
  }
  // ---- a1 tiles and the second edge linear.  Ring step s: tile s is in buffer 1 + s % 3; tile s + 2 is
  // fetched during step s, after the loads of the step's own inputs were consumed (the compiler drains
  // the whole vector-memory queue whenever it waits for a load behind an LDS-DMA), and every wave waits
  // for all its memory operations before the barrier that opens step s + 1.
  static_for<0, 8>([&](auto T) {
    constexpr int t = decltype(T)::value;
    if (t > 0 && !(ABL & 4)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    f32x16 u;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) u[4 * g + q] = ga[g][q] + gb[g][q];
    chain_groups<4>([&](auto J) { return lds_read128<t * 4096>(at[0][decltype(J)::value]); },
                    [&](auto J) { return rb[decltype(J)::value]; }, [&](auto) -> f32x16& { return u; });
    if (!(ABL & 8)) silu_tile(u);
    chain_groups<32>(
        [&](auto J) { constexpr int j = decltype(J)::value; return lds_read128<(j >> 2) * 4096>(at[1 + t % 3][j & 3]); },
        [&](auto J) { constexpr int g = decltype(J)::value & 3; return f32x4{u[4 * g], u[4 * g + 1], u[4 * g + 2], u[4 * g + 3]}; },
        [&](auto J) -> f32x16& { return msg[decltype(J)::value >> 2]; },
        [&](auto J) {   // the next step's inputs, one instruction per group
          constexpr int j = decltype(J)::value;
          if (j < 8) {
            if (t + 1 < 8 && !(ABL & 1)) {
              if (j < 4) ga[j & 3] = *reinterpret_cast<const f32x4*>(pa + 32 * (t + 1) + 8 * (j & 3));
              else gb[j & 3] = *reinterpret_cast<const f32x4*>(pb + 32 * (t + 1) + 8 * (j & 3));
            }
          } else if (j < 16 && !(ABL & 2)) {
            float* nb = ring + ((t + 2) % 3) * CH_TILE_FL;
            if (t + 2 < 8) cl.dma_tall(p.W2, 256, 32 * (t + 2), nb, j - 8);
            else cl.dma_wide(p.Wc1, 256, 32 * (t - 6), 128, 0, nb, j - 8);
          }
        });
  });

  stamp(2);
  // ---- LayerNorm over the 256 features of each edge (the lane and its partner lane ^ 32 hold them)
  {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += msg[o][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = msg[o][r] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(vec + 256 + o * 32 + 8 * g + 4 * h);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(vec + 512 + o * 32 + 8 * g + 4 * h);
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          y[k] = ((msg[o][4 * g + k] - mean) * rstd) * gm[k] + bt[k];
          msg[o][4 * g + k] = y[k];
        }
      }
  }

  stamp(3);
  // ---- coordinate gate: 128 hidden units in four WIDE tiles (ring steps 8..11)
  float cs = 0.f;
  float* const mrow = p.msg + (size_t)e * 256 + 4 * h;
  static_for<0, 4>([&](auto C) {
    constexpr int c = decltype(C)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c == 0) {   // Wc1 tile 2 (the later steps fetch behind their MFMAs; this one stores msg there)
#pragma unroll
      for (int n = 0; n < 8; ++n) cl.dma_wide(p.Wc1, 256, 64, 128, 0, ring + (10 % 3) * CH_TILE_FL, n);
    }
    f32x16 hid;
    f32x4 wv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(vec + 768 + c * 32 + 8 * g + 4 * h);
      wv[g] = *reinterpret_cast<const f32x4*>(vec + 896 + c * 32 + 8 * g + 4 * h);
      hid[4 * g] = b[0]; hid[4 * g + 1] = b[1]; hid[4 * g + 2] = b[2]; hid[4 * g + 3] = b[3];
    }
    chain_groups<32>(
        [&](auto J) { constexpr int j = decltype(J)::value; return lds_read128<(j >> 3) * 256>(aw[1 + (c + 8) % 3][j & 7]); },
        [&](auto J) {
          constexpr int m = decltype(J)::value;
          return f32x4{msg[m >> 2][4 * (m & 3)], msg[m >> 2][4 * (m & 3) + 1], msg[m >> 2][4 * (m & 3) + 2], msg[m >> 2][4 * (m & 3) + 3]};
        },
        [&](auto) -> f32x16& { return hid; },
        [&](auto J) {   // the first step stores msg, one 16-byte piece per group
          constexpr int j = decltype(J)::value;
          if (!REDUCE && c == 0 && valid)
            *reinterpret_cast<f32x4*>(mrow + 8 * j) =
                f32x4{msg[j >> 2][4 * (j & 3)], msg[j >> 2][4 * (j & 3) + 1], msg[j >> 2][4 * (j & 3) + 2], msg[j >> 2][4 * (j & 3) + 3]};
          if (c == 1 && j < 8) cl.dma_wide(p.Wc1, 256, 96, 128, 0, ring + (11 % 3) * CH_TILE_FL, j);   // Wc1 tile 3
        });
    silu_tile(hid);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 4; ++k) cs += hid[4 * g + k] * wv[g][k];
  });
  stamp(4);
  cs += __shfl_xor(cs, 32);
  const float cw = tanhf(cs + bc2);
  if (h == 0 && valid) {
    float* wr = p.wrel + (size_t)e * 4;
    wr[0] = relv[0] * cw; wr[1] = relv[1] * cw; wr[2] = relv[2] * cw;
  }
  stamp(5);
  if constexpr (REDUCE) {
    __syncthreads();   // every wave has finished its last weight tile: the four 32-KiB buffers are free
    float* const tile = smem + cl.wave * CH_TILE_FL;   // [32 edges][64 slots of 16 B], slot ^ (edge & 15)
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tile + fi * 256 + (((8 * o + 2 * g + h) ^ (fi & 15)) << 2)) =
            f32x4{msg[o][4 * g], msg[o][4 * g + 1], msg[o][4 * g + 2], msg[o][4 * g + 3]};
    const int dst_l = valid ? i_dst : -1;
    const int slab = blockIdx.x * 4 + cl.wave;
    int cur = -1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int d = __builtin_amdgcn_readlane(dst_l, r);
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * 256 + ((cl.lane ^ (r & 15)) << 2));
      if (d != cur) {   // wave-uniform: a new destination starts at row r
        if (cur >= 0) *reinterpret_cast<f32x4*>(p.part + (size_t)(slab + cur) * 256 + cl.lane * 4) = acc;
        acc = v;
        cur = d;
      } else {
        acc += v;
      }
    }
    if (cur >= 0) *reinterpret_cast<f32x4*>(p.part + (size_t)(slab + cur) * 256 + cl.lane * 4) = acc;
  }
}
