// Building blocks of the fused "chain" kernels (ffn_chain.h, edge_chain.h, ...): several Linear layers
// of one row slab evaluated back to back with the activations in registers.
//
// Every product runs TRANSPOSED on v_mfma_f32_32x32x2_f32: the A operand is a 32-row tile of the weight
// matrix (read from LDS), the B operand is the activation slab of the wave's 32 rows (registers), and
// the accumulator tile is [32 output features x 32 rows]:
//   lane (j, h) = (lane & 31, lane >> 5) holds row j; register 4g + q holds output feature 8g + 4h + q.
// A register slab in "B layout" holds, for row j, feature 8m + 4h + e in element e of group m; one MFMA
// group = one ds_read_b128 of the weight tile + 4 MFMAs (k pairs {8m + e, 8m + 4 + e}).  An accumulator
// tile IS a 32-feature slab in B layout (group g = registers 4g .. 4g + 3), so the output of one Linear
// feeds the next without any lane movement, and a row's LayerNorm statistics are a per-lane sum plus one
// exchange with lane ^ 32.
//
// Why this shape: on gfx950 the fp32 MFMAs and the vector ALU do not overlap - not within a wave and
// not between waves of one SIMD (csrc/tools/overlap_bench.hip: MFMA-only 237 us, VALU-only 69 us, both
// 293 us) - so a kernel's time is its MFMA cycles plus its VALU cycles, and what a fused kernel can win
// is (1) memory round trips and launch boundaries and (2) VALU instructions: in this layout a lane owns
// contiguous 16-byte runs of ONE row, so epilogues need no cross-lane reductions and a quarter of the
// load / store instructions of the row-major form.
//
// Weights stream through LDS in 32-KiB tiles fetched by LDS-DMA (global_load_lds_dwordx4), shared by the
// four waves of the workgroup (one wave per SIMD, 32 rows each):
//   WIDE tile  [32 weight rows x 256 k]   (1-KiB rows): one output tile over K = 256
//   TALL tile  [256 weight rows x 32 k]   (128-B rows): eight output tiles over K = 32
// LDS images are XOR-swizzled on the source side of the DMA so that ds_read_b128 is conflict-free:
//   WIDE: row i, 64 slots of 16 B: physical slot = logical ^ (i & 15)
//   TALL: row R, 8 slots:          physical slot = logical ^ ((R >> 1) & 7)
#pragma once
#include "gemm3.h"

#include <type_traits>
#include <utility>

typedef __attribute__((address_space(3))) float* lds_float_ptr;

template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<FIRST, LAST>(f); }

// ds_read_b128 at byte address addr + OFF, issued without the compiler's bookkeeping; the value
// may be used only after lds_wait_frag on it
template <int OFF>
__device__ __forceinline__ f32x4 lds_read128(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// waits until at most N LDS operations issued after the read of v are outstanding
template <int N>
__device__ __forceinline__ void lds_wait_frag(f32x4& v) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}

#define CH_TILE_FL 8192   // floats in one 32-KiB weight tile

struct ChainLane {
  int lane, wave, fi, h;
  unsigned a_wide[8];   // WIDE tile at LDS offset 0: fragment of group m at a_wide[m & 7] + 256 (m >> 3)
  unsigned a_tall[4];   // TALL tile at LDS offset 0: fragment of group (o, g) at a_tall[g] + 4096 o
  __device__ __forceinline__ void init(const float* smem, int tid) {
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    fi = lane & 31; h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(lds_float_ptr)smem;
    const int wx = (fi & 15) ^ h, tx = ((fi >> 1) & 7) ^ h;
#pragma unroll
    for (int k = 0; k < 8; ++k) a_wide[k] = lds0 + fi * 1024 + ((2 * k) ^ wx) * 16;
#pragma unroll
    for (int g = 0; g < 4; ++g) a_tall[g] = lds0 + fi * 128 + ((2 * g) ^ tx) * 16;
  }
  // This wave's piece n (0..7) of a WIDE tile: rows [row0, row0 + 32) (clamped below n_rows) x columns
  // [k0, k0 + 256) of W.  Piece = one 1-KiB row; lane l fills physical slot l.
  __device__ __forceinline__ void dma_wide(const float* W, int ldw, int row0, int n_rows, int k0, float* buf, int n) const {
    const int i = wave * 8 + n;
    int r = row0 + i;
    r = r < n_rows ? r : n_rows - 1;
    const int ls = lane ^ ((wave & 1) * 8) ^ n;
    lds_dma16(W + (size_t)r * ldw + k0 + ls * 4, buf + i * 256);
  }
  // This wave's piece n (0..7) of a TALL tile: rows [0, 256) x columns [k0, k0 + 32) of W.  Piece = eight
  // 128-B rows; lane l fills row l >> 3, physical slot l & 7.
  __device__ __forceinline__ void dma_tall(const float* W, int ldw, int k0, float* buf, int n) const {
    const int piece = wave * 8 + n;
    const int R = piece * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((R >> 1) & 7);
    lds_dma16(W + (size_t)R * ldw + k0 + ls * 4, buf + piece * 256);
  }
};

// N MFMA groups.  Group i: A fragment = rd(I) (an lds_read128 call), B values = getb(I) (f32x4),
// accumulator = getacc(I) (f32x16&).  The reads run three groups ahead of the MFMAs with counted waits:
// the compiler's wait-count pass treats an LDS-DMA in flight as a flat access and would drain the LDS
// queue before every use, which exposes a full LDS latency per group to the one wave on the SIMD.
// Nothing else may issue LDS or scalar-memory instructions inside (the counts assume it).
// hook(I) runs behind the MFMAs of group i: the place for vector-memory instructions (one or two per
// group).  A wave issues in order and is alone on its SIMD, so a global load, store or LDS-DMA issued
// between two steps holds up the matrix pipe for as long as the memory pipeline takes to accept it
// (100-200 cycles each when several go back to back); behind a group it has 256 cycles of MFMA shadow.
struct NoHook {
  template <class T> __device__ __forceinline__ void operator()(T) const {}
};
// SWAP exchanges the operand roles (A = the register values, B = the LDS fragment): the accumulator tile is
// then [32 rows x 32 weight rows], rows in the registers and the weight row (output feature) on the lane.
template <int N, bool SWAP = false, class Rd, class GetB, class GetAcc, class Hook = NoHook>
__device__ __forceinline__ void chain_groups(Rd&& rd, GetB&& getb, GetAcc&& getacc, Hook&& hook = NoHook{}) {
  f32x4 a[4];
  static_for<0, (N < 3 ? N : 3)>([&](auto J) { a[decltype(J)::value & 3] = rd(J); });
  static_for<0, N>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (i + 3 < N) a[(i + 3) & 3] = rd(std::integral_constant<int, i + 3>{});
    lds_wait_frag<(N - 1 - i < 3 ? N - 1 - i : 3)>(a[i & 3]);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 b = getb(I);
    f32x16& acc = getacc(I);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[e], a[i & 3][e], acc, 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3][e], b[e], acc, 0, 0, 0);
    hook(I);
    __builtin_amdgcn_sched_barrier(0);
  });
}
// 32 groups over a WIDE tile at LDS byte offset off: group m multiplies slab group m into ONE accumulator
template <class GetB>
__device__ __forceinline__ void chain_wide(const ChainLane& cl, unsigned off, f32x16& acc, GetB&& getb) {
  chain_groups<32>(
      [&](auto J) { constexpr int j = decltype(J)::value; return lds_read128<(j >> 3) * 256>(cl.a_wide[j & 7] + off); },
      getb, [&](auto) -> f32x16& { return acc; });
}
// 32 groups over a TALL tile at LDS byte offset off: the 32-feature slab u (an accumulator tile) into the
// eight accumulators out[0..7]
__device__ __forceinline__ void chain_tall(const ChainLane& cl, unsigned off, f32x16 (&out)[8], const f32x16& u) {
  chain_groups<32>(
      [&](auto J) { constexpr int j = decltype(J)::value; return lds_read128<(j >> 2) * 4096>(cl.a_tall[j & 3] + off); },
      [&](auto J) { constexpr int g = decltype(J)::value & 3; return f32x4{u[4 * g], u[4 * g + 1], u[4 * g + 2], u[4 * g + 3]}; },
      [&](auto J) -> f32x16& { return out[decltype(J)::value >> 2]; });
}

// SiLU of an accumulator tile, two elements per packed instruction where the ISA has one
__device__ __forceinline__ void silu_tile(f32x16& u) {
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const f32x2 x = {u[k], u[k + 1]};
    const f32x2 nx = x * -1.4426950408889634f;
    f32x2 d = {__builtin_amdgcn_exp2f(nx[0]), __builtin_amdgcn_exp2f(nx[1])};
    d = d + 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 y = x * r;
    u[k] = y[0]; u[k + 1] = y[1];
  }
}
