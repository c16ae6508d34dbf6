// Fused attention-layer tail, row form (32 consecutive atoms per wave): kernel instantiations.
#include "dit_chain.h"

void launch_dit_chain_row_heads(const DitChainArgs& a, dim3 grid, hipStream_t s);

void launch_dit_chain_row(const DitChainArgs& a, int tail, hipStream_t s) {
  const dim3 grid((a.M + 127) / 128);
  if (tail == DIT_TAIL_QKV) hipLaunchKernelGGL((dit_chain_kernel<DIT_TAIL_QKV>), grid, dim3(256), 0, s, a);
  else if (tail == DIT_TAIL_HEADS) launch_dit_chain_row_heads(a, grid, s);
  else hipLaunchKernelGGL((dit_chain_kernel<DIT_TAIL_NONE>), grid, dim3(256), 0, s, a);
}
