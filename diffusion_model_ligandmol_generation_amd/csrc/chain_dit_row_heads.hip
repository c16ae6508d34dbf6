// Fused attention-layer tail, row form, last layer (with the atom / coordinate heads): kernel instantiation.
#include "dit_chain.h"

void launch_dit_chain_row_heads(const DitChainArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((dit_chain_kernel<DIT_TAIL_HEADS>), grid, dim3(256), 0, s, a);
}
