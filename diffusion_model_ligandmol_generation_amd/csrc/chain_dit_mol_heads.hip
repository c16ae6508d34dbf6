// Fused attention layer, molecule-per-wave form, last layer (with the atom / coordinate heads): kernel instantiation.
#include "dit_chain.h"

void launch_dit_chain_mol_heads(const DitChainArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((dit_chain_kernel<DIT_TAIL_HEADS, true>), grid, dim3(256), 0, s, a);
}
