// Fused attention layer, molecule-per-wave form (attention inside the kernel): the instantiation without a tail
// and the launcher of both forms.  The instantiations with the output heads sit in translation units of their own
// (chain_dit_mol_heads.hip, chain_dit_row_heads.hip): each takes minutes to compile.
#include "dit_chain.h"

void launch_dit_chain_row(const DitChainArgs& a, int tail, hipStream_t s);
void launch_dit_chain_mol_heads(const DitChainArgs& a, dim3 grid, hipStream_t s);

void launch_dit_chain(const DitChainArgs& a, int tail, bool mol_wave, hipStream_t s) {
  if (a.M <= 0) return;
  if (!mol_wave) {
    launch_dit_chain_row(a, tail, s);
    return;
  }
  const dim3 grid((a.B + 3) / 4);
  if (tail == DIT_TAIL_HEADS) launch_dit_chain_mol_heads(a, grid, s);
  else hipLaunchKernelGGL((dit_chain_kernel<DIT_TAIL_NONE, true>), grid, dim3(256), 0, s, a);
}
