// Fused edge path and node update of the E(3) layers (inference path): kernel instantiations and launchers.
#include "edge_chain.h"
#include "node_chain.h"

static inline int ceil_div_i(long a, long b) { return (int)((a + b - 1) / b); }

void launch_edge_chain(const EdgeChainArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const bool geom = a.pos != nullptr;   // edge geometry and radial MLP inside the kernel
  if (a.part != nullptr) {
    if (geom) hipLaunchKernelGGL((edge_chain_kernel<false, 0, true, true>), dim3(a.n_slabs / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((edge_chain_kernel<false, 0, true>), dim3(a.n_slabs / 4), dim3(256), 0, s, a);
  } else {
    if (geom) hipLaunchKernelGGL((edge_chain_kernel<false, 0, false, true>), dim3(ceil_div_i(a.E, 128)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((edge_chain_kernel<false, 0>), dim3(ceil_div_i(a.E, 128)), dim3(256), 0, s, a);
  }
}

void launch_node_chain(const NodeChainArgs& a, int tail, hipStream_t s) {
  if (a.M <= 0) return;
  const dim3 grid(ceil_div_i(a.M, 128));
  if (tail == NODE_TAIL_PROJ) hipLaunchKernelGGL((node_chain_kernel<NODE_TAIL_PROJ>), grid, dim3(256), 0, s, a);
  else if (tail == NODE_TAIL_QKV) hipLaunchKernelGGL((node_chain_kernel<NODE_TAIL_QKV>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((node_chain_kernel<NODE_TAIL_NONE>), grid, dim3(256), 0, s, a);
}
