// Argument blocks and host-side launchers of the fused chain kernels (edge_chain.h, node_chain.h, dit_chain.h).
// The kernels are compiled in translation units of their own (chain_e3.hip, chain_dit_row*.hip,
// chain_dit_mol*.hip): each instantiation is ten thousand instructions of straight-line code and takes minutes
// to compile, so they build in parallel with the rest of the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct EdgeChainArgs {
  int E;
  const float* rbf;          // [E,32]
  const float* rel;          // [E,4]
  const int32_t* src; const int32_t* dst;
  const float* P;            // [N,512]
  const float* Wc; int ldwc; // rbf columns of the first edge linear: [256 x 32], leading dimension ldwc
  const float* W2; const float* b2; const float* gamma; const float* beta;
  const float* Wc1; const float* bc1; const float* wc2; const float* bc2;
  float* msg;                // [E,256]
  float* wrel;               // [E,4] (components 0..2 written)
  unsigned long long* stamps;   // diagnostic builds (DBG): [grid][8] shader-clock stamps of wave 0
  float* part;               // REDUCE form: [N + slabs, 256] per-slab partial sums instead of msg (see edge_chain.h)
  const int32_t* slab_e0;    // REDUCE form: first edge / edge count (<= 32) of every slab; n_slabs entries, the ones
  const int32_t* slab_cnt;   //   past the last molecule's slabs hold (0, 0)
  int n_slabs;               // REDUCE form: table length = grid size * 4
  // GEOM form: the edge geometry and the radial MLP (model1.py:88-93) run inside the kernel; rbf / rel are not read
  const float* pos;          // [N,3] coordinates this layer sees
  const float* rw0; const float* rb0;   // radial_basis.0: weight [32] (Linear(1, 32)), bias [32]
  const float* rW2; const float* rb2;   // radial_basis.2: weight [32,32], bias [32]
};

enum { NODE_TAIL_NONE = 0, NODE_TAIL_PROJ = 1, NODE_TAIL_QKV = 2 };
struct NodeChainArgs {
  int M;
  const float* h;       // [M,256]
  const float* agg;     // [M,256]
  float* hout;          // [M,256] (may be h)
  const float* Wn1; const float* b1; const float* Wn2; const float* b2; const float* gamma; const float* beta;
  // tail
  const float* Wt; const float* bt;   // TAIL_PROJ: next layer's edge_network.0 weight [256,544] / bias; TAIL_QKV: in_proj [768,256] / bias
  float* tout;                        // TAIL_PROJ: P [M,512]; TAIL_QKV: qkv [M,768]
  const float* mod;   // TAIL_QKV: first attention layer's adaLN row, shared by all molecules (shift_msa | scale_msa | ...)
  unsigned long long* stamps;   // diagnostic builds (DBG): [grid][8] shader-clock stamps of wave 0
};

enum { DIT_TAIL_NONE = 0, DIT_TAIL_QKV = 1, DIT_TAIL_HEADS = 2 };
struct DitChainArgs {
  int M;
  const float* att;     // [M,256] attention output (heads concatenated)
  const float* x;       // [M,256] residual stream in
  float* xout;          // [M,256] residual stream out (may be x)
  const float* Wo; const float* bo;
  const float* W1; const float* b1; const float* W2; const float* b2;
  const float* mod;   // this layer's adaLN row, shared by all molecules (one timestep per call): shift_msa | scale_msa | gate_msa | shift_mlp | scale_mlp | gate_mlp
  const float* mod_next;              // the next layer's rows (TAIL_QKV): shift_msa | scale_msa | ...
  const float* Wqkv; const float* bqkv; float* qkv;   // DIT_TAIL_QKV: next layer's in_proj, output [M,768]
  // DIT_TAIL_HEADS: hidden layers (Linear 256-256, GELU, LayerNorm) and output layers of the atom / coordinate heads
  const float* W0a; const float* b0a; const float* ga; const float* bea; const float* W3a; const float* b3a;
  const float* W0p; const float* b0p; const float* gp; const float* bep; const float* W3p; const float* b3p;
  const float* pos_upd;   // [M,3] accumulated coordinate updates of the E(3) layers
  float* atom_pred;       // [M,119]
  float* pos_pred;        // [M,3]
  // MOLWAVE: this layer's in-projection and the molecule boundaries
  const float* Wqkv_self; const float* bqkv_self; const int32_t* mol_ptr; int B;
  unsigned long long* stamps;   // diagnostic builds (DBG): [grid][8] shader-clock stamps of wave 0
};

// One launch each on stream s; grids follow from the argument blocks.
void launch_edge_chain(const EdgeChainArgs& a, hipStream_t s);                              // ceil(E / 128) workgroups; a.part != NULL: REDUCE form, n_slabs / 4
void launch_node_chain(const NodeChainArgs& a, int tail, hipStream_t s);                    // ceil(M / 128) workgroups
void launch_dit_chain(const DitChainArgs& a, int tail, bool mol_wave, hipStream_t s);       // ceil(M / 128), or ceil(B / 4) in the molecule-per-wave form
