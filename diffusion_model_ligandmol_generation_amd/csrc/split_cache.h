// Host-side record of the weight blocks already converted to the split-bf16 layout of gemm4.h
// during one library call (weights cannot change while a call runs, so entries stay valid for its
// duration; nothing survives the call).  Storage is a bump-allocated region of the forward workspace.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct SplitCache {
  struct Entry { const float* W; int ldw, K, n_valid, ncols; const uint16_t* blk; };
  uint16_t* base = nullptr;   // device region (bf16 elements)
  size_t cap = 0, used = 0;   // in bf16 elements
  int n = 0;
  Entry ent[192];
};
