#!/bin/bash
# Builds libligdiff.so for gfx950 next to the sources (in-tree, travels to the GPU box).
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in chain_dit_mol_heads chain_dit_mol chain_dit_row_heads chain_dit_row forward backward chain_e3 sampler api pack loss consistency optim; do
  if [ -f $f.hip ]; then
    hipcc $FLAGS -c $f.hip -o $f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libligdiff.so *.o
echo built ../libligdiff.so
