#!/bin/bash
# rocprofv3 kernel-trace and HBM-counter passes at the shapes of the benchmark's extra legs.
# Usage (GPU box, repo root): bash tools/profile_shapes.sh TAG SHAPE [SHAPE ...]   -> gpurun_out/TAG_SHAPE_*
set -e
tag=$1; shift
root=$(pwd)
export TMPDIR=/tmp
for shape in "$@"; do
  out=$root/gpurun_out/${tag}_${shape}
  rm -rf $out; mkdir -p $out
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 tools/pmc_shape.py $shape > $out/kt.log 2>&1
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 tools/pmc_shape.py $shape > $out/fetch.log 2>&1
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o pmc -- python3 tools/pmc_shape.py $shape > $out/write.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_summary.csv $out/fetch $out/write
  cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
  # keep only the summaries (the raw traces are large)
  rm -rf $out/kt $out/fetch $out/write
  echo "done $shape"
done
