#!/bin/bash
# rocprofv3 kernel-trace and HBM-counter passes at the headline shape (1024 x 29-atom template ligands).
# Usage (GPU box, repo root): bash tools/profile_headline.sh TAG   -> gpurun_out/TAG_headline/
set -e
tag=$1
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out/${tag}_headline
rm -rf $out; mkdir -p $out
# every batch as ONE library call: a kernel's average is the duration of a whole-batch launch
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --one-call > $out/kt.log 2>&1
cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
rm -rf $out/kt
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 tools/pmc_forward.py > $out/fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o pmc -- python3 tools/pmc_forward.py > $out/write.log 2>&1
python3 tools/pmc_summary.py $out/pmc_summary.csv $out/fetch $out/write
rm -rf $out/fetch $out/write
echo "done headline"
