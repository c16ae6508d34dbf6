# headline-only bench several times in one box session (run-to-run spread, cold-start effects); arguments of each run in RUNS, ';'-separated
IFS=';' read -ra runs <<< "${RUNS:-;;;}"
i=0
for a in "${runs[@]}" ""; do
i=$((i+1))
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i [$a]', round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['avg_launch_ms'],4))" || exit 1
done
