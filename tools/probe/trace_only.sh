cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4/profb2; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe --no-end-to-end --no-placement > $out/kt.log 2>&1 || echo "trace failed"
python3 tools/summarize_profile.py r04b --out $out/summary --stats $out/kt
grep -h "^{" $out/kt.log | tail -1 > $out/summary/r04b_bench_line_under_rocprofv3.json
rm -rf $out/kt
