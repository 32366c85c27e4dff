cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${OUT:-gpurun_out/r4/profd}; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe --no-end-to-end --no-placement > $out/kt.log 2>&1 || echo "trace failed"
python3 tools/summarize_profile.py ${TAG:-r04d} --out $out/summary --stats $out/kt
grep -h "^{" $out/kt.log | tail -1 > $out/summary/${TAG:-r04d}_bench_line_under_rocprofv3.json
rm -rf $out/kt
