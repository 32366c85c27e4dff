# the kernel trace of the driver's command WITH the start-up placement step (trace_only.sh runs without it, so that rocprofv3's own
# stats table holds the steps' launches only): the stats of the timed region are cut from the trace (the last steps x 8
# reconstruction dispatches, tools/summarize_profile.py --timed-launches):  TAG=r04i bash tools/probe/trace_with_placement.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${TAG:-r04i}; out=${OUT:-gpurun_out/r4/prof_$tag}; mkdir -p $out/summary
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-second-recipe --no-end-to-end --steps 20 --warmup 5 > $out/kt.log 2>&1 || echo "trace failed"
python3 tools/summarize_profile.py $tag --out $out/summary --stats $out/kt --timed-launches 160
grep -h "^{" $out/kt.log | tail -n 1 > $out/summary/${tag}_bench_line_under_rocprofv3.json
rm -rf $out/kt
cat $out/summary/${tag}_kernel_stats_timed_region.csv
