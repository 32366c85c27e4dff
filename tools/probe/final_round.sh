# the round's closing measurements in ONE call: the whole GPU suite, the kernel trace + stats of a bench run, vector instructions per
# wave, the driver's command and the default:  TAG=r04e bash tools/probe/final_round.sh
tag=${TAG:-r04e}
out=gpurun_out/r4/prof_$tag; mkdir -p $out/summary
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -2 $out/gpu_tests.log
OUT=$out TAG=$tag bash tools/probe/trace_only.sh &&
bash tools/probe/valu_pass.sh $out/valu > $out/summary/${tag}_valu_per_wave.txt 2>&1 &&
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver.log 2> $out/driver.err &&
grep -h "^{" $out/driver.log | tail -1 > $out/summary/${tag}_bench_line_driver_command.json &&
timeout -k 10 500 python bench.py > $out/default.log 2> $out/default.err &&
grep -h "^{" $out/default.log | tail -1 > $out/summary/${tag}_bench_line.json
ls $out/summary
