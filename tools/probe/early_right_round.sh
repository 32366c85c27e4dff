# LEON_EARLY_RIGHT (the luma parts' reference rows requested earlier, csrc/leon_kernels.h) measured against the tree's build on ONE box,
# two interleaved passes; the faster build (if it wins both passes by > 0.7 %) takes the tree's place ON THE BOX for the rest of the call:
# the whole GPU suite, the driver's command, the kernel trace, the default command.  Builds: tools/ab_build.sh erN -DLEON_EARLY_RIGHT=N
#   TAG=r04h bash tools/probe/early_right_round.sh
tag=${TAG:-r04h}
out=gpurun_out/r4/prof_$tag; mkdir -p $out/summary
variants=${VARIANTS:-"tree er1 er2 er3 er4"}     # VARIANTS="tree ilp memcl relax": builds with other scheduling strategies of the compiler
timeout -k 10 600 python tools/ab_run.py $variants $variants -- --no-end-to-end --warmup 5 > $out/ab.log 2>&1 || { tail -5 $out/ab.log; exit 1; }
cat $out/ab.log
python - $out/ab.log > $out/choice.txt <<'EOF'
import sys, collections
ms = collections.defaultdict(list)
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) > 2 and f[2].startswith("ms/step"):
        ms[f[0]].append(float(f[1]))
tree = ms.get("tree", [])
best = "tree"
for name, v in ms.items():
    if name == "tree" or len(v) != len(tree) or len(v) < 2:
        continue
    if all(a < b for a, b in zip(v, tree)) and sum(v) < 0.993 * sum(tree) and sum(v) < sum(ms[best]):
        best = name
print(best)
EOF
choice=$(cat $out/choice.txt); echo "choice: $choice"
if [ "$choice" = "tree" ] && [ -n "$STOP_IF_TREE" ]; then exit 0; fi
if [ "$choice" != "tree" ]; then cp build/ab/$choice/libleon_hip.so mpeg1video-decoder-webgl_amd/lib/libleon_hip.so; fi
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -n 2 $out/gpu_tests.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver.log 2> $out/driver.err &&
grep -h "^{" $out/driver.log | tail -n 1 > $out/summary/${tag}_bench_line_driver_command.json &&
OUT=$out TAG=$tag bash tools/probe/trace_only.sh &&
timeout -k 10 400 python bench.py > $out/default.log 2> $out/default.err &&
grep -h "^{" $out/default.log | tail -n 1 > $out/summary/${tag}_bench_line.json
ls $out/summary
