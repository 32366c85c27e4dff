for r in 1 2 3; do
for n in 2 3; do
echo "inflight $n"; python tools/ab_pipe.py tree -- --varied --loop 1920 --window 128 --inflight $n --gpu-parser --threads 16
done; done
