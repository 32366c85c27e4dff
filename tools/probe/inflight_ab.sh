# windows_in_flight 3 against 4 (and 2), alternating runs in one call: bash tools/probe/inflight_ab.sh
for r in 1 2 3 4; do
for n in ${INFLIGHT:-3 4}; do
echo "inflight $n"; python tools/ab_pipe.py tree -- --varied --loop 1920 --window 128 --inflight $n --gpu-parser --threads 16
done; done
