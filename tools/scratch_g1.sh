set -x
mkdir -p gpurun_out/s6
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s6/dp_test.log
timeout 300 python tools/dp_graph_micro.py > gpurun_out/s6/dp_graph_micro.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s6/bench_20_5.log 2>&1
timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/s6/bench_400.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s6/bench_20_5b.log 2>&1
tail -3 gpurun_out/s6/dp_test.log; cat gpurun_out/s6/dp_graph_micro.log | tail -8
for f in gpurun_out/s6/bench_*.log; do python - "$f" <<'P'
import sys,json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['ms_per_step'], d['value'])
P
done
