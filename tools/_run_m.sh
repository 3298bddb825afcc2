mkdir -p gpurun_out/r02_o
python -m pytest tests/test_gpu_streaming.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02_o/pytest.log
python bench.py --mode streaming --graph --no-cpu-baseline > gpurun_out/r02_o/bench_streaming_graph.json 2>> gpurun_out/r02_o/err.txt
python bench.py --mode streaming --graph --pipeline --no-cpu-baseline > gpurun_out/r02_o/bench_streaming_pipe.json 2>> gpurun_out/r02_o/err.txt
python bench.py --no-cpu-baseline --no-launch-timing > gpurun_out/r02_o/bench.json 2>> gpurun_out/r02_o/err.txt
python bench.py --no-cpu-baseline --no-launch-timing --overlap 2 > gpurun_out/r02_o/bench_ov2.json 2>> gpurun_out/r02_o/err.txt
python bench.py --no-cpu-baseline --no-launch-timing --overlap 4 > gpurun_out/r02_o/bench_ov4.json 2>> gpurun_out/r02_o/err.txt
cat gpurun_out/r02_o/pytest.log; tail -5 gpurun_out/r02_o/err.txt
python - <<'PY'
import json
for f in ['bench_streaming_graph','bench_streaming_pipe','bench','bench_ov2','bench_ov4']:
    d=json.load(open(f'gpurun_out/r02_o/{f}.json'))
    print(f, round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['whole_path_frac'],4))
PY
