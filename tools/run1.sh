set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "host_pipeline or mixed or ragged or prepared or tiny or argument" 2>&1 | tail -15 || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/bench_pipe.json 2> gpurun_out/bench_pipe.err || { tail -20 gpurun_out/bench_pipe.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_pipe.json')); print(d['value']); [print(k, v) for k,v in d['extras'].items()]"
