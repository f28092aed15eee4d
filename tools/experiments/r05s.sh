mkdir -p gpurun_out/r05s
python -m pytest tests -m gpu -x -q > gpurun_out/r05s/tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r05s/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-op-rates --no-sampling --no-box 2> gpurun_out/r05s/bench.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['other'])"
