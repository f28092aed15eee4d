mkdir -p gpurun_out/r05o
python -m pytest tests -m gpu -x -q -s > gpurun_out/r05o/tests.log 2>&1; echo "tests rc=$?"
grep -a "residual storage\|passed\|failed" gpurun_out/r05o/tests.log | tail -8
timeout -k 10 400 python tools/sampling_100k.py > gpurun_out/r05o/sampling_100k.md 2> gpurun_out/r05o/sampling.err; echo "sampling rc=$?"; cat gpurun_out/r05o/sampling_100k.md
timeout -k 10 300 python bench.py > gpurun_out/r05o/bench.json 2> gpurun_out/r05o/bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r05o/bench.json
