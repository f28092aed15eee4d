mkdir -p gpurun_out/r05t
python -m pytest tests -m gpu -q > gpurun_out/r05t/tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r05t/tests.log
