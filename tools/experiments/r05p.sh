mkdir -p gpurun_out/r05p
python -m pytest tests -m gpu -x -q > gpurun_out/r05p/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r05p/tests.log
timeout -k 10 400 python tools/sampling_100k.py > gpurun_out/r05p/sampling_100k.md 2> gpurun_out/r05p/sampling.err; echo "sampling rc=$?"; cat gpurun_out/r05p/sampling_100k.md
timeout -k 10 500 python tools/ddp_route_bench.py > gpurun_out/r05p/ddp_route.txt 2> gpurun_out/r05p/ddp_route.err; echo "ddp rc=$?"; cat gpurun_out/r05p/ddp_route.txt
V4H_STAGED_LATE=0 timeout -k 10 500 python tools/ddp_route_bench.py > gpurun_out/r05p/ddp_route_late0.txt 2> gpurun_out/r05p/ddp_route_late0.err; echo "ddp rc=$?"; cat gpurun_out/r05p/ddp_route_late0.txt
