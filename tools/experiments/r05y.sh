mkdir -p gpurun_out/r05y
timeout -k 10 400 python tools/comm_interference.py --steps 30 --rounds 2 --only "identity;nwg=16,reserve=0,gbps=300;nwg=16,reserve=0,gbps=150;nwg=8,reserve=0,gbps=300" > gpurun_out/r05y/f32.txt 2> gpurun_out/r05y/f32.err; tail -7 gpurun_out/r05y/f32.txt
timeout -k 10 400 python tools/comm_interference.py --steps 30 --rounds 2 --grad-allreduce bf16 --only "identity;nwg=16,reserve=0,gbps=300;nwg=16,reserve=0,gbps=150;nwg=8,reserve=0,gbps=300" > gpurun_out/r05y/bf16.txt 2> gpurun_out/r05y/bf16.err; tail -7 gpurun_out/r05y/bf16.txt
python -m pytest tests/test_hip_round4.py -q -k "staged_autograd" 2>&1 | tail -2
