cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
BABYSGL_TP_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > gpurun_out/r3h/b2.json 2> gpurun_out/r3h/b2.err; echo "rc=$?"; cut -c1-900 gpurun_out/r3h/b2.json; tail -3 gpurun_out/r3h/b2.err
timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 > gpurun_out/r3h/b2n.json 2> gpurun_out/r3h/b2n.err; echo "no-gloo rc=$?"; tail -2 gpurun_out/r3h/b2n.err
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > gpurun_out/r3h/b1.json 2> gpurun_out/r3h/b1.err; echo "rc=$?"; cut -c1-1500 gpurun_out/r3h/b1.json
