# two ranks on the one GPU of this box, gathering over gloo: exercises bench.py's N > 1 path end to end
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 3 --warmup 1 --frames 8 --backend gloo --no-cpu > gpurun_out/bench_n2_gloo.json 2> gpurun_out/bench_n2_gloo.err || { tail -20 gpurun_out/bench_n2_gloo.err; exit 1; }
tail -1 gpurun_out/bench_n2_gloo.json
