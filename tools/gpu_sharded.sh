#!/bin/bash
# two ranks sharing the one GPU (gloo rendezvous): sharded registration with the in-library exchange vs the host-driven collective
OUT=gpurun_out/$1; mkdir -p $OUT
export BENCH_DIST_BACKEND=gloo
for ex in library collective; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 5 --mode sharded --exchange $ex --no-cpu-baseline --loop-steps 0 > $OUT/sharded_$ex.json 2> $OUT/sharded_$ex.err
  tail -1 $OUT/sharded_$ex.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ex', round(d['value'],1), 'scans/s', round(d['ms_per_step'],3), 'ms', d['max_pose_error_vs_ground_truth_m'])" || tail -5 $OUT/sharded_$ex.err
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline --loop-steps 0 > $OUT/replicas2.json 2> $OUT/replicas2.err
tail -1 $OUT/replicas2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('replicas x2', round(d['value'],1), 'scans/s', round(d['ms_per_step'],3), 'ms')"
