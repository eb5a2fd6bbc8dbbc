#!/bin/bash
# Exercises bench.py's multi-rank path on a ONE-GPU box: two ranks share cuda:0 and talk over gloo (DC_BENCH_ONE_DEVICE=1).
# Not a performance number (the ranks time-share the GPU); it checks that the N > 1 flow runs and prints one JSON line.
mkdir -p gpurun_out/two
export DC_BENCH_ONE_DEVICE=1
for ov in 0 1; do
export DC_DP_OVERLAP=$ov
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$ov \
    bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/two/bench_$ov.json 2> gpurun_out/two/bench_$ov.err
echo "DC_DP_OVERLAP=$ov exit $?"
python - <<PY
import json
lines = [l for l in open('gpurun_out/two/bench_$ov.json') if l.startswith('{')]
print('json lines:', len(lines))
j = json.loads(lines[0])
print({k: j[k] for k in ('metric', 'value', 'n_gpus', 'steps', 'ms_per_step', 'scaling', 'nan_status', 'final_loss')})
PY
grep -v "hostname of the client socket" gpurun_out/two/bench_$ov.err | tail -3
done
