mkdir -p gpurun_out/r04am
ORBX_DIST_REHEARSE=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 > gpurun_out/r04am/rehearse2.json 2> gpurun_out/r04am/rehearse2.err || { tail -20 gpurun_out/r04am/rehearse2.err; exit 1; }
tail -1 gpurun_out/r04am/rehearse2.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'n_gpus', 'scaling')}, d['config'].get('parallelism'))
lb = d.get('local_ba', {})
print({k: (v if not isinstance(v, dict) else '...') for k, v in lb.items()})
"
