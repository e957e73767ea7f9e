# 1-GPU rehearsal of `bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per
# rank), with gloo instead of RCCL and both ranks on cuda:0: checks the multi-rank control flow (view sharding,
# gradient all-reduce, barriers, max-over-ranks timing, rank-0-only codec / CPU legs, ONE stdout line).  The numbers
# it prints are meaningless (two ranks share one GPU, the all-reduce goes through the host).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export CGS_BENCH_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --anchors ${1:-200000} > gpurun_out/bench_rehearsal.json 2> gpurun_out/bench_rehearsal.err
echo "rc=$? stdout_lines=$(wc -l < gpurun_out/bench_rehearsal.json)"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_rehearsal.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "n_gpus", "steps", "ms_per_step", "scaling")}, d["config"]["parallelism"], d["config"]["views_per_step"], bool(d.get("codec")), bool(d.get("cpu_baseline")))
PY
tail -5 gpurun_out/bench_rehearsal.err
