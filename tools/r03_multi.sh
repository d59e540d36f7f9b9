#!/bin/bash
# N > 1 code path of bench.py on ONE GPU (two ranks over gloo: RCCL refuses duplicate devices) + smoke()
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
AC_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --sweep-rows 2000000 > $O/bench_line_2proc_gloo_one_gpu.json 2> $O/bench_2proc_err.log; echo "rc=$?"
tail -3 $O/bench_2proc_err.log | cut -c1-300
python -c "
import json;d=json.load(open('$O/bench_line_2proc_gloo_one_gpu.json'))
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}, d.get('cfg2_sharded'))"
