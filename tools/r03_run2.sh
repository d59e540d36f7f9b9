#!/bin/bash
# round 3, GPU call 2: wave-specialised GEMM variants + per-workgroup stamps; per-shape tables inside the encoder; kernel trace
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
CFGS=1,2232,2262,1431,1432,1461,1462,2442,42261,42262,22262,22232,42441,42442,41462,41432
{
echo "### ring / wave-specialised planes GEMM sweep with per-workgroup stamps; cfg = nlw*10000 + tm*1000 + wmw*100 + ns*10 + pipe"
GEMM_BENCH_STAMPS=1 timeout 300 tools/ab/gemm_bench $CFGS 20 3 5141,2304,768,0,0,0 5141,768,768,0,1,0 5141,3072,768,2,0,1 5141,768,3072,0,1,0
timeout 300 tools/ab/gemm_bench $CFGS 10 3 20564,3072,1024,0,0,0 20564,1024,4096,0,1,0 8192,8192,8192,0,0,0
} > gpurun_out/r03/gemm_sweep2.txt 2>&1
{
echo "### per-shape tables inside the bert-base encoder (ragged 256 x 32 batch, 5141 token rows)"
timeout 600 python tools/encode_ab.py "base=" "n768_1461=768x768=1461;768x3072=1461" "n768_1461+qkv2232=768x768=1461;768x3072=1461;2304x768=2232" \
   "n768_1462=768x768=1462;768x3072=1462" "all_ws42262=768x768=42262;768x3072=42262;2304x768=42262;3072x768=42262" \
   "n768_ws41462=768x768=41462;768x3072=41462" "n768_ws42262=768x768=42262;768x3072=42262" "big_ws42442=2304x768=42442;3072x768=42442;768x768=1461;768x3072=1461"
} > gpurun_out/r03/encode_ab2.txt 2>&1
cd /tmp && ROUNDS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_enc -- python $GRAFT_REPO_ROOT/tools/encode_ab.py "base=" > /tmp/prof_enc.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_enc -name "*kernel_trace.csv" | head -1)
python - "$f" > gpurun_out/r03/encode_trace_base.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    key = (name[:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""), r.get("LDS_Block_Size", ""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/n/1e3:9.1f} us x {n:5d} = {t/1e6:8.3f} ms ({100*t/tot:5.1f}%)  grid {k[1]:>8} wg {k[2]:>4} lds {k[3]:>6}  {k[0]}")
PY
tail -12 gpurun_out/r03/encode_ab2.txt
