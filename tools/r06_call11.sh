#!/bin/bash
# round 6, call 11: host phases of the timed step + its launch sequence (kernel trace of the same probe)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
python tools/r06_step_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/step_host_phases.txt
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_step6; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $T -o t -- python $REPO/tools/r06_step_probe.py > /dev/null 2>&1
python $REPO/tools/r06_step_seq.py $(find $T -name "*kernel_trace.csv" | head -1) | tee $O/step_launch_sequence.txt | tail -50
