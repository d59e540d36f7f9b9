#!/bin/bash
# round 6, call 13: ac_bert_encode_cls_unpad (one call, no stream synchronisation): its equivalence test, the encoder test file, the step's
# host phases and launch sequence with it
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "unpad_one_call" 2>&1 | tail -25 | tee $O/pytest_unpad.txt
python tools/r06_step_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/step_host_phases_3.txt
AC_BERT_UNPAD_ONE_CALL=0 python tools/r06_step_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/step_host_phases_3.txt
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_step6; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $T -o t -- python $REPO/tools/r06_step_probe.py > /dev/null 2>&1
python $REPO/tools/r06_step_seq.py $(find $T -name "*kernel_trace.csv" | head -1) pack_prologue_kernel > $O/step_launch_sequence_3.txt; head -8 $O/step_launch_sequence_3.txt; tail -32 $O/step_launch_sequence_3.txt
cd $REPO
timeout 2400 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_classifier_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_encoder_files_3.txt
