#!/bin/bash
set -x
mkdir -p gpurun_out/r02
(time timeout 1500 python -m pytest tests -x -q -m gpu --durations=15) > gpurun_out/r02/tests_all.log 2>&1
tail -40 gpurun_out/r02/tests_all.log
