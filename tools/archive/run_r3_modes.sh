#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3modes; mkdir -p $O; P=tools/plan_mode_probe.py
{ timeout -k 10 100 python3 $P; timeout -k 10 100 python3 $P; GPU_MAX_HW_QUEUES=8 timeout -k 10 100 python3 $P; GPU_MAX_HW_QUEUES=8 timeout -k 10 100 python3 $P; GPU_MAX_HW_QUEUES=2 timeout -k 10 100 python3 $P; timeout -k 10 100 python3 $P --lg 20 --batch 4096; timeout -k 10 100 python3 $P --lg 17 --batch 32768; } > $O/plan_instances_after.txt 2>&1
cat $O/plan_instances_after.txt
