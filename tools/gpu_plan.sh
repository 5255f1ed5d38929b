#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout 900 python tools/probe_plan.py ${2:-16384} ${3:-ba100k} > $O/probe_plan.log 2>&1; tail -5 $O/probe_plan.log
