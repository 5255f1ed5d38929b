#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python tools/probe_classes.py ${2:-16384} > $O/probe_classes.log 2>&1; tail -16 $O/probe_classes.log
