#!/bin/bash
# Development aid: cycle stamps of lw_gas_kernel's phases (build with -DLW_EXP_CLOCK on the GPU box, one bench step).
cd wrfchem-arc-interactions_amd/csrc; touch lw_kernel.hip; make -s EXTRA="-DLW_EXP_CLOCK $1" > /dev/null 2>&1; cd ../..
python3 bench.py --steps 1 --warmup 0 --no-cpu --host-path-rows 0 --cloudy-pass 0 2>/dev/null | grep "^blk" | sort | head -12
cd wrfchem-arc-interactions_amd/csrc; touch lw_kernel.hip; make -s > /dev/null 2>&1
