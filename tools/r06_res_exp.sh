#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0 FOS_RESIDENT_WAIT_S=3 FOSHIP_LIB=firstordersolvers.jl_amd/csrc/libfoship_res_stamps.so
for cfg in "0 1" "1 1" "0 0" "1 0"; do
  set -- $cfg
  echo "== FOS_RES_FLAGS=$1 FOS_RES_UNCACHED=$2"
  FOS_RES_FLAGS=$1 FOS_RES_UNCACHED=$2 timeout 120 python tools/res_stamps.py 64 2>&1 | grep "wg 0 comm\|wg G-1 comm" -A99 | grep "mean over\|finer" | sed -n '1,2p;5,6p'
done
