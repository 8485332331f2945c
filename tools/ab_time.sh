#!/bin/bash
# same-box A/B of two builds of the library on the c3 recurrent sweep: mvsnet_amd/variants/lib_base.so against the working tree's
for rep in 1 2 3; do
  for L in base new; do
    if [ $L = base ]; then export MVS_LIB_PATH=mvsnet_amd/variants/lib_base.so; else unset MVS_LIB_PATH; fi
    python tools/gru_time.py --views ${AB_VIEWS:-1 4} --iters 6 2>&1 | grep "c3 sweep" | sed "s/^/$L: /" | cut -c1-105
  done
done
