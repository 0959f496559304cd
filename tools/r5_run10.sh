#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/step_layer_table.py 10 > gpurun_out/steptab_new.md 2>gpurun_out/steptab_new.err
ONET_FLAGS="CONVT_SLOTS=0,CONVT_BWD_SLOTS=0" timeout -k 10 300 python tools/step_layer_table.py 10 > gpurun_out/steptab_old.md 2>gpurun_out/steptab_old.err
grep -c . gpurun_out/steptab_new.md gpurun_out/steptab_old.md
