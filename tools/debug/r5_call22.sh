#!/bin/bash
O=gpurun_out/r5_22
mkdir -p $O
timeout 600 python tools/insitu_sweep.py --steps 6 --settings "default:;i3off:igemm3=0;i3all:igemm3=2;t128:igemm_tile=1;t256:igemm_tile=3;wsmall:wgrad_big=0;wbig:wgrad_big=1;nohalo:conv_halo=0;whalo0:wgrad_halo=0" --out $O/insitu3.json > $O/insitu3.log 2>&1; echo "rc=$?" >> $O/summary.txt
tail -10 $O/insitu3.log | cut -c1-160
