#!/bin/bash
O=gpurun_out/r5_9
mkdir -p $O
timeout 900 python tools/insitu_sweep.py --out $O/insitu.json > $O/insitu.log 2>&1; echo "insitu rc=$?" >> $O/summary.txt
tail -16 $O/insitu.log | cut -c1-160
cat $O/summary.txt
