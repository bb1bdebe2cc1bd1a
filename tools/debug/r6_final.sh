# end-of-round verification: GPU test tier, smoke(), the bench line, the round's profiles.  usage: r6_final.sh TAG
TAG=${1:-round6_c}
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench_line.json
bash tools/profile_round.sh $TAG stats_default stats sq traffic dominant > gpurun_out/${TAG}_profile.log 2>&1
timeout 600 python tools/low_occupancy.py $(find gpurun_out/profd_$TAG -name '*results.db' | head -1) > gpurun_out/${TAG}_low_occupancy.md 2>&1
ls -la gpurun_out/${TAG}_*
