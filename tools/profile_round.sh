#!/bin/bash
# Profiles of the bench command for one round: rocprofv3 kernel stats + PMC traffic (separate passes, MI355X_MICROARCH.md's recipe) per
# configuration -> gpurun_out/$R/, one JSON per configuration that bench.py reads its `traffic` / `decode.kernels_profiled` figures from
# (copy the results into profiles/ and commit them).   usage: R=r05 CFGS="c1 c2 c4" bash tools/profile_round.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${R:-r05}
mkdir -p gpurun_out/$R
for c in ${CFGS:-c1 c2 c4}; do
  case $c in
    c1) A=""; P="" ;;
    c2) A="--config fpn --batch 64"; P="--config fpn --batch 64" ;;
    c4) A="--config tracking --batch 32 --height 608 --width 1088"; P="--config tracking --batch 32 --size 608 1088" ;;
  esac
  CMD="python bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-also --no-accuracy"
  O=gpurun_out/$R/$c
  rm -rf $O; mkdir -p $O
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- $CMD > $O/bench_stats.json 2> $O/stats.err
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- $CMD > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- $CMD > /dev/null 2>&1
  python tools/rocpd_summary.py stats $O/stats > gpurun_out/$R/${R}_kernel_stats_$c.csv
  python tools/rocpd_summary.py pmc $O/fetch $O/write > gpurun_out/$R/${R}_pmc_traffic_$c.txt
  SHA=$(python -c "import bench; print(bench.sources_sha16())")
  python tools/rocpd_summary.py json $O/stats $O/fetch $O/write gpurun_out/$R/${R}_profile_$c.json "command=$CMD" config=$c sources_sha16=$SHA
  timeout 300 python tools/plan_profile.py $P > gpurun_out/$R/${R}_plan_per_launch_$c.txt 2>&1
  tail -1 $O/bench_stats.json | cut -c1-200 > gpurun_out/$R/${R}_bench_profiled_$c.head
  rm -rf $O/stats $O/fetch $O/write
  head -8 gpurun_out/$R/${R}_kernel_stats_$c.csv
done
