#!/bin/bash
# per-shape A/B of the 3x3 / stride-1 Winograd kernels on one box: 2 (fp32 MFMA), 5 / 6 (2-D F(2x2) fp16-split), 9 / 10 (row-Winograd, 8- / 4-row
# items), 0 = what AUTO picks; then SQ counters and the sustained clock (GRBM_GUI_ACTIVE / 8 / duration) of the row kernels.
# usage: ALGOS="109 110 0" SHAPES="head256 layer1" bash tools/winograd_variants.sh > gpurun_out/r04_winograd_variants.txt
cd $GRAFT_REPO_ROOT
SHAPES=${SHAPES:-layer1 layer1res layer2 l2nr layer3 layer4 neck0 neckup1 neckup2 head256 headfirst fpnfirst c4l1 c4l2 c4l3 c4l4 c4first c4head}
for a in ${ALGOS:-102 105 106 109 110 0}; do echo "== algo $a (cnl_conv_params.algo; 100 + v forces variant v where it can run)"; timeout 600 python tools/conv_bench.py $SHAPES --winograd --hints --relu-data --algo $a --reps 10 --check 2>&1 | grep kernel; done
for v in ${PMC_VARIANTS:-9 10}; do
  echo "== PMC winograd$v on the 256 -> 256 head block"
  bash tools/pmc_quick.sh winograd${v}_kernel -- python tools/conv_bench.py head256 --winograd --hints --relu-data --algo $((100 + v)) --reps 3
  echo "== clocks of winograd$v (GRBM_GUI_ACTIVE / 8 / duration)"
  ALGO=$((100 + v)) SHAPES="head256 layer3 l1nr" bash tools/clock_per_launch.sh
done
