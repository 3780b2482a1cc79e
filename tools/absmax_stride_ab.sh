#!/bin/bash
# A/B on one box: per-image maxima packed into one cache line (`make -C centernet-lightning_amd/csrc variant TAG=ams1 EXTRA=-DCNL_ABSMAX_STRIDE=1`, the layout
# of ABI <= 9) against the product (cnl_absmax_stride() = 32 floats: one line per image) — standalone with the array zeroed before every launch, per launch
# inside the network, and the C1 step.  -> profiles/r04_absmax_stride.txt
cd $GRAFT_REPO_ROOT
SH="layer1 layer1res layer2 layer3 layer4 neck0 head256"
for lib in ams1 product ams1 product; do
  if [ $lib != product ]; then export CENTERNET_GFX950_LIB=$PWD/tools/ablibs/libcnl_$lib.so; else unset CENTERNET_GFX950_LIB; fi
  echo "== $lib: standalone, y_absmax zeroed before every launch"; timeout 300 python tools/conv_bench.py $SH --winograd --hints --zero-ymax --relu-data --reps 20 2>&1 | grep kernel
  echo "== $lib: in the network"; timeout 300 python tools/plan_profile.py 2>&1 | grep -E "layer1.1.conv1|layer2.4.conv1|layer3.8.conv1|layer4.14.conv1|heads.heatmap.block|conv launches"
  timeout 600 python bench.py --no-cpu-baseline --no-variants --no-also --no-accuracy 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C1 $lib:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"
done
