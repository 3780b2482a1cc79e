#!/bin/bash
# A/B on one box: the library with unconditional max |y| atomics (`make -C centernet-lightning_amd/csrc variant TAG=nopeek EXTRA=-DCNL_NO_PEEK` -> tools/ablibs/libcnl_nopeek.so, the
# behaviour of rounds 1-3) against the product (peek before raising: cnl_common.h cnl::report_max) — per-layer standalone and the C1 step.
cd $GRAFT_REPO_ROOT
SH="layer1 layer1res layer2 layer3 layer4 neck0 head256 headfirst"
for i in 1 2; do
  echo "== every report an atomic (rounds 1-3)"; CENTERNET_GFX950_LIB=$PWD/tools/ablibs/libcnl_nopeek.so timeout 300 python tools/conv_bench.py $SH --winograd --hints --relu-data --reps 20 2>&1 | grep kernel
  echo "== peek, then raise (round 4)"; timeout 300 python tools/conv_bench.py $SH --winograd --hints --relu-data --reps 20 2>&1 | grep kernel
done
echo "== no report at all (y_absmax = NULL)"; timeout 300 python tools/conv_bench.py $SH --winograd --hints --no-ymax --relu-data --reps 20 2>&1 | grep kernel
for i in 1 2; do
  for lib in nopeek product; do
    if [ $lib = nopeek ]; then export CENTERNET_GFX950_LIB=$PWD/tools/ablibs/libcnl_nopeek.so; else unset CENTERNET_GFX950_LIB; fi
    timeout 600 python bench.py --no-cpu-baseline --no-variants --no-also --no-accuracy 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C1 $lib:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"
  done
done
