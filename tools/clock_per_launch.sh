#!/bin/bash
# cycles per launch (GRBM_GUI_ACTIVE / 8) and the sustained clock of the Winograd kernel a shape takes: ALGO=109 SHAPES="head256 layer3" [LIB=path.so]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in ${SHAPES:-head256}; do
  rm -rf gpurun_out/clk9
  if [ -n "$LIB" ]; then export CENTERNET_GFX950_LIB=$LIB; fi
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d gpurun_out/clk9 -o r -- python tools/conv_bench.py $s --winograd --hints --relu-data --algo ${ALGO:-109} --reps 5 > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
for f in glob.glob("gpurun_out/clk9/**/*_results.db", recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select kernel_name, count(*), avg(value), avg(end-start) from counters_collection where kernel_name like '%winograd%_kernel%' and kernel_name not like '%weights%' group by kernel_name"):
        print("$s ${LIB:-lib} %s: cycles %.4gM  dur %.1f us  clock %.3f GHz" % (r[0].split('(')[0][-30:], r[2]/8/1e6, r[3]/1e3, r[2]/8/r[3]))
PY
done
rm -rf gpurun_out/clk9
