cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# effective clock (GRBM_GUI_ACTIVE / duration) of the Winograd kernels on the 256 -> 256 head block: variants 2 (fp32 MFMA), 5 / 6 (fp16-split F(2x2)), 8 (F(4x4))
for v in 2 5 6 8; do
  rm -rf gpurun_out/clk$v
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/clk$v -o r -- python tools/conv_bench.py head256 --winograd --algo $((100 + v)) > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
for f in glob.glob("gpurun_out/clk$v/**/*_results.db", recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from counters_collection where kernel_name like '%winograd%_kernel%' and kernel_name not like '%weights%' group by kernel_name, counter_name"):
        print("variant $v", r[0][:40], r[1], "n=%d mean=%.4g dur_ns=%.4g -> %.3f GHz" % (r[2], r[3], r[4], r[3]/r[4]))
PY
done
