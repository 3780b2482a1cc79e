#!/bin/bash
# usage: tools/pmc_quick.sh <kernel-substring> -- <command...>   : a few SQ counter passes, per-dispatch means for that kernel
kern=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pq
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); d=gpurun_out/pq/$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d -o r -- "$@" > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
for f in glob.glob("$d/**/*_results.db", recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%$kern%' group by counter_name"): print("%-28s n=%d mean=%.4g" % r)
PY
done
