#!/bin/bash
# Same-box A/B of two checkouts of this repository (e.g. the previous round's final commit against the working tree): the default bench of
# each (C1 line + the `also` lines), alternating, ROUNDS times.  The older checkout is a git worktree under tools/_trace/ (ignored by git,
# shipped to the GPU box):   git worktree add -f tools/_trace/r04 <commit> && make -C tools/_trace/r04/centernet-lightning_amd/csrc
# usage (on the GPU box): OLD=tools/_trace/r04 ROUNDS=2 bash tools/round_ab.sh > gpurun_out/r05_round_ab.txt
cd ${GRAFT_REPO_ROOT:-.}
OLD=${OLD:-tools/_trace/r04}
for r in $(seq 1 ${ROUNDS:-2}); do
  for d in $OLD .; do
    echo "== round $r: $d ($(cd $d && git rev-parse --short HEAD 2>/dev/null || echo working-tree))"
    (cd $d && python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --no-accuracy 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('C1', d['value'], 'images/s', d['ms_per_step'], 'ms; decode_gpu_ms', d['decode_gpu_ms'], '; N=1 latency', d['latency_ms_N1']['default'])
for a in d.get('also', []): print(a['config']['workload'][:12], a['value'], 'images/s', a['ms_per_step'], 'ms')
")
  done
done
