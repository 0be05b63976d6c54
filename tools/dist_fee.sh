#!/bin/bash
# what the N > 1 step costs before a byte crosses a link: bench.py at world size 1, single graph vs the distributed step
# (--force-dist: phase graphs + RCCL collectives on one rank) under --comm / --phases / --cu-reserve variants, one box
COMMON="--no-cpu-baseline --fp32-steps 0 --second-steps 0 --profile-steps 0 --steps 30 --warmup 5"
run() { python3 bench.py $COMMON "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.3f ms  %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d.get('launch','')[:90]))" "$@"; }
for i in 1 2; do
run
run --force-dist --comm overlap
run --force-dist --comm overlap --cu-reserve 0
run --force-dist --comm overlap --phases 3
run --force-dist --comm overlap --phases 2 --cu-reserve 0
run --force-dist --comm tail
run --force-dist --comm tail --cu-reserve 0
done
