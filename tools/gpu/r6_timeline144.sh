#!/bin/bash
# round 6: per-workgroup cycle timelines (PRIMX_GEMM_PROF=1: synchronous launches, stamps of compute wave 0) of every GEMM of one configs[1]
# forward, the loader-wave 128 x 144 kernel included (K = 1152 family: proj / cproj producers <., 6>, to_q consumer <., 7>)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_GEMM_PROF=1 timeout 600 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 2 --warmup 1 --repeats 1 2> gpurun_out/r6_timeline144.err > /dev/null
grep "^gemm" gpurun_out/r6_timeline144.err | python -c "
import sys, re, collections, statistics
rows = collections.defaultdict(list)
for l in sys.stdin:
    m = re.match(r'(\S+) M=(\d+) N=(\d+) K=(\d+): (\d+) workgroups, events ([\d.]+) us, first start -> last end ([\d.]+) us, mean start offset ([\d.]+) us, shader clock ([\d.]+) GHz.*entry->tile0 (\d+) \| main loop (\d+) \| epilogue (\d+) \(LDS staging (\d+), read\+store issue (\d+)\)', l)
    if m: rows[(m.group(1),) + tuple(int(m.group(i)) for i in (2, 3, 4, 5))].append([float(m.group(i)) for i in range(6, 15)])
print('kernel M N K workgroups | launches | medians: events us, first->last us, start offset us, GHz | cycles: entry, main loop, epilogue (parking, walk issue)')
for k, v in sorted(rows.items()):
    med = [statistics.median(c) for c in zip(*v)]
    print(*k, '|', len(v), '|', *['%.1f' % x for x in med[:3]], '%.2f' % med[3], '|', *['%.0f' % x for x in med[4:]])
"
