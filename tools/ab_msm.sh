#!/bin/bash
# A/B of the accumulation kernel's knobs on the headline workload (run on the GPU box): prints value / latency / kernel time
for rw in 4 5; do for pf in 0 1; do
  KH_ROOM_WAVES=$rw KH_ACC_PREFETCH=$pf python bench.py --no-oplist --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('room_waves=$rw prefetch=$pf  value %.1f Mscalar/s  ms/step %.3f  sync %.3f  kernel %.3f  phases %s' % (d['value'], d['ms_per_step'], d['ms_per_step_synchronous'], d['roofline']['kernel_ms'], {k: round(v,3) for k,v in d['phases_ms'].items()}))"
done; done
