#!/bin/bash
# Lab: the hipGraph replay of the training step under the runtime's graph knobs (strings of libamdhip64.so: DEBUG_HIP_FORCE_GRAPH_QUEUES,
# DEBUG_HIP_GRAPH_BATCH_SIZE, DEBUG_CLR_GRAPH_PACKET_CAPTURE).  bench.py's graph child at 8 tiles: eager vs replay ms per step.
run() { env "$@" timeout -k 10 200 python bench.py --graph-child --batch 8 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())['tiles_8']
print('$*', 'eager', d['eager_ms_per_step'], 'replay', d['graph_ms_per_step'])"; }
run X=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=16
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run X=0
