#!/usr/bin/env python3
"""The per-scan chain of the handler on device-resident clouds - extractCloud -> downsampleCurrentScan -> scan2MapOptimization
(reference src/mapOptmization.cpp:257-265) - as bench.py's `chain` object reports it; one JSON line.

   python tools/bench_chain.py [frames = 50] [points per frame = 30000] [raw scan points = 120000]
   rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/bench_chain.py      (per-kernel breakdown)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = [int(v) for v in sys.argv[1:4]]
print(json.dumps(bench.chain_figures(0, *a)))
