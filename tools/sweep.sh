#!/bin/bash
# GPU box: window-length sweep of the hot path (tools/bench_cfg.py B N W), inputs resident: the reference's capture lengths
# (8192: iq_stream_client.py:459, 16384: buoy_node.py:364), SURVEY section 8d's secondary sweep at 8 buoys, 3-buoy shapes
cd $GRAFT_REPO_ROOT
for s in "3 256 16384" "3 512 16384" "3 1024 8192" "3 2048 4096" "3 4096 4096" "3 8192 1024" "3 16384 512" "3 65536 128" "3 262144 64" \
         "4 2048 4096" "8 256 8192" "8 512 8192" "8 1024 4096" "8 2048 2048" "8 4096 4096" "8 8192 512" "8 16384 256" "8 65536 64" "16 2048 1024" "16 4096 2048"; do
  python tools/bench_cfg.py $s 2>/dev/null
done
