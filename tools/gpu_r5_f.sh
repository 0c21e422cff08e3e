#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r5f; mkdir -p $o
timeout 60 python tools/dbg_sp_graph.py keep-graph > $o/sp_graph_keep.log 2>&1; echo "sp graph (graph alive at destroy) rc=$?"; tail -4 $o/sp_graph_keep.log
