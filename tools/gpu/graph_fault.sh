#!/bin/bash
# rocprofv3 --kernel-trace against hipGraph replay: a minimal program (tools/native/graph_repro.hip), graphs of N kernels with K
# bytes of by-value arguments.  (profiles/r04/hipgraph_frames.md)
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/graphfault
cd /tmp && export TMPDIR=/tmp
for CFG in "133 128 256 1" "8 4096 0 1" "133 100 256 0" "133 120 256 0"; do
  set -- $CFG
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/graphfault/c$1_$2_$3 --output-format csv -- $ROOT/tools/native/graph_repro $1 $2 $3 $4 > $ROOT/gpurun_out/graph_repro_$1_$2_$3.txt 2>&1
  echo "nodes=$1 launches=$2 argbytes=$3 sync_each=$4 rc=$? sigsegv=$(grep -c SIGSEGV $ROOT/gpurun_out/graph_repro_$1_$2_$3.txt) $(grep '^nodes' $ROOT/gpurun_out/graph_repro_$1_$2_$3.txt)"
done
exit 0
