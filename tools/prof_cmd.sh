#!/bin/bash
# GPU-box helper: rocprofv3 kernel trace of a command, summary printed + written to gpurun_out/<tag>.md (the
# trace database itself stays on the box).  usage: tools/prof_cmd.sh <tag> <command...>
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out /tmp/prof_$tag
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- "$@" > /tmp/prof_$tag/cmd.log 2>&1 )
tail -3 /tmp/prof_$tag/cmd.log
db=$(find /tmp/prof_$tag -name "*_results.db" | head -1)
python $root/tools/rocprof_summary.py "$db" $root/gpurun_out/$tag.md "$tag" | head -${PROF_LINES:-45}
