#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python tool. usage: bash tools/prof_cmd.sh <tag> <script.py> [args...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- python3 "$@" > gpurun_out/prof_${TAG}_stats.log 2>&1
