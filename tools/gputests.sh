#!/bin/bash
# tools/gputests.sh [pytest args] -- the GPU suite on the box, summary lines only (full log: gpurun_out/gputests.log)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/gputests.log 2>&1
rc=$?
grep -E "passed|failed|error" gpurun_out/gputests.log | tail -5
echo "pytest rc=$rc"
exit $rc
