#!/bin/bash
# perf-only gpurun call: leaves the 68 MB of golden fixtures out of the snapshot (the benches do not read them)
cp .gpurunignore /tmp/gpurunignore.bak
echo "tests/golden/" >> .gpurunignore
/usr/local/graft/bin/gpurun --timeout ${1:-900} -- 'bash tools/_gpu_call.sh'
rc=$?
cp /tmp/gpurunignore.bak .gpurunignore
exit $rc
