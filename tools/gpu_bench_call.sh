#!/bin/bash
# perf-only gpurun call: leaves the ~100 MB of golden fixtures out of the snapshot (the benches do not read them): the push drops from
# ~130 MiB (30-100 s, charged) to ~10 MiB (a few seconds). Pattern syntax that works in .gpurunignore (probed in round 4): a path WITHOUT a
# trailing slash ("tests/golden"), "dir/*", or a glob ("*.npz"); "dir/" with a trailing slash is silently ignored.
cp .gpurunignore /tmp/gpurunignore.bak
printf 'tests/golden\ntests/golden/*\n' >> .gpurunignore
/usr/local/graft/bin/gpurun --timeout ${1:-900} -- 'bash tools/_gpu_call.sh'
rc=$?
cp /tmp/gpurunignore.bak .gpurunignore
exit $rc
