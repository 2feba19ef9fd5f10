#!/bin/bash
# times tools/frozen_pa_loop.py (frozen-PA loss step, EXP_B x 200) under rocprofv3 for each experiment library given: kernel average in us
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" != "in-tree" ]; then export OPENDPD_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset OPENDPD_HIP_LIB; fi
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/tools/frozen_pa_loop.py > /tmp/pp.log 2>&1
  python3 - "$lib" <<'PY'
import csv, glob, sys
for f in glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("lossdx", "pipe", "gru16n")):
            print(f"{sys.argv[1]}: {r['Name'][:60]} avg {float(r['AverageNs'])/1e3:.1f} us (n={r['Calls']})", open('/tmp/pp.log').read().strip().splitlines()[-1])
PY
done
