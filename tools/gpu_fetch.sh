#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the blend kernels for the current build (two PMC passes over profiles/pmc_workload.py)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -o p -- python $GRAFT_REPO_ROOT/profiles/pmc_workload.py > /dev/null 2>&1
  python - $c <<'PY'
import csv, sys, collections
c=sys.argv[1]; d=collections.defaultdict(list)
for r in csv.DictReader(open(f'/tmp/pm_{c}/p_counter_collection.csv')):
    if 'blend' in r['Kernel_Name'] or 'row_sum' in r['Kernel_Name']: d[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k,v in d.items(): print(c, k, round(sum(v)/len(v)/1e3,1), 'MB raw')
PY
done
