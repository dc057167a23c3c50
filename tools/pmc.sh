#!/bin/bash
# Usage: bash tools/pmc.sh <tag> <bench flags> -- "<counter group 1>" "<counter group 2>" ...
# One rocprofv3 --pmc pass per group (PMC never combined with sys/hip traces).
set -e
TAG=$1; shift
BARGS=""
while [ "$1" != "--" ]; do BARGS="$BARGS $1"; shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "$@"; do
  echo "group $i: $G" >> $OUT/progress.log
  timeout -k 5 100 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile $BARGS > /dev/null 2> $OUT/g$i.log || { tail -5 $OUT/g$i.log; }
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
res=collections.defaultdict(dict)
for f in glob.glob(out+'/g*/*/*counter_collection.csv'):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in d.items():
        for c,vals in v.items():
            res[k][c]=max(vals)
for k in res:
    if 'meshletCull' in k or 'instanceClassify' in k and 'ILi0' in k:
        print(k[:80])
        for c in sorted(res[k]): print('   %-40s %.4g'%(c,res[k][c]))
PY
