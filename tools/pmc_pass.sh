#!/bin/bash
# usage: tools/pmc_pass.sh <kernel-substring> <counters...> -- <python script and args>
# one rocprofv3 --pmc pass (own run, kernel-trace only), prints the per-launch mean of every counter for kernels matching the substring
sub="$1"; shift
ctr=()
while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d /tmp/pp -- python3 "$@" > /dev/null 2>&1 < /dev/null
f=$(find /tmp/pp -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter file"; exit 1; }
python3 - "$f" "$sub" <<PY
import csv,sys,collections
d=collections.defaultdict(float); n=collections.defaultdict(int); t=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        d[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); t["ns"]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k in d: print(k, round(d[k]/n[k]), "launches", n[k])
if seen: print("avg_us", round(t["ns"]/len(seen)/1e3,1))
PY
