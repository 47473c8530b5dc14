#!/bin/bash
# usage: tools/pmc_multi.sh <kernel-substring> <counters...> -- <python script and args>
# like tools/pmc_pass.sh, but prints the per-launch means for EVERY kernel whose name contains the substring (one row per kernel)
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
import csv,sys,collections,re
d=collections.defaultdict(float); n=collections.defaultdict(int); t=collections.defaultdict(float); seen=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    kn=r["Kernel_Name"]
    if sys.argv[2] in kn:
        k=re.sub(r"\(.*", "", kn)[:60]
        d[(k,r["Counter_Name"])]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"]); t[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k in sorted(seen):
    print(k, "avg_us", round(t[k]/len(seen[k])/1e3,1), "launches", len(seen[k]), " ".join(f"{c}={round(d[(kk,c)]/n[(kk,c)])}" for (kk,c) in sorted(d) if kk==k))
PY
