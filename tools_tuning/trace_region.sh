# kernel timeline of the last region of short_region.py <schedule>; usage: trace_region.sh <schedule> <tag> [env...]
cd /tmp && export TMPDIR=/tmp
sched=$1; tag=$2
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools_tuning/short_region.py $sched > $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/trace_$tag/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last region = last N dispatches after the last big idle gap (> 200 us)
st=[int(r["Start_Timestamp"]) for r in rows]; en=[int(r["End_Timestamp"]) for r in rows]
cut=0
for i in range(1,len(rows)):
    if st[i]-max(en[:i][-50:])>150000: cut=i
t0=st[cut]
out=open("$GRAFT_REPO_ROOT/gpurun_out/trace_$tag.txt","w")
for r in rows[cut:]:
    name=r["Kernel_Name"].split("(")[0][:60]
    out.write("%8.1f %8.1f %7.1f  %s  q=%s\n"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,name,r.get("Queue_Id","")))
out.close()
PY
cat $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.txt
