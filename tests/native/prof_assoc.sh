cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/r3_assoc_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_assoc_prof -o t -- python3 $R/tests/native/perf_assoc_loop.py 100 > $O/r3_assoc_prof.log 2>&1
python3 - <<'PY'
import csv, os, collections
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out"
rows=list(csv.DictReader(open(O+"/r3_assoc_prof/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last run() of the script = the last ~100 frames: take the launches after the last detector kernel
last_det=max(i for i,r in enumerate(rows) if "conv_gemm_big" in r["Kernel_Name"] or "attention_x3" in r["Kernel_Name"])
tail=rows[last_det+1:]
agg=collections.OrderedDict()
for r in tail:
    k=r["Kernel_Name"][:100]; a=agg.setdefault(k,[0,0]); a[0]+=1; a[1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
tot=sum(v[1] for v in agg.values())
print("launches after the detector: %d, kernel time %.1f ms, span %.1f ms"%(len(tail),tot/1e6,(int(tail[-1]["End_Timestamp"])-int(tail[0]["Start_Timestamp"]))/1e6))
for k,(n,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:30]:
    print("%6d x %8.1f us avg  %8.2f ms  %s"%(n,t/n/1e3,t/1e6,k))
PY
rm -rf $O/r3_assoc_prof
