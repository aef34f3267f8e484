bash tools/gpu/prof_inbench.sh > gpurun_out/r12_prof.txt 2>&1; tail -4 gpurun_out/r12_prof.txt | cut -c1-900
python - <<'PY'
import csv,sys
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r["Name"].split("(")[0][-40:]+"|"+str(len(r["Name"]))]=(int(r["Calls"]),float(r["AverageNs"]),float(r["TotalDurationNs"]))
    return d
a=load("gpurun_out/prof_inbench/kernel_stats_inbench.csv"); b=load("gpurun_out/prof_inbench/kernel_stats_alone.csv")
rows=[]
for k in a:
    if k in b and b[k][2]>2e6: rows.append((b[k][2],k,a[k],b[k]))
rows.sort(reverse=True)
print("kernel | calls in-bench/alone | avg us in-bench/alone | ratio")
for t,k,x,y in rows[:25]: print(f"{k[:48]:48s} {x[0]:7d}/{y[0]:7d} {x[1]/1e3:9.1f}/{y[1]/1e3:9.1f} {x[1]/y[1]:.2f}")
PY
