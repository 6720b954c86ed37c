cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3m
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r3m/pmc -- python3 bench.py --workload merkle --steps 2 > gpurun_out/r3m/pmc.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3m/trace -- python3 bench.py --workload merkle --steps 2 > gpurun_out/r3m/trace.json 2>/dev/null
python3 - <<'PY'
import csv, glob, collections
f=glob.glob("gpurun_out/r3m/pmc/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_hash_parents" in k or "k_proofs_lds" in k:
        agg[(k.split("(")[0][-24:], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
f=glob.glob("gpurun_out/r3m/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_hash_parents" in r["Kernel_Name"] or "k_proofs_lds" in r["Kernel_Name"]]
agg=collections.defaultdict(list)
for r in rows: agg[(r["Kernel_Name"].split("(")[0][-24:], r["Grid_Size"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-int(kv[0][1])): print(k, len(v), "us avg", round(sum(v)/len(v),1))
PY
find gpurun_out/r3m -name "*.csv" -size +4M -delete
