cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r06
out=gpurun_out/pmc_c5sq
mkdir -p $out
run() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out/$name -o p -- python3 bench.py --config c5 --table-rows 4000000 --steps 2 --warmup 1 --train-only --no-phase-timing --no-other-configs < /dev/null > $out/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for path in glob.glob("gpurun_out/pmc_c5sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k=(r.get("Kernel_Name") or r.get("Kernel Name")).split("(")[0].replace("void ","").replace("satrans::","")[:40]
        a=acc[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("gpurun_out/r06/c5_sq_counters.txt","w") as f:
    for k in sorted(acc):
        if not k.startswith("gen_"): continue
        f.write(k+"\n")
        for c,(v,n) in sorted(acc[k].items()): f.write(f"   {c:28s} {v/n:16.1f}  (x{n})\n")
PY
rm -rf $out
cat gpurun_out/r06/c5_sq_counters.txt | head -150
