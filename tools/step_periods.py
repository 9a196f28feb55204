"""Step-to-step periods and per-kernel mean durations from a rocprofv3 --kernel-trace CSV of `bench.py --train-only
--no-phase-timing` (a step starts at its lazy_replay launch):  python tools/step_periods.py p_kernel_trace.csv"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "lazy_replay_kernel" in r["Kernel_Name"]]
per = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(heads, heads[1:])]
print("periods us:", " ".join(f"{p:.0f}" for p in per))
plain = [p for p in per if p < 2 * statistics.median(per)]
print(f"median {statistics.median(per):.1f} us, mean without the flush steps {statistics.mean(plain):.1f} us, mean {statistics.mean(per):.1f} us")
dur = collections.defaultdict(list)
for r in rows[heads[0]:heads[-1]]:
    name = r["Kernel_Name"].replace("void ", "").replace("satrans::", "").split("(")[0][:70]
    dur[(r.get("Queue_Id", "?"), name)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
n = len(per)
tot = collections.defaultdict(float)
for (q, name), v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"queue {q:>2s} {sum(v) / n:9.1f} us/step  {len(v) / n:5.2f} launches/step  mean {statistics.mean(v):8.1f}  {name}")
    tot[q] += sum(v) / n
print("busy us/step per queue:", {q: round(t, 1) for q, t in tot.items()})
