"""One training step as a timeline: every kernel of a step in dispatch order with its start (relative to the step), duration and
queue, from a rocprofv3 --kernel-trace CSV of `bench.py --train-only`.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o p -- python3 bench.py --train-only --no-phase-timing
    python tools/step_timeline.py /tmp/prof/p_kernel_trace.csv [step index, default 12] > profiles/rNN_step_timeline.txt

A step starts at its lazy_replay launch.  Back-to-back dependent launches show >= ~4.8 us each however little they do (the launch
floor); kernels of the side stream (next batch's ids -> rows, sort, bucketing) overlap the tail of the launch stream."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
heads = [i for i, r in enumerate(rows) if "lazy_replay_kernel" in r["Kernel_Name"]]
a, b = heads[which], heads[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"{'start us':>9s} {'dur us':>8s}  queue  kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("satrans::", "").split("(")[0][:84]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r.get('Queue_Id', '?'):>5s}  {name}")
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
