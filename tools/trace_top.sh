# per (kernel, grid) time breakdown of the bench step: tools/trace_top.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trace_top; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o out --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $* > $O/log 2>&1
python - <<EOF
import csv, glob, collections, re
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sprc" not in r["Kernel_Name"]: continue
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sprc::", "")[:60]
        key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))
        acc[key][0] += 1; acc[key][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(v[1] for v in acc.values())
print(f"total sprc kernel time {tot/4e6:.2f} ms per step (4 steps traced)")
for k, v in sorted(acc.items(), key=lambda x: -x[1][1])[:40]:
    print(f"{v[1]/4e6:8.3f} ms/step  {v[0]//4:4d} launches/step  avg {v[1]/v[0]/1e3:8.1f} us  grid {k[1]:>8} wg {k[2]:>4}  {k[0]}")
EOF
