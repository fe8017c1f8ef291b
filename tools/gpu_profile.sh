#!/bin/bash
# rocprofv3 kernel-trace + stats of a short bench run; summary copied to gpurun_out/prof_<tag>/
set -u
tag=${1:-r01}; shift || true
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out/prof_$tag"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python "$R/bench.py" "$@" > "$R/gpurun_out/prof_$tag/bench_under_profiler.log" 2>&1
echo "rocprof exit: $?" >> "$R/gpurun_out/prof_$tag/bench_under_profiler.log"
find /tmp/prof_$tag -name "*stats*.csv" -exec cp {} "$R/gpurun_out/prof_$tag/" \;
# per-kernel summary of the trace (count, avg/min/max us) independent of rocprof's own stats file
python - "$R/gpurun_out/prof_$tag" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob("/tmp/prof_*/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    with open(out + "/kernel_summary.csv", "w") as w:
        w.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            w.write(f"\"{k[:90]}\",{len(v)},{sum(v)/1e3:.3f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/tot:.2f}\n")
PY
ls -la /tmp/prof_$tag/* | head -20 >> "$R/gpurun_out/prof_$tag/bench_under_profiler.log"
