mkdir -p gpurun_out/r03/tts_trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/tts_trace -- python3 bench.py --workload tts --batch 32 --steps 30 --warmup 3 > gpurun_out/r03/tts_trace/bench.json 2> gpurun_out/r03/tts_trace/bench.err || { tail -5 gpurun_out/r03/tts_trace/bench.err; exit 1; }
cat gpurun_out/r03/tts_trace/bench.json | cut -c1-300
python tools/trace_table.py gpurun_out/r03/tts_trace > gpurun_out/r03/tts_trace/table.txt
f=$(find gpurun_out/r03/tts_trace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r03/tts_trace/kernel_stats.csv
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r03/tts_trace/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 20% of the trace = steady-state steps
n=len(rows); rows=rows[int(n*0.6):]
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
span=int(rows[-1]["End_Timestamp"])-int(rows[0]["Start_Timestamp"])
print("steady tail: %d launches, busy %.2f ms, span %.2f ms, busy/span %.2f, avg kernel %.2f us, avg pitch %.2f us"%(len(rows),busy/1e6,span/1e6,busy/span,busy/len(rows)/1e3,span/len(rows)/1e3))
PY
find gpurun_out/r03/tts_trace -name "*kernel_trace.csv" -size +30M -delete
