mkdir -p gpurun_out/r03
out=gpurun_out/r03/cu_mask2.txt; : > $out
run() { tag="$1"; shift; env "$@" timeout -k 10 200 python bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs "" --steps 150 > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - >> $out <<PY
import json
d=json.load(open("/tmp/b.json"))
print("$tag: B=64 ms_per_step %.3f  attn live frac %.3f" % (d["ms_per_step"], d["roofline"]["frac"]))
PY
tail -1 $out; }
run "default" A=1
run "enc 16 CUs/XCD, LM everywhere" DSM_ENC_CUS=16
run "enc 24 CUs/XCD, LM everywhere" DSM_ENC_CUS=24
run "enc 20 CUs/XCD, LM on the other 12" DSM_ENC_CUS=20 DSM_LM_CUS_EXCL=1
run "stream priorities" DSM_STREAM_PRIO=1
run "one LM group" DSM_LM_GROUPS=1
run "one LM group + priorities" DSM_LM_GROUPS=1 DSM_STREAM_PRIO=1
run "three LM groups" DSM_LM_GROUPS=3
