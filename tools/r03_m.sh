mkdir -p gpurun_out/r03
out=gpurun_out/r03/cu_mask.txt; : > $out
for cfgs in "0 0" "4 0" "8 0" "8 1" "4 1" "12 1" "16 1"; do set -- $cfgs
DSM_ENC_CUS=$1 DSM_LM_CUS_EXCL=$2 timeout -k 10 200 python bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs 400 --dot-mode 1 --steps 100 > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - >> $out <<PY
import json
d=json.load(open("/tmp/b.json"))
print("enc_cus $1 lm_excl $2: B=64 ms_per_step %.3f" % d["ms_per_step"], {k:round(v.get("ms_per_step",0),2) for k,v in d["capacity"]["legs"].items()})
PY
tail -1 $out
done
