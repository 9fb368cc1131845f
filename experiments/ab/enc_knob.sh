C64="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement --part enc"
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))"; }
for v in default 1 8 24 48 96; do
  if [ $v = default ]; then echo "enc B=64 default: $(bash -c "$C64" 2>/dev/null | ms)"; else echo "enc B=64 DSM_CHUNK_LOOP_MIN=$v: $(DSM_CHUNK_LOOP_MIN=$v bash -c "$C64" 2>/dev/null | ms)"; fi
done
