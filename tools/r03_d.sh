mkdir -p gpurun_out/r03
python - <<'PY'
import sys; sys.path.insert(0,'.')
import dsm_amd
from dsm_amd import synth
cfg = dsm_amd.config_stt_1b_en_fr()
print(synth.make_synth_weights(cfg, "/tmp/dsm_weights", tag="stt-1b-en_fr"))
PY
for b in 64 400 2048; do
  timeout -k 10 300 ./tools/host_path_bench /tmp/dsm_weights/stt-1b-en_fr.lm.safetensors /tmp/dsm_weights/stt-1b-en_fr.mimi.safetensors $b 16 8 > gpurun_out/r03/host_path_b$b.json 2> gpurun_out/r03/host_path_b$b.err; echo "rc=$?"; cat gpurun_out/r03/host_path_b$b.json; tail -2 gpurun_out/r03/host_path_b$b.err
done
