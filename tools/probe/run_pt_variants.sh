mkdir -p gpurun_out/r4
for v in ${PT_VARIANTS:-base vp1 vp3}; do
  export LEON_DEBUG_LIB=build/ab/$v/libleon_hip.so
  bash tools/probe/pipe_trace.sh gpurun_out/r4/pt_$v --varied --loop 240 > /dev/null
done
unset LEON_DEBUG_LIB
export PT_VARIANTS="${PT_VARIANTS:-base vp1 vp3}"
python - <<'PY'
import json
import os
for n in os.environ.get("PT_VARIANTS", "base vp1 vp3").split():
    d=json.load(open("gpurun_out/r4/pt_%s/summary.json"%n)); k=d["kernels"]
    print(n, round(d["pictures_per_s"]), "parse %.2f blocks %.2f index %.2f B %.2f" % tuple(k[x]["avg_ms"] for x in ("leon::k_vlc_parse","leon::k_vlc_blocks","leon::k_vlc_index","k_recon_display<3, true, false>")))
PY
