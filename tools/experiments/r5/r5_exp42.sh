# closing session: the uniform 16-out-regular graph through the library as it is, and with nearly every row in the sweep (short_row 8 / 2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp42; mkdir -p $out
for sr in 0 8 2; do
  GRAPHMAT_VERBOSE=1 timeout 900 python bench.py --graph uniform --scale 26 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --short-row $sr > $out/uniform_sr$sr.json 2> $out/uniform_sr$sr.err
  echo "short_row $sr: rc $?"; python - <<P
import json
try:
    d=json.loads(open("$out/uniform_sr$sr.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d.get("roofline",{}).get("per_kernel"))
except Exception as e: print("no line", e)
P
  grep -i "sweep\|tiles" $out/uniform_sr$sr.err | head -5 | cut -c1-250
done
