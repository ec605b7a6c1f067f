# closing session: the default bench line of the committed bench.py (with the extra leg on a graph without skew)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/s3bench; mkdir -p $out
( time python bench.py > $out/r05_bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3
python - <<P
import json
d=json.loads(open("$out/r05_bench_default.json").read().strip().splitlines()[-1])
print("default bench: ms", d["ms_per_step"], "value", d["value"], "traffic", d["roofline"].get("traffic"))
print(json.dumps(d["extra"]["pagerank_uniform25"])[:900])
print({k:(v.get("median_wall_ms") if isinstance(v,dict) else v) for k,v in d["extra"].items()})
P
