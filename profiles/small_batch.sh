mkdir -p gpurun_out/sb3
for w in 1 2 8; do
  for f in "" "--graph" "--x6" "--x6 --graph" "--split" "--split --graph" "--bf16" "--bf16 --graph"; do
    tag=$(echo "w${w}${f}" | tr -d ' ' | tr -- '-' '_')
    python bench.py --windows $w --steps 40 --warmup 8 --no-cpu-baseline --data synthetic $f 2>/dev/null | tail -1 > gpurun_out/sb3/$tag.json
  done
done
python - <<'P'
import json, glob, os
rows = []
for p in sorted(glob.glob("gpurun_out/sb3/*.json")):
    try: j = json.load(open(p))
    except Exception: continue
    rows.append({"run": os.path.basename(p)[:-5], "windows": j["config"]["windows_per_gpu"], "ms_per_forward": round(j["ms_per_step"], 4),
                 "samples_per_s": round(j["value"]), "dtype": j["dtype"], "kernel_ms": {k: round(v, 4) for k, v in j["kernel_ms"].items()}})
json.dump(rows, open("gpurun_out/sb3/summary.json", "w"), indent=1)
for r in rows: print(r["run"], r["ms_per_forward"])
P
