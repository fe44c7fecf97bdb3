mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_bench_n1.jsonl 2> gpurun_out/r04_bench_n1.err; echo rc=$?
tail -c 3000 gpurun_out/r04_bench_n1.err
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r04_bench_n1.jsonl") if x.startswith("{")][-1]
d = json.loads(l)
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus")})
print(d["roofline"]["frac"], d["roofline"].get("timed_schedule"))
print(d.get("cpu_baseline"))
print(d.get("seam"))
ow = d.get("other_workloads", {})
for k, v in ow.items():
    if not isinstance(v, dict) or k in ("shadowmap_1024_us",): print(k, v)
print(ow.get("render_cli_bench"))
PY
