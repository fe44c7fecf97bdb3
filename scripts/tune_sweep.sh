# quick sweep of the raytracer's scheduling knobs on the bench workload (Mrays/s per setting)
for t in '{}' '{"rmin":48}' '{"rmin":32}' '{"xmin":48,"rmin":48}' '{"xmin":32,"rmin":32}' '{"chunk":128}' '{"bpc":3}' '{"nocull":1}'; do
  v=$(timeout 120 python bench.py --no-cpu-baseline --no-extra --steps 60 --warmup 10 --tune "$t" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$t -> $v"
done
