for s in 2 3 2 3 4; do python bench.py --steps 20 --warmup 5 --streams $s --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json; python - <<'PY'
import json
d=json.loads(open('/tmp/b.json').read()); r=d["roofline"]
print("streams tried", d["config"]["streams_tried"], "used", d["config"]["streams"], "multi", r["multi_stream"]["regions_us"], "one", r["one_stream"]["regions_us"])
PY
done
