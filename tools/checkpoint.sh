# Full -m gpu suite + the default bench line; outputs under gpurun_out/$1 (run on the GPU box: gpurun -- bash tools/checkpoint.sh r02x)
O=gpurun_out/${1:-ckpt}
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12 > $O/pytest.txt
python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.txt
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 1), "host_path", round(d["host_path"]["ms_per_step"], 1),
      "f32_path", round(d["f32_path"]["ms_per_step"], 1), "frac", round(d["roofline"]["frac"], 3),
      "cpu", round(d["cpu_baseline"]["value"], 1))
print(d["kernel_ms_per_step"])
PY
