python -m nextbestpath_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_network.py -m gpu -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --layers 2> gpurun_out/layers7.log | tail -1 > gpurun_out/bench7.json
python - <<PY
import json
d=json.load(open("gpurun_out/bench7.json"))
print(d["value"], d["ms_per_step"], d["stages"]["nbp_forward"], d["roofline"])
PY
cat gpurun_out/layers7.log | grep -v amdgpu
