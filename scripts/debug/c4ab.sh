python -m pytest tests/test_gpu_ials.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python bench.py --legs c4 --no-cpu-baseline > gpurun_out/c4_s2.json 2>gpurun_out/c4_s2.err
IRSPACK_AMD_IALS_SHORT2=0 python bench.py --legs c4 --no-cpu-baseline > gpurun_out/c4_s1.json 2>gpurun_out/c4_s1.err
python - <<PY
import json
for f in ("c4_s2","c4_s1"):
    d=json.load(open("gpurun_out/%s.json"%f))
    c=d["secondary"]["c4"]
    print(f, c["cg"]["ms_per_epoch"], c["cg"]["kernels_ms_per_launch"])
PY
