bash tools/gpu_tests.sh r06k tests/test_config4.py
python - <<'PY'
import json, time, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from pisces_amd import engine
for k in range(2):
    r = bench.config4_sample(engine, torch)
    print("config4_sample:", {kk: r[kk] for kk in ("value", "frac", "seconds", "host_seconds")})
PY
