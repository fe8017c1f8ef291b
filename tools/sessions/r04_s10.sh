#!/bin/bash
# session 10: config 5's fp8 path at size: 14B-width block(s) at L = 75 600 vs the fp32 checker, and the 40-layer forward time
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/r04_s10; mkdir -p $out
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "fp8_linears" 2>&1 | tail -25 > $out/pytest_fp8_at_size.log; cat $out/pytest_fp8_at_size.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/fullsize_parity.json"))
for k, v in d.items():
    if "fp8" in k:
        print(k, {a: b for a, b in v.items() if a != "layers"}, [round(r["e_hip"], 4) for r in v["layers"]], [round(r["e_ac"], 4) for r in v["layers"]])
PY
for f in 0 2; do timeout 600 python tools/bench_wan14b.py --fp8_linear $f > $out/wan14b_720p_fp8_linear$f.json.log 2>&1; tail -1 $out/wan14b_720p_fp8_linear$f.json.log; done
