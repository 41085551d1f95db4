#!/bin/bash
# round 3, call o: direct 3x3 kernel for Cin % 8 != 0 (49-channel head input): bit-identity, microbench, end-to-end
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -2
timeout 200 python scripts/ubench/conv_bench.py --iters 10 --tag cin49_direct --shapes head49_512_60x80 l3_256_60x80 --out gpurun_out/cin49.jsonl 2>&1 | grep -v "Warn\|amdgpu.ids"
RFX_CONV_DIRECT=0 timeout 200 python scripts/ubench/conv_bench.py --iters 10 --tag cin49_generic --shapes head49_512_60x80 --out gpurun_out/cin49.jsonl 2>&1 | grep -v "Warn\|amdgpu.ids"
timeout 300 python bench.py --config 3 --steps 8 --warmup 3 --no-cpu-baseline --no-qs-leg > gpurun_out/bench_o.log 2> gpurun_out/bench_o.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/bench_o.log") if l.startswith("{")][0])
print(j["value"], j["ms_per_step"], j["roofline"]["all_conv_tflops"])
PY
