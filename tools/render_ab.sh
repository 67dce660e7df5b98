#!/bin/bash
# On the GPU box: render tests + probe timing with the shipped library, phase stamps with the stamps variant.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
timeout -k 10 300 python -m pytest tests/test_gpu_render.py -m gpu -q -x > gpurun_out/${TAG}_render_tests.log 2>&1
tail -2 gpurun_out/${TAG}_render_tests.log
timeout -k 10 120 python tools/render_probe.py 40 2>&1 | tail -1 | tee gpurun_out/${TAG}_render_probe.txt
NPP_AMD_LIB=$PWD/build_ab/libnpp_rstamps.so timeout -k 10 120 python tools/render_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_render_stamps.txt
