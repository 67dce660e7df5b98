#!/bin/bash
# Compile ONE step-kernel instantiation (G lanes per env, plain TU) and print its register / scratch / spill figures.
#   tools/regs_experiment.sh [G=16] [extra hipcc flags...]
G=${1:-16}; shift
cd "$(dirname "$0")/.."
mkdir -p /tmp/npp_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -Wno-unused-value -Wno-unused-result -I include -DNPP_TU=0 -DNPP_ONLY_G=$G "$@" -Rpass-analysis=kernel-resource-usage \
  -save-temps=obj -c nclone_amd/csrc/npp_kernels.hip -o /tmp/npp_exp/k_g$G.o 2> /tmp/npp_exp/k_g$G.log
python tools/kernel_resources.py /tmp/npp_exp/k_g$G.log | grep step_kernel
