#!/bin/bash
# Phase accounting of the two wide edge kernels (forward, then backward) at the H = 300 configuration.
# Needs lib/libgcn_uelb_hip_acc1.so / _acc2.so (builds with -DGCN_PHASE_ACC=1 / =2).  Usage on the GPU box: bash tools/phase_acc_both.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
GCN_UELB_LIB=graph-convnet-uelb_amd/lib/libgcn_uelb_hip_acc1.so timeout -k 10 200 python tools/phase_acc.py synthetic_100n_50c_h300_l30 fwd > gpurun_out/phase_acc_fwd_$tag.txt 2>&1 || { tail -5 gpurun_out/phase_acc_fwd_$tag.txt; exit 1; }
GCN_UELB_LIB=graph-convnet-uelb_amd/lib/libgcn_uelb_hip_acc2.so timeout -k 10 200 python tools/phase_acc.py synthetic_100n_50c_h300_l30 bwd > gpurun_out/phase_acc_bwd_$tag.txt 2>&1 || { tail -5 gpurun_out/phase_acc_bwd_$tag.txt; exit 1; }
cat gpurun_out/phase_acc_fwd_$tag.txt gpurun_out/phase_acc_bwd_$tag.txt
