#!/bin/bash
# routing tables (which kernel computes which convolution product) of the convolution models at the bench batch size
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 200 python -u scripts/route_table.py resnet50 256 > gpurun_out/routes_resnet50.txt 2>gpurun_out/routes.err \
 && timeout -k 10 300 python -u scripts/route_table.py inception_v3 256 > gpurun_out/routes_inception_v3.txt 2>>gpurun_out/routes.err \
 && timeout -k 10 200 python -u scripts/route_table.py resnet50 32 > gpurun_out/routes_resnet50_b32.txt 2>>gpurun_out/routes.err \
 && cat gpurun_out/routes_resnet50.txt && tail -3 gpurun_out/routes_inception_v3.txt || { tail -20 gpurun_out/routes.err; exit 1; }
