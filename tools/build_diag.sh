#!/bin/sh
# Diagnostics build of the library (-DMGCN_DIAG: ablation switches in the fused layer kernel); use with MGCN_LIB=...
cd "$(dirname "$0")/../kgc-gcn_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -DMGCN_DIAG \
  -o libmgcn_hip_diag.so csr_build.cpp ingest.cpp aggregate.hip dense.hip layer_fused.hip layer_fused2.hip layer_fused3.hip layer_fused4.hip train_layer.hip
