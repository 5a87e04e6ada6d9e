#!/bin/bash
# Diagnostic library for tools/gemm_stamps.py: the product GEMM source compiled with -DDS_GEMM_STAMPS.
set -e
cd "$(dirname "$0")/exp"; mkdir -p _build
C=../../dynamicscaler_amd/csrc
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -ffp-contract=on -I$C"
hipcc $F -DDS_GEMM_STAMPS $EXTRA -x hip -c $C/gemm.hip -o _build/gemm_st.o
hipcc $F -c $C/error.cpp -o _build/err.o
hipcc --offload-arch=gfx950 -shared -fPIC -o libgemm_stamps.so _build/gemm_st.o _build/err.o
