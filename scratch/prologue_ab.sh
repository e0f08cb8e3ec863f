#!/bin/bash
# in-kernel stamps of the per-iteration kernel's prologue (make timing build), run on the GPU box from the repo root
cd $GRAFT_REPO_ROOT
make -C sycl_points_amd/csrc timing -s 2>&1 | grep -E "error" | head -3
cp sycl_points_amd/lib/libsycl_points_amd.so /tmp/lib_keep.so
cp scratch/lib_timing.so sycl_points_amd/lib/libsycl_points_amd.so
SP_TM=1 python scratch/exp_prologue.py 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so
