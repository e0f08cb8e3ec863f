#!/bin/bash
# A/B of a scratch library (B) against the shipped one (A) on one box with any python script: bash scratch/ab_py.sh <lib> <script>
cd $GRAFT_REPO_ROOT
LIB=$1; SCRIPT=$2
cp sycl_points_amd/lib/libsycl_points_amd.so /tmp/lib_keep.so
for tag in A B A B A B; do
  if [ $tag = A ]; then cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so; else cp $LIB sycl_points_amd/lib/libsycl_points_amd.so; fi
  echo -n "$tag "; python $SCRIPT 2>/dev/null | tail -1
done
cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so
