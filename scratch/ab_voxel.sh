#!/bin/bash
# A/B of a scratch library (B) against the shipped one (A) on one box: the voxel stages alone, alternating.
cd $GRAFT_REPO_ROOT
LIB=$1
cp sycl_points_amd/lib/libsycl_points_amd.so /tmp/lib_keep.so
for tag in A B A B A B; do
  if [ $tag = A ]; then cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so; else cp $LIB sycl_points_amd/lib/libsycl_points_amd.so; fi
  echo -n "$tag "; python scratch/voxel_stage.py 2>/dev/null | tail -1
done
cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so
