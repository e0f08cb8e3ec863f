#!/bin/bash
# A/B of a scratch library (B) against the shipped one (A) on one box: the reference's example loop through the facade, alternating.
# usage (GPU box, repo root): bash scratch/ab_example.sh scratch/lib_b.so
cd $GRAFT_REPO_ROOT
LIB=$1
cp sycl_points_amd/lib/libsycl_points_amd.so /tmp/lib_keep.so
for tag in A B A B A B; do
  if [ $tag = A ]; then cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so; else cp $LIB sycl_points_amd/lib/libsycl_points_amd.so; fi
  (cd tests/cpp && ./example_registration ../golden/source.ply ../golden/target.ply 200 20 | grep -E "Downsampling|2a|2b|TOTAL|Registration|kNN" | awk -v t=$tag '{printf "%s %s | ", t, $0} END {print ""}')
done
cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so
