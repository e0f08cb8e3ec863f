// Reference-path forwarding header: algorithms/filter/voxel_downsampling.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/features.hpp"
