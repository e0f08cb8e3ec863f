// Reference-path forwarding header: algorithms/mapping/voxel_hash_map.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/mapping.hpp"
