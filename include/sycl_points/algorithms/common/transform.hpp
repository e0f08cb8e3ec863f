// Reference-path forwarding header: algorithms/common/transform.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/features.hpp"
