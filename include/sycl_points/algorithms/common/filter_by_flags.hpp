// Reference-path forwarding header: algorithms/common/filter_by_flags.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/features.hpp"
