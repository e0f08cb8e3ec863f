// Reference-path forwarding header: algorithms/feature/covariance.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/features.hpp"
