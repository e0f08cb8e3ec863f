// Reference-path forwarding header: utils/eigen_utils.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../amd/core.hpp"
