// Reference-path forwarding header: utils/sycl_utils.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../amd/core.hpp"
