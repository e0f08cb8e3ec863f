// Reference-path forwarding header: points/point_cloud.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../amd/core.hpp"
