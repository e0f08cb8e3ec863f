// Reference-path forwarding header: algorithms/knn/result.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/knn.hpp"
