// Reference-path forwarding header: algorithms/registration/linearized_result.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/registration.hpp"
