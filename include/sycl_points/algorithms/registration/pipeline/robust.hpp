// Reference-path forwarding header: algorithms/registration/pipeline/robust.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../../amd/registration.hpp"
