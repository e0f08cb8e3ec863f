// Reference-path forwarding header: algorithms/registration/registration_pipeline.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../../amd/registration.hpp"
