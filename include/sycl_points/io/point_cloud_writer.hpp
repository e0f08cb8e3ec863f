// Reference-path forwarding header: io/point_cloud_writer.hpp of fateshelled/sycl_points maps onto the MI355X facade.
#pragma once
#include "../amd/io.hpp"
