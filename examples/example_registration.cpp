// BASELINE config 1: the reference's cpp/examples/example_registration.cpp flow on the facade, same parameters
// (example_registration.cpp:29-55): box filter [0.5, 50], voxel 0.25, k = 10 covariances + normals, RegistrationPipeline
// with LM + GEMAN_MCCLURE + GICP, 3-level robust-scale annealing 10 -> 2.5, random sampling of 1000 source points.
// usage: example_registration <source.ply> <target.ply> [loops=10] [warmup=2] [--grid]
#include <chrono>
#include <cstdio>
#include <map>

#include "sycl_points/algorithms/feature/covariance.hpp"
#include "sycl_points/algorithms/filter/preprocess_filter.hpp"
#include "sycl_points/algorithms/filter/voxel_downsampling.hpp"
#include "sycl_points/algorithms/knn/grid.hpp"
#include "sycl_points/algorithms/knn/kdtree.hpp"
#include "sycl_points/algorithms/registration/registration_pipeline.hpp"
#include "sycl_points/io/point_cloud_reader.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: %s source.ply target.ply [loops=10] [warmup=2] [--grid]\n", argv[0]); return 2; }
    size_t LOOP = 10, WARM_UP = 2;
    bool use_grid = false;
    int pos = 0;
    for (int a = 3; a < argc; ++a) {
        if (std::string(argv[a]) == "--grid") use_grid = true;
        else if (pos++ == 0) LOOP = std::stoul(argv[a]);
        else WARM_UP = std::stoul(argv[a]);
    }
    const auto source_points = sycl_points::PointCloudReader::readFile(argv[1], false, false);
    const auto target_points = sycl_points::PointCloudReader::readFile(argv[2], false, false);
    sycl_points::sycl_utils::DeviceQueue queue(0);
    queue.print_device_info();
    std::printf("source %zu points, target %zu points\n", source_points.size(), target_points.size());

    const float voxel_size = 0.25f;
    const size_t num_neighbors = 10;
    namespace alg = sycl_points::algorithms;
    alg::registration::RegistrationPipelineParams pp;
    pp.registration.max_iterations = 10;
    pp.registration.max_correspondence_distance = 2.0f;
    pp.registration.optimization_method = alg::registration::OptimizationMethod::LEVENBERG_MARQUARDT;
    pp.registration.robust.type = alg::robust::RobustLossType::GEMAN_MCCLURE;
    pp.registration.robust.default_scale = 10.0f;
    pp.registration.reg_type = alg::registration::RegType::GICP;
    pp.robust.auto_scale = true;
    pp.robust.init_scale = 10.0f;
    pp.robust.min_scale = 2.5f;
    pp.robust.rotation_init_scale = 5.0f;
    pp.robust.rotation_min_scale = 2.5f;
    pp.robust.auto_scaling_iter = 3;
    const auto pipeline = std::make_shared<alg::registration::RegistrationPipeline>(queue, pp);
    const auto voxel_grid = std::make_shared<alg::filter::VoxelGrid>(queue, voxel_size);
    const auto preprocess = std::make_shared<alg::filter::PreprocessFilter>(queue);

    std::map<std::string, double> elapsed, detail;  // detail: parts of the reference's stages, printed beside them
    auto now = [] { return std::chrono::high_resolution_clock::now(); };
    auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    sycl_points::TransformMatrix T_final = sycl_points::TransformMatrix::Identity();
    for (size_t i = 0; i < LOOP + WARM_UP; ++i) {
        auto t0 = now();
        sycl_points::PointCloudShared source(queue, source_points), target(queue, target_points);
        const double dt_shared = us(t0, now());
        t0 = now();
        auto t1 = now();
        preprocess->box_filter(source, 0.5f, 50.0f);
        const double dt_box_s = us(t1, now());
        sycl_points::PointCloudShared source_ds(queue), target_ds(queue);
        t1 = now();
        voxel_grid->downsampling(source, source_ds);
        const double dt_vox_s = us(t1, now());
        t1 = now();
        preprocess->box_filter(target, 0.5f, 50.0f);
        const double dt_box_t = us(t1, now());
        t1 = now();
        voxel_grid->downsampling(target, target_ds);
        const double dt_vox_t = us(t1, now());
        const double dt_down = us(t0, now());
        t0 = now();
        std::shared_ptr<alg::knn::KNNBase> source_knn, target_knn;
        if (use_grid) {
            source_knn = alg::knn::GridKNN::build(queue, source_ds, 4.0f);
            target_knn = alg::knn::GridKNN::build(queue, target_ds, 4.0f);
        } else {
            source_knn = alg::knn::KDTree::build(queue, source_ds);
            target_knn = alg::knn::KDTree::build(queue, target_ds);
        }
        const double dt_build = us(t0, now());
        t0 = now();
        const auto source_neighbors = source_knn->knn_search(source_ds, num_neighbors);
        const auto target_neighbors = target_knn->knn_search(target_ds, num_neighbors);
        const double dt_knn = us(t0, now());
        t0 = now();
        alg::covariance::estimate_async(source_neighbors, source_ds).wait_and_throw();
        alg::covariance::estimate_async(target_neighbors, target_ds).wait_and_throw();
        const double dt_cov = us(t0, now());
        t0 = now();
        alg::covariance::estimate_normals_async(source_neighbors, source_ds).wait_and_throw();
        alg::covariance::estimate_normals_async(target_neighbors, target_ds).wait_and_throw();
        const double dt_normal = us(t0, now());
        t0 = now();
        const auto ret = pipeline->align(source_ds, target_ds, *target_knn, sycl_points::TransformMatrix::Identity());
        const double dt_reg = us(t0, now());
        if (i >= WARM_UP) {
            elapsed["1. to PointCloudShared"] += dt_shared;
            elapsed["2. Downsampling"] += dt_down;
            elapsed["3. KNN structure build"] += dt_build;
            elapsed["4. kNN Search"] += dt_knn;
            elapsed["5. compute Covariances"] += dt_cov;
            elapsed["6. compute Normals"] += dt_normal;
            elapsed["7. Registration"] += dt_reg;
            detail["2a. box filter (source + target)"] += dt_box_s + dt_box_t;
            detail["2b. voxel downsampling (source + target)"] += dt_vox_s + dt_vox_t;
        }
        if (i == LOOP + WARM_UP - 1) {
            T_final = ret.T.matrix();
            std::printf("downsampled: source %zu, target %zu; inliers %u\n", source_ds.size(), target_ds.size(), ret.inlier);
        }
    }
    std::printf("T_target_source =\n");
    for (int r = 0; r < 4; ++r) std::printf("  % .6f % .6f % .6f % .6f\n", T_final(r, 0), T_final(r, 1), T_final(r, 2), T_final(r, 3));
    double total = 0;
    for (auto& [k, v] : elapsed) { std::printf("%28s: %10.2f us\n", k.c_str(), v / LOOP); total += v / LOOP; }
    std::printf("%28s: %10.2f us\n", "TOTAL", total);
    for (auto& [k, v] : detail) std::printf("%44s: %10.2f us\n", k.c_str(), v / LOOP);
    std::printf("RESULT");
    for (int i = 0; i < 16; ++i) std::printf(" %.9g", T_final.data()[i]);
    std::printf("\n");
    return 0;
}
