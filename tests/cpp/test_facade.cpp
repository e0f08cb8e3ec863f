// C++ tests of the header facade (include/sycl_points/...), written against the reference's API and modelled on its
// own gtest files: cpp/tests/test_kdtree.cpp, test_downsampling_filters.cpp, test_preprocess_filter.cpp and the
// KNNBase-injection idiom of test_registration_pipeline.cpp:16-61. The CPU oracle (oracle/, test infrastructure) is the
// checker for registration. Runs on a GPU box; exit code 0 = all checks passed.
#include <array>
#include <cstring>
#include <cstdio>
#include <random>

#include "sycl_points/algorithms/common/transform.hpp"
#include "sycl_points/algorithms/feature/covariance.hpp"
#include "sycl_points/algorithms/filter/preprocess_filter.hpp"
#include "sycl_points/algorithms/filter/voxel_downsampling.hpp"
#include "sycl_points/algorithms/knn/bruteforce.hpp"
#include "sycl_points/algorithms/knn/grid.hpp"
#include "sycl_points/algorithms/knn/kdtree.hpp"
#include "sycl_points/algorithms/registration/registration_pipeline.hpp"
#include "sycl_points/algorithms/mapping/voxel_hash_map.hpp"
#include "sycl_points/io/point_cloud_reader.hpp"
#include "sycl_points/io/point_cloud_writer.hpp"
#include <fstream>

// ---- oracle (liboracle.so) entry points used as the checker
struct orc_reg_params {
    int reg_type, robust_type, optimization_method, max_iterations;
    float max_correspondence_distance, robust_default_scale, gn_lambda;
    float lm_init_lambda, lm_lambda_factor, lm_min_lambda, lm_max_lambda;
    int lm_max_inner_iterations;
    float crit_translation, crit_rotation;
    int auto_scale, auto_scaling_iter;
    float init_scale, min_scale;
    float dl_initial_radius, dl_min_radius, dl_max_radius, dl_eta1, dl_eta2, dl_gamma_decrease, dl_gamma_increase;  // 0: defaults
    int rot_enable; float rot_weight, rot_robust_default_scale;                  // rotation constraint (0: off)
    int dr_type; float dr_rot_threshold, dr_trans_threshold, dr_base_factor;    // degenerate regularisation (0: off)
    int mp_active; float mp_omega[36]; float mp_T_pred_inv[16];                  // MAP prior from orc_map_prior_update
};
struct orc_reg_result {
    float T[16]; float H[36]; float b[6]; float error; uint32_t inlier; int iterations; int converged;
    float H_raw[36]; float b_raw[6]; float error_raw;
};
extern "C" {
void orc_knn_bruteforce(const float* q, size_t nq, const float* t, size_t nt, size_t k, int32_t* idx, float* d2);
void orc_cov_estimate(const float* pts, size_t n, const int32_t* idx, size_t k, float* covs);
void orc_cov_estimate_robust(const float* pts, size_t n, const int32_t* idx, size_t k, int robust_type, float mad_scale,
                             float min_robust_scale, size_t max_iter, float* covs);
void orc_registration_align(const orc_reg_params* P, const float* src, const float* src_cov, size_t ns, const float* tgt,
                            const float* tgt_cov, const float* tgt_nrm, size_t nt, const float* init_T16, int nn_mode,
                            orc_reg_result* out, float* trace_T, int* trace_n, const void* prebuilt_nodes,
                            size_t prebuilt_n_nodes, float* trace_steps, int trace_steps_capacity, int* trace_steps_n);
void orc_se3_exp(const float* twist6, float* T16);
int orc_map_prior_update(const float* sig4, const float* H_raw36, float error_raw, uint32_t inlier, const float* T_prev16,
                         const float* T_pred16, float* omega36, float* T_pred_inv16);
}

using namespace sycl_points;
namespace alg = sycl_points::algorithms;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        ++g_checks;                                                                      \
        if (!(cond)) { ++g_failed; std::printf("  CHECK FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)
#define RUN(fn) do { std::printf("[ RUN  ] %s\n", #fn); const int before = g_failed; fn(); std::printf("[ %s ] %s\n", g_failed == before ? " OK " : "FAIL", #fn); } while (0)

static sycl_utils::DeviceQueue* Q = nullptr;

static void random_points(std::mt19937& gen, PointCloudCPU& c, size_t n, float range) {  // test_kdtree.cpp:69-75
    std::uniform_real_distribution<float> dist(-range, range);
    c.points->resize(n);
    for (size_t i = 0; i < n; ++i) { const float x = dist(gen), y = dist(gen), z = dist(gen); (*c.points)[i] = PointType(x, y, z, 1.0f); }
}

static void kdtree_grid_vs_bruteforce() {  // test_kdtree.cpp:301-317, 392-408
    std::mt19937 gen(1234);
    PointCloudCPU tc, qc;
    random_points(gen, tc, 1000, 10.0f);
    random_points(gen, qc, 100, 10.0f);
    PointCloudShared target(*Q, tc), query(*Q, qc);
    auto tree = alg::knn::KDTree::build(*Q, target);
    auto grid = alg::knn::GridKNN::build(*Q, target, 2.0f);
    for (size_t k : {1, 3, 5, 10, 20}) {
        auto kd = tree->knn_search(query, k);
        auto gr = grid->knn_search(query, k);
        auto bf = alg::knn::knn_search_bruteforce(*Q, query, target, k);
        CHECK(kd.query_size == 100 && kd.k == k);
        bool same = true;
        for (size_t i = 0; i < 100 * k; ++i)
            same = same && (*kd.indices)[i] == (*bf.indices)[i] && (*kd.distances)[i] == (*bf.distances)[i] &&
                   (*gr.indices)[i] == (*bf.indices)[i] && (*gr.distances)[i] == (*bf.distances)[i];
        CHECK(same);
    }
    {   // radius search: the grid against the KD-tree kernel (kdtree.hpp:574-719)
        alg::knn::KNNResult rk, rg;
        tree->radius_search_async(query, 10, 3.0f, rk).wait_and_throw();
        grid->radius_search_async(query, 10, 3.0f, rg).wait_and_throw();
        bool same = rk.indices->size() == rg.indices->size();
        size_t cut = 0;
        for (size_t i = 0; same && i < rk.indices->size(); ++i) {
            same = (*rk.indices)[i] == (*rg.indices)[i] && (*rk.distances)[i] == (*rg.distances)[i];
            cut += (*rg.indices)[i] < 0;
        }
        CHECK(same && cut > 0);
    }
    {   // lazy delete on both structures (test_kdtree.cpp:459-512): same neighbours of the kept points afterwards
        shared_vector<uint8_t> flags(1000, uint8_t(1), *Q);
        shared_vector<int32_t> new_idx(1000, int32_t(-1), *Q);
        int32_t next = 0;
        for (size_t i = 0; i < 1000; ++i) {
            if (i % 10 == 0) flags[i] = 0;
            else new_idx[i] = next++;
        }
        auto tree2 = alg::knn::KDTree::build(*Q, target);
        auto grid2 = alg::knn::GridKNN::build(*Q, target, 2.0f);
        tree2->remove_nodes_by_flags(flags, new_idx);
        grid2->remove_nodes_by_flags(flags, new_idx);
        CHECK(grid2->size() == 900);
        auto kd = tree2->knn_search(query, 5);
        auto gr = grid2->knn_search(query, 5);
        bool same = true;
        for (size_t i = 0; i < 100 * 5; ++i)
            same = same && (*kd.indices)[i] == (*gr.indices)[i] && (*kd.distances)[i] == (*gr.distances)[i];
        CHECK(same);
    }
    {   // nodes leave a tree that only has its device-built hierarchy, then k > 32 is asked for — the reference-topology
        // tree is built from the original points and replays the lazy delete: the kept points' neighbours, as a tree with the
        // reference order from the start gives them
        std::mt19937 g2(5);
        PointCloudCPU big;
        random_points(g2, big, 4000, 10.0f);
        PointCloudShared bigs(*Q, big);
        shared_vector<uint8_t> flags(4000, uint8_t(1), *Q);
        shared_vector<int32_t> new_idx(4000, int32_t(-1), *Q);
        int32_t next = 0;
        for (size_t i = 0; i < 4000; ++i) {
            if (i % 4 == 1) flags[i] = 0;
            else new_idx[i] = next++;
        }
        auto lazy = alg::knn::KDTree::build(*Q, bigs);
        auto ref = alg::knn::KDTree::build(*Q, bigs);
        ref->set_reference_tie_order(true);
        (void)ref->knn_search(query, 40);  // its reference-topology tree exists BEFORE the removal
        lazy->remove_nodes_by_flags(flags, new_idx);
        ref->remove_nodes_by_flags(flags, new_idx);
        auto a = lazy->knn_search(query, 40);
        auto b = ref->knn_search(query, 40);
        bool same = a.indices->size() == b.indices->size() && a.indices->size() == 100 * 40;
        for (size_t i = 0; same && i < a.indices->size(); ++i)
            same = (*a.indices)[i] == (*b.indices)[i] && (*a.distances)[i] == (*b.distances)[i];
        CHECK(same);
        bool no_removed = true;
        for (size_t i = 0; i < a.indices->size(); ++i) no_removed = no_removed && (*a.indices)[i] >= 0 && (*a.indices)[i] < next;
        CHECK(no_removed);
    }
    // SinglePoint (test_kdtree.cpp:358-389)
    PointCloudCPU one, q1;
    one.points->push_back(PointType(0, 0, 0, 1));
    q1.points->push_back(PointType(1, 1, 1, 1));
    PointCloudShared t1(*Q, one), qq(*Q, q1);
    auto r = alg::knn::KDTree::build(*Q, t1)->knn_search(qq, 1);
    CHECK((*r.indices)[0] == 0 && std::fabs((*r.distances)[0] - 3.0f) < 1e-6f);
    bool thrown = false;
    try { tree->knn_search(query, 101); } catch (const std::runtime_error&) { thrown = true; }  // kdtree.hpp:221-223
    CHECK(thrown);
}

static void voxelgrid_known_answer() {  // test_downsampling_filters.cpp:27-88
    PointCloudCPU c;
    const float xs[5] = {0.10f, 0.40f, 1.10f, 1.40f, 0.20f};
    const float rgb[5][3] = {{10, 20, 30}, {20, 40, 60}, {30, 60, 90}, {50, 70, 90}, {70, 80, 90}};
    const float inten[5] = {1, 3, 5, 7, 100}, ts[5] = {0, 2, 4, 6, 8};
    for (int i = 0; i < 5; ++i) {
        c.points->push_back(PointType(xs[i], 0, 0, 1));
        c.rgb->push_back(RGBType(rgb[i][0], rgb[i][1], rgb[i][2], 1.0f));
        c.intensities->push_back(inten[i]);
        c.timestamp_offsets->push_back(ts[i]);
    }
    PointCloudShared cloud(*Q, c), result(*Q);
    alg::filter::VoxelGrid vg(*Q, 1.0f);
    vg.set_min_voxel_count(2);
    vg.downsampling(cloud, result);
    CHECK(result.size() == 2);
    CHECK(result.has_rgb() && result.has_intensity() && result.has_timestamps());
    CHECK(std::fabs((*result.points)[0].x() - 0.233333f) < 1e-5f);
    CHECK(std::fabs((*result.points)[1].x() - 1.25f) < 1e-5f);
    CHECK(std::fabs((*result.intensities)[0] - 3.0f) < 1e-5f);
    CHECK(std::fabs((*result.timestamp_offsets)[0] - 3.333333f) < 1e-5f);
    CHECK(std::fabs((*result.rgb)[0].x() - 33.333333f) < 1e-4f && std::fabs((*result.rgb)[0].y() - 46.666667f) < 1e-4f);
    vg.downsampling(cloud, cloud);  // in place (voxel_downsampling.hpp:189-190)
    CHECK(cloud.size() == 2);
    bool thrown = false;
    try { alg::filter::VoxelGrid bad(*Q, 0.0f); } catch (const std::invalid_argument&) { thrown = true; }
    CHECK(thrown);
}

static void preprocess_filter() {  // test_preprocess_filter.cpp:29-99
    PointCloudCPU c;
    c.points->push_back(PointType(0.5f, 0, 0, 1));
    c.points->push_back(PointType(2.0f, 0, 0, 1));
    c.points->push_back(PointType(0, 0, 4.0f, 1));
    c.points->push_back(PointType(std::numeric_limits<float>::quiet_NaN(), 1.0f, 0, 1));
    for (int i = 0; i < 4; ++i) c.intensities->push_back(float(i + 1));
    PointCloudShared cloud(*Q, c);
    alg::filter::PreprocessFilter f(*Q);
    f.box_filter(cloud, 1.0f, 3.0f);
    CHECK(cloud.size() == 1 && cloud.has_intensity());
    CHECK((*cloud.points)[0].x() == 2.0f && (*cloud.intensities)[0] == 2.0f);
    PointCloudCPU d;
    for (int i = 0; i < 5; ++i) { d.points->push_back(PointType(float(i), 0, 0, 1)); d.intensities->push_back(float(i)); }
    PointCloudShared a(*Q, d), b(*Q, d);
    alg::filter::PreprocessFilter fa(*Q), fb(*Q);
    fa.set_random_seed(42);
    fb.set_random_seed(42);
    fa.random_sampling(a, 2);
    fb.random_sampling(b, 2);
    CHECK(a.size() == 2 && b.size() == 2);
    CHECK((*a.points)[0].x() == (*b.points)[0].x() && (*a.points)[1].x() == (*b.points)[1].x());
    PointCloudShared e(*Q, d);
    fa.random_sampling(e, 10);  // no-op when the request covers the input
    CHECK(e.size() == 5);
}

// The box filter reads a fresh scan out of its pinned host copy (no upload): the source container stays what it was, may be
// written right after the call, filtered again (host copy newer again) and used by a kernel that needs the device copy.
static void box_filter_reads_the_host_copy_in_place() {
    std::mt19937 gen(77);
    PointCloudCPU c;
    random_points(gen, c, 70001, 12.0f);  // 35 tiles, a ragged last one
    PointCloudShared scan(*Q, c), kept(*Q), again(*Q);
    alg::filter::PreprocessFilter f(*Q);
    auto expect = [&](const PointCloudShared& in, float mn, float mx) {
        std::vector<PointType> out;
        for (const PointType& p : in.points->host()) {
            const float l = std::max(std::fabs(p.x()), std::max(std::fabs(p.y()), std::fabs(p.z())));
            if (std::isfinite(p.x()) && std::isfinite(p.y()) && std::isfinite(p.z()) && std::isfinite(p.w()) && !(l < mn) && !(l > mx)) out.push_back(p);
        }
        return out;
    };
    auto same = [](const PointCloudShared& got, const std::vector<PointType>& want) {
        if (got.size() != want.size()) return false;
        const auto& h = got.points->host();
        for (size_t i = 0; i < want.size(); ++i)
            if (std::memcmp(&h[i], &want[i], sizeof(PointType)) != 0) return false;
        return true;
    };
    const std::vector<PointType> want1 = expect(scan, 2.0f, 9.0f);
    f.box_filter(scan, kept, 2.0f, 9.0f);
    CHECK(scan.size() == 70001 && same(kept, want1));
    for (size_t i = 0; i < 70001; i += 3) (*scan.points)[i] = PointType(3.0f, float(i % 7), -1.0f, 1.0f);  // the host copy changes
    const std::vector<PointType> want2 = expect(scan, 2.0f, 9.0f);
    f.box_filter(scan, again, 2.0f, 9.0f);
    CHECK(same(again, want2) && want2.size() != want1.size());
    CHECK(same(kept, want1));  // (the first result is its own container)
    // a kernel that needs the device copy of the unfiltered scan: it is uploaded now
    alg::transform::transform(scan, TransformMatrix::Identity());
    const PointType third(3.0f, 3.0f, -1.0f, 1.0f);
    CHECK(scan.size() == 70001 && std::memcmp(&scan.points->host()[3], &third, sizeof(PointType)) == 0);
    f.box_filter(scan, 2.0f, 9.0f);  // in place, from the device copy this time
    CHECK(same(scan, want2));
}

// A host KNNBase injected through the operator boundary, as the reference's DummyKNN (test_registration_pipeline.cpp:16-61).
class HostBruteForceKNN : public alg::knn::KNNBase {
public:
    HostBruteForceKNN(const sycl_utils::DeviceQueue& q, const PointCloudShared& t) : queue(q), target(t) {}
    sycl_utils::events knn_search_async(const PointCloudShared& queries, const size_t k, alg::knn::KNNResult& result,
                                        const std::vector<sycl_utils::event>& = {},
                                        const TransformMatrix& T = TransformMatrix::Identity()) const override {
        PointCloudShared moved = alg::transform::transform_copy(queries, T);
        if (result.indices == nullptr) result.allocate(queue, queries.size(), k); else result.resize(queries.size(), k);
        orc_knn_bruteforce(reinterpret_cast<const float*>(moved.points->data()), moved.size(),
                           reinterpret_cast<const float*>(target.points->data()), target.size(), k, result.indices->data(),
                           result.distances->data());
        ++calls;
        return sycl_utils::events();
    }
    sycl_utils::DeviceQueue queue;
    const PointCloudShared& target;
    mutable int calls = 0;
};

static void point_cloud_extend_erase() {  // points/point_cloud.hpp:307-368, 393-474
    auto make = [&](size_t n, float x0, bool with_ts, double t0) {
        PointCloudCPU c;
        for (size_t i = 0; i < n; ++i) {
            c.points->emplace_back(x0 + float(i), 0.0f, 0.0f, 1.0f);
            c.covs->push_back(Covariance::Identity() * (x0 + float(i)));
            c.intensities->push_back(x0 + float(i));
            if (with_ts) c.timestamp_offsets->push_back(float(i));
        }
        c.start_time_ms = with_ts ? t0 : 0.0;
        c.end_time_ms = with_ts ? t0 + double(n - 1) : 0.0;
        return c;
    };
    PointCloudShared a(*Q, make(5, 0.0f, true, 100.0)), b(*Q, make(3, 10.0f, true, 50.0));
    (void)a.points_device();  // the mirrors exist: extend / erase must keep host and device views consistent
    a.extend(b);
    CHECK(a.size() == 8 && a.has_cov() && a.has_intensity() && !a.has_normal() && !a.has_rgb());
    CHECK((*a.points)[5].x() == 10.0f && (*a.points)[7].x() == 12.0f && (*a.covs)[6](0, 0) == 11.0f && (*a.intensities)[7] == 12.0f);
    CHECK(!a.has_timestamps());  // the reference drops the timestamps of two merged, timestamped clouds (see merge_timestamp_offsets)
    PointCloudShared e(*Q);
    e += b;                       // an empty cloud adopts them
    CHECK(e.size() == 3 && e.has_timestamps() && e.start_time_ms == 50.0 && e.end_time_ms == 52.0 && (*e.timestamp_offsets)[2] == 2.0f);
    CHECK(!e.has_cov());          // attributes survive only when BOTH clouds have them (the empty cloud has none)
    e.erase(0, 1);
    CHECK(e.size() == 2 && (*e.points)[0].x() == 11.0f && e.has_timestamps() && e.end_time_ms == 52.0);
    e.erase(1, 2);
    CHECK(e.size() == 1 && e.end_time_ms == 51.0);
    a.erase(2, 6);
    CHECK(a.size() == 4 && (*a.points)[1].x() == 1.0f && (*a.points)[2].x() == 11.0f && a.has_cov() && (*a.covs)[3](1, 1) == 12.0f);
    // the device view follows: the L-infinity box filter on the device sees the four remaining points
    alg::filter::PreprocessFilter pf(*Q);
    pf.box_filter(a, 0.5f, 11.5f);
    CHECK(a.size() == 2 && (*a.points)[0].x() == 1.0f && (*a.points)[1].x() == 11.0f);
    a.clear();
    CHECK(a.size() == 0 && a.start_time_ms == 0.0 && a.end_time_ms == 0.0);
}

static void point_cloud_files() {  // io/point_cloud_reader.hpp: PLY (the bundled clouds) and PCD, ascii and binary
    const char* dir = std::getenv("SP_GOLDEN_DIR");
    if (dir) {
        const PointCloudCPU src = PointCloudReader::readFile(std::string(dir) + "/source.ply");
        CHECK(src.size() == 69792);  // cpp/data/source.ply
    }
    const float pts[3][4] = {{1.5f, -2.25f, 3.0f, 0.5f}, {0.0f, 1e-3f, -7.5f, 10.0f}, {100.0f, 200.0f, -300.0f, 255.0f}};
    {
        std::ofstream o("/tmp/sp_test_ascii.pcd");
        o << "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 3\nHEIGHT 1\n"
             "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 3\nDATA ascii\n";
        for (auto& p : pts) o << p[0] << " " << p[1] << " " << p[2] << " " << p[3] << "\n";
    }
    {
        std::ofstream o("/tmp/sp_test_binary.pcd", std::ios::binary);
        o << "VERSION 0.7\nFIELDS x y z rgb intensity\nSIZE 4 4 4 4 4\nTYPE F F F U F\nCOUNT 1 1 1 1 1\nWIDTH 3\nHEIGHT 1\nPOINTS 3\nDATA binary\n";
        for (auto& p : pts) {
            const uint32_t rgb = 0x00ff8800u;
            o.write(reinterpret_cast<const char*>(p), 12);
            o.write(reinterpret_cast<const char*>(&rgb), 4);
            o.write(reinterpret_cast<const char*>(&p[3]), 4);
        }
    }
    for (const char* name : {"/tmp/sp_test_ascii.pcd", "/tmp/sp_test_binary.pcd"}) {
        const PointCloudCPU c = PointCloudReader::readFile(name, false, true);
        CHECK(c.size() == 3 && c.intensities->size() == 3);
        bool same = c.size() == 3;
        for (size_t i = 0; same && i < 3; ++i)
            same = (*c.points)[i].x() == pts[i][0] && (*c.points)[i].y() == pts[i][1] && (*c.points)[i].z() == pts[i][2] &&
                   (*c.points)[i].w() == 1.0f && (*c.intensities)[i] == pts[i][3];
        CHECK(same);
    }
    bool threw = false;
    try { PointCloudReader::readFile("/tmp/sp_test_ascii.xyz"); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
}

static float max_abs_diff(const TransformMatrix& A, const float* colmajor16) {
    float m = 0;
    for (int i = 0; i < 16; ++i) m = std::max(m, std::fabs(A.data()[i] - colmajor16[i]));
    return m;
}

static void registration_matches_oracle() {
    const size_t n = 20000;
    std::mt19937 gen(1234);
    PointCloudCPU tc;
    random_points(gen, tc, n, 10.0f * std::cbrt(float(n) / 1e6f));
    const float twist[6] = {0.01f, -0.02f, 0.015f, 0.03f, -0.02f, 0.01f};
    TransformMatrix T_gt;
    orc_se3_exp(twist, T_gt.data());
    PointCloudShared target(*Q, tc);
    Eigen::Isometry3f iso(T_gt);
    PointCloudShared source = alg::transform::transform_copy(target, iso.inverse().matrix());
    std::mt19937 ngen(4321);
    std::normal_distribution<float> noise(0.0f, 0.005f);
    for (size_t i = 0; i < n; ++i) { auto& p = (*source.points)[i]; p.x() += noise(ngen); p.y() += noise(ngen); p.z() += noise(ngen); }

    auto tgrid20 = alg::knn::GridKNN::build(*Q, target, 8.0f);
    auto sgrid20 = alg::knn::GridKNN::build(*Q, source, 8.0f);
    alg::covariance::estimate_async(*tgrid20, target, 20).wait_and_throw();   // fused self-kNN + covariance
    alg::covariance::estimate_async(*sgrid20, source, 20).wait_and_throw();
    CHECK(target.has_cov() && source.has_cov());
    {   // the fused covariances equal the two-step path (KD-tree search, then estimate) bit for bit
        auto tree = alg::knn::KDTree::build(*Q, target);
        PointCloudShared t2(target);
        alg::covariance::estimate_async(tree->knn_search(t2, 20), t2).wait_and_throw();
        size_t diff = 0;
        for (size_t i = 0; i < n; ++i) diff += !((*t2.covs)[i] == (*target.covs)[i]);
        CHECK(diff <= n / 1000);  // only exact-distance ties may order differently
        // M-estimated covariances through the same seam (covariance.hpp:323-411), against the oracle bit for bit
        const auto nb = tree->knn_search(t2, 20);
        alg::covariance::estimate_robust_async(nb, t2, alg::robust::RobustLossType::CAUCHY, 1.0f, 0.1f, 2).wait_and_throw();
        std::vector<float> ref(16 * n);
        orc_cov_estimate_robust(reinterpret_cast<const float*>(t2.points->data()), n, nb.indices->data(), 20, 3, 1.0f, 0.1f,
                                2, ref.data());
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += std::memcmp((*t2.covs)[i].data(), ref.data() + 16 * i, 64) != 0;
        CHECK(bad == 0);
    }
    alg::registration::RegistrationParams p;
    p.max_iterations = 12;
    p.criteria.translation = 0.0f;
    p.criteria.rotation = 0.0f;
    orc_reg_params op{3, 0, 0, 12, 2.0f, 10.0f, 1.0f, 1.0f, 2.0f, 1e-6f, 1e3f, 10, 0.0f, 0.0f, 0, 4, 10.0f, 0.5f};
    orc_reg_result ref;
    const TransformMatrix I = TransformMatrix::Identity();
    orc_registration_align(&op, reinterpret_cast<const float*>(source.points->data()),
                           reinterpret_cast<const float*>(source.covs->data()), n,
                           reinterpret_cast<const float*>(target.points->data()),
                           reinterpret_cast<const float*>(target.covs->data()), nullptr, n, I.data(), 0, &ref, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
    auto tree = alg::knn::KDTree::build(*Q, target);
    auto grid = alg::knn::GridKNN::build(*Q, target);
    HostBruteForceKNN host_knn(*Q, target);
    alg::registration::Registration reg(*Q, p);
    const auto r_tree_grid = reg.align(source, target, *tree);     // a KDTree caller is served by an internal GridKNN
    CHECK(max_abs_diff(r_tree_grid.T.matrix(), ref.T) < 1e-5f && r_tree_grid.inlier == ref.inlier);
    reg.set_accelerate_kdtree(false);
    const auto r_tree = reg.align(source, target, *tree);          // generic path: KNNBase search + K11
    const auto r_grid = reg.align(source, target, *grid);          // fused path (GridKNN recognised)
    const auto r_host = reg.align(source, target, host_knn);       // injected host KNN through the same seam
    CHECK(max_abs_diff(r_tree.T.matrix(), ref.T) < 1e-5f);
    CHECK(max_abs_diff(r_grid.T.matrix(), ref.T) < 1e-5f);
    CHECK(max_abs_diff(r_host.T.matrix(), ref.T) < 1e-5f);
    CHECK(max_abs_diff(r_grid.T.matrix(), T_gt.data()) < 5e-4f);
    CHECK(r_tree.inlier == ref.inlier && r_grid.inlier == ref.inlier && host_knn.calls == 12);
    // the GridKNN + GICP + Gauss-Newton route runs the whole loop on the device (sp_gicp_align_fused); it must stop where
    // the reference's loop stops (criteria != 0) and agree with the host-driven loop (verbose forces that one)
    {
        alg::registration::RegistrationParams pc = p;
        pc.max_iterations = 30;
        pc.criteria.translation = 1e-4f;
        pc.criteria.rotation = 1e-4f;
        orc_reg_params opc = op;
        opc.max_iterations = 30; opc.crit_translation = 1e-4f; opc.crit_rotation = 1e-4f;
        orc_reg_result refc;
        orc_registration_align(&opc, reinterpret_cast<const float*>(source.points->data()),
                               reinterpret_cast<const float*>(source.covs->data()), n,
                               reinterpret_cast<const float*>(target.points->data()),
                               reinterpret_cast<const float*>(target.covs->data()), nullptr, n, I.data(), 0, &refc, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
        alg::registration::Registration regc(*Q, pc);
        const auto rc = regc.align(source, target, *grid);
        CHECK(rc.converged && refc.converged);
        CHECK((int)rc.iterations == refc.iterations);
        CHECK(max_abs_diff(rc.T.matrix(), refc.T) < 1e-5f);
        CHECK(rc.inlier == refc.inlier);
        regc.set_source_presorted(true);  // only changes which lane handles which point
        const auto rp = regc.align(source, target, *grid);
        CHECK(max_abs_diff(rp.T.matrix(), refc.T) < 1e-5f && rp.iterations == rc.iterations);
        alg::registration::RegistrationParams pv = pc;
        pv.verbose = true;
        alg::registration::Registration regv(*Q, pv);
        std::streambuf* old = std::cout.rdbuf(nullptr);  // silence the per-iteration lines
        const auto rv = regv.align(source, target, *grid);
        std::cout.rdbuf(old);
        CHECK(max_abs_diff(rv.T.matrix(), rc.T.matrix().data()) < 2e-6f && rv.iterations == rc.iterations);
    }
    // compute_error_frozen after align() (registration.hpp:350-359): the reference evaluates the neighbours its last
    // linearisation left behind. After the device-resident loop they live in the prepared source's correspondence cache,
    // frozen at the pose before the last update; the generic path keeps them in neighbors_. Both must give the same error
    // at the same trial pose (to the rounding of the two factor formulations) and the same inlier count.
    {
        alg::registration::Registration rf(*Q, p), rg(*Q, p);
        rg.set_accelerate_kdtree(false);
        const auto af = rf.align(source, target, *grid);   // device loop, prepared path
        const auto ag = rg.align(source, target, *tree);   // host loop, KNNBase + K11
        CHECK(max_abs_diff(af.T.matrix(), ag.T.matrix().data()) < 2e-6f);
        TransformMatrix trial = af.T.matrix();
        trial(0, 3) += 0.01f;
        const auto [ef, nf] = rf.compute_error_frozen(source, target, trial);
        const auto [eg, ng] = rg.compute_error_frozen(source, target, trial);
        CHECK(nf == ng && nf == af.inlier);
        CHECK(std::fabs(ef - eg) <= 2e-5f * std::fabs(eg));
        CHECK(ef > af.error);  // off the optimum
        alg::registration::Registration fresh(*Q, p);
        bool threw = false;
        try { (void)fresh.compute_error_frozen(source, target, trial); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);  // nothing linearised yet: a clear error, not a read of stale neighbours
    }
    // Powell dogleg (registration.hpp:897-965) against the oracle's restatement
    {
        alg::registration::RegistrationParams pd = p;
        pd.optimization_method = alg::registration::OptimizationMethod::POWELL_DOGLEG;
        pd.dogleg.initial_trust_region_radius = 0.05f;  // small enough for the Cauchy / dogleg branches to be taken
        orc_reg_params opd = op;
        opd.optimization_method = 2;
        opd.dl_initial_radius = 0.05f; opd.dl_min_radius = 1e-4f; opd.dl_max_radius = 10.0f; opd.dl_eta1 = 0.25f;
        opd.dl_eta2 = 0.75f; opd.dl_gamma_decrease = 0.25f; opd.dl_gamma_increase = 2.0f;
        orc_reg_result refd;
        orc_registration_align(&opd, reinterpret_cast<const float*>(source.points->data()),
                               reinterpret_cast<const float*>(source.covs->data()), n,
                               reinterpret_cast<const float*>(target.points->data()),
                               reinterpret_cast<const float*>(target.covs->data()), nullptr, n, I.data(), 0, &refd, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
        alg::registration::Registration regd(*Q, pd);
        const auto rd = regd.align(source, target, *tree);
        CHECK(max_abs_diff(rd.T.matrix(), refd.T) < 1e-5f);
        CHECK(rd.inlier == refd.inlier && (int)rd.iterations == refd.iterations);
        CHECK(max_abs_diff(rd.T.matrix(), T_gt.data()) < 5e-4f);
    }
    // default-off terms together (rotation constraint, NL-Reg, MAP prior) with LM, against the oracle's restatement
    {
        alg::registration::RegistrationParams p0 = p;
        p0.max_iterations = 3;
        alg::registration::Registration reg0(*Q, p0);
        const auto prev = reg0.align(source, target, *tree);
        CHECK(prev.error_raw < std::numeric_limits<float>::max() && prev.H_raw(0, 0) > 0.0f);
        const float dtw[6] = {0.002f, -0.001f, 0.001f, 0.01f, 0.0f, -0.005f};
        TransformMatrix E;
        orc_se3_exp(dtw, E.data());
        Eigen::Isometry3f T_pred = prev.T * Eigen::Isometry3f(E);
        alg::registration::RegistrationParams pt = p;
        pt.optimization_method = alg::registration::OptimizationMethod::LEVENBERG_MARQUARDT;
        pt.max_iterations = 8;
        pt.rotation_constraint.enable = true;
        pt.rotation_constraint.weight = 3.0f;
        pt.degenerate_reg.type = alg::registration::DegenerateRegularizationType_from_string("NL-REG");
        pt.degenerate_reg.rot_eigenvalue_threshold = 1e9f;  // every direction penalised: the selection cannot flip
        pt.degenerate_reg.trans_eigenvalue_threshold = 1e9f;
        pt.degenerate_reg.base_factor = 0.25f;
        pt.map_prior.enabled = true;
        alg::registration::Registration regt(*Q, pt);
        regt.set_map_prior_state(prev, T_pred);
        const TransformMatrix Tp = T_pred.matrix();
        const auto rt = regt.align(source, target, *tree, Tp);
        orc_reg_params ot = op;
        ot.optimization_method = 1; ot.max_iterations = 8;
        ot.rot_enable = 1; ot.rot_weight = 3.0f; ot.rot_robust_default_scale = 10.0f;
        ot.dr_type = 1; ot.dr_rot_threshold = 1e9f; ot.dr_trans_threshold = 1e9f; ot.dr_base_factor = 0.25f;
        const float sig[4] = {1.0f, 1.0f, 3.16e-2f, 1e-2f};
        float Hraw[36];  // the oracle takes column-major; H_raw is symmetric
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) Hraw[j * 6 + i] = prev.H_raw(i, j);
        const TransformMatrix Tprev = prev.T.matrix();
        ot.mp_active = orc_map_prior_update(sig, Hraw, prev.error_raw, prev.inlier, Tprev.data(), Tp.data(), ot.mp_omega,
                                            ot.mp_T_pred_inv);
        CHECK(ot.mp_active == 1);
        orc_reg_result reft;
        orc_registration_align(&ot, reinterpret_cast<const float*>(source.points->data()),
                               reinterpret_cast<const float*>(source.covs->data()), n,
                               reinterpret_cast<const float*>(target.points->data()),
                               reinterpret_cast<const float*>(target.covs->data()), nullptr, n, Tp.data(), 0, &reft, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
        CHECK(max_abs_diff(rt.T.matrix(), reft.T) < 1e-5f);
        CHECK(rt.inlier == reft.inlier && (int)rt.iterations == reft.iterations);
        CHECK(rt.H(0, 0) - rt.H_raw(0, 0) > 0.9f * 0.25f * float(rt.inlier));  // penalty (+ prior) in H, not in H_raw
        CHECK(std::fabs(rt.H_raw(0, 0) - reft.H_raw[0]) <= 5e-5f * std::fabs(reft.H_raw[0]));
        // without source covariances the rotation constraint is refused (registration.hpp:174-186)
        PointCloudShared bare(*Q);
        bare.resize_points(n);
        for (size_t i = 0; i < n; ++i) (*bare.points)[i] = (*source.points)[i];
        bool threw = false;
        try { regt.align(bare, target, *tree); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw);
    }
    // LM + Geman-McClure through the annealing pipeline (example_registration.cpp:31-45), against the oracle's restatement
    alg::registration::RegistrationPipelineParams pp;
    pp.registration.max_iterations = 10;
    pp.registration.optimization_method = alg::registration::OptimizationMethod::LEVENBERG_MARQUARDT;
    pp.registration.robust.type = alg::robust::RobustLossType::GEMAN_MCCLURE;
    pp.registration.criteria.translation = 0.0f;
    pp.registration.criteria.rotation = 0.0f;
    pp.robust.auto_scale = true; pp.robust.init_scale = 10.0f; pp.robust.min_scale = 2.5f; pp.robust.auto_scaling_iter = 3;
    pp.random_sampling.enable = false;
    alg::registration::RegistrationPipeline pipe(*Q, pp);
    const auto r_pipe = pipe.align(source, target, *grid);
    orc_reg_params op2{3, 4, 1, 10, 2.0f, 10.0f, 1.0f, 1.0f, 2.0f, 1e-6f, 1e3f, 10, 0.0f, 0.0f, 1, 3, 10.0f, 2.5f};
    orc_registration_align(&op2, reinterpret_cast<const float*>(source.points->data()),
                           reinterpret_cast<const float*>(source.covs->data()), n,
                           reinterpret_cast<const float*>(target.points->data()),
                           reinterpret_cast<const float*>(target.covs->data()), nullptr, n, I.data(), 0, &ref, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
    CHECK(max_abs_diff(r_pipe.T.matrix(), ref.T) < 1e-5f);
    // validate_params (registration.hpp:144-150)
    PointCloudShared nocov(*Q, tc);
    bool thrown = false;
    try { reg.align(nocov, target, *tree); } catch (const std::runtime_error&) { thrown = true; }
    CHECK(thrown);
    // random sampling default (num = 1000, mt19937(1234)) wires through
    alg::registration::RegistrationPipeline sampled(*Q);
    const auto r_sampled = sampled.align(source, target, *tree);
    CHECK(sampled.get_registration_input_point_cloud()->size() == 1000);
    // the pipeline's accessors (registration_pipeline.hpp:64-97): inlier ratio over the registration input, the deskewed cloud
    // (= the registration input: no velocity-update stage here), robust weights of that input at a pose
    CHECK(sampled.get_deskewed_point_cloud().get() == sampled.get_registration_input_point_cloud());
    CHECK(std::fabs(sampled.get_inlier_ratio(r_sampled) - float(r_sampled.inlier) / 1000.0f) < 1e-7f && r_sampled.inlier > 0);
    shared_vector<float> wts(*Q);
    sampled.compute_icp_robust_weights(target, *tree, r_sampled.T.matrix(), 10.0f, wts);
    CHECK(wts.size() == 1000);
    size_t ones = 0;
    for (size_t i = 0; i < wts.size(); ++i) ones += std::as_const(wts)[i] == 1.0f;  // robust NONE: weight 1 per inlier
    CHECK(ones == r_sampled.inlier);
    alg::registration::RegistrationPipelineParams pq;
    pq.registration = p;
    pq.random_sampling.enable = false;
    alg::registration::RegistrationPipeline from_backend(std::make_shared<alg::registration::Registration>(*Q, p), pq);
    const auto r_backend = from_backend.align(source, target, *grid);
    CHECK(max_abs_diff(r_backend.T.matrix(), r_grid.T.matrix().data()) == 0.0f);  // constructed around a given backend
}

// algorithms/mapping/voxel_hash_map.hpp through the facade, on the reference's own cases
// (cpp/tests/test_voxel_hash_map.cpp:98-147, 149-193, 315-344, 380-419, 455-502)
static void voxel_hash_map_known_answers() {
    auto make = [&](std::initializer_list<std::array<float, 3>> pts) {
        PointCloudCPU c;
        for (const auto& p : pts) c.points->emplace_back(p[0], p[1], p[2], 1.0f);
        return c;
    };
    bool threw = false;
    try { alg::mapping::VoxelHashMap bad(*Q, 0.0f); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    {
        alg::mapping::VoxelHashMap map(*Q, 0.1f);
        const PointCloudShared cloud(*Q, make({{0.02f, 0.02f, 0.0f}, {0.03f, 0.04f, 0.0f}, {0.11f, 0.02f, 0.0f}, {0.12f, 0.03f, 0.0f}}));
        map.add_point_cloud(cloud, Eigen::Isometry3f::Identity());
        PointCloudShared result(*Q);
        map.downsampling(result, Eigen::Vector3f(0.f, 0.f, 0.f));
        CHECK(result.size() == 2);
        if (result.size() == 2) {
            std::vector<PointType> v{(*result.points)[0], (*result.points)[1]};
            if (v[0].x() > v[1].x()) std::swap(v[0], v[1]);
            CHECK(std::fabs(v[0].x() - 0.025f) < 1e-5f && std::fabs(v[0].y() - 0.03f) < 1e-5f && v[0].w() == 1.0f);
            CHECK(std::fabs(v[1].x() - 0.115f) < 1e-5f && std::fabs(v[1].y() - 0.025f) < 1e-5f);
        }
        CHECK(!result.has_cov() && !result.has_rgb() && !result.has_intensity());
    }
    {
        alg::mapping::VoxelHashMap map(*Q, 0.5f);
        PointCloudCPU c = make({{0.f, 0.f, 0.f}, {0.1f, 0.f, 0.f}});
        c.rgb->emplace_back(0.2f, 0.4f, 0.6f, 1.0f);
        c.rgb->emplace_back(0.6f, 0.2f, 0.0f, 1.0f);
        *c.intensities = {10.0f, 20.0f};
        map.add_point_cloud(PointCloudShared(*Q, c), Eigen::Isometry3f::Identity());
        PointCloudShared result(*Q);
        map.downsampling(result, Eigen::Vector3f(0.f, 0.f, 0.f));
        CHECK(result.size() == 1 && result.has_rgb() && result.has_intensity());
        if (result.size() == 1 && result.has_rgb() && result.has_intensity()) {
            const RGBType col = (*result.rgb)[0];
            CHECK(std::fabs(col.x() - 0.4f) < 1e-5f && std::fabs(col.y() - 0.3f) < 1e-5f && std::fabs(col.z() - 0.3f) < 1e-5f &&
                  std::fabs(col.w() - 1.0f) < 1e-5f);
            CHECK(std::fabs((*result.intensities)[0] - 15.0f) < 1e-5f && std::fabs((*result.points)[0].x() - 0.05f) < 1e-5f);
        }
    }
    {
        alg::mapping::VoxelHashMap map(*Q, 0.2f);
        map.set_min_num_point(2);
        CHECK(map.get_min_num_point() == 2 && map.get_voxel_size() == 0.2f && map.get_max_staleness() == 100);
        map.add_point_cloud(PointCloudShared(*Q, make({{0.01f, 0.01f, 0.f}, {0.02f, 0.01f, 0.f}, {0.30f, 0.30f, 0.f}})),
                            Eigen::Isometry3f::Identity());
        PointCloudShared result(*Q);
        map.downsampling(result, Eigen::Vector3f(0.f, 0.f, 0.f));
        CHECK(result.size() == 1 && std::fabs((*result.points)[0].x() - 0.015f) < 1e-5f);
    }
    {
        alg::mapping::VoxelHashMap map(*Q, 0.5f);
        const PointCloudShared map_cloud(*Q, make({{0.1f, 0.1f, 0.0f}, {1.1f, 0.0f, 0.0f}}));
        map.add_point_cloud(map_cloud, Eigen::Isometry3f::Identity());
        const PointCloudShared query(*Q, make({{-0.9f, 0.1f, 0.0f}, {0.1f, 0.0f, 0.0f}, {1.0f, 0.0f, 0.0f}}));
        Eigen::Isometry3f pose = Eigen::Isometry3f::Identity();
        pose.matrix()(0, 3) = 1.0f;
        CHECK(std::fabs(map.compute_overlap_ratio(query, pose) - 2.0f / 3.0f) < 1e-5f);
        map.set_min_num_point(2);
        CHECK(std::fabs(map.compute_overlap_ratio(query, pose)) < 1e-5f);
        map.add_point_cloud(map_cloud, Eigen::Isometry3f::Identity());
        CHECK(std::fabs(map.compute_overlap_ratio(query, pose) - 2.0f / 3.0f) < 1e-5f);
    }
    {
        alg::mapping::VoxelHashMap map(*Q, 1.0f);
        map.set_rehash_threshold(0.0f);
        map.add_point_cloud(PointCloudShared(*Q, make({{0.5f, 0.5f, 0.5f}, {10.5f, 0.5f, 0.5f}, {20.5f, 0.5f, 0.5f}})),
                            Eigen::Isometry3f::Identity());
        map.add_point_cloud(PointCloudShared(*Q, make({{30.5f, 0.5f, 0.5f}, {40.5f, 0.5f, 0.5f}})), Eigen::Isometry3f::Identity());
        PointCloudShared result(*Q);
        map.downsampling(result, Eigen::Vector3f(0.f, 0.f, 0.f));
        CHECK(result.size() == 5);
        // written from shared memory and read back (io/point_cloud_writer.hpp on a PointCloudShared, test_file_io.cpp:107-128)
        PointCloudWriter::writeFile("/tmp/sp_vhm_result.pcd", result, true);
        const PointCloudCPU back = PointCloudReader::readFile("/tmp/sp_vhm_result.pcd");
        bool same = back.size() == result.size();
        for (size_t i = 0; same && i < back.size(); ++i)
            same = (*back.points)[i].x() == (*result.points)[i].x() && (*back.points)[i].y() == (*result.points)[i].y() &&
                   (*back.points)[i].z() == (*result.points)[i].z();
        CHECK(same);
        map.clear();
        map.downsampling(result, Eigen::Vector3f(0.f, 0.f, 0.f));
        CHECK(result.size() == 0);
    }
}

// Registration::align with the source sharded over a communicator (here a world of one rank: the all-reduce is the identity,
// so the sharded loop — fan-in row + sp_allreduce_rows per iteration — must reproduce the single-GPU result bit for bit)
static void sharded_align_one_rank() {
    const size_t n = 30000;
    std::mt19937 gen(99);
    std::uniform_real_distribution<float> U(-3.0f, 3.0f);
    std::normal_distribution<float> N(0.0f, 0.004f);
    PointCloudCPU tc, sc;
    float twist[6] = {0.01f, -0.02f, 0.015f, 0.03f, -0.02f, 0.01f}, Tgt[16];
    sp_se3_exp_host(twist, Tgt);
    Eigen::Isometry3f Tg;
    for (int i = 0; i < 16; ++i) Tg.matrix().data()[i] = Tgt[i];
    const Eigen::Isometry3f Tinv = Tg.inverse();
    for (size_t i = 0; i < n; ++i) {
        const PointType p(U(gen), U(gen), U(gen), 1.0f);
        tc.points->push_back(p);
        const auto& M = Tinv.matrix();
        sc.points->emplace_back(M(0, 0) * p.x() + M(0, 1) * p.y() + M(0, 2) * p.z() + M(0, 3) + N(gen),
                                M(1, 0) * p.x() + M(1, 1) * p.y() + M(1, 2) * p.z() + M(1, 3) + N(gen),
                                M(2, 0) * p.x() + M(2, 1) * p.y() + M(2, 2) * p.z() + M(2, 3) + N(gen), 1.0f);
    }
    PointCloudShared target(*Q, tc), source(*Q, sc);
    auto tgrid = alg::knn::GridKNN::build(*Q, target);
    auto sgrid = alg::knn::GridKNN::build(*Q, source);
    alg::covariance::estimate_async(*tgrid, target, 10).wait_and_throw();
    alg::covariance::estimate_async(*sgrid, source, 10).wait_and_throw();
    alg::registration::RegistrationParams params;
    params.max_iterations = 15;
    alg::registration::Registration single(*Q, params), sharded(*Q, params);
    const auto a = single.align(source, target, *tgrid);
    sycl_utils::Communicator comm(sycl_utils::Communicator::unique_id(), 0, 1);
    CHECK(comm.rank() == 0 && comm.world() == 1);
    sharded.set_communicator(comm.handle());
    const auto b = sharded.align(source, target, *tgrid);
    CHECK(a.converged && b.converged && a.iterations == b.iterations && a.inlier == b.inlier);
    bool same = true;
    for (int i = 0; i < 16; ++i) same = same && a.T.matrix().data()[i] == b.T.matrix().data()[i];
    CHECK(same);
    CHECK(max_abs_diff(a.T.matrix(), Tgt) < 2e-3f);
    {   // the direct exchange with a world of one rank: the same loop, rows through the (own) slot buffer
        sycl_utils::Exchange x(0, 1);
        x.connect(x.handle_bytes());
        CHECK(x.rank() == 0 && x.world() == 1);
        alg::registration::Registration direct(*Q, params);
        direct.set_exchange(x.handle());
        const auto c = direct.align(source, target, *tgrid);
        CHECK(c.converged && c.iterations == a.iterations && c.inlier == a.inlier);
        CHECK(max_abs_diff(a.T.matrix(), c.T.matrix().data()) < 2e-6f);
        const auto c2 = direct.align(source, target, *tgrid);  // a second alignment: the tags advance
        bool same2 = true;
        for (int i = 0; i < 16; ++i) same2 = same2 && c.T.matrix().data()[i] == c2.T.matrix().data()[i];
        CHECK(same2);
    }
    // a configuration that would leave the ranks with different results is refused, not run rank-locally
    params.optimization_method = alg::registration::OptimizationMethod::LEVENBERG_MARQUARDT;
    alg::registration::Registration lm(*Q, params);
    lm.set_communicator(comm.handle());
    bool threw = false;
    try { lm.align(source, target, *tgrid); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
}

// ---- the reference's pipeline seam tests, restated on the facade (cpp/tests/test_registration_pipeline.cpp): an injected
// RegistrationAligner lambda records what RegistrationPipeline / RobustAligner hand it.
namespace {
class DummyKNN : public alg::knn::KNNBase {  // test_registration_pipeline.cpp:16-23
public:
    sycl_utils::events knn_search_async(const PointCloudShared&, const size_t, alg::knn::KNNResult&,
                                        const std::vector<sycl_utils::event>& = std::vector<sycl_utils::event>(),
                                        const TransformMatrix& = TransformMatrix::Identity()) const override {
        return sycl_utils::events();
    }
};
PointCloudShared make_cloud(size_t size) {  // test_registration_pipeline.cpp:63-78
    PointCloudShared cloud(*Q);
    cloud.points->resize(size);
    cloud.intensities->resize(size);
    cloud.timestamp_offsets->resize(size);
    for (size_t i = 0; i < size; ++i) {
        cloud.points->data()[i] = PointType(static_cast<float>(i), static_cast<float>(i + 1), static_cast<float>(i + 2), 1.0f);
        cloud.intensities->data()[i] = static_cast<float>(i);
        cloud.timestamp_offsets->data()[i] = static_cast<float>(i) * 0.1f;
    }
    cloud.start_time_ms = 1.0;
    cloud.end_time_ms = 2.0;
    return cloud;
}
}  // namespace

static void pipeline_random_sampling_seam() {  // test_registration_pipeline.cpp:106-193, 195-...
    namespace reg = alg::registration;
    for (const auto& c : {std::array<size_t, 4>{1, 3, 6, 3}, std::array<size_t, 4>{0, 2, 5, 5}, std::array<size_t, 4>{1, 8, 5, 5}}) {
        // {enable, num, source size, expected registration input size}: limits / can be disabled / does not shrink a small cloud
        reg::RegistrationPipelineParams params;
        params.random_sampling.enable = c[0] != 0;
        params.random_sampling.num = c[1];
        size_t aligned = 0;
        bool has_i = false, has_t = false;
        auto aligner = [&](const PointCloudShared& source, const PointCloudShared&, const alg::knn::KNNBase&, const TransformMatrix&,
                           const reg::Registration::ExecutionOptions&) {
            aligned = source.size();
            has_i = source.has_intensity();
            has_t = source.has_timestamps();
            reg::RegistrationResult result;
            result.inlier = static_cast<uint32_t>(source.size());
            return result;
        };
        reg::RegistrationPipeline pipeline(aligner, params);
        const auto source = make_cloud(c[2]);
        const auto target = make_cloud(4);
        DummyKNN knn;
        const auto result = pipeline.align(source, target, knn);
        CHECK(result.inlier == c[3]);
        CHECK(aligned == c[3]);
        CHECK(pipeline.get_registration_input_point_cloud() != nullptr);
        CHECK(pipeline.get_registration_input_point_cloud()->size() == c[3]);
        CHECK(has_i && has_t);  // the sampled cloud keeps its attributes (:131-137)
        CHECK(pipeline.get_registration_input_point_cloud()->has_intensity());
        CHECK(pipeline.get_registration_input_point_cloud()->has_timestamps());
        CHECK(pipeline.get_deskewed_point_cloud().get() == pipeline.get_registration_input_point_cloud());  // :195-210
        CHECK(std::fabs(pipeline.get_inlier_ratio(result) - 1.0f) < 1e-6f);
    }
}

static void pipeline_robust_annealing_seam() {  // test_registration_pipeline.cpp:360-409
    namespace reg = alg::registration;
    reg::RegistrationPipelineParams params;
    params.registration.robust.type = alg::robust::RobustLossType::HUBER;
    params.registration.robust.default_scale = 8.0f;
    std::vector<float> fixed_scales;
    auto fixed_aligner = [&](const PointCloudShared&, const PointCloudShared&, const alg::knn::KNNBase&, const TransformMatrix&,
                             const reg::Registration::ExecutionOptions& options) {
        fixed_scales.push_back(options.robust_scale);
        return reg::RegistrationResult{};
    };
    reg::RegistrationPipeline fixed_pipeline(fixed_aligner, params);
    DummyKNN knn;
    fixed_pipeline.align(make_cloud(3), make_cloud(3), knn);
    CHECK(fixed_scales.size() == 1);
    CHECK(fixed_scales.size() == 1 && fixed_scales.front() == -1.0f);  // auto scaling off: the scale stays unset

    params.robust.auto_scale = true;
    params.robust.init_scale = 6.0f;
    params.robust.min_scale = 2.0f;
    params.robust.rotation_init_scale = 9.0f;
    params.robust.rotation_min_scale = 3.0f;
    params.robust.auto_scaling_iter = 3;
    std::vector<float> scales, rot_scales;
    auto annealed_aligner = [&](const PointCloudShared&, const PointCloudShared&, const alg::knn::KNNBase&, const TransformMatrix&,
                                const reg::Registration::ExecutionOptions& options) {
        scales.push_back(options.robust_scale);
        rot_scales.push_back(options.rotation_robust_scale);
        return reg::RegistrationResult{};
    };
    reg::RegistrationPipeline annealed_pipeline(annealed_aligner, params);
    annealed_pipeline.align(make_cloud(3), make_cloud(3), knn);
    CHECK(scales.size() == 3 && rot_scales.size() == 3);
    if (scales.size() == 3 && rot_scales.size() == 3) {
        CHECK(scales[0] == 6.0f);
        CHECK(std::fabs(scales[1] - std::sqrt(12.0f)) < 1e-5f);
        CHECK(std::fabs(scales[2] - 2.0f) < 1e-5f);
        CHECK(rot_scales[0] == 9.0f);
        CHECK(std::fabs(rot_scales[1] - std::sqrt(27.0f)) < 1e-5f);
        CHECK(std::fabs(rot_scales[2] - 3.0f) < 1e-5f);
    }
    // the same schedule through RobustAligner directly, and a fixed scale from the caller switches the annealing off (:52-55)
    reg::pipeline::RobustAligner ra(annealed_aligner, params);
    scales.clear();
    rot_scales.clear();
    reg::Registration::ExecutionOptions fixed;
    fixed.robust_scale = 4.0f;
    ra.align(make_cloud(3), make_cloud(3), knn, TransformMatrix::Identity(), fixed);
    CHECK(scales.size() == 1 && scales[0] == 4.0f);
}

static void kdtree_backend_on_the_bundled_scan() {
    // The reference's raw LiDAR scan (cpp/data/target.ply, 69 088 points on surfaces): its density varies by orders of magnitude,
    // the fullest cell of a 6-points-per-cell grid holds far more than 48 points, so KDTree answers from the device-built hierarchy
    // for every k <= 32 — the decision tests/test_gpu_facade.py holds the Python mirror to. A few hundred points: the host tree.
    const char* dir = std::getenv("SP_GOLDEN_DIR");
    if (!dir) { std::printf("  (SP_GOLDEN_DIR not set: skipped)\n"); return; }
    const auto scan = PointCloudReader::readFile(std::string(dir) + "/target.ply", *Q);
    CHECK(scan.size() == 69088);
    auto tree = alg::knn::KDTree::build(*Q, scan);
    using Backend = alg::knn::KDTree::Backend;
    for (size_t k : {1, 10, 20, 32}) CHECK(tree->backend_for(scan, k) == Backend::Hierarchy);
    CHECK(tree->backend_for(scan, 33) == Backend::HostTree);
    std::mt19937 gen(5);
    PointCloudCPU c;
    random_points(gen, c, 800, 10.0f);
    PointCloudShared small(*Q, c);
    CHECK(alg::knn::KDTree::build(*Q, small)->backend_for(small, 10) == Backend::HostTree);
    // a cloud of a few thousand points (the example's downsampled scans): exact brute force, no hierarchy is built; the lists
    // are the brute-force search's own (same tie rule as the hierarchy); a transform or a removed node ends the shortcut
    PointCloudCPU cm;
    random_points(gen, cm, 6000, 10.0f);
    PointCloudShared mid(*Q, cm);
    auto mt = alg::knn::KDTree::build(*Q, mid);
    CHECK(mt->backend_for(mid, 10) == Backend::BruteForce && mt->backend_for(mid, 20) == Backend::BruteForce);
    CHECK(mt->backend_for(mid, 24) == Backend::Hierarchy);  // fewer than k chunks of 256 targets
    TransformMatrix Tm = TransformMatrix::Identity();
    Tm(0, 3) = 0.5f;
    CHECK(mt->backend_for(mid, 10, Tm) == Backend::Hierarchy);
    const auto r_bf = mt->knn_search(mid, 10);
    const auto r_ref = alg::knn::knn_search_bruteforce(*Q, mid, mid, 10);
    bool same = true;
    for (size_t i = 0; i < 6000 * 10; ++i)
        same = same && (*r_bf.indices)[i] == (*r_ref.indices)[i] && (*r_bf.distances)[i] == (*r_ref.distances)[i];
    CHECK(same);
}

static void raw_scans_full_resolution_alignment() {
    // the reference's bundled scans as they are — 69 792 vs 69 088 points, no sampling, no voxel filter,
    // density varying by orders of magnitude (a cell of the in-loop grid holds up to 5000 points near the sensor) — through
    // Registration::align with the caller's KDTree, against ONE oracle alignment on the same inputs: pose within 1e-5, same
    // inliers. (How fast each search structure is on this cloud: profiles/r05_h_raw_scan_full_resolution_alignment.txt.)
    const char* dir = std::getenv("SP_GOLDEN_DIR");
    if (!dir) { std::printf("  (SP_GOLDEN_DIR not set: skipped)\n"); return; }
    auto source = PointCloudReader::readFile(std::string(dir) + "/source.ply", *Q);
    auto target = PointCloudReader::readFile(std::string(dir) + "/target.ply", *Q);
    CHECK(source.size() == 69792 && target.size() == 69088);
    auto source_tree = alg::knn::KDTree::build(*Q, source);
    auto target_tree = alg::knn::KDTree::build(*Q, target);
    alg::covariance::estimate_async(source_tree->knn_search(source, 20), source).wait_and_throw();
    alg::covariance::estimate_async(target_tree->knn_search(target, 20), target).wait_and_throw();
    alg::registration::RegistrationParams p;  // the reference's defaults: GICP, Gauss-Newton, max_corr 2.0, criteria 1e-3
    orc_reg_params op{3, 0, 0, 20, 2.0f, 10.0f, 1.0f, 1.0f, 2.0f, 1e-6f, 1e3f, 10, 1e-3f, 1e-3f, 0, 4, 10.0f, 0.5f};
    orc_reg_result ref;
    const TransformMatrix I = TransformMatrix::Identity();
    orc_registration_align(&op, reinterpret_cast<const float*>(source.points->data()), reinterpret_cast<const float*>(source.covs->data()),
                           source.size(), reinterpret_cast<const float*>(target.points->data()),
                           reinterpret_cast<const float*>(target.covs->data()), nullptr, target.size(), I.data(), 0, &ref, nullptr,
                           nullptr, nullptr, 0, nullptr, 0, nullptr);
    alg::registration::Registration reg(*Q, p);
    const auto r = reg.align(source, target, *target_tree);
    CHECK(max_abs_diff(r.T.matrix(), ref.T) < 1e-5f);
    CHECK(r.inlier == ref.inlier && (int)r.iterations == ref.iterations && r.converged == (ref.converged != 0));
    CHECK(ref.inlier > 60000);
}

static void kdtree_radius_and_lazy_delete_on_the_hierarchy() {
    // KDTree::radius_search_async / remove_nodes_by_flags (kdtree.hpp:574-765) on a cloud large enough for the device-built
    // hierarchy (no host tree is ever built): radius search = brute force's list cut at the radius; after a removal the tree
    // answers like brute force over the kept points under their new indices (test_kdtree.cpp:459-512).
    std::mt19937 gen(17);
    PointCloudCPU c, qc;
    random_points(gen, c, 20000, 5.0f);
    random_points(gen, qc, 500, 5.0f);
    PointCloudShared cloud(*Q, c), queries(*Q, qc);
    auto tree = alg::knn::KDTree::build(*Q, cloud);
    const size_t max_k = 10;
    const float radius = 0.4f;
    alg::knn::KNNResult rr;
    tree->radius_search_async(queries, max_k, radius, rr).wait_and_throw();
    auto bf = alg::knn::knn_search_bruteforce(*Q, queries, cloud, max_k);
    bool same = rr.query_size == queries.size() && rr.k == max_k;
    for (size_t i = 0; same && i < queries.size() * max_k; ++i) {
        const bool inside = (*bf.distances)[i] <= radius * radius;
        same = inside ? ((*rr.indices)[i] == (*bf.indices)[i] && (*rr.distances)[i] == (*bf.distances)[i])
                      : ((*rr.indices)[i] == -1 && (*rr.distances)[i] == std::numeric_limits<float>::max());
    }
    CHECK(same);
    CHECK(tree->backend_for(queries, 10) == alg::knn::KDTree::Backend::Hierarchy);
    // lazy delete: every third point goes
    shared_vector<uint8_t> flags(c.size(), uint8_t(1), *Q);
    shared_vector<int32_t> new_idx(c.size(), -1, *Q);
    PointCloudCPU kept;
    int32_t next = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        if (i % 3 == 0) { flags[i] = 0; continue; }
        new_idx[i] = next++;
        kept.points->push_back((*c.points)[i]);
    }
    tree->remove_nodes_by_flags(flags, new_idx);
    CHECK(!tree->pristine());
    CHECK(tree->backend_for(queries, 10) == alg::knn::KDTree::Backend::Hierarchy);  // still the hierarchy: no host build
    PointCloudShared kept_cloud(*Q, kept);
    auto kd = tree->knn_search(queries, max_k);
    auto bf2 = alg::knn::knn_search_bruteforce(*Q, queries, kept_cloud, max_k);
    same = true;
    for (size_t i = 0; same && i < queries.size() * max_k; ++i)
        same = (*kd.indices)[i] == (*bf2.indices)[i] && (*kd.distances)[i] == (*bf2.distances)[i];
    CHECK(same);
}

static void containers_read_from_another_queue() {
    // A container's device buffer goes back to the buffer cache when the container dies, tagged with events on EVERY live queue's
    // stream (core.hpp DeviceBufferCache): a search enqueued on the tree's queue may still be reading a query cloud that is bound
    // to another queue. Twenty times over: search asynchronously on queue B, drop the query cloud at once, take a new container
    // of the same size on queue A (the cache hands out the buffer just released) and overwrite it — the lists must be those of
    // a search that was waited for.
    sycl_utils::DeviceQueue qa(0), qb(0);
    std::mt19937 gen(5);
    PointCloudCPU tc, qc;
    random_points(gen, tc, 200000, 10.0f);
    random_points(gen, qc, 200000, 10.0f);
    PointCloudShared target(qb, tc);
    auto tree = alg::knn::KDTree::build(qb, target);
    alg::knn::KNNResult ref;
    {
        PointCloudShared queries(qa, qc);
        tree->knn_search_async(queries, 10, ref).wait_and_throw();
    }
    (void)ref.indices->host();
    bool same = true;
    for (int rep = 0; rep < 20 && same; ++rep) {
        alg::knn::KNNResult r;
        {
            PointCloudShared queries(qa, qc);
            (void)queries.points_device();
            tree->knn_search_async(queries, 10, r);  // not waited for
        }  // the query cloud's buffer is released here, the search may still be running on qb
        shared_vector<PointType> scribble(qc.size(), PointType(1e6f, 1e6f, 1e6f, 1.0f), qa);
        (void)scribble.device_data();  // same size: the buffer that was just released, overwritten on qa
        qb.wait();
        const auto& ri = r.indices->host();
        const auto& ei = ref.indices->host();
        for (size_t i = 0; same && i < ei.size(); i += 97) same = ri[i] == ei[i];
    }
    CHECK(same);
}

static void queues_come_and_go() {
    // Round 5's host-side machinery under churn: pooled device buffers tagged with the stream they were last used on (facade
    // cache and library pool), pinned host blocks from a pool, uploads nobody waits for, grid builds that return unsynchronised —
    // and queues that are created and destroyed around them (DeviceQueue's destructor retires its stream in both pools: a tag on a
    // dead stream must never be queried). Twelve times: a new queue, a cloud on it, box filter + voxel grid + k = 10 neighbours +
    // covariances + an alignment of a 1000-point sample against the rest, everything dropped with the queue. Every repetition's
    // results must be the first one's, bit for bit.
    std::mt19937 gen(11);
    PointCloudCPU c;
    random_points(gen, c, 90000, 12.0f);
    std::vector<float> first_pts, first_T;
    size_t first_n = 0;
    bool same = true;
    for (int rep = 0; rep < 12 && same; ++rep) {
        sycl_utils::DeviceQueue q(0);
        PointCloudShared cloud(q, c), boxed(q), ds(q);
        alg::filter::PreprocessFilter pre(q);
        pre.box_filter(cloud, boxed, 0.5f, 11.0f);
        alg::filter::VoxelGrid vg(q, 0.4f);
        vg.downsampling(boxed, ds);
        auto tree = alg::knn::KDTree::build(q, ds);
        const auto nb = tree->knn_search(ds, 10);
        alg::covariance::estimate_async(nb, ds).wait_and_throw();
        PointCloudShared sample(q);
        pre.set_random_seed(7);
        pre.random_sampling(ds, sample, 1000);
        alg::registration::RegistrationParams rp;
        rp.max_iterations = 5;
        auto reg = std::make_shared<alg::registration::Registration>(q, rp);
        TransformMatrix T0 = TransformMatrix::Identity();
        T0(0, 3) = 0.02f;
        const auto res = reg->align(sample, ds, *tree, T0);
        std::vector<float> pts(4 * ds.size()), Tm(16);
        const auto& hp = ds.points->host();
        for (size_t i = 0; i < ds.size(); ++i) { pts[4 * i] = hp[i].x(); pts[4 * i + 1] = hp[i].y(); pts[4 * i + 2] = hp[i].z(); pts[4 * i + 3] = hp[i].w(); }
        for (int i = 0; i < 16; ++i) Tm[i] = res.T.matrix().data()[i];
        if (rep == 0) { first_pts = pts; first_T = Tm; first_n = ds.size(); }
        same = ds.size() == first_n && pts == first_pts && Tm == first_T;
    }
    CHECK(same);
    CHECK(first_n > 5000);
}

static void large_containers_round_trip() {
    // Uploads and downloads of a megabyte or more go through two pinned staging buffers of 8 MB (core.hpp StagedCopy): sizes
    // that are no multiple of the chunk, one chunk exactly, several chunks; device-side modification in between (a transform).
    for (size_t n : {size_t(70000), size_t(524288), size_t(1300001)}) {  // 1.1 MB, 8 MB = one chunk, 20.8 MB
        PointCloudCPU c;
        c.points->resize(n);
        for (size_t i = 0; i < n; ++i) (*c.points)[i] = PointType((float)(i % 1000) * 0.25f, (float)(i % 777), (float)(i / 1000), 1.0f);
        PointCloudShared cloud(*Q, c);
        TransformMatrix T = TransformMatrix::Identity();
        T(0, 3) = 1.0f; T(1, 3) = -2.0f; T(2, 3) = 0.5f;
        alg::transform::transform(cloud, T);  // on the device: the host copy becomes stale and is read back below
        bool ok = cloud.size() == n;
        const auto& h = cloud.points->host();
        for (size_t i = 0; ok && i < n; i += (i < 100 || i + 100 >= n) ? 1 : 997)
            ok = h[i].x() == (float)(i % 1000) * 0.25f + 1.0f && h[i].y() == (float)(i % 777) - 2.0f && h[i].z() == (float)(i / 1000) + 0.5f;
        CHECK(ok);
    }
}

static void large_copies_on_queues_of_two_devices() {
    // The staging buffers and their events are per device, keyed by the STREAM's device: a process with a queue on
    // device 0 and one on device 1 copies more than a megabyte through each, alternately, whatever device is current. On a box
    // with one GPU both queues sit on device 0 (the keying is exercised, not the second device).
    const int second = sp_device_count() > 1 ? 1 : 0;
    sycl_points::sycl_utils::DeviceQueue q0(0), q1(second);
    const size_t n = 200000;  // 3.2 MB of points
    PointCloudCPU c;
    c.points->resize(n);
    for (size_t i = 0; i < n; ++i) (*c.points)[i] = PointType((float)i, (float)(i % 13), 2.0f, 1.0f);
    bool ok = true;
    for (int round = 0; round < 3 && ok; ++round) {
        PointCloudShared a(q0, c), b(q1, c);
        TransformMatrix T = TransformMatrix::Identity();
        T(0, 3) = (float)(round + 1);
        alg::transform::transform(b, T);  // (device 1's queue first: the current device is then not device 0)
        alg::transform::transform(a, T);
        const auto& ha = a.points->host();
        const auto& hb = b.points->host();
        for (size_t i = 0; ok && i < n; i += 991) ok = ha[i].x() == (float)i + (float)(round + 1) && hb[i].x() == ha[i].x();
    }
    CHECK(ok);
    throw_on_error(sp_set_device(0));
}

static void kdtree_self_knn_large_clouds() {
    // KDTree::knn_search on the tree's own cloud: from 32 k points on, a cloud of near-uniform density is answered by the grid's
    // lane-per-query selection, a clustered one (fullest cell over the limit) by the device-built hierarchy; both must give
    // brute force's lists.
    std::mt19937 gen(99);
    for (int clustered = 0; clustered < 2; ++clustered) {
        PointCloudCPU c;
        random_points(gen, c, clustered ? 20000 : 40000, 10.0f);
        if (clustered) {
            std::uniform_real_distribution<float> U(-0.05f, 0.05f);
            for (int i = 0; i < 20000; ++i) c.points->emplace_back(1.0f + U(gen), 2.0f + U(gen), -3.0f + U(gen), 1.0f);
        }
        PointCloudShared cloud(*Q, c);
        auto tree = alg::knn::KDTree::build(*Q, cloud);
        using Backend = alg::knn::KDTree::Backend;
        CHECK(tree->backend_for(cloud, 20) == (clustered ? Backend::Hierarchy : Backend::Grid));
        CHECK(tree->backend_for(cloud, 8) == (clustered ? Backend::Hierarchy : Backend::Grid));
        CHECK(tree->backend_for(cloud, 5) == Backend::Hierarchy);   // below the selection kernel's range
        CHECK(tree->backend_for(cloud, 40) == Backend::HostTree);   // beyond the hierarchy's k
        for (size_t k : {10, 20}) {
            auto kd = tree->knn_search(cloud, k);
            auto bf = alg::knn::knn_search_bruteforce(*Q, cloud, cloud, k);
            bool same = kd.query_size == cloud.size() && kd.k == k;
            for (size_t i = 0; same && i < cloud.size() * k; ++i)
                same = (*kd.indices)[i] == (*bf.indices)[i] && (*kd.distances)[i] == (*bf.distances)[i];
            CHECK(same);
        }
    }
}

int main() {
    sycl_utils::DeviceQueue queue(0);
    Q = &queue;
    queue.print_device_info();
    RUN(kdtree_grid_vs_bruteforce);
    RUN(kdtree_self_knn_large_clouds);
    RUN(containers_read_from_another_queue);
    RUN(queues_come_and_go);
    RUN(large_containers_round_trip);
    RUN(large_copies_on_queues_of_two_devices);
    RUN(kdtree_backend_on_the_bundled_scan);
    RUN(raw_scans_full_resolution_alignment);
    RUN(kdtree_radius_and_lazy_delete_on_the_hierarchy);
    RUN(voxelgrid_known_answer);
    RUN(preprocess_filter);
    RUN(box_filter_reads_the_host_copy_in_place);
    RUN(point_cloud_extend_erase);
    RUN(pipeline_random_sampling_seam);
    RUN(pipeline_robust_annealing_seam);
    RUN(point_cloud_files);
    RUN(voxel_hash_map_known_answers);
    RUN(sharded_align_one_rank);
    RUN(registration_matches_oracle);
    std::printf("%d checks, %d failed\n", g_checks, g_failed);
    return g_failed == 0 ? 0 : 1;
}
