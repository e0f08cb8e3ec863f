// sycl_points facade for MI355X — registration layer.
//   algorithms/robust/robust.hpp                         : RobustLossType (+ from_string)
//   algorithms/registration/factor.hpp                   : RegType (+ from_string)
//   algorithms/registration/registration_params.hpp      : RegistrationParams and friends
//   algorithms/registration/{linearized_result,result}.hpp
//   algorithms/registration/registration.hpp             : Registration (align, compute_linearized_result,
//                                                          compute_error_frozen, compute_icp_robust_weights)
//   algorithms/registration/pipeline/{aligner,robust}.hpp, registration_pipeline(_params).hpp
// Host control flow follows the reference; every per-point kernel is a C-ABI call. When the KNNBase handed to align()
// is a GridKNN and the factor is GICP, the iteration uses the prepared / fused kernel (sp_gicp_iteration_fused).
#pragma once
#include <cctype>
#include <cstring>
#include <functional>
#include <optional>
#include <vector>
#include <iostream>
#include <tuple>

#include "features.hpp"

namespace sycl_points {
namespace algorithms {

// robust::RobustLossType lives in features.hpp (covariance::estimate_robust_async needs it too)

namespace registration {

enum class RegType { POINT_TO_POINT = 0, POINT_TO_PLANE, POINT_TO_DISTRIBUTION, GICP, GENZ };  // factor.hpp:18-32
inline RegType RegType_from_string(const std::string& str) {
    std::string u = str;
    for (auto& c : u) c = (char)std::toupper((unsigned char)c);
    if (u == "POINT_TO_POINT") return RegType::POINT_TO_POINT;
    if (u == "POINT_TO_PLANE") return RegType::POINT_TO_PLANE;
    if (u == "GICP") return RegType::GICP;
    if (u == "GENZ") return RegType::GENZ;
    if (u == "POINT_TO_DISTRIBUTION" || u == "P2D") return RegType::POINT_TO_DISTRIBUTION;
    throw std::runtime_error("[RegType_from_string] Invalid RegType str '" + str + "'");
}
enum class OptimizationMethod { GAUSS_NEWTON = 0, LEVENBERG_MARQUARDT, POWELL_DOGLEG };
inline OptimizationMethod OptimizationMethod_from_string(const std::string& str) {
    std::string u = str;
    for (auto& c : u) c = (char)std::toupper((unsigned char)c);
    if (u == "GN" || u == "GAUSS_NEWTON") return OptimizationMethod::GAUSS_NEWTON;
    if (u == "LM" || u == "LEVENBERG_MARQUARDT") return OptimizationMethod::LEVENBERG_MARQUARDT;
    if (u == "DOGLEG" || u == "POWELL_DOGLEG") return OptimizationMethod::POWELL_DOGLEG;
    throw std::runtime_error("[OptimizationMethod_from_string] Invalid OptimizationMethod str [" + str + "]");
}

/// registration_params.hpp:41-114 (same fields, same defaults)
struct RegistrationConvergenceCriteria {
    float translation = 1e-3f;
    float rotation = 1e-3f;
};
struct RegistrationFactorParams {
    struct Robust { robust::RobustLossType type = robust::RobustLossType::NONE; float default_scale = 10.0f; };
    struct GenZ { float planarity_threshold = 0.2f; };
    struct RotationConstraint {  // registration_params.hpp:56-64
        struct Robust { float default_scale = 10.0f; };
        bool enable = false;
        float weight = 1.0f;
        Robust robust;
    };
    RegType reg_type = RegType::GICP;
    float max_correspondence_distance = 2.0f;
    Robust robust;
    RotationConstraint rotation_constraint;
    GenZ genz;
    bool verbose = false;
};
/// degenerate_regularization.hpp:14-46
enum class DegenerateRegularizationType { none = 0, nl_reg };
inline DegenerateRegularizationType DegenerateRegularizationType_from_string(const std::string& str) {
    std::string u = str;
    for (auto& c : u) c = (char)std::toupper((unsigned char)c);
    if (u == "NONE") return DegenerateRegularizationType::none;
    if (u == "NL-REG" || u == "NL_REG") return DegenerateRegularizationType::nl_reg;
    throw std::runtime_error("[DegenerateRegularizationType_from_string] Invalid DegenerateRegularizationType str [" + str + "]");
}
struct DegenerateRegularizationParams {
    DegenerateRegularizationType type = DegenerateRegularizationType::none;
    float rot_eigenvalue_threshold = 10.0f;
    float trans_eigenvalue_threshold = 1.0f;
    float base_factor = 1.0f;
};
/// map_prior.hpp:14-35
struct MapPriorParams {
    bool enabled = false;
    float rot_vel_sigma = 1.0f;
    float trans_vel_sigma = 1.0f;
    float rot_base_sigma = 3.16e-2f;
    float trans_base_sigma = 1e-2f;
};
struct RegistrationOptimizationParams {
    struct GaussNewton { float lambda = 1.0f; };
    struct LevenbergMarquardt {
        size_t max_inner_iterations = 10;
        float lambda_factor = 2.0f, init_lambda = 1.0f, max_lambda = 1e3f, min_lambda = 1e-6f;
    };
    struct Dogleg {  // registration_params.hpp:84-92
        float initial_trust_region_radius = 1.0f, min_trust_region_radius = 1e-4f, max_trust_region_radius = 10.0f;
        float eta1 = 0.25f, eta2 = 0.75f, gamma_decrease = 0.25f, gamma_increase = 2.0f;
    };
    GaussNewton gn;
    LevenbergMarquardt lm;
    Dogleg dogleg;
    OptimizationMethod optimization_method = OptimizationMethod::GAUSS_NEWTON;
};
struct RegistrationParams : public RegistrationFactorParams, public RegistrationOptimizationParams {
    using Criteria = RegistrationConvergenceCriteria;
    size_t max_iterations = 20;
    Criteria criteria;
    DegenerateRegularizationParams degenerate_reg;  // registration_params.hpp:111-112
    MapPriorParams map_prior;
};

/// linearized_result.hpp:12-24
struct LinearizedResult {
    Eigen::Matrix<float, 6, 6> H = Eigen::Matrix<float, 6, 6>::Zero();
    Eigen::Matrix<float, 6, 1> b = Eigen::Matrix<float, 6, 1>::Zero();
    float error = std::numeric_limits<float>::max();
    uint32_t inlier = 0;
};
/// result.hpp:12-28
struct RegistrationResult {
    using Ptr = std::shared_ptr<RegistrationResult>;
    Eigen::Isometry3f T = Eigen::Isometry3f::Identity();
    bool converged = false;
    size_t iterations = 0;
    Eigen::Matrix<float, 6, 6> H = Eigen::Matrix<float, 6, 6>::Zero();
    Eigen::Matrix<float, 6, 1> b = Eigen::Matrix<float, 6, 1>::Zero();
    float error = std::numeric_limits<float>::max();
    Eigen::Matrix<float, 6, 6> H_raw = Eigen::Matrix<float, 6, 6>::Zero();
    Eigen::Matrix<float, 6, 1> b_raw = Eigen::Matrix<float, 6, 1>::Zero();
    float error_raw = std::numeric_limits<float>::max();
    uint32_t inlier = 0;
};

/// registration.hpp:88-965
class Registration {
public:
    using Ptr = std::shared_ptr<Registration>;
    struct ExecutionOptions {  // registration.hpp:92-100
        float robust_scale;
        float rotation_robust_scale;
        float dt;
        TransformMatrix prev_pose;
        ExecutionOptions() : robust_scale(-1.0f), rotation_robust_scale(-1.0f), dt(0.1f), prev_pose(TransformMatrix::Identity()) {}
    };

    explicit Registration(const sycl_utils::DeviceQueue& queue, const RegistrationParams& params = RegistrationParams())
        : params_(params), queue_(queue) {
        hip_check(hipMalloc(&lin_dev_, sizeof(sp_linearized)), "hipMalloc");
        ws_bytes_ = sp_gicp_workspace_bytes(0);
        hip_check(hipMalloc(&ws_, ws_bytes_), "hipMalloc");
        hip_check(hipMalloc(&T_dev_, (16 + 8 + 4) * sizeof(float)), "hipMalloc");  // pose | delta[8] | iterations
        // read-backs (one 192-byte system or one error per optimiser step) land in pinned memory: a copy into pageable memory
        // is staged and blocks inside the runtime (2.24 -> 1.94 ms for the LM alignment of the reference's example)
        hip_check(hipHostMalloc(&pin_, 4096), "hipHostMalloc");
        hip_check(hipMalloc(&res_dev_, sizeof(sp_align_result)), "hipMalloc");
        // the device-resident optimiser's pose (64 bytes each way) and result block in host-mapped memory, when it is to be had:
        // the host writes the initial guess, the launch stores the block and then its `done` word, the host spins on that word —
        // no copy in, no copy out, no synchronisation
        void* m = nullptr;
        if (hipHostMalloc(&m, kMappedBytes, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, m, 0) == hipSuccess) { map_host_ = static_cast<char*>(m); map_dev_ = static_cast<char*>(d); }
            else { (void)hipGetLastError(); (void)hipHostFree(m); }
        } else {
            (void)hipGetLastError();
        }
    }
    ~Registration() {
        if (psrc_) sp_gicp_source_destroy(psrc_);
        if (ptgt_) sp_gicp_target_destroy(ptgt_);
        (void)hipFree(lin_dev_); (void)hipFree(ws_); (void)hipFree(T_dev_); (void)hipFree(res_dev_);
        if (pin_) (void)hipHostFree(pin_);
        if (map_host_) (void)hipHostFree(map_host_);
    }
    Registration(const Registration&) = delete;
    Registration& operator=(const Registration&) = delete;

    /// registration.hpp:124-126 / MapPrior::update (map_prior.hpp:97-174): once per frame, after motion prediction and
    /// before align().
    void set_map_prior_state(const RegistrationResult& prev_result, const Eigen::Isometry3f& T_pred) {
        const sp_map_prior_params mp{params_.map_prior.enabled ? 1 : 0, params_.map_prior.rot_vel_sigma,
                                     params_.map_prior.trans_vel_sigma, params_.map_prior.rot_base_sigma,
                                     params_.map_prior.trans_base_sigma};
        float H36[36];
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) H36[i * 6 + j] = prev_result.H_raw(i, j);
        const TransformMatrix Tprev = prev_result.T.matrix(), Tpred = T_pred.matrix();
        throw_on_error(sp_map_prior_update_host(&mp, H36, prev_result.error_raw, prev_result.inlier, Tprev.data(),
                                                Tpred.data(), &map_prior_));
    }

    /// registration.hpp:129-193
    void validate_params(const PointCloudShared& source, const PointCloudShared& target, RegistrationParams& params) const {
        if (params.reg_type == RegType::POINT_TO_PLANE && !target.has_normal()) {
            if (!target.has_cov())
                throw std::runtime_error("[Registration::validate_params] Normal vector or covariance matrices of target "
                                         "must be pre-computed before performing Point-to-Plane ICP matching.");
            covariance::extract_normals(target);
        }
        if (params.reg_type == RegType::GICP && (!source.has_cov() || !target.has_cov()))
            throw std::runtime_error("[Registration::validate_params] Covariance matrices of source and target must be "
                                     "pre-computed before performing GICP matching.");
        if (params.reg_type == RegType::GENZ) {
            if (!target.has_cov())
                throw std::runtime_error("[Registration::validate_params] Covariance matrices of target must be "
                                         "pre-computed before performing GenZ-ICP matching.");
            if (!target.has_normal()) covariance::extract_normals(target);
        }
        if (params.reg_type == RegType::POINT_TO_DISTRIBUTION && !target.has_cov())
            throw std::runtime_error("[Registration::validate_params] Covariance matrices of target must be pre-computed "
                                     "before performing Point-to-Distribution ICP matching.");
        if (params.rotation_constraint.enable) {
            if (!source.has_cov())
                throw std::runtime_error("[Registration::validate_params] Covariance matrices of source are required for "
                                         "performing rotation constraint matching.");
            if (!target.has_cov())
                throw std::runtime_error("[Registration::validate_params] Covariance matrices of target are required for "
                                         "performing rotation constraint matching.");
        }
        if (params.robust.type != robust::RobustLossType::NONE && params.robust.default_scale <= 0.0f) {
            std::cout << "[Caution] `robust.default_scale` must be greater than zero. Disable robust loss." << std::endl;
            params.robust.type = robust::RobustLossType::NONE;
        }
    }

    /// registration.hpp:201-276
    RegistrationResult align(const PointCloudShared& source, const PointCloudShared& target, const knn::KNNBase& target_knn,
                             const TransformMatrix& initial_guess = TransformMatrix::Identity(),
                             const ExecutionOptions& options = ExecutionOptions()) {
        RegistrationResult result;
        result.T.matrix() = initial_guess;
        if (source.size() == 0) return result;
        validate_params(source, target, params_);
        const float robust_scale = options.robust_scale > 0.0f ? options.robust_scale : params_.robust.default_scale;
        rotation_robust_scale_ = options.rotation_robust_scale > 0.0f ? options.rotation_robust_scale
                                                                      : params_.rotation_constraint.robust.default_scale;
        const bool pose_terms = params_.degenerate_reg.type != DegenerateRegularizationType::none || prior_active();
        float lm_lambda = params_.lm.init_lambda;
        float trust_region_radius = params_.dogleg.initial_trust_region_radius;
        const auto* grid = dynamic_cast<const knn::GridKNN*>(&target_knn);
        // A caller that still hands over the reference's KDTree gets the same correspondences (exact nearest neighbours;
        // only the order of exactly equidistant points can differ) from a GridKNN built on the target once per tree.
        if (grid == nullptr && accelerate_kdtree_)
            if (const auto* kd = dynamic_cast<const knn::KDTree*>(&target_knn)) grid = grid_for(*kd, target);
        // every factor type searches on the grid when there is one (4x faster per k = 1 search than the KD-tree kernel);
        // GICP additionally fuses search and linearisation
        const knn::KNNBase& nn = (grid != nullptr && grid->size() == target.size()) ? static_cast<const knn::KNNBase&>(*grid)
                                                                                     : target_knn;
        // (the rotation constraint reads the raw covariances: it runs on the unfused kernels)
        const bool fused = grid != nullptr && grid->size() == target.size() && !params_.rotation_constraint.enable &&
                           (params_.reg_type == RegType::GICP || params_.reg_type == RegType::POINT_TO_DISTRIBUTION);
        fused_loop_active_ = fused;  // the LM / dog-leg trial steps then read the frozen correspondences from the cache
        if (fused) prepare_fused(source, target, *grid, initial_guess);
        // GICP + Gauss-Newton on a GridKNN: the whole loop runs on the device (one launch per iteration, convergence
        // test included), the host reads the result back once — same arithmetic as the loop below.
        // (the host-side pose terms — degenerate regularisation, MAP prior — need the reduced system on the host)
        const bool on_device = fused && !params_.verbose && params_.max_iterations > 0 && !pose_terms;
        const bool sharded = comm_ != nullptr || xchg_ != nullptr;
        // Every optimiser as ONE launch that loops on the device and ONE read-back (sp_gicp_align_optimize): LM and dog-leg always
        // (the reference crosses host <-> device 2 + inner tries times per iteration, registration.hpp:830-964), Gauss-Newton when
        // the source fits one workgroup (the pipeline's default 1000-point sample: no launch per iteration, no counter between
        // steps). std::nullopt: the launch is not available now, or its bounded wait ran out — the loops below take over.
        // A target with crowded cells — a raw scan: thousands of returns in the sensor's own cell — and a source of up to 131072
        // points: that launch with a WAVE per source point (sp_gicp_source_set_wave_per_point = 2), for Gauss-Newton too: a lane
        // alone walks thousands of candidates per query there (the reference's bundled scans at full resolution: 0.21 against
        // 0.69 ms per iteration). The fullest cell is measured once per grid (a kernel and a read-back, cached by the library).
        const bool crowded = on_device && !sharded && crowded_target(*grid, source.size());
        if (on_device && !sharded) throw_on_error(sp_gicp_source_set_wave_per_point(psrc_, crowded ? 2 : 1));
        if (on_device && !sharded &&
            (params_.optimization_method != OptimizationMethod::GAUSS_NEWTON || source.size() <= 1024 || crowded)) {
            const float scales[1] = {robust_scale};
            if (auto r = align_optimize_on_device(initial_guess, scales, 1)) return *r;
            prepare_fused(source, target, *grid, initial_guess);  // (the correspondence cache of the abandoned launch is stale)
        }
        if (on_device && params_.optimization_method == OptimizationMethod::GAUSS_NEWTON)
            return align_on_device(source.size(), initial_guess, robust_scale);
        if (comm_ != nullptr || xchg_ != nullptr)
            throw std::runtime_error("[Registration::align] a communicator is set: only the device-resident Gauss-Newton loop "
                                     "(GICP / POINT_TO_DISTRIBUTION on a GridKNN, no host-side pose terms) is sharded");

        for (size_t iter = 0; iter < params_.max_iterations; ++iter) {
            LinearizedResult lin = fused ? linearize_fused(source.size(), result.T.matrix(), robust_scale)
                                         : linearize_generic(source, target, nn, result.T.matrix(), robust_scale);
            result.H_raw = lin.H; result.b_raw = lin.b; result.error_raw = lin.error;
            if (pose_terms) apply_pose_terms(lin, result.T.matrix(), initial_guess);  // registration.hpp:249-253
            switch (params_.optimization_method) {
                case OptimizationMethod::LEVENBERG_MARQUARDT:
                    optimize_levenberg_marquardt(source, target, result, lin, lm_lambda, iter, robust_scale);
                    break;
                case OptimizationMethod::GAUSS_NEWTON:
                    optimize_gauss_newton(result, lin, iter);
                    break;
                case OptimizationMethod::POWELL_DOGLEG:
                    optimize_powell_dogleg(source, target, result, lin, trust_region_radius, iter, robust_scale);
                    break;
            }
            if (result.converged) break;
        }
        return result;
    }

    /// MI355X extension: pipeline::RobustAligner's annealing loop (pipeline/robust.hpp:100-111) — one align() per robust scale,
    /// each starting from the pose the previous one ended on — with all levels in ONE launch and ONE read-back when align()
    /// would run on the device-resident optimiser (GICP / POINT_TO_DISTRIBUTION on a GridKNN or an accelerated KDTree, no
    /// rotation constraint, no host-side pose terms, not verbose); level by level through align() otherwise. Returns the last
    /// level's result, like the reference's loop.
    RegistrationResult align_levels(const PointCloudShared& source, const PointCloudShared& target, const knn::KNNBase& target_knn,
                                    const TransformMatrix& initial_guess, const ExecutionOptions& options,
                                    const std::vector<float>& robust_scales, const std::vector<float>& rotation_robust_scales) {
        RegistrationResult result;
        result.T.matrix() = initial_guess;
        if (source.size() == 0 || robust_scales.empty()) return result;
        const bool pose_terms = params_.degenerate_reg.type != DegenerateRegularizationType::none || prior_active();
        const bool candidate = robust_scales.size() > 1 && robust_scales.size() <= size_t(SP_OPT_MAX_LEVELS) && !params_.verbose &&
                               params_.max_iterations > 0 && !pose_terms && comm_ == nullptr && xchg_ == nullptr &&
                               !params_.rotation_constraint.enable &&
                               (params_.reg_type == RegType::GICP || params_.reg_type == RegType::POINT_TO_DISTRIBUTION);
        if (candidate) {
            validate_params(source, target, params_);
            const auto* grid = dynamic_cast<const knn::GridKNN*>(&target_knn);
            if (grid == nullptr && accelerate_kdtree_)
                if (const auto* kd = dynamic_cast<const knn::KDTree*>(&target_knn)) grid = grid_for(*kd, target);
            if (grid != nullptr && grid->size() == target.size() && params_.robust.type != robust::RobustLossType::NONE) {
                fused_loop_active_ = true;
                prepare_fused(source, target, *grid, initial_guess);
                const bool crowded = crowded_target(*grid, source.size());  // (see align())
                throw_on_error(sp_gicp_source_set_wave_per_point(psrc_, crowded ? 2 : 1));
                if (auto r = align_optimize_on_device(initial_guess, robust_scales.data(), (int)robust_scales.size())) return *r;
            }
        }
        for (size_t level = 0; level < robust_scales.size(); ++level) {  // pipeline/robust.hpp:100-111
            ExecutionOptions o = options;
            o.robust_scale = robust_scales[level];
            o.rotation_robust_scale = level < rotation_robust_scales.size() ? rotation_robust_scales[level] : options.rotation_robust_scale;
            result = align(source, target, target_knn, result.T.matrix(), o);
        }
        return result;
    }

    /// MI355X extension: with a communicator set, `source` is THIS rank's shard of the source cloud (target, its
    /// covariances and its KNN structure are replicated on every rank) and align() runs the Gauss-Newton loop through
    /// sp_gicp_align_sharded: one launch and one 128-byte RCCL all-reduce per iteration, the identical pose on every rank.
    /// Applies where align() runs entirely on the device (GICP / POINT_TO_DISTRIBUTION on a GridKNN or an accelerated KDTree,
    /// Gauss-Newton, no host-side pose terms); every other configuration throws, since a rank-local result would silently
    /// differ between ranks. nullptr (default): single GPU.
    void set_communicator(sp_comm* comm) { comm_ = comm; }
    /// The same sharded alignment with the rows exchanged directly between the ranks' slot buffers (Exchange, sp_xchg):
    /// sp_gicp_align_direct, no collective launch per iteration. Takes precedence over a communicator. A peer that does
    /// not deliver within the exchange's timeout makes align() throw (SP_ERR_RUNTIME) on every waiting rank.
    void set_exchange(sp_xchg* xchg) { xchg_ = xchg; }

    /// MI355X extension: when align() is given a KDTree (no nodes removed) and the factor is GICP, search on a GridKNN
    /// built from the target instead (default on; results agree to rounding with the KD-tree path).
    void set_accelerate_kdtree(bool v) { accelerate_kdtree_ = v; }

    /// MI355X extension: tell align() that source clouds arrive spatially ordered (GridKNN::order() / voxel-downsampled
    /// clouds), so the prepared path skips its per-alignment sort (SP_SOURCE_PRESORTED).
    void set_source_presorted(bool v) { source_presorted_ = v; }

    /// registration.hpp:312-324: with the pose at the start of the optimisation window the degenerate regularisation
    /// is applied to the result.
    LinearizedResult compute_linearized_result(const PointCloudShared& source, const PointCloudShared& target,
                                               const knn::KNNBase& target_knn, const TransformMatrix& pose,
                                               const TransformMatrix& initial_pose,
                                               const ExecutionOptions& options = ExecutionOptions()) {
        LinearizedResult lin = compute_linearized_result(source, target, target_knn, pose, options);
        regularize(lin, pose, initial_pose);
        return lin;
    }
    /// registration.hpp:326-331 (without degenerate regularisation)
    LinearizedResult compute_linearized_result(const PointCloudShared& source, const PointCloudShared& target,
                                               const knn::KNNBase& target_knn, const TransformMatrix& pose,
                                               const ExecutionOptions& options = ExecutionOptions()) {
        const float s = options.robust_scale > 0.0f ? options.robust_scale : params_.robust.default_scale;
        rotation_robust_scale_ = options.rotation_robust_scale > 0.0f ? options.rotation_robust_scale
                                                                      : params_.rotation_constraint.robust.default_scale;
        return linearize_generic(source, target, target_knn, pose, s);
    }
    /// registration.hpp:350-359
    std::tuple<float, uint32_t> compute_error_frozen(const PointCloudShared& source, const PointCloudShared& target,
                                                     const TransformMatrix& pose,
                                                     const ExecutionOptions& options = ExecutionOptions()) const {
        const float s = options.robust_scale > 0.0f ? options.robust_scale : params_.robust.default_scale;
        rotation_robust_scale_ = options.rotation_robust_scale > 0.0f ? options.rotation_robust_scale
                                                                      : params_.rotation_constraint.robust.default_scale;
        // The reference evaluates the neighbours the last linearisation left in neighbors_ (registration.hpp:350-359). When
        // that linearisation ran on the prepared path (align() on a GridKNN / accelerated KDTree) they live in the
        // correspondence cache of the prepared source, frozen at lin_T_, not in neighbors_.
        fused_loop_active_ = last_lin_fused_;
        if (!last_lin_fused_ && (neighbors_.indices == nullptr || neighbors_.indices->size() != source.size()))
            throw std::runtime_error("[Registration::compute_error_frozen] no correspondences for this source: call align() or "
                                     "compute_linearized_result() first");
        return compute_error(source, target, pose, s);
    }
    /// registration.hpp:279-294
    void compute_icp_robust_weights(const PointCloudShared& source, const PointCloudShared& target,
                                    const knn::KNNBase& target_knn, const TransformMatrix& pose, float robust_scale,
                                    shared_vector<float>& out) const {
        const size_t N = source.size();
        out.assign(N, 0.0f);
        if (N == 0) return;
        target_knn.nearest_neighbor_search_async(source, neighbors_, {}, pose);
        last_lin_fused_ = false;
        const sp_factor_params fp = factor_params(robust_scale);
        throw_on_error(sp_icp_robust_weights(source.points_device(), source.covs_device(), N, target.points_device(),
                                             target.covs_device(), target.normals_device(), neighbors_.indices->device_data(),
                                             neighbors_.distances->device_data(), pose.data(), 0, &fp,
                                             out.device_data_for_write(N), queue_.stream()));
        queue_.wait();
    }
    const RegistrationParams& params() const { return params_; }
    /// MI355X extension: steps the latest device-resident optimiser run executed (linearisations, trial evaluations)
    std::pair<uint32_t, uint32_t> last_optimizer_steps() const { return {last_opt_linearizations_, last_opt_trials_}; }

private:
    sp_factor_params factor_params(float robust_scale) const {
        return sp_factor_params{int(params_.reg_type), int(params_.robust.type), params_.max_correspondence_distance,
                                robust_scale, genz_alpha_, params_.genz.planarity_threshold,
                                params_.rotation_constraint.enable ? 1 : 0, params_.rotation_constraint.weight,
                                rotation_robust_scale_};
    }
    bool prior_active() const { return params_.map_prior.enabled && map_prior_.has_prior != 0; }
    float prior_error(const TransformMatrix& T) const {  // MapPrior::prior_error (map_prior.hpp:197-201)
        return prior_active() ? sp_map_prior_apply_host(&map_prior_, T.data(), nullptr, nullptr, nullptr) : 0.0f;
    }
    void regularize(LinearizedResult& lin, const TransformMatrix& T, const TransformMatrix& T_initial) const {
        if (params_.degenerate_reg.type == DegenerateRegularizationType::none) return;
        const sp_degenerate_reg_params dr{int(params_.degenerate_reg.type), params_.degenerate_reg.rot_eigenvalue_threshold,
                                          params_.degenerate_reg.trans_eigenvalue_threshold,
                                          params_.degenerate_reg.base_factor};
        float H36[36], b6[6];
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) H36[i * 6 + j] = lin.H(i, j);
            b6[i] = lin.b(i);
        }
        throw_on_error(sp_degenerate_regularize_host(&dr, H36, b6, lin.inlier, T.data(), T_initial.data()));
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) lin.H(i, j) = H36[i * 6 + j];
            lin.b(i) = b6[i];
        }
    }
    /// DegenerateRegularization::regularize then MapPrior::apply on the reduced system (registration.hpp:249-253)
    void apply_pose_terms(LinearizedResult& lin, const TransformMatrix& T, const TransformMatrix& T_initial) const {
        regularize(lin, T, T_initial);
        if (!prior_active()) return;
        float H36[36], b6[6];
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) H36[i * 6 + j] = lin.H(i, j);
            b6[i] = lin.b(i);
        }
        sp_map_prior_apply_host(&map_prior_, T.data(), H36, b6, &lin.error);
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) lin.H(i, j) = H36[i * 6 + j];
            lin.b(i) = b6[i];
        }
    }
    static LinearizedResult to_result(const sp_linearized& h) {
        LinearizedResult r;
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) r.H(i, j) = h.H[i * 6 + j];
            r.b(i) = h.b[i];
        }
        r.error = h.error;
        r.inlier = h.inlier;
        return r;
    }
    sp_linearized read_lin() const {  // the reference's wait_and_throw + toCPU(0) (registration.hpp:674-675)
        sp_linearized* const h = static_cast<sp_linearized*>(pin_);
        hip_check(hipMemcpyAsync(h, lin_dev_, sizeof *h, hipMemcpyDeviceToHost, queue_.stream()), "D2H");
        hip_check(hipStreamSynchronize(queue_.stream()), "sync");
        return *h;
    }
    float compute_genz_alpha(const PointCloudShared& target, size_t N) const {  // registration.hpp:464-511
        detail::DeviceScratch cnt(8);
        throw_on_error(sp_genz_counts(target.covs_device(), neighbors_.indices->device_data(),
                                      neighbors_.distances->device_data(), N, params_.max_correspondence_distance,
                                      params_.genz.planarity_threshold, static_cast<uint32_t*>(cnt.p), queue_.stream()));
        uint32_t h[2];
        hip_check(hipMemcpyAsync(h, cnt.p, 8, hipMemcpyDeviceToHost, queue_.stream()), "D2H");
        queue_.wait();
        return h[0] == 0 ? 1.0f : static_cast<float>(h[1]) / static_cast<float>(h[0]);
    }
    LinearizedResult linearize_generic(const PointCloudShared& source, const PointCloudShared& target,
                                       const knn::KNNBase& target_knn, const TransformMatrix& T, float robust_scale) {
        target_knn.nearest_neighbor_search_async(source, neighbors_, {}, T);
        last_lin_fused_ = false;
        if (params_.reg_type == RegType::GENZ) genz_alpha_ = compute_genz_alpha(target, source.size());
        const sp_factor_params fp = factor_params(robust_scale);
        throw_on_error(sp_gicp_linearize(source.points_device(), source.covs_device(), source.size(), target.points_device(),
                                         target.covs_device(), target.normals_device(), neighbors_.indices->device_data(),
                                         neighbors_.distances->device_data(), T.data(), 0, &fp, lin_dev_, ws_, ws_bytes_,
                                         queue_.stream()));
        return to_result(read_lin());
    }
    void prepare_fused(const PointCloudShared& source, const PointCloudShared& target, const knn::GridKNN& grid,
                       const TransformMatrix& T0) {
        // Reuse certificates (one k = 3 self-search of the target) pay when many source points are aligned to the target, or the
        // target is aligned to again (a submap, frame after frame). A target that is new and large beside its source — the
        // reference's example: a fresh 6 k-point target per frame, a 1000-point sample aligned to it — is prepared without them
        // (every linearisation then searches, seeded by the previous winner) and gets them if it comes back.
        const bool small_source = source.size() * 4 < target.size();
        if (ptgt_ == nullptr || ptgt_grid_id_ != grid.id()) {
            if (ptgt_) sp_gicp_target_destroy(ptgt_);
            ptgt_ = nullptr;
            if (small_source)
                throw_on_error(sp_gicp_target_create_plain(grid.handle(), target.covs_device(), target.size(), queue_.stream(), &ptgt_));
            else
                throw_on_error(sp_gicp_target_create(grid.handle(), target.covs_device(), target.size(), queue_.stream(), &ptgt_));
            ptgt_grid_id_ = grid.id();
            ptgt_covs_ = target.covs.get();  // (created with the GICP rows of these covariances)
            ptgt_covs_gen_ = target.covs->generation();
            ptgt_reg_ = int(RegType::GICP);
            ptgt_uses_ = 0;
        }
        ++ptgt_uses_;
        if ((ptgt_uses_ >= 2 || !small_source) && !sp_gicp_target_has_certificates(ptgt_)) {
            throw_on_error(sp_gicp_target_certify(ptgt_, target.covs_device(), queue_.stream()));
            ptgt_covs_ = target.covs.get();  // (the rows were rewritten from these covariances, for the factor they already served)
            ptgt_covs_gen_ = target.covs->generation();
        }
        // rows of the factor asked for: plane(Ct) for GICP, inverse(Ct) for point-to-distribution (factor.hpp:311-317) —
        // once per (target covariances, factor): a caller that aligns frame after frame against the same target does not pay
        // the 30 us per million points again (the container's generation moves whenever somebody may have written to it)
        if (ptgt_covs_ != target.covs.get() || ptgt_covs_gen_ != target.covs->generation() ||
            ptgt_reg_ != int(params_.reg_type)) {
            throw_on_error(sp_gicp_target_prepare(ptgt_, target.covs_device(), int(params_.reg_type), queue_.stream()));
            ptgt_covs_ = target.covs.get();
            ptgt_covs_gen_ = target.covs->generation();
            ptgt_reg_ = int(params_.reg_type);
        }
        if (psrc_ == nullptr || psrc_cap_ < source.size()) {
            if (psrc_) sp_gicp_source_destroy(psrc_);
            psrc_ = nullptr;
            throw_on_error(sp_gicp_source_create(source.size(), &psrc_));
            psrc_cap_ = source.size();
        }
        last_src_points_ = source.points_device();
        last_src_covs_ = source.covs_device();
        throw_on_error(sp_gicp_source_prepare(psrc_, ptgt_, last_src_points_, last_src_covs_, source.size(), T0.data(), 0,
                                              source_order(source.size()), queue_.stream()));
        neighbors_.indices == nullptr ? neighbors_.allocate(queue_, source.size(), 1) : neighbors_.resize(source.size(), 1);
    }
    /// How sp_gicp_source_prepare orders the source: as it is when the caller says it is spatially ordered — or when it is a few
    /// thousand points (the pipeline's random sample): the order only decides which lane handles which point, and the cell sort
    /// is seven launches for something a handful of waves do not notice.
    /// Whether a source of n points is better served by a wave per point against this target (measured: scratch/waveq_crossover.py,
    /// scratch/raw_scan_align.py, scratch/opt_example.py): a fullest cell of 512 points or more (a raw scan) for sources of up to
    /// 131072 points; of 32 or more (a voxel-downsampled scan: surfaces) up to 16384 — the reference example's whole 6 k-point
    /// source against its 6 k-point target: 0.28 against 0.37 ms per alignment. Up to 1024 points the launch takes that form by
    /// itself and the grid is not asked.
    static bool crowded_target(const knn::GridKNN& grid, size_t n) {
        if (n <= 1024 || n > 131072) return false;
        const uint32_t fullest = sp_grid_max_cell_points(grid.handle());
        return fullest >= 512 || (fullest >= 32 && n <= 16384);
    }
    int source_order(size_t n) const { return (source_presorted_ || n <= 4096) ? SP_SOURCE_PRESORTED : SP_SOURCE_SORT; }
    const knn::GridKNN* grid_for(const knn::KDTree& tree, const PointCloudShared& target) {
        if (!tree.pristine() || tree.size() != target.size() || target.size() == 0) return nullptr;
        if (kd_grid_ == nullptr || kd_grid_tree_id_ != tree.id()) {
            if (ptgt_) { sp_gicp_target_destroy(ptgt_); ptgt_ = nullptr; ptgt_grid_id_ = 0; }  // it borrows the old grid
            kd_grid_ = knn::GridKNN::build(queue_, target);
            kd_grid_tree_id_ = tree.id();
        }
        return kd_grid_.get();
    }
    sp_opt_params opt_params() const {
        return sp_opt_params{int(params_.optimization_method) == int(OptimizationMethod::GAUSS_NEWTON)          ? SP_OPT_GAUSS_NEWTON
                             : int(params_.optimization_method) == int(OptimizationMethod::LEVENBERG_MARQUARDT) ? SP_OPT_LEVENBERG_MARQUARDT
                                                                                                                 : SP_OPT_POWELL_DOGLEG,
                             (int)params_.max_iterations, params_.criteria.rotation, params_.criteria.translation, params_.gn.lambda,
                             (int)params_.lm.max_inner_iterations, params_.lm.lambda_factor, params_.lm.init_lambda,
                             params_.lm.max_lambda, params_.lm.min_lambda, params_.dogleg.initial_trust_region_radius,
                             params_.dogleg.min_trust_region_radius, params_.dogleg.max_trust_region_radius, params_.dogleg.eta1,
                             params_.dogleg.eta2, params_.dogleg.gamma_decrease, params_.dogleg.gamma_increase};
    }
    /// The prepared source / target are in place (prepare_fused): the whole optimiser loop — every level of `scales` — as one
    /// launch (sp_gicp_align_optimize) and one read-back. std::nullopt when the launch cannot be resident now or its bounded wait
    /// ran out (sp_align_result::status): the caller runs its per-step loop from the initial guess.
    std::optional<RegistrationResult> align_optimize_on_device(const TransformMatrix& initial_guess, const float* scales, int levels) {
        const sp_factor_params fp = factor_params(scales[0]);
        const sp_opt_params op = opt_params();
        sp_align_result* h;
        if (map_host_) {
            std::memcpy(map_host_, initial_guess.data(), 16 * sizeof(float));
            h = reinterpret_cast<sp_align_result*>(map_host_ + 256);
            volatile uint32_t* const done = &h->pad[0];
            *done = 0u;
            const int rc = sp_gicp_align_optimize(ptgt_, psrc_, reinterpret_cast<float*>(map_dev_), &fp, &op, scales, levels,
                                                  reinterpret_cast<sp_align_result*>(map_dev_ + 256), ws_, ws_bytes_, queue_.stream());
            if (rc == SP_ERR_RUNTIME && std::strstr(sp_last_error(), "not available") != nullptr) return std::nullopt;
            throw_on_error(rc);
            for (unsigned spins = 0; *done != SP_ALIGN_RESULT_DONE; ++spins) {
                if (spins > (1u << 24)) {  // (a fraction of a second: the device is busy elsewhere, or the launch failed — wait for the stream)
                    hip_check(hipStreamSynchronize(queue_.stream()), "sync");
                    break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            hip_check(hipMemcpyAsync(T_dev_, initial_guess.data(), 16 * sizeof(float), hipMemcpyHostToDevice, queue_.stream()), "H2D");
            const int rc = sp_gicp_align_optimize(ptgt_, psrc_, T_dev_, &fp, &op, scales, levels, res_dev_, ws_, ws_bytes_, queue_.stream());
            if (rc == SP_ERR_RUNTIME && std::strstr(sp_last_error(), "not available") != nullptr) return std::nullopt;
            throw_on_error(rc);
            h = reinterpret_cast<sp_align_result*>(static_cast<char*>(pin_) + 1024);
            hip_check(hipMemcpyAsync(h, res_dev_, sizeof *h, hipMemcpyDeviceToHost, queue_.stream()), "D2H");
            hip_check(hipStreamSynchronize(queue_.stream()), "sync");
        }
        if (h->status != 0) {  // not every workgroup of the launch was resident (another process / a CU mask holds compute units)
            throw_on_error(sp_gicp_source_set_persistent(psrc_, 0));  // this source: per-step launches from now on
            return std::nullopt;
        }
        RegistrationResult result;
        TransformMatrix T;
        for (int i = 0; i < 16; ++i) { T.data()[i] = h->T[i]; lin_T_.data()[i] = h->T_lin[i]; }
        last_lin_fused_ = true;
        result.T.matrix() = T;
        result.iterations = h->iterations;
        result.converged = h->converged != 0;
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) result.H(i, j) = h->H[i * 6 + j];
            result.b(i) = h->b[i];
        }
        result.error = h->error;
        result.inlier = h->inlier;
        result.H_raw = result.H; result.b_raw = result.b; result.error_raw = h->error_raw;
        last_opt_linearizations_ = h->linearizations;
        last_opt_trials_ = h->trials;
        return result;
    }
    RegistrationResult align_on_device(size_t N, const TransformMatrix& initial_guess, float robust_scale) {
        const sp_factor_params fp = factor_params(robust_scale);
        const sp_gn_params gn{params_.gn.lambda, params_.criteria.rotation, params_.criteria.translation};
        float* delta_dev = T_dev_ + 16;
        uint32_t* iters_dev = reinterpret_cast<uint32_t*>(T_dev_ + 24);
        hip_check(hipMemcpyAsync(T_dev_, initial_guess.data(), 16 * sizeof(float), hipMemcpyHostToDevice, queue_.stream()), "H2D");
        if (xchg_ != nullptr)  // source sharded over the ranks: rows stored into the peers' slot buffers
            throw_on_error(sp_gicp_align_direct(ptgt_, psrc_, T_dev_, &fp, &gn, (int)params_.max_iterations, xchg_, nullptr,
                                                nullptr, lin_dev_, delta_dev, iters_dev, ws_, ws_bytes_, queue_.stream()));
        else if (comm_ != nullptr)  // one 128-byte all-reduce per iteration
            throw_on_error(sp_gicp_align_sharded(ptgt_, psrc_, T_dev_, &fp, &gn, (int)params_.max_iterations, comm_, nullptr,
                                                 nullptr, lin_dev_, delta_dev, iters_dev, ws_, ws_bytes_, queue_.stream()));
        else
            // (no neighbour output: nothing reads neighbors_ on this path, and without it the kernels need not look for a
            // neighbour beyond max_correspondence_distance — with partial overlap that search was most of an iteration)
            throw_on_error(sp_gicp_align_fused(ptgt_, psrc_, T_dev_, &fp, &gn, (int)params_.max_iterations, nullptr, nullptr,
                                               lin_dev_, delta_dev, iters_dev, ws_, ws_bytes_, queue_.stream()));
        float* const host = reinterpret_cast<float*>(static_cast<char*>(pin_) + 256);  // pose | delta[8] | iterations | T_lin
        hip_check(hipMemcpyAsync(host, T_dev_, 28 * sizeof(float), hipMemcpyDeviceToHost, queue_.stream()), "D2H");
        // the pose the correspondence cache is frozen at (compute_error_frozen after align(), registration.hpp:350-359)
        throw_on_error(sp_gicp_align_linearization_pose(ws_, (int)params_.max_iterations - 1, host + 28, queue_.stream()));
        const sp_linearized h = read_lin();  // synchronises the stream
        if (xchg_ != nullptr)  // a row that did not arrive: the loop stopped, the pose is not the alignment's
            throw_on_error(sp_gicp_align_status(ws_, (int)params_.max_iterations - 1, queue_.stream()));
        const LinearizedResult lin = to_result(h);
        RegistrationResult result;
        TransformMatrix T;
        for (int i = 0; i < 16; ++i) T.data()[i] = host[i];
        for (int i = 0; i < 16; ++i) lin_T_.data()[i] = host[28 + i];
        last_lin_fused_ = true;
        result.T.matrix() = T;
        uint32_t iters;
        std::memcpy(&iters, &host[24], sizeof iters);
        if (iters == 0xffffffffu) {
            // the device-side tail of sp_gicp_align_fused ran into its bounded wait (not every workgroup was resident: another
            // process or a CU mask holds compute units) and left a NaN pose: once more from the initial guess, a launch per
            // iteration, and this prepared source stays on that form
            throw_on_error(sp_gicp_source_set_persistent(psrc_, 0));
            if (retried_without_persistent_)
                throw std::runtime_error("[Registration::align] the device-resident loop ran into its time limit");
            retried_without_persistent_ = true;
            throw_on_error(sp_gicp_source_prepare(psrc_, ptgt_, last_src_points_, last_src_covs_, N, initial_guess.data(), 0,
                                                  source_order(N), queue_.stream()));
            const RegistrationResult again = align_on_device(N, initial_guess, robust_scale);
            retried_without_persistent_ = false;
            return again;
        }
        result.iterations = iters > 0 ? iters - 1 : 0;  // index of the last iteration (registration.hpp:822)
        result.converged = host[16 + 6] > 0.5f;
        result.H = lin.H; result.b = lin.b; result.error = lin.error; result.inlier = lin.inlier;
        result.H_raw = lin.H; result.b_raw = lin.b; result.error_raw = lin.error;
        return result;
    }
    LinearizedResult linearize_fused(size_t N, const TransformMatrix& T, float robust_scale) {
        const sp_factor_params fp = factor_params(robust_scale);
        TransformMatrix Tc = T;
        lin_T_ = T;  // the pose the correspondences are frozen at (compute_error on the prepared path)
        last_lin_fused_ = true;
        throw_on_error(sp_gicp_iteration_fused(ptgt_, psrc_, Tc.data(), 0, &fp, nullptr, nullptr, nullptr, lin_dev_, nullptr, ws_,
                                               ws_bytes_, queue_.stream()));
        return to_result(read_lin());
    }
    std::tuple<float, uint32_t> compute_error(const PointCloudShared& source, const PointCloudShared& target,
                                              const TransformMatrix& T, float robust_scale) const {
        const sp_factor_params fp = factor_params(robust_scale);
        if (fused_loop_active_) {  // K12 over the correspondence cache: no gathers, no eigen-decompositions
            throw_on_error(sp_gicp_error_prepared(ptgt_, psrc_, lin_T_.data(), T.data(), 0, &fp, lin_dev_, ws_, ws_bytes_,
                                                  queue_.stream()));
            const sp_linearized h = read_lin();
            return {h.error, h.inlier};
        }
        throw_on_error(sp_gicp_error(source.points_device(), source.covs_device(), source.size(), target.points_device(),
                                     target.covs_device(), target.normals_device(), neighbors_.indices->device_data(),
                                     neighbors_.distances->device_data(), T.data(), 0, &fp, lin_dev_, ws_, ws_bytes_,
                                     queue_.stream()));
        const sp_linearized h = read_lin();
        return {h.error, h.inlier};
    }
    /// solve_linear_system + is_converged + pose update (registration.hpp:791-801, 407-410, 814): the host twin of the
    /// device solver; returns {delta, converged}.
    std::pair<Eigen::Matrix<float, 6, 1>, bool> gn_step(const LinearizedResult& lin, float lambda, TransformMatrix& T) const {
        sp_linearized h{};
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) h.H[i * 6 + j] = lin.H(i, j);
            h.b[i] = lin.b(i);
        }
        float d8[8];
        sp_gn_update_host(&h, T.data(), lambda, params_.criteria.rotation, params_.criteria.translation, d8);
        Eigen::Matrix<float, 6, 1> delta;
        for (int i = 0; i < 6; ++i) delta(i) = d8[i];
        return {delta, d8[6] > 0.5f};
    }
    void optimize_gauss_newton(RegistrationResult& result, const LinearizedResult& lin, size_t iter) const {  // :803-828
        TransformMatrix T = result.T.matrix();
        const auto [delta, conv] = gn_step(lin, params_.gn.lambda, T);
        result.converged = conv;
        result.T.matrix() = T;
        result.iterations = iter;
        result.H = lin.H; result.b = lin.b; result.error = lin.error; result.inlier = lin.inlier;
        if (params_.verbose)
            std::cout << "iter [" << iter << "] error: " << result.error << ", inlier: " << result.inlier << std::endl;
        (void)delta;
    }
    bool optimize_levenberg_marquardt(const PointCloudShared& source, const PointCloudShared& target,
                                      RegistrationResult& result, const LinearizedResult& lin, float& lambda, size_t iter,
                                      float robust_scale) const {  // registration.hpp:830-895
        const float current_error = lin.error;
        bool updated = false;
        float last_error = std::numeric_limits<float>::max();
        for (size_t i = 0; i < params_.lm.max_inner_iterations; ++i) {
            TransformMatrix new_T = result.T.matrix();
            const auto [delta, conv] = gn_step(lin, lambda, new_T);
            (void)delta;
            result.converged = conv;
            const auto [new_icp_error, inlier] = compute_error(source, target, new_T, robust_scale);
            const float new_error = new_icp_error + prior_error(new_T);  // registration.hpp:854
            if (new_error <= current_error) {
                result.T.matrix() = new_T; result.error = new_error; result.inlier = inlier; updated = true;
                lambda = std::clamp(lambda / params_.lm.lambda_factor, params_.lm.min_lambda, params_.lm.max_lambda);
                break;
            } else if (std::fabs(new_error - last_error) <= 1e-6f) {
                result.T.matrix() = new_T; result.error = new_error; result.inlier = inlier;
                break;
            } else {
                lambda = std::clamp(lambda * params_.lm.lambda_factor, params_.lm.min_lambda, params_.lm.max_lambda);
            }
            last_error = new_error;
        }
        result.iterations = iter;
        result.H = lin.H; result.b = lin.b;
        return updated;
    }

    bool optimize_powell_dogleg(const PointCloudShared& source, const PointCloudShared& target, RegistrationResult& result,
                                const LinearizedResult& lin, float& trust_region_radius, size_t iter,
                                float robust_scale) const {  // registration.hpp:897-965
        result.H = lin.H; result.b = lin.b; result.error = lin.error; result.inlier = lin.inlier; result.iterations = iter;
        const auto& dl = params_.dogleg;
        const auto clamp_radius = [&](float r) { return std::clamp(r, dl.min_trust_region_radius, dl.max_trust_region_radius); };
        trust_region_radius = clamp_radius(trust_region_radius);
        float H36[36], g6[6], p6[6], step_norm = 0.0f, predicted = 0.0f;
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) H36[i * 6 + j] = lin.H(i, j);
            g6[i] = lin.b(i);
        }
        sp_dogleg_step_host(H36, g6, trust_region_radius, p6, &step_norm, &predicted);  // dogleg_step.hpp:35-101
        if (predicted <= 0.0f) {
            trust_region_radius = clamp_radius(trust_region_radius * dl.gamma_decrease);
            return false;
        }
        TransformMatrix E, new_T;
        sp_se3_exp_host(p6, E.data());
        const TransformMatrix cur = result.T.matrix();
        sp_rigid_mul_host(cur.data(), E.data(), new_T.data());
        const auto [new_icp_error, inlier] = compute_error(source, target, new_T, robust_scale);
        const float new_error = new_icp_error + prior_error(new_T);  // registration.hpp:933
        const float rho = (lin.error - new_error) / predicted;
        if (params_.verbose)
            std::cout << "iter [" << iter << "] radius: " << trust_region_radius << ", rho: " << rho << ", error: " << new_error
                      << ", inlier: " << inlier << std::endl;
        if (rho < dl.eta1) {
            trust_region_radius = clamp_radius(trust_region_radius * dl.gamma_decrease);
            return false;
        }
        const float nr = std::sqrt(p6[0] * p6[0] + p6[1] * p6[1] + p6[2] * p6[2]);
        const float nt = std::sqrt(p6[3] * p6[3] + p6[4] * p6[4] + p6[5] * p6[5]);
        result.converged = nr < params_.criteria.rotation && nt < params_.criteria.translation;  // is_converged (:407-410)
        result.T.matrix() = new_T;
        result.error = new_error;
        result.inlier = inlier;
        if (rho > dl.eta2 && step_norm >= trust_region_radius * 0.99f)
            trust_region_radius = clamp_radius(trust_region_radius * dl.gamma_increase);
        return true;
    }

    RegistrationParams params_;
    sycl_utils::DeviceQueue queue_;
    mutable knn::KNNResult neighbors_;
    sp_linearized* lin_dev_ = nullptr;
    void* ws_ = nullptr;
    size_t ws_bytes_ = 0;
    float* T_dev_ = nullptr;
    void* pin_ = nullptr;  // 4 KB of pinned host memory: [0, 256) linear system, [256, 432) pose | delta | iterations | T_lin,
                           // [1024, 1024 + sizeof(sp_align_result)) the result block of the device-resident optimiser
    sp_align_result* res_dev_ = nullptr;
    static constexpr size_t kMappedBytes = 256 + ((sizeof(sp_align_result) + 255) / 256) * 256;  // pose | result block
    char* map_host_ = nullptr;  // host-mapped: [0, 64) the optimiser's pose, [256, ...) its result block
    char* map_dev_ = nullptr;   // ... as the device sees it
    float genz_alpha_ = 1.0f;
    mutable float rotation_robust_scale_ = 10.0f;  // resolved per call from ExecutionOptions (registration.hpp:219-221)
    sp_map_prior_state map_prior_{};               // MapPrior state (map_prior.hpp:203-210)
    sp_gicp_source* psrc_ = nullptr;
    size_t psrc_cap_ = 0;
    sp_gicp_target* ptgt_ = nullptr;
    uint64_t ptgt_grid_id_ = 0;  // GridKNN::id() the prepared target was built on
    const void* ptgt_covs_ = nullptr;  // the covariance container the prepared rows were computed from, ...
    uint64_t ptgt_covs_gen_ = 0;       // ... its generation then, ...
    int ptgt_reg_ = -1;                // ... and the factor they are the rows of
    unsigned ptgt_uses_ = 0;           // alignments against this prepared target so far (second use: certificates, prepare_fused)
    bool source_presorted_ = false;
    sp_comm* comm_ = nullptr;  // borrowed (set_communicator)
    sp_xchg* xchg_ = nullptr;  // borrowed (set_exchange)
    bool accelerate_kdtree_ = true;
    mutable bool fused_loop_active_ = false;  // align() is running its optimiser loop on the prepared path
    mutable bool last_lin_fused_ = false;     // the last linearisation left its correspondences in the prepared source's cache
    TransformMatrix lin_T_ = TransformMatrix::Identity();  // pose of the last fused linearisation
    const float* last_src_points_ = nullptr;  // device pointers of the source of the latest prepare_fused (valid during align())
    const float* last_src_covs_ = nullptr;
    bool retried_without_persistent_ = false;
    uint32_t last_opt_linearizations_ = 0, last_opt_trials_ = 0;  // steps of the latest device-resident optimiser run
    knn::GridKNN::Ptr kd_grid_;        // GridKNN standing in for the caller's KDTree (grid_for)
    uint64_t kd_grid_tree_id_ = 0;
};

// ------------------------------------------------------------------------------------------------ pipeline wrappers
namespace pipeline {
/// pipeline/aligner.hpp:13-15
using RegistrationAligner = std::function<RegistrationResult(const PointCloudShared&, const PointCloudShared&,
                                                             const knn::KNNBase&, const TransformMatrix&,
                                                             const Registration::ExecutionOptions&)>;
inline RegistrationAligner make_registration_aligner(const Registration::Ptr& registration) {
    return [registration](const PointCloudShared& s, const PointCloudShared& t, const knn::KNNBase& k,
                          const TransformMatrix& T, const Registration::ExecutionOptions& o) {
        return registration->align(s, t, k, T, o);
    };
}
}  // namespace pipeline

/// registration_pipeline_params.hpp:11-43
struct RegistrationRandomSamplingParams { bool enable = true; size_t num = 1000; };
struct RegistrationRobustScheduleParams {
    bool auto_scale = false;
    float init_scale = 10.0f, min_scale = 0.5f, rotation_init_scale = 10.0f, rotation_min_scale = 0.5f;
    size_t auto_scaling_iter = 4;
};
struct RegistrationPipelineParams {
    using RandomSampling = RegistrationRandomSamplingParams;
    using Robust = RegistrationRobustScheduleParams;
    RegistrationParams registration;
    RandomSampling random_sampling;
    Robust robust;
};

namespace pipeline {
/// pipeline/robust.hpp:17-128 — geometric annealing of the robust scale around the wrapped aligner.
class RobustAligner {
public:
    using Ptr = std::shared_ptr<RobustAligner>;
    RobustAligner(RegistrationAligner aligner, const RegistrationPipelineParams& p)
        : aligner_(std::move(aligner)), params_(p.registration), sched_(p.robust) {}
    /// pipeline/robust.hpp:32-33. With the backend known, the levels go to Registration::align_levels: one launch, one
    /// read-back for the whole schedule where the device-resident optimiser applies, the same loop as below otherwise.
    RobustAligner(const Registration::Ptr& registration, const RegistrationPipelineParams& p)
        : RobustAligner(make_registration_aligner(registration), p) {
        registration_ = registration;
    }
    RegistrationResult align(const PointCloudShared& source, const PointCloudShared& target, const knn::KNNBase& knn,
                             const TransformMatrix& initial_guess = TransformMatrix::Identity(),
                             const Registration::ExecutionOptions& options = Registration::ExecutionOptions()) const {
        RegistrationResult result;
        result.T.matrix() = initial_guess;
        if (source.size() == 0) return result;
        const bool fixed = options.robust_scale > 0.0f || options.rotation_robust_scale > 0.0f;
        bool autos = !fixed && params_.robust.type != robust::RobustLossType::NONE && sched_.auto_scale;
        if (autos && (sched_.min_scale <= 0.0f || sched_.min_scale >= sched_.init_scale)) autos = false;
        if (autos && (sched_.rotation_min_scale <= 0.0f || sched_.rotation_min_scale >= sched_.rotation_init_scale)) autos = false;
        if (autos && sched_.auto_scaling_iter == 0) autos = false;
        const size_t levels = autos ? std::max<size_t>(1, sched_.auto_scaling_iter) : 1;
        // pipeline/robust.hpp:83-98: both scales shrink geometrically from init to min over the levels
        float scale = options.robust_scale > 0.0f ? options.robust_scale
                                                  : (autos ? sched_.init_scale : params_.robust.default_scale);
        const float factor = levels > 1 ? std::pow(sched_.min_scale / sched_.init_scale, 1.0f / static_cast<float>(levels - 1)) : 1.0f;
        float rot_scale = options.rotation_robust_scale > 0.0f
                              ? options.rotation_robust_scale
                              : (autos ? sched_.rotation_init_scale : params_.rotation_constraint.robust.default_scale);
        const float rot_factor =
            levels > 1 ? std::pow(sched_.rotation_min_scale / sched_.rotation_init_scale, 1.0f / static_cast<float>(levels - 1)) : 1.0f;
        if (registration_ != nullptr) {
            std::vector<float> scales, rot_scales;
            for (size_t level = 0; level < levels; ++level) {
                scales.push_back(scale);
                rot_scales.push_back(rot_scale);
                scale *= factor;
                rot_scale *= rot_factor;
            }
            return registration_->align_levels(source, target, knn, initial_guess, options, scales, rot_scales);
        }
        for (size_t level = 0; level < levels; ++level) {
            auto o = options;
            o.robust_scale = scale;
            o.rotation_robust_scale = rot_scale;
            result = aligner_(source, target, knn, result.T.matrix(), o);
            scale *= factor;
            rot_scale *= rot_factor;
        }
        return result;
    }
    RegistrationAligner make_aligner() const {
        return [this](const PointCloudShared& s, const PointCloudShared& t, const knn::KNNBase& k, const TransformMatrix& T,
                      const Registration::ExecutionOptions& o) { return this->align(s, t, k, T, o); };
    }

private:
    RegistrationAligner aligner_;
    RegistrationParams params_;
    RegistrationRobustScheduleParams sched_;
    Registration::Ptr registration_;  // set when the wrapped aligner IS a Registration backend
};
}  // namespace pipeline

/// registration_pipeline.hpp:16-149 — optional random sampling of the source, then (annealed) alignment.
class RegistrationPipeline {
public:
    using Ptr = std::shared_ptr<RegistrationPipeline>;
    /// registration_pipeline.hpp:24-43: from an aligner callable, from a Registration backend, or from a queue
    RegistrationPipeline(pipeline::RegistrationAligner aligner, const RegistrationPipelineParams& p = RegistrationPipelineParams())
        : params_(p), aligner_(std::move(aligner)) {
        wrap_aligner();
    }
    RegistrationPipeline(const Registration::Ptr& registration, const RegistrationPipelineParams& p = RegistrationPipelineParams())
        : params_(p), registration_(registration), aligner_(pipeline::make_registration_aligner(registration)) {
        wrap_aligner();
    }
    RegistrationPipeline(const sycl_utils::DeviceQueue& queue, const RegistrationPipelineParams& p = RegistrationPipelineParams())
        : RegistrationPipeline(std::make_shared<Registration>(queue, p.registration), p) {}

    RegistrationResult align(const PointCloudShared& source, const PointCloudShared& target, const knn::KNNBase& target_knn,
                             const TransformMatrix& initial_guess = TransformMatrix::Identity(),
                             const Registration::ExecutionOptions& options = Registration::ExecutionOptions()) const {
        update_registration_input(source);
        return aligner_(*input_, target, target_knn, initial_guess, options);
    }
    const Registration::Ptr& registration() const { return registration_; }
    /// registration_pipeline.hpp:64-77: geometry ICP robust weights of the latest registration input
    void compute_icp_robust_weights(const PointCloudShared& target, const knn::KNNBase& target_knn, const TransformMatrix& pose,
                                    float robust_scale, shared_vector<float>& out) const {
        if (registration_ == nullptr)
            throw std::runtime_error("[RegistrationPipeline::compute_icp_robust_weights] Registration backend is not available.");
        const auto source = get_deskewed_point_cloud();
        if (source == nullptr)
            throw std::runtime_error("[RegistrationPipeline::compute_icp_robust_weights] Registration input point cloud is not available.");
        registration_->compute_icp_robust_weights(*source, target, target_knn, pose, robust_scale, out);
    }
    const PointCloudShared* get_registration_input_point_cloud() const { return input_.get(); }
    /// registration_pipeline.hpp:82-89. The velocity-update (deskew) stage is outside this library's scope,
    /// so this is always the registration input — what the reference returns with velocity_update disabled.
    const PointCloudShared::Ptr get_deskewed_point_cloud() const { return input_; }
    /// registration_pipeline.hpp:91-97
    float get_inlier_ratio(const RegistrationResult& result) const {
        const auto* in = get_registration_input_point_cloud();
        if (in && in->size() > 0) return static_cast<float>(result.inlier) / static_cast<float>(in->size());
        return 0.0f;
    }

private:
    void wrap_aligner() {  // registration_pipeline.hpp:100-119 (robust wrapper outermost)
        if (params_.robust.auto_scale) {
            // (no velocity-update wrapper sits between the two in this library, so the robust wrapper may talk to the backend)
            robust_ = registration_ != nullptr ? std::make_shared<pipeline::RobustAligner>(registration_, params_)
                                               : std::make_shared<pipeline::RobustAligner>(aligner_, params_);
            aligner_ = robust_->make_aligner();
        }
    }
    void update_registration_input(const PointCloudShared& source) const {  // registration_pipeline.hpp:121-140
        if (filter_ == nullptr || input_ == nullptr) {
            filter_ = std::make_shared<filter::PreprocessFilter>(source.queue);
            input_ = std::make_shared<PointCloudShared>(source.queue);
        }
        const auto& rs = params_.random_sampling;
        if (rs.enable && source.size() > rs.num) filter_->random_sampling(source, *input_, rs.num);
        else *input_ = source;  // shallow (registration_pipeline.hpp:138)
    }
    RegistrationPipelineParams params_;
    Registration::Ptr registration_;
    pipeline::RobustAligner::Ptr robust_;
    pipeline::RegistrationAligner aligner_;
    mutable filter::PreprocessFilter::Ptr filter_;
    mutable PointCloudShared::Ptr input_;
};

}  // namespace registration
}  // namespace algorithms
}  // namespace sycl_points
